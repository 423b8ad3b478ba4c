/* vslam_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain, single-threaded C++ restatement of the reference's per-frame hot path
 * (Ssellu/vslam-pose-estimation-framework, a ProSLAM fork).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (libvslam_hip.so) never does.
 *
 * PARITY STATUS: "parity unpinned" against the reference binary.  The reference cannot be built
 * here (needs OpenCV 3 + xfeatures2d, Eigen, srrg_core, srrg_hbst, yaml-cpp, easy_profiler, g2o,
 * Qt: none on this machine) and ships no tests, fixtures or golden vectors.
 * The third-party arithmetic on the path is restated from the published algorithms:
 *   - cv::FastFeatureDetector (FAST-9/16 + cornerScore + 3x3 strict NMS), OpenCV 3.x features2d
 *   - cv::xfeatures2d::BriefDescriptorExtractor(32): integral image + 9x9 box tests, 28 px border;
 *     the 256 test pairs are REPO-DEFINED (include/vslam_brief_pattern.h), OpenCV's table is absent
 *   - cv::norm(NORM_HAMMING), BFMatcher::knnMatch(k=2)
 *   - srrg_core::skew / v2t, Eigen fullPivLu / Isometry (branch "marchless", unpinned)
 * It is pinned instead by an independent numpy restatement (tests/golden/make_golden.py) whose
 * outputs are committed under tests/golden/.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../include/vslam_hip.h"
#include "../include/vslam_brief_pattern.h"
#include "../include/vslam_orb_pattern.h"
#include "../tools/synth/synth_scene.h"

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace {

typedef double real; /* src/types/definitions.h:52 */

/* ------------------------------------------------------------------------------------------
 * small fixed-size math (Eigen / srrg_core restated)
 * ---------------------------------------------------------------------------------------- */
struct Tf { /* row-major 3x4 [R|t], TransformMatrix3D (definitions.h:62) */
  real m[12];
};
inline Tf tf_identity() { Tf t; std::memset(t.m, 0, sizeof t.m); t.m[0] = t.m[5] = t.m[10] = 1; return t; }
inline void tf_apply(const Tf& T, const real p[3], real out[3]) {
  for (int i = 0; i < 3; ++i)
    out[i] = ((T.m[4 * i + 0] * p[0] + T.m[4 * i + 1] * p[1]) + T.m[4 * i + 2] * p[2]) + T.m[4 * i + 3];
}
inline Tf tf_mul(const Tf& A, const Tf& B) { /* A*B */
  Tf C;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      C.m[4 * i + j] = (A.m[4 * i + 0] * B.m[0 + j] + A.m[4 * i + 1] * B.m[4 + j]) + A.m[4 * i + 2] * B.m[8 + j];
    C.m[4 * i + 3] = ((A.m[4 * i + 0] * B.m[3] + A.m[4 * i + 1] * B.m[7]) + A.m[4 * i + 2] * B.m[11]) + A.m[4 * i + 3];
  }
  return C;
}
inline Tf tf_inverse(const Tf& A) { /* Isometry inverse: [R^T | -R^T t] */
  Tf C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[4 * i + j] = A.m[4 * j + i];
  for (int i = 0; i < 3; ++i)
    C.m[4 * i + 3] = -((C.m[4 * i + 0] * A.m[3] + C.m[4 * i + 1] * A.m[7]) + C.m[4 * i + 2] * A.m[11]);
  return C;
}
inline void mat3_mul_vec(const real K[9], const real p[3], real out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = (K[3 * i + 0] * p[0] + K[3 * i + 1] * p[1]) + K[3 * i + 2] * p[2];
}

/* srrg_core::v2t [recalled, SURVEY.md §8c]: translation = v[0:3]; rotation from the vector part of
 * a unit quaternion (w = sqrt(1-|q|^2)); Eigen Quaternion::toRotationMatrix formula. */
inline Tf v2t(const real v[6]) {
  Tf T;
  real qx = v[3], qy = v[4], qz = v[5], qw;
  const real n2 = (qx * qx + qy * qy) + qz * qz;
  if (n2 < 1) {
    qw = std::sqrt(1 - n2);
  } else {
    const real n = std::sqrt(n2);
    qx /= n; qy /= n; qz /= n; qw = 0;
  }
  const real tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const real twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const real txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const real tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  T.m[0] = 1 - (tyy + tzz); T.m[1] = txy - twz;       T.m[2] = txz + twy;
  T.m[4] = txy + twz;       T.m[5] = 1 - (txx + tzz); T.m[6] = tyz - twx;
  T.m[8] = txz - twy;       T.m[9] = tyz + twx;       T.m[10] = 1 - (txx + tyy);
  T.m[3] = v[0]; T.m[7] = v[1]; T.m[11] = v[2];
  return T;
}

/* Eigen::FullPivLU<Matrix6>::solve restated: Gaussian elimination with full pivoting, pivot =
 * first strict maximum of |a_ij| in column-major scan of the remaining corner. n <= 6. */
template <int N>
inline void full_piv_lu_solve(const real A_in[N * N], const real b_in[N], real x[N]) {
  real A[N * N], b[N];
  int colperm[N];
  std::memcpy(A, A_in, sizeof A);
  std::memcpy(b, b_in, sizeof b);
  for (int i = 0; i < N; ++i) colperm[i] = i;
  int rank = N;
  for (int k = 0; k < N; ++k) {
    int pr = k, pc = k;
    real best = 0;
    for (int j = k; j < N; ++j)
      for (int i = k; i < N; ++i) {
        const real a = std::fabs(A[i * N + j]);
        if (a > best) { best = a; pr = i; pc = j; }
      }
    if (best == 0) { rank = k; break; }
    if (pr != k) { for (int j = 0; j < N; ++j) std::swap(A[k * N + j], A[pr * N + j]); std::swap(b[k], b[pr]); }
    if (pc != k) { for (int i = 0; i < N; ++i) std::swap(A[i * N + k], A[i * N + pc]); std::swap(colperm[k], colperm[pc]); }
    for (int i = k + 1; i < N; ++i) {
      const real f = A[i * N + k] / A[k * N + k];
      A[i * N + k] = 0;
      for (int j = k + 1; j < N; ++j) A[i * N + j] -= f * A[k * N + j];
      b[i] -= f * b[k];
    }
  }
  real y[N];
  for (int i = 0; i < N; ++i) y[i] = 0;
  for (int i = rank - 1; i >= 0; --i) {
    real s = b[i];
    for (int j = i + 1; j < rank; ++j) s -= A[i * N + j] * y[j];
    y[i] = s / A[i * N + i];
  }
  for (int i = 0; i < N; ++i) x[colperm[i]] = y[i];
}

/* WorldMap::toOrientationRodrigues(R).norm() (src/types/world_map.h:143-147): the rotation angle
 * as cv::Rodrigues computes it (matrix -> vector branch) [recalled]. */
inline real rotation_angle(const Tf& T) {
  const real rx = T.m[9] - T.m[6], ry = T.m[2] - T.m[8], rz = T.m[4] - T.m[1];
  const real s = std::sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25);
  real c = ((T.m[0] + T.m[5]) + T.m[10] - 1) * 0.5;
  c = c > 1 ? 1 : (c < -1 ? -1 : c);
  if (s < 1e-5) return c > 0 ? 0.0 : M_PI;
  return std::acos(c);
}

/* ------------------------------------------------------------------------------------------
 * FAST-9/16 with non-max suppression: cv::FastFeatureDetector::detect on a ROI view
 * (reference call sites base_framepoint_generator.cpp:12-25 and :367; algorithm: OpenCV 3.x
 * features2d/src/fast.cpp FAST_t<16> and fast_score.cpp cornerScore<16>) [recalled].
 * ---------------------------------------------------------------------------------------- */
const int kFastDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kFastDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* returns the corner score (>= threshold) or 0 when (x,y) is not a FAST-9 corner */
inline int fast_corner_score(const uint8_t* img, int stride, int x, int y, int threshold) {
  const int v = img[y * stride + x];
  {
    /* high-speed rejection of fast.cpp: a 9-arc contains one pixel of every opposite pair */
    const uint8_t* c = img + y * stride + x;
    const int lo = v - threshold, hi = v + threshold;
    const int p0 = c[3 * stride], p8 = c[-3 * stride], p4 = c[3], p12 = c[-3];
    const bool dark = (p0 < lo || p8 < lo) && (p4 < lo || p12 < lo);
    const bool bright = (p0 > hi || p8 > hi) && (p4 > hi || p12 > hi);
    if (!dark && !bright) return 0;
  }
  int d[25];
  for (int k = 0; k < 16; ++k) d[k] = v - (int)img[(y + kFastDy[k]) * stride + (x + kFastDx[k])];
  for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
  /* corner iff >= 9 contiguous circle pixels are all darker (d > t) or all brighter (d < -t) */
  bool corner = false;
  int run_dark = 0, run_bright = 0;
  for (int k = 0; k < 25 && !corner; ++k) {
    run_dark = (d[k] > threshold) ? run_dark + 1 : 0;
    run_bright = (d[k] < -threshold) ? run_bright + 1 : 0;
    if (run_dark > 8 || run_bright > 8) corner = true;
  }
  if (!corner) return 0;
  /* cornerScore<16>: largest threshold for which the pixel stays a corner */
  int A = -1000, Bm = 1000;
  for (int s = 0; s < 16; ++s) {
    int mn = d[s], mx = d[s];
    for (int j = 1; j < 9; ++j) { mn = std::min(mn, d[s + j]); mx = std::max(mx, d[s + j]); }
    A = std::max(A, mn);
    Bm = std::min(Bm, mx);
  }
  const int a0 = std::max(threshold, A);
  const int b0 = std::min(-a0, Bm);
  return -b0 - 1;
}

struct Keypoint {
  int16_t x, y;
  int32_t score;
};

/* detect on ROI (rx,ry,rw,rh) of an image; coordinates returned relative to the ROI */
void fast_detect_roi(const uint8_t* img, int stride, int rx, int ry, int rw, int rh, int threshold,
                     std::vector<Keypoint>& out) {
  out.clear();
  threshold = std::min(std::max(threshold, 0), 255);
  if (rw < 7 || rh < 7) return;
  const uint8_t* roi = img + (size_t)ry * stride + rx;
  std::vector<uint8_t> score((size_t)rw * rh, 0);
  for (int y = 3; y < rh - 3; ++y)
    for (int x = 3; x < rw - 3; ++x) score[(size_t)y * rw + x] = (uint8_t)fast_corner_score(roi, stride, x, y, threshold);
  for (int y = 3; y < rh - 3; ++y)
    for (int x = 3; x < rw - 3; ++x) {
      const int s = score[(size_t)y * rw + x];
      if (!s) continue;
      bool keep = true;
      for (int dy = -1; dy <= 1 && keep; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          if (!dx && !dy) continue;
          if (s <= score[(size_t)(y + dy) * rw + (x + dx)]) { keep = false; break; }
        }
      if (keep) out.push_back(Keypoint{(int16_t)x, (int16_t)y, s});
    }
}

/* ------------------------------------------------------------------------------------------
 * BRIEF-32: cv::xfeatures2d::BriefDescriptorExtractorImpl::compute restated [recalled]:
 * integral image (CV_32S), KeyPointsFilter::runByImageBorder(28), pixelTests32 on 9x9 box sums.
 * Reference call site: base_framepoint_generator.cpp:431-438.
 * ---------------------------------------------------------------------------------------- */
int8_t kBriefPattern[256][4] = VSLAM_BRIEF_PATTERN_INIT;   /* run-time data: orc_set_brief_pattern (the C ABI's vslam_set_brief_pattern) */

void integral_image(const uint8_t* img, int rows, int cols, int stride, std::vector<int32_t>& sum) {
  sum.assign((size_t)(rows + 1) * (cols + 1), 0);
  for (int y = 0; y < rows; ++y) {
    int32_t acc = 0;
    for (int x = 0; x < cols; ++x) {
      acc += img[(size_t)y * stride + x];
      sum[(size_t)(y + 1) * (cols + 1) + (x + 1)] = sum[(size_t)y * (cols + 1) + (x + 1)] + acc;
    }
  }
}
inline int32_t smoothed_sum(const std::vector<int32_t>& sum, int cols, int y, int x) {
  const int W = cols + 1, h = VSLAM_BRIEF_KERNEL_HALF;
  return sum[(size_t)(y + h + 1) * W + (x + h + 1)] - sum[(size_t)(y + h + 1) * W + (x - h)] -
         sum[(size_t)(y - h) * W + (x + h + 1)] + sum[(size_t)(y - h) * W + (x - h)];
}
inline bool brief_inside(int rows, int cols, int x, int y) {
  const int b = VSLAM_BRIEF_BORDER;
  return x >= b && x < cols - b && y >= b && y < rows - b;
}
void brief_at(const std::vector<int32_t>& sum, int cols, int x, int y, uint8_t desc[32]) {
  std::memset(desc, 0, 32);
  for (int i = 0; i < 256; ++i) {
    const int32_t a = smoothed_sum(sum, cols, y + kBriefPattern[i][0], x + kBriefPattern[i][1]);
    const int32_t b = smoothed_sum(sum, cols, y + kBriefPattern[i][2], x + kBriefPattern[i][3]);
    if (a < b) desc[i >> 3] |= (uint8_t)(0x80u >> (i & 7));
  }
}

/* ------------------------------------------------------------------------------------------
 * ORB as descriptor extractor: cv::ORB::create()->compute() on provided keypoints [recalled, OpenCV 3.x
 * features2d/src/orb.cpp detectAndCompute(useProvidedKeypoints) + computeOrbDescriptors, imgproc smooth.cpp].
 * Reference call sites: base_framepoint_generator.cpp:190-196,219-224 (extractor), :431-438 (compute).
 * ---------------------------------------------------------------------------------------- */
int8_t kOrbPattern[256][4] = VSLAM_ORB_PATTERN_INIT;       /* run-time data: orc_set_orb_pattern */

/* getGaussianKernel(7, 2, CV_32F) -> fixed point for 8-bit images: cvRound(k * 256) (createSeparableLinearFilter, bits = 8) */
inline void gauss7_kernel_fixed(int32_t k[7]) {
  float cf[7];
  double sum = 0;
  const double scale2x = -0.5 / (2.0 * 2.0);
  for (int i = 0; i < 7; ++i) { const double x = i - 3.0; cf[i] = (float)std::exp(scale2x * x * x); sum += cf[i]; }
  sum = 1. / sum;
  for (int i = 0; i < 7; ++i) { cf[i] = (float)(cf[i] * sum); k[i] = (int32_t)std::lrint((double)cf[i] * 256.0); }
}
inline int reflect101(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }
/* GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1: integer row pass, integer column pass,
 * FixedPtCastEx<int, uchar>(16): (v + 2^15) >> 16, saturated */
void gaussian_blur7_u8(const uint8_t* img, int rows, int cols, int stride, std::vector<uint8_t>& out) {
  int32_t k[7];
  gauss7_kernel_fixed(k);
  std::vector<int32_t> tmp((size_t)rows * cols);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int32_t acc = 0;
      for (int i = 0; i < 7; ++i) acc += k[i] * (int32_t)img[(size_t)y * stride + reflect101(x + i - 3, cols)];
      tmp[(size_t)y * cols + x] = acc;
    }
  out.assign((size_t)rows * cols, 0);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int32_t acc = 0;
      for (int i = 0; i < 7; ++i) acc += k[i] * tmp[(size_t)reflect101(y + i - 3, rows) * cols + x];
      const int32_t v = (acc + (1 << 15)) >> 16;
      out[(size_t)y * cols + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}
inline bool orb_inside(int rows, int cols, int x, int y) { /* runByImageBorder(edgeThreshold = 31) */
  const int b = VSLAM_ORB_BORDER;
  return x >= b && x < cols - b && y >= b && y < rows - b;
}
/* rotation of the pattern by KeyPoint::angle: float angle *= (float)(CV_PI/180.f); a = (float)cos(angle), b = (float)sin(angle) */
inline void orb_rotation(float angle_degrees, float* a, float* b) {
  float angle = angle_degrees;
  angle *= (float)(3.1415926535897932384626433832795 / 180.f);
  *a = (float)std::cos(angle); *b = (float)std::sin(angle);
}
/* computeOrbDescriptors, WTA_K = 2: GET_VALUE(p) = blurred[cy + cvRound(px*b + py*a)][cx + cvRound(px*a - py*b)] */
void orb_at(const uint8_t* blur, int cols, int cx, int cy, float a, float b, uint8_t desc[32]) {
  std::memset(desc, 0, 32);
  for (int i = 0; i < 256; ++i) {
    int v[2];
    for (int h = 0; h < 2; ++h) {
      const float px = (float)kOrbPattern[i][2 * h], py = (float)kOrbPattern[i][2 * h + 1];
      const float xf = px * a - py * b, yf = px * b + py * a;
      const int ix = (int)std::lrint((double)xf), iy = (int)std::lrint((double)yf);
      v[h] = blur[(size_t)(cy + iy) * cols + (cx + ix)];
    }
    if (v[0] < v[1]) desc[i >> 3] |= (uint8_t)(1u << (i & 7));
  }
}

/* cv::norm(a, b, NORM_HAMMING) on 32 bytes (definitions.h:49) */
inline int hamming32(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

/* ------------------------------------------------------------------------------------------
 * data model (src/types/frame_point.h, frame.h restated as PODs)
 * ---------------------------------------------------------------------------------------- */
struct Feature { /* IntensityFeature (frame_point.h:18-35): row=(int)pt.y, col=(int)pt.x */
  int row, col;
  int score;
  uint8_t desc[32];
};

struct FeatureStore { /* IntensityFeatureMatcher (intensity_feature_matcher.{h,cpp}) */
  int rows = 0, cols = 0;
  std::vector<Feature> feats;  /* features of the image, id = position at setFeatures time  */
  std::vector<int> vec;        /* feature_vector: ids, pruned / sorted in place             */
  std::vector<int> lattice;    /* feature_lattice[row][col]: id or -1                       */
  void configure(int r, int c) { rows = r; cols = c; lattice.assign((size_t)r * c, -1); }
  /* setFeatures (intensity_feature_matcher.cpp:48-70): later duplicate overwrites the cell */
  void set_features(const std::vector<Feature>& f) {
    std::fill(lattice.begin(), lattice.end(), -1);
    feats = f;
    vec.resize(f.size());
    for (size_t i = 0; i < f.size(); ++i) { vec[i] = (int)i; lattice[(size_t)f[i].row * cols + f[i].col] = (int)i; }
  }
  /* sortFeatureVector (:72-79); ties (same pixel) broken by id to stay deterministic */
  void sort_vec() {
    std::sort(vec.begin(), vec.end(), [this](int a, int b) {
      const Feature &fa = feats[a], &fb = feats[b];
      if (fa.row != fb.row) return fa.row < fb.row;
      if (fa.col != fb.col) return fa.col < fb.col;
      return a < b;
    });
  }
  /* prune (:150-172): remove the given POSITIONS of vec, keep order */
  void prune_positions(const std::set<uint32_t>& pos) {
    size_t n = 0;
    for (size_t i = 0; i < vec.size(); ++i)
      if (!pos.count((uint32_t)i)) vec[n++] = vec[i];
    vec.resize(n);
  }
  /* getMatchingFeatureInRectangularRegion (:81-148); returns feature id or -1 */
  int match_in_region(int row_ref, int col_ref, const uint8_t* desc_ref, int r0, int r1, int c0, int c1,
                      real max_dist, bool by_appearance, real& dist_best) const {
    dist_best = max_dist;
    int best = -1;
    if (by_appearance) {
      for (int r = r0; r < r1; ++r)
        for (int c = c0; c < c1; ++c) {
          const int id = lattice[(size_t)r * cols + c];
          if (id < 0) continue;
          const real d = hamming32(desc_ref, feats[id].desc);
          if (d < dist_best) { dist_best = d; best = id; }
        }
    } else {
      uint32_t pix_best = 10000;
      for (int r = r0; r < r1; ++r)
        for (int c = c0; c < c1; ++c) {
          const int id = lattice[(size_t)r * cols + c];
          if (id < 0) continue;
          const real d = hamming32(desc_ref, feats[id].desc);
          if (d < max_dist) {
            const int32_t dr = row_ref - r, dc = col_ref - c;
            const uint32_t pix = (uint32_t)(dr * dr + dc * dc);
            if (pix < pix_best) { pix_best = pix; dist_best = d; best = id; }
          }
        }
    }
    return best;
  }
};

struct Meas { /* Landmark::Measurement (landmark.h:22-36) */
  int frame;
  real cam[3];
  real inv_depth;
};
struct Landmark { /* src/types/landmark.{h,cpp} (position part only) */
  real w[3];
  uint32_t updates;
  std::vector<Meas> meas;
};

struct Point { /* FramePoint (frame_point.h:39-203) */
  int xL, yL, xR, yR;       /* keypointLeft/Right().pt (integer valued)          */
  uint8_t dL[32], dR[32];   /* descriptorLeft/Right                              */
  int dist;                 /* descriptorDistanceTriangulation                   */
  int epi;                  /* epipolarOffset                                    */
  int prev;                 /* index of previous() in the previous frame, or -1  */
  int track_len;            /* trackLength                                       */
  int lm;                   /* landmark id (origin()->landmark()) or -1          */
  bool has_next;            /* next() != nullptr                                 */
  real cam[3];              /* cameraCoordinatesLeft                             */
  real cam_lm[3];           /* cameraCoordinatesLeftLandmark                     */
  real chi;                 /* aligner error of this point (readback only)       */
  uint8_t inlier;
};

struct FrameRec {
  Tf cam_to_world, world_to_cam;
  std::vector<Point> points;
};

/* ------------------------------------------------------------------------------------------
 * StereoUVAligner (src/aligners/stereouv_aligner.cpp:72-264, base_aligner.h:74-106)
 * ---------------------------------------------------------------------------------------- */
struct AlignerIO {
  int n = 0;
  std::vector<real> moving, fixed, omega, weight; /* n*3, n*4, n, n */
  Tf T;
  /* outputs */
  std::vector<real> errors;
  std::vector<uint8_t> inliers;
  int n_inliers = 0, n_outliers = 0, iterations = 0, converged = 0;
  real total_error = 0;
  real H[36];
  bool uvd = false;           /* UVDAligner: fixed = (u, v, depth, depth information), omega = u/v information */
};

struct AlignerParams {
  real K[9], baseline[3];
  int rows, cols;
  real min_depth, kernel, damping, delta;
  int max_it, min_inliers;
};

/* linearize (:72-187) */
void aligner_linearize(const AlignerParams& P, AlignerIO& io, bool ignore_outliers, real H[36], real b[6]) {
  std::memset(H, 0, 36 * sizeof(real));
  std::memset(b, 0, 6 * sizeof(real));
  io.n_inliers = 0;
  io.total_error = 0;
  for (int u = 0; u < io.n; ++u) {
    io.errors[u] = -1;
    io.inliers[u] = 0;
    real omega = io.omega[u];
    real p[3];
    tf_apply(io.T, &io.moving[3 * u], p);
    if (p[2] < P.min_depth) continue;
    real abcL[3], abcR[3];
    mat3_mul_vec(P.K, p, abcL);
    for (int i = 0; i < 3; ++i) abcR[i] = abcL[i] + P.baseline[i];
    const real cL = abcL[2], cR = abcR[2];
    const real uL = abcL[0] / cL, vL = abcL[1] / cL, uR = abcR[0] / cR, vR = abcR[1] / cR;
    if (uL < 0 || uL > P.cols || vL < 0 || vL > P.rows) continue;
    if (uR < 0 || uR > P.cols || vR < 0 || vR > P.rows) continue;
    const real e[4] = {uL - io.fixed[4 * u + 0], vL - io.fixed[4 * u + 1], uR - io.fixed[4 * u + 2],
                       vR - io.fixed[4 * u + 3]};
    const real chi = omega * (((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) + e[3] * e[3]);
    io.errors[u] = chi;
    if (chi > P.kernel) {
      if (ignore_outliers) continue;
      omega *= P.kernel / chi;
    } else {
      io.inliers[u] = 1;
      ++io.n_inliers;
    }
    io.total_error += chi;
    /* jacobian_transform = [w*I3 | -2*skew(p)] (:146-149); skew(p) = [[0,-pz,py],[pz,0,-px],[-py,px,0]] */
    const real w = io.weight[u];
    real Jt[3][6] = {{w, 0, 0, 0, 2 * p[2], -2 * p[1]},
                     {0, w, 0, -2 * p[2], 0, 2 * p[0]},
                     {0, 0, w, 2 * p[1], -2 * p[0], 0}};
    real KJ[3][6];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 6; ++j)
        KJ[i][j] = (P.K[3 * i + 0] * Jt[0][j] + P.K[3 * i + 1] * Jt[1][j]) + P.K[3 * i + 2] * Jt[2][j];
    const real icL = 1 / cL, icR = 1 / cR, icL2 = icL * icL, icR2 = icR * icR;
    const real JL[2][3] = {{icL, 0, -abcL[0] * icL2}, {0, icL, -abcL[1] * icL2}};
    const real JR[2][3] = {{icR, 0, -abcR[0] * icR2}, {0, icR, -abcR[1] * icR2}};
    real J[4][6];
    for (int j = 0; j < 6; ++j) {
      J[0][j] = (JL[0][0] * KJ[0][j] + JL[0][1] * KJ[1][j]) + JL[0][2] * KJ[2][j];
      J[1][j] = (JL[1][0] * KJ[0][j] + JL[1][1] * KJ[1][j]) + JL[1][2] * KJ[2][j];
      J[2][j] = (JR[0][0] * KJ[0][j] + JR[0][1] * KJ[1][j]) + JR[0][2] * KJ[2][j];
      J[3][j] = (JR[1][0] * KJ[0][j] + JR[1][1] * KJ[1][j]) + JR[1][2] * KJ[2][j];
    }
    for (int r = 0; r < 6; ++r) {
      for (int c = 0; c < 6; ++c)
        H[6 * r + c] += omega * (((J[0][r] * J[0][c] + J[1][r] * J[1][c]) + J[2][r] * J[2][c]) + J[3][r] * J[3][c]);
      b[r] += omega * (((J[0][r] * e[0] + J[1][r] * e[1]) + J[2][r] * e[2]) + J[3][r] * e[3]);
    }
  }
  io.n_outliers = io.n - io.n_inliers;
}

/* UVDAligner::linearize (uvd_aligner.cpp:72-171): residual (u, v, depth), diagonal information */
void aligner_linearize_uvd(const AlignerParams& P, AlignerIO& io, bool ignore_outliers, real H[36], real b[6]) {
  std::memset(H, 0, 36 * sizeof(real));
  std::memset(b, 0, 6 * sizeof(real));
  io.n_inliers = 0;
  io.total_error = 0;
  for (int u = 0; u < io.n; ++u) {
    io.errors[u] = -1;
    io.inliers[u] = 0;
    real w_uv = io.omega[u], w_d = io.fixed[4 * u + 3];
    real p[3];
    tf_apply(io.T, &io.moving[3 * u], p);
    if (p[2] <= P.min_depth) continue;                                                       /* :91 */
    real a[3];
    mat3_mul_vec(P.K, p, a);
    const real uu = a[0] / a[2], vv = a[1] / a[2];
    if (uu < 0 || uu > P.cols || vv < 0 || vv > P.rows) continue;                            /* :105-108 */
    const real e[3] = {uu - io.fixed[4 * u + 0], vv - io.fixed[4 * u + 1], p[2] - io.fixed[4 * u + 2]};
    const real chi = ((e[0] * w_uv) * e[0] + (e[1] * w_uv) * e[1]) + (e[2] * w_d) * e[2];   /* e^T Omega e, Omega diagonal */
    io.errors[u] = chi;
    if (chi > P.kernel) {
      if (ignore_outliers) continue;
      const real sc = P.kernel / chi;
      w_uv *= sc; w_d *= sc;
    } else {
      io.inliers[u] = 1;
      ++io.n_inliers;
    }
    io.total_error += chi;
    const real w = io.weight[u];
    real Jt[3][6] = {{w, 0, 0, 0, 2 * p[2], -2 * p[1]},
                     {0, w, 0, -2 * p[2], 0, 2 * p[0]},
                     {0, 0, w, 2 * p[1], -2 * p[0], 0}};
    real KJ[3][6];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 6; ++j)
        KJ[i][j] = (P.K[3 * i + 0] * Jt[0][j] + P.K[3 * i + 1] * Jt[1][j]) + P.K[3 * i + 2] * Jt[2][j];
    const real iz = 1 / p[2], iz2 = iz * iz;                                                 /* :140-141 */
    real J[3][6];
    for (int j = 0; j < 6; ++j) {
      J[0][j] = iz * KJ[0][j] + (-a[0] * iz2) * KJ[2][j];
      J[1][j] = iz * KJ[1][j] + (-a[1] * iz2) * KJ[2][j];
      J[2][j] = KJ[2][j];
    }
    for (int r = 0; r < 6; ++r) {
      for (int c = 0; c < 6; ++c)
        H[6 * r + c] += w_uv * (J[0][r] * J[0][c] + J[1][r] * J[1][c]) + w_d * (J[2][r] * J[2][c]);
      b[r] += w_uv * (J[0][r] * e[0] + J[1][r] * e[1]) + w_d * (J[2][r] * e[2]);
    }
  }
  io.n_outliers = io.n - io.n_inliers;
}

/* oneRound (:190-207) */
void aligner_one_round(const AlignerParams& P, AlignerIO& io, bool ignore_outliers) {
  real b[6];
  if (io.uvd) aligner_linearize_uvd(P, io, ignore_outliers, io.H, b);
  else aligner_linearize(P, io, ignore_outliers, io.H, b);
  for (int i = 0; i < 6; ++i) io.H[7 * i] += P.damping * io.n;
  real nb[6], dx[6];
  for (int i = 0; i < 6; ++i) nb[i] = -b[i];
  full_piv_lu_solve<6>(io.H, nb, dx);
  io.T = tf_mul(v2t(dx), io.T);
  /* R <- R - 0.5*R*(R^T R - I) */
  real R[9], RtR[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = io.T.m[4 * i + j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      RtR[3 * i + j] = (R[0 + i] * R[0 + j] + R[3 + i] * R[3 + j]) + R[6 + i] * R[6 + j];
      if (i == j) RtR[3 * i + j] -= 1;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      io.T.m[4 * i + j] = R[3 * i + j] - 0.5 * ((R[3 * i + 0] * RtR[j] + R[3 * i + 1] * RtR[3 + j]) + R[3 * i + 2] * RtR[6 + j]);
  ++io.iterations;
}

/* converge (:210-264) */
void aligner_converge(const AlignerParams& P, AlignerIO& io) {
  io.errors.assign(io.n, -1);
  io.inliers.assign(io.n, 0);
  io.iterations = 0;
  io.converged = 0;
  std::memset(io.H, 0, sizeof io.H);
  real total_error_previous = 0;
  for (int it = 0; it < P.max_it; ++it) {
    aligner_one_round(P, io, false);
    if (P.delta > std::fabs(total_error_previous - io.total_error)) {
      total_error_previous = io.total_error;
      if (io.n_inliers > (io.uvd ? 100 : P.min_inliers) && io.n_inliers > io.n_outliers) {   /* uvd_aligner.cpp:211 */
        for (int it2 = 0; it2 < P.max_it; ++it2) {
          aligner_one_round(P, io, true);
          if (std::fabs(total_error_previous - io.total_error) < P.delta) {
            total_error_previous = io.total_error;
            break;
          } else {
            total_error_previous = io.total_error;
          }
        }
      }
      io.converged = 1;
      break;
    } else {
      total_error_previous = io.total_error;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * one sequence: StereoFramePointGenerator + StereoUVAligner + PoseTracker3D control logic
 * ---------------------------------------------------------------------------------------- */
struct Region { int x, y, w, h; };

struct Stream {
  vslam_config cfg;
  /* BaseFramePointGenerator::configure (base_framepoint_generator.cpp:165-329) */
  std::vector<Region> regions;
  std::vector<int> thr;        /* FAST detector thresholds (int, FastDetector::setThreshold rint) */
  std::vector<real> thr_acc;   /* _detector_thresholds */
  int n_detections = 0;
  int rows_bin = 0, cols_bin = 0, target_kp = 0;
  uint32_t target_per_detector = 0;
  FeatureStore storeL, storeR;
  std::vector<int32_t> sumL, sumR; /* integral images of the current frame (recoverPoints) */
  std::vector<uint8_t> blurL, blurR; /* 7x7 Gaussian-blurred images of the current frame (ORB extractor) */
  std::vector<Feature> kpL, kpR;   /* Frame::keypoints/descriptors after the border filter */
  int n_detected_left = 0, n_detected_right_raw = 0, n_detected_left_raw = 0;
  real tau_tri;                /* _current_maximum_descriptor_distance_triangulation */
  /* PoseTracker3D state (pose_tracker_3d.h:79-113) */
  int status = VSLAM_LOCALIZING;
  Tf prior = tf_identity();    /* _previous_to_current_camera */
  int win = 0;                 /* _projection_tracking_distance_pixels */
  real tau_track = 0;          /* _current_descriptor_distance_tracking */
  real gen_tau_track = 0;      /* generator's _maximum_descriptor_distance_tracking (set per _track) */
  uint32_t n_tracked_landmarks = 0, n_tracked_points = 0, n_tracked_landmarks_prev = 0, n_active_landmarks = 0;
  Tf world_pose = tf_identity(); /* WorldMap::robot_to_world (identity robot offset) */
  std::vector<FrameRec> frames;
  std::vector<Landmark> landmarks;
  std::vector<int> lost;       /* _lost_points: indices into the previous frame's points */
  AlignerIO al;
  bool aligner_valid = false;  /* aligner ran on the current frame->points() (quirk B.3) */
  /* the 8 chronometers SLAMAssembly::printReport prints (slam_assembly.cpp:703-742; CREATE_CHRONOMETER in
   * base_framepoint_generator.h:232-233, stereo_framepoint_generator.h:81, pose_tracker_3d.h:123-127), seconds:
   * keypoint_detection, descriptor_extraction, point_triangulation, tracking, track_creation, pose_optimization,
   * landmark_optimization, point_recovery */
  double chrono[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  struct Chrono {
    double* acc; std::chrono::steady_clock::time_point t0;
    explicit Chrono(double* a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~Chrono() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  };
  vslam_frame_info info;
  std::vector<Tf> poses;

  void configure(const vslam_config& c) {
    cfg = c;
    const int nv = cfg.det_rows, nh = cfg.det_cols;
    const real ph = (real)cfg.rows / nv, pw = (real)cfg.cols / nh;
    regions.clear();
    for (int r = 0; r < nv; ++r)
      for (int cc = 0; cc < nh; ++cc) {
        int off_w = nh > 1 ? 2 : 0, off_h = nv > 1 ? 2 : 0, off_r = 0, off_c = 0;
        if (r > 0) { off_r = -off_h; if (r < nv - 1) off_h *= 2; }
        if (cc > 0) { off_c = -off_w; if (cc < nh - 1) off_w *= 2; }
        Region R;
        R.x = (int)(std::round(cc * pw) + off_c);
        R.y = (int)(std::round(r * ph) + off_r);
        R.w = (int)(pw + off_w);
        R.h = (int)(ph + off_h);
        regions.push_back(R);
      }
    thr.assign(regions.size(), cfg.detector_threshold_minimum);
    thr_acc.assign(regions.size(), 0);
    n_detections = 0;
    cols_bin = (int)(std::floor((real)cfg.cols / cfg.bin_size_pixels) + 1);
    rows_bin = (int)(std::floor((real)cfg.rows / cfg.bin_size_pixels) + 1);
    target_kp = cols_bin * rows_bin;
    target_per_detector = (uint32_t)((real)target_kp / (real)regions.size());
    storeL.configure(cfg.rows, cfg.cols);
    storeR.configure(cfg.rows, cfg.cols);
    reset();
  }
  void reset() {
    /* PoseTracker3D::configure (pose_tracker_3d.cpp:11-21) */
    status = VSLAM_LOCALIZING;
    prior = tf_identity();
    win = cfg.maximum_projection_tracking_distance_pixels;
    tau_track = cfg.minimum_descriptor_distance_tracking;
    tau_tri = 0.1 * 256;
    thr.assign(regions.size(), cfg.detector_threshold_minimum);
    thr_acc.assign(regions.size(), 0);
    n_detections = 0;
    n_tracked_landmarks = n_tracked_points = n_tracked_landmarks_prev = n_active_landmarks = 0;
    world_pose = tf_identity();
    frames.clear();
    landmarks.clear();
    lost.clear();
    poses.clear();
    aligner_valid = false;
    al.weight.clear();           /* a new StereoUVAligner: empty _weights_translation */
    std::memset(&info, 0, sizeof info);
  }

  /* detectKeypoints (base_framepoint_generator.cpp:355-429): per-region FAST + controller */
  void detect_keypoints(const uint8_t* img, int stride, std::vector<Keypoint>& out, int& n_raw) {
    out.clear();
    std::vector<Keypoint> kps;
    for (size_t i = 0; i < regions.size(); ++i) {
      const Region& R = regions[i];
      fast_detect_roi(img, stride, R.x, R.y, R.w, R.h, thr[i], kps);
      real t = thr[i];
      const real target = (real)target_per_detector;
      const real delta = ((real)kps.size() - target) / target;
      if (delta < -cfg.target_number_of_keypoints_tolerance) {
        const real change = std::max(delta, -cfg.detector_threshold_maximum_change);
        t = t + std::min(change * t, -1.0);
        if (t < cfg.detector_threshold_minimum) t = cfg.detector_threshold_minimum;
      } else if (delta > cfg.target_number_of_keypoints_tolerance) {
        const real change = std::min(delta, cfg.detector_threshold_maximum_change);
        t += std::max(change * t, 1.0);
        if (t > cfg.detector_threshold_maximum) t = cfg.detector_threshold_maximum;
      }
      thr_acc[i] += t;
      for (Keypoint k : kps) { k.x = (int16_t)(k.x + R.x); k.y = (int16_t)(k.y + R.y); out.push_back(k); }
    }
    ++n_detections;
    n_raw = (int)out.size();
  }
  /* adjustDetectorThresholds (:440-459) */
  void adjust_thresholds() {
    for (size_t i = 0; i < regions.size(); ++i) {
      thr_acc[i] /= n_detections;
      thr[i] = (int)std::rint(thr_acc[i]);
      thr_acc[i] = 0;
    }
    n_detections = 0;
  }
  /* computeDescriptors (:431-438): border filter + BRIEF */
  void compute_descriptors(const uint8_t* img, int stride, const std::vector<Keypoint>& kps,
                           std::vector<int32_t>& sum, std::vector<uint8_t>& blur, std::vector<Feature>& out) {
    out.clear();
    if (cfg.descriptor_type == VSLAM_DESCRIPTOR_ORB) {   /* cv::ORB::create() as extractor (:190-196,219-224) */
      gaussian_blur7_u8(img, cfg.rows, cfg.cols, stride, blur);
      float a, b;
      orb_rotation(-1.f, &a, &b);                        /* FAST keypoints: KeyPoint::angle = -1, never recomputed */
      for (const Keypoint& k : kps) {
        if (!orb_inside(cfg.rows, cfg.cols, k.x, k.y)) continue;
        Feature f;
        f.row = k.y; f.col = k.x; f.score = k.score;
        orb_at(blur.data(), cfg.cols, k.x, k.y, a, b, f.desc);
        out.push_back(f);
      }
      return;
    }
    integral_image(img, cfg.rows, cfg.cols, stride, sum);
    for (const Keypoint& k : kps) {
      if (!brief_inside(cfg.rows, cfg.cols, k.x, k.y)) continue;
      Feature f;
      f.row = k.y; f.col = k.x; f.score = k.score;
      brief_at(sum, cfg.cols, k.x, k.y, f.desc);
      out.push_back(f);
    }
  }
  /* the extractor on the 71 x 71 region around a projected landmark (stereo_framepoint_generator.cpp:773-812): the pattern
   * stays 14 px and more inside the region, so its border handling never reaches a tap — same bytes as on the whole image */
  void describe_at(bool right, int x, int y, uint8_t desc[32]) const {
    if (cfg.descriptor_type == VSLAM_DESCRIPTOR_ORB) {
      float a, b;
      orb_rotation(-1.f, &a, &b);                        /* keypoint_buffer = the previous point's FAST keypoint: angle -1 */
      orb_at((right ? blurR : blurL).data(), cfg.cols, x, y, a, b, desc);
    } else {
      brief_at(right ? sumR : sumL, cfg.cols, x, y, desc);
    }
  }

  /* StereoFramePointGenerator::initialize (stereo_framepoint_generator.cpp:73-133) */
  void initialize(const uint8_t* L, const uint8_t* R, int stride, bool extract, int frame_status) {
    if (extract) {
      std::vector<Keypoint> kl, kr;
      { Chrono t(&chrono[0]); detect_keypoints(L, stride, kl, n_detected_left_raw); detect_keypoints(R, stride, kr, n_detected_right_raw); }
      adjust_thresholds();
      { Chrono t(&chrono[1]); compute_descriptors(L, stride, kl, sumL, blurL, kpL); compute_descriptors(R, stride, kr, sumR, blurR, kpR); }
      n_detected_left = (int)kpL.size();
      if (frame_status == VSLAM_LOCALIZING) {
        tau_tri = std::min(0.1 * 256, cfg.maximum_matching_distance_triangulation);
      } else {
        const real ratio = std::min((real)n_detected_left / target_kp, 1.0);
        tau_tri = std::max(ratio * cfg.maximum_matching_distance_triangulation, 0.1 * 256);
      }
    }
    storeL.set_features(kpL);
    storeR.set_features(kpR);
  }

  /* getPointInLeftCamera (:871-895) */
  void triangulate(int xL, int yL, int xR, int yR, real out[3]) const {
    const real bx = cfg.baseline_h[0], fx = cfg.K[0], fy = cfg.K[4], cx = cfg.K[2], cy = cfg.K[5];
    out[2] = bx / (real)(xR - xL);
    out[0] = 1 / fx * ((real)xL - cx) * out[2];
    out[1] = 1 / fy * ((real)(yL + yR) / 2.0 - cy) * out[2];
  }

  static bool to_int32(real v, int32_t& out) { /* C++ double->int32 truncation, guarded */
    if (!(v > -2147483648.0 && v < 2147483648.0)) return false;
    out = (int32_t)v;
    return true;
  }

  /* StereoFramePointGenerator::track (:464-681) */
  void track(FrameRec& cur, FrameRec& prev, const Tf& T, bool by_appearance) {
    std::vector<Point>& pts = cur.points;
    pts.clear();
    lost.clear();
    std::set<uint32_t> matchedL, matchedR;
    n_tracked_landmarks = 0;
    const int rows = cfg.rows, cols = cfg.cols, d = win;
    for (size_t ip = 0; ip < prev.points.size(); ++ip) {
      Point& pp = prev.points[ip];
      real q[3], uvw[3];
      tf_apply(T, pp.cam, q);
      mat3_mul_vec(cfg.K, q, uvw);
      if (!(uvw[2] > 0)) continue; /* quirk B.5: no positivity check in the reference */
      int32_t col, row;
      if (!to_int32(uvw[0] / uvw[2], col) || !to_int32(uvw[1] / uvw[2], row)) continue;
      if (col < 0 || col > cols || row < 0 || row > rows) continue;
      real dist_best = gen_tau_track;
      int r0 = std::max(row - d, 0), r1 = std::min(row + d + 1, rows);
      int c0 = std::max(col - d, 0), c1 = std::min(col + d + 1, cols);
      const int fl = storeL.match_in_region(row, col, pp.dL, r0, r1, c0, c1, gen_tau_track, by_appearance, dist_best);
      if (fl >= 0) {
        const Feature& FL = storeL.feats[fl];
        const float ex = (float)col - (float)FL.col, ey = (float)row - (float)FL.row; /* cv::Point2f */
        real uvwR[3];
        for (int i = 0; i < 3; ++i) uvwR[i] = uvw[i] + cfg.baseline_h[i];
        int32_t colR, rowR;
        if (!to_int32(uvwR[0] / uvwR[2] - ex, colR) || !to_int32(uvwR[1] / uvwR[2] - ey, rowR)) continue;
        if (colR < 0 || colR > cols || rowR < 0 || rowR > rows) continue;
        const int32_t k = (int32_t)std::fabs((real)pp.epi);
        r0 = std::max(rowR - k, 0); r1 = std::min(rowR + k + 1, rows);
        c0 = std::max(colR - d, 0); c1 = std::min(colR + d + 1, FL.col);
        const int fr = storeR.match_in_region(rowR, colR, FL.desc, r0, r1, c0, c1, tau_tri, true, dist_best);
        if (fr >= 0) {
          const Feature& FR = storeR.feats[fr];
          if (FL.col - FR.col < cfg.minimum_disparity_pixels) continue;
          if (hamming32(FR.desc, pp.dR) > gen_tau_track) continue;
          for (int c = FR.col + 1; c < FL.col; ++c) {
            int& cell = storeR.lattice[(size_t)FR.row * cols + c];
            if (cell >= 0) { matchedR.insert((uint32_t)cell); cell = -1; }
          }
          Point np;
          std::memset(&np, 0, sizeof np);
          np.xL = FL.col; np.yL = FL.row; np.xR = FR.col; np.yR = FR.row;
          std::memcpy(np.dL, FL.desc, 32); std::memcpy(np.dR, FR.desc, 32);
          np.dist = (int)dist_best;
          triangulate(np.xL, np.yL, np.xR, np.yR, np.cam);
          np.prev = (int)ip;
          np.track_len = pp.track_len + 1;
          np.lm = pp.lm;
          np.epi = FR.row - FL.row;
          np.has_next = false;
          np.chi = -1; np.inlier = 0;
          pp.has_next = true;
          pts.push_back(np);
          matchedL.insert((uint32_t)fl);
          matchedR.insert((uint32_t)fr);
          storeL.lattice[(size_t)FL.row * cols + FL.col] = -1;
          storeR.lattice[(size_t)FR.row * cols + FR.col] = -1;
          if (pp.lm >= 0) ++n_tracked_landmarks;
        }
      }
      if (!pp.has_next) lost.push_back((int)ip);
    }
    /* ids == positions here (fresh stores) */
    storeL.prune_positions(matchedL);
    storeR.prune_positions(matchedR);
  }

  /* StereoFramePointGenerator::compute (:135-462) */
  int compute(FrameRec& cur) {
    std::vector<Point>& pts = cur.points;
    const size_t n_tracked = pts.size();
    const int bin = cfg.bin_size_pixels;
    std::vector<int> bin_map; /* index into `cand` (>=0) or -(tracked index)-2, -1 empty */
    auto bin_of = [&](int row, int col, int& rb, int& cb) {
      rb = (int)std::rint((real)row / bin);
      cb = (int)std::rint((real)col / bin);
      rb = std::min(rb, rows_bin - 1); /* quirk B.6: clamp the latent out-of-bounds index */
      cb = std::min(cb, cols_bin - 1);
    };
    if (cfg.enable_keypoint_binning) {
      bin_map.assign((size_t)rows_bin * cols_bin, -1);
      for (size_t i = 0; i < n_tracked; ++i) {
        int rb, cb;
        bin_of(pts[i].yL, pts[i].xL, rb, cb);
        bin_map[(size_t)rb * cols_bin + cb] = -(int)i - 2;
      }
    }
    storeL.sort_vec();
    storeR.sort_vec();
    std::vector<Point> cand;
    std::vector<int> offsets;
    offsets.push_back(0);
    for (int u = 1; u <= cfg.maximum_epipolar_search_offset_pixels; ++u) { offsets.push_back(u); offsets.push_back(-u); }
    for (int o : offsets) {
      std::vector<int>& FLv = storeL.vec;
      std::vector<int>& FRv = storeR.vec;
      std::set<uint32_t> mL, mR;
      uint32_t iR = 0;
      for (uint32_t iL = 0; iL < FLv.size(); iL++) {
        if (iR == FRv.size()) break;
        while (storeL.feats[FLv[iL]].row < storeR.feats[FRv[iR]].row + o) { iL++; if (iL == FLv.size()) break; }
        if (iL == FLv.size()) break;
        const Feature& fL = storeL.feats[FLv[iL]];
        while (fL.row > storeR.feats[FRv[iR]].row + o) { iR++; if (iR == FRv.size()) break; }
        if (iR == FRv.size()) break;
        uint32_t is = iR;
        real best = tau_tri;
        uint32_t ibest = 0;
        while (fL.row == storeR.feats[FRv[is]].row + o) {
          if (fL.col - storeR.feats[FRv[is]].col < 0) break;
          const real dd = hamming32(fL.desc, storeR.feats[FRv[is]].desc);
          if (dd < best) { best = dd; ibest = is; }
          is++;
          if (is == FRv.size()) break;
        }
        if (best < tau_tri) {
          const Feature& fR = storeR.feats[FRv[ibest]];
          if (fL.col - fR.col < cfg.minimum_disparity_pixels) continue;
          Point np;
          std::memset(&np, 0, sizeof np);
          np.xL = fL.col; np.yL = fL.row; np.xR = fR.col; np.yR = fR.row;
          std::memcpy(np.dL, fL.desc, 32); std::memcpy(np.dR, fR.desc, 32);
          np.dist = (int)best;
          triangulate(np.xL, np.yL, np.xR, np.yR, np.cam);
          np.prev = -1; np.track_len = 0; np.lm = -1; np.epi = o; np.has_next = false; np.chi = -1;
          if (cfg.enable_keypoint_binning) {
            int rb, cb;
            bin_of(fL.row, fL.col, rb, cb);
            int& cell = bin_map[(size_t)rb * cols_bin + cb];
            if (cell != -1) {
              if (cell >= 0) { /* occupant is an untracked candidate */
                const Point& c = cand[cell];
                if ((np.xL - np.xR) > (c.xL - c.xR) && np.dist <= c.dist) cell = (int)cand.size();
              } /* tracked occupant (previous() != null): never replaced */
            } else {
              cell = (int)cand.size();
            }
          }
          cand.push_back(np);
          mL.insert(iL);
          mR.insert(ibest);
          storeL.lattice[(size_t)fL.row * cfg.cols + fL.col] = -1;
          storeR.lattice[(size_t)fR.row * cfg.cols + fR.col] = -1;
          iR = ibest + 1;
        }
      }
      storeL.prune_positions(mL);
      storeR.prune_positions(mR);
    }
    int added = 0;
    if (cfg.enable_keypoint_binning) {
      for (int r = 0; r < rows_bin; ++r)
        for (int c = 0; c < cols_bin; ++c) {
          const int cell = bin_map[(size_t)r * cols_bin + c];
          if (cell >= 0) { pts.push_back(cand[cell]); ++added; }
        }
    } else {
      for (const Point& p : cand) { pts.push_back(p); ++added; }
    }
    return added;
  }

  /* StereoFramePointGenerator::recoverPoints (:683-869) */
  int recover(FrameRec& cur, FrameRec& prev) {
    int recovered = 0;
    const real* K = cfg.K;
    for (int ip : lost) {
      Point& pp = prev.points[ip];
      if (pp.lm < 0) continue;
      real pc[3], uL[3], uR[3];
      tf_apply(cur.world_to_cam, landmarks[pp.lm].w, pc);
      mat3_mul_vec(K, pc, uL);
      for (int i = 0; i < 3; ++i) uR[i] = uL[i] + cfg.baseline_h[i];
      if (uL[2] < cfg.minimum_depth_meters || uL[2] > cfg.maximum_depth_meters ||
          uR[2] < cfg.minimum_depth_meters || uR[2] > cfg.maximum_depth_meters) continue;
      const float pLx = (float)std::rint(uL[0] / uL[2]), pLy = (float)std::rint(uL[1] / uL[2]);
      const float pRx = (float)std::rint(uR[0] / uR[2]), pRy = (float)std::rint(uR[1] / uR[2]);
      const float border = 5 * 7.f; /* 5*keypoint.size, FAST size = 7 */
      if (pLx < border + 1 || pLx > cfg.cols - border - 1 || pRx < border + 1 || pRx > cfg.cols - border - 1 ||
          pLy < border + 1 || pLy > cfg.rows - border - 1 || pRy < border + 1 || pRy > cfg.rows - border - 1) continue;
      const int xL = (int)pLx, yL = (int)pLy, xR = (int)pRx, yR = (int)pRy;
      uint8_t dL[32], dR[32];
      describe_at(false, xL, yL, dL);
      if (hamming32(pp.dL, dL) > gen_tau_track) continue;
      describe_at(true, xR, yR, dR);
      if ((real)(pLx - pRx) < cfg.minimum_disparity_pixels) continue;
      if (hamming32(pp.dR, dR) > gen_tau_track) continue;
      const int dtri = hamming32(dL, dR);
      if (dtri > tau_tri) continue;
      Point np;
      std::memset(&np, 0, sizeof np);
      np.xL = xL; np.yL = yL; np.xR = xR; np.yR = yR;
      std::memcpy(np.dL, dL, 32); std::memcpy(np.dR, dR, 32);
      np.dist = dtri;
      triangulate(xL, yL, xR, yR, np.cam);
      np.prev = ip; np.track_len = pp.track_len + 1; np.lm = pp.lm; np.epi = 0; np.has_next = false; np.chi = -1;
      pp.has_next = true;
      cur.points.push_back(np);
      ++recovered;
    }
    return recovered;
  }

  /* StereoUVAligner::initialize (stereouv_aligner.cpp:10-69) + converge */
  void align(FrameRec& cur, FrameRec& prev, bool inverse_depth) {
    Chrono timer(&chrono[5]);
    const int n = (int)cur.points.size();
    al.n = n;
    al.moving.resize(3 * n); al.fixed.resize(4 * n); al.omega.resize(n);
    /* _weights_translation.resize(n, 1) (stereouv_aligner.cpp:22): a std::vector resize keeps the elements it already
     * holds and gives 1 only to the ones it appends; they are rewritten only when inverse depth is enabled (:57-61).
     * The vector is a member of the aligner, so the Localizing frames after a breakTrack (inverse depth off,
     * pose_tracker_3d.cpp:124) run with the weights the last Tracking frame left at the same indices. */
    aligner_resize_weights(n);
    for (int u = 0; u < n; ++u) {
      const Point& p = cur.points[u];
      const Point& pp = prev.points[p.prev];
      al.fixed[4 * u + 0] = p.xL; al.fixed[4 * u + 1] = p.yL; al.fixed[4 * u + 2] = p.xR; al.fixed[4 * u + 3] = p.yR;
      real om = 1;
      if (pp.lm >= 0) {
        for (int i = 0; i < 3; ++i) al.moving[3 * u + i] = pp.cam_lm[i];
        om *= (1 + std::log((real)landmarks[pp.lm].updates));
      } else {
        for (int i = 0; i < 3; ++i) al.moving[3 * u + i] = pp.cam[i];
      }
      al.omega[u] = om;
      if (inverse_depth) aligner_set_weight(u, p.cam[2]);
    }
    al.T = prior;
    AlignerParams P = aligner_params();
    aligner_converge(P, al);
    for (int u = 0; u < n; ++u) { cur.points[u].chi = al.errors[u]; cur.points[u].inlier = al.inliers[u]; }
    aligner_valid = true;
  }
  void aligner_resize_weights(int n) { al.weight.resize(n, 1.0); }                                   /* :22 */
  void aligner_set_weight(int u, real depth) { al.weight[u] = std::min(cfg.maximum_reliable_depth_meters / depth, 1.0); } /* :60 */
  AlignerParams aligner_params() const {
    AlignerParams P;
    std::memcpy(P.K, cfg.K, sizeof P.K);
    std::memcpy(P.baseline, cfg.baseline_h, sizeof P.baseline);
    P.rows = cfg.rows; P.cols = cfg.cols;
    P.min_depth = cfg.minimum_depth_meters; /* setMinimumReliableDepthMeters, slam_assembly.cpp:70 */
    P.kernel = cfg.aligner_maximum_error_kernel; P.damping = cfg.aligner_damping;
    P.delta = cfg.aligner_error_delta_for_convergence;
    P.max_it = cfg.aligner_maximum_number_of_iterations; P.min_inliers = cfg.aligner_minimum_number_of_inliers;
    return P;
  }

  /* PoseTracker3D::_track (pose_tracker_3d.cpp:225-298) */
  void tracker_track(FrameRec& cur, FrameRec& prev, bool by_appearance) {
    Chrono timer(&chrono[3]);
    if (by_appearance) win = cfg.maximum_projection_tracking_distance_pixels;
    aligner_valid = false;
    gen_tau_track = tau_track; /* setMaximumDescriptorDistanceTracking (:238) */
    track(cur, prev, prior, by_appearance);
    n_tracked_points = (uint32_t)cur.points.size();
    adapt_search(prev.points.size());
    ++info.track_attempts;
  }
  /* the adaptive part of _track (:240-288): search window and descriptor distance from the tracking statistics */
  void adapt_search(size_t n_previous_points) {
    const real tracking_ratio = (real)n_tracked_points / (real)n_previous_points;
    const real landmark_per_point = (real)n_tracked_landmarks / (real)n_tracked_points;
    const real success_ratio = (real)n_tracked_points / (real)target_kp;
    const int wmax = cfg.maximum_projection_tracking_distance_pixels, wmin = cfg.minimum_projection_tracking_distance_pixels;
    if (tracking_ratio < cfg.good_tracking_ratio / 2) {
      if (win < wmax) win = (int32_t)std::min(win * 1 / cfg.tunnel_vision_ratio, (real)wmax);
    } else {
      if (win > wmin) win = (int32_t)std::max(win * cfg.tunnel_vision_ratio, (real)wmin);
    }
    if (tracking_ratio < cfg.good_tracking_ratio ||
        n_tracked_points < (uint32_t)cfg.aligner_minimum_number_of_inliers ||
        (landmark_per_point < 0.5 && success_ratio < 0.25)) {
      tau_track += 5;
      if (tau_track > cfg.maximum_descriptor_distance_tracking) tau_track = cfg.maximum_descriptor_distance_tracking;
    } else {
      tau_track -= 5;
      if (tau_track < cfg.minimum_descriptor_distance_tracking) tau_track = cfg.minimum_descriptor_distance_tracking;
    }
  }
  void fallback(FrameRec& cur, FrameRec& prev) { /* _fallbackEstimate (:551-566) */
    prior = tf_identity();
    set_pose(cur, prev.cam_to_world);
    info.fallback = 1;
  }
  void set_pose(FrameRec& f, const Tf& c2w) { f.cam_to_world = c2w; f.world_to_cam = tf_inverse(c2w); }
  /* accept-or-fallback block shared by :139-159 and :372-388 */
  void accept_motion(FrameRec& cur, FrameRec& prev) {
    const Tf& T = al.T;
    const real dang = rotation_angle(T);
    const real dtr = std::sqrt((T.m[3] * T.m[3] + T.m[7] * T.m[7]) + T.m[11] * T.m[11]);
    if (dang > cfg.minimum_delta_angular_for_movement || dtr > cfg.minimum_delta_translational_for_movement) {
      prior = T;
      set_pose(cur, tf_mul(prev.cam_to_world, tf_inverse(prior)));
    } else {
      fallback(cur, prev);
    }
  }
  void break_track(FrameRec& cur, FrameRec& prev) { /* breakTrack (:422-435) */
    status = VSLAM_LOCALIZING;
    set_pose(cur, prev.cam_to_world);
    prior = tf_identity();
    n_tracked_points = 0;
    info.track_broken = 1;
  }
  /* _registerRecursive (:300-419) */
  void register_recursive(FrameRec& cur, FrameRec& prev, const uint8_t* L, const uint8_t* R, int stride, int rec) {
    const real rel = (real)n_tracked_landmarks / (real)n_tracked_landmarks_prev;
    if (n_tracked_landmarks == 0 || rel < 0.1) {
      if (rec < 2) {
        prior = tf_identity();
        initialize(L, R, stride, false, cur_status_at_start);
        tracker_track(cur, prev, true);
        register_recursive(cur, prev, L, R, stride, rec + 1);
      } else {
        break_track(cur, prev);
      }
      return;
    }
    align(cur, prev, true);
    if ((uint32_t)al.n_inliers > (uint32_t)cfg.minimum_number_of_landmarks_to_track) {
      accept_motion(cur, prev);
    } else {
      if (rec < 2) {
        if (win < cfg.maximum_projection_tracking_distance_pixels) ++win;
        initialize(L, R, stride, false, cur_status_at_start);
        tracker_track(cur, prev, false);
        register_recursive(cur, prev, L, R, stride, rec + 1);
      } else {
        break_track(cur, prev);
      }
    }
  }
  int cur_status_at_start = VSLAM_LOCALIZING;

  /* the selection rule of _prunePoints (:439-466) */
  bool prune_keeps(real average_error, real error, uint8_t inlier) const {
    if (average_error < cfg.aligner_maximum_error_kernel) return inlier != 0;
    return error != -1 && error < 100 * cfg.aligner_maximum_error_kernel;
  }
  /* _prunePoints (:437-472) */
  void prune(FrameRec& cur, FrameRec& prev) {
    std::vector<Point> kept;
    if (aligner_valid) {
      const real avg = al.total_error / (real)al.n;
      for (size_t i = 0; i < cur.points.size(); ++i) {
        const bool keep = prune_keeps(avg, al.errors[i], al.inliers[i]);
        if (keep) kept.push_back(cur.points[i]);
        else prev.points[cur.points[i].prev].has_next = false; /* FramePoint::clear (frame_point.cpp:57-82) */
      }
    } else {
      /* quirk B.3: the reference reads stale aligner buffers here; defined rule: drop all */
      for (const Point& p : cur.points) prev.points[p.prev].has_next = false;
    }
    cur.points.swap(kept);
    n_tracked_points = (uint32_t)cur.points.size();
  }

  /* Landmark::Landmark (landmark.cpp:8-33) */
  int create_landmark(int frame_index, int point_index) {
    Landmark lm;
    lm.w[0] = lm.w[1] = lm.w[2] = 0;
    int f = frame_index, i = point_index;
    while (i >= 0) {
      Point& p = frames[f].points[i];
      p.lm = (int)landmarks.size();
      Meas m;
      m.frame = f;
      for (int k = 0; k < 3; ++k) m.cam[k] = p.cam[k];
      m.inv_depth = 1 / p.cam[2];
      lm.meas.push_back(m);
      real wpt[3];
      tf_apply(frames[f].cam_to_world, p.cam, wpt);
      for (int k = 0; k < 3; ++k) lm.w[k] += wpt[k];
      i = p.prev;
      --f;
    }
    for (int k = 0; k < 3; ++k) lm.w[k] /= (real)lm.meas.size();
    lm.updates = (uint32_t)lm.meas.size();
    landmarks.push_back(lm);
    return (int)landmarks.size() - 1;
  }
  /* Landmark::update (landmark.cpp:66-167) */
  void update_landmark(Landmark& lm, int frame_index, const Point& p) {
    Meas nm;
    nm.frame = frame_index;
    for (int k = 0; k < 3; ++k) nm.cam[k] = p.cam[k];
    nm.inv_depth = 1 / p.cam[2];
    lm.meas.push_back(nm);
    real w[3] = {lm.w[0], lm.w[1], lm.w[2]};
    real err_prev = 0;
    for (int it = 0; it < cfg.landmark_maximum_number_of_iterations; ++it) {
      real H[9] = {0}, b[3] = {0};
      real err = 0;
      uint32_t n_out = 0;
      for (const Meas& m : lm.meas) {
        const Tf& W2C = frames[m.frame].world_to_cam;
        real s[3];
        tf_apply(W2C, w, s);
        if (s[2] <= 0) { ++n_out; continue; }
        const real e[3] = {s[0] - m.cam[0], s[1] - m.cam[1], s[2] - m.cam[2]};
        real om = m.inv_depth;
        const real e2 = om * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
        err += e2;
        if (e2 > cfg.landmark_maximum_error_squared_meters) { om *= cfg.landmark_maximum_error_squared_meters / e2; ++n_out; }
        /* J = R (world_to_cam.linear()); H += J^T om J ; b += J^T om e */
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c)
            H[3 * r + c] += om * ((W2C.m[0 + r] * W2C.m[0 + c] + W2C.m[4 + r] * W2C.m[4 + c]) + W2C.m[8 + r] * W2C.m[8 + c]);
          b[r] += om * ((W2C.m[0 + r] * e[0] + W2C.m[4 + r] * e[1]) + W2C.m[8 + r] * e[2]);
        }
      }
      real nb[3] = {-b[0], -b[1], -b[2]}, dx[3];
      full_piv_lu_solve<3>(H, nb, dx);
      for (int k = 0; k < 3; ++k) w[k] += dx[k];
      if (std::fabs(err - err_prev) < 1e-5 || it == 999) {
        const uint32_t n_in = (uint32_t)lm.meas.size() - n_out;
        if (n_in > lm.updates) {
          for (int k = 0; k < 3; ++k) lm.w[k] = w[k];
          lm.updates = n_in;
        } else if (n_in < n_out) {
          real acc[3] = {0, 0, 0};
          for (const Meas& m : lm.meas) {
            real wp[3];
            tf_apply(frames[m.frame].cam_to_world, m.cam, wp);
            for (int k = 0; k < 3; ++k) acc[k] += wp[k];
          }
          for (int k = 0; k < 3; ++k) lm.w[k] = acc[k] / (real)lm.meas.size();
        }
        break;
      }
      err_prev = err;
    }
  }
  /* _updatePoints (pose_tracker_3d.cpp:475-520) */
  void update_points(int frame_index) {
    FrameRec& cur = frames[frame_index];
    n_active_landmarks = 0;
    for (size_t i = 0; i < cur.points.size(); ++i) {
      Point& p = cur.points[i];
      if ((uint32_t)p.track_len < (uint32_t)cfg.minimum_track_length_for_landmark_creation) continue;
      if (p.lm < 0) {
        create_landmark(frame_index, (int)i);
      } else {
        update_landmark(landmarks[p.lm], frame_index, p);
      }
      tf_apply(cur.world_to_cam, landmarks[p.lm].w, p.cam_lm);
      ++n_active_landmarks;
    }
  }

  /* PoseTracker3D::compute (pose_tracker_3d.cpp:32-222) */
  void process(const uint8_t* L, const uint8_t* R, int stride) {
    const int findex = (int)frames.size();
    std::memset(&info, 0, sizeof info);
    info.frame_index = findex + 1;
    info.status_at_start = status;
    cur_status_at_start = status;
    n_tracked_points = 0;
    frames.emplace_back();
    FrameRec& cur = frames[findex];
    set_pose(cur, world_pose);
    const bool has_prev = findex > 0;
    initialize(L, R, stride, true, status);
    if (has_prev) {
      FrameRec& prev = frames[findex - 1];
      for (Point& p : prev.points) p.has_next = false;
      tracker_track(cur, prev, status == VSLAM_LOCALIZING);
      if (status == VSLAM_LOCALIZING) {
        if (n_tracked_points < (uint32_t)cfg.minimum_number_of_landmarks_to_track) {
          fallback(cur, prev);
        } else {
          align(cur, prev, false);
          if ((uint32_t)al.n_inliers < (uint32_t)cfg.minimum_number_of_landmarks_to_track) fallback(cur, prev);
          else accept_motion(cur, prev);
        }
      } else {
        register_recursive(cur, prev, L, R, stride, 0);
      }
    }
    world_pose = cur.cam_to_world;
    info.n_tracked = (int)cur.points.size();
    info.n_lost = (int)lost.size();
    info.n_tracked_landmarks = (int)n_tracked_landmarks;
    if (has_prev) {
      FrameRec& prev = frames[findex - 1];
      info.aligner_ran = aligner_valid ? 1 : 0;
      if (aligner_valid) {
        info.aligner_iterations = al.iterations; info.aligner_converged = al.converged;
        info.n_inliers = al.n_inliers; info.n_outliers = al.n_outliers; info.total_error = al.total_error;
      }
      prune(cur, prev);
      info.n_after_prune = (int)cur.points.size();
      if (cfg.enable_landmark_recovery) {
        { Chrono t(&chrono[7]); info.n_recovered = recover(cur, prev); }
        n_tracked_points = (uint32_t)cur.points.size();
      }
    }
    { Chrono t(&chrono[6]); update_points(findex); }
    if (n_active_landmarks > (uint32_t)cfg.minimum_number_of_landmarks_to_track) status = VSLAM_TRACKING;
    { const double before = chrono[2]; { Chrono t(&chrono[2]); info.n_new_stereo = compute(cur); } chrono[4] += chrono[2] - before; }
    n_tracked_landmarks_prev = n_active_landmarks;
    /* report */
    info.status = status;
    info.n_keypoints_left = (int)kpL.size(); info.n_keypoints_right = (int)kpR.size();
    info.n_detected_left = n_detected_left_raw; info.n_detected_right = n_detected_right_raw;
    for (size_t i = 0; i < thr.size() && i < VSLAM_MAX_REGIONS; ++i) info.thresholds[i] = thr[i];
    info.n_active_landmarks = (int)n_active_landmarks;
    info.n_points = (int)cur.points.size();
    info.window_pixels = win;
    info.tau_track = tau_track;
    info.tau_triangulation = tau_tri;
    std::memcpy(info.camera_left_to_world, cur.cam_to_world.m, sizeof(real) * 12);
    std::memcpy(info.previous_to_current, prior.m, sizeof(real) * 12);
    poses.push_back(cur.cam_to_world);
  }
};

} /* namespace */

struct orc_ctx {
  std::vector<Stream> streams;
  std::vector<uint8_t> active;   /* a stream whose sequence has ended is skipped (vslam_set_stream_active) */
  std::string err;
};

/* ------------------------------------------------------------------------------------------
 * C entry points: same shapes as include/vslam_hip.h, prefix orc_
 * ---------------------------------------------------------------------------------------- */
static void fill_common_defaults(vslam_config* c) {
  std::memset(c, 0, sizeof *c);
  c->det_rows = 1; c->det_cols = 1;
  c->detector_threshold_minimum = 20; c->detector_threshold_maximum = 100;
  c->detector_threshold_maximum_change = 0.1; c->target_number_of_keypoints_tolerance = 0.1;
  c->bin_size_pixels = 15; c->enable_keypoint_binning = 1;
  c->minimum_projection_tracking_distance_pixels = 15; c->maximum_projection_tracking_distance_pixels = 50;
  c->minimum_descriptor_distance_tracking = 25.6; c->maximum_descriptor_distance_tracking = 51.2;
  c->maximum_reliable_depth_meters = 15; c->maximum_depth_meters = 1000; c->minimum_depth_meters = 0.1;
  c->maximum_matching_distance_triangulation = 51.2; c->minimum_disparity_pixels = 1;
  c->maximum_epipolar_search_offset_pixels = 0;
  c->minimum_track_length_for_landmark_creation = 1; c->minimum_number_of_landmarks_to_track = 5;
  c->tunnel_vision_ratio = 0.5; c->good_tracking_ratio = 0.2; c->enable_landmark_recovery = 1;
  c->minimum_delta_angular_for_movement = 0.001; c->minimum_delta_translational_for_movement = 0.01;
  c->aligner_error_delta_for_convergence = 1e-3; c->aligner_maximum_error_kernel = 4; c->aligner_damping = 5;
  c->aligner_maximum_number_of_iterations = 1000; c->aligner_minimum_number_of_inliers = 100;
  c->landmark_maximum_error_squared_meters = 25; c->landmark_maximum_number_of_iterations = 100;
  c->max_keypoints = 16384; c->max_points = 8192; c->max_history_frames = 512;
}
ORC_API void orc_default_config_kitti(vslam_config* c) { /* configurations/configuration_kitti.yaml:49-134 */
  fill_common_defaults(c);
  c->rows = 376; c->cols = 1241;
  const double K[9] = {718.856, 0, 607.1928, 0, 718.856, 185.2157, 0, 0, 1};
  std::memcpy(c->K, K, sizeof K);
  c->baseline_h[0] = -386.1448; /* KITTI-00 calib.txt P1(0,3) */
}
ORC_API void orc_default_config_euroc(vslam_config* c) { /* configurations/configuration_euroc.yaml:47-116 */
  fill_common_defaults(c);
  c->rows = 480; c->cols = 752;
  const double K[9] = {458.654, 0, 367.215, 0, 457.296, 248.375, 0, 0, 1};
  std::memcpy(c->K, K, sizeof K);
  c->baseline_h[0] = -458.654 * 0.11;
  c->det_rows = 2; c->det_cols = 2;
  c->detector_threshold_minimum = 10; c->detector_threshold_maximum = 30; c->detector_threshold_maximum_change = 1.0;
  c->bin_size_pixels = 20;
  c->minimum_descriptor_distance_tracking = 25; c->maximum_descriptor_distance_tracking = 50;
  c->maximum_reliable_depth_meters = 5; c->maximum_depth_meters = 100;
  c->maximum_matching_distance_triangulation = 50;
  c->minimum_track_length_for_landmark_creation = 2; c->good_tracking_ratio = 0.25;
  c->aligner_damping = 0;
  c->descriptor_type = VSLAM_DESCRIPTOR_ORB; /* configuration_euroc.yaml:52 "ORB-256" -> cv::ORB::create() (:219-224) */
}

ORC_API int orc_create(const vslam_config* cfg, int /*device*/, int n_streams, orc_ctx** out) {
  if (!cfg || !out || n_streams < 1 || cfg->rows < 64 || cfg->cols < 64) return VSLAM_ERR_INVALID;
  if (cfg->det_rows * cfg->det_cols > VSLAM_MAX_REGIONS || cfg->det_rows < 1 || cfg->det_cols < 1) return VSLAM_ERR_INVALID;
  if (!(-cfg->baseline_h[0] / cfg->K[0] > 0)) return VSLAM_ERR_INVALID; /* stereo_framepoint_generator.cpp:28-34 */
  orc_ctx* c = new orc_ctx;
  c->streams.resize(n_streams);
  for (Stream& s : c->streams) s.configure(*cfg);
  *out = c;
  return VSLAM_OK;
}
ORC_API void orc_destroy(orc_ctx* c) { delete c; }
ORC_API int orc_reset(orc_ctx* c) { for (Stream& s : c->streams) s.reset(); c->active.clear(); return VSLAM_OK; }
ORC_API int orc_process_host(orc_ctx* c, const uint8_t* L, const uint8_t* R, int32_t stride, size_t image_stride) {
  if (!c || !L || !R) return VSLAM_ERR_INVALID;
  for (size_t s = 0; s < c->streams.size(); ++s)
    if (c->active.empty() || c->active[s]) c->streams[s].process(L + s * image_stride, R + s * image_stride, stride);
  return VSLAM_OK;
}
ORC_API int orc_set_stream_active(orc_ctx* c, int s, int on) {
  if (!c || s < 0 || s >= (int)c->streams.size()) return VSLAM_ERR_INVALID;
  if (c->active.empty()) c->active.assign(c->streams.size(), 1);
  c->active[s] = on ? 1 : 0;
  return VSLAM_OK;
}
ORC_API int orc_reset_stream(orc_ctx* c, int s) {
  if (!c || s < 0 || s >= (int)c->streams.size()) return VSLAM_ERR_INVALID;
  c->streams[s].reset();
  return VSLAM_OK;
}
ORC_API int orc_get_frame_info(orc_ctx* c, int s, vslam_frame_info* out) {
  if (!c || s < 0 || s >= (int)c->streams.size() || !out) return VSLAM_ERR_INVALID;
  *out = c->streams[s].info;
  return VSLAM_OK;
}
ORC_API int orc_get_keypoints(orc_ctx* c, int s, int side, int32_t cap, int32_t* n, int16_t* xy, int32_t* score, uint8_t* desc) {
  if (!c || s < 0 || s >= (int)c->streams.size() || !n) return VSLAM_ERR_INVALID;
  const std::vector<Feature>& k = side ? c->streams[s].kpR : c->streams[s].kpL;
  *n = (int32_t)k.size();
  if ((int32_t)k.size() > cap) return VSLAM_ERR_CAPACITY;
  for (size_t i = 0; i < k.size(); ++i) {
    if (xy) { xy[2 * i] = (int16_t)k[i].col; xy[2 * i + 1] = (int16_t)k[i].row; }
    if (score) score[i] = k[i].score;
    if (desc) std::memcpy(desc + 32 * i, k[i].desc, 32);
  }
  return VSLAM_OK;
}
ORC_API int orc_get_points(orc_ctx* c, int s, int32_t cap, int32_t* n, int16_t* kp, int32_t* meta, double* cam, double* lm) {
  if (!c || s < 0 || s >= (int)c->streams.size() || !n) return VSLAM_ERR_INVALID;
  Stream& st = c->streams[s];
  if (st.frames.empty()) { *n = 0; return VSLAM_OK; }
  const std::vector<Point>& p = st.frames.back().points;
  *n = (int32_t)p.size();
  if ((int32_t)p.size() > cap) return VSLAM_ERR_CAPACITY;
  for (size_t i = 0; i < p.size(); ++i) {
    if (kp) { kp[4 * i] = (int16_t)p[i].xL; kp[4 * i + 1] = (int16_t)p[i].yL; kp[4 * i + 2] = (int16_t)p[i].xR; kp[4 * i + 3] = (int16_t)p[i].yR; }
    if (meta) {
      meta[6 * i + 0] = p[i].dist; meta[6 * i + 1] = p[i].epi; meta[6 * i + 2] = p[i].prev; meta[6 * i + 3] = p[i].track_len;
      meta[6 * i + 4] = p[i].lm >= 0 ? (int32_t)st.landmarks[p[i].lm].updates : 0;
      meta[6 * i + 5] = p[i].xL - p[i].xR;
    }
    if (cam) for (int k = 0; k < 3; ++k) cam[3 * i + k] = p[i].cam[k];
    if (lm) for (int k = 0; k < 3; ++k) lm[3 * i + k] = p[i].lm >= 0 ? st.landmarks[p[i].lm].w[k] : 0.0;
  }
  return VSLAM_OK;
}
ORC_API int orc_get_aligner_result(orc_ctx* c, int s, int32_t cap, int32_t* n, double* chi, uint8_t* inlier, double T[12], double H[36]) {
  if (!c || s < 0 || s >= (int)c->streams.size() || !n) return VSLAM_ERR_INVALID;
  const AlignerIO& a = c->streams[s].al;
  *n = a.n;
  if (a.n > cap) return VSLAM_ERR_CAPACITY;
  for (int i = 0; i < a.n; ++i) { if (chi) chi[i] = a.errors[i]; if (inlier) inlier[i] = a.inliers[i]; }
  if (T) std::memcpy(T, a.T.m, sizeof(double) * 12);
  if (H) std::memcpy(H, a.H, sizeof(double) * 36);
  return VSLAM_OK;
}
/* _weights_translation as the last StereoUVAligner::initialize of the stream left it */
ORC_API int orc_get_aligner_weights(orc_ctx* c, int s, int32_t cap, int32_t* n, double* weight) {
  if (!c || s < 0 || s >= (int)c->streams.size() || !n) return VSLAM_ERR_INVALID;
  const AlignerIO& a = c->streams[s].al;
  *n = (int32_t)a.weight.size();
  if (*n > cap) return VSLAM_ERR_CAPACITY;
  for (int i = 0; i < *n; ++i) if (weight) weight[i] = a.weight[i];
  return VSLAM_OK;
}
/* the reference's 8 chronometers, summed over the streams of the context (seconds of host time) */
ORC_API int orc_get_timers(orc_ctx* c, double seconds[8]) {
  if (!c || !seconds) return VSLAM_ERR_INVALID;
  for (int k = 0; k < 8; ++k) { seconds[k] = 0; for (const Stream& s : c->streams) seconds[k] += s.chrono[k]; }
  return VSLAM_OK;
}
ORC_API int orc_get_poses(orc_ctx* c, int s, int32_t first, int32_t nf, double* out) {
  if (!c || s < 0 || s >= (int)c->streams.size() || !out) return VSLAM_ERR_INVALID;
  const std::vector<Tf>& p = c->streams[s].poses;
  if (first < 0 || first + nf > (int32_t)p.size()) return VSLAM_ERR_INVALID;
  for (int i = 0; i < nf; ++i) std::memcpy(out + 12 * i, p[first + i].m, sizeof(double) * 12);
  return VSLAM_OK;
}

/* ---- stand-alone stages ------------------------------------------------------------------ */
ORC_API int orc_fast_detect(orc_ctx*, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t rx, int32_t ry,
                            int32_t rw, int32_t rh, int32_t threshold, int32_t cap, int32_t* n, int16_t* xy, int32_t* score) {
  if (!img || !n || rx < 0 || ry < 0 || rx + rw > cols || ry + rh > rows) return VSLAM_ERR_INVALID;
  std::vector<Keypoint> k;
  fast_detect_roi(img, stride, rx, ry, rw, rh, threshold, k);
  *n = (int32_t)k.size();
  if ((int32_t)k.size() > cap) return VSLAM_ERR_CAPACITY;
  for (size_t i = 0; i < k.size(); ++i) { xy[2 * i] = k[i].x; xy[2 * i + 1] = k[i].y; if (score) score[i] = k[i].score; }
  return VSLAM_OK;
}
ORC_API int orc_brief_describe(orc_ctx*, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t n,
                               const int16_t* xy, uint8_t* keep, uint8_t* desc) {
  if (!img || !xy || !keep || !desc) return VSLAM_ERR_INVALID;
  std::vector<int32_t> sum;
  integral_image(img, rows, cols, stride, sum);
  for (int i = 0; i < n; ++i) {
    const int x = xy[2 * i], y = xy[2 * i + 1];
    keep[i] = brief_inside(rows, cols, x, y) ? 1 : 0;
    if (keep[i]) brief_at(sum, cols, x, y, desc + 32 * i);
    else std::memset(desc + 32 * i, 0, 32);
  }
  return VSLAM_OK;
}
ORC_API int orc_gaussian_blur7_u8(const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, uint8_t* out) {
  if (!img || !out || rows < 4 || cols < 4 || stride < cols) return VSLAM_ERR_INVALID;
  std::vector<uint8_t> b;
  gaussian_blur7_u8(img, rows, cols, stride, b);
  std::memcpy(out, b.data(), b.size());
  return VSLAM_OK;
}
ORC_API int orc_orb_describe(const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t n, const int16_t* xy, float angle_degrees,
                             uint8_t* keep, uint8_t* desc) {
  if (!img || !xy || !keep || !desc || n < 0 || rows < 4 || cols < 4 || stride < cols) return VSLAM_ERR_INVALID;
  std::vector<uint8_t> blur;
  gaussian_blur7_u8(img, rows, cols, stride, blur);
  float a, b;
  orb_rotation(angle_degrees, &a, &b);
  for (int i = 0; i < n; ++i) {
    const int x = xy[2 * i], y = xy[2 * i + 1];
    keep[i] = orb_inside(rows, cols, x, y) ? 1 : 0;
    if (keep[i]) orb_at(blur.data(), cols, x, y, a, b, desc + 32 * i);
    else std::memset(desc + 32 * i, 0, 32);
  }
  return VSLAM_OK;
}
/* knnMatch(k=2) of the use_matches block (stereo_framepoint_generator.cpp:168-206) */
ORC_API int orc_knn2(orc_ctx*, int norm, int32_t nq, const uint8_t* q, int32_t nt, const uint8_t* t, int32_t* idx, float* dist) {
  if (!q || !t || !idx || !dist || norm < 0 || norm > 3) return VSLAM_ERR_INVALID;
  /* matcher type -> norm (stereo_framepoint_generator.cpp:175-197): 0 HAMMING on the bytes; 1 L2, 2 L1, 3 SL2 on the
   * descriptors converted to CV_32F (:199-205).  The float sums are exact: integers below 2^24. */
  std::vector<float> qf, tf;
  if (norm != 0) {
    qf.resize((size_t)nq * 32); tf.resize((size_t)nt * 32);
    for (size_t k = 0; k < qf.size(); ++k) qf[k] = (float)q[k];   /* convertTo(temp_des, CV_32F) */
    for (size_t k = 0; k < tf.size(); ++k) tf[k] = (float)t[k];
  }
  for (int i = 0; i < nq; ++i) {
    float b0 = 3.0e38f, b1 = 3.0e38f;
    int i0 = -1, i1 = -1;
    for (int j = 0; j < nt; ++j) {
      float d;
      if (norm == 0) d = (float)hamming32(q + 32 * i, t + 32 * j);
      else {
        const float* a = &qf[(size_t)32 * i]; const float* c = &tf[(size_t)32 * j];
        d = 0;
        if (norm == 2) for (int k = 0; k < 32; ++k) d += std::fabs(a[k] - c[k]);
        else for (int k = 0; k < 32; ++k) { const float e = a[k] - c[k]; d += e * e; }
      }
      if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
      else if (d < b1) { b1 = d; i1 = j; }
    }
    idx[2 * i] = i0; idx[2 * i + 1] = i1;
    dist[2 * i] = i0 < 0 ? 0.f : (norm != 1 ? b0 : std::sqrt(b0));
    dist[2 * i + 1] = i1 < 0 ? 0.f : (norm != 1 ? b1 : std::sqrt(b1));
  }
  return VSLAM_OK;
}
/* The reference's dead block of compute() (stereo_framepoint_generator.cpp:168-273), for the CPU baseline's cost only — every
 * result is discarded there and here:
 *   knnMatch(k = 2) of the left against the right descriptors converted to CV_32F (:199-206) with the given norm, and — with
 *   `homography` != 0 — cv::findHomography(left points, right points of every first match, LMEDS, 1, mask, 1000, 0.99)
 *   (:232-262 with configuration_kitti.yaml:99-103) restated from OpenCV's published algorithm [recalled: calib3d
 *   fundam.cpp / ptsetreg.cpp]: niters = RANSACUpdateNumIters(0.99, 0.45, 4, 1000) minimal samples of 4 correspondences, a
 *   direct linear 4-point homography each, the median of the squared reprojection errors over ALL correspondences as the
 *   score; then the inlier mask (sigma = 2.5 * 1.4826 * (1 + 5 / (n - 4)) * sqrt(median)), a normalised DLT re-fit on the
 *   inliers (9 x 9 normal matrix, smallest eigenvector by Jacobi sweeps) and 10 Levenberg-Marquardt rounds on the 8
 *   free parameters.  Used by bench.py's cpu_baseline leg only; returns the number of query rows. */
namespace {
bool homography4(const double* x, const double* y, const double* u, const double* v, double H[9]) {   /* 8 x 8 linear system, h33 = 1 */
  double A[8][9];
  for (int i = 0; i < 4; ++i) {
    const double r0[9] = {x[i], y[i], 1, 0, 0, 0, -u[i] * x[i], -u[i] * y[i], u[i]};
    const double r1[9] = {0, 0, 0, x[i], y[i], 1, -v[i] * x[i], -v[i] * y[i], v[i]};
    for (int k = 0; k < 9; ++k) { A[2 * i][k] = r0[k]; A[2 * i + 1][k] = r1[k]; }
  }
  for (int c = 0; c < 8; ++c) {
    int p = c;
    for (int r = c + 1; r < 8; ++r) if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
    if (std::fabs(A[p][c]) < 1e-12) return false;
    if (p != c) for (int k = 0; k < 9; ++k) std::swap(A[p][k], A[c][k]);
    for (int r = 0; r < 8; ++r) if (r != c) { const double f = A[r][c] / A[c][c]; for (int k = c; k < 9; ++k) A[r][k] -= f * A[c][k]; }
  }
  for (int k = 0; k < 8; ++k) H[k] = A[k][8] / A[k][k];
  H[8] = 1;
  return true;
}
inline double reproj2(const double H[9], double x, double y, double u, double v) {
  const double w = 1.0 / (H[6] * x + H[7] * y + H[8]);
  const double du = (H[0] * x + H[1] * y + H[2]) * w - u, dv = (H[3] * x + H[4] * y + H[5]) * w - v;
  return du * du + dv * dv;
}
void smallest_eigenvector9(double M[9][9], double out[9]) {   /* cyclic Jacobi on a symmetric 9 x 9 */
  double V[9][9];
  for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) V[i][j] = i == j;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0;
    for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) off += M[i][j] * M[i][j];
    if (off < 1e-30) break;
    for (int p = 0; p < 8; ++p) for (int q = p + 1; q < 9; ++q) {
      if (std::fabs(M[p][q]) < 1e-300) continue;
      const double th = (M[q][q] - M[p][p]) / (2 * M[p][q]);
      const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1)), cs = 1 / std::sqrt(t * t + 1), sn = t * cs;
      for (int k = 0; k < 9; ++k) { const double a = M[k][p], b = M[k][q]; M[k][p] = cs * a - sn * b; M[k][q] = sn * a + cs * b; }
      for (int k = 0; k < 9; ++k) { const double a = M[p][k], b = M[q][k]; M[p][k] = cs * a - sn * b; M[q][k] = sn * a + cs * b; }
      for (int k = 0; k < 9; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = cs * a - sn * b; V[k][q] = sn * a + cs * b; }
    }
  }
  int m = 0;
  for (int i = 1; i < 9; ++i) if (M[i][i] < M[m][m]) m = i;
  for (int k = 0; k < 9; ++k) out[k] = V[k][m];
}
int dead_find_homography_lmeds(const std::vector<double>& px, const std::vector<double>& py, const std::vector<double>& qx, const std::vector<double>& qy) {
  const int n = (int)px.size();
  if (n < 5) return 0;
  const int niters = std::min(1000, (int)std::lround(std::log(1 - 0.99) / std::log(1 - std::pow(1 - 0.45, 4))));
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  auto next = [&rng](int m) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (int)((rng >> 33) % (uint64_t)m); };
  std::vector<float> err(n), tmp(n);
  double best[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, best_med = 1e300;
  for (int it = 0; it < niters; ++it) {
    int id[4];
    for (int k = 0; k < 4; ++k) { bool again; do { id[k] = next(n); again = false; for (int j = 0; j < k; ++j) again |= id[j] == id[k]; } while (again); }
    double x[4], y[4], u[4], v[4], H[9];
    for (int k = 0; k < 4; ++k) { x[k] = px[id[k]]; y[k] = py[id[k]]; u[k] = qx[id[k]]; v[k] = qy[id[k]]; }
    if (!homography4(x, y, u, v, H)) continue;
    for (int i = 0; i < n; ++i) err[i] = (float)reproj2(H, px[i], py[i], qx[i], qy[i]);
    tmp = err;
    std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
    if (tmp[n / 2] < best_med) { best_med = tmp[n / 2]; std::memcpy(best, H, sizeof best); }
  }
  const double sigma = 2.5 * 1.4826 * (1 + 5.0 / (n - 4)) * std::sqrt(best_med), thr = std::max(sigma * sigma, 1e-6);
  std::vector<int> inl;
  for (int i = 0; i < n; ++i) if (reproj2(best, px[i], py[i], qx[i], qy[i]) <= thr) inl.push_back(i);
  if (inl.size() < 4) return (int)inl.size();
  double M[9][9] = {{0}};
  for (int i : inl) {
    const double r0[9] = {px[i], py[i], 1, 0, 0, 0, -qx[i] * px[i], -qx[i] * py[i], -qx[i]};
    const double r1[9] = {0, 0, 0, px[i], py[i], 1, -qy[i] * px[i], -qy[i] * py[i], -qy[i]};
    for (int a = 0; a < 9; ++a) for (int b2 = a; b2 < 9; ++b2) M[a][b2] += r0[a] * r0[b2] + r1[a] * r1[b2];
  }
  for (int a = 0; a < 9; ++a) for (int b2 = 0; b2 < a; ++b2) M[a][b2] = M[b2][a];
  double H[9];
  smallest_eigenvector9(M, H);
  if (std::fabs(H[8]) > 1e-12) for (int k = 0; k < 9; ++k) H[k] /= H[8];
  for (int round = 0; round < 10; ++round) {   /* Levenberg-Marquardt on h0..h7 (h8 = 1) */
    double JtJ[8][9] = {{0}};
    for (int i : inl) {
      const double x = px[i], y = py[i], w = 1.0 / (H[6] * x + H[7] * y + 1), a = (H[0] * x + H[1] * y + H[2]) * w, b2 = (H[3] * x + H[4] * y + H[5]) * w;
      const double ju[8] = {x * w, y * w, w, 0, 0, 0, -a * x * w, -a * y * w}, jv[8] = {0, 0, 0, x * w, y * w, w, -b2 * x * w, -b2 * y * w};
      const double eu = a - qx[i], ev = b2 - qy[i];
      for (int r = 0; r < 8; ++r) { for (int cc = 0; cc < 8; ++cc) JtJ[r][cc] += ju[r] * ju[cc] + jv[r] * jv[cc]; JtJ[r][8] -= ju[r] * eu + jv[r] * ev; }
    }
    for (int r = 0; r < 8; ++r) JtJ[r][r] *= 1.001;
    bool ok = true;
    for (int c = 0; c < 8 && ok; ++c) {
      int p = c;
      for (int r = c + 1; r < 8; ++r) if (std::fabs(JtJ[r][c]) > std::fabs(JtJ[p][c])) p = r;
      if (std::fabs(JtJ[p][c]) < 1e-300) { ok = false; break; }
      if (p != c) for (int k = 0; k < 9; ++k) std::swap(JtJ[p][k], JtJ[c][k]);
      for (int r = 0; r < 8; ++r) if (r != c) { const double f = JtJ[r][c] / JtJ[c][c]; for (int k = c; k < 9; ++k) JtJ[r][k] -= f * JtJ[c][k]; }
    }
    if (!ok) break;
    for (int k = 0; k < 8; ++k) H[k] += JtJ[k][8] / JtJ[k][k];
  }
  volatile double sink = H[0];
  (void)sink;
  return (int)inl.size();
}
}  // namespace
/* the descriptor test pairs as run-time data (vslam_set_brief_pattern / vslam_set_orb_pattern of the C ABI; `device` unused) */
ORC_API int orc_set_brief_pattern(int, const int8_t* pairs) {
  if (!pairs) return VSLAM_ERR_INVALID;
  for (int i = 0; i < 1024; ++i) if (pairs[i] < -24 || pairs[i] > 24) return VSLAM_ERR_INVALID;
  std::memcpy(kBriefPattern, pairs, 1024);
  return VSLAM_OK;
}
ORC_API int orc_set_orb_pattern(int, const int8_t* pairs) {
  if (!pairs) return VSLAM_ERR_INVALID;
  for (int i = 0; i < 1024; ++i) if (pairs[i] < -15 || pairs[i] > 15) return VSLAM_ERR_INVALID;     /* a 31 x 31 patch, as the HIP library checks it */
  std::memcpy(kOrbPattern, pairs, 1024);
  return VSLAM_OK;
}
ORC_API int orc_get_brief_pattern(int, int8_t* out) { if (!out) return VSLAM_ERR_INVALID; std::memcpy(out, kBriefPattern, 1024); return VSLAM_OK; }
ORC_API int orc_get_orb_pattern(int, int8_t* out) { if (!out) return VSLAM_ERR_INVALID; std::memcpy(out, kOrbPattern, 1024); return VSLAM_OK; }
ORC_API int orc_dead_knn_match(orc_ctx* c, int s, int norm, int homography) {
  if (!c || s < 0 || s >= (int)c->streams.size()) return VSLAM_ERR_INVALID;
  const Stream& st = c->streams[s];
  const int nq = (int)st.kpL.size(), nt = (int)st.kpR.size();
  std::vector<uint8_t> dq((size_t)std::max(nq, 1) * 32), dt((size_t)std::max(nt, 1) * 32);
  for (int i = 0; i < nq; ++i) std::memcpy(&dq[(size_t)32 * i], st.kpL[i].desc, 32);
  for (int i = 0; i < nt; ++i) std::memcpy(&dt[(size_t)32 * i], st.kpR[i].desc, 32);
  std::vector<int32_t> idx((size_t)std::max(nq, 1) * 2);
  std::vector<float> dist((size_t)std::max(nq, 1) * 2);
  const int rc = orc_knn2(c, norm, nq, dq.data(), nt, dt.data(), idx.data(), dist.data());
  if (rc != VSLAM_OK) return rc;
  if (homography && nt > 0) {
    std::vector<double> px, py, qx, qy;
    for (int i = 0; i < nq; ++i) {
      const int j = idx[2 * (size_t)i];
      if (j < 0 || j >= nt) continue;
      px.push_back(st.kpL[i].col); py.push_back(st.kpL[i].row); qx.push_back(st.kpR[j].col); qy.push_back(st.kpR[j].row);
    }
    (void)dead_find_homography_lmeds(px, py, qx, qy);
  }
  return nq;
}
ORC_API int orc_align_points(orc_ctx* c, int32_t n, const double* moving, const double* fixed, const double* omega,
                             const double* weight, const double T_init[12], double T_out[12], double* chi, uint8_t* inlier,
                             int32_t* n_inliers, double* total_error, int32_t* iterations, double H_out[36]) {
  if (!c || c->streams.empty() || n < 0) return VSLAM_ERR_INVALID;
  AlignerIO io;
  io.n = n;
  io.moving.assign(moving, moving + 3 * n); io.fixed.assign(fixed, fixed + 4 * n);
  io.omega.assign(omega, omega + n); io.weight.assign(weight, weight + n);
  std::memcpy(io.T.m, T_init, sizeof(double) * 12);
  AlignerParams P = c->streams[0].aligner_params();
  aligner_converge(P, io);
  if (T_out) std::memcpy(T_out, io.T.m, sizeof(double) * 12);
  for (int i = 0; i < n; ++i) { if (chi) chi[i] = io.errors[i]; if (inlier) inlier[i] = io.inliers[i]; }
  if (n_inliers) *n_inliers = io.n_inliers;
  if (total_error) *total_error = io.total_error;
  if (iterations) *iterations = io.iterations;
  if (H_out) std::memcpy(H_out, io.H, sizeof(double) * 36);
  return VSLAM_OK;
}
/* UVDAligner on caller-provided correspondences (uvd_aligner.cpp): fixed n*3 = (u, v, depth) */
ORC_API int orc_align_points_uvd(orc_ctx* c, int32_t n, const double* moving, const double* fixed_uvd, const double* omega_uv,
                                 const double* omega_depth, const double* weight, const double T_init[12], double T_out[12],
                                 double* chi, uint8_t* inlier, int32_t* n_inliers, double* total_error, int32_t* iterations,
                                 double H_out[36]) {
  if (!c || c->streams.empty() || n < 0) return VSLAM_ERR_INVALID;
  AlignerIO io;
  io.uvd = true;
  io.n = n;
  io.moving.assign(moving, moving + 3 * n);
  io.fixed.resize((size_t)4 * n);
  for (int i = 0; i < n; ++i) { for (int k = 0; k < 3; ++k) io.fixed[4 * i + k] = fixed_uvd[3 * i + k]; io.fixed[4 * i + 3] = omega_depth[i]; }
  io.omega.assign(omega_uv, omega_uv + n); io.weight.assign(weight, weight + n);
  std::memcpy(io.T.m, T_init, sizeof(double) * 12);
  AlignerParams P = c->streams[0].aligner_params();
  aligner_converge(P, io);
  if (T_out) std::memcpy(T_out, io.T.m, sizeof(double) * 12);
  for (int i = 0; i < n; ++i) { if (chi) chi[i] = io.errors[i]; if (inlier) inlier[i] = io.inliers[i]; }
  if (n_inliers) *n_inliers = io.n_inliers;
  if (total_error) *total_error = io.total_error;
  if (iterations) *iterations = io.iterations;
  if (H_out) std::memcpy(H_out, io.H, sizeof(double) * 36);
  return VSLAM_OK;
}
/* one linearize() call: H (36), b (6), total error, inliers — for the golden first-iteration check */
ORC_API int orc_align_linearize(orc_ctx* c, int32_t n, const double* moving, const double* fixed, const double* omega,
                                const double* weight, const double T[12], int ignore_outliers, double H[36], double b[6],
                                double* total_error, int32_t* n_inliers, double* chi, uint8_t* inlier) {
  if (!c || c->streams.empty()) return VSLAM_ERR_INVALID;
  AlignerIO io;
  io.n = n;
  io.moving.assign(moving, moving + 3 * n); io.fixed.assign(fixed, fixed + 4 * n);
  io.omega.assign(omega, omega + n); io.weight.assign(weight, weight + n);
  io.errors.assign(n, -1); io.inliers.assign(n, 0);
  std::memcpy(io.T.m, T, sizeof(double) * 12);
  AlignerParams P = c->streams[0].aligner_params();
  aligner_linearize(P, io, ignore_outliers != 0, H, b);
  *total_error = io.total_error; *n_inliers = io.n_inliers;
  for (int i = 0; i < n; ++i) { if (chi) chi[i] = io.errors[i]; if (inlier) inlier[i] = io.inliers[i]; }
  return VSLAM_OK;
}
/* threshold controller on scripted counts (base_framepoint_generator.cpp:382-415,440-459), 1 region */
ORC_API int orc_controller_run(const vslam_config* cfg, int32_t n_frames, const int32_t* counts_left,
                               const int32_t* counts_right, int32_t target_per_detector, int32_t* thresholds_out) {
  int thr = cfg->detector_threshold_minimum;
  for (int f = 0; f < n_frames; ++f) {
    real acc = 0;
    const int32_t counts[2] = {counts_left[f], counts_right[f]};
    for (int s = 0; s < 2; ++s) {
      real t = thr;
      const real delta = ((real)counts[s] - (real)target_per_detector) / (real)target_per_detector;
      if (delta < -cfg->target_number_of_keypoints_tolerance) {
        t = t + std::min(std::max(delta, -cfg->detector_threshold_maximum_change) * t, -1.0);
        if (t < cfg->detector_threshold_minimum) t = cfg->detector_threshold_minimum;
      } else if (delta > cfg->target_number_of_keypoints_tolerance) {
        t += std::max(std::min(delta, cfg->detector_threshold_maximum_change) * t, 1.0);
        if (t > cfg->detector_threshold_maximum) t = cfg->detector_threshold_maximum;
      }
      acc += t;
    }
    thr = (int)std::rint(acc / 2);
    thresholds_out[f] = thr;
  }
  return VSLAM_OK;
}
/* stereo sweep + binning on caller-provided features (compute(), :135-462), no tracked points.
 * featL/featR: n*2 int32 (row,col); returns matches as (iL_id, iR_id, dist, epi) in output order */
ORC_API int orc_stereo_match(const vslam_config* cfg, double tau_tri, int32_t nL, const int32_t* rcL, const uint8_t* dL,
                             int32_t nR, const int32_t* rcR, const uint8_t* dR, int32_t cap, int32_t* n_out, int32_t* out4) {
  Stream s;
  s.configure(*cfg);
  s.tau_tri = tau_tri;
  std::vector<Feature> fl(nL), fr(nR);
  for (int i = 0; i < nL; ++i) { fl[i].row = rcL[2 * i]; fl[i].col = rcL[2 * i + 1]; fl[i].score = 0; std::memcpy(fl[i].desc, dL + 32 * i, 32); }
  for (int i = 0; i < nR; ++i) { fr[i].row = rcR[2 * i]; fr[i].col = rcR[2 * i + 1]; fr[i].score = 0; std::memcpy(fr[i].desc, dR + 32 * i, 32); }
  s.storeL.set_features(fl);
  s.storeR.set_features(fr);
  FrameRec cur;
  s.compute(cur);
  *n_out = (int32_t)cur.points.size();
  if ((int32_t)cur.points.size() > cap) return VSLAM_ERR_CAPACITY;
  for (size_t i = 0; i < cur.points.size(); ++i) {
    const Point& p = cur.points[i];
    /* recover ids by coordinates (unique pixels in the fixtures) */
    int il = -1, ir = -1;
    for (int k = 0; k < nL; ++k) if (fl[k].row == p.yL && fl[k].col == p.xL) il = k;
    for (int k = 0; k < nR; ++k) if (fr[k].row == p.yR && fr[k].col == p.xR) ir = k;
    out4[4 * i + 0] = il; out4[4 * i + 1] = ir; out4[4 * i + 2] = p.dist; out4[4 * i + 3] = p.epi;
  }
  return VSLAM_OK;
}

/* StereoFramePointGenerator::track (:464-681) on caller-provided data: previous points (left-camera coordinates,
 * both descriptors, epipolar offset), the motion prior T, window d, both feature sets.  out4 = (previous index, left
 * feature, right feature, L-R distance) per tracked point in order; lost = indices of the lost-eligible points. */
ORC_API int orc_track_match(const vslam_config* cfg, const double T[12], int32_t d, double tau_track, double tau_tri, int32_t by_appearance,
                            int32_t nP, const double* cam, const uint8_t* pdL, const uint8_t* pdR, const int32_t* epi,
                            int32_t nL, const int32_t* rcL, const uint8_t* dL, int32_t nR, const int32_t* rcR, const uint8_t* dR,
                            int32_t* n_tracked, int32_t* out4, int32_t* n_lost, int32_t* lost) {
  Stream s;
  s.configure(*cfg);
  s.tau_tri = tau_tri; s.gen_tau_track = tau_track; s.win = d;
  std::vector<Feature> fl(nL), fr(nR);
  for (int i = 0; i < nL; ++i) { fl[i].row = rcL[2 * i]; fl[i].col = rcL[2 * i + 1]; fl[i].score = 0; std::memcpy(fl[i].desc, dL + 32 * i, 32); }
  for (int i = 0; i < nR; ++i) { fr[i].row = rcR[2 * i]; fr[i].col = rcR[2 * i + 1]; fr[i].score = 0; std::memcpy(fr[i].desc, dR + 32 * i, 32); }
  s.storeL.set_features(fl);
  s.storeR.set_features(fr);
  FrameRec prev, cur;
  prev.points.resize(nP);
  for (int i = 0; i < nP; ++i) {
    Point& p = prev.points[i];
    std::memset(&p, 0, sizeof p);
    for (int k = 0; k < 3; ++k) p.cam[k] = cam[3 * i + k];
    std::memcpy(p.dL, pdL + 32 * i, 32); std::memcpy(p.dR, pdR + 32 * i, 32);
    p.epi = epi[i]; p.prev = -1; p.lm = -1; p.has_next = false;
  }
  Tf Tt;
  std::memcpy(&Tt, T, sizeof(double) * 12);
  s.track(cur, prev, Tt, by_appearance != 0);
  *n_tracked = (int32_t)cur.points.size();
  for (size_t i = 0; i < cur.points.size(); ++i) {
    const Point& p = cur.points[i];
    int il = -1, ir = -1;   /* ids by coordinates (unique pixels in the fixtures) */
    for (int k = 0; k < nL; ++k) if (fl[k].row == p.yL && fl[k].col == p.xL) il = k;
    for (int k = 0; k < nR; ++k) if (fr[k].row == p.yR && fr[k].col == p.xR) ir = k;
    out4[4 * i + 0] = p.prev; out4[4 * i + 1] = il; out4[4 * i + 2] = ir; out4[4 * i + 3] = p.dist;
  }
  *n_lost = (int32_t)s.lost.size();
  for (size_t i = 0; i < s.lost.size(); ++i) lost[i] = s.lost[i];
  return VSLAM_OK;
}

/* StereoFramePointGenerator::recoverPoints (:683-869) on caller-provided data: the lost points' landmarks (world), their last
 * descriptors, the frame's world_to_camera_left and both images.  Test infrastructure for the independent fixture
 * (tests/golden/stereo_recover.npz); the HIP path is compared with this code frame by frame in the pipeline tests. */
ORC_API int orc_stereo_recover(const vslam_config* cfg, const uint8_t* imgL, const uint8_t* imgR, int32_t stride, const double w2c[12], int32_t n,
                               const uint8_t* has_lm, const double* lmw, const uint8_t* pdL, const uint8_t* pdR, double tau_track, double tau_tri,
                               int32_t* n_rec, int32_t* rec_index, int32_t* rec_xy4, int32_t* rec_dist, uint8_t* rec_desc, double* rec_xyz) {
  Stream s;
  s.configure(*cfg);
  integral_image(imgL, cfg->rows, cfg->cols, stride, s.sumL);
  integral_image(imgR, cfg->rows, cfg->cols, stride, s.sumR);
  if (cfg->descriptor_type == VSLAM_DESCRIPTOR_ORB) {
    gaussian_blur7_u8(imgL, cfg->rows, cfg->cols, stride, s.blurL);
    gaussian_blur7_u8(imgR, cfg->rows, cfg->cols, stride, s.blurR);
  }
  s.gen_tau_track = tau_track; s.tau_tri = tau_tri;
  FrameRec prev, cur;
  std::memcpy(cur.world_to_cam.m, w2c, sizeof(double) * 12);
  cur.cam_to_world = tf_inverse(cur.world_to_cam);
  prev.points.resize(n);
  for (int i = 0; i < n; ++i) {
    Point& p = prev.points[i];
    std::memset(&p, 0, sizeof p);
    std::memcpy(p.dL, pdL + 32 * i, 32); std::memcpy(p.dR, pdR + 32 * i, 32);
    p.lm = -1; p.prev = -1;
    if (has_lm[i]) {
      Landmark L;
      for (int k = 0; k < 3; ++k) L.w[k] = lmw[3 * i + k];
      L.updates = 1;
      s.landmarks.push_back(L);
      p.lm = (int)s.landmarks.size() - 1;
    }
    s.lost.push_back(i);
  }
  *n_rec = s.recover(cur, prev);
  for (size_t k = 0; k < cur.points.size(); ++k) {
    const Point& q = cur.points[k];
    rec_index[k] = q.prev;
    rec_xy4[4 * k] = q.xL; rec_xy4[4 * k + 1] = q.yL; rec_xy4[4 * k + 2] = q.xR; rec_xy4[4 * k + 3] = q.yR;
    rec_dist[k] = q.dist;
    std::memcpy(rec_desc + 64 * k, q.dL, 32); std::memcpy(rec_desc + 64 * k + 32, q.dR, 32);
    for (int j = 0; j < 3; ++j) rec_xyz[3 * k + j] = q.cam[j];
  }
  return VSLAM_OK;
}

/* PoseTracker3D control arithmetic on scripted inputs (test infrastructure for tests/golden/tracker.npz): the search
 * adaptation of _track (:240-288) over a sequence of (previous points, tracked points, tracked landmarks, by_appearance), and the
 * selection rule of _prunePoints (:439-466). */
ORC_API int orc_track_adapt(const vslam_config* cfg, int32_t n, const int32_t* n_prev, const int32_t* n_tracked, const int32_t* n_landmarks,
                            const int32_t* by_appearance, int32_t win0, double tau0, int32_t* win_out, double* tau_out) {
  Stream s;
  s.configure(*cfg);
  s.win = win0; s.tau_track = tau0;
  for (int i = 0; i < n; ++i) {
    if (by_appearance[i]) s.win = cfg->maximum_projection_tracking_distance_pixels;   /* :228-230 */
    s.n_tracked_points = (uint32_t)n_tracked[i]; s.n_tracked_landmarks = (uint32_t)n_landmarks[i];
    s.adapt_search((size_t)n_prev[i]);
    win_out[i] = s.win; tau_out[i] = s.tau_track;
  }
  return VSLAM_OK;
}
/* The translation weights over a sequence of StereoUVAligner::initialize calls on ONE aligner object
 * (stereouv_aligner.cpp:22,57-61): call k has n[k] measurements with depths depth[off_k ..] and the
 * enable_inverse_depth_as_information flag inverse_depth[k]; out receives the n[k] weights after each call. */
ORC_API int orc_aligner_weights(const vslam_config* cfg, int32_t n_calls, const int32_t* n, const int32_t* inverse_depth,
                                const double* depth, double* out) {
  Stream s;
  s.configure(*cfg);
  size_t off = 0;
  for (int k = 0; k < n_calls; ++k) {
    s.aligner_resize_weights(n[k]);
    if (inverse_depth[k]) for (int u = 0; u < n[k]; ++u) s.aligner_set_weight(u, depth[off + u]);
    for (int u = 0; u < n[k]; ++u) out[off + u] = s.al.weight[u];
    off += (size_t)n[k];
  }
  return VSLAM_OK;
}
ORC_API int orc_prune_select(const vslam_config* cfg, int32_t n, double total_error, const double* errors, const uint8_t* inliers, uint8_t* keep) {
  Stream s;
  s.configure(*cfg);
  const real avg = total_error / (real)n;   /* averageError() = E / M (base_aligner.h) */
  for (int i = 0; i < n; ++i) keep[i] = s.prune_keeps(avg, errors[i], inliers[i]) ? 1 : 0;
  return VSLAM_OK;
}

/* Landmark::update (landmark.cpp:66-167) on caller-provided measurement lists (last measurement of a list = the new one) */
ORC_API int orc_landmark_update(const vslam_config* cfg, int32_t n, const int32_t* offsets, const int32_t* frame_of, int32_t n_frames,
                                const double* w2c, const double* c2w, const double* cam, double* world, int32_t* updates) {
  Stream s;
  s.configure(*cfg);
  s.frames.resize(n_frames);
  for (int f = 0; f < n_frames; ++f) {
    std::memcpy(s.frames[f].world_to_cam.m, w2c + 12 * f, sizeof(double) * 12);
    std::memcpy(s.frames[f].cam_to_world.m, c2w + 12 * f, sizeof(double) * 12);
  }
  for (int i = 0; i < n; ++i) {
    const int a = offsets[i], b = offsets[i + 1];
    if (b <= a) continue;
    Landmark lm;
    for (int k = 0; k < 3; ++k) lm.w[k] = world[3 * i + k];
    lm.updates = (uint32_t)updates[i];
    for (int m = a; m < b - 1; ++m) {
      Meas q;
      q.frame = frame_of[m];
      for (int k = 0; k < 3; ++k) q.cam[k] = cam[3 * m + k];
      q.inv_depth = 1 / cam[3 * m + 2];
      lm.meas.push_back(q);
    }
    Point p;
    std::memset(&p, 0, sizeof p);
    for (int k = 0; k < 3; ++k) p.cam[k] = cam[3 * (b - 1) + k];
    s.update_landmark(lm, frame_of[b - 1], p);
    for (int k = 0; k < 3; ++k) world[3 * i + k] = lm.w[k];
    updates[i] = (int32_t)lm.updates;
  }
  return VSLAM_OK;
}

/* ---- RGB-D components (DepthFramePointGenerator) ------------------------------------------ */
/* _computeDepthMap (depth_framepoint_generator.cpp:410-485), bilateral filter off (configuration_{icl,tum,xtion}.yaml) */
ORC_API int orc_depth_space_map(const vslam_depth_params* p, const uint16_t* depth, int32_t stride, float* space,
                                int16_t* row_map, int16_t* col_map) {
  const int rows = p->rows, cols = p->cols;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {                                   /* :428-432 */
      float* d = space + ((size_t)r * cols + c) * 3;
      d[0] = 0; d[1] = 0; d[2] = (float)p->maximum_depth_meters;
      if (row_map) row_map[(size_t)r * cols + c] = -1;                 /* :434-438 */
      if (col_map) col_map[(size_t)r * cols + c] = -1;
    }
  Tf r2l;
  std::memcpy(r2l.m, p->right_to_left, sizeof r2l.m);
  for (int r = 0; r < rows; ++r) {
    const uint16_t* raw = depth + (size_t)r * stride;
    for (int c = 0; c < cols; ++c) {
      if (!raw[c]) continue;                                            /* :449 */
      const real dm = raw[c] * p->depth_scale_factor_intensity_to_meters;   /* :452 */
      const real ph[3] = {c * dm, r * dm, dm};
      real pr[3], pl[3], px[3];
      mat3_mul_vec(p->K_right_inverse, ph, pr);                        /* :454 */
      tf_apply(r2l, pr, pl);                                            /* :456 */
      const real zl = pl[2];
      if (zl <= 0) continue;                                            /* :458-460 */
      mat3_mul_vec(p->K_left, pl, px);                                  /* :462 */
      const real u = px[0] / px[2], v = px[1] / px[2];                  /* :463 */
      const int32_t dr = (int32_t)std::round(v), dc = (int32_t)std::round(u);   /* :466-467 */
      if (dr < 0 || dr >= rows || dc < 0 || dc >= cols) continue;       /* :470-474 */
      float* d = space + ((size_t)dr * cols + dc) * 3;
      if (d[2] > zl) {                                                  /* :479: stored float against the new double */
        d[0] = (float)pl[0]; d[1] = (float)pl[1]; d[2] = (float)pl[2];
        if (row_map) row_map[(size_t)dr * cols + dc] = (int16_t)r;
        if (col_map) col_map[(size_t)dr * cols + dc] = (int16_t)c;
      }
    }
  }
  return VSLAM_OK;
}

/* compute (:45-164) on caller-provided features; bins: -1 empty, -2 owned by a tracked point, >= 0 new feature */
ORC_API int orc_depth_compute(const vslam_depth_params* p, const float* space, int32_t nF, const int32_t* rcF, int32_t nT,
                              const int32_t* rcT, int32_t cap, int32_t* n_new, int32_t* new_feat, double* new_xyz,
                              int32_t* n_temp, int32_t* temp_feat, double* temp_xyz) {
  const int rows = p->rows, cols = p->cols, bin = p->bin_size_pixels;
  const int rows_bin = p->enable_keypoint_binning ? rows / bin + 1 : 0, cols_bin = p->enable_keypoint_binning ? cols / bin + 1 : 0;   /* base_framepoint_generator.cpp:304-305 */
  std::vector<int32_t> bins((size_t)(rows_bin + 1) * (cols_bin + 1), -1);   /* +1: rint() can reach the grid size */
  auto bin_of = [&](int row, int col) -> int32_t& {
    const int rb = (int)std::rint((real)row / bin), cb = (int)std::rint((real)col / bin);   /* :59-60, :113-114 */
    return bins[(size_t)rb * (cols_bin + 1) + cb];
  };
  if (p->enable_keypoint_binning) for (int i = 0; i < nT; ++i) bin_of(rcT[2 * i], rcT[2 * i + 1]) = -2;   /* :57-63 */
  std::vector<int32_t> fresh;        /* framepoints_new, feature order */
  std::vector<real> depth_of(nF, 0);
  int nt = 0;
  for (int i = 0; i < nF; ++i) {
    const int row = rcF[2 * i], col = rcF[2 * i + 1];
    const float* d = space + ((size_t)row * cols + col) * 3;
    if (d[2] < p->minimum_depth_meters) continue;                                            /* :80 */
    if (d[2] >= p->maximum_depth_meters && p->enable_point_triangulation) {                  /* :89-101 */
      const real m = p->maximum_depth_meters;
      const real ph[3] = {col * m, row * m, m};
      real x[3];
      mat3_mul_vec(p->K_left_inverse, ph, x);
      if (nt < cap) { temp_feat[nt] = i; for (int k = 0; k < 3; ++k) temp_xyz[3 * nt + k] = x[k]; }
      ++nt;
      continue;
    }
    fresh.push_back(i);                                                                      /* :104-108 */
    depth_of[i] = (real)d[2];
    if (p->enable_keypoint_binning) {                                                        /* :111-130 */
      int32_t& b = bin_of(row, col);
      if (b != -1) { if (b >= 0 && depth_of[i] < depth_of[b]) b = i; }
      else b = i;
    }
  }
  std::vector<int32_t> out;
  if (p->enable_keypoint_binning) {                                                          /* :141-157 */
    for (int rb = 0; rb < rows_bin; ++rb)
      for (int cb = 0; cb < cols_bin; ++cb) { const int32_t b = bins[(size_t)rb * (cols_bin + 1) + cb]; if (b >= 0) out.push_back(b); }
  } else {
    out = fresh;                                                                             /* :160-162 */
  }
  *n_new = (int32_t)out.size(); *n_temp = nt;
  if ((int32_t)out.size() > cap || nt > cap) return VSLAM_ERR_CAPACITY;
  for (size_t k = 0; k < out.size(); ++k) {
    const int i = out[k];
    const float* d = space + ((size_t)rcF[2 * i] * cols + rcF[2 * i + 1]) * 3;
    new_feat[k] = i;
    for (int q = 0; q < 3; ++q) new_xyz[3 * k + q] = (real)d[q];
  }
  return VSLAM_OK;
}

/* DepthFramePointGenerator::track (:166-287) on caller-provided data.  Previous points = points + temporary points of the
 * previous frame, in that order (:181-184); flags bit0 = has a landmark, bit1 = hasUnreliableDepth.  out2 = (previous index,
 * left feature) per tracked point with its measured coordinates; temp2 = the same for the points whose pixel has no depth
 * (:231-238, triangulation enabled); lost as in :264-268. */
ORC_API int orc_depth_track(const vslam_depth_params* p, const float* space, const double T[12], int32_t d, double tau, int32_t by_appearance,
                            int32_t nP, const double* cam, const uint8_t* pdesc, const uint8_t* pflags, int32_t nL, const int32_t* rcL,
                            const uint8_t* dL, int32_t* n_tracked, int32_t* out2, double* xyz, int32_t* n_temp, int32_t* temp2,
                            int32_t* n_lost, int32_t* lost, int32_t* n_tracked_landmarks) {
  const int rows = p->rows, cols = p->cols;
  FeatureStore store;
  store.configure(rows, cols);
  std::vector<Feature> fl(nL);
  for (int i = 0; i < nL; ++i) { fl[i].row = rcL[2 * i]; fl[i].col = rcL[2 * i + 1]; fl[i].score = 0; std::memcpy(fl[i].desc, dL + 32 * i, 32); }
  store.set_features(fl);
  Tf Tt;
  std::memcpy(Tt.m, T, sizeof Tt.m);
  int nt = 0, ntmp = 0, nl = 0, nlm = 0;
  for (int i = 0; i < nP; ++i) {
    real q[3], uvw[3];
    tf_apply(Tt, cam + 3 * i, q);                                       /* :197 */
    mat3_mul_vec(p->K_left, q, uvw);                                    /* :200 */
    if (!(uvw[2] > 0)) continue;                                        /* quirk B.5 (as in the stereo track): the reference divides regardless */
    const real uc = uvw[0] / uvw[2], ur = uvw[1] / uvw[2];
    if (!(uc > -2147483648.0 && uc < 2147483648.0 && ur > -2147483648.0 && ur < 2147483648.0)) continue;
    const int32_t col = (int32_t)uc, row = (int32_t)ur;                 /* :201-202 */
    if (col < 0 || col > cols || row < 0 || row > rows) continue;       /* :205-208 */
    const int r0 = std::max(row - d, 0), r1 = std::min(row + d + 1, rows);   /* :214-217 */
    const int c0 = std::max(col - d, 0), c1 = std::min(col + d + 1, cols);
    real dist_best;
    const int f = store.match_in_region(row, col, pdesc + 32 * i, r0, r1, c0, c1, tau, by_appearance != 0, dist_best);   /* :220-229 */
    bool has_next = false;
    if (f >= 0) {
      const float* dp = space + ((size_t)fl[f].row * cols + fl[f].col) * 3;    /* :235 */
      if (dp[2] < p->minimum_depth_meters) continue;                           /* :238-240 */
      store.lattice[(size_t)fl[f].row * cols + fl[f].col] = -1;                /* :243-244 */
      if (dp[2] >= p->maximum_depth_meters && p->enable_point_triangulation) { /* :247-256 */
        temp2[2 * ntmp] = i; temp2[2 * ntmp + 1] = f; ++ntmp;
        continue;
      }
      out2[2 * nt] = i; out2[2 * nt + 1] = f;                                  /* :259-273 */
      for (int k = 0; k < 3; ++k) xyz[3 * nt + k] = (real)dp[k];
      ++nt;
      has_next = true;
      if (pflags[i] & 1) ++nlm;                                                /* :275-277 */
    }
    if (!has_next && !(pflags[i] & 2)) lost[nl++] = i;                         /* :281-284 */
  }
  *n_tracked = nt; *n_temp = ntmp; *n_lost = nl; *n_tracked_landmarks = nlm;
  return VSLAM_OK;
}

/* DepthFramePointGenerator::recoverPoints (:289-407) on caller-provided data */
ORC_API int orc_depth_recover(const vslam_depth_params* p, const float* space, const uint8_t* img, int32_t stride, const double w2c[12],
                              int32_t n, const uint8_t* has_lm, const double* lm, const uint8_t* pdesc, float kp_size, double tau,
                              int32_t* n_rec, int32_t* rec_index, float* rec_xy, uint8_t* rec_desc, double* rec_xyz) {
  const int rows = p->rows, cols = p->cols;
  const bool orb = p->descriptor_type == VSLAM_DESCRIPTOR_ORB;   /* the configured _descriptor_extractor (:369-373) */
  std::vector<int32_t> sum;
  std::vector<uint8_t> blur;
  float oa = 1, ob = 0;
  if (orb) { gaussian_blur7_u8(img, rows, cols, stride, blur); orb_rotation(-1.f, &oa, &ob); }   /* the lost point's FAST keypoint: angle -1 */
  else integral_image(img, rows, cols, stride, sum);
  Tf W;
  std::memcpy(W.m, w2c, sizeof W.m);
  int nr = 0;
  for (int i = 0; i < n; ++i) {
    if (!has_lm[i]) continue;                                                     /* :305-307 */
    real pc[3], pi[3];
    tf_apply(W, lm + 3 * i, pc);                                                  /* :317 */
    mat3_mul_vec(p->K_left, pc, pi);                                              /* :325 */
    const real x = pi[0] / pi[2], y = pi[1] / pi[2];                              /* :326 */
    if (!(x >= 0 && x <= cols && y >= 0 && y <= rows)) continue;                  /* :329-332 (NaN never passes) */
    const float px = (float)x, py = (float)y;                                     /* :335 */
    const float fr = std::rint(py), fc = std::rint(px);                           /* :338 */
    if (!(fr >= 0 && fr < rows && fc >= 0 && fc < cols)) continue;                /* out-of-bounds read upstream */
    const float* dp = space + ((size_t)(int)fr * cols + (int)fc) * 3;
    if (dp[2] < p->minimum_depth_meters || dp[2] >= p->maximum_depth_meters) continue;   /* :341-344 */
    const float rbc = 5 * kp_size;                                                /* :347 */
    if (px <= rbc + 1 || px >= cols - rbc - 1 || py <= rbc + 1 || py >= rows - rbc - 1) continue;   /* :352-359 */
    const float cxf = px - rbc, cyf = py - rbc;                                   /* :362 */
    const int ox = (int)std::lrint(cxf), oy = (int)std::lrint(cyf);               /* cv::Rect_<float> -> cv::Rect: saturate_cast = cvRound */
    const int bx = ox + (int)(rbc + 0.5f), by = oy + (int)(rbc + 0.5f);           /* BRIEF rounds the keypoint (rbc, rbc) of the ROI */
    if (!(orb ? orb_inside(rows, cols, bx, by) : brief_inside(rows, cols, bx, by))) continue;   /* :376-378 (cannot happen behind the border gate) */
    uint8_t d[32];
    if (orb) orb_at(blur.data(), cols, bx, by, oa, ob, d);                        /* ORB::compute rounds the keypoint of the region the same way */
    else brief_at(sum, cols, bx, by, d);                                          /* :369-373 */
    if (hamming32(pdesc + 32 * i, d) > tau) continue;                             /* :381-383 */
    rec_index[nr] = i;
    rec_xy[2 * nr] = rbc + cxf; rec_xy[2 * nr + 1] = rbc + cyf;                   /* :384 keypoint.pt += corner_left */
    std::memcpy(rec_desc + 32 * nr, d, 32);
    for (int k = 0; k < 3; ++k) rec_xyz[3 * nr + k] = (real)dp[k];                /* :390 */
    ++nr;
  }
  *n_rec = nr;
  return VSLAM_OK;
}

/* getPointInCamera (base_framepoint_generator.cpp:461-494).  JacobiSVD::solve of the 3x2 system restated as QR of the
 * two columns followed by the triangular solve, minimum-norm when the columns are parallel to rounding (Eigen's rank
 * rule: singular values <= 2 eps * largest count as zero; here sigma_min ~ r00 r11 / sigma_max against the Frobenius
 * bound of sigma_max). */
static void point_in_camera(const float xp[2], const float xc[2], const Tf& T, const real K[9], real out[3]) {
  const real a0 = (xp[0] - K[2]) / K[0], b0 = (xp[1] - K[5]) / K[4];     /* :470-473 */
  const real a1 = (xc[0] - K[2]) / K[0], b1 = (xc[1] - K[5]) / K[4];
  const real x0[3] = {a0, b0, 1}, x1[3] = {a1, b1, 1};
  real c0[3], c1[3] = {a1, b1, 1}, t[3];
  for (int i = 0; i < 3; ++i) { c0[i] = -((T.m[4 * i] * x0[0] + T.m[4 * i + 1] * x0[1]) + T.m[4 * i + 2] * x0[2]); t[i] = T.m[4 * i + 3]; }   /* :480-483 */
  const real r00 = std::sqrt((c0[0] * c0[0] + c0[1] * c0[1]) + c0[2] * c0[2]);
  real q0[3];
  for (int i = 0; i < 3; ++i) q0[i] = c0[i] / r00;
  const real r01 = (q0[0] * c1[0] + q0[1] * c1[1]) + q0[2] * c1[2];
  real v[3];
  for (int i = 0; i < 3; ++i) v[i] = c1[i] - r01 * q0[i];
  const real r11 = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  const real g0 = (q0[0] * t[0] + q0[1] * t[1]) + q0[2] * t[2];
  const real fro2 = (r00 * r00 + r01 * r01) + r11 * r11;
  real z0, z1;
  if (r00 * r11 > 2 * 2.220446049250313e-16 * fro2) {
    const real g1 = ((v[0] * t[0] + v[1] * t[1]) + v[2] * t[2]) / r11;
    z1 = g1 / r11;
    z0 = (g0 - r01 * z1) / r00;
  } else {   /* rank 1: minimum-norm solution of r00 z0 + r01 z1 = g0 */
    const real n2 = r00 * r00 + r01 * r01;
    z0 = g0 * r00 / n2; z1 = g0 * r01 / n2;
  }
  const real pp[3] = {x0[0] * z0, x0[1] * z0, x0[2] * z0}, pc[3] = {x1[0] * z1, x1[1] * z1, x1[2] * z1};   /* :489-490 */
  real moved[3];
  tf_apply(T, pp, moved);
  for (int i = 0; i < 3; ++i) out[i] = (pc[i] + moved[i]) / 2.0;        /* :493 */
}
ORC_API int orc_point_in_camera(int32_t n, const float* xy_prev, const float* xy_cur, const double T[12], const double K[9], double* out) {
  Tf Tt;
  std::memcpy(Tt.m, T, sizeof Tt.m);
  for (int i = 0; i < n; ++i) point_in_camera(xy_prev + 2 * i, xy_cur + 2 * i, Tt, K, out + 3 * i);
  return VSLAM_OK;
}

/* ---- OrbDetector components (base_framepoint_generator.cpp:52-70: cv::ORB as a detector) [recalled: OpenCV 3.x] ---- */
static inline int cv_round(double v) { return (int)std::lrint(v); }               /* cvRound: half to even */
static inline short sat_short(float v) { const int i = (int)std::lrint(v); return (short)std::min(std::max(i, -32768), 32767); }
/* imgproc resize.cpp, INTER_LINEAR, 8UC1: resizeGeneric_<HResizeLinear<uchar,int,short,2048>, VResizeLinear<uchar,int,short,FixedPtCast<22>>> */
static void resize_linear_u8(const uint8_t* src, int rows, int cols, int stride, uint8_t* dst, int drows, int dcols) {
  const double sx_scale = (double)cols / dcols, sy_scale = (double)rows / drows;
  std::vector<int> xofs(dcols), yofs(drows);
  std::vector<short> alpha(2 * dcols), beta(2 * drows);
  int xmax = dcols;
  for (int dx = 0; dx < dcols; ++dx) {
    float fx = (float)((dx + 0.5) * sx_scale - 0.5);
    int sx = (int)std::floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= cols - 1) { xmax = std::min(xmax, dx); fx = 0; sx = cols - 1; }
    xofs[dx] = sx;
    alpha[2 * dx] = sat_short((1.f - fx) * 2048); alpha[2 * dx + 1] = sat_short(fx * 2048);
  }
  for (int dy = 0; dy < drows; ++dy) {
    float fy = (float)((dy + 0.5) * sy_scale - 0.5);
    int sy = (int)std::floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = sat_short((1.f - fy) * 2048); beta[2 * dy + 1] = sat_short(fy * 2048);
  }
  std::vector<int> r0(dcols), r1(dcols);
  auto hrow = [&](int sy, std::vector<int>& D) {
    const uint8_t* S = src + (size_t)std::min(std::max(sy, 0), rows - 1) * stride;
    for (int dx = 0; dx < dcols; ++dx)
      D[dx] = dx < xmax ? S[xofs[dx]] * alpha[2 * dx] + S[xofs[dx] + 1] * alpha[2 * dx + 1] : S[xofs[dx]] * 2048;
  };
  for (int dy = 0; dy < drows; ++dy) {
    hrow(yofs[dy], r0); hrow(yofs[dy] + 1, r1);
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    for (int dx = 0; dx < dcols; ++dx)
      dst[(size_t)dy * dcols + dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}
ORC_API int orc_resize_linear_u8(const uint8_t* src, int32_t rows, int32_t cols, int32_t stride, uint8_t* dst, int32_t drows, int32_t dcols) {
  if (!src || !dst || rows < 2 || cols < 2 || drows < 1 || dcols < 1) return VSLAM_ERR_INVALID;
  resize_linear_u8(src, rows, cols, stride, dst, drows, dcols);
  return VSLAM_OK;
}
/* core mathfuncs fastAtan2 (float, degrees) */
static float fast_atan2f(float y, float x) {
  const float s = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) { c = ay / (ax + (float)2.220446049250313e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
  else { c = ax / (ay + (float)2.220446049250313e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}
static void orb_umax(int half, std::vector<int>& umax) {   /* orb.cpp computeKeyPoints: circular patch rows */
  umax.assign(half + 2, 0);
  const int vmax = (int)std::floor(half * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
  for (int v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)half * half - v * v));
  for (int v = half, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
}
static float harris_response(const uint8_t* img, int stride, int x0, int y0) {   /* orb.cpp HarrisResponses, blockSize 7, k 0.04 */
  const int r = 3;
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) {
      const uint8_t* q = img + (size_t)(y0 - r + i) * stride + (x0 - r + j);
      const int Ix = (q[1] - q[-1]) * 2 + (q[-stride + 1] - q[-stride - 1]) + (q[stride + 1] - q[stride - 1]);
      const int Iy = (q[stride] - q[-stride]) * 2 + (q[stride - 1] - q[-stride - 1]) + (q[stride + 1] - q[-stride + 1]);
      a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
  const float scale = 1.f / ((1 << 2) * 7 * 255.f), scale_sq_sq = scale * scale * scale * scale;
  return ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}
static float ic_angle(const uint8_t* img, int stride, int x0, int y0, int half, const std::vector<int>& umax) {   /* orb.cpp ICAngles */
  const uint8_t* center = img + (size_t)y0 * stride + x0;
  int m_01 = 0, m_10 = 0;
  for (int u = -half; u <= half; ++u) m_10 += u * center[u];
  for (int v = 1; v <= half; ++v) {
    int v_sum = 0;
    const int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      const int vp = center[u + v * stride], vm = center[u - v * stride];
      v_sum += vp - vm;
      m_10 += u * (vp + vm);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2f((float)m_01, (float)m_10);
}
ORC_API int orc_harris_angle(const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t n, const int16_t* xy, float* response, float* angle) {
  if (!img || n < 0 || (n && (!xy || !response || !angle))) return VSLAM_ERR_INVALID;
  std::vector<int> umax;
  orb_umax(15, umax);
  for (int i = 0; i < n; ++i) {
    const int x = xy[2 * i], y = xy[2 * i + 1];
    if (x < 16 || y < 16 || x >= cols - 16 || y >= rows - 16) return VSLAM_ERR_INVALID;
    response[i] = harris_response(img, stride, x, y);
    angle[i] = ic_angle(img, stride, x, y, 15, umax);
  }
  return VSLAM_OK;
}
/* KeyPointsFilter::retainBest: keeps every keypoint whose response ties the n-th; order = input order (nth_element's is unspecified) */
template <typename KP> static void retain_best(std::vector<KP>& k, int n) {
  if (n >= (int)k.size()) return;
  if (n == 0) { k.clear(); return; }
  std::vector<float> r(k.size());
  for (size_t i = 0; i < k.size(); ++i) r[i] = k[i].response;
  std::nth_element(r.begin(), r.begin() + (n - 1), r.end(), std::greater<float>());
  const float amb = r[n - 1];
  size_t m = 0;
  for (size_t i = 0; i < k.size(); ++i) if (k[i].response >= amb) k[m++] = k[i];
  k.resize(m);
}
struct OrbKp { float x, y, response, angle; };
ORC_API int orc_orb_detect(const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t nfeatures, float scale_factor, int32_t nlevels,
                           int32_t edge, int32_t patch, int32_t fast_threshold, int32_t cap, int32_t* n, float* out) {
  if (!img || !n || nlevels < 1 || nlevels > 16 || nfeatures < 0 || patch < 3 || patch > 63 || !(scale_factor > 1.f) || edge < patch / 2 + 1 || edge < 4 ||
      rows < 2 * edge + 8 || cols < 2 * edge + 8) return VSLAM_ERR_INVALID;
  /* features per level (orb.cpp computeKeyPoints) */
  std::vector<int> per(nlevels);
  {
    const float factor = (float)(1.0 / scale_factor);
    float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) { per[l] = cv_round(nd); sum += per[l]; nd *= factor; }
    per[nlevels - 1] = std::max(nfeatures - sum, 0);
  }
  const int half = patch / 2;
  std::vector<int> umax;
  orb_umax(half, umax);
  std::vector<uint8_t> prev(img, img + 0), cur;
  const uint8_t* lev = img;
  int lrows = rows, lcols = cols, lstride = stride;
  int total = 0;
  for (int l = 0; l < nlevels; ++l) {
    const float sc = (float)std::pow((double)scale_factor, (double)l);   /* getScale(level, 0, scaleFactor) */
    if (l > 0) {
      const int nr = cv_round(rows / sc), nc = cv_round(cols / sc);       /* Size sz(cvRound(cols/scale), cvRound(rows/scale)) */
      if (nr < 2 * edge + 8 || nc < 2 * edge + 8) break;                  /* nothing can survive the border filter */
      cur.assign((size_t)nr * nc, 0);
      resize_linear_u8(lev, lrows, lcols, lstride, cur.data(), nr, nc);   /* level l from level l-1 */
      prev.swap(cur);
      lev = prev.data(); lrows = nr; lcols = nc; lstride = nc;
    }
    std::vector<Keypoint> k;
    fast_detect_roi(lev, lstride, 0, 0, lcols, lrows, fast_threshold, k);   /* FastFeatureDetector::create(fastThreshold, true) */
    std::vector<OrbKp> kp;
    for (const Keypoint& q : k)                                             /* runByImageBorder(edgeThreshold) */
      if (q.x >= edge && q.x < lcols - edge && q.y >= edge && q.y < lrows - edge) kp.push_back({(float)q.x, (float)q.y, (float)q.score, -1.f});
    retain_best(kp, 2 * per[l]);                                            /* HARRIS_SCORE: twice the budget on the FAST score first */
    for (OrbKp& q : kp) q.response = harris_response(lev, lstride, (int)q.x, (int)q.y);
    retain_best(kp, per[l]);
    for (OrbKp& q : kp) q.angle = ic_angle(lev, lstride, (int)q.x, (int)q.y, half, umax);
    for (const OrbKp& q : kp) {
      if (total < cap) {
        float* o = out + 6 * (size_t)total;
        o[0] = l ? q.x * sc : q.x; o[1] = l ? q.y * sc : q.y; o[2] = patch * sc; o[3] = q.angle; o[4] = q.response; o[5] = (float)l;
      }
      ++total;
    }
  }
  *n = total;
  return total > cap ? VSLAM_ERR_CAPACITY : VSLAM_OK;
}

/* cv::ORB::create()->compute() on keypoints that carry an octave and an angle — what _descriptor_extractor->compute() does with an
 * OrbDetector's keypoints (base_framepoint_generator.cpp:431-438) [recalled: orb.cpp detectAndCompute, useProvidedKeypoints]:
 * runByImageBorder(edgeThreshold 31) on the level-0 coordinates (the Rect::contains test rounds the float point), a pyramid up to the
 * highest octave present (level l from level l-1, INTER_LINEAR, size cvRound(cols / scale) x cvRound(rows / scale), scale =
 * (float)pow(scaleFactor, l)), a 7x7 Gaussian per level, and per keypoint the 256 tests steered by ITS angle around
 * (cvRound(x * (1 / scale)), cvRound(y * (1 / scale))) of its level.  kp6: x, y, size, angle, response, octave per keypoint (orc_orb_detect).
 * A keypoint whose pattern would leave its level (never an OrbDetector's: they keep 31 px from their level's border) is removed too. */
ORC_API int orc_orb_describe_keypoints(const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t n, const float* kp6, float scale_factor,
                                       uint8_t* keep, uint8_t* desc) {
  if (!img || n < 0 || rows < 4 || cols < 4 || stride < cols || !(scale_factor > 1.f) || (n && (!kp6 || !keep || !desc))) return VSLAM_ERR_INVALID;
  int top = 0;
  for (int i = 0; i < n; ++i) { const int o = (int)kp6[6 * i + 5]; if (o < 0 || o > 15) return VSLAM_ERR_INVALID; top = std::max(top, o); }
  struct Level { std::vector<uint8_t> img, blur; int rows, cols; float scale; };
  std::vector<Level> L(top + 1);
  for (int l = 0; l <= top; ++l) {
    L[l].scale = (float)std::pow((double)scale_factor, (double)l);
    if (l == 0) {
      L[0].rows = rows; L[0].cols = cols;
      L[0].img.resize((size_t)rows * cols);
      for (int y = 0; y < rows; ++y) std::memcpy(&L[0].img[(size_t)y * cols], img + (size_t)y * stride, cols);
    } else {
      L[l].rows = cv_round(rows / L[l].scale); L[l].cols = cv_round(cols / L[l].scale);
      if (L[l].rows < 8 || L[l].cols < 8) return VSLAM_ERR_INVALID;
      L[l].img.resize((size_t)L[l].rows * L[l].cols);
      resize_linear_u8(L[l - 1].img.data(), L[l - 1].rows, L[l - 1].cols, L[l - 1].cols, L[l].img.data(), L[l].rows, L[l].cols);
    }
    gaussian_blur7_u8(L[l].img.data(), L[l].rows, L[l].cols, L[l].cols, L[l].blur);
  }
  const int reach = 23;   /* the rotated 31 x 31 pattern: |offset| <= cvRound(15 sqrt 2) = 21, + slack */
  for (int i = 0; i < n; ++i) {
    const float* k = kp6 + 6 * (size_t)i;
    const Level& lv = L[(int)k[5]];
    const float inv = 1.f / lv.scale;
    const int cx = cv_round(k[0] * inv), cy = cv_round(k[1] * inv);
    const bool in = orb_inside(rows, cols, cv_round(k[0]), cv_round(k[1])) && cx >= reach && cy >= reach && cx < lv.cols - reach && cy < lv.rows - reach;
    keep[i] = in ? 1 : 0;
    if (in) { float a, b; orb_rotation(k[3], &a, &b); orb_at(lv.blur.data(), lv.cols, cx, cy, a, b, desc + 32 * (size_t)i); }
    else std::memset(desc + 32 * (size_t)i, 0, 32);
  }
  return VSLAM_OK;
}

/* ---- synthetic data + trajectory error (test / bench infrastructure) ---------------------- */
ORC_API void orc_synth_default_kitti(synth_scene* s) { synth_default_kitti(s); }
ORC_API void orc_synth_default_euroc(synth_scene* s) { synth_default_euroc(s); }
ORC_API void orc_synth_pose(const synth_scene* s, int k, double cam_to_world[12]) {
  double R[9], t[3];
  synth_pose(s, k, R, t);
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) cam_to_world[4 * i + j] = R[3 * i + j]; cam_to_world[4 * i + 3] = t[i]; }
}
ORC_API void orc_synth_render_depth(const synth_scene* s, int frame, double unit_m, uint16_t* depth, int32_t stride) {
  double R[9], t[3];
  synth_pose(s, frame, R, t);
  for (int y = 0; y < s->rows; ++y)
    for (int x = 0; x < s->cols; ++x) depth[(size_t)y * stride + x] = synth_depth(s, R, t, x, y, unit_m);
}
ORC_API void orc_synth_render(const synth_scene* s, int frame, uint8_t* left, uint8_t* right, int32_t stride) {
  double R[9], t[3];
  synth_pose(s, frame, R, t);
  for (int y = 0; y < s->rows; ++y)
    for (int x = 0; x < s->cols; ++x) {
      left[(size_t)y * stride + x] = synth_pixel(s, R, t, frame, 0, x, y);
      right[(size_t)y * stride + x] = synth_pixel(s, R, t, frame, 1, x, y);
    }
}
