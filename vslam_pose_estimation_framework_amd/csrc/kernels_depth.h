// kernels_depth.h — RGB-D components: the pieces of DepthFramePointGenerator as stand-alone
// kernels.  gfx950, wave64.  HBM-bound byte / float work: one thread per pixel, coalesced rows, no LDS needed.
//
// Reference code replaced (paths relative to the reference root):
//   k_depth_*          DepthFramePointGenerator::_computeDepthMap   framepoint_generation/depth_framepoint_generator.cpp:410-485
//   k_depth_compute    DepthFramePointGenerator::compute            framepoint_generation/depth_framepoint_generator.cpp:45-164
//   k_point_in_camera  BaseFramePointGenerator::getPointInCamera    framepoint_generation/base_framepoint_generator.cpp:461-494
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/vslam_hip.h"

// ---- space map --------------------------------------------------------------------------------------------------
// The reference fills the map with a serial z-buffer whose test compares the STORED FLOAT with the new DOUBLE depth
// (:479), in scan order.  Its fixed point, per destination pixel, with F0 = float(maximum_depth), fl() = rounding to
// float and S = the sources whose fl(z) equals Fmin = min(F0, min fl(z)):
//     winner = the last source of S (scan order) with z < Fmin, or — if Fmin < F0 and no later such source exists — the
//     first source of S
// (a source replaces the entry whenever the stored float is strictly above its double depth; once the minimum float is
// stored only sources that round UP to it still pass).  Three order-free passes reproduce it exactly:
//   min   atomicMin of (float bits << 32 | source index): Fmin and the first source of S (positive floats order like
//         integers)
//   pick  sources of S with z < Fmin other than the first: atomicMax of the index ("last"); rare
//   write one thread per destination recomputes its winner (same arithmetic, same bits) and stores it
struct DepthSource { double pl[3]; int dest; };

__device__ __forceinline__ bool depth_source(const vslam_depth_params& p, const uint16_t* depth, int stride, int r, int c, DepthSource& s) {
  const unsigned raw = depth[(size_t)r * stride + c];
  if (!raw) return false;                                                                   // :449
  const double dm = (double)(int)raw * p.depth_scale_factor_intensity_to_meters;            // :452
  const double ph[3] = {c * dm, r * dm, dm};
  double pr[3], px[3];
  const double* Ki = p.K_right_inverse;
  const double* T = p.right_to_left;
  const double* K = p.K_left;
  // Pinhole matrices and a depth image registered to the colour image (identity offset) — every shipped RGB-D configuration: the products with
  // the structural zeros are exact zeros and x + 0 == x, so leaving them out gives the same bits as the full triple product below with a
  // third of its multiplications (uniform branch: the parameters are kernel arguments)
  const bool plain = Ki[1] == 0 && Ki[3] == 0 && Ki[6] == 0 && Ki[7] == 0 && Ki[8] == 1 && K[1] == 0 && K[3] == 0 && K[6] == 0 && K[7] == 0 && K[8] == 1 &&
                     T[0] == 1 && T[1] == 0 && T[2] == 0 && T[3] == 0 && T[4] == 0 && T[5] == 1 && T[6] == 0 && T[7] == 0 && T[8] == 0 && T[9] == 0 && T[10] == 1 && T[11] == 0;
  if (plain) {
    pr[0] = Ki[0] * ph[0] + Ki[2] * ph[2]; pr[1] = Ki[4] * ph[1] + Ki[5] * ph[2]; pr[2] = ph[2];
    s.pl[0] = pr[0]; s.pl[1] = pr[1]; s.pl[2] = pr[2];
    if (s.pl[2] <= 0) return false;
    px[0] = K[0] * s.pl[0] + K[2] * s.pl[2]; px[1] = K[4] * s.pl[1] + K[5] * s.pl[2]; px[2] = s.pl[2];
  } else {
  for (int i = 0; i < 3; ++i) pr[i] = (Ki[3 * i] * ph[0] + Ki[3 * i + 1] * ph[1]) + Ki[3 * i + 2] * ph[2];   // :454
  for (int i = 0; i < 3; ++i) s.pl[i] = ((T[4 * i] * pr[0] + T[4 * i + 1] * pr[1]) + T[4 * i + 2] * pr[2]) + T[4 * i + 3];   // :456
  if (s.pl[2] <= 0) return false;                                                           // :458-460
  for (int i = 0; i < 3; ++i) px[i] = (K[3 * i] * s.pl[0] + K[3 * i + 1] * s.pl[1]) + K[3 * i + 2] * s.pl[2];   // :462
  }
  const double u = px[0] / px[2], v = px[1] / px[2];                                        // :463
  const double ru = round(u), rv = round(v);                                                // :466-467 (half away from zero)
  if (!(rv >= 0 && rv < p.rows && ru >= 0 && ru < p.cols)) return false;                    // :470-474
  s.dest = (int)rv * p.cols + (int)ru;
  return true;
}

// (the four kernels take a batch of equally sized images: blockIdx.z — blockIdx.y for the 1-D init — is the image, every array advanced by
// rows * cols elements per image; the stand-alone entry launches one image)
__global__ __launch_bounds__(256) void k_depth_init(int n, uint32_t f0_bits, unsigned long long* key, int32_t* last) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const size_t zo = (size_t)blockIdx.y * n;
  if (i < n) { key[zo + i] = ((unsigned long long)f0_bits << 32) | 0xffffffffull; last[zo + i] = -1; }
}

// min + first in ONE 64-bit atomic per source: key = float bits of the depth << 32 | source index
// gate (may be null): one int per image; the kernel skips an image whose entry is 0 (k_depth_direct found that the general z-buffer is not needed)
__global__ __launch_bounds__(256) void k_depth_min(const vslam_depth_params p, const uint16_t* depth, int stride, unsigned long long* key, const int32_t* gate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.cols) return;
  if (gate && !gate[blockIdx.z]) return;
  depth += (size_t)blockIdx.z * p.rows * stride; key += (size_t)blockIdx.z * p.rows * p.cols;
  for (int r = blockIdx.y; r < p.rows; r += gridDim.y) {     // (a gated launch uses a few row blocks only: it almost never has work)
    DepthSource s;
    if (!depth_source(p, depth, stride, r, c, s)) continue;
    atomicMin(&key[s.dest], ((unsigned long long)__float_as_uint((float)s.pl[2]) << 32) | (unsigned)(r * p.cols + c));
  }
}

// "last": only sources that tie the minimum float, lie strictly below it as doubles and are not the first one already
// recorded — with one source per destination (the usual case) no atomic is issued here at all
__global__ __launch_bounds__(256) void k_depth_pick(const vslam_depth_params p, const uint16_t* depth, int stride, uint32_t f0_bits,
                                                    const unsigned long long* key, int32_t* last, const int32_t* gate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.cols) return;
  if (gate && !gate[blockIdx.z]) return;
  { const size_t zo = (size_t)blockIdx.z * p.rows * p.cols; depth += (size_t)blockIdx.z * p.rows * stride; key += zo; last += zo; }
  for (int r = blockIdx.y; r < p.rows; r += gridDim.y) {
    DepthSource s;
    if (!depth_source(p, depth, stride, r, c, s)) continue;
    const unsigned long long k = key[s.dest];
    const uint32_t m = (uint32_t)(k >> 32);
    if (__float_as_uint((float)s.pl[2]) != m) continue;
    const int idx = r * p.cols + c;
    if (!(s.pl[2] < (double)__uint_as_float(m))) continue;
    if (m < f0_bits && (uint32_t)idx == (uint32_t)k) continue;   // the first source wins anyway unless a later one passes
    atomicMax(&last[s.dest], idx);
  }
}

// rearm: the z-buffer entries are put back to their initial values once read (the device-resident loop initialises them once and lets every
// frame leave them ready for the next: one kernel and one pass over the buffers less per frame)
__global__ __launch_bounds__(256) void k_depth_write(const vslam_depth_params p, const uint16_t* depth, int stride, uint32_t f0_bits,
                                                     unsigned long long* key, int32_t* last, float* space,
                                                     int16_t* row_map, int16_t* col_map, int rearm, const int32_t* gate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.cols) return;
  if (gate && !gate[blockIdx.z]) return;
  { const size_t zo = (size_t)blockIdx.z * p.rows * p.cols; depth += (size_t)blockIdx.z * p.rows * stride; key += zo; last += zo; space += 3 * zo; row_map += zo; col_map += zo; }
  for (int r = blockIdx.y; r < p.rows; r += gridDim.y) {
    const int d = r * p.cols + c;
    const unsigned long long k = key[d];
    int win = last[d];
    if (rearm) { key[d] = ((unsigned long long)f0_bits << 32) | 0xffffffffull; last[d] = -1; }
    if ((uint32_t)(k >> 32) < f0_bits && (uint32_t)k != 0xffffffffu) win = max(win, (int)(uint32_t)k);
    float o[3] = {0.f, 0.f, __uint_as_float(f0_bits)};                                       // :428-432
    int sr = -1, sc = -1;
    if (win >= 0) {
      sr = win / p.cols; sc = win - sr * p.cols;
      DepthSource s;
      depth_source(p, depth, stride, sr, sc, s);
      o[0] = (float)s.pl[0]; o[1] = (float)s.pl[1]; o[2] = (float)s.pl[2];                  // :480-482
    }
    space[3 * (size_t)d] = o[0]; space[3 * (size_t)d + 1] = o[1]; space[3 * (size_t)d + 2] = o[2];
    row_map[d] = (int16_t)sr; col_map[d] = (int16_t)sc;                                      // :483-484
  }
}

// The z-buffer without a z-buffer: when every source pixel projects onto ITSELF (a depth image registered to the colour image and one camera
// matrix: every shipped RGB-D configuration) a destination has at most one source, and the three passes above reduce, per pixel, to
//     the source wins iff fl(z) < F0, or fl(z) == F0 and z < F0 as doubles          (min: key = (fl(z), index) when fl(z) <= F0; pick: the tie case)
// One pass, no atomics (the three passes are bound by 64-bit atomics: 430 + 160 + 240 us for 256 images of 620 x 188).  Whether the premise
// holds is CHECKED, not assumed: a source that lands on another pixel raises its image's flag, and the general passes — gated on that flag —
// then recompute the image from scratch.
__global__ __launch_bounds__(256) void k_depth_direct(const vslam_depth_params p, const uint16_t* depth, int stride, uint32_t f0_bits, float* space,
                                                      int16_t* row_map, int16_t* col_map, int32_t* cross) {
  __shared__ float tile[256 * 3];
  const int c = blockIdx.x * 256 + threadIdx.x;
  { const size_t zo = (size_t)blockIdx.z * p.rows * p.cols; depth += (size_t)blockIdx.z * p.rows * stride; space += 3 * zo; row_map += zo; col_map += zo; }
  const int c0 = blockIdx.x * 256, nvalid = min(256, p.cols - c0) * 3;
  // a workgroup walks down its 256-column strip (a workgroup per row segment is 144 k workgroups of 256 pixels for 256 small images: the
  // dispatch, not the arithmetic, was the cost)
  for (int r = blockIdx.y; r < p.rows; r += gridDim.y) {
    const int d = r * p.cols + c;
    DepthSource s;
    float o[3] = {0.f, 0.f, __uint_as_float(f0_bits)};                                       // :428-432
    int sr = -1, sc = -1;
    if (c < p.cols && depth_source(p, depth, stride, r, c, s)) {
      if (s.dest != d) {
        if (!__hip_atomic_load(&cross[blockIdx.z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&cross[blockIdx.z], 1);
      } else {
        const uint32_t zb = __float_as_uint((float)s.pl[2]);
        if (zb < f0_bits || (zb == f0_bits && s.pl[2] < (double)__uint_as_float(f0_bits))) {
          o[0] = (float)s.pl[0]; o[1] = (float)s.pl[1]; o[2] = (float)s.pl[2];               // :480-482
          sr = r; sc = c;
        }
      }
    }
    // the three floats of a pixel through LDS: the workgroup's 256 x 3 floats leave as three fully coalesced rows of stores instead of three
    // stride-12-byte ones
    __syncthreads();
    tile[3 * threadIdx.x] = o[0]; tile[3 * threadIdx.x + 1] = o[1]; tile[3 * threadIdx.x + 2] = o[2];
    __syncthreads();
    float* out = space + 3 * ((size_t)r * p.cols + c0);
    for (int k = threadIdx.x; k < nvalid; k += 256) out[k] = tile[k];
    if (c < p.cols) { row_map[d] = (int16_t)sr; col_map[d] = (int16_t)sc; }                  // :483-484
  }
}

// ---- compute ----------------------------------------------------------------------------------------------------
// One 1024-thread workgroup per image (the unit of the reference's loop; ~2000 features).  The serial bin competition
// "an untracked occupant is replaced by a strictly lower depth" (:117-129) ends with the FIRST feature of the bin's
// minimum depth: atomicMin of (float bits of the depth << 32 | feature index); bins owned by a tracked point hold 0.
// cls: 0 dropped, 1 new point, 2 temporary point (depth >= maximum, triangulation enabled).
__device__ __forceinline__ int depth_bin(const vslam_depth_params& p, int row, int col, int cols_bin1) {
  const int rb = (int)rint((double)row / p.bin_size_pixels), cb = (int)rint((double)col / p.bin_size_pixels);   // :59-60, :113-114
  return rb * cols_bin1 + cb;
}

// body (1024 threads, sh = 17 ints of LDS): also called by the device-resident RGB-D loop (kernels_rgbd.h) with counts it reads
// from device memory
__device__ __forceinline__ void depth_compute_body(const vslam_depth_params& p, const float* space, int nF, const int32_t* rcF, int nT,
                                                   const int32_t* rcT, unsigned long long* bins, int n_bins, int rows_bin, int cols_bin,
                                                   int cap, int32_t* counts, int32_t* new_feat, double* new_xyz, int32_t* temp_feat,
                                                   double* temp_xyz, uint8_t* cls, int* sh) {
  const int tid = threadIdx.x;
  const int cols_bin1 = cols_bin + 1;   // rint() can reach the grid size (latent overflow upstream): spare row / column
  const int NT = blockDim.x;       // 1024 (stand-alone entries) or 512 (device-resident loop on small images)
  for (int i = tid; i < n_bins; i += NT) bins[i] = ~0ull;
  __syncthreads();
  if (p.enable_keypoint_binning)
    for (int i = tid; i < nT; i += NT) bins[depth_bin(p, rcT[2 * i], rcT[2 * i + 1], cols_bin1)] = 0ull;     // :57-63
  __syncthreads();
  for (int i = tid; i < nF; i += NT) {
    const int row = rcF[2 * i], col = rcF[2 * i + 1];
    const float z = space[3 * ((size_t)row * p.cols + col) + 2];
    uint8_t k = 0;
    if (!((double)z < p.minimum_depth_meters)) {                                                               // :80
      if ((double)z >= p.maximum_depth_meters && p.enable_point_triangulation) k = 2;                          // :89
      else {
        k = 1;
        if (p.enable_keypoint_binning)
          atomicMin(&bins[depth_bin(p, row, col, cols_bin1)], ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)i);
      }
    }
    cls[i] = k;
  }
  __syncthreads();
  // temporary points, feature order (:92-100)
  int n_temp = 0;
  for (int base = 0; base < nF; base += NT) {
    const int i = base + tid;
    const int mine = (i < nF && cls[i] == 2) ? 1 : 0;
    int total;
    const int at = n_temp + block_exclusive_scan(mine, sh, &total);
    if (mine && at < cap) {
      const double m = p.maximum_depth_meters;
      const double ph[3] = {rcF[2 * i + 1] * m, rcF[2 * i] * m, m};
      const double* Ki = p.K_left_inverse;
      temp_feat[at] = i;
      for (int q = 0; q < 3; ++q) temp_xyz[3 * (size_t)at + q] = (Ki[3 * q] * ph[0] + Ki[3 * q + 1] * ph[1]) + Ki[3 * q + 2] * ph[2];
    }
    n_temp += total;
  }
  // new points: bin grid row-major (:141-157) or feature order (:160-162)
  int n_new = 0;
  const int n_scan = p.enable_keypoint_binning ? rows_bin * cols_bin : nF;
  for (int base = 0; base < n_scan; base += NT) {
    const int j = base + tid;
    int feat = -1;
    if (j < n_scan) {
      if (p.enable_keypoint_binning) {
        const int rb = j / cols_bin, cb = j - rb * cols_bin;
        const unsigned long long key = bins[rb * cols_bin1 + cb];
        if (key != ~0ull && key != 0ull) feat = (int)(key & 0xffffffffu);
      } else if (cls[j] == 1) {
        feat = j;
      }
    }
    int total;
    const int at = n_new + block_exclusive_scan(feat >= 0 ? 1 : 0, sh, &total);
    if (feat >= 0 && at < cap) {
      const float* d = space + 3 * ((size_t)rcF[2 * feat] * p.cols + rcF[2 * feat + 1]);
      new_feat[at] = feat;
      for (int q = 0; q < 3; ++q) new_xyz[3 * (size_t)at + q] = (double)d[q];
    }
    n_new += total;
  }
  if (tid == 0) { counts[0] = n_new; counts[1] = n_temp; }
}
__global__ __launch_bounds__(1024) void k_depth_compute(const vslam_depth_params p, const float* space, int nF, const int32_t* rcF, int nT,
                                                        const int32_t* rcT, unsigned long long* bins, int n_bins, int rows_bin, int cols_bin,
                                                        int cap, int32_t* counts, int32_t* new_feat, double* new_xyz, int32_t* temp_feat,
                                                        double* temp_xyz, uint8_t* cls) {
  __shared__ int sh[17];
  depth_compute_body(p, space, nF, rcF, nT, rcT, bins, n_bins, rows_bin, cols_bin, cap, counts, new_feat, new_xyz, temp_feat, temp_xyz, cls, sh);
}

// ---- midpoint triangulation -------------------------------------------------------------------------------------
// The 3x2 least-squares system [-R x0 | x1] z = t (:480-486): QR of the two columns, triangular solve; minimum-norm
// solution when the rays are parallel to rounding (JacobiSVD's rank rule: sigma_min <= 2 eps sigma_max).
__device__ __forceinline__ void point_in_camera_one(const float* xp2, const float* xc2, const double* T, const double* K, double* out3) {
  const double a0 = ((double)xp2[0] - K[2]) / K[0], b0 = ((double)xp2[1] - K[5]) / K[4];   // :470-473
  const double a1 = ((double)xc2[0] - K[2]) / K[0], b1 = ((double)xc2[1] - K[5]) / K[4];
  const double x0[3] = {a0, b0, 1}, x1[3] = {a1, b1, 1};
  double c0[3], t[3];
  for (int k = 0; k < 3; ++k) { c0[k] = -((T[4 * k] * x0[0] + T[4 * k + 1] * x0[1]) + T[4 * k + 2] * x0[2]); t[k] = T[4 * k + 3]; }
  const double r00 = sqrt((c0[0] * c0[0] + c0[1] * c0[1]) + c0[2] * c0[2]);
  double q0[3], v[3];
  for (int k = 0; k < 3; ++k) q0[k] = c0[k] / r00;
  const double r01 = (q0[0] * x1[0] + q0[1] * x1[1]) + q0[2] * x1[2];
  for (int k = 0; k < 3; ++k) v[k] = x1[k] - r01 * q0[k];
  const double r11 = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  const double g0 = (q0[0] * t[0] + q0[1] * t[1]) + q0[2] * t[2];
  const double fro2 = (r00 * r00 + r01 * r01) + r11 * r11;
  double z0, z1;
  if (r00 * r11 > 2 * 2.220446049250313e-16 * fro2) {
    const double g1 = ((v[0] * t[0] + v[1] * t[1]) + v[2] * t[2]) / r11;
    z1 = g1 / r11;
    z0 = (g0 - r01 * z1) / r00;
  } else {
    const double n2 = r00 * r00 + r01 * r01;
    z0 = g0 * r00 / n2; z1 = g0 * r01 / n2;
  }
  const double pp[3] = {x0[0] * z0, x0[1] * z0, x0[2] * z0};                                        // :489
  for (int k = 0; k < 3; ++k) {
    const double moved = ((T[4 * k] * pp[0] + T[4 * k + 1] * pp[1]) + T[4 * k + 2] * pp[2]) + T[4 * k + 3];
    out3[k] = (x1[k] * z1 + moved) / 2.0;                                                           // :490-493
  }
}
__global__ __launch_bounds__(256) void k_point_in_camera(int n, const float* xp, const float* xc, const double* Tg, const double* Kg, double* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double T[12], K[9];
  for (int k = 0; k < 12; ++k) T[k] = Tg[k];
  for (int k = 0; k < 9; ++k) K[k] = Kg[k];
  point_in_camera_one(xp + 2 * (size_t)i, xc + 2 * (size_t)i, T, K, out + 3 * (size_t)i);
}

// ---- track ------------------------------------------------------------------------------------------------------
// The reference walks the previous points in order; a matched feature leaves the lattice, so a later point that would
// have picked it takes its next-best one (:243-244).  Order-exact in parallel, as in the stereo tracker: iterate
//   pick(i) = best feature of point i's window among those no EARLIER point holds;  hold(f) = min { i : pick(i) = f }
// from "nobody holds anything" until nothing changes (point 0 is final after one sweep, point 1 after two, ...: the fixed
// point is the serial outcome; a handful of sweeps in practice).  Two kernels: k_depth_track_candidates (wide) sorts every
// point's window candidates once — features are row-major with a (row, 16-px cell) CSR, so a window scan reads only the cells
// it overlaps —, k_depth_track (one workgroup per image, one thread per previous point and sweep) walks those lists.
// A match on a pixel below the minimum depth is dropped WITHOUT taking the feature (:238-240).
#define VS_DT_K 6
struct DepthTrack {
  vslam_depth_params p;
  double T[12];
  int d, by_app;
  double tau;
  int nP, nL, CW;
  const double* cam; const uint8_t* pdesc; const uint8_t* pflags;
  const int16_t* kxy;        // [nL][2] x, y  (features sorted row-major)
  const uint8_t* desc;       // [nL][32]
  const int32_t* rowcell;    // [rows][CW + 1]
  const float* space;
  const uint8_t* fvis;       // [nL] 0: another feature was written onto this one's lattice cell (never found by track()); null = all visible
  int32_t* hold;             // [2][nL]
  int32_t* pick;             // [nP]  feature, -1 none, -2 projection outside
  unsigned long long* cand;  // [nP][VS_DT_K + 1] the K best keys of the point's window (ascending) + the candidate count
  int32_t* counts;           // tracked, temporary, lost, tracked landmarks
  int32_t* out2; double* xyz; int32_t* temp2; int32_t* lost;
};

// full window scan of one point under the current holds (used when its candidate list is exhausted or incomplete)
__device__ __forceinline__ int depth_track_best(const DepthTrack& a, int i, int row, int col, const uint32_t* pd, const int32_t* hold) {
  const int rows = a.p.rows, cols = a.p.cols;
  const int r0 = max(row - a.d, 0), r1 = min(row + a.d + 1, rows);                  // :214-217
  const int c0 = max(col - a.d, 0), c1 = min(col + a.d + 1, cols);
  if (c1 <= c0) return -1;
  const int cl = c0 >> 4, ch = ((c1 - 1) >> 4) + 1;
  unsigned long long best = ~0ull;    // (primary << 32 | feature): row-major index == the reference's scan order for ties
  for (int r = r0; r < r1; ++r) {
    const int lo = a.rowcell[(size_t)r * (a.CW + 1) + cl], hi = a.rowcell[(size_t)r * (a.CW + 1) + ch];
    for (int k = lo; k < hi; ++k) {
      const int x = a.kxy[2 * k];
      if (x < c0 || x >= c1) continue;
      if (a.fvis && !a.fvis[k]) continue;                                           // overwritten in the lattice
      if (hold[k] < i) continue;                                                    // an earlier point removed it from the lattice
      const uint32_t* kd = reinterpret_cast<const uint32_t*>(a.desc + (size_t)32 * k);
      int h = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) h += __popc(pd[u] ^ kd[u]);
      if (!((double)h < a.tau)) continue;                                           // intensity_feature_matcher.cpp:100-121
      unsigned prim;
      if (a.by_app) prim = (unsigned)h;
      else { const int dr = row - r, dc = col - x; prim = (unsigned)(dr * dr + dc * dc); if (prim >= 10000u) continue; }
      const unsigned long long key = ((unsigned long long)prim << 32) | (unsigned)k;
      if (key < best) best = key;
    }
  }
  return best == ~0ull ? -1 : (int)(best & 0xffffffffu);
}

// projection of previous point i (:197-208): false = outside / behind (neither tracked nor lost)
__device__ __forceinline__ bool depth_track_project(const DepthTrack& a, int i, int* row, int* col) {
  double q[3], uvw[3];
  const double* cm = a.cam + 3 * (size_t)i;
  for (int k = 0; k < 3; ++k) q[k] = ((a.T[4 * k] * cm[0] + a.T[4 * k + 1] * cm[1]) + a.T[4 * k + 2] * cm[2]) + a.T[4 * k + 3];   // :197
  for (int k = 0; k < 3; ++k) uvw[k] = (a.p.K_left[3 * k] * q[0] + a.p.K_left[3 * k + 1] * q[1]) + a.p.K_left[3 * k + 2] * q[2];     // :200
  if (!(uvw[2] > 0)) return false;
  const double uc = uvw[0] / uvw[2], ur = uvw[1] / uvw[2];
  if (!(uc > -2147483648.0 && uc < 2147483648.0 && ur > -2147483648.0 && ur < 2147483648.0)) return false;
  *col = (int)uc; *row = (int)ur;                                                   // :201-202 (truncation)
  return !(*col < 0 || *col > a.p.cols || *row < 0 || *row > a.p.rows);             // :205-208
}

// Wide first pass (nothing is held yet): one 16-lane group per previous point, the lanes split the window rows, every
// candidate key goes to the group's LDS buffer, a rank sort orders them and the VS_DT_K best + the count are kept for the
// sweeps of k_depth_track.  More than VS_DT_CAP candidates: the list is declared incomplete (count = huge, no keys), the
// resolution kernel then rescans that window itself.
#define VS_DT_CAP 32
__device__ __forceinline__ void depth_track_candidates_body(const DepthTrack& a, unsigned long long (*keys)[VS_DT_CAP], int* cnt) {
  const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
  for (int i = blockIdx.x * 16 + g; i < a.nP; i += gridDim.x * 16) {
    unsigned long long* list = a.cand + (size_t)i * (VS_DT_K + 1);
    int row = 0, col = 0;
    const bool ok = depth_track_project(a, i, &row, &col);
    if (lane == 0) cnt[g] = 0;
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      const int rows = a.p.rows, cols = a.p.cols;
      const int r0 = max(row - a.d, 0), r1 = min(row + a.d + 1, rows);
      const int c0 = max(col - a.d, 0), c1 = min(col + a.d + 1, cols);
      if (c1 > c0) {
        uint32_t pd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pd[u] = reinterpret_cast<const uint32_t*>(a.pdesc + (size_t)32 * i)[u];
        const int cl = c0 >> 4, ch = ((c1 - 1) >> 4) + 1;
        for (int r = r0 + lane; r < r1; r += 16) {
          const int lo = a.rowcell[(size_t)r * (a.CW + 1) + cl], hi = a.rowcell[(size_t)r * (a.CW + 1) + ch];
          for (int k = lo; k < hi; ++k) {
            const int x = a.kxy[2 * k];
            if (x < c0 || x >= c1) continue;
            if (a.fvis && !a.fvis[k]) continue;
            const uint32_t* kd = reinterpret_cast<const uint32_t*>(a.desc + (size_t)32 * k);
            int h = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) h += __popc(pd[u] ^ kd[u]);
            if (!((double)h < a.tau)) continue;
            unsigned prim;
            if (a.by_app) prim = (unsigned)h;
            else { const int dr = row - r, dc = col - x; prim = (unsigned)(dr * dr + dc * dc); if (prim >= 10000u) continue; }
            const int slot = atomicAdd(&cnt[g], 1);
            if (slot < VS_DT_CAP) keys[g][slot] = ((unsigned long long)prim << 32) | (unsigned)k;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int n = __hip_atomic_load(&cnt[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (n <= VS_DT_CAP) {
      for (int j = lane; j < n; j += 16) {
        const unsigned long long mine = __hip_atomic_load(&keys[g][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int rank = 0;
        for (int u = 0; u < n; ++u) rank += __hip_atomic_load(&keys[g][u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < mine ? 1 : 0;
        if (rank < VS_DT_K) list[rank] = mine;
      }
      if (lane == 0) list[VS_DT_K] = (unsigned long long)n;
    } else if (lane == 0) {
      list[VS_DT_K] = 0x7fffffffull;          // incomplete: k_depth_track rescans
    }
    if (lane == 0) a.pick[i] = ok ? -1 : -2;  // -2: projection outside
    __builtin_amdgcn_wave_barrier();
  }
}
__global__ __launch_bounds__(256) void k_depth_track_candidates(const DepthTrack a) {
  __shared__ unsigned long long keys[16][VS_DT_CAP];
  __shared__ int cnt[16];
  depth_track_candidates_body(a, keys, cnt);
}

// body (1024 threads; sh = 17 ints, changed_p = one int of LDS)
__device__ __forceinline__ void depth_track_body(const DepthTrack& a, int* sh, int* changed_p) {
  int& changed = *changed_p;
  const int tid = threadIdx.x;
  int32_t* hold = a.hold;
  int32_t* next = a.hold + a.nL;
  const int NT = blockDim.x;
  for (int k = tid; k < a.nL; k += NT) hold[k] = 0x7fffffff;
  __syncthreads();
  for (int sweep = 0; sweep <= a.nP; ++sweep) {
    for (int k = tid; k < a.nL; k += NT) next[k] = 0x7fffffff;
    if (tid == 0) changed = 0;
    __syncthreads();
    for (int i = tid; i < a.nP; i += NT) {
      int f = a.pick[i] == -2 ? -2 : -1;
      if (f != -2) {
        const unsigned long long* list = a.cand + (size_t)i * (VS_DT_K + 1);
        const int cnt = (int)list[VS_DT_K];
        bool found = false;
        if (cnt != 0x7fffffff)
          for (int u = 0; u < VS_DT_K && u < cnt; ++u) {
            const int k = (int)(list[u] & 0xffffffffu);
            if (hold[k] >= i) { f = k; found = true; break; }      // the best candidate no earlier point holds
          }
        if (!found && cnt > VS_DT_K) {                               // list exhausted or incomplete: scan the window
          int row = 0, col = 0;
          depth_track_project(a, i, &row, &col);
          uint32_t pd[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) pd[u] = reinterpret_cast<const uint32_t*>(a.pdesc + (size_t)32 * i)[u];
          f = depth_track_best(a, i, row, col, pd, hold);
        }
      }
      if (f >= 0) {
        const float z = a.space[3 * ((size_t)a.kxy[2 * f + 1] * a.p.cols + a.kxy[2 * f]) + 2];
        if (!((double)z < a.p.minimum_depth_meters)) atomicMin(&next[f], i);        // :238-244
      }
      a.pick[i] = f;
    }
    __syncthreads();
    for (int k = tid; k < a.nL; k += NT) if (next[k] != hold[k]) changed = 1;
    __syncthreads();
    const bool again = changed != 0;
    int32_t* t = hold; hold = next; next = t;
    __syncthreads();
    if (!again) break;
  }
  // outcome of every point under the final holds, lists in the order of the previous points
  int n_trk = 0, n_tmp = 0, n_lost = 0, n_lm = 0;
  for (int base = 0; base < a.nP; base += NT) {
    const int i = base + tid;
    int kind = 0;   // 1 tracked, 2 temporary, 3 lost
    int f = -1;
    if (i < a.nP) {
      f = a.pick[i];
      if (f >= 0) {
        const float z = a.space[3 * ((size_t)a.kxy[2 * f + 1] * a.p.cols + a.kxy[2 * f]) + 2];
        if ((double)z < a.p.minimum_depth_meters) kind = 0;                                            // :238-240: neither tracked nor lost
        else if ((double)z >= a.p.maximum_depth_meters && a.p.enable_point_triangulation) kind = 2;   // :247-256
        else kind = 1;
      } else if (f == -1 && !(a.pflags[i] & 2)) {
        kind = 3;                                                                                      // :281-284
      }
    }
    int total;
    const int at1 = n_trk + block_exclusive_scan(kind == 1 ? 1 : 0, sh, &total);
    n_trk += total;
    const int at2 = n_tmp + block_exclusive_scan(kind == 2 ? 1 : 0, sh, &total);
    n_tmp += total;
    const int at3 = n_lost + block_exclusive_scan(kind == 3 ? 1 : 0, sh, &total);
    n_lost += total;
    block_exclusive_scan((kind == 1 && (a.pflags[min(i, a.nP - 1)] & 1)) ? 1 : 0, sh, &total);
    n_lm += total;
    if (kind == 1) {
      const float* dp = a.space + 3 * ((size_t)a.kxy[2 * f + 1] * a.p.cols + a.kxy[2 * f]);
      a.out2[2 * at1] = i; a.out2[2 * at1 + 1] = f;
      for (int k = 0; k < 3; ++k) a.xyz[3 * (size_t)at1 + k] = (double)dp[k];
    } else if (kind == 2) {
      a.temp2[2 * at2] = i; a.temp2[2 * at2 + 1] = f;
    } else if (kind == 3) {
      a.lost[at3] = i;
    }
  }
  if (tid == 0) { a.counts[0] = n_trk; a.counts[1] = n_tmp; a.counts[2] = n_lost; a.counts[3] = n_lm; }
}
__global__ __launch_bounds__(1024) void k_depth_track(const DepthTrack a) {
  __shared__ int sh[17];
  __shared__ int changed;
  depth_track_body(a, sh, &changed);
}

// ---- recoverPoints ----------------------------------------------------------------------------------------------
// One thread per lost point: projection of its landmark, field-of-view, depth and border gates (:317-359), the pixel
// BRIEF has to be evaluated at (ROI origin = rounded corner: cv::Rect_<float> -> cv::Rect) and the keypoint the new
// feature gets (:384).  k_brief_at then describes the pixels; k_depth_recover_finish applies the descriptor gate
// (:381-383) and emits the survivors in the order of the lost list.
struct DepthRecover {
  vslam_depth_params p;
  double w2c[12];
  float kp_size;
  double tau;
  int n;
  const uint8_t* has_lm; const double* lm; const uint8_t* pdesc;
  const float* space;
  int16_t* bxy;       // [n][2] pixel for BRIEF (0, 0 = rejected: outside the descriptor border, k_brief_at keeps 0)
  float* kxy;         // [n][2] keypoint of the recovered feature
  int32_t* cell;      // [n] space-map index of the depth lookup, -1 rejected
  const uint8_t* keep; const uint8_t* desc;   // k_brief_at outputs
  int32_t* count; int32_t* rec_index; float* rec_xy; uint8_t* rec_desc; double* rec_xyz;
};

__device__ __forceinline__ void depth_recover_project_one(const DepthRecover& a, int i) {
  int cell = -1, bx = 0, by = 0;
  float kx = 0.f, ky = 0.f;
  if (a.has_lm[i]) {                                                              // :305-307
    const double* X = a.lm + 3 * (size_t)i;
    double pc[3], pi[3];
    for (int k = 0; k < 3; ++k) pc[k] = ((a.w2c[4 * k] * X[0] + a.w2c[4 * k + 1] * X[1]) + a.w2c[4 * k + 2] * X[2]) + a.w2c[4 * k + 3];   // :317
    for (int k = 0; k < 3; ++k) pi[k] = (a.p.K_left[3 * k] * pc[0] + a.p.K_left[3 * k + 1] * pc[1]) + a.p.K_left[3 * k + 2] * pc[2];       // :325
    const double x = pi[0] / pi[2], y = pi[1] / pi[2];                            // :326
    const int rows = a.p.rows, cols = a.p.cols;
    if (x >= 0 && x <= cols && y >= 0 && y <= rows) {                             // :329-332
      const float px = (float)x, py = (float)y;                                   // :335
      const float fr = rintf(py), fc = rintf(px);                                 // :338
      if (fr >= 0 && fr < rows && fc >= 0 && fc < cols) {
        const int cl = (int)fr * cols + (int)fc;
        const float z = a.space[3 * (size_t)cl + 2];
        if (!((double)z < a.p.minimum_depth_meters || (double)z >= a.p.maximum_depth_meters)) {   // :341-344
          const float rbc = 5 * a.kp_size;                                        // :347
          if (!(px <= rbc + 1 || px >= cols - rbc - 1 || py <= rbc + 1 || py >= rows - rbc - 1)) {   // :352-359
            const float cxf = px - rbc, cyf = py - rbc;                           // :362
            bx = (int)rintf(cxf) + (int)(rbc + 0.5f); by = (int)rintf(cyf) + (int)(rbc + 0.5f);
            kx = rbc + cxf; ky = rbc + cyf;                                       // :384
            cell = cl;
          }
        }
      }
    }
  }
  a.cell[i] = cell;
  a.bxy[2 * i] = (int16_t)(cell >= 0 ? bx : 0); a.bxy[2 * i + 1] = (int16_t)(cell >= 0 ? by : 0);
  a.kxy[2 * i] = kx; a.kxy[2 * i + 1] = ky;
}
__global__ __launch_bounds__(256) void k_depth_recover_project(const DepthRecover a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  depth_recover_project_one(a, i);
}

__device__ __forceinline__ void depth_recover_finish_body(const DepthRecover& a, int* sh) {
  const int tid = threadIdx.x;
  int n_rec = 0;
  const int NT = blockDim.x;
  for (int base = 0; base < a.n; base += NT) {
    const int i = base + tid;
    int ok = 0;
    if (i < a.n && a.cell[i] >= 0 && a.keep[i]) {
      const uint32_t* d = reinterpret_cast<const uint32_t*>(a.desc + (size_t)32 * i);
      const uint32_t* q = reinterpret_cast<const uint32_t*>(a.pdesc + (size_t)32 * i);
      int h = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) h += __popc(d[u] ^ q[u]);
      ok = !((double)h > a.tau);                                                  // :381-383
    }
    int total;
    const int at = n_rec + block_exclusive_scan(ok, sh, &total);
    n_rec += total;
    if (ok) {
      a.rec_index[at] = i;
      a.rec_xy[2 * at] = a.kxy[2 * i]; a.rec_xy[2 * at + 1] = a.kxy[2 * i + 1];
      for (int u = 0; u < 8; ++u) reinterpret_cast<uint32_t*>(a.rec_desc + (size_t)32 * at)[u] = reinterpret_cast<const uint32_t*>(a.desc + (size_t)32 * i)[u];
      for (int k = 0; k < 3; ++k) a.rec_xyz[3 * (size_t)at + k] = (double)a.space[3 * (size_t)a.cell[i] + k];   // :390
    }
  }
  if (tid == 0) a.count[0] = n_rec;
}
__global__ __launch_bounds__(1024) void k_depth_recover_finish(const DepthRecover a) {
  __shared__ int sh[17];
  depth_recover_finish_body(a, sh);
}
