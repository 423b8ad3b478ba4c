// kernels_frame.h — per-stream kernels: temporal-track candidate search (K4) and the frame kernel
// (K5): order-exact track resolution, StereoUVAligner, tracker control logic, prune, recovery,
// landmark refinement, stereo sweep + binning.  One 512-thread workgroup owns one stream, so the
// reference's sequential per-frame control flow (PoseTracker3D::compute) runs on the device with
// workgroup barriers only: no inter-workgroup hand-off, no host round trip.
//
// Reference code replaced:
//   K4/K5 track      StereoFramePointGenerator::track           stereo_framepoint_generator.cpp:464-681
//                    getMatchingFeatureInRectangularRegion       intensity_feature_matcher.cpp:81-148
//   K5 align         StereoUVAligner::initialize/linearize/oneRound/converge  stereouv_aligner.cpp:10-264
//   K5 control       PoseTracker3D::compute/_track/_registerRecursive/_prunePoints/_updatePoints
//                                                                pose_tracker_3d.cpp:32-566
//   K5 recover       StereoFramePointGenerator::recoverPoints    stereo_framepoint_generator.cpp:683-869
//   K5 landmarks     Landmark::Landmark / Landmark::update       types/landmark.cpp:8-33,66-167
//   K5 stereo        StereoFramePointGenerator::compute          stereo_framepoint_generator.cpp:135-462
#pragma once
#include "kernels_image.h"
#include "dev_math.h"

// Fine-grained phase clocks of the frame kernel (tools/probe/phase_clocks_per_stream.py): compiled in only with -DVS_PROFILE_PHASES, every
// stamp is a global read-modify-write by thread 0 and costs about a microsecond.
#ifdef VS_PROFILE_PHASES
#define VS_PHASE_BEGIN(var) unsigned long long var = wall_clock64()
#define VS_PHASE_STAMP(k, var) do { if (threadIdx.x == 0) { const unsigned long long tn_ = wall_clock64(); b.st[s].dbg[k] += tn_ - var; var = tn_; } } while (0)
#define VS_PHASE_COUNT(k) do { if (threadIdx.x == 0) b.st[s].dbg[k] += 1; } while (0)
#else
#define VS_PHASE_BEGIN(var) do { } while (0)
#define VS_PHASE_STAMP(k, var) do { } while (0)
#define VS_PHASE_COUNT(k) do { } while (0)
#endif

#define META 6
#define M_DIST 0
#define M_EPI 1
#define M_PREV 2
#define M_TLEN 3
#define M_LMUP 4
#define M_NEXT 5

struct PtView {  // frame points of stream s, buffer pb
  int16_t* kp; uint8_t* desc; int32_t* meta; double* cam; double* camlm; double* lm; int32_t* n; uint16_t* trail;
};
__device__ __forceinline__ PtView pts_of(const DevCfg& c, const DevBuf& b, int s, int pb) {
  const size_t o = ((size_t)s * 2 + pb) * c.MAXP;
  PtView v;
  v.kp = b.p_kp + o * 4; v.desc = b.p_desc + o * 64; v.meta = b.p_meta + o * META;
  v.cam = b.p_cam + o * 3; v.camlm = b.p_camlm + o * 3; v.lm = b.p_lm + o * 3; v.n = b.n_points + s * 2 + pb;
  v.trail = b.p_trail + (c.trail ? o * VS_TRAIL : (size_t)0);
  return v;
}
__device__ __forceinline__ double* hpose_of(const DevCfg& c, const DevBuf& b, int s, int f) {
  return b.h_pose + ((size_t)s * c.HCAP + (f % c.HCAP)) * 24;
}
__device__ __forceinline__ double* hcam_of(const DevCfg& c, const DevBuf& b, int s, int f) {
  return b.h_cam + ((size_t)s * c.HCAP + (f % c.HCAP)) * (size_t)c.MAXP * 4;   // x, y, z, 1 / z per point
}
__device__ __forceinline__ int32_t* hprev_of(const DevCfg& c, const DevBuf& b, int s, int f) {
  return b.h_prev + ((size_t)s * c.HCAP + (f % c.HCAP)) * (size_t)c.MAXP;
}
__device__ __forceinline__ int ld_relaxed(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// getPointInLeftCamera (stereo_framepoint_generator.cpp:871-895)
__device__ __forceinline__ void triangulate(const DevCfg& c, int xL, int yL, int xR, int yR, double* o) {
  const double bx = c.c.baseline_h[0], fx = c.c.K[0], fy = c.c.K[4], cx = c.c.K[2], cy = c.c.K[5];
  o[2] = bx / (double)(xR - xL);
  o[0] = 1 / fx * ((double)xL - cx) * o[2];
  o[1] = 1 / fy * ((double)(yL + yR) / 2.0 - cy) * o[2];
}

// ----------------------------------------------------------------------------------------------
// projection of previous point i into the current left image (track(), :494-513)
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ bool project_prev(const DevCfg& c, const double* T, const double* cam, double* uvw, int* row, int* col) {
  double q[3];
  tf_apply(T, cam, q);
  mat3_mul_vec(c.c.K, q, uvw);
  if (!(uvw[2] > 0)) return false;
  if (!to_int32(uvw[0] / uvw[2], col) || !to_int32(uvw[1] / uvw[2], row)) return false;
  if (*col < 0 || *col > c.c.cols || *row < 0 || *row > c.c.rows) return false;
  return true;
}

// triangulation-distance rule of initialize() (stereo_framepoint_generator.cpp:109-125); frame status = tracker status
__device__ __forceinline__ double tau_tri_rule(const DevCfg& c, int status, int n_left) {
  if (status == VSLAM_LOCALIZING) return fmin(0.1 * 256, c.c.maximum_matching_distance_triangulation);
  const double ratio = fmin((double)n_left / (double)c.target_kp, 1.0);
  return fmax(ratio * c.c.maximum_matching_distance_triangulation, 0.1 * 256);
}

// Per previous point i the wide candidate kernel leaves everything of track() that depends only on the motion prior:
//   proj[i]    = {row, col, n_candidates (-1: projection outside the image), |epipolar offset|,
//                 n_right (see below), x of the first left candidate, -, -}
//   proj_q[i]  = right-image projection (u/w, v/w) before the left-match correction (:541-556)
//   cand_key[i][0..15] = the left features inside the search window below the descriptor distance, as sorted keys
//                (primary << 16 | feature index): primary = Hamming distance (appearance mode) or squared pixel distance
//                (projection mode, < 10000 only).  Features are stored row-major, so ordering by index is the reference's
//                (row, col) tie-break and the first key whose feature is still present IS the reference's match.
//   cand_rkey[i][0..7] = for the FIRST left candidate (the match unless an earlier point removed it): the right features
//                its search (:541-590) can return, sorted by (distance, index) — the first one still present is the
//                reference's pick — as (reject << 31 | distance << 16 | index); reject = the pair fails the disparity or
//                the previous-right-descriptor gate (:597-608), i.e. the point is neither tracked nor lost.
//                n_right = -1: the right projection leaves the image (never tracked, never lost); 9 = more than 8.
// One group of 16 lanes per point (four points per wavefront: in Tracking state the window is ~21 rows): the lanes split
// the window rows; the row/cell CSR bounds each row to the 16-px cells the window overlaps; a rank sort through LDS
// orders the (at most 16) keys.  `lane` is the lane inside the group, `cw` the group's LDS slot.
#define VS_CGL 16
#define VS_MAXRCAND 8
struct CandWave {
  int cnt, rcnt;
  uint32_t keys[VS_MAXCAND];
  uint32_t rkeys[VS_MAXRCAND];
  int32_t kxy[VS_MAXCAND];          // x | y << 16 and descriptor of every left candidate: the right search of the first
  uint32_t kdesc[VS_MAXCAND][8];    // one starts from LDS instead of two more HBM round trips
};

__device__ __forceinline__ void candidates_wave(const DevCfg& c, const DevBuf& b, int s, int pb_prev, int i, int lane, CandWave* cw,
                                const double* T, int d, double tau, double tau_tri, int by_app) {
  const PtView pv = pts_of(c, b, s, pb_prev);
  const size_t gi = (size_t)s * c.MAXP + i;
  double uvw[3];
  int row, col;
  const bool ok = project_prev(c, T, pv.cam + 3 * (size_t)i, uvw, &row, &col);
  if (lane == 0) { cw->cnt = 0; cw->rcnt = 0; }
  if (!ok) {
    if (lane == 0) *reinterpret_cast<int4*>(b.proj + gi * 8) = make_int4(row, col, -1, 0);
    return;
  }
  const int rows = c.c.rows, cols = c.c.cols;
  const int r0 = max(row - d, 0), r1 = min(row + d + 1, rows);
  const int c0 = max(col - d, 0), c1 = min(col + d + 1, cols);
  const int kk = (int)fabs((double)pv.meta[(size_t)i * META + M_EPI]);
  uint32_t pd[8], prd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    pd[k] = reinterpret_cast<const uint32_t*>(pv.desc + (size_t)64 * i)[k];
    prd[k] = reinterpret_cast<const uint32_t*>(pv.desc + (size_t)64 * i + 32)[k];
  }
  const int32_t* rowcell = rowcell_of(c, b, s, 0);
  const int16_t* kxy = kpxy_of(c, b, s, 0);
  const uint8_t* desc = desc_of(c, b, s, 0);
  if (c1 > c0) {
    const int cl = c0 >> 4, ch = ((c1 - 1) >> 4) + 1;
    for (int r = r0 + lane; r < r1; r += VS_CGL) {
      const int lo = rowcell[(size_t)r * (c.CW + 1) + cl], hi = rowcell[(size_t)r * (c.CW + 1) + ch];
      for (int k = lo; k < hi; ++k) {
        // coordinates and descriptor in flight together (the descriptor of a feature outside the column range is wasted)
        const int32_t xy = *reinterpret_cast<const int32_t*>(kxy + 2 * k);
        uint32_t kd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kd[u] = reinterpret_cast<const uint32_t*>(desc + (size_t)32 * k)[u];
        const int x = (int16_t)(xy & 0xFFFF);
        if (x < c0 || x >= c1) continue;
        const int h = hamming32(pd, kd);
        if (!((double)h < tau)) continue;
        unsigned prim;
        if (by_app) prim = (unsigned)h;
        else { const int dr = row - r, dc = col - x; prim = (unsigned)(dr * dr + dc * dc); if (prim >= 10000u) continue; }
        const int slot = atomicAdd(&cw->cnt, 1);
        if (slot < VS_MAXCAND) {
          cw->keys[slot] = (prim << 16) | (unsigned)k;
          cw->kxy[slot] = xy;
#pragma unroll
          for (int u = 0; u < 8; ++u) cw->kdesc[slot][u] = kd[u];
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int cnt = __hip_atomic_load(&cw->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  uint32_t first = 0xFFFFFFFFu;   // smallest key == first left candidate
  int first_slot = 0;
  if (cnt <= VS_MAXCAND) {
    for (int j = 0; j < cnt; ++j) {
      const uint32_t kj = __hip_atomic_load(&cw->keys[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (kj < first) { first = kj; first_slot = j; }
    }
    if (lane < cnt) {
      const uint32_t mine = __hip_atomic_load(&cw->keys[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int rank = 0;
      for (int j = 0; j < cnt; ++j) rank += __hip_atomic_load(&cw->keys[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < mine ? 1 : 0;
      b.cand_key[gi * VS_MAXCAND + rank] = mine;
    }
  }
  double uR[3];
  for (int k = 0; k < 3; ++k) uR[k] = uvw[k] + c.c.baseline_h[k];
  const double qx = uR[0] / uR[2], qy = uR[1] / uR[2];
  // ---- right candidates of the first left candidate -------------------------------------------------------------------
  int n_right = 9, flx = 0;
  if (cnt >= 1 && cnt <= VS_MAXCAND) {
    const int fxy = __hip_atomic_load(&cw->kxy[first_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    flx = (int16_t)(fxy & 0xFFFF);
    const int fly = fxy >> 16;
    const float ex = (float)col - (float)flx, ey = (float)row - (float)fly;
    int colR, rowR;
    if (!to_int32(qx - ex, &colR) || !to_int32(qy - ey, &rowR) || colR < 0 || colR > cols || rowR < 0 || rowR > rows) {
      n_right = -1;
    } else {
      const int rr0 = max(rowR - kk, 0), rr1 = min(rowR + kk + 1, rows);
      const int rc0 = max(colR - d, 0), rc1 = min(colR + d + 1, flx);
      if (rc1 > rc0) {
        uint32_t ld[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ld[k] = __hip_atomic_load(&cw->kdesc[first_slot][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int32_t* rowcellR = rowcell_of(c, b, s, 1);
        const int16_t* kxyR = kpxy_of(c, b, s, 1);
        const uint8_t* descR = desc_of(c, b, s, 1);
        for (int r = rr0 + lane; r < rr1; r += VS_CGL) {
          const int lo = rowcellR[(size_t)r * (c.CW + 1) + (rc0 >> 4)], hi = rowcellR[(size_t)r * (c.CW + 1) + ((rc1 - 1) >> 4) + 1];
          for (int g = lo; g < hi; ++g) {
            const int gx = kxyR[2 * g];
            uint32_t gd[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) gd[k] = reinterpret_cast<const uint32_t*>(descR + (size_t)32 * g)[k];
            if (gx < rc0 || gx >= rc1) continue;
            const int h = hamming32(ld, gd);
            if (!((double)h < tau_tri)) continue;
            const bool rej = ((double)(flx - gx) < c.c.minimum_disparity_pixels) || ((double)hamming32(gd, prd) > tau);
            const int slot = atomicAdd(&cw->rcnt, 1);
            if (slot < VS_MAXRCAND) cw->rkeys[slot] = (rej ? 0x80000000u : 0u) | ((unsigned)h << 16) | (unsigned)g;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int rc = __hip_atomic_load(&cw->rcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      n_right = min(rc, 9);
      if (rc <= VS_MAXRCAND && lane < rc) {
        const uint32_t mine = __hip_atomic_load(&cw->rkeys[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int rank = 0;
        for (int j = 0; j < rc; ++j)
          rank += (__hip_atomic_load(&cw->rkeys[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 0x7FFFFFFFu) < (mine & 0x7FFFFFFFu) ? 1 : 0;
        b.cand_rkey[gi * VS_MAXRCAND + rank] = mine;
      }
    }
  }
  if (lane == 0) {
    *reinterpret_cast<int4*>(b.proj + gi * 8) = make_int4(row, col, cnt, kk);
    *reinterpret_cast<int4*>(b.proj + gi * 8 + 4) = make_int4(n_right, flx, 0, 0);
    b.proj_q[gi * 2] = qx; b.proj_q[gi * 2 + 1] = qy;
  }
  __builtin_amdgcn_wave_barrier();
}

// XCD-aware labels for (blocks-per-stream, streams) grids: workgroups are dealt round-robin over the 8 XCDs, so with the plain
// grid the blocks of ONE stream land on all eight and every L2 fetches that stream's descriptors, keypoints and row / cell CSR.
// Re-labelled, the gridDim.x blocks of stream s all run on XCD s mod 8 — the XCD its frame workgroup (block s of k_frame) and
// its two k_emit workgroups run on, whose L2 then already holds / will want the same lines.  Speed only, never correctness.
__device__ __forceinline__ void xcd_stream_block(int* bx, int* sy, int rot = 0) {
  const int gx = gridDim.x, ns = gridDim.y;
  const int lin = blockIdx.y * gx + blockIdx.x, per = (ns >> 3) * gx;      // blocks per XCD among the first 8 * (ns / 8) streams
  *bx = blockIdx.x; *sy = blockIdx.y;
  if (lin < (per << 3)) { const int x = (lin + rot) & 7, j = lin >> 3, q = j / gx; *sy = x + 8 * q; *bx = j - q * gx; }   // rot: dev_types.h xcd_rot
}

__global__ __launch_bounds__(256, 8) void k_track_candidates(const DevCfg c, const DevBuf b, int mode) {
  // mode < 0: fused path (appearance iff the tracker is Localizing, window forced to max in that case);
  // mode 0/1: stage path, window and distance exactly as set through vslam_set_tracker_state
  __shared__ CandWave cw[256 / VS_CGL];
  int bx, sy;
  xcd_stream_block(&bx, &sy, b.xcd_rot);
  const int s = b.s0 + sy;
  if (!vs_active(b, s)) return;
  const StreamState& st = b.st[s];
  if (!st.has_prev) return;
  const int lane = threadIdx.x % VS_CGL, w = threadIdx.x / VS_CGL;
  const int wave = bx * (256 / VS_CGL) + w, nwaves = gridDim.x * (256 / VS_CGL);
  const int pb_prev = st.cur;  // the previous frame's points: buffer that was current last frame
  const int P = b.n_points[s * 2 + pb_prev];
  const int by_app = mode < 0 ? (st.status == VSLAM_LOCALIZING) : mode;
  const int d = (mode < 0 && by_app) ? c.c.maximum_projection_tracking_distance_pixels : st.win;
  double T[12];
  for (int k = 0; k < 12; ++k) T[k] = st.prior[k];
  const double tau_tri = mode < 0 ? tau_tri_rule(c, st.status, b.n_kp[s * 2]) : st.tau_tri;
  for (int i = wave; i < P; i += nwaves) candidates_wave(c, b, s, pb_prev, i, lane, &cw[w], T, d, st.tau_track, tau_tri, by_app);
}

// ==============================================================================================
// frame kernel pieces (all called by the whole workgroup of stream s)
// ==============================================================================================
struct FrameShared {
  int scan[17];
  int flag;
  int n_trk, n_lost, n_lm, n_cur, n_cand, n_proj;
  int status, win, attempts, broken, fallback, aligner_ran;
  double tau_track;
  double prior[12];      // _previous_to_current_camera of the frame in flight
  double T[12];          // aligner estimate
  double Tprev[12];      // estimate the LAST round linearized at (VS_ALIGN_LDS builds recompute errors / inlier flags from it)
  double H[36];
  double bvec[6];
  double E, Eprev;
  int inl, outl, its, conv;
  double red4[VS_WG / 64][4][32];
  unsigned long long key;
};

__device__ __forceinline__ unsigned long long key3(unsigned a, int row, int col) {
  return ((unsigned long long)a << 32) | ((unsigned long long)(unsigned)row << 16) | (unsigned)col;
}

// "who removed which feature" + the right keypoint coordinates of the stream: in LDS when they fit the arena (the
// usual case), in HBM otherwise; same code through flat pointers
struct TrackTables {
  int32_t* killL; int32_t* killR;
  const uint16_t* xyR;     // [nR][2] x, y
};
__device__ __forceinline__ int ld_kill(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// evaluation of one previous point against the current kill state: what the sequential loop body
// of track() would do if all earlier points had the outcomes recorded in `kill`.
// out = fl, fr, dist, flag (bit0 success, bit1 lost-eligible), x of fl, y of fr (for the parallax clearing)
__device__ __forceinline__ void evaluate_point(const DevCfg& c, const DevBuf& b, int s, int pb_prev, int i, const TrackTables& tt, int d,
                               double tau_track, double tau_tri, int by_app, int* out) {
  const size_t gi = (size_t)s * c.MAXP + i;
  out[0] = -1; out[1] = -1; out[2] = 0; out[3] = 0; out[4] = 0; out[5] = 0;
  // everything whose address is known up front, in flight together
  const int4 pr = *reinterpret_cast<const int4*>(b.proj + gi * 8);
  const int4 pr2 = *reinterpret_cast<const int4*>(b.proj + gi * 8 + 4);
  uint4 rv[VS_MAXRCAND / 4];
#pragma unroll
  for (int k = 0; k < VS_MAXRCAND / 4; ++k) rv[k] = reinterpret_cast<const uint4*>(b.cand_rkey + gi * VS_MAXRCAND)[k];
  const double2 q = *reinterpret_cast<const double2*>(b.proj_q + gi * 2);
  uint4 kv[VS_MAXCAND / 4];
#pragma unroll
  for (int k = 0; k < VS_MAXCAND / 4; ++k) kv[k] = reinterpret_cast<const uint4*>(b.cand_key + gi * VS_MAXCAND)[k];
  const PtView pv = pts_of(c, b, s, pb_prev);
  uint32_t prd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) prd[k] = reinterpret_cast<const uint32_t*>(pv.desc + (size_t)64 * i + 32)[k];
  const int cnt = pr.z;
  if (cnt < 0) return;  // not in image: neither tracked nor lost (:508-513)
  const int row = pr.x, col = pr.y;
  const int16_t* kxyL = kpxy_of(c, b, s, 0);
  const uint8_t* descL = desc_of(c, b, s, 0);
  const uint8_t* descR = desc_of(c, b, s, 1);
  const int rows = c.c.rows, cols = c.c.cols, CW1 = c.CW + 1;
  // ---- left search: first surviving key -------------------------------------------------------
  int fl = -1;
  if (cnt <= VS_MAXCAND) {
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(kv);
#pragma unroll
    for (int k = 0; k < VS_MAXCAND; ++k) {
      const int f = (int)(keys[k] & 0xFFFFu);
      if (fl < 0 && k < cnt && ld_kill(tt.killL + f) >= i) fl = f;
    }
  } else {
    // candidate list overflowed: exact serial rescan of the window
    unsigned long long best = ~0ull;
    uint32_t pd[8];
    for (int k = 0; k < 8; ++k) pd[k] = reinterpret_cast<const uint32_t*>(pv.desc + (size_t)64 * i)[k];
    const int r0 = max(row - d, 0), r1 = min(row + d + 1, rows), c0 = max(col - d, 0), c1 = min(col + d + 1, cols);
    const int32_t* rowcell = rowcell_of(c, b, s, 0);
    if (c1 > c0)
      for (int r = r0; r < r1; ++r) {
        const int lo = rowcell[(size_t)r * CW1 + (c0 >> 4)], hi = rowcell[(size_t)r * CW1 + ((c1 - 1) >> 4) + 1];
        for (int f = lo; f < hi; ++f) {
          const int fx = kxyL[2 * f];
          if (fx < c0 || fx >= c1) continue;
          if (ld_kill(tt.killL + f) < i) continue;
          const int h = hamming32(pd, reinterpret_cast<const uint32_t*>(descL + (size_t)32 * f));
          if (!((double)h < tau_track)) continue;
          unsigned prim;
          if (by_app) prim = (unsigned)h;
          else { const int dr = row - r, dc = col - fx; prim = (unsigned)(dr * dr + dc * dc); if (prim >= 10000u) continue; }
          const unsigned long long key = key3(prim, r, fx);
          if (key < best) { best = key; fl = f; }
        }
      }
  }
  if (fl < 0) { out[3] = 2; return; }  // no left match: lost-eligible
  // ---- right search, usual case: the first left candidate survived and its right candidates are tabulated -------------
  if (cnt <= VS_MAXCAND && fl == (int)(reinterpret_cast<const uint32_t*>(kv)[0] & 0xFFFFu) && pr2.x <= VS_MAXRCAND) {
    if (pr2.x < 0) return;                       // right projection outside the image: continue (:553-556)
    const uint32_t* rk = reinterpret_cast<const uint32_t*>(rv);
    uint32_t pick = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < VS_MAXRCAND; ++k)
      if (pick == 0xFFFFFFFFu && k < pr2.x && ld_kill(tt.killR + (rk[k] & 0xFFFFu)) >= i) pick = rk[k];
    if (pick == 0xFFFFFFFFu) { out[3] = 2; return; }   // no right match: lost-eligible
    if (pick & 0x80000000u) return;                      // disparity / previous-right-descriptor gate: continue
    const int fr = (int)(pick & 0xFFFFu);
    out[0] = fl; out[1] = fr; out[2] = (int)((pick >> 16) & 0x7FFFu); out[3] = 1; out[4] = pr2.y; out[5] = tt.xyR[2 * fr + 1];
    return;
  }
  // ---- right search (:541-590) --------------------------------------------------------------
  const int fxy = *reinterpret_cast<const int32_t*>(kxyL + 2 * fl);
  uint32_t ld[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ld[k] = reinterpret_cast<const uint32_t*>(descL + (size_t)32 * fl)[k];
  const int flx = (int16_t)(fxy & 0xFFFF), fly = fxy >> 16;
  const float ex = (float)col - (float)flx, ey = (float)row - (float)fly;
  int colR, rowR;
  if (!to_int32(q.x - ex, &colR) || !to_int32(q.y - ey, &rowR)) return;
  if (colR < 0 || colR > cols || rowR < 0 || rowR > rows) return;
  const int kk = pr.w;
  const int rr0 = max(rowR - kk, 0), rr1 = min(rowR + kk + 1, rows);
  const int rc0 = max(colR - d, 0), rc1 = min(colR + d + 1, flx);
  double dbest = tau_tri;
  int fr = -1;
  uint32_t rd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) rd[k] = 0;
  if (rc1 > rc0) {
    const int32_t* rowcellR = rowcell_of(c, b, s, 1);
    for (int r = rr0; r < rr1; ++r) {
      const int lo = rowcellR[(size_t)r * CW1 + (rc0 >> 4)], hi = rowcellR[(size_t)r * CW1 + ((rc1 - 1) >> 4) + 1];
      for (int g = lo; g < hi; ++g) {
        const int gx = tt.xyR[2 * g];
        if (gx < rc0 || gx >= rc1) continue;
        if (ld_kill(tt.killR + g) < i) continue;
        uint32_t gd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gd[k] = reinterpret_cast<const uint32_t*>(descR + (size_t)32 * g)[k];
        const double h = (double)hamming32(ld, gd);
        if (h < dbest) {
          dbest = h; fr = g;
#pragma unroll
          for (int k = 0; k < 8; ++k) rd[k] = gd[k];
        }
      }
    }
  }
  if (fr < 0) { out[3] = 2; return; }  // no right match: lost-eligible
  const int frx = tt.xyR[2 * fr];
  if ((double)(flx - frx) < c.c.minimum_disparity_pixels) return;  // continue: not lost (:597-600)
  if ((double)hamming32(rd, prd) > tau_track) return;               // continue (:603-608)
  out[0] = fl; out[1] = fr; out[2] = (int)dbest; out[3] = 1; out[4] = flx; out[5] = tt.xyR[2 * fr + 1];
}

// Order-exact resolution of track(): Jacobi iteration on "who removed which lattice cell".
// kill[f] = smallest index of a previous point whose (tentative) success removes feature f; point i
// sees f as present iff kill[f] >= i.  A fixed point equals the sequential result (induction on i:
// point 0 never depends on others; if all j < i are final, the kills i sees are final).
__device__ __forceinline__ void wg_track_resolve(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int pb_prev, unsigned char* arena,
                                 int d, double tau_track, double tau_tri, int by_app) {
  const int tid = threadIdx.x;
  const PtView pv = pts_of(c, b, s, pb_prev);
  const int P = *pv.n;
  const int nL = b.n_kp[s * 2], nR = b.n_kp[s * 2 + 1];
  const int16_t* kxyR = kpxy_of(c, b, s, 1);
  TrackTables tt;
  const bool in_lds = (size_t)nL * 4 + (size_t)nR * 8 <= VS_ARENA;
  if (in_lds) {
    tt.killL = reinterpret_cast<int32_t*>(arena);
    tt.killR = tt.killL + nL;
    uint32_t* xy = reinterpret_cast<uint32_t*>(tt.killR + nR);
    for (int g = tid; g < nR; g += VS_WG) xy[g] = reinterpret_cast<const uint32_t*>(kxyR)[g];
    tt.xyR = reinterpret_cast<const uint16_t*>(xy);
  } else {
    tt.killL = kill_of(c, b, s, 0);
    tt.killR = kill_of(c, b, s, 1);
    tt.xyR = reinterpret_cast<const uint16_t*>(kxyR);
  }
  int32_t* res = b.res + (size_t)s * c.MAXP * 8;
  for (int i = tid; i < P; i += VS_WG) { res[8 * i] = -1; res[8 * i + 1] = -1; res[8 * i + 2] = 0; res[8 * i + 3] = 0; }
  __syncthreads();
  for (int iter = 0; iter <= P + 1; ++iter) {
    VS_PHASE_COUNT(7);
    for (int f = tid; f < nL; f += VS_WG) __hip_atomic_store(tt.killL + f, 0x7FFFFFFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int f = tid; f < nR; f += VS_WG) __hip_atomic_store(tt.killR + f, 0x7FFFFFFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (iter > 0) {
      for (int i = tid; i < P; i += VS_WG) {
        const int4 r4 = *reinterpret_cast<const int4*>(res + 8 * i);
        if (r4.w & 1) {
          const int fl = r4.x, fr = r4.y;
          const int lim = res[8 * i + 4], rowR = res[8 * i + 5];
          atomicMin(tt.killL + fl, i);
          atomicMin(tt.killR + fr, i);
          // parallax clearing (:612-621): right features strictly between fR.col and fL.col on fR.row
          for (int g = fr + 1; g < nR && tt.xyR[2 * g + 1] == rowR && tt.xyR[2 * g] < lim; ++g) atomicMin(tt.killR + g, i);
        }
      }
      __syncthreads();
    }
    int changed = 0;
    for (int i = tid; i < P; i += VS_WG) {
      int o[6];
      evaluate_point(c, b, s, pb_prev, i, tt, d, tau_track, tau_tri, by_app, o);
      const int4 old = *reinterpret_cast<const int4*>(res + 8 * i);
      if (o[0] != old.x || o[1] != old.y || o[3] != old.w) changed = 1;
      *reinterpret_cast<int4*>(res + 8 * i) = make_int4(o[0], o[1], o[2], o[3]);
      res[8 * i + 4] = o[4]; res[8 * i + 5] = o[5];
    }
    if (!__syncthreads_or(changed)) break;
  }
  // used flags == every feature some final success removed (matched_indices_* + prune, :646-672)
  uint8_t* usedL = used_of(c, b, s, 0);
  uint8_t* usedR = used_of(c, b, s, 1);
  for (int f = tid; f < nL; f += VS_WG) usedL[f] = ld_kill(tt.killL + f) != 0x7FFFFFFF;
  for (int f = tid; f < nR; f += VS_WG) usedR[f] = ld_kill(tt.killR + f) != 0x7FFFFFFF;
  // compaction in previous-point order: tracked list, lost list, landmark count
  int32_t* trk = b.trk + (size_t)s * c.MAXP * 4;
  int32_t* lost = b.lost + (size_t)s * c.MAXP;
  const int per = (P + VS_WG - 1) / VS_WG;
  const int i0 = tid * per, i1 = min(i0 + per, P);
  int nt = 0, nl = 0, nlm = 0;
  for (int i = i0; i < i1; ++i) {
    const int fl = res[8 * i + 3];
    if (fl & 1) { ++nt; if (pv.meta[(size_t)i * META + M_LMUP] > 0) ++nlm; }
    else if ((fl & 2) && !pv.meta[(size_t)i * META + M_NEXT]) ++nl;
  }
  int tot_t, tot_l, tot_lm;
  int ot = block_exclusive_scan(nt, sh.scan, &tot_t);
  int ol = block_exclusive_scan(nl, sh.scan, &tot_l);
  block_exclusive_scan(nlm, sh.scan, &tot_lm);
  for (int i = i0; i < i1; ++i) {
    const int fl = res[8 * i + 3];
    if (fl & 1) {
      trk[4 * ot] = i; trk[4 * ot + 1] = res[8 * i]; trk[4 * ot + 2] = res[8 * i + 1]; trk[4 * ot + 3] = res[8 * i + 2];
      ++ot;
      pv.meta[(size_t)i * META + M_NEXT] = 1;
    } else if ((fl & 2) && !pv.meta[(size_t)i * META + M_NEXT]) {
      lost[ol++] = i;
    }
  }
  if (tid == 0) { sh.n_trk = tot_t; sh.n_lost = tot_l; sh.n_lm = tot_lm; }
  __syncthreads();
}

// ----------------------------------------------------------------------------------------------
// StereoUVAligner
// ----------------------------------------------------------------------------------------------
// Symmetric elimination without pivoting (LDL^T) of the damped normal equations, every lane redundantly, all
// indices static (registers only).  H is symmetric positive definite whenever the alignment is well posed, where
// unpivoted elimination is backward stable, so the solution equals the reference's full-pivot LU up to rounding
// (same class of difference as the H,b summation order).  Returns false when a pivot is not safely positive; the
// caller then falls back to the exact full-pivot wave solver (rank-deficient systems keep reference semantics).
__device__ __forceinline__ bool ldlt_solve6(const double* Hs, const double* rhs, double* x) {
  // upper triangle only (21 entries; the eliminated matrix stays symmetric, so A[i][k] below the diagonal is A[k][i])
  double A[6][6], y[6], invd[6];
  double dmax = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = i; j < 6; ++j) A[i][j] = Hs[6 * i + j];
    y[i] = rhs[i];
    dmax = fmax(dmax, fabs(A[i][i]));
  }
  bool ok = dmax > 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double d = A[k][k];
    if (!(d > 1e-12 * dmax)) ok = false;
    const double inv = 1.0 / d;
    invd[k] = inv;
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const double f = A[k][i] * inv;
#pragma unroll
      for (int j = i; j < 6; ++j) A[i][j] -= f * A[k][j];
      y[i] -= f * y[k];
    }
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double sv = y[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) sv -= A[i][j] * x[j];
    x[i] = sv * invd[i];
  }
  return ok;
}

// Eigen::FullPivLU<Matrix6>::solve on one wavefront.  Element (row i, col j) of the matrix sits in lane
// 6*j+i (column-major, so "first strict maximum in column-major order" is the lowest set bit of a ballot),
// the right-hand side in lanes 36..41.  Arithmetic per element is the serial algorithm's (same operands,
// same order), so the result is bit-identical to dev_math.h full_piv_solve<6> / the CPU oracle.
__device__ __forceinline__ void wave_solve6(double a, int lane, double* x, unsigned long long* lds_key) {
  const int j = lane / 6, i = lane - 6 * j;
  const int row = lane < 36 ? i : (lane < 42 ? lane - 36 : -1);
  unsigned perm = 0x543210u;
  int rank = 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (rank == 6) {
      const bool elig = lane < 36 && i >= k && j >= k;
      const double v = elig ? fabs(a) : -1.0;
      // maximum of |a| over the remaining corner: the bit pattern of a non-negative double orders like an integer
      if (lane == 0) *lds_key = 0ull;
      __builtin_amdgcn_wave_barrier();
      if (elig) atomicMax(lds_key, (unsigned long long)__double_as_longlong(v));
      __builtin_amdgcn_wave_barrier();
      const double m = __longlong_as_double((long long)__hip_atomic_load(lds_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      __builtin_amdgcn_wave_barrier();
      if (!(m > 0)) {
        rank = k;
      } else {
        const unsigned long long bal = __ballot(elig && v == m);
        const int pl = __ffsll((long long)bal) - 1;
        const int pc = pl / 6, pr = pl - 6 * pc;
        if (pr != k) {
          int src = lane;
          if (row == k) src = lane - k + pr; else if (row == pr) src = lane - pr + k;
          a = __shfl(a, src, 64);
        }
        if (pc != k) {
          int src = lane;
          if (lane < 36) { if (j == k) src = pc * 6 + i; else if (j == pc) src = k * 6 + i; }
          a = __shfl(a, src, 64);
          const unsigned pk = (perm >> (4 * k)) & 15u, pp = (perm >> (4 * pc)) & 15u;
          perm = (perm & ~((15u << (4 * k)) | (15u << (4 * pc)))) | (pp << (4 * k)) | (pk << (4 * pc));
        }
        const double akk = __shfl(a, k * 6 + k, 64);
        const double aik = __shfl(a, k * 6 + (row < 0 ? 0 : row), 64);
        const double akj = __shfl(a, lane < 36 ? j * 6 + k : 36 + k, 64);
        if (row > k) {
          const double f = aik / akk;
          if (lane < 36) { if (j == k) a = 0; else if (j > k) a = a - f * akj; }
          else a = a - f * akj;
        }
      }
    }
  }
  double y[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int ii = 5; ii >= 0; --ii) {
    double sv = __shfl(a, 36 + ii, 64);
#pragma unroll
    for (int jj = ii + 1; jj < 6; ++jj) {
      const double aij = __shfl(a, jj * 6 + ii, 64);
      if (jj < rank) sv -= aij * y[jj];
    }
    const double aii = __shfl(a, ii * 6 + ii, 64);
    if (ii < rank) y[ii] = sv / aii;
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) x[q] = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int dst = (int)((perm >> (4 * k)) & 15u);
#pragma unroll
    for (int q = 0; q < 6; ++q) if (dst == q) x[q] = y[k];
  }
}

#define NACC 29   // 21 upper-triangular H + 6 b + E + inlier count
#define VS_ALCACHE 1   // measurements per thread kept in registers across rounds (the first chunk of VS_WG measurements)

struct AlignPoint { double m[3], f[4], om, wt; };

// row-wise (16-lane) inclusive add by DPP row_shr 1,2,4,8: lanes 15,31,47,63 end up with their row's sum
__device__ __forceinline__ double dpp_row_sum(double v) {
#define VS_DPP_STEP(ctrl)                                                                      \
  {                                                                                            \
    const long long bits = __double_as_longlong(v);                                           \
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xFFFFFFFFll), ctrl, 0xF, 0xF, true); \
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), ctrl, 0xF, 0xF, true);   \
    v += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);                           \
  }
  VS_DPP_STEP(0x111) VS_DPP_STEP(0x112) VS_DPP_STEP(0x114) VS_DPP_STEP(0x118)
#undef VS_DPP_STEP
  return v;
}

// One measurement of the linearization (StereoUVAligner::linearize body, stereouv_aligner.cpp:81-185; UVDAligner,
// uvd_aligner.cpp:78-171) as its residual rows: Jacobian rows J (4 x 6, stereo; 3 x 6, RGB-D), residual e, row
// weights (w_uv for the image rows, w_d for the depth row of the RGB-D model) and its share of E / the inlier count.
// A measurement that is skipped, ignored or out of range (`have` false) gets zero weights and FINITE rows, so that every
// lane can take part in the wave-wide reduction of the products without branches.
struct AlignRows { double J[4][6]; double e[4]; double w_uv, w_d, E, cnt; };

template <bool UVD>
__device__ __forceinline__ void align_rows(const DevCfg& c, const double* T, const AlignPoint& P, bool have, bool ignore_outliers,
                                           AlignRows& R, double* chi_out, uint8_t* inl_out) {
  const double* K = c.c.K;
  const bool pinhole = K[1] == 0 && K[3] == 0 && K[6] == 0 && K[7] == 0 && K[8] == 1;   // uniform
  double chi_w = -1;
  uint8_t inl_w = 0;
  double w_uv = P.om, w_d = UVD ? P.f[3] : 0.0;
  double p[3];
  tf_apply(T, P.m, p);
  bool skip = !have || (UVD ? p[2] <= c.c.minimum_depth_meters : p[2] < c.c.minimum_depth_meters);
  double aL[3], aR[3];
  mat3_mul_vec(K, p, aL);
  for (int k = 0; k < 3; ++k) aR[k] = aL[k] + c.c.baseline_h[k];
  const double cL = aL[2], cR = aR[2];
  const double uL = aL[0] / cL, vL = aL[1] / cL, uR = aR[0] / cR, vR = aR[1] / cR;
  if (!skip) {
    if (uL < 0 || uL > c.c.cols || vL < 0 || vL > c.c.rows) skip = true;
    if (!UVD && (uR < 0 || uR > c.c.cols || vR < 0 || vR > c.c.rows)) skip = true;
  }
  bool use = false;
  R.E = 0; R.cnt = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) R.e[k] = 0;
  if (!skip) {
    double chi;
    if constexpr (UVD) {
      R.e[0] = uL - P.f[0]; R.e[1] = vL - P.f[1]; R.e[2] = p[2] - P.f[2];
      chi = ((R.e[0] * w_uv) * R.e[0] + (R.e[1] * w_uv) * R.e[1]) + (R.e[2] * w_d) * R.e[2];
    } else {
      R.e[0] = uL - P.f[0]; R.e[1] = vL - P.f[1]; R.e[2] = uR - P.f[2]; R.e[3] = vR - P.f[3];
      chi = w_uv * (((R.e[0] * R.e[0] + R.e[1] * R.e[1]) + R.e[2] * R.e[2]) + R.e[3] * R.e[3]);
    }
    chi_w = chi;
    use = true;
    if (chi > c.c.aligner_maximum_error_kernel) {
      if (ignore_outliers) use = false;
      else { const double sc = c.c.aligner_maximum_error_kernel / chi; w_uv *= sc; w_d *= sc; }
    } else {
      inl_w = 1;
      R.cnt = 1.0;
    }
    if (use) R.E = chi;
  }
  *chi_out = chi_w;
  *inl_out = inl_w;
  R.w_uv = use ? w_uv : 0.0;
  R.w_d = use ? w_d : 0.0;
  // rows: finite for every lane (unused measurements: unit depths, their weights are zero)
  const double cLs = use ? cL : 1.0, cRs = use ? cR : 1.0;
  {
    // The Jacobian products (and the H, b products formed from them) use explicit fused multiply-adds: their sums
    // already differ from the reference's serial order by rounding (parallel reduction), a gate-free part of the
    // computation.  Explicit fma() rather than a contraction pragma: every instantiation of this code (fused kernel,
    // stage kernel, stand-alone aligner) rounds identically.  Everything that feeds a comparison (projection, chi, the
    // kernel test above) stays unfused like the oracle.
    const double wt = P.wt;
    // K * [w*I3 | -2*skew(p)]
    const double Jt[3][6] = {{wt, 0, 0, 0, 2 * p[2], -2 * p[1]}, {0, wt, 0, -2 * p[2], 0, 2 * p[0]}, {0, 0, wt, 2 * p[1], -2 * p[0], 0}};
    double KJ[3][6];
    if (pinhole) {
      // K = [fx 0 cx; 0 fy cy; 0 0 1]: the products with the structural zeros of K and Jt are exact zeros, and adding an
      // exact zero is exact, so dropping them leaves every KJ entry bit-identical to the full triple product
      const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
      KJ[0][0] = fx * Jt[0][0]; KJ[0][1] = 0;                KJ[0][2] = cx * Jt[2][2];
      KJ[0][3] = cx * Jt[2][3];  KJ[0][4] = fma(fx, Jt[0][4], cx * Jt[2][4]); KJ[0][5] = fx * Jt[0][5];
      KJ[1][0] = 0;              KJ[1][1] = fy * Jt[1][1];  KJ[1][2] = cy * Jt[2][2];
      KJ[1][3] = fma(fy, Jt[1][3], cy * Jt[2][3]); KJ[1][4] = cy * Jt[2][4]; KJ[1][5] = fy * Jt[1][5];
#pragma unroll
      for (int j = 0; j < 6; ++j) KJ[2][j] = Jt[2][j];
    } else {
      // general K: the products with the structural zeros of Jt are exact zeros and are left out (x + 0 == x), which also
      // keeps a dozen loop-invariant K * 0 values from being hoisted in front of the round loop and held in registers
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double k0 = K[3 * i], k1 = K[3 * i + 1], k2 = K[3 * i + 2];
        KJ[i][0] = k0 * Jt[0][0]; KJ[i][1] = k1 * Jt[1][1]; KJ[i][2] = k2 * Jt[2][2];
        KJ[i][3] = k1 * Jt[1][3] + k2 * Jt[2][3];
        KJ[i][4] = k0 * Jt[0][4] + k2 * Jt[2][4];
        KJ[i][5] = k0 * Jt[0][5] + k1 * Jt[1][5];
      }
    }
    if constexpr (UVD) {
      const double iz = 1 / (use ? p[2] : 1.0), iz2 = iz * iz;
      const double j0 = -aL[0] * iz2, j1 = -aL[1] * iz2;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        R.J[0][j] = fma(iz, KJ[0][j], j0 * KJ[2][j]);
        R.J[1][j] = fma(iz, KJ[1][j], j1 * KJ[2][j]);
        R.J[2][j] = KJ[2][j];
        R.J[3][j] = 0;
      }
    } else {
      const double icL = 1 / cLs, icR = 1 / cRs, icL2 = icL * icL, icR2 = icR * icR;
      const double jl0 = -aL[0] * icL2, jl1 = -aL[1] * icL2, jr0 = -aR[0] * icR2, jr1 = -aR[1] * icR2;
      // rows of the projection Jacobian times KJ; the reference's explicit 0 * x terms are exact zeros (finite x)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        R.J[0][j] = fma(icL, KJ[0][j], jl0 * KJ[2][j]);
        R.J[1][j] = fma(icL, KJ[1][j], jl1 * KJ[2][j]);
        R.J[2][j] = fma(icR, KJ[0][j], jr0 * KJ[2][j]);
        R.J[3][j] = fma(icR, KJ[1][j], jr1 * KJ[2][j]);
      }
    }
  }
}

// accumulator q of the measurement: 0..20 upper triangle of H = J^T W J (row-major), 21..26 b = J^T W e, 27 E, 28 inliers
template <bool UVD, int Q>
__device__ __forceinline__ double align_entry(const AlignRows& R) {
  if constexpr (Q < 21) {
    constexpr int r = Q < 6 ? 0 : Q < 11 ? 1 : Q < 15 ? 2 : Q < 18 ? 3 : Q < 20 ? 4 : 5;
    constexpr int first = r == 0 ? 0 : r == 1 ? 6 : r == 2 ? 11 : r == 3 ? 15 : r == 4 ? 18 : 20;
    constexpr int cc = r + (Q - first);
    if constexpr (UVD) return fma(R.w_uv, fma(R.J[1][r], R.J[1][cc], R.J[0][r] * R.J[0][cc]), R.w_d * (R.J[2][r] * R.J[2][cc]));
    else return R.w_uv * fma(R.J[3][r], R.J[3][cc], fma(R.J[2][r], R.J[2][cc], fma(R.J[1][r], R.J[1][cc], R.J[0][r] * R.J[0][cc])));
  } else if constexpr (Q < 27) {
    constexpr int r = Q - 21;
    if constexpr (UVD) return fma(R.w_uv, fma(R.J[1][r], R.e[1], R.J[0][r] * R.e[0]), R.w_d * (R.J[2][r] * R.e[2]));
    else return R.w_uv * fma(R.J[3][r], R.e[3], fma(R.J[2][r], R.e[2], fma(R.J[1][r], R.e[1], R.J[0][r] * R.e[0])));
  } else if constexpr (Q == 27) {
    return R.E;
  } else {
    return R.cnt;
  }
}

// four accumulators Q0 .. Q0+3 of every lane's measurement, summed over each 16-lane row as a REDUCE-SCATTER: inside a quad the lanes first
// exchange halves (quad_perm [1,0,3,2]: even lanes keep accumulators 0 and 1 and hand over 2 and 3, odd lanes the reverse), then halves of that
// (quad_perm [2,3,0,1]), so that every lane ends up with ONE accumulator's quad total — 3 transfers instead of 8 —, and only that one value goes
// through row_shr 4 and 8.  5 value transfers per group instead of 16 (round 2: every accumulator through row_shr 1, 2, 4, 8).  Lanes 12 .. 15 of
// a row hold the row totals of accumulators 0, 2, 1, 3 and put them into LDS (first chunk of measurements) or add them to what is there.
#define VS_DPP_F64(dst, src, ctrl)                                                                                   \
  {                                                                                                                  \
    const long long bits_ = __double_as_longlong(src);                                                               \
    const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(bits_ & 0xFFFFFFFFll), ctrl, 0xF, 0xF, true);               \
    const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(bits_ >> 32), ctrl, 0xF, 0xF, true);                        \
    dst = __longlong_as_double(((long long)hi_ << 32) | (unsigned)lo_);                                              \
  }
template <bool UVD, int Q0>
__device__ __forceinline__ void align_reduce4(const AlignRows& R, double (*red)[32], int lane, bool first_chunk) {
  double v[4];
  v[0] = align_entry<UVD, Q0>(R);
  v[1] = Q0 + 1 < NACC ? align_entry<UVD, (Q0 + 1 < NACC ? Q0 + 1 : 0)>(R) : 0.0;
  v[2] = Q0 + 2 < NACC ? align_entry<UVD, (Q0 + 2 < NACC ? Q0 + 2 : 0)>(R) : 0.0;
  v[3] = Q0 + 3 < NACC ? align_entry<UVD, (Q0 + 3 < NACC ? Q0 + 3 : 0)>(R) : 0.0;
  const bool odd = lane & 1, hi = lane & 2;
  // lanes l and l ^ 1: the even one collects accumulators 0, 1, the odd one 2, 3
  const double keep0 = odd ? v[2] : v[0], keep1 = odd ? v[3] : v[1], send0 = odd ? v[0] : v[2], send1 = odd ? v[1] : v[3];
  double r0, r1;
  VS_DPP_F64(r0, send0, 0xB1) VS_DPP_F64(r1, send1, 0xB1)       // quad_perm [1,0,3,2]
  const double a0 = keep0 + r0, a1 = keep1 + r1;
  // lanes l and l ^ 2: the lower one keeps the first of its two, the upper one the second
  const double keep = hi ? a1 : a0, send = hi ? a0 : a1;
  double r2;
  VS_DPP_F64(r2, send, 0x4E)                                     // quad_perm [2,3,0,1]
  double q = keep + r2;                                           // quad total of accumulator {0, 2, 1, 3}[lane & 3]
  double t;
  VS_DPP_F64(t, q, 0x114) q += t;                                 // row_shr 4 (lanes 0 .. 3 of the row receive zero)
  VS_DPP_F64(t, q, 0x118) q += t;                                 // row_shr 8
  if ((lane & 15) >= 12) {
    const int idx = Q0 + ((lane & 1) << 1) + ((lane >> 1) & 1);   // lanes 12, 13, 14, 15 -> accumulators 0, 2, 1, 3
    if (idx < NACC) { if (first_chunk) red[lane >> 4][idx] = q; else red[lane >> 4][idx] += q; }
  }
}
#undef VS_DPP_F64

__device__ __forceinline__ void load_align_point(const DevCfg& c, const DevBuf& b, int s, int u, AlignPoint& P) {
  const double* moving = b.al_moving + (size_t)s * c.MAXP * 3;
  const double* fixed = b.al_fixed + (size_t)s * c.MAXP * 4;
  for (int k = 0; k < 3; ++k) P.m[k] = moving[3 * (size_t)u + k];
  for (int k = 0; k < 4; ++k) P.f[k] = fixed[4 * (size_t)u + k];
  P.om = (b.al_omega + (size_t)s * c.MAXP)[u];
  P.wt = (b.al_weight + (size_t)s * c.MAXP)[u];
}

// second half of a round: totals of the 29 sums, damping, 6 x 6 solve, pose update, re-orthonormalisation (oneRound, :190-207)
__device__ __forceinline__ void wg_round_solve(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int n, const double* T) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  VS_PHASE_BEGIN(tp0);
  if (w == 0) {
    // lane k < 29 owns total k; the 6x6 system then lives one element per lane (column-major: lane = 6*col+row,
    // right-hand side in lanes 36..41) and is solved by wave_solve6 without leaving registers.
    double tot = 0;
    const int nw = min((n + 63) >> 6, VS_WG / 64);
    if (lane < NACC) for (int ww = 0; ww < nw; ++ww) tot += ((sh.red4[ww][0][lane] + sh.red4[ww][1][lane]) + sh.red4[ww][2][lane]) + sh.red4[ww][3][lane];
    const int j = lane / 6, i = lane - 6 * j;
    int src = 0;
    if (lane < 36) { const int r = min(i, j), cc = max(i, j); src = r * 6 - (r * (r - 1)) / 2 + (cc - r); }
    else if (lane < 42) src = 21 + (lane - 36);
    double a = __shfl(tot, src, 64);
    if (lane < 36 && i == j) a += c.c.aligner_damping * n;   // oneRound (:196)
    if (lane >= 36 && lane < 42) a = -a;
    if (lane < 36) sh.H[6 * i + j] = a;
    if (lane == 27) sh.E = tot;
    if (lane == 28) { sh.inl = (int)tot; sh.outl = n - (int)tot; }
    if (lane >= 36 && lane < 42) sh.bvec[lane - 36] = a;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    double dx[6];
    VS_PHASE_STAMP(9, tp0);
    if (!ldlt_solve6(sh.H, sh.bvec, dx)) {
#ifdef VS_PROFILE_PHASES
      if (lane == 0) b.st[s].dbg[7] += 100;   // counts the full-pivot fallbacks (reads as 1 "us" each in vslam_debug_ticks)
#endif
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));   // opaque: the fallback's lane-index arithmetic stays here instead of being hoisted out of the round loop
      wave_solve6(a, lane_o, dx, &sh.key);
    }
    VS_PHASE_STAMP(10, tp0);
    if (lane == 0) {
      double D[12], Tn[12];
      v2t(dx, D);
      tf_mul(D, T, Tn);
      double R[9], RtR[9];
#pragma unroll
      for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) R[3 * ii + jj] = Tn[4 * ii + jj];
#pragma unroll
      for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          RtR[3 * ii + jj] = (R[ii] * R[jj] + R[3 + ii] * R[3 + jj]) + R[6 + ii] * R[6 + jj];
          if (ii == jj) RtR[3 * ii + jj] -= 1;
        }
#pragma unroll
      for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
          Tn[4 * ii + jj] = R[3 * ii + jj] - 0.5 * ((R[3 * ii] * RtR[jj] + R[3 * ii + 1] * RtR[3 + jj]) + R[3 * ii + 2] * RtR[6 + jj]);
#pragma unroll
      for (int k = 0; k < 12; ++k) { sh.Tprev[k] = T[k]; sh.T[k] = Tn[k]; }
      ++sh.its;
    }
  }
  __syncthreads();
  VS_PHASE_STAMP(11, tp0);
}

template <bool UVD>
__device__ __forceinline__ void wg_one_round(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int n, bool ignore_outliers,
                             const AlignPoint* cache, double* chi_reg, uint8_t* inl_reg) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double* chi_o = b.al_chi + (size_t)s * c.MAXP;
  uint8_t* inl_o = b.al_inl + (size_t)s * c.MAXP;
  VS_PHASE_BEGIN(tp0);
  double T[12];
  for (int k = 0; k < 12; ++k) T[k] = sh.T[k];
  // one measurement per thread and chunk: its rows, then the 29 products reduced four at a time — nothing but the rows
  // stays live (29 fp64 accumulators per thread were what pushed the kernel to 256 VGPRs and into scratch)
  for (int base = 0; base < n; base += VS_WG) {
    const int u = base + tid;
    const bool have = u < n;
    AlignRows R;
    // a wavefront without a measurement in this chunk (M ~ 240 of 512 lanes: half of them) skips the rows as well as the sums: it
    // would add exact zeros, and its ~600 fp64 instructions would share a SIMD's issue with a wavefront that does have work
#ifdef VS_ALIGN_NO_WAVE_SKIP      // probe builds: every wavefront evaluates the rows (round 3 behaviour)
    const bool wave_has = true;
#else
    const bool wave_has = base + w * 64 < n;
#endif
    if (!wave_has) {
      if (base == 0) { chi_reg[0] = -1; inl_reg[0] = 0; }
    } else if (base == 0) {
      align_rows<UVD>(c, T, cache[0], have, ignore_outliers, R, &chi_reg[0], &inl_reg[0]);   // stored after the last round
    } else {
      AlignPoint P;
      P.m[0] = P.m[1] = P.m[2] = 0; P.f[0] = P.f[1] = P.f[2] = P.f[3] = 0; P.om = 0; P.wt = 0;
      if (have) load_align_point(c, b, s, u, P);
      double chi_w; uint8_t inl_w;
      align_rows<UVD>(c, T, P, have, ignore_outliers, R, &chi_w, &inl_w);
      if (have) { chi_o[u] = chi_w; inl_o[u] = inl_w; }
    }
    if (base + w * 64 < n) {   // waves without a measurement in this chunk would add exact zeros
      double (*red)[32] = sh.red4[w];
      const bool fc = base == 0;
      align_reduce4<UVD, 0>(R, red, lane, fc);  align_reduce4<UVD, 4>(R, red, lane, fc);  align_reduce4<UVD, 8>(R, red, lane, fc);
      align_reduce4<UVD, 12>(R, red, lane, fc); align_reduce4<UVD, 16>(R, red, lane, fc); align_reduce4<UVD, 20>(R, red, lane, fc);
      align_reduce4<UVD, 24>(R, red, lane, fc); align_reduce4<UVD, 28>(R, red, lane, fc);
    }
  }
  __syncthreads();
  VS_PHASE_STAMP(5, tp0);
  wg_round_solve(c, b, s, sh, n, T);
}

// Small-register form of the round (VS_ALIGN_LDS builds: the co-scheduled frame kernel must live in 128 VGPRs).  The measurements
// are staged in LDS once per converge() instead of being held in registers across the rounds (those beyond the LDS capacity are
// re-read from HBM, L2 hits), no error / inlier value is carried from round to round — after the last round they are recomputed
// from sh.Tprev, the estimate that round linearized at — and there is ONE code path per chunk.  Same arithmetic per measurement,
// same reduction, same bits as wg_one_round.
template <bool UVD>
__device__ __forceinline__ void wg_one_round_lds(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int n, bool ignore_outliers,
                                                 const AlignPoint* lp, int cap) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  VS_PHASE_BEGIN(tp0);
  for (int base = 0; base < n; base += VS_WG) {
    const int u = base + tid;
    const bool have = u < n;
    AlignPoint P;
    P.m[0] = P.m[1] = P.m[2] = 0; P.f[0] = P.f[1] = P.f[2] = P.f[3] = 0; P.om = 0; P.wt = 0;
    if (have) { if (u < cap) P = lp[u]; else load_align_point(c, b, s, u, P); }
    AlignRows R;
    double chi_w; uint8_t inl_w;
    const bool wave_has = base + w * 64 < n;      // see wg_one_round
    if (wave_has) align_rows<UVD>(c, sh.T, P, have, ignore_outliers, R, &chi_w, &inl_w);
    if (wave_has) {
      double (*red)[32] = sh.red4[w];
      const bool fc = base == 0;
      align_reduce4<UVD, 0>(R, red, lane, fc);  align_reduce4<UVD, 4>(R, red, lane, fc);  align_reduce4<UVD, 8>(R, red, lane, fc);
      align_reduce4<UVD, 12>(R, red, lane, fc); align_reduce4<UVD, 16>(R, red, lane, fc); align_reduce4<UVD, 20>(R, red, lane, fc);
      align_reduce4<UVD, 24>(R, red, lane, fc); align_reduce4<UVD, 28>(R, red, lane, fc);
    }
  }
  __syncthreads();
  VS_PHASE_STAMP(5, tp0);
  wg_round_solve(c, b, s, sh, n, sh.T);
}

template <bool UVD>
__device__ __forceinline__ void wg_align_converge_lds(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int n, const double* T_init,
                                                      unsigned char* lds, int lds_bytes) {
  const int tid = threadIdx.x;
  AlignPoint* lp = reinterpret_cast<AlignPoint*>(lds);
  const int cap = lds_bytes / (int)sizeof(AlignPoint);
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 12; ++k) { sh.T[k] = T_init[k]; sh.Tprev[k] = T_init[k]; }
    sh.its = 0; sh.conv = 0; sh.Eprev = 0; sh.E = 0; sh.inl = 0; sh.outl = n;
    for (int k = 0; k < 36; ++k) sh.H[k] = 0;
  }
  for (int u = tid; u < n && u < cap; u += VS_WG) { AlignPoint P; load_align_point(c, b, s, u, P); lp[u] = P; }
  __syncthreads();
  const int max_it = c.c.aligner_maximum_number_of_iterations;
  const double delta = c.c.aligner_error_delta_for_convergence;
  double e_prev = 0;
  int it = 0, it2 = 0;
  bool refine = false;
  while (max_it > 0) {
    wg_one_round_lds<UVD>(c, b, s, sh, n, refine, lp, cap);
    const double E = sh.E;
    if (!refine) {
      ++it;
      if (delta > fabs(e_prev - E)) {
        e_prev = E;
        const int min_inl = UVD ? 100 : c.c.aligner_minimum_number_of_inliers;
        if (sh.inl > min_inl && sh.inl > sh.outl && max_it > 0) { refine = true; continue; }
        if (tid == 0) sh.conv = 1;
        break;
      }
      e_prev = E;
      if (it >= max_it) break;
    } else {
      ++it2;
      const bool done = fabs(e_prev - E) < delta;
      e_prev = E;
      if (done || it2 >= max_it) { if (tid == 0) sh.conv = 1; break; }
    }
  }
  // errors / inlier flags of the last linearization: one more evaluation of the rows at the estimate it used
  double* chi_o = b.al_chi + (size_t)s * c.MAXP;
  uint8_t* inl_o = b.al_inl + (size_t)s * c.MAXP;
  const bool ran = sh.its > 0;
  for (int u = tid; u < n; u += VS_WG) {
    double chi_w = -1; uint8_t inl_w = 0;
    if (ran) {
      AlignPoint P;
      if (u < cap) P = lp[u]; else load_align_point(c, b, s, u, P);
      AlignRows R;
      align_rows<UVD>(c, sh.Tprev, P, true, false, R, &chi_w, &inl_w);
    }
    chi_o[u] = chi_w; inl_o[u] = inl_w;
  }
  __syncthreads();
}

// converge (:210-264) on the aligner SoA of stream s (n measurements), starting from T_init
template <bool UVD = false>
__device__ __forceinline__ void wg_align_converge(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int n, const double* T_init,
                                                  unsigned char* lds = nullptr, int lds_bytes = 0) {
#ifdef VS_ALIGN_LDS
  if (lds) { wg_align_converge_lds<UVD>(c, b, s, sh, n, T_init, lds, lds_bytes); return; }
#endif
  (void)lds; (void)lds_bytes;
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 12; ++k) sh.T[k] = T_init[k];
    sh.its = 0; sh.conv = 0; sh.Eprev = 0; sh.E = 0; sh.inl = 0; sh.outl = n;
    for (int k = 0; k < 36; ++k) sh.H[k] = 0;
  }
  __syncthreads();
  const int max_it = c.c.aligner_maximum_number_of_iterations;
  const double delta = c.c.aligner_error_delta_for_convergence;
  AlignPoint cache[VS_ALCACHE];
  double chi_reg[VS_ALCACHE];
  uint8_t inl_reg[VS_ALCACHE];
#pragma unroll
  for (int q = 0; q < VS_ALCACHE; ++q) {
    const int u = tid + q * VS_WG;
    chi_reg[q] = -1; inl_reg[q] = 0;
    // threads without a measurement carry zeros: their (finite) rows enter the reduction with zero weight
    cache[q].m[0] = cache[q].m[1] = cache[q].m[2] = 0; cache[q].f[0] = cache[q].f[1] = cache[q].f[2] = cache[q].f[3] = 0;
    cache[q].om = 0; cache[q].wt = 0;
    if (u < n) load_align_point(c, b, s, u, cache[q]);
  }
  // converge() as one loop (single inlined copy of the round): outer rounds use the saturated kernel, after the
  // first convergence inlier-only rounds follow while they keep changing the error (:216-255)
  double e_prev = 0;
  int it = 0, it2 = 0;
  bool refine = false;
  while (max_it > 0) {   // maximum_number_of_iterations 0: no round at all (:216), errors stay -1, T = T_init, not converged
    wg_one_round<UVD>(c, b, s, sh, n, refine, cache, chi_reg, inl_reg);
    const double E = sh.E;
    if (!refine) {
      ++it;
      if (delta > fabs(e_prev - E)) {
        e_prev = E;
        // inlier-only rounds: StereoUVAligner asks for more than the configured minimum (stereouv_aligner.cpp:224),
        // UVDAligner for more than a hard-wired 100 (uvd_aligner.cpp:211)
        const int min_inl = UVD ? 100 : c.c.aligner_minimum_number_of_inliers;
        if (sh.inl > min_inl && sh.inl > sh.outl && max_it > 0) { refine = true; continue; }
        if (tid == 0) sh.conv = 1;
        break;
      }
      e_prev = E;
      if (it >= max_it) break;
    } else {
      ++it2;
      const bool done = fabs(e_prev - E) < delta;
      e_prev = E;
      if (done || it2 >= max_it) { if (tid == 0) sh.conv = 1; break; }
    }
  }
  // errors / inlier flags of the last linearization (the rounds themselves keep them in registers: a global store per
  // round would put an HBM round trip on every barrier)
#pragma unroll
  for (int q = 0; q < VS_ALCACHE; ++q) {
    const int u = tid + q * VS_WG;
    if (u < n) { (b.al_chi + (size_t)s * c.MAXP)[u] = chi_reg[q]; (b.al_inl + (size_t)s * c.MAXP)[u] = inl_reg[q]; }
  }
  __syncthreads();
}

// _weights_translation of StereoUVAligner::initialize (stereouv_aligner.cpp:22,57-61).  The vector is a member of the aligner and
// `resize(n, 1)` only initialises the elements it appends: with inverse depth enabled every weight is rewritten, without it the
// first min(previous size, n) weights are whatever the previous initialize() left there — the Localizing frames after a
// breakTrack (pose_tracker_3d.cpp:124) run with the inverse-depth weights of the last Tracking frame.  `old` is the stored
// weight, `wprev` the vector's size before the call.
__device__ __forceinline__ double al_weight_rule(bool inverse_depth, int u, int wprev, double old, double depth, double max_reliable) {
  if (inverse_depth) return fmin(max_reliable / depth, 1.0);
  return u < wprev ? old : 1.0;
}

// initialize (:10-69) on the tracked list, then converge
__device__ __forceinline__ void wg_align(const DevCfg& c, const DevBuf& b, int s, FrameShared& sh, int pb_prev, bool inverse_depth,
                         const double* T_init, unsigned char* lds = nullptr, int lds_bytes = 0) {
  const int tid = threadIdx.x;
  const int n = sh.n_trk;
  const PtView pv = pts_of(c, b, s, pb_prev);
  const int32_t* trk = b.trk + (size_t)s * c.MAXP * 4;
  const int16_t* kxyL = kpxy_of(c, b, s, 0);
  const int16_t* kxyR = kpxy_of(c, b, s, 1);
  double* moving = b.al_moving + (size_t)s * c.MAXP * 3;
  double* fixed = b.al_fixed + (size_t)s * c.MAXP * 4;
  double* omega = b.al_omega + (size_t)s * c.MAXP;
  double* weight = b.al_weight + (size_t)s * c.MAXP;
  const int wprev = b.st[s].al_wsize;
  __syncthreads();
  if (tid == 0) b.st[s].al_wsize = n;
  for (int u = tid; u < n; u += VS_WG) {
    const int ip = trk[4 * u], fl = trk[4 * u + 1], fr = trk[4 * u + 2];
    const int xL = kxyL[2 * fl], yL = kxyL[2 * fl + 1], xR = kxyR[2 * fr], yR = kxyR[2 * fr + 1];
    fixed[4 * (size_t)u] = xL; fixed[4 * (size_t)u + 1] = yL; fixed[4 * (size_t)u + 2] = xR; fixed[4 * (size_t)u + 3] = yR;
    const int lmup = pv.meta[(size_t)ip * META + M_LMUP];
    double om = 1;
    if (lmup > 0) {
      for (int k = 0; k < 3; ++k) moving[3 * (size_t)u + k] = pv.camlm[3 * (size_t)ip + k];
      om *= (1 + log((double)lmup));
    } else {
      for (int k = 0; k < 3; ++k) moving[3 * (size_t)u + k] = pv.cam[3 * (size_t)ip + k];
    }
    omega[u] = om;
    double cam[3];
    triangulate(c, xL, yL, xR, yR, cam);
    weight[u] = al_weight_rule(inverse_depth, u, wprev, (!inverse_depth && u < wprev) ? weight[u] : 1.0, cam[2], c.c.maximum_reliable_depth_meters);
  }
  wg_align_converge(c, b, s, sh, n, T_init, lds, lds_bytes);
}

// stand-alone aligner on caller-provided correspondences (vslam_align_points, vslam_align_points_uvd)
#ifdef VS_ALIGN_WAVES
#define VS_ALIGN_BOUNDS __launch_bounds__(VS_WG, VS_ALIGN_WAVES)
#else
#define VS_ALIGN_BOUNDS __launch_bounds__(VS_WG)
#endif
// known-answer entry (vslam_aligner_weights): a sequence of initialize() calls on one aligner, weights after each call
__global__ __launch_bounds__(256) void k_aligner_weights(int n_calls, const int32_t* n, const int32_t* inverse_depth, const double* depth,
                                                          double max_reliable, double* weight, double* out) {
  __shared__ int wsize;
  if (threadIdx.x == 0) wsize = 0;
  __syncthreads();
  size_t off = 0;
  for (int k = 0; k < n_calls; ++k) {
    const int wprev = wsize, nk = n[k];
    __syncthreads();
    if (threadIdx.x == 0) wsize = nk;
    for (int u = threadIdx.x; u < nk; u += blockDim.x) {
      weight[u] = al_weight_rule(inverse_depth[k] != 0, u, wprev, weight[u], depth[off + u], max_reliable);
      out[off + u] = weight[u];
    }
    off += (size_t)nk;
    __syncthreads();
  }
}

template <bool UVD>
__global__ VS_ALIGN_BOUNDS void k_align_points(const DevCfg c, const DevBuf b, int n, const double* T_init) {
  __shared__ FrameShared sh;
  double T0[12];
  for (int k = 0; k < 12; ++k) T0[k] = T_init[k];
#ifdef VS_ALIGN_LDS
  __shared__ __align__(16) unsigned char al_lds[VS_ARENA];
  wg_align_converge<UVD>(c, b, 0, sh, n, T0, al_lds, VS_ARENA);
#else
  wg_align_converge<UVD>(c, b, 0, sh, n, T0);
#endif
  if (threadIdx.x == 0) {
    StreamState& st = b.st[0];
    st.al_n = n; st.al_inliers = sh.inl; st.al_outliers = sh.outl; st.al_iterations = sh.its; st.al_converged = sh.conv;
    st.al_total_error = sh.E;
    for (int k = 0; k < 12; ++k) st.al_T[k] = sh.T[k];
    for (int k = 0; k < 36; ++k) st.al_H[k] = sh.H[k];
  }
}
