// kernels_rgbd.h — RGB-D mode as a DEVICE-RESIDENT loop: PoseTracker3D::compute with a DepthFramePointGenerator and a
// UVDAligner plugged in (pose_tracker_3d.cpp:32-566, depth_framepoint_generator.cpp:24-407, uvd_aligner.cpp), one sequence.  gfx950, wave64.
//
// Everything the tracker carries from frame to frame lives in HBM — the two frame lists (framepoints followed by the temporary points, ping-pong),
// the per-frame history ring the landmark refinement reads (camera coordinates, poses), every track's trail of predecessor indices, the tracker
// scalars (RgbdState) — and every step of a frame is a kernel on ONE HIP stream: the host enqueues the frame and reads one small state block back
// at its end.  The host-driven loop of csrc/rgbd_tracker.h (same results, every entry point a round trip) stays as the second implementation
// the tests compare with.
//
//   k_rgbd_begin            frame scalars
//   k_depth_init/min/pick/write    space map                                                      (kernels_depth.h)
//   k_fast_box, k_emit, k_brief | k_gauss7 + k_orb_describe   the image pipeline on ONE image     (kernels_image.h; k_emit run_controller = 2)
//   k_rgbd_track_candidates wide: window candidates of every previous point
//   k_rgbd_track            feature order of the reference (detector regions row-major, row-major inside a region); order-exact resolution
//                           (depth_track_body), then _track's bookkeeping: framepoints, temporary points, links, lost list,
//                           window / descriptor-distance adaptation, and what the registration does next
//   k_rgbd_align            UVDAligner::initialize + converge (wg_align_converge<UVD>), then accept / fall back / ask for another attempt
//        ... a frame whose registration asks for another attempt (pose_tracker_3d.cpp:333-418, rare) gets the block from the image pipeline to
//        k_rgbd_align enqueued again by the host (it reads RgbdState::done); the tail below is enqueued optimistically and skips itself until then.
//        detectKeypoints appends to the frame's keypoint vector (base_framepoint_generator.cpp:422; nothing clears it between the initialize()
//        calls of one frame), so such an attempt works on the UNION of the frame's detections: k_rgbd_save_features keeps the list so far,
//        k_rgbd_merge_features merges the new detection into it (row-major like k_emit's, equal pixels in attempt order, CSR = the sum of the
//        two, the reference's vector order [earlier attempts..., this detection], the lattice's last-writer-wins as visibility flags)
//   k_rgbd_prune            _prunePoints; projection of the lost points' landmarks
//   k_rgbd_describe_at, k_rgbd_recover_finish      recoverPoints: descriptors at the projections, gates, new framepoints
//   k_rgbd_landmarks        wide, eight lanes per framepoint: Landmark::Landmark / Landmark::update of its track, measurements in the reference's order
//   k_rgbd_finish           temporary points triangulated, compute() on the unmatched features, lists joined, history, trails, frame info
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_depth.h"

#define RGBD_F_LM 1      // FramePoint::landmark() is set (the point went through Landmark::Landmark or Landmark::update)
#define RGBD_F_UNREL 2   // hasUnreliableDepth (inherited along a track)
#define RGBD_F_NEXT 4    // next() is set: a point of the following frame links here (survives re-registrations)
#define RGBD_F_CHAIN 8   // origin()->landmark() is set: the track has a landmark

struct RgbdList {
  float* xy;          // [MAXP][2] keypoint
  uint8_t* desc;      // [MAXP][32]
  double* cam;        // [MAXP][3] camera coordinates
  int32_t* prev;      // [MAXP] index in the previous frame's list (framepoints followed by temporary points), -1 none
  int32_t* tlen;      // [MAXP] trackLength
  uint8_t* flags;     // [MAXP] RGBD_F_*
  double* lmw;        // [MAXP][3] the track's landmark: world coordinates
  int32_t* lmu;       // [MAXP] its number of updates
  int32_t* lmm;       // [MAXP] track length of the point the landmark was created at (order of its first measurements)
  uint16_t* trail;    // [MAXP][TR] index of the track's point in frame f-1, f-2, ... (valid entries: min(trackLength, TR))
};

struct RgbdState {
  // tracker state carried from frame to frame
  int32_t status, win, frame_count, n_lm_prev, wsize, error_flags;
  int32_t last_points, last_all;            // previous frame: framepoints, framepoints + temporary points
  double tau_track, prior[12], world[12];
  // frame in flight
  int32_t n_points, n_temps, n_detected, n_raw, n_lost, n_tracked, n_tracked_lm, attempts, recursion;
  int32_t do_align, inverse_depth, done, tail_done, next_by_app, status0;
  int32_t aligner_valid, al_inliers, al_iterations, al_n;
  int32_t n_registered, n_after_prune, n_recovered, n_active, n_new, fallback, broken;
  double al_total, al_T[12], c2w[12], w2c[12];
  int32_t tcounts[4], ccounts[2], rcount, n_rem;
  int32_t n_temporary;
  int32_t n_acc, merged;       // features of the frame's earlier attempts kept by k_rgbd_save_features; 1: the live feature list is a merged one (fvis / order are the merge's)
  vslam_frame_info info;
};

struct RgbdBuf {
  vslam_depth_params p;
  RgbdState* st;
  RgbdList fl[2];
  RgbdList tmp;               // temporary points of the frame in flight (trail unused)
  int32_t TR, H;              // trail entries per point, frames of history
  int32_t MAXP, NMAX, npx, nbins, n_streams;   // capacities per sequence: every array below holds n_streams slices of its per-sequence size
  // features of the last initialize(): the inner context's keypoints / descriptors / CSR of image 0, plus
  int32_t* order;             // [NMAX] feature index (row-major numbering) of the j-th feature in the reference's order
  uint8_t* matched;           // [NMAX]
  // the frame's keypoint vector over re-registration attempts: the list of the earlier attempts (a_*: a copy of the inner context's arrays, its order
  // and visibility), the new detection while it is merged in (t_*), the merged list's visibility (fvis: 0 = a later feature sits on the same pixel)
  int16_t* a_kxy; uint8_t* a_desc; uint8_t* a_score; int32_t* a_rowcell; int32_t* a_order; uint8_t* a_vis;
  int16_t* t_kxy; uint8_t* t_desc; uint8_t* t_score; int32_t* t_order; int32_t* m_rank;
  uint8_t* fvis;
  int32_t ncsr;               // rows * (CW + 1): entries of a row / cell CSR
  // space map
  const uint16_t* depth; unsigned long long* dkey; int32_t* dlast; float* space; int16_t* row_map; int16_t* col_map;
  // track scratch
  int32_t* hold; int32_t* pick; unsigned long long* cand; int32_t* out2; double* xyz; int32_t* temp2; int32_t* lost_raw;
  // lost list handed from track to recoverPoints
  int32_t* lost; uint8_t* lost_has; double* lost_lm; uint8_t* lost_desc;
  // recovery scratch
  int16_t* rbxy; float* rkxy; int32_t* rcell; uint8_t* rkeep; uint8_t* rdesc; int32_t* ridx; float* rxy; uint8_t* rrdesc; double* rxyz;
  // compute scratch
  int32_t* rcF; int32_t* remf; int32_t* rcT; unsigned long long* bins; uint8_t* cls; int32_t* new_feat; double* new_xyz; int32_t* temp_feat; double* temp_xyz;
  // aligner weights (UVDAligner::_weights_translation: a member vector that is never cleared)
  double* weights;
  // history ring
  double* h_cam;              // [H][MAXP][4] camera coordinates and 1 / z (Measurement::inverse_depth_meters, divided once)
  double* h_pose;             // [H][24] camera_to_world, world_to_camera
  double* pose_log;           // [VS_POSE_LOG][12]
  int32_t* cross;             // [n_streams] k_depth_direct: some depth pixel of the image projects onto another pixel (the general z-buffer runs)
};

// One context tracks n_streams independent sequences (one workgroup per sequence in the single-workgroup kernels, blockIdx.y in the wide ones):
// the kernels receive the whole table and take their sequence's slice of every array first; the code below that line is written for ONE sequence.
__device__ __forceinline__ void rgbd_list_at(RgbdList& l, size_t s, size_t P, size_t TR) {
  l.xy += s * P * 2; l.desc += s * P * 32; l.cam += s * P * 3; l.prev += s * P; l.tlen += s * P; l.flags += s * P;
  l.lmw += s * P * 3; l.lmu += s * P; l.lmm += s * P;
  if (l.trail) l.trail += s * P * TR;
}
__device__ __forceinline__ RgbdBuf rgbd_stream(const RgbdBuf& a, int stream) {
  RgbdBuf r = a;
  const size_t s = (size_t)stream, P = (size_t)a.MAXP, N = (size_t)a.NMAX, X = (size_t)a.npx;
  r.st += s;
  rgbd_list_at(r.fl[0], s, P, a.TR); rgbd_list_at(r.fl[1], s, P, a.TR); rgbd_list_at(r.tmp, s, P, a.TR);
  r.order += s * N; r.matched += s * N;
  r.a_kxy += s * N * 2; r.a_desc += s * N * 32; r.a_score += s * N; r.a_rowcell += s * (size_t)a.ncsr; r.a_order += s * N; r.a_vis += s * N;
  r.t_kxy += s * N * 2; r.t_desc += s * N * 32; r.t_score += s * N; r.t_order += s * N; r.m_rank += s * N * 2; r.fvis += s * N;
  r.depth += s * X; r.dkey += s * X; r.dlast += s * X; r.space += s * X * 3; r.row_map += s * X; r.col_map += s * X;
  r.hold += s * N * 2; r.pick += s * P; r.cand += s * P * (VS_DT_K + 1); r.out2 += s * P * 2; r.xyz += s * P * 3; r.temp2 += s * P * 2; r.lost_raw += s * P;
  r.lost += s * P; r.lost_has += s * P; r.lost_lm += s * P * 3; r.lost_desc += s * P * 32;
  r.rbxy += s * P * 2; r.rkxy += s * P * 2; r.rcell += s * P; r.rkeep += s * P; r.rdesc += s * P * 32; r.ridx += s * P; r.rxy += s * P * 2;
  r.rrdesc += s * P * 32; r.rxyz += s * P * 3;
  r.rcF += s * N * 2; r.remf += s * N; r.rcT += s * P * 2; r.bins += s * (size_t)a.nbins; r.cls += s * N; r.new_feat += s * N; r.new_xyz += s * N * 3;
  r.temp_feat += s * N; r.temp_xyz += s * N * 3;
  r.weights += s * P;
  r.h_cam += s * (size_t)a.H * P * 4; r.h_pose += s * (size_t)a.H * 24; r.pose_log += s * (size_t)VS_POSE_LOG * 12;
  r.cross += s;
  return r;
}
// (field by field: indexing r.fl[] with a run-time value would force the whole per-sequence table — a local value since the batch — into scratch)
__device__ __forceinline__ RgbdList rgbd_pick(const RgbdBuf& r, bool second) {
  RgbdList l;
  l.xy = second ? r.fl[1].xy : r.fl[0].xy; l.desc = second ? r.fl[1].desc : r.fl[0].desc; l.cam = second ? r.fl[1].cam : r.fl[0].cam;
  l.prev = second ? r.fl[1].prev : r.fl[0].prev; l.tlen = second ? r.fl[1].tlen : r.fl[0].tlen; l.flags = second ? r.fl[1].flags : r.fl[0].flags;
  l.lmw = second ? r.fl[1].lmw : r.fl[0].lmw; l.lmu = second ? r.fl[1].lmu : r.fl[0].lmu; l.lmm = second ? r.fl[1].lmm : r.fl[0].lmm;
  l.trail = second ? r.fl[1].trail : r.fl[0].trail;
  return l;
}
__device__ __forceinline__ RgbdList rgbd_cur(const RgbdBuf& r) { return rgbd_pick(r, (r.st->frame_count & 1) != 0); }
__device__ __forceinline__ RgbdList rgbd_prev(const RgbdBuf& r) { return rgbd_pick(r, (r.st->frame_count & 1) == 0); }

__device__ __forceinline__ void rgbd_copy_desc(uint8_t* dst, const uint8_t* src) {
  const uint4 a = reinterpret_cast<const uint4*>(src)[0], b = reinterpret_cast<const uint4*>(src)[1];
  reinterpret_cast<uint4*>(dst)[0] = a; reinterpret_cast<uint4*>(dst)[1] = b;
}

// ---- frame start (PoseTracker3D::compute up to the first initialize) ---------------------------------------------------------------------
__global__ void k_rgbd_begin(const RgbdBuf all) {
  if (threadIdx.x != 0) return;
  const RgbdBuf r = rgbd_stream(all, blockIdx.x);
  RgbdState& st = *r.st;
  vslam_frame_info z = {};
  st.info = z;
  st.status0 = st.status;
  for (int k = 0; k < 12; ++k) st.c2w[k] = st.world[k];
  tf_inverse(st.world, st.w2c);
  st.n_points = 0; st.n_temps = 0; st.n_lost = 0; st.n_tracked = 0; st.n_tracked_lm = 0; st.attempts = 0; st.recursion = 0;
  st.do_align = 0; st.inverse_depth = 0; st.aligner_valid = 0; st.al_inliers = 0; st.al_iterations = 0; st.al_total = 0; st.al_n = 0;
  st.n_registered = 0; st.n_after_prune = 0; st.n_recovered = 0; st.n_active = 0; st.n_new = 0; st.fallback = 0; st.broken = 0;
  st.tail_done = 0; st.n_acc = 0; st.merged = 0;
  st.done = st.frame_count == 0 ? 1 : 0;                 // the first frame has nothing to register against
  st.next_by_app = st.status == VSLAM_LOCALIZING ? 1 : 0;  // _track(..., _status == Localizing)
}

// ---- features of one initialize() ------------------------------------------------------------------------------------------------------------
// detectKeypoints concatenates the regions' keypoints in region order (base_framepoint_generator.cpp:355-429); k_emit leaves them row-major.
// A corner lies in exactly one region's FAST-valid area (the regions overlap by 2-4 px, FAST's border is 3), so the reference's order is a
// stable partition of the row-major list by region.  First part of k_rgbd_track (1024 threads).
// out[j] = index (row-major numbering) of the j-th keypoint of ONE detection in the reference's order
__device__ __forceinline__ void rgbd_region_order(const DevCfg& c, const int16_t* kxy, int n, int32_t* out, int* sh) {
  const int tid = threadIdx.x, NT = blockDim.x;
  if (c.n_regions == 1) {
    for (int i = tid; i < n; i += NT) out[i] = i;
    return;
  }
  int base = 0;
  for (int q = 0; q < c.n_regions; ++q) {
    const DevRegion R = c.regions[q];
    for (int i0 = 0; i0 < n; i0 += NT) {
      const int i = i0 + tid;
      int in = 0;
      if (i < n) { const int x = kxy[2 * i], y = kxy[2 * i + 1]; in = (x >= R.x + 3 && x < R.x + R.w - 3 && y >= R.y + 3 && y < R.y + R.h - 3) ? 1 : 0; }
      int total;
      const int at = base + block_exclusive_scan(in, sh, &total);
      if (in) out[at] = i;
      base += total;
    }
  }
}
__device__ __forceinline__ void rgbd_features(const DevCfg& c, const DevBuf& b, const RgbdBuf& r, int sq, int* sh) {
  RgbdState& st = *r.st;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int n = b.n_kp[2 * sq];
  for (int i = tid; i < n; i += NT) r.matched[i] = 0;                       // setFeatures: a fresh store per initialize()
  if (!st.merged) rgbd_region_order(c, kpxy_of(c, b, sq, 0), n, r.order, sh);   // a merged list carries the order k_rgbd_merge_features gave it
  if (tid == 0) {
    st.n_detected = n;                                                      // _number_of_detected_keypoints = keypointsLeft().size(): all attempts'
    int raw = 0;
    for (int q = 0; q < c.n_regions; ++q) raw += b.iinfo[sq].raw_count[0][q];
    st.n_raw = raw;
  }
  __syncthreads();
}

// ---- the frame's keypoint vector over re-registration attempts -------------------------------------------------------------------------------
// Before attempt 2 / 3 detects: keep the list so far (the inner context's keypoint arrays of image 0 are about to be overwritten).
__global__ __launch_bounds__(1024) void k_rgbd_save_features(const DevCfg c, const DevBuf b, const RgbdBuf all) {
  const int sq = blockIdx.x;
  if (!vs_active(b, sq)) return;
  const RgbdBuf r = rgbd_stream(all, sq);
  RgbdState& st = *r.st;
  if (st.done) return;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int n = b.n_kp[2 * sq];
  const int16_t* kxy = kpxy_of(c, b, sq, 0);
  const uint8_t* desc = desc_of(c, b, sq, 0);
  const uint8_t* ksc = kpscore_of(c, b, sq, 0);
  const int32_t* rc = rowcell_of(c, b, sq, 0);
  for (int i = tid; i < n; i += NT) {
    reinterpret_cast<int32_t*>(r.a_kxy)[i] = reinterpret_cast<const int32_t*>(kxy)[i];
    r.a_score[i] = ksc[i]; r.a_order[i] = r.order[i]; r.a_vis[i] = st.merged ? r.fvis[i] : (uint8_t)1;
  }
  for (int i = tid; i < 2 * n; i += NT) reinterpret_cast<uint4*>(r.a_desc)[i] = reinterpret_cast<const uint4*>(desc)[i];
  for (int i = tid; i < r.ncsr; i += NT) r.a_rowcell[i] = rc[i];
  __syncthreads();
  if (tid == 0) st.n_acc = n;
}

// After the attempt's k_emit + descriptors: the inner context holds the new detection B (row-major, CSR); merge the kept list A into it.
// Row-major by (row, column), equal pixels in vector order (A's first); rank of A[i] = i + #{B < A[i]}, of B[j] = j + #{A <= B[j]}.
__device__ __forceinline__ uint32_t rgbd_pixel_key(const int16_t* kxy, int i) { return ((uint32_t)(uint16_t)kxy[2 * i + 1] << 16) | (uint16_t)kxy[2 * i]; }
__global__ __launch_bounds__(1024) void k_rgbd_merge_features(const DevCfg c, const DevBuf b, const RgbdBuf all) {
  __shared__ int sh[17];
  const int sq = blockIdx.x;
  if (!vs_active(b, sq)) return;
  const RgbdBuf r = rgbd_stream(all, sq);
  RgbdState& st = *r.st;
  if (st.done) return;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int nA = st.n_acc, nB = b.n_kp[2 * sq];
  if (nA + nB > c.NMAX) {                     // the frame fails with VSLAM_ERR_CAPACITY (bit 0, like k_emit's overflow)
    if (tid == 0) { atomicOr(&st.error_flags, 1); st.merged = 0; }
    return;
  }
  int16_t* kxy = kpxy_of(c, b, sq, 0);
  uint8_t* desc = desc_of(c, b, sq, 0);
  uint8_t* ksc = kpscore_of(c, b, sq, 0);
  int32_t* rc = rowcell_of(c, b, sq, 0);
  rgbd_region_order(c, kxy, nB, r.t_order, sh);
  for (int i = tid; i < nB; i += NT) { reinterpret_cast<int32_t*>(r.t_kxy)[i] = reinterpret_cast<const int32_t*>(kxy)[i]; r.t_score[i] = ksc[i]; }
  for (int i = tid; i < 2 * nB; i += NT) reinterpret_cast<uint4*>(r.t_desc)[i] = reinterpret_cast<const uint4*>(desc)[i];
  __syncthreads();
  auto place = [&](int at, const int16_t* sxy, const uint8_t* sdesc, const uint8_t* ssc, int i, uint8_t vis) {
    reinterpret_cast<int32_t*>(kxy)[at] = reinterpret_cast<const int32_t*>(sxy)[i];
    rgbd_copy_desc(desc + (size_t)32 * at, sdesc + (size_t)32 * i);
    ksc[at] = ssc[i]; r.fvis[at] = vis;
  };
  for (int i = tid; i < nA; i += NT) {
    const uint32_t key = rgbd_pixel_key(r.a_kxy, i);
    int lo = 0, hi = nB;                                   // lower bound: B's entries with a smaller key
    while (lo < hi) { const int m = (lo + hi) >> 1; if (rgbd_pixel_key(r.t_kxy, m) < key) lo = m + 1; else hi = m; }
    const bool covered = lo < nB && rgbd_pixel_key(r.t_kxy, lo) == key;      // this detection found the pixel again: it owns the lattice cell now
    r.m_rank[i] = i + lo;
    place(i + lo, r.a_kxy, r.a_desc, r.a_score, i, (uint8_t)((r.a_vis[i] && !covered) ? 1 : 0));
  }
  for (int j = tid; j < nB; j += NT) {
    const uint32_t key = rgbd_pixel_key(r.t_kxy, j);
    int lo = 0, hi = nA;                                   // upper bound: A's entries with a key <= this one
    while (lo < hi) { const int m = (lo + hi) >> 1; if (rgbd_pixel_key(r.a_kxy, m) <= key) lo = m + 1; else hi = m; }
    r.m_rank[c.NMAX + j] = j + lo;
    place(j + lo, r.t_kxy, r.t_desc, r.t_score, j, 1);
  }
  for (int i = tid; i < r.ncsr; i += NT) rc[i] = min(rc[i] + r.a_rowcell[i], c.NMAX);
  __syncthreads();
  // keypointsLeft(): [earlier attempts' keypoints in their order, this detection region-major]
  for (int j = tid; j < nA; j += NT) r.order[j] = r.m_rank[r.a_order[j]];
  for (int j = tid; j < nB; j += NT) r.order[nA + j] = r.m_rank[c.NMAX + r.t_order[j]];
  if (tid == 0) { b.n_kp[2 * sq] = nA + nB; st.merged = 1; }
}

// ---- _track (pose_tracker_3d.cpp:225-298) around DepthFramePointGenerator::track (:166-287) --------------------------------------------------
__device__ __forceinline__ void rgbd_track_args(const DevCfg& c, const DevBuf& b, const RgbdBuf& r, int sq, DepthTrack& a) {
  const RgbdState& st = *r.st;
  const RgbdList pv = rgbd_prev(r);
  a.p = r.p;
  for (int k = 0; k < 12; ++k) a.T[k] = st.prior[k];
  a.by_app = st.next_by_app;
  a.d = a.by_app ? c.c.maximum_projection_tracking_distance_pixels : st.win;   // :229-231
  a.tau = c.c.minimum_descriptor_distance_tracking;
  a.nP = st.last_all; a.nL = b.n_kp[2 * sq]; a.CW = c.CW;
  a.cam = pv.cam; a.pdesc = pv.desc; a.pflags = pv.flags;
  a.kxy = kpxy_of(c, b, sq, 0); a.desc = desc_of(c, b, sq, 0); a.rowcell = rowcell_of(c, b, sq, 0);
  a.space = r.space; a.fvis = st.merged ? r.fvis : nullptr;       // one detection: FAST leaves one feature per pixel; a merged list: the lattice's last writer
  a.hold = r.hold; a.pick = r.pick; a.cand = r.cand; a.counts = r.st->tcounts; a.out2 = r.out2; a.xyz = r.xyz; a.temp2 = r.temp2; a.lost = r.lost_raw;
}

__global__ __launch_bounds__(256) void k_rgbd_track_candidates(const DevCfg c, const DevBuf b, const RgbdBuf all) {
  __shared__ unsigned long long keys[16][VS_DT_CAP];
  __shared__ int cnt[16];
  const int sq = blockIdx.y;
  if (!vs_active(b, sq)) return;
  const RgbdBuf r = rgbd_stream(all, sq);
  if (r.st->done) return;
  DepthTrack a;
  rgbd_track_args(c, b, r, sq, a);
  depth_track_candidates_body(a, keys, cnt);
}

__device__ __forceinline__ void rgbd_fallback(RgbdState& st) {        // _fallbackEstimate
  tf_identity(st.prior);
  for (int k = 0; k < 12; ++k) st.c2w[k] = st.world[k];
  tf_inverse(st.world, st.w2c);
  st.fallback = 1;
}
__device__ __forceinline__ void rgbd_break_track(RgbdState& st) {     // breakTrack
  st.status = VSLAM_LOCALIZING;
  for (int k = 0; k < 12; ++k) st.c2w[k] = st.world[k];
  tf_inverse(st.world, st.w2c);
  tf_identity(st.prior);
  st.n_tracked = 0;
  st.broken = 1;
}
__device__ __forceinline__ void rgbd_accept(const DevCfg& c, RgbdState& st) {
  const double* T = st.al_T;
  const double dt = sqrt((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
  if (rotation_angle(T) > c.c.minimum_delta_angular_for_movement || dt > c.c.minimum_delta_translational_for_movement) {
    for (int k = 0; k < 12; ++k) st.prior[k] = T[k];
    double inv[12];
    tf_inverse(st.prior, inv);
    tf_mul(st.world, inv, st.c2w);
    tf_inverse(st.c2w, st.w2c);
  } else {
    rgbd_fallback(st);
  }
}

__global__ __launch_bounds__(1024) void k_rgbd_track(const DevCfg c, const DevBuf b, const RgbdBuf all) {
  __shared__ int sh[17];
  __shared__ int changed;
  const int sq = blockIdx.x;
  if (!vs_active(b, sq)) return;
  const RgbdBuf r = rgbd_stream(all, sq);
  RgbdState& st = *r.st;
  rgbd_features(c, b, r, sq, sh);      // also on the first frame: compute() walks the features in this order
  if (st.done) return;
  const int tid = threadIdx.x, NT = blockDim.x;
  DepthTrack a;
  rgbd_track_args(c, b, r, sq, a);
  depth_track_body(a, sh, &changed);
  __syncthreads();
  const RgbdList cur = rgbd_cur(r), pv = rgbd_prev(r), tp = r.tmp;
  const int nt = st.tcounts[0], nl = st.tcounts[2], nlm = st.tcounts[3];
  int ntmp = st.tcounts[1];
  const int t0 = st.n_temps;
  if (t0 + ntmp > c.MAXP) { ntmp = c.MAXP - t0; if (tid == 0) atomicOr(&st.error_flags, 2); }
  const int16_t* kxy = a.kxy;
  // frame->points().clear(); one framepoint per tracked point (Frame::createFramepoint + setPrevious, frame_point.cpp:43-55)
  for (int u = tid; u < nt; u += NT) {
    const int i = r.out2[2 * u], f = r.out2[2 * u + 1];
    cur.xy[2 * u] = (float)kxy[2 * f]; cur.xy[2 * u + 1] = (float)kxy[2 * f + 1];
    rgbd_copy_desc(cur.desc + (size_t)32 * u, a.desc + (size_t)32 * f);
    for (int k = 0; k < 3; ++k) cur.cam[3 * (size_t)u + k] = r.xyz[3 * (size_t)u + k];
    const int fl = pv.flags[i];
    cur.prev[u] = i; cur.tlen[u] = pv.tlen[i] + 1;
    cur.flags[u] = (uint8_t)((fl & RGBD_F_UNREL) | ((fl & RGBD_F_LM) ? RGBD_F_CHAIN : 0));
    for (int k = 0; k < 3; ++k) cur.lmw[3 * (size_t)u + k] = pv.lmw[3 * (size_t)i + k];
    cur.lmu[u] = pv.lmu[i]; cur.lmm[u] = pv.lmm[i];
    pv.flags[i] = (uint8_t)(fl | RGBD_F_NEXT);
    r.matched[f] = 1;
  }
  // matches on pixels without a depth measurement: temporary points (:247-256); they are NOT cleared between registration attempts
  for (int u = tid; u < ntmp; u += NT) {
    const int i = r.temp2[2 * u], f = r.temp2[2 * u + 1], j = t0 + u;
    tp.xy[2 * j] = (float)kxy[2 * f]; tp.xy[2 * j + 1] = (float)kxy[2 * f + 1];
    rgbd_copy_desc(tp.desc + (size_t)32 * j, a.desc + (size_t)32 * f);
    for (int k = 0; k < 3; ++k) tp.cam[3 * (size_t)j + k] = 0.0;
    const int fl = pv.flags[i];
    tp.prev[j] = i; tp.tlen[j] = pv.tlen[i] + 1;
    tp.flags[j] = RGBD_F_UNREL;
    tp.lmu[j] = 0; tp.lmm[j] = 0;
    pv.flags[i] = (uint8_t)(fl | RGBD_F_NEXT);
    r.matched[f] = 1;
  }
  __syncthreads();
  // lost list: a previous point an EARLIER attempt of this frame linked is not lost (its next() is set)
  int n_lost = 0;
  for (int u0 = 0; u0 < nl; u0 += NT) {
    const int u = u0 + tid;
    int i = -1, keep = 0;
    if (u < nl) { i = r.lost_raw[u]; keep = (pv.flags[i] & RGBD_F_NEXT) ? 0 : 1; }
    int total;
    const int at = n_lost + block_exclusive_scan(keep, sh, &total);
    if (keep) {
      r.lost[at] = i;
      r.lost_has[at] = (pv.flags[i] & RGBD_F_LM) ? 1 : 0;
      for (int k = 0; k < 3; ++k) r.lost_lm[3 * (size_t)at + k] = pv.lmw[3 * (size_t)i + k];
      rgbd_copy_desc(r.lost_desc + (size_t)32 * at, pv.desc + (size_t)32 * i);
    }
    n_lost += total;
  }
  if (tid == 0) {
    const int wmax = c.c.maximum_projection_tracking_distance_pixels, wmin = c.c.minimum_projection_tracking_distance_pixels;
    if (a.by_app) st.win = wmax;
    st.n_points = nt; st.n_temps = t0 + ntmp; st.n_lost = n_lost; st.n_tracked = nt; st.n_tracked_lm = nlm;
    const double ratio = (double)nt / (double)st.last_points;
    const double lm_per_point = (double)nlm / (double)nt, success = (double)nt / (double)c.target_kp;
    int win = st.win;
    if (ratio < c.c.good_tracking_ratio / 2) { if (win < wmax) win = (int)fmin(win * 1 / c.c.tunnel_vision_ratio, (double)wmax); }
    else if (win > wmin) win = (int)fmax(win * c.c.tunnel_vision_ratio, (double)wmin);
    st.win = win;
    if (ratio < c.c.good_tracking_ratio || nt < c.c.aligner_minimum_number_of_inliers || (lm_per_point < 0.5 && success < 0.25))
      st.tau_track = fmin(st.tau_track + 5, c.c.maximum_descriptor_distance_tracking);
    else st.tau_track = fmax(st.tau_track - 5, c.c.minimum_descriptor_distance_tracking);
    st.aligner_valid = 0;
    st.attempts += 1;
    // what the registration does with this track() (compute :57-76, _registerRecursive :300-418)
    st.do_align = 0;
    if (st.status0 == VSLAM_LOCALIZING) {
      if (nt < c.c.minimum_number_of_landmarks_to_track) { rgbd_fallback(st); st.done = 1; }
      else { st.do_align = 1; st.inverse_depth = 0; }
    } else {
      const double rel = (double)nlm / (double)st.n_lm_prev;
      if (nlm == 0 || rel < 0.1) {
        if (st.recursion < 2) { tf_identity(st.prior); st.next_by_app = 1; st.recursion += 1; }
        else { rgbd_break_track(st); st.done = 1; }
      } else { st.do_align = 1; st.inverse_depth = 1; }
    }
  }
}

// ---- UVDAligner::initialize (uvd_aligner.cpp:11-69) + converge, then the registration's verdict ------------------------------------------
__global__ VS_ALIGN_BOUNDS void k_rgbd_align(const DevCfg c, const DevBuf b, const RgbdBuf all) {
  __shared__ FrameShared sh;
  const int sq = blockIdx.x;
  if (!vs_active(b, sq)) return;
  const RgbdBuf r = rgbd_stream(all, sq);
  RgbdState& st = *r.st;
  if (st.done || !st.do_align) return;
  double* al_fixed = b.al_fixed + (size_t)sq * c.MAXP * 4;
  double* al_moving = b.al_moving + (size_t)sq * c.MAXP * 3;
  double* al_omega = b.al_omega + (size_t)sq * c.MAXP;
  double* al_weight = b.al_weight + (size_t)sq * c.MAXP;
  const int tid = threadIdx.x;
  const RgbdList cur = rgbd_cur(r), pv = rgbd_prev(r);
  const int n = st.n_points, old = st.wsize, inverse = st.inverse_depth;
  for (int u = tid; u < n; u += VS_WG) {
    const double z = cur.cam[3 * (size_t)u + 2];
    const int ip = cur.prev[u];
    double w = u < old ? r.weights[u] : 1.0, wd = 10.0;          // _weights_translation.resize(n, 1) keeps what it holds
    if (cur.flags[u] & RGBD_F_UNREL) { w = 0; wd = 0; }
    else if (inverse) w = c.c.maximum_reliable_depth_meters / z;
    r.weights[u] = w;
    al_fixed[4 * (size_t)u] = (double)cur.xy[2 * u]; al_fixed[4 * (size_t)u + 1] = (double)cur.xy[2 * u + 1];
    al_fixed[4 * (size_t)u + 2] = z; al_fixed[4 * (size_t)u + 3] = wd;
    for (int k = 0; k < 3; ++k) al_moving[3 * (size_t)u + k] = pv.cam[3 * (size_t)ip + k];    // the current point has no landmark yet (:38)
    al_omega[u] = 1.0; al_weight[u] = w;
  }
  double T0[12];
  for (int k = 0; k < 12; ++k) T0[k] = st.prior[k];
  __threadfence_block();
  __syncthreads();
  wg_align_converge<true>(c, b, sq, sh, n, T0);
  __syncthreads();
  if (tid == 0) {
    st.wsize = n;
    st.do_align = 0; st.aligner_valid = 1; st.al_n = n;
    st.al_inliers = sh.inl; st.al_total = sh.E; st.al_iterations = sh.its;
    for (int k = 0; k < 12; ++k) st.al_T[k] = sh.T[k];
    const int need = c.c.minimum_number_of_landmarks_to_track;
    if (st.status0 == VSLAM_LOCALIZING) {
      if (sh.inl < need) rgbd_fallback(st); else rgbd_accept(c, st);
      st.done = 1;
    } else if (sh.inl > need) {
      rgbd_accept(c, st);
      st.done = 1;
    } else if (st.recursion < 2) {
      if (st.win < c.c.maximum_projection_tracking_distance_pixels) st.win += 1;
      st.next_by_app = 0; st.recursion += 1;
    } else {
      rgbd_break_track(st);
      st.done = 1;
    }
  }
}

// ---- the tail: enqueued after every attempt, runs once the registration is done ----------------------------------------------------------
__device__ __forceinline__ bool rgbd_tail_on(const RgbdState& st) { return st.done && !st.tail_done; }

struct RgbdPoint { float xy[2]; uint4 d0, d1; double cam[3]; int32_t prev, tlen, lmu, lmm; uint8_t flags; double lmw[3]; };
__device__ __forceinline__ void rgbd_load(const RgbdList& l, int i, RgbdPoint& q) {
  q.xy[0] = l.xy[2 * i]; q.xy[1] = l.xy[2 * i + 1];
  q.d0 = reinterpret_cast<const uint4*>(l.desc + (size_t)32 * i)[0]; q.d1 = reinterpret_cast<const uint4*>(l.desc + (size_t)32 * i)[1];
  for (int k = 0; k < 3; ++k) { q.cam[k] = l.cam[3 * (size_t)i + k]; q.lmw[k] = l.lmw[3 * (size_t)i + k]; }
  q.prev = l.prev[i]; q.tlen = l.tlen[i]; q.lmu = l.lmu[i]; q.lmm = l.lmm[i]; q.flags = l.flags[i];
}
__device__ __forceinline__ void rgbd_store(const RgbdList& l, int i, const RgbdPoint& q) {
  l.xy[2 * i] = q.xy[0]; l.xy[2 * i + 1] = q.xy[1];
  reinterpret_cast<uint4*>(l.desc + (size_t)32 * i)[0] = q.d0; reinterpret_cast<uint4*>(l.desc + (size_t)32 * i)[1] = q.d1;
  for (int k = 0; k < 3; ++k) { l.cam[3 * (size_t)i + k] = q.cam[k]; l.lmw[3 * (size_t)i + k] = q.lmw[k]; }
  l.prev[i] = q.prev; l.tlen[i] = q.tlen; l.lmu[i] = q.lmu; l.lmm[i] = q.lmm; l.flags[i] = q.flags;
}

// DepthFramePointGenerator::recoverPoints (:289-407)
__device__ __forceinline__ void rgbd_recover_args(const DevCfg& c, const RgbdBuf& r, DepthRecover& a) {
  const RgbdState& st = *r.st;
  a.p = r.p;
  for (int k = 0; k < 12; ++k) a.w2c[k] = st.w2c[k];
  a.kp_size = 7.f; a.tau = c.c.minimum_descriptor_distance_tracking; a.n = st.n_lost;
  a.has_lm = r.lost_has; a.lm = r.lost_lm; a.pdesc = r.lost_desc; a.space = r.space;
  a.bxy = r.rbxy; a.kxy = r.rkxy; a.cell = r.rcell; a.keep = r.rkeep; a.desc = r.rdesc;
  a.count = &r.st->rcount; a.rec_index = r.ridx; a.rec_xy = r.rxy; a.rec_desc = r.rrdesc; a.rec_xyz = r.rxyz;
}
__device__ __forceinline__ bool rgbd_recover_on(const DevCfg& c, const RgbdState& st) {
  return rgbd_tail_on(st) && st.frame_count > 0 && c.c.enable_landmark_recovery && st.n_lost > 0;
}
// _prunePoints (:437-472): without a fresh aligner result every tracked point is dropped.  Order-preserving compaction in place, 1024
// points per pass (a pass reads its points before it writes, and writes never reach the next pass's points).
__global__ __launch_bounds__(1024) void k_rgbd_prune(const DevCfg c, const DevBuf b, const RgbdBuf all) {
  __shared__ int sh[17];
  const int sq = blockIdx.x;
  const RgbdBuf r = rgbd_stream(all, sq);
  RgbdState& st = *r.st;
  if (!rgbd_tail_on(st)) return;
  const double* al_chi = b.al_chi + (size_t)sq * c.MAXP;
  const uint8_t* al_inl = b.al_inl + (size_t)sq * c.MAXP;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int n = st.n_points;
  if (tid == 0) st.n_registered = n;
  if (st.frame_count == 0) return;
  // recoverPoints, first step: projection of the lost points' landmarks with the frame's pose (independent of the pruning)
  if (c.c.enable_landmark_recovery && st.n_lost > 0) {
    DepthRecover a;
    rgbd_recover_args(c, r, a);
    for (int i = tid; i < a.n; i += NT) depth_recover_project_one(a, i);
  }
  const RgbdList cur = rgbd_cur(r);
  const bool valid = st.aligner_valid != 0;
  const double kern = c.c.aligner_maximum_error_kernel;
  const double avg = valid ? st.al_total / (double)n : 0;
  int out = 0;
  for (int u0 = 0; u0 < n; u0 += NT) {
    const int u = u0 + tid;
    int keep = 0;
    RgbdPoint q;
    if (u < n) {
      rgbd_load(cur, u, q);
      if (valid) { const double chi = al_chi[u]; keep = avg < kern ? (al_inl[u] != 0) : (chi != -1 && chi < 100 * kern); }
    }
    int total;
    const int at = out + block_exclusive_scan(keep, sh, &total);
    if (keep) rgbd_store(cur, at, q);
    out += total;
    __syncthreads();
  }
  if (tid == 0) { st.n_points = out; st.n_after_prune = out; }
}

// descriptors at the projected pixels: BRIEF on the box image / steered ORB tests on the Gaussian image the image pipeline left
__global__ __launch_bounds__(256) void k_rgbd_describe_at(const DevCfg c, const DevBuf b, const RgbdBuf all) {
  const int sq = blockIdx.y;
  const RgbdBuf r = rgbd_stream(all, sq);
  const RgbdState& st = *r.st;
  if (!rgbd_recover_on(c, st)) return;
  const int n = st.n_lost, rows = c.c.rows, cols = c.c.cols;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  if (r.p.descriptor_type == VSLAM_DESCRIPTOR_ORB) {
    const uint8_t* blur = blur_of(c, b, sq, 0);
    const OrbTaps t = orb_taps(lane, c.orb_cos, c.orb_sin, c.bstride);
    for (int i = wave; i < n; i += nwaves) {
      const int x = r.rbxy[2 * i], y = r.rbxy[2 * i + 1];
      const bool in = x >= VSLAM_ORB_BORDER && x < cols - VSLAM_ORB_BORDER && y >= VSLAM_ORB_BORDER && y < rows - VSLAM_ORB_BORDER;
      if (lane == 0) r.rkeep[i] = in ? 1 : 0;
      unsigned long long d[4] = {0ull, 0ull, 0ull, 0ull};
      if (in) orb_wave(blur + (size_t)y * c.bstride + x, t, d);
      const unsigned long long dv = lane == 0 ? d[0] : (lane == 1 ? d[1] : (lane == 2 ? d[2] : d[3]));
      if (lane < 4) reinterpret_cast<unsigned long long*>(r.rdesc + (size_t)32 * i)[lane] = dv;
    }
  } else {
    const uint16_t* box = box_of(c, b, sq, 0);
    for (int i = wave; i < n; i += nwaves) {
      const int x = r.rbxy[2 * i], y = r.rbxy[2 * i + 1];
      const bool in = x >= VSLAM_BRIEF_BORDER && x < cols - VSLAM_BRIEF_BORDER && y >= VSLAM_BRIEF_BORDER && y < rows - VSLAM_BRIEF_BORDER;
      if (lane == 0) r.rkeep[i] = in ? 1 : 0;
      if (in) brief_wave(box, c.bstride, x, y, lane, r.rdesc + (size_t)32 * i);
      else if (lane < 4) reinterpret_cast<unsigned long long*>(r.rdesc + (size_t)32 * i)[lane] = 0ull;
    }
  }
}
__global__ __launch_bounds__(1024) void k_rgbd_recover_finish(const DevCfg c, const RgbdBuf all) {
  __shared__ int sh[17];
  const RgbdBuf r = rgbd_stream(all, blockIdx.x);
  RgbdState& st = *r.st;
  if (!rgbd_recover_on(c, st)) return;
  const int tid = threadIdx.x, NT = blockDim.x;
  DepthRecover a;
  rgbd_recover_args(c, r, a);
  depth_recover_finish_body(a, sh);
  __syncthreads();
  const RgbdList cur = rgbd_cur(r), pv = rgbd_prev(r);
  int nr = st.rcount;
  const int n0 = st.n_points;
  if (n0 + nr > c.MAXP) { nr = c.MAXP - n0; if (tid == 0) atomicOr(&st.error_flags, 2); }
  for (int k = tid; k < nr; k += NT) {
    const int i = r.lost[r.ridx[k]], j = n0 + k;
    cur.xy[2 * j] = r.rxy[2 * k]; cur.xy[2 * j + 1] = r.rxy[2 * k + 1];
    rgbd_copy_desc(cur.desc + (size_t)32 * j, r.rrdesc + (size_t)32 * k);
    for (int q = 0; q < 3; ++q) cur.cam[3 * (size_t)j + q] = r.rxyz[3 * (size_t)k + q];
    const int fl = pv.flags[i];
    cur.prev[j] = i; cur.tlen[j] = pv.tlen[i] + 1;
    cur.flags[j] = (uint8_t)((fl & RGBD_F_UNREL) | ((fl & RGBD_F_LM) ? RGBD_F_CHAIN : 0));
    for (int q = 0; q < 3; ++q) cur.lmw[3 * (size_t)j + q] = pv.lmw[3 * (size_t)i + q];
    cur.lmu[j] = pv.lmu[i]; cur.lmm[j] = pv.lmm[i];
    pv.flags[i] = (uint8_t)(fl | RGBD_F_NEXT);
  }
  __syncthreads();
  if (tid == 0) { st.n_points = n0 + nr; st.n_recovered = nr; }
}

// _updatePoints (:475-520).  Measurement k of a track is its point in frame f - k: the point itself (k = 0), its predecessor (k = 1), then the
// predecessor's trail — direct addresses, no link walk.  Landmark::Landmark (landmark.cpp:8-33) sums the world coordinates from the newest point
// back to the origin; Landmark::update (:66-167) runs Gauss-Newton over _measurements in THEIR order: the creation's (newest first: the point the
// landmark was created at, back to the origin), then one per later frame, the current point last.
//
// Eight lanes per framepoint.  A Gauss-Newton round over a long track is a chain of ~65 dependent fp64 operations per measurement on one lane
// (measured: 72 us per frame for 60-frame tracks with one lane per point) — but only the thirteen ADDITIONS into H, b and the error have to
// happen in the list's order.  So the eight lanes evaluate eight consecutive measurements at once (projection, residual, kernel, the products
// om * R^T R and om * R^T e), park the results in LDS, and every lane then adds the eight contributions in list order: same operations on the
// same operands in the same order as the serial loop, an eighth of its multiplications on the critical path.
#define RGBD_LM_NP 64     // world_to_camera (and R^T R) of the newest RGBD_LM_NP frames staged in LDS, one copy for the workgroup
#define RGBD_LM_G 8       // lanes per framepoint
#define RGBD_LM_PTS (256 / RGBD_LM_G)
#define RGBD_LM_PRE 64    // measurements of a track kept in LDS across the rounds (each is two dependent HBM loads: trail entry, then the ring)
struct RgbdPoseLds { double w2c[12]; double rtr[9]; };
struct RgbdLmTerm { double e2, h[6], b[3]; int kind, pad; };   // kind 0: behind the camera (an outlier, nothing added), 1: inlier, 2: outlier with a saturated kernel
__global__ __launch_bounds__(256) void k_rgbd_landmarks(const DevCfg c, const RgbdBuf all) {
  const RgbdBuf r = rgbd_stream(all, blockIdx.y);
  __shared__ RgbdPoseLds s_pose[RGBD_LM_NP];
  __shared__ RgbdLmTerm s_term[RGBD_LM_PTS][RGBD_LM_G];
  __shared__ double s_meas[RGBD_LM_PTS][RGBD_LM_PRE][4];     // the track's newest measurements, fetched once for all Gauss-Newton rounds
  RgbdState& st = *r.st;
  if (!rgbd_tail_on(st)) return;
  const int n = st.n_points;
  if ((int)(blockIdx.x * RGBD_LM_PTS) >= n) return;      // (the grid's x extent is a guess of the host: blocks loop over the points below)
  const int f = st.frame_count, H = r.H;
  for (int t = threadIdx.x; t < RGBD_LM_NP * 12; t += 256) {
    const int k = t / 12, e = t - 12 * k;
    if (k <= f && k < H) s_pose[k].w2c[e] = k == 0 ? st.w2c[e] : r.h_pose[(size_t)((f - k) % H) * 24 + 12 + e];
  }
  __syncthreads();
  // J^T J of a measurement (J = the rotation of world_to_camera) depends on the frame only: once per frame instead of once per measurement and round
  for (int t = threadIdx.x; t < RGBD_LM_NP * 9; t += 256) {
    const int k = t / 9, e = t - 9 * k, rr = e / 3, cc = e - 3 * rr;
    if (k <= f && k < H) { const double* W = s_pose[k].w2c; s_pose[k].rtr[e] = (W[rr] * W[cc] + W[4 + rr] * W[4 + cc]) + W[8 + rr] * W[8 + cc]; }
  }
  __syncthreads();
  const int g = threadIdx.x / RGBD_LM_G, gl = threadIdx.x % RGBD_LM_G;
  int n_active = 0;
  for (int i = blockIdx.x * RGBD_LM_PTS + g; i < n; i += gridDim.x * RGBD_LM_PTS) {
  bool active = false;
  {
    const RgbdList cur = rgbd_cur(r), pv = rgbd_prev(r);
    const int T = cur.tlen[i], fl = cur.flags[i];
    if (!(T < c.c.minimum_track_length_for_landmark_creation || (fl & RGBD_F_UNREL))) {
      active = gl == 0;
      const int TR = r.TR, MAXP = c.MAXP;
      const int p1 = cur.prev[i];
      const double own[4] = {cur.cam[3 * (size_t)i], cur.cam[3 * (size_t)i + 1], cur.cam[3 * (size_t)i + 2], 1 / cur.cam[3 * (size_t)i + 2]};
      auto cam_of = [&](int k, double* o) {     // x, y, z, 1 / z of measurement k
        if (k == 0) { o[0] = own[0]; o[1] = own[1]; o[2] = own[2]; o[3] = own[3]; return; }
        const int idx = k == 1 ? p1 : (int)pv.trail[(size_t)p1 * TR + (k - 2)];
        const double2* src = reinterpret_cast<const double2*>(r.h_cam + ((size_t)((f - k) % H) * MAXP + idx) * 4);
        const double2 a = src[0], bq = src[1];
        o[0] = a.x; o[1] = a.y; o[2] = bq.x; o[3] = bq.y;
      };
      auto pose_of = [&](int k) -> const double* { return k == 0 ? st.c2w : r.h_pose + (size_t)((f - k) % H) * 24; };          // camera_to_world
      int len = T + 1;                               // measurements of the track, this frame's included
      const int reach = min(min(H - 1, TR + 1), f);  // oldest k that can still be addressed
      if (len - 1 > reach) { len = reach + 1; if (gl == 0) atomicOr(&st.error_flags, 4); }
      double world[3];
      int updates = cur.lmu[i], lmm = cur.lmm[i];
      if (!(fl & RGBD_F_CHAIN)) {
        // (every lane of the group computes the same short sum)
        double acc[3] = {0, 0, 0};
        for (int k = 0; k < len; ++k) {
          double m[4], w[3];
          cam_of(k, m);
          tf_apply(pose_of(k), m, w);
          for (int q = 0; q < 3; ++q) acc[q] = acc[q] + w[q];
        }
        for (int q = 0; q < 3; ++q) world[q] = acc[q] / (double)len;
        updates = len; lmm = T;
      } else {
        // _measurements in their order: k = T - m .. T (the creation), then k = T - m - 1 .. 0
        const int m0 = min(max(T - lmm, 0), len - 1);
        auto k_at = [&](int j) { return j <= len - 1 - m0 ? m0 + j : len - 1 - j; };   // j-th measurement -> k
        double wv[3] = {cur.lmw[3 * (size_t)i], cur.lmw[3 * (size_t)i + 1], cur.lmw[3 * (size_t)i + 2]};
        for (int q = 0; q < 3; ++q) world[q] = wv[q];
        const double kern = c.c.landmark_maximum_error_squared_meters;
        double err_prev = 0;
        RgbdLmTerm* terms = s_term[g];
        // every lane fetches the measurements it will evaluate (list positions gl, gl + 8, ...), all loads in flight together, once
        double (*meas)[4] = s_meas[g];
        for (int j = gl; j < len && j < RGBD_LM_PRE; j += RGBD_LM_G) {
          double mc[4];
          cam_of(k_at(j), mc);
          meas[j][0] = mc[0]; meas[j][1] = mc[1]; meas[j][2] = mc[2]; meas[j][3] = mc[3];
        }
        for (int it = 0; it < c.c.landmark_maximum_number_of_iterations; ++it) {
          double Hm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bv[3] = {0, 0, 0};
          double err = 0;
          int n_out = 0;
          for (int j0 = 0; j0 < len; j0 += RGBD_LM_G) {
            // this lane's measurement of the batch
            const int j = j0 + gl;
            RgbdLmTerm t;
            t.kind = -1;
            if (j < len) {
              const int k = k_at(j);
              double mc[4];
              if (j < RGBD_LM_PRE) { mc[0] = meas[j][0]; mc[1] = meas[j][1]; mc[2] = meas[j][2]; mc[3] = meas[j][3]; }     // written by this very lane
              else cam_of(k, mc);
              const double* W = s_pose[0].w2c;
              const double* RtR = s_pose[0].rtr;
              double rtr_far[9];
              if (k < RGBD_LM_NP) { W = s_pose[k].w2c; RtR = s_pose[k].rtr; }
              else {     // a track older than the staged poses: the same expressions from HBM
                W = r.h_pose + (size_t)((f - k) % H) * 24 + 12;
                for (int e = 0; e < 9; ++e) { const int rr = e / 3, cc = e - 3 * rr; rtr_far[e] = (W[rr] * W[cc] + W[4 + rr] * W[4 + cc]) + W[8 + rr] * W[8 + cc]; }
                RtR = rtr_far;
              }
              double sp[3];
              tf_apply(W, wv, sp);
              if (sp[2] <= 0) { t.kind = 0; }
              else {
                const double er[3] = {sp[0] - mc[0], sp[1] - mc[1], sp[2] - mc[2]};
                double om = mc[3];
                t.e2 = om * ((er[0] * er[0] + er[1] * er[1]) + er[2] * er[2]);
                t.kind = 1;
                if (t.e2 > kern) { om *= kern / t.e2; t.kind = 2; }
                // R^T R is symmetric to the bit (its entries are sums of commuting products): six products instead of nine
                t.h[0] = om * RtR[0]; t.h[1] = om * RtR[1]; t.h[2] = om * RtR[2]; t.h[3] = om * RtR[4]; t.h[4] = om * RtR[5]; t.h[5] = om * RtR[8];
                for (int rr = 0; rr < 3; ++rr) t.b[rr] = om * ((W[rr] * er[0] + W[4 + rr] * er[1]) + W[8 + rr] * er[2]);
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();     // the previous batch's terms have been read by every lane of the group
            terms[gl] = t;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // every lane adds the batch in list order (its own copy of the accumulators: no broadcast afterwards); plain LDS loads, so that
            // the next terms are on their way while one is being added (volatile reads cost one LDS round trip EACH: 88 per batch)
            const int nb_ = min(RGBD_LM_G, len - j0);
#pragma unroll
            for (int u = 0; u < RGBD_LM_G; ++u) {
              if (u < nb_) {
                const RgbdLmTerm q = terms[u];
                if (q.kind == 0) { ++n_out; }
                else {
                  err += q.e2;
                  if (q.kind == 2) ++n_out;
                  Hm[0] += q.h[0]; Hm[4] += q.h[3]; Hm[8] += q.h[5];
                  { const double h01 = q.h[1], h02 = q.h[2], h12 = q.h[4]; Hm[1] += h01; Hm[3] += h01; Hm[2] += h02; Hm[6] += h02; Hm[5] += h12; Hm[7] += h12; }
                  bv[0] += q.b[0]; bv[1] += q.b[1]; bv[2] += q.b[2];
                }
              }
            }
          }
          double nb[3] = {-bv[0], -bv[1], -bv[2]}, dx[3];
          full_piv_solve_regs<3>(Hm, nb, dx);
          for (int q = 0; q < 3; ++q) wv[q] += dx[q];
          if (fabs(err - err_prev) < 1e-5 || it == 999) {
            const int n_in = len - n_out;
            if ((unsigned)n_in > (unsigned)updates) {
              for (int q = 0; q < 3; ++q) world[q] = wv[q];
              updates = n_in;
            } else if (n_in < n_out) {
              double acc[3] = {0, 0, 0};
              for (int j = 0; j < len; ++j) {
                const int k = k_at(j);
                double mc[4], wp[3];
                cam_of(k, mc);
                tf_apply(pose_of(k), mc, wp);
                for (int q = 0; q < 3; ++q) acc[q] += wp[q];
              }
              for (int q = 0; q < 3; ++q) world[q] = acc[q] / (double)len;
            }
            break;
          }
          err_prev = err;
        }
      }
      if (gl == 0) {
        cur.lmu[i] = updates; cur.lmm[i] = lmm;
        for (int q = 0; q < 3; ++q) cur.lmw[3 * (size_t)i + q] = world[q];
        cur.flags[i] = (uint8_t)(fl | RGBD_F_LM | RGBD_F_CHAIN);
      }
    }
  }
  n_active += active ? 1 : 0;
  }
  // (lanes of a wavefront run different numbers of points: a plain per-lane count, summed once)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n_active += __shfl_xor(n_active, o, 64);
  if ((threadIdx.x & 63) == 0 && n_active) atomicAdd(&st.n_active, n_active);
}

// The rest of the frame in one workgroup: temporary points triangulated with the accepted motion (:524-545), status, compute() on the features
// track() left unmatched (depth_framepoint_generator.cpp:45-164), the frame's list closed (framepoints followed by temporary points), history,
// trails, frame info.
__global__ __launch_bounds__(1024) void k_rgbd_finish(const DevCfg c, const DevBuf b, const RgbdBuf all) {
  __shared__ int sh[17];
  const int sq = blockIdx.x;
  const RgbdBuf r = rgbd_stream(all, sq);
  RgbdState& st = *r.st;
  if (!rgbd_tail_on(st)) return;
  const int tid = threadIdx.x, NT = blockDim.x;
  const RgbdList cur = rgbd_cur(r), pv = rgbd_prev(r), tp = r.tmp;
  const int f = st.frame_count;
  // ---- temporary points: midpoint triangulation between the previous and the current keypoint; kept if in front of the camera
  {
    const int n = st.n_temps;
    double T[12], K[9];
    for (int k = 0; k < 12; ++k) T[k] = st.prior[k];
    for (int k = 0; k < 9; ++k) K[k] = c.c.K[k];
    int out = 0;
    for (int u0 = 0; u0 < n; u0 += NT) {
      const int u = u0 + tid;
      int keep = 0;
      RgbdPoint q;
      if (u < n) {
        rgbd_load(tp, u, q);
        double tri[3];
        point_in_camera_one(pv.xy + 2 * (size_t)q.prev, q.xy, T, K, tri);
        if (!(tri[2] <= 0)) { keep = 1; q.cam[0] = tri[0]; q.cam[1] = tri[1]; q.cam[2] = tri[2]; }
      }
      int total;
      const int at = out + block_exclusive_scan(keep, sh, &total);
      if (keep) rgbd_store(tp, at, q);
      out += total;
      __syncthreads();
    }
    if (tid == 0) {
      st.n_temps = out;
      if (st.n_active > c.c.minimum_number_of_landmarks_to_track) st.status = VSLAM_TRACKING;     // :105-107
    }
    __syncthreads();
  }
  // ---- compute(): unmatched features in the reference's order, the bins the frame's points own
  const int16_t* kxy = kpxy_of(c, b, sq, 0);
  const uint8_t* kdesc = desc_of(c, b, sq, 0);
  int nF = 0;
  {
    const int nd = st.n_detected;
    for (int j0 = 0; j0 < nd; j0 += NT) {
      const int j = j0 + tid;
      int fi = -1, rem = 0;
      if (j < nd) { fi = r.order[j]; rem = r.matched[fi] ? 0 : 1; }
      int total;
      const int at = nF + block_exclusive_scan(rem, sh, &total);
      if (rem) { r.rcF[2 * at] = kxy[2 * fi + 1]; r.rcF[2 * at + 1] = kxy[2 * fi]; r.remf[at] = fi; }
      nF += total;
    }
  }
  const int nT = st.n_points;
  for (int i = tid; i < nT; i += NT) { r.rcT[2 * i] = (int32_t)cur.xy[2 * i + 1]; r.rcT[2 * i + 1] = (int32_t)cur.xy[2 * i]; }
  __syncthreads();
  const int rows_bin = r.p.enable_keypoint_binning ? r.p.rows / r.p.bin_size_pixels + 1 : 0;
  const int cols_bin = r.p.enable_keypoint_binning ? r.p.cols / r.p.bin_size_pixels + 1 : 0;
  const int n_bins = (rows_bin + 1) * (cols_bin + 1);
  depth_compute_body(r.p, r.space, nF, r.rcF, nT, r.rcT, r.bins, n_bins, rows_bin, cols_bin, max(nF, 1), st.ccounts, r.new_feat, r.new_xyz, r.temp_feat,
                     r.temp_xyz, r.cls, sh);
  __syncthreads();
  int nn = st.ccounts[0], nq = st.ccounts[1];
  const int t0 = st.n_temps;
  if (nT + nn > c.MAXP) { nn = c.MAXP - nT; if (tid == 0) atomicOr(&st.error_flags, 2); }
  if (nT + nn + t0 + nq > c.MAXP) { nq = max(c.MAXP - nT - nn - t0, 0); if (tid == 0) atomicOr(&st.error_flags, 2); }
  const int n_all = min(nT + nn + t0 + nq, c.MAXP);
  for (int k = tid; k < nn; k += NT) {
    const int g = r.remf[r.new_feat[k]], j = nT + k;
    cur.xy[2 * j] = (float)kxy[2 * g]; cur.xy[2 * j + 1] = (float)kxy[2 * g + 1];
    rgbd_copy_desc(cur.desc + (size_t)32 * j, kdesc + (size_t)32 * g);
    for (int q = 0; q < 3; ++q) { cur.cam[3 * (size_t)j + q] = r.new_xyz[3 * (size_t)k + q]; cur.lmw[3 * (size_t)j + q] = 0; }
    cur.prev[j] = -1; cur.tlen[j] = 0; cur.flags[j] = 0; cur.lmu[j] = 0; cur.lmm[j] = 0;
  }
  // ---- the frame's list: framepoints, then the temporary points of track() (triangulated), then compute()'s
  const int np = nT + nn;
  for (int u = tid; u < t0 && np + u < c.MAXP; u += NT) { RgbdPoint q; rgbd_load(tp, u, q); rgbd_store(cur, np + u, q); }
  for (int k = tid; k < nq; k += NT) {
    const int g = r.remf[r.temp_feat[k]], j = np + t0 + k;
    cur.xy[2 * j] = (float)kxy[2 * g]; cur.xy[2 * j + 1] = (float)kxy[2 * g + 1];
    rgbd_copy_desc(cur.desc + (size_t)32 * j, kdesc + (size_t)32 * g);
    for (int q = 0; q < 3; ++q) { cur.cam[3 * (size_t)j + q] = r.temp_xyz[3 * (size_t)k + q]; cur.lmw[3 * (size_t)j + q] = 0; }
    cur.prev[j] = -1; cur.tlen[j] = 0; cur.flags[j] = RGBD_F_UNREL; cur.lmu[j] = 0; cur.lmm[j] = 0;
  }
  __syncthreads();
  // ---- history ring and trails
  {
    double* hc = r.h_cam + (size_t)(f % r.H) * c.MAXP * 4;
    for (int i = tid; i < n_all; i += NT) {
      const double x = cur.cam[3 * (size_t)i], y = cur.cam[3 * (size_t)i + 1], z = cur.cam[3 * (size_t)i + 2];
      reinterpret_cast<double2*>(hc + 4 * (size_t)i)[0] = make_double2(x, y);
      reinterpret_cast<double2*>(hc + 4 * (size_t)i)[1] = make_double2(z, 1 / z);
    }
    if (tid < 12) { r.h_pose[(size_t)(f % r.H) * 24 + tid] = st.c2w[tid]; r.h_pose[(size_t)(f % r.H) * 24 + 12 + tid] = st.w2c[tid]; }
    if (tid < 12 && f < VS_POSE_LOG) r.pose_log[(size_t)f * 12 + tid] = st.c2w[tid];
    // 16 lanes per point: entry 0 = the predecessor, entries 1.. = the predecessor's trail
    const int lane = tid & 15, g = tid >> 4, TR = r.TR;
    for (int i = g; i < np; i += NT / 16) {
      const int ip = cur.prev[i];
      if (ip < 0 || (cur.flags[i] & RGBD_F_UNREL)) continue;
      const int cnt = min(cur.tlen[i], TR);
      uint16_t* dst = cur.trail + (size_t)i * TR;
      const uint16_t* src = pv.trail + (size_t)ip * TR;
      for (int k = lane; k < cnt; k += 16) dst[k] = k == 0 ? (uint16_t)ip : src[k - 1];
    }
  }
  __syncthreads();
  if (tid == 0) {
    st.error_flags |= b.st[sq].error_flags & 1;     // keypoint capacity (k_emit)
    vslam_frame_info& o = st.info;
    o.frame_index = f + 1; o.status = st.status; o.status_at_start = st.status0;
    o.n_keypoints_left = st.n_detected; o.n_detected_left = st.n_raw;
    for (int q = 0; q < c.n_regions; ++q) o.thresholds[q] = b.st[sq].thr[q];
    o.track_attempts = st.attempts; o.n_tracked = st.n_registered; o.n_lost = st.n_lost; o.n_tracked_landmarks = st.n_tracked_lm;
    o.aligner_ran = st.aligner_valid ? 1 : 0;
    o.aligner_iterations = st.aligner_valid ? st.al_iterations : 0;
    o.n_inliers = st.aligner_valid ? st.al_inliers : 0; o.n_outliers = st.aligner_valid ? st.al_n - st.al_inliers : 0;
    o.total_error = st.aligner_valid ? st.al_total : 0;
    o.n_after_prune = st.n_after_prune; o.n_recovered = st.n_recovered; o.n_active_landmarks = st.n_active; o.n_new_stereo = nn;
    o.n_points = np; o.track_broken = st.broken; o.fallback = st.fallback; o.window_pixels = st.win; o.error_flags = st.error_flags;
    o.tau_track = st.tau_track;
    for (int k = 0; k < 12; ++k) { o.camera_left_to_world[k] = st.c2w[k]; o.previous_to_current[k] = st.prior[k]; st.world[k] = st.c2w[k]; }
    st.n_temporary = t0 + nq;
    st.n_new = nn;
    st.n_lm_prev = st.n_active;
    st.n_points = np; st.n_temps = t0 + nq;
    st.last_points = np; st.last_all = n_all;
    st.frame_count = f + 1;
    st.tail_done = 1;
    *r.cross = 0;              // the space map has been consumed: the next frame's k_depth_direct starts from "no crossing source"
  }
}
