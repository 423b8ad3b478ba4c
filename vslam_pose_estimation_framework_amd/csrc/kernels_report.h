// kernels_report.h — stage reports for a host that keeps the reference's object model (shim/proslam_hip_plugin.h).
//
// The reference's tracker reads results back after EVERY plug-in call (keypoints after initialize(), the tracked list after
// track(), errors()/inliers() after converge(), the point list after recoverPoints() and compute()).  Read back array by array
// (count first, then one copy per array, each with its own synchronisation, through pageable memory) that is ~25 host round trips
// per frame — more wall time than the kernels (profiles/r04_shim_path.json).  Instead ONE small kernel per stage packs exactly the
// live elements of what the caller reads next into a pinned, device-mapped host buffer (the GPU writes over PCIe; sizes are known
// on the device only), and the host synchronises the frame stream once.  Layout: a fixed header, then arrays at offsets that depend
// on the capacities only (ReportLayout, computed on the host).
#pragma once
#include "dev_types.h"

enum { VS_REPORT_KEYPOINTS = 1, VS_REPORT_TRACK = 2, VS_REPORT_ALIGNER = 3, VS_REPORT_POINTS = 4,
       VS_REPORT_KEYPOINTS_XY = 5 /* coordinates and scores only, as soon as k_emit has written them: published through seq_xy */ };

struct ReportLayout {   // byte offsets into the report buffer (64 B aligned)
  uint32_t kp_xy[2], kp_score[2], desc[2];     // per side: int16 [n][2], u8 [n], u8 [n][32]
  uint32_t trk, lost;                           // int32 [n_trk][4], int32 [n_lost]
  uint32_t chi, inl;                            // double [al_n], u8 [al_n]
  uint32_t p_kp, p_meta, p_cam, p_desc;         // int16 [n][4], int32 [n][6], double [n][3], u8 [n][64]
  uint32_t total;
};

struct ReportHeader {
  int32_t what, stream, in_progress, frame_count;
  int32_t n_kp[2];
  int32_t n_trk, n_lost, n_tracked_landmarks;
  int32_t al_n, al_inliers, al_outliers, al_iterations, al_converged;
  int32_t n_points, n_after_prune;
  int32_t seq, seq_xy;                          // seq: the launch that wrote this report (StageIo::seq); written LAST, system scope: the host may poll it.
                                                // seq_xy: the same for the early coordinates-only keypoint report (the full report follows behind k_brief)
  double al_total_error;
  double al_T[12];
  double al_H[36];
  unsigned long long ticks[5];                  // the stream's in-kernel chronometers (100 MHz ticks, accumulated)
  vslam_frame_info info;
};

// nbytes from src (device) to dst (mapped host memory) by gsz lanes (lane gid).  The report offsets are 64 B aligned; the source arrays
// start at (stream, side) * capacity elements, which is 16 B aligned for the usual capacities (multiples of 64) but only as aligned as
// the caller's max_keypoints / max_points make it: the widest unit both pointers allow is used (16 B, 4 B or single bytes).
__device__ __forceinline__ void report_copy(void* dst, const void* src, size_t nbytes, size_t gid, size_t gsz) {
  const uintptr_t both = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst);
  const unsigned char* s1 = reinterpret_cast<const unsigned char*>(src);
  unsigned char* d1 = reinterpret_cast<unsigned char*>(dst);
  size_t done = 0;
  if ((both & 15) == 0) {
    const size_t n16 = nbytes >> 4;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = gid; i < n16; i += gsz) d4[i] = s4[i];
    done = n16 << 4;
  } else if ((both & 3) == 0) {
    const size_t n4 = nbytes >> 2;
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
    for (size_t i = gid; i < n4; i += gsz) d4[i] = s4[i];
    done = n4 << 2;
  }
  for (size_t i = done + gid; i < nbytes; i += gsz) d1[i] = s1[i];
}

// what a stage launch carries besides its stage number (by value in the kernel arguments): tracker-owned state the caller set
// since the last launch (vslam_set_tracker_state / vslam_set_pose of a one-stream context are folded into the next stage launch
// instead of a one-thread kernel each) and the report the stage packs when it is done (no separate report launch)
struct StageIo {
  int32_t set_flags;            // bit 0: status / window / tau / prior, bit 1: pose
  int32_t status, win;
  double tau;
  double prior[12], pose[12];
  int32_t report, report_in_progress, report_stream, seq;     // VS_REPORT_* (0 = none) for stream report_stream; seq stamps the header
  ReportLayout L;
  unsigned char* out;
};

// packs `what` of stream s with gsz lanes; `first` lanes (one workgroup) also write the header
__device__ __forceinline__ void report_body(const DevCfg& c, const DevBuf& b, int s, int what, int in_progress, int seq, const ReportLayout& L,
                                            unsigned char* out, size_t gid, size_t gsz, bool first, int lane, int lanes) {
  const StreamState& st = b.st[s];
  ReportHeader* h = reinterpret_cast<ReportHeader*>(out);
  const int pb = in_progress ? (st.cur ^ 1) : st.cur;
  int n_points = 0;
  if (what == VS_REPORT_POINTS) n_points = in_progress ? st.n_cur : (st.has_prev ? b.n_points[s * 2 + st.cur] : 0);
  if (n_points > c.MAXP) n_points = c.MAXP;
  if (first && what == VS_REPORT_KEYPOINTS_XY) {
    if (lane == 0) { h->n_kp[0] = b.n_kp[s * 2]; h->n_kp[1] = b.n_kp[s * 2 + 1]; }       // nothing else of the header: the full report owns it
  } else if (first) {
    if (lane == 0) {
      h->what = what; h->stream = s; h->in_progress = in_progress; h->frame_count = st.frame_count;
      h->n_kp[0] = b.n_kp[s * 2]; h->n_kp[1] = b.n_kp[s * 2 + 1];
      h->n_trk = st.n_trk; h->n_lost = st.n_lost; h->n_tracked_landmarks = st.n_tracked_landmarks;
      h->al_n = st.al_n; h->al_inliers = st.al_inliers; h->al_outliers = st.al_outliers; h->al_iterations = st.al_iterations;
      h->al_converged = st.al_converged; h->n_points = n_points; h->n_after_prune = st.n_after_prune; h->al_total_error = st.al_total_error;
      for (int k = 0; k < 5; ++k) h->ticks[k] = st.ticks[k];
    }
    if (lane < 12) h->al_T[lane] = st.al_T[lane];
    if (lane < 36) h->al_H[lane] = st.al_H[lane];
    const uint32_t* is = reinterpret_cast<const uint32_t*>(b.info + s);
    uint32_t* id = reinterpret_cast<uint32_t*>(&h->info);
    for (int k = lane; k < (int)(sizeof(vslam_frame_info) / 4); k += lanes) id[k] = is[k];
  }
  if (what == VS_REPORT_KEYPOINTS || what == VS_REPORT_KEYPOINTS_XY) {
    for (int side = 0; side < 2; ++side) {
      int n = b.n_kp[s * 2 + side];
      if (n > c.NMAX) n = c.NMAX;
      const size_t o = ((size_t)s * 2 + side) * c.NMAX;
      report_copy(out + L.kp_xy[side], b.kp_xy + o * 2, (size_t)n * 4, gid, gsz);
      report_copy(out + L.kp_score[side], b.kp_score + o, (size_t)n, gid, gsz);
      if (what == VS_REPORT_KEYPOINTS) report_copy(out + L.desc[side], b.desc + o * 32, (size_t)n * 32, gid, gsz);
    }
  } else if (what == VS_REPORT_TRACK) {
    report_copy(out + L.trk, b.trk + (size_t)s * c.MAXP * 4, (size_t)st.n_trk * 16, gid, gsz);
    report_copy(out + L.lost, b.lost + (size_t)s * c.MAXP, (size_t)st.n_lost * 4, gid, gsz);
  } else if (what == VS_REPORT_ALIGNER) {
    report_copy(out + L.chi, b.al_chi + (size_t)s * c.MAXP, (size_t)st.al_n * 8, gid, gsz);
    report_copy(out + L.inl, b.al_inl + (size_t)s * c.MAXP, (size_t)st.al_n, gid, gsz);
  } else if (what == VS_REPORT_POINTS) {
    // finished frame (in_progress 0): kp, meta, cam of every point (the caller has the descriptors: they are its features').
    // Frame in assembly (1, after prune + recovery): kp of every point (the caller checks the survivors' order), meta / cam / desc
    // of the RECOVERED points only (index >= n_after_prune) — the survivors are objects the caller already holds.
    const size_t o = ((size_t)s * 2 + pb) * c.MAXP;
    int first = in_progress ? st.n_after_prune : 0;
    if (first > n_points) first = n_points;
    const size_t nf = (size_t)(n_points - first);
    report_copy(out + L.p_kp, b.p_kp + o * 4, (size_t)n_points * 8, gid, gsz);
    // cam rows are 24 B: start the 16 B copy at an even point index
    const size_t fe = (size_t)first & ~(size_t)1;
    report_copy(out + L.p_cam + fe * 24, b.p_cam + (o + fe) * 3, ((size_t)n_points - fe) * 24, gid, gsz);
    if (in_progress) report_copy(out + L.p_desc + (size_t)first * 64, b.p_desc + (o + first) * 64, nf * 64, gid, gsz);
    // meta in the PUBLIC layout of vslam_get_points: (hamming_LR, epipolar_offset, previous_index, track_length, landmark_updates,
    // disparity) — the device's sixth field (has_next) stays internal
    const int32_t* m = b.p_meta + o * 6;
    const int16_t* kp = b.p_kp + o * 4;
    int32_t* d = reinterpret_cast<int32_t*>(out + L.p_meta);
    for (size_t i = (size_t)first * 6 + gid; i < (size_t)n_points * 6; i += gsz) {
      const size_t p = i / 6, f = i - p * 6;
      d[i] = f < 5 ? m[i] : (int32_t)kp[4 * p] - (int32_t)kp[4 * p + 2];
    }
  }
}

// completion flag of a report, after every lane's stores: release at system scope so that a host polling the header sees the
// arrays complete (the host does not have to wait for the runtime to notice that the kernel has ended)
__device__ __forceinline__ void report_publish(unsigned char* out, int seq, bool xy = false) {
  ReportHeader* h = reinterpret_cast<ReportHeader*>(out);
  __hip_atomic_store(xy ? &h->seq_xy : &h->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_report(const DevCfg c, const DevBuf b, int s, int what, int in_progress, int seq, const ReportLayout L,
                                                unsigned char* out, unsigned int* done) {
  report_body(c, b, s, what, in_progress, seq, L, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x,
              blockIdx.x == 0, (int)threadIdx.x, (int)blockDim.x);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int arrived = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == gridDim.x - 1) { __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); report_publish(out, seq, what == VS_REPORT_KEYPOINTS_XY); }   // the last block
  }
}
