// vslam_hip.hip — host side of libvslam_hip.so: context / device-buffer management, kernel launches
// and read-back behind the C ABI of include/vslam_hip.h.  gfx950 only; there is no CPU fallback:
// every entry point fails with VSLAM_ERR_NO_DEVICE / VSLAM_ERR_HIP when the GPU path is unusable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kernels_frame2.h"
#include "kernels_depth.h"
#include "kernels_orb.h"
#include "kernels_landmark.h"
#include "kernels_report.h"

#define VS_API extern "C" __attribute__((visibility("default")))

static thread_local std::string g_create_error;

#ifndef VS_SPLIT4_MAX_STREAMS
#define VS_SPLIT4_MAX_STREAMS 96  // up to this many streams the frame runs as launch sequence 4 (phase 0 | wide recovery kernel | phase 4 | phase 2 with the landmark
                                  // refinement in workgroups of its own in the same launch).  Measured, ms per step fused / sequence 4: 1 stream 0.255 (two launches) /
                                  // 0.223, 4: 0.296 / 0.258, 11: 0.343 / 0.292, 32: 0.400 / 0.353, 64: 0.491 / 0.452, 96: 0.584 / 0.561, 128: 0.664 / 0.667, 157: 0.73 / 0.81
#endif
struct vslam_ctx {
  DevCfg cfg;
  DevBuf buf;
  int device = 0;
  int B = 0;
  hipStream_t stream = nullptr;       // tracker kernels (k_track_candidates, k_frame, stages) + read-back
  hipStream_t stream_img = nullptr;   // image pipeline (k_fast_box, k_emit, k_brief) + uploads
  bool own_stream = false;
  // image products are double-buffered: frame t+1 is detected/described while frame t is tracked
  struct ImgSet { uint16_t* box; uint8_t* score8; unsigned long long* mask; int16_t* kp_xy; uint8_t* kp_score; uint8_t* desc;
                  int32_t* n_kp; int32_t* rowcell; uint8_t* used; uint8_t* sdist; ImgInfo* iinfo; } sets[2];
  int parity = 0, last_set = 0;
  // streams are processed in G independent groups, each with its own pair of HIP streams: a slow stream only
  // delays its own group, the other groups' kernels fill the idle CUs
  // st_img: image pipeline of even steps (and uploads), st_img2: image pipeline of odd steps, so that BRIEF of step t
  // overlaps FAST of step t+1 (FAST(t+1) only waits for the threshold controller in k_emit(t))
  struct Group { int s0, n; hipStream_t st_frm, st_img, st_img2; hipEvent_t ev_img[2], ev_frm[2], ev_emit[2]; bool frm_pending[2], emit_pending[2];
                 int q0_frm = 0, q0_img = 0, q0_img2 = 0; };   // XCD that block 0 of a launch on the queue runs on (calibrate_queues)
  std::vector<Group> groups;
  std::string err;
  std::vector<void*> allocs;
  DevCfg* d_cfg = nullptr;            // device-resident copies read by k_frame through the constant address space
  DevBuf* d_bufs = nullptr;           // [2 product sets][groups]
  uint8_t* upload[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [step parity][left/right]
  int up_stride = 0;
  size_t up_stream_stride = 0;
  bool frame_begun = false;
  bool timers = false;
  double timer_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  struct EvRec { hipEvent_t a, b; int k; bool count; };
  std::vector<EvRec> evrec;
  struct EvShared { hipEvent_t a, b; int k; };      // interval whose start event belongs to an EvRec (only b returns to the pool)
  std::vector<EvShared> evshared;
  std::vector<hipEvent_t> evpool;
  double kern_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int kern_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // RGB-D components: the space map of the last vslam_depth_space_map call stays resident for vslam_depth_compute
  struct DepthMap { int rows = 0, cols = 0; uint16_t* depth = nullptr; unsigned long long* key = nullptr; int32_t* last = nullptr;
                    float* space = nullptr; int16_t* row_map = nullptr; int16_t* col_map = nullptr; bool valid = false; } dm;
  // scratch contexts of the stand-alone entry points (one per distinct configuration), kept for reuse: creating one costs
  // ~45 device allocations plus streams and events — several milliseconds, which the host-driven RGB-D loop would pay
  // five times per frame
  struct Scratch { vslam_ctx* t; vslam_config cfg; bool busy; size_t base_allocs; };
  std::vector<Scratch> scratch;
  // per-call device scratch of the stand-alone entry points: blocks kept between calls and handed out by bumping an offset
  // (tmp_get / tmp_reset below) — hipMalloc and hipFree cost tens of microseconds each, hipFree synchronises the device, and the
  // host-driven RGB-D loop would pay ~60 of them per frame
  struct Tmp { std::vector<std::pair<char*, size_t>> blocks; size_t used = 0; } tmp;
  // stage reports (kernels_report.h): pinned, device-mapped host buffer the report kernel packs a stage's results into; pinned
  // staging of the stage path's host images (a pageable hipMemcpyAsync of 2 x 467 KB costs ~0.24 ms of host time)
  unsigned char* report = nullptr; unsigned char* report_dev = nullptr; ReportLayout rl;
  unsigned int* report_done = nullptr;                 // arrival counter of the multi-block report kernel (device)
  // stage path of a one-stream context: the image pipeline runs on the frame queue itself (the caller waits for every stage, so a
  // second queue buys no overlap and costs an event round trip per frame) and is timed by three events instead of two per kernel
  hipStream_t img_override = nullptr;
  bool xcd_affinity = true;      // VSLAM_XCD_AFFINITY=0: block ids as the runtime deals them (measurement aid)
  int xcd_skew = 0;              // VSLAM_XCD_SKEW=k: the image queues' streams k XCDs away from their frame workgroups (measurement aid)
  bool img_on_frm_queue = false;
  int report_seq = 0;                                  // stamps every report launch; the header carries it back
  int report_xy_seq = -1;                              // the early coordinates-only keypoint report of the frame in flight (-1: none)
  int report_have = 0, report_have_ip = 0, report_have_stream = -1, report_have_seq = -1;   // what the LAST launch on the frame queue packed (0: nothing)
  // setters of a one-stream context wait here for the next stage launch (StageIo); flush_pending() launches them on their own
  struct Pending { int flags = 0; int status = 0, win = 0; double tau = 0; double prior[12], pose[12]; } pend;
  unsigned char* pin_img[2] = {nullptr, nullptr}; size_t pin_img_bytes = 0;     // [step parity]: left | right
  hipEvent_t pin_ev[2] = {nullptr, nullptr}; bool pin_used[2] = {false, false};
  int split = 0;   // 0: one frame launch; 1: three phase launches with wide recovery / landmark kernels in between (measured slower);
                   // 2: two phase launches around the wide recovery kernel
                   // 4: phase launches around the wide recovery kernel, the landmark refinement in workgroups of its own inside the last one (fastest up to VS_SPLIT4_MAX_STREAMS streams)
  bool lm_published = false;                            // vslam_prune_recover has published the frame's history (one stream): vslam_compute runs the landmark refinement beside the stereo stage
  int sticky = VSLAM_OK;
};

static int fail(vslam_ctx* c, int code, const std::string& msg) {
  if (c) { c->err = msg; if (code == VSLAM_ERR_HIP) c->sticky = code; }
  else g_create_error = msg;
  return code;
}
#define HIP_TRY(ctx, expr)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(ctx, VSLAM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <typename T>
static hipError_t dalloc(vslam_ctx* c, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
  if (e == hipSuccess) { c->allocs.push_back(q); *p = (T*)q; }
  return e;
}

// ---- per-call device scratch ------------------------------------------------------------------------------
static hipError_t tmp_get(vslam_ctx* c, void** p, size_t bytes) {
  bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
  auto& T = c->tmp;
  if (T.blocks.empty() || T.used + bytes > T.blocks.back().second) {
    (void)hipSetDevice(c->device);     // the caller's thread may have another device current (torch switches it)
    const size_t want = std::max<size_t>(bytes, T.blocks.empty() ? ((size_t)1 << 20) : 2 * T.blocks.back().second);
    void* q = nullptr;
    const hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) return e;
    T.blocks.push_back({(char*)q, want});
    T.used = 0;
  }
  *p = T.blocks.back().first + T.used;
  T.used += bytes;
  return hipSuccess;
}
template <typename T>
static hipError_t tmp_alloc(vslam_ctx* c, T** p, size_t count) { return tmp_get(c, (void**)p, count * sizeof(T)); }
// start of an entry point: everything handed out before is dead (every entry synchronises before it returns its results); blocks that
// had to be chained during a call are merged into one, so that a steady caller allocates nothing
static void tmp_reset(vslam_ctx* c) {
  if (!c) return;
  auto& T = c->tmp;
  if (T.blocks.size() > 1) {
    (void)hipSetDevice(c->device);     // entries call tmp_reset first: the merged block must live on the context's device
    size_t total = 0;
    for (auto& b : T.blocks) { total += b.second; (void)hipFree(b.first); }
    T.blocks.clear();
    void* q = nullptr;
    if (hipMalloc(&q, total) == hipSuccess) T.blocks.push_back({(char*)q, total});
  }
  T.used = 0;
}
static void tmp_free(vslam_ctx* c) {
  for (auto& b : c->tmp.blocks) (void)hipFree(b.first);
  c->tmp.blocks.clear(); c->tmp.used = 0;
}

// ---- optional per-kernel timing (HIP events on the context stream) -----------------------------------
static hipEvent_t ev_get(vslam_ctx* c) {
  if (!c->evpool.empty()) { hipEvent_t e = c->evpool.back(); c->evpool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
struct KernelTimer {
  vslam_ctx* c; int k; hipStream_t st; bool count; hipEvent_t a = nullptr;
  KernelTimer(vslam_ctx* c_, int k_, hipStream_t st_, bool count_ = true, bool enabled_ = true) : c(c_), k(k_), st(st_), count(count_) { if (c->timers && enabled_) { a = ev_get(c); (void)hipEventRecord(a, st); } }
  ~KernelTimer() { if (a) { hipEvent_t b = ev_get(c); (void)hipEventRecord(b, st); c->evrec.push_back({a, b, k, count}); } }
};
static void sync_all(vslam_ctx* c);
static int flush_pending(vslam_ctx* c);
static void harvest_events(vslam_ctx* c) {
  sync_all(c);
  for (auto& r : c->evshared) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->kern_ms[r.k] += ms; c->kern_n[r.k] += 1; }
    c->evpool.push_back(r.b);
  }
  c->evshared.clear();
  for (auto& r : c->evrec) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->kern_ms[r.k] += ms; if (r.count) c->kern_n[r.k] += 1; }
    c->evpool.push_back(r.a); c->evpool.push_back(r.b);
  }
  c->evrec.clear();
}


static void sync_all(vslam_ctx* c) {
  for (auto& g : c->groups) { (void)hipStreamSynchronize(g.st_img); (void)hipStreamSynchronize(g.st_img2); (void)hipStreamSynchronize(g.st_frm); }
}
static int group_of(const vslam_ctx* c, int s) {
  for (size_t i = 0; i < c->groups.size(); ++i) if (s >= c->groups[i].s0 && s < c->groups[i].s0 + c->groups[i].n) return (int)i;
  return 0;
}
static DevBuf buf_set(const vslam_ctx* c, int set, int s0 = 0, int q0 = 0) {
  DevBuf b = c->buf;
  b.s0 = s0;
  b.xcd_rot = c->xcd_affinity ? ((q0 - s0) & 7) : 0;     // dev_types.h: stream s on physical XCD s % 8 whatever queue the launch goes to
  const vslam_ctx::ImgSet& q = c->sets[set];
  b.box = q.box; b.score8 = q.score8; b.mask = q.mask; b.kp_xy = q.kp_xy; b.kp_score = q.kp_score; b.desc = q.desc;
  b.n_kp = q.n_kp; b.rowcell = q.rowcell; b.used = q.used; b.sdist = q.sdist; b.iinfo = q.iinfo;
  return b;
}

// ---- defaults (configurations/configuration_{kitti,euroc}.yaml, src/types/parameters.h) -----------
static void common_defaults(vslam_config* c) {
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
VS_API void vslam_default_config_kitti(vslam_config* c) {
  common_defaults(c);
  c->rows = 376; c->cols = 1241;
  const double K[9] = {718.856, 0, 607.1928, 0, 718.856, 185.2157, 0, 0, 1};
  std::memcpy(c->K, K, sizeof K);
  c->baseline_h[0] = -386.1448;
}
VS_API void vslam_default_config_euroc(vslam_config* c) {
  common_defaults(c);
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
  c->descriptor_type = VSLAM_DESCRIPTOR_ORB;   // configuration_euroc.yaml:52 "ORB-256": unknown to the parser -> cv::ORB::create() (:219-224)
}

VS_API const char* vslam_last_error(const vslam_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

// ---- ORB extractor constants, computed on the host with OpenCV's own expressions [recalled: orb.cpp, smooth.cpp] --------------
static void orb_rotation_host(float angle_degrees, float* a, float* b) {
  float angle = angle_degrees;
  angle *= (float)(3.1415926535897932384626433832795 / 180.f);
  *a = (float)std::cos(angle); *b = (float)std::sin(angle);
}
static void gauss7_kernel_host(int32_t k4[4]) {   // getGaussianKernel(7, 2, CV_32F) -> cvRound(k * 256): centre .. outermost tap
  float cf[7];
  double sum = 0;
  for (int i = 0; i < 7; ++i) { const double x = i - 3.0; cf[i] = (float)std::exp(-0.5 / 4.0 * x * x); sum += cf[i]; }
  sum = 1. / sum;
  for (int i = 0; i < 4; ++i) k4[i] = (int32_t)std::lrint((double)(float)(cf[3 + i] * sum) * 256.0);
}
// ---- configure (BaseFramePointGenerator::configure, base_framepoint_generator.cpp:229-329) ----------
static void derive_cfg(const vslam_config& in, int n_streams, DevCfg* d) {
  std::memset(d, 0, sizeof *d);
  d->c = in;
  d->TX = (in.cols + VS_TILE_W - 1) / VS_TILE_W;
  d->CW = d->TX * 4;
  d->bstride = d->TX * VS_TILE_W;
  const int nv = in.det_rows, nh = in.det_cols;
  const double ph = (double)in.rows / nv, pw = (double)in.cols / nh;
  int k = 0;
  for (int r = 0; r < nv; ++r)
    for (int cc = 0; cc < nh; ++cc) {
      int off_w = nh > 1 ? 2 : 0, off_h = nv > 1 ? 2 : 0, off_r = 0, off_c = 0;
      if (r > 0) { off_r = -off_h; if (r < nv - 1) off_h *= 2; }
      if (cc > 0) { off_c = -off_w; if (cc < nh - 1) off_w *= 2; }
      d->regions[k].x = (int)(std::round(cc * pw) + off_c);
      d->regions[k].y = (int)(std::round(r * ph) + off_r);
      d->regions[k].w = (int)(pw + off_w);
      d->regions[k].h = (int)(ph + off_h);
      ++k;
    }
  d->n_regions = k;
  d->cols_bin = (int)(std::floor((double)in.cols / in.bin_size_pixels) + 1);
  d->rows_bin = (int)(std::floor((double)in.rows / in.bin_size_pixels) + 1);
  d->target_kp = d->cols_bin * d->rows_bin;
  d->target_per_detector = (int)((double)d->target_kp / (double)d->n_regions);
  d->n_offsets = 0;
  d->offsets[d->n_offsets++] = 0;
  for (int u = 1; u <= in.maximum_epipolar_search_offset_pixels; ++u) { d->offsets[d->n_offsets++] = u; d->offsets[d->n_offsets++] = -u; }
  orb_rotation_host(-1.f, &d->orb_cos, &d->orb_sin);   // FAST keypoints: KeyPoint::angle = -1, never recomputed by ORB::compute
  gauss7_kernel_host(d->gauss7);
  d->NMAX = in.max_keypoints;
  d->MAXP = in.max_points;
  d->HCAP = in.max_history_frames;
  d->trail = in.max_points <= 65535 ? 1 : 0;
  d->n_streams = n_streams;
}

// PoseTracker3D::configure (pose_tracker_3d.cpp:11-21) + a fresh generator / aligner / world map for one stream
static void fresh_stream_state(const vslam_ctx* c, StreamState& x) {
  std::memset(&x, 0, sizeof x);
  for (int r = 0; r < c->cfg.n_regions; ++r) x.thr[r] = c->cfg.c.detector_threshold_minimum;
  x.status = VSLAM_LOCALIZING;
  x.win = c->cfg.c.maximum_projection_tracking_distance_pixels;
  x.tau_track = c->cfg.c.minimum_descriptor_distance_tracking;
  x.tau_tri = 0.1 * 256;
  tf_identity(x.prior);
  tf_identity(x.pose);
}
static int upload_buffer_tables(vslam_ctx* c) {
  const size_t G = c->groups.size();
  std::vector<DevBuf> hb(2 * G);
  for (int q = 0; q < 2; ++q) for (size_t g = 0; g < G; ++g) hb[q * G + g] = buf_set(c, q, c->groups[g].s0, c->groups[g].q0_frm);
  HIP_TRY(c, hipMemcpy(c->d_bufs, hb.data(), sizeof(DevBuf) * 2 * G, hipMemcpyHostToDevice));
  return VSLAM_OK;
}
static int init_state(vslam_ctx* c) {
  c->pend.flags = 0;          // a reset drops setters that were waiting for a stage launch: the fresh state is the state
  c->report_have = 0;
  std::vector<StreamState> st(c->B);
  for (int s = 0; s < c->B; ++s) fresh_stream_state(c, st[s]);
  bool all_active = true;
  for (int s = 0; s < c->B; ++s) all_active = all_active && ((c->buf.active[s >> 5] >> (s & 31)) & 1u);
  if (!all_active) {   // a reset of the whole context re-activates every stream
    sync_all(c);
    std::memset(c->buf.active, 0xff, sizeof c->buf.active);
    int rc = upload_buffer_tables(c);
    if (rc != VSLAM_OK) return rc;
  }
  HIP_TRY(c, hipMemcpyAsync(c->buf.st, st.data(), sizeof(StreamState) * c->B, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemsetAsync(c->buf.info, 0, sizeof(vslam_frame_info) * c->B, c->stream));
  HIP_TRY(c, hipMemsetAsync(c->buf.n_points, 0, sizeof(int32_t) * c->B * 2, c->stream));
  sync_all(c);
  for (int q = 0; q < 2; ++q) {
    HIP_TRY(c, hipMemsetAsync(c->sets[q].n_kp, 0, sizeof(int32_t) * c->B * 2, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->sets[q].iinfo, 0, sizeof(ImgInfo) * c->B, c->stream));
    for (auto& g : c->groups) { g.frm_pending[q] = false; g.emit_pending[q] = false; }
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->parity = 0; c->last_set = 0;
  c->frame_begun = false;
  return VSLAM_OK;
}

static void destroy_streams(vslam_ctx* c) {
  for (auto& g : c->groups) {
    for (int q = 0; q < 2; ++q) { if (g.ev_img[q]) (void)hipEventDestroy(g.ev_img[q]); if (g.ev_frm[q]) (void)hipEventDestroy(g.ev_frm[q]); if (g.ev_emit[q]) (void)hipEventDestroy(g.ev_emit[q]); }
    if (c->own_stream) { if (g.st_img != g.st_frm) (void)hipStreamDestroy(g.st_img); if (g.st_img2 != g.st_img) (void)hipStreamDestroy(g.st_img2); (void)hipStreamDestroy(g.st_frm); }
  }
  c->groups.clear();
}
static int create_internal(const vslam_config* cfg, int device, int n_streams, vslam_ctx** out);
static int init_state(vslam_ctx* c);
// check a scratch context of configuration `cfg` out of the parent's pool (fresh stream state, pristine DevCfg) / back in
static int scratch_get(vslam_ctx* parent, const vslam_config& cfg, vslam_ctx** out) {
  for (auto& e : parent->scratch)
    if (!e.busy && std::memcmp(&e.cfg, &cfg, sizeof cfg) == 0) {
      derive_cfg(cfg, 1, &e.t->cfg);            // stand-alone entries edit the detector regions of their scratch DevCfg
      e.t->err.clear(); e.t->sticky = VSLAM_OK; e.t->timers = false;
      const int rc = init_state(e.t);
      if (rc != VSLAM_OK) { parent->err = e.t->err; return rc; }
      e.busy = true;
      *out = e.t;
      return VSLAM_OK;
    }
  vslam_ctx* t = nullptr;
  const int rc = create_internal(&cfg, parent->device, 1, &t);
  if (rc != VSLAM_OK) { parent->err = g_create_error; return rc; }
  if (parent->scratch.size() >= 12) {           // bound the pool: drop an idle entry
    for (size_t i = 0; i < parent->scratch.size(); ++i)
      if (!parent->scratch[i].busy) { vslam_destroy(parent->scratch[i].t); parent->scratch.erase(parent->scratch.begin() + i); break; }
  }
  parent->scratch.push_back({t, cfg, true, t->allocs.size()});
  *out = t;
  return VSLAM_OK;
}
static void scratch_put(vslam_ctx* parent, vslam_ctx* t) {
  if (!t) return;
  for (auto& e : parent->scratch)
    if (e.t == t) {
      sync_all(t);
      for (size_t i = e.base_allocs; i < t->allocs.size(); ++i) (void)hipFree(t->allocs[i]);   // per-call extras (dalloc on the scratch)
      t->allocs.resize(e.base_allocs);
      e.busy = false;
      return;
    }
  vslam_destroy(t);
}
// Workgroup b of a launch runs on XCD (q0 + b) % 8 with q0 a property of the hardware queue behind the HIP stream (constant from launch
// to launch, idle or loaded: tools/probe/xcd_map.hip).  One one-block launch per queue reads it.
__global__ void k_xcc_probe(int* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) *out = (int)(v & 7u);
}
static int calibrate_queues(vslam_ctx* c) {
  if (const char* e = getenv("VSLAM_XCD_AFFINITY")) c->xcd_affinity = atoi(e) != 0;
  if (const char* e = getenv("VSLAM_XCD_SKEW")) c->xcd_skew = atoi(e) & 7;
  int* d = nullptr;
  if (hipMalloc(&d, 3 * sizeof(int)) != hipSuccess) return VSLAM_OK;     // affinity is an optimisation: without it rot stays 0
  for (auto& g : c->groups) {
    int h[3] = {0, 0, 0};
    hipStream_t q[3] = {g.st_frm, g.st_img, g.st_img2};
    bool ok = true;
    for (int k = 0; k < 3 && ok; ++k) { hipLaunchKernelGGL(k_xcc_probe, dim3(1), dim3(64), 0, q[k], d + k); ok = hipStreamSynchronize(q[k]) == hipSuccess; }
    if (ok && hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) { g.q0_frm = h[0]; g.q0_img = (h[1] + c->xcd_skew) & 7; g.q0_img2 = (h[2] + c->xcd_skew) & 7; }
  }
  (void)hipFree(d);
  return VSLAM_OK;
}

static int create_internal(const vslam_config* cfg, int device, int n_streams, vslam_ctx** out) {
  if (!cfg || !out || n_streams < 1) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: null argument or n_streams < 1");
  if (n_streams > VS_MAX_STREAMS) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: more than 4096 streams in one context");
  if (cfg->rows < 1 || cfg->cols < 1 || cfg->cols > 32767 || cfg->rows > 32767) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: invalid image dimensions");
  if (cfg->det_rows < 1 || cfg->det_cols < 1 || cfg->det_rows * cfg->det_cols > VSLAM_MAX_REGIONS) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: invalid detector grid");
  if (!(-cfg->baseline_h[0] / cfg->K[0] > 0)) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: invalid baseline (m), verify intrinsic camera parameters");
  if (cfg->maximum_epipolar_search_offset_pixels < 0 || cfg->maximum_epipolar_search_offset_pixels > VSLAM_MAX_EPI) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: epipolar offset out of range");
  if (cfg->descriptor_type != VSLAM_DESCRIPTOR_BRIEF && cfg->descriptor_type != VSLAM_DESCRIPTOR_ORB) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: unknown descriptor_type");
  if (cfg->max_keypoints < 64 || cfg->max_keypoints > 65535 || cfg->max_points < 64 || cfg->max_history_frames < 2 || cfg->bin_size_pixels < 1) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: invalid capacities");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, VSLAM_ERR_NO_DEVICE, "vslam_create: no HIP device available (the HIP path has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, VSLAM_ERR_NO_DEVICE, "vslam_create: device ordinal out of range");
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, VSLAM_ERR_NO_DEVICE, "vslam_create: hipSetDevice failed");
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_frame)) != hipSuccess)
    return fail(nullptr, VSLAM_ERR_NO_DEVICE, "vslam_create: no gfx950 kernel image for this device");
  vslam_ctx* c = new vslam_ctx;
  c->device = device;
  c->B = n_streams;
  derive_cfg(*cfg, n_streams, &c->cfg);
  {
    int G = 1;   // measured on MI355X: concurrent HIP streams did not overlap the per-group kernels (1 group is fastest)
    if (const char* e = getenv("VSLAM_GROUPS")) G = atoi(e);
    G = std::max(1, std::min(std::min(G, 16), n_streams));
    for (int g = 0; g < G; ++g) {
      vslam_ctx::Group q;
      for (int k = 0; k < 2; ++k) { q.ev_img[k] = nullptr; q.ev_frm[k] = nullptr; q.ev_emit[k] = nullptr; }
      q.s0 = (int)((long long)n_streams * g / G);
      q.n = (int)((long long)n_streams * (g + 1) / G) - q.s0;
      // VSLAM_PRIO (measurement aid): 1 = frame stream at the highest queue priority, image stream at the lowest; 2 = the reverse
      int p_lo = 0, p_hi = 0, prio = 0;
      if (const char* e = getenv("VSLAM_PRIO")) prio = atoi(e);
      (void)hipDeviceGetStreamPriorityRange(&p_lo, &p_hi);
      const int pf = prio == 1 ? p_hi : (prio == 2 ? p_lo : 0), pi = prio == 1 ? p_lo : (prio == 2 ? p_hi : 0);
      bool ok = hipStreamCreateWithPriority(&q.st_frm, hipStreamNonBlocking, pf) == hipSuccess &&
                hipStreamCreateWithPriority(&q.st_img, hipStreamNonBlocking, pi) == hipSuccess &&
                hipStreamCreateWithPriority(&q.st_img2, hipStreamNonBlocking, pi) == hipSuccess;
      // a second image stream (BRIEF(t) overlapping FAST(t+1)) measured slower on MI355X: opt-in only
      if (ok && !(getenv("VSLAM_IMG_STREAMS") && atoi(getenv("VSLAM_IMG_STREAMS")) == 2)) { (void)hipStreamDestroy(q.st_img2); q.st_img2 = q.st_img; }
      // VSLAM_IMG_STREAMS=0: everything on one HIP stream (no overlap) — measurement aid for stand-alone kernel times
      if (ok && getenv("VSLAM_IMG_STREAMS") && atoi(getenv("VSLAM_IMG_STREAMS")) == 0) { (void)hipStreamDestroy(q.st_img); q.st_img = q.st_img2 = q.st_frm; }
      for (int k = 0; k < 2 && ok; ++k)
        ok = hipEventCreateWithFlags(&q.ev_img[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&q.ev_frm[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&q.ev_emit[k], hipEventDisableTiming) == hipSuccess;
      q.frm_pending[0] = q.frm_pending[1] = false;
      q.emit_pending[0] = q.emit_pending[1] = false;
      if (!ok) { c->own_stream = true; destroy_streams(c); delete c; return fail(nullptr, VSLAM_ERR_HIP, "hipStreamCreate failed"); }
      c->groups.push_back(q);
    }
    c->stream = c->groups[0].st_frm;
    c->stream_img = c->groups[0].st_img;
    c->own_stream = true;
    c->split = n_streams <= VS_SPLIT4_MAX_STREAMS ? 4 : 0;
    if (const char* e = getenv("VSLAM_SPLIT")) c->split = std::max(0, std::min(4, atoi(e)));
    if (c->split == 4 && c->groups.size() != 1) c->split = 2;     // one stream group only
  }
  const DevCfg& d = c->cfg;
  DevBuf& b = c->buf;
  std::memset(&b, 0, sizeof b);
  std::memset(b.active, 0xff, sizeof b.active);
  const size_t B = n_streams, S2 = B * 2, rows = cfg->rows, N = d.NMAX, P = d.MAXP, Hc = d.HCAP;
  hipError_t e = hipSuccess;
#define A(field, count) if (e == hipSuccess) e = dalloc(c, &b.field, (count))
  A(box, S2 * rows * d.bstride); A(score8, S2 * rows * d.bstride); A(mask, S2 * rows * d.TX);
  A(kp_xy, S2 * N * 2); A(kp_score, S2 * N); A(desc, S2 * N * 32); A(n_kp, S2);
  A(rowcell, S2 * rows * (d.CW + 1)); A(used, S2 * N); A(kill, S2 * N);
  A(st, B); A(info, B); A(pose_log, B * VS_POSE_LOG * 12);
  A(p_kp, S2 * P * 4); A(p_desc, S2 * P * 64); A(p_meta, S2 * P * META); A(p_cam, S2 * P * 3); A(p_camlm, S2 * P * 3);
  A(p_lm, S2 * P * 3); A(n_points, S2); A(p_trail, d.trail ? S2 * P * VS_TRAIL : (size_t)64);
  A(proj, B * P * 8); A(proj_q, B * P * 2); A(cand_key, B * P * VS_MAXCAND); A(cand_rkey, B * P * VS_MAXRCAND);
  A(res, B * P * 8); A(trk, B * P * 4); A(lost, B * P);
  A(al_moving, B * P * 3); A(al_fixed, B * P * 4); A(al_omega, B * P); A(al_weight, B * P); A(al_chi, B * P); A(al_inl, B * P);
  A(rec, B * P * 6); A(rec_desc, B * P * 64);
  A(st_match, B * N * 3); A(sc, B * N * 4); A(bin_occ, B * (size_t)d.rows_bin * d.cols_bin); A(sdist, B * N * 16); A(bin_aux, B * (2 * ((size_t)d.rows_bin * d.cols_bin + 1) + N));
  A(h_pose, B * Hc * 24); A(h_cam, B * Hc * P * 4); A(h_prev, B * Hc * P);
#undef A
  for (int q = 0; q < 2 && e == hipSuccess; ++q) {
    vslam_ctx::ImgSet& t = c->sets[q];
    if (q == 0) { t = {b.box, b.score8, b.mask, b.kp_xy, b.kp_score, b.desc, b.n_kp, b.rowcell, b.used, b.sdist, nullptr}; }
    else {
      e = dalloc(c, &t.box, S2 * rows * d.bstride);
      if (e == hipSuccess) e = dalloc(c, &t.score8, S2 * rows * d.bstride);
      if (e == hipSuccess) e = dalloc(c, &t.mask, S2 * rows * d.TX);
      if (e == hipSuccess) e = dalloc(c, &t.kp_xy, S2 * N * 2);
      if (e == hipSuccess) e = dalloc(c, &t.kp_score, S2 * N);
      if (e == hipSuccess) e = dalloc(c, &t.desc, S2 * N * 32);
      if (e == hipSuccess) e = dalloc(c, &t.n_kp, S2);
      if (e == hipSuccess) e = dalloc(c, &t.rowcell, S2 * rows * (d.CW + 1));
      if (e == hipSuccess) e = dalloc(c, &t.used, S2 * N);
      if (e == hipSuccess) e = dalloc(c, &t.sdist, B * N * 16);
    }
    if (e == hipSuccess) e = dalloc(c, &t.iinfo, B);
  }
  if (e == hipSuccess) b.iinfo = c->sets[0].iinfo;
  c->up_stride = d.bstride;
  c->up_stream_stride = (size_t)rows * d.bstride;
  for (int q = 0; q < 2; ++q)
    for (int d2 = 0; d2 < 2; ++d2)
      if (e == hipSuccess) e = dalloc(c, &c->upload[q][d2], B * c->up_stream_stride);
  if (e != hipSuccess) {
    std::string msg = std::string("vslam_create: hipMalloc failed: ") + hipGetErrorString(e);
    for (void* p : c->allocs) (void)hipFree(p);
    destroy_streams(c);     // every group's streams and events, not only the first group's
    delete c;
    return fail(nullptr, VSLAM_ERR_HIP, msg);
  }
  // score8 must read 0 where no corner was ever written only through the mask, box/mask are fully
  // rewritten every frame; nothing else needs initialisation besides the stream state.
  for (int i = 0; i < 6; ++i) (void)hipEventCreate(&c->ev[i]);
  {
    // the frame kernel's view of the configuration and of the buffer table (image pointers excluded: it never reads them)
    const size_t G = c->groups.size();
    calibrate_queues(c);
    e = dalloc(c, &c->d_cfg, 1);
    if (e == hipSuccess) e = dalloc(c, &c->d_bufs, 2 * G);
    if (e == hipSuccess) e = hipMemcpy(c->d_cfg, &c->cfg, sizeof(DevCfg), hipMemcpyHostToDevice);
    if (e == hipSuccess && upload_buffer_tables(c) != VSLAM_OK) e = hipErrorUnknown;
    if (e != hipSuccess) {
      std::string msg = std::string("vslam_create: device tables: ") + hipGetErrorString(e);
      for (void* p : c->allocs) (void)hipFree(p);
      for (int i = 0; i < 6; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
      destroy_streams(c);
      delete c;
      return fail(nullptr, VSLAM_ERR_HIP, msg);
    }
  }
  int rc = init_state(c);
  if (rc != VSLAM_OK) {
    g_create_error = c->err;
    for (void* p : c->allocs) (void)hipFree(p);
    for (int i = 0; i < 6; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    destroy_streams(c);
    delete c;
    return rc;
  }
  *out = c;
  return VSLAM_OK;
}

static void depth_map_free(vslam_ctx* c) {
  vslam_ctx::DepthMap& m = c->dm;
  (void)hipFree(m.depth); (void)hipFree(m.key); (void)hipFree(m.last); (void)hipFree(m.space);
  (void)hipFree(m.row_map); (void)hipFree(m.col_map);
  m = vslam_ctx::DepthMap();
}
VS_API int vslam_create(const vslam_config* cfg, int device, int n_streams, vslam_ctx** out) {
  // a tracker needs room for a keypoint (descriptor border 28 / 31 px); the scratch contexts of the stand-alone entries accept
  // any image, a tiny one simply has no valid pixel
  if (cfg && (cfg->rows < 16 || cfg->cols < 16)) return fail(nullptr, VSLAM_ERR_INVALID, "vslam_create: invalid image dimensions");
  return create_internal(cfg, device, n_streams, out);
}
VS_API void vslam_destroy(vslam_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  sync_all(c);
  for (auto& e : c->scratch) vslam_destroy(e.t);
  c->scratch.clear();
  for (void* p : c->allocs) (void)hipFree(p);
  tmp_free(c);
  depth_map_free(c);
  if (c->report) (void)hipHostFree(c->report);
  for (int q = 0; q < 2; ++q) { if (c->pin_img[q]) (void)hipHostFree(c->pin_img[q]); if (c->pin_ev[q]) (void)hipEventDestroy(c->pin_ev[q]); }
  for (int i = 0; i < 6; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  harvest_events(c);
  for (hipEvent_t e : c->evpool) (void)hipEventDestroy(e);
  for (auto& g : c->groups) {
    for (int q = 0; q < 2; ++q) { (void)hipEventDestroy(g.ev_img[q]); (void)hipEventDestroy(g.ev_frm[q]); (void)hipEventDestroy(g.ev_emit[q]); }
    if (c->own_stream) { if (g.st_img != g.st_frm) (void)hipStreamDestroy(g.st_img); if (g.st_img2 != g.st_img) (void)hipStreamDestroy(g.st_img2); (void)hipStreamDestroy(g.st_frm); }
  }
  delete c;
}
VS_API int vslam_reset(vslam_ctx* c) {
  if (!c) return VSLAM_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  return init_state(c);
}
// ---- per-stream lifetime: whole sequences of different lengths on the streams of one context (exact mode) --------------
VS_API int vslam_set_stream_active(vslam_ctx* c, int s, int active) {
  if (!c) return VSLAM_ERR_INVALID;
  if (s < 0 || s >= c->B) return fail(c, VSLAM_ERR_INVALID, "stream index out of range");
  if (c->frame_begun) return fail(c, VSLAM_ERR_STATE, "vslam_set_stream_active called inside a frame (between vslam_frame_begin and vslam_stereo_new)");
  const uint32_t bit = 1u << (s & 31);
  const bool was = (c->buf.active[s >> 5] & bit) != 0;
  if (was == (active != 0)) return VSLAM_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  sync_all(c);                       // the buffer tables in flight still carry the old mask
  if (active) c->buf.active[s >> 5] |= bit; else c->buf.active[s >> 5] &= ~bit;
  return upload_buffer_tables(c);
}
VS_API int vslam_reset_streams(vslam_ctx* c, int32_t n, const int32_t* streams) {
  if (!c || n < 0 || (n && !streams)) return VSLAM_ERR_INVALID;
  for (int i = 0; i < n; ++i) if (streams[i] < 0 || streams[i] >= c->B) return fail(c, VSLAM_ERR_INVALID, "stream index out of range");
  if (c->frame_begun) return fail(c, VSLAM_ERR_STATE, "vslam_reset_stream called inside a frame");
  if (n == 0) return VSLAM_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->B == 1) { c->pend.flags = 0; c->report_have = 0; }     // the one stream starts over: pending setters belong to the old sequence
  // no host synchronisation: each half of the state is reset in order on the HIP stream(s) that own it, one launch per
  // half for up to 63 streams of a group
  for (auto& g : c->groups) {
    ResetList l;
    l.n = 0;
    auto flush = [&]() -> int {
      if (!l.n) return VSLAM_OK;
      hipLaunchKernelGGL(k_reset_stream_img, dim3(1), dim3(64), 0, g.st_img, c->cfg, c->buf, l);
      if (g.st_img2 != g.st_img) {   // two image streams alternate: the second one must see the reset as well
        hipEvent_t e = ev_get(c);
        HIP_TRY(c, hipEventRecord(e, g.st_img));
        HIP_TRY(c, hipStreamWaitEvent(g.st_img2, e, 0));
        c->evpool.push_back(e);
      }
      hipLaunchKernelGGL(k_reset_stream_trk, dim3(1), dim3(64), 0, g.st_frm, c->cfg, c->buf, l);
      l.n = 0;
      return VSLAM_OK;
    };
    for (int i = 0; i < n; ++i) {
      if (streams[i] < g.s0 || streams[i] >= g.s0 + g.n) continue;
      l.ids[l.n++] = streams[i];
      if (l.n == 63) { int rc = flush(); if (rc) return rc; }
    }
    int rc = flush();
    if (rc) return rc;
  }
  HIP_TRY(c, hipGetLastError());
  return VSLAM_OK;
}
VS_API int vslam_reset_stream(vslam_ctx* c, int s) { const int32_t id = s; return vslam_reset_streams(c, 1, &id); }
VS_API int vslam_copy_current_poses_device(vslam_ctx* c, double* dst) {
  if (!c || !dst) return VSLAM_ERR_INVALID;
  { int rc = flush_pending(c); if (rc) return rc; }
  for (auto& g : c->groups)
    hipLaunchKernelGGL(k_gather_poses, dim3((g.n * 12 + 255) / 256), dim3(256), 0, g.st_frm, buf_set(c, c->last_set, g.s0), g.n, dst + (size_t)g.s0 * 12);
  HIP_TRY(c, hipGetLastError());
  return VSLAM_OK;
}
VS_API int vslam_set_hip_stream(vslam_ctx* c, void* s) {
  if (!c) return VSLAM_ERR_INVALID;
  sync_all(c);
  for (auto& g : c->groups) {
    for (int q = 0; q < 2; ++q) { (void)hipEventDestroy(g.ev_img[q]); (void)hipEventDestroy(g.ev_frm[q]); (void)hipEventDestroy(g.ev_emit[q]); }
    if (c->own_stream) { if (g.st_img != g.st_frm) (void)hipStreamDestroy(g.st_img); if (g.st_img2 != g.st_img) (void)hipStreamDestroy(g.st_img2); (void)hipStreamDestroy(g.st_frm); }
  }
  c->groups.clear();
  // one caller stream: a single group, image pipeline and tracker run back to back on it
  vslam_ctx::Group q;
  q.s0 = 0; q.n = c->B; q.st_frm = (hipStream_t)s; q.st_img = (hipStream_t)s; q.st_img2 = (hipStream_t)s;
  for (int k = 0; k < 2; ++k) { (void)hipEventCreateWithFlags(&q.ev_img[k], hipEventDisableTiming); (void)hipEventCreateWithFlags(&q.ev_frm[k], hipEventDisableTiming); (void)hipEventCreateWithFlags(&q.ev_emit[k], hipEventDisableTiming); }
  q.frm_pending[0] = q.frm_pending[1] = false;
  q.emit_pending[0] = q.emit_pending[1] = false;
  c->groups.push_back(q);
  c->stream = q.st_frm; c->stream_img = q.st_img;
  c->own_stream = false;
  calibrate_queues(c);      // the caller's queue has its own first XCD
  // the frame kernel's buffer table for the single group
  return upload_buffer_tables(c);
}
VS_API int vslam_synchronize(vslam_ctx* c) {
  if (!c) return VSLAM_ERR_INVALID;
  for (auto& g : c->groups) { HIP_TRY(c, hipStreamSynchronize(g.st_img)); HIP_TRY(c, hipStreamSynchronize(g.st_img2)); HIP_TRY(c, hipStreamSynchronize(g.st_frm)); }
  return c->sticky;
}


// ---- launches ------------------------------------------------------------------------------------
static int launch_image_pipeline(vslam_ctx* c) {
  const DevCfg& d = c->cfg;
  const int set = c->parity;
  for (auto& g : c->groups) {
    hipStream_t st = c->img_override ? c->img_override : (set ? g.st_img2 : g.st_img);
    const DevBuf bs = buf_set(c, set, g.s0, c->img_override ? g.q0_frm : (set ? g.q0_img2 : g.q0_img));
    if (!c->img_override && c->img_on_frm_queue) {
      // the last frame's image pipeline ran on the frame queue (stage path) and left no event behind: a caller that switches to
      // the fused path mid-sequence pays one synchronisation here, once
      HIP_TRY(c, hipStreamSynchronize(g.st_frm));
      c->img_on_frm_queue = false;
    }
    if (c->img_override) c->img_on_frm_queue = true;
    const bool coarse = c->img_override != nullptr;     // stage path: detection = [k_fast_box .. k_emit], extraction = k_brief: three events
    // the image products of this set were last read by the frame kernel two steps ago; the detector thresholds come
    // from the controller in k_emit of the previous step (other image stream)
    if (g.frm_pending[set] && st != g.st_frm) HIP_TRY(c, hipStreamWaitEvent(st, g.ev_frm[set], 0));
    if (g.emit_pending[set ^ 1] && (g.st_img != g.st_img2 || c->img_override)) HIP_TRY(c, hipStreamWaitEvent(st, g.ev_emit[set ^ 1], 0));
    dim3 g1(d.TX, (d.c.rows + VS_TILE_H - 1) / VS_TILE_H, 2 * g.n);
    const bool orb = d.c.descriptor_type == VSLAM_DESCRIPTOR_ORB;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (coarse && c->timers) { e0 = ev_get(c); (void)hipEventRecord(e0, st); }
    { KernelTimer t(c, 0, st, true, !coarse); hipLaunchKernelGGL(k_fast_box, g1, dim3(256), VS_FB_DYN_LDS, st, c->cfg, bs); }
    { KernelTimer t(c, 1, st, true, !coarse); hipLaunchKernelGGL(k_emit, dim3(g.n, 2), dim3(512), 0, st, c->cfg, bs, orb ? (int)VSLAM_ORB_BORDER : (int)VSLAM_BRIEF_BORDER, 1); }
    if (e0) { e1 = ev_get(c); (void)hipEventRecord(e1, st); c->evrec.push_back({e0, e1, 0, true}); c->kern_n[1] += 1; }
    if (c->img_override && c->report && c->B == 1) {
      // stage path with a view reader: coordinates and scores leave for the host as soon as k_emit has written them, so that the caller
      // builds its cv::KeyPoint lists while k_brief / k_stereo_dist / k_begin still run (vslam_view_keypoints_xy)
      c->report_xy_seq = ++c->report_seq;
      hipLaunchKernelGGL(k_report, dim3(8), dim3(256), 0, st, c->cfg, bs, 0, (int)VS_REPORT_KEYPOINTS_XY, 0, c->report_xy_seq, c->rl, c->report_dev, c->report_done);
    }
    if (g.st_img != g.st_img2 && !c->img_override) { HIP_TRY(c, hipEventRecord(g.ev_emit[set], st)); g.emit_pending[set] = true; }
    else g.emit_pending[set] = false;
    if (orb) {   // cv::ORB::create() as extractor: Gaussian image (in the box image's memory), steered tests per keypoint
      KernelTimer t(c, 2, st, true, !coarse);
      Gauss7 gk; for (int i = 0; i < 4; ++i) gk.k[i] = d.gauss7[i];
      hipLaunchKernelGGL(k_gauss7, g1, dim3(256), 0, st, c->cfg, bs, gk);
      hipLaunchKernelGGL(k_orb_describe, dim3((d.c.cols + VS_BT_W - 1) / VS_BT_W, (d.c.rows + VS_BT_H - 1) / VS_BT_H, 2 * g.n), dim3(256), 0, st, c->cfg, bs, d.orb_cos, d.orb_sin);
    } else {
      dim3 g3((d.c.cols + VS_BT_W - 1) / VS_BT_W, (d.c.rows + VS_BT_H - 1) / VS_BT_H, 2 * g.n);
      KernelTimer t(c, 2, st, true, !coarse); hipLaunchKernelGGL(k_brief, g3, dim3(256), 0, st, c->cfg, bs);
    }
    if (e1) { hipEvent_t e2 = ev_get(c); (void)hipEventRecord(e2, st); hipEvent_t e1b = e1; c->evshared.push_back({e1b, e2, 2}); }
    // left-right descriptor distances of the first epipolar pass: a product of the images alone, so it is computed
    // here, wide, instead of inside the per-stream frame workgroup
    { KernelTimer t(c, 7, st, true, !coarse); hipLaunchKernelGGL(k_stereo_dist, dim3((d.NMAX + 255) / 256, g.n), dim3(256), 0, st, c->cfg, bs); }
    HIP_TRY(c, hipGetLastError());
    if (st != g.st_frm) { HIP_TRY(c, hipEventRecord(g.ev_img[set], st)); HIP_TRY(c, hipStreamWaitEvent(g.st_frm, g.ev_img[set], 0)); }
  }
  c->last_set = set;
  return VSLAM_OK;
}
static int frame_done(vslam_ctx* c) {
  const int set = c->last_set;
  for (auto& g : c->groups)
    if (g.st_img != g.st_frm) { HIP_TRY(c, hipEventRecord(g.ev_frm[set], g.st_frm)); g.frm_pending[set] = true; }
  c->parity = set ^ 1;
  return VSLAM_OK;
}
// blocks per stream of the candidate kernel (16 point groups each).  Streams differ in cost by an order of magnitude (a
// Localizing stream searches 101 x 101 windows by appearance, a Tracking stream ~31 x 31), so the points are spread over
// many small blocks — about one previous point per 16-lane group at ~700 points — and the hardware scheduler balances
// them: 0.154 -> 0.086 ms back to back at 160 streams of mixed phase against 12 blocks per stream (profiles/r02_*).
static int cand_blocks(const vslam_ctx* c, int n_streams) {
  if (const char* e = getenv("VSLAM_CAND_GX")) return std::max(1, atoi(e));
  (void)c;
  return std::max(4, std::min(128, 7040 / std::max(n_streams, 1)));
}
static int launch_frame(vslam_ctx* c) {
  size_t gi = 0;
  for (auto& g : c->groups) {
    const DevBuf bs = buf_set(c, c->last_set, g.s0, g.q0_frm);
    ConstDevCfg* kc = (ConstDevCfg*)c->d_cfg;
    ConstDevBuf* kb = (ConstDevBuf*)(c->d_bufs + c->last_set * c->groups.size() + gi++);
    const int gx = cand_blocks(c, g.n);
    { KernelTimer t(c, 3, g.st_frm); hipLaunchKernelGGL(k_track_candidates, dim3(gx, g.n), dim3(256), 0, g.st_frm, c->cfg, bs, -1); }
    if (!c->split || (c->split == 2 && !c->cfg.c.enable_landmark_recovery)) {
      KernelTimer t(c, 4, g.st_frm);
      hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, -1);
    } else if (c->split == 3) {
      // registration + prune on CUs of its own (the aligner needs the whole register file), then the tail co-scheduled with the
      // image pipeline of the next frame (k_tail)
      { KernelTimer t(c, 4, g.st_frm); hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, 0); }
      if (c->cfg.c.enable_landmark_recovery) { KernelTimer t(c, 5, g.st_frm); hipLaunchKernelGGL(k_recover_brief, dim3(std::max(4, std::min(64, 1024 / std::max(g.n, 1))), g.n), dim3(256), 0, g.st_frm, c->cfg, bs); }
      { KernelTimer t(c, 6, g.st_frm); hipLaunchKernelGGL(k_tail, dim3(g.n), dim3(VS_TAIL_WG), 0, g.st_frm, kc, kb); }
    } else if (c->split == 4) {
      // few streams on an otherwise idle chip: the landmark refinement (a serial chain per track) leaves the frame's critical path — it runs in
      // workgroups of its own beside the stereo sweep, inside the frame's last launch
      { KernelTimer t(c, 4, g.st_frm, false); hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, 0); }
      if (c->cfg.c.enable_landmark_recovery) { KernelTimer t(c, 5, g.st_frm); hipLaunchKernelGGL(k_recover_brief, dim3(std::max(4, std::min(64, 1024 / std::max(g.n, 1))), g.n), dim3(256), 0, g.st_frm, c->cfg, bs); }
      { KernelTimer t(c, 4, g.st_frm, false); hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, 4); }
      // phase 2 and the landmark refinement in ONE launch: n frame workgroups + G refinement workgroups per stream (k_tail_lm)
      { KernelTimer t(c, 4, g.st_frm); const int G = std::max(1, std::min(16, 64 / std::max(g.n, 1)));
        hipLaunchKernelGGL(k_tail_lm, dim3(g.n * (1 + G)), dim3(VS_WG), 0, g.st_frm, kc, kb, g.n, G); }
    } else if (c->split == 2) {
      { KernelTimer t(c, 4, g.st_frm, false); hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, 0); }
      { KernelTimer t(c, 5, g.st_frm); hipLaunchKernelGGL(k_recover_brief, dim3(std::max(4, std::min(64, 1024 / std::max(g.n, 1))), g.n), dim3(256), 0, g.st_frm, c->cfg, bs); }
      { KernelTimer t(c, 4, g.st_frm); hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, 3); }
    } else {
      { KernelTimer t(c, 4, g.st_frm, false); hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, 0); }
      if (c->cfg.c.enable_landmark_recovery) { KernelTimer t(c, 5, g.st_frm); hipLaunchKernelGGL(k_recover_brief, dim3(std::max(4, std::min(64, 1024 / std::max(g.n, 1))), g.n), dim3(256), 0, g.st_frm, c->cfg, bs); }
      { KernelTimer t(c, 4, g.st_frm, false); hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, 1); }
      { KernelTimer t(c, 6, g.st_frm); hipLaunchKernelGGL(k_update_landmarks, dim3((c->cfg.MAXP + 255) / 256, g.n), dim3(256), 0, g.st_frm, c->cfg, bs, 1); }
      { KernelTimer t(c, 4, g.st_frm); hipLaunchKernelGGL(k_frame, dim3(g.n), dim3(VS_WG), 0, g.st_frm, kc, kb, 2); }
    }
  }
  HIP_TRY(c, hipGetLastError());
  return frame_done(c);
}
static int set_images_device(vslam_ctx* c, const uint8_t* L, const uint8_t* R, int32_t row_stride, size_t image_stride) {
  if (!L || !R) return fail(c, VSLAM_ERR_INVALID, "called with empty frame");  // stereo_framepoint_generator.cpp:75-78
  if (row_stride < c->cfg.c.cols) return fail(c, VSLAM_ERR_INVALID, "row stride smaller than image width");
  c->buf.img[0] = L; c->buf.img[1] = R;
  c->buf.img_row_stride = row_stride;
  c->buf.img_stream_stride = image_stride;
  return VSLAM_OK;
}
static int upload_images(vslam_ctx* c, const uint8_t* L, const uint8_t* R, int32_t row_stride, size_t image_stride) {
  if (!L || !R) return fail(c, VSLAM_ERR_INVALID, "called with empty frame");
  if (row_stride < c->cfg.c.cols) return fail(c, VSLAM_ERR_INVALID, "row stride smaller than image width");
  // Host images of all streams in one (nearly) dense block: one copy per side, the caller's strides kept on the device
  // (2 B strided 2-D copies per step cost more in submission than in transfer).
  const size_t span = (size_t)(c->B - 1) * image_stride + (size_t)(c->cfg.c.rows - 1) * row_stride + c->cfg.c.cols;   // last byte the caller owns
  const size_t dense = (size_t)c->B * c->cfg.c.rows * c->cfg.c.cols;
  const bool ordered = c->B == 1 || image_stride >= (size_t)c->cfg.c.rows * row_stride;
  if (c->groups.size() == 1 && ordered && span <= (size_t)c->B * c->up_stream_stride && span <= dense + dense / 8) {
    hipStream_t st = c->img_override ? c->img_override : (c->parity ? c->groups[0].st_img2 : c->groups[0].st_img);
    // A small pageable source (the literal drop-in: one cv::Mat pair per call) goes through pinned memory of the context: the
    // runtime's own staging of a pageable hipMemcpyAsync costs ~0.12 ms of host time per 467 KB image here, a memcpy into a pinned
    // buffer + a true asynchronous copy ~0.03 ms; the left image's DMA runs while the right one is being staged.
    if (span <= ((size_t)4 << 20)) {
      hipPointerAttribute_t at;
      const bool pinned = hipPointerGetAttributes(&at, L) == hipSuccess && at.type == hipMemoryTypeHost;
      if (!pinned) {
        (void)hipGetLastError();   // "invalid value" for a plain malloc'ed pointer is the expected answer, not an error of this call
        const size_t half = (span + 255) & ~(size_t)255;
        if (c->pin_img_bytes < 2 * half) {
          for (int q = 0; q < 2; ++q) {
            if (c->pin_ev[q]) HIP_TRY(c, hipEventSynchronize(c->pin_ev[q]));
            if (c->pin_img[q]) { (void)hipHostFree(c->pin_img[q]); c->pin_img[q] = nullptr; }
            void* h = nullptr;
            HIP_TRY(c, hipHostMalloc(&h, 2 * half, hipHostMallocDefault));
            c->pin_img[q] = (unsigned char*)h;
            if (!c->pin_ev[q]) HIP_TRY(c, hipEventCreateWithFlags(&c->pin_ev[q], hipEventDisableTiming));
          }
          c->pin_img_bytes = 2 * half;
        } else if (c->pin_used[c->parity]) {
          HIP_TRY(c, hipEventSynchronize(c->pin_ev[c->parity]));     // the copy that last read this staging buffer (two frames ago)
        }
        unsigned char* stage = c->pin_img[c->parity];
        std::memcpy(stage, L, span);
        HIP_TRY(c, hipMemcpyAsync(c->upload[c->parity][0], stage, span, hipMemcpyHostToDevice, st));
        std::memcpy(stage + half, R, span);
        HIP_TRY(c, hipMemcpyAsync(c->upload[c->parity][1], stage + half, span, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipEventRecord(c->pin_ev[c->parity], st));
        c->pin_used[c->parity] = true;
        return set_images_device(c, c->upload[c->parity][0], c->upload[c->parity][1], row_stride, image_stride);
      }
    }
    HIP_TRY(c, hipMemcpyAsync(c->upload[c->parity][0], L, span, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->upload[c->parity][1], R, span, hipMemcpyHostToDevice, st));
    return set_images_device(c, c->upload[c->parity][0], c->upload[c->parity][1], row_stride, image_stride);
  }
  if (row_stride <= c->up_stride) {
    // one contiguous copy per image, rows keep the caller's stride (a pitched host-to-device copy is issued row by row
    // by the runtime: measured 0.13 GB/s against 43 GB/s for the plain copy)
    for (int s = 0; s < c->B; ++s) {
      const vslam_ctx::Group& gg = c->groups[group_of(c, s)];
      hipStream_t st = c->parity ? gg.st_img2 : gg.st_img;
      const size_t bytes = (size_t)(c->cfg.c.rows - 1) * row_stride + c->cfg.c.cols;
      HIP_TRY(c, hipMemcpyAsync(c->upload[c->parity][0] + s * c->up_stream_stride, L + s * image_stride, bytes, hipMemcpyHostToDevice, st));
      HIP_TRY(c, hipMemcpyAsync(c->upload[c->parity][1] + s * c->up_stream_stride, R + s * image_stride, bytes, hipMemcpyHostToDevice, st));
    }
    return set_images_device(c, c->upload[c->parity][0], c->upload[c->parity][1], row_stride, c->up_stream_stride);
  }
  for (int s = 0; s < c->B; ++s) {
    const vslam_ctx::Group& gg = c->groups[group_of(c, s)];
    hipStream_t st = c->parity ? gg.st_img2 : gg.st_img;
    HIP_TRY(c, hipMemcpy2DAsync(c->upload[c->parity][0] + s * c->up_stream_stride, c->up_stride, L + s * image_stride, row_stride,
                                c->cfg.c.cols, c->cfg.c.rows, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpy2DAsync(c->upload[c->parity][1] + s * c->up_stream_stride, c->up_stride, R + s * image_stride, row_stride,
                                c->cfg.c.cols, c->cfg.c.rows, hipMemcpyHostToDevice, st));
  }
  return set_images_device(c, c->upload[c->parity][0], c->upload[c->parity][1], c->up_stride, c->up_stream_stride);
}

VS_API int vslam_process_device(vslam_ctx* c, const uint8_t* L, const uint8_t* R, int32_t row_stride, size_t image_stride) {
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  int rc = set_images_device(c, L, R, row_stride, image_stride);
  if (rc != VSLAM_OK) return rc;
  rc = flush_pending(c);
  if (rc != VSLAM_OK) return rc;
  rc = launch_image_pipeline(c);
  if (rc != VSLAM_OK) return rc;
  return launch_frame(c);
}
VS_API int vslam_process_host(vslam_ctx* c, const uint8_t* L, const uint8_t* R, int32_t row_stride, size_t image_stride) {
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = upload_images(c, L, R, row_stride, image_stride);
  if (rc != VSLAM_OK) return rc;
  rc = flush_pending(c);
  if (rc != VSLAM_OK) return rc;
  rc = launch_image_pipeline(c);
  if (rc != VSLAM_OK) return rc;
  return launch_frame(c);
}

// ---- read-back -----------------------------------------------------------------------------------
static int check_stream(vslam_ctx* c, int s) {
  if (!c) return VSLAM_ERR_INVALID;
  if (s < 0 || s >= c->B) return fail(c, VSLAM_ERR_INVALID, "stream index out of range");
  { int rc = flush_pending(c); if (rc) return rc; }
  sync_all(c);   // read-back: every group's queued work must have finished
  return VSLAM_OK;
}
template <typename T>
static hipError_t d2h(vslam_ctx* c, T* dst, const T* src, size_t count) {
  if (!dst || !count) return hipSuccess;
  return hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, c->stream);
}
VS_API int vslam_get_frame_info(vslam_ctx* c, int s, vslam_frame_info* out) {
  int rc = check_stream(c, s);
  if (rc) return rc;
  if (!out) return fail(c, VSLAM_ERR_INVALID, "null output");
  HIP_TRY(c, d2h(c, out, c->buf.info + s, 1));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (out->error_flags) { c->err = "device buffer capacity exceeded (error_flags != 0)"; }
  return VSLAM_OK;
}
VS_API int vslam_get_keypoints(vslam_ctx* c, int s, int side, int32_t cap, int32_t* n, int16_t* xy, int32_t* score, uint8_t* desc) {
  int rc = check_stream(c, s);
  if (rc) return rc;
  if (!n || side < 0 || side > 1) return fail(c, VSLAM_ERR_INVALID, "bad argument");
  int32_t cnt = 0;
  const vslam_ctx::ImgSet& iset = c->sets[c->last_set];
  HIP_TRY(c, hipStreamSynchronize(c->stream_img));
  HIP_TRY(c, d2h(c, &cnt, iset.n_kp + s * 2 + side, 1));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *n = cnt;
  if (cnt > cap) return fail(c, VSLAM_ERR_CAPACITY, "keypoint output capacity too small");
  const size_t o = ((size_t)s * 2 + side) * c->cfg.NMAX;
  std::vector<uint8_t> sc(cnt);
  HIP_TRY(c, d2h(c, xy, iset.kp_xy + o * 2, (size_t)cnt * 2));
  HIP_TRY(c, d2h(c, sc.data(), iset.kp_score + o, (size_t)cnt));
  HIP_TRY(c, d2h(c, desc, iset.desc + o * 32, (size_t)cnt * 32));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (score) for (int i = 0; i < cnt; ++i) score[i] = sc[i];
  return VSLAM_OK;
}
static int get_points_impl(vslam_ctx* c, int s, int in_progress, int32_t cap, int32_t* n, int16_t* kp, int32_t* meta, double* cam, double* lm,
                           uint8_t* desc) {
  int rc = check_stream(c, s);
  if (rc) return rc;
  if (!n) return fail(c, VSLAM_ERR_INVALID, "bad argument");
  StreamState st;
  HIP_TRY(c, d2h(c, &st, c->buf.st + s, 1));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int32_t cnt = 0;
  const int pb = in_progress ? (st.cur ^ 1) : st.cur;
  if (in_progress) cnt = st.n_cur;
  else {
    HIP_TRY(c, d2h(c, &cnt, c->buf.n_points + s * 2 + st.cur, 1));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!st.has_prev) cnt = 0;
  }
  *n = cnt;
  if (cnt > cap) return fail(c, VSLAM_ERR_CAPACITY, "point output capacity too small");
  const size_t o = ((size_t)s * 2 + pb) * c->cfg.MAXP;
  std::vector<int32_t> m((size_t)cnt * META);
  std::vector<int16_t> k((size_t)cnt * 4);
  HIP_TRY(c, d2h(c, k.data(), c->buf.p_kp + o * 4, (size_t)cnt * 4));
  HIP_TRY(c, d2h(c, m.data(), c->buf.p_meta + o * META, (size_t)cnt * META));
  HIP_TRY(c, d2h(c, cam, c->buf.p_cam + o * 3, (size_t)cnt * 3));
  HIP_TRY(c, d2h(c, lm, c->buf.p_lm + o * 3, (size_t)cnt * 3));
  HIP_TRY(c, d2h(c, desc, c->buf.p_desc + o * 64, (size_t)cnt * 64));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < cnt; ++i) {
    if (kp) for (int q = 0; q < 4; ++q) kp[4 * i + q] = k[4 * i + q];
    if (meta) {
      meta[6 * i + 0] = m[META * i + M_DIST]; meta[6 * i + 1] = m[META * i + M_EPI]; meta[6 * i + 2] = m[META * i + M_PREV];
      meta[6 * i + 3] = m[META * i + M_TLEN]; meta[6 * i + 4] = m[META * i + M_LMUP]; meta[6 * i + 5] = k[4 * i] - k[4 * i + 2];
    }
    if (lm && m[META * i + M_LMUP] == 0) { lm[3 * i] = lm[3 * i + 1] = lm[3 * i + 2] = 0; }
  }
  return VSLAM_OK;
}
VS_API int vslam_get_points(vslam_ctx* c, int s, int32_t cap, int32_t* n, int16_t* kp, int32_t* meta, double* cam, double* lm) {
  return get_points_impl(c, s, 0, cap, n, kp, meta, cam, lm, nullptr);
}
VS_API int vslam_get_frame_points(vslam_ctx* c, int s, int in_progress, int32_t cap, int32_t* n, int16_t* kp, int32_t* meta, double* cam,
                                  double* lm, uint8_t* desc) {
  return get_points_impl(c, s, in_progress, cap, n, kp, meta, cam, lm, desc);
}
VS_API int vslam_get_track_result(vslam_ctx* c, int s, int32_t cap, int32_t* n_tracked, int32_t* out4, int32_t* n_lost, int32_t* lost) {
  int rc = check_stream(c, s);
  if (rc) return rc;
  if (!n_tracked || !n_lost) return fail(c, VSLAM_ERR_INVALID, "bad argument");
  StreamState st;
  HIP_TRY(c, d2h(c, &st, c->buf.st + s, 1));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *n_tracked = st.n_trk; *n_lost = st.n_lost;
  if (st.n_trk > cap || st.n_lost > cap) return fail(c, VSLAM_ERR_CAPACITY, "track output capacity too small");
  HIP_TRY(c, d2h(c, out4, c->buf.trk + (size_t)s * c->cfg.MAXP * 4, (size_t)st.n_trk * 4));
  HIP_TRY(c, d2h(c, lost, c->buf.lost + (size_t)s * c->cfg.MAXP, (size_t)st.n_lost));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VSLAM_OK;
}
VS_API int vslam_get_aligner_result(vslam_ctx* c, int s, int32_t cap, int32_t* n, double* chi, uint8_t* inlier, double T[12], double H[36]) {
  int rc = check_stream(c, s);
  if (rc) return rc;
  if (!n) return fail(c, VSLAM_ERR_INVALID, "bad argument");
  StreamState st;
  HIP_TRY(c, d2h(c, &st, c->buf.st + s, 1));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *n = st.al_n;
  if (st.al_n > cap) return fail(c, VSLAM_ERR_CAPACITY, "aligner output capacity too small");
  HIP_TRY(c, d2h(c, chi, c->buf.al_chi + (size_t)s * c->cfg.MAXP, (size_t)st.al_n));
  HIP_TRY(c, d2h(c, inlier, c->buf.al_inl + (size_t)s * c->cfg.MAXP, (size_t)st.al_n));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (T) std::memcpy(T, st.al_T, sizeof(double) * 12);
  if (H) std::memcpy(H, st.al_H, sizeof(double) * 36);
  return VSLAM_OK;
}
VS_API int vslam_get_aligner_weights(vslam_ctx* c, int s, int32_t cap, int32_t* n, double* weight) {
  int rc = check_stream(c, s);
  if (rc) return rc;
  if (!n) return fail(c, VSLAM_ERR_INVALID, "bad argument");
  StreamState st;
  HIP_TRY(c, d2h(c, &st, c->buf.st + s, 1));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *n = st.al_wsize;
  if (st.al_wsize > cap) return fail(c, VSLAM_ERR_CAPACITY, "aligner weight output capacity too small");
  HIP_TRY(c, d2h(c, weight, c->buf.al_weight + (size_t)s * c->cfg.MAXP, (size_t)st.al_wsize));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VSLAM_OK;
}
VS_API int vslam_aligner_weights(vslam_ctx* c, int32_t n_calls, const int32_t* n, const int32_t* inverse_depth, const double* depth, double* out) {
  tmp_reset(c);
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (n_calls < 0 || (n_calls && (!n || !inverse_depth))) return fail(c, VSLAM_ERR_INVALID, "aligner_weights: bad argument");
  size_t total = 0; int nmax = 0;
  for (int k = 0; k < n_calls; ++k) { if (n[k] < 0) return fail(c, VSLAM_ERR_INVALID, "aligner_weights: negative size"); total += (size_t)n[k]; nmax = std::max(nmax, n[k]); }
  if (total && (!depth || !out)) return fail(c, VSLAM_ERR_INVALID, "aligner_weights: bad argument");
  if (!n_calls || !total) return VSLAM_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  int32_t *dn = nullptr, *di = nullptr; double *dd = nullptr, *dw = nullptr, *dout = nullptr;
  hipError_t e = tmp_get(c, (void**)&dn, (size_t)n_calls * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&di, (size_t)n_calls * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dd, total * 8);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dw, (size_t)nmax * 8);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dout, total * 8);
  if (e == hipSuccess) e = hipMemcpyAsync(dn, n, (size_t)n_calls * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(di, inverse_depth, (size_t)n_calls * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dd, depth, total * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_aligner_weights, dim3(1), dim3(256), 0, c->stream, n_calls, dn, di, dd, c->cfg.c.maximum_reliable_depth_meters, dw, dout);
    e = hipMemcpyAsync(out, dout, total * 8, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}
// ---- RGB-D components (DepthFramePointGenerator pieces, stand-alone) ------------------------------------------------
static int make_scratch_ctx(vslam_ctx* parent, int rows, int cols, int nmax, int maxp, vslam_ctx** out);
static int depth_params_ok(vslam_ctx* c, const vslam_depth_params* p) {
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!p || p->rows <= 0 || p->cols <= 0 || p->rows > 32767 || p->cols > 32767) return fail(c, VSLAM_ERR_INVALID, "depth: image size out of range");
  if (!(p->maximum_depth_meters > 0) || (p->enable_keypoint_binning && p->bin_size_pixels <= 0)) return fail(c, VSLAM_ERR_INVALID, "depth: bad parameters");
  return VSLAM_OK;
}
VS_API int vslam_depth_space_map(vslam_ctx* c, const vslam_depth_params* p, const uint16_t* depth, int32_t row_stride, float* space,
                                 int16_t* row_map, int16_t* col_map) {
  int rc = depth_params_ok(c, p);
  if (rc != VSLAM_OK) return rc;
  if (!depth) return fail(c, VSLAM_ERR_INVALID, "depth tracker requires a 16bit mono image to encode depth");   // :411-413
  if (row_stride < p->cols) return fail(c, VSLAM_ERR_INVALID, "row stride smaller than image width");
  HIP_TRY(c, hipSetDevice(c->device));
  vslam_ctx::DepthMap& m = c->dm;
  const size_t n = (size_t)p->rows * p->cols;
  if (m.rows != p->rows || m.cols != p->cols) {
    depth_map_free(c);
    hipError_t e = hipMalloc((void**)&m.depth, n * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc((void**)&m.key, n * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&m.last, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&m.space, n * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m.row_map, n * sizeof(int16_t));
    if (e == hipSuccess) e = hipMalloc((void**)&m.col_map, n * sizeof(int16_t));
    if (e != hipSuccess) { depth_map_free(c); return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e)); }
    m.rows = p->rows; m.cols = p->cols;
  }
  m.valid = false;
  // rows re-packed on the device side of the copy (dense device image, stride = cols)
  if (row_stride == p->cols) HIP_TRY(c, hipMemcpyAsync(m.depth, depth, n * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
  else HIP_TRY(c, hipMemcpy2DAsync(m.depth, (size_t)p->cols * 2, depth, (size_t)row_stride * 2, (size_t)p->cols * 2, p->rows, hipMemcpyHostToDevice, c->stream));
  const float f0 = (float)p->maximum_depth_meters;
  uint32_t f0_bits;
  std::memcpy(&f0_bits, &f0, 4);
  const dim3 grid((p->cols + 255) / 256, p->rows);
  hipLaunchKernelGGL(k_depth_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (int)n, f0_bits, m.key, m.last);
  hipLaunchKernelGGL(k_depth_min, grid, dim3(256), 0, c->stream, *p, m.depth, p->cols, m.key, (const int32_t*)nullptr);
  hipLaunchKernelGGL(k_depth_pick, grid, dim3(256), 0, c->stream, *p, m.depth, p->cols, f0_bits, m.key, m.last, (const int32_t*)nullptr);
  hipLaunchKernelGGL(k_depth_write, grid, dim3(256), 0, c->stream, *p, m.depth, p->cols, f0_bits, m.key, m.last, m.space, m.row_map, m.col_map, 0, (const int32_t*)nullptr);
  HIP_TRY(c, hipGetLastError());
  if (space) HIP_TRY(c, hipMemcpyAsync(space, m.space, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (row_map) HIP_TRY(c, hipMemcpyAsync(row_map, m.row_map, n * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  if (col_map) HIP_TRY(c, hipMemcpyAsync(col_map, m.col_map, n * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  m.valid = true;
  return VSLAM_OK;
}
VS_API int vslam_depth_compute(vslam_ctx* c, const vslam_depth_params* p, const float* space, int32_t nF, const int32_t* rcF, int32_t nT,
                               const int32_t* rcT, int32_t cap, int32_t* n_new, int32_t* new_feat, double* new_xyz, int32_t* n_temp,
                               int32_t* temp_feat, double* temp_xyz) {
  tmp_reset(c);
  int rc = depth_params_ok(c, p);
  if (rc != VSLAM_OK) return rc;
  if (nF < 0 || nT < 0 || cap < 0 || !n_new || !n_temp || (nF && !rcF) || (nT && !rcT) || (cap && (!new_feat || !new_xyz || !temp_feat || !temp_xyz)))
    return fail(c, VSLAM_ERR_INVALID, "depth_compute: bad argument");
  for (int i = 0; i < nF; ++i) if (rcF[2 * i] < 0 || rcF[2 * i] >= p->rows || rcF[2 * i + 1] < 0 || rcF[2 * i + 1] >= p->cols) return fail(c, VSLAM_ERR_INVALID, "depth_compute: feature outside the image");
  for (int i = 0; i < nT; ++i) if (rcT[2 * i] < 0 || rcT[2 * i] >= p->rows || rcT[2 * i + 1] < 0 || rcT[2 * i + 1] >= p->cols) return fail(c, VSLAM_ERR_INVALID, "depth_compute: point outside the image");
  if (!space && !(c->dm.valid && c->dm.rows == p->rows && c->dm.cols == p->cols)) return fail(c, VSLAM_ERR_STATE, "depth_compute: no resident space map of this size");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = (size_t)p->rows * p->cols;
  const int rows_bin = p->enable_keypoint_binning ? p->rows / p->bin_size_pixels + 1 : 0;   // base_framepoint_generator.cpp:304-305
  const int cols_bin = p->enable_keypoint_binning ? p->cols / p->bin_size_pixels + 1 : 0;
  const int n_bins = (rows_bin + 1) * (cols_bin + 1);
  float* dspace = nullptr; int32_t *dF = nullptr, *dT = nullptr, *dcnt = nullptr, *dnf = nullptr, *dtf = nullptr;
  double *dnx = nullptr, *dtx = nullptr; unsigned long long* dbins = nullptr; uint8_t* dcls = nullptr;
  const size_t capa = std::max(cap, 1);
  hipError_t e = hipSuccess;
  if (space) { e = tmp_get(c, (void**)&dspace, n * 3 * sizeof(float)); if (e == hipSuccess) e = hipMemcpyAsync(dspace, space, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream); }
  if (e == hipSuccess) e = tmp_get(c, (void**)&dF, std::max(nF, 1) * 2 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dT, std::max(nT, 1) * 2 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dcnt, 2 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dnf, capa * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dtf, capa * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dnx, capa * 3 * sizeof(double));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dtx, capa * 3 * sizeof(double));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dbins, (size_t)n_bins * sizeof(unsigned long long));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dcls, std::max(nF, 1));
  if (e == hipSuccess && nF) e = hipMemcpyAsync(dF, rcF, (size_t)nF * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && nT) e = hipMemcpyAsync(dT, rcT, (size_t)nT * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
  int32_t cnt[2] = {0, 0};
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_depth_compute, dim3(1), dim3(1024), 0, c->stream, *p, space ? dspace : c->dm.space, nF, dF, nT, dT, dbins, n_bins,
                       rows_bin, cols_bin, cap, dcnt, dnf, dnx, dtf, dtx, dcls);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(cnt, dcnt, sizeof cnt, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) {
    *n_new = cnt[0]; *n_temp = cnt[1];
    const int a = std::min(cnt[0], cap), b = std::min(cnt[1], cap);
    if (a) { e = hipMemcpy(new_feat, dnf, (size_t)a * sizeof(int32_t), hipMemcpyDeviceToHost); if (e == hipSuccess) e = hipMemcpy(new_xyz, dnx, (size_t)a * 3 * sizeof(double), hipMemcpyDeviceToHost); }
    if (e == hipSuccess && b) { e = hipMemcpy(temp_feat, dtf, (size_t)b * sizeof(int32_t), hipMemcpyDeviceToHost); if (e == hipSuccess) e = hipMemcpy(temp_xyz, dtx, (size_t)b * 3 * sizeof(double), hipMemcpyDeviceToHost); }
  }

  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  if (cnt[0] > cap || cnt[1] > cap) return fail(c, VSLAM_ERR_CAPACITY, "depth_compute: output capacity too small");
  return VSLAM_OK;
}
VS_API int vslam_depth_track(vslam_ctx* c, const vslam_depth_params* p, const float* space, const double T[12], int32_t d, double tau,
                             int32_t by_appearance, int32_t nP, const double* cam, const uint8_t* pdesc, const uint8_t* pflags, int32_t nL,
                             const int32_t* rcL, const uint8_t* dL, int32_t* n_tracked, int32_t* out2, double* xyz, int32_t* n_temp,
                             int32_t* temp2, int32_t* n_lost, int32_t* lost, int32_t* n_tracked_landmarks) {
  tmp_reset(c);
  int rc = depth_params_ok(c, p);
  if (rc != VSLAM_OK) return rc;
  if (!T || d < 0 || nP < 0 || nL < 0 || !n_tracked || !n_temp || !n_lost || !n_tracked_landmarks || (nP && (!cam || !pdesc || !pflags || !out2 || !xyz || !temp2 || !lost)) ||
      (nL && (!rcL || !dL)))
    return fail(c, VSLAM_ERR_INVALID, "depth_track: bad argument");
  if (!space && !(c->dm.valid && c->dm.rows == p->rows && c->dm.cols == p->cols)) return fail(c, VSLAM_ERR_STATE, "depth_track: no resident space map of this size");
  const int rows = p->rows, cols = p->cols, CW = (cols + 15) / 16, CW1 = CW + 1;
  // features row-major + (row, 16-px cell) CSR, as the image pipeline leaves them (k_emit).  Several features on ONE pixel (an OrbDetector
  // finds a corner on more than one pyramid level): setFeatures (intensity_feature_matcher.cpp:48-70) writes them into the lattice in list
  // order, so only the LAST one can ever be found through the lattice — the others stay in the feature vector (compute() still sees them) but
  // are invisible to track(), also after the last one has been taken.
  std::vector<int> ord(nL);
  for (int i = 0; i < nL; ++i) {
    ord[i] = i;
    if (rcL[2 * i] < 0 || rcL[2 * i] >= rows || rcL[2 * i + 1] < 0 || rcL[2 * i + 1] >= cols) return fail(c, VSLAM_ERR_INVALID, "feature outside the image");
  }
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return rcL[2 * a] != rcL[2 * b] ? rcL[2 * a] < rcL[2 * b] : (rcL[2 * a + 1] != rcL[2 * b + 1] ? rcL[2 * a + 1] < rcL[2 * b + 1] : a < b); });
  std::vector<int16_t> xy((size_t)std::max(nL, 1) * 2);
  std::vector<uint8_t> ds((size_t)std::max(nL, 1) * 32);
  std::vector<uint8_t> vis(std::max(nL, 1), 1);
  bool duplicates = false;
  std::vector<int32_t> rowcell((size_t)rows * CW1);
  for (int k = 0; k < nL; ++k) {
    xy[2 * k] = (int16_t)rcL[2 * ord[k] + 1]; xy[2 * k + 1] = (int16_t)rcL[2 * ord[k]];
    std::memcpy(&ds[(size_t)32 * k], dL + (size_t)32 * ord[k], 32);
    if (k + 1 < nL && rcL[2 * ord[k]] == rcL[2 * ord[k + 1]] && rcL[2 * ord[k] + 1] == rcL[2 * ord[k + 1] + 1]) { vis[k] = 0; duplicates = true; }
  }
  for (int r = 0, k = 0; r < rows; ++r)
    for (int cc = 0; cc < CW1; ++cc) {
      while (k < nL && (xy[2 * k + 1] < r || (xy[2 * k + 1] == r && xy[2 * k] < 16 * cc))) ++k;
      rowcell[(size_t)r * CW1 + cc] = k;
    }
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = (size_t)rows * cols, P1 = std::max(nP, 1), L1 = std::max(nL, 1);
  DepthTrack a;
  std::memset(&a, 0, sizeof a);
  a.p = *p; std::memcpy(a.T, T, sizeof a.T); a.d = d; a.by_app = by_appearance ? 1 : 0; a.tau = tau; a.nP = nP; a.nL = nL; a.CW = CW;
  float* dspace = nullptr; double *dcam = nullptr, *dxyz = nullptr; uint8_t *dpd = nullptr, *dpf = nullptr, *dds = nullptr; int16_t* dxy = nullptr;
  int32_t *drc = nullptr, *dhold = nullptr, *dpick = nullptr, *dcnt = nullptr, *dout2 = nullptr, *dtmp2 = nullptr, *dlost = nullptr;
  unsigned long long* dcand = nullptr;
  uint8_t* dvis = nullptr;
  hipError_t e = hipSuccess;
  if (space) { e = tmp_get(c, (void**)&dspace, n * 3 * sizeof(float)); if (e == hipSuccess) e = hipMemcpyAsync(dspace, space, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream); }
  if (e == hipSuccess && duplicates) { e = tmp_get(c, (void**)&dvis, L1); if (e == hipSuccess) e = hipMemcpyAsync(dvis, vis.data(), (size_t)nL, hipMemcpyHostToDevice, c->stream); }
  if (e == hipSuccess) e = tmp_get(c, (void**)&dcand, P1 * (VS_DT_K + 1) * sizeof(unsigned long long));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dcam, P1 * 3 * sizeof(double));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dxyz, P1 * 3 * sizeof(double));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dpd, P1 * 32);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dpf, P1);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dds, L1 * 32);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dxy, L1 * 2 * sizeof(int16_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&drc, rowcell.size() * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dhold, L1 * 2 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dpick, P1 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dcnt, 4 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dout2, P1 * 2 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dtmp2, P1 * 2 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dlost, P1 * sizeof(int32_t));
  if (e == hipSuccess && nP) e = hipMemcpyAsync(dcam, cam, (size_t)nP * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && nP) e = hipMemcpyAsync(dpd, pdesc, (size_t)nP * 32, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && nP) e = hipMemcpyAsync(dpf, pflags, (size_t)nP, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && nL) e = hipMemcpyAsync(dds, ds.data(), (size_t)nL * 32, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && nL) e = hipMemcpyAsync(dxy, xy.data(), (size_t)nL * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(drc, rowcell.data(), rowcell.size() * 4, hipMemcpyHostToDevice, c->stream);
  int32_t cnt[4] = {0, 0, 0, 0};
  if (e == hipSuccess) {
    a.cam = dcam; a.pdesc = dpd; a.pflags = dpf; a.kxy = dxy; a.desc = dds; a.rowcell = drc; a.space = space ? dspace : c->dm.space; a.fvis = dvis;
    a.hold = dhold; a.pick = dpick; a.cand = dcand; a.counts = dcnt; a.out2 = dout2; a.xyz = dxyz; a.temp2 = dtmp2; a.lost = dlost;
    if (nP) hipLaunchKernelGGL(k_depth_track_candidates, dim3(std::min(1024, (nP + 15) / 16)), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_depth_track, dim3(1), dim3(1024), 0, c->stream, a);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(cnt, dcnt, sizeof cnt, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // also: the host staging vectors may go out of scope now
  if (e == hipSuccess) {
    *n_tracked = cnt[0]; *n_temp = cnt[1]; *n_lost = cnt[2]; *n_tracked_landmarks = cnt[3];
    if (cnt[0]) { e = hipMemcpy(out2, dout2, (size_t)cnt[0] * 8, hipMemcpyDeviceToHost); if (e == hipSuccess) e = hipMemcpy(xyz, dxyz, (size_t)cnt[0] * 24, hipMemcpyDeviceToHost); }
    if (e == hipSuccess && cnt[1]) e = hipMemcpy(temp2, dtmp2, (size_t)cnt[1] * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && cnt[2]) e = hipMemcpy(lost, dlost, (size_t)cnt[2] * 4, hipMemcpyDeviceToHost);
    for (int u = 0; u < cnt[0]; ++u) out2[2 * u + 1] = ord[out2[2 * u + 1]];     // back to the caller's feature numbering
    for (int u = 0; u < cnt[1]; ++u) temp2[2 * u + 1] = ord[temp2[2 * u + 1]];
  }

  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}
VS_API int vslam_depth_recover(vslam_ctx* c, const vslam_depth_params* p, const float* space, const uint8_t* img, int32_t row_stride,
                               const double w2c[12], int32_t n, const uint8_t* has_lm, const double* lm, const uint8_t* pdesc, float kp_size,
                               double tau, int32_t* n_rec, int32_t* rec_index, float* rec_xy, uint8_t* rec_desc, double* rec_xyz) {
  tmp_reset(c);
  int rc = depth_params_ok(c, p);
  if (rc != VSLAM_OK) return rc;
  if (!img || !w2c || n < 0 || !n_rec || (n && (!has_lm || !lm || !pdesc || !rec_index || !rec_xy || !rec_desc || !rec_xyz)))
    return fail(c, VSLAM_ERR_INVALID, "depth_recover: bad argument");
  if (row_stride < p->cols) return fail(c, VSLAM_ERR_INVALID, "row stride smaller than image width");
  if (!space && !(c->dm.valid && c->dm.rows == p->rows && c->dm.cols == p->cols)) return fail(c, VSLAM_ERR_STATE, "depth_recover: no resident space map of this size");
  *n_rec = 0;
  if (n == 0) return VSLAM_OK;
  // box image of the left image through the image pipeline's own kernel (scratch context of the image size)
  vslam_ctx* t = nullptr;
  rc = make_scratch_ctx(c, p->rows, p->cols, 64, 64, &t);
  if (rc != VSLAM_OK) return rc;
  const size_t npx = (size_t)p->rows * p->cols;
  DepthRecover a;
  std::memset(&a, 0, sizeof a);
  a.p = *p; std::memcpy(a.w2c, w2c, sizeof a.w2c); a.kp_size = kp_size; a.tau = tau; a.n = n;
  float *dspace = nullptr, *dkxy = nullptr, *drxy = nullptr; double *dlm = nullptr, *drxyz = nullptr; uint8_t *dhl = nullptr, *dpd = nullptr, *dkeep = nullptr, *ddesc = nullptr, *drdesc = nullptr;
  int16_t* dbxy = nullptr; int32_t *dcell = nullptr, *dcnt = nullptr, *dridx = nullptr;
  hipError_t e = hipSuccess;
  if (space) { e = tmp_alloc(c, &dspace, npx * 3); if (e == hipSuccess) e = hipMemcpyAsync(dspace, space, npx * 3 * sizeof(float), hipMemcpyHostToDevice, t->stream_img); }
  if (e == hipSuccess) e = tmp_alloc(c, &dkxy, (size_t)n * 2);
  if (e == hipSuccess) e = tmp_alloc(c, &drxy, (size_t)n * 2);
  if (e == hipSuccess) e = tmp_alloc(c, &dlm, (size_t)n * 3);
  if (e == hipSuccess) e = tmp_alloc(c, &drxyz, (size_t)n * 3);
  if (e == hipSuccess) e = tmp_alloc(c, &dhl, (size_t)n);
  if (e == hipSuccess) e = tmp_alloc(c, &dpd, (size_t)n * 32);
  if (e == hipSuccess) e = tmp_alloc(c, &dkeep, (size_t)n);
  if (e == hipSuccess) e = tmp_alloc(c, &ddesc, (size_t)n * 32);
  if (e == hipSuccess) e = tmp_alloc(c, &drdesc, (size_t)n * 32);
  if (e == hipSuccess) e = tmp_alloc(c, &dbxy, (size_t)n * 2);
  if (e == hipSuccess) e = tmp_alloc(c, &dcell, (size_t)n);
  if (e == hipSuccess) e = tmp_alloc(c, &dcnt, 1);
  if (e == hipSuccess) e = tmp_alloc(c, &dridx, (size_t)n);
  if (e == hipSuccess) e = hipMemcpyAsync(dhl, has_lm, (size_t)n, hipMemcpyHostToDevice, t->stream_img);
  if (e == hipSuccess) e = hipMemcpyAsync(dlm, lm, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, t->stream_img);
  if (e == hipSuccess) e = hipMemcpyAsync(dpd, pdesc, (size_t)n * 32, hipMemcpyHostToDevice, t->stream_img);
  rc = e == hipSuccess ? upload_images(t, img, img, row_stride, 0) : fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  if (rc == VSLAM_OK) {
    if (!space) (void)hipStreamSynchronize(c->stream);   // the resident map was written on the parent's stream
    a.has_lm = dhl; a.lm = dlm; a.pdesc = dpd; a.space = space ? dspace : c->dm.space; a.bxy = dbxy; a.kxy = dkxy; a.cell = dcell;
    a.keep = dkeep; a.desc = ddesc; a.count = dcnt; a.rec_index = dridx; a.rec_xy = drxy; a.rec_desc = drdesc; a.rec_xyz = drxyz;
    hipLaunchKernelGGL(k_depth_recover_project, dim3((n + 255) / 256), dim3(256), 0, t->stream_img, a);
    if (p->descriptor_type == VSLAM_DESCRIPTOR_ORB) {
      // cv::ORB::create() as extractor: Gaussian image (in the scratch context's box memory), steered tests at the rounded pixels
      uint8_t* dblur = reinterpret_cast<uint8_t*>(t->buf.box);
      Gauss7 gk; for (int i = 0; i < 4; ++i) gk.k[i] = t->cfg.gauss7[i];
      hipLaunchKernelGGL(k_gauss7_plain, dim3((p->cols + VS_TILE_W - 1) / VS_TILE_W, (p->rows + VS_TILE_H - 1) / VS_TILE_H), dim3(256), 0, t->stream_img,
                         t->buf.img[0], t->buf.img_row_stride, p->rows, p->cols, gk, dblur, t->cfg.bstride);
      hipLaunchKernelGGL(k_orb_at, dim3(std::min(64, (n + 3) / 4)), dim3(256), 0, t->stream_img, dblur, t->cfg.bstride, p->rows, p->cols, n, dbxy, t->cfg.orb_cos, t->cfg.orb_sin, dkeep, ddesc);
    } else {
      dim3 g1(t->cfg.TX, (p->rows + VS_TILE_H - 1) / VS_TILE_H, 2);
      hipLaunchKernelGGL(k_fast_box, g1, dim3(256), VS_FB_DYN_LDS, t->stream_img, t->cfg, t->buf);
      hipLaunchKernelGGL(k_brief_at, dim3(std::min(64, (n + 3) / 4)), dim3(256), 0, t->stream_img, t->buf.box, t->cfg.bstride, p->rows, p->cols, n, dbxy, dkeep, ddesc);
    }
    hipLaunchKernelGGL(k_depth_recover_finish, dim3(1), dim3(1024), 0, t->stream_img, a);
    e = hipGetLastError();
    int32_t cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, dcnt, 4, hipMemcpyDeviceToHost, t->stream_img);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream_img);
    if (e == hipSuccess && cnt) {
      e = hipMemcpy(rec_index, dridx, (size_t)cnt * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(rec_xy, drxy, (size_t)cnt * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(rec_desc, drdesc, (size_t)cnt * 32, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(rec_xyz, drxyz, (size_t)cnt * 24, hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess) *n_rec = cnt;
    else rc = fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  }
  scratch_put(c, t);
  return rc;
}
VS_API int vslam_point_in_camera(vslam_ctx* c, int32_t n, const float* xp, const float* xc, const double T[12], const double K[9], double* out) {
  tmp_reset(c);
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (n < 0 || !T || !K || (n && (!xp || !xc || !out))) return fail(c, VSLAM_ERR_INVALID, "point_in_camera: bad argument");
  if (n == 0) return VSLAM_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  float *dp = nullptr, *dc = nullptr; double *dT = nullptr, *dK = nullptr, *dout = nullptr;
  hipError_t e = tmp_get(c, (void**)&dp, (size_t)n * 2 * sizeof(float));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dc, (size_t)n * 2 * sizeof(float));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dT, 12 * sizeof(double));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dK, 9 * sizeof(double));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dout, (size_t)n * 3 * sizeof(double));
  if (e == hipSuccess) e = hipMemcpyAsync(dp, xp, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dc, xc, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dT, T, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dK, K, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_point_in_camera, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dp, dc, dT, dK, dout);
    e = hipMemcpyAsync(out, dout, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}

VS_API int vslam_landmark_update(vslam_ctx* c, int32_t n, const int32_t* offsets, const int32_t* frame_of, int32_t n_frames, const double* w2c,
                                 const double* c2w, const double* cam, double* world, int32_t* updates) {
  tmp_reset(c);
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (n < 0 || n_frames < 0 || (n && (!offsets || !world || !updates))) return fail(c, VSLAM_ERR_INVALID, "landmark_update: bad argument");
  if (n == 0) return VSLAM_OK;
  const int M = offsets[n];
  if (M < 0 || (M && (!frame_of || !w2c || !c2w || !cam))) return fail(c, VSLAM_ERR_INVALID, "landmark_update: bad argument");
  for (int i = 0; i < n; ++i) if (offsets[i] > offsets[i + 1] || offsets[i] < 0) return fail(c, VSLAM_ERR_INVALID, "landmark_update: offsets not ascending");
  for (int m = 0; m < M; ++m) if (frame_of[m] < 0 || frame_of[m] >= n_frames) return fail(c, VSLAM_ERR_INVALID, "landmark_update: frame index out of range");
  HIP_TRY(c, hipSetDevice(c->device));
  int32_t *doff = nullptr, *dfo = nullptr, *dup = nullptr; double *dw2c = nullptr, *dc2w = nullptr, *dcam = nullptr, *dworld = nullptr;
  hipError_t e = tmp_get(c, (void**)&doff, (size_t)(n + 1) * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dfo, std::max<size_t>(M, 1) * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dup, (size_t)n * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dw2c, std::max<size_t>(n_frames, 1) * 96);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dc2w, std::max<size_t>(n_frames, 1) * 96);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dcam, std::max<size_t>(M, 1) * 24);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dworld, (size_t)n * 24);
  if (e == hipSuccess) e = hipMemcpyAsync(doff, offsets, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && M) e = hipMemcpyAsync(dfo, frame_of, (size_t)M * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dup, updates, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && n_frames) e = hipMemcpyAsync(dw2c, w2c, (size_t)n_frames * 96, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && n_frames) e = hipMemcpyAsync(dc2w, c2w, (size_t)n_frames * 96, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && M) e = hipMemcpyAsync(dcam, cam, (size_t)M * 24, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dworld, world, (size_t)n * 24, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_landmark_update, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, doff, dfo, dw2c, dc2w, dcam, dworld, dup,
                       c->cfg.c.landmark_maximum_number_of_iterations, c->cfg.c.landmark_maximum_error_squared_meters);
    e = hipMemcpyAsync(world, dworld, (size_t)n * 24, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(updates, dup, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}

// ---- OrbDetector components ---------------------------------------------------------------------------------------
VS_API int vslam_resize_linear_u8(vslam_ctx* c, const uint8_t* src, int32_t rows, int32_t cols, int32_t row_stride, uint8_t* dst, int32_t drows,
                                  int32_t dcols) {
  tmp_reset(c);
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!src || !dst || rows < 2 || cols < 2 || drows < 1 || dcols < 1 || row_stride < cols) return fail(c, VSLAM_ERR_INVALID, "resize: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  uint8_t *ds = nullptr, *dd = nullptr;
  hipError_t e = tmp_get(c, (void**)&ds, (size_t)rows * row_stride);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dd, (size_t)drows * dcols);
  if (e == hipSuccess) e = hipMemcpyAsync(ds, src, (size_t)(rows - 1) * row_stride + cols, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_resize_linear_u8, dim3((dcols + 255) / 256, drows), dim3(256), 0, c->stream, ds, rows, cols, row_stride, dd, drows, dcols, dcols);
    e = hipMemcpyAsync(dst, dd, (size_t)drows * dcols, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}
static OrbUmax orb_umax_table(int half) {   // orb.cpp computeKeyPoints: row half-widths of the circular patch
  OrbUmax t;
  std::memset(&t, 0, sizeof t);
  const int vmax = (int)std::floor(half * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
  for (int v = 0; v <= vmax; ++v) t.v[v] = (int)std::lrint(std::sqrt((double)half * half - v * v));
  for (int v = half, v0 = 0; v >= vmin; --v) { while (t.v[v0] == t.v[v0 + 1]) ++v0; t.v[v] = v0; ++v0; }
  return t;
}
VS_API int vslam_harris_angle(vslam_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t row_stride, int32_t n, const int16_t* xy,
                              float* response, float* angle) {
  tmp_reset(c);
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!img || n < 0 || rows < 33 || cols < 33 || row_stride < cols || (n && (!xy || !response || !angle))) return fail(c, VSLAM_ERR_INVALID, "harris_angle: bad argument");
  for (int i = 0; i < n; ++i)
    if (xy[2 * i] < 16 || xy[2 * i + 1] < 16 || xy[2 * i] >= cols - 16 || xy[2 * i + 1] >= rows - 16) return fail(c, VSLAM_ERR_INVALID, "harris_angle: keypoint closer than 16 px to the border");
  if (n == 0) return VSLAM_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  uint8_t* di = nullptr; int16_t* dxy = nullptr; float *dr = nullptr, *da = nullptr; int32_t* dn = nullptr;
  hipError_t e = tmp_get(c, (void**)&di, (size_t)rows * row_stride);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dxy, (size_t)n * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dr, (size_t)n * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&da, (size_t)n * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dn, 4);
  if (e == hipSuccess) e = hipMemcpyAsync(di, img, (size_t)(rows - 1) * row_stride + cols, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dxy, xy, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dn, &n, 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    const int blocks = std::min(256, (n + 3) / 4);
    hipLaunchKernelGGL(k_orb_harris, dim3(blocks), dim3(256), 0, c->stream, di, row_stride, dn, dxy, dr);
    hipLaunchKernelGGL(k_orb_angle, dim3(blocks), dim3(256), 0, c->stream, di, row_stride, dn, dxy, dr, 15, orb_umax_table(15), da, (float*)nullptr,
                       (const int32_t*)nullptr, 0, 1.f, 0, 31);
    e = hipMemcpyAsync(response, dr, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(angle, da, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}
VS_API int vslam_orb_detect(vslam_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t row_stride, int32_t nfeatures, float scale_factor,
                            int32_t nlevels, int32_t edge, int32_t patch, int32_t fast_threshold, int32_t cap, int32_t* n, float* keypoints) {
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!img || !n || nlevels < 1 || nlevels > 16 || nfeatures < 0 || patch < 3 || patch > 63 || cap < 0 || (cap && !keypoints) || row_stride < cols ||
      !(scale_factor > 1.f) || edge < patch / 2 + 1 || edge < 4 || rows < 2 * edge + 8 || cols < 2 * edge + 8 || rows > 32767 || cols > 32767)
    return fail(c, VSLAM_ERR_INVALID, "orb_detect: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  // features per level (orb.cpp computeKeyPoints), float arithmetic as upstream
  std::vector<int> per(nlevels);
  {
    const float factor = (float)(1.0 / scale_factor);
    float nd = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) { per[l] = (int)std::lrint(nd); sum += per[l]; nd *= factor; }
    per[nlevels - 1] = std::max(nfeatures - sum, 0);
  }
  const int half = patch / 2;
  const OrbUmax um = orb_umax_table(half);
  hipStream_t st = c->stream;
  std::vector<void*> tmp;
  auto dmal = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, std::max<size_t>(bytes, 4)) != hipSuccess) return nullptr; tmp.push_back(q); return q; };
  float* dout = (float*)dmal((size_t)std::max(cap, 1) * 6 * sizeof(float));
  int32_t* dtotal = (int32_t*)dmal(4);
  int rc = VSLAM_OK;
  hipError_t e = (dout && dtotal) ? hipMemsetAsync(dtotal, 0, 4, st) : hipErrorOutOfMemory;
  const uint8_t* lev = nullptr;
  int lrows = rows, lcols = cols, lstride = (cols + 63) & ~63;
  std::vector<vslam_ctx*> scratch;
  if (e == hipSuccess) {
    uint8_t* d0 = (uint8_t*)dmal((size_t)rows * lstride);
    if (!d0) e = hipErrorOutOfMemory;
    else e = hipMemcpy2DAsync(d0, lstride, img, row_stride, cols, rows, hipMemcpyHostToDevice, st);
    lev = d0;
  }
  for (int l = 0; l < nlevels && e == hipSuccess && rc == VSLAM_OK; ++l) {
    const float sc = (float)std::pow((double)scale_factor, (double)l);
    if (l > 0) {
      const int nr = (int)std::lrint(rows / sc), nc = (int)std::lrint(cols / sc);
      if (nr < 2 * edge + 8 || nc < 2 * edge + 8) break;
      const int ns = (nc + 63) & ~63;
      uint8_t* dl = (uint8_t*)dmal((size_t)nr * ns);
      if (!dl) { e = hipErrorOutOfMemory; break; }
      hipLaunchKernelGGL(k_resize_linear_u8, dim3((nc + 255) / 256, nr), dim3(256), 0, st, lev, lrows, lcols, lstride, dl, nr, nc, ns);
      lev = dl; lrows = nr; lcols = nc; lstride = ns;
    }
    // FAST-9/16 + NMS + border filter through the image pipeline's own kernels on a scratch context of the level's size
    vslam_ctx* t = nullptr;
    rc = make_scratch_ctx(c, lrows, lcols, 65535, 64, &t);
    if (rc != VSLAM_OK) break;
    scratch.push_back(t);
    t->cfg.n_regions = 1;
    t->cfg.regions[0].x = 0; t->cfg.regions[0].y = 0; t->cfg.regions[0].w = lcols; t->cfg.regions[0].h = lrows;
    StreamState sst;
    e = hipMemcpy(&sst, t->buf.st, sizeof sst, hipMemcpyDeviceToHost);
    sst.thr[0] = fast_threshold;
    if (e == hipSuccess) e = hipMemcpy(t->buf.st, &sst, sizeof sst, hipMemcpyHostToDevice);
    if (e != hipSuccess) break;
    rc = set_images_device(t, lev, lev, lstride, 0);
    if (rc != VSLAM_OK) break;
    const int N = t->cfg.NMAX;
    int16_t* xy1 = (int16_t*)dmal((size_t)N * 4); int16_t* xy2 = (int16_t*)dmal((size_t)N * 4);
    float* r1 = (float*)dmal((size_t)N * 4); float* r2 = (float*)dmal((size_t)N * 4); float* rh = (float*)dmal((size_t)N * 4);
    int32_t* n1 = (int32_t*)dmal(4); int32_t* n2 = (int32_t*)dmal(4);
    if (!xy1 || !xy2 || !r1 || !r2 || !rh || !n1 || !n2) { e = hipErrorOutOfMemory; break; }
    dim3 g1(t->cfg.TX, (lrows + VS_TILE_H - 1) / VS_TILE_H, 1);
    hipLaunchKernelGGL(k_fast_box, g1, dim3(256), VS_FB_DYN_LDS, st, t->cfg, t->buf);
    hipLaunchKernelGGL(k_emit, dim3(1, 1), dim3(512), 0, st, t->cfg, t->buf, edge, 0);                                  // runByImageBorder(edgeThreshold)
    hipLaunchKernelGGL(k_orb_select<uint8_t>, dim3(1), dim3(1024), 0, st, t->buf.n_kp, t->buf.kp_xy, t->buf.kp_score, 2 * per[l], n1, xy1, r1, N);   // retainBest(2 n) on the FAST score
    hipLaunchKernelGGL(k_orb_harris, dim3(256), dim3(256), 0, st, lev, lstride, n1, xy1, rh);
    hipLaunchKernelGGL(k_orb_select<float>, dim3(1), dim3(1024), 0, st, n1, xy1, rh, per[l], n2, xy2, r2, N);            // retainBest(n) on the Harris response
    hipLaunchKernelGGL(k_orb_angle, dim3(256), dim3(256), 0, st, lev, lstride, n2, xy2, r2, half, um, (float*)nullptr, dout, dtotal, cap, sc, l, patch);
    hipLaunchKernelGGL(k_orb_advance, dim3(1), dim3(1), 0, st, dtotal, n2);
    e = hipGetLastError();
  }
  int32_t total = 0;
  if (e == hipSuccess && rc == VSLAM_OK) e = hipMemcpyAsync(&total, dtotal, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess && rc == VSLAM_OK) {
    *n = total;
    const int m = std::min(total, cap);
    if (m) e = hipMemcpy(keypoints, dout, (size_t)m * 6 * sizeof(float), hipMemcpyDeviceToHost);
    for (vslam_ctx* t : scratch) { int32_t cnt = 0; if (hipMemcpy(&cnt, t->buf.n_kp, 4, hipMemcpyDeviceToHost) == hipSuccess && cnt >= t->cfg.NMAX) rc = fail(c, VSLAM_ERR_CAPACITY, "orb_detect: more than 65535 FAST corners on a level"); }
    if (rc == VSLAM_OK && total > cap) rc = fail(c, VSLAM_ERR_CAPACITY, "orb_detect: output capacity too small");
  }
  for (vslam_ctx* t : scratch) scratch_put(c, t);
  for (void* q : tmp) (void)hipFree(q);
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return rc;
}

VS_API int vslam_get_poses(vslam_ctx* c, int s, int32_t first, int32_t nf, double* out) {
  int rc = check_stream(c, s);
  if (rc) return rc;
  if (!out || first < 0 || nf < 0 || first + nf > VS_POSE_LOG) return fail(c, VSLAM_ERR_INVALID, "bad pose range");
  HIP_TRY(c, d2h(c, out, c->buf.pose_log + ((size_t)s * VS_POSE_LOG + first) * 12, (size_t)nf * 12));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VSLAM_OK;
}
VS_API int vslam_copy_poses_device(vslam_ctx* c, int32_t first, int32_t nf, double* dst) {
  if (!c || !dst || first < 0 || nf < 0 || first + nf > VS_POSE_LOG) return VSLAM_ERR_INVALID;
  sync_all(c);
  HIP_TRY(c, hipMemcpy2DAsync(dst, (size_t)nf * 12 * sizeof(double), c->buf.pose_log + (size_t)first * 12,
                              (size_t)VS_POSE_LOG * 12 * sizeof(double), (size_t)nf * 12 * sizeof(double), c->B,
                              hipMemcpyDeviceToDevice, c->stream));
  return VSLAM_OK;
}
VS_API int vslam_get_timers(vslam_ctx* c, double seconds[8]) {
  if (!c || !seconds) return VSLAM_ERR_INVALID;
  harvest_events(c);
  std::vector<StreamState> st(c->B);
  HIP_TRY(c, hipMemcpy(st.data(), c->buf.st, sizeof(StreamState) * c->B, hipMemcpyDeviceToHost));
  double ph[5] = {0, 0, 0, 0, 0};
  for (int s = 0; s < c->B; ++s) for (int k = 0; k < 5; ++k) ph[k] += (double)st[s].ticks[k] * 1e-8 / c->B;  // 100 MHz ticks
  seconds[0] = (c->kern_ms[0] + c->kern_ms[1]) * 1e-3;  // keypoint_detection: FAST/NMS + emission/controller
  seconds[1] = c->kern_ms[2] * 1e-3;                    // descriptor_extraction
  seconds[2] = ph[4];                                   // point_triangulation (compute())
  seconds[3] = c->kern_ms[3] * 1e-3 + ph[0];            // tracking: candidate search + resolution
  seconds[4] = ph[4];                                   // track_creation (tracker's timer around compute())
  seconds[5] = ph[1];                                   // pose_optimization
  seconds[6] = ph[3] + c->kern_ms[6] * 1e-3;            // landmark_optimization (in-kernel part + wide kernel)
  seconds[7] = ph[2] + c->kern_ms[5] * 1e-3;            // point_recovery
  return VSLAM_OK;
}
VS_API int vslam_get_kernel_times(vslam_ctx* c, double ms[8], int32_t launches[8]) {
  if (!c || !ms || !launches) return VSLAM_ERR_INVALID;
  harvest_events(c);
  for (int k = 0; k < 8; ++k) { ms[k] = c->kern_ms[k]; launches[k] = c->kern_n[k]; }
  return VSLAM_OK;
}
VS_API int vslam_enable_timers(vslam_ctx* c, int on) {
  if (!c) return VSLAM_ERR_INVALID;
  harvest_events(c);
  if (on && !c->timers) { for (int k = 0; k < 8; ++k) { c->kern_ms[k] = 0; c->kern_n[k] = 0; } }
  c->timers = on != 0;
  return VSLAM_OK;
}

// ---- stand-alone kernels ---------------------------------------------------------------------------
static int make_scratch_ctx(vslam_ctx* parent, int rows, int cols, int nmax, int maxp, vslam_ctx** out) {
  vslam_config cfg = parent->cfg.c;
  cfg.rows = rows; cfg.cols = cols; cfg.det_rows = 1; cfg.det_cols = 1;
  cfg.descriptor_type = VSLAM_DESCRIPTOR_BRIEF;   // the stand-alone FAST / BRIEF entries need the box image whatever the parent uses
  cfg.max_keypoints = std::max(64, nmax); cfg.max_points = std::max(64, maxp); cfg.max_history_frames = 2;
  return scratch_get(parent, cfg, out);
}
VS_API int vslam_fast_detect(vslam_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t rx, int32_t ry,
                             int32_t rw, int32_t rh, int32_t threshold, int32_t cap, int32_t* n, int16_t* xy, int32_t* score) {
  if (!c || !img || !n) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (rx < 0 || ry < 0 || rw < 1 || rh < 1 || rx + rw > cols || ry + rh > rows || cap < 0 || (cap && (!xy))) return fail(c, VSLAM_ERR_INVALID, "ROI outside the image");
  HIP_TRY(c, hipSetDevice(c->device));
  vslam_ctx* t = nullptr;
  int rc = make_scratch_ctx(c, rows, cols, std::min(rows * cols, 65535), 64, &t);   // 16-bit feature indices; independent of `cap`: one pooled scratch context serves every call
  if (rc != VSLAM_OK) return rc;
  t->cfg.n_regions = 1;
  t->cfg.regions[0].x = rx; t->cfg.regions[0].y = ry; t->cfg.regions[0].w = rw; t->cfg.regions[0].h = rh;
  StreamState st;
  hipError_t e = hipMemcpy(&st, t->buf.st, sizeof st, hipMemcpyDeviceToHost);
  st.thr[0] = threshold;
  if (e == hipSuccess) e = hipMemcpy(t->buf.st, &st, sizeof st, hipMemcpyHostToDevice);
  rc = e == hipSuccess ? upload_images(t, img, img, stride, 0) : fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  if (rc == VSLAM_OK) {
    dim3 g1(t->cfg.TX, (rows + VS_TILE_H - 1) / VS_TILE_H, 2);
    hipLaunchKernelGGL(k_fast_box, g1, dim3(256), VS_FB_DYN_LDS, t->stream_img, t->cfg, t->buf);
    hipLaunchKernelGGL(k_emit, dim3(1, 2), dim3(512), 0, t->stream_img, t->cfg, t->buf, 0, 0);
    int32_t cnt = 0;
    rc = vslam_get_keypoints(t, 0, 0, cap, &cnt, xy, score, nullptr);
    *n = cnt;
    if (rc == VSLAM_OK) {
      // more corners in the ROI than the scratch buffers hold (k_emit clamps and raises error bit 0): not a silent truncation
      ImgInfo ii;
      if (hipMemcpy(&ii, t->sets[t->last_set].iinfo, sizeof ii, hipMemcpyDeviceToHost) == hipSuccess && ii.raw_count[0][0] > cnt) {
        *n = ii.raw_count[0][0];
        rc = fail(c, VSLAM_ERR_CAPACITY, "fast_detect: more corners than the output capacity (65535 at most)");
      }
    }
    if (rc == VSLAM_OK) for (int i = 0; i < cnt; ++i) { xy[2 * i] = (int16_t)(xy[2 * i] - rx); xy[2 * i + 1] = (int16_t)(xy[2 * i + 1] - ry); }
    else if (rc != VSLAM_ERR_CAPACITY || c->err.empty()) c->err = t->err;
  }
  scratch_put(c, t);
  return rc;
}
VS_API int vslam_brief_describe(vslam_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t n,
                                const int16_t* xy, uint8_t* keep, uint8_t* desc) {
  tmp_reset(c);
  if (!c || !img || !xy || !keep || !desc || n < 0) return VSLAM_ERR_INVALID;
  vslam_ctx* t = nullptr;
  int rc = make_scratch_ctx(c, rows, cols, 64, 64, &t);
  if (rc != VSLAM_OK) return rc;
  int16_t* dxy = nullptr; uint8_t* dkeep = nullptr; uint8_t* ddesc = nullptr;
  hipError_t e = tmp_alloc(c, &dxy, (size_t)n * 2);
  if (e == hipSuccess) e = tmp_alloc(c, &dkeep, (size_t)n);
  if (e == hipSuccess) e = tmp_alloc(c, &ddesc, (size_t)n * 32);
  if (e == hipSuccess && n) e = hipMemcpyAsync(dxy, xy, (size_t)n * 2 * sizeof(int16_t), hipMemcpyHostToDevice, t->stream_img);
  rc = e == hipSuccess ? upload_images(t, img, img, stride, 0) : fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  if (rc == VSLAM_OK && n) {
    dim3 g1(t->cfg.TX, (rows + VS_TILE_H - 1) / VS_TILE_H, 2);
    hipLaunchKernelGGL(k_fast_box, g1, dim3(256), VS_FB_DYN_LDS, t->stream_img, t->cfg, t->buf);
    hipLaunchKernelGGL(k_brief_at, dim3(std::min(64, (n + 3) / 4)), dim3(256), 0, t->stream_img, t->buf.box, t->cfg.bstride, rows, cols,
                       n, dxy, dkeep, ddesc);
    e = hipMemcpyAsync(keep, dkeep, (size_t)n, hipMemcpyDeviceToHost, t->stream_img);
    if (e == hipSuccess) e = hipMemcpyAsync(desc, ddesc, (size_t)n * 32, hipMemcpyDeviceToHost, t->stream_img);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream_img);
    if (e != hipSuccess) rc = fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  }
  scratch_put(c, t);
  return rc;
}
// cv::ORB::create()->compute() pieces, stand-alone (known-answer tests)
static int orb_blur_device(vslam_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, uint8_t** dimg, uint8_t** dblur) {
  hipError_t e = tmp_get(c, (void**)dimg, (size_t)rows * stride);     // per-call scratch: the caller has reset the arena
  if (e == hipSuccess) e = tmp_get(c, (void**)dblur, (size_t)rows * cols);
  if (e == hipSuccess) e = hipMemcpyAsync(*dimg, img, (size_t)(rows - 1) * stride + cols, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  Gauss7 gk; for (int i = 0; i < 4; ++i) gk.k[i] = c->cfg.gauss7[i];
  hipLaunchKernelGGL(k_gauss7_plain, dim3((cols + VS_TILE_W - 1) / VS_TILE_W, (rows + VS_TILE_H - 1) / VS_TILE_H), dim3(256), 0, c->stream, *dimg, stride, rows, cols, gk, *dblur, cols);
  return VSLAM_OK;
}
VS_API int vslam_gaussian_blur7_u8(vslam_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, uint8_t* out) {
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!img || !out || rows < 4 || cols < 4 || stride < cols) return fail(c, VSLAM_ERR_INVALID, "gaussian_blur7: bad argument");   // one reflection per border
  HIP_TRY(c, hipSetDevice(c->device));
  tmp_reset(c);
  uint8_t *dimg = nullptr, *dblur = nullptr;
  int rc = orb_blur_device(c, img, rows, cols, stride, &dimg, &dblur);
  hipError_t e = hipSuccess;
  if (rc == VSLAM_OK) e = hipMemcpyAsync(out, dblur, (size_t)rows * cols, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (rc != VSLAM_OK) return rc;
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}
VS_API int vslam_orb_describe(vslam_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t n, const int16_t* xy,
                              float angle_degrees, uint8_t* keep, uint8_t* desc) {
  tmp_reset(c);
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!img || n < 0 || rows < 4 || cols < 4 || stride < cols || (n && (!xy || !keep || !desc))) return fail(c, VSLAM_ERR_INVALID, "orb_describe: bad argument");
  if (n == 0) return VSLAM_OK;
  if (rows < 2 * VSLAM_ORB_BORDER + 1 || cols < 2 * VSLAM_ORB_BORDER + 1) {   // no pixel is 31 px away from every border: all keypoints removed
    std::memset(keep, 0, (size_t)n);
    std::memset(desc, 0, (size_t)n * 32);
    return VSLAM_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  uint8_t *dimg = nullptr, *dblur = nullptr, *dkeep = nullptr, *ddesc = nullptr; int16_t* dxy = nullptr;
  int rc = orb_blur_device(c, img, rows, cols, stride, &dimg, &dblur);
  hipError_t e = hipSuccess;
  if (rc == VSLAM_OK) {
    e = tmp_get(c, (void**)&dxy, (size_t)n * 4);
    if (e == hipSuccess) e = tmp_get(c, (void**)&dkeep, (size_t)n);
    if (e == hipSuccess) e = tmp_get(c, (void**)&ddesc, (size_t)n * 32);
    if (e == hipSuccess) e = hipMemcpyAsync(dxy, xy, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      float a, b;
      orb_rotation_host(angle_degrees, &a, &b);
      hipLaunchKernelGGL(k_orb_at, dim3(std::min(64, (n + 3) / 4)), dim3(256), 0, c->stream, dblur, cols, rows, cols, n, dxy, a, b, dkeep, ddesc);
      e = hipMemcpyAsync(keep, dkeep, (size_t)n, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(desc, ddesc, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  if (rc != VSLAM_OK) return rc;
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}
// cv::ORB::create()->compute() on an OrbDetector's keypoints: a pyramid up to the highest octave present (level l from level l-1, as the detector
// builds it), the 7x7 Gaussian per level, the steered tests per keypoint at its level.  Positions, border filter and rotations are host arithmetic
// (float products rounded half-to-even, cos / sin through the host libm as OpenCV evaluates them).
VS_API int vslam_orb_describe_keypoints(vslam_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t n, const float* kp6,
                                        float scale_factor, uint8_t* keep, uint8_t* desc) {
  tmp_reset(c);
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!img || n < 0 || rows < 4 || cols < 4 || stride < cols || !(scale_factor > 1.f) || (n && (!kp6 || !keep || !desc))) return fail(c, VSLAM_ERR_INVALID, "orb_describe_keypoints: bad argument");
  if (n == 0) return VSLAM_OK;
  int top = 0;
  for (int i = 0; i < n; ++i) { const int o = (int)kp6[6 * (size_t)i + 5]; if (o < 0 || o > 15) return fail(c, VSLAM_ERR_INVALID, "orb_describe_keypoints: octave out of range"); top = std::max(top, o); }
  HIP_TRY(c, hipSetDevice(c->device));
  OrbLevels L;
  std::memset(&L, 0, sizeof L);
  float scale[16];
  uint8_t* raw[16];
  hipError_t e = hipSuccess;
  for (int l = 0; l <= top; ++l) {      // every level is validated BEFORE the first launch: an error return must not leave kernels running on the arena
    scale[l] = (float)std::pow((double)scale_factor, (double)l);
    L.rows[l] = l ? (int)std::lrint(rows / scale[l]) : rows; L.cols[l] = l ? (int)std::lrint(cols / scale[l]) : cols;
    if (L.rows[l] < 8 || L.cols[l] < 8) return fail(c, VSLAM_ERR_INVALID, "orb_describe_keypoints: pyramid level smaller than 8 pixels");
  }
  for (int l = 0; l <= top && e == hipSuccess; ++l) {
    L.stride[l] = L.cols[l];
    uint8_t* blur = nullptr;
    e = tmp_get(c, (void**)&raw[l], l ? (size_t)L.rows[l] * L.cols[l] : (size_t)rows * stride);
    if (e == hipSuccess) e = tmp_get(c, (void**)&blur, (size_t)L.rows[l] * L.cols[l]);
    L.blur[l] = blur;
    if (e != hipSuccess) break;
    const int lstride = l ? L.cols[l] : stride;
    if (l == 0) e = hipMemcpyAsync(raw[0], img, (size_t)(rows - 1) * stride + cols, hipMemcpyHostToDevice, c->stream);
    else hipLaunchKernelGGL(k_resize_linear_u8, dim3((L.cols[l] + 255) / 256, L.rows[l]), dim3(256), 0, c->stream, raw[l - 1], L.rows[l - 1], L.cols[l - 1],
                            l == 1 ? stride : L.cols[l - 1], raw[l], L.rows[l], L.cols[l], L.cols[l]);
    Gauss7 gk; for (int i = 0; i < 4; ++i) gk.k[i] = c->cfg.gauss7[i];
    hipLaunchKernelGGL(k_gauss7_plain, dim3((L.cols[l] + VS_TILE_W - 1) / VS_TILE_W, (L.rows[l] + VS_TILE_H - 1) / VS_TILE_H), dim3(256), 0, c->stream, raw[l], lstride,
                       L.rows[l], L.cols[l], gk, blur, L.cols[l]);
  }
  std::vector<int32_t> pos((size_t)n * 3);
  std::vector<float> ab((size_t)n * 2);
  const int reach = 23;   // the rotated 31 x 31 pattern reaches cvRound(15 sqrt 2) = 21 pixels
  for (int i = 0; i < n; ++i) {
    const float* k = kp6 + 6 * (size_t)i;
    const int lv = (int)k[5];
    const float inv = 1.f / scale[lv];
    const int cx = (int)std::lrint(k[0] * inv), cy = (int)std::lrint(k[1] * inv);
    const int x0 = (int)std::lrint(k[0]), y0 = (int)std::lrint(k[1]);
    const bool in = x0 >= VSLAM_ORB_BORDER && x0 < cols - VSLAM_ORB_BORDER && y0 >= VSLAM_ORB_BORDER && y0 < rows - VSLAM_ORB_BORDER &&   // runByImageBorder(31) at level 0
                    cx >= reach && cy >= reach && cx < L.cols[lv] - reach && cy < L.rows[lv] - reach;
    pos[3 * (size_t)i] = cx; pos[3 * (size_t)i + 1] = cy; pos[3 * (size_t)i + 2] = in ? lv : -1;
    orb_rotation_host(k[3], &ab[2 * (size_t)i], &ab[2 * (size_t)i + 1]);
  }
  int32_t* dpos = nullptr; float* dab = nullptr; uint8_t *dkeep = nullptr, *ddesc = nullptr;
  if (e == hipSuccess) e = tmp_get(c, (void**)&dpos, pos.size() * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dab, ab.size() * 4);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dkeep, (size_t)n);
  if (e == hipSuccess) e = tmp_get(c, (void**)&ddesc, (size_t)n * 32);
  if (e == hipSuccess) e = hipMemcpyAsync(dpos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dab, ab.data(), ab.size() * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_orb_at_levels, dim3(std::min(64, (n + 3) / 4)), dim3(256), 0, c->stream, L, n, dpos, dab, dkeep, ddesc);
    e = hipMemcpyAsync(keep, dkeep, (size_t)n, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(desc, ddesc, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);    // also: the host staging vectors may go out of scope now
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}
// ---- descriptor test pairs as run-time data (the tables are __constant__ arrays of this module: one copy per device) ------
static int pattern_io(int device, int which, const int8_t* in, int8_t* out) {
  if ((!in && !out) || device < 0) { g_create_error = "pattern: bad argument"; return VSLAM_ERR_INVALID; }
  if (in) {
    for (int i = 0; i < 256; ++i) {
      const int8_t* q = in + 4 * i;
      if (which == 0) {
        for (int k = 0; k < 4; ++k) if (q[k] < -VSLAM_BRIEF_PATCH_HALF || q[k] > VSLAM_BRIEF_PATCH_HALF) { g_create_error = "brief pattern: offset beyond the 48 px patch"; return VSLAM_ERR_INVALID; }
      } else {
        // a 31 x 31 patch: |x|, |y| <= 15 (OpenCV's bit_pattern_31_ reaches (12, -13), radius 17.7).  Where the reach matters: the
        // tiled extractor (k_orb_describe) stages a 16 px margin and rotates by the FAST keypoints' fixed -1 degree, so a rotated,
        // rounded offset is at most rint(15 cos 1 + 15 sin 1) = 15; the kernels that rotate by arbitrary angles gather from the
        // whole image behind the 31 px border, and 15 sqrt 2 < 22.
        for (int k = 0; k < 4; ++k) if (q[k] < -15 || q[k] > 15) { g_create_error = "orb pattern: offset beyond the 31 px patch (|x|, |y| <= 15)"; return VSLAM_ERR_INVALID; }
      }
    }
  }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "pattern: no such HIP device"; return VSLAM_ERR_NO_DEVICE; }
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && in) e = which == 0 ? hipMemcpyToSymbol(HIP_SYMBOL(c_brief), in, 1024) : hipMemcpyToSymbol(HIP_SYMBOL(c_orb), in, 1024);
  if (e == hipSuccess && out) e = which == 0 ? hipMemcpyFromSymbol(out, HIP_SYMBOL(c_brief), 1024) : hipMemcpyFromSymbol(out, HIP_SYMBOL(c_orb), 1024);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { g_create_error = std::string("pattern: ") + hipGetErrorString(e); return VSLAM_ERR_HIP; }
  return VSLAM_OK;
}
VS_API int vslam_set_brief_pattern(int device, const int8_t* pairs) { return pattern_io(device, 0, pairs, nullptr); }
VS_API int vslam_set_orb_pattern(int device, const int8_t* pairs) { return pattern_io(device, 1, pairs, nullptr); }
VS_API int vslam_get_brief_pattern(int device, int8_t* out) { return pattern_io(device, 0, nullptr, out); }
VS_API int vslam_get_orb_pattern(int device, int8_t* out) { return pattern_io(device, 1, nullptr, out); }

VS_API int vslam_knn2(vslam_ctx* c, int norm, int32_t nq, const uint8_t* q, int32_t nt, const uint8_t* t, int32_t* idx, float* dist) {
  tmp_reset(c);
  if (!c || !q || !t || !idx || !dist || nq < 0 || nt < 0 || norm < 0 || norm > 3) return VSLAM_ERR_INVALID;
  if (nq == 0) return VSLAM_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  uint8_t *dq = nullptr, *dt = nullptr; int32_t* di = nullptr; float* dd = nullptr;
  hipError_t e = tmp_get(c, (void**)&dq, (size_t)nq * 32);
  if (e == hipSuccess) e = tmp_get(c, (void**)&dt, std::max<size_t>((size_t)nt * 32, 32));
  if (e == hipSuccess) e = tmp_get(c, (void**)&di, (size_t)nq * 2 * sizeof(int32_t));
  if (e == hipSuccess) e = tmp_get(c, (void**)&dd, (size_t)nq * 2 * sizeof(float));
  if (e == hipSuccess) e = hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && nt) e = hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_knn2, dim3((nq + 15) / 16), dim3(256), 0, c->stream, norm, nq, dq, nt, dt, di, dd);
    e = hipMemcpyAsync(idx, di, (size_t)nq * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(dist, dd, (size_t)nq * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}
static int align_points_impl(vslam_ctx* c, bool uvd, int32_t n, const double* moving, const double* fixed4, const double* omega,
                             const double* weight, const double T_init[12], double T_out[12], double* chi, uint8_t* inlier,
                             int32_t* n_inliers, double* total_error, int32_t* iterations, double H_out[36]) {
  tmp_reset(c);
  vslam_ctx* t = nullptr;
  vslam_config cfg = c->cfg.c;
  cfg.max_points = (std::max(64, n) + 1023) / 1024 * 1024; cfg.max_keypoints = 64; cfg.max_history_frames = 2;   // rounded: one pooled scratch context serves every call
  int rc = scratch_get(c, cfg, &t);
  if (rc != VSLAM_OK) return rc;
  double* dT = nullptr;
  hipError_t e = tmp_alloc(c, &dT, 12);
  if (e == hipSuccess) e = hipMemcpyAsync(dT, T_init, 12 * sizeof(double), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(t->buf.al_moving, moving, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(t->buf.al_fixed, fixed4, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(t->buf.al_omega, omega, (size_t)n * sizeof(double), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(t->buf.al_weight, weight, (size_t)n * sizeof(double), hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) {
    if (uvd) hipLaunchKernelGGL(k_align_points<true>, dim3(1), dim3(VS_WG), 0, t->stream, t->cfg, t->buf, n, dT);
    else hipLaunchKernelGGL(k_align_points<false>, dim3(1), dim3(VS_WG), 0, t->stream, t->cfg, t->buf, n, dT);
    StreamState st;
    e = hipMemcpyAsync(&st, t->buf.st, sizeof st, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess && chi && n) e = hipMemcpyAsync(chi, t->buf.al_chi, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess && inlier && n) e = hipMemcpyAsync(inlier, t->buf.al_inl, (size_t)n, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e == hipSuccess) {
      if (T_out) std::memcpy(T_out, st.al_T, sizeof(double) * 12);
      if (H_out) std::memcpy(H_out, st.al_H, sizeof(double) * 36);
      if (n_inliers) *n_inliers = st.al_inliers;
      if (total_error) *total_error = st.al_total_error;
      if (iterations) *iterations = st.al_iterations;
    }
  }
  if (e != hipSuccess) rc = fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  scratch_put(c, t);
  return rc;
}
VS_API int vslam_align_points(vslam_ctx* c, int32_t n, const double* moving, const double* fixed, const double* omega,
                              const double* weight, const double T_init[12], double T_out[12], double* chi, uint8_t* inlier,
                              int32_t* n_inliers, double* total_error, int32_t* iterations, double H_out[36]) {
  if (!c || n < 0 || !moving || !fixed || !omega || !weight || !T_init) return VSLAM_ERR_INVALID;
  return align_points_impl(c, false, n, moving, fixed, omega, weight, T_init, T_out, chi, inlier, n_inliers, total_error, iterations, H_out);
}
VS_API int vslam_align_points_uvd(vslam_ctx* c, int32_t n, const double* moving, const double* fixed_uvd, const double* omega_uv,
                                  const double* omega_depth, const double* weight, const double T_init[12], double T_out[12],
                                  double* chi, uint8_t* inlier, int32_t* n_inliers, double* total_error, int32_t* iterations,
                                  double H_out[36]) {
  if (!c || n < 0 || !moving || !fixed_uvd || !omega_uv || !omega_depth || !weight || !T_init) return VSLAM_ERR_INVALID;
  std::vector<double> f4((size_t)std::max(n, 1) * 4);   // (u, v, depth, depth information) per measurement
  for (int i = 0; i < n; ++i) { f4[4 * (size_t)i] = fixed_uvd[3 * (size_t)i]; f4[4 * (size_t)i + 1] = fixed_uvd[3 * (size_t)i + 1]; f4[4 * (size_t)i + 2] = fixed_uvd[3 * (size_t)i + 2]; f4[4 * (size_t)i + 3] = omega_depth[i]; }
  return align_points_impl(c, true, n, moving, f4.data(), omega_uv, weight, T_init, T_out, chi, inlier, n_inliers, total_error, iterations, H_out);
}

// features of one image of scratch context t as the image pipeline would leave them: row-major order, coordinates,
// descriptors, cleared used flags, the row / 16-px-cell CSR.  order[k] = caller index of sorted feature k.
static int upload_features(vslam_ctx* c, vslam_ctx* t, int side, int n, const int32_t* rcx, const uint8_t* dx, std::vector<int>& ord) {
  const DevCfg& dc = t->cfg;
  const int rows = dc.c.rows, cols = dc.c.cols, CW1 = dc.CW + 1;
  ord.resize(n);
  for (int i = 0; i < n; ++i) {
    ord[i] = i;
    if (rcx[2 * i] < 0 || rcx[2 * i] >= rows || rcx[2 * i + 1] < 0 || rcx[2 * i + 1] >= cols) return fail(c, VSLAM_ERR_INVALID, "feature outside the image");
  }
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return rcx[2 * a] != rcx[2 * b] ? rcx[2 * a] < rcx[2 * b] : rcx[2 * a + 1] < rcx[2 * b + 1]; });
  std::vector<int16_t> xy((size_t)std::max(n, 1) * 2);
  std::vector<uint8_t> ds((size_t)std::max(n, 1) * 32), used((size_t)std::max(n, 1), 0);
  std::vector<int32_t> rowcell((size_t)rows * CW1);
  for (int k = 0; k < n; ++k) {
    xy[2 * k] = (int16_t)rcx[2 * ord[k] + 1]; xy[2 * k + 1] = (int16_t)rcx[2 * ord[k]];
    std::memcpy(&ds[(size_t)32 * k], dx + (size_t)32 * ord[k], 32);
  }
  int k = 0;
  for (int r = 0; r < rows; ++r)
    for (int cc = 0; cc < CW1; ++cc) {
      while (k < n && (xy[2 * k + 1] < r || (xy[2 * k + 1] == r && xy[2 * k] < 16 * cc))) ++k;
      rowcell[(size_t)r * CW1 + cc] = k;
    }
  const size_t N = dc.NMAX;
  hipError_t e = hipSuccess;
  if (n) e = hipMemcpyAsync(t->buf.kp_xy + side * N * 2, xy.data(), (size_t)n * 4, hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(t->buf.desc + side * N * 32, ds.data(), (size_t)n * 32, hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(t->buf.used + side * N, used.data(), (size_t)n, hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(t->buf.rowcell + (size_t)side * rows * CW1, rowcell.data(), rowcell.size() * 4, hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(t->buf.n_kp + side, &n, 4, hipMemcpyHostToDevice, t->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(t->stream);   // the host vectors go out of scope
  if (e != hipSuccess) return fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  return VSLAM_OK;
}

VS_API int vslam_track_match(vslam_ctx* c, const double T[12], int32_t d, double tau_track, double tau_tri, int32_t by_appearance,
                             int32_t nP, const double* cam, const uint8_t* pdL, const uint8_t* pdR, const int32_t* epi,
                             int32_t nL, const int32_t* rcL, const uint8_t* dL, int32_t nR, const int32_t* rcR, const uint8_t* dR,
                             int32_t* n_tracked, int32_t* out4, int32_t* n_lost, int32_t* lost) {
  if (!c || !T || nP < 0 || nL < 0 || nR < 0 || !n_tracked || !out4 || !n_lost || !lost) return VSLAM_ERR_INVALID;
  if ((nP && (!cam || !pdL || !pdR || !epi)) || (nL && (!rcL || !dL)) || (nR && (!rcR || !dR))) return VSLAM_ERR_INVALID;
  vslam_ctx* t = nullptr;
  vslam_config cfg = c->cfg.c;
  cfg.max_points = std::max(64, nP); cfg.max_keypoints = std::max(64, std::max(nL, nR)); cfg.max_history_frames = 2;
  int rc = scratch_get(c, cfg, &t);
  if (rc != VSLAM_OK) return rc;
  std::vector<int> order[2];
  rc = upload_features(c, t, 0, nL, rcL, dL, order[0]);
  if (rc == VSLAM_OK) rc = upload_features(c, t, 1, nR, rcR, dR, order[1]);
  hipError_t e = hipSuccess;
  if (rc == VSLAM_OK) {
    // previous points in point buffer 0
    std::vector<uint8_t> pdesc((size_t)std::max(nP, 1) * 64);
    std::vector<int32_t> meta((size_t)std::max(nP, 1) * META, 0);
    for (int i = 0; i < nP; ++i) {
      std::memcpy(&pdesc[(size_t)64 * i], pdL + (size_t)32 * i, 32); std::memcpy(&pdesc[(size_t)64 * i + 32], pdR + (size_t)32 * i, 32);
      meta[(size_t)i * META + M_EPI] = epi[i]; meta[(size_t)i * META + M_PREV] = -1;
    }
    if (nP) e = hipMemcpyAsync(t->buf.p_cam, cam, (size_t)nP * 3 * sizeof(double), hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess && nP) e = hipMemcpyAsync(t->buf.p_desc, pdesc.data(), (size_t)nP * 64, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess && nP) e = hipMemcpyAsync(t->buf.p_meta, meta.data(), (size_t)nP * META * 4, hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t->buf.n_points, &nP, 4, hipMemcpyHostToDevice, t->stream);
    StreamState st;
    if (e == hipSuccess) e = hipMemcpyAsync(&st, t->buf.st, sizeof st, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e == hipSuccess) {
      st.has_prev = 1; st.cur = 0; st.win = d; st.tau_track = tau_track; st.tau_tri = tau_tri;
      st.status = by_appearance ? VSLAM_LOCALIZING : VSLAM_TRACKING;
      std::memcpy(st.prior, T, sizeof(double) * 12);
      e = hipMemcpyAsync(t->buf.st, &st, sizeof st, hipMemcpyHostToDevice, t->stream);
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_track_candidates, dim3(16, 1), dim3(256), 0, t->stream, t->cfg, t->buf, by_appearance ? 1 : 0);
      hipLaunchKernelGGL(k_stage, dim3(1), dim3(VS_WG), 0, t->stream, t->cfg, t->buf, (int)VS_STAGE_TRACK, by_appearance ? 1 : 0, StageIo{});
      e = hipMemcpyAsync(&st, t->buf.st, sizeof st, hipMemcpyDeviceToHost, t->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    }
    if (e == hipSuccess) {
      std::vector<int32_t> trk((size_t)std::max(st.n_trk, 1) * 4), ls((size_t)std::max(st.n_lost, 1));
      if (st.n_trk) e = hipMemcpy(trk.data(), t->buf.trk, (size_t)st.n_trk * 16, hipMemcpyDeviceToHost);
      if (e == hipSuccess && st.n_lost) e = hipMemcpy(ls.data(), t->buf.lost, (size_t)st.n_lost * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess) {
        *n_tracked = st.n_trk; *n_lost = st.n_lost;
        for (int u = 0; u < st.n_trk; ++u) {
          out4[4 * u] = trk[4 * u]; out4[4 * u + 1] = order[0][trk[4 * u + 1]]; out4[4 * u + 2] = order[1][trk[4 * u + 2]]; out4[4 * u + 3] = trk[4 * u + 3];
        }
        for (int u = 0; u < st.n_lost; ++u) lost[u] = ls[u];
      }
    }
    if (e != hipSuccess) rc = fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  }
  scratch_put(c, t);
  return rc;
}

VS_API int vslam_stereo_match(vslam_ctx* c, double tau_tri, int32_t nL, const int32_t* rcL, const uint8_t* dL, int32_t nR,
                              const int32_t* rcR, const uint8_t* dR, int32_t cap, int32_t* n_out, int32_t* out4) {
  if (!c || nL < 0 || nR < 0 || cap < 0 || !n_out || (cap && !out4) || (nL && (!rcL || !dL)) || (nR && (!rcR || !dR))) return VSLAM_ERR_INVALID;
  vslam_ctx* t = nullptr;
  vslam_config cfg = c->cfg.c;
  cfg.max_keypoints = std::max(64, std::max(nL, nR)); cfg.max_points = std::max(64, nL); cfg.max_history_frames = 2;
  int rc = scratch_get(c, cfg, &t);
  if (rc != VSLAM_OK) return rc;
  std::vector<int> order[2];
  rc = upload_features(c, t, 0, nL, rcL, dL, order[0]);
  if (rc == VSLAM_OK) rc = upload_features(c, t, 1, nR, rcR, dR, order[1]);
  if (rc == VSLAM_OK) {
    StreamState st;
    hipError_t e = hipMemcpyAsync(&st, t->buf.st, sizeof st, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e == hipSuccess) {
      st.tau_tri = tau_tri; st.n_cur = 0; st.cur = 0;
      e = hipMemcpyAsync(t->buf.st, &st, sizeof st, hipMemcpyHostToDevice, t->stream);
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_stereo_dist, dim3((t->cfg.NMAX + 255) / 256, 1), dim3(256), 0, t->stream, t->cfg, t->buf);
      hipLaunchKernelGGL(k_stage, dim3(1), dim3(VS_WG), 0, t->stream, t->cfg, t->buf, (int)VS_STAGE_STEREO, 0, StageIo{});
      e = hipMemcpyAsync(&st, t->buf.st, sizeof st, hipMemcpyDeviceToHost, t->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    }
    if (e == hipSuccess) {
      const int n = st.n_new;
      *n_out = n;
      if (n > cap) rc = fail(c, VSLAM_ERR_CAPACITY, "stereo_match: output capacity too small");
      else if (n) {
        // the new points were written to point buffer 1 (current = previous ^ 1)
        const size_t P = t->cfg.MAXP;
        std::vector<int16_t> kp((size_t)n * 4);
        std::vector<int32_t> meta((size_t)n * META);
        e = hipMemcpy(kp.data(), t->buf.p_kp + P * 4, (size_t)n * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(meta.data(), t->buf.p_meta + P * META, (size_t)n * META * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n && e == hipSuccess; ++i) {
          int il = -1, ir = -1;   // ids by coordinates (one feature per pixel)
          for (int k = 0; k < nL; ++k) if (rcL[2 * k] == kp[4 * i + 1] && rcL[2 * k + 1] == kp[4 * i]) il = k;
          for (int k = 0; k < nR; ++k) if (rcR[2 * k] == kp[4 * i + 3] && rcR[2 * k + 1] == kp[4 * i + 2]) ir = k;
          out4[4 * i] = il; out4[4 * i + 1] = ir; out4[4 * i + 2] = meta[(size_t)i * META + M_DIST]; out4[4 * i + 3] = meta[(size_t)i * META + M_EPI];
        }
      }
    }
    if (e != hipSuccess) rc = fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  }
  scratch_put(c, t);
  return rc;
}

VS_API int vslam_stereo_recover(vslam_ctx* c, const uint8_t* imgL, const uint8_t* imgR, int32_t row_stride, const double w2c[12], int32_t n,
                                const uint8_t* has_lm, const double* lm, const uint8_t* pdL, const uint8_t* pdR, double tau_track, double tau_tri,
                                int32_t* n_rec, int32_t* rec_index, int32_t* rec_xy4, int32_t* rec_dist, uint8_t* rec_desc, double* rec_xyz) {
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!imgL || !imgR || !w2c || n < 0 || !n_rec || (n && (!has_lm || !lm || !pdL || !pdR || !rec_index || !rec_xy4 || !rec_dist || !rec_desc || !rec_xyz)))
    return fail(c, VSLAM_ERR_INVALID, "stereo_recover: bad argument");
  if (row_stride < c->cfg.c.cols) return fail(c, VSLAM_ERR_INVALID, "row stride smaller than image width");
  *n_rec = 0;
  if (n == 0) return VSLAM_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  vslam_ctx* t = nullptr;
  vslam_config cfg = c->cfg.c;
  cfg.det_rows = 1; cfg.det_cols = 1; cfg.max_keypoints = 64; cfg.max_points = (std::max(64, n) + 1023) & ~1023; cfg.max_history_frames = 2;
  int rc = scratch_get(c, cfg, &t);
  if (rc != VSLAM_OK) return rc;
  const size_t P = t->cfg.MAXP;
  std::vector<uint8_t> desc((size_t)n * 64);
  std::vector<int32_t> meta((size_t)n * META, 0), lost((size_t)n);
  for (int i = 0; i < n; ++i) {
    std::memcpy(&desc[(size_t)64 * i], pdL + (size_t)32 * i, 32);
    std::memcpy(&desc[(size_t)64 * i + 32], pdR + (size_t)32 * i, 32);
    meta[(size_t)i * META + M_LMUP] = has_lm[i] ? 1 : 0;
    meta[(size_t)i * META + M_PREV] = -1;
    lost[i] = i;
  }
  hipStream_t q = t->stream_img;
  hipError_t e = hipMemcpyAsync(t->buf.p_desc, desc.data(), desc.size(), hipMemcpyHostToDevice, q);
  if (e == hipSuccess) e = hipMemcpyAsync(t->buf.p_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, q);
  if (e == hipSuccess) e = hipMemcpyAsync(t->buf.p_lm, lm, (size_t)n * 24, hipMemcpyHostToDevice, q);
  if (e == hipSuccess) e = hipMemcpyAsync(t->buf.lost, lost.data(), (size_t)n * 4, hipMemcpyHostToDevice, q);
  rc = e == hipSuccess ? upload_images(t, imgL, imgR, row_stride, 0) : fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  if (rc == VSLAM_OK) {
    const int rows = t->cfg.c.rows;
    if (t->cfg.c.descriptor_type == VSLAM_DESCRIPTOR_ORB) {
      Gauss7 gk; for (int i = 0; i < 4; ++i) gk.k[i] = t->cfg.gauss7[i];
      hipLaunchKernelGGL(k_gauss7, dim3(t->cfg.TX, (rows + VS_TILE_H - 1) / VS_TILE_H, 2), dim3(256), 0, q, t->cfg, t->buf, gk);
    } else {
      hipLaunchKernelGGL(k_fast_box, dim3(t->cfg.TX, (rows + VS_TILE_H - 1) / VS_TILE_H, 2), dim3(256), VS_FB_DYN_LDS, q, t->cfg, t->buf);
    }
    RecoverAlone a;
    std::memcpy(a.w2c, w2c, sizeof a.w2c); a.tau_track = tau_track; a.tau_tri = tau_tri; a.n = n;
    hipLaunchKernelGGL(k_recover_alone, dim3(1), dim3(VS_WG), 0, q, t->cfg, t->buf, a);
    e = hipGetLastError();
    StreamState st;
    if (e == hipSuccess) e = hipMemcpyAsync(&st, t->buf.st, sizeof st, hipMemcpyDeviceToHost, q);
    if (e == hipSuccess) e = hipStreamSynchronize(q);
    if (e == hipSuccess && st.n_cur > 0) {
      const int k = st.n_cur;
      std::vector<int16_t> kp((size_t)k * 4);
      std::vector<int32_t> m((size_t)k * META);
      e = hipMemcpy(kp.data(), t->buf.p_kp + P * 4, (size_t)k * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(m.data(), t->buf.p_meta + P * META, (size_t)k * META * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(rec_desc, t->buf.p_desc + P * 64, (size_t)k * 64, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(rec_xyz, t->buf.p_cam + P * 3, (size_t)k * 24, hipMemcpyDeviceToHost);
      for (int i = 0; i < k && e == hipSuccess; ++i) {
        rec_index[i] = m[(size_t)i * META + M_PREV]; rec_dist[i] = m[(size_t)i * META + M_DIST];
        for (int j = 0; j < 4; ++j) rec_xy4[4 * i + j] = kp[4 * (size_t)i + j];
      }
      if (e == hipSuccess) *n_rec = k;
    }
    if (e != hipSuccess) rc = fail(c, VSLAM_ERR_HIP, hipGetErrorString(e));
  } else if (c->err.empty()) c->err = t->err;
  scratch_put(c, t);
  return rc;
}

// ---- stage entry points (the reference's plug-in virtuals; control flow stays with the caller) ----------
static StageIo stage_io(vslam_ctx* c, int report, int in_progress) {
  StageIo io;
  std::memset(&io, 0, sizeof io);
  if (c->B == 1 && c->pend.flags) {
    io.set_flags = c->pend.flags; io.status = c->pend.status; io.win = c->pend.win; io.tau = c->pend.tau;
    std::memcpy(io.prior, c->pend.prior, sizeof io.prior); std::memcpy(io.pose, c->pend.pose, sizeof io.pose);
    c->pend.flags = 0;
  }
  c->report_have = 0;
  if (report && c->report) {
    io.report = report; io.report_in_progress = in_progress; io.report_stream = 0; io.seq = ++c->report_seq; io.L = c->rl; io.out = c->report_dev;
    c->report_have = report; c->report_have_ip = in_progress; c->report_have_stream = 0; c->report_have_seq = io.seq;
  }
  return io;
}
// setters that were not folded into a stage launch (the next launch is not a stage kernel, or a getter reads the state)
static int flush_pending(vslam_ctx* c) {
  if (!c->pend.flags) return VSLAM_OK;
  const int fl = c->pend.flags;
  c->pend.flags = 0;
  hipStream_t q = c->groups[0].st_frm;
  if (fl & 1) { D12 p; std::memcpy(p.v, c->pend.prior, sizeof p.v); hipLaunchKernelGGL(k_set_tracker_state, dim3(1), dim3(1), 0, q, c->buf, 0, c->pend.status, c->pend.win, c->pend.tau, p); }
  if (fl & 2) { D12 p; std::memcpy(p.v, c->pend.pose, sizeof p.v); hipLaunchKernelGGL(k_set_pose, dim3(1), dim3(1), 0, q, c->buf, 0, p); }
  HIP_TRY(c, hipGetLastError());
  return VSLAM_OK;
}
static int launch_begin(vslam_ctx* c) {
  const StageIo io = stage_io(c, 0, 0);
  for (auto& g : c->groups) hipLaunchKernelGGL(k_begin, dim3(g.n), dim3(256), 0, g.st_frm, c->cfg, buf_set(c, c->last_set, g.s0, g.q0_frm), io);
  HIP_TRY(c, hipGetLastError());
  return VSLAM_OK;
}
static int launch_stage(vslam_ctx* c, int stage, int arg, int report = 0, int in_progress = 0) {
  const StageIo io = stage_io(c, report, in_progress);
  for (auto& g : c->groups) hipLaunchKernelGGL(k_stage, dim3(g.n), dim3(VS_WG), 0, g.st_frm, c->cfg, buf_set(c, c->last_set, g.s0, g.q0_frm), stage, arg, io);
  HIP_TRY(c, hipGetLastError());
  return VSLAM_OK;
}
VS_API int vslam_frame_begin(vslam_ctx* c, const uint8_t* L, const uint8_t* R, int32_t row_stride, size_t image_stride, int on_device) {
  if (!c) return VSLAM_ERR_INVALID;
  if (c->sticky != VSLAM_OK) return c->sticky;
  HIP_TRY(c, hipSetDevice(c->device));
  c->img_override = (c->B == 1 && c->groups.size() == 1) ? c->groups[0].st_frm : nullptr;
  c->report_xy_seq = -1;
  c->lm_published = false;
  int rc = on_device ? set_images_device(c, L, R, row_stride, image_stride) : upload_images(c, L, R, row_stride, image_stride);
  if (rc == VSLAM_OK) rc = launch_image_pipeline(c);
  c->img_override = nullptr;
  if (rc != VSLAM_OK) return rc;
  rc = launch_begin(c);
  c->frame_begun = rc == VSLAM_OK;
  if (rc == VSLAM_OK && c->report) {
    // a caller that reads stage views wants the keypoints next (initialize() fills Frame::keypoints / descriptors): packed right
    // behind k_begin, no host round trip in between
    const vslam_ctx::Group& g = c->groups[0];
    const int seq = ++c->report_seq;
    hipLaunchKernelGGL(k_report, dim3(32), dim3(256), 0, g.st_frm, c->cfg, buf_set(c, c->last_set, g.s0), 0, (int)VS_REPORT_KEYPOINTS, 0, seq, c->rl, c->report_dev, c->report_done);
    HIP_TRY(c, hipGetLastError());
    c->report_have = VS_REPORT_KEYPOINTS; c->report_have_ip = 0; c->report_have_stream = 0; c->report_have_seq = seq;
  }
  return rc;
}
VS_API int vslam_frame_finish(vslam_ctx* c) {
  if (!c) return VSLAM_ERR_INVALID;
  if (!c->frame_begun) return fail(c, VSLAM_ERR_STATE, "vslam_frame_finish called before vslam_frame_begin");
  c->frame_begun = false;
  int rc = flush_pending(c);
  return rc == VSLAM_OK ? launch_frame(c) : rc;
}
#define NEED_FRAME(name) if (!c) return VSLAM_ERR_INVALID; if (!c->frame_begun) return fail(c, VSLAM_ERR_STATE, name " called before vslam_frame_begin")
VS_API int vslam_frame_restore(vslam_ctx* c) {
  // initialize(frame, false) only rebuilds the two feature stores; the device stores are rebuilt from the
  // keypoint arrays by every vslam_track call (kill / used flags are recomputed), so nothing to launch.
  NEED_FRAME("vslam_frame_restore");
  return VSLAM_OK;
}
VS_API int vslam_track(vslam_ctx* c, int by_appearance) {
  NEED_FRAME("vslam_track");
  { int rc = flush_pending(c); if (rc) return rc; }     // the candidate kernel reads prior / window / distance before the stage kernel runs
  for (auto& g : c->groups) {
    const int gx = cand_blocks(c, g.n);
    hipLaunchKernelGGL(k_track_candidates, dim3(gx, g.n), dim3(256), 0, g.st_frm, c->cfg, buf_set(c, c->last_set, g.s0, g.q0_frm), by_appearance ? 1 : 0);
  }
  return launch_stage(c, VS_STAGE_TRACK, by_appearance ? 1 : 0, VS_REPORT_TRACK);
}
VS_API int vslam_align(vslam_ctx* c, int inverse_depth) { NEED_FRAME("vslam_align"); return launch_stage(c, VS_STAGE_ALIGN, inverse_depth, VS_REPORT_ALIGNER); }
VS_API int vslam_prune_recover(vslam_ctx* c) {
  NEED_FRAME("vslam_prune_recover");
  if (!c->cfg.c.enable_landmark_recovery) return launch_stage(c, VS_STAGE_PRUNE_RECOVER, 0, VS_REPORT_POINTS, 1);
  // with recovery: prune + projection | descriptors of the projected points, wide | append + report — the per-point patch reads of the
  // descriptors go through every CU's memory pipe instead of one (59 -> ~25 us for one stream)
  int rc = launch_stage(c, VS_STAGE_PRUNE_PROJECT, 1);
  if (rc != VSLAM_OK) return rc;
  for (auto& g : c->groups)
    hipLaunchKernelGGL(k_recover_brief, dim3(std::max(4, std::min(64, 1024 / std::max(g.n, 1))), g.n), dim3(256), 0, g.st_frm, c->cfg, buf_set(c, c->last_set, g.s0, g.q0_frm));
  HIP_TRY(c, hipGetLastError());
  // one stream: the stage also publishes the frame's history, so that vslam_compute can run the landmark refinement beside the stereo stage
  // instead of in front of it
  const bool side = c->B == 1;
  rc = launch_stage(c, VS_STAGE_RECOVER_APPEND, side ? 3 : 1, VS_REPORT_POINTS, 1);
  c->lm_published = rc == VSLAM_OK && side;
  return rc;
}
VS_API int vslam_update_points(vslam_ctx* c) { NEED_FRAME("vslam_update_points"); c->lm_published = false; return launch_stage(c, VS_STAGE_UPDATE, 0); }
VS_API int vslam_stereo_new(vslam_ctx* c) {
  NEED_FRAME("vslam_stereo_new");
  c->frame_begun = false;  // compute() is the last call PoseTracker3D::compute makes on a frame
  int rc = launch_stage(c, VS_STAGE_STEREO, 0, VS_REPORT_POINTS, 0);
  return rc == VSLAM_OK ? frame_done(c) : rc;
}
VS_API int vslam_compute(vslam_ctx* c) {     // vslam_update_points + vslam_stereo_new in one launch
  NEED_FRAME("vslam_compute");
  c->frame_begun = false;
  if (c->lm_published) {
    // one stream, its history already published by vslam_prune_recover: the landmark refinement (lm_teams_body, the frame workgroup's refinement
    // spread over several workgroups) runs BESIDE the stereo stage in the same launch (k_stage_lm); the stage only counts the active landmarks.
    // The report — it carries the landmark update counts — is packed by the next launch on the queue.
    c->lm_published = false;
    vslam_ctx::Group& g = c->groups[0];
    const StageIo io = stage_io(c, 0, 0);
    { hipLaunchKernelGGL(k_stage_lm, dim3(g.n * (1 + 16)), dim3(VS_WG), 0, g.st_frm, c->cfg, buf_set(c, c->last_set, g.s0, g.q0_frm), (int)VS_STAGE_STEREO_COUNT, 0, io, g.n, 16); }
    HIP_TRY(c, hipGetLastError());
    if (c->report) {
      const int seq = ++c->report_seq;
      hipLaunchKernelGGL(k_report, dim3(16), dim3(256), 0, g.st_frm, c->cfg, buf_set(c, c->last_set, g.s0), 0, (int)VS_REPORT_POINTS, 0, seq, c->rl, c->report_dev, c->report_done);
      HIP_TRY(c, hipGetLastError());
      c->report_have = VS_REPORT_POINTS; c->report_have_ip = 0; c->report_have_stream = 0; c->report_have_seq = seq;
    }
    return frame_done(c);
  }
  int rc = launch_stage(c, VS_STAGE_COMPUTE, 0, VS_REPORT_POINTS, 0);
  return rc == VSLAM_OK ? frame_done(c) : rc;
}
// the setters are queued on the stream's frame queue, in order with the stage launches around them: no synchronisation
static int check_stream_index(vslam_ctx* c, int s) {
  if (!c) return VSLAM_ERR_INVALID;
  if (s < 0 || s >= c->B) return fail(c, VSLAM_ERR_INVALID, "stream index out of range");
  return VSLAM_OK;
}
VS_API int vslam_set_tracker_state(vslam_ctx* c, int s, int status, const double prior[12], int win, double tau) {
  int rc = check_stream_index(c, s);
  if (rc) return rc;
  if (!prior) return fail(c, VSLAM_ERR_INVALID, "null prior");
  if (c->B == 1) {      // rides with the next stage launch (StageIo)
    c->pend.flags |= 1; c->pend.status = status; c->pend.win = win; c->pend.tau = tau; std::memcpy(c->pend.prior, prior, sizeof c->pend.prior);
    return VSLAM_OK;
  }
  D12 p;
  std::memcpy(p.v, prior, sizeof p.v);
  hipLaunchKernelGGL(k_set_tracker_state, dim3(1), dim3(1), 0, c->groups[group_of(c, s)].st_frm, c->buf, s, status, win, tau, p);
  HIP_TRY(c, hipGetLastError());
  return VSLAM_OK;
}
VS_API int vslam_set_pose(vslam_ctx* c, int s, const double pose[12]) {
  int rc = check_stream_index(c, s);
  if (rc) return rc;
  if (!pose) return fail(c, VSLAM_ERR_INVALID, "null pose");
  if (c->B == 1) { c->pend.flags |= 2; std::memcpy(c->pend.pose, pose, sizeof c->pend.pose); return VSLAM_OK; }
  D12 p;
  std::memcpy(p.v, pose, sizeof p.v);
  hipLaunchKernelGGL(k_set_pose, dim3(1), dim3(1), 0, c->groups[group_of(c, s)].st_frm, c->buf, s, p);
  HIP_TRY(c, hipGetLastError());
  return VSLAM_OK;
}


// ---- pinned host memory for the caller's images -------------------------------------------------------------------------------------
VS_API int vslam_host_alloc(void** out, size_t bytes) {
  if (!out || !bytes) return VSLAM_ERR_INVALID;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, VSLAM_ERR_HIP, "vslam_host_alloc: hipHostMalloc failed"); }
  *out = p;
  return VSLAM_OK;
}
VS_API void vslam_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ---- stage views: one report kernel + one synchronisation of the stream's frame queue per stage (kernels_report.h) -------------
static uint32_t rl_take(uint32_t* off, size_t bytes) { const uint32_t o = *off; *off = (uint32_t)((o + bytes + 63) & ~(size_t)63); return o; }
static int report_ready(vslam_ctx* c) {
  if (c->report) return VSLAM_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  ReportLayout& L = c->rl;
  uint32_t off = (uint32_t)((sizeof(ReportHeader) + 255) & ~(size_t)255);
  const size_t N = c->cfg.NMAX, P = c->cfg.MAXP;
  for (int d = 0; d < 2; ++d) { L.kp_xy[d] = rl_take(&off, N * 4); L.kp_score[d] = rl_take(&off, N); L.desc[d] = rl_take(&off, N * 32); }
  L.trk = rl_take(&off, P * 16); L.lost = rl_take(&off, P * 4);
  L.chi = rl_take(&off, P * 8); L.inl = rl_take(&off, P);
  L.p_kp = rl_take(&off, P * 8); L.p_meta = rl_take(&off, P * 24); L.p_cam = rl_take(&off, P * 24); L.p_desc = rl_take(&off, P * 64);
  L.total = off;
  void* h = nullptr;
  // coherent (fine-grained) on purpose: the GPU's stores go out over PCIe as they are issued and the completion flag's system-scope
  // release orders them for a host that polls it mid-kernel; with any other flag set and no coherence flag, HIP's default is a
  // NON-coherent mapping whose lines may sit in the GPU's L2 until the kernel ends
  HIP_TRY(c, hipHostMalloc(&h, L.total, hipHostMallocMapped | hipHostMallocCoherent));
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return fail(c, VSLAM_ERR_HIP, "hipHostGetDevicePointer(report buffer)"); }
  std::memset(h, 0, L.total);
  if (dalloc(c, &c->report_done, 1) != hipSuccess || hipMemset(c->report_done, 0, sizeof(unsigned int)) != hipSuccess) { (void)hipHostFree(h); return fail(c, VSLAM_ERR_HIP, "report counter"); }
  c->report = (unsigned char*)h; c->report_dev = (unsigned char*)d;
  return VSLAM_OK;
}
// packs `what` of stream s and waits for it: everything queued on the stream's frame queue before (the image pipeline is ordered
// before it by the frame's event) has finished when this returns
static int report_run(vslam_ctx* c, int s, int what, int in_progress, const ReportHeader** hdr) {
  int rc = check_stream_index(c, s);
  if (rc) return rc;
  if (c->sticky != VSLAM_OK) return c->sticky;
  rc = report_ready(c);
  if (rc) return rc;
  const vslam_ctx::Group& g = c->groups[group_of(c, s)];
  int seq = c->report_have_seq;
  const bool folded = c->report_have == what && c->report_have_ip == in_progress && c->report_have_stream == s && !c->pend.flags;
  if (!folded) {      // the stage was launched before the report buffer existed, or something else ran since: pack it now
    rc = flush_pending(c);
    if (rc) return rc;
    seq = ++c->report_seq;
    const int blocks = what == VS_REPORT_KEYPOINTS ? 32 : (what == VS_REPORT_POINTS ? 16 : 4);
    hipLaunchKernelGGL(k_report, dim3(blocks), dim3(256), 0, g.st_frm, c->cfg, buf_set(c, c->last_set, g.s0), s, what, in_progress, seq, c->rl, c->report_dev, c->report_done);
    HIP_TRY(c, hipGetLastError());
    c->report_have = what; c->report_have_ip = in_progress; c->report_have_stream = s; c->report_have_seq = seq;
  }
  // the report's completion flag (its seq, stored last with system-scope release) is polled in the pinned buffer: the caller's
  // thread sees the stage end a few microseconds after the kernel's last store instead of waiting for the runtime's own
  // completion path (~10-15 us per synchronisation, five per frame).  Bounded: after ~0.1 s without the flag the queue is
  // synchronised the ordinary way (an inactive stream never writes a report: that is the STATE error below)
  const ReportHeader* h = reinterpret_cast<const ReportHeader*>(c->report);
  bool seen = false;
  for (long spin = 0; spin < 4000000L; ++spin) {
    if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
    __builtin_ia32_pause();
  }
  if (!seen) HIP_TRY(c, hipStreamSynchronize(g.st_frm));
  *hdr = h;
  if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) != seq || h->what != what) return fail(c, VSLAM_ERR_STATE, "stage report is stale (the stream is inactive?)");
  if ((*hdr)->info.error_flags) c->err = "device buffer capacity exceeded (error_flags != 0)";
  return VSLAM_OK;
}
VS_API int vslam_view_keypoints(vslam_ctx* c, int s, vslam_keypoints_view* out);
// polls a report flag (bounded), falling back to an ordinary synchronisation of the queue
static int report_wait(vslam_ctx* c, const vslam_ctx::Group& g, const int32_t* flag, int seq) {
  for (long spin = 0; spin < 4000000L; ++spin) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return VSLAM_OK;
    __builtin_ia32_pause();
  }
  HIP_TRY(c, hipStreamSynchronize(g.st_frm));
  return VSLAM_OK;
}
VS_API int vslam_view_keypoints_xy(vslam_ctx* c, int s, vslam_keypoints_view* out) {
  if (!c || !out) return VSLAM_ERR_INVALID;
  int rc = check_stream_index(c, s);
  if (rc) return rc;
  if (c->sticky != VSLAM_OK) return c->sticky;
  if (!c->report || s != 0 || c->report_xy_seq < 0 || !c->frame_begun) return vslam_view_keypoints(c, s, out);   // no early report in flight: the full one
  const ReportHeader* h = reinterpret_cast<const ReportHeader*>(c->report);
  rc = report_wait(c, c->groups[0], &h->seq_xy, c->report_xy_seq);
  if (rc) return rc;
  if (__atomic_load_n(&h->seq_xy, __ATOMIC_ACQUIRE) != c->report_xy_seq) return fail(c, VSLAM_ERR_STATE, "early keypoint report is stale (the stream is inactive?)");
  for (int d = 0; d < 2; ++d) {
    out->n[d] = std::min(h->n_kp[d], c->cfg.NMAX);
    out->xy[d] = reinterpret_cast<const int16_t*>(c->report + c->rl.kp_xy[d]);
    out->score[d] = c->report + c->rl.kp_score[d];
    out->desc[d] = nullptr;                       // not there yet: vslam_view_keypoints
  }
  return VSLAM_OK;
}
VS_API int vslam_view_keypoints(vslam_ctx* c, int s, vslam_keypoints_view* out) {
  if (!c || !out) return VSLAM_ERR_INVALID;
  const ReportHeader* h = nullptr;
  int rc = report_run(c, s, VS_REPORT_KEYPOINTS, 0, &h);
  if (rc) return rc;
  for (int d = 0; d < 2; ++d) {
    out->n[d] = std::min(h->n_kp[d], c->cfg.NMAX);
    out->xy[d] = reinterpret_cast<const int16_t*>(c->report + c->rl.kp_xy[d]);
    out->score[d] = c->report + c->rl.kp_score[d];
    out->desc[d] = c->report + c->rl.desc[d];
  }
  return VSLAM_OK;
}
VS_API int vslam_view_track(vslam_ctx* c, int s, vslam_track_view* out) {
  if (!c || !out) return VSLAM_ERR_INVALID;
  const ReportHeader* h = nullptr;
  int rc = report_run(c, s, VS_REPORT_TRACK, 0, &h);
  if (rc) return rc;
  out->n_tracked = h->n_trk; out->n_lost = h->n_lost; out->n_tracked_landmarks = h->n_tracked_landmarks;
  out->tracked4 = reinterpret_cast<const int32_t*>(c->report + c->rl.trk);
  out->lost = reinterpret_cast<const int32_t*>(c->report + c->rl.lost);
  return VSLAM_OK;
}
VS_API int vslam_view_aligner(vslam_ctx* c, int s, vslam_aligner_view* out) {
  if (!c || !out) return VSLAM_ERR_INVALID;
  const ReportHeader* h = nullptr;
  int rc = report_run(c, s, VS_REPORT_ALIGNER, 0, &h);
  if (rc) return rc;
  out->n = h->al_n; out->n_inliers = h->al_inliers; out->n_outliers = h->al_outliers; out->iterations = h->al_iterations;
  out->converged = h->al_converged; out->total_error = h->al_total_error;
  out->chi = reinterpret_cast<const double*>(c->report + c->rl.chi);
  out->inlier = c->report + c->rl.inl;
  std::memcpy(out->T, h->al_T, sizeof out->T);
  std::memcpy(out->H, h->al_H, sizeof out->H);
  return VSLAM_OK;
}
VS_API int vslam_view_points(vslam_ctx* c, int s, int in_progress, vslam_points_view* out) {
  if (!c || !out) return VSLAM_ERR_INVALID;
  const ReportHeader* h = nullptr;
  int rc = report_run(c, s, VS_REPORT_POINTS, in_progress ? 1 : 0, &h);
  if (rc) return rc;
  out->n = h->n_points;
  out->kp = reinterpret_cast<const int16_t*>(c->report + c->rl.p_kp);
  out->meta = reinterpret_cast<const int32_t*>(c->report + c->rl.p_meta);
  out->cam = reinterpret_cast<const double*>(c->report + c->rl.p_cam);
  out->desc = in_progress ? c->report + c->rl.p_desc : nullptr;
  out->first_full = in_progress ? std::min(h->n_after_prune, h->n_points) : 0;
  out->info = h->info;
  // the generator's chronometers from the same report (no further copy): accumulated seconds like vslam_get_timers
  const double inv = 1e-8;
  out->seconds_tracking = (double)h->ticks[0] * inv; out->seconds_pose_optimization = (double)h->ticks[1] * inv;
  out->seconds_point_recovery = (double)h->ticks[2] * inv; out->seconds_landmark_optimization = (double)h->ticks[3] * inv;
  out->seconds_point_triangulation = (double)h->ticks[4] * inv;
  return VSLAM_OK;
}

// ---- RGB-D mode -----------------------------------------------------------------------------------------------------------------
// Two implementations behind the same entry points: the device-resident loop (csrc/rgbd_device.h + kernels_rgbd.h; the default) and the
// host-driven loop over the library's own stand-alone entry points (csrc/rgbd_tracker.h; VSLAM_RGBD_HOST=1), kept as the cross-check.
#include "rgbd_tracker.h"
#include "rgbd_device.h"
struct vslam_rgbd {
  bool on_host = false, host_pending = false;
  int host_rc = 0;
  vs_rgbd::Tracker t;
  vs_rgbd::DeviceTracker d;
  std::string& err() { return on_host ? t.err : d.err; }
};
static thread_local std::string g_rgbd_error;
VS_API const char* vslam_rgbd_last_error(const vslam_rgbd* r) { return r ? (r->on_host ? r->t.err.c_str() : r->d.err.c_str()) : g_rgbd_error.c_str(); }
static int rgbd_create(const vslam_config* cfg, const vslam_depth_params* p, int device, int n_streams, vslam_rgbd** out);
VS_API int vslam_rgbd_wait(vslam_rgbd* r);
VS_API int vslam_rgbd_get_frame_info(vslam_rgbd* r, vslam_frame_info* out, int32_t* n_temporary);
VS_API int vslam_rgbd_get_points(vslam_rgbd* r, int32_t cap, int32_t* n, float* xy, double* cam, int32_t* meta4, uint8_t* desc);
VS_API int vslam_rgbd_create(const vslam_config* cfg, const vslam_depth_params* p, int device, vslam_rgbd** out) { return rgbd_create(cfg, p, device, 1, out); }
VS_API int vslam_rgbd_create_batch(const vslam_config* cfg, const vslam_depth_params* p, int device, int32_t n_streams, vslam_rgbd** out) {
  return rgbd_create(cfg, p, device, n_streams, out);
}
static int rgbd_create(const vslam_config* cfg, const vslam_depth_params* p, int device, int n_streams, vslam_rgbd** out) {
  if (!cfg || !p || !out) { g_rgbd_error = "vslam_rgbd_create: null argument"; return VSLAM_ERR_INVALID; }
  vslam_rgbd* r = new vslam_rgbd;
  if (const char* e = std::getenv("VSLAM_RGBD_HOST")) r->on_host = std::atoi(e) != 0;
  // detector_type ORB (no shipped configuration): the OrbDetector is a host-driven sequence of per-level kernels (vslam_orb_detect) and several
  // features can share a pixel — the device-resident loop's image pipeline is FAST's; the host-driven loop serves this mode
  if (p->detector_type == VSLAM_DETECTOR_ORB) r->on_host = true;
  else if (p->detector_type != VSLAM_DETECTOR_FAST) { g_rgbd_error = "vslam_rgbd_create: unknown detector_type"; delete r; return VSLAM_ERR_INVALID; }
  if (r->on_host && n_streams != 1) { g_rgbd_error = "vslam_rgbd_create_batch: the host-driven loop (VSLAM_RGBD_HOST=1, detector_type ORB) tracks one sequence per object"; delete r; return VSLAM_ERR_INVALID; }
  const int rc = r->on_host ? r->t.create(*cfg, *p, device) : r->d.create(*cfg, *p, device, n_streams);
  if (rc != VSLAM_OK) { g_rgbd_error = r->err(); delete r; return rc; }
  *out = r;
  return VSLAM_OK;
}
VS_API void vslam_rgbd_destroy(vslam_rgbd* r) { delete r; }
VS_API int vslam_rgbd_reset(vslam_rgbd* r) {
  if (!r) return VSLAM_ERR_INVALID;
  if (r->on_host) { r->t.reset(); return VSLAM_OK; }
  return r->d.reset();
}
VS_API int vslam_rgbd_process_host(vslam_rgbd* r, const uint8_t* left, int32_t lstride, const uint16_t* depth, int32_t dstride) {
  if (!r) return VSLAM_ERR_INVALID;
  if (!left || !depth) { r->err() = "called with empty frame"; return VSLAM_ERR_INVALID; }   // depth_framepoint_generator.cpp:48-50
  const int cols = r->on_host ? r->t.cfg.cols : r->d.cfg.cols;
  if (lstride < cols || dstride < cols) { r->err() = "row stride smaller than image width"; return VSLAM_ERR_INVALID; }
  return r->on_host ? r->t.process(left, lstride, depth, dstride) : r->d.process(left, lstride, depth, dstride);
}
VS_API int vslam_rgbd_submit_host(vslam_rgbd* r, const uint8_t* left, int32_t lstride, const uint16_t* depth, int32_t dstride) {
  if (!r) return VSLAM_ERR_INVALID;
  if (!left || !depth) { r->err() = "called with empty frame"; return VSLAM_ERR_INVALID; }
  const int cols = r->on_host ? r->t.cfg.cols : r->d.cfg.cols;
  if (lstride < cols || dstride < cols) { r->err() = "row stride smaller than image width"; return VSLAM_ERR_INVALID; }
  if (r->on_host) { r->host_rc = r->t.process(left, lstride, depth, dstride); r->host_pending = true; return r->host_rc; }   // the host-driven loop has nothing to overlap
  return r->d.submit(left, lstride, depth, dstride);
}
VS_API int vslam_rgbd_wait(vslam_rgbd* r) {
  if (!r) return VSLAM_ERR_INVALID;
  if (r->on_host) {
    if (!r->host_pending) { r->t.err = "RGB-D tracker: no frame in flight"; return VSLAM_ERR_STATE; }
    r->host_pending = false;
    return r->host_rc;
  }
  return r->d.wait();
}
VS_API int vslam_rgbd_submit_batch_host(vslam_rgbd* r, const uint8_t* left, int32_t lstride, size_t left_stream_stride, const uint16_t* depth, int32_t dstride,
                                        size_t depth_stream_stride) {
  if (!r) return VSLAM_ERR_INVALID;
  if (r->on_host) { r->t.err = "batch entry points need the device-resident loop"; return VSLAM_ERR_STATE; }
  if (!left || !depth) { r->d.err = "called with empty frame"; return VSLAM_ERR_INVALID; }
  if (lstride < r->d.cfg.cols || dstride < r->d.cfg.cols) { r->d.err = "row stride smaller than image width"; return VSLAM_ERR_INVALID; }
  return r->d.submit(left, lstride, depth, dstride, left_stream_stride, depth_stream_stride);
}
VS_API int vslam_rgbd_process_batch_host(vslam_rgbd* r, const uint8_t* left, int32_t lstride, size_t left_stream_stride, const uint16_t* depth, int32_t dstride,
                                         size_t depth_stream_stride) {
  const int rc = vslam_rgbd_submit_batch_host(r, left, lstride, left_stream_stride, depth, dstride, depth_stream_stride);
  return rc != VSLAM_OK ? rc : vslam_rgbd_wait(r);
}
VS_API int vslam_rgbd_submit_batch_device(vslam_rgbd* r, const uint8_t* left, int32_t lstride, size_t left_stream_stride, const uint16_t* depth, int32_t dstride,
                                          size_t depth_stream_stride) {
  if (!r) return VSLAM_ERR_INVALID;
  if (r->on_host) { r->t.err = "device images need the device-resident loop"; return VSLAM_ERR_STATE; }
  if (!left || !depth) { r->d.err = "called with empty frame"; return VSLAM_ERR_INVALID; }
  if (lstride < r->d.cfg.cols || dstride < r->d.cfg.cols) { r->d.err = "row stride smaller than image width"; return VSLAM_ERR_INVALID; }
  return r->d.submit(left, lstride, depth, dstride, left_stream_stride, depth_stream_stride, true);
}
VS_API int vslam_rgbd_get_frame_info_stream(vslam_rgbd* r, int32_t stream, vslam_frame_info* out, int32_t* n_temporary) {
  if (!r || !out) return VSLAM_ERR_INVALID;
  if (r->on_host) return stream == 0 ? vslam_rgbd_get_frame_info(r, out, n_temporary) : VSLAM_ERR_INVALID;
  if (stream < 0 || stream >= r->d.B) { r->d.err = "stream index out of range"; return VSLAM_ERR_INVALID; }
  if (r->d.frame_in_flight()) { r->d.err = "RGB-D tracker: a frame is in flight (call vslam_rgbd_wait first)"; return VSLAM_ERR_STATE; }
  *out = r->d.hosts[stream].info;
  if (n_temporary) *n_temporary = r->d.hosts[stream].n_temporary;
  return VSLAM_OK;
}
VS_API int vslam_rgbd_get_points_stream(vslam_rgbd* r, int32_t stream, int32_t cap, int32_t* n, float* xy, double* cam, int32_t* meta4, uint8_t* desc) {
  if (!r || !n) return VSLAM_ERR_INVALID;
  if (r->on_host) return stream == 0 ? vslam_rgbd_get_points(r, cap, n, xy, cam, meta4, desc) : VSLAM_ERR_INVALID;
  return r->d.get_points(stream, cap, n, xy, cam, meta4, desc);
}
VS_API int vslam_rgbd_get_frame_info(vslam_rgbd* r, vslam_frame_info* out, int32_t* n_temporary) {
  if (!r || !out) return VSLAM_ERR_INVALID;
  if (r->on_host) { *out = r->t.info; if (n_temporary) *n_temporary = r->t.n_temporary; return VSLAM_OK; }
  if (r->d.frame_in_flight()) { r->d.err = "RGB-D tracker: a frame is in flight (call vslam_rgbd_wait first)"; return VSLAM_ERR_STATE; }
  *out = r->d.host.info;
  if (n_temporary) *n_temporary = r->d.host.n_temporary;
  return VSLAM_OK;
}
VS_API int vslam_rgbd_get_points(vslam_rgbd* r, int32_t cap, int32_t* n, float* xy, double* cam, int32_t* meta4, uint8_t* desc) {
  if (!r || !n) return VSLAM_ERR_INVALID;
  if (!r->on_host) return r->d.get_points(0, cap, n, xy, cam, meta4, desc);
  if (r->t.info.frame_index == 0) { *n = 0; return VSLAM_OK; }
  const vs_rgbd::Fr& f = r->t.current();
  *n = (int32_t)f.points.size();
  if (*n > cap) { r->t.err = "point output capacity too small"; return VSLAM_ERR_CAPACITY; }
  for (int i = 0; i < *n; ++i) {
    const vs_rgbd::Pt& q = r->t.point(f.points[i]);
    if (xy) { xy[2 * i] = q.xy[0]; xy[2 * i + 1] = q.xy[1]; }
    if (cam) for (int k = 0; k < 3; ++k) cam[3 * i + k] = q.cam[k];
    if (meta4) { meta4[4 * i] = r->t.previous_index(q); meta4[4 * i + 1] = q.track_len; meta4[4 * i + 2] = q.landmark >= 0 ? r->t.landmarks()[q.landmark].updates : 0; meta4[4 * i + 3] = q.unreliable ? 1 : 0; }
    if (desc) std::memcpy(desc + (size_t)32 * i, q.desc, 32);
  }
  return VSLAM_OK;
}

// ---- pose all-gather on RCCL (loaded lazily: the single-GPU path has no dependency on librccl.so) ----------------------------
#include <dlfcn.h>
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, struct Id128, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
struct Id128 { char internal[VSLAM_COMM_ID_BYTES]; };   // ncclUniqueId: passed BY VALUE to ncclCommInitRank
RcclApi g_rccl;
thread_local std::string g_comm_error;
int comm_fail(int code, const std::string& msg) { g_comm_error = msg; return code; }
int rccl_load() {
  if (g_rccl.lib) return VSLAM_OK;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return comm_fail(VSLAM_ERR_NO_DEVICE, std::string("librccl.so not found: ") + dlerror());
  RcclApi a;
  a.lib = h;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy || !a.GetErrorString) return comm_fail(VSLAM_ERR_NO_DEVICE, "librccl.so lacks an expected symbol");
  g_rccl = a;
  return VSLAM_OK;
}
}  // namespace
struct vslam_comm { void* comm = nullptr; int rank = 0, nranks = 1, device = 0; };
VS_API const char* vslam_comm_last_error(void) { return g_comm_error.c_str(); }
VS_API int vslam_comm_available(int device) {
  int rc = rccl_load();
  if (rc != VSLAM_OK) return rc;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return comm_fail(VSLAM_ERR_NO_DEVICE, "vslam_comm_available: no such HIP device");
  return VSLAM_OK;
}
VS_API int vslam_comm_unique_id(uint8_t id[VSLAM_COMM_ID_BYTES]) {
  if (!id) return comm_fail(VSLAM_ERR_INVALID, "null id");
  int rc = rccl_load();
  if (rc != VSLAM_OK) return rc;
  Id128 u;
  const int r = g_rccl.GetUniqueId(&u);
  if (r != 0) return comm_fail(VSLAM_ERR_HIP, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
  std::memcpy(id, u.internal, VSLAM_COMM_ID_BYTES);
  return VSLAM_OK;
}
VS_API int vslam_comm_init(int rank, int nranks, const uint8_t id[VSLAM_COMM_ID_BYTES], int device, vslam_comm** out) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return comm_fail(VSLAM_ERR_INVALID, "vslam_comm_init: bad argument");
  int rc = rccl_load();
  if (rc != VSLAM_OK) return rc;
  if (hipSetDevice(device) != hipSuccess) return comm_fail(VSLAM_ERR_NO_DEVICE, "vslam_comm_init: hipSetDevice failed");
  Id128 u;
  std::memcpy(u.internal, id, VSLAM_COMM_ID_BYTES);
  vslam_comm* c = new vslam_comm;
  c->rank = rank; c->nranks = nranks; c->device = device;
  const int r = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
  if (r != 0) { delete c; return comm_fail(VSLAM_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
  *out = c;
  return VSLAM_OK;
}
VS_API int vslam_allgather_poses(vslam_comm* c, const double* send, double* recv, size_t count, void* stream) {
  if (!c || !send || !recv) return comm_fail(VSLAM_ERR_INVALID, "vslam_allgather_poses: bad argument");
  if (count == 0) return VSLAM_OK;
  if (hipSetDevice(c->device) != hipSuccess) return comm_fail(VSLAM_ERR_NO_DEVICE, "hipSetDevice failed");
  const int r = g_rccl.AllGather(send, recv, count, /*ncclDouble*/ 8, c->comm, (hipStream_t)stream);
  if (r != 0) return comm_fail(VSLAM_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
  return VSLAM_OK;
}
VS_API void vslam_comm_destroy(vslam_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

// profiling aid (not part of the ABI): mean per-stream ticks of the fine-grained phase stamps, in microseconds
VS_API int vslam_debug_ticks(vslam_ctx* c, double us[12]) {
  if (!c || !us) return VSLAM_ERR_INVALID;
  harvest_events(c);
  std::vector<StreamState> st(c->B);
  HIP_TRY(c, hipMemcpy(st.data(), c->buf.st, sizeof(StreamState) * c->B, hipMemcpyDeviceToHost));
  for (int k = 0; k < 12; ++k) { double a = 0; for (int s = 0; s < c->B; ++s) a += (double)st[s].dbg[k]; us[k] = a * 1e-2 / c->B; }
  return VSLAM_OK;
}
// profiling aid (not part of the ABI): per stream, the 5 chronometer tick counters followed by the 12 phase stamps
// (cumulative, 100 MHz ticks)
VS_API int vslam_debug_stream_ticks(vslam_ctx* c, unsigned long long* out /* [B][17] */) {
  if (!c || !out) return VSLAM_ERR_INVALID;
  std::vector<StreamState> st(c->B);
  HIP_TRY(c, hipMemcpy(st.data(), c->buf.st, sizeof(StreamState) * c->B, hipMemcpyDeviceToHost));
  for (int s = 0; s < c->B; ++s) {
    for (int k = 0; k < 5; ++k) out[17 * s + k] = st[s].ticks[k];
    for (int k = 0; k < 12; ++k) out[17 * s + 5 + k] = st[s].dbg[k];
  }
  return VSLAM_OK;
}
