// rgbd_device.h — host side of the device-resident RGB-D loop (kernels_rgbd.h): buffers, the per-frame launch sequence on one HIP stream,
// one small read-back per frame.  Included by vslam_hip.hip after the context code (it uses create_internal / buf_set / fail of that file).
//
// A frame is: two host-to-device copies (image, depth), ~17 kernel launches on two streams (the space map runs beside the image pipeline), one 1.2 KB device-to-host copy, one stream synchronisation.
// A context tracks n_streams independent sequences side by side (vslam_rgbd_create_batch): the same ~17 launches serve all of them — one
// workgroup per sequence in the single-workgroup kernels, a grid dimension in the wide ones —, a step takes the time of its slowest sequence.  When
// the registration asks for another attempt (RgbdState::done still 0 after the tail skipped itself: pose_tracker_3d.cpp:333-418, a lost
// track), the block image pipeline .. aligner .. tail is enqueued again, at most twice.
#pragma once
#include "kernels_rgbd.h"

namespace vs_rgbd {

class DeviceTracker {
public:
  vslam_ctx* ic = nullptr;       // inner context: image pipeline buffers, aligner SoA, detector thresholds (one stream per sequence)
  vslam_config cfg;
  vslam_depth_params p;
  std::string err;
  int B = 1;                     // sequences tracked side by side
  std::vector<RgbdState> hosts;  // last state block of every sequence
  RgbdState host;                // == hosts[0] (the one-sequence entry points)
  bool failed = false;
  std::string failed_why;

  ~DeviceTracker() { release(); }

  int create(const vslam_config& c, const vslam_depth_params& dp, int device, int n_streams = 1) {
    cfg = c; p = dp; B = n_streams;
    if (B < 1 || B > 1024) { err = "RGB-D mode: 1 .. 1024 sequences per context"; return VSLAM_ERR_INVALID; }
    if (cfg.det_rows < 1 || cfg.det_cols < 1 || cfg.det_rows * cfg.det_cols > VSLAM_MAX_REGIONS) { err = "RGB-D mode: bad detector grid"; return VSLAM_ERR_INVALID; }
    if (p.rows != cfg.rows || p.cols != cfg.cols) { err = "RGB-D mode: depth parameters and configuration disagree on the image size"; return VSLAM_ERR_INVALID; }
    if (cfg.max_points > 65535) { err = "RGB-D mode: max_points above 65535 (16-bit trail indices)"; return VSLAM_ERR_INVALID; }
    vslam_config in = cfg;
    in.descriptor_type = p.descriptor_type;      // the extractor initialize() uses (depth_framepoint_generator.cpp:24-44 -> computeDescriptors)
    in.max_history_frames = 2;                    // the stereo tracker's history ring is not used in this mode
    int rc = create_internal(&in, device, B, &ic);
    if (rc != VSLAM_OK) { err = vslam_last_error(nullptr); return rc; }
    q = ic->stream_img;
    ic->cfg.mono = 1;                             // the tile kernels run ONE image per sequence (grid z = sequences); k_emit's one-image controller
    const size_t MAXP = ic->cfg.MAXP, NMAX = ic->cfg.NMAX, npx = (size_t)p.rows * p.cols, nB = (size_t)B;
    if (const char* e = std::getenv("VSLAM_RGBD_WG")) { const int v = std::atoi(e); if (v == 256 || v == 512 || v == 1024) wg1 = v; }
    H = std::max(4, std::min(cfg.max_history_frames, 512));   // measurements of a track the landmark refinement can address
    TR = H - 2;
    const int rows_bin = p.enable_keypoint_binning ? p.rows / std::max(p.bin_size_pixels, 1) + 1 : 0, cols_bin = p.enable_keypoint_binning ? p.cols / std::max(p.bin_size_pixels, 1) + 1 : 0;
    const size_t nbins = (size_t)(rows_bin + 1) * (cols_bin + 1);
    std::memset(&rb, 0, sizeof rb);
    rb.p = p; rb.TR = TR; rb.H = H;
    rb.MAXP = (int32_t)MAXP; rb.NMAX = (int32_t)NMAX; rb.npx = (int32_t)npx; rb.nbins = (int32_t)nbins; rb.n_streams = B;
    hipError_t e = hipSuccess;
    auto A = [&](auto** ptr, size_t count) { if (e == hipSuccess) e = dalloc(ic, ptr, count * nB); };      // n_streams slices of the per-sequence size
    A(&rb.st, 1);
    for (RgbdList* l : {&rb.fl[0], &rb.fl[1], &rb.tmp}) {
      A(&l->xy, MAXP * 2); A(&l->desc, MAXP * 32); A(&l->cam, MAXP * 3); A(&l->prev, MAXP); A(&l->tlen, MAXP); A(&l->flags, MAXP);
      A(&l->lmw, MAXP * 3); A(&l->lmu, MAXP); A(&l->lmm, MAXP);
      if (l != &rb.tmp) A(&l->trail, MAXP * (size_t)TR);
    }
    A(&rb.order, NMAX); A(&rb.matched, NMAX);
    rb.ncsr = (int32_t)((size_t)ic->cfg.c.rows * (ic->cfg.CW + 1));
    A(&rb.a_kxy, NMAX * 2); A(&rb.a_desc, NMAX * 32); A(&rb.a_score, NMAX); A(&rb.a_rowcell, (size_t)rb.ncsr); A(&rb.a_order, NMAX); A(&rb.a_vis, NMAX);
    A(&rb.t_kxy, NMAX * 2); A(&rb.t_desc, NMAX * 32); A(&rb.t_score, NMAX); A(&rb.t_order, NMAX); A(&rb.m_rank, NMAX * 2); A(&rb.fvis, NMAX);
    A(&d_depth, npx); A(&rb.dkey, npx); A(&rb.dlast, npx); A(&rb.space, npx * 3); A(&rb.row_map, npx); A(&rb.col_map, npx);
    A(&rb.hold, NMAX * 2); A(&rb.pick, MAXP); A(&rb.cand, MAXP * (VS_DT_K + 1)); A(&rb.out2, MAXP * 2); A(&rb.xyz, MAXP * 3); A(&rb.temp2, MAXP * 2); A(&rb.lost_raw, MAXP);
    A(&rb.lost, MAXP); A(&rb.lost_has, MAXP); A(&rb.lost_lm, MAXP * 3); A(&rb.lost_desc, MAXP * 32);
    A(&rb.rbxy, MAXP * 2); A(&rb.rkxy, MAXP * 2); A(&rb.rcell, MAXP); A(&rb.rkeep, MAXP); A(&rb.rdesc, MAXP * 32); A(&rb.ridx, MAXP); A(&rb.rxy, MAXP * 2);
    A(&rb.rrdesc, MAXP * 32); A(&rb.rxyz, MAXP * 3);
    A(&rb.rcF, NMAX * 2); A(&rb.remf, NMAX); A(&rb.rcT, MAXP * 2); A(&rb.bins, nbins); A(&rb.cls, NMAX);
    A(&rb.new_feat, NMAX); A(&rb.new_xyz, NMAX * 3); A(&rb.temp_feat, NMAX); A(&rb.temp_xyz, NMAX * 3);
    A(&rb.weights, MAXP); A(&rb.cross, 1);
    A(&rb.h_cam, (size_t)H * MAXP * 4); A(&rb.h_pose, (size_t)H * 24); A(&rb.pose_log, (size_t)VS_POSE_LOG * 12);
    if (e == hipSuccess) e = hipHostMalloc((void**)&pinned, sizeof(RgbdState) * nB, hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&q2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_depth, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
    if (const char* g = std::getenv("VSLAM_RGBD_GRAPH")) use_graph = std::atoi(g) != 0;
    if (e != hipSuccess) { err = std::string("RGB-D mode: ") + hipGetErrorString(e); release(); return VSLAM_ERR_HIP; }
    rb.depth = d_depth;
    {   // the premise of k_depth_direct, as kernels_depth.h depth_source tests it
      const double* Ki = p.K_right_inverse; const double* K = p.K_left; const double* T = p.right_to_left;
      depth_plain = Ki[1] == 0 && Ki[3] == 0 && Ki[6] == 0 && Ki[7] == 0 && Ki[8] == 1 && K[1] == 0 && K[3] == 0 && K[6] == 0 && K[7] == 0 && K[8] == 1 &&
                    T[0] == 1 && T[1] == 0 && T[2] == 0 && T[3] == 0 && T[4] == 0 && T[5] == 1 && T[6] == 0 && T[7] == 0 && T[8] == 0 && T[9] == 0 && T[10] == 1 && T[11] == 0;
      if (const char* e2 = std::getenv("VSLAM_RGBD_DEPTH_DIRECT")) depth_plain = depth_plain && std::atoi(e2) != 0;
      if (const char* e3 = std::getenv("VSLAM_RGBD_DEPTH_FORCE_CROSS")) force_cross = std::atoi(e3) != 0;
    }
    hosts.resize(B);
    return reset();
  }

  int reset() {
    if (!ic) return VSLAM_ERR_STATE;
    (void)hipSetDevice(ic->device);
    (void)hipStreamSynchronize(q);
    if (q2) (void)hipStreamSynchronize(q2);
    int rc = init_state(ic);                                                 // detector thresholds back to the minimum (FastDetector of a fresh generator)
    if (rc != VSLAM_OK) { err = ic->err; return rc; }
    RgbdState s;
    std::memset(&s, 0, sizeof s);
    s.status = VSLAM_LOCALIZING;
    s.win = cfg.maximum_projection_tracking_distance_pixels;
    s.tau_track = cfg.minimum_descriptor_distance_tracking;
    tf_identity(s.prior); tf_identity(s.world);
    for (int i = 0; i < B; ++i) hosts[i] = s;
    host = s;
    hipError_t e = hipMemcpy(rb.st, hosts.data(), sizeof(RgbdState) * (size_t)B, hipMemcpyHostToDevice);
    if (e != hipSuccess) { err = hipGetErrorString(e); return VSLAM_ERR_HIP; }
    {   // the space maps' z-buffers: initialised here, every frame's k_depth_write puts them back
      const size_t npx = (size_t)p.rows * p.cols;
      const float f0 = (float)p.maximum_depth_meters;
      uint32_t f0_bits;
      std::memcpy(&f0_bits, &f0, 4);
      hipLaunchKernelGGL(k_depth_init, dim3((unsigned)((npx + 255) / 256), B), dim3(256), 0, q, (int)npx, f0_bits, rb.dkey, rb.dlast);
      (void)hipMemsetAsync(rb.cross, 0, sizeof(int32_t) * (size_t)B, q);
      if (hipStreamSynchronize(q) != hipSuccess) { err = "RGB-D reset: space-map buffers"; return VSLAM_ERR_HIP; }
    }
    failed = false; failed_why.clear(); pending = false;
    return VSLAM_OK;
  }

  int process(const uint8_t* left, int32_t lstride, const uint16_t* depth, int32_t dstride, size_t left_stream_stride = 0, size_t depth_stream_stride = 0) {
    const int rc = submit(left, lstride, depth, dstride, left_stream_stride, depth_stream_stride);
    return rc != VSLAM_OK ? rc : wait();
  }
  // submit(): copies the frame(s) in and enqueues the kernels; returns without waiting.  wait(): the result (and, for the rare frame whose
  // registration asks for another attempt, the further attempts).  left / depth: n_streams images, `*_stream_stride` bytes / elements apart.
  // on_device: left / depth are DEVICE pointers (images already in HBM, e.g. written by a capture pipeline): nothing is copied, the kernels
  // read them where they are; the depth images must then be dense per sequence (depth_stream_stride == rows * depth_row_stride)
  int submit(const uint8_t* left, int32_t lstride, const uint16_t* depth, int32_t dstride, size_t left_stream_stride = 0, size_t depth_stream_stride = 0, bool on_device = false) {
    if (!left || !depth) { err = "called with empty frame"; return VSLAM_ERR_INVALID; }
    if (on_device && B > 1 && depth_stream_stride != (size_t)p.rows * dstride) { err = "RGB-D batch on device images: depth images must be dense per sequence"; return VSLAM_ERR_INVALID; }
    if (failed) { err = "RGB-D tracker: an earlier frame failed (" + failed_why + "); reset() before the next frame"; return VSLAM_ERR_STATE; }
    if (pending) { err = "RGB-D tracker: the previous frame has not been waited for"; return VSLAM_ERR_STATE; }
    if (B > 1 && (left_stream_stride < (size_t)(p.rows - 1) * lstride + p.cols || depth_stream_stride < (size_t)(p.rows - 1) * dstride + p.cols)) {
      err = "RGB-D batch: stream strides smaller than an image"; return VSLAM_ERR_INVALID;
    }
    const int rc = submit_frame(left, lstride, depth, dstride, left_stream_stride, depth_stream_stride, on_device);
    if (rc != VSLAM_OK) { failed = true; failed_why = err; }
    else pending = true;
    return rc;
  }
  int wait() {
    if (!pending) { err = "RGB-D tracker: no frame in flight"; return VSLAM_ERR_STATE; }
    pending = false;
    const int rc = finish_frame();
    if (rc != VSLAM_OK) { failed = true; failed_why = err; }
    return rc;
  }

  bool frame_in_flight() const { return pending; }
  int get_points(int stream, int32_t cap, int32_t* n, float* xy, double* cam, int32_t* meta4, uint8_t* desc) {
    if (stream < 0 || stream >= B) { err = "stream index out of range"; return VSLAM_ERR_INVALID; }
    // a frame between submit() and wait() is rewriting the lists these copies read (and flags of the previous list): not readable
    if (pending) { err = "RGB-D tracker: a frame is in flight (call vslam_rgbd_wait first)"; return VSLAM_ERR_STATE; }
    const RgbdState& hs = hosts[stream];
    if (hs.frame_count == 0) { *n = 0; return VSLAM_OK; }
    const int np = hs.last_points;
    *n = np;
    if (np > cap) { err = "point output capacity too small"; return VSLAM_ERR_CAPACITY; }
    if (np == 0) return VSLAM_OK;
    const RgbdList& l = rb.fl[(hs.frame_count - 1) & 1];
    const size_t P = (size_t)rb.MAXP, o = (size_t)stream * P;
    hipError_t e = hipSuccess;
    if (xy) e = hipMemcpy(xy, l.xy + o * 2, (size_t)np * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && cam) e = hipMemcpy(cam, l.cam + o * 3, (size_t)np * 24, hipMemcpyDeviceToHost);
    if (e == hipSuccess && desc) e = hipMemcpy(desc, l.desc + o * 32, (size_t)np * 32, hipMemcpyDeviceToHost);
    if (e == hipSuccess && meta4) {
      std::vector<int32_t> prev(np), tlen(np), lmu(np); std::vector<uint8_t> fl(np);
      e = hipMemcpy(prev.data(), l.prev + o, (size_t)np * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(tlen.data(), l.tlen + o, (size_t)np * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(lmu.data(), l.lmu + o, (size_t)np * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(fl.data(), l.flags + o, (size_t)np, hipMemcpyDeviceToHost);
      for (int i = 0; i < np && e == hipSuccess; ++i) {
        meta4[4 * i] = prev[i]; meta4[4 * i + 1] = tlen[i]; meta4[4 * i + 2] = (fl[i] & RGBD_F_LM) ? lmu[i] : 0; meta4[4 * i + 3] = (fl[i] & RGBD_F_UNREL) ? 1 : 0;
      }
    }
    if (e != hipSuccess) { err = hipGetErrorString(e); return VSLAM_ERR_HIP; }
    return VSLAM_OK;
  }

private:
  RgbdBuf rb;
  hipStream_t q = nullptr, q2 = nullptr;      // q: everything of a frame; q2: the depth image's copy and the space map, beside the image pipeline
  hipEvent_t ev_depth = nullptr;
  int H = 0, TR = 0;
  int wg1 = 1024;                // threads of the single-workgroup kernels (VSLAM_RGBD_WG = 256 | 512 | 1024; measured at 620 x 188, ~700 features: 0.256 / 0.258 / 0.265 ms per frame for 1024 / 512 / 256: their barriers are not what a frame waits for)
  uint16_t* d_depth = nullptr;   // [B][rows * cols], rows re-packed
  const uint16_t* depth_src = nullptr; int32_t depth_src_stride = 0;   // what the space-map kernels read this frame: d_depth, or the caller's device images
  uint8_t* d_img = nullptr; size_t img_stream = 0;     // [B][img_stream] bytes, the caller's row stride kept
  RgbdState* pinned = nullptr;   // [B]
  bool depth_pending = false, pending = false, depth_plain = false, force_cross = false;
  DevBuf bs;                      // the inner context's buffer table with this frame's image pointers
  // VSLAM_RGBD_GRAPH=1 (opt-in): the frame's launch sequence (depth map on q2 beside the image pipeline on q, registration, tail, the state
  // block's copy out) captured once into a hipGraph and replayed — two copies and ONE launch per frame instead of ~22.  Measured on MI355X /
  // ROCm 7.2 (tests/validation/rgbd_submit_time.py): submit() takes 80 us of host time either way (the two pageable copies; the runtime replays a
  // graph node by node) and the frame 0.31 against 0.30 ms: the frame is bound by its dependent kernels, not by their launches.  Kept as a
  // tested switch, captured again when the image stride or buffer changes.
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  bool use_graph = false;
  int32_t graph_stride = -1;
  const uint8_t* graph_img = nullptr;
  hipEvent_t ev_fork = nullptr;

  void release() {
    // an un-waited frame may still be copying into `pinned` and running on q / q2: both queues drain before anything is freed
    if (q) (void)hipStreamSynchronize(q);
    if (q2) (void)hipStreamSynchronize(q2);
    pending = false;
    if (pinned) { (void)hipHostFree(pinned); pinned = nullptr; }
    drop_graph();
    if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
    if (q2) { (void)hipStreamDestroy(q2); q2 = nullptr; }
    if (ev_depth) { (void)hipEventDestroy(ev_depth); ev_depth = nullptr; }
    if (d_img) { (void)hipFree(d_img); d_img = nullptr; }
    if (ic) { vslam_destroy(ic); ic = nullptr; }      // frees everything dalloc() registered
  }
  int hip_fail(hipError_t e, const char* where) { err = std::string(where) + ": " + hipGetErrorString(e); return VSLAM_ERR_HIP; }
  void all_active(DevBuf& b) const {
    std::memset(b.active, 0, sizeof b.active);
    for (int s = 0; s < B; ++s) b.active[s >> 5] |= 1u << (s & 31);
  }

  // initialize() .. registration of one attempt: the image pipeline on ONE image per sequence (DevCfg::mono: grid z = sequences; k_emit with
  // its one-image controller), track, aligner.  Sequences whose bit in b.active is cleared are skipped by every kernel.
  // again: attempt 2 / 3 of a frame — its keypoint vector keeps the earlier attempts' keypoints (frame_->keypointsLeft() is appended to and never
  // cleared between the initialize() calls of one frame: base_framepoint_generator.cpp:422, pose_tracker_3d.cpp:320,402): the list so far is
  // saved before the detection overwrites it and merged with the new one before anything reads the features.
  void enqueue_attempt(const DevBuf& b, bool again = false) {
    const DevCfg& d = ic->cfg;
    const dim3 g1(d.TX, (d.c.rows + VS_TILE_H - 1) / VS_TILE_H, B);
    if (again) hipLaunchKernelGGL(k_rgbd_save_features, dim3(B), dim3(1024), 0, q, d, b, rb);
    hipLaunchKernelGGL(k_fast_box, g1, dim3(256), VS_FB_DYN_LDS, q, d, b);
    const bool orb = d.c.descriptor_type == VSLAM_DESCRIPTOR_ORB;
    hipLaunchKernelGGL(k_emit, dim3(B, 1), dim3(512), 0, q, d, b, orb ? (int)VSLAM_ORB_BORDER : (int)VSLAM_BRIEF_BORDER, 2);
    const dim3 g3((d.c.cols + VS_BT_W - 1) / VS_BT_W, (d.c.rows + VS_BT_H - 1) / VS_BT_H, B);
    if (orb) {
      Gauss7 gk; for (int i = 0; i < 4; ++i) gk.k[i] = d.gauss7[i];
      hipLaunchKernelGGL(k_gauss7, g1, dim3(256), 0, q, d, b, gk);
      hipLaunchKernelGGL(k_orb_describe, g3, dim3(256), 0, q, d, b, d.orb_cos, d.orb_sin);
    } else {
      hipLaunchKernelGGL(k_brief, g3, dim3(256), 0, q, d, b);
    }
    if (again) hipLaunchKernelGGL(k_rgbd_merge_features, dim3(B), dim3(1024), 0, q, d, b, rb);
    if (depth_pending) { (void)hipStreamWaitEvent(q, ev_depth, 0); depth_pending = false; }     // the space map is first read here
    hipLaunchKernelGGL(k_rgbd_track_candidates, dim3(std::max(8, std::min(256, 4096 / B)), B), dim3(256), 0, q, d, b, rb);
    hipLaunchKernelGGL(k_rgbd_track, dim3(B), dim3(wg1), 0, q, d, b, rb);
    hipLaunchKernelGGL(k_rgbd_align, dim3(B), dim3(VS_WG), 0, q, d, b, rb);
  }
  void enqueue_tail(const DevBuf& b) {
    const DevCfg& d = ic->cfg;
    hipLaunchKernelGGL(k_rgbd_prune, dim3(B), dim3(wg1), 0, q, d, b, rb);
    hipLaunchKernelGGL(k_rgbd_describe_at, dim3(std::max(4, std::min(64, 1024 / B)), B), dim3(256), 0, q, d, b, rb);
    hipLaunchKernelGGL(k_rgbd_recover_finish, dim3(B), dim3(wg1), 0, q, d, rb);
    // 99 KB of LDS per workgroup: one per CU.  One sequence: a workgroup per 32 framepoints; many: a few workgroups per sequence that loop
    const int lm_gx = std::max(2, std::min((d.MAXP + RGBD_LM_PTS - 1) / RGBD_LM_PTS, 256 / B));
    hipLaunchKernelGGL(k_rgbd_landmarks, dim3(lm_gx, B), dim3(256), 0, q, d, rb);
    hipLaunchKernelGGL(k_rgbd_finish, dim3(B), dim3(wg1), 0, q, d, b, rb);
  }

  int submit_frame(const uint8_t* left, int32_t lstride, const uint16_t* depth, int32_t dstride, size_t lss, size_t dss, bool on_device) {
    (void)hipSetDevice(ic->device);
    const int rows = p.rows, cols = p.cols;
    if (on_device) {
      use_graph = false;                       // the captured graph holds the staging buffers' addresses
      bs = buf_set(ic, 0, 0);
      bs.img[0] = left; bs.img[1] = left; bs.img_row_stride = lstride; bs.img_stream_stride = lss;
      all_active(bs);
      depth_src = depth; depth_src_stride = dstride;
      enqueue_first_attempt(false);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(pinned, rb.st, sizeof(RgbdState) * (size_t)B, hipMemcpyDeviceToHost, q);
      if (e != hipSuccess) { ic->sticky = VSLAM_ERR_HIP; return hip_fail(e, "RGB-D frame"); }
      return VSLAM_OK;
    }
    depth_src = d_depth; depth_src_stride = cols;
    // inputs: the caller's row stride kept on the device for the image, the depth image re-packed
    const size_t ib = (size_t)(rows - 1) * lstride + cols;                 // bytes of one image the caller owns
    // device bytes per sequence: the caller's own stream stride when the images lie in one (nearly) dense block — then ONE copy brings all of
    // them in (a pageable copy costs ~40 us of host time whatever its size: 128 sequences are 10 ms of copies one by one, 0.6 ms as a block)
    const bool dense = B > 1 && lss >= (size_t)rows * lstride && lss <= (size_t)rows * lstride + 4096;
    const size_t need = dense ? lss : (((size_t)rows * lstride + 255) & ~(size_t)255);
    if (need != img_stream || !d_img) {
      (void)hipStreamSynchronize(q);
      if (d_img) (void)hipFree(d_img);
      d_img = nullptr; img_stream = 0;
      const hipError_t e = hipMalloc((void**)&d_img, need * (size_t)B + 64);
      if (e != hipSuccess) return hip_fail(e, "image buffer");
      img_stream = need;
    }
    hipError_t e = hipSuccess;
    // the depth images go in on the second stream, beside the images' copy (replaying a captured graph: on the first, the graph forks itself)
    hipStream_t qd = use_graph ? q : q2;
    if (dense) e = hipMemcpyAsync(d_img, left, (size_t)(B - 1) * lss + ib, hipMemcpyHostToDevice, q);
    else for (int s = 0; s < B && e == hipSuccess; ++s) e = hipMemcpyAsync(d_img + (size_t)s * img_stream, left + (size_t)s * lss, ib, hipMemcpyHostToDevice, q);
    if (e == hipSuccess) {
      if (dstride == cols && (B == 1 || dss == (size_t)rows * cols)) e = hipMemcpyAsync(d_depth, depth, (size_t)B * rows * cols * 2, hipMemcpyHostToDevice, qd);
      else for (int s = 0; s < B && e == hipSuccess; ++s) {
        const uint16_t* dsrc = depth + (size_t)s * dss;
        uint16_t* ddst = d_depth + (size_t)s * rows * cols;
        if (dstride == cols) e = hipMemcpyAsync(ddst, dsrc, (size_t)rows * cols * 2, hipMemcpyHostToDevice, qd);
        else e = hipMemcpy2DAsync(ddst, (size_t)cols * 2, dsrc, (size_t)dstride * 2, (size_t)cols * 2, rows, hipMemcpyHostToDevice, qd);
      }
    }
    if (e != hipSuccess) return hip_fail(e, "image / depth upload");
    bs = buf_set(ic, 0, 0);
    bs.img[0] = d_img; bs.img[1] = d_img; bs.img_row_stride = lstride; bs.img_stream_stride = img_stream;
    all_active(bs);
    if (use_graph && (!graph_exec || graph_stride != lstride || graph_img != d_img)) capture_graph(lstride);
    if (use_graph && graph_exec) {
      e = hipGraphLaunch(graph_exec, q);
      if (e != hipSuccess) { ic->sticky = VSLAM_ERR_HIP; return hip_fail(e, "RGB-D frame (graph launch)"); }
      return VSLAM_OK;
    }
    enqueue_first_attempt(false);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(pinned, rb.st, sizeof(RgbdState) * (size_t)B, hipMemcpyDeviceToHost, q);
    if (e != hipSuccess) { ic->sticky = VSLAM_ERR_HIP; return hip_fail(e, "RGB-D frame"); }
    return VSLAM_OK;
  }

  // everything of a frame's first attempt after the uploads: frame scalars, the space maps on q2 beside the image pipeline, registration, tail
  void enqueue_first_attempt(bool fork) {
    const int rows = p.rows, cols = p.cols;
    hipLaunchKernelGGL(k_rgbd_begin, dim3(B), dim3(64), 0, q, rb);
    // _computeDepthMap (once per frame: every initialize() of the frame sees the same depth image) on the second stream, joined before the
    // first reader of the map (the candidate kernel)
    if (fork) { (void)hipEventRecord(ev_fork, q); (void)hipStreamWaitEvent(q2, ev_fork, 0); }     // inside a capture: q2 joins the graph here
    const float f0 = (float)p.maximum_depth_meters;
    uint32_t f0_bits;
    std::memcpy(&f0_bits, &f0, 4);
    const dim3 grid((cols + 255) / 256, rows, B);
    if (depth_src != d_depth) { (void)hipEventRecord(ev_fork, q); (void)hipStreamWaitEvent(q2, ev_fork, 0); }   // device images: whatever wrote them was ordered before q
    // pinhole matrices, identity offset: one direct pass; the general three passes stay enqueued behind it and skip every image whose depth
    // pixels all landed on themselves (k_depth_direct checks that and raises rb.cross[image] otherwise)
    const int32_t* gate = nullptr;
    if (depth_plain) {
      if (force_cross) (void)hipMemsetAsync(rb.cross, 1, sizeof(int32_t) * (size_t)B, q2);     // test switch: the general passes run behind the direct one
      const dim3 dgrid((cols + 255) / 256, std::min(rows, std::max(8, 4096 / B)), B);      // one sequence: a workgroup per row segment; many: workgroups walk down their strip
      hipLaunchKernelGGL(k_depth_direct, dgrid, dim3(256), 0, q2, p, depth_src, depth_src_stride, f0_bits, rb.space, rb.row_map, rb.col_map, rb.cross);
      gate = rb.cross;
    }
    const dim3 ggrid((cols + 255) / 256, gate ? std::min(rows, 4) : rows, B);     // gated: a token grid (the kernels loop over the rows); it almost never has work
    hipLaunchKernelGGL(k_depth_min, ggrid, dim3(256), 0, q2, p, depth_src, depth_src_stride, rb.dkey, gate);
    hipLaunchKernelGGL(k_depth_pick, ggrid, dim3(256), 0, q2, p, depth_src, depth_src_stride, f0_bits, rb.dkey, rb.dlast, gate);
    hipLaunchKernelGGL(k_depth_write, ggrid, dim3(256), 0, q2, p, depth_src, depth_src_stride, f0_bits, rb.dkey, rb.dlast, rb.space, rb.row_map, rb.col_map, 1, gate);   // leaves the z-buffer initialised for the next frame
    (void)hipEventRecord(ev_depth, q2);
    depth_pending = true;
    enqueue_attempt(bs);
    enqueue_tail(bs);
  }

  void drop_graph() {
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
  }
  // stream capture of enqueue_first_attempt + the state block's copy out; any failure leaves the direct launches in charge
  void capture_graph(int32_t lstride) {
    drop_graph();
    (void)hipStreamSynchronize(q);
    hipError_t e = hipStreamBeginCapture(q, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { (void)hipGetLastError(); use_graph = false; return; }
    enqueue_first_attempt(true);
    (void)hipMemcpyAsync(pinned, rb.st, sizeof(RgbdState) * (size_t)B, hipMemcpyDeviceToHost, q);
    e = hipStreamEndCapture(q, &graph);
    depth_pending = false;
    if (e == hipSuccess) e = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess || !graph_exec) { (void)hipGetLastError(); drop_graph(); use_graph = false; return; }
    graph_stride = lstride; graph_img = d_img;
  }

  int finish_frame() {
    (void)hipSetDevice(ic->device);
    hipError_t e = hipSuccess;
    bool all_done = false;
    for (int attempt = 0; attempt < 3 && !all_done; ++attempt) {
      if (attempt) {
        // some sequence's registration asked for another attempt: initialize() .. aligner .. tail once more, for those sequences only (the
        // others' detector thresholds and features must not move: their bits in the activity mask are cleared)
        DevBuf b2 = bs;
        std::memset(b2.active, 0, sizeof b2.active);
        for (int s = 0; s < B; ++s) if (!hosts[s].tail_done) b2.active[s >> 5] |= 1u << (s & 31);
        enqueue_attempt(b2, true);
        enqueue_tail(b2);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(pinned, rb.st, sizeof(RgbdState) * (size_t)B, hipMemcpyDeviceToHost, q);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(q);
      if (e != hipSuccess) { ic->sticky = VSLAM_ERR_HIP; return hip_fail(e, "RGB-D frame"); }
      all_done = true;
      for (int s = 0; s < B; ++s) { hosts[s] = pinned[s]; all_done = all_done && hosts[s].tail_done; }
    }
    host = hosts[0];
    if (!all_done) { err = "RGB-D frame: registration did not finish in three attempts"; return VSLAM_ERR_STATE; }
    // bit 0: more corners than max_keypoints (k_emit), bit 1: more points than max_points — results would be truncated: the frame fails.
    // bit 2 (a track longer than the history ring: its oldest measurements are left out of the landmark refinement) is reported in
    // vslam_frame_info::error_flags only, like the stereo tracker does.
    for (int s = 0; s < B; ++s)
      if (hosts[s].error_flags & 3) {
        err = std::string("RGB-D frame: capacity exceeded (") + ((hosts[s].error_flags & 1) ? "max_keypoints " : "") + ((hosts[s].error_flags & 2) ? "max_points" : "") + ")" +
              (B > 1 ? " in sequence " + std::to_string(s) : std::string());
        return VSLAM_ERR_CAPACITY;
      }
    return VSLAM_OK;
  }
};

}  // namespace vs_rgbd
