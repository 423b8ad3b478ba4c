// rgbd_tracker.h — WHY IT SHIPS IN THE PRODUCT LIBRARY: it is the only RGB-D loop that serves detector_type ORB (the device-resident
// loop of rgbd_device.h runs the FAST detector grid only) and the independent second implementation the device loop is fuzzed against.
// RGB-D mode: PoseTracker3D with a DepthFramePointGenerator and a UVDAligner plugged in
// (slam_assembly.cpp _createDepthTracker; pose_tracker_3d.cpp:32-566; depth_framepoint_generator.cpp:24-407; uvd_aligner.cpp),
// as a HOST-DRIVEN loop: the tracker's control flow and the object bookkeeping (framepoints, links, temporary points, landmarks)
// run here in C++, every data-parallel step is one of the library's own device entry points (vslam_depth_space_map /
// _compute / _track / _recover, vslam_fast_detect, vslam_brief_describe | vslam_orb_describe, vslam_align_points_uvd,
// vslam_landmark_update, vslam_point_in_camera).  The device-resident version of this loop is csrc/rgbd_device.h + kernels_rgbd.h (the
// default behind vslam_rgbd_*); this one is its cross-check (VSLAM_RGBD_HOST=1) and serves detector_type ORB.  Detector grids of any shape (configuration_icl.yaml:57-58 runs 2 x 2; tum and
// xtion 1 x 1): one FAST detection, one threshold and one controller per region, keypoints in region-major order.
//
// Re-registration attempts of a frame ACCUMULATE its keypoints, as upstream: detectKeypoints appends to frame_->keypointsLeft(), which is
// never cleared between the initialize() calls of one frame (base_framepoint_generator.cpp:422, pose_tracker_3d.cpp:320,402), so attempts 2
// and 3 describe, store and track against the union of all attempts' keypoints, duplicates included (the lattice keeps the last feature
// written to a pixel, the feature vector keeps all: intensity_feature_matcher.cpp:48-70); cv::ORB::compute regroups a keypoint vector that
// is not sorted by pyramid level (stable, level-major).  Device loop, this loop and the checker (tests/rgbd_loop.py) agree on it.
// Reference behaviour kept (docs/rounds/ lists the citations): initialize() detects and runs the controller on
// EVERY call (also on re-registration); temporary points are not cleared between re-registrations; hasUnreliableDepth is
// inherited along a track; the aligner never sees landmarks (it asks the current point, which has none yet) and its
// translation weights live in a member vector that is never reset; a previous point linked by an earlier registration
// attempt is not reported lost by a later one.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vslam_hip.h"

namespace vs_rgbd {

struct Pt {
  float xy[2]; uint8_t desc[32]; double cam[3];
  int prev = -1, next = -1, origin = -1, track_len = 0, landmark = -1, frame = 0;
  bool unreliable = false;
};
struct Meas { int frame; double cam[3]; };
struct Lm { double world[3]; int updates = 0; std::vector<Meas> meas; };
struct Fr { double c2w[12], w2c[12]; std::vector<int> points, temps; };

inline void tf_inv(const double* T, double* o) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[4 * i + j] = T[4 * j + i];
  for (int i = 0; i < 3; ++i) o[4 * i + 3] = -((o[4 * i] * T[3] + o[4 * i + 1] * T[7]) + o[4 * i + 2] * T[11]);
}
inline void tf_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) C[4 * i + j] = (A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j]) + A[4 * i + 2] * B[8 + j];
    C[4 * i + 3] = ((A[4 * i] * B[3] + A[4 * i + 1] * B[7]) + A[4 * i + 2] * B[11]) + A[4 * i + 3];
  }
}
inline void tf_apply(const double* T, const double* p, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = ((T[4 * i] * p[0] + T[4 * i + 1] * p[1]) + T[4 * i + 2] * p[2]) + T[4 * i + 3];
}
inline void tf_id(double* T) { std::memset(T, 0, 96); T[0] = T[5] = T[10] = 1; }
inline double rot_angle(const double* T) {
  const double rx = T[9] - T[6], ry = T[2] - T[8], rz = T[4] - T[1];
  const double s = std::sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25);
  double c = ((T[0] + T[5]) + T[10] - 1) * 0.5;
  c = c > 1 ? 1 : (c < -1 ? -1 : c);
  if (s < 1e-5) return c > 0 ? 0.0 : 3.14159265358979323846;
  return std::acos(c);
}

class Tracker {
public:
  vslam_ctx* ctx = nullptr;
  vslam_config cfg;
  vslam_depth_params p;
  std::string err;
  vslam_frame_info info;
  int n_temporary = 0, threshold = 0;

  int create(const vslam_config& c, const vslam_depth_params& dp, int device) {
    cfg = c; p = dp;
    if (cfg.det_rows < 1 || cfg.det_cols < 1 || cfg.det_rows * cfg.det_cols > VSLAM_MAX_REGIONS) { err = "RGB-D mode: bad detector grid"; return VSLAM_ERR_INVALID; }
    if (p.rows != cfg.rows || p.cols != cfg.cols) { err = "RGB-D mode: depth parameters and configuration disagree on the image size"; return VSLAM_ERR_INVALID; }
    int rc = vslam_create(&cfg, device, 1, &ctx);
    if (rc != VSLAM_OK) { err = vslam_last_error(nullptr); return rc; }
    if (const char* e = std::getenv("VSLAM_RGBD_COMPACT")) compact_period = std::atoi(e);   // frames between two compactions of the point pool (tests: a small number; 0 = never)
    reset();
    return VSLAM_OK;
  }
  ~Tracker() { if (ctx) vslam_destroy(ctx); }
  void reset() {
    status = VSLAM_LOCALIZING; tf_id(prior); tf_id(world); win = cfg.maximum_projection_tracking_distance_pixels;
    tau_track = cfg.minimum_descriptor_distance_tracking;
    target = (cfg.cols / cfg.bin_size_pixels + 1) * (cfg.rows / cfg.bin_size_pixels + 1);
    // BaseFramePointGenerator::configure (base_framepoint_generator.cpp:229-312): detector regions with their overlaps, one
    // threshold per region starting at the minimum, the per-region keypoint target
    regions.clear();
    const int nv = cfg.det_rows, nh = cfg.det_cols;
    const double ph = (double)cfg.rows / nv, pw = (double)cfg.cols / nh;
    for (int r = 0; r < nv; ++r)
      for (int c = 0; c < nh; ++c) {
        int off_w = nh > 1 ? 2 : 0, off_h = nv > 1 ? 2 : 0, off_r = 0, off_c = 0;
        if (r > 0) { off_r = -off_h; if (r < nv - 1) off_h *= 2; }
        if (c > 0) { off_c = -off_w; if (c < nh - 1) off_w *= 2; }
        regions.push_back({(int)(std::round(c * pw) + off_c), (int)(std::round(r * ph) + off_r), (int)(pw + off_w), (int)(ph + off_h)});
      }
    thr.assign(regions.size(), cfg.detector_threshold_minimum);
    target_per_detector = (int)((double)target / (double)regions.size());
    pool.clear(); lms.clear(); frames.clear(); lost.clear(); weights.clear(); n_lm_prev = 0; failed = false; failed_why.clear();
    std::memset(&info, 0, sizeof info);
  }

  // PoseTracker3D::compute.  A stage that fails (capacity, HIP error) leaves the frame half registered — points linked, landmarks
  // partly updated — and there is no cheap way back: the tracker refuses further frames until reset() instead of tracking against
  // a broken history (the reference throws out of compute() and the run ends, app.cpp:128).
  int process(const uint8_t* left, int32_t lstride, const uint16_t* depth, int32_t dstride) {
    if (!left || !depth) { err = "called with empty frame"; return VSLAM_ERR_INVALID; }
    if (failed) { err = "RGB-D tracker: an earlier frame failed (" + failed_why + "); reset() before the next frame"; return VSLAM_ERR_STATE; }
    const int rc = process_frame(left, lstride, depth, dstride);
    if (rc != VSLAM_OK) { failed = true; failed_why = err; }
    return rc;
  }

private:
  bool failed = false;
  std::string failed_why;
  int compact_period = 32;
  int process_frame(const uint8_t* left, int32_t lstride, const uint16_t* depth, int32_t dstride) {
    img = left; img_stride = lstride; dep = depth; dep_stride = dstride;
    std::memset(&info, 0, sizeof info);
    info.status_at_start = status;
    const int fi = (int)frames.size();
    frames.emplace_back();
    set_pose(frames[fi], world);
    n_tracked = 0; n_tracked_lm = 0; aligner_valid = false; lost.clear(); attempts = 0;
    int rc = initialize(true);
    if (rc) return rc;
    if (fi > 0) {
      rc = track(fi, status == VSLAM_LOCALIZING);
      if (rc) return rc;
      if (status == VSLAM_LOCALIZING) {
        if (n_tracked < cfg.minimum_number_of_landmarks_to_track) fallback(fi);
        else {
          rc = align(fi, false);
          if (rc) return rc;
          if (al_inliers < cfg.minimum_number_of_landmarks_to_track) fallback(fi); else accept(fi);
        }
      } else {
        rc = register_recursive(fi, 0);
        if (rc) return rc;
      }
    }
    std::memcpy(world, frames[fi].c2w, 96);
    info.n_tracked = (int)frames[fi].points.size(); info.n_lost = (int)lost.size(); info.n_tracked_landmarks = n_tracked_lm;
    info.aligner_ran = aligner_valid ? 1 : 0;
    info.n_inliers = aligner_valid ? al_inliers : 0; info.n_outliers = aligner_valid ? (int)al_chi.size() - al_inliers : 0;
    info.aligner_iterations = aligner_valid ? al_iterations : 0; info.total_error = aligner_valid ? al_total : 0;
    if (fi > 0) {
      prune(fi);
      info.n_after_prune = (int)frames[fi].points.size();
      if (cfg.enable_landmark_recovery) { rc = recover(fi); if (rc) return rc; }
    }
    rc = update_points(fi);
    if (rc) return rc;
    if (n_active > cfg.minimum_number_of_landmarks_to_track) status = VSLAM_TRACKING;
    rc = compute(fi);
    if (rc) return rc;
    n_lm_prev = n_active;
    info.frame_index = fi + 1; info.status = status; info.n_keypoints_left = n_detected; info.n_detected_left = n_raw;
    for (size_t r = 0; r < thr.size(); ++r) info.thresholds[r] = thr[r];
    info.track_attempts = attempts; info.n_active_landmarks = n_active; info.n_points = (int)frames[fi].points.size();
    info.window_pixels = win; info.tau_track = tau_track;
    std::memcpy(info.camera_left_to_world, frames[fi].c2w, 96); std::memcpy(info.previous_to_current, prior, 96);
    n_temporary = (int)frames[fi].temps.size(); threshold = thr[0];
    if (compact_period > 0 && (fi + 1) % compact_period == 0) compact(fi);
    return VSLAM_OK;
  }

public:
  const Fr& current() const { return frames.back(); }
  const Pt& point(int id) const { return pool[id]; }
  const std::vector<Lm>& landmarks() const { return lms; }
  //! index of a point's predecessor in the previous frame's list (points followed by temporary points), -1 if none
  int previous_index(const Pt& q) const {
    if (q.prev < 0 || frames.size() < 2) return -1;
    const Fr& pf = frames[frames.size() - 2];
    for (size_t i = 0; i < pf.points.size(); ++i) if (pf.points[i] == q.prev) return (int)i;
    for (size_t i = 0; i < pf.temps.size(); ++i) if (pf.temps[i] == q.prev) return (int)(pf.points.size() + i);
    return -1;
  }

private:
  struct Region { int x, y, w, h; };
  std::vector<Region> regions; std::vector<int> thr;   // _detector_regions, the FastDetector thresholds in effect
  int target_per_detector = 0;
  int status = VSLAM_LOCALIZING, win = 0, target = 0, n_lm_prev = 0, n_tracked = 0, n_tracked_lm = 0, n_active = 0, n_detected = 0, n_raw = 0, attempts = 0;
  double tau_track = 0, prior[12], world[12];
  std::vector<Pt> pool; std::vector<Lm> lms; std::vector<Fr> frames; std::vector<int> lost;
  std::vector<double> weights;          // UVDAligner::_weights_translation: a member, resize(n, 1) keeps what it holds (uvd_aligner.cpp:22)
  bool aligner_valid = false; int al_inliers = 0, al_iterations = 0; double al_total = 0, al_T[12];
  std::vector<double> al_chi; std::vector<uint8_t> al_inl;
  // the current frame's inputs and features
  const uint8_t* img = nullptr; int32_t img_stride = 0; const uint16_t* dep = nullptr; int32_t dep_stride = 0;
  std::vector<float> fxy;        // keypoint.pt of every feature (integers for FAST, level coordinates x scale for an OrbDetector)
  std::vector<uint8_t> fdesc; std::vector<int32_t> frc, flevel; std::vector<uint8_t> matched;

  int fail(int rc, const char* where) { err = std::string(where) + ": " + vslam_last_error(ctx); return rc; }
  static void set_pose(Fr& f, const double* c2w) { std::memcpy(f.c2w, c2w, 96); tf_inv(c2w, f.w2c); }
  int new_point(const float* xy, const uint8_t* desc, const double* cam, int frame, int prev, bool unreliable) {
    Pt q;
    q.xy[0] = xy[0]; q.xy[1] = xy[1]; std::memcpy(q.desc, desc, 32);
    for (int k = 0; k < 3; ++k) q.cam[k] = cam[k];
    q.frame = frame; q.unreliable = unreliable;
    const int id = (int)pool.size();
    q.origin = id;
    if (prev >= 0) {   // FramePoint::setPrevious (frame_point.cpp:43-55)
      pool[prev].next = id; q.prev = prev; q.unreliable = pool[prev].unreliable || unreliable;
      q.track_len = pool[prev].track_len + 1; q.origin = pool[prev].origin;
    }
    pool.push_back(q);
    return id;
  }
  void clear_point(int id) {   // FramePoint::clear
    Pt& q = pool[id];
    if (q.prev >= 0) { pool[q.prev].next = -1; q.prev = -1; }
    q.landmark = -1; q.next = -1; q.track_len = 0; q.origin = id;
  }

  // Bounded memory on long sequences: every 32 frames the point pool keeps only the points of the newest frames — the previous
  // frame's lists feed the next track() and the lost list, a landmark is created from a chain of minimum_track_length + 1 points
  // (pose_tracker_3d.cpp:485-511) — and the landmarks those points still refer to.  Older points are unreachable from then on:
  // links into them become "none" (the track length stays a number), an origin that is dropped moves to the oldest kept ancestor
  // (same landmark: a track's landmark is written to every point of its chain).  Frame POSES are kept for the whole run (the
  // landmark measurements name them); their point lists are not.  Results are unchanged (tests/test_rgbd_mode.py runs across a
  // compaction).
  void compact(int fi) {
    const int keep = std::max(3, cfg.minimum_track_length_for_landmark_creation + 2);
    const int first = fi - keep + 1;
    if (first <= 0) return;
    std::vector<int> new_id(pool.size(), -1);
    std::vector<Pt> np;
    for (int f = first; f <= fi; ++f)
      for (std::vector<int>* list : {&frames[f].points, &frames[f].temps})
        for (int& id : *list) { new_id[id] = (int)np.size(); np.push_back(pool[id]); id = new_id[id]; }
    for (int& id : lost) id = id >= 0 ? new_id[id] : -1;
    lost.erase(std::remove(lost.begin(), lost.end(), -1), lost.end());
    std::vector<int> lm_id(lms.size(), -1);
    std::vector<Lm> nl;
    // (links are remapped in a second pass: np[] still holds OLD indices in prev / next / origin)
    for (size_t k = 0; k < np.size(); ++k) {
      Pt& q = np[k];
      q.prev = q.prev >= 0 ? new_id[q.prev] : -1;
      q.next = q.next >= 0 ? new_id[q.next] : -1;
    }
    for (size_t k = 0; k < np.size(); ++k) {   // origins: the dropped ones walk forward to the first kept point of the chain
      Pt& q = np[k];
      const int mapped = q.origin >= 0 ? new_id[q.origin] : -1;
      if (mapped >= 0) { q.origin = mapped; continue; }
      int a = (int)k;
      while (np[a].prev >= 0) a = np[a].prev;
      q.origin = a;
    }
    for (Pt& q : np)
      if (q.landmark >= 0) {
        if (lm_id[q.landmark] < 0) { lm_id[q.landmark] = (int)nl.size(); nl.push_back(std::move(lms[q.landmark])); }
        q.landmark = lm_id[q.landmark];
      }
    for (int f = 0; f < first; ++f) { std::vector<int>().swap(frames[f].points); std::vector<int>().swap(frames[f].temps); }
    pool.swap(np);
    lms.swap(nl);
  }

  // DepthFramePointGenerator::initialize (depth_framepoint_generator.cpp:24-44): depth map, FAST + controller over ONE image,
  // descriptors; runs in full on every call (extract_features_ is ignored upstream)
  int initialize(bool first = false) {
    int rc = vslam_depth_space_map(ctx, &p, dep, dep_stride, nullptr, nullptr, nullptr);
    if (rc) return fail(rc, "depth_space_map");
    const int cap = 65535;
    auto controller = [&](size_t r, int nr) {
      double t = (double)thr[r];
      const double delta = ((double)nr - target_per_detector) / target_per_detector;
      if (delta < -cfg.target_number_of_keypoints_tolerance) {
        t += std::min(std::max(delta, -cfg.detector_threshold_maximum_change) * t, -1.0); t = std::max(t, (double)cfg.detector_threshold_minimum);
      } else if (delta > cfg.target_number_of_keypoints_tolerance) {
        t += std::max(std::min(delta, cfg.detector_threshold_maximum_change) * t, 1.0); t = std::min(t, (double)cfg.detector_threshold_maximum);
      }
      thr[r] = (int)std::rint(t / 1);
    };
    if (first) { fxy.clear(); fdesc.clear(); frc.clear(); flevel.clear(); }     // a new Frame: empty keypointsLeft(); otherwise this detection is appended
    if (p.detector_type == VSLAM_DETECTOR_ORB) {
      // OrbDetector (base_framepoint_generator.cpp:52-70): cv::ORB::create(5000, 1.2f, 8, 31, 0, 2, HARRIS_SCORE, 31, threshold)->detect on the
      // region's VIEW of the image (its pyramid is the region's), keypoint.pt += region corner (float), lists concatenated; then the configured
      // extractor on the whole image: ORB::compute (the keypoint's pyramid level and angle) or BRIEF (level 0, pixel (int)(pt + 0.5))
      std::vector<float> kps((size_t)cap * 6);
      int32_t n = 0;
      for (size_t r = 0; r < regions.size(); ++r) {
        const Region& R = regions[r];
        int32_t nr = 0;
        rc = vslam_orb_detect(ctx, img + (size_t)R.y * img_stride + R.x, R.h, R.w, img_stride, 5000, 1.2f, 8, 31, 31, thr[r], cap - n, &nr, kps.data() + 6 * (size_t)n);
        if (rc) return fail(rc, "orb_detect");
        for (int i = n; i < n + nr; ++i) { kps[6 * (size_t)i] = kps[6 * (size_t)i] + (float)R.x; kps[6 * (size_t)i + 1] = kps[6 * (size_t)i + 1] + (float)R.y; }
        controller(r, nr);
        n += nr;
      }
      n_raw = n;
      std::vector<uint8_t> keep(std::max(n, 1)), desc((size_t)std::max(n, 1) * 32);
      if (p.descriptor_type == VSLAM_DESCRIPTOR_ORB) {
        rc = vslam_orb_describe_keypoints(ctx, img, cfg.rows, cfg.cols, img_stride, n, kps.data(), 1.2f, keep.data(), desc.data());
      } else {
        std::vector<int16_t> px((size_t)std::max(n, 1) * 2);
        for (int i = 0; i < n; ++i) { px[2 * i] = (int16_t)(int)(kps[6 * (size_t)i] + 0.5f); px[2 * i + 1] = (int16_t)(int)(kps[6 * (size_t)i + 1] + 0.5f); }
        rc = vslam_brief_describe(ctx, img, cfg.rows, cfg.cols, img_stride, n, px.data(), keep.data(), desc.data());
      }
      if (rc) return fail(rc, "describe");
      for (int i = 0; i < n; ++i) {
        if (!keep[i]) continue;
        const float x = kps[6 * (size_t)i], y = kps[6 * (size_t)i + 1];
        fxy.push_back(x); fxy.push_back(y);
        frc.push_back((int32_t)y); frc.push_back((int32_t)x);          // IntensityFeature: row = (int)pt.y, col = (int)pt.x
        fdesc.insert(fdesc.end(), desc.begin() + (size_t)32 * i, desc.begin() + (size_t)32 * i + 32);
        flevel.push_back((int32_t)kps[6 * (size_t)i + 5]);
      }
    } else {
    std::vector<int16_t> xy((size_t)cap * 2); std::vector<int32_t> score(cap);
    int32_t n = 0;
    // detectKeypoints (base_framepoint_generator.cpp:355-429): region by region (row-major over the grid) with the region's own
    // threshold, its controller against the per-region target, coordinates shifted by the region's corner, lists concatenated;
    // adjustDetectorThresholds (:440-459) over ONE detection per frame: the new threshold is rint of the controller's value
    for (size_t r = 0; r < regions.size(); ++r) {
      const Region& R = regions[r];
      int32_t nr = 0;
      rc = vslam_fast_detect(ctx, img, cfg.rows, cfg.cols, img_stride, R.x, R.y, R.w, R.h, thr[r], cap - n, &nr, xy.data() + 2 * (size_t)n, score.data() + n);
      if (rc) return fail(rc, "fast_detect");
      for (int i = n; i < n + nr; ++i) { xy[2 * i] = (int16_t)(xy[2 * i] + R.x); xy[2 * i + 1] = (int16_t)(xy[2 * i + 1] + R.y); }
      controller(r, nr);
      n += nr;
    }
    n_raw = n;
    std::vector<uint8_t> keep(std::max(n, 1)), desc((size_t)std::max(n, 1) * 32);
    rc = p.descriptor_type == VSLAM_DESCRIPTOR_ORB ? vslam_orb_describe(ctx, img, cfg.rows, cfg.cols, img_stride, n, xy.data(), -1.f, keep.data(), desc.data())
                                                   : vslam_brief_describe(ctx, img, cfg.rows, cfg.cols, img_stride, n, xy.data(), keep.data(), desc.data());
    if (rc) return fail(rc, "describe");
    for (int i = 0; i < n; ++i) {
      if (!keep[i]) continue;
      fxy.push_back((float)xy[2 * i]); fxy.push_back((float)xy[2 * i + 1]);
      frc.push_back(xy[2 * i + 1]); frc.push_back(xy[2 * i]);
      fdesc.insert(fdesc.end(), desc.begin() + (size_t)32 * i, desc.begin() + (size_t)32 * i + 32);
      flevel.push_back(0);
    }
    }
    // ORB::compute on the frame's whole keypoint vector: earlier attempts' keypoints get the same descriptors again; a vector that is not
    // sorted by level is regrouped level-major, stable (only an OrbDetector's accumulated list is not)
    if (p.descriptor_type == VSLAM_DESCRIPTOR_ORB && !std::is_sorted(flevel.begin(), flevel.end())) {
      const size_t n = flevel.size();
      std::vector<int> o(n);
      for (size_t i = 0; i < n; ++i) o[i] = (int)i;
      std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return flevel[a] < flevel[b]; });
      std::vector<float> xy2(2 * n); std::vector<int32_t> rc2(2 * n), lv2(n); std::vector<uint8_t> ds2(32 * n);
      for (size_t i = 0; i < n; ++i) {
        xy2[2 * i] = fxy[2 * o[i]]; xy2[2 * i + 1] = fxy[2 * o[i] + 1]; rc2[2 * i] = frc[2 * o[i]]; rc2[2 * i + 1] = frc[2 * o[i] + 1]; lv2[i] = flevel[o[i]];
        std::memcpy(&ds2[32 * i], &fdesc[(size_t)32 * o[i]], 32);
      }
      fxy.swap(xy2); frc.swap(rc2); flevel.swap(lv2); fdesc.swap(ds2);
    }
    n_detected = (int)fxy.size() / 2;
    matched.assign(n_detected, 0);
    return VSLAM_OK;
  }

  // PoseTracker3D::_track (:225-298) around DepthFramePointGenerator::track (:166-287)
  int track(int fi, bool by_appearance) {
    if (by_appearance) win = cfg.maximum_projection_tracking_distance_pixels;
    Fr& cur = frames[fi]; const Fr& prev = frames[fi - 1];
    std::vector<int> prevlist(prev.points);
    prevlist.insert(prevlist.end(), prev.temps.begin(), prev.temps.end());
    const int nP = (int)prevlist.size();
    std::vector<double> cam((size_t)std::max(nP, 1) * 3); std::vector<uint8_t> pd((size_t)std::max(nP, 1) * 32), fl(std::max(nP, 1));
    for (int i = 0; i < nP; ++i) {
      const Pt& q = pool[prevlist[i]];
      for (int k = 0; k < 3; ++k) cam[3 * i + k] = q.cam[k];
      std::memcpy(&pd[(size_t)32 * i], q.desc, 32);
      fl[i] = (uint8_t)((q.landmark >= 0 ? 1 : 0) | (q.unreliable ? 2 : 0));
    }
    std::vector<int32_t> out2((size_t)std::max(nP, 1) * 2), tmp2((size_t)std::max(nP, 1) * 2), ls(std::max(nP, 1));
    std::vector<double> xyz((size_t)std::max(nP, 1) * 3);
    int32_t nt = 0, ntmp = 0, nl = 0, nlm = 0;
    int rc = vslam_depth_track(ctx, &p, nullptr, prior, win, cfg.minimum_descriptor_distance_tracking, by_appearance ? 1 : 0, nP, cam.data(), pd.data(),
                               fl.data(), n_detected, frc.data(), fdesc.data(), &nt, out2.data(), xyz.data(), &ntmp, tmp2.data(), &nl, ls.data(), &nlm);
    if (rc) return fail(rc, "depth_track");
    cur.points.clear();
    std::fill(matched.begin(), matched.end(), 0);
    for (int u = 0; u < nt; ++u) {
      const int f = out2[2 * u + 1];
      const float fx[2] = {fxy[2 * f], fxy[2 * f + 1]};
      cur.points.push_back(new_point(fx, &fdesc[(size_t)32 * f], &xyz[3 * u], fi, prevlist[out2[2 * u]], false));
      matched[f] = 1;
    }
    const double zero[3] = {0, 0, 0};
    for (int u = 0; u < ntmp; ++u) {   // matches on pixels without depth: temporary points (:247-256); never cleared between attempts
      const int f = tmp2[2 * u + 1];
      const float fx[2] = {fxy[2 * f], fxy[2 * f + 1]};
      cur.temps.push_back(new_point(fx, &fdesc[(size_t)32 * f], zero, fi, prevlist[tmp2[2 * u]], true));
      matched[f] = 1;
    }
    lost.clear();
    for (int u = 0; u < nl; ++u) if (pool[prevlist[ls[u]]].next < 0) lost.push_back(prevlist[ls[u]]);   // next() survives earlier attempts
    n_tracked_lm = nlm; n_tracked = nt;
    const double ratio = (double)n_tracked / (double)prev.points.size();
    const double lm_per_point = (double)n_tracked_lm / (double)n_tracked, success = (double)n_tracked / target;
    const int wmax = cfg.maximum_projection_tracking_distance_pixels, wmin = cfg.minimum_projection_tracking_distance_pixels;
    if (ratio < cfg.good_tracking_ratio / 2) { if (win < wmax) win = (int)std::min(win * 1 / cfg.tunnel_vision_ratio, (double)wmax); }
    else if (win > wmin) win = (int)std::max(win * cfg.tunnel_vision_ratio, (double)wmin);
    if (ratio < cfg.good_tracking_ratio || n_tracked < cfg.aligner_minimum_number_of_inliers || (lm_per_point < 0.5 && success < 0.25))
      tau_track = std::min(tau_track + 5, cfg.maximum_descriptor_distance_tracking);
    else tau_track = std::max(tau_track - 5, cfg.minimum_descriptor_distance_tracking);
    aligner_valid = false;
    ++attempts;
    return VSLAM_OK;
  }

  // UVDAligner::initialize (uvd_aligner.cpp:11-69) + converge
  int align(int fi, bool inverse_depth) {
    const Fr& cur = frames[fi];
    const int n = (int)cur.points.size();
    weights.resize(n, 1.0);
    std::vector<double> moving((size_t)std::max(n, 1) * 3), fixed((size_t)std::max(n, 1) * 3), wuv(std::max(n, 1), 1.0), wd(std::max(n, 1), 10.0);
    for (int u = 0; u < n; ++u) {
      const Pt& q = pool[cur.points[u]];
      fixed[3 * u] = (double)q.xy[0]; fixed[3 * u + 1] = (double)q.xy[1]; fixed[3 * u + 2] = q.cam[2];
      for (int k = 0; k < 3; ++k) moving[3 * u + k] = pool[q.prev].cam[k];   // the current point has no landmark yet (:38): always the previous point
      if (q.unreliable) { weights[u] = 0; wd[u] = 0; }
      else if (inverse_depth) weights[u] = cfg.maximum_reliable_depth_meters / q.cam[2];
    }
    al_chi.assign(std::max(n, 1), 0); al_inl.assign(std::max(n, 1), 0);
    double H[36];
    int rc = vslam_align_points_uvd(ctx, n, moving.data(), fixed.data(), wuv.data(), wd.data(), weights.data(), prior, al_T, al_chi.data(), al_inl.data(),
                                    &al_inliers, &al_total, &al_iterations, H);
    if (rc) return fail(rc, "align_points_uvd");
    al_chi.resize(n); al_inl.resize(n);
    aligner_valid = true;
    return VSLAM_OK;
  }
  void accept(int fi) {
    const double dt = std::sqrt((al_T[3] * al_T[3] + al_T[7] * al_T[7]) + al_T[11] * al_T[11]);
    if (rot_angle(al_T) > cfg.minimum_delta_angular_for_movement || dt > cfg.minimum_delta_translational_for_movement) {
      std::memcpy(prior, al_T, 96);
      double inv[12], c2w[12];
      tf_inv(prior, inv); tf_mul(frames[fi - 1].c2w, inv, c2w);
      set_pose(frames[fi], c2w);
    } else fallback(fi);
  }
  void fallback(int fi) { tf_id(prior); set_pose(frames[fi], frames[fi - 1].c2w); info.fallback = 1; }
  void break_track(int fi) { status = VSLAM_LOCALIZING; set_pose(frames[fi], frames[fi - 1].c2w); tf_id(prior); n_tracked = 0; info.track_broken = 1; }
  int register_recursive(int fi, int recursion) {
    const double rel = (double)n_tracked_lm / (double)n_lm_prev;
    if (n_tracked_lm == 0 || rel < 0.1) {
      if (recursion < 2) {
        tf_id(prior);
        int rc = initialize(); if (rc) return rc;
        rc = track(fi, true); if (rc) return rc;
        return register_recursive(fi, recursion + 1);
      }
      break_track(fi);
      return VSLAM_OK;
    }
    int rc = align(fi, true);
    if (rc) return rc;
    if (al_inliers > cfg.minimum_number_of_landmarks_to_track) { accept(fi); return VSLAM_OK; }
    if (recursion < 2) {
      if (win < cfg.maximum_projection_tracking_distance_pixels) ++win;
      rc = initialize(); if (rc) return rc;
      rc = track(fi, false); if (rc) return rc;
      return register_recursive(fi, recursion + 1);
    }
    break_track(fi);
    return VSLAM_OK;
  }

  // _prunePoints (:437-472); without a fresh aligner result every tracked point is dropped
  void prune(int fi) {
    Fr& cur = frames[fi];
    std::vector<int> kept;
    const int n = (int)cur.points.size();
    const double avg = aligner_valid ? al_total / (double)n : 0;
    for (int u = 0; u < n; ++u) {
      bool keep = false;
      if (aligner_valid) keep = avg < cfg.aligner_maximum_error_kernel ? al_inl[u] != 0 : (al_chi[u] != -1 && al_chi[u] < 100 * cfg.aligner_maximum_error_kernel);
      if (keep) kept.push_back(cur.points[u]); else clear_point(cur.points[u]);
    }
    cur.points.swap(kept);
  }

  // DepthFramePointGenerator::recoverPoints (:289-407)
  int recover(int fi) {
    Fr& cur = frames[fi];
    const int n = (int)lost.size();
    if (!n) return VSLAM_OK;
    std::vector<uint8_t> has(n), pd((size_t)n * 32), rdesc((size_t)n * 32);
    std::vector<double> lw((size_t)n * 3, 0.0), rxyz((size_t)n * 3);
    std::vector<int32_t> ridx(n); std::vector<float> rxy((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
      const Pt& q = pool[lost[i]];
      has[i] = q.landmark >= 0 ? 1 : 0;
      if (q.landmark >= 0) for (int k = 0; k < 3; ++k) lw[3 * i + k] = lms[q.landmark].world[k];
      std::memcpy(&pd[(size_t)32 * i], q.desc, 32);
    }
    int32_t nr = 0;
    int rc = vslam_depth_recover(ctx, &p, nullptr, img, img_stride, cur.w2c, n, has.data(), lw.data(), pd.data(), 7.f, cfg.minimum_descriptor_distance_tracking,
                                 &nr, ridx.data(), rxy.data(), rdesc.data(), rxyz.data());
    if (rc) return fail(rc, "depth_recover");
    for (int k = 0; k < nr; ++k) cur.points.push_back(new_point(&rxy[2 * k], &rdesc[(size_t)32 * k], &rxyz[3 * k], fi, lost[ridx[k]], false));
    info.n_recovered = nr;
    return VSLAM_OK;
  }

  // PoseTracker3D::_updatePoints (:475-548): landmark creation (Landmark::Landmark, landmark.cpp:8-33) on the host, refinement
  // (Landmark::update, :66-167) batched through vslam_landmark_update; temporary points triangulated with the refined motion
  int update_points(int fi) {
    Fr& cur = frames[fi];
    std::vector<std::pair<int, int>> todo;   // (landmark, point)
    n_active = 0;
    for (int id : cur.points) {
      Pt& q = pool[id];
      if (q.track_len < cfg.minimum_track_length_for_landmark_creation || q.unreliable) continue;
      int lm = pool[q.origin].landmark;
      if (lm < 0) {
        lm = (int)lms.size();
        lms.emplace_back();
        Lm& L = lms.back();
        double acc[3] = {0, 0, 0};
        int len = 0;
        for (int t = id; t >= 0; t = pool[t].prev) {   // newest first, as the constructor walks the chain
          pool[t].landmark = lm;
          Meas m; m.frame = pool[t].frame; for (int k = 0; k < 3; ++k) m.cam[k] = pool[t].cam[k];
          L.meas.push_back(m);
          double w[3];
          tf_apply(frames[pool[t].frame].c2w, pool[t].cam, w);
          for (int k = 0; k < 3; ++k) acc[k] = acc[k] + w[k];
          ++len;
        }
        for (int k = 0; k < 3; ++k) L.world[k] = acc[k] / len;
        L.updates = len;
      } else {
        todo.emplace_back(lm, id);
      }
      ++n_active;
    }
    if (!todo.empty()) {
      std::vector<int> used;
      for (auto& lq : todo) for (const Meas& m : lms[lq.first].meas) used.push_back(m.frame);
      used.push_back(fi);
      std::sort(used.begin(), used.end()); used.erase(std::unique(used.begin(), used.end()), used.end());
      auto remap = [&used](int frame) { return (int)(std::lower_bound(used.begin(), used.end(), frame) - used.begin()); };   // `used` is sorted, every frame asked for is in it
      std::vector<double> w2c(used.size() * 12), c2w(used.size() * 12);
      for (size_t i = 0; i < used.size(); ++i) { std::memcpy(&w2c[12 * i], frames[used[i]].w2c, 96); std::memcpy(&c2w[12 * i], frames[used[i]].c2w, 96); }
      std::vector<int32_t> off(1, 0), frame_of, upd; std::vector<double> cams, wld;
      for (auto& lq : todo) {
        const Lm& L = lms[lq.first];
        for (const Meas& m : L.meas) { frame_of.push_back(remap(m.frame)); cams.insert(cams.end(), m.cam, m.cam + 3); }
        frame_of.push_back(remap(fi)); cams.insert(cams.end(), pool[lq.second].cam, pool[lq.second].cam + 3);
        off.push_back((int32_t)frame_of.size());
        wld.insert(wld.end(), L.world, L.world + 3); upd.push_back(L.updates);
      }
      int rc = vslam_landmark_update(ctx, (int32_t)todo.size(), off.data(), frame_of.data(), (int32_t)used.size(), w2c.data(), c2w.data(), cams.data(), wld.data(), upd.data());
      if (rc) return fail(rc, "landmark_update");
      for (size_t k = 0; k < todo.size(); ++k) {
        Lm& L = lms[todo[k].first];
        for (int q = 0; q < 3; ++q) L.world[q] = wld[3 * k + q];
        L.updates = upd[k];
        Meas m; m.frame = fi; for (int q = 0; q < 3; ++q) m.cam[q] = pool[todo[k].second].cam[q];
        L.meas.push_back(m);
        pool[todo[k].second].landmark = todo[k].first;
      }
    }
    if (!cur.temps.empty()) {
      const int n = (int)cur.temps.size();
      std::vector<float> xp((size_t)n * 2), xc((size_t)n * 2); std::vector<double> tri((size_t)n * 3);
      for (int i = 0; i < n; ++i) {
        const Pt& q = pool[cur.temps[i]];
        xp[2 * i] = pool[q.prev].xy[0]; xp[2 * i + 1] = pool[q.prev].xy[1]; xc[2 * i] = q.xy[0]; xc[2 * i + 1] = q.xy[1];
      }
      int rc = vslam_point_in_camera(ctx, n, xp.data(), xc.data(), prior, cfg.K, tri.data());
      if (rc) return fail(rc, "point_in_camera");
      std::vector<int> kept;
      for (int i = 0; i < n; ++i) {
        if (tri[3 * i + 2] <= 0) continue;
        for (int k = 0; k < 3; ++k) pool[cur.temps[i]].cam[k] = tri[3 * i + k];
        kept.push_back(cur.temps[i]);
      }
      cur.temps.swap(kept);
    }
    return VSLAM_OK;
  }

  // DepthFramePointGenerator::compute (:45-164) on the features track() left unmatched
  int compute(int fi) {
    Fr& cur = frames[fi];
    std::vector<int> rem;
    for (int f = 0; f < n_detected; ++f) if (!matched[f]) rem.push_back(f);
    const int nF = (int)rem.size(), nT = (int)cur.points.size();
    std::vector<int32_t> rc((size_t)std::max(nF, 1) * 2), trc((size_t)std::max(nT, 1) * 2);
    for (int i = 0; i < nF; ++i) { rc[2 * i] = frc[2 * rem[i]]; rc[2 * i + 1] = frc[2 * rem[i] + 1]; }
    for (int i = 0; i < nT; ++i) { const Pt& q = pool[cur.points[i]]; trc[2 * i] = (int32_t)q.xy[1]; trc[2 * i + 1] = (int32_t)q.xy[0]; }
    const int cap = std::max(nF, 1);
    std::vector<int32_t> nf(cap), tf_(cap); std::vector<double> nx((size_t)cap * 3), tx((size_t)cap * 3);
    int32_t nn = 0, nt = 0;
    int r = vslam_depth_compute(ctx, &p, nullptr, nF, rc.data(), nT, trc.data(), cap, &nn, nf.data(), nx.data(), &nt, tf_.data(), tx.data());
    if (r) return fail(r, "depth_compute");
    for (int k = 0; k < nn; ++k) {
      const int g = rem[nf[k]];
      const float fx[2] = {fxy[2 * g], fxy[2 * g + 1]};
      cur.points.push_back(new_point(fx, &fdesc[(size_t)32 * g], &nx[3 * k], fi, -1, false));
    }
    for (int k = 0; k < nt; ++k) {
      const int g = rem[tf_[k]];
      const float fx[2] = {fxy[2 * g], fxy[2 * g + 1]};
      cur.temps.push_back(new_point(fx, &fdesc[(size_t)32 * g], &tx[3 * k], fi, -1, true));
    }
    info.n_new_stereo = nn;
    return VSLAM_OK;
  }
};

}  // namespace vs_rgbd
