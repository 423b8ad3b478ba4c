// proslam_hip_plugin.h — header-only C++ shim: the reference's plug-in classes on top of libvslam_hip.so.
//
// Meant to be compiled inside the reference tree (its headers: OpenCV 3, Eigen, srrg_core); INTEGRATION.md shows the two
// lines of SLAMAssembly::_createStereoTracker (src/system/slam_assembly.cpp:61-76) a maintainer changes.  Those
// dependencies are absent from this repository's image, so here the header is compile- and run-checked against minimal
// declaration stubs of the interfaces it touches (tests/shim_stubs/, test-only): a signature drift against
// base_framepoint_generator.h:119-145 / base_aligner.h:26-48 / frame.h fails the CPU test suite.
//
// The classes derive from the CONCRETE reference classes because SLAMAssembly down-casts the generator
// (slam_assembly.h:101, slam_assembly.cpp:690-691,717-719) and PoseTracker3D keeps its own control flow
// (pose_tracker_3d.cpp:32-566): every virtual below is one or two C calls; results are materialised into the host
// objects the rest of the reference reads (Frame::keypoints/descriptors, FramePoint via Frame::createFramepoint —
// its constructor is protected, types/frame_point.h:45-58 — and BaseAligner's protected result members).
//
// Host and device run the same frame side by side: the device keeps its own framepoints / landmarks (they feed its
// tracker and aligner), the host keeps the reference's objects (they feed the map, relocalization, viewers).  Both
// apply the same rules, so they stay index-aligned: Frame::points()[i] on the host is point i of the device's frame;
// every materialisation step below checks the counts and throws on divergence.  Open loop only (loop closing rewrites
// host poses the device does not see).
#pragma once
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "aligners/stereouv_aligner.h"
#include "framepoint_generation/stereo_framepoint_generator.h"
#include "vslam_hip.h"

namespace proslam {

// host-time breakdown of the plug-in calls (tests/cpp/bench_shim.cpp builds with -DPROSLAM_HIP_PROFILE); compiled out otherwise
#ifdef PROSLAM_HIP_PROFILE
#include <chrono>
struct HipProfile {
  enum { BEGIN_CALL, KEYPOINTS_WAIT, KEYPOINTS_HOST, TRACK_CALL, TRACK_WAIT, TRACK_HOST, ALIGN_CALL, ALIGN_WAIT, ALIGN_HOST, PRUNE_CALL, PRUNE_WAIT, PRUNE_HOST,
         COMPUTE_CALL, COMPUTE_WAIT, COMPUTE_HOST, CHRONO, N };
  static double* acc() { static double a[N] = {0}; return a; }
  static const char* name(int i) {
    static const char* n[N] = {"begin_call", "keypoints_wait", "keypoints_host", "track_call", "track_wait", "track_host", "align_call", "align_wait", "align_host",
                               "prune_call", "prune_wait", "prune_host", "compute_call", "compute_wait", "compute_host", "chronometers"};
    return n[i];
  }
  int k; std::chrono::steady_clock::time_point t0;
  explicit HipProfile(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {}
  ~HipProfile() { acc()[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
#define HIP_PROFILE(K) HipProfile hip_profile_##K(HipProfile::K)
#else
#define HIP_PROFILE(K) do {} while (0)
#endif

inline void hipCheck(vslam_ctx* ctx, int rc, const char* where) {
  if (rc != VSLAM_OK) throw std::runtime_error(std::string(where) + "|" + vslam_last_error(ctx));
}

class HipStereoUVAligner;

//! one device context shared by the generator and the aligner of a tracker (n_streams = 1)
struct HipContext {
  vslam_ctx* ctx = nullptr;
  vslam_config config;
  int device = 0;
  //! parameters the generator / aligner objects do not own (set them before configure(); null: configuration_kitti.yaml values)
  const PoseTracker3DParameters* tracker_parameters = nullptr;   // tracking: (parameters.h:262-300)
  const LandmarkParameters* landmark_parameters = nullptr;       // world_map: landmark (parameters.h:97-112)
  HipStereoUVAligner* aligner = nullptr;
  //! the 256 test pairs of the descriptor extractors (null: the library's own tables).  An OpenCV build has them in
  //! xfeatures2d/src/generated_32.i (BRIEF, as {y1, x1, y2, x2}) and features2d/src/orb.cpp bit_pattern_31_ (ORB, {x1, y1, x2, y2});
  //! passing them makes Frame::descriptorsLeft/Right bit-compatible with the reference's own extractors (include/vslam_hip.h)
  const int8_t* brief_pattern = nullptr;
  const int8_t* orb_pattern = nullptr;
  //! tracker-owned state last pushed to the device (Frame::status, window, descriptor distance)
  int status = VSLAM_LOCALIZING;
  int32_t window_pixels = 0;
  double tau_track = 0;
  HipContext() { vslam_default_config_kitti(&config); }
  ~HipContext() { if (ctx) vslam_destroy(ctx); }
  HipContext(const HipContext&) = delete;
  HipContext& operator=(const HipContext&) = delete;
};

template <typename Transform>
inline void hipToArray(const Transform& T, double* out) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = T.matrix()(i, j);
}

class HipStereoUVAligner : public StereoUVAligner {
public:
  HipStereoUVAligner(AlignerParameters* parameters_, HipContext* hip_) : StereoUVAligner(parameters_), _hip(hip_) { _hip->aligner = this; }

  //! creates the device context once every parameter is known (call after the generator's configure)
  void configure() override {
    vslam_config& c = _hip->config;
    c.aligner_error_delta_for_convergence = _parameters->error_delta_for_convergence;
    c.aligner_maximum_error_kernel = _parameters->maximum_error_kernel;
    c.aligner_damping = _parameters->damping;
    c.aligner_maximum_number_of_iterations = (int32_t)_parameters->maximum_number_of_iterations;
    c.aligner_minimum_number_of_inliers = (int32_t)_parameters->minimum_number_of_inliers;
    c.minimum_depth_meters = _minimum_reliable_depth_meters;         // setMinimumReliableDepthMeters (slam_assembly.cpp:70)
    c.maximum_reliable_depth_meters = _maximum_reliable_depth_meters;
    if (const PoseTracker3DParameters* t = _hip->tracker_parameters) {
      c.minimum_track_length_for_landmark_creation = (int32_t)t->minimum_track_length_for_landmark_creation;
      c.minimum_number_of_landmarks_to_track = (int32_t)t->minimum_number_of_landmarks_to_track;
      c.tunnel_vision_ratio = t->tunnel_vision_ratio;
      c.good_tracking_ratio = t->good_tracking_ratio;
      c.enable_landmark_recovery = t->enable_landmark_recovery ? 1 : 0;
      c.minimum_delta_angular_for_movement = t->minimum_delta_angular_for_movement;
      c.minimum_delta_translational_for_movement = t->minimum_delta_translational_for_movement;
    }
    if (const LandmarkParameters* l = _hip->landmark_parameters) {
      c.landmark_maximum_error_squared_meters = l->maximum_error_squared_meters;
      c.landmark_maximum_number_of_iterations = (int32_t)l->maximum_number_of_iterations;
    }
    if (!_hip->ctx) {
      if (_hip->brief_pattern) hipCheck(nullptr, vslam_set_brief_pattern(_hip->device, _hip->brief_pattern), "HipStereoUVAligner::configure|brief pattern");
      if (_hip->orb_pattern) hipCheck(nullptr, vslam_set_orb_pattern(_hip->device, _hip->orb_pattern), "HipStereoUVAligner::configure|orb pattern");
      hipCheck(nullptr, vslam_create(&c, _hip->device, 1, &_hip->ctx), "HipStereoUVAligner::configure");
    }
  }

  //! StereoUVAligner::initialize (stereouv_aligner.cpp:10-69): the correspondences are already on the device (the list
  //! the last track() left); the motion prior travels with the tracker state
  void initialize(const Frame* frame_previous_, const Frame* frame_current_, const TransformMatrix3D& previous_to_current_) override {
    _frame_previous = frame_previous_; _frame_current = frame_current_; _previous_to_current = previous_to_current_;
    _number_of_measurements = (Count)_frame_current->points().size();
    double prior[12];
    hipToArray(previous_to_current_, prior);
    hipCheck(_hip->ctx, vslam_set_tracker_state(_hip->ctx, 0, _hip->status, prior, _hip->window_pixels, _hip->tau_track), "HipStereoUVAligner::initialize");
  }
  void linearize(const bool&) override {}   // folded into converge() on the device
  void oneRound(const bool&) override {}

  //! StereoUVAligner::converge (:210-264): one launch; results into the base-class members the tracker reads
  //! (errors(), inliers(), numberOfInliers(), totalError(), previousToCurrent(): base_aligner.h:37-48)
  void converge() override {
    { HIP_PROFILE(ALIGN_CALL); hipCheck(_hip->ctx, vslam_align(_hip->ctx, _parameters->enable_inverse_depth_as_information ? 1 : 0), "HipStereoUVAligner::converge"); }
    vslam_aligner_view v;      // one packed read-back, one synchronisation (include/vslam_hip.h: stage views)
    { HIP_PROFILE(ALIGN_WAIT); hipCheck(_hip->ctx, vslam_view_aligner(_hip->ctx, 0, &v), "HipStereoUVAligner::converge"); }
    HIP_PROFILE(ALIGN_HOST);
    const int32_t n = v.n;
    if (n != (int32_t)_number_of_measurements) throw std::runtime_error("HipStereoUVAligner::converge|host and device disagree on the number of measurements");
    _errors.assign(v.chi, v.chi + n);
    _inliers.resize(n);
    for (int32_t u = 0; u < n; ++u) _inliers[u] = v.inlier[u] != 0;
    _number_of_inliers = (Count)v.n_inliers; _number_of_outliers = (Count)v.n_outliers; _total_error = v.total_error;
    _has_system_converged = v.converged != 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) _previous_to_current.matrix()(i, j) = v.T[4 * i + j];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { _H(i, j) = v.H[6 * i + j]; _information_matrix(i, j) = v.H[6 * i + j]; }
  }

  //! a new track() invalidates the previous converge(): PoseTracker3D::_prunePoints would otherwise read the stale
  //! errors()/inliers() of another point list (pose_tracker_3d.cpp:441-470).  Defined behaviour, same as the device
  //!: without a fresh aligner result every tracked point is dropped — all outliers, zero error.
  void invalidate(const Count& number_of_points_) {
    _number_of_measurements = number_of_points_;
    _errors.assign(number_of_points_, -1.0);
    _inliers.assign(number_of_points_, false);
    _number_of_inliers = 0; _number_of_outliers = number_of_points_; _total_error = 0; _has_system_converged = false;
  }

private:
  HipContext* _hip;
};

class HipStereoFramePointGenerator : public StereoFramePointGenerator {
public:
  HipStereoFramePointGenerator(StereoFramePointGeneratorParameters* parameters_, HipContext* hip_)
      : StereoFramePointGenerator(parameters_), _hip(hip_) {}

  //! BaseFramePointGenerator::configure + StereoFramePointGenerator::configure: parameters -> vslam_config
  void configure() override {
    StereoFramePointGenerator::configure();  // keeps the inherited members (bins, targets, chronometers) valid
    vslam_config& c = _hip->config;
    c.rows = _number_of_rows_image; c.cols = _number_of_cols_image;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.K[3 * i + j] = _camera_left->cameraMatrix()(i, j);
    for (int i = 0; i < 3; ++i) c.baseline_h[i] = _camera_right->baselineHomogeneous()(i);
    StereoFramePointGeneratorParameters* p = parameters();
    c.det_rows = (int32_t)p->number_of_detectors_vertical; c.det_cols = (int32_t)p->number_of_detectors_horizontal;
    c.detector_threshold_minimum = (int32_t)p->detector_threshold_minimum; c.detector_threshold_maximum = (int32_t)p->detector_threshold_maximum;
    c.detector_threshold_maximum_change = p->detector_threshold_maximum_change;
    c.target_number_of_keypoints_tolerance = p->target_number_of_keypoints_tolerance;
    c.bin_size_pixels = (int32_t)p->bin_size_pixels; c.enable_keypoint_binning = p->enable_keypoint_binning ? 1 : 0;
    c.minimum_projection_tracking_distance_pixels = p->minimum_projection_tracking_distance_pixels;
    c.maximum_projection_tracking_distance_pixels = p->maximum_projection_tracking_distance_pixels;
    c.minimum_descriptor_distance_tracking = p->minimum_descriptor_distance_tracking;
    c.maximum_descriptor_distance_tracking = p->maximum_descriptor_distance_tracking;
    c.maximum_reliable_depth_meters = p->maximum_reliable_depth_meters; c.maximum_depth_meters = p->maximum_depth_meters;
    c.minimum_depth_meters = p->minimum_depth_meters;
    c.maximum_matching_distance_triangulation = p->maximum_matching_distance_triangulation;
    c.minimum_disparity_pixels = p->minimum_disparity_pixels;
    c.maximum_epipolar_search_offset_pixels = p->maximum_epipolar_search_offset_pixels;
    // descriptor extractor as BaseFramePointGenerator::configure selects it (base_framepoint_generator.cpp:184-224): BRIEF only
    // by name (and only in a build with opencv_contrib); "ORB" and every string the parser does not know end at cv::ORB::create()
    c.descriptor_type = p->descriptor_type == "BRIEF" ? VSLAM_DESCRIPTOR_BRIEF : VSLAM_DESCRIPTOR_ORB;
    _hip->window_pixels = p->maximum_projection_tracking_distance_pixels;    // PoseTracker3D::configure (pose_tracker_3d.cpp:11-21)
    _hip->tau_track = p->minimum_descriptor_distance_tracking;
    // tracker / aligner / landmark values are filled by HipStereoUVAligner::configure, which creates the device context
  }

  //! StereoFramePointGenerator::initialize (stereo_framepoint_generator.cpp:73-133)
  void initialize(Frame* frame_, const bool& extract_features_ = true) override {
    if (!frame_) throw std::runtime_error("HipStereoFramePointGenerator::initialize|called with empty frame");
    if (!_hip->ctx) throw std::runtime_error("HipStereoFramePointGenerator::initialize|no device context: configure the aligner first");
    if (!extract_features_) { hipCheck(_hip->ctx, vslam_frame_restore(_hip->ctx), "HipStereoFramePointGenerator::initialize"); return; }
    if (!_timers_enabled) {   // device-side chronometers (HIP events + in-kernel clocks) behind getTimeConsumptionSeconds_*()
      hipCheck(_hip->ctx, vslam_enable_timers(_hip->ctx, 1), "HipStereoFramePointGenerator::initialize");
      _timers_enabled = true;
    }
    const cv::Mat& L = frame_->intensityImageLeft();
    const cv::Mat& R = frame_->intensityImageRight();
    if (!L.data || !R.data) throw std::runtime_error("HipStereoFramePointGenerator::initialize|called with empty frame");
    // the frame is created with the tracker's status and the world map's pose (pose_tracker_3d.cpp:66-74): both enter
    // through the setters BEFORE the frame begins (the triangulation-distance rule of :109-125 reads the status)
    _hip->status = frame_->status() == Frame::Localizing ? VSLAM_LOCALIZING : VSLAM_TRACKING;
    pushState(TransformMatrix3D::Identity());
    double pose[12];
    hipToArray(frame_->cameraLeftToWorld(), pose);
    hipCheck(_hip->ctx, vslam_set_pose(_hip->ctx, 0, pose), "HipStereoFramePointGenerator::initialize");
    { HIP_PROFILE(BEGIN_CALL); hipCheck(_hip->ctx, vslam_frame_begin(_hip->ctx, L.data, R.data, (int32_t)static_cast<size_t>(L.step), 0, 0), "HipStereoFramePointGenerator::initialize"); }
    _pruned = false; _computed = false;
    downloadCoordinates(frame_);   // Frame::keypointsLeft/Right now; descriptorsLeft/Right by the next call on this frame (track(): around the device's tracking stage)
  }

  //! StereoFramePointGenerator::track (:464-681)
  void track(Frame* frame_, Frame* frame_previous_, const TransformMatrix3D& camera_left_previous_in_current_,
             FramePointPointerVector& lost_points_, const bool track_by_appearance_ = true) override {
    if (!frame_ || !frame_previous_) throw std::runtime_error("HipStereoFramePointGenerator::track|called with invalid frames");
    // setProjectionTrackingDistancePixels / setMaximumDescriptorDistanceTracking were called by the tracker (:237-238)
    _hip->status = frame_->status() == Frame::Localizing ? VSLAM_LOCALIZING : VSLAM_TRACKING;
    _hip->window_pixels = _projection_tracking_distance_pixels;
    _hip->tau_track = _maximum_descriptor_distance_tracking;
    const bool copy = _descriptors_pending;
    vslam_keypoints_view keypoint_view;
    if (copy) waitDescriptors(keypoint_view);  // the descriptors' report is read (header) before the tracking stage's report replaces it
    { HIP_PROFILE(TRACK_CALL);
      pushState(camera_left_previous_in_current_);
      hipCheck(_hip->ctx, vslam_track(_hip->ctx, track_by_appearance_ ? 1 : 0), "HipStereoFramePointGenerator::track"); }
    if (copy) copyDescriptors(frame_, keypoint_view);   // descriptor rows and feature stores, while the device tracks
    materializeTrackedPoints(frame_, frame_previous_, lost_points_);
    if (_hip->aligner) _hip->aligner->invalidate((Count)frame_->points().size());
  }

  //! StereoFramePointGenerator::recoverPoints (:683-869).  PoseTracker3D::_prunePoints has just pruned the host list; the
  //! device prunes its own by the same rule and recovers in the same launch
  void recoverPoints(Frame* current_frame_, const FramePointPointerVector& lost_points_) const override {
    (void)lost_points_;   // the device kept the lost list of its own track()
    const_cast<HipStereoFramePointGenerator*>(this)->ensureKeypoints(current_frame_);
    pruneOnDevice(current_frame_);
    materializeRecoveredPoints(current_frame_);
  }

  //! StereoFramePointGenerator::compute (:135-462); PoseTracker3D::_updatePoints has just run on the host objects — the
  //! device updates its landmarks here (they feed the next frame's aligner)
  void compute(Frame* frame_) override {
    if (!frame_) throw std::runtime_error("HipStereoFramePointGenerator::compute|called with empty frame");
    ensureKeypoints(frame_);                 // first frame: no track() has filled Frame::keypoints yet
    if (!_pruned) pruneOnDevice(frame_);     // recovery disabled or no previous frame: _prunePoints alone
    { HIP_PROFILE(COMPUTE_CALL);
      // landmark update + stereo sweep, one launch — already under way when recoverPoints() ran (it starts them as soon as it has read the
      // recovered points' report): PoseTracker3D::compute calls _updatePoints (host objects only) and this function next, unconditionally
      if (!_computed) hipCheck(_hip->ctx, vslam_compute(_hip->ctx), "HipStereoFramePointGenerator::compute"); }
    materializeNewPoints(frame_);  // Frame::createFramepoint(feature_left, feature_right, distance, xyz); chronometers
  }

  HipContext* hip() { return _hip; }
  //! the device's per-frame report of the frame last finished (counters the tests compare with the fused path)
  vslam_frame_info frameInfo() const { vslam_frame_info info; hipCheck(_hip->ctx, vslam_get_frame_info(_hip->ctx, 0, &info), "frameInfo"); return info; }
  //! the same report as compute() received it with the frame's last stage view (no device round trip)
  const vslam_frame_info& lastFrameInfo() const { return _last_info; }

private:
  //! SLAMAssembly::printReport reads getTimeConsumptionSeconds_keypoint_detection / _descriptor_extraction (base generator,
  //! base_framepoint_generator.h:232-233) and _point_triangulation (stereo_framepoint_generator.h:81) from the generator
  //! (slam_assembly.cpp:709-719); CREATE_CHRONOMETER makes the members protected (definitions.h:144-146), so they are set here
  //! from the device's own clocks: accumulated seconds since the context was created, like CHRONOMETER_STOP accumulates
  void updateChronometers(const vslam_points_view& view_) {
    HIP_PROFILE(CHRONO);
    double ms[8]; int32_t launches[8];
    hipCheck(_hip->ctx, vslam_get_kernel_times(_hip->ctx, ms, launches), "HipStereoFramePointGenerator|timers");
    _time_consumption_seconds_keypoint_detection = (ms[0] + ms[1]) * 1e-3;     // FAST / NMS + emission / threshold controller
    _time_consumption_seconds_descriptor_extraction = ms[2] * 1e-3;
    _time_consumption_seconds_point_triangulation = view_.seconds_point_triangulation;
  }
  void pushState(const TransformMatrix3D& prior_) const {
    double prior[12];
    hipToArray(prior_, prior);
    hipCheck(_hip->ctx, vslam_set_tracker_state(_hip->ctx, 0, _hip->status, prior, _hip->window_pixels, _hip->tau_track), "HipStereoFramePointGenerator|state");
  }
  //! getPointInLeftCamera (:871-895) from the configuration, the arithmetic of the device's triangulation
  PointCoordinates pointInLeftCamera(const double xL, const double yL, const double xR, const double yR) const {
    const vslam_config& c = _hip->config;
    const double z = c.baseline_h[0] / (xR - xL);
    return PointCoordinates(1 / c.K[0] * (xL - c.K[2]) * z, 1 / c.K[4] * ((yL + yR) / 2.0 - c.K[5]) * z, z);
  }
  void pruneOnDevice(Frame* frame_) const {
    HIP_PROFILE(PRUNE_CALL);
    double pose[12];
    hipToArray(frame_->cameraLeftToWorld(), pose);   // Frame::setRobotToWorld happened on the host (pose_tracker_3d.cpp:151-152,175)
    hipCheck(_hip->ctx, vslam_set_pose(_hip->ctx, 0, pose), "HipStereoFramePointGenerator|prune");
    hipCheck(_hip->ctx, vslam_prune_recover(_hip->ctx), "HipStereoFramePointGenerator|prune");
    _pruned = true;
  }

  //! IntensityFeature objects (frame_point.h:18-35: a cv::KeyPoint copy + a cv::Mat row header each) are what
  //! Frame::createFramepoint takes; only the features that end up in a framepoint (~1 in 3) need one, so they are made on first
  //! use.  The store is reserved for every feature up front: pointers handed out stay valid for the frame.
  struct LazyFeatures {
    std::vector<IntensityFeature> store;
    std::vector<int32_t> slot;
    const std::vector<cv::KeyPoint>* keypoints = nullptr;
    const cv::Mat* descriptors = nullptr;
    void reset(const std::vector<cv::KeyPoint>& keypoints_, const cv::Mat& descriptors_) {
      keypoints = &keypoints_; descriptors = &descriptors_;
      store.clear(); store.reserve(keypoints_.size());
      slot.assign(keypoints_.size(), -1);
    }
    int32_t size() const { return (int32_t)slot.size(); }
    const IntensityFeature* get(int32_t i) {
      if (slot[i] < 0) { slot[i] = (int32_t)store.size(); store.emplace_back((*keypoints)[i], descriptors->row(i), (size_t)i); }
      return &store[slot[i]];
    }
  };
  //! stage views -> cv::KeyPoint(x, y, 7, -1, score) + n x 32 CV_8U descriptor rows (frame.h:64-67).  The device lists are
  //! image row-major: (row << 16 | col) rises strictly, so the feature at a pixel is found by bisection (compute() needs it for
  //! the new points).
  void materializeCoordinates(const vslam_keypoints_view& view_, int side_, std::vector<cv::KeyPoint>& keypoints_, std::vector<uint32_t>& pixel_keys_) const {
    const int32_t n = view_.n[side_];
    const int16_t* xy = view_.xy[side_];
    const uint8_t* score = view_.score[side_];
    keypoints_.resize(n);
    pixel_keys_.resize(n);
    uint32_t last = 0;
    for (int32_t i = 0; i < n; ++i) {
      keypoints_[i] = cv::KeyPoint((float)xy[2 * i], (float)xy[2 * i + 1], 7.f, -1.f, (float)score[i]);   // FAST: size 7, no angle
      const uint32_t key = ((uint32_t)(uint16_t)xy[2 * i + 1] << 16) | (uint16_t)xy[2 * i];
      if (i && key <= last) throw std::runtime_error("HipStereoFramePointGenerator|keypoints are not in image row-major order");
      pixel_keys_[i] = last = key;
    }
  }
  //! Frame::keypointsLeft/Right + descriptorsLeft/Right in three steps that interleave with the device: (1) initialize(): coordinates and scores
  //! arrive right after the detector (vslam_view_keypoints_xy) and the cv::KeyPoint lists are built while the device still describes; (2) track()
  //! waits for the descriptors' report, (3) launches the tracking stage and copies the descriptor rows / resets the feature stores while it runs.
  //! A stage launch leaves the keypoint regions of the report buffer alone (they are rewritten by the next frame's keypoint report only).
  //! PoseTracker3D::compute reads none of this between initialize() and _track().
  void downloadCoordinates(Frame* frame_) {
    vslam_keypoints_view view;
    { HIP_PROFILE(KEYPOINTS_WAIT); hipCheck(_hip->ctx, vslam_view_keypoints_xy(_hip->ctx, 0, &view), "HipStereoFramePointGenerator|keypoints"); }
    HIP_PROFILE(KEYPOINTS_HOST);
    materializeCoordinates(view, 0, frame_->keypointsLeft(), _pixel_left);
    materializeCoordinates(view, 1, frame_->keypointsRight(), _pixel_right);
    _descriptors_pending = true;
  }
  void waitDescriptors(vslam_keypoints_view& view_) {
    HIP_PROFILE(KEYPOINTS_WAIT);
    hipCheck(_hip->ctx, vslam_view_keypoints(_hip->ctx, 0, &view_), "HipStereoFramePointGenerator|descriptors");
  }
  void copyDescriptors(Frame* frame_, const vslam_keypoints_view& view_) {
    HIP_PROFILE(KEYPOINTS_HOST);
    cv::Mat* descriptors[2] = {&frame_->descriptorsLeft(), &frame_->descriptorsRight()};
    const std::vector<cv::KeyPoint>* keypoints[2] = {&frame_->keypointsLeft(), &frame_->keypointsRight()};
    for (int side = 0; side < 2; ++side) {
      const int32_t n = view_.n[side];
      if (n != (int32_t)keypoints[side]->size()) throw std::runtime_error("HipStereoFramePointGenerator|keypoint reports of one frame disagree");
      *descriptors[side] = cv::Mat(n > 0 ? n : 1, VSLAM_DESC_BYTES, CV_8UC1);
      if (n > 0) std::memcpy(descriptors[side]->ptr<uint8_t>(0), view_.desc[side], (size_t)n * VSLAM_DESC_BYTES);   // freshly created Mat: dense rows
    }
    _features_left.reset(frame_->keypointsLeft(), frame_->descriptorsLeft());
    _features_right.reset(frame_->keypointsRight(), frame_->descriptorsRight());
    _number_of_detected_keypoints = (Count)_features_left.size();
    _descriptors_pending = false;
  }
  //! every path that needs the frame's descriptors without a track() in front of it (first frame; a caller that skips track())
  void ensureKeypoints(Frame* frame_) {
    if (!_descriptors_pending) return;
    vslam_keypoints_view view;
    waitDescriptors(view);
    copyDescriptors(frame_, view);
  }
  static int32_t featureAt(const std::vector<uint32_t>& pixel_keys_, int16_t x_, int16_t y_) {
    const uint32_t key = ((uint32_t)(uint16_t)y_ << 16) | (uint16_t)x_;
    const auto it = std::lower_bound(pixel_keys_.begin(), pixel_keys_.end(), key);
    return it != pixel_keys_.end() && *it == key ? (int32_t)(it - pixel_keys_.begin()) : -1;
  }

  //! the tracked list of the device -> Frame::createFramepoint(left, right, distance, xyz, previous) in the order of the
  //! previous points, FramePoint::setEpipolarOffset, the lost list, the tracking statistics (:632-668)
  void materializeTrackedPoints(Frame* frame_, Frame* previous_, FramePointPointerVector& lost_) {
    FramePointPointerVector& previous_points(previous_->points());
    vslam_track_view view;
    { HIP_PROFILE(TRACK_WAIT); hipCheck(_hip->ctx, vslam_view_track(_hip->ctx, 0, &view), "HipStereoFramePointGenerator::track"); }
    HIP_PROFILE(TRACK_HOST);
    const int32_t n_tracked = view.n_tracked, n_lost = view.n_lost;
    const int32_t* out4 = view.tracked4;
    FramePointPointerVector& points(frame_->points());
    points.resize(n_tracked);
    _number_of_tracked_landmarks = 0;
    real accumulated_descriptor_distance = 0;
    for (int32_t u = 0; u < n_tracked; ++u) {
      const int32_t ip = out4[4 * u], fl = out4[4 * u + 1], fr = out4[4 * u + 2], dist = out4[4 * u + 3];
      if (ip < 0 || ip >= (int32_t)previous_points.size() || fl < 0 || fl >= (int32_t)_features_left.size() || fr < 0 || fr >= (int32_t)_features_right.size())
        throw std::runtime_error("HipStereoFramePointGenerator::track|host and device frames diverged");
      const IntensityFeature* feature_left = _features_left.get(fl);
      const IntensityFeature* feature_right = _features_right.get(fr);
      FramePoint* point_previous = previous_points[ip];
      FramePoint* framepoint = frame_->createFramepoint(feature_left, feature_right, (real)dist,
                                                        pointInLeftCamera(feature_left->keypoint.pt.x, feature_left->keypoint.pt.y,
                                                                          feature_right->keypoint.pt.x, feature_right->keypoint.pt.y),
                                                        point_previous);
      framepoint->setEpipolarOffset(feature_right->row - feature_left->row);
      accumulated_descriptor_distance += dist;
      points[u] = framepoint;
      if (point_previous->landmark()) ++_number_of_tracked_landmarks;
    }
    lost_.resize(n_lost);
    for (int32_t u = 0; u < n_lost; ++u) {
      if (view.lost[u] < 0 || view.lost[u] >= (int32_t)previous_points.size()) throw std::runtime_error("HipStereoFramePointGenerator::track|bad lost index");
      lost_[u] = previous_points[view.lost[u]];
    }
    previous_->setAverageDescriptorDistanceTracking(accumulated_descriptor_distance / n_tracked);
  }

  //! after vslam_prune_recover the device's frame holds the survivors of _prunePoints followed by the recovered points
  //! (:841-856: features owned by the framepoint, previous = the lost point)
  void materializeRecoveredPoints(Frame* frame_) const {
    Frame* previous = frame_->previous();
    if (!previous) return;
    vslam_points_view view;
    { HIP_PROFILE(PRUNE_WAIT); hipCheck(_hip->ctx, vslam_view_points(_hip->ctx, 0, 1, &view), "HipStereoFramePointGenerator::recoverPoints"); }
    HIP_PROFILE(PRUNE_HOST);
    const int32_t n = view.n;
    FramePointPointerVector& points(frame_->points());
    const int32_t n_kept = (int32_t)points.size();
    if (n < n_kept || view.first_full != n_kept) throw std::runtime_error("HipStereoFramePointGenerator::recoverPoints|host and device frames diverged (prune)");
    for (int32_t i = 0; i < n_kept; ++i)
      if (points[i]->keypointLeft().pt.x != (float)view.kp[4 * i] || points[i]->keypointLeft().pt.y != (float)view.kp[4 * i + 1])
        throw std::runtime_error("HipStereoFramePointGenerator::recoverPoints|host and device frames diverged (prune order)");
    // The frame's point list is final on the device: its landmark update (PoseTracker3D::_updatePoints, the next thing the tracker does on the
    // host objects) and the stereo sweep of compute() start NOW and run while the recovered points are materialised below.  The stage view's
    // memory is reused by that launch's report, so the few recovered entries are copied out first.
    const int32_t n_rec = n - n_kept;
    _rec_kp.assign(view.kp + 4 * (size_t)n_kept, view.kp + 4 * (size_t)n);
    _rec_meta.assign(view.meta + 6 * (size_t)n_kept, view.meta + 6 * (size_t)n);
    _rec_cam.assign(view.cam + 3 * (size_t)n_kept, view.cam + 3 * (size_t)n);
    _rec_desc.assign(view.desc + 64 * (size_t)n_kept, view.desc + 64 * (size_t)n);
    hipCheck(_hip->ctx, vslam_compute(_hip->ctx), "HipStereoFramePointGenerator::recoverPoints");
    _computed = true;
    points.resize(n);
    for (int32_t k = 0; k < n_rec; ++k) {
      const int32_t ip = _rec_meta[6 * k + 2];
      if (ip < 0 || ip >= (int32_t)previous->points().size()) throw std::runtime_error("HipStereoFramePointGenerator::recoverPoints|bad previous index");
      FramePoint* point_previous = previous->points()[ip];
      cv::Mat descriptor_left(1, VSLAM_DESC_BYTES, CV_8UC1), descriptor_right(1, VSLAM_DESC_BYTES, CV_8UC1);
      std::memcpy(descriptor_left.ptr<uint8_t>(0), &_rec_desc[(size_t)64 * k], VSLAM_DESC_BYTES);
      std::memcpy(descriptor_right.ptr<uint8_t>(0), &_rec_desc[(size_t)64 * k + VSLAM_DESC_BYTES], VSLAM_DESC_BYTES);
      cv::KeyPoint keypoint_left(point_previous->keypointLeft()), keypoint_right(point_previous->keypointRight());
      keypoint_left.pt.x = _rec_kp[4 * k]; keypoint_left.pt.y = _rec_kp[4 * k + 1];
      keypoint_right.pt.x = _rec_kp[4 * k + 2]; keypoint_right.pt.y = _rec_kp[4 * k + 3];
      const IntensityFeature feature_left(keypoint_left, descriptor_left, 0), feature_right(keypoint_right, descriptor_right, 0);
      points[n_kept + k] = frame_->createFramepoint(&feature_left, &feature_right, (real)_rec_meta[6 * k],
                                                    PointCoordinates(_rec_cam[3 * k], _rec_cam[3 * k + 1], _rec_cam[3 * k + 2]), point_previous);
    }
  }

  //! the finished device frame = the host list so far + the new stereo points in emission order
  void materializeNewPoints(Frame* frame_) {
    vslam_points_view view;
    { HIP_PROFILE(COMPUTE_WAIT); hipCheck(_hip->ctx, vslam_view_points(_hip->ctx, 0, 0, &view), "HipStereoFramePointGenerator::compute"); }
    HIP_PROFILE(COMPUTE_HOST);
    const int32_t n = view.n;
    const int16_t* kp = view.kp; const int32_t* meta = view.meta; const double* cam = view.cam;
    FramePointPointerVector& points(frame_->points());
    const int32_t n_old = (int32_t)points.size();
    if (n < n_old) throw std::runtime_error("HipStereoFramePointGenerator::compute|host and device frames diverged");
    points.resize(n);
    for (int32_t i = n_old; i < n; ++i) {
      const int32_t fl = featureAt(_pixel_left, kp[4 * i], kp[4 * i + 1]), fr = featureAt(_pixel_right, kp[4 * i + 2], kp[4 * i + 3]);
      if (fl < 0 || fr < 0) throw std::runtime_error("HipStereoFramePointGenerator::compute|new point without a feature");
      FramePoint* framepoint = frame_->createFramepoint(_features_left.get(fl), _features_right.get(fr), (real)meta[6 * i],
                                                        PointCoordinates(cam[3 * i], cam[3 * i + 1], cam[3 * i + 2]));
      framepoint->setEpipolarOffset(meta[6 * i + 1]);
      points[i] = framepoint;
    }
    _last_info = view.info;
    updateChronometers(view);
  }

  HipContext* _hip;
  mutable bool _pruned = false, _computed = false;
  bool _descriptors_pending = false;   // Frame::descriptorsLeft/Right and the feature stores of the frame are still to be filled
  mutable std::vector<int16_t> _rec_kp; mutable std::vector<int32_t> _rec_meta; mutable std::vector<double> _rec_cam; mutable std::vector<uint8_t> _rec_desc;   // recovered points' report entries (copied out of the stage view)
  bool _timers_enabled = false;
  LazyFeatures _features_left, _features_right;                            // keypoints + descriptors of the current frame
  std::vector<uint32_t> _pixel_left, _pixel_right;                         // (row << 16 | col) of feature i, strictly rising
  vslam_frame_info _last_info{};                                           // the device's report of the frame last finished
};

}  // namespace proslam
