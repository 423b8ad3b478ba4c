// proslam_hip_plugin.h — header-only C++ shim: the reference's plug-in classes on top of libvslam_hip.so.
//
// Compiles ONLY inside the reference tree (needs its headers: OpenCV 3, Eigen, srrg_core); it is not built in
// this repository (those dependencies are absent here).  INTEGRATION.md shows the two lines of
// SLAMAssembly::_createStereoTracker (src/system/slam_assembly.cpp:61-76) a maintainer changes.
//
// The classes derive from the CONCRETE reference classes because SLAMAssembly down-casts the generator
// (slam_assembly.h:101, slam_assembly.cpp:690-691,717-719) and PoseTracker3D keeps its own control flow
// (pose_tracker_3d.cpp:32-566): every virtual below is one C call; results are materialised into the host
// objects the rest of the reference reads (Frame::keypoints/descriptors, FramePoint via Frame::createFramepoint,
// BaseAligner's protected result members).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "aligners/stereouv_aligner.h"
#include "framepoint_generation/stereo_framepoint_generator.h"
#include "vslam_hip.h"

namespace proslam {

inline void hipCheck(vslam_ctx* ctx, int rc, const char* where) {
  if (rc != VSLAM_OK) throw std::runtime_error(std::string(where) + "|" + vslam_last_error(ctx));
}

//! one device context shared by the generator and the aligner of a tracker (n_streams = 1)
struct HipContext {
  vslam_ctx* ctx = nullptr;
  vslam_config config;
  ~HipContext() { if (ctx) vslam_destroy(ctx); }
};

class HipStereoFramePointGenerator : public StereoFramePointGenerator {
public:
  HipStereoFramePointGenerator(StereoFramePointGeneratorParameters* parameters_, HipContext* hip_)
      : StereoFramePointGenerator(parameters_), _hip(hip_) {}

  //! BaseFramePointGenerator::configure + StereoFramePointGenerator::configure: parameters -> vslam_config
  void configure() override {
    StereoFramePointGenerator::configure();  // keeps the inherited members (bins, targets, chronometers) valid
    vslam_config& c = _hip->config;
    vslam_default_config_kitti(&c);
    c.rows = _number_of_rows_image; c.cols = _number_of_cols_image;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.K[3 * i + j] = _camera_left->cameraMatrix()(i, j);
    for (int i = 0; i < 3; ++i) c.baseline_h[i] = _camera_right->baselineHomogeneous()(i);
    StereoFramePointGeneratorParameters* p = static_cast<StereoFramePointGeneratorParameters*>(parameters());
    c.det_rows = p->number_of_detectors_vertical; c.det_cols = p->number_of_detectors_horizontal;
    c.detector_threshold_minimum = p->detector_threshold_minimum; c.detector_threshold_maximum = p->detector_threshold_maximum;
    c.detector_threshold_maximum_change = p->detector_threshold_maximum_change;
    c.target_number_of_keypoints_tolerance = p->target_number_of_keypoints_tolerance;
    c.bin_size_pixels = p->bin_size_pixels; c.enable_keypoint_binning = p->enable_keypoint_binning;
    c.minimum_projection_tracking_distance_pixels = p->minimum_projection_tracking_distance_pixels;
    c.maximum_projection_tracking_distance_pixels = p->maximum_projection_tracking_distance_pixels;
    c.minimum_descriptor_distance_tracking = p->minimum_descriptor_distance_tracking;
    c.maximum_descriptor_distance_tracking = p->maximum_descriptor_distance_tracking;
    c.maximum_reliable_depth_meters = p->maximum_reliable_depth_meters; c.maximum_depth_meters = p->maximum_depth_meters;
    c.minimum_depth_meters = p->minimum_depth_meters;
    c.maximum_matching_distance_triangulation = p->maximum_matching_distance_triangulation;
    c.minimum_disparity_pixels = p->minimum_disparity_pixels;
    c.maximum_epipolar_search_offset_pixels = p->maximum_epipolar_search_offset_pixels;
    // tracker / aligner / landmark values are filled by HipStereoUVAligner::configure before vslam_create
  }

  //! StereoFramePointGenerator::initialize (stereo_framepoint_generator.cpp:73-133)
  void initialize(Frame* frame_, const bool& extract_features_ = true) override {
    if (!frame_) throw std::runtime_error("HipStereoFramePointGenerator::initialize|called with empty frame");
    if (!extract_features_) { hipCheck(_hip->ctx, vslam_frame_restore(_hip->ctx), "initialize"); return; }
    const cv::Mat& L = frame_->intensityImageLeft();
    const cv::Mat& R = frame_->intensityImageRight();
    // the tracker status of the frame and its pose enter through the setters
    double prior[12]; toArray(TransformMatrix3D::Identity(), prior);
    hipCheck(_hip->ctx, vslam_frame_begin(_hip->ctx, L.data, R.data, (int32_t)L.step, 0, 0), "initialize");
    downloadKeypoints(frame_);   // Frame::keypointsLeft/Right + descriptorsLeft/Right for downstream consumers
  }

  //! StereoFramePointGenerator::track (:464-681)
  void track(Frame* frame_, Frame* frame_previous_, const TransformMatrix3D& camera_left_previous_in_current_,
             FramePointPointerVector& lost_points_, const bool track_by_appearance_ = true) override {
    if (!frame_ || !frame_previous_) throw std::runtime_error("HipStereoFramePointGenerator::track|called with invalid frames");
    double prior[12]; toArray(camera_left_previous_in_current_, prior);
    hipCheck(_hip->ctx, vslam_set_tracker_state(_hip->ctx, 0, frame_->status() == Frame::Localizing ? VSLAM_LOCALIZING : VSLAM_TRACKING,
                                                prior, _projection_tracking_distance_pixels, _maximum_descriptor_distance_tracking), "track");
    hipCheck(_hip->ctx, vslam_track(_hip->ctx, track_by_appearance_ ? 1 : 0), "track");
    vslam_frame_info info;
    hipCheck(_hip->ctx, vslam_get_frame_info(_hip->ctx, 0, &info), "track");
    _number_of_tracked_landmarks = info.n_tracked_landmarks;
    materializeTrackedPoints(frame_, frame_previous_, lost_points_, info);  // Frame::createFramepoint(..., previous)
  }

  //! StereoFramePointGenerator::recoverPoints (:683-869) — runs together with the tracker's _prunePoints on the device
  void recoverPoints(Frame* current_frame_, const FramePointPointerVector& lost_points_) const override {
    double pose[12]; toArray(current_frame_->cameraLeftToWorld(), pose);
    hipCheck(_hip->ctx, vslam_set_pose(_hip->ctx, 0, pose), "recoverPoints");
    hipCheck(_hip->ctx, vslam_prune_recover(_hip->ctx), "recoverPoints");
    materializeRecoveredPoints(current_frame_, lost_points_);
  }

  //! StereoFramePointGenerator::compute (:135-462); the landmark refinement of PoseTracker3D::_updatePoints has to
  //! precede it on the device (it feeds the next frame's aligner), so it is issued here
  void compute(Frame* frame_) override {
    if (!frame_) throw std::runtime_error("HipStereoFramePointGenerator::compute|called with empty frame");
    hipCheck(_hip->ctx, vslam_update_points(_hip->ctx), "compute");
    hipCheck(_hip->ctx, vslam_stereo_new(_hip->ctx), "compute");
    materializeNewPoints(frame_);  // Frame::createFramepoint(feature_left, feature_right, distance, xyz)
  }

  HipContext* hip() { return _hip; }

private:
  static void toArray(const TransformMatrix3D& T, double* out) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = T.matrix()(i, j);
  }
  // The four helpers below copy SoA read-backs (vslam_get_keypoints / vslam_get_points) into the reference's host
  // objects: cv::KeyPoint(x, y, 7.f, -1, score), 1x32 CV_8U descriptor rows, IntensityFeature pairs handed to
  // Frame::createFramepoint (types/frame.cpp:61-84), FramePoint::setEpipolarOffset, lost list from previous points
  // whose next() stayed null.  Bodies are mechanical; see INTEGRATION.md §3 for the field mapping.
  void downloadKeypoints(Frame* frame_);
  void materializeTrackedPoints(Frame* frame_, Frame* previous_, FramePointPointerVector& lost_, const vslam_frame_info& info_);
  void materializeRecoveredPoints(Frame* frame_, const FramePointPointerVector& lost_) const;
  void materializeNewPoints(Frame* frame_);
  HipContext* _hip;
};

class HipStereoUVAligner : public StereoUVAligner {
public:
  HipStereoUVAligner(AlignerParameters* parameters_, HipContext* hip_) : StereoUVAligner(parameters_), _hip(hip_) {}

  //! creates the device context once every parameter is known (called after the generator's configure)
  void configure() override {
    vslam_config& c = _hip->config;
    c.aligner_error_delta_for_convergence = _parameters->error_delta_for_convergence;
    c.aligner_maximum_error_kernel = _parameters->maximum_error_kernel;
    c.aligner_damping = _parameters->damping;
    c.aligner_maximum_number_of_iterations = _parameters->maximum_number_of_iterations;
    c.aligner_minimum_number_of_inliers = _parameters->minimum_number_of_inliers;
    c.minimum_depth_meters = _minimum_reliable_depth_meters;
    c.maximum_reliable_depth_meters = _maximum_reliable_depth_meters;
    if (!_hip->ctx) hipCheck(nullptr, vslam_create(&c, 0, 1, &_hip->ctx), "HipStereoUVAligner::configure");
  }

  //! StereoUVAligner::initialize (stereouv_aligner.cpp:10-69): the correspondences are already on the device
  void initialize(const Frame* frame_previous_, const Frame* frame_current_, const TransformMatrix3D& previous_to_current_) override {
    _frame_previous = frame_previous_; _frame_current = frame_current_; _previous_to_current = previous_to_current_;
    _number_of_measurements = _frame_current->points().size();
  }
  void linearize(const bool&) override {}   // folded into converge() on the device
  void oneRound(const bool&) override {}

  //! StereoUVAligner::converge (:210-264): one launch; results into the base-class members the tracker reads
  void converge() override {
    hipCheck(_hip->ctx, vslam_align(_hip->ctx, _parameters->enable_inverse_depth_as_information ? 1 : 0), "converge");
    std::vector<double> chi(_number_of_measurements);
    std::vector<uint8_t> inl(_number_of_measurements);
    double T[12], H[36];
    int32_t n = 0;
    hipCheck(_hip->ctx, vslam_get_aligner_result(_hip->ctx, 0, (int32_t)_number_of_measurements, &n, chi.data(), inl.data(), T, H), "converge");
    vslam_frame_info info;
    hipCheck(_hip->ctx, vslam_get_frame_info(_hip->ctx, 0, &info), "converge");
    _errors.assign(chi.begin(), chi.end());
    _inliers.resize(n);
    for (int32_t u = 0; u < n; ++u) _inliers[u] = inl[u] != 0;
    _number_of_inliers = info.n_inliers; _number_of_outliers = info.n_outliers; _total_error = info.total_error;
    _has_system_converged = info.aligner_converged != 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) _previous_to_current.matrix()(i, j) = T[4 * i + j];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { _H(i, j) = H[6 * i + j]; _information_matrix(i, j) = H[6 * i + j]; }
  }

private:
  HipContext* _hip;
};

}  // namespace proslam
