// proslam_hip_mirror.hpp — TEST-ONLY host mirror above the C ABI, in C++14, mirroring the reference's plug-in interface for the hot
// path (same class and method names, argument meaning and error behaviour) without its OpenCV / Eigen / srrg
// dependencies, which are absent from this image.  Inside the reference tree use shim/proslam_hip_plugin.h instead
// (it derives from the reference's own classes).  Single sequence (n_streams = 1), as the reference.
//
//   proslam::StereoFramePointGenerator   src/framepoint_generation/stereo_framepoint_generator.h:7-82
//   proslam::StereoUVAligner             src/aligners/stereouv_aligner.h:7-39, base_aligner.h:37-48
//   proslam::PoseTracker3D               src/position_tracking/pose_tracker_3d.h:14-134
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vslam_hip.h"

namespace proslam_hip {

typedef double real;

struct TransformMatrix3D {  // row-major 3x4 [R|t]
  real m[12];
  static TransformMatrix3D Identity() { TransformMatrix3D T; std::memset(T.m, 0, sizeof T.m); T.m[0] = T.m[5] = T.m[10] = 1; return T; }
  TransformMatrix3D operator*(const TransformMatrix3D& B) const {
    TransformMatrix3D C;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) C.m[4 * i + j] = (m[4 * i] * B.m[j] + m[4 * i + 1] * B.m[4 + j]) + m[4 * i + 2] * B.m[8 + j];
      C.m[4 * i + 3] = ((m[4 * i] * B.m[3] + m[4 * i + 1] * B.m[7]) + m[4 * i + 2] * B.m[11]) + m[4 * i + 3];
    }
    return C;
  }
  TransformMatrix3D inverse() const {
    TransformMatrix3D C;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[4 * i + j] = m[4 * j + i];
    for (int i = 0; i < 3; ++i) C.m[4 * i + 3] = -((C.m[4 * i] * m[3] + C.m[4 * i + 1] * m[7]) + C.m[4 * i + 2] * m[11]);
    return C;
  }
  real translationNorm() const { return std::sqrt((m[3] * m[3] + m[7] * m[7]) + m[11] * m[11]); }
  real rotationAngle() const {  // WorldMap::toOrientationRodrigues(linear()).norm()
    const real rx = m[9] - m[6], ry = m[2] - m[8], rz = m[4] - m[1];
    const real s = std::sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25);
    real c = ((m[0] + m[5]) + m[10] - 1) * 0.5;
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    if (s < 1e-5) return c > 0 ? 0.0 : 3.14159265358979323846;
    return std::acos(c);
  }
};

//! Frame (types/frame.h) reduced to what crosses the plug-in boundary
struct Frame {
  enum Status { Localizing = VSLAM_LOCALIZING, Tracking = VSLAM_TRACKING };
  const uint8_t* intensity_image_left = nullptr;
  const uint8_t* intensity_image_right = nullptr;
  int32_t row_stride = 0;
  Status status = Localizing;
  TransformMatrix3D camera_left_to_world = TransformMatrix3D::Identity();
  Frame* previous = nullptr;
  int32_t number_of_points = 0;   // points().size()
  vslam_frame_info info;
};

inline void check(vslam_ctx* ctx, int rc, const char* where) {
  if (rc != VSLAM_OK) throw std::runtime_error(std::string(where) + "|" + vslam_last_error(ctx));
}

//! shared device context of the generator and the aligner of one tracker
struct HipContext {
  vslam_ctx* ctx = nullptr;
  vslam_config config;
  explicit HipContext(const vslam_config& config_, int device_ = 0) : config(config_) { check(nullptr, vslam_create(&config, device_, 1, &ctx), "HipContext"); }
  ~HipContext() { if (ctx) vslam_destroy(ctx); }
  HipContext(const HipContext&) = delete;
  HipContext& operator=(const HipContext&) = delete;
};

class StereoFramePointGenerator {
public:
  explicit StereoFramePointGenerator(HipContext* hip_) : _hip(hip_) {}
  void configure() {
    _projection_tracking_distance_pixels = _hip->config.maximum_projection_tracking_distance_pixels;
    _maximum_descriptor_distance_tracking = _hip->config.maximum_descriptor_distance_tracking;
    _target_number_of_keypoints = (_hip->config.cols / _hip->config.bin_size_pixels + 1) * (_hip->config.rows / _hip->config.bin_size_pixels + 1);
  }
  //! initializes the framepoint generator (detects keypoints and computes descriptors in both images)
  void initialize(Frame* frame_, const bool& extract_features_ = true) {
    if (!frame_) throw std::runtime_error("StereoFramePointGenerator::initialize|called with empty frame");
    if (!extract_features_) { check(_hip->ctx, vslam_frame_restore(_hip->ctx), "initialize"); return; }
    check(_hip->ctx, vslam_frame_begin(_hip->ctx, frame_->intensity_image_left, frame_->intensity_image_right, frame_->row_stride,
                                       0, 0), "StereoFramePointGenerator::initialize");
  }
  //! computes tracks between previous and current framepoints
  void track(Frame* frame_, Frame* frame_previous_, const TransformMatrix3D& camera_left_previous_in_current_,
             int32_t& number_of_lost_points_, const bool track_by_appearance_ = true) {
    if (!frame_ || !frame_previous_) throw std::runtime_error("StereoFramePointGenerator::track|called with invalid frames");
    check(_hip->ctx, vslam_set_tracker_state(_hip->ctx, 0, frame_->status, camera_left_previous_in_current_.m,
                                             _projection_tracking_distance_pixels, _maximum_descriptor_distance_tracking), "track");
    check(_hip->ctx, vslam_track(_hip->ctx, track_by_appearance_ ? 1 : 0), "StereoFramePointGenerator::track");
    check(_hip->ctx, vslam_get_frame_info(_hip->ctx, 0, &frame_->info), "track");
    _number_of_tracked_landmarks = frame_->info.n_tracked_landmarks;
    frame_->number_of_points = frame_->info.n_tracked;
    number_of_lost_points_ = frame_->info.n_lost;
  }
  //! _prunePoints of the tracker + recoverPoints, evaluated together on the device with the refined pose
  void recoverPoints(Frame* current_frame_) const {
    check(_hip->ctx, vslam_set_pose(_hip->ctx, 0, current_frame_->camera_left_to_world.m), "recoverPoints");
    check(_hip->ctx, vslam_prune_recover(_hip->ctx), "StereoFramePointGenerator::recoverPoints");
  }
  //! computes the remaining (new) framepoints by exhaustive rigid stereo matching
  void compute(Frame* frame_) {
    if (!frame_) throw std::runtime_error("StereoFramePointGenerator::compute|called with empty frame");
    check(_hip->ctx, vslam_stereo_new(_hip->ctx), "StereoFramePointGenerator::compute");
    check(_hip->ctx, vslam_get_frame_info(_hip->ctx, 0, &frame_->info), "compute");
    frame_->number_of_points = frame_->info.n_points;
  }
  void setProjectionTrackingDistancePixels(const int32_t& v_) { _projection_tracking_distance_pixels = v_; }
  void setMaximumDescriptorDistanceTracking(const real& v_) { _maximum_descriptor_distance_tracking = v_; }
  const int32_t& numberOfTrackedLandmarks() const { return _number_of_tracked_landmarks; }
  const int32_t& targetNumberOfKeypoints() const { return _target_number_of_keypoints; }
  const vslam_config* parameters() const { return &_hip->config; }
  HipContext* hip() const { return _hip; }

private:
  HipContext* _hip;
  int32_t _projection_tracking_distance_pixels = 0;
  real _maximum_descriptor_distance_tracking = 0;
  int32_t _number_of_tracked_landmarks = 0;
  int32_t _target_number_of_keypoints = 0;
};

class StereoUVAligner {
public:
  explicit StereoUVAligner(HipContext* hip_) : _hip(hip_) {}
  bool enable_inverse_depth_as_information = true;   // AlignerParameters::enable_inverse_depth_as_information
  void configure() {}
  void initialize(const Frame* frame_previous_, const Frame* frame_current_, const TransformMatrix3D& previous_to_current_) {
    _frame_previous = frame_previous_; _frame_current = frame_current_; _previous_to_current = previous_to_current_;
    _number_of_measurements = frame_current_->number_of_points;
  }
  void converge() {
    check(_hip->ctx, vslam_set_tracker_state(_hip->ctx, 0, _frame_current->status, _previous_to_current.m, _window, _tau), "converge");
    check(_hip->ctx, vslam_align(_hip->ctx, enable_inverse_depth_as_information ? 1 : 0), "StereoUVAligner::converge");
    _errors.assign(_number_of_measurements, -1);
    std::vector<uint8_t> inl(_number_of_measurements);
    int32_t n = 0;
    real H[36];
    check(_hip->ctx, vslam_get_aligner_result(_hip->ctx, 0, _number_of_measurements, &n, _errors.data(), inl.data(), _previous_to_current.m, H), "converge");
    _inliers.assign(inl.begin(), inl.end());
    vslam_frame_info info;
    check(_hip->ctx, vslam_get_frame_info(_hip->ctx, 0, &info), "converge");
    _number_of_inliers = info.n_inliers; _number_of_outliers = info.n_outliers; _total_error = info.total_error;
    _has_system_converged = info.aligner_converged != 0;
  }
  //! the tracker-owned window / descriptor distance travel with every state push (the device keeps one state block)
  void setTrackerWindow(int32_t window_, real tau_) { _window = window_; _tau = tau_; }
  const std::vector<real>& errors() const { return _errors; }
  const std::vector<bool>& inliers() const { return _inliers; }
  int32_t numberOfInliers() const { return _number_of_inliers; }
  int32_t numberOfOutliers() const { return _number_of_outliers; }
  real totalError() const { return _total_error; }
  bool hasSystemConverged() const { return _has_system_converged; }
  const TransformMatrix3D& previousToCurrent() const { return _previous_to_current; }
  int32_t minimumNumberOfInliers() const { return _hip->config.aligner_minimum_number_of_inliers; }

private:
  HipContext* _hip;
  const Frame* _frame_previous = nullptr;
  const Frame* _frame_current = nullptr;
  TransformMatrix3D _previous_to_current = TransformMatrix3D::Identity();
  int32_t _number_of_measurements = 0, _number_of_inliers = 0, _number_of_outliers = 0, _window = 0;
  real _total_error = 0, _tau = 0;
  bool _has_system_converged = false;
  std::vector<real> _errors;
  std::vector<bool> _inliers;
};

//! PoseTracker3D (pose_tracker_3d.cpp) with the reference's control flow; owns both plug-ins like the reference
class PoseTracker3D {
public:
  PoseTracker3D(StereoFramePointGenerator* generator_, StereoUVAligner* aligner_) : _framepoint_generator(generator_), _pose_optimizer(aligner_) {}
  ~PoseTracker3D() { delete _framepoint_generator; delete _pose_optimizer; }
  void configure() {
    _previous_to_current_camera = TransformMatrix3D::Identity();
    _projection_tracking_distance_pixels = _framepoint_generator->parameters()->maximum_projection_tracking_distance_pixels;
    _current_descriptor_distance_tracking = _framepoint_generator->parameters()->minimum_descriptor_distance_tracking;
  }
  void setIntensityImageLeft(const uint8_t* image_, int32_t stride_) { _intensity_image_left = image_; _stride = stride_; }
  void setImageSecondary(const uint8_t* image_) { _image_secondary = image_; }
  const Frame& currentFrame() const { return _frames[_current]; }
  Frame::Status status() const { return _status; }

  void compute() {  // :32-222
    const vslam_config& p = *_framepoint_generator->parameters();
    vslam_ctx* ctx = _framepoint_generator->hip()->ctx;
    _number_of_tracked_points = 0;
    const bool has_previous = _has_frame;
    _current ^= 1;
    Frame* current_frame = &_frames[_current];
    Frame* previous_frame = has_previous ? &_frames[_current ^ 1] : nullptr;
    *current_frame = Frame();
    current_frame->intensity_image_left = _intensity_image_left; current_frame->intensity_image_right = _image_secondary;
    current_frame->row_stride = _stride; current_frame->status = _status; current_frame->previous = previous_frame;
    current_frame->camera_left_to_world = _robot_to_world;
    _pushState(ctx, current_frame);
    check(ctx, vslam_set_pose(ctx, 0, current_frame->camera_left_to_world.m), "compute");
    _framepoint_generator->initialize(current_frame);
    if (previous_frame) {
      _track(previous_frame, current_frame, _status == Frame::Localizing);
      if (_status == Frame::Localizing) {
        if (_number_of_tracked_points < p.minimum_number_of_landmarks_to_track) {
          _fallbackEstimate(current_frame, previous_frame);
        } else {
          _pose_optimizer->enable_inverse_depth_as_information = false;
          _runAligner(previous_frame, current_frame);
          if (_pose_optimizer->numberOfInliers() < p.minimum_number_of_landmarks_to_track) _fallbackEstimate(current_frame, previous_frame);
          else _acceptMotion(current_frame, previous_frame);
        }
      } else {
        _registerRecursive(previous_frame, current_frame, 0);
      }
    }
    _robot_to_world = current_frame->camera_left_to_world;
    if (previous_frame) _framepoint_generator->recoverPoints(current_frame);   // _prunePoints + recoverPoints
    else check(ctx, vslam_set_pose(ctx, 0, current_frame->camera_left_to_world.m), "compute");
    check(ctx, vslam_update_points(ctx), "PoseTracker3D::_updatePoints");
    check(ctx, vslam_get_frame_info(ctx, 0, &current_frame->info), "compute");
    _number_of_active_landmarks = current_frame->info.n_active_landmarks;
    if (_number_of_active_landmarks > p.minimum_number_of_landmarks_to_track) _status = Frame::Tracking;
    current_frame->status = _status;
    _pushState(ctx, current_frame);
    _framepoint_generator->compute(current_frame);
    _number_of_tracked_landmarks_previous = _number_of_active_landmarks;
    _has_frame = true;
  }

private:
  void _pushState(vslam_ctx* ctx, const Frame* frame_) {
    _pose_optimizer->setTrackerWindow(_projection_tracking_distance_pixels, _current_descriptor_distance_tracking);
    check(ctx, vslam_set_tracker_state(ctx, 0, _status, _previous_to_current_camera.m, _projection_tracking_distance_pixels,
                                       _current_descriptor_distance_tracking), "PoseTracker3D");
    (void)frame_;
  }
  void _runAligner(Frame* previous_frame_, Frame* current_frame_) {
    _pose_optimizer->setTrackerWindow(_projection_tracking_distance_pixels, _current_descriptor_distance_tracking);
    _pose_optimizer->initialize(previous_frame_, current_frame_, _previous_to_current_camera);
    _pose_optimizer->converge();
  }
  void _track(Frame* previous_frame_, Frame* current_frame_, const bool& track_by_appearance_) {  // :225-298
    const vslam_config& p = *_framepoint_generator->parameters();
    if (track_by_appearance_) _projection_tracking_distance_pixels = p.maximum_projection_tracking_distance_pixels;
    _framepoint_generator->setProjectionTrackingDistancePixels(_projection_tracking_distance_pixels);
    _framepoint_generator->setMaximumDescriptorDistanceTracking(_current_descriptor_distance_tracking);
    int32_t lost = 0;
    _framepoint_generator->track(current_frame_, previous_frame_, _previous_to_current_camera, lost, track_by_appearance_);
    _number_of_tracked_landmarks = _framepoint_generator->numberOfTrackedLandmarks();
    _number_of_tracked_points = current_frame_->number_of_points;
    const real tracking_ratio = static_cast<real>(_number_of_tracked_points) / previous_frame_->number_of_points;
    const real landmark_per_point = static_cast<real>(_number_of_tracked_landmarks) / _number_of_tracked_points;
    const real tracking_success_ratio = static_cast<real>(_number_of_tracked_points) / _framepoint_generator->targetNumberOfKeypoints();
    if (tracking_ratio < p.good_tracking_ratio / 2) {
      if (_projection_tracking_distance_pixels < p.maximum_projection_tracking_distance_pixels)
        _projection_tracking_distance_pixels = std::min(_projection_tracking_distance_pixels * 1 / p.tunnel_vision_ratio,
                                                        static_cast<real>(p.maximum_projection_tracking_distance_pixels));
    } else {
      if (_projection_tracking_distance_pixels > p.minimum_projection_tracking_distance_pixels)
        _projection_tracking_distance_pixels = std::max(_projection_tracking_distance_pixels * p.tunnel_vision_ratio,
                                                        static_cast<real>(p.minimum_projection_tracking_distance_pixels));
    }
    if (tracking_ratio < p.good_tracking_ratio || _number_of_tracked_points < _pose_optimizer->minimumNumberOfInliers() ||
        (landmark_per_point < 0.5 && tracking_success_ratio < 0.25)) {
      _current_descriptor_distance_tracking += 5;
      if (_current_descriptor_distance_tracking > p.maximum_descriptor_distance_tracking) _current_descriptor_distance_tracking = p.maximum_descriptor_distance_tracking;
    } else {
      _current_descriptor_distance_tracking -= 5;
      if (_current_descriptor_distance_tracking < p.minimum_descriptor_distance_tracking) _current_descriptor_distance_tracking = p.minimum_descriptor_distance_tracking;
    }
  }
  void _acceptMotion(Frame* current_frame_, Frame* previous_frame_) {  // :139-159, :372-388
    const vslam_config& p = *_framepoint_generator->parameters();
    const TransformMatrix3D& previous_to_current_camera = _pose_optimizer->previousToCurrent();
    const real delta_angular = previous_to_current_camera.rotationAngle();
    const real delta_translational = previous_to_current_camera.translationNorm();
    if (delta_angular > p.minimum_delta_angular_for_movement || delta_translational > p.minimum_delta_translational_for_movement) {
      _previous_to_current_camera = previous_to_current_camera;
      current_frame_->camera_left_to_world = previous_frame_->camera_left_to_world * _previous_to_current_camera.inverse();
    } else {
      _fallbackEstimate(current_frame_, previous_frame_);
    }
  }
  void _registerRecursive(Frame* previous_frame_, Frame* current_frame_, const int32_t& recursion_) {  // :300-419
    const vslam_config& p = *_framepoint_generator->parameters();
    const real relative = static_cast<real>(_number_of_tracked_landmarks) / _number_of_tracked_landmarks_previous;
    if (_number_of_tracked_landmarks == 0 || relative < 0.1) {
      if (recursion_ < 2) {
        _previous_to_current_camera = TransformMatrix3D::Identity();
        _framepoint_generator->initialize(current_frame_, false);
        _track(previous_frame_, current_frame_, true);
        _registerRecursive(previous_frame_, current_frame_, recursion_ + 1);
      } else {
        breakTrack(current_frame_, previous_frame_);
      }
      return;
    }
    _pose_optimizer->enable_inverse_depth_as_information = true;
    _runAligner(previous_frame_, current_frame_);
    if (_pose_optimizer->numberOfInliers() > p.minimum_number_of_landmarks_to_track) {
      _acceptMotion(current_frame_, previous_frame_);
    } else if (recursion_ < 2) {
      if (_projection_tracking_distance_pixels < p.maximum_projection_tracking_distance_pixels) ++_projection_tracking_distance_pixels;
      _framepoint_generator->initialize(current_frame_, false);
      _track(previous_frame_, current_frame_, false);
      _registerRecursive(previous_frame_, current_frame_, recursion_ + 1);
    } else {
      breakTrack(current_frame_, previous_frame_);
    }
  }
  void breakTrack(Frame* frame_, Frame* previous_frame_) {  // :422-435
    _status = Frame::Localizing;
    frame_->camera_left_to_world = previous_frame_->camera_left_to_world;
    _previous_to_current_camera = TransformMatrix3D::Identity();
    _number_of_tracked_points = 0;
  }
  void _fallbackEstimate(Frame* current_frame_, Frame* previous_frame_) {  // :551-566
    _previous_to_current_camera = TransformMatrix3D::Identity();
    current_frame_->camera_left_to_world = previous_frame_->camera_left_to_world;
  }

  StereoFramePointGenerator* _framepoint_generator;
  StereoUVAligner* _pose_optimizer;
  Frame::Status _status = Frame::Localizing;
  int32_t _number_of_tracked_landmarks = 0, _number_of_tracked_points = 0, _number_of_tracked_landmarks_previous = 0, _number_of_active_landmarks = 0;
  int32_t _projection_tracking_distance_pixels = 0;
  real _current_descriptor_distance_tracking = 0;
  TransformMatrix3D _previous_to_current_camera = TransformMatrix3D::Identity();
  TransformMatrix3D _robot_to_world = TransformMatrix3D::Identity();
  const uint8_t* _intensity_image_left = nullptr;
  const uint8_t* _image_secondary = nullptr;
  int32_t _stride = 0;
  Frame _frames[2];
  int _current = 0;
  bool _has_frame = false;
};

}  // namespace proslam_hip
