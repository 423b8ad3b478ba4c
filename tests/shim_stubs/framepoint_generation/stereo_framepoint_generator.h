// TEST-ONLY declaration stubs (see types/definitions.h): BaseFramePointGenerator / StereoFramePointGenerator.
#pragma once
#include "types/frame.h"

namespace proslam {
class BaseFramePointGenerator {
public:
  BaseFramePointGenerator(BaseFramePointGeneratorParameters* parameters_) : _parameters(parameters_) {}
  virtual void configure() {
    _number_of_rows_image = (int32_t)_camera_left->numberOfImageRows(); _number_of_cols_image = (int32_t)_camera_left->numberOfImageCols();
    _target_number_of_keypoints = (Count)((_number_of_cols_image / (int32_t)_parameters->bin_size_pixels + 1) * (_number_of_rows_image / (int32_t)_parameters->bin_size_pixels + 1));
  }
  virtual ~BaseFramePointGenerator() {}
  BaseFramePointGeneratorParameters* parameters() { return _parameters; }
  virtual void initialize(Frame* frame_, const bool& extract_features_ = true) = 0;
  virtual void compute(Frame* frame_) = 0;
  virtual void track(Frame* frame_, Frame* frame_previous_, const TransformMatrix3D& camera_left_previous_in_current_,
                     FramePointPointerVector& lost_points_, const bool track_by_appearance_ = true) = 0;
  virtual void recoverPoints(Frame* current_frame_, const FramePointPointerVector& lost_points_) const = 0;
  void setCameraLeft(const Camera* camera_left_) { _camera_left = camera_left_; }
  const Count& targetNumberOfKeypoints() const { return _target_number_of_keypoints; }
  void setProjectionTrackingDistancePixels(const int32_t& d_) { _projection_tracking_distance_pixels = d_; }
  void setMaximumDescriptorDistanceTracking(const real& d_) { _maximum_descriptor_distance_tracking = d_; }
  const Count& numberOfDetectedKeypoints() const { return _number_of_detected_keypoints; }
  const Count& numberOfTrackedLandmarks() const { return _number_of_tracked_landmarks; }
protected:
  const Camera* _camera_left = nullptr;
  int32_t _number_of_rows_image = 0, _number_of_cols_image = 0;
  Count _target_number_of_keypoints = 0, _number_of_detected_keypoints = 0;
  int32_t _projection_tracking_distance_pixels = 0; real _maximum_descriptor_distance_tracking = 0;
  Count _number_of_tracked_landmarks = 0;
  CREATE_CHRONOMETER(keypoint_detection)       // base_framepoint_generator.h:232-233
  CREATE_CHRONOMETER(descriptor_extraction)
private:
  BaseFramePointGeneratorParameters* _parameters;
};
class StereoFramePointGenerator : public BaseFramePointGenerator {
public:
  StereoFramePointGenerator(StereoFramePointGeneratorParameters* parameters_) : BaseFramePointGenerator(parameters_), _parameters(parameters_) {}
  virtual void configure() { BaseFramePointGenerator::configure(); }
  virtual ~StereoFramePointGenerator() {}
  StereoFramePointGeneratorParameters* parameters() { return _parameters; }
  virtual void initialize(Frame*, const bool& = true) override {}
  virtual void compute(Frame*) override {}
  virtual void track(Frame*, Frame*, const TransformMatrix3D&, FramePointPointerVector&, const bool = true) override {}
  virtual void recoverPoints(Frame*, const FramePointPointerVector&) const override {}
  void setCameraRight(const Camera* camera_right_) { _camera_right = camera_right_; }
protected:
  const Camera* _camera_right = nullptr;
  CREATE_CHRONOMETER(point_triangulation)      // stereo_framepoint_generator.h:81
private:
  StereoFramePointGeneratorParameters* _parameters;
};
}  // namespace proslam
