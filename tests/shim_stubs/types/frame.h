// TEST-ONLY declaration stubs (see definitions.h): Camera, IntensityFeature, FramePoint, Frame as the shim sees them.
#pragma once
#include "definitions.h"

namespace proslam {
class Frame;
class Landmark { public: Count updates = 0; };   // the shim only tests FramePoint::landmark() for null

class Camera {
public:
  Camera(const Count& rows_, const Count& cols_, const CameraMatrix& camera_matrix_) : _rows(rows_), _cols(cols_), _camera_matrix(camera_matrix_) {}
  const Count& numberOfImageRows() const { return _rows; }
  const Count& numberOfImageCols() const { return _cols; }
  const CameraMatrix& cameraMatrix() const { return _camera_matrix; }
  const Vector3& baselineHomogeneous() const { return _baseline_homogeneous; }
  void setBaselineHomogeneous(const Vector3& b_) { _baseline_homogeneous = b_; }
  const TransformMatrix3D& cameraToRobot() const { return _camera_to_robot; }
  const TransformMatrix3D& robotToCamera() const { return _robot_to_camera; }
private:
  Count _rows, _cols; CameraMatrix _camera_matrix; Vector3 _baseline_homogeneous;
  TransformMatrix3D _camera_to_robot = TransformMatrix3D::Identity(), _robot_to_camera = TransformMatrix3D::Identity();
};

struct IntensityFeature {
  IntensityFeature() : row(0), col(0), index_in_vector(0) {}
  IntensityFeature(const cv::KeyPoint& keypoint_, const cv::Mat& descriptor_, const size_t& index_in_vector_)
      : keypoint(keypoint_), descriptor(descriptor_), row((int32_t)keypoint_.pt.y), col((int32_t)keypoint_.pt.x), index_in_vector(index_in_vector_) {}
  cv::KeyPoint keypoint; cv::Mat descriptor; int32_t row, col; size_t index_in_vector;
};

class FramePoint {
protected:
  FramePoint(const IntensityFeature* l_, const IntensityFeature* r_, const real& d_, Frame* frame_)
      : _keypoint_left(l_->keypoint), _keypoint_right(r_->keypoint), _descriptor_left(l_->descriptor), _descriptor_right(r_->descriptor),
        _descriptor_distance_triangulation(d_), _frame(frame_) { _origin = this; }
public:
  FramePoint* previous() const { return _previous; }
  FramePoint* next() const { return _next; }
  void setPrevious(FramePoint* previous_) { previous_->_next = this; _previous = previous_; _track_length = previous_->_track_length + 1; _origin = previous_->_origin; }
  void clear() { if (_previous) { _previous->_next = nullptr; _previous = nullptr; } if (_next) { _next->_previous = nullptr; if (_next->_origin == this) _next->_origin = _next; }
                 _landmark = nullptr; _next = nullptr; _track_length = 0; _origin = this; }
  FramePoint* origin() { return _origin; }
  Landmark* landmark() { return _landmark; }
  const Landmark* landmark() const { return _landmark; }
  void setLandmark(Landmark* l_) { _landmark = l_; }
  const Count trackLength() const { return _track_length; }
  void setEpipolarOffset(const int32_t& e_) { _epipolar_offset = e_; }
  const int32_t& epipolarOffset() const { return _epipolar_offset; }
  const PointCoordinates cameraCoordinatesLeft() const { return _camera_coordinates_left; }
  void setCameraCoordinatesLeft(const PointCoordinates& c_) { _camera_coordinates_left = c_; }
  const cv::KeyPoint& keypointLeft() const { return _keypoint_left; }
  const cv::KeyPoint& keypointRight() const { return _keypoint_right; }
  const cv::Mat& descriptorLeft() const { return _descriptor_left; }
  const cv::Mat& descriptorRight() const { return _descriptor_right; }
  const real& descriptorDistanceTriangulation() const { return _descriptor_distance_triangulation; }
protected:
  const cv::KeyPoint _keypoint_left, _keypoint_right; const cv::Mat _descriptor_left, _descriptor_right;
  real _descriptor_distance_triangulation; Frame* _frame;
  FramePoint *_previous = nullptr, *_next = nullptr, *_origin = nullptr; Landmark* _landmark = nullptr;
  Count _track_length = 0; int32_t _epipolar_offset = 0; PointCoordinates _camera_coordinates_left;
  friend Frame;
};
typedef std::vector<FramePoint*> FramePointPointerVector;

class Frame {
public:
  enum Status {Localizing, Tracking};
  Frame(Frame* previous_, const TransformMatrix3D& robot_to_world_) : _previous(previous_) { setRobotToWorld(robot_to_world_); }
  ~Frame() { for (FramePoint* p : _created_points) delete p; }
  Frame* previous() { return _previous; }
  std::vector<cv::KeyPoint>& keypointsLeft() { return _keypoints_left; }
  std::vector<cv::KeyPoint>& keypointsRight() { return _keypoints_right; }
  cv::Mat& descriptorsLeft() { return _descriptors_left; }
  cv::Mat& descriptorsRight() { return _descriptors_right; }
  const Camera* cameraLeft() const { return _camera_left; }
  void setCameraLeft(const Camera* c_) { _camera_left = c_; }
  const Camera* cameraRight() const { return _camera_right; }
  void setCameraRight(const Camera* c_) { _camera_right = c_; }
  const TransformMatrix3D& robotToWorld() const { return _robot_to_world; }
  void setRobotToWorld(const TransformMatrix3D& T_, const bool = false) { _robot_to_world = T_; _camera_left_to_world = T_; _world_to_camera_left = T_.inverse(); }
  const TransformMatrix3D& cameraLeftToWorld() const { return _camera_left_to_world; }
  const TransformMatrix3D& worldToCameraLeft() const { return _world_to_camera_left; }
  const FramePointPointerVector& points() const { return _active_points; }
  FramePointPointerVector& points() { return _active_points; }
  FramePoint* createFramepoint(const IntensityFeature* l_, const IntensityFeature* r_, const real& d_, const PointCoordinates& c_, FramePoint* previous_point_ = nullptr) {
    FramePoint* p = new FramePoint(l_, r_, d_, this);
    p->setCameraCoordinatesLeft(c_);
    if (previous_point_) p->setPrevious(previous_point_);
    _created_points.push_back(p);
    return p;
  }
  const cv::Mat& intensityImageLeft() const { return _intensity_image_left; }
  void setIntensityImageLeft(const cv::Mat i_) { _intensity_image_left = i_; }
  const cv::Mat& intensityImageRight() const { return _intensity_image_right; }
  void setIntensityImageRight(const cv::Mat i_) { _intensity_image_right = i_; }
  const Status& status() const { return _status; }
  void setStatus(const Status& s_) { _status = s_; }
  void setAverageDescriptorDistanceTracking(const real& d_) { _average_descriptor_distance = d_; }
  const real& averageDescriptorDistanceTracking() const { return _average_descriptor_distance; }
private:
  Status _status = Localizing; Frame* _previous = nullptr;
  std::vector<cv::KeyPoint> _keypoints_left, _keypoints_right; cv::Mat _descriptors_left, _descriptors_right;
  real _average_descriptor_distance = 0; FramePointPointerVector _created_points, _active_points;
  TransformMatrix3D _robot_to_world, _camera_left_to_world, _world_to_camera_left;
  const Camera *_camera_left = nullptr, *_camera_right = nullptr; cv::Mat _intensity_image_left, _intensity_image_right;
};
}  // namespace proslam
