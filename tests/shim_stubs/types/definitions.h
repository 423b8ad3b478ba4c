// TEST-ONLY declaration stubs.  The shim (shim/proslam_hip_plugin.h) is written against the reference's headers
// (OpenCV 3, Eigen, srrg_core), none of which exist in this image; these files declare — with the reference's names and
// signatures, and nothing else — the few members of those interfaces the shim touches, so that a drift between the shim
// and the interface it subclasses fails the CPU test suite and so that the shim can be driven on the GPU box.  They are
// not shipped, not an oracle/_ref build, and implement no algorithm of the reference: cv::Mat is a byte matrix, the
// "Eigen" types are fixed-size arrays with operator().
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8UC1 0
namespace cv {
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct KeyPoint {
  Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x_, float y_, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
      : pt(x_, y_), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
struct MatStep { size_t v = 0; operator size_t() const { return v; } };
class Mat {   // 8-bit single-channel matrix with shared storage (header copies and row() alias the data, as cv::Mat does)
public:
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  MatStep step;
  Mat() {}
  Mat(int rows_, int cols_, int /*type*/) : rows(rows_), cols(cols_), _store(new std::vector<uint8_t>((size_t)rows_ * cols_, 0)) { data = _store->data(); step.v = (size_t)cols_; }
  Mat(int rows_, int cols_, int /*type*/, void* data_, size_t step_) : rows(rows_), cols(cols_), data((uint8_t*)data_) { step.v = step_; }
  Mat row(int r) const { Mat m; m.rows = 1; m.cols = cols; m.data = data + (size_t)r * step.v; m.step = step; m._store = _store; return m; }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step.v); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step.v); }
  void release() { _store.reset(); data = nullptr; rows = cols = 0; }
private:
  std::shared_ptr<std::vector<uint8_t>> _store;
};
}  // namespace cv

// definitions.h:143-146 (timing): the members behind getTimeConsumptionSeconds_*() are protected
#define CREATE_CHRONOMETER(NAME) \
  protected: double _time_consumption_seconds_##NAME = 0; \
  public: const double getTimeConsumptionSeconds_##NAME() const {return _time_consumption_seconds_##NAME;}

namespace proslam {
typedef double real;
typedef uint32_t Identifier;
typedef uint32_t Index;
typedef uint32_t Count;

template <int R, int C>
struct StubMatrix {
  real v[R * C];
  StubMatrix() { std::memset(v, 0, sizeof v); }
  StubMatrix(real a, real b, real c) { static_assert(R * C == 3, "3-vector"); v[0] = a; v[1] = b; v[2] = c; }
  real& operator()(int i, int j) { return v[i * C + j]; }
  const real& operator()(int i, int j) const { return v[i * C + j]; }
  real& operator()(int i) { return v[i]; }
  const real& operator()(int i) const { return v[i]; }
  real x() const { return v[0]; }
  real y() const { return v[1]; }
  real z() const { return v[2]; }
  static StubMatrix Zero() { return StubMatrix(); }
  static StubMatrix Identity() { StubMatrix m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1; return m; }
};
typedef StubMatrix<3, 1> PointCoordinates;
typedef StubMatrix<3, 1> ImageCoordinates;
typedef StubMatrix<3, 1> Vector3;
typedef StubMatrix<3, 3> Matrix3;
typedef StubMatrix<3, 3> CameraMatrix;
typedef StubMatrix<6, 6> Matrix6;

struct TransformMatrix3D {   // Eigen::Transform<real, 3, Eigen::Isometry>: 3x4 [R|t]
  StubMatrix<3, 4> m;
  StubMatrix<3, 4>& matrix() { return m; }
  const StubMatrix<3, 4>& matrix() const { return m; }
  static TransformMatrix3D Identity() { TransformMatrix3D T; T.m(0, 0) = T.m(1, 1) = T.m(2, 2) = 1; return T; }
  TransformMatrix3D operator*(const TransformMatrix3D& B) const {
    TransformMatrix3D C;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) C.m(i, j) = (m(i, 0) * B.m(0, j) + m(i, 1) * B.m(1, j)) + m(i, 2) * B.m(2, j);
      C.m(i, 3) = ((m(i, 0) * B.m(0, 3) + m(i, 1) * B.m(1, 3)) + m(i, 2) * B.m(2, 3)) + m(i, 3);
    }
    return C;
  }
  PointCoordinates operator*(const PointCoordinates& p) const {
    PointCoordinates q;
    for (int i = 0; i < 3; ++i) q(i) = ((m(i, 0) * p(0) + m(i, 1) * p(1)) + m(i, 2) * p(2)) + m(i, 3);
    return q;
  }
  TransformMatrix3D inverse() const {
    TransformMatrix3D C;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m(i, j) = m(j, i);
    for (int i = 0; i < 3; ++i) C.m(i, 3) = -((C.m(i, 0) * m(0, 3) + C.m(i, 1) * m(1, 3)) + C.m(i, 2) * m(2, 3));
    return C;
  }
};

// parameters.h: the members the shim maps into vslam_config (defaults as in the header)
struct AlignerParameters {
  real error_delta_for_convergence = 1e-5, maximum_error_kernel = 10, damping = 0;
  Count maximum_number_of_iterations = 1000, minimum_number_of_inliers = 100;
  bool enable_inverse_depth_as_information = true;
};
struct LandmarkParameters { real maximum_error_squared_meters = 25; Count maximum_number_of_iterations = 100; };
struct BaseFramePointGeneratorParameters {
  std::string detector_type = "FAST", descriptor_type = "ORB";
  real target_number_of_keypoints_tolerance = 0.1; uint32_t detector_threshold_minimum = 20, detector_threshold_maximum = 100;
  real detector_threshold_maximum_change = 0.1; uint32_t number_of_detectors_vertical = 1, number_of_detectors_horizontal = 1;
  int32_t minimum_projection_tracking_distance_pixels = 15, maximum_projection_tracking_distance_pixels = 50;
  real minimum_descriptor_distance_tracking = 25.6, maximum_descriptor_distance_tracking = 51.2;
  real maximum_reliable_depth_meters = 15, maximum_depth_meters = 1000, minimum_depth_meters = 0.1;
  bool enable_keypoint_binning = true; Count bin_size_pixels = 15;
};
struct StereoFramePointGeneratorParameters : BaseFramePointGeneratorParameters {
  real maximum_matching_distance_triangulation = 51.2, minimum_disparity_pixels = 1; int32_t maximum_epipolar_search_offset_pixels = 0;
};
struct PoseTracker3DParameters {
  Count minimum_track_length_for_landmark_creation = 2, minimum_number_of_landmarks_to_track = 10;
  real tunnel_vision_ratio = 0.75, good_tracking_ratio = 0.3; bool enable_landmark_recovery = true;
  real minimum_delta_angular_for_movement = 0.001, minimum_delta_translational_for_movement = 0.01;
  AlignerParameters* aligner = nullptr;
};
}  // namespace proslam
