// TEST-ONLY declaration stubs (see types/definitions.h): BaseAligner / BaseFrameAligner / AlignerWorkspace<6,4> / StereoUVAligner.
#pragma once
#include "types/frame.h"

namespace proslam {
class BaseAligner {
public:
  BaseAligner(AlignerParameters* parameters_) : _parameters(parameters_) {}
  virtual void configure() {}
  virtual ~BaseAligner() {}
  virtual void linearize(const bool& ignore_outliers_) = 0;
  virtual void oneRound(const bool& ignore_outliers_) = 0;
  virtual void converge() = 0;
  const std::vector<real>& errors() const { return _errors; }
  const std::vector<bool>& inliers() const { return _inliers; }
  const Count numberOfInliers() const { return _number_of_inliers; }
  const Count numberOfOutliers() const { return _number_of_outliers; }
  const Count numberOfCorrespondences() const { return _number_of_measurements; }
  const real totalError() const { return _total_error; }
  const real averageError() const { return _total_error / _number_of_measurements; }
  const bool hasSystemConverged() const { return _has_system_converged; }
  AlignerParameters* parameters() { return _parameters; }
  void setMinimumReliableDepthMeters(const real& d_) { _minimum_reliable_depth_meters = d_; }
  void setMaximumReliableDepthMeters(const real& d_) { _maximum_reliable_depth_meters = d_; }
protected:
  std::vector<real> _errors; std::vector<bool> _inliers;
  Count _number_of_inliers = 0, _number_of_outliers = 0, _number_of_measurements = 0;
  bool _has_system_converged = false; real _total_error = 0;
  real _minimum_reliable_depth_meters = 0.01, _maximum_reliable_depth_meters = 15;
  AlignerParameters* _parameters = 0;
};
template <Count states_, Count dimension_>
class AlignerWorkspace {
protected:
  StubMatrix<states_, states_> _H, _information_matrix = StubMatrix<states_, states_>::Identity();
};
class BaseFrameAligner : public BaseAligner {
public:
  BaseFrameAligner(AlignerParameters* parameters_) : BaseAligner(parameters_) {}
  virtual void initialize(const Frame* frame_previous_, const Frame* frame_current_, const TransformMatrix3D& previous_to_current_) = 0;
  const TransformMatrix3D& previousToCurrent() const { return _previous_to_current; }
protected:
  const Frame* _frame_current = 0; const Frame* _frame_previous = 0;
  TransformMatrix3D _previous_to_current = TransformMatrix3D::Identity();
};
class StereoUVAligner : public BaseFrameAligner, public AlignerWorkspace<6, 4> {
public:
  StereoUVAligner(AlignerParameters* parameters_) : BaseFrameAligner(parameters_) {}
  virtual ~StereoUVAligner() {}
  virtual void initialize(const Frame*, const Frame*, const TransformMatrix3D&) {}
  virtual void linearize(const bool&) {}
  virtual void oneRound(const bool&) {}
  virtual void converge() {}
};
}  // namespace proslam
