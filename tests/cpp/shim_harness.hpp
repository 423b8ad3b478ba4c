// shim_harness.hpp — the caller side of the plug-in boundary for the shim's tests and its throughput bench: what a tracker does
// between the virtual calls (PoseTracker3D::compute, pose_tracker_3d.cpp:32-222: host bookkeeping only; the control flow is
// re-stated here because the reference's tracker cannot be compiled in this image).  Test infrastructure, not product.
#pragma once
#include <chrono>
#include <cmath>
#include <memory>

#include "proslam_hip_plugin.h"

using namespace proslam;

static double rotationAngle(const TransformMatrix3D& T) {   // |Rodrigues(R)|
  const double rx = T.m(2, 1) - T.m(1, 2), ry = T.m(0, 2) - T.m(2, 0), rz = T.m(1, 0) - T.m(0, 1);
  const double s = std::sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25);
  double c = ((T.m(0, 0) + T.m(1, 1)) + T.m(2, 2) - 1) * 0.5;
  c = c > 1 ? 1 : (c < -1 ? -1 : c);
  if (s < 1e-5) return c > 0 ? 0.0 : 3.14159265358979323846;
  return std::acos(c);
}

// the caller side of the plug-in boundary: what a tracker does between the virtual calls (host bookkeeping only)
struct Harness {
  HipStereoFramePointGenerator* generator; HipStereoUVAligner* aligner;
  const PoseTracker3DParameters* p; const StereoFramePointGeneratorParameters* gp; const Camera *left, *right;
  Frame::Status status = Frame::Localizing;
  TransformMatrix3D prior = TransformMatrix3D::Identity(), robot_to_world = TransformMatrix3D::Identity();
  int32_t window; double tau; Count tracked_landmarks = 0, tracked_points = 0, tracked_landmarks_previous = 0, active_landmarks = 0;
  std::vector<std::unique_ptr<Frame>> frames; std::vector<std::unique_ptr<Landmark>> landmarks; FramePointPointerVector lost;
  // wall time per plug-in call of the caller thread (seconds, accumulated): initialize, track, aligner initialize + converge,
  // host prune, recoverPoints, host landmark bookkeeping, compute
  enum { T_INITIALIZE, T_TRACK, T_ALIGN, T_PRUNE, T_RECOVER, T_UPDATE, T_COMPUTE, T_STAGES };
  double seconds[T_STAGES] = {0, 0, 0, 0, 0, 0, 0};
  struct Clock {
    double* acc; std::chrono::steady_clock::time_point t0;
    explicit Clock(double* a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~Clock() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  };

  void track(Frame* previous, Frame* current, bool by_appearance) {
    if (by_appearance) window = gp->maximum_projection_tracking_distance_pixels;
    generator->setProjectionTrackingDistancePixels(window);
    generator->setMaximumDescriptorDistanceTracking(tau);
    { Clock c(&seconds[T_TRACK]); generator->track(current, previous, prior, lost, by_appearance); }
    tracked_landmarks = generator->numberOfTrackedLandmarks();
    tracked_points = (Count)current->points().size();
    const double ratio = (double)tracked_points / previous->points().size();
    const double lm_per_point = (double)tracked_landmarks / tracked_points;
    const double success = (double)tracked_points / generator->targetNumberOfKeypoints();
    if (ratio < p->good_tracking_ratio / 2) { if (window < gp->maximum_projection_tracking_distance_pixels) window = (int32_t)std::min(window * 1 / p->tunnel_vision_ratio, (double)gp->maximum_projection_tracking_distance_pixels); }
    else if (window > gp->minimum_projection_tracking_distance_pixels) window = (int32_t)std::max(window * p->tunnel_vision_ratio, (double)gp->minimum_projection_tracking_distance_pixels);
    if (ratio < p->good_tracking_ratio || tracked_points < aligner->parameters()->minimum_number_of_inliers || (lm_per_point < 0.5 && success < 0.25))
      tau = std::min(tau + 5, gp->maximum_descriptor_distance_tracking);
    else tau = std::max(tau - 5, gp->minimum_descriptor_distance_tracking);
  }
  void fallback(Frame* current, Frame* previous) { prior = TransformMatrix3D::Identity(); current->setRobotToWorld(previous->robotToWorld()); }
  void accept(Frame* current, Frame* previous) {
    const TransformMatrix3D& T = aligner->previousToCurrent();
    const double dt = std::sqrt((T.m(0, 3) * T.m(0, 3) + T.m(1, 3) * T.m(1, 3)) + T.m(2, 3) * T.m(2, 3));
    if (rotationAngle(T) > p->minimum_delta_angular_for_movement || dt > p->minimum_delta_translational_for_movement) {
      prior = T;
      current->setRobotToWorld(previous->cameraLeftToWorld() * prior.inverse());
    } else fallback(current, previous);
  }
  void align(Frame* previous, Frame* current, bool inverse_depth) {
    aligner->parameters()->enable_inverse_depth_as_information = inverse_depth;
    Clock c(&seconds[T_ALIGN]);
    aligner->initialize(previous, current, prior);
    aligner->converge();
  }
  void registerRecursive(Frame* previous, Frame* current, int recursion) {
    const double relative = (double)tracked_landmarks / tracked_landmarks_previous;
    if (tracked_landmarks == 0 || relative < 0.1) {
      if (recursion < 2) { prior = TransformMatrix3D::Identity(); generator->initialize(current, false); track(previous, current, true); registerRecursive(previous, current, recursion + 1); }
      else breakTrack(current, previous);
      return;
    }
    align(previous, current, true);
    if (aligner->numberOfInliers() > p->minimum_number_of_landmarks_to_track) accept(current, previous);
    else if (recursion < 2) {
      if (window < gp->maximum_projection_tracking_distance_pixels) ++window;
      generator->initialize(current, false); track(previous, current, false); registerRecursive(previous, current, recursion + 1);
    } else breakTrack(current, previous);
  }
  void breakTrack(Frame* current, Frame* previous) { status = Frame::Localizing; current->setRobotToWorld(previous->robotToWorld()); prior = TransformMatrix3D::Identity(); tracked_points = 0; }
  void prune(Frame* frame) {   // the selection rule of _prunePoints on the aligner's result members
    Count kept = 0;
    FramePointPointerVector& pts = frame->points();
    const bool good = aligner->averageError() < aligner->parameters()->maximum_error_kernel;
    for (Index i = 0; i < pts.size(); ++i) {
      const bool keep = good ? (bool)aligner->inliers()[i] : (aligner->errors()[i] != -1 && aligner->errors()[i] < 100 * aligner->parameters()->maximum_error_kernel);
      if (keep) pts[kept++] = pts[i]; else pts[i]->clear();
    }
    pts.resize(kept);
    tracked_points = kept;
  }
  void updatePoints(Frame* frame) {   // which points carry a landmark after this frame (the optimisation itself runs on the device)
    active_landmarks = 0;
    for (FramePoint* point : frame->points()) {
      if (point->trackLength() < p->minimum_track_length_for_landmark_creation) continue;
      Landmark* landmark = point->origin()->landmark();
      if (!landmark) { landmarks.emplace_back(new Landmark()); landmark = landmarks.back().get(); for (FramePoint* q = point; q; q = q->previous()) { q->setLandmark(landmark); ++landmark->updates; } }
      else { point->setLandmark(landmark); ++landmark->updates; }
      ++active_landmarks;
    }
  }
  Frame* step(uint8_t* L, uint8_t* R, int rows, int cols, size_t row_stride = 0) {
    if (!row_stride) row_stride = (size_t)cols;
    Frame* previous = frames.empty() ? nullptr : frames.back().get();
    frames.emplace_back(new Frame(previous, robot_to_world));
    Frame* current = frames.back().get();
    current->setCameraLeft(left); current->setCameraRight(right);
    current->setIntensityImageLeft(cv::Mat(rows, cols, CV_8UC1, L, row_stride)); current->setIntensityImageRight(cv::Mat(rows, cols, CV_8UC1, R, row_stride));
    current->setStatus(status);
    tracked_points = 0;
    { Clock c(&seconds[T_INITIALIZE]); generator->initialize(current); }
    if (previous) {
      track(previous, current, status == Frame::Localizing);
      if (status == Frame::Localizing) {
        if (tracked_points < p->minimum_number_of_landmarks_to_track) fallback(current, previous);
        else { align(previous, current, false); if (aligner->numberOfInliers() < p->minimum_number_of_landmarks_to_track) fallback(current, previous); else accept(current, previous); }
      } else registerRecursive(previous, current, 0);
    }
    robot_to_world = current->robotToWorld();
    if (previous) {
      { Clock c(&seconds[T_PRUNE]); prune(current); }
      if (p->enable_landmark_recovery) { Clock c(&seconds[T_RECOVER]); generator->recoverPoints(current, lost); tracked_points = (Count)current->points().size(); }
    }
    { Clock c(&seconds[T_UPDATE]); updatePoints(current); }
    if (active_landmarks > p->minimum_number_of_landmarks_to_track) status = Frame::Tracking;
    { Clock c(&seconds[T_COMPUTE]); generator->compute(current); }
    current->setStatus(status);
    tracked_landmarks_previous = active_landmarks;
    // frames are kept for the whole run: origin() / previous() chains of the tracks reach back into them
    return current;
  }
};

