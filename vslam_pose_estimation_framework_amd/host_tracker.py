"""Host-side mirror of the reference's PoseTracker3D (src/position_tracking/pose_tracker_3d.{h,cpp}) on top of
the stage-granular C ABI — the control flow the shim (shim/proslam_hip_plugin.h) leaves to the reference's own
tracker: one C call per plug-in virtual (initialize / track / converge / recoverPoints / compute).

Same method names and decision logic as the reference; used by the tests to show that the staged boundary and
the fused device path (`vslam_process_*`) are the same computation.  Single stream (the drop-in case)."""
import ctypes as C
import math

import numpy as np

from .capi import LOCALIZING, TRACKING


def _identity():
    return [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]


def _mul(A, B):
    out = [0.0] * 12
    for i in range(3):
        for j in range(3):
            out[4 * i + j] = (A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j]) + A[4 * i + 2] * B[8 + j]
        out[4 * i + 3] = ((A[4 * i] * B[3] + A[4 * i + 1] * B[7]) + A[4 * i + 2] * B[11]) + A[4 * i + 3]
    return out


def _inverse(A):
    out = [0.0] * 12
    for i in range(3):
        for j in range(3):
            out[4 * i + j] = A[4 * j + i]
    for i in range(3):
        out[4 * i + 3] = -((out[4 * i] * A[3] + out[4 * i + 1] * A[7]) + out[4 * i + 2] * A[11])
    return out


def _rotation_angle(T):
    rx, ry, rz = T[9] - T[6], T[2] - T[8], T[4] - T[1]
    s = math.sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25)
    c = ((T[0] + T[5]) + T[10] - 1) * 0.5
    c = max(-1.0, min(1.0, c))
    if s < 1e-5:
        return 0.0 if c > 0 else math.pi
    return math.acos(c)


def _div(a, b):
    """C++ double division semantics (inf / nan instead of ZeroDivisionError)."""
    if b == 0:
        return float("nan") if a == 0 else math.copysign(float("inf"), a)
    return a / b


class PoseTracker3D(object):
    """pose_tracker_3d.h:14-134 — state and methods named as in the reference."""

    def __init__(self, api):
        self.api = api
        self.cfg = api.cfg
        self.configure()

    def _target_number_of_keypoints(self):
        c = self.cfg
        return (c.cols // c.bin_size_pixels + 1) * (c.rows // c.bin_size_pixels + 1)

    def configure(self):  # :11-21
        c = self.cfg
        self._status = LOCALIZING
        self._previous_to_current_camera = _identity()
        self._projection_tracking_distance_pixels = c.maximum_projection_tracking_distance_pixels
        self._current_descriptor_distance_tracking = c.minimum_descriptor_distance_tracking
        self._number_of_tracked_landmarks_previous = 0
        self._number_of_tracked_landmarks = 0
        self._number_of_tracked_points = 0
        self._number_of_active_landmarks = 0
        self._robot_to_world = _identity()          # WorldMap::robot_to_world
        self._previous_pose = None                  # previous_frame->cameraLeftToWorld()
        self._previous_points = 0                   # previous_frame->points().size()
        self._frame_pose = _identity()
        self.fallback = 0
        self.track_broken = 0

    # -- helpers ---------------------------------------------------------------------------------------
    def _push_state(self):
        prior = (C.c_double * 12)(*self._previous_to_current_camera)
        self.api.check(self.api.fn("set_tracker_state")(self.api.ctx, C.c_int(0), C.c_int(self._status), prior,
                                                        C.c_int(self._projection_tracking_distance_pixels),
                                                        C.c_double(self._current_descriptor_distance_tracking)))

    def _set_frame_pose(self, pose):
        self._frame_pose = list(pose)
        self.api.check(self.api.fn("set_pose")(self.api.ctx, C.c_int(0), (C.c_double * 12)(*pose)))

    # -- PoseTracker3D::compute (:32-222) -----------------------------------------------------------------
    def compute(self, left, right):
        api, c = self.api, self.cfg
        self.fallback = 0
        self.track_broken = 0
        self._number_of_tracked_points = 0
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        stride = left.shape[1]
        has_previous = self._previous_pose is not None
        self._push_state()                                  # frame created with the tracker status
        self._set_frame_pose(self._robot_to_world)
        # _framepoint_generator->initialize(current_frame)
        api.check(api.fn("frame_begin")(api.ctx, left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p),
                                        C.c_int32(stride), C.c_size_t(left.shape[0] * stride), C.c_int(0)))
        if has_previous:
            self._track(self._status == LOCALIZING)
            if self._status == LOCALIZING:
                if self._number_of_tracked_points < c.minimum_number_of_landmarks_to_track:
                    self._fallbackEstimate()
                else:
                    inliers, T = self._align(False)
                    if inliers < c.minimum_number_of_landmarks_to_track:
                        self._fallbackEstimate()
                    else:
                        self._acceptMotion(T)
            else:
                self._registerRecursive(0)
        self._robot_to_world = list(self._frame_pose)
        self._set_frame_pose(self._frame_pose)
        if has_previous:
            api.check(api.fn("prune_recover")(api.ctx))     # _prunePoints + recoverPoints
        api.check(api.fn("update_points")(api.ctx))          # _updatePoints
        self._number_of_active_landmarks = api.frame_info(0).n_active_landmarks
        if self._number_of_active_landmarks > c.minimum_number_of_landmarks_to_track:
            self._status = TRACKING
        self._push_state()
        api.check(api.fn("stereo_new")(api.ctx))             # _framepoint_generator->compute(current_frame)
        self._number_of_tracked_landmarks_previous = self._number_of_active_landmarks
        fi = api.frame_info(0)
        self._previous_pose = list(self._frame_pose)
        self._previous_points = fi.n_points
        return fi

    # -- _track (:225-298) ------------------------------------------------------------------------------------
    def _track(self, track_by_appearance):
        api, c = self.api, self.cfg
        if track_by_appearance:
            self._projection_tracking_distance_pixels = c.maximum_projection_tracking_distance_pixels
        self._push_state()    # setProjectionTrackingDistancePixels / setMaximumDescriptorDistanceTracking
        api.check(api.fn("track")(api.ctx, C.c_int(1 if track_by_appearance else 0)))
        fi = api.frame_info(0)
        self._number_of_tracked_landmarks = fi.n_tracked_landmarks
        self._number_of_tracked_points = fi.n_tracked
        tracking_ratio = _div(float(self._number_of_tracked_points), float(self._previous_points))
        landmark_per_point = _div(float(self._number_of_tracked_landmarks), float(self._number_of_tracked_points))
        success_ratio = float(self._number_of_tracked_points) / self._target_number_of_keypoints()
        wmax, wmin = c.maximum_projection_tracking_distance_pixels, c.minimum_projection_tracking_distance_pixels
        w = self._projection_tracking_distance_pixels
        if tracking_ratio < c.good_tracking_ratio / 2:
            if w < wmax:
                w = int(min(w * 1 / c.tunnel_vision_ratio, float(wmax)))
        else:
            if w > wmin:
                w = int(max(w * c.tunnel_vision_ratio, float(wmin)))
        self._projection_tracking_distance_pixels = w
        if (tracking_ratio < c.good_tracking_ratio or self._number_of_tracked_points < c.aligner_minimum_number_of_inliers
                or (landmark_per_point < 0.5 and success_ratio < 0.25)):
            self._current_descriptor_distance_tracking = min(self._current_descriptor_distance_tracking + 5,
                                                             c.maximum_descriptor_distance_tracking)
        else:
            self._current_descriptor_distance_tracking = max(self._current_descriptor_distance_tracking - 5,
                                                             c.minimum_descriptor_distance_tracking)

    def _align(self, enable_inverse_depth_as_information):
        api = self.api
        self._push_state()   # _pose_optimizer->initialize(previous, current, _previous_to_current_camera)
        api.check(api.fn("align")(api.ctx, C.c_int(1 if enable_inverse_depth_as_information else 0)))
        fi = api.frame_info(0)
        T = (C.c_double * 12)()
        n = C.c_int32()
        api.check(api.fn("get_aligner_result")(api.ctx, C.c_int(0), C.c_int32(self.cfg.max_points), C.byref(n), None, None, T, None))
        return fi.n_inliers, list(T)

    def _acceptMotion(self, T):  # :139-159, :372-388
        c = self.cfg
        delta_angular = _rotation_angle(T)
        delta_translational = math.sqrt((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11])
        if delta_angular > c.minimum_delta_angular_for_movement or delta_translational > c.minimum_delta_translational_for_movement:
            self._previous_to_current_camera = list(T)
            self._frame_pose = _mul(self._previous_pose, _inverse(self._previous_to_current_camera))
        else:
            self._fallbackEstimate()

    # -- _registerRecursive (:300-419) ----------------------------------------------------------------------------
    def _registerRecursive(self, recursion):
        api, c = self.api, self.cfg
        relative = _div(float(self._number_of_tracked_landmarks), float(self._number_of_tracked_landmarks_previous))
        if self._number_of_tracked_landmarks == 0 or relative < 0.1:
            if recursion < 2:
                self._previous_to_current_camera = _identity()
                api.check(api.fn("frame_restore")(api.ctx))   # initialize(current_frame, false)
                self._track(True)
                self._registerRecursive(recursion + 1)
            else:
                self.breakTrack()
            return
        inliers, T = self._align(True)
        if inliers > c.minimum_number_of_landmarks_to_track:
            self._acceptMotion(T)
        else:
            if recursion < 2:
                if self._projection_tracking_distance_pixels < c.maximum_projection_tracking_distance_pixels:
                    self._projection_tracking_distance_pixels += 1
                api.check(api.fn("frame_restore")(api.ctx))
                self._track(False)
                self._registerRecursive(recursion + 1)
            else:
                self.breakTrack()

    def _fallbackEstimate(self):  # :551-566
        self._previous_to_current_camera = _identity()
        self._frame_pose = list(self._previous_pose)
        self.fallback = 1

    def breakTrack(self):  # :422-435
        self._status = LOCALIZING
        self._frame_pose = list(self._previous_pose)
        self._previous_to_current_camera = _identity()
        self._number_of_tracked_points = 0
        self.track_broken = 1
