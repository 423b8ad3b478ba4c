"""ctypes view of include/vslam_hip.h.

`CApi(path, prefix)` binds one shared library that exports the C ABI with the given symbol
prefix.  The product binds `libvslam_hip.so` with prefix ``vslam_`` (see ``hip.py``); the test
suite binds the CPU oracle with prefix ``orc_`` through ``tests/_oracle.py``.  Nothing in this
package loads the oracle.
"""
import ctypes as C
import numpy as np

MAX_REGIONS = 16

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_STATE = 0, -1, -2, -3, -4, -5
LOCALIZING, TRACKING = 0, 1


class Config(C.Structure):
    """struct vslam_config (include/vslam_hip.h)."""
    _fields_ = [
        ("rows", C.c_int32), ("cols", C.c_int32),
        ("K", C.c_double * 9), ("baseline_h", C.c_double * 3),
        ("det_rows", C.c_int32), ("det_cols", C.c_int32),
        ("detector_threshold_minimum", C.c_int32), ("detector_threshold_maximum", C.c_int32),
        ("detector_threshold_maximum_change", C.c_double),
        ("target_number_of_keypoints_tolerance", C.c_double),
        ("bin_size_pixels", C.c_int32), ("enable_keypoint_binning", C.c_int32),
        ("minimum_projection_tracking_distance_pixels", C.c_int32),
        ("maximum_projection_tracking_distance_pixels", C.c_int32),
        ("minimum_descriptor_distance_tracking", C.c_double),
        ("maximum_descriptor_distance_tracking", C.c_double),
        ("maximum_reliable_depth_meters", C.c_double),
        ("maximum_depth_meters", C.c_double),
        ("minimum_depth_meters", C.c_double),
        ("maximum_matching_distance_triangulation", C.c_double),
        ("minimum_disparity_pixels", C.c_double),
        ("maximum_epipolar_search_offset_pixels", C.c_int32),
        ("minimum_track_length_for_landmark_creation", C.c_int32),
        ("minimum_number_of_landmarks_to_track", C.c_int32),
        ("tunnel_vision_ratio", C.c_double), ("good_tracking_ratio", C.c_double),
        ("enable_landmark_recovery", C.c_int32),
        ("minimum_delta_angular_for_movement", C.c_double),
        ("minimum_delta_translational_for_movement", C.c_double),
        ("aligner_error_delta_for_convergence", C.c_double),
        ("aligner_maximum_error_kernel", C.c_double),
        ("aligner_damping", C.c_double),
        ("aligner_maximum_number_of_iterations", C.c_int32),
        ("aligner_minimum_number_of_inliers", C.c_int32),
        ("landmark_maximum_error_squared_meters", C.c_double),
        ("landmark_maximum_number_of_iterations", C.c_int32),
        ("max_keypoints", C.c_int32), ("max_points", C.c_int32), ("max_history_frames", C.c_int32),
        ("descriptor_type", C.c_int32),
    ]

    def copy(self):
        c = Config()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Config))
        return c


class FrameInfo(C.Structure):
    """struct vslam_frame_info (include/vslam_hip.h)."""
    _fields_ = [
        ("frame_index", C.c_int32), ("status", C.c_int32), ("status_at_start", C.c_int32),
        ("n_keypoints_left", C.c_int32), ("n_keypoints_right", C.c_int32),
        ("n_detected_left", C.c_int32), ("n_detected_right", C.c_int32),
        ("thresholds", C.c_int32 * MAX_REGIONS),
        ("track_attempts", C.c_int32), ("n_tracked", C.c_int32), ("n_lost", C.c_int32),
        ("n_tracked_landmarks", C.c_int32), ("aligner_ran", C.c_int32),
        ("aligner_iterations", C.c_int32), ("aligner_converged", C.c_int32),
        ("n_inliers", C.c_int32), ("n_outliers", C.c_int32), ("total_error", C.c_double),
        ("n_after_prune", C.c_int32), ("n_recovered", C.c_int32),
        ("n_active_landmarks", C.c_int32), ("n_new_stereo", C.c_int32), ("n_points", C.c_int32),
        ("track_broken", C.c_int32), ("fallback", C.c_int32), ("window_pixels", C.c_int32),
        ("error_flags", C.c_int32), ("tau_track", C.c_double), ("tau_triangulation", C.c_double),
        ("camera_left_to_world", C.c_double * 12), ("previous_to_current", C.c_double * 12),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


class KeypointsView(C.Structure):
    """struct vslam_keypoints_view (stage views: pointers into the context's pinned report buffer, valid until the next view call)."""
    _fields_ = [("n", C.c_int32 * 2), ("xy", C.POINTER(C.c_int16) * 2), ("score", C.POINTER(C.c_uint8) * 2), ("desc", C.POINTER(C.c_uint8) * 2)]


class TrackView(C.Structure):
    _fields_ = [("n_tracked", C.c_int32), ("n_lost", C.c_int32), ("n_tracked_landmarks", C.c_int32),
                ("tracked4", C.POINTER(C.c_int32)), ("lost", C.POINTER(C.c_int32))]


class AlignerView(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_inliers", C.c_int32), ("n_outliers", C.c_int32), ("iterations", C.c_int32), ("converged", C.c_int32),
                ("total_error", C.c_double), ("chi", C.POINTER(C.c_double)), ("inlier", C.POINTER(C.c_uint8)), ("T", C.c_double * 12), ("H", C.c_double * 36)]


class PointsView(C.Structure):
    _fields_ = [("n", C.c_int32), ("first_full", C.c_int32), ("kp", C.POINTER(C.c_int16)), ("meta", C.POINTER(C.c_int32)), ("cam", C.POINTER(C.c_double)),
                ("desc", C.POINTER(C.c_uint8)), ("info", FrameInfo), ("seconds_tracking", C.c_double), ("seconds_pose_optimization", C.c_double),
                ("seconds_point_recovery", C.c_double), ("seconds_landmark_optimization", C.c_double), ("seconds_point_triangulation", C.c_double)]


class DepthParams(C.Structure):
    """struct vslam_depth_params (include/vslam_hip.h)."""
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("K_left", C.c_double * 9), ("K_left_inverse", C.c_double * 9),
                ("K_right_inverse", C.c_double * 9), ("right_to_left", C.c_double * 12),
                ("depth_scale_factor_intensity_to_meters", C.c_double), ("minimum_depth_meters", C.c_double),
                ("maximum_depth_meters", C.c_double), ("enable_point_triangulation", C.c_int32),
                ("enable_keypoint_binning", C.c_int32), ("bin_size_pixels", C.c_int32), ("descriptor_type", C.c_int32),
                ("detector_type", C.c_int32)]

    @staticmethod
    def make(rows, cols, K_left, K_left_inverse, K_right_inverse, right_to_left, scale=1e-3, min_depth=0.1, max_depth=10.0,
             triangulation=1, binning=1, bin_px=6, descriptor=0, detector=0):
        p = DepthParams()
        p.rows, p.cols = int(rows), int(cols)
        for name, a in (("K_left", K_left), ("K_left_inverse", K_left_inverse), ("K_right_inverse", K_right_inverse), ("right_to_left", right_to_left)):
            flat = np.ascontiguousarray(a, np.float64).ravel()
            getattr(p, name)[:] = list(flat)
        p.depth_scale_factor_intensity_to_meters = scale
        p.minimum_depth_meters, p.maximum_depth_meters = min_depth, max_depth
        p.enable_point_triangulation, p.enable_keypoint_binning, p.bin_size_pixels = int(triangulation), int(binning), int(bin_px)
        p.descriptor_type = int(descriptor)
        p.detector_type = int(detector)      # 0 FAST, 1 ORB (OrbDetector)
        return p


class VslamError(RuntimeError):
    """Raised for a negative vslam_status (the reference throws std::runtime_error)."""

    def __init__(self, code, text):
        super().__init__("vslam status %d: %s" % (code, text))
        self.code = code


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


class CApi(object):
    """One loaded library + one context (n_streams independent sequences)."""

    def __init__(self, lib_path, prefix):
        self.lib = C.CDLL(lib_path)
        self.prefix = prefix
        self.ctx = None

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    # -- lifetime ---------------------------------------------------------------------------
    def default_config(self, which="kitti"):
        cfg = Config()
        self.fn("default_config_" + which)(C.byref(cfg))
        return cfg

    def create(self, cfg, device=0, n_streams=1):
        self.destroy()
        ctx = C.c_void_p()
        f = self.fn("create")
        f.restype = C.c_int
        rc = f(C.byref(cfg), C.c_int(device), C.c_int(n_streams), C.byref(ctx))
        if rc != OK:
            raise VslamError(rc, self.last_error(None))
        self.ctx = ctx
        self.cfg = cfg.copy()
        self.n_streams = n_streams
        return self

    def destroy(self):
        if self.ctx is not None:
            self.fn("destroy")(self.ctx)
            self.ctx = None

    def last_error(self, ctx):
        if not self.has("last_error"):
            return ""
        f = self.fn("last_error")
        f.restype = C.c_char_p
        s = f(ctx)
        return s.decode() if s else ""

    def check(self, rc):
        if rc != OK:
            raise VslamError(rc, self.last_error(self.ctx))

    def reset(self):
        self.check(self.fn("reset")(self.ctx))

    def set_stream_active(self, stream, active=True):
        self.check(self.fn("set_stream_active")(self.ctx, C.c_int(stream), C.c_int(1 if active else 0)))

    def reset_stream(self, stream):
        self.check(self.fn("reset_stream")(self.ctx, C.c_int(stream)))

    def reset_streams(self, streams):
        ids = np.ascontiguousarray(streams, np.int32)
        if len(ids):
            self.check(self.fn("reset_streams")(self.ctx, C.c_int32(len(ids)), _p(ids, C.c_int32)))

    # -- whole frame --------------------------------------------------------------------------
    def process_host(self, left, right):
        """left/right: uint8 arrays [n_streams, rows, stride] (C-contiguous)."""
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        if left.ndim == 2:
            left, right = left[None], right[None]
        assert left.shape == right.shape and left.shape[0] == self.n_streams
        assert left.shape[1] == self.cfg.rows and left.shape[2] >= self.cfg.cols
        stride = left.shape[2]
        self.check(self.fn("process_host")(self.ctx, _p(left, C.c_uint8), _p(right, C.c_uint8),
                                           C.c_int32(stride), C.c_size_t(left.shape[1] * stride)))

    def process_device(self, left_ptr, right_ptr, row_stride, image_stride):
        self.check(self.fn("process_device")(self.ctx, C.c_void_p(left_ptr), C.c_void_p(right_ptr),
                                             C.c_int32(row_stride), C.c_size_t(image_stride)))

    def synchronize(self):
        self.check(self.fn("synchronize")(self.ctx))

    # -- readback -----------------------------------------------------------------------------
    def frame_info(self, stream=0):
        fi = FrameInfo()
        self.check(self.fn("get_frame_info")(self.ctx, C.c_int(stream), C.byref(fi)))
        return fi

    def keypoints(self, stream=0, side=0):
        cap = int(self.cfg.max_keypoints)
        n = C.c_int32()
        xy = np.zeros((cap, 2), np.int16)
        score = np.zeros(cap, np.int32)
        desc = np.zeros((cap, 32), np.uint8)
        self.check(self.fn("get_keypoints")(self.ctx, C.c_int(stream), C.c_int(side), C.c_int32(cap), C.byref(n),
                                            _p(xy, C.c_int16), _p(score, C.c_int32), _p(desc, C.c_uint8)))
        k = n.value
        return xy[:k].copy(), score[:k].copy(), desc[:k].copy()

    def points(self, stream=0):
        cap = int(self.cfg.max_points)
        n = C.c_int32()
        kp = np.zeros((cap, 4), np.int16)
        meta = np.zeros((cap, 6), np.int32)
        cam = np.zeros((cap, 3), np.float64)
        lm = np.zeros((cap, 3), np.float64)
        self.check(self.fn("get_points")(self.ctx, C.c_int(stream), C.c_int32(cap), C.byref(n), _p(kp, C.c_int16),
                                         _p(meta, C.c_int32), _p(cam, C.c_double), _p(lm, C.c_double)))
        k = n.value
        return dict(kp=kp[:k].copy(), meta=meta[:k].copy(), cam=cam[:k].copy(), lm=lm[:k].copy())

    def aligner_result(self, stream=0):
        cap = int(self.cfg.max_points)
        n = C.c_int32()
        chi = np.zeros(cap, np.float64)
        inl = np.zeros(cap, np.uint8)
        T = np.zeros(12, np.float64)
        H = np.zeros(36, np.float64)
        self.check(self.fn("get_aligner_result")(self.ctx, C.c_int(stream), C.c_int32(cap), C.byref(n),
                                                 _p(chi, C.c_double), _p(inl, C.c_uint8), _p(T, C.c_double),
                                                 _p(H, C.c_double)))
        k = n.value
        return dict(chi=chi[:k].copy(), inlier=inl[:k].copy(), T=T.reshape(3, 4), H=H.reshape(6, 6))

    def aligner_weights_of(self, stream=0):
        """StereoUVAligner::_weights_translation as the stream's last initialize() left it."""
        cap = int(self.cfg.max_points)
        n = C.c_int32()
        w = np.zeros(cap, np.float64)
        self.check(self.fn("get_aligner_weights")(self.ctx, C.c_int(stream), C.c_int32(cap), C.byref(n), _p(w, C.c_double)))
        return w[:n.value].copy()

    def aligner_weights(self, cfg, sizes, inverse_depth, depth):
        """Weights after each of a sequence of StereoUVAligner::initialize calls on one aligner (known-answer tests)."""
        n = np.ascontiguousarray(sizes, np.int32); inv = np.ascontiguousarray(inverse_depth, np.int32)
        d = np.ascontiguousarray(depth, np.float64)
        out = np.zeros(max(int(n.sum()), 1), np.float64)
        first = (self.ctx,) if self.prefix == "vslam_" else (C.byref(cfg),)
        self.check(self.fn("aligner_weights")(*first, C.c_int32(n.shape[0]), _p(n, C.c_int32), _p(inv, C.c_int32), _p(d, C.c_double),
                                              _p(out, C.c_double)))
        return out[:int(n.sum())]

    def poses(self, stream, first, count):
        out = np.zeros((count, 12), np.float64)
        self.check(self.fn("get_poses")(self.ctx, C.c_int(stream), C.c_int32(first), C.c_int32(count),
                                        _p(out, C.c_double)))
        return out.reshape(count, 3, 4)

    # -- stage views (copies of what the pointers show, taken at once: the buffer is reused by the next view call) --------------
    def view_keypoints(self, stream=0):
        v = KeypointsView()
        self.check(self.fn("view_keypoints")(self.ctx, C.c_int(stream), C.byref(v)))
        out = []
        for d in (0, 1):
            n = v.n[d]
            out.append((np.ctypeslib.as_array(v.xy[d], (max(n, 1), 2))[:n].copy(), np.ctypeslib.as_array(v.score[d], (max(n, 1),))[:n].copy(),
                        np.ctypeslib.as_array(v.desc[d], (max(n, 1), 32))[:n].copy()))
        return out

    def view_keypoints_xy(self, stream=0):
        """Coordinates and scores as soon as the detector has written them (descriptors: view_keypoints afterwards)."""
        v = KeypointsView()
        self.check(self.fn("view_keypoints_xy")(self.ctx, C.c_int(stream), C.byref(v)))
        out = []
        for d in (0, 1):
            n = v.n[d]
            out.append((np.ctypeslib.as_array(v.xy[d], (max(n, 1), 2))[:n].copy(), np.ctypeslib.as_array(v.score[d], (max(n, 1),))[:n].copy(), bool(v.desc[d])))
        return out

    def view_track(self, stream=0):
        v = TrackView()
        self.check(self.fn("view_track")(self.ctx, C.c_int(stream), C.byref(v)))
        return dict(n_tracked_landmarks=v.n_tracked_landmarks, tracked4=np.ctypeslib.as_array(v.tracked4, (max(v.n_tracked, 1), 4))[:v.n_tracked].copy(),
                    lost=np.ctypeslib.as_array(v.lost, (max(v.n_lost, 1),))[:v.n_lost].copy())

    def view_aligner(self, stream=0):
        v = AlignerView()
        self.check(self.fn("view_aligner")(self.ctx, C.c_int(stream), C.byref(v)))
        return dict(n_inliers=v.n_inliers, n_outliers=v.n_outliers, iterations=v.iterations, converged=v.converged, total_error=v.total_error,
                    chi=np.ctypeslib.as_array(v.chi, (max(v.n, 1),))[:v.n].copy(), inlier=np.ctypeslib.as_array(v.inlier, (max(v.n, 1),))[:v.n].copy(),
                    T=np.array(v.T).reshape(3, 4), H=np.array(v.H).reshape(6, 6))

    def view_points(self, stream=0, in_progress=False):
        v = PointsView()
        self.check(self.fn("view_points")(self.ctx, C.c_int(stream), C.c_int(1 if in_progress else 0), C.byref(v)))
        n, f = v.n, v.first_full
        out = dict(n=n, first_full=f, kp=np.ctypeslib.as_array(v.kp, (max(n, 1), 4))[:n].copy(), meta=np.ctypeslib.as_array(v.meta, (max(n, 1), 6))[:n].copy(),
                   cam=np.ctypeslib.as_array(v.cam, (max(n, 1), 3))[:n].copy(), info=FrameInfo.from_buffer_copy(bytes(v.info)),
                   seconds=[v.seconds_tracking, v.seconds_pose_optimization, v.seconds_point_recovery, v.seconds_landmark_optimization, v.seconds_point_triangulation])
        out["desc"] = np.ctypeslib.as_array(v.desc, (max(n, 1), 64))[:n].copy() if in_progress else None
        return out

    # -- stand-alone stages ---------------------------------------------------------------------
    def fast_detect(self, image, roi, threshold, cap=65536):
        image = np.ascontiguousarray(image, np.uint8)
        rows, stride = image.shape
        x, y, w, h = roi
        n = C.c_int32()
        xy = np.zeros((cap, 2), np.int16)
        score = np.zeros(cap, np.int32)
        self.check(self.fn("fast_detect")(self.ctx, _p(image, C.c_uint8), C.c_int32(rows), C.c_int32(stride),
                                          C.c_int32(stride), C.c_int32(x), C.c_int32(y), C.c_int32(w), C.c_int32(h),
                                          C.c_int32(threshold), C.c_int32(cap), C.byref(n), _p(xy, C.c_int16),
                                          _p(score, C.c_int32)))
        return xy[:n.value].copy(), score[:n.value].copy()

    def brief_describe(self, image, xy):
        image = np.ascontiguousarray(image, np.uint8)
        xy = np.ascontiguousarray(xy, np.int16)
        rows, stride = image.shape
        n = xy.shape[0]
        keep = np.zeros(n, np.uint8)
        desc = np.zeros((n, 32), np.uint8)
        self.check(self.fn("brief_describe")(self.ctx, _p(image, C.c_uint8), C.c_int32(rows), C.c_int32(stride),
                                             C.c_int32(stride), C.c_int32(n), _p(xy, C.c_int16), _p(keep, C.c_uint8),
                                             _p(desc, C.c_uint8)))
        return keep, desc

    # -- descriptor test pairs (run-time data, one table per device) --------------------------------
    def get_pattern(self, which, device=0):
        """which: "brief" ({y1, x1, y2, x2}) or "orb" ({x1, y1, x2, y2}); returns int8 [256, 4]."""
        out = np.zeros((256, 4), np.int8)
        rc = getattr(self.lib, self.prefix + "get_%s_pattern" % which)(C.c_int(device), _p(out, C.c_int8))
        if rc != 0:
            raise VslamError(rc, "get_%s_pattern" % which)
        return out

    def set_pattern(self, which, pairs, device=0):
        pairs = np.ascontiguousarray(pairs, np.int8).reshape(256, 4)
        rc = getattr(self.lib, self.prefix + "set_%s_pattern" % which)(C.c_int(device), _p(pairs, C.c_int8))
        if rc != 0:
            raise VslamError(rc, "set_%s_pattern: table rejected" % which)

    def knn2(self, query, train, norm=0):
        query = np.ascontiguousarray(query, np.uint8)
        train = np.ascontiguousarray(train, np.uint8)
        nq, nt = query.shape[0], train.shape[0]
        idx = np.zeros((nq, 2), np.int32)
        dist = np.zeros((nq, 2), np.float32)
        self.check(self.fn("knn2")(self.ctx, C.c_int(norm), C.c_int32(nq), _p(query, C.c_uint8), C.c_int32(nt),
                                   _p(train, C.c_uint8), _p(idx, C.c_int32), _p(dist, C.c_float)))
        return idx, dist

    def align_points(self, moving, fixed, omega, weight, T_init):
        moving = np.ascontiguousarray(moving, np.float64)
        fixed = np.ascontiguousarray(fixed, np.float64)
        omega = np.ascontiguousarray(omega, np.float64)
        weight = np.ascontiguousarray(weight, np.float64)
        T_init = np.ascontiguousarray(T_init, np.float64).reshape(12)
        n = moving.shape[0]
        T = np.zeros(12, np.float64)
        chi = np.zeros(n, np.float64)
        inl = np.zeros(n, np.uint8)
        ninl, its = C.c_int32(), C.c_int32()
        err = C.c_double()
        H = np.zeros(36, np.float64)
        self.check(self.fn("align_points")(self.ctx, C.c_int32(n), _p(moving, C.c_double), _p(fixed, C.c_double),
                                           _p(omega, C.c_double), _p(weight, C.c_double), _p(T_init, C.c_double),
                                           _p(T, C.c_double), _p(chi, C.c_double), _p(inl, C.c_uint8), C.byref(ninl),
                                           C.byref(err), C.byref(its), _p(H, C.c_double)))
        return dict(T=T.reshape(3, 4), chi=chi, inlier=inl, n_inliers=ninl.value, total_error=err.value,
                    iterations=its.value, H=H.reshape(6, 6))


    def align_points_uvd(self, moving, fixed_uvd, omega_uv, omega_depth, weight, T_init):
        moving = np.ascontiguousarray(moving, np.float64)
        fixed = np.ascontiguousarray(fixed_uvd, np.float64)
        ouv = np.ascontiguousarray(omega_uv, np.float64)
        od = np.ascontiguousarray(omega_depth, np.float64)
        weight = np.ascontiguousarray(weight, np.float64)
        T_init = np.ascontiguousarray(T_init, np.float64).reshape(12)
        n = moving.shape[0]
        T = np.zeros(12, np.float64)
        chi = np.zeros(n, np.float64)
        inl = np.zeros(n, np.uint8)
        ninl, its = C.c_int32(), C.c_int32()
        err = C.c_double()
        H = np.zeros(36, np.float64)
        self.check(self.fn("align_points_uvd")(self.ctx, C.c_int32(n), _p(moving, C.c_double), _p(fixed, C.c_double),
                                               _p(ouv, C.c_double), _p(od, C.c_double), _p(weight, C.c_double),
                                               _p(T_init, C.c_double), _p(T, C.c_double), _p(chi, C.c_double),
                                               _p(inl, C.c_uint8), C.byref(ninl), C.byref(err), C.byref(its),
                                               _p(H, C.c_double)))
        return dict(T=T.reshape(3, 4), chi=chi, inlier=inl, n_inliers=ninl.value, total_error=err.value,
                    iterations=its.value, H=H.reshape(6, 6))

    def track_match(self, T, d, tau_track, tau_tri, by_appearance, cam, prev_desc_left, prev_desc_right, epi,
                    rc_left, desc_left, rc_right, desc_right):
        """StereoFramePointGenerator::track on caller-provided data (known-answer tests): returns (tracked [n][4], lost)."""
        T = np.ascontiguousarray(T, np.float64).reshape(12)
        cam = np.ascontiguousarray(cam, np.float64)
        pdl = np.ascontiguousarray(prev_desc_left, np.uint8); pdr = np.ascontiguousarray(prev_desc_right, np.uint8)
        epi = np.ascontiguousarray(epi, np.int32)
        rcl = np.ascontiguousarray(rc_left, np.int32); dl = np.ascontiguousarray(desc_left, np.uint8)
        rcr = np.ascontiguousarray(rc_right, np.int32); dr = np.ascontiguousarray(desc_right, np.uint8)
        n = cam.shape[0]
        out = np.zeros((max(n, 1), 4), np.int32)
        lost = np.zeros(max(n, 1), np.int32)
        nt, nl = C.c_int32(), C.c_int32()
        self.check(self.fn("track_match")(self.ctx, _p(T, C.c_double), C.c_int32(int(d)), C.c_double(float(tau_track)),
                                          C.c_double(float(tau_tri)), C.c_int32(int(by_appearance)), C.c_int32(n),
                                          _p(cam, C.c_double), _p(pdl, C.c_uint8), _p(pdr, C.c_uint8), _p(epi, C.c_int32),
                                          C.c_int32(rcl.shape[0]), _p(rcl, C.c_int32), _p(dl, C.c_uint8),
                                          C.c_int32(rcr.shape[0]), _p(rcr, C.c_int32), _p(dr, C.c_uint8),
                                          C.byref(nt), _p(out, C.c_int32), C.byref(nl), _p(lost, C.c_int32)))
        return out[:nt.value].copy(), lost[:nl.value].copy()


    def stereo_match(self, tau_tri, rc_left, desc_left, rc_right, desc_right, cap=8192):
        """StereoFramePointGenerator::compute on caller-provided features: (left, right, distance, offset) rows."""
        rcl = np.ascontiguousarray(rc_left, np.int32); dl = np.ascontiguousarray(desc_left, np.uint8)
        rcr = np.ascontiguousarray(rc_right, np.int32); dr = np.ascontiguousarray(desc_right, np.uint8)
        out = np.zeros((cap, 4), np.int32)
        n = C.c_int32()
        self.check(self.fn("stereo_match")(self.ctx, C.c_double(float(tau_tri)), C.c_int32(rcl.shape[0]), _p(rcl, C.c_int32),
                                           _p(dl, C.c_uint8), C.c_int32(rcr.shape[0]), _p(rcr, C.c_int32), _p(dr, C.c_uint8),
                                           C.c_int32(cap), C.byref(n), _p(out, C.c_int32)))
        return out[:n.value].copy()

    def stereo_recover(self, image_left, image_right, w2c, has_landmark, landmark_world, prev_desc_left, prev_desc_right, tau_track, tau_tri):
        """StereoFramePointGenerator::recoverPoints on caller-provided lost points (known-answer tests): dict of
        index / xy4 / dist / desc / xyz rows of the recovered points in lost-list order."""
        L = np.ascontiguousarray(image_left, np.uint8); R = np.ascontiguousarray(image_right, np.uint8)
        w = np.ascontiguousarray(w2c, np.float64).reshape(12)
        hl = np.ascontiguousarray(has_landmark, np.uint8); lm = np.ascontiguousarray(landmark_world, np.float64)
        pdl = np.ascontiguousarray(prev_desc_left, np.uint8); pdr = np.ascontiguousarray(prev_desc_right, np.uint8)
        n = hl.shape[0]
        idx = np.zeros(max(n, 1), np.int32); xy4 = np.zeros((max(n, 1), 4), np.int32); dist = np.zeros(max(n, 1), np.int32)
        desc = np.zeros((max(n, 1), 64), np.uint8); xyz = np.zeros((max(n, 1), 3), np.float64)
        k = C.c_int32()
        head = (self.ctx,) if self.prefix == "vslam_" else (C.byref(self.cfg),)
        self.check(self.fn("stereo_recover")(*head, _p(L, C.c_uint8), _p(R, C.c_uint8), C.c_int32(L.shape[1]), _p(w, C.c_double), C.c_int32(n),
                                             _p(hl, C.c_uint8), _p(lm, C.c_double), _p(pdl, C.c_uint8), _p(pdr, C.c_uint8),
                                             C.c_double(float(tau_track)), C.c_double(float(tau_tri)), C.byref(k), _p(idx, C.c_int32),
                                             _p(xy4, C.c_int32), _p(dist, C.c_int32), _p(desc, C.c_uint8), _p(xyz, C.c_double)))
        k = k.value
        return dict(index=idx[:k].copy(), xy4=xy4[:k].copy(), dist=dist[:k].copy(), desc=desc[:k].copy(), xyz=xyz[:k].copy())

    # -- RGB-D components.  The oracle's entry points (prefix orc_) take no context. --------------------------------
    def _ctx_args(self):
        return (self.ctx,) if self.prefix == "vslam_" else ()

    def depth_space_map(self, params, depth):
        depth = np.ascontiguousarray(depth, np.uint16)
        rows, cols = depth.shape
        space = np.zeros((rows, cols, 3), np.float32)
        rmap = np.zeros((rows, cols), np.int16); cmap = np.zeros((rows, cols), np.int16)
        self.check(self.fn("depth_space_map")(*self._ctx_args(), C.byref(params), _p(depth, C.c_uint16), C.c_int32(cols),
                                              _p(space, C.c_float), _p(rmap, C.c_int16), _p(cmap, C.c_int16)))
        return space, rmap, cmap

    def depth_compute(self, params, space, rc_features, rc_tracked, cap=8192):
        """space: host map or None (HIP only: the map the last depth_space_map call left on the device)."""
        rcf = np.ascontiguousarray(rc_features, np.int32).reshape(-1, 2)
        rct = np.ascontiguousarray(rc_tracked, np.int32).reshape(-1, 2)
        sp = None if space is None else np.ascontiguousarray(space, np.float32)
        nf, xyz = np.zeros(cap, np.int32), np.zeros((cap, 3), np.float64)
        tf_, txyz = np.zeros(cap, np.int32), np.zeros((cap, 3), np.float64)
        nn, nt = C.c_int32(), C.c_int32()
        self.check(self.fn("depth_compute")(*self._ctx_args(), C.byref(params), None if sp is None else _p(sp, C.c_float),
                                            C.c_int32(rcf.shape[0]), _p(rcf, C.c_int32), C.c_int32(rct.shape[0]),
                                            _p(rct, C.c_int32), C.c_int32(cap), C.byref(nn), _p(nf, C.c_int32),
                                            _p(xyz, C.c_double), C.byref(nt), _p(tf_, C.c_int32), _p(txyz, C.c_double)))
        return nf[:nn.value].copy(), xyz[:nn.value].copy(), tf_[:nt.value].copy(), txyz[:nt.value].copy()

    def depth_track(self, params, space, T, d, tau, by_appearance, cam, prev_desc, prev_flags, rc_left, desc_left):
        """DepthFramePointGenerator::track on caller-provided data: (tracked [n][2], xyz [n][3], temporary [m][2], lost, landmarks)."""
        T = np.ascontiguousarray(T, np.float64).reshape(12)
        cam = np.ascontiguousarray(cam, np.float64).reshape(-1, 3)
        pd = np.ascontiguousarray(prev_desc, np.uint8); pf = np.ascontiguousarray(prev_flags, np.uint8)
        rc = np.ascontiguousarray(rc_left, np.int32).reshape(-1, 2); dl = np.ascontiguousarray(desc_left, np.uint8)
        sp = None if space is None else np.ascontiguousarray(space, np.float32)
        nP = cam.shape[0]
        out2, xyz = np.zeros((max(nP, 1), 2), np.int32), np.zeros((max(nP, 1), 3), np.float64)
        tmp2, lost = np.zeros((max(nP, 1), 2), np.int32), np.zeros(max(nP, 1), np.int32)
        nt, ntmp, nl, nlm = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self.check(self.fn("depth_track")(*self._ctx_args(), C.byref(params), None if sp is None else _p(sp, C.c_float), _p(T, C.c_double),
                                          C.c_int32(int(d)), C.c_double(float(tau)), C.c_int32(int(by_appearance)), C.c_int32(nP),
                                          _p(cam, C.c_double), _p(pd, C.c_uint8), _p(pf, C.c_uint8), C.c_int32(rc.shape[0]),
                                          _p(rc, C.c_int32), _p(dl, C.c_uint8), C.byref(nt), _p(out2, C.c_int32), _p(xyz, C.c_double),
                                          C.byref(ntmp), _p(tmp2, C.c_int32), C.byref(nl), _p(lost, C.c_int32), C.byref(nlm)))
        return out2[:nt.value].copy(), xyz[:nt.value].copy(), tmp2[:ntmp.value].copy(), lost[:nl.value].copy(), nlm.value

    def depth_recover(self, params, space, image_left, world_to_camera_left, has_landmark, landmark_world, prev_desc, keypoint_size, tau):
        """DepthFramePointGenerator::recoverPoints on caller-provided data: (lost-list index, keypoint xy, descriptor, xyz)."""
        img = np.ascontiguousarray(image_left, np.uint8)
        w2c = np.ascontiguousarray(world_to_camera_left, np.float64).reshape(12)
        hl = np.ascontiguousarray(has_landmark, np.uint8); lm = np.ascontiguousarray(landmark_world, np.float64).reshape(-1, 3)
        pd = np.ascontiguousarray(prev_desc, np.uint8)
        sp = None if space is None else np.ascontiguousarray(space, np.float32)
        n = lm.shape[0]
        idx, xy = np.zeros(max(n, 1), np.int32), np.zeros((max(n, 1), 2), np.float32)
        desc, xyz = np.zeros((max(n, 1), 32), np.uint8), np.zeros((max(n, 1), 3), np.float64)
        nr = C.c_int32()
        self.check(self.fn("depth_recover")(*self._ctx_args(), C.byref(params), None if sp is None else _p(sp, C.c_float), _p(img, C.c_uint8),
                                            C.c_int32(img.shape[1]), _p(w2c, C.c_double), C.c_int32(n), _p(hl, C.c_uint8), _p(lm, C.c_double),
                                            _p(pd, C.c_uint8), C.c_float(float(keypoint_size)), C.c_double(float(tau)), C.byref(nr),
                                            _p(idx, C.c_int32), _p(xy, C.c_float), _p(desc, C.c_uint8), _p(xyz, C.c_double)))
        k = nr.value
        return idx[:k].copy(), xy[:k].copy(), desc[:k].copy(), xyz[:k].copy()

    def landmark_update(self, cfg, offsets, frame_of, world_to_camera, camera_to_world, cam, world, updates):
        """Landmark::update for a batch of landmarks (last measurement of each list = the new observation)."""
        off = np.ascontiguousarray(offsets, np.int32); fo = np.ascontiguousarray(frame_of, np.int32)
        w2c = np.ascontiguousarray(world_to_camera, np.float64).reshape(-1, 12); c2w = np.ascontiguousarray(camera_to_world, np.float64).reshape(-1, 12)
        cm = np.ascontiguousarray(cam, np.float64).reshape(-1, 3)
        w = np.array(world, np.float64).reshape(-1, 3).copy(); u = np.array(updates, np.int32).copy()
        first = (self.ctx,) if self.prefix == "vslam_" else (C.byref(cfg),)
        self.check(self.fn("landmark_update")(*first, C.c_int32(w.shape[0]), _p(off, C.c_int32), _p(fo, C.c_int32), C.c_int32(w2c.shape[0]),
                                              _p(w2c, C.c_double), _p(c2w, C.c_double), _p(cm, C.c_double), _p(w, C.c_double), _p(u, C.c_int32)))
        return w, u

    # -- OrbDetector components -------------------------------------------------------------------------------------
    def resize_linear_u8(self, image, dst_rows, dst_cols):
        img = np.ascontiguousarray(image, np.uint8)
        dst = np.zeros((int(dst_rows), int(dst_cols)), np.uint8)
        self.check(self.fn("resize_linear_u8")(*self._ctx_args(), _p(img, C.c_uint8), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                                               C.c_int32(img.shape[1]), _p(dst, C.c_uint8), C.c_int32(dst.shape[0]), C.c_int32(dst.shape[1])))
        return dst

    def harris_angle(self, image, xy):
        img = np.ascontiguousarray(image, np.uint8)
        pts = np.ascontiguousarray(xy, np.int16).reshape(-1, 2)
        resp, ang = np.zeros(pts.shape[0], np.float32), np.zeros(pts.shape[0], np.float32)
        self.check(self.fn("harris_angle")(*self._ctx_args(), _p(img, C.c_uint8), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                                           C.c_int32(img.shape[1]), C.c_int32(pts.shape[0]), _p(pts, C.c_int16), _p(resp, C.c_float),
                                           _p(ang, C.c_float)))
        return resp, ang

    def orb_detect(self, image, nfeatures=5000, scale_factor=1.2, nlevels=8, edge_threshold=31, patch_size=31, fast_threshold=20, cap=20000):
        """cv::ORB::detect (HARRIS_SCORE): rows of (x, y, size, angle, response, octave)."""
        img = np.ascontiguousarray(image, np.uint8)
        out = np.zeros((cap, 6), np.float32)
        n = C.c_int32()
        self.check(self.fn("orb_detect")(*self._ctx_args(), _p(img, C.c_uint8), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                                         C.c_int32(img.shape[1]), C.c_int32(int(nfeatures)), C.c_float(float(scale_factor)), C.c_int32(int(nlevels)),
                                         C.c_int32(int(edge_threshold)), C.c_int32(int(patch_size)), C.c_int32(int(fast_threshold)),
                                         C.c_int32(cap), C.byref(n), _p(out, C.c_float)))
        return out[:n.value].copy()

    def gaussian_blur7_u8(self, image):
        img = np.ascontiguousarray(image, np.uint8)
        out = np.zeros_like(img)
        self.check(self.fn("gaussian_blur7_u8")(*self._ctx_args(), _p(img, C.c_uint8), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                                                C.c_int32(img.shape[1]), _p(out, C.c_uint8)))
        return out

    def orb_describe(self, image, xy, angle_degrees=-1.0):
        """cv::ORB::create()->compute() at integer keypoints with one angle: (keep, descriptors)."""
        img = np.ascontiguousarray(image, np.uint8)
        pts = np.ascontiguousarray(xy, np.int16).reshape(-1, 2)
        n = pts.shape[0]
        keep = np.zeros(n, np.uint8)
        desc = np.zeros((n, 32), np.uint8)
        self.check(self.fn("orb_describe")(*self._ctx_args(), _p(img, C.c_uint8), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                                           C.c_int32(img.shape[1]), C.c_int32(n), _p(pts, C.c_int16), C.c_float(float(angle_degrees)),
                                           _p(keep, C.c_uint8), _p(desc, C.c_uint8)))
        return keep, desc

    def orb_describe_keypoints(self, image, keypoints, scale_factor=1.2):
        """cv::ORB::create()->compute() on keypoints with octave and angle (rows of orb_detect): (keep, descriptors)."""
        img = np.ascontiguousarray(image, np.uint8)
        kp = np.ascontiguousarray(keypoints, np.float32).reshape(-1, 6)
        n = kp.shape[0]
        keep = np.zeros(max(n, 1), np.uint8)
        desc = np.zeros((max(n, 1), 32), np.uint8)
        self.check(self.fn("orb_describe_keypoints")(*self._ctx_args(), _p(img, C.c_uint8), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                                                     C.c_int32(img.shape[1]), C.c_int32(n), _p(kp, C.c_float), C.c_float(float(scale_factor)),
                                                     _p(keep, C.c_uint8), _p(desc, C.c_uint8)))
        return keep[:n], desc[:n]

    def point_in_camera(self, xy_previous, xy_current, T, K):
        xp = np.ascontiguousarray(xy_previous, np.float32).reshape(-1, 2)
        xc = np.ascontiguousarray(xy_current, np.float32).reshape(-1, 2)
        T = np.ascontiguousarray(T, np.float64).reshape(12); K = np.ascontiguousarray(K, np.float64).reshape(9)
        out = np.zeros((xp.shape[0], 3), np.float64)
        self.check(self.fn("point_in_camera")(*self._ctx_args(), C.c_int32(xp.shape[0]), _p(xp, C.c_float), _p(xc, C.c_float),
                                              _p(T, C.c_double), _p(K, C.c_double), _p(out, C.c_double)))
        return out

def _extra_methods():
    def copy_poses_device(self, first, count, dst_ptr):
        self.check(self.fn("copy_poses_device")(self.ctx, C.c_int32(first), C.c_int32(count), C.c_void_p(dst_ptr)))

    def copy_current_poses_device(self, dst_ptr):
        self.check(self.fn("copy_current_poses_device")(self.ctx, C.c_void_p(dst_ptr)))

    def enable_timers(self, on=True):
        self.check(self.fn("enable_timers")(self.ctx, C.c_int(1 if on else 0)))

    def kernel_times(self):
        ms = (C.c_double * 8)()
        n = (C.c_int32 * 8)()
        self.check(self.fn("get_kernel_times")(self.ctx, ms, n))
        names = ["k_fast_box", "k_emit", "k_brief", "k_track_candidates", "k_frame", "k_recover_brief",
                 "k_update_landmarks", "k_stereo_dist"]
        return {names[i]: (ms[i], n[i]) for i in range(8)}

    def timers(self):
        sec = (C.c_double * 8)()
        self.check(self.fn("get_timers")(self.ctx, sec))
        names = ["keypoint_detection", "descriptor_extraction", "point_triangulation", "tracking", "track_creation",
                 "pose_optimization", "landmark_optimization", "point_recovery"]
        return {names[i]: sec[i] for i in range(8)}

    def set_hip_stream(self, stream_ptr):
        self.check(self.fn("set_hip_stream")(self.ctx, C.c_void_p(stream_ptr)))

    for f in (copy_poses_device, copy_current_poses_device, enable_timers, kernel_times, timers, set_hip_stream):
        setattr(CApi, f.__name__, f)


_extra_methods()


class RgbdTracker(object):
    """ctypes view of vslam_rgbd_* (RGB-D mode end to end inside libvslam_hip.so: the device-resident loop, or the host-driven loop over the
    stand-alone entry points when VSLAM_RGBD_HOST=1 is set while the tracker is created)."""

    def __init__(self, api, cfg, params, device=0):
        self.lib = api.lib
        self.cfg = cfg.copy()
        self.lib.vslam_rgbd_last_error.restype = C.c_char_p
        self.h = C.c_void_p()
        rc = self.lib.vslam_rgbd_create(C.byref(cfg), C.byref(params), C.c_int(device), C.byref(self.h))
        if rc != OK:
            raise VslamError(rc, self.lib.vslam_rgbd_last_error(None).decode())

    def _check(self, rc):
        if rc != OK:
            raise VslamError(rc, self.lib.vslam_rgbd_last_error(self.h).decode())

    def reset(self):
        self._check(self.lib.vslam_rgbd_reset(self.h))

    def process(self, left, depth, cols=None):
        """left / depth: 2-D arrays; cols: image width when the arrays carry padding columns (row stride = array width)."""
        left = np.ascontiguousarray(left, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        self._check(self.lib.vslam_rgbd_process_host(self.h, _p(left, C.c_uint8), C.c_int32(left.shape[1]), _p(depth, C.c_uint16), C.c_int32(depth.shape[1])))
        fi = FrameInfo()
        nt = C.c_int32()
        self._check(self.lib.vslam_rgbd_get_frame_info(self.h, C.byref(fi), C.byref(nt)))
        return fi, nt.value

    def submit(self, left, depth):
        """First half of process(): copies the frame in and enqueues it.  The arrays are kept alive until wait()."""
        left = np.ascontiguousarray(left, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        self._inflight = (left, depth)
        self._check(self.lib.vslam_rgbd_submit_host(self.h, _p(left, C.c_uint8), C.c_int32(left.shape[1]), _p(depth, C.c_uint16), C.c_int32(depth.shape[1])))

    def wait(self):
        self._check(self.lib.vslam_rgbd_wait(self.h))
        self._inflight = None
        fi = FrameInfo()
        nt = C.c_int32()
        self._check(self.lib.vslam_rgbd_get_frame_info(self.h, C.byref(fi), C.byref(nt)))
        return fi, nt.value

    def points(self):
        cap = int(self.cfg.max_points) * 4
        n = C.c_int32()
        xy = np.zeros((cap, 2), np.float32); cam = np.zeros((cap, 3), np.float64); meta = np.zeros((cap, 4), np.int32); desc = np.zeros((cap, 32), np.uint8)
        self._check(self.lib.vslam_rgbd_get_points(self.h, C.c_int32(cap), C.byref(n), _p(xy, C.c_float), _p(cam, C.c_double), _p(meta, C.c_int32), _p(desc, C.c_uint8)))
        k = n.value
        return dict(xy=xy[:k].copy(), cam=cam[:k].copy(), meta=meta[:k].copy(), desc=desc[:k].copy())

    def destroy(self):
        if self.h:
            self.lib.vslam_rgbd_destroy(self.h)
            self.h = None


class RgbdBatch(object):
    """ctypes view of vslam_rgbd_create_batch / _process_batch_host: n_streams sequences of one camera and configuration in one context."""

    def __init__(self, api, cfg, params, n_streams, device=0):
        self.lib = api.lib
        self.cfg = cfg.copy()
        self.n = int(n_streams)
        self.lib.vslam_rgbd_last_error.restype = C.c_char_p
        self.h = C.c_void_p()
        rc = self.lib.vslam_rgbd_create_batch(C.byref(cfg), C.byref(params), C.c_int(device), C.c_int32(self.n), C.byref(self.h))
        if rc != OK:
            raise VslamError(rc, self.lib.vslam_rgbd_last_error(None).decode())

    def _check(self, rc):
        if rc != OK:
            raise VslamError(rc, self.lib.vslam_rgbd_last_error(self.h).decode())

    def reset(self):
        self._check(self.lib.vslam_rgbd_reset(self.h))

    def submit(self, left, depth):
        """left: [n_streams, rows, stride] uint8, depth: [n_streams, rows, stride] uint16 (kept alive until wait())."""
        left = np.ascontiguousarray(left, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        assert left.shape[0] == self.n and depth.shape[0] == self.n
        self._inflight = (left, depth)
        self._check(self.lib.vslam_rgbd_submit_batch_host(self.h, _p(left, C.c_uint8), C.c_int32(left.shape[2]), C.c_size_t(left.shape[1] * left.shape[2]),
                                                          _p(depth, C.c_uint16), C.c_int32(depth.shape[2]), C.c_size_t(depth.shape[1] * depth.shape[2])))

    def submit_device(self, left_ptr, left_row_stride, left_stream_stride, depth_ptr, depth_row_stride, depth_stream_stride):
        """Images already in HBM: device addresses (e.g. torch tensors' data_ptr()), strides in bytes / depth in elements."""
        self._check(self.lib.vslam_rgbd_submit_batch_device(self.h, C.c_void_p(left_ptr), C.c_int32(left_row_stride), C.c_size_t(left_stream_stride),
                                                            C.c_void_p(depth_ptr), C.c_int32(depth_row_stride), C.c_size_t(depth_stream_stride)))

    def wait(self, infos=True):
        self._check(self.lib.vslam_rgbd_wait(self.h))
        self._inflight = None
        return [self.frame_info(s) for s in range(self.n)] if infos else None

    def process(self, left, depth):
        self.submit(left, depth)
        return self.wait()

    def frame_info(self, s):
        fi = FrameInfo()
        nt = C.c_int32()
        self._check(self.lib.vslam_rgbd_get_frame_info_stream(self.h, C.c_int32(s), C.byref(fi), C.byref(nt)))
        return fi, nt.value

    def points(self, s):
        cap = int(self.cfg.max_points) * 4
        n = C.c_int32()
        xy = np.zeros((cap, 2), np.float32); cam = np.zeros((cap, 3), np.float64); meta = np.zeros((cap, 4), np.int32); desc = np.zeros((cap, 32), np.uint8)
        self._check(self.lib.vslam_rgbd_get_points_stream(self.h, C.c_int32(s), C.c_int32(cap), C.byref(n), _p(xy, C.c_float), _p(cam, C.c_double), _p(meta, C.c_int32),
                                                          _p(desc, C.c_uint8)))
        k = n.value
        return dict(xy=xy[:k].copy(), cam=cam[:k].copy(), meta=meta[:k].copy(), desc=desc[:k].copy())

    def destroy(self):
        if self.h:
            self.lib.vslam_rgbd_destroy(self.h)
            self.h = None
