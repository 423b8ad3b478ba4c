"""RGB-D mode of the tracker, as a plain Python loop over the stand-alone entry points (TEST INFRASTRUCTURE).

A second, independent statement of what PoseTracker3D does with a DepthFramePointGenerator + UVDAligner plugged in
(slam_assembly.cpp `_createDepthTracker`; pose_tracker_3d.cpp:32-566; depth_framepoint_generator.cpp:24-407;
uvd_aligner.cpp:11-69), written over an object with the CApi interface: run over the CPU oracle it is the checker for the
product's C++ loop (csrc/rgbd_tracker.h behind vslam_rgbd_*), run over libvslam_hip.so it chains the device kernels exactly
as that loop does.  Detector grids of any shape (configuration_icl.yaml runs 2 x 2, tum and xtion 1 x 1).

Reference behaviour restated here (file:line relative to the reference root):
  * initialize() ignores `extract_features`: every re-registration detects again with the thresholds the controller has
    meanwhile moved, and runs the controller again (depth_framepoint_generator.cpp:24-44; one image per adjust: the mean is
    over ONE detection).  detectKeypoints APPENDS to frame->keypointsLeft() (base_framepoint_generator.cpp:422), which nothing clears
    between the initialize() calls of one frame (pose_tracker_3d.cpp:320,402): the second and third attempt describe, store
    (setFeatures: the lattice keeps the LAST feature written to a pixel, the vector keeps all) and track against the union of
    all attempts' keypoints, duplicates included, and _number_of_detected_keypoints is the accumulated count.  cv::ORB::compute
    regroups keypoints that are not sorted by pyramid level (stable, level-major) — the accumulated list of an OrbDetector.
  * track() works on previous points + previous temporary points (:181-184); matches on pixels without depth become
    temporary points of the current frame (:247-256) and the temporary list is NOT cleared between re-registrations.
  * a point inherits hasUnreliableDepth from its predecessor (frame_point.cpp:43-55) even when its own depth is measured;
    unreliable points get no landmark (pose_tracker_3d.cpp:490), weigh 0 in the aligner (uvd_aligner.cpp:55-61) and are
    never reported lost (:281-284).
  * UVDAligner::initialize reads the CURRENT point's landmark (:38), which nothing has set at that time: the moving point is
    always previous->cameraCoordinatesLeft(), the information always diag(1, 1, 10); the translation weights live in a member
    vector that `resize(n, 1)` does not reset (:22,62-66).
  * temporary points are triangulated with the refined motion in _updatePoints (pose_tracker_3d.cpp:524-545) and dropped when
    the result lies behind the camera."""
import math

import numpy as np

LOCALIZING, TRACKING = 0, 1


def _inv(T):
    o = np.zeros((3, 4)); o[:, :3] = T[:, :3].T; o[:, 3] = -(T[:, :3].T @ T[:, 3]); return o


def _mul(A, B):
    o = np.zeros((3, 4)); o[:, :3] = A[:, :3] @ B[:, :3]; o[:, 3] = A[:, :3] @ B[:, 3] + A[:, 3]; return o


def _apply(T, p):
    return T[:, :3] @ p + T[:, 3]


def _rotation_angle(T):
    rx, ry, rz = T[2, 1] - T[1, 2], T[0, 2] - T[2, 0], T[1, 0] - T[0, 1]
    s = math.sqrt(((rx * rx + ry * ry) + rz * rz) * 0.25)
    c = max(-1.0, min(1.0, ((T[0, 0] + T[1, 1]) + T[2, 2] - 1) * 0.5))
    if s < 1e-5:
        return 0.0 if c > 0 else math.pi
    return math.acos(c)


def _div(a, b):
    if b == 0:
        return float("nan") if a == 0 else math.copysign(float("inf"), a)
    return a / b


class Point(object):
    __slots__ = ("xy", "desc", "cam", "previous", "next", "origin", "track_len", "landmark", "unreliable", "frame")

    def __init__(self, xy, desc, cam, frame, previous=None, unreliable=False):
        self.xy = np.array(xy, np.float32); self.desc = desc; self.cam = np.array(cam, np.float64); self.frame = frame
        self.previous = None; self.next = None; self.origin = self; self.track_len = 0; self.landmark = None
        self.unreliable = unreliable
        if previous is not None:                          # FramePoint::setPrevious
            previous.next = self; self.previous = previous
            self.unreliable = previous.unreliable or unreliable
            self.track_len = previous.track_len + 1; self.origin = previous.origin

    def clear(self):                                      # FramePoint::clear
        if self.previous is not None:
            self.previous.next = None; self.previous = None
        self.landmark = None; self.next = None; self.track_len = 0; self.origin = self

    @property
    def row(self):
        return int(self.xy[1])

    @property
    def col(self):
        return int(self.xy[0])


class Landmark(object):
    def __init__(self):
        self.world = np.zeros(3); self.updates = 0; self.meas = []      # (frame index, camera coordinates)


class Frame(object):
    def __init__(self, index, c2w):
        self.index = index; self.points = []; self.temps = []; self.set_pose(c2w)

    def set_pose(self, c2w):
        self.c2w = np.array(c2w, np.float64).reshape(3, 4).copy(); self.w2c = _inv(self.c2w)


class RgbdTracker(object):
    def __init__(self, api, cfg, params):
        self.api, self.cfg, self.p = api, cfg, params
        self.status = LOCALIZING
        self.prior = np.hstack([np.eye(3), np.zeros((3, 1))])
        self.win = cfg.maximum_projection_tracking_distance_pixels
        self.tau_track = cfg.minimum_descriptor_distance_tracking
        self.target = (cfg.cols // cfg.bin_size_pixels + 1) * (cfg.rows // cfg.bin_size_pixels + 1)
        # BaseFramePointGenerator::configure (base_framepoint_generator.cpp:229-312): the detector regions (A.1 of SURVEY.md)
        nv, nh = cfg.det_rows, cfg.det_cols
        ph, pw = cfg.rows / nv, cfg.cols / nh
        self.regions = []
        for r in range(nv):
            for cc in range(nh):
                off_w, off_h, off_r, off_c = (2 if nh > 1 else 0), (2 if nv > 1 else 0), 0, 0
                if r > 0:
                    off_r = -off_h
                    if r < nv - 1:
                        off_h *= 2
                if cc > 0:
                    off_c = -off_w
                    if cc < nh - 1:
                        off_w *= 2
                self.regions.append((int(math.floor(cc * pw + 0.5)) + off_c, int(math.floor(r * ph + 0.5)) + off_r, int(pw + off_w), int(ph + off_h)))
        self.thr = [cfg.detector_threshold_minimum] * len(self.regions)
        self.per_detector = int(float(self.target) / len(self.regions))
        self.world = np.hstack([np.eye(3), np.zeros((3, 1))])
        self.frames, self.landmarks, self.lost = [], [], []
        self.n_lm_prev = 0
        self.weights = []                                   # UVDAligner::_weights_translation (a member: survives calls)
        self.K = np.array(list(cfg.K)).reshape(3, 3)
        self.info = {}

    # -- DepthFramePointGenerator::initialize (every call: detection + controller + descriptors) ---------------------------------
    def initialize(self, left, depth, first=True):
        api, c = self.api, self.cfg
        self.space, _, _ = api.depth_space_map(self.p, depth)
        if first:                                               # a new Frame: empty keypointsLeft()
            self.acc_xy = np.zeros((0, 2), np.float32); self.acc_desc = np.zeros((0, 32), np.uint8); self.acc_level = np.zeros(0, np.int32)
        parts = []
        orb_detector = getattr(self.p, "detector_type", 0) == 1
        for r, (rx, ry, rw, rh) in enumerate(self.regions):     # detectKeypoints: region-major, per-region threshold and controller
            if orb_detector:
                # OrbDetector (base_framepoint_generator.cpp:52-70): cv::ORB::create(5000, 1.2f, 8, 31, 0, 2, HARRIS_SCORE, 31, threshold) on the
                # region's view of the image; keypoint.pt += region corner (:418-419, float addition)
                kps = api.orb_detect(left[ry:ry + rh, rx:rx + rw], 5000, 1.2, 8, 31, 31, self.thr[r])
                n = len(kps)
                if n:
                    kps = kps.copy(); kps[:, 0] = kps[:, 0] + np.float32(rx); kps[:, 1] = kps[:, 1] + np.float32(ry)
                    parts.append(kps)
            else:
                pxy, _ = api.fast_detect(left, (rx, ry, rw, rh), self.thr[r])
                n = len(pxy)
                if n:
                    parts.append(pxy.astype(np.int32) + np.array([rx, ry], np.int32))
            t = float(self.thr[r])                              # base_framepoint_generator.cpp:382-415
            delta = (float(n) - self.per_detector) / self.per_detector
            if delta < -c.target_number_of_keypoints_tolerance:
                t = t + min(max(delta, -c.detector_threshold_maximum_change) * t, -1.0); t = max(t, float(c.detector_threshold_minimum))
            elif delta > c.target_number_of_keypoints_tolerance:
                t = t + max(min(delta, c.detector_threshold_maximum_change) * t, 1.0); t = min(t, float(c.detector_threshold_maximum))
            self.thr[r] = int(np.rint(t / 1))                   # adjustDetectorThresholds over ONE detection (:440-459)
        if orb_detector:
            kps = np.concatenate(parts) if parts else np.zeros((0, 6), np.float32)
            if self.p.descriptor_type == 1:                     # ORB::compute: the keypoint's pyramid level and angle
                keep, desc = api.orb_describe_keypoints(left, kps, 1.2)
            else:                                               # BriefDescriptorExtractor: level 0, pixel (int)(pt + 0.5)
                keep, desc = api.brief_describe(left, np.floor(kps[:, :2] + np.float32(0.5)).astype(np.int16))
            sel = keep.astype(bool)
            new_xy = kps[sel][:, :2].astype(np.float32); new_desc = desc[sel]; new_level = kps[sel][:, 5].astype(np.int32)
        else:
            xy = (np.concatenate(parts) if parts else np.zeros((0, 2), np.int32)).astype(np.int16)
            keep, desc = (api.orb_describe(left, xy, -1.0) if self.p.descriptor_type == 1 else api.brief_describe(left, xy))
            sel = keep.astype(bool)
            new_xy = xy[sel].astype(np.float32); new_desc = desc[sel]; new_level = np.zeros(int(sel.sum()), np.int32)
        # the frame's keypoint vector: earlier attempts' keypoints (already filtered and described: the extractor gives them the same
        # descriptors again) followed by this detection's; ORB::compute regroups by level when the vector is not level-sorted
        self.acc_xy = np.concatenate([self.acc_xy, new_xy.reshape(-1, 2)]); self.acc_desc = np.concatenate([self.acc_desc, new_desc.reshape(-1, 32)])
        self.acc_level = np.concatenate([self.acc_level, new_level])
        self.detections = ([] if first else self.detections) + [len(new_level)]   # keypoints each initialize() of the frame added (kept by the extractor's border filter)
        if self.p.descriptor_type == 1 and len(self.acc_level) and np.any(np.diff(self.acc_level) < 0):
            o = np.argsort(self.acc_level, kind="stable")
            self.acc_xy, self.acc_desc, self.acc_level = self.acc_xy[o], self.acc_desc[o], self.acc_level[o]
        self.feat_xy = self.acc_xy.copy(); self.feat_desc = self.acc_desc.copy()
        # IntensityFeature: row = (int)pt.y, col = (int)pt.x (frame_point.h:18-35)
        self.feat_rc = np.stack([self.feat_xy[:, 1], self.feat_xy[:, 0]], axis=1).astype(np.int32) if len(self.feat_xy) else np.zeros((0, 2), np.int32)
        self.matched = np.zeros(len(self.feat_xy), bool)
        self.n_detected = len(self.feat_xy)

    # -- PoseTracker3D::_track + DepthFramePointGenerator::track ---------------------------------------------------------------------
    def track(self, cur, prev, by_appearance):
        api, c = self.api, self.cfg
        if by_appearance:
            self.win = c.maximum_projection_tracking_distance_pixels
        prevlist = prev.points + prev.temps
        cam = np.array([q.cam for q in prevlist]).reshape(-1, 3)
        pdesc = np.array([q.desc for q in prevlist], np.uint8).reshape(-1, 32)
        flags = np.array([(1 if q.landmark is not None else 0) | (2 if q.unreliable else 0) for q in prevlist], np.uint8)
        tr, xyz, tmp, lost, nlm = api.depth_track(self.p, self.space, self.prior, self.win, c.minimum_descriptor_distance_tracking,
                                                  1 if by_appearance else 0, cam, pdesc, flags, self.feat_rc, self.feat_desc)
        cur.points = []
        self.matched[:] = False                              # a fresh feature store per initialize(); matched = tracked or temporary
        for (ip, f), x in zip(tr, xyz):
            cur.points.append(Point(self.feat_xy[f], self.feat_desc[f], x, cur.index, previous=prevlist[ip]))
            self.matched[f] = True
        for ip, f in tmp:
            cur.temps.append(Point(self.feat_xy[f], self.feat_desc[f], (0, 0, 0), cur.index, previous=prevlist[ip], unreliable=True))
            self.matched[f] = True
        # a previous point that an earlier registration attempt of this frame linked keeps its next() (never reset: quirk B.5),
        # so it is not reported lost now even if this attempt did not find it
        self.lost = [prevlist[i] for i in lost if prevlist[i].next is None]
        self.n_tracked_landmarks = int(nlm)
        self.n_tracked = len(cur.points)
        ratio = _div(float(self.n_tracked), float(len(prev.points)))
        lm_per_point = _div(float(self.n_tracked_landmarks), float(self.n_tracked))
        success = float(self.n_tracked) / self.target
        wmax, wmin = c.maximum_projection_tracking_distance_pixels, c.minimum_projection_tracking_distance_pixels
        if ratio < c.good_tracking_ratio / 2:
            if self.win < wmax:
                self.win = int(min(self.win * 1 / c.tunnel_vision_ratio, float(wmax)))
        elif self.win > wmin:
            self.win = int(max(self.win * c.tunnel_vision_ratio, float(wmin)))
        if ratio < c.good_tracking_ratio or self.n_tracked < c.aligner_minimum_number_of_inliers or (lm_per_point < 0.5 and success < 0.25):
            self.tau_track = min(self.tau_track + 5, c.maximum_descriptor_distance_tracking)
        else:
            self.tau_track = max(self.tau_track - 5, c.minimum_descriptor_distance_tracking)
        self.aligner_valid = False
        self.info["track_attempts"] = self.info.get("track_attempts", 0) + 1

    # -- UVDAligner::initialize + converge ----------------------------------------------------------------------------------------------
    def align(self, cur, inverse_depth):
        c = self.cfg
        n = len(cur.points)
        if n < len(self.weights):
            del self.weights[n:]
        else:
            self.weights.extend([1.0] * (n - len(self.weights)))
        moving, fixed = np.zeros((n, 3)), np.zeros((n, 3))
        w_uv, w_d = np.ones(n), 10.0 * np.ones(n)
        for u, q in enumerate(cur.points):
            fixed[u] = (float(q.xy[0]), float(q.xy[1]), q.cam[2])
            moving[u] = q.previous.cam
            if q.unreliable:
                self.weights[u] = 0.0; w_d[u] = 0.0
            elif inverse_depth:
                self.weights[u] = c.maximum_reliable_depth_meters / q.cam[2]
        r = self.api.align_points_uvd(moving, fixed, w_uv, w_d, np.array(self.weights, np.float64), self.prior)
        self.al = r
        self.aligner_valid = True
        return r

    def accept(self, cur, prev):
        c, T = self.cfg, self.al["T"]
        dt = math.sqrt((T[0, 3] * T[0, 3] + T[1, 3] * T[1, 3]) + T[2, 3] * T[2, 3])
        if _rotation_angle(T) > c.minimum_delta_angular_for_movement or dt > c.minimum_delta_translational_for_movement:
            self.prior = T.copy()
            cur.set_pose(_mul(prev.c2w, _inv(self.prior)))
        else:
            self.fallback(cur, prev)

    def fallback(self, cur, prev):
        self.prior = np.hstack([np.eye(3), np.zeros((3, 1))]); cur.set_pose(prev.c2w); self.info["fallback"] = 1

    def break_track(self, cur, prev):
        self.status = LOCALIZING; cur.set_pose(prev.c2w); self.prior = np.hstack([np.eye(3), np.zeros((3, 1))])
        self.n_tracked = 0; self.info["track_broken"] = 1

    def register_recursive(self, cur, prev, left, depth, recursion):
        c = self.cfg
        rel = _div(float(self.n_tracked_landmarks), float(self.n_lm_prev))
        if self.n_tracked_landmarks == 0 or rel < 0.1:
            if recursion < 2:
                self.prior = np.hstack([np.eye(3), np.zeros((3, 1))])
                self.initialize(left, depth, first=False); self.track(cur, prev, True); self.register_recursive(cur, prev, left, depth, recursion + 1)
            else:
                self.break_track(cur, prev)
            return
        r = self.align(cur, True)
        if r["n_inliers"] > c.minimum_number_of_landmarks_to_track:
            self.accept(cur, prev)
        elif recursion < 2:
            if self.win < c.maximum_projection_tracking_distance_pixels:
                self.win += 1
            self.initialize(left, depth, first=False); self.track(cur, prev, False); self.register_recursive(cur, prev, left, depth, recursion + 1)
        else:
            self.break_track(cur, prev)

    # -- _prunePoints -------------------------------------------------------------------------------------------------------------------------
    def prune(self, cur):
        c = self.cfg
        kept = []
        if not self.aligner_valid:                          # defined behaviour: no fresh aligner result -> all dropped
            for q in cur.points:
                q.clear()
            cur.points = []
            return
        n = len(cur.points)
        avg = self.al["total_error"] / n if n else float("nan")
        for u, q in enumerate(cur.points):
            if avg < c.aligner_maximum_error_kernel:
                keep = bool(self.al["inlier"][u])
            else:
                keep = self.al["chi"][u] != -1 and self.al["chi"][u] < 100 * c.aligner_maximum_error_kernel
            if keep:
                kept.append(q)
            else:
                q.clear()
        cur.points = kept

    # -- DepthFramePointGenerator::recoverPoints ----------------------------------------------------------------------------------------------
    def recover(self, cur, left):
        lost = self.lost
        if not lost:
            return 0
        has_lm = np.array([1 if q.landmark is not None else 0 for q in lost], np.uint8)
        lm = np.array([q.landmark.world if q.landmark is not None else (0, 0, 0) for q in lost], np.float64).reshape(-1, 3)
        pd = np.array([q.desc for q in lost], np.uint8).reshape(-1, 32)
        idx, xy, desc, xyz = self.api.depth_recover(self.p, self.space, left, cur.w2c, has_lm, lm, pd, 7.0, self.cfg.minimum_descriptor_distance_tracking)
        for k in range(len(idx)):
            cur.points.append(Point(xy[k], desc[k], xyz[k], cur.index, previous=lost[idx[k]]))
        return len(idx)

    # -- _updatePoints ----------------------------------------------------------------------------------------------------------------------------
    def update_points(self, cur):
        c = self.cfg
        todo = []                                           # (landmark, point) whose landmark is refined with the new measurement
        active = 0
        for q in cur.points:
            if q.track_len < c.minimum_track_length_for_landmark_creation or q.unreliable:
                continue
            lm = q.origin.landmark
            if lm is None:                                  # Landmark::Landmark: mean of the track's world coordinates
                lm = Landmark()
                self.landmarks.append(lm)
                acc = np.zeros(3)
                t = q
                chain = []
                while t is not None:
                    t.landmark = lm; chain.append(t); t = t.previous
                for t in chain:                              # newest first, as the constructor walks the chain
                    lm.meas.append((t.frame, t.cam.copy()))
                    acc = acc + _apply(self.frames[t.frame].c2w, t.cam)
                lm.world = acc / len(chain); lm.updates = len(chain)
            else:
                todo.append((lm, q))
            active += 1
        if todo:
            frames_used = sorted({f for lm, q in todo for f, _ in lm.meas} | {cur.index})
            remap = {f: i for i, f in enumerate(frames_used)}
            w2c = np.array([self.frames[f].w2c for f in frames_used]); c2w = np.array([self.frames[f].c2w for f in frames_used])
            offsets, frame_of, cams = [0], [], []
            for lm, q in todo:
                for f, cam in lm.meas:
                    frame_of.append(remap[f]); cams.append(cam)
                frame_of.append(remap[cur.index]); cams.append(q.cam)
                offsets.append(len(frame_of))
            world = np.array([lm.world for lm, q in todo]); upd = np.array([lm.updates for lm, q in todo], np.int32)
            w, u = self.api.landmark_update(c, np.array(offsets, np.int32), np.array(frame_of, np.int32), w2c, c2w, np.array(cams), world, upd)
            for k, (lm, q) in enumerate(todo):
                lm.world = w[k].copy(); lm.updates = int(u[k]); lm.meas.append((cur.index, q.cam.copy())); q.landmark = lm
        self.n_active = active
        # temporary points: midpoint triangulation with the refined motion (:524-545)
        if cur.temps:
            xp = np.array([q.previous.xy for q in cur.temps], np.float32); xc = np.array([q.xy for q in cur.temps], np.float32)
            tri = self.api.point_in_camera(xp, xc, self.prior, self.K)
            kept = []
            for q, x in zip(cur.temps, tri):
                if x[2] <= 0:
                    continue
                q.cam = x.copy(); kept.append(q)
            cur.temps = kept

    # -- DepthFramePointGenerator::compute ----------------------------------------------------------------------------------------------------------
    def compute(self, cur):
        remaining = np.nonzero(~self.matched)[0]
        rc = self.feat_rc[remaining]
        tracked_rc = np.array([(q.row, q.col) for q in cur.points], np.int32).reshape(-1, 2)
        new, xyz, temp, txyz = self.api.depth_compute(self.p, self.space, rc, tracked_rc)
        for k, f in enumerate(new):
            g = remaining[f]
            cur.points.append(Point(self.feat_xy[g], self.feat_desc[g], xyz[k], cur.index))
        for k, f in enumerate(temp):
            g = remaining[f]
            cur.temps.append(Point(self.feat_xy[g], self.feat_desc[g], txyz[k], cur.index, unreliable=True))
        return len(new), len(temp)

    # -- PoseTracker3D::compute -------------------------------------------------------------------------------------------------------------------------
    def process(self, left, depth):
        c = self.cfg
        self.info = {"status_at_start": self.status, "fallback": 0, "track_broken": 0, "track_attempts": 0}
        prev = self.frames[-1] if self.frames else None
        cur = Frame(len(self.frames), self.world)
        self.frames.append(cur)
        self.n_tracked = 0; self.n_tracked_landmarks = 0; self.aligner_valid = False; self.lost = []
        self.initialize(left, depth)
        if prev is not None:
            for q in prev.points + prev.temps:
                q.next = None
            self.track(cur, prev, self.status == LOCALIZING)
            if self.status == LOCALIZING:
                if self.n_tracked < c.minimum_number_of_landmarks_to_track:
                    self.fallback(cur, prev)
                else:
                    r = self.align(cur, False)
                    if r["n_inliers"] < c.minimum_number_of_landmarks_to_track:
                        self.fallback(cur, prev)
                    else:
                        self.accept(cur, prev)
            else:
                self.register_recursive(cur, prev, left, depth, 0)
        self.world = cur.c2w.copy()
        info = self.info
        info["n_tracked"] = len(cur.points); info["n_lost"] = len(self.lost); info["n_tracked_landmarks"] = self.n_tracked_landmarks
        info["aligner_ran"] = int(self.aligner_valid)
        info["n_inliers"] = int(self.al["n_inliers"]) if self.aligner_valid else 0
        info["aligner_iterations"] = int(self.al["iterations"]) if self.aligner_valid else 0
        n_rec = 0
        if prev is not None:
            self.prune(cur)
            info["n_after_prune"] = len(cur.points)
            if c.enable_landmark_recovery:
                n_rec = self.recover(cur, left)
        else:
            info["n_after_prune"] = 0
        info["n_recovered"] = n_rec
        self.update_points(cur)
        if self.n_active > c.minimum_number_of_landmarks_to_track:
            self.status = TRACKING
        n_new, n_temp_new = self.compute(cur)
        self.n_lm_prev = self.n_active
        info.update(status=self.status, n_keypoints=self.n_detected, n_active_landmarks=self.n_active, n_new=n_new, n_points=len(cur.points),
                    n_temporary=len(cur.temps), threshold=self.thr[0], thresholds=list(self.thr), window_pixels=self.win, tau_track=self.tau_track, pose=cur.c2w.copy(),
                    prior=self.prior.copy())
        return info
