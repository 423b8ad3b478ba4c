"""RGB-D mode end to end: PoseTracker3D with a DepthFramePointGenerator and a UVDAligner.

Two independent statements of the same tracker are compared on rendered image + depth sequences:
  * the product (vslam_rgbd_*): the device-resident loop inside libvslam_hip.so (csrc/rgbd_device.h + kernels_rgbd.h: the tracker's state
    stays in HBM, a frame is one launch sequence and one small read-back), and its second implementation, the C++ host loop over the
    library's stand-alone device entry points (csrc/rgbd_tracker.h, VSLAM_RGBD_HOST=1) — both run here;
  * the checker: tests/rgbd_loop.py, a plain Python loop, run over the CPU oracle's stand-alone functions.
Counters and point lists must agree exactly, poses within 1e-4 relative Frobenius (north_star).  The Python loop is also run
over the HIP entry points (same control flow, device kernels), which separates a kernel difference from a loop difference."""
import numpy as np
import pytest

from rgbd_loop import RgbdTracker as PyLoop
from vslam_pose_estimation_framework_amd import hip
from vslam_pose_estimation_framework_amd.capi import DepthParams, RgbdBatch, RgbdTracker

POSE_RTOL = 1e-4


# The three RGB-D configurations of the reference (configurations/configuration_{icl,tum,xtion}.yaml), every value of their
# base_framepoint_generation / depth_framepoint_generation / tracking / landmark sections that the path reads.  The synthetic scene is
# a street, not a room: the three metric depth limits are the yaml's x DEPTH_SCALE (reliable 2.5 m -> 10 m, maximum 10 m -> 40 m),
# everything else is the yaml's number.  xtion's motion_model CAMERA_ODOMETRY needs an odometry input the path does not have:
# CONSTANT_VELOCITY is used (pose_tracker_3d.cpp:44-48), as for the other two.
DEPTH_SCALE = 4.0
YAML = {
    "icl": dict(thr=(5, 100), max_change=1.0, grid=(2, 2), win=(5, 25), desc=(25, 50), depth=(2.5, 10.0, 0.001), bin=25, tri=0,
                lm_err=0.5, min_lm=10, tunnel=0.5, good=0.25, delta_move=(0.0, 0.0), kernel=5),
    "tum": dict(thr=(10, 100), max_change=0.5, grid=(1, 1), win=(10, 50), desc=(40, 40), depth=(2.5, 10.0, 0.1), bin=15, tri=1,
                lm_err=1.0, min_lm=5, tunnel=0.75, good=0.25, delta_move=(0.001, 0.01), kernel=10),
    "xtion": dict(thr=(10, 100), max_change=0.5, grid=(1, 1), win=(5, 10), desc=(25, 50), depth=(2.5, 8.0, 0.1), bin=10, tri=1,
                  lm_err=4.0, min_lm=25, tunnel=0.75, good=0.5, delta_move=(0.001, 0.01), kernel=10),
}


def setup(o, which="tum", scale=0.5, descriptor=1, seed=23, max_depth=None):
    """`which` configuration's values on the synthetic street scene; depth beyond the maximum reads as "no measurement", so far
    features become temporary points where the configuration triangulates them (tum, xtion) and are skipped where it does not (icl)."""
    y = YAML[which]
    scene = o.scene_kitti(scale=scale, seed=seed)
    scene.speed_m = 0.25; scene.sway_m = 0.4
    if which == "xtion":
        scene.speed_m = 0.08; scene.sway_m = 0.15          # a 10 px search window: hand-held sensor speeds
    cfg = o.config_for_scene(scene)
    cfg.det_rows, cfg.det_cols = y["grid"]
    cfg.detector_threshold_minimum, cfg.detector_threshold_maximum = y["thr"]
    cfg.detector_threshold_maximum_change = y["max_change"]; cfg.target_number_of_keypoints_tolerance = 0.1
    cfg.minimum_projection_tracking_distance_pixels, cfg.maximum_projection_tracking_distance_pixels = y["win"]
    cfg.minimum_descriptor_distance_tracking, cfg.maximum_descriptor_distance_tracking = y["desc"]
    max_depth = y["depth"][1] * DEPTH_SCALE if max_depth is None else max_depth
    cfg.maximum_reliable_depth_meters = y["depth"][0] * DEPTH_SCALE; cfg.maximum_depth_meters = max_depth; cfg.minimum_depth_meters = y["depth"][2]
    cfg.enable_keypoint_binning = 1; cfg.bin_size_pixels = y["bin"]
    cfg.minimum_track_length_for_landmark_creation = 2; cfg.minimum_number_of_landmarks_to_track = y["min_lm"]
    cfg.tunnel_vision_ratio = y["tunnel"]; cfg.good_tracking_ratio = y["good"]
    cfg.minimum_delta_angular_for_movement, cfg.minimum_delta_translational_for_movement = y["delta_move"]
    cfg.aligner_error_delta_for_convergence = 1e-5; cfg.aligner_maximum_error_kernel = y["kernel"]; cfg.aligner_damping = 0
    cfg.aligner_maximum_number_of_iterations = 1000; cfg.aligner_minimum_number_of_inliers = 0
    cfg.landmark_maximum_error_squared_meters = y["lm_err"]
    cfg.enable_landmark_recovery = 1
    cfg.descriptor_type = descriptor
    K = np.array([[scene.fx, 0, scene.cx], [0, scene.fy, scene.cy], [0, 0, 1.0]])
    p = DepthParams.make(scene.rows, scene.cols, K, np.linalg.inv(K), np.linalg.inv(K), np.eye(4)[:3], 2e-3, y["depth"][2], max_depth, y["tri"], 1,
                         y["bin"], descriptor)
    return scene, cfg, p


def test_python_loop_over_the_oracle_tracks_the_scene():
    """CPU: the checker loop itself locks on, keeps landmarks, recovers points, carries temporary points and follows the ground
    truth (so that agreeing with it means something)."""
    from _oracle import Oracle
    o = Oracle()
    scene, cfg, p = setup(o)
    o.create(cfg, 0, 1)
    tr = PyLoop(o, cfg, p)
    g0 = o.gt_pose(scene, 0)
    rec = tmp = 0
    for k in range(10):
        L, _ = o.render(scene, k)
        info = tr.process(L, o.render_depth(scene, k, 2e-3))
        rec += info["n_recovered"]; tmp += info["n_temporary"]
    G = o.gt_pose(scene, 9)
    Grel = np.hstack([g0[:, :3].T @ G[:, :3], (g0[:, :3].T @ (G[:, 3] - g0[:, 3]))[:, None]])
    assert info["status"] == 1 and info["n_tracked"] > 50 and info["n_active_landmarks"] > 50
    assert rec > 20 and tmp > 5
    assert np.abs(info["pose"] - Grel).max() < 0.1          # 2.25 m of path
    o.destroy()


@pytest.mark.parametrize("which", ["icl", "tum", "xtion"])
def test_python_loop_over_the_oracle_runs_every_configuration(which):
    """CPU: the checker loop over the oracle with each configuration's values (icl: 2 x 2 detector grid, no triangulation of
    points without depth) locks on and keeps landmarks; with a 2 x 2 grid the four thresholds move on their own."""
    from _oracle import Oracle
    o = Oracle()
    scene, cfg, p = setup(o, which)
    o.create(cfg, 0, 1)
    tr = PyLoop(o, cfg, p)
    assert len(tr.regions) == cfg.det_rows * cfg.det_cols
    if which == "icl":     # 620 x 188 cut 2 x 2: overlaps of 2 px towards the neighbours (A.1 of SURVEY.md)
        assert tr.regions == [(0, 0, 312, 96), (308, 0, 312, 96), (0, 92, 312, 96), (308, 92, 312, 96)]
    tmp = 0
    for k in range(8):
        L, _ = o.render(scene, k)
        info = tr.process(L, o.render_depth(scene, k, 2e-3))
        tmp += info["n_temporary"]
    assert info["status"] == 1 and info["n_tracked"] > 30 and info["n_active_landmarks"] > 30, info
    assert (tmp > 0) == bool(YAML[which]["tri"])
    if which == "icl":
        assert len(set(info["thresholds"])) > 1, info["thresholds"]
    o.destroy()


def test_python_loop_over_the_oracle_keeps_the_union_of_keypoints_over_reregistrations():
    """CPU: a jump in the sequence loses the track — _registerRecursive calls initialize() twice more (pose_tracker_3d.cpp:320,402), detectKeypoints
    appends to the frame's keypoint vector every time (base_framepoint_generator.cpp:422) and setFeatures stores all of them: the checker's feature
    list of such a frame is the union of three detections, corners found again sit on one pixel twice (only the last one is in the lattice), and
    compute() turns features of every attempt into framepoints."""
    from _oracle import Oracle
    o = Oracle()
    scene, cfg, p = setup(o, "icl", descriptor=0, max_depth=30.0, seed=41)
    cfg.minimum_number_of_landmarks_to_track = 30
    o.create(cfg, 0, 1)
    tr = PyLoop(o, cfg, p)
    try:
        for k in [0, 1, 2, 3, 4, 5, 6, 7, 8]:
            info = tr.process(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3))
            assert len(tr.detections) == max(info["track_attempts"], 1) and info["n_keypoints"] == sum(tr.detections)
        info = tr.process(o.render(scene, 16)[0], o.render_depth(scene, 16, 2e-3))            # the jump
        rc = tr.feat_rc
        assert info["track_attempts"] == 3 and info["track_broken"] == 1, info
        assert len(tr.detections) == 3 and min(tr.detections) > 100 and info["n_keypoints"] == len(rc) == sum(tr.detections), (info["n_keypoints"], tr.detections)
        pixels = rc[:, 0].astype(np.int64) * 100000 + rc[:, 1]
        assert len(np.unique(pixels)) < len(rc)                                              # the same corner, detected again
        new = [q for q in tr.frames[-1].points if q.previous is None]
        pts = np.array([q.row * 100000 + q.col for q in new], np.int64)
        assert len(new) == info["n_new"] > 0 and len(np.unique(pts)) <= len(pts)
    finally:
        o.destroy()


def test_python_loop_over_the_oracle_with_the_orb_detector():
    """The checker loop itself with detector_type ORB (CPU only): keypoints of several pyramid levels, ORB::compute on their own level and
    angle, features sharing pixels — the scene is tracked."""
    from _oracle import Oracle
    o = Oracle()
    scene, cfg, p = setup(o, "tum", descriptor=1, seed=67)
    p.detector_type = 1
    o.create(cfg, 0, 1)
    tr = PyLoop(o, cfg, p)
    try:
        for k in range(5):
            info = tr.process(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3))
        rc = tr.feat_rc
        assert info["status"] == 1 and info["n_tracked"] > 40 and info["n_keypoints"] > 400, info
        assert len(rc) > len(np.unique(rc[:, 0].astype(np.int64) * 100000 + rc[:, 1]))     # several features on one pixel
        assert not np.array_equal(tr.feat_xy, np.floor(tr.feat_xy))                           # sub-pixel keypoints of the higher levels
    finally:
        o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["device", "host"])
@pytest.mark.parametrize("which,descriptor,max_depth,seed", [("tum", 1, None, 23), ("tum", 0, 25.0, 31), ("icl", 1, None, 29), ("xtion", 1, None, 37)])
def test_rgbd_tracker_matches_the_checker_loop(which, descriptor, max_depth, seed, impl, monkeypatch):
    from _oracle import Oracle
    monkeypatch.setenv("VSLAM_RGBD_HOST", "1" if impl == "host" else "0")
    # the product loop compacts its point pool every 4 frames here (32 by default): three compactions inside the 12 frames, the
    # checker loop keeps everything — dropping the unreachable points and landmarks must not change a single result
    monkeypatch.setenv("VSLAM_RGBD_COMPACT", "4")
    o = Oracle()
    scene, cfg, p = setup(o, which, descriptor=descriptor, max_depth=max_depth, seed=seed)
    o.create(cfg, 0, 1)
    g = hip.load()
    g.create(cfg, 0, 1)
    ref = PyLoop(o, cfg, p)                 # Python loop over the oracle
    mid = PyLoop(g, cfg, p)                 # Python loop over the HIP entry points
    prod = RgbdTracker(g, cfg, p)           # the product: C++ loop inside libvslam_hip.so
    try:
        seen_temp = seen_rec = 0
        for k in range(12):
            L, _ = o.render(scene, k)
            D = o.render_depth(scene, k, 2e-3)
            a = ref.process(L, D)
            b = mid.process(L, D)
            fi, n_temp = prod.process(L, D)
            for name, field in (("status", "status"), ("n_keypoints", "n_keypoints_left"), ("n_tracked", "n_tracked"), ("n_lost", "n_lost"),
                                ("n_tracked_landmarks", "n_tracked_landmarks"), ("aligner_ran", "aligner_ran"), ("n_inliers", "n_inliers"),
                                ("aligner_iterations", "aligner_iterations"), ("n_after_prune", "n_after_prune"), ("n_recovered", "n_recovered"),
                                ("n_active_landmarks", "n_active_landmarks"), ("n_new", "n_new_stereo"), ("n_points", "n_points"),
                                ("window_pixels", "window_pixels"), ("track_attempts", "track_attempts"), ("fallback", "fallback"),
                                ("track_broken", "track_broken"), ("status_at_start", "status_at_start")):
                assert a[name] == b[name] == getattr(fi, field), (k, name, a[name], b[name], getattr(fi, field))
            assert a["thresholds"] == b["thresholds"] == list(fi.thresholds)[:len(a["thresholds"])] and a["n_temporary"] == b["n_temporary"] == n_temp
            assert a["tau_track"] == fi.tau_track
            To, Tg = a["pose"], np.array(fi.camera_left_to_world).reshape(3, 4)
            assert np.linalg.norm(Tg - To) / np.linalg.norm(To) <= POSE_RTOL
            assert np.linalg.norm(b["pose"] - To) / np.linalg.norm(To) <= POSE_RTOL
            pts = prod.points()
            cur = ref.frames[-1]
            assert len(pts["xy"]) == len(cur.points)
            prevlist = (ref.frames[-2].points + ref.frames[-2].temps) if k else []
            for i, q in enumerate(cur.points):
                assert np.array_equal(pts["xy"][i].view(np.uint32), q.xy.view(np.uint32)), (k, i)
                np.testing.assert_array_equal(pts["desc"][i], q.desc)
                np.testing.assert_allclose(pts["cam"][i], q.cam, rtol=1e-12, atol=0)
                want_prev = prevlist.index(q.previous) if q.previous is not None else -1
                assert pts["meta"][i, 0] == want_prev and pts["meta"][i, 1] == q.track_len and pts["meta"][i, 3] == int(q.unreliable)
                assert pts["meta"][i, 2] == (q.landmark.updates if q.landmark is not None else 0)
            seen_temp += n_temp; seen_rec += fi.n_recovered
        assert fi.status == 1 and fi.n_tracked > 30 and (seen_temp > 0) == bool(YAML[which]["tri"]) and seen_rec > 0
    finally:
        prod.destroy(); g.destroy(); o.destroy()


@pytest.mark.gpu
def test_rgbd_tracker_argument_errors():
    from _oracle import Oracle
    from vslam_pose_estimation_framework_amd.capi import VslamError
    o = Oracle()
    scene, cfg, p = setup(o)
    g = hip.load()
    bad = cfg.copy(); bad.det_rows = 0
    with pytest.raises(VslamError):
        RgbdTracker(g, bad, p)
    t = RgbdTracker(g, cfg, p)
    rc = g.lib.vslam_rgbd_process_host(t.h, None, 0, None, 0)
    assert rc == -1 and b"empty frame" in g.lib.vslam_rgbd_last_error(t.h)
    t.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["device", "host"])
def test_rgbd_degenerate_inputs(impl, monkeypatch):
    """Featureless frames, a depth image without a single measurement (every feature becomes a temporary point), a depth image
    that comes back, a scene cut: product loop and checker loop stay identical and nothing raises."""
    from _oracle import Oracle
    monkeypatch.setenv("VSLAM_RGBD_HOST", "1" if impl == "host" else "0")
    o = Oracle()
    scene, cfg, p = setup(o, "tum", descriptor=0, max_depth=30.0, seed=41)
    scene2 = o.scene_kitti(scale=0.5, seed=977)
    o.create(cfg, 0, 1)
    g = hip.load()
    g.create(cfg, 0, 1)
    ref = PyLoop(o, cfg, p)
    prod = RgbdTracker(g, cfg, p)
    blank = np.full((cfg.rows, cfg.cols), 90, np.uint8)
    nodepth = np.zeros((cfg.rows, cfg.cols), np.uint16)
    frames = [(blank, nodepth)]
    for k in range(3):
        frames.append((o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)))
    frames.append((o.render(scene, 3)[0], nodepth))                      # images go on, the depth sensor drops out
    frames.append((o.render(scene, 4)[0], o.render_depth(scene, 4, 2e-3)))
    for k in range(3):
        frames.append((o.render(scene2, 50 + k)[0], o.render_depth(scene2, 50 + k, 2e-3)))   # another world: the track is lost
    frames.append((blank, nodepth))
    try:
        temps = 0
        for k, (L, D) in enumerate(frames):
            a = ref.process(L, D)
            fi, n_temp = prod.process(L, D)
            for name, field in (("status", "status"), ("n_keypoints", "n_keypoints_left"), ("n_tracked", "n_tracked"), ("n_inliers", "n_inliers"),
                                ("n_after_prune", "n_after_prune"), ("n_recovered", "n_recovered"), ("n_active_landmarks", "n_active_landmarks"),
                                ("n_new", "n_new_stereo"), ("n_points", "n_points"), ("track_attempts", "track_attempts"),
                                ("track_broken", "track_broken"), ("fallback", "fallback")):
                assert a[name] == getattr(fi, field), (k, name, a[name], getattr(fi, field))
            assert a["n_temporary"] == n_temp
            To, Tg = a["pose"], np.array(fi.camera_left_to_world).reshape(3, 4)
            assert np.linalg.norm(Tg - To) / np.linalg.norm(To) <= POSE_RTOL
            temps = max(temps, n_temp)
        assert temps > 100          # the frame without depth: features without a measurement are carried as temporary points
    finally:
        prod.destroy(); g.destroy(); o.destroy()


def _run(prod, frames):
    out = []
    for L, D in frames:
        fi, nt = prod.process(L, D)
        out.append((fi.status, fi.n_keypoints_left, fi.n_tracked, fi.n_lost, fi.n_inliers, fi.aligner_iterations, fi.n_after_prune, fi.n_recovered,
                    fi.n_active_landmarks, fi.n_new_stereo, fi.n_points, fi.track_attempts, fi.window_pixels, fi.tau_track, nt, list(fi.thresholds)[:4],
                    tuple(fi.camera_left_to_world)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("graph", ["0", "1", "general-depth", "gated-depth"])
def test_rgbd_device_loop_strides_reset_and_second_tracker(graph, monkeypatch):
    """The device-resident loop with padded rows (image stride != width, depth stride != width: the caller's strides are kept on the device / the
    depth image is re-packed), after reset(), and as a second tracker in the same process: every frame's counters and pose identical to the dense
    first run, bit for bit.  graph = 1: the frame's launch sequence replayed from a captured hipGraph (captured again when the stride changes)."""
    from _oracle import Oracle
    monkeypatch.setenv("VSLAM_RGBD_HOST", "0")
    monkeypatch.setenv("VSLAM_RGBD_GRAPH", graph if graph in ("0", "1") else "0")
    # the space map: the direct pass (default where every depth pixel projects onto itself), the general z-buffer alone, and the general passes
    # opened behind the direct one as if a crossing source had been seen — all three must give the first run's results
    o = Oracle()
    scene, cfg, p = setup(o, "tum", descriptor=1, seed=53)
    g = hip.load()
    frames = [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(10)]
    ref = RgbdTracker(g, cfg, p)                   # the default paths (created before the switches below are set)
    baseline = _run(ref, frames)
    ref.destroy()
    if graph == "general-depth":
        monkeypatch.setenv("VSLAM_RGBD_DEPTH_DIRECT", "0")
    if graph == "gated-depth":
        monkeypatch.setenv("VSLAM_RGBD_DEPTH_FORCE_CROSS", "1")
    padded = []
    for L, D in frames:
        Lp = np.full((cfg.rows, cfg.cols + 12), 7, np.uint8); Lp[:, :cfg.cols] = L
        Dp = np.full((cfg.rows, cfg.cols + 5), 999, np.uint16); Dp[:, :cfg.cols] = D
        padded.append((Lp, Dp))
    a = RgbdTracker(g, cfg, p)
    b = RgbdTracker(g, cfg, p)
    try:
        first = _run(a, frames)
        assert first == baseline
        assert first[-1][0] == 1 and first[-1][2] > 50
        a.reset()
        # padded rows: process() passes the array width as the stride, the image width comes from the configuration
        again = _run(a, padded)
        assert again == first
        assert _run(b, frames) == first
        pa, pb = a.points(), b.points()
        for name in ("xy", "cam", "meta", "desc"):
            np.testing.assert_array_equal(pa[name], pb[name])
    finally:
        a.destroy(); b.destroy(); o.destroy()


@pytest.mark.gpu
def test_rgbd_device_loop_capacity_and_short_history(monkeypatch):
    """Capacities of the device-resident loop: more points than max_points fails the frame with VSLAM_ERR_CAPACITY and the tracker refuses
    further frames until reset(); a history ring shorter than the tracks raises bit 2 of error_flags (oldest measurements left out of the
    landmark refinement, as in the stereo tracker) and the tracker carries on."""
    from _oracle import Oracle
    from vslam_pose_estimation_framework_amd.capi import VslamError, ERR_CAPACITY, ERR_STATE
    monkeypatch.setenv("VSLAM_RGBD_HOST", "0")
    o = Oracle()
    scene, cfg, p = setup(o, "tum", descriptor=0, seed=59)
    g = hip.load()
    frames = [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(12)]
    small = cfg.copy(); small.max_points = 64
    t = RgbdTracker(g, small, p)
    try:
        with pytest.raises(VslamError) as e:
            for L, D in frames[:3]:
                t.process(L, D)
        assert e.value.code == ERR_CAPACITY and "max_points" in str(e.value)
        with pytest.raises(VslamError) as e:
            t.process(*frames[0])
        assert e.value.code == ERR_STATE and "reset()" in str(e.value)
        t.reset()
        with pytest.raises(VslamError) as e:
            t.process(*frames[0])
        assert e.value.code == ERR_CAPACITY          # the same frame overflows again: the state was reset, the capacity was not
    finally:
        t.destroy()
    # a lost track: the frame's second / third detection is appended to its keypoint vector — the union must fit max_keypoints too.
    # With the detector threshold pinned every detection of a frame finds the same corners: the vector grows to 2 x and 3 x one detection.
    pinned = cfg.copy(); pinned.detector_threshold_minimum = pinned.detector_threshold_maximum = 25
    jump = (o.render(scene, 40)[0], o.render_depth(scene, 40, 2e-3))
    full = RgbdTracker(g, pinned, p)
    try:
        singles = [full.process(L, D)[0].n_keypoints_left for L, D in frames[:4]]
        fi, _ = full.process(*jump)
        assert fi.track_attempts == 3 and fi.n_keypoints_left % 3 == 0 and fi.n_keypoints_left > max(singles), (fi.track_attempts, fi.n_keypoints_left, singles)
        union = fi.n_keypoints_left
    finally:
        full.destroy()
    tight = pinned.copy(); tight.max_keypoints = max(max(singles), 2 * union // 3) + 1      # every frame and two detections of the jump frame fit, three do not
    assert tight.max_keypoints < union
    t = RgbdTracker(g, tight, p)
    try:
        for L, D in frames[:4]:
            t.process(L, D)
        with pytest.raises(VslamError) as e:
            t.process(*jump)
        assert e.value.code == ERR_CAPACITY and "max_keypoints" in str(e.value)
    finally:
        t.destroy()
    short = cfg.copy(); short.max_history_frames = 5
    t = RgbdTracker(g, short, p)
    try:
        flags = []
        for L, D in frames:
            fi, _ = t.process(L, D)
            flags.append(fi.error_flags)
        assert flags[3] == 0 and flags[-1] == 4 and fi.status == 1 and fi.n_tracked > 30 and fi.n_active_landmarks > 30, (flags, fi.status, fi.n_tracked)
    finally:
        t.destroy(); o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["device", "host"])
def test_rgbd_reregistration_paths(impl, monkeypatch):
    """_registerRecursive's branches (pose_tracker_3d.cpp:300-418) on the icl configuration with a stricter landmark minimum: a second
    attempt by projection with a wider window after too few aligner inliers (:377-399), a second attempt by appearance after too few tracked
    landmarks (:333-352), three attempts and breakTrack after a jump in the sequence — every attempt detects again with the thresholds the
    controller has moved meanwhile.  Product loop == checker loop on every counter."""
    from _oracle import Oracle
    monkeypatch.setenv("VSLAM_RGBD_HOST", "1" if impl == "host" else "0")
    o = Oracle()
    scene, cfg, p = setup(o, "icl", descriptor=0, max_depth=30.0, seed=41)
    cfg.minimum_number_of_landmarks_to_track = 30
    o.create(cfg, 0, 1)
    g = hip.load()
    ref = PyLoop(o, cfg, p)
    prod = RgbdTracker(g, cfg, p)
    try:
        seen, union = set(), []
        for k in [0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 17, 18]:
            L, D = o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)
            a = ref.process(L, D)
            fi, n_temp = prod.process(L, D)
            for name, field in (("status", "status"), ("n_keypoints", "n_keypoints_left"), ("n_tracked", "n_tracked"), ("n_lost", "n_lost"),
                                ("n_tracked_landmarks", "n_tracked_landmarks"), ("aligner_ran", "aligner_ran"), ("n_inliers", "n_inliers"),
                                ("aligner_iterations", "aligner_iterations"), ("n_after_prune", "n_after_prune"), ("n_recovered", "n_recovered"),
                                ("n_active_landmarks", "n_active_landmarks"), ("n_new", "n_new_stereo"), ("n_points", "n_points"),
                                ("window_pixels", "window_pixels"), ("track_attempts", "track_attempts"), ("fallback", "fallback"),
                                ("track_broken", "track_broken"), ("status_at_start", "status_at_start")):
                assert a[name] == getattr(fi, field), (k, name, a[name], getattr(fi, field))
            assert a["thresholds"] == list(fi.thresholds)[:len(a["thresholds"])] and a["n_temporary"] == n_temp and a["tau_track"] == fi.tau_track
            To, Tg = a["pose"], np.array(fi.camera_left_to_world).reshape(3, 4)
            assert np.linalg.norm(Tg - To) / np.linalg.norm(To) <= POSE_RTOL
            pts, cur = prod.points(), ref.frames[-1]          # the frame's points, one by one (new points of duplicated keypoints included)
            assert len(pts["xy"]) == len(cur.points)
            for i, q in enumerate(cur.points):
                assert np.array_equal(pts["xy"][i].view(np.uint32), q.xy.view(np.uint32)), (k, i)
                np.testing.assert_array_equal(pts["desc"][i], q.desc)
                np.testing.assert_allclose(pts["cam"][i], q.cam, rtol=1e-12, atol=0)
            if fi.track_attempts > 1:
                # keypointsLeft() is appended to by every initialize() of the frame (base_framepoint_generator.cpp:422): the union of the
                # attempts' detections, corners found again on the same pixel twice in it
                rc = ref.feat_rc
                union.append((fi.track_attempts, len(rc), len(rc) - len(np.unique(rc[:, 0].astype(np.int64) * 100000 + rc[:, 1]))))
                assert fi.n_keypoints_left == len(rc)
            if fi.track_attempts == 2 and fi.aligner_ran and fi.window_pixels < cfg.maximum_projection_tracking_distance_pixels // 2:
                seen.add("second attempt by projection")
            if fi.track_attempts == 2 and fi.aligner_ran and fi.window_pixels >= cfg.maximum_projection_tracking_distance_pixels // 2:
                seen.add("second attempt by appearance")
            if fi.track_attempts == 3 and fi.track_broken:
                seen.add("track broken after three attempts")
        assert seen == {"second attempt by projection", "second attempt by appearance", "track broken after three attempts"}, seen
        assert any(a == 3 for a, _, _ in union) and all(d > 0 for _, _, d in union), union     # every repeated detection found some corners again
    finally:
        prod.destroy(); o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("descriptor", [1, 0])
def test_rgbd_tracker_with_the_orb_detector(descriptor, monkeypatch):
    """detector_type ORB (base_framepoint_generator.cpp:52-70, :242-247): one cv::ORB::create(5000, 1.2, 8, 31, 0, 2, HARRIS_SCORE, 31, thr) per
    detector region with the controller on its FAST threshold, then the configured extractor — ORB::compute on the keypoints' own pyramid
    level and angle, or BRIEF at level 0 — and a feature lattice in which keypoints of several levels share pixels.  Product (the host-driven
    loop serves this mode) == the checker loop over the oracle, 2 x 2 detector grid, eight frames."""
    from _oracle import Oracle
    monkeypatch.setenv("VSLAM_RGBD_HOST", "0")       # the library switches to the host-driven loop by itself
    o = Oracle()
    scene, cfg, p = setup(o, "tum", descriptor=descriptor, seed=67)
    cfg.det_rows, cfg.det_cols = 2, 2
    p.detector_type = 1
    o.create(cfg, 0, 1)
    g = hip.load()
    ref = PyLoop(o, cfg, p)
    prod = RgbdTracker(g, cfg, p)
    try:
        shared = 0
        for k in range(8):
            L, D = o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)
            a = ref.process(L, D)
            fi, n_temp = prod.process(L, D)
            for name, field in (("status", "status"), ("n_keypoints", "n_keypoints_left"), ("n_tracked", "n_tracked"), ("n_lost", "n_lost"),
                                ("n_tracked_landmarks", "n_tracked_landmarks"), ("aligner_ran", "aligner_ran"), ("n_inliers", "n_inliers"),
                                ("aligner_iterations", "aligner_iterations"), ("n_after_prune", "n_after_prune"), ("n_recovered", "n_recovered"),
                                ("n_active_landmarks", "n_active_landmarks"), ("n_new", "n_new_stereo"), ("n_points", "n_points"),
                                ("window_pixels", "window_pixels"), ("track_attempts", "track_attempts"), ("fallback", "fallback"),
                                ("track_broken", "track_broken"), ("status_at_start", "status_at_start")):
                assert a[name] == getattr(fi, field), (k, name, a[name], getattr(fi, field))
            assert a["thresholds"] == list(fi.thresholds)[:len(a["thresholds"])] and a["n_temporary"] == n_temp and a["tau_track"] == fi.tau_track
            To, Tg = a["pose"], np.array(fi.camera_left_to_world).reshape(3, 4)
            assert np.linalg.norm(Tg - To) / np.linalg.norm(To) <= POSE_RTOL
            pts = prod.points()
            cur = ref.frames[-1]
            assert len(pts["xy"]) == len(cur.points)
            for i, q in enumerate(cur.points):
                assert np.array_equal(pts["xy"][i].view(np.uint32), q.xy.view(np.uint32)), (k, i)
                np.testing.assert_array_equal(pts["desc"][i], q.desc)
                np.testing.assert_allclose(pts["cam"][i], q.cam, rtol=1e-12, atol=0)
            rc = ref.feat_rc
            shared += len(rc) - len(np.unique(rc[:, 0].astype(np.int64) * 100000 + rc[:, 1]))
        assert fi.status == 1 and fi.n_tracked > 20 and fi.n_keypoints_left > 300, (fi.status, fi.n_tracked, fi.n_keypoints_left)
        assert shared > 0                                  # keypoints of several pyramid levels did land on the same pixel
        assert len(set(a["thresholds"])) > 1               # the four detectors' FAST thresholds moved apart
    finally:
        prod.destroy(); o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("descriptor", [1, 0])
def test_rgbd_orb_detector_reregistration_keeps_the_union_of_keypoints(descriptor, monkeypatch):
    """A lost track with the OrbDetector: the second and third initialize() of the frame append their detections to keypointsLeft()
    (base_framepoint_generator.cpp:422); cv::ORB::compute then regroups the no longer level-sorted vector level-major (extractor ORB), BRIEF
    leaves the order alone.  Product (host-driven loop) == checker loop over the oracle, counters, poses and every point."""
    from _oracle import Oracle
    monkeypatch.setenv("VSLAM_RGBD_HOST", "0")
    o = Oracle()
    scene, cfg, p = setup(o, "tum", descriptor=descriptor, seed=67)
    p.detector_type = 1
    o.create(cfg, 0, 1)
    g = hip.load()
    ref = PyLoop(o, cfg, p)
    prod = RgbdTracker(g, cfg, p)
    try:
        most, levels = 0, 0
        for k in [0, 1, 2, 3, 4, 5, 19, 20, 21]:
            L, D = o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)
            a = ref.process(L, D)
            fi, n_temp = prod.process(L, D)
            for name, field in (("status", "status"), ("n_keypoints", "n_keypoints_left"), ("n_tracked", "n_tracked"), ("n_lost", "n_lost"),
                                ("n_tracked_landmarks", "n_tracked_landmarks"), ("aligner_ran", "aligner_ran"), ("n_inliers", "n_inliers"),
                                ("n_after_prune", "n_after_prune"), ("n_recovered", "n_recovered"), ("n_active_landmarks", "n_active_landmarks"),
                                ("n_new", "n_new_stereo"), ("n_points", "n_points"), ("window_pixels", "window_pixels"),
                                ("track_attempts", "track_attempts"), ("fallback", "fallback"), ("track_broken", "track_broken")):
                assert a[name] == getattr(fi, field), (k, name, a[name], getattr(fi, field))
            assert a["thresholds"] == list(fi.thresholds)[:len(a["thresholds"])] and a["n_temporary"] == n_temp
            To, Tg = a["pose"], np.array(fi.camera_left_to_world).reshape(3, 4)
            assert np.linalg.norm(Tg - To) / np.linalg.norm(To) <= POSE_RTOL
            pts, cur = prod.points(), ref.frames[-1]
            assert len(pts["xy"]) == len(cur.points)
            for i, q in enumerate(cur.points):
                assert np.array_equal(pts["xy"][i].view(np.uint32), q.xy.view(np.uint32)), (k, i)
                np.testing.assert_array_equal(pts["desc"][i], q.desc)
                np.testing.assert_allclose(pts["cam"][i], q.cam, rtol=1e-12, atol=0)
            if fi.track_attempts > 1:
                most = max(most, fi.track_attempts)
                levels = max(levels, int(ref.acc_level.max()))
                if descriptor == 1:
                    assert np.all(np.diff(ref.acc_level) >= 0)          # ORB::compute left the union level-major
        assert most >= 2 and levels >= 1, (most, levels)
    finally:
        prod.destroy(); o.destroy()


@pytest.mark.gpu
def test_rgbd_device_loop_equals_host_loop_at_full_resolution(monkeypatch):
    """1241 x 376: ~1800 features and ~1300 framepoints per frame — more than one 1024-thread pass in every single-workgroup kernel of the
    device-resident loop (track bookkeeping, prune, recovery, compute, list closing).  Both product loops on the same ten frames: frame
    counters, thresholds, poses and the complete point lists identical."""
    from _oracle import Oracle
    o = Oracle()
    scene, cfg, p = setup(o, "tum", scale=1.0, descriptor=1, seed=71)
    g = hip.load()
    frames = [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(10)]
    monkeypatch.setenv("VSLAM_RGBD_HOST", "0")
    dev = RgbdTracker(g, cfg, p)
    monkeypatch.setenv("VSLAM_RGBD_HOST", "1")
    host = RgbdTracker(g, cfg, p)
    try:
        for k, (L, D) in enumerate(frames):
            fa, na = dev.process(L, D)
            fb, nb = host.process(L, D)
            for name, _ in fa._fields_:
                va, vb = getattr(fa, name), getattr(fb, name)
                if hasattr(va, "__len__"):
                    va, vb = list(va), list(vb)
                if name in ("camera_left_to_world", "previous_to_current", "total_error"):
                    np.testing.assert_allclose(np.array(va), np.array(vb), rtol=1e-9, atol=1e-12, err_msg="%d %s" % (k, name))
                else:
                    assert va == vb, (k, name, va, vb)
            assert na == nb
            pa, pb = dev.points(), host.points()
            np.testing.assert_array_equal(pa["xy"].view(np.uint32), pb["xy"].view(np.uint32))
            np.testing.assert_array_equal(pa["desc"], pb["desc"])
            np.testing.assert_array_equal(pa["meta"], pb["meta"])
            np.testing.assert_allclose(pa["cam"], pb["cam"], rtol=1e-12, atol=0)
        assert fa.status == 1 and fa.n_points > 1024 and fa.n_keypoints_left > 1500 and fa.n_tracked > 600, (fa.n_points, fa.n_keypoints_left, fa.n_tracked)
    finally:
        dev.destroy(); host.destroy(); o.destroy()


@pytest.mark.gpu
def test_rgbd_three_trackers_in_flight_together(monkeypatch):
    """Three sequences on one GPU: three tracker objects, every frame submitted for all of them before any is waited for
    (vslam_rgbd_submit_host / vslam_rgbd_wait) — each tracker's results are those of the same sequence run alone; a second submit without a
    wait, and a wait without a submit, are refused."""
    from _oracle import Oracle
    from vslam_pose_estimation_framework_amd.capi import VslamError, ERR_STATE
    monkeypatch.setenv("VSLAM_RGBD_HOST", "0")
    o = Oracle()
    g = hip.load()
    worlds = []
    for i, which in enumerate(("tum", "icl", "xtion")):
        scene, cfg, p = setup(o, which, seed=83 + i)
        worlds.append((cfg, p, [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(8)]))
    alone = []
    for cfg, p, frames in worlds:
        t = RgbdTracker(g, cfg, p)
        alone.append(_run(t, frames))
        t.destroy()
    trackers = [RgbdTracker(g, cfg, p) for cfg, p, _ in worlds]
    try:
        together = [[] for _ in trackers]
        for f in range(8):
            for i, t in enumerate(trackers):
                t.submit(*worlds[i][2][f])
            if f == 3:
                with pytest.raises(VslamError) as e:
                    trackers[0].submit(*worlds[0][2][f])
                assert e.value.code == ERR_STATE
                with pytest.raises(VslamError) as e:          # the frame in flight is rewriting the lists: not readable before wait()
                    trackers[0].points()
                assert e.value.code == ERR_STATE
            for i, t in enumerate(trackers):
                fi, nt = t.wait()
                together[i].append((fi.status, fi.n_keypoints_left, fi.n_tracked, fi.n_lost, fi.n_inliers, fi.aligner_iterations, fi.n_after_prune, fi.n_recovered,
                                    fi.n_active_landmarks, fi.n_new_stereo, fi.n_points, fi.track_attempts, fi.window_pixels, fi.tau_track, nt,
                                    list(fi.thresholds)[:4], tuple(fi.camera_left_to_world)))
        assert together == alone
        with pytest.raises(VslamError) as e:
            trackers[1].wait()
        assert e.value.code == ERR_STATE
        trackers[2].submit(*worlds[2][2][0])          # destroyed with a frame in flight: release() drains its queues first
    finally:
        for t in trackers:
            t.destroy()
        o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("which,descriptor", [("tum", 1), ("icl", 0)])
def test_rgbd_batch_of_sequences_equals_the_sequences_alone(which, descriptor, monkeypatch):
    """vslam_rgbd_create_batch: five sequences (five worlds, different camera speeds) in ONE context, one launch sequence per step for all of them.
    Sequence 2 jumps in the middle (lost track: two more registration attempts, breakTrack) and sequence 4 loses its depth image for a frame,
    while the others track on — every sequence's frame info and complete point lists equal those of the same sequence run alone."""
    from _oracle import Oracle
    monkeypatch.setenv("VSLAM_RGBD_HOST", "0")
    o = Oracle()
    g = hip.load()
    n, B = 10, 5
    worlds = []
    for i in range(B):
        scene, cfg, p = setup(o, which, descriptor=descriptor, seed=101 + 13 * i)
        scene.speed_m = scene.speed_m * (0.7 + 0.15 * i)
        ks = list(range(n)) if i != 2 else [0, 1, 2, 3, 4, 20, 21, 22, 23, 24]
        frames = [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in ks]
        if i == 4:
            frames[6] = (frames[6][0], np.zeros_like(frames[6][1]))
        worlds.append(frames)
    alone, alone_pts = [], []
    for frames in worlds:
        t = RgbdTracker(g, cfg, p)
        rec, pts = [], []
        for L, D in frames:
            fi, nt = t.process(L, D)
            rec.append((fi, nt)); pts.append(t.points())
        alone.append(rec); alone_pts.append(pts)
        t.destroy()
    batch = RgbdBatch(g, cfg, p, B)
    try:
        attempts = set()
        for f in range(n):
            L = np.stack([worlds[i][f][0] for i in range(B)]); D = np.stack([worlds[i][f][1] for i in range(B)])
            res = batch.process(L, D)
            for i, (fi, nt) in enumerate(res):
                fa, na = alone[i][f]
                for name, _ in fa._fields_:
                    va, vb = getattr(fa, name), getattr(fi, name)
                    if hasattr(va, "__len__"):
                        va, vb = list(va), list(vb)
                    assert va == vb, (f, i, name, va, vb)
                assert na == nt
                pa, pb = alone_pts[i][f], batch.points(i)
                for key in ("xy", "cam", "meta", "desc"):
                    np.testing.assert_array_equal(pa[key], pb[key], err_msg="frame %d sequence %d %s" % (f, i, key))
                attempts.add((i, fi.track_attempts))
        assert (2, 3) in attempts                 # the jumping sequence needed all three attempts while the others went on undisturbed
        assert sum(fi.status == 1 for fi, _ in res) >= 3
    finally:
        batch.destroy(); o.destroy()


@pytest.mark.gpu
def test_rgbd_batch_on_device_images_and_strides(monkeypatch):
    """The batch fed with images that already live in HBM (vslam_rgbd_submit_batch_device: no copy, the kernels read the caller's tensors) and
    with padded host images (stream and row strides that do not form one dense block): both equal the dense host path, frame by frame."""
    import torch
    from _oracle import Oracle
    monkeypatch.setenv("VSLAM_RGBD_HOST", "0")
    o = Oracle()
    g = hip.load()
    B, n = 3, 8
    seqs = []
    for i in range(B):
        scene, cfg, p = setup(o, "tum", descriptor=1, seed=301 + i)
        seqs.append([(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(n)])
    rows, cols = int(cfg.rows), int(cfg.cols)
    dev = torch.device("cuda", 0)
    a, b, c = RgbdBatch(g, cfg, p, B), RgbdBatch(g, cfg, p, B), RgbdBatch(g, cfg, p, B)
    try:
        for f in range(n):
            L = np.stack([seqs[i][f][0] for i in range(B)]); D = np.stack([seqs[i][f][1] for i in range(B)])
            ra = a.process(L, D)
            Ld = torch.from_numpy(L).to(dev); Dd = torch.from_numpy(D.view(np.int16)).to(dev)
            torch.cuda.synchronize()
            b.submit_device(Ld.data_ptr(), cols, rows * cols, Dd.data_ptr(), cols, rows * cols)
            rb_ = b.wait()
            Lp = np.full((B, rows + 3, cols + 20), 9, np.uint8); Lp[:, :rows, :cols] = L        # padded rows AND three spare rows between the sequences
            Dp = np.full((B, rows + 1, cols + 6), 77, np.uint16); Dp[:, :rows, :cols] = D
            rc_ = c.process(Lp, Dp)
            for i in range(B):
                for other in (rb_, rc_):
                    fa, fb = ra[i][0], other[i][0]
                    for name, _ in fa._fields_:
                        va, vb = getattr(fa, name), getattr(fb, name)
                        if hasattr(va, "__len__"):
                            va, vb = list(va), list(vb)
                        assert va == vb, (f, i, name)
                    assert ra[i][1] == other[i][1]
                pa, pb, pc = a.points(i), b.points(i), c.points(i)
                for key in ("xy", "cam", "meta", "desc"):
                    np.testing.assert_array_equal(pa[key], pb[key]); np.testing.assert_array_equal(pa[key], pc[key])
        assert all(fi.status == 1 and fi.n_tracked > 50 for fi, _ in ra)
    finally:
        a.destroy(); b.destroy(); c.destroy(); o.destroy()
