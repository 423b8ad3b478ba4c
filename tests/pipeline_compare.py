"""Frame-by-frame comparison of the HIP path with the CPU oracle on the same seeded synthetic sequences (shared by the
GPU parity tests).  Integer / byte / index results must be bit-exact; poses within 1e-4 relative Frobenius (north_star)."""
import numpy as np

from vslam_pose_estimation_framework_amd import hip

POSE_RTOL = 1e-4  # north_star: pose within 1e-4 relative Frobenius

INT_FIELDS = ["frame_index", "status", "status_at_start", "n_keypoints_left", "n_keypoints_right", "n_detected_left",
              "n_detected_right", "track_attempts", "n_tracked", "n_lost", "n_tracked_landmarks", "aligner_ran", "n_inliers",
              "n_outliers", "n_after_prune", "n_recovered", "n_active_landmarks", "n_new_stereo", "n_points",
              "track_broken", "fallback", "window_pixels", "error_flags"]


def compare_frame(o, g, s, k, tag="", sg=None, identical=False):
    """Stream s of the checker `o` against stream sg (default: s) of `g` after the same frame.  identical: `o` is another HIP
    context that ran the same images (one stream alone, say): every float must then be equal bit for bit as well."""
    sg = s if sg is None else sg
    s_o = s
    fo, fg = o.frame_info(s_o), g.frame_info(sg)
    for name in INT_FIELDS:
        assert getattr(fo, name) == getattr(fg, name), "%s frame %d stream %d: %s oracle=%s hip=%s" % (
            tag, k, s, name, getattr(fo, name), getattr(fg, name))
    assert list(fo.thresholds) == list(fg.thresholds)
    assert fo.tau_track == fg.tau_track and fo.tau_triangulation == fg.tau_triangulation
    for side in (0, 1):
        xo, sco, do = o.keypoints(s_o, side)
        xg, scg, dg = g.keypoints(sg, side)
        # oracle order is detector-region-major, the device order is image row-major: same for 1x1 grids
        io = np.lexsort((xo[:, 0], xo[:, 1]))
        ig = np.lexsort((xg[:, 0], xg[:, 1]))
        np.testing.assert_array_equal(xo[io], xg[ig])
        np.testing.assert_array_equal(sco[io], scg[ig])
        np.testing.assert_array_equal(do[io], dg[ig])
    po, pg = o.points(s_o), g.points(sg)
    np.testing.assert_array_equal(po["kp"], pg["kp"])
    np.testing.assert_array_equal(po["meta"], pg["meta"])
    if identical:
        np.testing.assert_array_equal(pg["cam"], po["cam"])
        np.testing.assert_array_equal(pg["lm"], po["lm"])
        assert list(fo.camera_left_to_world) == list(fg.camera_left_to_world) and list(fo.previous_to_current) == list(fg.previous_to_current)
    np.testing.assert_allclose(pg["cam"], po["cam"], rtol=1e-13, atol=0)
    np.testing.assert_allclose(pg["lm"], po["lm"], rtol=1e-6, atol=1e-6)
    To = np.array(fo.camera_left_to_world).reshape(3, 4)
    Tg = np.array(fg.camera_left_to_world).reshape(3, 4)
    assert np.linalg.norm(Tg - To) / np.linalg.norm(To) <= POSE_RTOL
    Po = np.array(fo.previous_to_current).reshape(3, 4)
    Pg = np.array(fg.previous_to_current).reshape(3, 4)
    assert np.linalg.norm(Pg - Po) / np.linalg.norm(Po) <= POSE_RTOL
    if fo.aligner_ran:
        ao, ag = o.aligner_result(s_o), g.aligner_result(sg)
        np.testing.assert_array_equal(ao["inlier"], ag["inlier"])
        np.testing.assert_allclose(ag["chi"], ao["chi"], rtol=1e-6, atol=1e-6)
        if identical:
            np.testing.assert_array_equal(ag["chi"], ao["chi"])
            np.testing.assert_array_equal(ag["H"], ao["H"])
        assert fo.aligner_iterations == fg.aligner_iterations
    np.testing.assert_array_equal(o.aligner_weights_of(s_o), g.aligner_weights_of(sg))   # persistent _weights_translation


def create_hip(cfg, n_streams, split=None):
    """HIP context; split: value of VSLAM_SPLIT while the context is created (None = the library's own choice)."""
    import os
    old = os.environ.get("VSLAM_SPLIT")
    if split is not None:
        os.environ["VSLAM_SPLIT"] = str(split)
    try:
        g = hip.load()
        g.create(cfg, 0, n_streams)
    finally:
        if split is not None:
            if old is None:
                del os.environ["VSLAM_SPLIT"]
            else:
                os.environ["VSLAM_SPLIT"] = old
    return g


def run_sequence(oracle_cls, scene_kw, n_frames, n_streams=1, which="kitti", cfg_edit=None, seeds=None, scene="kitti", after=None):
    """scene: "kitti" (street canyon, planar motion) or "euroc" (752x480 hall, 6-DoF motion); which: the default
    configuration the run starts from; after(o, g): called with both contexts still alive after the last frame."""
    o = oracle_cls()
    scenes = []
    for s in range(n_streams):
        make = o.scene_euroc if scene == "euroc" else o.scene_kitti
        sc = make(scale=scene_kw.get("scale", 0.5), seed=(seeds[s] if seeds else 7 + s))
        for k_, v_ in scene_kw.items():
            if k_ != "scale":
                setattr(sc, k_, v_)
        scenes.append(sc)
    cfg = o.config_for_scene(scenes[0], which)
    if cfg_edit:
        cfg_edit(cfg)
    o.create(cfg, 0, n_streams)
    # every launch sequence of the frame: the library's choice for this stream count (up to 64 streams sequence 4: phase launches around the wide
    # recovery kernel with the landmark kernel on a second queue beside the last phase; one fused launch above) and, forced through VSLAM_SPLIT:
    # the fused launch (0), the two launches around the wide recovery kernel (2), and registration + wide recovery + the frame's tail as the small
    # co-schedulable kernel (3; k_tail: 256 threads, stereo sweep band by band, bin competition on 16-bit tables, landmark cache one measurement deep)
    g = create_hip(cfg, n_streams)
    g2 = create_hip(cfg, n_streams, split=0)
    g3 = create_hip(cfg, n_streams, split=3)
    g4 = create_hip(cfg, n_streams, split=2)
    try:
        for k in range(n_frames):
            imgs = [o.render(sc, k) for sc in scenes]
            L = np.stack([im[0] for im in imgs])
            R = np.stack([im[1] for im in imgs])
            o.process_host(L, R)
            for h in (g, g2, g3, g4):
                h.process_host(L, R)
                for s in range(n_streams):
                    compare_frame(o, h, s, k)
        if after:
            after(o, g)
    finally:
        g.destroy()
        g2.destroy()
        g3.destroy()
        g4.destroy()
        o.destroy()


