"""One whole KITTI-00-length synthetic sequence (4541 frames, full resolution, configuration_kitti.yaml values) through the HIP path
and through the CPU oracle: every frame's counters, thresholds, tracker state and pose compared.  Prints one JSON line.
Usage: python tests/validation/full_sequence_parity.py [frames]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from vslam_pose_estimation_framework_amd import hip
from _oracle import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
INT_FIELDS = ["status", "n_keypoints_left", "n_keypoints_right", "n_detected_left", "n_detected_right", "n_tracked", "n_lost",
              "n_tracked_landmarks", "aligner_ran", "aligner_iterations", "n_inliers", "n_outliers", "track_attempts", "n_after_prune",
              "n_recovered", "n_active_landmarks", "n_new_stereo", "n_points", "track_broken", "fallback", "window_pixels", "error_flags"]
o = Oracle()
sc = o.scene_kitti(scale=1.0, seed=7)
cfg = o.config_for_scene(sc, "kitti")
cfg.max_keypoints, cfg.max_points, cfg.max_history_frames = 8192, 4096, 512
o.create(cfg, 0, 1)
g = hip.load(); g.create(cfg, 0, 1)
mism, worst, t_o, t_g, flags = 0, 0.0, 0.0, 0.0, 0
for k in range(N):
    L, R = o.render(sc, k)
    t0 = time.perf_counter(); o.process_host(L[None], R[None]); t_o += time.perf_counter() - t0
    t0 = time.perf_counter(); g.process_host(L[None], R[None]); fg = g.frame_info(0); t_g += time.perf_counter() - t0
    fo = o.frame_info(0)
    for name in INT_FIELDS:
        if getattr(fo, name) != getattr(fg, name):
            mism += 1
    if fo.tau_track != fg.tau_track or list(fo.thresholds) != list(fg.thresholds):
        mism += 1
    po, pg = np.array(fo.camera_left_to_world), np.array(fg.camera_left_to_world)
    worst = max(worst, float(np.linalg.norm(po - pg) / np.linalg.norm(po)))
    flags |= fg.error_flags
    if k % 500 == 0:
        print("frame %d mismatches %d worst pose %.3e" % (k, mism, worst), file=sys.stderr, flush=True)
po = o.points(0); pg = g.points(0)
same_points = all(np.array_equal(po[key], pg[key]) for key in ("kp", "meta")) and np.allclose(po["cam"], pg["cam"], rtol=0, atol=1e-9)
print(json.dumps({"frames": N, "resolution": [cfg.cols, cfg.rows], "integer_field_mismatches": mism, "max_pose_rel_frobenius": worst,
                  "last_frame_points_identical": bool(same_points), "error_flags": int(flags), "final_status": int(fg.status),
                  "oracle_seconds": round(t_o, 1), "hip_seconds_incl_host_copy_and_readback": round(t_g, 1)}))
o.destroy(); g.destroy()
