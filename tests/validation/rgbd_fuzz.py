#!/usr/bin/env python3
"""Device-resident RGB-D loop against the host-driven loop on many worlds: seeds x configurations x extractors, camera speeds drawn at random,
occasional jumps in the sequence (re-registrations, broken tracks) and depth drop-outs.  Every frame: all frame-info fields and the complete
point lists must be identical.  usage: rgbd_fuzz.py [runs] [frames]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle  # renderer only
from test_rgbd_mode import setup
from vslam_pose_estimation_framework_amd import hip
from vslam_pose_estimation_framework_amd.capi import RgbdTracker

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(2026)
o = Oracle(); g = hip.load()
bad, total_frames, attempts, broken = [], 0, {0: 0, 1: 0, 2: 0, 3: 0}, 0
t0 = time.time()
for run in range(runs):
    which = ("tum", "icl", "xtion")[run % 3]
    descriptor = int(rng.integers(0, 2))
    seed = int(rng.integers(1, 100000))
    scene, cfg, p = setup(o, which, descriptor=descriptor, seed=seed)
    scene.speed_m = float(scene.speed_m * rng.uniform(0.5, 1.8)); scene.sway_m = float(scene.sway_m * rng.uniform(0.5, 1.5))
    if rng.random() < 0.3:
        cfg.minimum_number_of_landmarks_to_track = int(rng.integers(15, 45))
    ks, k = [], 0
    for f in range(n):
        ks.append(k)
        k += 1 if rng.random() > 0.08 else int(rng.integers(4, 20))
    drop = set(int(v) for v in rng.choice(n, 2, replace=False)) if rng.random() < 0.4 else set()
    os.environ["VSLAM_RGBD_HOST"] = "0"; dev = RgbdTracker(g, cfg, p)
    os.environ["VSLAM_RGBD_HOST"] = "1"; host = RgbdTracker(g, cfg, p)
    try:
        for f, kf in enumerate(ks):
            L = o.render(scene, kf)[0]
            D = o.render_depth(scene, kf, 2e-3) if f not in drop else np.zeros((cfg.rows, cfg.cols), np.uint16)
            fa, na = dev.process(L, D); fb, nb = host.process(L, D)
            total_frames += 1
            attempts[min(int(fa.track_attempts), 3)] += 1
            broken += fa.track_broken
            diff = [name for name, _ in fa._fields_ if (list(getattr(fa, name)) if hasattr(getattr(fa, name), "__len__") else getattr(fa, name)) !=
                    (list(getattr(fb, name)) if hasattr(getattr(fb, name), "__len__") else getattr(fb, name))
                    and name not in ("camera_left_to_world", "previous_to_current", "total_error")]
            Ta, Tb = np.array(fa.camera_left_to_world), np.array(fb.camera_left_to_world)
            if np.linalg.norm(Ta - Tb) > 1e-9 * max(1.0, np.linalg.norm(Tb)) or na != nb:
                diff.append("pose/temps")
            pa, pb = dev.points(), host.points()
            if not (np.array_equal(pa["xy"].view(np.uint32), pb["xy"].view(np.uint32)) and np.array_equal(pa["desc"], pb["desc"]) and
                    np.array_equal(pa["meta"], pb["meta"]) and np.allclose(pa["cam"], pb["cam"], rtol=1e-12, atol=0)):
                diff.append("points")
            if diff:
                bad.append({"run": run, "config": which, "descriptor": descriptor, "seed": seed, "frame": f, "fields": diff})
                break
    finally:
        dev.destroy(); host.destroy()
print(json.dumps({"runs": runs, "frames": total_frames, "mismatching_runs": bad, "track_attempts_histogram": attempts, "broken_tracks": int(broken),
                  "seconds": round(time.time() - t0, 1)}))
