#!/usr/bin/env python3
"""RGB-D mode, milliseconds per frame of the two product loops behind vslam_rgbd_* on the same rendered sequence: the device-resident loop
(csrc/rgbd_device.h + kernels_rgbd.h) and the host-driven loop over the stand-alone entry points (csrc/rgbd_tracker.h, VSLAM_RGBD_HOST=1).
Both must report identical frame counters; the wall clock is around vslam_rgbd_process_host (host images, copies included).
usage: rgbd_bench.py [icl|tum|xtion] [frames] [scale] [both|device|host]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle  # renderer only
from test_rgbd_mode import setup
from vslam_pose_estimation_framework_amd import hip
from vslam_pose_estimation_framework_amd.capi import RgbdTracker

which = sys.argv[1] if len(sys.argv) > 1 else "tum"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
o = Oracle()
scene, cfg, p = setup(o, which, scale=scale)
g = hip.load()
frames = [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(n)]
out = {"config": which, "rows": int(cfg.rows), "cols": int(cfg.cols), "frames": n}
infos = {}
impls = ("device", "host") if len(sys.argv) <= 4 or sys.argv[4] == "both" else (sys.argv[4],)
for impl in impls:
    os.environ["VSLAM_RGBD_HOST"] = "1" if impl == "host" else "0"
    t = RgbdTracker(g, cfg, p)
    rec, times = [], []
    for L, D in frames:
        t0 = time.perf_counter()
        fi, n_temp = t.process(L, D)
        times.append(time.perf_counter() - t0)
        rec.append((fi.status, fi.n_keypoints_left, fi.n_tracked, fi.n_inliers, fi.n_after_prune, fi.n_recovered, fi.n_active_landmarks, fi.n_new_stereo,
                    fi.n_points, fi.track_attempts, n_temp, tuple(np.array(fi.camera_left_to_world).round(9))))
    t.destroy()
    infos[impl] = rec
    warm = times[8:]
    out[impl] = {"ms_per_frame_mean": 1e3 * float(np.mean(warm)), "ms_per_frame_median": 1e3 * float(np.median(warm)), "ms_per_frame_min": 1e3 * float(np.min(warm))}
    out[impl]["last"] = dict(n_keypoints=rec[-1][1], n_tracked=rec[-1][2], n_active_landmarks=rec[-1][6], n_points=rec[-1][8])
if len(impls) < 2:
    print(json.dumps(out)); sys.exit(0)
same = all(a[:11] == b[:11] and np.allclose(a[11], b[11], rtol=0, atol=1e-6) for a, b in zip(infos["device"], infos["host"]))
out["identical_counters"] = bool(same)
out["speedup_median"] = out["host"]["ms_per_frame_median"] / out["device"]["ms_per_frame_median"]
print(json.dumps(out))
