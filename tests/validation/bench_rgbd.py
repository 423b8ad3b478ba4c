#!/usr/bin/env python3
"""Per-frame time of the RGB-D mode (vslam_rgbd_process_host: host-driven loop over the device entry points) on rendered
image + depth frames of the half-size street scene, and at 640 x 480."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle import Oracle  # noqa: E402  (renderer only)
from test_rgbd_mode import setup  # noqa: E402
from vslam_pose_estimation_framework_amd import hip  # noqa: E402
from vslam_pose_estimation_framework_amd.capi import RgbdTracker  # noqa: E402

o = Oracle()
for scale in (0.5, 0.52):
    scene, cfg, p = setup(o, scale=scale)
    g = hip.load()
    t = RgbdTracker(g, cfg, p)
    frames = [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(24)]
    for L, D in frames[:4]:
        t.process(L, D)
    t0 = time.perf_counter()
    for L, D in frames[4:]:
        fi, nt = t.process(L, D)
    dt = (time.perf_counter() - t0) / 20
    print("%d x %d: %.2f ms per frame (%.0f frames/s), tracked %d, points %d" % (cfg.cols, cfg.rows, dt * 1e3, 1 / dt, fi.n_tracked, fi.n_points))
    t.destroy()
