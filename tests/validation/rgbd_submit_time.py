import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle
from test_rgbd_mode import setup
from vslam_pose_estimation_framework_amd import hip
from vslam_pose_estimation_framework_amd.capi import RgbdTracker
os.environ["VSLAM_RGBD_HOST"] = "0"
o = Oracle(); g = hip.load()
scene, cfg, p = setup(o, "tum", seed=23)
frames = [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(40)]
for graph in ("1", "0"):
    os.environ["VSLAM_RGBD_GRAPH"] = graph
    t = RgbdTracker(g, cfg, p)
    ts, tw = [], []
    for L, D in frames:
        a = time.perf_counter(); t.submit(L, D); b = time.perf_counter(); t.wait(); c = time.perf_counter()
        ts.append(b - a); tw.append(c - b)
    t.destroy()
    print("graph", graph, "submit us median %.1f  wait us median %.1f" % (1e6 * np.median(ts[8:]), 1e6 * np.median(tw[8:])))
