#!/usr/bin/env python3
"""RGB-D mode with K sequences on one GPU: K tracker objects (vslam_rgbd_*: one sequence each, the device-resident loop on HIP streams of its own),
every frame submitted for all of them before any is waited for (vslam_rgbd_submit_host / vslam_rgbd_wait), so that their launch sequences overlap
on the GPU.  Prints aggregate frames/s for K = 1, 2, 4, ... against one tracker stepping alone; the sequences are K different synthetic worlds.
usage: rgbd_multi.py [icl|tum|xtion] [frames] [max K]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle  # renderer only
from test_rgbd_mode import setup
from vslam_pose_estimation_framework_amd import hip
from vslam_pose_estimation_framework_amd.capi import RgbdTracker

which = sys.argv[1] if len(sys.argv) > 1 else "tum"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 16
os.environ["VSLAM_RGBD_HOST"] = "0"
o = Oracle()
g = hip.load()
worlds = []
for i in range(kmax):
    scene, cfg, p = setup(o, which, seed=23 + 7 * i)
    worlds.append((cfg, p, [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(n)]))
out = {"config": which, "rows": int(worlds[0][0].rows), "cols": int(worlds[0][0].cols), "frames_per_sequence": n, "runs": []}
K = 1
while K <= kmax:
    trackers = [RgbdTracker(g, worlds[i][0], worlds[i][1]) for i in range(K)]
    t_warm = 8
    last = None
    for f in range(n):
        if f == t_warm:
            t0 = time.perf_counter()
        for i, t in enumerate(trackers):
            t.submit(*worlds[i][2][f])
        last = [t.wait() for t in trackers]
    dt = time.perf_counter() - t0
    for t in trackers:
        t.destroy()
    out["runs"].append({"trackers": K, "frames_per_s": K * (n - t_warm) / dt, "ms_per_frame_aggregate": 1e3 * dt / (K * (n - t_warm)),
                        "ms_per_step": 1e3 * dt / (n - t_warm), "tracking": int(sum(fi.status == 1 for fi, _ in last))})
    # the same K sequences, one host thread per tracker (ctypes releases the GIL inside the library): the submission cost of a frame
    # (~20 launches with ~2.6 KB of arguments each, two pageable copies) is then paid in parallel as well
    import threading
    trackers = [RgbdTracker(g, worlds[i][0], worlds[i][1]) for i in range(K)]
    for i, t in enumerate(trackers):
        for f in range(t_warm):
            t.process(*worlds[i][2][f])
    def work(i):
        t = trackers[i]
        for f in range(t_warm, n):
            t.process(*worlds[i][2][f])
    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    for t in trackers:
        t.destroy()
    out["runs"][-1]["threads_frames_per_s"] = K * (n - t_warm) / dt
    K *= 2
out["speedup_at_max"] = out["runs"][-1]["frames_per_s"] / out["runs"][0]["frames_per_s"]
print(json.dumps(out))
