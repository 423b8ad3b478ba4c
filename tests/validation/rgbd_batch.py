#!/usr/bin/env python3
"""RGB-D mode, B sequences in ONE context (vslam_rgbd_create_batch): one launch sequence per step for all of them.  B different synthetic worlds;
prints frames/s for B = 1, 2, 4, ... with host images (vslam_rgbd_process_batch_host: every step copies B images + B depth images over PCIe
from pageable memory) and with the images already in HBM (vslam_rgbd_submit_batch_device + vslam_rgbd_wait: the tracker alone).
usage: rgbd_batch.py [icl|tum|xtion] [frames] [max B] [scale]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle  # renderer only
from test_rgbd_mode import setup
from vslam_pose_estimation_framework_amd import hip
from vslam_pose_estimation_framework_amd.capi import RgbdBatch

which = sys.argv[1] if len(sys.argv) > 1 else "tum"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bmax = int(sys.argv[3]) if len(sys.argv) > 3 else 64
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
os.environ["VSLAM_RGBD_HOST"] = "0"
o = Oracle(); g = hip.load()
# a handful of rendered worlds, reused round-robin with a shifted start so that the sequences of a batch differ
W = 8
worlds = []
for i in range(W):
    scene, cfg, p = setup(o, which, scale=scale, seed=23 + 7 * i)
    worlds.append([(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(n)])
cfg.max_points = 4096; cfg.max_keypoints = 8192; cfg.max_history_frames = 64       # memory of B sequences: ring of 64 frames each
out = {"config": which, "rows": int(cfg.rows), "cols": int(cfg.cols), "frames_per_sequence": n, "runs": []}
B = 1
while B <= bmax:
    L = [np.stack([worlds[i % W][f][0] for i in range(B)]) for f in range(n)]
    D = [np.stack([worlds[i % W][f][1] for i in range(B)]) for f in range(n)]
    t = RgbdBatch(g, cfg, p, B)
    warm = 6
    for f in range(n):
        if f == warm:
            t0 = time.perf_counter()
        res = t.process(L[f], D[f])
    dt = time.perf_counter() - t0
    t.destroy()
    row = {"sequences": B, "host_images": {"frames_per_s": B * (n - warm) / dt, "ms_per_step": 1e3 * dt / (n - warm), "us_per_frame": 1e6 * dt / (B * (n - warm))},
           "tracking": int(sum(fi.status == 1 for fi, _ in res)), "mean_tracked": float(np.mean([fi.n_tracked for fi, _ in res]))}
    # the same sequences with the images resident in HBM
    import torch
    dev = torch.device("cuda", 0)
    Ld = [torch.from_numpy(a).to(dev) for a in L]; Dd = [torch.from_numpy(a.view(np.int16)).to(dev) for a in D]
    torch.cuda.synchronize()
    t = RgbdBatch(g, cfg, p, B)
    rows, cols = int(cfg.rows), int(cfg.cols)
    for f in range(n):
        if f == warm:
            t0 = time.perf_counter()
        t.submit_device(Ld[f].data_ptr(), cols, rows * cols, Dd[f].data_ptr(), cols, rows * cols)
        res2 = t.wait(infos=(f == n - 1))
    dt = time.perf_counter() - t0
    t.destroy()
    same = all(a[0].n_points == b[0].n_points and a[0].n_tracked == b[0].n_tracked and tuple(a[0].camera_left_to_world) == tuple(b[0].camera_left_to_world) for a, b in zip(res, res2))
    row["device_images"] = {"frames_per_s": B * (n - warm) / dt, "ms_per_step": 1e3 * dt / (n - warm), "us_per_frame": 1e6 * dt / (B * (n - warm)), "same_results_as_host_images": bool(same)}
    del Ld, Dd
    out["runs"].append(row)
    B *= 2
out["speedup_at_max"] = out["runs"][-1]["device_images"]["frames_per_s"] / out["runs"][0]["device_images"]["frames_per_s"]
print(json.dumps(out))
