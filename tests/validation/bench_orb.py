"""Kernel times of vslam_orb_detect (the reference's OrbDetector parameters: 5000 features, 1.2, 8 levels, edge 31, patch 31)
on a KITTI-sized synthetic frame; run under rocprofv3 --kernel-trace.  Checks GPU == oracle bit for bit."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from vslam_pose_estimation_framework_amd import hip
from _oracle import Oracle
api = hip.load(); api.create(api.default_config("kitti"), 0, 1)
orc = Oracle()
scene = orc.scene_kitti()
left, right = orc.render(scene, 40)
for thr in (20, 12):
    for _ in range(3):
        g = api.orb_detect(left, 5000, 1.2, 8, 31, 31, thr)
    o = orc.orb_detect(left, 5000, 1.2, 8, 31, 31, thr)
    same = g.shape == o.shape and bool(np.array_equal(g.view(np.uint32), o.view(np.uint32)))
    print("threshold %d: %d keypoints, per level %s, gpu == oracle: %s" % (thr, len(g), np.bincount(g[:, 5].astype(int), minlength=8).tolist(), same), flush=True)
    assert same
