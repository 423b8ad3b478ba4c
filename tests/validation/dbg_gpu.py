# GPU-side debug driver: runs oracle and HIP side by side, prints per-frame summaries and first mismatches
import sys, os, time, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from _oracle import Oracle
from vslam_pose_estimation_framework_amd import hip
import test_hip_parity as T
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
o = Oracle(); sc = o.scene_kitti(scale=scale); cfg = o.config_for_scene(sc)
o.create(cfg, 0, 1)
g = hip.load(); g.create(cfg, 0, 1)
for k in range(nf):
    L, R = o.render(sc, k)
    o.process_host(L, R)
    t0 = time.time(); g.process_host(L, R); g.synchronize(); t1 = time.time()
    fo, fg = o.frame_info(0), g.frame_info(0)
    d = [(n, getattr(fo, n), getattr(fg, n)) for n in T.INT_FIELDS if getattr(fo, n) != getattr(fg, n)]
    print("frame", k, "hip %.1f ms" % ((t1 - t0) * 1e3), "kp", fg.n_keypoints_left, fg.n_keypoints_right, "trk", fg.n_tracked, "inl", fg.n_inliers,
          "pts", fg.n_points, "it", fo.aligner_iterations, fg.aligner_iterations, "DIFF" if d else "same-ints", d[:8])
    try:
        T.compare_frame(o, g, 0, k)
        print("   frame parity OK")
    except AssertionError as e:
        print("   MISMATCH:", str(e)[:1500])
        break
