# two identical fused contexts + one staged on the same frames: report the first field that differs
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from _oracle import Oracle
from vslam_pose_estimation_framework_amd import hip
from vslam_pose_estimation_framework_amd.host_tracker import PoseTracker3D
o = Oracle(); reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
for rep in range(reps):
    sc = o.scene_kitti(scale=0.5, seed=21 + rep); cfg = o.config_for_scene(sc)
    a = hip.load(); a.create(cfg, 0, 1); b = hip.load(); b.create(cfg, 0, 1); c = hip.load(); c.create(cfg, 0, 1)
    tr = PoseTracker3D(c)
    for k in range(12):
        L, R = o.render(sc, k)
        a.process_host(L, R); b.process_host(L, R); fc = tr.compute(L, R)
        fa, fb = a.frame_info(0), b.frame_info(0)
        for nm, _ in fa._fields_:
            va, vb, vc = getattr(fa, nm), getattr(fb, nm), getattr(fc, nm)
            if hasattr(va, '__len__'): va, vb, vc = list(va), list(vb), list(vc)
            if va != vb: print("rep", rep, "frame", k, "FUSED-vs-FUSED differ:", nm, va, vb); bad += 1
            if va != vc and nm not in ("fallback", "track_broken", "aligner_iterations", "aligner_converged", "status_at_start"):
                print("rep", rep, "frame", k, "FUSED-vs-STAGED differ:", nm, va, vc); bad += 1
        pa, pb, pc = a.points(0), b.points(0), c.points(0)
        for key in pa:
            if not np.array_equal(pa[key], pb[key]): print("rep", rep, "frame", k, "points", key, "differ fused/fused"); bad += 1
            if not np.array_equal(pa[key], pc[key]): print("rep", rep, "frame", k, "points", key, "differ fused/staged",
                                                         np.argwhere(pa[key] != pc[key])[:3].tolist() if pa[key].shape == pc[key].shape else (pa[key].shape, pc[key].shape)); bad += 1
        if bad > 6: break
    a.destroy(); b.destroy(); c.destroy()
    if bad > 6: break
print("done, mismatches:", bad)
