"""Creates and destroys contexts in a loop and watches free device memory and whether torch can still initialise afterwards."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from vslam_pose_estimation_framework_amd import hip
from _oracle import Oracle
hiprt = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    rc = hiprt.hipMemGetInfo(C.byref(f), C.byref(t))
    return rc, f.value / 2**20
o = Oracle(); sc = o.scene_kitti(scale=0.5); cfg = o.config_for_scene(sc)
L, R = o.render(sc, 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(N):
    g = hip.load(); g.create(cfg, 0, 1 + (i % 3))
    Ls = np.stack([L] * g.n_streams); Rs = np.stack([R] * g.n_streams)
    g.process_host(Ls, Rs); g.frame_info(0)
    if i % 3 == 0:
        g.fast_detect(L, (0, 0, cfg.cols, cfg.rows), 20)
    g.destroy()
    if i % 10 == 0:
        print(i, free_mb(), flush=True)
print("final", free_mb())
import torch
try:
    x = torch.zeros(4, device="cuda"); print("torch ok", x.sum().item())
except Exception as e:
    print("torch FAILED:", str(e)[:200])
print("fds", len(os.listdir("/proc/self/fd")))
