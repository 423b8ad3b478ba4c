"""Phase durations of k_frame under the real overlap with the image pipeline (no host synchronisation between steps; the
in-kernel clocks accumulate, read once at the end) next to the synchronised figures of tools/dbg_tail.py.  Needs the
library built with EXTRA=-DVS_PROFILE_PHASES.  Usage: python tools/dbg_overlap.py [B] [K]"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from vslam_pose_estimation_framework_amd import hip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
api = hip.load(); sy = synth.Synth(); scene = sy.scene_kitti(7); cfg = synth.config_for_scene(api, scene)
cfg.max_history_frames = K + 2
stride = 1280; img = cfg.rows * stride
dev = torch.device("cuda", 0)
Lb = torch.empty((K, B, cfg.rows, stride), dtype=torch.uint8, device=dev); Rb = torch.empty_like(Lb)
for s in range(B):
    sy.render_device(scene, 28 * s, K, Lb[0, s].data_ptr(), Rb[0, s].data_ptr(), stride, B * img, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
api.create(cfg, 0, B)
names = ["track", "align", "recover", "landmark", "stereo"]
def ticks():
    buf = (C.c_ulonglong * (B * 17))()
    api.lib.vslam_debug_stream_ticks(api.ctx, buf)
    return np.frombuffer(buf, dtype=np.uint64).reshape(B, 17).astype(np.float64)
for mode in ("synchronised", "overlapped"):
    api.reset()
    for k in range(4):
        api.process_device(Lb[k].data_ptr(), Rb[k].data_ptr(), stride, img)
    api.synchronize()
    t0 = ticks()
    for k in range(4, K):
        api.process_device(Lb[k].data_ptr(), Rb[k].data_ptr(), stride, img)
        if mode == "synchronised":
            api.synchronize()
    api.synchronize()
    d = (ticks() - t0) * 1e-2 / (K - 4)
    print("%-12s" % mode, " ".join("%s %.1f" % (n, d[:, i].mean()) for i, n in enumerate(names)), "| total %.1f" % d[:, 13].mean(),
          "| prune %.1f" % d[:, 11].mean(), "| aligner compute %.1f gather %.1f solve %.1f update %.1f" % tuple(d[:, 5 + k].mean() for k in (5, 9, 10, 11)))
