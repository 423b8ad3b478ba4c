import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from vslam_pose_estimation_framework_amd import hip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
api = hip.load(); sy = synth.Synth(); scene = sy.scene_kitti(7); cfg = synth.config_for_scene(api, scene)
cfg.max_keypoints = 8192; cfg.max_points = 4096; cfg.max_history_frames = K + 2
stride = 1280; img = cfg.rows * stride
dev = torch.device("cuda", 0)
Lb = torch.empty((K, B, cfg.rows, stride), dtype=torch.uint8, device=dev); Rb = torch.empty_like(Lb)
for s in range(B):
    sy.render_device(scene, 40 * s, K, Lb[0, s].data_ptr(), Rb[0, s].data_ptr(), stride, B * img, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
api.create(cfg, 0, B)
for k in range(K): api.process_device(Lb[k].data_ptr(), Rb[k].data_ptr(), stride, img)
api.synchronize()
us = (C.c_double * 12)(); api.lib.vslam_debug_ticks(api.ctx, us)
print("dbg us per stream per frame:", [round(v / K, 1) for v in us])
print("chrono ms/frame:", {k: round(v / K * 1e3, 3) for k, v in api.timers().items()})
