"""In-kernel phase clocks of the frame kernel for ONE stream (exact mode).  Needs the profiling build:
  make -C vslam_pose_estimation_framework_amd/csrc variant NAME=libvslam_hip_prof.so EXTRA=-DVS_PROFILE_PHASES
  VSLAM_HIP_LIB=$PWD/vslam_pose_estimation_framework_amd/csrc/libvslam_hip_prof.so python tools/probe/phases_single_stream.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from vslam_pose_estimation_framework_amd import hip, synth
B = 1; K = 300
api = hip.load(); sy = synth.Synth(); scene = sy.scene_kitti(7); cfg = synth.config_for_scene(api, scene)
cfg.max_keypoints = 8192; cfg.max_points = 4096; cfg.max_history_frames = 512
stride = 1280; img = cfg.rows * stride
dev = torch.device("cuda", 0)
Lb = torch.empty((K, B, cfg.rows, stride), dtype=torch.uint8, device=dev); Rb = torch.empty_like(Lb)
sy.render_device(scene, 0, K, Lb[0, 0].data_ptr(), Rb[0, 0].data_ptr(), stride, B * img, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
api.create(cfg, 0, B)
buf = (C.c_ulonglong * (B * 17))()
for k in range(50):
    api.process_device(Lb[k].data_ptr(), Rb[k].data_ptr(), stride, img)
api.synchronize(); api.lib.vslam_debug_stream_ticks(api.ctx, buf)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).copy()
its = 0
for k in range(50, K):
    api.process_device(Lb[k].data_ptr(), Rb[k].data_ptr(), stride, img)
    if k % 10 == 0:
        api.synchronize(); its += api.frame_info(0).aligner_iterations
api.synchronize(); api.lib.vslam_debug_stream_ticks(api.ctx, buf)
d = (np.frombuffer(buf, dtype=np.uint64).astype(np.float64) - a) * 1e-2 / (K - 50)
print("chronometer clocks, us per frame [track, align, recover, landmark, stereo]:", [round(v, 1) for v in d[:5]])
print("phase stamps, us per frame [0..11]:", [round(v, 1) for v in d[5:]])
print("mean aligner iterations (sampled):", its / ((K - 50) // 10))
