import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from vslam_pose_estimation_framework_amd import hip, synth
api = hip.load(); sy = synth.Synth(); scene = sy.scene_kitti(7); cfg = synth.config_for_scene(api, scene)
B, K = 8, 30
cfg.max_history_frames = K + 2
stride = 1280; img = cfg.rows * stride
dev = torch.device("cuda", 0)
Lb = torch.empty((K, B, cfg.rows, stride), dtype=torch.uint8, device=dev); Rb = torch.empty_like(Lb)
for s in range(B):
    sy.render_device(scene, 200 * s, K, Lb[0, s].data_ptr(), Rb[0, s].data_ptr(), stride, B * img, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
api.create(cfg, 0, B)
acc = []
for k in range(K):
    api.process_device(Lb[k].data_ptr(), Rb[k].data_ptr(), stride, img); api.synchronize()
    if k >= 10:
        for s in range(B):
            fi = api.frame_info(s)
            acc.append((fi.n_tracked, fi.n_lost, fi.n_recovered, fi.n_points, fi.n_active_landmarks, fi.n_after_prune))
a = np.array(acc, float)
print("mean tracked %.0f lost %.0f recovered %.1f points %.0f active_lm %.0f after_prune %.0f" % tuple(a.mean(axis=0)))
