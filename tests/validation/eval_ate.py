"""ATE of the chunked (frame-sharded) GPU run vs the sequential GPU run vs the CPU oracle, all against the synthetic
ground truth of the KITTI-00-shaped sequence.  Usage: python tests/validation/eval_ate.py [frames] [streams] [overlaps...]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from vslam_pose_estimation_framework_amd import hip, synth, sharding, evaluation as ev

total = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
overlaps = [int(v) for v in sys.argv[3:]] or [10, 6, 4]
with_oracle = os.environ.get("ATE_ORACLE_FRAMES", "0")
api = hip.load(); sy = synth.Synth(); scene = sy.scene_kitti(7); cfg = synth.config_for_scene(api, scene)
cfg.max_keypoints = 8192; cfg.max_points = 4096
stride = 1280; img = cfg.rows * stride; dev = torch.device("cuda", 0)
gt = np.array([sy.gt_pose(scene, k) for k in range(total)])
out = {"frames": total}

def render(first, n):
    L = torch.empty((n, cfg.rows, stride), dtype=torch.uint8, device=dev); R = torch.empty_like(L)
    sy.render_device(scene, first, n, L.data_ptr(), R.data_ptr(), stride, img, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return L, R

# sequential, one stream (exact mode)
cfg.max_history_frames = 512
api.create(cfg, 0, 1)
t0 = time.time(); CH = 256; flags = 0
for f0 in range(0, total, CH):
    n = min(CH, total - f0); L, R = render(f0, n)
    for k in range(n): api.process_device(L[k].data_ptr(), R[k].data_ptr(), stride, img)
    api.synchronize(); flags |= api.frame_info(0).error_flags
seq = api.poses(0, 0, total); out["sequential_s"] = round(time.time() - t0, 2); out["sequential_error_flags"] = flags
out["ate_sequential_aligned"] = ev.ate_rmse(seq, gt); out["ate_sequential_raw_first_frame_aligned"] = ev.ate_rmse(
    np.array([ev.mul34(gt[0], T) for T in seq]), gt, align=False)
api.destroy()

for ov in overlaps:
    plan, Lc = sharding.plan_chunks(total, B, ov)
    steps = max(e - s for (s, f, e) in plan)
    cfg.max_history_frames = steps + 2
    api.create(cfg, 0, B)
    Lb = torch.empty((steps, B, cfg.rows, stride), dtype=torch.uint8, device=dev); Rb = torch.empty_like(Lb)
    for s_, (st, fi, en) in enumerate(plan):
        sy.render_device(scene, st, steps, Lb[0, s_].data_ptr(), Rb[0, s_].data_ptr(), stride, B * img, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for k in range(steps): api.process_device(Lb[k].data_ptr(), Rb[k].data_ptr(), stride, img)
    api.synchronize()
    chunks = [api.poses(s_, 0, en - st) for s_, (st, fi, en) in enumerate(plan)]
    asm = sharding.assemble_trajectory(chunks, plan)
    a = ev.ate_rmse(asm, gt)
    out["ate_chunked_B%d_overlap%d" % (B, ov)] = a
    out["rel_diff_vs_sequential_overlap%d" % ov] = (a - out["ate_sequential_aligned"]) / out["ate_sequential_aligned"]
    api.destroy(); del Lb, Rb

nor = int(with_oracle)
if nor > 0:
    from _oracle import Oracle
    o = Oracle(); o.create(cfg, 0, 1); t0 = time.time()
    for f0 in range(0, nor, CH):
        n = min(CH, nor - f0); L, R = render(f0, n); Lh, Rh = L.cpu().numpy(), R.cpu().numpy()
        for k in range(n): o.process_host(Lh[k], Rh[k])
    op = o.poses(0, 0, nor); out["oracle_frames"] = nor; out["oracle_s"] = round(time.time() - t0, 1)
    out["ate_oracle_aligned"] = ev.ate_rmse(op, gt[:nor]); out["ate_gpu_sequential_same_frames"] = ev.ate_rmse(seq[:nor], gt[:nor])
    out["max_pose_rel_frobenius_gpu_vs_oracle"] = float(max(np.linalg.norm(seq[k] - op[k]) / np.linalg.norm(op[k]) for k in range(nor)))
path = float(np.sum(np.linalg.norm(np.diff(gt[:, :, 3], axis=0), axis=1))); out["path_length_m"] = path
print(json.dumps(out))
