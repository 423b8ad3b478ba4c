"""Is the chunked (frame-sharded) trajectory systematically worse than the sequential one?  Open-loop drift is a random
walk: one sequence gives one realization of it, and two runs of equal quality differ by far more than 1 % in ATE.  This
tool repeats the comparison of tests/validation/eval_ate.py over several scene seeds (same path, different texture / noise, i.e.
different keypoints) and reports the distribution of ATE(chunked) / ATE(sequential).
Usage: python tools/eval_ate_seeds.py [streams] [overlap] [seeds...]"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vslam_pose_estimation_framework_amd import hip, synth, sharding, evaluation as ev

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
ov = int(sys.argv[2]) if len(sys.argv) > 2 else 5
seeds = [int(v) for v in sys.argv[3:]] or [7, 11, 13, 17, 19, 23]
total = 4541
api = hip.load(); sy = synth.Synth()
stride = 1280; dev = torch.device("cuda", 0)
rows_out = []
for seed in seeds:
    scene = sy.scene_kitti(seed); cfg = synth.config_for_scene(api, scene)
    cfg.max_keypoints = 8192; cfg.max_points = 4096
    img = cfg.rows * stride
    gt = np.array([sy.gt_pose(scene, k) for k in range(total)])
    # sequential, one stream (exact mode)
    cfg.max_history_frames = 512
    api.create(cfg, 0, 1)
    CH = 256; flags = 0
    for f0 in range(0, total, CH):
        n = min(CH, total - f0)
        L = torch.empty((n, cfg.rows, stride), dtype=torch.uint8, device=dev); R = torch.empty_like(L)
        sy.render_device(scene, f0, n, L.data_ptr(), R.data_ptr(), stride, img, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for k in range(n): api.process_device(L[k].data_ptr(), R[k].data_ptr(), stride, img)
        api.synchronize(); flags |= api.frame_info(0).error_flags
    seq = api.poses(0, 0, total)
    api.destroy()
    a_seq = ev.ate_rmse(seq, gt)
    # chunked, B streams
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
    flags_c = 0
    for s_ in range(B): flags_c |= api.frame_info(s_).error_flags
    chunks = [api.poses(s_, 0, en - st) for s_, (st, fi, en) in enumerate(plan)]
    api.destroy(); del Lb, Rb
    a_chk = ev.ate_rmse(sharding.assemble_trajectory(chunks, plan), gt)
    seam = {str(k): ev.ate_rmse(sharding.assemble_trajectory(chunks, plan, seam_frames=k), gt) / a_seq for k in (2, 3, 5) if k <= ov}
    rows_out.append({"seed": seed, "ate_sequential": a_seq, "ate_chunked": a_chk, "ratio": a_chk / a_seq, "ratio_by_seam_frames": seam,
                     "error_flags": int(flags | flags_c)})
    print(json.dumps(rows_out[-1]), flush=True)
r = np.array([x["ratio"] for x in rows_out]); lr = np.log(r)
print(json.dumps({"streams": B, "overlap": ov, "frames": total, "seeds": seeds, "ratio_mean": float(r.mean()), "ratio_geomean": float(np.exp(lr.mean())),
                  "ratio_min": float(r.min()), "ratio_max": float(r.max()), "log_ratio_std": float(lr.std(ddof=1)) if len(r) > 1 else None,
                  "log_ratio_stderr": float(lr.std(ddof=1) / np.sqrt(len(r))) if len(r) > 1 else None,
                  "ate_sequential_mean": float(np.mean([x["ate_sequential"] for x in rows_out])),
                  "ate_chunked_mean": float(np.mean([x["ate_chunked"] for x in rows_out])),
                  "ratio_geomean_by_seam_frames": {k: float(np.exp(np.mean([np.log(x["ratio_by_seam_frames"][k]) for x in rows_out]))) for k in rows_out[0]["ratio_by_seam_frames"]}}))
