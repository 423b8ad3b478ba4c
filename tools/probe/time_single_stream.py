"""Single-stream (exact mode) latency: per-kernel HIP-event times and the wall time per frame of ONE stream, for the fused
frame kernel and for the phase-split launch sequence (VSLAM_SPLIT).  Usage: python tools/probe/time_single_stream.py [streams] [frames]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from vslam_pose_estimation_framework_amd import hip, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
api = hip.load(); sy = synth.Synth()
dev = torch.device("cuda", 0)
scene = sy.scene_kitti(7); cfg = synth.config_for_scene(api, scene, "kitti")
cfg.max_keypoints = 8192; cfg.max_points = 4096; cfg.max_history_frames = 512
stride = 1280; img = cfg.rows * stride
L = torch.empty((N, B, cfg.rows, stride), dtype=torch.uint8, device=dev); R = torch.empty_like(L)
for s in range(B):
    sc = sy.scene_kitti(7 + 13 * s)
    sy.render_device(sc, 0, N, L[0, s].data_ptr(), R[0, s].data_ptr(), stride, B * img, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
api.create(cfg, 0, B)
for timers in (False, True):
    api.reset() if hasattr(api, "reset") else None
    api.enable_timers(timers)
    for k in range(40):
        api.process_device(L[k].data_ptr(), R[k].data_ptr(), stride, img)
    api.synchronize()
    t0 = time.perf_counter()
    for k in range(40, N):
        api.process_device(L[k].data_ptr(), R[k].data_ptr(), stride, img)
    api.synchronize()
    dt = time.perf_counter() - t0
    out = {"streams": B, "split": os.environ.get("VSLAM_SPLIT", "0"), "timers": timers, "ms_per_frame": round(dt / (N - 40) * 1e3, 4)}
    if timers:
        out["kernels_ms"] = {k: round(ms / max(n, 1), 4) for k, (ms, n) in api.kernel_times().items() if n > 0}
        out["chronometers_us_per_frame"] = {k: round(v / N * 1e6, 1) for k, v in api.timers().items()}
    print(json.dumps(out), flush=True)
api.destroy()
