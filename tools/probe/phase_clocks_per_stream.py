"""Per-stream, per-step phase durations of k_frame (in-kernel clocks; build the library with
`make -C vslam_pose_estimation_framework_amd/csrc clean all EXTRA=-DVS_PROFILE_PHASES` first): where the slowest stream of a step spends
its time.  Usage: python tools/probe/phase_clocks_per_stream.py [B] [K]      (HEAVY=1: the scene of bench.py --contrast 2 --speed 0.3 --bin 11;
VSLAM_HIP_LIB=<variant library> selects a probe build without touching the product's)"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from vslam_pose_estimation_framework_amd import hip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
api = hip.load(); sy = synth.Synth(); scene = sy.scene_kitti(7)
if os.environ.get("HEAVY"):
    scene.speed_m = 0.3; scene.contrast = 2.0
cfg = synth.config_for_scene(api, scene)
if os.environ.get("HEAVY"):
    cfg.bin_size_pixels = 11; cfg.max_keypoints = 16384; cfg.max_points = 8192
cfg.max_history_frames = K + 2
stride = 1280; img = cfg.rows * stride
dev = torch.device("cuda", 0)
Lb = torch.empty((K, B, cfg.rows, stride), dtype=torch.uint8, device=dev); Rb = torch.empty_like(Lb)
for s in range(B):
    sy.render_device(scene, 28 * s, K, Lb[0, s].data_ptr(), Rb[0, s].data_ptr(), stride, B * img, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
api.create(cfg, 0, B)
buf = (C.c_ulonglong * (B * 17))()
prev = np.zeros((B, 17))
names = ["track", "align", "recover", "landmark", "stereo"]
rows = []
for k in range(K):
    api.process_device(Lb[k].data_ptr(), Rb[k].data_ptr(), stride, img)
    api.synchronize()
    api.lib.vslam_debug_stream_ticks(api.ctx, buf)
    cur = np.frombuffer(buf, dtype=np.uint64).reshape(B, 17).astype(np.float64)
    d = (cur - prev) * 1e-2   # us
    prev = cur
    tot = d[:, 5 + 8]
    w = int(np.argmax(tot))
    info = [api.frame_info(s) for s in (w,)]
    rows.append(d)
    if k >= 2:
        print("step %2d  mean %6.0f  p90 %6.0f  max %6.0f us (stream %3d: " % (k, tot.mean(), np.percentile(tot, 90), tot.max(), w)
              + " ".join("%s %.0f" % (n, d[w, i]) for i, n in enumerate(names))
              + " | its %d attempts %d trk %d pts %d)" % (info[0].aligner_iterations, info[0].track_attempts, info[0].n_tracked, info[0].n_points))
d = np.stack(rows[2:])
print("mean over streams/steps (us):", {n: round(float(d[:, :, i].mean()), 1) for i, n in enumerate(names)}, "total", round(float(d[:, :, 13].mean()), 1))
print("mean of per-step max total:", round(float(d[:, :, 13].max(axis=1).mean()), 1))
print("track: jacobi iterations/frame (x100 ticks->count)", round(float(d[:, :, 12].mean()) * 100, 2), "iterate us", round(float(d[:, :, 14].mean()), 1), "flags+compaction us", round(float(d[:, :, 15].mean()), 1))
print("stereo cumulative: after sub/prepass", round(float(d[:, :, 10].mean()), 1), "after step A", round(float(d[:, :, 6].mean()), 1))
print("stereo staging us", round(float(d[:, :, 16].mean()), 1))
print("aligner per frame (us): compute+reduce+sync %.1f, gather %.1f, solve %.1f, update+sync %.1f" % tuple(float(d[:, :, 5 + k].mean()) for k in (5, 9, 10, 11)))
print("prune (dbg6)", round(float(d[:, :, 11].mean()), 1), "stereo stamps dbg0-4:", [round(float(d[:, :, 5 + i].mean()), 1) for i in range(5)])
