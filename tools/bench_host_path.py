"""PCIe-inclusive rate of the hot path: the same workload as bench.py (KITTI-00-shaped, B streams), but every step's
2 x B images are handed over as HOST buffers (`vslam_process_host`) instead of being resident in HBM.  Never the
bench `value` — the figure a caller without device-resident images should expect.
Usage: python tools/bench_host_path.py [B] [K]   ->  one JSON line per host-buffer kind."""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from vslam_pose_estimation_framework_amd import hip, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
api = hip.load(); sy = synth.Synth(); scene = sy.scene_kitti(7); cfg = synth.config_for_scene(api, scene, "kitti")
cfg.max_keypoints = 8192; cfg.max_points = 4096; cfg.max_history_frames = K + 2
dev = torch.device("cuda", 0)
api.create(cfg, 0, B)


def run(kind, stride, pinned):
    img = cfg.rows * stride
    Ld = torch.empty((K, B, cfg.rows, stride), dtype=torch.uint8, device=dev); Rd = torch.empty_like(Ld)
    for s in range(B):
        sy.render_device(scene, 28 * s, K, Ld[0, s].data_ptr(), Rd[0, s].data_ptr(), stride, B * img, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Lh = torch.empty(Ld.shape, dtype=torch.uint8, pin_memory=pinned); Rh = torch.empty(Rd.shape, dtype=torch.uint8, pin_memory=pinned)
    Lh.copy_(Ld); Rh.copy_(Rd)
    del Ld, Rd
    fn = api.fn("process_host")

    def steps(n):
        for k in range(n):
            api.check(fn(api.ctx, C.cast(Lh[k].data_ptr(), C.POINTER(C.c_uint8)), C.cast(Rh[k].data_ptr(), C.POINTER(C.c_uint8)),
                         C.c_int32(stride), C.c_size_t(img)))
    api.reset(); steps(2); api.synchronize(); api.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); steps(K); api.synchronize(); dt = time.perf_counter() - t0
    info = api.frame_info(0)
    mb = 2.0 * B * cfg.rows * cfg.cols / 1e6
    print(json.dumps({"host_buffers": kind, "streams": B, "steps": K, "ms_per_step": round(dt / K * 1e3, 3),
                      "pairs_per_s": round(B * K / dt, 1), "image_MB_per_step": round(mb, 1),
                      "pcie_GBs": round(mb * K / dt / 1e3, 2), "error_flags": int(info.error_flags)}), flush=True)


run("pinned, 64-byte-aligned rows (one copy per side)", ((cfg.cols + 63) // 64) * 64, True)
run("pinned, dense rows (row stride = cols, unaligned)", cfg.cols, True)
run("pageable, 64-byte-aligned rows", ((cfg.cols + 63) // 64) * 64, False)
