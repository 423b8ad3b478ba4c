#!/usr/bin/env python3
"""bench.py — stereo frames/s of the hot path (FAST+BRIEF+stereo/temporal matching+StereoUVAligner) on MI355X.

One "step" = one pass of the whole per-frame hot path over a batch of B stereo pairs: one pair for each of
the B streams (chunks of the sequence) a GPU owns.  Workload (BASELINE.json configs[1]): KITTI-00-shaped
synthetic sequence (1241x376, 4541 frames), configuration_kitti.yaml values (bin 15 -> ~2158 keypoints per
image), cut into B contiguous chunks that start `overlap` frames early.  With N GPUs every
rank owns B further chunks (weak scaling: per-GPU work fixed), no data-path collective; the only exchange
is one all-gather of the per-frame 3x4 poses at the end of the timed region.

Prints ONE JSON line on rank 0.  `value` counts unique frames only: B*N*K/t * L/(L+overlap).  Exactly K steps are
timed for any K: the job is L+overlap steps long (the default K) and starts over (reset, rewind) when K exceeds it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vslam_pose_estimation_framework_amd import hip, sharding, synth  # noqa: E402

SEQ_FRAMES = 4541          # KITTI odometry sequence 00
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PMC_SUMMARY = "r01_h_pmc_traffic.json"   # tools/pmc_traffic.sh of this build (rocprofv3 --pmc, separate passes)


def load_oracle():
    """CPU oracle, used ONLY for the cpu_baseline leg (timing + sample parity check)."""
    so = os.path.join(ROOT, "oracle", "libvslam_oracle.so")
    try:
        from vslam_pose_estimation_framework_amd.capi import CApi
        return CApi(so, "orc_")
    except OSError:
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-B"])
        from vslam_pose_estimation_framework_amd.capi import CApi
        return CApi(so, "orc_")


def algorithmic_bytes(cfg, B, stats):
    """SURVEY.md §8(d) per-frame figures x the frames one launch processes (B)."""
    W, H = cfg.cols, cfg.rows
    N, P, M, I = stats["N"], stats["P"], stats["M"], stats["I"]
    per_frame = {
        "k_fast_box": 2 * W * H,                                   # each image byte read once
        "k_emit": 2 * W * H / 8 + 2 * N * 5,                        # corner masks in, keypoints out
        "k_brief": 2 * N * (4 + 32) + 2 * N * 512 * 2,             # keypoints in, descriptors out, 512 u16 taps
        "k_track_candidates": P * (24 + 32) + P * 32 * 4,           # previous points + in-window descriptors
        "k_frame": P * (24 + 64 + 8) + P * 64 + I * M * 64 + M * 9 + 2 * N * 32 + 96,
        "k_recover_brief": P * 0.5 * (64 + 2 * 512 * 2),               # lost points: previous descriptors + 2 x 512 taps
        "k_update_landmarks": M * (24 + 8) * 4,                        # a few measurements per tracked point
        "k_stereo_dist": 2 * N * 32 + N * 16,                          # descriptors in, 16 distances per left feature out
    }
    return {k: v * B for k, v in per_frame.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0 = the whole chunked job)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("VSLAM_BENCH_STREAMS", "160")))
    ap.add_argument("--overlap", type=int, default=10, help="warm-up frames per chunk (SURVEY.md 8e default)")
    ap.add_argument("--cpu-frames", type=int, default=240)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one rank per GPU (RCCL).  VSLAM_BENCH_BACKEND=gloo is a rehearsal aid only: several ranks on the one GPU of a
    # test box, same code path, the single all-gather staged through the host
    backend = os.environ.get("VSLAM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B = args.streams
    overlap = args.overlap
    L = -(-SEQ_FRAMES // B)                      # unique frames per chunk
    job_steps = L + overlap
    K = args.steps if args.steps > 0 else job_steps
    KB = min(K, job_steps)                       # frames held per chunk; K > job_steps: the job is run again (reset + rewind)
    W = max(0, args.warmup)

    api = hip.load()
    sy = synth.Synth()
    scene = sy.scene_kitti(seed=7)
    cfg = synth.config_for_scene(api, scene, "kitti")
    cfg.max_keypoints = 8192
    cfg.max_points = 4096
    cfg.max_history_frames = job_steps + 2
    stride = ((cfg.cols + 63) // 64) * 64
    img_bytes = cfg.rows * stride

    # ---- inputs resident in HBM: [step][stream][rows][stride] --------------------------------------------
    Lbuf = torch.empty((KB, B, cfg.rows, stride), dtype=torch.uint8, device=dev)
    Rbuf = torch.empty((KB, B, cfg.rows, stride), dtype=torch.uint8, device=dev)
    starts = []
    for s in range(B):
        gc = rank * B + s                        # global chunk index; rank r continues the virtual sequence
        start = max(0, gc * L - overlap)
        starts.append(start)
        sy.render_device(scene, start, KB, Lbuf[0, s].data_ptr(), Rbuf[0, s].data_ptr(), stride, B * img_bytes,
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()

    api.create(cfg, dev_index, B)

    def run_steps(n):
        for k in range(n):
            j = k % KB
            if j == 0 and k > 0:
                api.reset()                          # the whole job again: every chunk re-localises from its first frame
            api.process_device(Lbuf[j].data_ptr(), Rbuf[j].data_ptr(), stride, img_bytes)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # ---- warm-up (untimed), then the timed region ---------------------------------------------------------
    run_steps(min(W, K))
    api.synchronize()
    api.reset()
    pose_send = torch.zeros((B, KB, 12), dtype=torch.float64, device=dev)
    sharding.gather_poses(pose_send)              # untimed: RCCL sets its all-gather channels up on first use
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(K)
    api.copy_poses_device(0, min(KB, ((K - 1) % KB) + 1), pose_send.data_ptr())
    api.synchronize()
    all_poses = sharding.gather_poses(pose_send)          # RCCL all-gather (no-op for one GPU)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- frame statistics of the timed pass (sizes the algorithmic-byte figures) ----------------------------
    infos = [api.frame_info(s) for s in range(B)]
    flags = max(fi.error_flags for fi in infos)
    stats = {
        "N": float(np.mean([0.5 * (fi.n_keypoints_left + fi.n_keypoints_right) for fi in infos])),
        "P": float(np.mean([fi.n_points for fi in infos])),
        "M": float(np.mean([fi.n_tracked for fi in infos])),
        "I": float(np.mean([max(fi.aligner_iterations, 1) for fi in infos])),
    }

    # ---- instrumented pass: per-kernel device time with HIP events on the context stream ----------------------
    api.reset()
    api.enable_timers(True)
    run_steps(K)
    api.synchronize()
    ktimes = api.kernel_times()
    chrono = api.timers()
    api.enable_timers(False)

    out = None
    if rank == 0:
        frames_per_step = B * world
        eff = L / float(L + overlap)
        value = frames_per_step * K / elapsed * eff
        # the context processes its streams in G independent groups: one launch covers B/G streams
        groups = max(1, ktimes["k_frame"][1] // max(K, 1))
        abytes = algorithmic_bytes(cfg, B / groups, stats)
        kern = {}
        for name, (ms, n) in ktimes.items():
            avg_ms = ms / max(n, 1)
            kern[name] = {"avg_ms": round(avg_ms, 4), "launches": n,
                          "achieved_GBs": round(abytes[name] / (avg_ms * 1e-3) / 1e9, 2) if avg_ms > 0 else None}
        kern = {k_: v_ for k_, v_ in kern.items() if v_["launches"] > 0}
        dom = max(ktimes.items(), key=lambda kv: kv[1][0])[0]
        # HBM traffic of the dominant kernel: PMC counters cannot be read in-process; taken from the committed
        # rocprofv3 --pmc summary (separate FETCH_SIZE / WRITE_SIZE passes of this same command) when it matches B
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", PMC_SUMMARY)))
            if B == 160 and groups == 1 and dom in pm["per_launch_KB"]:
                traffic = int((pm["per_launch_KB"][dom]["FETCH_SIZE"] + pm["per_launch_KB"][dom]["WRITE_SIZE"]) * 1024)
        except (OSError, KeyError, ValueError):
            pass
        dom_avg_s = ktimes[dom][0] / max(ktimes[dom][1], 1) * 1e-3
        achieved = abytes[dom] / dom_avg_s / 1e9
        out = {
            "metric": "stereo frames/sec on KITTI-00 at 1/2/4/8 MI355X; ATE vs reference",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "KITTI-00-shaped synthetic stereo (1241x376, 4541 frames), configuration_kitti.yaml "
                                   "values, bin 15 (target 2158 kp/image), FAST+BRIEF-32, open loop",
                       "streams_per_gpu": B, "chunk_frames": L, "chunk_overlap": overlap,
                       "frames_per_step": frames_per_step, "stream_groups": groups, "streams_per_launch": B // groups,
                       "unique_frame_fraction": round(eff, 4),
                       "parallelism": "frame-sharded chunks, %d per GPU x %d GPU" % (B, world),
                       "mean_keypoints_per_image": round(stats["N"], 1), "mean_points_per_frame": round(stats["P"], 1),
                       "mean_tracked": round(stats["M"], 1), "mean_aligner_iterations": round(stats["I"], 1),
                       "error_flags": flags},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(abytes[dom]), "avg_launch_ms": round(dom_avg_s * 1e3, 4)},
            "kernels": kern,
            "chronometers_s": {k: round(v, 4) for k, v in chrono.items()},
        }

    # ---- CPU baseline: the oracle (a port of the reference path) on a bounded sample, rank 0, N=1 only ---------
    if rank == 0 and world == 1 and not args.no_cpu:
        orc = load_oracle()
        n_chunks = max(1, min(B, args.cpu_frames // KB)) if KB <= args.cpu_frames else 1
        per_chunk = min(KB, args.cpu_frames)
        orc.create(cfg, 0, 1)
        chk = hip.load()
        chk.create(cfg, dev_index, 1)
        cpu_t = 0.0
        mism, max_rel, nfr = 0, 0.0, 0
        for sidx in range(n_chunks):
            orc.reset()
            chk.reset()
            Lh = Lbuf[:per_chunk, sidx].cpu().numpy()
            Rh = Rbuf[:per_chunk, sidx].cpu().numpy()
            for k in range(per_chunk):
                a = time.perf_counter()
                orc.process_host(Lh[k], Rh[k])
                cpu_t += time.perf_counter() - a
                chk.process_host(Lh[k], Rh[k])
                fo, fg = orc.frame_info(0), chk.frame_info(0)
                for name in ("n_keypoints_left", "n_keypoints_right", "n_tracked", "n_inliers", "n_points", "status",
                             "n_recovered", "n_new_stereo", "window_pixels"):
                    if getattr(fo, name) != getattr(fg, name):
                        mism += 1
                To = np.array(fo.camera_left_to_world)
                Tg = np.array(fg.camera_left_to_world)
                max_rel = max(max_rel, float(np.linalg.norm(Tg - To) / np.linalg.norm(To)))
                nfr += 1
        cpu_fps = nfr / cpu_t
        out["cpu_baseline"] = {"value": round(cpu_fps, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d chunks x %d frames of the same synthetic sequence (oracle/libvslam_oracle.so, "
                                         "g++ -O3, 1 thread, process() time only)" % (n_chunks, per_chunk),
                               "parity_on_sample": {"frames": nfr, "int_field_mismatches": mism,
                                                    "max_pose_rel_frobenius": max_rel}}
        out["speedup_vs_cpu_port"] = round(out["value"] / cpu_fps, 1)
        orc.destroy()
        chk.destroy()

    if rank == 0:
        print(json.dumps(out))
    api.destroy()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
