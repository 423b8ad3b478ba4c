#!/usr/bin/env python3
"""bench.py — stereo frames/s of the hot path (FAST+BRIEF+stereo/temporal matching+StereoUVAligner) on MI355X.

One "step" = one pass of the whole per-frame hot path over a batch of B stereo pairs: one pair for each of the B
streams a GPU owns.

--mode chunks (default; BASELINE.json configs[1]): KITTI-00-shaped synthetic sequence (1241x376, 4541 frames),
  configuration_kitti.yaml values (bin 15 -> target 2158 keypoints per image), cut into B contiguous chunks that start
  `overlap` frames early (default 6, see --overlap).  The chunks run as a steady-state pipeline: stream s is `phase_s` frames into
  its chunk when the timed region starts and restarts (vslam_reset_stream, asynchronous) whenever its chunk ends, so ANY
  window of K steps sees the stationary mix of warm-up and unique frames.  `value` counts only the unique frames
  (frames inside their chunk's own range) produced inside the timed region.  With N GPUs every rank owns B further
  chunks (weak scaling), no data-path collective; one all-gather of the per-step 3x4 poses ends the timed region.
--mode sequences (configs[2] / configs[4], the exact mode): whole KITTI-shaped sequences, one per stream, assigned to
  ranks longest first (sharding.plan_sequences); N = 4: 00+02+05+06, otherwise 00..10 (with --bin 11 for config #5).

Prints ONE JSON line on rank 0.  At N = 1 the line also carries: `exact_mode` (whole sequences per stream),
`pcie_inclusive` (host images through vslam_process_host), `cpu_baseline` (the oracle on the host cores)."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0 = one whole chunk job / whole sequences)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["chunks", "sequences"], default="chunks")
    ap.add_argument("--sequences", default="", help="--mode sequences: comma-separated KITTI sequence numbers")
    ap.add_argument("--bin", type=int, default=15, help="bin_size_pixels (15: ~2158 kp/image, 22: ~1026, 11: ~3955)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("VSLAM_BENCH_STREAMS", "160")))
    ap.add_argument("--scene", choices=["kitti", "euroc"], default="kitti", help="euroc: MH_01-shaped 752x480 scene with configuration_euroc.yaml values (not the headline workload)")
    ap.add_argument("--contrast", type=float, default=0.0, help="texture contrast of the synthetic scene (0 = the scene's default 1.0; 2.0: ~60 %% more corners, stereo points and tracks per frame: with --speed 0.3 --bin 11 a workload of the size SURVEY.md 8(d) assumed; not the headline)")
    ap.add_argument("--speed", type=float, default=0.0, help="camera speed of the synthetic scene in m/frame (0 = the scene's default 0.9; slower = more of the points tracked)")
    ap.add_argument("--overlap", type=int, default=6, help="warm-up frames per chunk.  SURVEY.md 8e proposed 10 without data; measured (48 sensor-noise seeds each, profiles/r03_ate_noise_overlaps48.json + r03_ate_noise_seeds48.json): 2, 3, 4, 5, 6, 8 and 10 all lie inside the sequential run's own ATE spread (|Welch t| <= 1.3), so the default is three times the two frames a chunk needs before its poses are aligned against landmarks")
    ap.add_argument("--cpu-frames", type=int, default=240)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0 = min(nproc, 16): the box's CPU share)")
    ap.add_argument("--exact-frames", type=int, default=1200, help="frames per sequence of the exact-mode legs (0 = whole sequences)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="chunks at N > 1: weak = one KITTI-00-shaped sequence per GPU (per-GPU work fixed); strong = the ONE sequence's chunk plan spread over the GPUs")
    ap.add_argument("--no-ate", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-exact", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--no-shim", action="store_true")
    ap.add_argument("--only-shim", action="store_true", help="print the shim_path block alone (one JSON line) and stop")
    return ap.parse_args(argv)


def self_launch():
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) and N > 1: start N ranks of this script, one per
    GPU, and relay rank 0's JSON line.  Runs BEFORE torch / libvslam_hip.so are imported: the parent never touches the GPU.  Under a
    launcher, WORLD_SIZE must equal --gpus (exit 2 otherwise)."""
    from vslam_pose_estimation_framework_amd import launch       # standard library only
    args = parse_args()
    if launch.check_world(args.gpus) is None and args.gpus > 1:
        sys.exit(launch.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    return args


ARGS = self_launch() if __name__ == "__main__" else None

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vslam_pose_estimation_framework_amd import buildinfo, hip, sharding, synth  # noqa: E402

KITTI_FRAMES = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]   # odometry sequences 00..10
SEQ_FRAMES = KITTI_FRAMES[0]
EUROC_MH01_FRAMES = 3682   # MH_01_easy stereo pairs
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PMC_SUMMARY = "r04_pmc_traffic.json"   # tools/pmc_traffic.sh (rocprofv3 --pmc, separate passes); carries the source hash of its build
SQ_SUMMARY = "r04_sq_counters.json"    # tools/pmc_sq.sh (SQ / GRBM counters per kernel, kernels serialised on one HIP stream); carries the source hash as well
ATE_NOISE_STUDY = ["r04_ate_noise_rel48.json", "r03_ate_noise_seeds48.json", "r03_ate_noise_overlaps48.json", "r03_ate_noise_seeds16.json", "r03_ate_noise_seeds.json"]   # tools/eval_ate_noise.py: sequential ATE spread under sensor noise vs chunked (48 seeds for the default configuration and for B = 160 at overlaps 2 .. 8, 16 for B = 144 / 160, 8 seeds for the B x overlap grid)
METRIC = "stereo frames/sec on KITTI-00 at 1/2/4/8 MI355X; ATE vs reference"
KERNELS = ["k_fast_box", "k_emit", "k_brief", "k_track_candidates", "k_frame", "k_recover_brief", "k_update_landmarks", "k_stereo_dist"]


def load_oracle(native=False):
    """CPU oracle, used ONLY for the cpu_baseline leg (timing + sample parity check).  native: a -march=native build made on
    this host; falls back to the portable build that travelled with the repository."""
    from vslam_pose_estimation_framework_amd.capi import CApi
    odir = os.path.join(ROOT, "oracle")
    if native:
        so = os.path.join(odir, "libvslam_oracle_native.so")
        try:
            subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++14", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
                                   "-shared", "-o", so, os.path.join(odir, "vslam_oracle.cpp")], stderr=subprocess.DEVNULL, timeout=180)
            return CApi(so, "orc_"), "-O3 -march=native"
        except (OSError, subprocess.SubprocessError):
            pass
    so = os.path.join(odir, "libvslam_oracle.so")
    try:
        return CApi(so, "orc_"), "-O3 -march=x86-64-v3"
    except OSError:
        subprocess.check_call(["make", "-C", odir, "-s", "-B"])
        return CApi(so, "orc_"), "-O3 -march=x86-64-v3"


def algorithmic_bytes(cfg, B, stats):
    """SURVEY.md §8(d) per-frame figures x the frames one launch processes (B)."""
    W, H = cfg.cols, cfg.rows
    N, P, M, I = stats["N"], stats["P"], stats["M"], stats["I"]
    per_frame = {
        "k_fast_box": 2 * W * H,                                   # each image byte read once
        "k_emit": 2 * W * H / 8 + 2 * N * 5,                        # corner masks in, keypoints out
        "k_brief": 2 * N * (4 + 32),                                # keypoints in, descriptors out (§8d: 2·N·36)
        "k_track_candidates": P * (24 + 32) + P * 32 * 4,           # previous points + in-window descriptors
        "k_frame": P * (24 + 64 + 8) + P * 64 + I * M * 64 + M * 9 + 2 * N * 32 + 96,
        "k_recover_brief": stats.get("R", 0.0) * (64 + 2 * 512 * 2),   # projected lost landmarks: previous descriptors + 2 x 512 box taps
        "k_update_landmarks": M * (24 + 8) * 4,                        # a few measurements per tracked point
        "k_stereo_dist": 2 * N * 32 + N * 16,                          # descriptors in, 16 distances per left feature out
    }
    # one launch of the fused frame kernel also does the recovery descriptors and the landmark refinement (their own kernels when
    # the frame is split into phase launches).  SURVEY.md 8(d) lists neither: they are reported separately, never inside `frac`.
    extra = {"k_frame": (per_frame["k_recover_brief"] + per_frame["k_update_landmarks"]) * B} if stats.get("fused", True) else {}
    # SURVEY.md 8(d) asks for BOTH figures of the frame path: with the aligner's I·M·64 "re-read per iteration" term and without it
    # (those re-reads are register / LDS resident: M < 512 lanes, one measurement per lane, nothing of them reaches HBM)
    extra["k_frame_hbm_only"] = (per_frame["k_frame"] - I * M * 64) * B
    return {k: v * B for k, v in per_frame.items()}, extra


def frame_stats(api, streams):
    infos = [api.frame_info(s) for s in streams]
    return {
        "N": float(np.mean([0.5 * (fi.n_keypoints_left + fi.n_keypoints_right) for fi in infos])),
        "P": float(np.mean([fi.n_points for fi in infos])),
        "M": float(np.mean([fi.n_tracked for fi in infos])),
        "I": float(np.mean([max(fi.aligner_iterations, 1) for fi in infos])),
        "R": float(np.mean([fi.n_recovered for fi in infos])) / 0.87,     # projected landmarks: 87 % of them pass the descriptor gates
    }, max(fi.error_flags for fi in infos)


def kernel_report(api, cfg, B, stats, launches_per_step_hint):
    ktimes = api.kernel_times()
    groups = max(1, launches_per_step_hint)
    abytes, extra = algorithmic_bytes(cfg, B / groups, stats)
    kern = {}
    for name, (ms, n) in ktimes.items():
        if n <= 0:
            continue
        avg_ms = ms / n
        kern[name] = {"avg_ms": round(avg_ms, 4), "launches": n,
                      "achieved_GBs": round(abytes[name] / (avg_ms * 1e-3) / 1e9, 2) if avg_ms > 0 else None}
    dom = max(ktimes.items(), key=lambda kv: kv[1][0])[0]
    dom_avg_s = ktimes[dom][0] / max(ktimes[dom][1], 1) * 1e-3
    return kern, dom, dom_avg_s, abytes, extra


def sq_utilisation():
    """Per-kernel VALU / SALU issue utilisation from the committed SQ counter summary (tools/pmc_sq.sh), only when it was recorded on
    THIS build.  valu_util = SQ_ACTIVE_INST_VALU / (32 x GRBM_GUI_ACTIVE): SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles, what
    one wave64 VALU instruction holds its SIMD's issue for), GRBM_GUI_ACTIVE is summed over the 8 XCDs, the chip has 1024 SIMDs:
    busy SIMD-cycles 4 x ACTIVE over available SIMD-cycles (GRBM / 8) x 1024.  salu_util = SQ_INSTS_SALU / (32 x GRBM_GUI_ACTIVE): one
    scalar instruction per cycle and CU (256 CUs).  Each kernel alone on the chip (the counters need serialised kernels)."""
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", SQ_SUMMARY)))
    except (OSError, ValueError):
        return {}, "no counter file profiles/%s" % SQ_SUMMARY
    here = buildinfo.source_sha16()
    if sq.get("source_sha16") != here:
        return {}, "profiles/%s was recorded on source %s, this build is %s: not reported" % (SQ_SUMMARY, sq.get("source_sha16"), here)
    out = {}
    for k, c in sq.get("per_kernel", {}).items():
        g = c.get("GRBM_GUI_ACTIVE", 0.0)
        if g > 0 and "SQ_ACTIVE_INST_VALU" in c and "SQ_INSTS_SALU" in c:
            out[k] = {"valu_util": round(c["SQ_ACTIVE_INST_VALU"] / (32.0 * g), 4), "salu_util": round(c["SQ_INSTS_SALU"] / (32.0 * g), 4)}
            if "SQ_ACTIVE_INST_LDS" in c:
                out[k]["lds_util"] = round(c["SQ_ACTIVE_INST_LDS"] / (32.0 * g), 4)
    return out, None


def pmc_traffic(kernel, streams):
    """HBM bytes per launch of `kernel` from the committed counter summary (tools/pmc_traffic.sh: FETCH_SIZE + WRITE_SIZE, separate
    rocprofv3 --pmc passes) — only when that file was recorded on THIS build (same source hash) at this stream count."""
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    try:
        pm = json.load(open(path))
    except (OSError, ValueError):
        return None, "no counter file profiles/%s" % PMC_SUMMARY
    here = buildinfo.source_sha16()
    if pm.get("source_sha16") != here:
        return None, "profiles/%s was recorded on source %s, this build is %s: not reported" % (PMC_SUMMARY, pm.get("source_sha16"), here)
    if pm.get("streams") != streams or kernel not in pm.get("per_launch_KB", {}):
        return None, "profiles/%s holds no %s at %d streams" % (PMC_SUMMARY, kernel, streams)
    k = pm["per_launch_KB"][kernel]
    return int((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024), None


class Bench(object):
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        # one rank per GPU (RCCL).  VSLAM_BENCH_BACKEND=gloo is a rehearsal aid only: several ranks on the one GPU of a
        # test box, same code path, the single all-gather staged through the host
        self.backend = os.environ.get("VSLAM_BENCH_BACKEND", "nccl")
        if self.backend == "nccl" and self.world > torch.cuda.device_count():
            raise SystemExit("bench.py: %d ranks on RCCL need %d GPUs, this node shows %d (one process per GPU; VSLAM_BENCH_BACKEND=gloo "
                             "rehearses several ranks on one card)" % (self.world, self.world, torch.cuda.device_count()))
        self.dev_index = local_rank if self.backend == "nccl" else local_rank % torch.cuda.device_count()
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        if self.world > 1:
            import torch.distributed as dist
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
        self.sy = synth.Synth()
        self.euroc = args.scene == "euroc"
        # weak scaling: every GPU drives through its own world (rank 0: the single-GPU scene)
        seed = 7 + (101 * self.rank if args.scaling == "weak" else 0)
        self.scene = self.sy.scene_euroc(seed=seed) if self.euroc else self.sy.scene_kitti(seed=seed)
        if args.speed > 0:
            self.scene.speed_m = args.speed
        if args.contrast > 0:
            self.scene.contrast = args.contrast
        self.api = hip.load()
        self.cfg = synth.config_for_scene(self.api, self.scene, "euroc" if self.euroc else "kitti")
        if not self.euroc or args.bin != 15:
            self.cfg.bin_size_pixels = args.bin      # euroc: configuration_euroc.yaml's bin 20 unless --bin is given
        self.seq_frames = EUROC_MH01_FRAMES if self.euroc else SEQ_FRAMES
        self.cfg.max_keypoints = 8192 if args.bin >= 15 else 16384
        self.cfg.max_points = 4096 if args.bin >= 15 else 8192
        self.stride = ((self.cfg.cols + 63) // 64) * 64
        self.img_bytes = self.cfg.rows * self.stride

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier()

    def max_over_ranks(self, seconds):
        if self.world > 1:
            tt = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            return float(tt.item())
        return seconds

    def sum_over_ranks(self, value):
        if self.world > 1:
            tt = torch.tensor([float(value)], dtype=torch.float64, device=self.dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.SUM)
            return float(tt.item())
        return float(value)

    def render(self, scene, first, n, L, R, s, n_slots):
        """frames first..first+n-1 of `scene` into slabs 0..n-1 of L/R [slab][stream] at stream slot s."""
        self.sy.render_device(scene, first, n, L[0, s].data_ptr(), R[0, s].data_ptr(), self.stride, n_slots * self.img_bytes,
                              torch.cuda.current_stream().cuda_stream)

    # ------------------------------------------------------------------------------------------------------------------
    def open_pose_comm(self):
        """N > 1 on RCCL: the pose all-gather of the timed region goes through the C ABI (vslam_comm_init / vslam_allgather_poses, RCCL
        called by libvslam_hip.so itself — what a C++ caller uses).  The communicator is formed BEFORE anything is timed, in a worker
        thread under a time limit, and its result on a probe block is compared with torch.distributed's.  Whatever goes wrong on any
        rank (librccl.so not loadable, ncclCommInitRank refused, no communicator inside the limit, a result that differs), the ranks
        agree over torch.distributed and ALL of them let torch.distributed (RCCL as well) carry the timed all-gather; the JSON line
        and stderr say which path ran and why.  VSLAM_BENCH_C_ABI_COMM=0 skips the attempt."""
        if self.world == 1:
            return None, "not used (one GPU)"
        if self.backend != "nccl":
            return None, "torch.distributed %s (rehearsal backend)" % self.backend
        if os.environ.get("VSLAM_BENCH_C_ABI_COMM", "1") == "0":
            return None, "torch.distributed nccl (RCCL); the C-ABI all-gather was switched off: VSLAM_BENCH_C_ABI_COMM=0"
        box = {}

        def form():
            try:
                torch.cuda.set_device(self.dev_index)
                comm = sharding.PoseComm(self.api, self.rank, self.world, self.dev_index)
                probe = torch.arange(2 * 3 * 12, dtype=torch.float64, device=self.dev).reshape(2, 3, 12) + 1000.0 * self.rank
                got = comm.allgather(probe)
                want = torch.cat([probe - 1000.0 * self.rank + 1000.0 * r for r in range(self.world)])     # rank-major blocks
                box["comm"] = comm
                box["r"] = "ok" if torch.equal(got, want) else "MISMATCH against the expected rank-major blocks"
            except Exception as e:
                box["r"] = "failed: %s" % str(e)[:300]
        th = threading.Thread(target=form, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("VSLAM_BENCH_COMM_TIMEOUT", "90")))
        verdict = box.get("r", "timeout: the RCCL communicator did not form inside the limit (the worker thread is abandoned)")
        have = verdict == "ok"
        if sharding.all_ranks_ok(have):
            return box["comm"], "C ABI (vslam_allgather_poses: RCCL called by libvslam_hip.so), probe block verified"
        if have:
            box["comm"].destroy()
        sys.stderr.write("bench.py rank %d: C-ABI pose all-gather not used (%s): torch.distributed carries the all-gather\n" % (self.rank, verdict))
        sys.stderr.flush()
        return None, "torch.distributed nccl (RCCL); C-ABI communicator unavailable on some rank, this rank: %s" % verdict[:160]

    def run_chunks(self):
        a, api, cfg = self.args, self.api, self.cfg
        overlap, world, rank = a.overlap, self.world, self.rank
        job = sharding.chunk_job(self.seq_frames, a.streams, overlap, rank, world, a.scaling)
        B, L, J = job["n_streams"], job["L"], job["J"]
        K = a.steps if a.steps > 0 else J
        W = max(0, a.warmup)
        cfg.max_history_frames = J + 2
        starts, phase = job["starts"], job["phase"]
        # inputs resident in HBM: slab j, stream s = chunk frame (j + phase_s) % J of stream s
        Lbuf = torch.empty((J, B, cfg.rows, self.stride), dtype=torch.uint8, device=self.dev)
        Rbuf = torch.empty_like(Lbuf)
        for s in range(B):
            p = phase[s]
            self.render(self.scene, starts[s] + p, J - p, Lbuf, Rbuf, s, B)                    # slabs 0 .. J-p-1
            if p:
                self.render(self.scene, starts[s], p, Lbuf[J - p:], Rbuf[J - p:], s, B)        # slabs J-p .. J-1
        torch.cuda.synchronize()
        api.create(cfg, self.dev_index, B)
        comm, comm_note = self.open_pose_comm()
        counter = [0]

        def run_steps(n):
            for _ in range(n):
                k = counter[0]
                j = k % J
                if k > 0:                                # streams whose chunk starts over: fresh sequences, queued asynchronously
                    api.reset_streams(sharding.chunk_job_restarts(job, k))
                api.process_device(Lbuf[j].data_ptr(), Rbuf[j].data_ptr(), self.stride, self.img_bytes)
                counter[0] = k + 1

        def allgather(t):
            return comm.allgather(t) if comm is not None else sharding.gather_poses(t)

        # pre-roll: every stream passes one restart so that the timed region starts in the pipeline's steady state
        preroll = J
        run_steps(preroll + W)
        api.synchronize()
        pose_send = torch.zeros((K, job["streams_padded"], 12), dtype=torch.float64, device=self.dev)    # same shape on every rank (a strong plan can be ragged)
        allgather(pose_send)                          # untimed: RCCL sets its all-gather channels up on first use
        self.barrier()
        torch.cuda.synchronize()
        k_first = counter[0]
        t0 = time.perf_counter()
        for i in range(K):
            run_steps(1)
            api.copy_current_poses_device(pose_send[i].data_ptr())
        api.synchronize()
        allgather(pose_send)                          # ONE RCCL all-gather of the per-step poses ends the timed region (no-op for one GPU)
        torch.cuda.synchronize()
        self.barrier()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        unique = self.sum_over_ranks(sharding.chunk_job_unique_frames(job, k_first, K))
        frames = self.sum_over_ranks(B * K)
        stats, flags = frame_stats(api, range(B))
        if comm is not None:
            comm.destroy()

        # instrumented pass: per-kernel device time with HIP events on the context's HIP streams
        api.enable_timers(True)
        run_steps(K)
        api.synchronize()
        kern, dom, dom_avg_s, abytes, extra = kernel_report(api, cfg, B, stats, 1)
        chrono = api.timers()
        api.enable_timers(False)
        self.Lbuf, self.Rbuf, self.J, self.B = Lbuf, Rbuf, J, B
        self.starts, self.phase, self.job = starts, phase, job
        self.run_steps, self.counter = run_steps, counter
        if rank != 0:
            return None
        traffic, traffic_note = pmc_traffic(dom, B)
        achieved = abytes[dom] / dom_avg_s / 1e9
        if self.euroc:
            what = ("EuRoC-MH_01-shaped synthetic stereo (752x480, 3682 frames, 6-DoF), configuration_euroc.yaml values (2x2 FAST detectors, "
                    "bin %d: target %d kp/image, ORB extractor on the FAST keypoints), open loop; ")
        else:
            what = "KITTI-00-shaped synthetic stereo (1241x376, 4541 frames), configuration_kitti.yaml values, bin %d (target %d kp/image), FAST+BRIEF-32, open loop; "
        what = what % (cfg.bin_size_pixels, (cfg.cols // cfg.bin_size_pixels + 1) * (cfg.rows // cfg.bin_size_pixels + 1))
        if world > 1 and a.scaling == "weak":
            what = "%d x " % world + what.replace("; ", " — one such sequence per GPU, each on its own synthetic world; ", 1)
        if world > 1 and a.scaling == "strong":
            what += "the ONE sequence's %d chunks spread over %d GPUs (%d per GPU); " % (a.streams, world, B)
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(abytes[dom]), "avg_launch_ms": round(dom_avg_s * 1e3, 4),
                "algorithmic_bytes": "SURVEY.md 8(d) rows of the kernel x the %d frames one launch processes" % B}
        roof["achieved_hbm_only"] = round(extra["k_frame_hbm_only"] / dom_avg_s / 1e9, 2) if dom == "k_frame" else roof["achieved"]
        roof["frac_hbm_only"] = round(roof["achieved_hbm_only"] / HBM_PEAK_GBS, 5)
        roof["frac_note"] = ("frac: SURVEY.md 8(d) bytes including the aligner's I*M*64 per-iteration re-reads; frac_hbm_only: without them (they are "
                             "register / LDS resident and never reach HBM).  The step is VALU / SALU issue-bound, not HBM-bound: see kernels{}.valu_util")
        if traffic is not None:
            roof["traffic_over_algorithmic"] = round(traffic / abytes[dom], 3)
            roof["traffic_counters"] = "FETCH_SIZE + WRITE_SIZE as reported; wide (16 B/lane) reads are tallied at half their bytes on gfx950, the kernel's loads are of mixed width: FETCH is a lower bound"
        if traffic_note:
            roof["traffic_note"] = traffic_note
        util, util_note = sq_utilisation()
        for k_, u_ in util.items():
            if k_ in kern:
                kern[k_].update(u_)
        if util_note:
            roof["utilisation_note"] = util_note
        if dom in extra:      # what the fused launch does beyond 8(d)'s rows, labelled and kept out of `frac`
            roof["achieved_incl_recovery"] = round((abytes[dom] + extra[dom]) / dom_avg_s / 1e9, 2)
            roof["achieved_incl_recovery_note"] = ("+ recovery descriptors (projected landmarks x (64 + 2 x 512 box taps x 2 B)) and landmark "
                                                   "refinement measurements, which the fused k_frame launch also does and 8(d) does not list")
        return {
            "metric": METRIC, "value": round(unique / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "frames_processed": int(frames), "unique_frames_timed": int(unique), "raw_pairs_per_s": round(frames / elapsed, 1),
            "config": {"workload": what + "chunks as a steady-state pipeline (streams staggered over the %d-step chunk job, restart when a chunk ends)" % J,
                       "mode": "chunks", "streams_per_gpu": B, "chunks_planned": int(a.streams), "chunk_frames": L, "chunk_overlap": overlap, "chunk_overlap_note": "warm-up frames per chunk; SURVEY.md 8e proposed 10 without data, the 48-seed ATE study (profiles/r03_ate_noise_overlaps48.json) finds 2 .. 10 equivalent; the same command at 2 / 4 / 6 / 10: profiles/r03_bench_overlap_sweep.json", "chunk_job_steps": J,
                       "preroll_steps": preroll, "frames_per_step": int(frames // K),
                       "unique_frame_fraction": round(unique / frames, 4),
                       "parallelism": "frame-sharded chunks, %d per GPU x %d GPU (%s scaling)" % (B, world, a.scaling),
                       "mean_keypoints_per_image": round(stats["N"], 1), "mean_points_per_frame": round(stats["P"], 1),
                       "mean_tracked": round(stats["M"], 1), "mean_aligner_iterations": round(stats["I"], 1),
                       "scene_speed_m_per_frame": round(float(self.scene.speed_m), 3), "scene_contrast": round(float(self.scene.contrast), 3), "error_flags": flags},
            "roofline": roof,
            "kernels": kern,
            "chronometers_s": {k: round(v, 4) for k, v in chrono.items()},
            "pose_allgather": comm_note,
            "build": {"source_sha16": buildinfo.source_sha16(), "library_sha16": buildinfo.library_sha16()},
        }

    # ------------------------------------------------------------------------------------------------------------------
    def ate_leg(self):
        """The accuracy half of the metric for the TIMED configuration: the chunk pipeline keeps running for two more chunk jobs so
        that every stream passes through one whole chunk (restart to end), the chunks are chained by sharding.assemble_trajectory
        and compared with the synthetic ground truth; the same for the sequential run (one stream, whole sequence — the exact mode,
        whose trajectory is the CPU port's to rounding).  ATE-RMSE both as the closed-form SE3 fit and as the reference tool
        defines it (executables/trajectory_analyzer.cpp:212-309: start-point shift, 100 robust Gauss-Newton rounds, RMSE)."""
        from vslam_pose_estimation_framework_amd import evaluation as ev
        api, cfg, job = self.api, self.cfg, self.job
        J, B, total = job["J"], job["n_streams"], self.seq_frames
        rec = torch.zeros((2 * J, B, 12), dtype=torch.float64, device=self.dev)
        k0 = self.counter[0]
        for i in range(2 * J):
            self.run_steps(1)
            api.copy_current_poses_device(rec[i].data_ptr())
        api.synchronize()
        rec = rec.cpu().numpy()
        chunks = []
        for s in range(B):
            i0 = (-(k0 + job["phase"][s])) % J            # the step at which stream s starts its chunk over
            chunks.append(rec[i0:i0 + J, s].reshape(J, 3, 4))
        traj = sharding.assemble_trajectory(chunks, job["plan"][:B])     # (the planned chunks behind the end of the sequence are empty and were not run)
        gt = np.array([self.sy.gt_pose(self.scene, k) for k in range(total)])
        seq_api = hip.load()
        cfg.max_history_frames = 512
        seq_api.create(cfg, self.dev_index, 1)
        t0 = time.perf_counter()
        for f0 in range(0, total, 256):
            n = min(256, total - f0)
            Ls = torch.empty((n, 1, cfg.rows, self.stride), dtype=torch.uint8, device=self.dev)
            Rs = torch.empty_like(Ls)
            self.render(self.scene, f0, n, Ls, Rs, 0, 1)
            torch.cuda.synchronize()
            for k in range(n):
                seq_api.process_device(Ls[k].data_ptr(), Rs[k].data_ptr(), self.stride, self.img_bytes)
            seq_api.synchronize()
        seq_s = time.perf_counter() - t0
        seq = seq_api.poses(0, 0, total)
        seq_api.destroy()
        cfg.max_history_frames = J + 2

        def analyzer(est):
            p = np.asarray(est).reshape(-1, 3, 4)[:, :, 3]
            g = gt[:, :, 3]
            p = p - p[0] + g[0]
            T, _ = ev.align_robust_icp(p, g)
            return ev.rmse(p @ T[:3, :3].T + T[:3, 3], g)
        a_c, a_s = ev.ate_rmse(traj, gt), ev.ate_rmse(seq, gt)
        out = {"unit": "m", "frames": total, "path_length_m": round(float(np.sum(np.linalg.norm(np.diff(gt[:, :, 3], axis=0), axis=1))), 1),
               "chunked": round(a_c, 4), "sequential": round(a_s, 4), "chunked_over_sequential": round(a_c / a_s, 4),
               "trajectory_analyzer": {"chunked": round(analyzer(traj), 4), "sequential": round(analyzer(seq), 4)},
               "definition": "ATE-RMSE of the camera positions against the synthetic ground truth after a closed-form SE3 fit; trajectory_analyzer: "
                             "the reference tool's own alignment (trajectory_analyzer.cpp:212-309 as restated in evaluation.py)",
               "chunked_is": "the timed configuration itself: %d chunks of %d frames + %d warm-up frames, chained at the seams" % (B, job["L"], self.args.overlap),
               "sequential_is": "the same images as ONE stream (exact mode: identical to the CPU port to rounding), %.2f s for the whole sequence" % seq_s}
        # metrics that can resolve a seam (ATE is a random walk in the noise): KITTI's relative errors over 100 .. 800 m sub-trajectories
        # and the relative-pose error of the frame-to-frame motions AT the seams, chunked against sequential at the very same frames
        kc, ks = ev.kitti_relative_errors(traj, gt), ev.kitti_relative_errors(seq, gt)
        out["kitti_relative"] = {"chunked": {"t_rel_percent": round(kc["t_rel_percent"], 4), "r_rel_deg_per_m": round(kc["r_rel_deg_per_m"], 6)},
                                 "sequential": {"t_rel_percent": round(ks["t_rel_percent"], 4), "r_rel_deg_per_m": round(ks["r_rel_deg_per_m"], 6)},
                                 "t_rel_chunked_over_sequential": round(kc["t_rel_percent"] / ks["t_rel_percent"], 4),
                                 "r_rel_chunked_over_sequential": round(kc["r_rel_deg_per_m"] / ks["r_rel_deg_per_m"], 4), "segments": kc["segments"],
                                 "definition": "KITTI odometry devkit: mean over sub-trajectories of 100 .. 800 m starting every 10 frames (evaluation.kitti_relative_errors)"}
        out["seam"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in ev.seam_report(traj, seq, gt, job["plan"]).items()}
        key = "B%d_ov%d" % (self.args.streams, self.args.overlap)
        for name in ATE_NOISE_STUDY:      # where single runs sit in the pipeline's own spread under sensor noise (tools/eval_ate_noise.py)
            try:
                full = json.load(open(os.path.join(ROOT, "profiles", name)))["summary"]
                st = full["ate"]
            except (OSError, KeyError, ValueError):
                continue
            if key not in st["chunked"] and name != ATE_NOISE_STUDY[-1]:
                continue
            out["noise_study"] = {"file": "profiles/" + name, "noise_seeds": full.get("noise_seeds"), "sequential_mean": round(st["sequential"]["mean"], 3),
                                  "sequential_std": round(st["sequential"]["std"], 3)}
            if key in st["chunked"]:
                c = st["chunked"][key]
                out["noise_study"].update({"chunked_mean": round(c["mean"], 3), "chunked_std": round(c["std"], 3),
                                           "mean_shift_in_sequential_sigmas": round(c["mean_shift_in_sequential_sigmas"], 3),
                                           "welch_t": round(c["welch_t"], 3), "chunked_runs_inside_sequential_range": c["inside_sequential_range"]})
                for m in ("t_rel", "r_rel"):          # the study's relative errors (r04 onwards)
                    if m in full and key in full[m]["chunked"]:
                        q = full[m]["chunked"][key]
                        out["noise_study"][m] = {"sequential_mean": round(full[m]["sequential"]["mean"], 5), "chunked_mean": round(q["mean"], 5),
                                                 "ratio_of_means": round(q["ratio_of_means"], 4), "welch_t": round(q["welch_t"], 3)}
                if "seam" in full and key in full["seam"]:
                    out["noise_study"]["seam"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in full["seam"][key].items()}
            break
        return out

    # ------------------------------------------------------------------------------------------------------------------
    def run_sequences(self, seq_ids, steps, warmup, api=None, label=None):
        """Exact mode: whole sequences, one per stream, longest first over the ranks; every stream runs start to end on its
        own (results identical to the sequential run), shorter ones switch off when they end.  steps = 0: whole sequences."""
        cfg, world, rank = self.cfg, self.world, self.rank
        lengths = [KITTI_FRAMES[i] for i in seq_ids]
        ranks, load = sharding.plan_sequences(lengths, world)
        mine = ranks[rank]
        B = max(1, max(len(r) for r in ranks))       # same stream count on every rank (idle slots switched off)
        my_len = [lengths[i] for i in mine] + [0] * (B - len(mine))
        total_steps = max(load) if max(len(r) for r in ranks) == 1 else max(lengths)
        K = min(steps, total_steps) if steps > 0 else total_steps
        Wm = min(max(0, warmup), 128, K)
        KB = K
        cfg.max_history_frames = min(K + Wm + 2, 512)
        api = api or hip.load()
        Lbuf = torch.empty((KB, B, cfg.rows, self.stride), dtype=torch.uint8, device=self.dev)
        Rbuf = torch.empty_like(Lbuf)
        scenes = []
        for slot, i in enumerate(mine):
            sc = self.sy.scene_kitti(seed=7 + 13 * seq_ids[i])       # one world per sequence
            sc.speed_m = self.args.speed if self.args.speed > 0 else 0.7 + 0.05 * (seq_ids[i] % 5)
            scenes.append(sc)
            n = min(KB, my_len[slot])
            for f0 in range(0, n, 512):
                self.render(sc, f0, min(512, n - f0), Lbuf[f0:], Rbuf[f0:], slot, B)
        torch.cuda.synchronize()
        api.create(cfg, self.dev_index, B)

        def run(n_steps):
            live = [s for s in range(B) if my_len[s] > 0]
            for s in range(B):
                api.set_stream_active(s, my_len[s] > 0)
            done = 0
            for k in range(n_steps):
                ended = [s for s in live if my_len[s] <= k]
                for s in ended:
                    api.set_stream_active(s, False)
                    live.remove(s)
                if not live:
                    break
                api.process_device(Lbuf[k].data_ptr(), Rbuf[k].data_ptr(), self.stride, self.img_bytes)
                done += len(live)
            return done

        run(Wm)
        api.synchronize()
        api.reset()
        pose_send = torch.zeros((B, K, 12), dtype=torch.float64, device=self.dev)
        sharding.gather_poses(pose_send)
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames = run(K)
        api.copy_poses_device(0, K, pose_send.data_ptr())
        api.synchronize()
        sharding.gather_poses(pose_send)
        torch.cuda.synchronize()
        self.barrier()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        frames = self.sum_over_ranks(frames)
        live = [s for s in range(B) if my_len[s] >= K] or [0]
        stats, flags = frame_stats(api, live)
        out = {"sequences": ["%02d" % i for i in seq_ids], "frames": int(frames), "seconds": round(elapsed, 4),
               "frames_per_s": round(frames / elapsed, 2), "steps": K, "streams_per_gpu": B,
               "rank_loads": load, "ms_per_step": round(elapsed / max(K, 1) * 1e3, 4),
               "mean_keypoints_per_image": round(stats["N"], 1), "mean_tracked": round(stats["M"], 1), "error_flags": flags}
        if label:
            out["label"] = label
        return out, api, (Lbuf, Rbuf)

    def main_sequences(self):
        a = self.args
        seq_ids = [0, 2, 5, 6] if self.world == 4 else list(range(11))
        if a.sequences:
            seq_ids = [int(x) for x in a.sequences.split(",")]
        res, api, _ = self.run_sequences(seq_ids, a.steps, a.warmup, api=self.api)
        if self.rank != 0:
            return None
        K = res["steps"]
        return {
            "metric": METRIC, "value": res["frames_per_s"], "unit": "frames/s", "n_gpus": self.world, "steps": K, "warmup": max(0, a.warmup),
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic", "frames_processed": res["frames"], "unique_frames_timed": res["frames"],
            "config": {"workload": "KITTI-shaped synthetic sequences %s (1241x376, KITTI odometry lengths), configuration_kitti.yaml values, "
                                   "bin %d, whole sequences one per stream (exact mode)" % ("+".join(res["sequences"]), a.bin),
                       "mode": "sequences", "streams_per_gpu": res["streams_per_gpu"], "rank_frame_loads": res["rank_loads"],
                       "parallelism": "sequence-sharded, longest first, %d GPU" % self.world,
                       "mean_keypoints_per_image": res["mean_keypoints_per_image"], "mean_tracked": res["mean_tracked"],
                       "error_flags": res["error_flags"]},
            "roofline": None,
        }

    # ------------------------------------------------------------------------------------------------------------------
    def pcie_leg(self, steps=16):
        """The chunk job again with HOST images (vslam_process_host): every pair crosses PCIe first.  Never `value`."""
        cfg, B, J = self.cfg, self.B, self.J
        n = min(steps, J)
        Lh = torch.empty((n, B, cfg.rows, self.stride), dtype=torch.uint8).pin_memory()
        Rh = torch.empty_like(Lh).pin_memory()
        Lh.copy_(self.Lbuf[:n]); Rh.copy_(self.Rbuf[:n])
        torch.cuda.synchronize()
        api = hip.load()
        api.create(cfg, self.dev_index, B)
        lp, rp = Lh.data_ptr(), Rh.data_ptr()
        slab = B * self.img_bytes

        def run():
            for k in range(n):
                api.check(api.fn("process_host")(api.ctx, C.c_void_p(lp + k * slab), C.c_void_p(rp + k * slab), C.c_int32(self.stride),
                                                 C.c_size_t(self.img_bytes)))
            api.synchronize()
        run()
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        api.destroy()
        pairs = n * B / dt
        L = -(-self.seq_frames // B)
        return {"pairs_per_s": round(pairs, 1), "unique_frames_per_s": round(pairs * L / J, 1), "steps": n,
                "host_to_device_GBs": round(pairs * 2 * self.img_bytes / 1e9, 2),
                "note": "pinned host images, one hipMemcpyAsync per side and step; PCIe-bound, never `value`"}

    # ------------------------------------------------------------------------------------------------------------------
    def shim_leg(self, frames=320, warmup=20):
        """The real drop-in path: shim/proslam_hip_plugin.h's HipStereoFramePointGenerator / HipStereoUVAligner driven
        as PoseTracker3D::compute drives its plug-ins (initialize -> track -> aligner -> prune / recoverPoints -> compute), one
        stream, HOST images (cv::Mat of the Frame), host objects materialised — tests/cpp/bench_shim.cpp, a separate C++ process
        built against the declaration stubs.  Never `value`."""
        cfg = self.cfg
        exe = os.path.join(ROOT, "tests", "cpp", "bench_shim")
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s", "bench_shim"])
        n = frames + warmup
        Ls = torch.empty((n, 1, cfg.rows, self.stride), dtype=torch.uint8, device=self.dev)
        Rs = torch.empty_like(Ls)
        self.render(self.scene, 0, n, Ls, Rs, 0, 1)
        torch.cuda.synchronize()
        both = torch.stack([Ls[:, 0, :, :cfg.cols], Rs[:, 0, :, :cfg.cols]], dim=1).contiguous().cpu().numpy()      # [n][2][rows][cols], dense rows like a cv::Mat
        del Ls, Rs
        import tempfile
        with tempfile.NamedTemporaryFile(prefix="vslam_shim_", suffix=".bin", dir=os.environ.get("TMPDIR", "/tmp"), delete=False) as fh:
            both.tofile(fh)
            path = fh.name
        try:
            cmd = [exe, path, str(cfg.rows), str(cfg.cols), str(cfg.cols), str(n), str(cfg.bin_size_pixels), str(warmup)]
            prof = os.environ.get("VSLAM_SHIM_ROCPROF")        # profiling aid: per-kernel device time of the shim loop (tools/profile_shim.sh)
            if prof:
                cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", prof, "-o", "shim", "--output-format", "csv", "--"] + cmd
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                return {"error": "bench_shim rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-300:])}
            out = json.loads(lines[-1])
            # the same loop with the frame's images in pinned memory (vslam_host_alloc): no staging copy on the way to the device
            p2 = subprocess.run([exe, path, str(cfg.rows), str(cfg.cols), str(cfg.cols), str(n), str(cfg.bin_size_pixels), str(warmup), "1"],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
            l2 = [ln for ln in p2.stdout.splitlines() if ln.startswith("{")]
            if p2.returncode == 0 and l2:
                r2 = json.loads(l2[-1])
                out["pinned_images"] = {k: r2[k] for k in ("ms_per_frame", "frames_per_s", "stage_ms", "fused_host_images_ms_per_frame", "last_frame_identical_to_fused")}
        finally:
            os.unlink(path)
        out["what"] = ("shim/proslam_hip_plugin.h driven like PoseTracker3D::compute (tests/cpp/bench_shim.cpp, stubs of the reference headers), ONE stream, "
                       "host images %dx%d, bin %d, host objects materialised; the fused figure beside it is vslam_process_host on the same images" % (cfg.cols, cfg.rows, cfg.bin_size_pixels))
        return out

    # ------------------------------------------------------------------------------------------------------------------
    def cpu_leg(self):
        """The oracle (a port of the reference path) on bounded samples of the same images, on this host's cores."""
        a, cfg, B, J = self.args, self.cfg, self.B, self.J
        orc, flags = load_oracle(native=True)
        # chunk frames in job order: stream s, slab j holds chunk frame (j + phase_s) % J -> undo the stagger
        n_chunks = max(1, min(B, a.cpu_frames // J))

        def chunk_images(s, count):
            idx = [(f - self.phase[s]) % J for f in range(count)]
            Lh = self.Lbuf[idx, s].cpu().numpy()
            Rh = self.Rbuf[idx, s].cpu().numpy()
            return Lh, Rh
        chk = hip.load()
        chk.create(cfg, self.dev_index, 1)
        orc.create(cfg, 0, 1)
        cpu_t, knn_t, nfr, mism, max_rel = 0.0, 0.0, 0, 0, 0.0
        frame_ms = []      # per-frame process() time like SLAMAssembly::printReport (slam_assembly.cpp:644-668): mean / median / min / max
        samples = []
        for sidx in range(n_chunks):
            orc.reset(); chk.reset()
            Lh, Rh = chunk_images(sidx, J)
            samples.append((Lh, Rh))
            for k in range(J):
                t = time.perf_counter()
                orc.process_host(Lh[k], Rh[k])
                frame_ms.append((time.perf_counter() - t) * 1e3)
                cpu_t += frame_ms[-1] * 1e-3
                t = time.perf_counter()
                orc.fn("dead_knn_match")(orc.ctx, C.c_int(0), C.c_int(1), C.c_int(1))   # use_matches: knnMatch(k=2) on floats (L2) + findHomography(LMEDS)
                knn_t += time.perf_counter() - t
                chk.process_host(Lh[k], Rh[k])
                fo, fg = orc.frame_info(0), chk.frame_info(0)
                for name in ("n_keypoints_left", "n_keypoints_right", "n_tracked", "n_inliers", "n_points", "status",
                             "n_recovered", "n_new_stereo", "window_pixels"):
                    if getattr(fo, name) != getattr(fg, name):
                        mism += 1
                To, Tg = np.array(fo.camera_left_to_world), np.array(fg.camera_left_to_world)
                max_rel = max(max_rel, float(np.linalg.norm(Tg - To) / np.linalg.norm(To)))
                nfr += 1
        sec = (C.c_double * 8)()
        orc.fn("get_timers")(orc.ctx, sec)
        names = ["keypoint_detection", "descriptor_extraction", "point_triangulation", "tracking", "track_creation",
                 "pose_optimization", "landmark_optimization", "point_recovery"]
        modules = {names[i]: round(sec[i] / nfr * 1e3, 4) for i in range(8)}
        chk.destroy(); orc.destroy()
        cpu_fps = nfr / cpu_t
        # all host cores: one independent chunk per thread (ctypes releases the GIL), each with its own oracle context
        nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncpu = a.cpu_threads if a.cpu_threads > 0 else min(nproc, 16)    # a one-GPU box owns a 16-CPU share of its host
        per = min(J, 24)
        ctxs = []
        from vslam_pose_estimation_framework_amd.capi import CApi
        for t in range(ncpu):
            o = CApi(orc.lib._name, "orc_")
            o.create(cfg, 0, 1)
            ctxs.append(o)

        def work(t):
            Lh, Rh = samples[t % len(samples)]
            for k in range(per):
                ctxs[t].process_host(Lh[k], Rh[k])
        th = [threading.Thread(target=work, args=(t,)) for t in range(ncpu)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        all_dt = time.perf_counter() - t0
        for o in ctxs:
            o.destroy()
        return {"value": round(cpu_fps, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": "%d chunks x %d frames of the same synthetic sequence (oracle/vslam_oracle.cpp, g++ %s, 1 thread, "
                          "process() time only)" % (n_chunks, J, flags),
                "with_dead_knn_match": {"value": round(nfr / (cpu_t + knn_t), 2), "unit": "frames/s",
                                        "note": "plus the reference's dead block of compute() (stereo_framepoint_generator.cpp:168-273, "
                                                "use_matches: true, results discarded): knnMatch(k=2) on CV_32F descriptors and "
                                                "findHomography(LMEDS, 0.99, 1000) of the first matches, both restated in the oracle"},
                "all_cores": {"value": round(ncpu * per / all_dt, 2), "unit": "frames/s", "cores": ncpu, "nproc_visible": nproc,
                              "sample": "%d threads x %d frames, one independent chunk per thread" % (ncpu, per)},
                "module_ms_per_frame": modules,
                "frame_ms": {"mean": round(float(np.mean(frame_ms)), 3), "median": round(float(np.median(frame_ms)), 3),
                             "min": round(float(np.min(frame_ms)), 3), "max": round(float(np.max(frame_ms)), 3)},
                "parity_on_sample": {"frames": nfr, "int_field_mismatches": mism, "max_pose_rel_frobenius": max_rel}}


def main():
    args = ARGS if ARGS is not None else parse_args()
    b = Bench(args)
    if args.only_shim:
        print(json.dumps(b.shim_leg()), flush=True)
        return
    if args.mode == "sequences":
        out = b.main_sequences()
    else:
        out = b.run_chunks()
        if b.rank == 0 and b.world == 1:
            if not args.no_ate and not b.euroc:
                out["ate"] = b.ate_leg()
            if not args.no_pcie:
                out["pcie_inclusive"] = b.pcie_leg()
            if not args.no_shim and not b.euroc:
                out["shim_path"] = b.shim_leg()
            if not args.no_cpu:
                out["cpu_baseline"] = b.cpu_leg()
                out["speedup_vs_cpu_port"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            b.Lbuf = b.Rbuf = None
            b.api.destroy()
            torch.cuda.empty_cache()
            if not args.no_exact:
                # exact mode (whole sequence per stream, results identical to the sequential reference run)
                one, api1, _ = b.run_sequences([0], args.exact_frames, 100, label="KITTI-00-shaped alone: ONE stream, the literal drop-in")
                api1.destroy()
                torch.cuda.empty_cache()
                allseq, api2, _ = b.run_sequences(list(range(11)), args.exact_frames, 100, label="KITTI 00-10 lengths as 11 streams")
                api2.destroy()
                out["exact_mode"] = {"single_sequence": one, "eleven_sequences": allseq}
                if "cpu_baseline" in out:
                    out["exact_mode"]["single_sequence_speedup_vs_cpu_port"] = round(one["frames_per_s"] / out["cpu_baseline"]["value"], 1)
    if b.rank == 0:
        print(json.dumps(out), flush=True)
    if b.world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
