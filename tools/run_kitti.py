#!/usr/bin/env python3
"""Run the MI355X front end on a KITTI odometry folder and write the trajectory (the I/O formats
either side of the hot path; executables/test_stereo_frontend.cpp:106-111,256-312 of the reference is the template).

    python tools/run_kitti.py <sequence dir with image_0/ image_1/ calib.txt [times.txt]> [--out traj.txt]
                              [--format kitti|tum] [--gt poses.txt] [--max-frames N] [--config kitti|euroc]
                              [--chunks B [--overlap 6]]   frame-sharded mode: B chunks side by side (approximate at the seams)
    python tools/run_kitti.py <EuRoC dir with mav0/cam0 mav0/cam1> --format tum --out traj.txt   (ground truth found in mav0/)

The sequence runs in exact mode (one stream, whole sequence, bit-for-bit the reference port's arithmetic); images are
uploaded frame by frame through vslam_process_host.  With --gt (KITTI 3x4 rows) the ATE-RMSE after rigid alignment is
printed, as trajectory_analyzer.cpp:212-284 computes it."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from vslam_pose_estimation_framework_amd import evaluation, hip, io_formats  # noqa: E402


def run_chunked(api, cfg, seq, n, n_chunks, overlap, device, log):
    """Frame-sharded mode (bench.py's headline mode) on a recorded sequence: `n_chunks` contiguous chunks, each
    started `overlap` frames early, run side by side as the streams of one context; the chunk trajectories are chained at the seams
    (sharding.assemble_trajectory).  Approximate at the seams — DESIGN.md section 9 has the accuracy study."""
    from vslam_pose_estimation_framework_amd import sharding
    plan, _ = sharding.plan_chunks(n, n_chunks, overlap)
    steps = max(e - s for (s, f, e) in plan)
    cfg.max_history_frames = steps + 2
    api.create(cfg, device, len(plan))
    rows, cols = int(cfg.rows), int(cfg.cols)
    Lb = np.zeros((len(plan), rows, cols), np.uint8)
    Rb = np.zeros_like(Lb)
    live = [True] * len(plan)
    for k in range(steps):
        for c, (st, fi, en) in enumerate(plan):
            if st + k < en:
                Lb[c], Rb[c] = seq.pair(st + k)
            elif live[c]:
                api.set_stream_active(c, False)
                live[c] = False
        api.process_host(Lb, Rb)
        if k % 20 == 19 or k == steps - 1:
            log("step %5d of %d (%d chunks side by side)" % (k + 1, steps, sum(live)))
    api.synchronize()
    flags = 0
    for c in range(len(plan)):
        flags |= api.frame_info(c).error_flags
    chunks = [api.poses(c, 0, en - st) for c, (st, fi, en) in enumerate(plan)]
    return np.asarray(sharding.assemble_trajectory(chunks, plan)).reshape(n, 12), flags


def run(seq_dir, out_path=None, fmt="kitti", gt_path=None, max_frames=0, which="kitti", device=0, log=print, layout="kitti", asl_gt=None,
        chunks=0, overlap=6):
    euroc = layout == "euroc" or os.path.isdir(os.path.join(seq_dir, "mav0"))
    seq = io_formats.EurocSequence(seq_dir) if euroc else io_formats.KittiSequence(seq_dir)
    n = len(seq) if max_frames <= 0 else min(len(seq), max_frames)
    if n == 0:
        raise SystemExit("no images under %s" % seq_dir)
    left, right = seq.pair(0)
    api = hip.load()
    cfg = api.default_config("euroc" if euroc and which == "kitti" else which)
    if euroc:
        cal = seq.calibration()       # a rectified export may carry its P0 / P1; otherwise the EuRoC values of the default config
        if cal is not None:
            io_formats.apply_calib(cfg, cal[0], cal[1], left.shape[0], left.shape[1])
        else:
            cfg.rows, cfg.cols = int(left.shape[0]), int(left.shape[1])
    else:
        io_formats.apply_calib(cfg, seq.K, seq.baseline, left.shape[0], left.shape[1])
    t0 = time.perf_counter()
    flags = 0
    if chunks > 1:
        poses, flags = run_chunked(api, cfg, seq, n, chunks, overlap, device, log)
    else:
        cfg.max_history_frames = 512
        api.create(cfg, device, 1)
        for k in range(n):
            if k:
                left, right = seq.pair(k)
            api.process_host(left, right)
            if k % 100 == 99 or k == n - 1:
                fi = api.frame_info(0)
                flags |= fi.error_flags
                log("frame %6d  status %s  points %5d  tracked %5d  inliers %5d" % (
                    k, "tracking" if fi.status == 1 else "localizing", fi.n_points, fi.n_tracked, fi.n_inliers))
        poses = api.poses(0, 0, n)
    dt = time.perf_counter() - t0
    api.destroy()
    log("%d frames in %.2f s (%.1f frames/s incl. PNG decode and upload), error flags %d" % (n, dt, n / dt, flags))
    if out_path:
        if fmt == "tum":
            io_formats.write_trajectory_tum(out_path, poses, seq.times[:n])
        else:
            io_formats.write_trajectory_kitti(out_path, poses)
        log("trajectory (%s) -> %s" % (fmt, out_path))
    result = {"frames": n, "seconds": dt, "error_flags": flags, "poses": poses}
    if gt_path:
        gt = io_formats.read_trajectory_kitti(gt_path)[:n]
        result["ate_rmse_aligned"] = evaluation.ate_rmse(poses[:len(gt)], gt)
        log("ATE-RMSE after rigid alignment: %.4f m over %d frames" % (result["ate_rmse_aligned"], len(gt)))
    asl = asl_gt or (seq.ground_truth_path if euroc else None)
    if asl and out_path and fmt == "tum":
        r = evaluation.trajectory_analyzer(out_path, asl)     # executables/trajectory_analyzer.cpp: -tum <out> -asl <ground truth>
        result["trajectory_analyzer"] = {k: r[k] for k in ("correspondences", "raw_rmse", "optimal_rmse")}
        log("trajectory_analyzer: %d interpolated positions, raw RMSE %.4f m, optimal RMSE %.4f m" % (r["correspondences"], r["raw_rmse"], r["optimal_rmse"]))
    return result


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("sequence")
    ap.add_argument("--out", default=None)
    ap.add_argument("--format", choices=("kitti", "tum"), default="kitti")
    ap.add_argument("--gt", default=None)
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--config", choices=("kitti", "euroc"), default="kitti")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--chunks", type=int, default=0, help="frame-sharded mode: cut the sequence into this many chunks and run them side by side (0 / 1: exact mode, one stream)")
    ap.add_argument("--overlap", type=int, default=6, help="warm-up frames per chunk in frame-sharded mode")
    ap.add_argument("--layout", choices=("kitti", "euroc"), default="kitti", help="folder layout (a folder with mav0/ is taken as EuRoC / ASL)")
    ap.add_argument("--asl-gt", default=None, help="ASL ground-truth csv for the trajectory_analyzer step (needs --format tum --out)")
    a = ap.parse_args()
    run(a.sequence, a.out, a.format, a.gt, a.max_frames, a.config, a.device, layout=a.layout, asl_gt=a.asl_gt, chunks=a.chunks, overlap=a.overlap)


if __name__ == "__main__":
    main()
