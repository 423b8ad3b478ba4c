#!/usr/bin/env python3
"""Run the MI355X RGB-D front end (vslam_rgbd_*: PoseTracker3D with DepthFramePointGenerator + UVDAligner) on a TUM RGB-D folder — the layout
the TUM benchmark and ICL-NUIM ship (`rgb.txt`, `depth.txt`, `rgb/`, `depth/`, optional `groundtruth.txt`) — and write the trajectory in the
reference's TUM format (WorldMap::writeTrajectoryTUM, world_map.cpp:222-258).

    python tools/run_rgbd.py <folder> [--config icl|tum|xtion] [--intrinsics freiburg1|freiburg2|freiburg3|icl|fx,fy,cx,cy]
                             [--depth-unit 0.0002] [--out traj.txt] [--max-frames N] [--descriptor ORB|BRIEF] [--detector FAST|ORB]

--config picks the values of configurations/configuration_{icl,tum,xtion}.yaml the path reads (table below: detector grid and thresholds,
tracking windows and descriptor distances, depth limits, bin size, triangulation of points without depth, landmark / aligner settings); the
camera comes from --intrinsics (the depth image is registered to the colour image in these data sets: one camera matrix, identity offset).
Colour images are converted like cv::imread(IMREAD_GRAYSCALE).  With a groundtruth.txt in the folder (or --gt) and --out, the reference's
trajectory_analyzer (executables/trajectory_analyzer.cpp, restated in evaluation.py) reports the RMSE after its alignment."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from vslam_pose_estimation_framework_amd import evaluation, hip, io_formats  # noqa: E402
from vslam_pose_estimation_framework_amd.capi import DepthParams, RgbdTracker  # noqa: E402

# configurations/configuration_{icl,tum,xtion}.yaml: base_framepoint_generation / depth_framepoint_generation / tracking / landmark values
YAML = {
    "icl": dict(thr=(5, 100), max_change=1.0, grid=(2, 2), win=(5, 25), desc=(25, 50), depth=(2.5, 10.0, 0.001), bin=25, tri=0,
                lm_err=0.5, min_lm=10, tunnel=0.5, good=0.25, delta_move=(0.0, 0.0), kernel=5),
    "tum": dict(thr=(10, 100), max_change=0.5, grid=(1, 1), win=(10, 50), desc=(40, 40), depth=(2.5, 10.0, 0.1), bin=15, tri=1,
                lm_err=1.0, min_lm=5, tunnel=0.75, good=0.25, delta_move=(0.001, 0.01), kernel=10),
    "xtion": dict(thr=(10, 100), max_change=0.5, grid=(1, 1), win=(5, 10), desc=(25, 50), depth=(2.5, 8.0, 0.1), bin=10, tri=1,
                  lm_err=4.0, min_lm=25, tunnel=0.75, good=0.5, delta_move=(0.001, 0.01), kernel=10),
}


def configure(api, which, rows, cols, K, depth_unit, descriptor=1, detector=0, depth_scale=1.0):
    """vslam_config + vslam_depth_params with `which` configuration's values; depth_scale multiplies the three metric depth limits (the
    synthetic street scenes of the tests are larger than a room)."""
    y = YAML[which]
    cfg = api.default_config("kitti")
    cfg.rows, cfg.cols = int(rows), int(cols)
    for i in range(9):
        cfg.K[i] = float(np.asarray(K).reshape(9)[i])
    cfg.baseline_h[0] = -float(K[0][0]) * 0.1        # unused in this mode (one camera); vslam_create wants a valid stereo baseline
    cfg.det_rows, cfg.det_cols = y["grid"]
    cfg.detector_threshold_minimum, cfg.detector_threshold_maximum = y["thr"]
    cfg.detector_threshold_maximum_change = y["max_change"]; cfg.target_number_of_keypoints_tolerance = 0.1
    cfg.minimum_projection_tracking_distance_pixels, cfg.maximum_projection_tracking_distance_pixels = y["win"]
    cfg.minimum_descriptor_distance_tracking, cfg.maximum_descriptor_distance_tracking = y["desc"]
    cfg.maximum_reliable_depth_meters = y["depth"][0] * depth_scale; cfg.maximum_depth_meters = y["depth"][1] * depth_scale
    cfg.minimum_depth_meters = y["depth"][2]
    cfg.enable_keypoint_binning = 1; cfg.bin_size_pixels = y["bin"]
    cfg.minimum_track_length_for_landmark_creation = 2; cfg.minimum_number_of_landmarks_to_track = y["min_lm"]
    cfg.tunnel_vision_ratio = y["tunnel"]; cfg.good_tracking_ratio = y["good"]
    cfg.minimum_delta_angular_for_movement, cfg.minimum_delta_translational_for_movement = y["delta_move"]
    cfg.aligner_error_delta_for_convergence = 1e-5; cfg.aligner_maximum_error_kernel = y["kernel"]; cfg.aligner_damping = 0
    cfg.aligner_maximum_number_of_iterations = 1000; cfg.aligner_minimum_number_of_inliers = 0
    cfg.landmark_maximum_error_squared_meters = y["lm_err"]
    cfg.enable_landmark_recovery = 1
    cfg.descriptor_type = descriptor
    K = np.asarray(K, np.float64).reshape(3, 3)
    p = DepthParams.make(rows, cols, K, np.linalg.inv(K), np.linalg.inv(K), np.eye(4)[:3], depth_unit, y["depth"][2], y["depth"][1] * depth_scale,
                         y["tri"], 1, y["bin"], descriptor, detector)
    return cfg, p


def run(folder, which="tum", intrinsics="freiburg1", depth_unit=io_formats.TUM_DEPTH_UNIT_M, out_path=None, max_frames=0, descriptor=1, detector=0,
        gt_path=None, device=0, depth_scale=1.0, log=print):
    seq = io_formats.TumRgbdSequence(folder)
    n = len(seq) if max_frames <= 0 else min(len(seq), max_frames)
    if n == 0:
        raise SystemExit("no associated rgb / depth pairs under %s" % folder)
    fx, fy, cx, cy = io_formats.TUM_INTRINSICS[intrinsics] if intrinsics in io_formats.TUM_INTRINSICS else [float(v) for v in intrinsics.split(",")]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    gray, depth = seq.frame(0)
    api = hip.load()
    cfg, p = configure(api, which, gray.shape[0], gray.shape[1], K, depth_unit, descriptor, detector, depth_scale)
    tr = RgbdTracker(api, cfg, p, device)
    poses, flags = [], 0
    t0 = time.perf_counter()
    try:
        for k in range(n):
            if k:
                gray, depth = seq.frame(k)
            fi, n_temp = tr.process(gray, depth)
            poses.append(np.array(fi.camera_left_to_world))
            flags |= fi.error_flags
            if k % 100 == 99 or k == n - 1:
                log("frame %6d  status %s  points %5d (+%d temporary)  tracked %5d  inliers %5d  landmarks %5d" % (
                    k, "tracking" if fi.status == 1 else "localizing", fi.n_points, n_temp, fi.n_tracked, fi.n_inliers, fi.n_active_landmarks))
    finally:
        tr.destroy()
    dt = time.perf_counter() - t0
    poses = np.array(poses).reshape(-1, 3, 4)
    log("%d frames in %.2f s (%.1f frames/s incl. PNG decode and upload), error flags %d" % (n, dt, n / dt, flags))
    result = {"frames": n, "seconds": dt, "error_flags": flags, "poses": poses, "times": seq.times[:n]}
    if out_path:
        io_formats.write_trajectory_tum(out_path, poses, seq.times[:n])
        log("trajectory (tum) -> %s" % out_path)
        gt = gt_path or seq.ground_truth_path
        if gt:
            # executables/trajectory_analyzer.cpp's pipeline (time-stamp interpolation :152-205, start-point shift + 100 robust rounds :212-284)
            # with the ground truth read from the benchmark's own `timestamp tx ty tz qx qy qz qw` file instead of an ASL csv
            t_s, p_s = evaluation.read_trajectory_tum(out_path)
            rows = io_formats.read_tum_list(gt)
            t_g = np.array([t for t, _ in rows]); p_g = np.array([[float(v) for v in a[:3]] for _, a in rows]).reshape(-1, 3)
            meas, ref = evaluation.interpolate_correspondences(t_s, p_s, t_g, p_g)
            if len(meas):
                T, _ = evaluation.align_robust_icp(meas, ref)
                moved = meas @ T[:3, :3].T + T[:3, 3]
                result["trajectory_analyzer"] = {"correspondences": len(meas), "raw_rmse": evaluation.rmse(meas, ref), "optimal_rmse": evaluation.rmse(moved, ref)}
                log("trajectory_analyzer: %d interpolated positions, raw RMSE %.4f m, optimal RMSE %.4f m" % (
                    len(meas), result["trajectory_analyzer"]["raw_rmse"], result["trajectory_analyzer"]["optimal_rmse"]))
    return result


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("folder")
    ap.add_argument("--config", choices=sorted(YAML), default="tum")
    ap.add_argument("--intrinsics", default="freiburg1")
    ap.add_argument("--depth-unit", type=float, default=io_formats.TUM_DEPTH_UNIT_M, help="metres per depth count (TUM / ICL: 1/5000)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--gt", default=None)
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--descriptor", choices=("ORB", "BRIEF"), default="ORB", help='the configurations say "ORB-256": cv::ORB::create() as extractor')
    ap.add_argument("--detector", choices=("FAST", "ORB"), default="FAST")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    run(a.folder, a.config, a.intrinsics, a.depth_unit, a.out, a.max_frames, 1 if a.descriptor == "ORB" else 0, 1 if a.detector == "ORB" else 0, a.gt, a.device)


if __name__ == "__main__":
    main()
