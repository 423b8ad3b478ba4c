"""Trajectory error.

* `trajectory_analyzer(...)` restates the reference's tool (executables/trajectory_analyzer.cpp:59-309): TUM trajectory against
  an ASL / EuRoC ground-truth csv — nearest ground-truth sample within 1 s, linear interpolation to the measurement's time
  stamp, start-point shift, raw RMSE, then its iterative robust alignment (100 Gauss-Newton rounds on T in SE3, kernel
  1 m^2, Jacobian [I | -2 skew(T p)], LDL^T solve, v2t update, re-orthonormalisation) and the RMSE after it.
* `ate_rmse(...)` is the closed-form companion (Kabsch / Umeyama without scale) used by the benchmarks on trajectories that
  are already sample-aligned (synthetic ground truth, KITTI pose files)."""
import numpy as np


def to44(T):
    M = np.eye(4)
    M[:3, :] = np.asarray(T).reshape(3, 4)
    return M


def inv34(T):
    T = np.asarray(T).reshape(3, 4)
    o = np.zeros((3, 4))
    o[:, :3] = T[:, :3].T
    o[:, 3] = -T[:, :3].T @ T[:, 3]
    return o


def mul34(A, B):
    A = np.asarray(A).reshape(3, 4)
    B = np.asarray(B).reshape(3, 4)
    o = np.zeros((3, 4))
    o[:, :3] = A[:, :3] @ B[:, :3]
    o[:, 3] = A[:, :3] @ B[:, 3] + A[:, 3]
    return o


def align_se3(est_xyz, gt_xyz):
    """R, t minimising sum |R est + t - gt|^2."""
    mu_e, mu_g = est_xyz.mean(0), gt_xyz.mean(0)
    H = (est_xyz - mu_e).T @ (gt_xyz - mu_g)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, mu_g - R @ mu_e


def ate_rmse(est_poses, gt_poses, align=True):
    est = np.asarray(est_poses).reshape(-1, 3, 4)[:, :, 3]
    gt = np.asarray(gt_poses).reshape(-1, 3, 4)[:, :, 3]
    if align and len(est) >= 3:
        R, t = align_se3(est, gt)
        est = est @ R.T + t
    return float(np.sqrt(((est - gt) ** 2).sum(1).mean()))


def write_trajectory_kitti(path, poses):
    """WorldMap::writeTrajectoryKITTI (src/types/world_map.cpp:184-214): one row of 12 numbers per frame."""
    with open(path, "w") as f:
        for T in np.asarray(poses).reshape(-1, 12):
            f.write(" ".join("%.9f" % v for v in T) + "\n")


# ---- executables/trajectory_analyzer.cpp restated -----------------------------------------------------------------------
def read_trajectory_tum(path, skip=0):
    """(timestamps, positions) of a TUM file `t x y z qx qy qz qw` (:67-107); `skip` poses are cut from both ends (:95-107)."""
    ts, xyz = [], []
    skipped = 0
    with open(path) as f:
        for line in f:
            v = line.split()
            if len(v) < 8:
                raise RuntimeError("unable to parse pose lines")
            if skipped >= skip:
                ts.append(float(v[0])); xyz.append([float(v[1]), float(v[2]), float(v[3])])
            else:
                skipped += 1
    if skip >= len(ts):
        raise RuntimeError("insufficient number of measurements for number_of_poses_to_skip: %d" % skip)
    n = len(ts) - skip
    return np.array(ts[:n]), np.array(xyz[:n]).reshape(-1, 3)


def read_ground_truth_asl(path):
    """(timestamps in seconds, positions) of an ASL / EuRoC ground-truth csv: `#` comment lines, then
    `timestamp_ns, p_x, p_y, p_z, ...` (:117-146; only the position columns are read)."""
    ts, xyz = [], []
    with open(path) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            v = line.split(",")
            ts.append(int(v[0]) / 1e9)
            xyz.append([float(v[1]), float(v[2]), float(v[3])])
    return np.array(ts), np.array(xyz).reshape(-1, 3)


def _skew(p):
    return np.array([[0.0, -p[2], p[1]], [p[2], 0.0, -p[0]], [-p[1], p[0], 0.0]])


def _v2t(v):
    """srrg_core::v2t: translation v[0:3], rotation from the vector part of a unit quaternion v[3:6]."""
    q = np.array(v[3:6], float)
    n2 = float(q @ q)
    if n2 < 1:
        w = np.sqrt(1 - n2)
    else:
        q = q / np.sqrt(n2); w = 0.0
    x, y, z = q
    T = np.eye(4)
    T[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                 [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                 [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    T[:3, 3] = v[0:3]
    return T


def interpolate_correspondences(t_slam, p_slam, t_gt, p_gt):
    """:152-200.  For every measurement: the ground-truth sample closest in time (strictly closer than 1 s, first one wins);
    measurements whose closest sample is index 0 are skipped ("until we arrive at the ground truth timestamp"); the ground
    truth is interpolated linearly to the measurement's time stamp between that sample and its neighbour on the measurement's
    side; the interpolated position at the FIRST measurement (if it is not skipped) shifts every measurement.  A closest
    sample that is the last one and still earlier than the measurement has no right neighbour (an out-of-bounds read
    upstream): the measurement is skipped.  Returns (measured positions incl. shift, interpolated ground truth)."""
    meas, ref = [], []
    shift = np.zeros(3)
    for i in range(len(t_slam)):
        d = np.abs(t_slam[i] - t_gt)
        best, best_d = 0, 1.0
        for j in range(len(t_gt)):
            if d[j] < best_d:
                best_d, best = d[j], j
        if best == 0:
            continue
        if t_gt[best] < t_slam[i]:
            if best + 1 >= len(t_gt):
                continue
            a, b = best, best + 1
        else:
            a, b = best - 1, best
        g = p_gt[a] + (t_slam[i] - t_gt[a]) / (t_gt[b] - t_gt[a]) * (p_gt[b] - p_gt[a])
        if i == 0:
            shift = g.copy()
        meas.append(p_slam[i] + shift)
        ref.append(g)
    return np.array(meas).reshape(-1, 3), np.array(ref).reshape(-1, 3)


def rmse(a, b):
    """getAbsoluteTranslationRootMeanSquaredError (:288-309)."""
    return float(np.sqrt(((np.asarray(a) - np.asarray(b)) ** 2).sum(1).mean()))


def align_robust_icp(meas, ref, iterations=100, kernel=1.0):
    """:212-276: T (4x4) that moves the measured positions onto the reference, by the tool's own iteration."""
    T = np.eye(4)
    log = []
    meas = np.asarray(meas, float).reshape(-1, 3)
    ref = np.asarray(ref, float).reshape(-1, 3)
    for _ in range(iterations):
        # all correspondences of a round at once (the tool's loop :222-262, one numpy expression per statement)
        S = meas @ T[:3, :3].T + T[:3, 3]
        E = S - ref
        e2 = (E * E).sum(1)
        w = np.where(e2 > kernel, kernel / np.where(e2 > 0, e2, 1.0), 1.0)
        inliers = int((e2 <= kernel).sum())
        total = float(e2.sum())
        J = np.zeros((len(S), 3, 6))
        J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1.0
        J[:, 0, 4] = 2 * S[:, 2]; J[:, 0, 5] = -2 * S[:, 1]
        J[:, 1, 3] = -2 * S[:, 2]; J[:, 1, 5] = 2 * S[:, 0]
        J[:, 2, 3] = 2 * S[:, 1]; J[:, 2, 4] = -2 * S[:, 0]
        H = np.einsum("n,nij,nik->jk", w, J, J)
        b = np.einsum("n,nij,ni->j", w, J, E)
        # Eigen's H.ldlt().solve(-b); the least-squares solve covers the rank-deficient start (all points coincide)
        dx = np.linalg.lstsq(H, -b, rcond=None)[0]
        T = _v2t(dx) @ T
        R = T[:3, :3]
        RtR = R.T @ R
        RtR[np.diag_indices(3)] -= 1
        T[:3, :3] = R - 0.5 * R @ RtR
        log.append((total, inliers))
    return T, log


def trajectory_analyzer(tum_path, asl_path, skip=0):
    """The tool end to end: {"correspondences", "raw_rmse", "optimal_rmse", "transform", "iterations"}."""
    t_s, p_s = read_trajectory_tum(tum_path, skip)
    t_g, p_g = read_ground_truth_asl(asl_path)
    meas, ref = interpolate_correspondences(t_s, p_s, t_g, p_g)
    if len(meas) == 0:
        raise RuntimeError("no interpolated positions: the trajectories do not overlap in time")
    raw = rmse(meas, ref)
    T, log = align_robust_icp(meas, ref)
    moved = meas @ T[:3, :3].T + T[:3, 3]
    return {"correspondences": len(meas), "raw_rmse": raw, "optimal_rmse": rmse(moved, ref), "transform": T, "iterations": log}


# ---- relative errors: metrics that resolve a seam ------------------------------------------------------------------------------
# ATE of open-loop odometry is a random walk in the measurement noise (29 % run-to-run spread of the sequential
# pipeline alone), so it cannot carry a 1 % criterion.  The KITTI odometry benchmark's own metric — translation / rotation error of
# sub-trajectories, relative to their length — and the relative-pose error of single frame-to-frame motions do not accumulate.
KITTI_LENGTHS_M = (100.0, 200.0, 300.0, 400.0, 500.0, 600.0, 700.0, 800.0)


def _rotation_angle(R):
    return float(np.arccos(max(-1.0, min(1.0, 0.5 * (np.trace(R) - 1.0)))))


def relative_pose(poses, i, j):
    """Motion from frame i to frame j in frame i's coordinates: inv(T_i) T_j (camera-to-world poses)."""
    return mul34(inv34(poses[i]), poses[j])


def kitti_relative_errors(est_poses, gt_poses, lengths=KITTI_LENGTHS_M, step=10):
    """The KITTI odometry devkit's evaluation (evaluate_odometry.cpp calcSequenceErrors, restated from its published definition): for
    every `step`-th first frame and every sub-trajectory length, the last frame is the first one whose ground-truth path length
    from the first frame reaches the length; error = inv(gt motion) * estimated motion; t_err = |translation| / length,
    r_err = rotation angle / length.  Returns {"t_rel_percent", "r_rel_deg_per_m", "segments", "per_length": {len: (t %, r deg/m, n)}}."""
    est = np.asarray(est_poses, float).reshape(-1, 3, 4)
    gt = np.asarray(gt_poses, float).reshape(-1, 3, 4)
    n = min(len(est), len(gt))
    dist = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(gt[:n, :, 3], axis=0), axis=1))])
    per = {}
    t_all, r_all = [], []
    for L in lengths:
        t_l, r_l = [], []
        for first in range(0, n, step):
            last = int(np.searchsorted(dist, dist[first] + L, side="left"))
            if last >= n:
                break
            err = mul34(inv34(relative_pose(gt, first, last)), relative_pose(est, first, last))
            t_l.append(np.linalg.norm(err[:, 3]) / L)
            r_l.append(_rotation_angle(err[:, :3]) / L)
        if t_l:
            per[float(L)] = (100.0 * float(np.mean(t_l)), float(np.degrees(np.mean(r_l))), len(t_l))
            t_all += t_l
            r_all += r_l
    if not t_all:
        return {"t_rel_percent": None, "r_rel_deg_per_m": None, "segments": 0, "per_length": {}}
    return {"t_rel_percent": 100.0 * float(np.mean(t_all)), "r_rel_deg_per_m": float(np.degrees(np.mean(r_all))), "segments": len(t_all), "per_length": per}


def relative_pose_errors(est_poses, ref_poses, frames, delta=1):
    """Relative-pose error of the motions frame-delta -> frame, for the given frames: (translation errors in m, rotation errors in
    rad) of inv(ref motion) * est motion.  `ref` is the ground truth, or another run of the same images."""
    est = np.asarray(est_poses, float).reshape(-1, 3, 4)
    ref = np.asarray(ref_poses, float).reshape(-1, 3, 4)
    te, re = [], []
    for f in frames:
        if f - delta < 0 or f >= min(len(est), len(ref)):
            continue
        err = mul34(inv34(relative_pose(ref, f - delta, f)), relative_pose(est, f - delta, f))
        te.append(np.linalg.norm(err[:, 3]))
        re.append(_rotation_angle(err[:, :3]))
    return np.array(te), np.array(re)


def seam_frames(plan):
    """First own frame of every chunk but the first (sharding.plan_chunks): the motions frame-1 -> frame cross a seam."""
    return [first for c, (start, first, end) in enumerate(plan) if c > 0 and end > first]


def seam_report(chunked, sequential, gt, plan):
    """Does a seam cost accuracy?  Relative-pose error against the ground truth of the frame-to-frame motions AT the seams, for the
    chunked run and for the sequential run at the very same frames (rms), their ratio, and the chunked run's seam motions against
    the sequential run's own."""
    seams = seam_frames(plan)
    tc, rc = relative_pose_errors(chunked, gt, seams)
    ts, rs = relative_pose_errors(sequential, gt, seams)
    td, rd = relative_pose_errors(chunked, sequential, seams)
    rms = lambda v: float(np.sqrt(np.mean(np.square(v)))) if len(v) else None      # noqa: E731
    return {"seams": len(seams), "chunked_rpe_trans_rms_m": rms(tc), "sequential_rpe_trans_rms_m": rms(ts),
            "chunked_rpe_rot_rms_deg": None if not len(rc) else float(np.degrees(rms(rc))), "sequential_rpe_rot_rms_deg": None if not len(rs) else float(np.degrees(rms(rs))),
            "rpe_trans_ratio": None if not len(tc) or rms(ts) == 0 else rms(tc) / rms(ts), "rpe_rot_ratio": None if not len(rc) or rms(rs) == 0 else rms(rc) / rms(rs),
            "chunked_vs_sequential_trans_rms_m": rms(td), "chunked_vs_sequential_rot_rms_deg": None if not len(rd) else float(np.degrees(rms(rd)))}
