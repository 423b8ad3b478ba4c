"""Trajectory error: ATE-RMSE of the translation after a rigid (SE3, no scale) alignment, and raw.

Follows what the reference's trajectory_analyzer prints (executables/trajectory_analyzer.cpp:207,284-309:
RMSE of position differences, raw and after aligning the trajectories); the alignment is the closed-form
least-squares solution (Kabsch/Umeyama without scale) instead of its 100-iteration robust ICP."""
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
