"""tools/run_kitti.py end to end on a KITTI-layout folder (image_0/ image_1/ calib.txt times.txt) written by this test
from the synthetic renderer: folder reader + calib parser -> HIP front end -> trajectory writers -> ATE tool."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
def test_run_kitti_folder_end_to_end(tmp_path):
    import run_kitti
    from _oracle import Oracle
    from vslam_pose_estimation_framework_amd import io_formats as io
    o = Oracle()
    scene = o.scene_kitti(scale=0.5, seed=9)
    n = 16
    seq = tmp_path / "seq"
    (seq / "image_0").mkdir(parents=True)
    (seq / "image_1").mkdir(parents=True)
    gt = []
    for k in range(n):
        L, R = o.render(scene, k)
        io.write_png_gray8(str(seq / "image_0" / ("%06d.png" % k)), L)
        io.write_png_gray8(str(seq / "image_1" / ("%06d.png" % k)), R)
        gt.append(np.array(o.gt_pose(scene, k)).reshape(12))
    fx, cx, cy, bx = scene.fx, scene.cx, scene.cy, -scene.fx * scene.baseline_m
    with open(seq / "calib.txt", "w") as f:
        f.write("P0: %r 0 %r 0 0 %r %r 0 0 0 1 0\n" % (fx, cx, scene.fy, cy))
        f.write("P1: %r 0 %r %r 0 %r %r 0 0 0 1 0\n" % (fx, cx, bx, scene.fy, cy))
    with open(seq / "times.txt", "w") as f:
        f.write("\n".join("%.6f" % (0.1 * k) for k in range(n)) + "\n")
    gt0 = np.array(gt)
    io.write_trajectory_kitti(str(tmp_path / "gt.txt"), gt0)
    out = str(tmp_path / "traj.txt")
    res = run_kitti.run(str(seq), out, "kitti", str(tmp_path / "gt.txt"), log=lambda *_: None)
    assert res["frames"] == n and res["error_flags"] == 0
    # the same sequence through the oracle: identical poses (exact mode), so the written file equals the oracle's trajectory
    cfg = o.config_for_scene(scene)
    o.create(cfg, 0, 1)
    for k in range(n):
        o.process_host(*o.render(scene, k))
    ref = np.array(o.poses(0, 0, n)).reshape(n, 12)
    o.destroy()
    got = io.read_trajectory_kitti(out).reshape(n, 12)
    assert np.abs(got - ref).max() < 1e-6            # %.9f text round trip
    assert res["ate_rmse_aligned"] < 0.5             # metres over 16 frames (~15 m of path)
    out_tum = str(tmp_path / "traj_tum.txt")
    run_kitti.run(str(seq), out_tum, "tum", log=lambda *_: None, max_frames=5)
    rows = [ln.split() for ln in open(out_tum).read().splitlines()]
    assert len(rows) == 5 and all(len(r) == 8 for r in rows)
    # frame-sharded mode on the same folder: 3 chunks with 3 warm-up frames side by side, chained at the seams; every frame gets a
    # pose, the first chunk is the sequential run itself, and the whole trajectory stays close to it (approximate at the seams)
    out_ch = str(tmp_path / "traj_chunks.txt")
    rc = run_kitti.run(str(seq), out_ch, "kitti", str(tmp_path / "gt.txt"), log=lambda *_: None, chunks=3, overlap=3)
    assert rc["frames"] == n and rc["error_flags"] == 0
    ch = io.read_trajectory_kitti(out_ch).reshape(n, 12)
    assert np.abs(ch[:6] - ref[:6]).max() < 1e-6      # chunk 0 = frames 0 .. 5, no seam before it
    assert np.abs(ch.reshape(n, 3, 4)[:, :, 3] - ref.reshape(n, 3, 4)[:, :, 3]).max() < 0.25
    assert rc["ate_rmse_aligned"] < 0.5


@pytest.mark.gpu
def test_run_euroc_folder_end_to_end(tmp_path):
    """An ASL / EuRoC-layout folder (mav0/cam0|cam1/data.csv + PNGs + ground-truth csv) written from the EuRoC-shaped scene:
    folder reader -> HIP front end with the EuRoC configuration -> TUM trajectory -> the trajectory_analyzer restatement."""
    import run_kitti
    from _oracle import Oracle
    from vslam_pose_estimation_framework_amd import io_formats as io
    o = Oracle()
    scene = o.scene_euroc(seed=5)
    n = 14
    base = tmp_path / "MH" / "mav0"
    for cam in ("cam0", "cam1"):
        (base / cam / "data").mkdir(parents=True)
    stamps = [1403636579763555584 + 50_000_000 * k for k in range(n)]
    lines = ["#timestamp [ns],filename"]
    for k, ts in enumerate(stamps):
        L, R = o.render(scene, k)
        io.write_png_gray8(str(base / "cam0" / "data" / ("%d.png" % ts)), L)
        io.write_png_gray8(str(base / "cam1" / "data" / ("%d.png" % ts)), R)
        lines.append("%d,%d.png" % (ts, ts))
    for cam in ("cam0", "cam1"):
        (base / cam / "data.csv").write_text("\n".join(lines) + "\n")
    (base / "state_groundtruth_estimate0").mkdir()
    with open(base / "state_groundtruth_estimate0" / "data.csv", "w") as f:
        f.write("#timestamp [ns], p_x, p_y, p_z\n")
        # ground truth at 4x the camera rate, from two frames before the first image
        for j in range(-8, 4 * n + 8):
            k = j / 4.0
            k0 = int(np.floor(k)); a = k - k0
            p0 = np.array(o.gt_pose(scene, k0))[:, 3]; p1 = np.array(o.gt_pose(scene, k0 + 1))[:, 3]
            p = p0 + a * (p1 - p0)
            f.write("%d,%.9f,%.9f,%.9f\n" % (stamps[0] + int(round(k * 50_000_000)), p[0], p[1], p[2]))
    out = str(tmp_path / "traj_tum.txt")
    res = run_kitti.run(str(tmp_path / "MH"), out, "tum", log=lambda *_: None)
    assert res["frames"] == n and res["error_flags"] == 0
    ta = res["trajectory_analyzer"]
    assert ta["correspondences"] == n
    path = float(np.linalg.norm(np.array(o.gt_pose(scene, n - 1))[:, 3] - np.array(o.gt_pose(scene, 0))[:, 3]))
    assert ta["optimal_rmse"] < 0.05 * path + 0.02, (ta, path)          # a few cm over ~0.5 m of flight
    assert ta["optimal_rmse"] <= ta["raw_rmse"] + 1e-9
