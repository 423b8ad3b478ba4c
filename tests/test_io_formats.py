"""I/O formats next to the hot path: KITTI folder + calib parser, PNG, trajectory writers."""
import os

import numpy as np

from vslam_pose_estimation_framework_amd import io_formats as io


def test_png_roundtrip_all_filter_types(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    p = str(tmp_path / "a.png")
    io.write_png_gray8(p, img)
    np.testing.assert_array_equal(io.read_png_gray8(p), img)
    # hand-filtered file exercising sub / up / average / paeth
    import struct, zlib
    h, w = img.shape
    raw = bytearray()
    prev = np.zeros(w, np.int32)
    for y in range(h):
        ft = y % 5
        cur = img[y].astype(np.int32)
        line = np.zeros(w, np.int32)
        for x in range(w):
            left = int(cur[x - 1]) if x else 0
            up = int(prev[x])
            ul = int(prev[x - 1]) if x else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = left
            elif ft == 2: pred = up
            elif ft == 3: pred = (left + up) >> 1
            else:
                pp = left + up - ul
                pa, pb, pc = abs(pp - left), abs(pp - up), abs(pp - ul)
                pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
            line[x] = (int(cur[x]) - pred) & 255
        raw += bytes([ft]) + bytes(line.astype(np.uint8))
        prev = cur
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    q = str(tmp_path / "b.png")
    open(q, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                        chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    np.testing.assert_array_equal(io.read_png_gray8(q), img)


def test_kitti_folder_and_calibration(tmp_path):
    root = tmp_path / "00"
    (root / "image_0").mkdir(parents=True)
    (root / "image_1").mkdir()
    rng = np.random.default_rng(2)
    frames = [(rng.integers(0, 256, (24, 40), dtype=np.uint8), rng.integers(0, 256, (24, 40), dtype=np.uint8)) for _ in range(3)]
    for k, (L, R) in enumerate(frames):
        io.write_png_gray8(str(root / "image_0" / ("%06d.png" % k)), L)
        io.write_png_gray8(str(root / "image_1" / ("%06d.png" % k)), R)
    (root / "calib.txt").write_text(
        "P0: 7.188560000000e+02 0.000000000000e+00 6.071928000000e+02 0.000000000000e+00 0.000000000000e+00 "
        "7.188560000000e+02 1.852157000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00\n"
        "P1: 7.188560000000e+02 0.000000000000e+00 6.071928000000e+02 -3.861448000000e+02 0.000000000000e+00 "
        "7.188560000000e+02 1.852157000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00\n")
    (root / "times.txt").write_text("0.0\n0.1\n0.2\n")
    seq = io.KittiSequence(str(root))
    assert len(seq) == 3 and seq.times == [0.0, 0.1, 0.2]
    assert seq.K[0, 0] == 718.856 and seq.K[0, 2] == 607.1928 and seq.K[1, 1] == 718.856 and seq.K[1, 2] == 185.2157
    assert seq.baseline[0] == -386.1448 and seq.baseline[1] == 0
    L, R = seq.pair(2)
    np.testing.assert_array_equal(L, frames[2][0])
    np.testing.assert_array_equal(R, frames[2][1])


def test_trajectory_writers(tmp_path):
    rng = np.random.default_rng(4)
    poses = []
    for _ in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        poses.append(np.hstack([R, rng.normal(size=(3, 1))]))
    poses = np.array(poses)
    pk = str(tmp_path / "trajectory_kitti.txt")
    io.write_trajectory_kitti(pk, poses)
    first = open(pk).readline()
    assert first.endswith(" \n") and len(first.split()) == 12 and all(len(v.split(".")[1]) == 9 for v in first.split())
    np.testing.assert_allclose(io.read_trajectory_kitti(pk), poses, atol=1e-9)
    pt = str(tmp_path / "trajectory_tum.txt")
    io.write_trajectory_tum(pt, poses, [0.1 * k for k in range(5)])
    rows = np.loadtxt(pt)
    assert rows.shape == (5, 8)
    for k in range(5):
        x, y, z, w = rows[k, 4:]
        assert abs(x * x + y * y + z * z + w * w - 1) < 1e-8
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(R, poses[k][:, :3], atol=1e-8)
        np.testing.assert_allclose(rows[k, 1:4], poses[k][:, 3], atol=1e-9)


def _write_asl(path, t_ns, xyz):
    with open(path, "w") as f:
        f.write("#timestamp [ns], p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z []\n")
        for t, p in zip(t_ns, xyz):
            f.write("%d,%.9f,%.9f,%.9f,1.0,0.0,0.0,0.0\n" % (t, p[0], p[1], p[2]))


def test_trajectory_analyzer_known_answer(tmp_path):
    """executables/trajectory_analyzer.cpp restated (time-stamp interpolation + its 100-round robust alignment): a trajectory
    that IS the ground truth moved by a known SE3 must come back with ~zero optimal RMSE, the raw RMSE must equal the
    hand-computed one, and on exact sample-aligned data the closed-form Kabsch ATE agrees."""
    from vslam_pose_estimation_framework_amd import evaluation as ev
    rng = np.random.default_rng(4)
    n_gt = 400
    t_gt = 1.4e18 + np.arange(n_gt) * 5_000_000            # 200 Hz ground truth, nanoseconds
    s = np.linspace(0, 12, n_gt)
    p_gt = np.stack([3 * np.cos(0.5 * s), 2 * np.sin(0.7 * s), 0.3 * s], axis=1)
    # SLAM samples at 20 Hz, between ground-truth samples (offset 1.3 ms), expressed in another frame
    idx = np.arange(5, n_gt - 5, 10)
    t_s = t_gt[idx] / 1e9 + 0.0013
    lerp = p_gt[idx] + (0.0013 / 0.005) * (p_gt[idx + 1] - p_gt[idx])     # what the tool must interpolate
    ang = 0.3
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    t0 = np.array([0.4, -0.2, 0.1])
    p_s = (lerp - t0) @ R                                    # lerp = R p_s + t0
    poses = np.zeros((len(idx), 3, 4)); poses[:, :, :3] = np.eye(3); poses[:, :, 3] = p_s
    tum = str(tmp_path / "traj.txt"); asl = str(tmp_path / "gt.csv")
    io.write_trajectory_tum(tum, poses, t_s)
    _write_asl(asl, t_gt, p_gt)
    r = ev.trajectory_analyzer(tum, asl)
    assert r["correspondences"] == len(idx)
    # raw RMSE: measurements shifted by the first interpolated ground-truth position, nothing else
    want_raw = np.sqrt((((p_s + lerp[0]) - lerp) ** 2).sum(1).mean())
    assert abs(r["raw_rmse"] - want_raw) < 1e-6
    assert r["optimal_rmse"] < 1e-6 and r["optimal_rmse"] < 1e-3 * r["raw_rmse"]
    np.testing.assert_allclose(r["transform"][:3, :3], R, atol=1e-6)
    # with noise the tool's robust optimum and the closed form agree closely (all points are inliers of the 1 m^2 kernel)
    noisy = poses.copy(); noisy[:, :, 3] += rng.normal(scale=0.02, size=(len(idx), 3))
    io.write_trajectory_tum(tum, noisy, t_s)
    r2 = ev.trajectory_analyzer(tum, asl)
    gtp = np.zeros_like(poses); gtp[:, :, 3] = lerp
    closed = ev.ate_rmse(noisy, gtp)
    assert abs(r2["optimal_rmse"] - closed) < 2e-4 and 0.02 < closed < 0.05
    assert r2["iterations"][-1][1] == len(idx)
    # -skip cuts both ends; measurements before the ground truth starts are dropped (closest sample = index 0)
    r3 = ev.trajectory_analyzer(tum, asl, skip=3)
    assert r3["correspondences"] == len(idx) - 6
    early = noisy[:4].copy()
    io.write_trajectory_tum(tum, np.concatenate([early, noisy]), np.concatenate([t_s[:4] - 5.0, t_s]))
    assert ev.trajectory_analyzer(tum, asl)["correspondences"] == len(idx)
    # the command-line front end
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "trajectory_analyzer.py"), "-tum", tum, "-asl", asl], capture_output=True, text=True)
    assert out.returncode == 0 and "optimal RMSE" in out.stderr and out.stdout.count("iteration:") == 100


def test_euroc_folder(tmp_path):
    """mav0/cam0|cam1/data.csv + data/<stamp>.png: pairs formed on equal time stamps, unmatched frames dropped, ground truth found."""
    rng = np.random.default_rng(6)
    base = tmp_path / "MH_01" / "mav0"
    stamps = [1403636579763555584 + 50_000_000 * k for k in range(5)]
    imgs = {}
    for cam, keep in (("cam0", stamps), ("cam1", stamps[:2] + stamps[3:])):     # cam1 lost one frame
        (base / cam / "data").mkdir(parents=True)
        with open(base / cam / "data.csv", "w") as f:
            f.write("#timestamp [ns],filename\n")
            for ts in keep:
                f.write("%d,%d.png\n" % (ts, ts))
                im = rng.integers(0, 256, (20, 32), dtype=np.uint8)
                imgs[(cam, ts)] = im
                io.write_png_gray8(str(base / cam / "data" / ("%d.png" % ts)), im)
    (base / "state_groundtruth_estimate0").mkdir()
    _write_asl(str(base / "state_groundtruth_estimate0" / "data.csv"), stamps, np.zeros((5, 3)))
    seq = io.EurocSequence(str(tmp_path / "MH_01"))
    assert len(seq) == 4 and seq.stamps == stamps[:2] + stamps[3:]
    L, R = seq.pair(2)
    np.testing.assert_array_equal(L, imgs[("cam0", stamps[3])])
    np.testing.assert_array_equal(R, imgs[("cam1", stamps[3])])
    assert abs(seq.times[1] - seq.times[0] - 0.05) < 1e-6      # seconds as doubles at 1.4e9: ~2e-7 resolution
    assert seq.ground_truth_path.endswith("state_groundtruth_estimate0/data.csv") and seq.calibration() is None
