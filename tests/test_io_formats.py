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
