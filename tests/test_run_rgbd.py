"""TUM RGB-D folder support (SURVEY.md 8f rows 2 + 4): the general PNG reader (16-bit depth, colour), time-stamp association, and
tools/run_rgbd.py end to end on a folder written by the test from the synthetic renderer."""
import os
import struct
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from vslam_pose_estimation_framework_amd import io_formats as io  # noqa: E402


def _encode_with_filters(img, filters):
    """A PNG whose rows use the given filter types in turn (what real encoders emit): test encoder, independent of write_png."""
    if img.dtype == np.uint16:
        rows, depth, ctype, bpp = img.astype(">u2").view(np.uint8).reshape(img.shape[0], -1), 16, 0, 2
    elif img.ndim == 3:
        rows, depth, ctype, bpp = img.reshape(img.shape[0], -1), 8, 2, 3
    else:
        rows, depth, ctype, bpp = img, 8, 0, 1
    h, stride = rows.shape
    prev = np.zeros(stride, np.int32)
    raw = b""
    for y in range(h):
        cur = rows[y].astype(np.int32)
        ft = filters[y % len(filters)]
        out = np.zeros(stride, np.int32)
        for x in range(stride):
            a = int(cur[x - bpp]) if x >= bpp else 0
            b = int(prev[x]); c = int(prev[x - bpp]) if x >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = b
            elif ft == 3: pred = (a + b) >> 1
            else:
                pp = a + b - c
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[x] = (int(cur[x]) - pred) & 255
        raw += bytes([ft]) + out.astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    w = img.shape[1]
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def test_png_reader_all_filters_and_formats(tmp_path):
    rng = np.random.default_rng(3)
    images = [rng.integers(0, 256, (12, 19), dtype=np.uint8), rng.integers(0, 65536, (11, 14), dtype=np.uint16),
              rng.integers(0, 256, (10, 13, 3), dtype=np.uint8)]
    for img in images:
        p = str(tmp_path / "a.png")
        io.write_png(p, img)
        back = io.read_png(p)
        assert back.dtype == img.dtype and np.array_equal(back, img)
        with open(p, "wb") as f:
            f.write(_encode_with_filters(img, [0, 1, 2, 3, 4]))
        back = io.read_png(p)
        assert back.dtype == img.dtype and np.array_equal(back, img)
    # the 8-bit grayscale reader of the stereo path and the general one agree
    p = str(tmp_path / "g.png")
    io.write_png_gray8(p, images[0])
    assert np.array_equal(io.read_png(p), io.read_png_gray8(p))
    with pytest.raises(RuntimeError):
        open(p, "wb").write(b"not a png at all"); io.read_png(p)


def test_gray_conversion_and_association():
    # cv::cvtColor(BGR2GRAY) fixed point: grey stays grey, pure channels get 4899 / 9617 / 1868 of 16384 (rounded)
    g = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(io.rgb_to_gray_opencv(np.stack([g, g, g], axis=2)), g)
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], np.uint8)
    assert list(io.rgb_to_gray_opencv(px)[0]) == [76, 150, 29, (10 * 4899 + 200 * 9617 + 30 * 1868 + 8192) >> 14]
    # associate.py: closest pairs first, every stamp once, sorted by the first list; nothing beyond max_difference
    assert io.associate([0.0, 0.1, 0.2, 0.31], [0.105, 0.19, 0.5, 0.012]) == [(0, 3), (1, 0), (2, 1)]
    assert io.associate([1.0], [1.03]) == [] and io.associate([1.0], [1.03], max_difference=0.05) == [(0, 0)]
    assert io.associate([0.10, 0.11], [0.104]) == [(0, 0)]        # one depth image cannot serve two colour images


def _write_tum_folder(root, o, scene, n, unit, skew=0.004):
    (root / "rgb").mkdir(parents=True); (root / "depth").mkdir()
    rgb_lines, dep_lines, gt_lines, frames = ["# color images", "# timestamp filename"], ["# depth maps"], ["# ground truth trajectory", "# timestamp tx ty tz qx qy qz qw"], []
    for k in range(n):
        L = o.render(scene, k)[0]
        D = o.render_depth(scene, k, unit)
        t = 1305031100.0 + k / 30.0
        io.write_png(str(root / "rgb" / ("%.6f.png" % t)), np.stack([L, L, L], axis=2))        # colour file with grey content
        io.write_png(str(root / "depth" / ("%.6f.png" % (t + skew))), D)
        rgb_lines.append("%.6f rgb/%.6f.png" % (t, t)); dep_lines.append("%.6f depth/%.6f.png" % (t + skew, t + skew))
        T = np.array(o.gt_pose(scene, k)).reshape(3, 4)
        q = io.rotation_to_quaternion(T[:, :3])
        gt_lines.append("%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f" % ((t, T[0, 3], T[1, 3], T[2, 3]) + tuple(q)))
        frames.append((L, D))
    (root / "rgb.txt").write_text("\n".join(rgb_lines) + "\n")
    (root / "depth.txt").write_text("\n".join(dep_lines) + "\n")
    (root / "groundtruth.txt").write_text("\n".join(gt_lines) + "\n")
    return frames


def test_tum_folder_reader(tmp_path):
    from _oracle import Oracle
    o = Oracle()
    scene = o.scene_kitti(scale=0.25, seed=5)
    frames = _write_tum_folder(tmp_path / "seq", o, scene, 4, 2e-3)
    seq = io.TumRgbdSequence(str(tmp_path / "seq"))
    assert len(seq) == 4 and seq.ground_truth_path.endswith("groundtruth.txt") and abs(seq.times[1] - seq.times[0] - 1 / 30.0) < 1e-6
    for k, (L, D) in enumerate(frames):
        g, d = seq.frame(k)
        assert g.dtype == np.uint8 and d.dtype == np.uint16 and np.array_equal(g, L) and np.array_equal(d, D)
    o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["tum", "icl"])
def test_run_rgbd_folder_end_to_end(which, tmp_path):
    """Folder reader -> vslam_rgbd_* (device-resident loop) -> TUM trajectory -> the analyzer's RMSE; the poses equal a direct run of the
    tracker on the rendered arrays, and the trajectory follows the ground truth."""
    import run_rgbd
    from _oracle import Oracle
    from vslam_pose_estimation_framework_amd import hip
    from vslam_pose_estimation_framework_amd.capi import RgbdTracker
    o = Oracle()
    scene = o.scene_kitti(scale=0.5, seed=13)
    scene.speed_m = 0.25; scene.sway_m = 0.4
    n, unit = 14, 2e-3
    frames = _write_tum_folder(tmp_path / "seq", o, scene, n, unit)
    intr = "%r,%r,%r,%r" % (scene.fx, scene.fy, scene.cx, scene.cy)
    out = str(tmp_path / "traj.txt")
    res = run_rgbd.run(str(tmp_path / "seq"), which, intr, unit, out, depth_scale=4.0, log=lambda *_: None)
    assert res["frames"] == n and res["error_flags"] == 0
    g = hip.load()
    K = np.array([[scene.fx, 0, scene.cx], [0, scene.fy, scene.cy], [0, 0, 1.0]])
    cfg, p = run_rgbd.configure(g, which, scene.rows, scene.cols, K, unit, 1, 0, 4.0)
    tr = RgbdTracker(g, cfg, p)
    try:
        for k, (L, D) in enumerate(frames):
            fi, _ = tr.process(L, D)
            np.testing.assert_array_equal(np.array(fi.camera_left_to_world).reshape(3, 4), res["poses"][k])
        assert fi.status == 1 and fi.n_tracked > 20
    finally:
        tr.destroy(); o.destroy()
    lines = open(out).read().splitlines()
    assert len(lines) == n and all(len(ln.split()) == 8 and ln.endswith(" ") for ln in lines)      # world_map.cpp:222-258: "%.9f " per value
    assert abs(float(lines[3].split()[0]) - (1305031100.0 + 3 / 30.0)) < 1e-5
    ta = res["trajectory_analyzer"]
    assert ta["correspondences"] >= n - 2 and ta["optimal_rmse"] < 0.05, ta      # 3.25 m of path
