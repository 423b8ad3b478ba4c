"""Data formats on either side of the hot path.

* KITTI odometry folder (`image_0/`, `image_1/`, `calib.txt`, `times.txt`) -> stereo pairs + camera values, with the
  calibration parsed exactly like the reference's harness (executables/test_stereo_frontend.cpp:281-312: first line =
  left projection matrix P0 -> fx, cx, fy, cy; second line P1 -> b_x = P1(0,3)).
* trajectory writers equal to WorldMap::writeTrajectoryKITTI / writeTrajectoryTUM (src/types/world_map.cpp:184-258):
  fixed notation, 9 digits, one trailing blank before the newline.
* a dependency-free reader/writer for 8-bit grayscale PNG (what KITTI ships): no OpenCV/PIL in this image; the general reader
  (8 / 16-bit grayscale, 8-bit RGB / RGBA) serves the RGB-D data sets.
* TUM RGB-D folder (`rgb.txt`, `depth.txt`, `rgb/`, `depth/`, optional `groundtruth.txt`; also how ICL-NUIM is distributed) -> gray image +
  16-bit depth pairs, associated on time stamps like the benchmark's associate.py."""
import os
import struct
import zlib

import numpy as np


# ---- KITTI calibration -------------------------------------------------------------------------------------
def parse_kitti_calib(path):
    """Returns (K 3x3, baseline_homogeneous 3) as getCameraCalibrationMatrixKITTI does."""
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines()]
    if not lines or not lines[0].strip():
        raise RuntimeError("invalid camera calibration file provided")
    a = lines[0].split()      # "P0:" fx 0 cx 0 0 fy cy 0 0 0 1 0
    K = np.eye(3)
    K[0, 0] = float(a[1]); K[0, 2] = float(a[3]); K[1, 1] = float(a[6]); K[1, 2] = float(a[7])
    b = lines[1].split()      # "P1:" fx 0 cx bx ...
    baseline = np.zeros(3)
    baseline[0] = float(b[4])
    return K, baseline


def apply_calib(cfg, K, baseline, rows, cols):
    cfg.rows, cfg.cols = int(rows), int(cols)
    for i in range(9):
        cfg.K[i] = float(K.reshape(9)[i])
    for i in range(3):
        cfg.baseline_h[i] = float(baseline[i])
    return cfg


# ---- minimal PNG (8-bit grayscale, non-interlaced) --------------------------------------------------------------
def write_png_gray8(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png_gray8(path):
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise RuntimeError("not a PNG: " + path)
    pos, idat, w = 8, [], None
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if t == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if depth != 8 or ctype != 0 or interlace != 0:
                raise RuntimeError("only 8-bit grayscale non-interlaced PNG is supported: " + path)
        elif t == b"IDAT":
            idat.append(body)
        elif t == b"IEND":
            break
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, w + 1)
    out = np.zeros((h, w), np.uint8)
    prev = np.zeros(w, np.int32)
    for y in range(h):
        ft = int(raw[y, 0])
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = np.cumsum(line) & 255
        else:  # 3 (average) and 4 (Paeth) are sequential in x
            cur = np.zeros(w, np.int32)
            left = 0
            upleft = 0
            for x in range(w):
                up = int(prev[x])
                if ft == 3:
                    pred = (left + up) >> 1
                else:
                    p = left + up - upleft
                    pa, pb, pc = abs(p - left), abs(p - up), abs(p - upleft)
                    pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else upleft)
                left = (int(line[x]) + pred) & 255
                cur[x] = left
                upleft = up
        out[y] = cur
        prev = cur
    return out


# ---- general PNG reader / writer (non-interlaced; gray 8 / 16 bit, RGB / RGBA 8 bit) ----------------------------------------------------
def _png_chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def write_png(path, img):
    """img: (h, w) uint8 | (h, w) uint16 (big-endian samples in the file, as PNG wants) | (h, w, 3) uint8."""
    img = np.asarray(img)
    if img.ndim == 2 and img.dtype == np.uint16:
        h, w = img.shape; depth, ctype = 16, 0
        rows = img.astype(">u2")
    elif img.ndim == 2:
        h, w = img.shape; depth, ctype = 8, 0
        rows = np.ascontiguousarray(img, np.uint8)
    elif img.ndim == 3 and img.shape[2] == 3:
        h, w = img.shape[:2]; depth, ctype = 8, 2
        rows = np.ascontiguousarray(img, np.uint8)
    else:
        raise ValueError("write_png: unsupported array")
    raw = b"".join(b"\x00" + rows[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                _png_chunk(b"IDAT", zlib.compress(raw, 6)) + _png_chunk(b"IEND", b""))


def _png_unfilter(raw, h, stride, bpp):
    """Reverses the five PNG row filters; raw: h rows of 1 + stride bytes; bpp: bytes per complete pixel."""
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = int(raw[y, 0])
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:        # Sub: a running sum per byte lane of the pixel
            cur = line.copy()
            for o in range(bpp):
                cur[o::bpp] = np.cumsum(line[o::bpp]) & 255
        else:                # 3 (average) and 4 (Paeth) are sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                left = int(cur[x - bpp]) if x >= bpp else 0
                up = int(prev[x])
                upleft = int(prev[x - bpp]) if x >= bpp else 0
                if ft == 3:
                    pred = (left + up) >> 1
                elif ft == 4:
                    pp = left + up - upleft
                    pa, pb, pc = abs(pp - left), abs(pp - up), abs(pp - upleft)
                    pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else upleft)
                else:
                    raise RuntimeError("bad PNG filter type %d" % ft)
                cur[x] = (int(line[x]) + pred) & 255
        out[y] = cur
        prev = cur
    return out


def read_png(path):
    """(h, w) uint8 / uint16 for grayscale, (h, w, 3 | 4) uint8 for RGB / RGBA; non-interlaced files only."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise RuntimeError("not a PNG: " + path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif t == b"IDAT":
            idat.append(body)
        elif t == b"IEND":
            break
    if hdr is None:
        raise RuntimeError("PNG without IHDR: " + path)
    w, h, depth, ctype, _, _, interlace = hdr
    channels = {0: 1, 2: 3, 6: 4}.get(ctype)
    if interlace != 0 or channels is None or depth not in (8, 16) or (depth == 16 and channels != 1):
        raise RuntimeError("unsupported PNG (colour type %d, %d bit, interlace %d): %s" % (ctype, depth, interlace, path))
    bpp = channels * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    px = _png_unfilter(raw, h, stride, bpp)
    if depth == 16:
        return px.reshape(h, w, 2).astype(np.uint16)[:, :, 0] * 256 + px.reshape(h, w, 2)[:, :, 1]
    return px.reshape(h, w) if channels == 1 else px.reshape(h, w, channels)


def rgb_to_gray_opencv(rgb):
    """cv::imread(..., IMREAD_GRAYSCALE) / cvtColor(BGR2GRAY) on 8-bit data [recalled: fixed point, 14 fractional bits]:
    (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14."""
    a = np.asarray(rgb).astype(np.int32)
    return ((a[:, :, 0] * 4899 + a[:, :, 1] * 9617 + a[:, :, 2] * 1868 + 8192) >> 14).astype(np.uint8)


# ---- TUM RGB-D folder ----------------------------------------------------------------------------------------------------------------
# The RGB-D benchmark's camera intrinsics (fx, fy, cx, cy) and its 16-bit depth unit (1 / 5000 m); ICL-NUIM is distributed in the same layout.
TUM_INTRINSICS = {"freiburg1": (517.3, 516.5, 318.6, 255.3), "freiburg2": (520.9, 521.0, 325.1, 249.7), "freiburg3": (535.4, 539.2, 320.1, 247.6),
                  "icl": (481.2, 480.0, 319.5, 239.5)}
TUM_DEPTH_UNIT_M = 1.0 / 5000.0


def read_tum_list(path):
    """`timestamp value...` lines, `#` comments: [(float timestamp, [tokens])]."""
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line[0] == "#":
                continue
            a = line.replace(",", " ").split()
            out.append((float(a[0]), a[1:]))
    return out


def associate(first, second, max_difference=0.02, offset=0.0):
    """The benchmark's associate.py: all pairs closer than max_difference, best first, every time stamp used once; sorted by the first."""
    cand = sorted((abs(a - (b + offset)), i, j) for i, a in enumerate(first) for j, b in enumerate(second) if abs(a - (b + offset)) < max_difference)
    used_a, used_b, pairs = set(), set(), []
    for _, i, j in cand:
        if i not in used_a and j not in used_b:
            used_a.add(i); used_b.add(j); pairs.append((i, j))
    return sorted(pairs)


class TumRgbdSequence(object):
    """`<root>/rgb.txt`, `<root>/depth.txt` (`timestamp path`), images under `<root>/rgb/`, `<root>/depth/`; optional
    `<root>/groundtruth.txt` (`timestamp tx ty tz qx qy qz qw`: the trajectory_analyzer's TUM format).  frame(k) = (gray u8, depth u16)."""

    def __init__(self, root, max_difference=0.02):
        self.root = root
        rgb = read_tum_list(os.path.join(root, "rgb.txt"))
        dep = read_tum_list(os.path.join(root, "depth.txt"))
        pairs = associate([t for t, _ in rgb], [t for t, _ in dep], max_difference)
        self.times = [rgb[i][0] for i, _ in pairs]
        self.rgb = [rgb[i][1][0] for i, _ in pairs]
        self.depth = [dep[j][1][0] for _, j in pairs]
        gt = os.path.join(root, "groundtruth.txt")
        self.ground_truth_path = gt if os.path.exists(gt) else None

    def __len__(self):
        return len(self.times)

    def frame(self, k):
        img = read_png(os.path.join(self.root, self.rgb[k]))
        gray = rgb_to_gray_opencv(img[:, :, :3]) if img.ndim == 3 else (img if img.dtype == np.uint8 else (img >> 8).astype(np.uint8))
        depth = read_png(os.path.join(self.root, self.depth[k]))
        if depth.dtype != np.uint16:
            raise RuntimeError("depth image is not 16-bit: " + self.depth[k])
        return gray, depth


# ---- KITTI odometry sequence folder ---------------------------------------------------------------------------------
class KittiSequence(object):
    """<root>/image_0/000000.png, <root>/image_1/000000.png, <root>/calib.txt, <root>/times.txt"""

    def __init__(self, root):
        self.root = root
        self.K, self.baseline = parse_kitti_calib(os.path.join(root, "calib.txt"))
        names = sorted(n for n in os.listdir(os.path.join(root, "image_0")) if n.endswith(".png"))
        self.names = names
        tpath = os.path.join(root, "times.txt")
        self.times = [float(v) for v in open(tpath).read().split()] if os.path.exists(tpath) else list(range(len(names)))

    def __len__(self):
        return len(self.names)

    def pair(self, k):
        left = read_png_gray8(os.path.join(self.root, "image_0", self.names[k]))
        right = read_png_gray8(os.path.join(self.root, "image_1", self.names[k]))
        return left, right


# ---- EuRoC / ASL folder (mav0/cam0|cam1/data.csv + data/<timestamp>.png) ----------------------------------------------------------
class EurocSequence(object):
    """`<root>/mav0/cam0/data.csv` and `cam1/data.csv` list `timestamp_ns,filename` (one `#` header line); images live in
    `camN/data/`.  Pairs are formed on equal time stamps (the two cameras are hardware-synchronised); `times` in seconds.
    Calibration: configuration_euroc.yaml's data set is rectified upstream by the message converter
    (executables/srrg_proslam_synchronizer / ROS rectification), so K / baseline come from the caller or from
    `vslam_default_config_euroc`; `calibration()` reads them from an optional `calib.txt` in KITTI P0/P1 form when a
    rectified export carries one.  Ground truth for trajectory_analyzer: `mav0/state_groundtruth_estimate0/data.csv` or
    `mav0/leica0/data.csv` (`ground_truth_path`)."""

    def __init__(self, root):
        self.root = root
        base = os.path.join(root, "mav0") if os.path.isdir(os.path.join(root, "mav0")) else root
        self.base = base
        left = self._listing(os.path.join(base, "cam0"))
        right = dict(self._listing(os.path.join(base, "cam1")))
        self.stamps, self.left, self.right = [], [], []
        for ts, name in left:
            if ts in right:
                self.stamps.append(ts); self.left.append(name); self.right.append(right[ts])
        self.times = [ts / 1e9 for ts in self.stamps]

    @staticmethod
    def _listing(cam_dir):
        path = os.path.join(cam_dir, "data.csv")
        out = []
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line or line[0] == "#":
                    continue
                a = line.split(",")
                out.append((int(a[0]), a[1].strip() if len(a) > 1 and a[1].strip() else a[0].strip() + ".png"))
        return out

    def __len__(self):
        return len(self.stamps)

    def pair(self, k):
        return (read_png_gray8(os.path.join(self.base, "cam0", "data", self.left[k])),
                read_png_gray8(os.path.join(self.base, "cam1", "data", self.right[k])))

    def calibration(self):
        path = os.path.join(self.root, "calib.txt")
        return parse_kitti_calib(path) if os.path.exists(path) else None

    @property
    def ground_truth_path(self):
        for sub in ("state_groundtruth_estimate0", "leica0"):
            p = os.path.join(self.base, sub, "data.csv")
            if os.path.exists(p):
                return p
        return None


# ---- trajectory writers -----------------------------------------------------------------------------------------------
def write_trajectory_kitti(path, poses):
    with open(path, "w") as f:
        for T in np.asarray(poses, np.float64).reshape(-1, 12):
            f.write("".join("%.9f " % v for v in T) + "\n")


def rotation_to_quaternion(R):
    """Eigen::Quaternion(Matrix3) (xyzw returned)."""
    R = np.asarray(R, np.float64).reshape(3, 3)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        x, y, z = (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = [0.0, 0.0, 0.0]
        q[i] = 0.5 * t
        t = 0.5 / t
        w = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
        x, y, z = q
    return x, y, z, w


def write_trajectory_tum(path, poses, timestamps):
    with open(path, "w") as f:
        for T, ts in zip(np.asarray(poses, np.float64).reshape(-1, 3, 4), timestamps):
            x, y, z, w = rotation_to_quaternion(T[:, :3])
            vals = (ts, T[0, 3], T[1, 3], T[2, 3], x, y, z, w)
            f.write("".join("%.9f " % v for v in vals) + "\n")


def read_trajectory_kitti(path):
    return np.loadtxt(path).reshape(-1, 3, 4)
