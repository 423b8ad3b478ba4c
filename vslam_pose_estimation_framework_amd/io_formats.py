"""Data formats on either side of the hot path.

* KITTI odometry folder (`image_0/`, `image_1/`, `calib.txt`, `times.txt`) -> stereo pairs + camera values, with the
  calibration parsed exactly like the reference's harness (executables/test_stereo_frontend.cpp:281-312: first line =
  left projection matrix P0 -> fx, cx, fy, cy; second line P1 -> b_x = P1(0,3)).
* trajectory writers equal to WorldMap::writeTrajectoryKITTI / writeTrajectoryTUM (src/types/world_map.cpp:184-258):
  fixed notation, 9 digits, one trailing blank before the newline.
* a dependency-free reader/writer for 8-bit grayscale PNG (what KITTI ships): no OpenCV/PIL in this image."""
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
