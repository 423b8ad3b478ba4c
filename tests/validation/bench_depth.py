"""Kernel times of the RGB-D space map (vslam_depth_space_map: k_depth_init/min/pick/write) at sensor size (640x480) and
at a large size (4096x3072) where the kernels are bandwidth-bound; run under rocprofv3 --kernel-trace --stats.
Also checks the GPU map against the oracle at both sizes (bit-exact)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from vslam_pose_estimation_framework_amd import hip
from vslam_pose_estimation_framework_amd.capi import DepthParams
from _oracle import Oracle
api = hip.load(); api.create(api.default_config("kitti"), 0, 1)
orc = Oracle()
rng = np.random.default_rng(5)
for rows, cols in ((480, 640), (3072, 4096)):
    f = 525.0 * cols / 640
    Kr = np.array([[f, 0, cols / 2 - 0.5], [0, f, rows / 2 - 0.5], [0, 0, 1]])
    Kl = np.array([[f * 0.98, 0, cols / 2 + 1.5], [0, f * 0.98, rows / 2 - 2.5], [0, 0, 1]])
    ang = 0.01
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    r2l = np.hstack([R, np.array([[0.025], [0.0], [0.001]])])
    depth = rng.integers(500, 8000, (rows, cols)).astype(np.uint16)
    depth[rng.random((rows, cols)) < 0.1] = 0
    p = DepthParams.make(rows, cols, Kl, np.linalg.inv(Kl), np.linalg.inv(Kr), r2l, 1e-3, 0.1, 10.0, 1, 1, 10)
    for _ in range(5):
        space, rmap, cmap = api.depth_space_map(p, depth)
    so, ro, co = orc.depth_space_map(p, depth)
    same = bool(np.array_equal(space.view(np.uint32), so.view(np.uint32)) and np.array_equal(rmap, ro) and np.array_equal(cmap, co))
    print(rows, cols, "filled", int((rmap >= 0).sum()), "of", rows * cols, "gpu == oracle:", same, flush=True)
    assert same

# ---- compute / track / recover at sensor size: ~1500 features, ~700 previous points, ~300 lost landmarks ---------------
rows, cols = 480, 640
f = 525.0
K = np.array([[f, 0, 319.5], [0, f, 239.5], [0, 0, 1]])
depth = rng.integers(500, 8000, (rows, cols)).astype(np.uint16)
depth[rng.random((rows, cols)) < 0.1] = 0
p = DepthParams.make(rows, cols, K, np.linalg.inv(K), np.linalg.inv(K), np.eye(4)[:3], 1e-3, 0.1, 10.0, 1, 1, 10)
space, _, _ = api.depth_space_map(p, depth)
flat = np.sort(rng.choice(rows * cols, 1500, replace=False))
feats = np.stack([flat // cols, flat % cols], axis=1).astype(np.int32)
fdesc = rng.integers(0, 256, (1500, 32), dtype=np.uint8)
sel = rng.choice(1500, 700, replace=False)
cam = np.zeros((700, 3)); pdesc = np.zeros((700, 32), np.uint8)
for j, k in enumerate(sel):
    z = float(rng.uniform(0.8, 6.0)); r, c = feats[k] + rng.integers(-3, 4, 2)
    cam[j] = [(c - 319.5) * z / f, (r - 239.5) * z / f, z]
    bits = np.unpackbits(fdesc[k]); bits[rng.choice(256, int(rng.integers(0, 30)), replace=False)] ^= 1; pdesc[j] = np.packbits(bits)
flags = (rng.random(700) < 0.6).astype(np.uint8)
img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
lm = np.stack([rng.uniform(-2, 2, 300), rng.uniform(-1.5, 1.5, 300), rng.uniform(1.0, 6.0, 300)], axis=1)
for _ in range(5):
    new, xyz, temp, txyz = api.depth_compute(p, None, feats, feats[sel[:200]])
    tr, txyz2, tmp, lost, nlm = api.depth_track(p, None, np.eye(4)[:3], 7, 35.0, 0, cam, pdesc, flags, feats, fdesc)
    idx, xy, d, x3 = api.depth_recover(p, None, img, np.eye(4)[:3], np.ones(300, np.uint8), lm, rng.integers(0, 256, (300, 32), dtype=np.uint8), 7.0, 256.0)
o_new, _, o_temp, _ = orc.depth_compute(p, space, feats, feats[sel[:200]])
o_tr, _, o_tmp, o_lost, o_nlm = orc.depth_track(p, space, np.eye(4)[:3], 7, 35.0, 0, cam, pdesc, flags, feats, fdesc)
assert np.array_equal(new, o_new) and np.array_equal(temp, o_temp) and np.array_equal(tr, o_tr) and np.array_equal(tmp, o_tmp) and np.array_equal(lost, o_lost)
print("sensor-size compute/track/recover: new %d temp %d | tracked %d temporary %d lost %d | recovered %d of 300; gpu == oracle: True" % (
    len(new), len(temp), len(tr), len(tmp), len(lost), len(idx)), flush=True)
