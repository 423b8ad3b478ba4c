#!/usr/bin/env python3
"""Generate tests/golden/*.npz: known-answer vectors for the hot path.

The reference ships no tests or fixtures and cannot be built or imported here
(C++ on OpenCV3/Eigen/srrg_*; nothing was attempted or denied), so these vectors come from an
INDEPENDENT numpy / pure-Python restatement of the same published algorithms, written without
looking at oracle/vslam_oracle.cpp's code paths: FAST uses the literal OpenCV loop structure
(threshold table + run counting + cornerScore's two min/max sweeps), BRIEF sums 9x9 windows
directly instead of using an integral image, the aligner is written with numpy matrices
(skew, K, projection Jacobians) and solved with numpy.linalg.  The oracle and the HIP path are
both checked against these files.

Run:  python tests/golden/make_golden.py        (rewrites the .npz files deterministically)
"""
import os
import re
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


# ---------------------------------------------------------------------------------------------
def hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def gen_hamming(rng):
    a = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    a[64] = 0; b[64] = 0
    a[65] = 0; b[65] = 255
    a[66] = 255; b[66] = 255
    a[67] = 0; b[67] = 0; b[67, 31] = 1
    a[68] = 0; b[68] = 0; b[68, 0] = 0x80
    a[69] = 0xAA; b[69] = 0x55
    d = np.array([hamming(a[i], b[i]) for i in range(70)], np.int32)
    q = rng.integers(0, 256, (257, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (129, 32), dtype=np.uint8)
    t[5] = t[77]            # exact tie between two train rows -> lowest index first
    q[3] = t[5]             # zero distance
    q[200] = q[3]
    D = np.array([[hamming(q[i], t[j]) for j in range(129)] for i in range(257)], np.int64)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    dist_h = np.take_along_axis(D, order, 1).astype(np.float32)
    qs, ts = q.astype(np.int64), t.astype(np.int64)
    D2 = ((qs[:, None, :] - ts[None, :, :]) ** 2).sum(-1)
    order2 = np.argsort(D2, axis=1, kind="stable")[:, :2]
    dist_l2 = np.sqrt(np.take_along_axis(D2, order2, 1).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "hamming.npz"), a=a, b=b, d=d, q=q, t=t,
                        idx_h=order.astype(np.int32), dist_h=dist_h, idx_l2=order2.astype(np.int32),
                        dist_l2=dist_l2.astype(np.float32))


# ---------------------------------------------------------------------------------------------
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2),
          (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]  # (dx, dy), OpenCV makeOffsets(16)


def corner_score16(img, x, y, threshold):
    """OpenCV fast_score.cpp cornerScore<16>, literal."""
    v = int(img[y, x])
    d = [v - int(img[y + CIRCLE[k % 16][1], x + CIRCLE[k % 16][0]]) for k in range(25)]
    a0 = threshold
    for k in range(0, 16, 2):
        a = min(d[k + 1], d[k + 2], d[k + 3])
        if a <= a0:
            continue
        a = min(a, d[k + 4], d[k + 5], d[k + 6], d[k + 7], d[k + 8])
        a0 = max(a0, min(a, d[k]))
        a0 = max(a0, min(a, d[k + 9]))
    b0 = -a0
    for k in range(0, 16, 2):
        b = max(d[k + 1], d[k + 2], d[k + 3], d[k + 4], d[k + 5])
        if b >= b0:
            continue
        b = max(b, d[k + 6], d[k + 7], d[k + 8])
        b0 = min(b0, max(b, d[k]))
        b0 = min(b0, max(b, d[k + 9]))
    return -b0 - 1


def fast9_16(img, threshold):
    """OpenCV fast.cpp FAST_t<16> with nonmaxSuppression=true; returns list of (x, y, score)."""
    rows, cols = img.shape
    threshold = min(max(threshold, 0), 255)
    scores = np.zeros((rows, cols), np.int32)
    for y in range(3, rows - 3):
        for x in range(3, cols - 3):
            v = int(img[y, x])
            tab = []
            for k in range(25):
                p = int(img[y + CIRCLE[k % 16][1], x + CIRCLE[k % 16][0]])
                tab.append(1 if p < v - threshold else (2 if p > v + threshold else 0))
            is_corner = False
            for flag in (1, 2):
                count = 0
                for k in range(25):
                    if tab[k] == flag:
                        count += 1
                        if count > 8:
                            is_corner = True
                            break
                    else:
                        count = 0
                if is_corner:
                    break
            if is_corner:
                scores[y, x] = corner_score16(img, x, y, threshold) & 0xFF  # stored as uchar
    out = []
    for y in range(3, rows - 3):
        for x in range(3, cols - 3):
            s = scores[y, x]
            if s == 0:
                continue
            nb = scores[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, int(s)))
    return out


def block_image(rng, rows, cols, cell):
    g = rng.integers(20, 236, (rows // cell + 2, cols // cell + 2))
    img = np.kron(g, np.ones((cell, cell), np.int64))[:rows, :cols]
    img = img + rng.integers(-3, 4, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def gen_fast(rng):
    imgs, names = [], []
    # 1: isolated bright pixel blob on dark ground, 2: straight edge (no corner), 3: L corner
    a = np.full((32, 40), 40, np.uint8); a[15:18, 19:22] = 200
    b = np.full((32, 40), 40, np.uint8); b[:, 20:] = 180
    c = np.full((32, 40), 40, np.uint8); c[16:, 20:] = 180
    # 4: plateau: two equal-score neighbours must BOTH be suppressed (strict >)
    d = np.full((32, 40), 40, np.uint8); d[15, 19] = 220; d[15, 20] = 220
    # 5: exactly 8 contiguous brighter pixels (not a corner) vs 9 (corner)
    e8 = np.full((16, 16), 100, np.uint8); e9 = np.full((16, 16), 100, np.uint8)
    for k in range(8):
        e8[8 + CIRCLE[k][1], 8 + CIRCLE[k][0]] = 160
    for k in range(9):
        e9[8 + CIRCLE[k][1], 8 + CIRCLE[k][0]] = 160
    imgs += [a, b, c, d, e8, e9]
    names += ["blob", "edge", "lcorner", "plateau", "arc8", "arc9"]
    imgs.append(block_image(rng, 48, 64, 6)); names.append("blocks6")
    imgs.append(block_image(rng, 64, 96, 9)); names.append("blocks9")
    imgs.append(rng.integers(0, 256, (40, 56), dtype=np.uint8)); names.append("noise")
    out = {}
    for img, name in zip(imgs, names):
        out["img_" + name] = img
        for thr in (10, 20, 50):
            kp = fast9_16(img, thr)
            out["kp_%s_%d" % (name, thr)] = np.array(kp, np.int32).reshape(-1, 3)
    # ROI semantics: detection on a sub-rectangle uses the ROI's own 3 px border
    big = block_image(rng, 64, 96, 7)
    roi = (10, 7, 60, 40)
    sub = big[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]]
    out["img_roi"] = big
    out["roi"] = np.array(roi, np.int32)
    out["kp_roi_20"] = np.array(fast9_16(sub, 20), np.int32).reshape(-1, 3)
    np.savez_compressed(os.path.join(HERE, "fast.npz"), **out)


# ---------------------------------------------------------------------------------------------
def read_brief_pattern():
    txt = open(os.path.join(ROOT, "include", "vslam_brief_pattern.h")).read()
    nums = re.findall(r"\{(-?\d+),(-?\d+),(-?\d+),(-?\d+)\}", txt)
    assert len(nums) == 256
    return np.array(nums, np.int64)


def brief32(img, x, y, pat):
    im = img.astype(np.int64)

    def box(yy, xx):
        return int(im[yy - 4:yy + 5, xx - 4:xx + 5].sum())
    bits = [1 if box(y + p[0], x + p[1]) < box(y + p[2], x + p[3]) else 0 for p in pat]
    return np.packbits(np.array(bits, np.uint8))  # MSB first within each byte


def gen_brief(rng):
    pat = read_brief_pattern()
    img = block_image(rng, 96, 128, 5)
    pts = [(28, 28), (99, 67), (64, 48), (27, 40), (40, 27), (100, 50), (50, 68), (70, 33), (31, 60), (90, 29)]
    pts += [(int(rng.integers(28, 100)), int(rng.integers(28, 68))) for _ in range(22)]
    keep, desc = [], []
    for (x, y) in pts:
        inside = (28 <= x < 128 - 28) and (28 <= y < 96 - 28)
        keep.append(1 if inside else 0)
        desc.append(brief32(img, x, y, pat) if inside else np.zeros(32, np.uint8))
    np.savez_compressed(os.path.join(HERE, "brief.npz"), img=img, xy=np.array(pts, np.int16),
                        keep=np.array(keep, np.uint8), desc=np.array(desc, np.uint8))


# ---------------------------------------------------------------------------------------------
def gen_controller():
    # base_framepoint_generator.cpp:382-415,440-459 with the KITTI yaml values, one region
    tol, maxchg, tmin, tmax, target = 0.1, 0.1, 20, 100, 2158
    counts = [(5005, 4900), (4600, 4700), (3000, 3100), (2200, 2150), (2158, 2158), (1200, 1100), (100, 50),
              (0, 0), (2500, 1800), (9000, 9000), (9000, 9000), (9000, 9000), (2380, 2380), (1941, 1941)]
    counts += [(20000, 20000)] * 20 + [(10, 10)] * 25
    thr = tmin
    out = []
    for (cl, cr) in counts:
        acc = 0.0
        for c in (cl, cr):
            t = float(thr)
            delta = (float(c) - target) / target
            if delta < -tol:
                t = t + min(max(delta, -maxchg) * t, -1.0)
                t = max(t, float(tmin))
            elif delta > tol:
                t = t + max(min(delta, maxchg) * t, 1.0)
                t = min(t, float(tmax))
            acc += t
        thr = int(np.rint(acc / 2))  # std::rint: half to even, as numpy
        out.append(thr)
    np.savez_compressed(os.path.join(HERE, "controller.npz"), counts=np.array(counts, np.int32),
                        thresholds=np.array(out, np.int32), target=np.int32(target))


# ---------------------------------------------------------------------------------------------
KITTI_K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
KITTI_B = np.array([-386.1448, 0, 0.0])


def skew(p):
    return np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0.0]])


def v2t(v):
    q = np.array(v[3:6], float)
    n2 = q @ q
    if n2 < 1:
        w = np.sqrt(1 - n2)
    else:
        q = q / np.sqrt(n2)
        w = 0.0
    x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = v[0:3]
    return T


def linearize(T, moving, fixed, omega, weight, ignore_outliers, kernel=4.0, min_depth=0.1, rows=376, cols=1241):
    H = np.zeros((6, 6)); b = np.zeros(6); E = 0.0; ninl = 0
    n = len(moving)
    chi_out = -np.ones(n); inl = np.zeros(n, np.uint8)
    for u in range(n):
        p = T[:3, :3] @ moving[u] + T[:3, 3]
        if p[2] < min_depth:
            continue
        abcL = KITTI_K @ p
        abcR = abcL + KITTI_B
        uvL = abcL[:2] / abcL[2]
        uvR = abcR[:2] / abcR[2]
        if uvL[0] < 0 or uvL[0] > cols or uvL[1] < 0 or uvL[1] > rows:
            continue
        if uvR[0] < 0 or uvR[0] > cols or uvR[1] < 0 or uvR[1] > rows:
            continue
        e = np.concatenate([uvL, uvR]) - fixed[u]
        om = omega[u]
        chi = om * (e @ e)
        chi_out[u] = chi
        if chi > kernel:
            if ignore_outliers:
                continue
            om = om * kernel / chi
        else:
            inl[u] = 1
            ninl += 1
        E += chi
        Jt = np.hstack([weight[u] * np.eye(3), -2 * skew(p)])
        KJ = KITTI_K @ Jt
        JL = np.array([[1 / abcL[2], 0, -abcL[0] / abcL[2] ** 2], [0, 1 / abcL[2], -abcL[1] / abcL[2] ** 2]])
        JR = np.array([[1 / abcR[2], 0, -abcR[0] / abcR[2] ** 2], [0, 1 / abcR[2], -abcR[1] / abcR[2] ** 2]])
        J = np.vstack([JL @ KJ, JR @ KJ])
        H += om * (J.T @ J)
        b += om * (J.T @ e)
    return H, b, E, ninl, chi_out, inl


def converge(T, moving, fixed, omega, weight, damping=5.0, delta=1e-3, max_it=1000, min_inl=100):
    its = 0
    Eprev = 0.0

    def one_round(T, ignore):
        H, b, E, ninl, chi, inl = linearize(T, moving, fixed, omega, weight, ignore)
        H = H + damping * len(moving) * np.eye(6)
        dx = np.linalg.solve(H, -b)
        T = v2t(dx) @ T
        R = T[:3, :3]
        T[:3, :3] = R - 0.5 * R @ (R.T @ R - np.eye(3))
        return T, H, E, ninl, chi, inl
    for it in range(max_it):
        T, H, E, ninl, chi, inl = one_round(T, False); its += 1
        if delta > abs(Eprev - E):
            Eprev = E
            if ninl > min_inl and ninl > len(moving) - ninl:
                for it2 in range(max_it):
                    T, H, E, ninl, chi, inl = one_round(T, True); its += 1
                    conv = abs(Eprev - E) < delta
                    Eprev = E
                    if conv:
                        break
            break
        Eprev = E
    return T, H, E, ninl, chi, inl, its


def gen_aligner(rng):
    out = {}
    for name, n, noise, outlier_frac in (("m64_clean", 64, 0.0, 0.0), ("m512_noisy", 512, 0.4, 0.1),
                                         ("m300_pixel", 300, -1.0, 0.05)):
        # points in the previous camera frame, KITTI-like depth range
        X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-1.5, 1.6, n), rng.uniform(5, 45, n)], 1)
        # true motion previous->current: 0.9 m forward, small yaw
        vtrue = np.array([0.02, -0.01, -0.9, 0.001, 0.012, -0.0005])
        Ttrue = v2t(vtrue)
        P = (Ttrue[:3, :3] @ X.T).T + Ttrue[:3, 3]
        aL = (KITTI_K @ P.T).T
        aR = aL + KITTI_B
        fixed = np.hstack([aL[:, :2] / aL[:, 2:3], aR[:, :2] / aR[:, 2:3]])
        if noise > 0:
            fixed = fixed + rng.normal(0, noise, fixed.shape)
        elif noise < 0:
            fixed = np.rint(fixed)           # integer keypoints, as FAST delivers
        nout = int(outlier_frac * n)
        if nout:
            fixed[:nout] += rng.uniform(-30, 30, (nout, 4))
        # a few degenerate rows: behind the camera / projecting outside the image
        X[-1] = [0.0, 0.0, -3.0]
        X[-2] = [60.0, 0.0, 6.0]
        omega = np.where(rng.random(n) < 0.5, 1.0, 1.0 + np.log(rng.integers(2, 30, n)))
        weight = np.minimum(15.0 / P[:, 2], 1.0)
        T0 = np.eye(4)
        H, b, E, ninl, chi, inl = linearize(T0, X, fixed, omega, weight, False)
        Tc, Hc, Ec, ninlc, chic, inlc, its = converge(T0.copy(), X, fixed, omega, weight)
        out.update({name + "_moving": X, name + "_fixed": fixed, name + "_omega": omega, name + "_weight": weight,
                    name + "_H0": H, name + "_b0": b, name + "_E0": E, name + "_ninl0": np.int32(ninl),
                    name + "_chi0": chi, name + "_inl0": inl, name + "_T": Tc[:3, :], name + "_ninl": np.int32(ninlc),
                    name + "_E": Ec, name + "_its": np.int32(its), name + "_inl": inlc, name + "_Ttrue": Ttrue[:3, :]})
    np.savez_compressed(os.path.join(HERE, "aligner.npz"), **out)


# ---- UVDAligner (RGB-D mode, uvd_aligner.cpp) ---------------------------------------------------------------------
def linearize_uvd(T, moving, fixed, w_uv, w_d, weight, ignore_outliers, kernel=4.0, min_depth=0.1, rows=376, cols=1241):
    H = np.zeros((6, 6)); b = np.zeros(6); E = 0.0; ninl = 0
    n = len(moving)
    chi_out = -np.ones(n); inl = np.zeros(n, np.uint8)
    for u in range(n):
        p = T[:3, :3] @ moving[u] + T[:3, 3]
        if p[2] <= min_depth:
            continue
        a = KITTI_K @ p
        uv = a[:2] / a[2]
        if uv[0] < 0 or uv[0] > cols or uv[1] < 0 or uv[1] > rows:
            continue
        e = np.array([uv[0] - fixed[u, 0], uv[1] - fixed[u, 1], p[2] - fixed[u, 2]])
        Om = np.diag([w_uv[u], w_uv[u], w_d[u]])
        chi = e @ Om @ e
        chi_out[u] = chi
        if chi > kernel:
            if ignore_outliers:
                continue
            Om = Om * (kernel / chi)
        else:
            inl[u] = 1
            ninl += 1
        E += chi
        Jt = np.hstack([weight[u] * np.eye(3), -2 * skew(p)])
        iz = 1 / p[2]
        Jp = np.array([[iz, 0, -a[0] * iz * iz], [0, iz, -a[1] * iz * iz], [0, 0, 1.0]])
        J = Jp @ KITTI_K @ Jt
        H += J.T @ Om @ J
        b += J.T @ Om @ e
    return H, b, E, ninl, chi_out, inl


def converge_uvd(T, moving, fixed, w_uv, w_d, weight, damping=5.0, delta=1e-3, max_it=1000):
    its = 0
    Eprev = 0.0

    def one_round(T, ignore):
        H, b, E, ninl, chi, inl = linearize_uvd(T, moving, fixed, w_uv, w_d, weight, ignore)
        H = H + damping * len(moving) * np.eye(6)
        dx = np.linalg.solve(H, -b)
        T = v2t(dx) @ T
        R = T[:3, :3]
        T[:3, :3] = R - 0.5 * R @ (R.T @ R - np.eye(3))
        return T, H, E, ninl, chi, inl
    for it in range(max_it):
        T, H, E, ninl, chi, inl = one_round(T, False); its += 1
        if delta > abs(Eprev - E):
            Eprev = E
            if ninl > 100 and ninl > len(moving) - ninl:          # uvd_aligner.cpp:211
                for it2 in range(max_it):
                    T, H, E, ninl, chi, inl = one_round(T, True); its += 1
                    conv = abs(Eprev - E) < delta
                    Eprev = E
                    if conv:
                        break
            break
        Eprev = E
    return T, H, E, ninl, chi, inl, its


def gen_aligner_uvd(rng):
    out = {}
    for name, n, noise_px, noise_d, outlier_frac in (("m80_clean", 80, 0.0, 0.0, 0.0), ("m400_noisy", 400, 0.4, 0.02, 0.1)):
        X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-1.2, 1.2, n), rng.uniform(0.8, 8.0, n)], 1)
        vtrue = np.array([0.01, -0.005, -0.06, 0.002, 0.01, -0.001])
        Ttrue = v2t(vtrue)
        P = (Ttrue[:3, :3] @ X.T).T + Ttrue[:3, 3]
        a = (KITTI_K @ P.T).T
        fixed = np.hstack([a[:, :2] / a[:, 2:3], P[:, 2:3]])
        if noise_px > 0:
            fixed[:, :2] += rng.normal(0, noise_px, (n, 2))
            fixed[:, 2] += rng.normal(0, noise_d, n)
        nout = int(outlier_frac * n)
        if nout:
            fixed[:nout, :2] += rng.uniform(-25, 25, (nout, 2))
        X[-1] = [0.0, 0.0, -2.0]          # behind the camera
        X[-2] = [40.0, 0.0, 3.0]          # projects outside the image
        upd = rng.integers(0, 12, n)
        w_uv = np.where(rng.random(n) < 0.5, 1.0, 1.0 + upd)        # (1 + numberOfUpdates) for landmarks (:44)
        unreliable = rng.random(n) < 0.15
        w_d = np.where(unreliable, 0.0, 10.0 * w_uv)                # :52-61
        weight = np.where(unreliable, 0.0, np.minimum(5.0 / P[:, 2], 1.0))
        T0 = np.eye(4)
        H, b, E, ninl, chi, inl = linearize_uvd(T0, X, fixed, w_uv, w_d, weight, False)
        Tc, Hc, Ec, ninlc, chic, inlc, its = converge_uvd(T0.copy(), X, fixed, w_uv, w_d, weight)
        out.update({name + "_moving": X, name + "_fixed": fixed, name + "_w_uv": w_uv, name + "_w_d": w_d, name + "_weight": weight,
                    name + "_E0": E, name + "_ninl0": np.int32(ninl), name + "_T": Tc[:3, :], name + "_ninl": np.int32(ninlc),
                    name + "_E": Ec, name + "_its": np.int32(its), name + "_inl": inlc, name + "_chi": chic,
                    name + "_Ttrue": Ttrue[:3, :]})
    np.savez_compressed(os.path.join(HERE, "aligner_uvd.npz"), **out)


# ---------------------------------------------------------------------------------------------
def stereo_sweep(rcL, dL, rcR, dR, tau, min_disp, offsets):
    """compute() restated row by row (stereo_framepoint_generator.cpp:278-426): rows are independent,
    inside a row the right cursor only moves forward past accepted matches."""
    aliveL = np.ones(len(rcL), bool); aliveR = np.ones(len(rcR), bool)
    out = []
    for o in offsets:
        orderL = sorted([i for i in range(len(rcL)) if aliveL[i]], key=lambda i: (rcL[i][0], rcL[i][1]))
        orderR = sorted([i for i in range(len(rcR)) if aliveR[i]], key=lambda i: (rcR[i][0], rcR[i][1]))
        rows = sorted(set(rcL[i][0] for i in orderL))
        matched = []
        exhausted = False
        for r in rows:
            if exhausted:
                break
            Ls = [i for i in orderL if rcL[i][0] == r]
            Rs = [j for j in orderR if rcR[j][0] + o == r]
            start = 0
            for i in Ls:
                best, bj = tau, -1
                k = start
                while k < len(Rs):
                    j = Rs[k]
                    if rcL[i][1] - rcR[j][1] < 0:
                        break
                    d = hamming(dL[i], dR[j])
                    if d < best:
                        best, bj = d, k
                    k += 1
                if bj >= 0:
                    j = Rs[bj]
                    if rcL[i][1] - rcR[j][1] < min_disp:
                        continue
                    matched.append((i, j, best, o))
                    start = bj + 1
                    # the reference stops the whole sweep once the right list is exhausted
                    if start == len(Rs) and j == orderR[-1]:
                        exhausted = True
                        break
        for (i, j, d, oo) in matched:
            aliveL[i] = False; aliveR[j] = False
        out += matched
    return out


def gen_stereo(rng):
    cases = {}
    # hand-made: ties (first/lowest col wins), col_R > col_L stop, min-disparity continue w/o advance,
    # ordering constraint (cursor jumps behind the accepted right feature)
    base = rng.integers(0, 256, (8, 32), dtype=np.uint8)

    def near(d, k):
        x = d.copy()
        bits = np.unpackbits(x)
        bits[:k] ^= 1
        return np.packbits(bits)
    rcL = [(10, 100), (10, 140), (10, 180), (12, 50), (12, 60), (20, 300), (20, 301), (25, 90)]
    dL = [base[0], base[1], base[2], base[3], base[3], base[4], base[5], base[6]]
    rcR = [(10, 60), (10, 90), (10, 139), (10, 150), (12, 50), (12, 55), (20, 250), (20, 300), (20, 400), (25, 95), (30, 5)]
    dR = [near(base[0], 5), near(base[0], 5), near(base[1], 3), near(base[2], 2), near(base[3], 1), near(base[3], 1),
          near(base[4], 4), near(base[5], 4), base[5], base[6], base[7]]
    cases["hand"] = (rcL, dL, rcR, dR)
    # random: 40 rows, a few features per row, right = left shifted by a disparity with bit noise
    rcL, dL, rcR, dR = [], [], [], []
    for r in range(30, 70):
        cols = sorted(set(int(c) for c in rng.integers(40, 600, rng.integers(1, 7))))
        for c in cols:
            d = rng.integers(0, 256, 32, dtype=np.uint8)
            rcL.append((r, c)); dL.append(d)
            if rng.random() < 0.8:
                disp = int(rng.integers(0, 35))
                rr = r + (int(rng.integers(-1, 2)) if rng.random() < 0.3 else 0)
                rcR.append((rr, c - disp)); dR.append(near(d, int(rng.integers(0, 40))))
        for _ in range(int(rng.integers(0, 3))):
            rcR.append((r, int(rng.integers(0, 640)))); dR.append(rng.integers(0, 256, 32, dtype=np.uint8))
    # unique pixels only (a pixel holds one feature after FAST NMS)
    seen, keep = set(), []
    for k, p in enumerate(rcR):
        if p not in seen:
            seen.add(p); keep.append(k)
    rcR = [rcR[k] for k in keep]; dR = [dR[k] for k in keep]
    cases["random"] = (rcL, dL, rcR, dR)
    out = {}
    for name, (rcL, dL, rcR, dR) in cases.items():
        for epi in (0, 1):
            offsets = [0] + [s * u for u in range(1, epi + 1) for s in (1, -1)]
            m = stereo_sweep(rcL, dL, rcR, dR, 25.6 if name == "hand" else 30.0, 1.0, offsets)
            out["%s_epi%d_matches" % (name, epi)] = np.array(m, np.int32).reshape(-1, 4)
        out[name + "_rcL"] = np.array(rcL, np.int32); out[name + "_dL"] = np.array(dL, np.uint8)
        out[name + "_rcR"] = np.array(rcR, np.int32); out[name + "_dR"] = np.array(dR, np.uint8)
    out["hand_tau"] = np.float64(25.6); out["random_tau"] = np.float64(30.0)
    np.savez_compressed(os.path.join(HERE, "stereo.npz"), **out)


# ---- temporal tracking (StereoFramePointGenerator::track + getMatchingFeatureInRectangularRegion) --------------------
def _trunc32(v):
    """C++ double -> int32 conversion (toward zero); None outside the int32 range / NaN."""
    if not (v > -2147483648.0 and v < 2147483648.0):
        return None
    return int(v)


def _match_in_region(lattice, feats, row_ref, col_ref, desc_ref, r0, r1, c0, c1, max_dist, by_appearance):
    """intensity_feature_matcher.cpp:81-148 on a dict lattice {(row, col): id}; returns (id or -1, distance)."""
    best, dist_best = -1, max_dist
    pix_best = 10000
    for r in range(r0, r1):
        for c in range(c0, c1):
            fid = lattice.get((r, c), -1)
            if fid < 0:
                continue
            dd = float(hamming(desc_ref, feats[fid][2]))
            if by_appearance:
                if dd < dist_best:
                    dist_best, best = dd, fid
            elif dd < max_dist:
                pix = (row_ref - r) ** 2 + (col_ref - c) ** 2
                if pix < pix_best:
                    pix_best, dist_best, best = pix, dd, fid
    return best, dist_best


def track_ref(K, bh, rows, cols, T, prev, featsL, featsR, d, tau_track, tau_tri, by_appearance, min_disp):
    """stereo_framepoint_generator.cpp:464-681, independent of the C++ oracle.
    prev: list of (cam xyz, descL, descR, epipolar offset); feats*: list of (row, col, desc).
    Returns (tracked [(prev, fl, fr, dist)], lost [prev indices])."""
    latL = {}
    for i, f in enumerate(featsL):
        latL[(f[0], f[1])] = i
    latR = {}
    for i, f in enumerate(featsR):
        latR[(f[0], f[1])] = i
    tracked, lost = [], []
    R, t = np.asarray(T[:, :3], np.float64), np.asarray(T[:, 3], np.float64)
    for ip, (cam, pdL, pdR, epi) in enumerate(prev):
        q = np.array([(R[i, 0] * cam[0] + R[i, 1] * cam[1]) + R[i, 2] * cam[2] + t[i] for i in range(3)])
        uvw = np.array([(K[i, 0] * q[0] + K[i, 1] * q[1]) + K[i, 2] * q[2] for i in range(3)])
        if not (uvw[2] > 0):        # the oracle's documented deviation B.5 (the reference divides regardless)
            continue
        col, row = _trunc32(uvw[0] / uvw[2]), _trunc32(uvw[1] / uvw[2])
        if col is None or row is None or col < 0 or col > cols or row < 0 or row > rows:
            continue
        has_next = False
        r0, r1 = max(row - d, 0), min(row + d + 1, rows)
        c0, c1 = max(col - d, 0), min(col + d + 1, cols)
        fl, _ = _match_in_region(latL, featsL, row, col, pdL, r0, r1, c0, c1, tau_track, by_appearance)
        if fl >= 0:
            FL = featsL[fl]
            ex = np.float32(col) - np.float32(FL[1])
            ey = np.float32(row) - np.float32(FL[0])
            uR = uvw + bh
            colR, rowR = _trunc32(uR[0] / uR[2] - float(ex)), _trunc32(uR[1] / uR[2] - float(ey))
            if colR is None or rowR is None or colR < 0 or colR > cols or rowR < 0 or rowR > rows:
                continue                                # :561 — skips the lost list too
            k = int(abs(float(epi)))
            rr0, rr1 = max(rowR - k, 0), min(rowR + k + 1, rows)
            rc0, rc1 = max(colR - d, 0), min(colR + d + 1, FL[1])
            fr, dist = _match_in_region(latR, featsR, rowR, colR, FL[2], rr0, rr1, rc0, rc1, tau_tri, True)
            if fr >= 0:
                FR = featsR[fr]
                if FL[1] - FR[1] < min_disp:
                    continue                            # :599
                if float(hamming(FR[2], pdR)) > tau_track:
                    continue                            # :607
                for c in range(FR[1] + 1, FL[1]):       # parallax clearing :612-621
                    latR.pop((FR[0], c), None)
                tracked.append((ip, fl, fr, int(dist)))
                has_next = True
                latL.pop((FL[0], FL[1]), None)
                latR.pop((FR[0], FR[1]), None)
        if not has_next:
            lost.append(ip)
    return tracked, lost


def gen_track(rng):
    """Small images packed with features so that order-dependent conflicts, parallax clearing, both search modes,
    the truncating projections and the inclusive image gate all occur."""
    rows, cols = 96, 160
    f, cx, cy, base = 120.0, 80.0, 48.0, 0.4
    K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1.0]])
    bh = np.array([-f * base, 0.0, 0.0])
    out = {"K": K, "bh": bh, "rows": np.int32(rows), "cols": np.int32(cols)}

    def near(dsc, k):
        bits = np.unpackbits(dsc)
        idx = rng.choice(256, size=k, replace=False)
        bits[idx] ^= 1
        return np.packbits(bits)
    n_case = 0
    for by_app in (1, 0):
        for d in (3, 9):
            for trial in range(3):
                T = np.eye(4)[:3]
                T = T.copy()
                T[:, 3] = rng.normal(0, 0.02, 3)
                ang = rng.normal(0, 0.01)
                T[0, 0], T[0, 2], T[2, 0], T[2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
                prev, featsL, featsR = [], [], []
                usedL, usedR = set(), set()
                nP = 70
                for ip in range(nP):
                    z = float(rng.uniform(2.0, 25.0))
                    u, v = float(rng.uniform(-6, cols + 6)), float(rng.uniform(-6, rows + 6))
                    cam = np.array([(u - cx) * z / f, (v - cy) * z / f, z])
                    dLp = rng.integers(0, 256, 32, dtype=np.uint8)
                    dRp = near(dLp, int(rng.integers(0, 20)))
                    prev.append((cam, dLp, dRp, int(rng.integers(-1, 2))))
                    # a few current features around the projection, left and right
                    for _ in range(int(rng.integers(0, 4))):
                        r = int(round(v)) + int(rng.integers(-d - 1, d + 2)); c = int(round(u)) + int(rng.integers(-d - 1, d + 2))
                        if 0 <= r < rows and 0 <= c < cols and (r, c) not in usedL:
                            usedL.add((r, c)); featsL.append((r, c, near(dLp, int(rng.integers(0, 45)))))
                            disp = f * base / z
                            for _ in range(int(rng.integers(0, 3))):
                                rr = r + int(rng.integers(-1, 2)); cc = int(round(c - disp)) + int(rng.integers(-3, 4))
                                if 0 <= rr < rows and 0 <= cc < cols and (rr, cc) not in usedR:
                                    usedR.add((rr, cc)); featsR.append((rr, cc, near(featsL[-1][2], int(rng.integers(0, 35)))))
                # clutter
                for _ in range(60):
                    r, c = int(rng.integers(0, rows)), int(rng.integers(0, cols))
                    if (r, c) not in usedL:
                        usedL.add((r, c)); featsL.append((r, c, rng.integers(0, 256, 32, dtype=np.uint8)))
                    r, c = int(rng.integers(0, rows)), int(rng.integers(0, cols))
                    if (r, c) not in usedR:
                        usedR.add((r, c)); featsR.append((r, c, rng.integers(0, 256, 32, dtype=np.uint8)))
                tau_track, tau_tri = 50.0, 40.0
                tr, lost = track_ref(K, bh, rows, cols, T, prev, featsL, featsR, d, tau_track, tau_tri, bool(by_app), 1.0)
                key = "c%d_" % n_case
                out[key + "T"] = T; out[key + "d"] = np.int32(d); out[key + "by_app"] = np.int32(by_app)
                out[key + "tau_track"] = np.float64(tau_track); out[key + "tau_tri"] = np.float64(tau_tri)
                out[key + "cam"] = np.array([p[0] for p in prev]); out[key + "pdL"] = np.array([p[1] for p in prev], np.uint8)
                out[key + "pdR"] = np.array([p[2] for p in prev], np.uint8); out[key + "epi"] = np.array([p[3] for p in prev], np.int32)
                out[key + "rcL"] = np.array([(a[0], a[1]) for a in featsL], np.int32); out[key + "dL"] = np.array([a[2] for a in featsL], np.uint8)
                out[key + "rcR"] = np.array([(a[0], a[1]) for a in featsR], np.int32); out[key + "dR"] = np.array([a[2] for a in featsR], np.uint8)
                out[key + "tracked"] = np.array(tr, np.int32).reshape(-1, 4); out[key + "lost"] = np.array(lost, np.int32)
                n_case += 1
    out["n_cases"] = np.int32(n_case)
    np.savez_compressed(os.path.join(HERE, "track.npz"), **out)
    return n_case


# ---------------------------------------------------------------------------------------------
# RGB-D components (DepthFramePointGenerator): pure-Python loops over small images, numpy scalars for the float / double
# distinction of the z-buffer test, numpy.linalg.lstsq (SVD) for the midpoint triangulation.
def _c_round(x):
    return int(np.floor(x + 0.5)) if x >= 0 else -int(np.floor(-x + 0.5))


def depth_space_map_ref(depth, Kl, Kri, r2l, scale, max_depth):
    rows, cols = depth.shape
    space = np.zeros((rows, cols, 3), np.float32)
    space[:, :, 2] = np.float32(max_depth)
    rmap = -np.ones((rows, cols), np.int16); cmap = -np.ones((rows, cols), np.int16)
    R = r2l[:, :3]; t = r2l[:, 3]
    for r in range(rows):
        for c in range(cols):
            raw = int(depth[r, c])
            if raw == 0:
                continue
            dm = raw * scale
            ph = np.array([c * dm, r * dm, dm])
            pr = np.array([(Kri[i, 0] * ph[0] + Kri[i, 1] * ph[1]) + Kri[i, 2] * ph[2] for i in range(3)])
            pl = np.array([((R[i, 0] * pr[0] + R[i, 1] * pr[1]) + R[i, 2] * pr[2]) + t[i] for i in range(3)])
            if pl[2] <= 0:
                continue
            px = np.array([(Kl[i, 0] * pl[0] + Kl[i, 1] * pl[1]) + Kl[i, 2] * pl[2] for i in range(3)])
            dr, dc = _c_round(px[1] / px[2]), _c_round(px[0] / px[2])
            if dr < 0 or dr >= rows or dc < 0 or dc >= cols:
                continue
            if float(space[dr, dc, 2]) > pl[2]:
                space[dr, dc] = pl.astype(np.float32)
                rmap[dr, dc] = r; cmap[dr, dc] = c
    return space, rmap, cmap


def depth_compute_ref(space, feats, tracked, Kli, min_depth, max_depth, triangulate, binning, bin_px):
    bins = {}
    if binning:
        for (row, col) in tracked:
            bins[(round(row / bin_px), round(col / bin_px))] = "tracked"   # Python round == rint (half to even)
    fresh, temp, depth_of = [], [], {}
    for i, (row, col) in enumerate(feats):
        d = space[row, col]
        if float(d[2]) < min_depth:
            continue
        if float(d[2]) >= max_depth and triangulate:
            ph = np.array([col * max_depth, row * max_depth, max_depth])
            temp.append((i, [(Kli[k, 0] * ph[0] + Kli[k, 1] * ph[1]) + Kli[k, 2] * ph[2] for k in range(3)]))
            continue
        fresh.append(i); depth_of[i] = float(d[2])
        if binning:
            key = (round(row / bin_px), round(col / bin_px))
            if key in bins:
                if bins[key] != "tracked" and depth_of[i] < depth_of[bins[key]]:
                    bins[key] = i
            else:
                bins[key] = i
    if binning:
        rows_bin, cols_bin = space.shape[0] // bin_px + 1, space.shape[1] // bin_px + 1
        out = [bins[(rb, cb)] for rb in range(rows_bin) for cb in range(cols_bin) if (rb, cb) in bins and bins[(rb, cb)] != "tracked"]
    else:
        out = fresh
    return out, temp


def point_in_camera_ref(xp, xc, T, K):
    a0, b0 = (float(xp[0]) - K[0, 2]) / K[0, 0], (float(xp[1]) - K[1, 2]) / K[1, 1]
    a1, b1 = (float(xc[0]) - K[0, 2]) / K[0, 0], (float(xc[1]) - K[1, 2]) / K[1, 1]
    x0 = np.array([a0, b0, 1.0]); x1 = np.array([a1, b1, 1.0])
    A = np.stack([-T[:, :3] @ x0, x1], axis=1)
    z = np.linalg.lstsq(A, T[:, 3], rcond=None)[0]
    return (x1 * z[1] + (T[:, :3] @ (x0 * z[0]) + T[:, 3])) / 2.0


def depth_track_ref(K, rows, cols, space, T, prev, feats, d, tau, by_appearance, min_depth, max_depth, triangulate):
    """depth_framepoint_generator.cpp:166-287, independent of the C++ oracle.  prev: (cam xyz, desc, has_landmark,
    unreliable_depth); feats: (row, col, desc).  Returns tracked [(prev, feature)], temporary [(prev, feature)], lost, landmarks."""
    lat = {(f[0], f[1]): i for i, f in enumerate(feats)}
    tracked, temp, lost, n_lm = [], [], [], 0
    R, t = np.asarray(T[:, :3], np.float64), np.asarray(T[:, 3], np.float64)
    for ip, (cam, pd, has_lm, unreliable) in enumerate(prev):
        q = np.array([(R[i, 0] * cam[0] + R[i, 1] * cam[1]) + R[i, 2] * cam[2] + t[i] for i in range(3)])
        uvw = np.array([(K[i, 0] * q[0] + K[i, 1] * q[1]) + K[i, 2] * q[2] for i in range(3)])
        if not (uvw[2] > 0):        # same documented deviation as the stereo track (B.5)
            continue
        col, row = _trunc32(uvw[0] / uvw[2]), _trunc32(uvw[1] / uvw[2])
        if col is None or row is None or col < 0 or col > cols or row < 0 or row > rows:
            continue
        r0, r1 = max(row - d, 0), min(row + d + 1, rows)
        c0, c1 = max(col - d, 0), min(col + d + 1, cols)
        f, _ = _match_in_region(lat, feats, row, col, pd, r0, r1, c0, c1, tau, by_appearance)
        has_next = False
        if f >= 0:
            z = float(space[feats[f][0], feats[f][1], 2])
            if z < min_depth:
                continue
            lat.pop((feats[f][0], feats[f][1]), None)
            if z >= max_depth and triangulate:
                temp.append((ip, f))
                continue
            tracked.append((ip, f))
            has_next = True
            if has_lm:
                n_lm += 1
        if not has_next and not unreliable:
            lost.append(ip)
    return tracked, temp, lost, n_lm


def depth_track_space(zmap, cx, cy, f):
    """space map of the track fixture from its depth channel (the .npz stores only z; tests rebuild x, y the same way)"""
    rows, cols = zmap.shape
    space = np.zeros((rows, cols, 3), np.float32)
    space[:, :, 2] = zmap
    space[:, :, 0] = ((np.arange(cols)[None, :] - cx) * zmap / f).astype(np.float32)
    space[:, :, 1] = ((np.arange(rows)[:, None] - cy) * zmap / f).astype(np.float32)
    return space


def gen_depth_track(rng):
    """Crowded small images: several previous points compete for the same feature (order decides), features on pixels
    without depth (temporary points), below the minimum depth (skipped without consuming the feature), both search modes."""
    rows, cols = 72, 96
    f_, cx, cy = 90.0, 48.0, 36.0
    K = np.array([[f_, 0, cx], [0, f_, cy], [0, 0, 1.0]])
    out = {"K": K, "rows": np.int32(rows), "cols": np.int32(cols)}

    def near(dsc, k):
        bits = np.unpackbits(dsc)
        bits[rng.choice(256, size=k, replace=False)] ^= 1
        return np.packbits(bits)
    n_case = 0
    for by_app in (1, 0):
        for d in (2, 7):
            for tri in (1, 0):
                T = np.eye(4)[:3].copy()
                T[:, 3] = rng.normal(0, 0.02, 3)
                ang = rng.normal(0, 0.01)
                T[0, 0], T[0, 2], T[2, 0], T[2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
                zmap = (np.round(rng.uniform(0.5, 6.0, (rows, cols)) * 16) / 16).astype(np.float32)   # coarse: the fixture compresses
                zmap[rng.random((rows, cols)) < 0.2] = np.float32(10.0)        # no measurement: maximum depth
                zmap[rng.random((rows, cols)) < 0.1] = np.float32(0.05)        # below the minimum depth
                space = depth_track_space(zmap, cx, cy, f_)
                prev, feats, used = [], [], set()
                for ip in range(90):
                    z = float(rng.uniform(0.8, 6.0))
                    u, v = float(rng.uniform(-4, cols + 4)), float(rng.uniform(-4, rows + 4))
                    cam = np.array([(u - cx) * z / f_, (v - cy) * z / f_, z])
                    dp = rng.integers(0, 256, 32, dtype=np.uint8)
                    prev.append((cam, dp, int(rng.random() < 0.5), int(rng.random() < 0.15)))
                    for _ in range(int(rng.integers(0, 3))):
                        r = int(round(v)) + int(rng.integers(-d - 1, d + 2)); c = int(round(u)) + int(rng.integers(-d - 1, d + 2))
                        if 0 <= r < rows and 0 <= c < cols and (r, c) not in used:
                            used.add((r, c)); feats.append((r, c, near(dp, int(rng.integers(0, 40)))))
                    if ip % 5 == 4:      # a rival: a second previous point with nearly the same appearance and position
                        prev.append((cam + rng.normal(0, 0.01, 3), near(dp, 3), 1, 0))
                for _ in range(50):
                    r, c = int(rng.integers(0, rows)), int(rng.integers(0, cols))
                    if (r, c) not in used:
                        used.add((r, c)); feats.append((r, c, rng.integers(0, 256, 32, dtype=np.uint8)))
                tau = 35.0
                tr, tmp, lost, nlm = depth_track_ref(K, rows, cols, space, T, prev, feats, d, tau, bool(by_app), 0.1, 10.0, tri)
                key = "c%d_" % n_case
                out[key + "T"] = T; out[key + "d"] = np.int32(d); out[key + "by_app"] = np.int32(by_app); out[key + "tri"] = np.int32(tri)
                out[key + "tau"] = np.float64(tau); out[key + "zmap"] = zmap
                out[key + "cam"] = np.array([q[0] for q in prev]); out[key + "pd"] = np.array([q[1] for q in prev], np.uint8)
                out[key + "flags"] = np.array([q[2] | (q[3] << 1) for q in prev], np.uint8)
                out[key + "rc"] = np.array([(a[0], a[1]) for a in feats], np.int32); out[key + "desc"] = np.array([a[2] for a in feats], np.uint8)
                out[key + "tracked"] = np.array(tr, np.int32).reshape(-1, 2); out[key + "temp"] = np.array(tmp, np.int32).reshape(-1, 2)
                out[key + "lost"] = np.array(lost, np.int32); out[key + "n_lm"] = np.int32(nlm)
                n_case += 1
    out["n_cases"] = np.int32(n_case)
    np.savez_compressed(os.path.join(HERE, "depth_track.npz"), **out)
    return n_case


def depth_recover_ref(K, rows, cols, space, img, pat, w2c, lost, kp_size, tau, min_depth, max_depth):
    """depth_framepoint_generator.cpp:289-407; lost: (has_landmark, world xyz, previous descriptor).  Floats as np.float32."""
    out = []
    R, t = w2c[:, :3], w2c[:, 3]
    f32 = np.float32
    for i, (has_lm, X, pd) in enumerate(lost):
        if not has_lm:
            continue
        pc = np.array([((R[k, 0] * X[0] + R[k, 1] * X[1]) + R[k, 2] * X[2]) + t[k] for k in range(3)])
        pi = np.array([(K[k, 0] * pc[0] + K[k, 1] * pc[1]) + K[k, 2] * pc[2] for k in range(3)])
        with np.errstate(divide="ignore", invalid="ignore"):
            x, y = pi[0] / pi[2], pi[1] / pi[2]
        if not (x >= 0 and x <= cols and y >= 0 and y <= rows):
            continue
        px, py = f32(x), f32(y)
        fr, fc = int(round(float(py))), int(round(float(px)))       # rint: half to even
        if not (0 <= fr < rows and 0 <= fc < cols):
            continue
        z = float(space[fr, fc, 2])
        if z < min_depth or z >= max_depth:
            continue
        rbc = f32(5) * f32(kp_size)
        if px <= rbc + f32(1) or px >= f32(cols) - rbc - f32(1) or py <= rbc + f32(1) or py >= f32(rows) - rbc - f32(1):
            continue
        cxf, cyf = f32(px - rbc), f32(py - rbc)
        bx, by = int(round(float(cxf))) + int(float(rbc) + 0.5), int(round(float(cyf))) + int(float(rbc) + 0.5)
        d = brief32(img, bx, by, pat)
        if hamming(pd, d) > tau:
            continue
        out.append((i, f32(rbc + cxf), f32(rbc + cyf), d, space[fr, fc].astype(np.float64)))
    return out


def gen_depth_recover(rng):
    pat = read_brief_pattern()
    rows, cols = 150, 200
    f_, cx, cy = 160.0, 99.5, 74.5
    K = np.array([[f_, 0, cx], [0, f_, cy], [0, 0, 1.0]])
    img = block_image(rng, rows, cols, 4)
    zmap = (np.round(rng.uniform(0.5, 6.0, (rows, cols)) * 16) / 16).astype(np.float32)
    zmap[rng.random((rows, cols)) < 0.15] = np.float32(10.0)
    zmap[rng.random((rows, cols)) < 0.05] = np.float32(0.05)
    space = depth_track_space(zmap, cx, cy, f_)
    ang = 0.02
    w2c = np.hstack([np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]), np.array([[0.03], [-0.02], [0.05]])])
    c2w_R, c2w_t = w2c[:, :3].T, -w2c[:, :3].T @ w2c[:, 3]
    lost = []
    for i in range(140):
        z = float(rng.uniform(0.8, 6.0))
        # sub-pixel projections all over the image, some in the border band, some outside; a few exactly on .5
        u = float(rng.uniform(25, cols - 25)); v = float(rng.uniform(25, rows - 25))
        if i % 10 == 3:
            u = float(rng.uniform(-5, cols + 5)); v = float(rng.uniform(-5, rows + 5))
        if i % 9 == 0:
            u, v = np.floor(u) + 0.5, np.floor(v) + 0.5
        pc = np.array([(u - cx) * z / f_, (v - cy) * z / f_, z])
        X = c2w_R @ pc + c2w_t
        has_lm = int(rng.random() < 0.85)
        # the previous descriptor: BRIEF near the projection with some bits flipped (or random clutter)
        bx, by = int(np.clip(round(u), 28, cols - 29)), int(np.clip(round(v), 28, rows - 29))
        base = brief32(img, bx, by, pat)
        bits = np.unpackbits(base)
        bits[rng.choice(256, size=int(rng.integers(0, 50)), replace=False)] ^= 1
        lost.append((has_lm, X, np.packbits(bits)))
    tau = 35.0
    rec = depth_recover_ref(K, rows, cols, space, img, pat, w2c, lost, 7.0, tau, 0.1, 10.0)
    out = {"K": K, "img": img, "zmap": zmap, "w2c": w2c, "tau": np.float64(tau),
           "has_lm": np.array([q[0] for q in lost], np.uint8), "lm": np.array([q[1] for q in lost]),
           "pd": np.array([q[2] for q in lost], np.uint8),
           "rec_index": np.array([r[0] for r in rec], np.int32), "rec_xy": np.array([[r[1], r[2]] for r in rec], np.float32).reshape(-1, 2),
           "rec_desc": np.array([r[3] for r in rec], np.uint8).reshape(-1, 32), "rec_xyz": np.array([r[4] for r in rec], np.float64).reshape(-1, 3)}
    np.savez_compressed(os.path.join(HERE, "depth_recover.npz"), **out)
    return len(rec)


def gen_depth(rng):
    out = {}
    rows, cols = 40, 56
    Kr = np.array([[60.0, 0, 27.5], [0, 60.0, 19.5], [0, 0, 1]])
    cases = {
        # registered RGB-D (identity, same K): every pixel maps onto itself
        "registered": (Kr.copy(), np.hstack([np.eye(3), np.zeros((3, 1))])),
        # the left camera sees the scene at 0.55x: ~3.3 depth pixels land on one cell, equal raw depths collide
        "shrunk": (np.array([[33.0, 0, 27.5], [0, 33.0, 19.5], [0, 0, 1]]), np.hstack([np.eye(3), np.zeros((3, 1))])),
        # small rotation + baseline: general forward warp with holes and occlusions
        "offset": (np.array([[58.0, 0, 28.0], [0, 58.0, 19.0], [0, 0, 1]]), None),
    }
    ang = 0.03
    Rz = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    cases["offset"] = (cases["offset"][0], np.hstack([Rz, np.array([[0.05], [-0.01], [0.02]])]))
    for name, (Kl, r2l) in cases.items():
        depth = rng.integers(400, 9000, (rows, cols)).astype(np.uint16)
        depth[rng.random((rows, cols)) < 0.15] = 0                         # holes
        depth[rng.random((rows, cols)) < 0.05] = 12000                     # beyond maximum_depth (10 m)
        depth[10:14, 20:30] = 2500                                         # a fronto-parallel patch: exact depth ties
        depth[20:23, 5:15] = 1001                                          # 1.001 m: float(z) != z, both rounding directions
        depth[25:28, 30:40] = 1003
        depth[0, 0] = 50                                                   # below minimum_depth
        Kri = np.linalg.inv(Kr)
        space, rmap, cmap = depth_space_map_ref(depth, Kl, Kri, r2l, 1e-3, 10.0)
        out[name + "_depth"] = depth; out[name + "_Kl"] = Kl; out[name + "_Kri"] = Kri; out[name + "_r2l"] = r2l
        out[name + "_space"] = space; out[name + "_rmap"] = rmap; out[name + "_cmap"] = cmap
        # features: a random subset of pixels in row-major order; tracked points: a few more
        flat = np.sort(rng.choice(rows * cols, 300, replace=False))
        feats = [(int(v // cols), int(v % cols)) for v in flat[:260]]
        tracked = [(int(v // cols), int(v % cols)) for v in flat[260:]]
        Kli = np.linalg.inv(Kl)
        out[name + "_feats"] = np.array(feats, np.int32); out[name + "_tracked"] = np.array(tracked, np.int32); out[name + "_Kli"] = Kli
        for tag, (tri, binning) in {"bin_tri": (1, 1), "nobin_tri": (1, 0), "bin_notri": (0, 1)}.items():
            new, temp = depth_compute_ref(space, feats, tracked, Kli, 0.1, 10.0, tri, binning, 6)
            out["%s_%s_new" % (name, tag)] = np.array(new, np.int32)
            out["%s_%s_temp" % (name, tag)] = np.array([i for i, _ in temp], np.int32)
            out["%s_%s_temp_xyz" % (name, tag)] = np.array([x for _, x in temp], np.float64).reshape(-1, 3)
    # midpoint triangulation: points seen from two poses (well-conditioned), plus near-degenerate pairs
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
    n = 64
    v = np.array([0.08, -0.03, 0.05, 0.01, -0.02, 0.015])
    T = v2t(v)[:3, :]
    P = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1.5, 8, n)], axis=1)
    Pc = (T[:, :3] @ P.T).T + T[:, 3]
    xp = (K @ P.T).T; xp = (xp[:, :2] / xp[:, 2:]).astype(np.float32)
    xc = (K @ Pc.T).T; xc = (xc[:, :2] / xc[:, 2:] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    out["tri_K"] = K; out["tri_T"] = T; out["tri_xp"] = xp; out["tri_xc"] = xc
    out["tri_xyz"] = np.array([point_in_camera_ref(xp[i], xc[i], T, K) for i in range(n)])
    np.savez_compressed(os.path.join(HERE, "depth.npz"), **out)
    return len(cases)


# ---------------------------------------------------------------------------------------------
# OrbDetector components: cv::resize INTER_LINEAR (8UC1), orb.cpp HarrisResponses / ICAngles / computeKeyPoints [recalled],
# with numpy arrays where the C++ has loops and numpy float32 scalars for the float arithmetic.
def resize_linear_ref(src, drows, dcols):
    rows, cols = src.shape
    S = src.astype(np.int64)

    def coeffs(n_dst, n_src):
        ofs, a0, a1, nmax = [], [], [], n_dst
        sc = n_src / n_dst
        for d in range(n_dst):
            f = np.float32((d + 0.5) * sc - 0.5)
            s0 = int(np.floor(f))
            f = np.float32(f - np.float32(s0))
            ofs.append(s0); a0.append(int(np.rint(np.float32(np.float32(1) - f) * np.float32(2048)))); a1.append(int(np.rint(f * np.float32(2048))))
        return ofs, a0, a1
    xo, xa0, xa1 = coeffs(dcols, cols)
    yo, yb0, yb1 = coeffs(drows, rows)
    # horizontal pass for every source row (the left edge clamps the offset with weight 1 on the first pixel, the right
    # edge replicates the last pixel with the full weight)
    H = np.zeros((rows, dcols), np.int64)
    for dx in range(dcols):
        sx, a0, a1 = xo[dx], xa0[dx], xa1[dx]
        if sx < 0:
            H[:, dx] = S[:, 0] * 2048 + S[:, 1] * 0
        elif sx >= cols - 1:
            H[:, dx] = S[:, cols - 1] * 2048
        else:
            H[:, dx] = S[:, sx] * a0 + S[:, sx + 1] * a1
    out = np.zeros((drows, dcols), np.uint8)
    for dy in range(drows):
        r0 = H[min(max(yo[dy], 0), rows - 1)]; r1 = H[min(max(yo[dy] + 1, 0), rows - 1)]
        out[dy] = ((((yb0[dy] * (r0 >> 4)) >> 16) + ((yb1[dy] * (r1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    return out


def fast_atan2_ref(y, x):
    f = np.float32
    s = f(180.0 / np.pi)
    p1, p3, p5, p7 = f(0.9997878412794807) * s, f(-0.3258083974640975) * s, f(0.1555786518463281) * s, f(-0.04432655554792128) * s
    ax, ay = f(abs(x)), f(abs(y))
    eps = f(2.220446049250313e-16)
    if ax >= ay:
        c = ay / (ax + eps); c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps); c2 = c * c
        a = f(90) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = f(180) - a
    if y < 0:
        a = f(360) - a
    return f(a)


def orb_umax_ref(half):
    umax = [0] * (half + 2)
    vmax = int(np.floor(np.float32(half) * np.sqrt(np.float32(2)) / np.float32(2) + np.float32(1)))
    vmin = int(np.ceil(np.float32(half) * np.sqrt(np.float32(2)) / np.float32(2)))
    for v in range(vmax + 1):
        umax[v] = int(np.rint(np.sqrt(float(half * half - v * v))))
    v0 = 0
    for v in range(half, vmin - 1, -1):
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
    return umax


def harris_ref(img, x0, y0):
    I = img.astype(np.int64)
    P = I[y0 - 4:y0 + 5, x0 - 4:x0 + 5]      # 9x9 around the 7x7 block
    Ix = (P[1:-1, 2:] - P[1:-1, :-2]) * 2 + (P[:-2, 2:] - P[:-2, :-2]) + (P[2:, 2:] - P[2:, :-2])
    Iy = (P[2:, 1:-1] - P[:-2, 1:-1]) * 2 + (P[2:, :-2] - P[:-2, :-2]) + (P[2:, 2:] - P[:-2, 2:])
    a, b, c = int((Ix * Ix).sum()), int((Iy * Iy).sum()), int((Ix * Iy).sum())
    f = np.float32
    scale = f(1) / (f(4 * 7) * f(255))
    ssq = scale * scale * scale * scale
    return f((f(a) * f(b) - f(c) * f(c) - f(0.04) * (f(a) + f(b)) * (f(a) + f(b))) * ssq)


def ic_angle_ref(img, x0, y0, half, umax):
    I = img.astype(np.int64)
    m10 = sum(u * int(I[y0, x0 + u]) for u in range(-half, half + 1))
    m01 = 0
    for v in range(1, half + 1):
        d = umax[v]
        vp = I[y0 + v, x0 - d:x0 + d + 1]; vm = I[y0 - v, x0 - d:x0 + d + 1]
        us = np.arange(-d, d + 1)
        m01 += v * int((vp - vm).sum())
        m10 += int((us * (vp + vm)).sum())
    return fast_atan2_ref(np.float32(m01), np.float32(m10))


def retain_best_ref(kps, n):
    """kps: list of dicts with 'response'; keeps every keypoint tying the n-th response, input order."""
    if n >= len(kps):
        return kps
    if n == 0:
        return []
    amb = sorted((k["response"] for k in kps), reverse=True)[n - 1]
    return [k for k in kps if k["response"] >= amb]


def orb_detect_ref(img, nfeatures, scale_factor, nlevels, edge, patch, fast_thr):
    f = np.float32
    factor = f(1.0 / scale_factor)
    nd = f(nfeatures) * (f(1) - factor) / (f(1) - f(np.power(float(factor), float(nlevels))))
    per, tot = [], 0
    for l in range(nlevels - 1):
        per.append(int(np.rint(nd))); tot += per[-1]; nd = f(nd * factor)
    per.append(max(nfeatures - tot, 0))
    half = patch // 2
    umax = orb_umax_ref(half)
    rows, cols = img.shape
    lev = img
    out = []
    for l in range(nlevels):
        sc = f(np.power(float(f(scale_factor)), float(l)))
        if l > 0:
            nr, nc = int(np.rint(rows / float(sc))), int(np.rint(cols / float(sc)))
            if nr < 2 * edge + 8 or nc < 2 * edge + 8:
                break
            lev = resize_linear_ref(lev, nr, nc)
        lr, lc = lev.shape
        kps = [dict(x=x, y=y, response=np.float32(s)) for (x, y, s) in fast9_16(lev, fast_thr)
               if edge <= x < lc - edge and edge <= y < lr - edge]
        kps = retain_best_ref(kps, 2 * per[l])
        for k in kps:
            k["response"] = harris_ref(lev, k["x"], k["y"])
        kps = retain_best_ref(kps, per[l])
        for k in kps:
            ang = ic_angle_ref(lev, k["x"], k["y"], half, umax)
            out.append([f(k["x"]) * sc if l else f(k["x"]), f(k["y"]) * sc if l else f(k["y"]), f(patch) * sc, ang, k["response"], f(l)])
    return np.array(out, np.float32).reshape(-1, 6)


def gen_orb(rng):
    img = block_image(rng, 150, 190, 6)
    out = {"img": img}
    for name, (dr, dc) in {"down12": (125, 158), "down2": (75, 95), "odd": (101, 77), "up": (180, 228)}.items():
        out["resize_" + name] = resize_linear_ref(img, dr, dc)
    pts = np.stack([rng.integers(16, 190 - 16, 48), rng.integers(16, 150 - 16, 48)], axis=1).astype(np.int16)
    umax = orb_umax_ref(15)
    out["ha_xy"] = pts
    out["ha_response"] = np.array([harris_ref(img, int(x), int(y)) for x, y in pts], np.float32)
    out["ha_angle"] = np.array([ic_angle_ref(img, int(x), int(y), 15, umax) for x, y in pts], np.float32)
    out["orb_a"] = orb_detect_ref(img, 60, 1.2, 4, 31, 31, 20)       # budgets bind on every level
    out["orb_b"] = orb_detect_ref(img, 5000, 1.2, 8, 31, 31, 12)     # the reference's OrbDetector parameters: nothing is cut
    np.savez_compressed(os.path.join(HERE, "orb.npz"), **out)
    return len(out["orb_a"]), len(out["orb_b"])


# ---------------------------------------------------------------------------------------------
# Landmark::update (types/landmark.cpp:66-167) with numpy matrices and numpy.linalg.solve
def landmark_update_ref(w2c, c2w, meas, world, updates, max_iter=100, kernel=25.0):
    """meas: list of (frame, cam xyz); last = new observation.  Returns (world, updates)."""
    w = np.array(world, float)
    prev = 0.0
    for it in range(max_iter):
        H = np.zeros((3, 3)); b = np.zeros(3); err = 0.0; n_out = 0
        for (f, mc) in meas:
            R, t = w2c[f][:, :3], w2c[f][:, 3]
            s_ = R @ w + t
            if s_[2] <= 0:
                n_out += 1
                continue
            e = s_ - mc
            om = 1.0 / mc[2]
            e2 = om * float(e @ e)
            err += e2
            if e2 > kernel:
                om *= kernel / e2
                n_out += 1
            H += om * (R.T @ R); b += om * (R.T @ e)
        if np.any(H != 0):              # H == 0 (every measurement skipped): fullPivLu().solve returns 0
            w = w + np.linalg.solve(H, -b)
        if abs(err - prev) < 1e-5 or it == 999:
            n_in = len(meas) - n_out
            if n_in > updates:
                return w, n_in
            if n_in < n_out:
                acc = np.zeros(3)
                for (f, mc) in meas:
                    acc += c2w[f][:, :3] @ mc + c2w[f][:, 3]
                return acc / len(meas), updates
            return np.array(world, float), updates
        prev = err
    return np.array(world, float), updates


def gen_landmark(rng):
    n_frames = 24
    w2c, c2w = [], []
    for f in range(n_frames):
        v = np.array([0.02 * f + rng.normal(0, 0.01), rng.normal(0, 0.01), 0.9 * f + rng.normal(0, 0.02), rng.normal(0, 0.004), 0.01 * np.sin(f / 5.0), rng.normal(0, 0.004)])
        C = v2t(v)                      # camera to world
        c2w.append(C[:3, :]); w2c.append(np.linalg.inv(C)[:3, :])
    offsets, frame_of, cam, world, updates = [0], [], [], [], []
    ref_world, ref_updates = [], []
    for i in range(60):
        f0 = int(rng.integers(0, n_frames - 4)); length = int(rng.integers(3, min(12, n_frames - f0)))
        X = c2w[f0][:, :3] @ np.array([rng.uniform(-6, 6), rng.uniform(-2, 2), rng.uniform(4, 40)]) + c2w[f0][:, 3]
        kind = i % 6
        meas = []
        for k in range(length):
            f = f0 + k
            pc = w2c[f][:, :3] @ X + w2c[f][:, 3]
            noise = rng.normal(0, 0.02 * max(pc[2], 1.0) / 10, 3)
            if kind == 3 and k % 2 == 0:
                noise += rng.normal(0, 8.0, 3)                     # gross outliers: kernel saturation
            mc = pc + noise
            mc[2] = max(mc[2], 0.3)
            meas.append((f, mc))
        if kind == 4:
            meas = meas[::-1]                                      # creation order (newest first) as the constructor leaves it
        w0 = X + rng.normal(0, 0.3, 3)
        up0 = length - 1 if kind != 5 else length + 5             # kind 5: already better supported, the estimate is not taken
        if kind == 2:
            w0 = X + np.array([0, 0, -200.0])                      # behind most cameras: outliers dominate -> reset to the mean
        wr, ur = landmark_update_ref(w2c, c2w, meas, w0, up0)
        offsets.append(offsets[-1] + len(meas)); frame_of += [m[0] for m in meas]; cam += [m[1] for m in meas]
        world.append(w0); updates.append(up0); ref_world.append(wr); ref_updates.append(ur)
    np.savez_compressed(os.path.join(HERE, "landmark.npz"), w2c=np.array(w2c), c2w=np.array(c2w), offsets=np.array(offsets, np.int32),
                        frame_of=np.array(frame_of, np.int32), cam=np.array(cam), world=np.array(world), updates=np.array(updates, np.int32),
                        ref_world=np.array(ref_world), ref_updates=np.array(ref_updates, np.int32))
    return sum(1 for a, b in zip(updates, ref_updates) if a != b), len(updates)


# ---------------------------------------------------------------------------------------------
# StereoFramePointGenerator::recoverPoints (stereo_framepoint_generator.cpp:683-869)
def stereo_recover_ref(K, bh, rows, cols, imgL, imgR, pat, w2c, lost, kp_size, tau_track, tau_tri, min_depth, max_depth, min_disp):
    """lost: (has_landmark, world xyz, descL, descR).  Returns [(index, xL, yL, xR, yR, dist, descL, descR, xyz)]."""
    out = []
    R, t = w2c[:, :3], w2c[:, 3]
    f32 = np.float32
    for i, (has_lm, X, pdL, pdR) in enumerate(lost):
        if not has_lm:
            continue
        pc = np.array([((R[k, 0] * X[0] + R[k, 1] * X[1]) + R[k, 2] * X[2]) + t[k] for k in range(3)])
        uL = np.array([(K[k, 0] * pc[0] + K[k, 1] * pc[1]) + K[k, 2] * pc[2] for k in range(3)])
        uR = uL + bh
        if uL[2] < min_depth or uL[2] > max_depth or uR[2] < min_depth or uR[2] > max_depth:
            continue
        pLx, pLy = f32(np.rint(uL[0] / uL[2])), f32(np.rint(uL[1] / uL[2]))
        pRx, pRy = f32(np.rint(uR[0] / uR[2])), f32(np.rint(uR[1] / uR[2]))
        rbc = f32(5) * f32(kp_size)
        lo, hx, hy = rbc + f32(1), f32(cols) - rbc - f32(1), f32(rows) - rbc - f32(1)
        if pLx < lo or pLx > hx or pRx < lo or pRx > hx or pLy < lo or pLy > hy or pRy < lo or pRy > hy:
            continue
        xL, yL, xR, yR = int(pLx), int(pLy), int(pRx), int(pRy)     # integer-valued: ROI corner and keypoint round trivially
        dL = brief32(imgL, xL, yL, pat)
        if hamming(pdL, dL) > tau_track:
            continue
        dR = brief32(imgR, xR, yR, pat)
        if float(pLx - pRx) < min_disp:
            continue
        if hamming(pdR, dR) > tau_track:
            continue
        dist = hamming(dL, dR)
        if dist > tau_tri:
            continue
        z = bh[0] / float(xR - xL)                                    # getPointInLeftCamera :871-895
        xyz = np.array([1 / K[0, 0] * (xL - K[0, 2]) * z, 1 / K[1, 1] * ((yL + yR) / 2.0 - K[1, 2]) * z, z])
        out.append((i, xL, yL, xR, yR, dist, dL, dR, xyz))
    return out


def gen_stereo_recover(rng):
    pat = read_brief_pattern()
    rows, cols = 150, 220
    f_, cx, cy, base = 160.0, 109.5, 74.5, 0.5
    K = np.array([[f_, 0, cx], [0, f_, cy], [0, 0, 1.0]])
    bh = np.array([-f_ * base, 0.0, 0.0])
    imgL = block_image(rng, rows, cols, 4)
    imgR = np.roll(imgL, -6, axis=1)                                  # a fronto-parallel world at disparity 6 ...
    imgR = np.clip(imgR.astype(np.int64) + rng.integers(-4, 5, imgR.shape), 0, 255).astype(np.uint8)   # ... plus sensor noise
    ang = -0.015
    w2c = np.hstack([np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]), np.array([[-0.02], [0.01], [0.04]])])
    c2w_R, c2w_t = w2c[:, :3].T, -w2c[:, :3].T @ w2c[:, 3]
    lost = []
    for i in range(160):
        u = float(rng.uniform(30, cols - 30)); v = float(rng.uniform(30, rows - 30))
        if i % 10 == 3:
            u = float(rng.uniform(-5, cols + 5)); v = float(rng.uniform(-5, rows + 5))
        z = f_ * base / 6.0 if i % 4 else float(rng.uniform(2.0, 60.0))   # most landmarks at the depth the right image shows
        if i % 17 == 0:
            z = 200.0                                                 # disparity below the minimum
        pc = np.array([(u - cx) * z / f_, (v - cy) * z / f_, z])
        X = c2w_R @ pc + c2w_t
        bx, by = int(np.clip(round(u), 28, cols - 29)), int(np.clip(round(v), 28, rows - 29))
        bxr = int(np.clip(round(u - f_ * base / z), 28, cols - 29))

        def flip(d, k):
            bits = np.unpackbits(d)
            bits[rng.choice(256, size=k, replace=False)] ^= 1
            return np.packbits(bits)
        lost.append((int(rng.random() < 0.9), X, flip(brief32(imgL, bx, by, pat), int(rng.integers(0, 45))),
                     flip(brief32(imgR, bxr, by, pat), int(rng.integers(0, 45)))))
    tau_track, tau_tri = 35.0, 60.0
    rec = stereo_recover_ref(K, bh, rows, cols, imgL, imgR, pat, w2c, lost, 7.0, tau_track, tau_tri, 0.1, 1000.0, 1.0)
    out = {"K": K, "bh": bh, "imgL": imgL, "imgR": imgR, "w2c": w2c, "tau_track": np.float64(tau_track), "tau_tri": np.float64(tau_tri),
           "has_lm": np.array([q[0] for q in lost], np.uint8), "lm": np.array([q[1] for q in lost]),
           "pdL": np.array([q[2] for q in lost], np.uint8), "pdR": np.array([q[3] for q in lost], np.uint8),
           "rec_index": np.array([r[0] for r in rec], np.int32), "rec_xy4": np.array([r[1:5] for r in rec], np.int32).reshape(-1, 4),
           "rec_dist": np.array([r[5] for r in rec], np.int32),
           "rec_desc": np.array([np.concatenate([r[6], r[7]]) for r in rec], np.uint8).reshape(-1, 64),
           "rec_xyz": np.array([r[8] for r in rec], np.float64).reshape(-1, 3)}
    np.savez_compressed(os.path.join(HERE, "stereo_recover.npz"), **out)
    return len(rec)


# ---------------------------------------------------------------------------------------------
# PoseTracker3D control arithmetic (pose_tracker_3d.cpp:240-288 and :439-466) on scripted inputs; configuration_kitti.yaml values
def track_adapt_ref(steps, win, tau, target, wmin=15, wmax=50, tmin=25.6, tmax=51.2, good=0.2, tunnel=0.5, min_inliers=100):
    out = []
    for (n_prev, n_trk, n_lm, by_app) in steps:
        if by_app:
            win = wmax
        ratio = n_trk / n_prev
        lm_per_pt = n_lm / n_trk if n_trk else float("nan")
        success = n_trk / target
        if ratio < good / 2:
            if win < wmax:
                win = int(min(win * 1 / tunnel, float(wmax)))
        else:
            if win > wmin:
                win = int(max(win * tunnel, float(wmin)))
        if ratio < good or n_trk < min_inliers or (lm_per_pt < 0.5 and success < 0.25):
            tau = min(tau + 5, tmax)
        else:
            tau = max(tau - 5, tmin)
        out.append((win, tau))
    return out


def gen_tracker(rng):
    target = (1241 // 15 + 1) * (376 // 15 + 1)          # base_framepoint_generator.cpp:304-308, bin 15
    steps = []
    for i in range(60):
        n_prev = int(rng.integers(300, 2200))
        mode = 0 if i < 6 else i % 6                     # six good frames first: the descriptor distance reaches its lower clamp
        frac = [0.9, 0.5, 0.15, 0.05, 0.3, 0.19][mode] * float(rng.uniform(0.9, 1.1))
        n_trk = max(1, int(n_prev * frac))
        if mode == 4:
            n_trk = int(rng.integers(40, 100))             # below the aligner's minimum number of inliers
        n_lm = int(n_trk * float(rng.uniform(0.1, 0.95)))
        steps.append((n_prev, n_trk, n_lm, int(i % 11 == 0)))
    res = track_adapt_ref(steps, 20, 40.0, target)
    n = 300
    err = rng.uniform(0, 30, n); err[rng.random(n) < 0.1] = -1.0; err[rng.random(n) < 0.05] = rng.uniform(390, 420, 1)[0]
    inl = ((err >= 0) & (err <= 4.0)).astype(np.uint8)
    out = {"steps": np.array(steps, np.int32), "win": np.array([r[0] for r in res], np.int32), "tau": np.array([r[1] for r in res], np.float64),
           "errors": err, "inliers": inl}
    for name, total in (("low", 2.0 * n), ("high", 9.0 * n)):   # average error below / above the kernel (4)
        avg = total / n
        keep = inl.astype(bool) if avg < 4.0 else ((err != -1) & (err < 100 * 4.0))
        out["total_" + name] = np.float64(total); out["keep_" + name] = keep.astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "tracker.npz"), **out)
    return len(steps)


# ---------------------------------------------------------------------------------------------
# StereoUVAligner::_weights_translation over a sequence of initialize() calls on ONE aligner (stereouv_aligner.cpp:22,57-61):
# a Python list stands in for the std::vector member; `resize(n, 1)` truncates or appends ones, never touches what it keeps.
def aligner_weights_ref(calls, max_reliable=15.0):
    w = []                                              # the member vector, empty in a new aligner
    out = []
    for n, inverse_depth, depth in calls:
        if n < len(w):
            del w[n:]                                   # vector::resize shrinks ...
        else:
            w.extend([1.0] * (n - len(w)))              # ... or appends the fill value
        if inverse_depth:
            for u in range(n):
                w[u] = min(max_reliable / depth[u], 1.0)
        out.append(list(w))
    return out


def gen_aligner_weights(rng):
    # Tracking frames (inverse depth on) with varying point counts, a track break, then Localizing frames (off) with FEWER and
    # with MORE points than the last Tracking frame, Tracking again, a second break after a shrink (grow-back must give ones).
    script = [(300, 1), (420, 1), (380, 1), (150, 0), (150, 0), (520, 0), (90, 0), (200, 0), (610, 1), (10, 1), (64, 0), (0, 0), (33, 0), (33, 1)]
    calls = []
    for n, inv in script:
        depth = rng.uniform(1.5, 60.0, n)                # both sides of the 15 m reliable depth
        calls.append((n, inv, depth))
    res = aligner_weights_ref(calls)
    out = {"sizes": np.array([c[0] for c in calls], np.int32), "inverse_depth": np.array([c[1] for c in calls], np.int32),
           "depth": np.concatenate([c[2] for c in calls]), "weights": np.concatenate([np.array(r, np.float64) for r in res])}
    np.savez_compressed(os.path.join(HERE, "aligner_weights.npz"), **out)
    return len(calls)


# ---------------------------------------------------------------------------------------------
# cv::ORB::create()->compute() on provided keypoints [recalled]: GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) in the 8-bit
# fixed-point form of OpenCV's separable filter, then 256 steered intensity tests per keypoint (WTA_K = 2) with the
# repo-defined pair table of include/vslam_orb_pattern.h.
def read_orb_pattern():
    text = open(os.path.join(ROOT, "include", "vslam_orb_pattern.h")).read()
    body = text[text.index("VSLAM_ORB_PATTERN_INIT"):]
    q = re.findall(r"\{(-?\d+),(-?\d+),(-?\d+),(-?\d+)\}", body)
    assert len(q) == 256
    return np.array(q, np.int32)


def gauss7_fixed():
    x = np.arange(7, dtype=np.float64) - 3.0
    cf = np.exp(-0.5 / 4.0 * x * x).astype(np.float32)          # stored as float (CV_32F kernel)
    s = 1.0 / float(np.sum(cf.astype(np.float64)))
    cf = (cf.astype(np.float64) * s).astype(np.float32)
    return np.rint(cf.astype(np.float64) * 256.0).astype(np.int64)


def gaussian_blur7_ref(img):
    k = gauss7_fixed()
    p = np.pad(img.astype(np.int64), 3, mode="reflect")          # numpy 'reflect' == BORDER_REFLECT_101
    rows, cols = img.shape
    h = sum(k[i] * p[3:3 + rows, i:i + cols] for i in range(7))   # row pass on the image rows only
    hp = np.pad(h, ((3, 3), (0, 0)), mode="reflect")
    v = sum(k[i] * hp[i:i + rows, :] for i in range(7))
    return np.clip((v + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def orb_describe_ref(blur, x, y, angle_degrees, pat):
    ang = np.float32(angle_degrees) * np.float32(np.pi / np.float32(180.0))
    a, b = np.float32(np.cos(np.float64(ang))), np.float32(np.sin(np.float64(ang)))
    d = np.zeros(32, np.uint8)
    for i in range(256):
        v = []
        for h in range(2):
            px, py = np.float32(pat[i][2 * h]), np.float32(pat[i][2 * h + 1])
            xf = np.float32(px * a) - np.float32(py * b)
            yf = np.float32(px * b) + np.float32(py * a)
            v.append(int(blur[y + int(np.rint(yf)), x + int(np.rint(xf))]))
        if v[0] < v[1]:
            d[i >> 3] |= 1 << (i & 7)
    return d


def gen_orb_descriptor(rng):
    pat = read_orb_pattern()
    img = block_image(rng, 150, 190, 7)
    img = np.clip(img.astype(np.int32) + rng.integers(-12, 13, img.shape), 0, 255).astype(np.uint8)
    flat = np.full((40, 52), 201, np.uint8)                       # constant image: the 257/256 kernel gain shows (201 -> 203)
    out = {"img": img, "blur": gaussian_blur7_ref(img), "flat": flat, "flat_blur": gaussian_blur7_ref(flat), "kernel": gauss7_fixed().astype(np.int32)}
    pts = [(31, 31), (158, 118), (30, 60), (159, 60), (60, 119), (95, 75), (100, 31), (31, 118)]
    pts += [(int(rng.integers(31, 159)), int(rng.integers(31, 119))) for _ in range(40)]
    xy = np.array(pts, np.int16)
    out["xy"] = xy
    for name, ang in (("fast", -1.0), ("a0", 0.0), ("a37", 37.25), ("a180", 180.0), ("a301", 301.5)):
        keep = np.array([1 if (31 <= x < 190 - 31 and 31 <= y < 150 - 31) else 0 for x, y in pts], np.uint8)
        desc = np.zeros((len(pts), 32), np.uint8)
        for i, (x, y) in enumerate(pts):
            if keep[i]:
                desc[i] = orb_describe_ref(out["blur"], x, y, ang, pat)
        out["keep"] = keep
        out["desc_" + name] = desc
        out["angle_" + name] = np.float32(ang)
    np.savez_compressed(os.path.join(HERE, "orb_descriptor.npz"), **out)
    return len(pts)


def main():
    import sys
    if len(sys.argv) > 1:                                 # regenerate selected fixtures only: make_golden.py aligner_weights ...
        streams = {"aligner_weights": 20261013, "orb_descriptor": 20261014}
        for name in sys.argv[1:]:
            globals()["gen_" + name](np.random.default_rng(streams[name]))
        return
    rng = np.random.default_rng(20261003)
    gen_hamming(rng)
    gen_fast(rng)
    gen_brief(rng)
    gen_controller()
    gen_aligner(rng)
    gen_stereo(rng)
    gen_track(np.random.default_rng(20261004))   # own stream: added later, the fixtures above stay byte-identical
    gen_aligner_uvd(np.random.default_rng(20261005))
    gen_depth(np.random.default_rng(20261006))
    gen_depth_track(np.random.default_rng(20261007))
    gen_depth_recover(np.random.default_rng(20261008))
    gen_orb(np.random.default_rng(20261009))
    gen_landmark(np.random.default_rng(20261010))
    gen_stereo_recover(np.random.default_rng(20261011))
    gen_tracker(np.random.default_rng(20261012))
    gen_aligner_weights(np.random.default_rng(20261013))
    gen_orb_descriptor(np.random.default_rng(20261014))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
