"""Stand-alone entries on seeded random shapes: HIP against the oracle, bit for bit.  Sizes straddle the kernels' tile geometry
(64 x 48 FAST tiles, 128 x 64 BRIEF tiles, 3 / 28 / 31 px borders), ROIs sit anywhere inside the image, images smaller than a
border yield nothing."""
import numpy as np
import pytest

from vslam_pose_estimation_framework_amd import hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    api = hip.load()
    api.create(api.default_config("kitti"), 0, 1)
    yield api
    api.destroy()


def _image(rng, rows, cols):
    # blocks of random grey + noise: corners at block junctions, no plateaus
    bs = int(rng.integers(3, 9))
    base = rng.integers(20, 236, (rows // bs + 2, cols // bs + 2))
    img = np.kron(base, np.ones((bs, bs), np.int64))[:rows, :cols]
    return np.clip(img + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)


SHAPES = [(5, 40), (7, 7), (8, 9), (47, 63), (48, 64), (49, 65), (57, 57), (63, 200), (97, 129), (130, 70), (200, 333), (376, 1241), (480, 752)]


def test_fast_detect_random_shapes_and_rois(gpu, oracle):
    rng = np.random.default_rng(2026)
    total = 0
    for rows, cols in SHAPES:
        img = _image(rng, rows, cols)
        rois = [(0, 0, cols, rows)]
        for _ in range(3):
            w = int(rng.integers(1, cols + 1)); h = int(rng.integers(1, rows + 1))
            rois.append((int(rng.integers(0, cols - w + 1)), int(rng.integers(0, rows - h + 1)), w, h))
        for roi in rois:
            for thr in (int(rng.integers(1, 12)), int(rng.integers(12, 60))):
                a = gpu.fast_detect(img, roi, thr)
                b = oracle.fast_detect(img, roi, thr)
                np.testing.assert_array_equal(a[0], b[0], err_msg="%dx%d roi %s thr %d" % (rows, cols, roi, thr))
                np.testing.assert_array_equal(a[1], b[1], err_msg="%dx%d roi %s thr %d scores" % (rows, cols, roi, thr))
                total += len(a[0])
    assert total > 20000


def test_descriptors_and_blur_random_shapes(gpu, oracle):
    rng = np.random.default_rng(2027)
    kept = 0
    for rows, cols in SHAPES:
        img = _image(rng, rows, cols)
        np.testing.assert_array_equal(gpu.gaussian_blur7_u8(img), oracle.gaussian_blur7_u8(img), err_msg="blur %dx%d" % (rows, cols))
        n = int(rng.integers(1, 400))
        xy = np.stack([rng.integers(0, cols, n), rng.integers(0, rows, n)], 1).astype(np.int16)   # also points inside the borders
        for name in ("brief_describe", "orb_describe"):
            ka, da = getattr(gpu, name)(img, xy)
            kb, db = getattr(oracle, name)(img, xy)
            np.testing.assert_array_equal(ka, kb, err_msg="%s keep %dx%d" % (name, rows, cols))
            np.testing.assert_array_equal(da[ka > 0], db[kb > 0], err_msg="%s %dx%d" % (name, rows, cols))
            kept += int(ka.sum())
        for angle in (0.0, 37.5, 181.0, 359.0):
            ka, da = gpu.orb_describe(img, xy, angle)
            kb, db = oracle.orb_describe(img, xy, angle)
            np.testing.assert_array_equal(da[ka > 0], db[kb > 0], err_msg="orb angle %g %dx%d" % (angle, rows, cols))
    assert kept > 800


def test_knn2_random_sizes(gpu, oracle):
    rng = np.random.default_rng(2028)
    for nq, nt in [(1, 1), (1, 2), (2, 1), (3, 300), (17, 16), (255, 257), (300, 3), (1000, 999)]:
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        t[: min(nt, nq) // 3] = q[: min(nt, nq) // 3]            # exact matches and ties
        for norm in (0, 1, 2, 3):
            ia, da = gpu.knn2(q, t, norm)
            ib, db = oracle.knn2(q, t, norm)
            np.testing.assert_array_equal(ia, ib, err_msg="knn2 %dx%d norm %d" % (nq, nt, norm))
            np.testing.assert_array_equal(da, db, err_msg="knn2 %dx%d norm %d distances" % (nq, nt, norm))
