#!/usr/bin/env python3
"""Where the RGB-D host loop's per-frame time goes: the checker loop (tests/rgbd_loop.py) over the HIP entry points with a wall
clock around every entry-point call (ctypes overhead included: ~10 us per call)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle import Oracle  # renderer only
from test_rgbd_mode import setup
from rgbd_loop import RgbdTracker as PyLoop
from vslam_pose_estimation_framework_amd import hip

o = Oracle()
scene, cfg, p = setup(o, sys.argv[1] if len(sys.argv) > 1 else "tum")
g = hip.load(); g.create(cfg, 0, 1)
acc = collections.defaultdict(lambda: [0.0, 0])
class Timed(object):
    def __init__(self, api): self._api = api
    def __getattr__(self, name):
        f = getattr(self._api, name)
        if not callable(f) or name.startswith("_"): return f
        def w(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); d = time.perf_counter() - t
            acc[name][0] += d; acc[name][1] += 1
            return r
        return w
tr = PyLoop(Timed(g), cfg, p)
frames = [(o.render(scene, k)[0], o.render_depth(scene, k, 2e-3)) for k in range(24)]
for L, D in frames[:4]: tr.process(L, D)
acc.clear()
t0 = time.perf_counter()
for L, D in frames[4:]: info = tr.process(L, D)
tot = (time.perf_counter() - t0) / 20
print("python loop: %.2f ms per frame" % (tot * 1e3))
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-22s %6.1f us per frame  (%.1f calls per frame, %.1f us per call)" % (k, s / 20 * 1e6, n / 20, s / n * 1e6))
