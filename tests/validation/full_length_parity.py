#!/usr/bin/env python3
"""Exact mode over WHOLE KITTI-shaped sequences at 1241 x 376 against the CPU oracle, frame by frame (a one-off validation beyond the suite's
1000-frame test; the oracle is the checker here, as in the tests): every frame's integer counters, detector thresholds, tracker state and pose,
the complete comparison (keypoints, descriptors, framepoint tuples, landmarks, aligner results) every 250 frames and on the last frame.
  full_length_parity.py            KITTI-00's 4541 frames as one stream (configs[1], the literal drop-in)
  full_length_parity.py config3    sequences 00 + 02 + 05 + 06 as four streams of their own lengths (configs[2]: one GPU's view)
Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hip_configs import _long_run
from vslam_pose_estimation_framework_amd import buildinfo

KITTI_FRAMES = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]   # odometry sequences 00..10

which = sys.argv[1] if len(sys.argv) > 1 else "config2"
if which == "config3":
    seqs = [0, 2, 5, 6]
    lengths = [KITTI_FRAMES[q] for q in seqs]
    seeds, speeds = [1000 * q + 7 for q in seqs], [0.9, 0.8, 1.0, 0.85]
else:
    seqs, lengths, seeds, speeds = [0], [4541], [7], [0.9]
t0 = time.time()
worst, stats = _long_run(lengths, seeds, speeds, full_every=250)
print(json.dumps({"what": "exact mode, whole sequences at 1241 x 376, HIP vs CPU oracle frame by frame", "sequences": seqs, "frames": lengths,
                  "integer_mismatches": 0, "worst_relative_pose_error": worst, "pose_tolerance": 1e-4, "stats": stats,
                  "wall_seconds": round(time.time() - t0, 1), "source_sha16": buildinfo.source_sha16()}))
