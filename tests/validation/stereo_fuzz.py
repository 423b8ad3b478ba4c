#!/usr/bin/env python3
"""The stereo pipeline against the CPU oracle on random worlds and random configurations (the checker is the oracle, as in the tests): image sizes,
detector grids, bin sizes, threshold ranges, epipolar offsets, extractor, recovery / binning switches, damping, track length for landmarks,
1 - 3 streams — every frame compared completely (tests/pipeline_compare.py: integers, keypoints, descriptors, framepoint tuples, aligner results,
all four launch sequences).  usage: stereo_fuzz.py [runs] [frames]"""
import json, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import Oracle
from pipeline_compare import run_sequence

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(40404)
bad, frames, t0 = [], 0, time.time()
for run in range(runs):
    scene = "euroc" if rng.random() < 0.3 else "kitti"
    scale = float(rng.choice([0.35, 0.5, 0.6, 0.75]))
    n_streams = int(rng.choice([1, 1, 2, 3]))
    seeds = [int(v) for v in rng.integers(1, 100000, n_streams)]
    edits = dict(det=tuple(int(v) for v in rng.choice([[1, 1], [2, 2], [1, 3], [3, 2], [2, 1]])), bin=int(rng.choice([11, 15, 22, 30])),
                 thr=(int(rng.integers(5, 25)), int(rng.integers(40, 120))), epi=int(rng.choice([0, 0, 1, 3])), desc=int(rng.integers(0, 2)),
                 recover=int(rng.random() < 0.8), binning=int(rng.random() < 0.8), damping=float(rng.choice([0, 5, 50])),
                 min_track=int(rng.integers(1, 4)), max_change=float(rng.choice([0.1, 0.5, 1.0])), speed=float(rng.uniform(0.3, 1.4)))

    def cfg_edit(c, e=edits):
        c.det_rows, c.det_cols = e["det"]; c.bin_size_pixels = e["bin"]
        c.detector_threshold_minimum, c.detector_threshold_maximum = e["thr"]; c.detector_threshold_maximum_change = e["max_change"]
        c.maximum_epipolar_search_offset_pixels = e["epi"]; c.descriptor_type = e["desc"]
        c.enable_landmark_recovery = e["recover"]; c.enable_keypoint_binning = e["binning"]; c.aligner_damping = e["damping"]
        c.minimum_track_length_for_landmark_creation = e["min_track"]
    try:
        run_sequence(Oracle, dict(scale=scale, speed_m=edits["speed"]), n, n_streams=n_streams, which=scene, cfg_edit=cfg_edit, seeds=seeds, scene=scene)
        frames += n * n_streams
    except Exception as ex:
        bad.append({"run": run, "scene": scene, "scale": scale, "streams": n_streams, "seeds": seeds, "edits": edits,
                    "error": (str(ex) or traceback.format_exc())[:400]})
print(json.dumps({"runs": runs, "stream_frames": frames, "mismatching_runs": bad, "seconds": round(time.time() - t0, 1)}))
