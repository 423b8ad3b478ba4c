#!/usr/bin/env python3
"""Harness fixture: per-frame tracker counters and poses of a 30-frame synthetic sequence
(half-resolution KITTI-shaped scene, seed 21, configuration_kitti.yaml values) -> tests/golden/harness.npz.

Unlike the other fixtures this one is produced BY the CPU oracle (oracle/libvslam_oracle.so): it is a regression
record of the whole PoseTracker3D::compute harness, not an independent restatement.  The images come from the
repository's deterministic renderer (tools/synth/synth_scene.h) and are re-rendered by the tests.
Run from the repository root:  python tests/golden/make_harness.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

FIELDS = ["status", "n_keypoints_left", "n_keypoints_right", "n_detected_left", "n_detected_right", "track_attempts",
          "n_tracked", "n_lost", "n_tracked_landmarks", "aligner_ran", "aligner_iterations", "n_inliers", "n_outliers",
          "n_after_prune", "n_recovered", "n_active_landmarks", "n_new_stereo", "n_points", "track_broken", "fallback",
          "window_pixels"]
N_FRAMES, SCALE, SEED = 30, 0.5, 21


def run(api_factory, render_from):
    """Counters [N_FRAMES][len(FIELDS)], thresholds, tau_track and poses of the sequence through `api_factory()`."""
    scene = render_from.scene_kitti(scale=SCALE, seed=SEED)
    cfg = render_from.config_for_scene(scene)
    api = api_factory()
    api.create(cfg, 0, 1)
    counters = np.zeros((N_FRAMES, len(FIELDS)), np.int32)
    thr = np.zeros(N_FRAMES, np.int32)
    tau = np.zeros(N_FRAMES, np.float64)
    poses = np.zeros((N_FRAMES, 12), np.float64)
    for k in range(N_FRAMES):
        L, R = render_from.render(scene, k)
        api.process_host(L, R)
        fi = api.frame_info(0)
        counters[k] = [getattr(fi, f) for f in FIELDS]
        thr[k] = fi.thresholds[0]
        tau[k] = fi.tau_track
        poses[k] = np.array(fi.camera_left_to_world)
    api.destroy()
    return counters, thr, tau, poses


def main():
    from _oracle import Oracle
    o = Oracle()
    counters, thr, tau, poses = run(Oracle, o)
    np.savez_compressed(os.path.join(HERE, "harness.npz"), fields=np.array(FIELDS), counters=counters, thresholds=thr,
                        tau_track=tau, poses=poses, n_frames=np.int32(N_FRAMES), scale=np.float64(SCALE), seed=np.int32(SEED))
    print("harness.npz:", counters.shape, "last frame", dict(zip(FIELDS, counters[-1].tolist())))


if __name__ == "__main__":
    main()
