#!/usr/bin/env python3
"""The reference's trajectory_analyzer (executables/trajectory_analyzer.cpp) on this repository's restatement:

    python tools/trajectory_analyzer.py -tum <trajectory.txt> -asl <ground_truth.csv> [-skip <integer>]

prints the number of interpolated positions, the raw RMSE, the per-iteration log of the robust alignment and the optimal RMSE."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vslam_pose_estimation_framework_amd import evaluation  # noqa: E402


def main(argv):
    if len(argv) < 5:
        print("usage: ./trajectory_analyzer -tum <trajectory.txt> -asl <ground_truth.txt> [-skip <integer>]", file=sys.stderr)
        return 0
    tum = asl = None
    skip = 0
    i = 1
    while i < len(argv):
        if argv[i] == "-tum" and i + 1 < len(argv):
            tum = argv[i + 1]; i += 1
        elif argv[i] == "-asl" and i + 1 < len(argv):
            asl = argv[i + 1]; i += 1
        elif argv[i] == "-skip" and i + 1 < len(argv):
            skip = int(argv[i + 1]); i += 1
        i += 1
    r = evaluation.trajectory_analyzer(tum, asl, skip)
    print("interpolated positions: %d" % r["correspondences"], file=sys.stderr)
    print("raw RMSE: %.9g" % r["raw_rmse"], file=sys.stderr)
    for k, (total, inliers) in enumerate(r["iterations"]):
        print("iteration: %03d total error (m^2): %12.3f (inliers: %4d/%4d=%4.2f)" % (k, total, inliers, r["correspondences"], inliers / r["correspondences"]))
    print("optimal RMSE: %.9g" % r["optimal_rmse"], file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
