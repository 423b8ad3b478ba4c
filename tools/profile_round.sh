#!/bin/bash
# Everything the round's profiles/ entries come from, run on the GPU box from the repository root:  bash tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average durations)
#   2. HBM traffic per kernel launch (FETCH_SIZE / WRITE_SIZE, separate --pmc passes): tools/pmc_traffic.sh
#   3. instruction mix / LDS / wait counters, kernels back to back on one stream: tools/pmc_sq.sh
#   4. the N x M 2-NN kernel at N = M = 2158 for the four norms (rocprofv3 --stats)
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --no-shim --steps 35 > $OUT/stats.log 2>&1
echo "stats rc=$?"
bash tools/pmc_traffic.sh $TAG > $OUT/pmc_traffic.log 2>&1; echo "traffic rc=$?"
bash tools/pmc_sq.sh $TAG > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/knn -- python3 tools/probe/bench_knn.py > $OUT/knn.log 2>&1
echo "knn rc=$?"
python3 vslam_pose_estimation_framework_amd/buildinfo.py > $OUT/build.json
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for sub in ("stats", "knn"):
    for f in glob.glob(out + "/" + sub + "/*/*kernel_stats.csv"):
        rows = list(csv.DictReader(open(f)))
        print("==", sub, f.split("/")[-1])
        for r in rows:
            if r["Name"].startswith("k_") or "knn" in r["Name"]:
                print("%-60s calls %6s avg_ns %12s total_ns %14s pct %s" % (r["Name"][:60], r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
PY
