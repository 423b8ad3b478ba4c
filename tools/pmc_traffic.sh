#!/bin/bash
# HBM traffic per kernel launch from rocprofv3 PMC counters (separate passes, kernel-trace only — see
# /opt/skills/guides/MI355X_MICROARCH.md).  Run on the GPU box from the repo root:  bash tools/pmc_traffic.sh <tag>
set -e
TAG=${1:-run}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --no-shim --steps 8 > $OUT/$C.log 2>&1
done
python3 - "$OUT" << 'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.getcwd())
from vslam_pose_estimation_framework_amd import buildinfo
out = sys.argv[1]
res = collections.defaultdict(dict)
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(out + "/" + cname + "/*/*counter_collection.csv")
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname: continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        if k.startswith("k_") and not k.startswith("k_synth"): res[k][cname] = round(v / n, 1)
streams = None
for line in open(out + "/FETCH_SIZE.log"):          # the bench line of the profiled run says how many streams it ran
    if line.startswith("{") and "streams_per_gpu" in line:
        streams = json.loads(line)["config"]["streams_per_gpu"]
json.dump({"streams": streams, "source_sha16": buildinfo.source_sha16(), "library_sha16": buildinfo.library_sha16(), "command": "python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --no-shim --steps 8", "counters": "FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, kernel-trace only; KB per launch as reported (see MI355X_MICROARCH.md for the fetch under-count of wide loads)", "per_launch_KB": res}, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
