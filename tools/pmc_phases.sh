#!/bin/bash
# Where does the frame kernel's HBM traffic come from?  The same bench command with VSLAM_SPLIT=1: the frame runs as three phase
# launches of k_frame (0: track resolution + aligner + prune, 1: recovery gates + landmark bookkeeping, 2: landmark refinement tail +
# stereo sweep + binning + report) around the wide k_recover_brief and k_update_landmarks kernels, so FETCH_SIZE / WRITE_SIZE
# (separate rocprofv3 --pmc passes, kernel-trace only) can be read per phase.  k_frame dispatches are told apart by their order
# inside a step (p0, p1, p2).  Run on the GPU box from the repo root:  bash tools/pmc_phases.sh <tag>
set -e
TAG=${1:-run}
OUT=$PWD/gpurun_out/pmcph_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VSLAM_SPLIT=1
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
    rows = []
    for f in glob.glob(out + "/" + cname + "/*/*counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == cname]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    acc = collections.defaultdict(lambda: [0.0, 0])
    nframe = 0
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if k == "k_frame":
            k = "k_frame_phase%d" % (nframe % 3)
            nframe += 1
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        if k.startswith("k_") and not k.startswith("k_synth"): res[k][cname] = round(v / n, 1)
        if k.startswith("k_frame_phase"): res[k]["launches"] = n
streams = None
for line in open(out + "/FETCH_SIZE.log"):
    if line.startswith("{") and "streams_per_gpu" in line:
        streams = json.loads(line)["config"]["streams_per_gpu"]
json.dump({"streams": streams, "source_sha16": buildinfo.source_sha16(), "launch_sequence": "VSLAM_SPLIT=1: k_frame phase 0 / k_recover_brief / k_frame phase 1 / k_update_landmarks / k_frame phase 2",
           "counters": "FETCH_SIZE / WRITE_SIZE as reported (KB per launch), separate --pmc passes; wide 16 B/lane reads are tallied at half their bytes on gfx950 (MI355X_MICROARCH.md)",
           "per_launch_KB": res}, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
