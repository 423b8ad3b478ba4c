#!/bin/bash
# Instruction mix / unit activity per kernel (rocprofv3 PMC, kernel-trace only, one counter group per pass), kernels run
# back to back on one HIP stream (VSLAM_IMG_STREAMS=0) so that every counter belongs to one kernel.
set -e
TAG=${1:-run}
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VSLAM_IMG_STREAMS=0
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -- python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --no-shim --steps 6 > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - "$OUT" << 'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.getcwd())
from vslam_pose_estimation_framework_amd import buildinfo
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_") or k.startswith("k_synth"): continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
res = {k: {cn: round(v[0] / v[1], 1) for cn, v in d.items()} for k, d in acc.items()}
json.dump({"source_sha16": buildinfo.source_sha16(), "command": "VSLAM_IMG_STREAMS=0 python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --no-shim --steps 6",
           "note": "per kernel launch, summed over the chip; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs",
           "per_kernel": res}, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
