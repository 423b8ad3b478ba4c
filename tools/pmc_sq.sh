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
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -- python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --steps 6 > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - "$OUT" << 'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_") or k.startswith("k_synth"): continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
res = {k: {cn: round(v[0] / v[1], 1) for cn, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
