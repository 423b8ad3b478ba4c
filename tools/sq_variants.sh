#!/bin/bash
# SQ instruction counters of k_fast_box (and friends) for library variants: bash tools/sq_variants.sh <tag> lib1.so lib2.so ...
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VSLAM_IMG_STREAMS=0
for L in "$@"; do
  n=$(basename $L .so)
  VSLAM_HIP_LIB=$GRAFT_REPO_ROOT/vslam_pose_estimation_framework_amd/csrc/$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/$n -- python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --steps 4 > $OUT/$n.log 2>&1 || echo "$n failed"
done
python3 - $OUT "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for L in sys.argv[2:]:
    n = L[:-3]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(out + "/" + n + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k in ("k_fast_box", "k_brief", "k_emit"):
        if k in acc:
            d = {c: v[0] / v[1] for c, v in acc[k].items()}
            w = d.get("SQ_WAVES", 1)
            print("%-22s %-12s VALU/wave %6.1f SALU/wave %6.1f LDS/wave %5.1f" % (n, k, d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_SALU", 0) / w, d.get("SQ_INSTS_LDS", 0) / w))
PY
