#!/bin/bash
# bash tools/probe_lib_args.sh <tag> <lib.so> "--streams 160" "--streams 224" ...
TAG=$1; LIB=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for A in "$@"; do
  i=$((i+1))
  VSLAM_HIP_LIB=$PWD/vslam_pose_estimation_framework_amd/csrc/$LIB timeout -k 10 200 python bench.py --no-cpu --no-exact --no-pcie --no-ate $A > $OUT/$LIB.$i.json 2> $OUT/$LIB.$i.err
  python - $OUT/$LIB.$i.json "$LIB $A" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value %.0f raw %.0f ms/step %.4f"%(d["value"], d["raw_pairs_per_s"], d["ms_per_step"]), " ".join("%s=%.4f"%(k.replace("k_",""),v["avg_ms"]) for k,v in d["kernels"].items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
done
