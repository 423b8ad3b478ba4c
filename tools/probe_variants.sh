#!/bin/bash
# back-to-back kernel times (one HIP stream) of library variants: bash tools/probe_variants.sh <tag> lib1.so lib2.so ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for L in "$@"; do
  n=$(basename $L .so)
  VSLAM_HIP_LIB=$PWD/vslam_pose_estimation_framework_amd/csrc/$L VSLAM_IMG_STREAMS=0 timeout -k 10 200 python bench.py --no-cpu --no-exact --no-pcie --no-ate --steps 35 > $OUT/$n.b2b.json 2> $OUT/$n.b2b.err
  VSLAM_HIP_LIB=$PWD/vslam_pose_estimation_framework_amd/csrc/$L timeout -k 10 200 python bench.py --no-cpu --no-exact --no-pcie --no-ate --steps 35 > $OUT/$n.ovl.json 2> $OUT/$n.ovl.err
  python - $OUT/$n <<'PY'
import json,sys
for mode in ("b2b","ovl"):
    try:
        d=json.load(open(sys.argv[1]+"."+mode+".json"))
        print(sys.argv[1].split("/")[-1], mode, "value %.0f ms/step %.4f"%(d["value"], d["ms_per_step"]), " ".join("%s=%.4f"%(k.replace("k_",""),v["avg_ms"]) for k,v in d["kernels"].items()))
    except Exception as e:
        print(sys.argv[1], mode, "FAILED", e)
PY
done
