#!/bin/bash
# overlapped bench under different environment settings: bash tools/probe_env.sh <tag> "VAR=a" "VAR=b" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for E in "$@"; do
  i=$((i+1))
  env $E timeout -k 10 200 python bench.py --no-cpu --no-exact --no-pcie --no-ate --steps 35 > $OUT/e$i.json 2> $OUT/e$i.err
  python - $OUT/e$i.json "$E" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value %.0f ms/step %.4f"%(d["value"], d["ms_per_step"]), " ".join("%s=%.4f"%(k.replace("k_",""),v["avg_ms"]) for k,v in d["kernels"].items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
