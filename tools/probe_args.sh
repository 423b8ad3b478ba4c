#!/bin/bash
# overlapped bench with different bench arguments: bash tools/probe_args.sh <tag> "--streams 128" "--streams 192" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for A in "$@"; do
  i=$((i+1))
  timeout -k 10 200 python bench.py --no-cpu --no-exact --no-pcie --no-ate $A > $OUT/a$i.json 2> $OUT/a$i.err
  python - $OUT/a$i.json "$A" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "value %.0f raw %.0f ms/step %.4f uniq %.3f"%(d["value"], d["raw_pairs_per_s"], d["ms_per_step"], d["config"]["unique_frame_fraction"]), " ".join("%s=%.4f"%(k.replace("k_",""),v["avg_ms"]) for k,v in d["kernels"].items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
