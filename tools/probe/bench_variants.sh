#!/bin/bash
# The bench's timed loop under environment variants (one line each):  bash tools/probe/bench_variants.sh "VAR=a" "VAR=b OTHER=c" ...
for V in "$@"; do
  env $V python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --no-shim --steps 70 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$V', 'value', d['value'], 'ms/step', d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done
