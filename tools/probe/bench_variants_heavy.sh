#!/bin/bash
# The bench's timed loop on the heavier scene (--contrast 2 --speed 0.3 --bin 11) under environment variants:  bash tools/probe/bench_variants_heavy.sh "VSLAM_SPLIT=0" "VSLAM_SPLIT=2" ...
for V in "$@"; do
  env $V python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --no-shim --steps 40 --contrast 2 --speed 0.3 --bin 11 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$V', 'value', d['value'], 'ms/step', d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done
