#!/bin/bash
# Does it matter that a stream's image products and its frame workgroup meet in the SAME XCD's L2?  The bench's timed loop with the
# relabelling off (VSLAM_XCD_AFFINITY=0), calibrated (default) and deliberately skewed by k XCDs (VSLAM_XCD_SKEW=k).
for V in "VSLAM_XCD_AFFINITY=0" "VSLAM_XCD_SKEW=0" "VSLAM_XCD_SKEW=1" "VSLAM_XCD_SKEW=2" "VSLAM_XCD_SKEW=4" "VSLAM_XCD_SKEW=0" "VSLAM_XCD_AFFINITY=0"; do
  env $V python3 bench.py --no-cpu --no-exact --no-pcie --no-ate --no-shim --steps 70 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$V', 'value', d['value'], 'ms/step', d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done
