#!/bin/bash
# Launch-sequence experiment: the fused frame kernel against registration + co-scheduled tail (VSLAM_SPLIT=3).
OUT=$PWD/gpurun_out/${1:-cosched}
mkdir -p $OUT
run() {  # name, env...
  local n=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu --no-exact --no-pcie --no-ate --steps 39 > $OUT/$n.json 2> $OUT/$n.err || { echo "$n failed"; tail -3 $OUT/$n.err; return; }
  python -c "
import json; d=json.load(open('$OUT/$n.json')); ch=d['chronometers_s']
print('%-14s' % '$n', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()}, 'flags', d['config']['error_flags'])"
}
run split0 VSLAM_SPLIT=0
run split3 VSLAM_SPLIT=3
run split3_b2b VSLAM_SPLIT=3 VSLAM_IMG_STREAMS=0
run split2 VSLAM_SPLIT=2
VSLAM_SPLIT=3 timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py -m gpu -x -q -k "pipeline_parity or config1 or config5 or config4 or config3 or degenerate or capacity or history" > $OUT/tests.log 2>&1; echo "split3 parity rc=$?"; tail -3 $OUT/tests.log
VSLAM_SPLIT=3 VSLAM_HIP_LIB=$PWD/vslam_pose_estimation_framework_amd/csrc/libvslam_hip_sp.so timeout -k 10 200 python tools/probe/phases_single_stream.py 2>&1 | tail -3
