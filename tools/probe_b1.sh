#!/bin/bash
# one stream alone (exact mode's shape) for library variants: bash tools/probe_b1.sh m9 m13 ...  (csrc/libvslam_hip_<name>.so)
for L in "$@"; do
VSLAM_HIP_LIB=$PWD/vslam_pose_estimation_framework_amd/csrc/libvslam_hip_$L.so python bench.py --streams 1 --steps 1500 --warmup 100 --no-cpu --no-exact --no-pcie --no-ate | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=1', '$L', d['value'], d['ms_per_step'], d['kernels']['k_frame']['avg_ms'])"
done
