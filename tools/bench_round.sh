#!/bin/bash
# the round's bench records (run on the GPU box from the repository root): bash tools/bench_round.sh r02
TAG=${1:-r02}
OUT=gpurun_out/bench_$TAG; mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/default.json 2> $OUT/default.err; echo "default rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err; echo "driver rc=$?"
timeout -k 10 300 python bench.py --mode sequences --sequences 0,2,5,6 --steps 0 > $OUT/seq_00_02_05_06.json 2> $OUT/seq4.err; echo "seq4 rc=$?"
timeout -k 10 400 python bench.py --mode sequences --bin 11 --steps 1200 > $OUT/seq_all_bin11.json 2> $OUT/seq11.err; echo "seq11 rc=$?"
timeout -k 10 300 python bench.py --bin 22 --no-exact --no-pcie --cpu-frames 200 > $OUT/chunks_bin22.json 2> $OUT/bin22.err; echo "bin22 rc=$?"
timeout -k 10 300 python bench.py --bin 11 --no-exact --no-pcie --no-cpu > $OUT/chunks_bin11.json 2> $OUT/bin11.err; echo "bin11 rc=$?"
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "frames", d.get("frames_processed"), d["config"].get("mode"), "kp", d["config"].get("mean_keypoints_per_image"))
    except Exception as e:
        print(f, "FAILED", e)
PY
