mkdir -p gpurun_out/r2c
timeout -k 10 300 python bench.py --mode sequences --steps 60 --warmup 3 > gpurun_out/r2c/seq_n1.json 2> gpurun_out/r2c/seq_n1.err; echo seq rc=$?
VSLAM_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --streams 48 > gpurun_out/r2c/chunks_n2_gloo.json 2> gpurun_out/r2c/chunks_n2_gloo.err; echo chunks2 rc=$?
VSLAM_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 4 --mode sequences --steps 40 --warmup 2 > gpurun_out/r2c/seq_n4_gloo.json 2> gpurun_out/r2c/seq_n4_gloo.err; echo seq4 rc=$?
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-exact --no-pcie --cpu-threads 64 > gpurun_out/r2c/bench_20.json 2> gpurun_out/r2c/bench_20.err; echo b20 rc=$?
tail -n 3 gpurun_out/r2c/*.err; cat gpurun_out/r2c/*.json
