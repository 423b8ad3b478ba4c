"""bench.py --gpus N means N ranks: the self-launcher, on CPU with gloo."""
import io
import json
import os
import subprocess
import sys

import pytest

from vslam_pose_estimation_framework_amd import launch, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_launch_worker.py")


def test_launch_two_ranks_relays_rank0_line():
    out = io.StringIO()
    rc = launch.launch_ranks(WORKER, ["--gpus", "2"], 2, out=out)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"n_gpus": 2, "shape": [6, 157, 12], "rank_means": [0.0, 1.0]}


def test_world_size_must_equal_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=2 but --gpus 4" in p.stderr and p.stdout == ""


def test_bench_self_launches_before_touching_the_gpu():
    """`python bench.py --gpus 2` without a launcher: two ranks are started (on this CPU-only container each rank then ends with
    bench.py's "needs an MI355X" message — the parent relays the failure as a non-zero exit and prints no JSON line)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the two-rank run itself is a gpu-marked test")
    assert p.returncode != 0 and "needs an MI355X" in p.stderr and "{" not in p.stdout
    # the ranks were started by torch.distributed.run: its failure report names both local ranks (the second one may be terminated by
    # the launcher before it prints its own message, once the first has failed)
    assert "ChildFailedError" in p.stderr and "local_rank: 0" in p.stderr and "local_rank: 1" in p.stderr


def test_rank_command_is_the_drivers_launcher():
    cmd = launch.rank_command("bench.py", ["--gpus", "8", "--steps", "5"], 8, 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["bench.py", "--gpus", "8", "--steps", "5"]


def test_strong_plan_with_an_empty_rank_raises_on_every_rank():
    for rank in range(8):
        with pytest.raises(ValueError, match="without a chunk"):
            sharding.chunk_job(4541, 4, 6, rank, 8, "strong")
    # 157 live chunks over 8 ranks: 20 per rank, the last rank 17 — nobody is empty
    jobs = [sharding.chunk_job(4541, 160, 6, r, 8, "strong") for r in range(8)]
    assert [j["n_streams"] for j in jobs] == [20] * 7 + [17] and all(j["streams_padded"] == 20 for j in jobs)


@pytest.mark.gpu
def test_bench_gpus_2_gloo_rehearsal_on_one_card():
    """`python bench.py --gpus 2` under VSLAM_BENCH_BACKEND=gloo: two ranks on the one card of a test box, the whole N > 1 code path
    (weak chunk jobs, sums over ranks, the pose all-gather) with gloo in RCCL's place; the line says n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VSLAM_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--streams", "24"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["frames_processed"] == 2 * 4 * rec["config"]["streams_per_gpu"]
