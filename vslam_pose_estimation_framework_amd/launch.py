"""One process per GPU: start N ranks of a script and relay rank 0's output (frame-sharded ranks + one pose all-gather).

Standard library only, and nothing here touches the GPU: the parent must not have initialised HIP when it starts the children (a
process that has may not be replaced or forked safely on this pool), so `bench.py --gpus N` calls this BEFORE it imports torch or
loads libvslam_hip.so.  The children are started through `python -m torch.distributed.run` (the launcher the driver itself uses),
rendezvous on 127.0.0.1."""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_command(script, argv, n_ranks, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), script] + list(argv)


def launch_ranks(script, argv, n_ranks, env=None, out=None):
    """Runs `script argv` as n_ranks ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run), copies the
    children's stdout (rank 0 prints the one JSON line) to `out` line by line as it arrives, returns the launcher's exit code."""
    if int(n_ranks) < 1:
        raise ValueError("n_ranks must be >= 1")
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver supports dmabuf IPC only (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = sys.stdout if out is None else out
    p = subprocess.Popen(rank_command(script, argv, n_ranks, free_port()), stdout=subprocess.PIPE, env=env, universal_newlines=True)
    try:
        for line in p.stdout:
            out.write(line)
            out.flush()
        return p.wait()
    except BaseException:
        p.kill()          # the exact child we started (its process group is torch.distributed.run's to clean up)
        p.wait()
        raise


def check_world(requested_gpus):
    """Under a launcher (WORLD_SIZE set): the rank count must be the one the command line asks for.  Returns WORLD_SIZE, or None
    when no launcher is present; raises SystemExit(2) on a mismatch."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is None:
        return None
    if int(ws) != int(requested_gpus):
        sys.stderr.write("WORLD_SIZE=%s but --gpus %d: start exactly one rank per GPU asked for\n" % (ws, requested_gpus))
        raise SystemExit(2)
    return int(ws)
