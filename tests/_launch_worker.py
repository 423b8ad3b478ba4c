"""Rank body for tests/test_launch.py: what bench.py does around its measurement, on CPU (gloo)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from vslam_pose_estimation_framework_amd import launch, sharding  # noqa: E402

gpus = int(sys.argv[sys.argv.index("--gpus") + 1])
launch.check_world(gpus)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
job = sharding.chunk_job(4541, 160, 6, rank, world, "weak")
send = torch.full((3, job["streams_padded"], 12), float(rank), dtype=torch.float64)
allp = sharding.gather_poses(send)
dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "shape": list(allp.shape), "rank_means": [float(allp[3 * r:3 * r + 3].mean()) for r in range(world)]}), flush=True)
dist.destroy_process_group()
