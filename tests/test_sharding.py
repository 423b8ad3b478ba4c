"""Frame-sharding: chunk planning, trajectory assembly and the pose all-gather (gloo, CPU)."""
import os
import socket

import numpy as np
import pytest

from vslam_pose_estimation_framework_amd import evaluation as ev
from vslam_pose_estimation_framework_amd import sharding


def random_se3(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return np.hstack([R, rng.normal(size=(3, 1))])


def test_plan_covers_every_frame_once():
    for total, n, ov in ((4541, 64, 10), (100, 7, 3), (10, 4, 5), (5, 8, 2)):
        plan, L = sharding.plan_chunks(total, n, ov)
        covered = []
        for (start, first, end) in plan:
            assert 0 <= start <= first <= end <= total and first - start <= ov
            covered += list(range(first, end))
        assert covered == list(range(total))


def test_sequence_assignment_lpt():
    """Exact mode: whole sequences per stream; KITTI 00-10 lengths on 8 and 4 ranks."""
    kitti = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]
    ranks, load = sharding.plan_sequences(kitti, 8)
    assert sorted(i for r in ranks for i in r) == list(range(11))
    assert max(load) == 4661 and sum(load) == sum(kitti)          # the longest sequence bounds the makespan
    ranks4, load4 = sharding.plan_sequences([kitti[i] for i in (0, 2, 5, 6)], 4)
    assert all(len(r) == 1 for r in ranks4) and sorted(load4) == [1101, 2761, 4541, 4661]
    ranks2, load2 = sharding.plan_sequences(kitti, 2)
    assert abs(load2[0] - load2[1]) <= min(kitti)                 # LPT balances two ranks to within the smallest job
    assert sharding.plan_sequences([], 3) == ([[], [], []], [0, 0, 0])


def test_assembly_is_exact_for_exact_chunks():
    rng = np.random.default_rng(3)
    total = 57
    G = [random_se3(rng)]
    for _ in range(total - 1):
        step = np.hstack([np.eye(3), rng.normal(scale=0.3, size=(3, 1))])
        G.append(ev.mul34(G[-1], step))
    G = np.array(G)
    plan, _ = sharding.plan_chunks(total, 5, 4)
    chunks = []
    for (start, first, end) in plan:
        W = random_se3(rng)                        # every chunk lives in its own world frame
        chunks.append(np.array([ev.mul34(W, G[f]) for f in range(start, end)]))
    for seam in (1, 3, 9):                         # seam transform from 1, 3 or all (4) warm-up frames: exact data, same answer
        A = sharding.assemble_trajectory(chunks, plan, seam_frames=seam)
        rel = ev.mul34(G[0], ev.inv34(A[0]))       # assembled trajectory is in chunk 0's frame
        for f in range(total):
            np.testing.assert_allclose(ev.mul34(rel, A[f]), G[f], atol=1e-9)
        assert ev.ate_rmse(A, G) < 1e-9
    # noisy warm-up estimates: the seam from several frames is a proper rigid transform and averages the noise
    noisy = [c.copy() for c in chunks]
    for c in noisy[1:]:
        c[:4, :, 3] += rng.normal(scale=0.05, size=(4, 3))
    e1 = ev.ate_rmse(sharding.assemble_trajectory(noisy, plan, seam_frames=1), G)
    e4 = ev.ate_rmse(sharding.assemble_trajectory(noisy, plan, seam_frames=4), G)
    A4 = sharding.assemble_trajectory(noisy, plan, seam_frames=4)
    for f in range(total):
        np.testing.assert_allclose(A4[f][:, :3] @ A4[f][:, :3].T, np.eye(3), atol=1e-9)
    assert e4 < e1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = torch.full((3, 5, 12), float(rank), dtype=torch.float64)
    local[:, :, 0] += torch.arange(3, dtype=torch.float64)[:, None]
    out = sharding.gather_poses(local)
    q.put((rank, out.numpy()))
    dist.destroy_process_group()


def test_pose_allgather_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        out = results[r]
        assert out.shape == (6, 5, 12)
        # rank-major order: chunks 0..2 from rank 0, 3..5 from rank 1
        assert (out[:3, :, 1] == 0).all() and (out[3:, :, 1] == 1).all()
        np.testing.assert_array_equal(out[:, 0, 0], [0, 1, 2, 1, 2, 3])
    np.testing.assert_array_equal(results[0], results[1])


def _id_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    uid = np.zeros(128, np.uint8)
    if rank == 0:
        uid[:] = (np.arange(128) * 7 + 3) % 251          # stands in for ncclGetUniqueId's bytes (needs a GPU)
    got = sharding.exchange_unique_id(uid, rank)
    # the send buffer layout of vslam_copy_poses_device: [streams][frames][12], gathered rank-major
    import torch
    send = torch.arange(2 * 3 * 12, dtype=torch.float64).reshape(2, 3, 12) + 1000.0 * rank
    allp = sharding.gather_poses(send).numpy()
    q.put((rank, got, allp))
    dist.destroy_process_group()


def test_comm_id_exchange_and_pose_layout_two_ranks_gloo():
    """What a two-process caller of the C-ABI collective does around it, on CPU: rank 0's 128-byte communicator id reaches rank 1
    unchanged, and the [streams][frames][12] send blocks come back rank-major (the order ncclAllGather gives)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (u, a) for r, u, a in (q.get(timeout=60) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = ((np.arange(128) * 7 + 3) % 251).astype(np.uint8)
    np.testing.assert_array_equal(res[0][0], want)
    np.testing.assert_array_equal(res[1][0], want)
    for r in range(2):
        a = res[r][1]
        assert a.shape == (4, 3, 12)
        np.testing.assert_array_equal(a[:2].ravel(), np.arange(72.0))
        np.testing.assert_array_equal(a[2:].ravel(), np.arange(72.0) + 1000.0)


def test_chunk_job_accounting_single_rank():
    """bench.py's chunk pipeline as numbers: over one whole chunk job every frame of the sequence is counted exactly once,
    whatever window of J steps is taken; the strong plan is the single-GPU plan cut into rank shares."""
    for total, B, ov in ((4541, 144, 10), (4541, 36, 40), (3682, 144, 10), (100, 7, 3)):
        job = sharding.chunk_job(total, B, ov)
        J = job["J"]
        live = sum(1 for (st, fi, en) in job["plan"] if en > fi)     # chunks behind the end of the sequence get no stream
        assert J == job["L"] + ov and job["n_streams"] == live <= B and job["plan"] == sharding.plan_chunks(total, B, ov)[0]
        assert live == -(-total // job["L"]) and job["streams_padded"] == live
        B = live
        for k0 in (0, 5, J, 3 * J + 1):
            assert sharding.chunk_job_unique_frames(job, k0, J) == total
        # every stream restarts exactly once per J steps, at the step its chunk frame index wraps to 0
        hits = [0] * B
        for k in range(1, J + 1):
            for s in sharding.chunk_job_restarts(job, k):
                hits[s] += 1
                assert (k + job["phase"][s]) % J == 0
        assert hits == [1] * B
    for world in (2, 4, 8):
        got = []
        for r in range(world):
            job = sharding.chunk_job(4541, 144, 10, r, world, "strong")
            assert job["n_streams"] <= job["streams_padded"] == -(-142 // world) and job["plan"] == sharding.plan_chunks(4541, 144, 10)[0]
            got += job["chunk_ids"]
        assert got == list(range(142))         # 4541 frames in chunks of 32: 142 live chunks of the 144 planned
    # a ragged strong plan (157 live chunks on 8 ranks: 7 x 20 + 17): every rank sends pose blocks of the same row length
    jobs = [sharding.chunk_job(4541, 160, 6, r, 8, "strong") for r in range(8)]
    assert [j["n_streams"] for j in jobs] == [20] * 7 + [17] and all(j["streams_padded"] == 20 for j in jobs)
    assert sum(sharding.chunk_job_unique_frames(j, 3, j["J"]) for j in jobs) == 4541


def _bench_accounting_worker(rank, world, port, scaling, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    job = sharding.chunk_job(4541, 144, 10, rank, world, scaling)
    K, k_first = job["J"], job["J"] + 2             # one whole chunk job timed after pre-roll + warm-up, as bench.py does
    t = torch.tensor([float(sharding.chunk_job_unique_frames(job, k_first, K)), float(job["n_streams"] * K)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)          # bench.py: sum_over_ranks(unique), sum_over_ranks(frames)
    ok = sharding.all_ranks_ok(rank == 0)            # one rank reporting a failure makes every rank see it
    ok_all = sharding.all_ranks_ok(True)
    # the [K][B][12] per-step pose blocks come back rank-major from the single all-gather
    send = torch.full((K, job["streams_padded"], 12), float(rank), dtype=torch.float64)
    allp = sharding.gather_poses(send)
    q.put((rank, t.tolist(), ok, ok_all, tuple(allp.shape), float(allp[K:].mean()) if world > 1 else 0.0))
    dist.destroy_process_group()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_accounting_two_ranks_gloo(scaling):
    """What `value` counts at N = 2 (bench.py --scaling weak | strong), on CPU: weak = two KITTI-00-shaped sequences (one per
    rank) of 4541 unique frames each, strong = the ONE sequence's 144 chunks split 72 / 72 -> 4541 unique frames in total."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_accounting_worker, args=(r, 2, port, scaling, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, (unique, frames), ok, ok_all, shape, mean_other in res:
        if scaling == "weak":
            assert unique == 2 * 4541 and frames == 2 * 142 * 42 and shape == (2 * 42, 142, 12)      # 142 live chunks of the 144 planned
        else:
            assert unique == 4541 and frames == 142 * 42 and shape == (2 * 42, 71, 12)
        assert ok is False and ok_all is True
        assert mean_other == 1.0


@pytest.mark.gpu
def test_c_abi_pose_allgather_single_rank_rccl():
    """vslam_comm_unique_id / vslam_comm_init / vslam_allgather_poses on the GPU: librccl.so is loaded by the library itself, a
    one-rank communicator gathers the context's own pose block (vslam_copy_poses_device) bit for bit."""
    import torch
    from vslam_pose_estimation_framework_amd import hip, synth
    api = hip.load()
    sy = synth.Synth()
    scene = sy.scene_kitti(seed=3)
    cfg = synth.config_for_scene(api, scene)
    cfg.max_keypoints, cfg.max_points, cfg.max_history_frames = 8192, 4096, 16
    B, K = 3, 5
    dev = torch.device("cuda", 0)
    stride = 1280
    L = torch.empty((K, B, cfg.rows, stride), dtype=torch.uint8, device=dev)
    R = torch.empty_like(L)
    for s in range(B):
        sy.render_device(scene, 10 * s, K, L[0, s].data_ptr(), R[0, s].data_ptr(), stride, B * cfg.rows * stride, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    api.create(cfg, 0, B)
    for k in range(K):
        api.process_device(L[k].data_ptr(), R[k].data_ptr(), stride, cfg.rows * stride)
    send = torch.zeros((B, K, 12), dtype=torch.float64, device=dev)
    api.copy_poses_device(0, K, send.data_ptr())
    api.synchronize()
    comm = sharding.PoseComm(api, 0, 1, 0)
    out = comm.allgather(send)
    comm.destroy()
    assert out.shape == (B, K, 12)
    assert torch.equal(out, send)
    host = np.stack([api.poses(s, 0, K).reshape(K, 12) for s in range(B)])
    np.testing.assert_array_equal(out.cpu().numpy(), host)
    api.destroy()


def test_chunked_oracle_trajectory_matches_sequential(oracle):
    """Chunks with a warm-up overlap, chained by sharding.assemble_trajectory, against the sequential run and
    the synthetic ground truth (CPU oracle; the device runs the same chunks as independent streams)."""
    from _oracle import Oracle
    o = Oracle()
    sc = o.scene_kitti(scale=0.4, seed=5)
    cfg = o.config_for_scene(sc)
    total, n_chunks, overlap = 30, 3, 6
    frames = [o.render(sc, k) for k in range(total)]
    gt = np.array([o.gt_pose(sc, k) for k in range(total)])
    o.create(cfg, 0, 1)
    for L, R in frames:
        o.process_host(L, R)
    seq = o.poses(0, 0, total)
    plan, _ = sharding.plan_chunks(total, n_chunks, overlap)
    chunks = []
    for (start, first, end) in plan:
        o.reset()
        for k in range(start, end):
            o.process_host(*frames[k])
        chunks.append(o.poses(0, 0, end - start))
    o.destroy()
    asm = sharding.assemble_trajectory(chunks, plan)
    ate_seq, ate_chunk = ev.ate_rmse(seq, gt), ev.ate_rmse(asm, gt)
    path = float(np.linalg.norm(gt[-1, :, 3] - gt[0, :, 3]))
    assert ate_seq < 0.01 * path and ate_chunk < 0.01 * path, (ate_seq, ate_chunk, path)
    assert abs(ate_chunk - ate_seq) < 0.005 * path
