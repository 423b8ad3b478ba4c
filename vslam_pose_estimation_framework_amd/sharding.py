"""Frame-sharding of a sequence into chunks (streams) across streams-per-GPU and GPUs, and trajectory
assembly from per-chunk poses.

The reference is a single process; frames of one sequence depend on each other (threshold feedback,
tracker state, landmarks), so a sequence is cut into contiguous chunks that start `overlap` frames early:
each chunk re-localises on its overlap frames and contributes only its own range.  Chunks are chained by
the rigid transform that maps the chunk's estimate of the last frame of the preceding range onto the
preceding chunk's estimate of that frame.  The only inter-GPU exchange is one all-gather of the per-chunk
3x4 poses (96 B/frame)."""
import numpy as np

from .evaluation import inv34, mul34


def plan_chunks(total_frames, n_chunks, overlap):
    """[(first_frame_processed, first_unique_frame, end_frame)] for every chunk."""
    L = -(-total_frames // n_chunks)
    plan = []
    for c in range(n_chunks):
        first_unique = min(c * L, total_frames)
        end = min((c + 1) * L, total_frames)
        start = max(0, first_unique - overlap)
        plan.append((start, first_unique, end))
    return plan, L


def chunk_job(seq_frames, streams, overlap, rank=0, world=1, scaling="weak"):
    """The steady-state chunk pipeline one rank of bench.py runs, as plain numbers.

    weak   every rank owns one whole `seq_frames`-frame sequence of its own (N GPUs = N KITTI-00-shaped drives), cut into
           `streams` chunks: per-GPU work is fixed.
    strong ONE `seq_frames`-frame sequence: the plan of `streams` chunks is the single-GPU plan and rank r runs chunks
           r*streams/world .. (r+1)*streams/world - 1 of it (the assembled trajectory does not depend on the GPU count).
    Chunks that lie behind the end of the sequence (ceil effects) are dropped: n_streams can be smaller than `streams`.
    Stream s is `phase[s]` frames into its chunk when the pipeline starts and restarts whenever its chunk job of J = L + overlap
    steps ends.  Returns a dict: L, J, n_streams (of this rank), chunk_ids (global chunk of every local stream), plan (the
    global (start, first_unique, end) list), starts / first_unique / end_unique / phase per local stream, streams_padded (the largest
    n_streams over the ranks: the row length of the pose blocks the ranks all-gather)."""
    if scaling not in ("weak", "strong"):
        raise ValueError("scaling must be weak or strong")
    n_chunks = int(streams)
    plan, L = plan_chunks(seq_frames, n_chunks, overlap)
    # chunks behind the end of the sequence are empty (4541 frames in 160 chunks of 29: chunks 157 .. 159) and get no stream: the
    # plan, and with it the assembled trajectory, is the one of `streams` chunks; only the live ones are run
    n_live = min(n_chunks, -(-int(seq_frames) // L))
    if scaling == "weak":
        ids = list(range(n_live))
        per = n_live
    else:
        per = -(-n_live // world)
        if (world - 1) * per >= n_live:
            # a rank without a chunk would create no context and leave its peers waiting in the collectives: every rank computes the
            # same plan, so every rank raises here, before anything is launched
            raise ValueError("strong chunk plan: %d live chunks (of %d planned) over %d ranks leaves rank %d and above without a chunk; "
                             "use more --streams or fewer GPUs" % (n_live, n_chunks, world, -(-n_live // per)))
        ids = list(range(rank * per, min((rank + 1) * per, n_live)))
    J = L + overlap
    B = len(ids)
    job = {"L": L, "J": J, "n_streams": B, "streams_padded": per, "chunk_ids": ids, "plan": plan, "scaling": scaling, "seq_frames": seq_frames,
           "starts": [], "first_unique": [], "end_unique": [], "phase": []}
    for s, c in enumerate(ids):
        job["starts"].append(max(0, c * L - overlap))         # every chunk job is J steps long: the same pipeline period for all
        job["first_unique"].append(c * L)
        job["end_unique"].append(min((c + 1) * L, seq_frames))
        job["phase"].append((s * J) // max(B, 1))
    return job


def chunk_job_unique_frames(job, k0, n):
    """Frames inside their chunk's own range among pipeline steps k0 .. k0+n-1 of this rank (what `value` counts)."""
    J = job["J"]
    cnt = 0
    for k in range(k0, k0 + n):
        for s in range(job["n_streams"]):
            f = job["starts"][s] + (k + job["phase"][s]) % J
            if job["first_unique"][s] <= f < job["end_unique"][s]:
                cnt += 1
    return cnt


def chunk_job_restarts(job, k):
    """Local streams whose chunk starts over at pipeline step k (k > 0)."""
    J = job["J"]
    return [s for s in range(job["n_streams"]) if (k + job["phase"][s]) % J == 0]


def plan_sequences(lengths, world):
    """Whole sequences onto `world` ranks, longest-processing-time first (the exact mode — every
    sequence runs start to end on one stream of one GPU, results identical to the sequential reference).
    Returns (per-rank lists of sequence indices, per-rank frame totals).  KITTI 00-10 on 8 ranks: makespan 4661."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    ranks = [[] for _ in range(world)]
    load = [0] * world
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        ranks[r].append(i)
        load[r] += int(lengths[i])
    return ranks, load


def _seam_anchor(G, P, start, first, seam_frames):
    """Rigid transform chunk-world -> global from the last `seam_frames` warm-up frames, which both the preceding range (G) and
    this chunk (P) estimate: chordal mean of the rotations G_f P_f^-1, then the translation that fits the camera centres."""
    k = max(1, min(seam_frames, first - start))
    fs = range(first - k, first)
    if k == 1:
        return mul34(G[first - 1], inv34(P[first - 1 - start]))
    M = np.zeros((3, 3))
    for f in fs:
        M += G[f][:, :3] @ P[f - start][:, :3].T
    U, _, Vt = np.linalg.svd(M)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    Rm = U @ D @ Vt
    t = np.mean([G[f][:, 3] - Rm @ P[f - start][:, 3] for f in fs], axis=0)
    return np.hstack([Rm, t.reshape(3, 1)])


def assemble_trajectory(chunk_poses, plan, seam_frames=1):
    """chunk_poses[c]: array [n_processed_c, 3, 4] (camera-to-chunk-world), plan from plan_chunks.
    Returns [total, 3, 4] in the first chunk's world frame.  seam_frames: how many of the chunk's last warm-up frames the
    seam transform is estimated from (1 = the frame before the chunk's own range alone)."""
    total = plan[-1][2]
    G = np.zeros((total, 3, 4))
    anchor = np.hstack([np.eye(3), np.zeros((3, 1))])
    for c, (start, first, end) in enumerate(plan):
        P = np.asarray(chunk_poses[c]).reshape(-1, 3, 4)
        if end <= first:
            continue
        if c > 0 and first > start:
            anchor = _seam_anchor(G, P, start, first, seam_frames)
        elif c > 0:
            anchor = mul34(G[first - 1], inv34(P[0])) if first > 0 else anchor
        for f in range(first, end):
            G[f] = mul34(anchor, P[f - start])
    return G


def gather_poses(local_poses, group=None):
    """All-gather of per-rank pose blocks [chunks_per_rank, frames, 12] (torch tensor, any device) ->
    [world*chunks_per_rank, frames, 12].  One collective per run; RCCL on GPUs, gloo in CPU tests."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_poses
    world = dist.get_world_size(group)
    out = torch.empty((world * local_poses.shape[0],) + tuple(local_poses.shape[1:]), dtype=local_poses.dtype,
                      device=local_poses.device)
    dist.all_gather_into_tensor(out, local_poses.contiguous(), group=group)
    return out


def exchange_unique_id(unique_id, rank, group=None):
    """Rank 0's communicator id (any byte string, 128 id bytes + status here) to every rank over the launcher's own channel
    (torch.distributed here; MPI or a file work as well).  unique_id: numpy uint8 array, filled on rank 0."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return unique_id
    t = torch.from_numpy(np.ascontiguousarray(unique_id, np.uint8).copy())
    on_gpu = dist.get_backend(group) == "nccl"
    if on_gpu:
        t = t.cuda()
    dist.broadcast(t, src=0, group=group)
    return t.cpu().numpy()


def all_ranks_ok(ok, group=None):
    """Logical AND of a per-rank status over the launcher's channel (torch.distributed; True without a process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class PoseComm(object):
    """The pose all-gather behind the C ABI (vslam_comm_init / vslam_allgather_poses: RCCL called by libvslam_hip.so itself),
    what a C++ caller of the library uses; `gather_poses` above is the Python convenience over torch.distributed."""

    def __init__(self, api, rank, world, device, group=None):
        import ctypes as C
        import numpy as np
        self.api, self.rank, self.world = api, rank, world
        lib = api.lib
        lib.vslam_comm_last_error.restype = C.c_char_p
        # every rank checks its local preconditions (librccl.so, the device) and the ranks agree on the result BEFORE anyone
        # enters the collective ncclCommInitRank: a rank that cannot take part would leave its peers waiting there for ever
        ok = lib.vslam_comm_available(C.c_int(device)) == 0
        why = "" if ok else lib.vslam_comm_last_error().decode()
        if not all_ranks_ok(ok, group):
            raise RuntimeError("vslam_comm_available failed on %s%s" % ("this rank: " if not ok else "another rank", why))
        # 128 id bytes + 4 status bytes in one broadcast: when rank 0 cannot make an id every rank learns it (and raises) instead
        # of waiting in a collective that rank 0 never joins
        msg = np.zeros(132, np.uint8)
        err = ""
        if rank == 0:
            rc = lib.vslam_comm_unique_id(msg.ctypes.data_as(C.c_void_p))
            if rc != 0:
                err = lib.vslam_comm_last_error().decode()
                msg[128:132] = np.frombuffer(np.int32(rc).tobytes(), np.uint8)
        msg = np.ascontiguousarray(exchange_unique_id(msg, rank, group), np.uint8)
        rc0 = int(np.frombuffer(msg[128:132].tobytes(), np.int32)[0])
        if rc0 != 0:
            raise RuntimeError("vslam_comm_unique_id on rank 0: %d %s" % (rc0, err))
        uid = np.ascontiguousarray(msg[:128])
        self.comm = C.c_void_p()
        rc = lib.vslam_comm_init(C.c_int(rank), C.c_int(world), uid.ctypes.data_as(C.c_void_p), C.c_int(device), C.byref(self.comm))
        why = "" if rc == 0 else lib.vslam_comm_last_error().decode()
        if not all_ranks_ok(rc == 0, group):     # ncclCommInitRank returned everywhere: one more agreement, so that all ranks raise together
            if rc == 0:
                lib.vslam_comm_destroy(self.comm)
                self.comm = None
            raise RuntimeError("vslam_comm_init failed on %s%s" % ("this rank: %d " % rc if rc != 0 else "another rank", why))

    def allgather(self, send):
        """send: contiguous float64 CUDA tensor; returns [world * send.shape[0], ...] ordered by rank (synchronised)."""
        import ctypes as C
        import torch
        send = send.contiguous()
        out = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        stream = torch.cuda.current_stream(send.device)
        rc = self.api.lib.vslam_allgather_poses(self.comm, C.c_void_p(send.data_ptr()), C.c_void_p(out.data_ptr()), C.c_size_t(send.numel()),
                                                C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError("vslam_allgather_poses: %d %s" % (rc, self.api.lib.vslam_comm_last_error().decode()))
        stream.synchronize()
        return out

    def destroy(self):
        if self.comm:
            self.api.lib.vslam_comm_destroy(self.comm)
            self.comm = None
