"""Multi-GPU driver: trajectories are independent, so they shard embarrassingly -- contiguous blocks of trajectory ids per
rank, one process per GPU, no collective on the data path.  The single collective is the collect step the north star
names: an all-gather of the fused poses (RCCL over xGMI; torch.distributed backend "nccl" IS RCCL on ROCm), issued per
chunk so that the receive buffer stays bounded (C5's 560 GB result cannot be gathered in one piece)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world, local_rank); initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:                     # GSF_DIST_BACKEND=gloo: rehearse the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("GSF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of trajectory ids for `rank`; blocks differ by at most one trajectory."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _gather(x, world, out=None):
    """one all_gather_into_tensor; output is the concatenation along dim 0, returned viewed as (world, ...)"""
    x = x.contiguous()
    if out is None:
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x)
    return out.view((world,) + tuple(x.shape))


def allgather_poses(pos, quat, chunk_trajs=None, sink=None):
    """All-gather the fused poses of every rank.  pos/quat: trajectory-LAST or trajectory-FIRST tensors whose rank-local
    shape is identical on all ranks (pad the last shard).  Gathers whole tensors when chunk_trajs is None; otherwise walks
    the trajectory axis (dim 0) in chunks of chunk_trajs and hands each gathered chunk to `sink(k, pos_all, quat_all)`
    (e.g. a checksum / ATE reduction / host drain) so the receive buffer stays chunk-sized.
    Returns the gathered (world, ...) tensors in the unchunked form, None otherwise."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if chunk_trajs is None:
        if world == 1:
            return pos.unsqueeze(0), quat.unsqueeze(0)
        return _gather(pos, world), _gather(quat, world)
    n = pos.shape[0]
    for k, lo in enumerate(range(0, n, chunk_trajs)):
        hi = min(n, lo + chunk_trajs)
        pc, qc = pos[lo:hi].contiguous(), quat[lo:hi].contiguous()
        if world == 1:
            pa, qa = pc.unsqueeze(0), qc.unsqueeze(0)
        else:
            pa, qa = _gather(pc, world), _gather(qc, world)
        if sink is not None:
            sink(k, pa, qa)
    return None


def max_over_ranks(x, device):
    """max of a python float over ranks (the bench's step time)"""
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and torch.device(device).type == "cuda":
            dist.barrier(device_ids=[torch.device(device).index or 0])
        else:
            dist.barrier()
