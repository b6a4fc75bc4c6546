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
        if backend is None:                     # fewer GPUs than ranks (a one-GPU box): gloo rehearsal, ranks share the GPUs
            ngpu = torch.cuda.device_count()
            backend = os.environ.get("GSF_DIST_BACKEND") or ("nccl" if ngpu >= world else "gloo")
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


def _staged():
    """gloo moves host memory: device tensors are staged through the CPU (the rehearsal path on a box with fewer GPUs than ranks)"""
    return dist.get_backend() == "gloo"


def all_gather_flat(out, x):
    """out[world * x.numel()] <- every rank's contiguous x (RCCL on device tensors; CPU-staged under gloo)"""
    if _staged() and x.is_cuda:
        oc = torch.empty(out.shape, dtype=out.dtype, device="cpu")
        dist.all_gather_into_tensor(oc, x.cpu())
        out.copy_(oc)
    else:
        dist.all_gather_into_tensor(out, x)
    return out


def all_reduce(t, op=None):
    """in-place all-reduce (default SUM) of a tensor; CPU-staged under gloo"""
    op = op or dist.ReduceOp.SUM
    if _staged() and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def _gather(x, world, out=None):
    """one all-gather; output is the concatenation along dim 0, returned viewed as (world, ...)"""
    x = x.contiguous()
    if out is None:
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    all_gather_flat(out, x)
    return out.view((world,) + tuple(x.shape))


def allgather_poses(pos, quat, chunk_trajs=None, sink=None, traj_dim=0):
    """All-gather the fused poses of every rank.  pos/quat: tensors whose rank-local shape is identical on all ranks (pad the
    last shard) with the trajectory axis at `traj_dim`: 0 for trajectory-major (B, N, 3)/(B, N, 4), -1 for time-major
    (N, 3, B)/(N, 4, B).  Returns (world, ...) tensors of the rank-local shapes when chunk_trajs is None; otherwise walks the
    trajectory axis in chunks of chunk_trajs and hands each gathered chunk to `sink(k, pos_all, quat_all)` (e.g. a checksum / error
    reduction / host drain) so the receive buffer stays chunk-sized (C5's 560 GB result cannot be gathered in one
    piece), and returns None."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    for name, x in (("pos", pos), ("quat", quat)):
        if not (-x.dim() <= traj_dim < x.dim()):
            raise ValueError(f"allgather_poses: traj_dim {traj_dim} out of range for {name} of shape {tuple(x.shape)}")
    if pos.shape[traj_dim] != quat.shape[traj_dim]:
        raise ValueError("allgather_poses: pos and quat disagree on the number of trajectories")

    def gather(x):
        return x.unsqueeze(0) if world == 1 else _gather(x, world)

    if chunk_trajs is None:
        return gather(pos), gather(quat)
    n = pos.shape[traj_dim]
    for k, lo in enumerate(range(0, n, int(chunk_trajs))):
        ln = min(n, lo + int(chunk_trajs)) - lo
        pa, qa = gather(pos.narrow(traj_dim, lo, ln).contiguous()), gather(quat.narrow(traj_dim, lo, ln).contiguous())
        if sink is not None:
            sink(k, pa, qa)
    return None


class PoseCollector:
    """The collect step with the library's own RCCL communicator (include/gsf.h: gsf_comm_*, gsf_allgather_poses): one process
    per GPU, the 128-byte id travels through torch.distributed's store.  mode 0 = one ncclAllGather, mode 1 = direct exchange with
    every peer (all seven xGMI links of a GPU busy at once).  Launches are asynchronous on `stream` (a torch.cuda.Stream)."""

    def __init__(self, device, stream=None):
        import ctypes as C

        from . import _lib
        if not dist.is_initialized():
            raise RuntimeError("PoseCollector needs an initialised torch.distributed process group (to ship the RCCL id)")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._L = _lib.load()
        v = C.c_int32(0)
        _lib.check(self._L.gsf_comm_rccl_version(C.byref(v)))
        self.rccl_version = int(v.value)                         # e.g. 22105: recorded by the bench next to the leg it ran
        stream = stream or torch.cuda.current_stream(device)
        self.stream = stream
        self.ctx = _lib.Context(torch.device(device).index or 0, stream.cuda_stream)
        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            _lib.check(self._L.gsf_comm_unique_id(buf))
            ident[0] = bytes(buf)
        dist.broadcast_object_list(ident, src=0)
        comm = C.c_void_p()
        idbuf = (C.c_uint8 * 128).from_buffer_copy(ident[0])
        _lib.check(self._L.gsf_comm_init_rank(self.ctx.handle, idbuf, self.world, self.rank, C.byref(comm)))
        self._comm = comm

    def allgather(self, send, recv, mode=1, chunk_count=0):
        """recv[world * send.numel()] <- every rank's `send` (contiguous float64 device tensors), asynchronously on self.stream"""
        import ctypes as C

        from . import _lib
        if send.dtype != torch.float64 or recv.dtype != torch.float64 or not send.is_contiguous() or not recv.is_contiguous():
            raise ValueError("PoseCollector.allgather: contiguous float64 tensors expected")
        if recv.numel() != self.world * send.numel():
            raise ValueError("PoseCollector.allgather: recv must hold world * send.numel() elements")
        _lib.check(self._L.gsf_allgather_poses(self.ctx.handle, self._comm, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                               send.numel(), int(mode), int(chunk_count)))

    def close(self):
        if getattr(self, "_comm", None):
            self.stream.synchronize()
            self._L.gsf_comm_destroy(self._comm)
            self._comm = None


def max_over_ranks(x, device):
    """max of a python float over ranks (the bench's step time)"""
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device="cpu" if _staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    if dist.is_initialized():
        if not _staged() and device is not None and torch.device(device).type == "cuda":
            dist.barrier(device_ids=[torch.device(device).index or 0])
        else:
            dist.barrier()
