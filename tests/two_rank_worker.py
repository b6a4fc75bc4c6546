"""Helper of tests/test_gpu_parity.py::test_two_rank_real_kernels_gathered_equals_unsharded -- ONE rank of a 2-rank job on a
one-GPU box (gloo rehearsal: ranks share the GPU).  Each rank fuses ITS shard with the real kernels, the shards are collected
with the product's all-gather helpers, rank 0 compares with the unsharded result bit for bit (SURVEY 8e shard invariance)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from gps_optimize_slam_amd import batch as B  # noqa: E402
from gps_optimize_slam_amd import distributed as D  # noqa: E402


def main():
    out_path, nb, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    rank, world, _ = D.init_from_env()
    res = {"rank": rank, "world": world, "backend": torch.distributed.get_backend()}
    for layout, traj_dim in ((B.LAYOUT_TRAJ_MAJOR, 0), (B.LAYOUT_TIME_MAJOR, -1)):
        shard = B.TrajectoryBatch.synthetic(nb, N, layout=layout, seed=77, traj0=rank * nb)
        mine, R, t, s = B.fuse_pipeline_batch(shard)
        pa, qa = D.allgather_poses(mine.pos, mine.quat, traj_dim=traj_dim)                    # (world, ...) of the rank-local shapes
        sums = []
        D.allgather_poses(mine.pos, mine.quat, chunk_trajs=37, traj_dim=traj_dim,
                          sink=lambda k, p, q: sums.append(int(p.contiguous().view(torch.int64).sum().item()) + int(q.contiguous().view(torch.int64).sum().item())))
        flat = torch.empty((world * mine.buf.numel(),), dtype=torch.float64, device="cuda")
        D.all_gather_flat(flat, mine.buf)                                                      # the bench's one-call collect
        if rank == 0:
            full = B.TrajectoryBatch.synthetic(world * nb, N, layout=layout, seed=77, traj0=0)
            ref, _, _, _ = B.fuse_pipeline_batch(full)
            torch.cuda.synchronize()
            cat_dim = 0 if traj_dim == 0 else 2
            res[f"layout{layout}_pos_equal"] = bool(torch.equal(torch.cat(list(pa), dim=cat_dim), ref.pos))
            res[f"layout{layout}_quat_equal"] = bool(torch.equal(torch.cat(list(qa), dim=cat_dim), ref.quat))
            tot = (int(ref.pos.contiguous().view(torch.int64).sum().item()) + int(ref.quat.contiguous().view(torch.int64).sum().item()))
            res[f"layout{layout}_chunked_checksum_equal"] = (sum(sums) - tot) % (1 << 64) == 0
            P = nb * N
            blocks = flat.view(world, -1)
            ok = True
            for r in range(world):
                lo, hi = r * nb, (r + 1) * nb
                if layout == B.LAYOUT_TRAJ_MAJOR:
                    ok &= bool(torch.equal(blocks[r][:P * 3].view(nb, N, 3), ref.pos[lo:hi])) and bool(torch.equal(blocks[r][P * 3:].view(nb, N, 4), ref.quat[lo:hi]))
                else:
                    ok &= bool(torch.equal(blocks[r][:P * 3].view(N, 3, nb), ref.pos[:, :, lo:hi])) and bool(torch.equal(blocks[r][P * 3:].view(N, 4, nb), ref.quat[:, :, lo:hi]))
            res[f"layout{layout}_flat_blocks_equal"] = ok
            res[f"layout{layout}_finite"] = bool(torch.isfinite(ref.pos).all().item())
    D.barrier()
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
