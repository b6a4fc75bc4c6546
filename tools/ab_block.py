"""A/B of the workgroup-per-trajectory kernel (gsf_set_option block_kernel=1) against the wave-per-trajectory kernel (=0) on one box:
outputs compared (orientations and status bit for bit, positions to 1e-9 m) and kernel times of both, K4 alone and the fused pipeline.
usage: python tools/ab_block.py [quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gps_optimize_slam_amd import batch as B  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def run(ctx, bt, which, block):
    ctx.set_option("block_kernel", block)
    o = B.FusedPoses(0, bt.B, bt.N, "cuda")
    r = (B.fuse_pipeline_batch if which == "pipe" else B.ekf_fuse_batch)(bt, out=o)
    torch.cuda.synchronize()
    return o, r


def compare(ctx, Bn, N, seed, which):
    bt = B.TrajectoryBatch.synthetic(Bn, N, layout=0, seed=seed)
    o0, r0 = run(ctx, bt, which, 0)
    o1, r1 = run(ctx, bt, which, 1)
    p0, q0 = o0.pos.cpu().numpy(), o0.quat.cpu().numpy()
    p1, q1 = o1.pos.cpu().numpy(), o1.quat.cpu().numpy()
    nan_same = np.array_equal(np.isnan(p0), np.isnan(p1))
    dp = np.nanmax(np.abs(p0 - p1)) if np.isfinite(p0).any() else 0.0
    dq = np.nanmax(np.abs(q0 - q1)) if np.isfinite(q0).any() else 0.0
    st_same = bool(np.array_equal(o0.status.cpu().numpy(), o1.status.cpu().numpy()))
    if which == "pipe":
        dR = max(float(np.nanmax(np.abs(a.cpu().numpy() - c.cpu().numpy()))) if np.isfinite(a.cpu().numpy()).any() else 0.0 for a, c in zip(r0[1:], r1[1:]))
        print(f"    fit: max|d(R,t,s)|={dR:.3e}", flush=True)
    print(f"cmp {which:5s} B={Bn:6d} N={N:5d}: max|dp|={dp:.3e} max|dq|={dq:.3e} nan_same={nan_same} status_same={st_same}", flush=True)
    return dp, dq, nan_same, st_same


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ctx = B.context()
    bad = 0
    for which in ("ekf", "pipe"):
        for (Bn, N, seed) in ((1000, 271, 20250523), (64, 65, 3), (300, 128, 4), (200, 640, 5), (2000, 1000, 6), (77, 1024, 7), (50, 333, 8)):
            dp, dq, ns, ss = compare(ctx, Bn, N, seed, which)
            if not (dp < 1e-8 and dq < 1e-12 and ns and ss is not False):
                bad += 1
    print("MISMATCHES", bad, flush=True)
    res = []
    bt = B.TrajectoryBatch.synthetic(1000, 271, layout=0, seed=20250523)
    o = B.FusedPoses(0, 1000, 271, "cuda")
    for blk in (0, 1):
        ctx.set_option("block_kernel", blk)
        res.append((f"c2_ekf_blk{blk}", timed(lambda: B.ekf_fuse_batch(bt, out=o), 500)))
        res.append((f"c2_pipe_blk{blk}", timed(lambda: B.fuse_pipeline_batch(bt, out=o), 500)))
    del bt, o
    if not quick:
        bt = B.TrajectoryBatch.synthetic(100_000, 1000, layout=0, seed=1)
        o = B.FusedPoses(0, 100_000, 1000, "cuda")
        for blk in (0, 1):
            ctx.set_option("block_kernel", blk)
            res.append((f"c3_ekf_blk{blk}", timed(lambda: B.ekf_fuse_batch(bt, out=o), 10)))
            res.append((f"c3_pipe_blk{blk}", timed(lambda: B.fuse_pipeline_batch(bt, out=o), 10)))
    for k, v in res:
        print(f"{k:18s} {v:10.2f} us", flush=True)


if __name__ == "__main__":
    main()
