"""C2-sized A/B of the kernel choices for 271-pose tracks: wave kernel, 5-wave workgroups, 4-wave workgroups with the tail pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
ctx = B.context()
def outs(bt, which, blk, var):
    ctx.set_option("block_kernel", blk); ctx.set_option("ekf_variant", var)
    o = B.FusedPoses(0, bt.B, bt.N, "cuda")
    r = (B.fuse_pipeline_batch if which == "pipe" else B.ekf_fuse_batch)(bt, out=o)
    torch.cuda.synchronize()
    return o.pos.cpu().numpy(), o.quat.cpu().numpy(), o.status.cpu().numpy()
for (Bn, N, seed) in ((1000, 271, 20250523), (500, 200, 3), (300, 150, 4), (200, 530, 5), (64, 129, 9)):
    bt = B.TrajectoryBatch.synthetic(Bn, N, layout=0, seed=seed)
    for which in ("ekf", "pipe"):
        p0, q0, s0 = outs(bt, which, 0, 0)
        for var in (7, 8):
            p1, q1, s1 = outs(bt, which, 1, var)
            print(f"cmp {which:4s} B={Bn} N={N} variant {var}: dp={np.nanmax(np.abs(p0-p1)):.2e} dq={np.nanmax(np.abs(q0-q1)):.2e} nan_same={np.array_equal(np.isnan(p0),np.isnan(p1))} status_same={np.array_equal(s0,s1)}", flush=True)
for Bn in (256, 768, 1000, 2000):
    bt = B.TrajectoryBatch.synthetic(Bn, 271, layout=0, seed=20250523)
    o = B.FusedPoses(0, Bn, 271, "cuda")
    row = []
    for blk, var in ((0, 0), (1, 7), (1, 0)):
        ctx.set_option("block_kernel", blk); ctx.set_option("ekf_variant", var)
        row.append(timed(lambda: B.ekf_fuse_batch(bt, out=o), 300))
        row.append(timed(lambda: B.fuse_pipeline_batch(bt, out=o), 300))
    print(f"B={Bn:5d} N=271: wave ekf {row[0]:6.2f} pipe {row[1]:6.2f} | block5 ekf {row[2]:6.2f} pipe {row[3]:6.2f} | block4+tail ekf {row[4]:6.2f} pipe {row[5]:6.2f} us", flush=True)
