import os, sys
sys.path.insert(0, os.getcwd())
import torch
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
for Bn in (128, 256, 512, 768, 1000, 1536, 2000):
    bt = B.TrajectoryBatch.synthetic(Bn, 271, layout=0, seed=20250523)
    o = B.FusedPoses(0, Bn, 271, "cuda")
    row = []
    for blk in (0, 1):
        ctx.set_option("block_kernel", blk)
        row.append(timed(lambda: B.ekf_fuse_batch(bt, out=o), 300))
        row.append(timed(lambda: B.fuse_pipeline_batch(bt, out=o), 300))
    print(f"B={Bn:5d} N=271: wave ekf {row[0]:7.2f} pipe {row[1]:7.2f} | block ekf {row[2]:7.2f} pipe {row[3]:7.2f} us", flush=True)
