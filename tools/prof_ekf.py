"""Profiling driver (run under rocprofv3 on the GPU box): a few launches of K4 / the fused pipeline on one workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gps_optimize_slam_amd import batch as B
shape = sys.argv[1] if len(sys.argv) > 1 else "100000x1000"
which = sys.argv[2] if len(sys.argv) > 2 else "both"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nb, n = (int(x) for x in shape.split("x"))
bt = B.TrajectoryBatch.synthetic(nb, n, layout=B.LAYOUT_TIME_MAJOR, seed=1)
o = B.FusedPoses(bt.layout, nb, n, "cuda")
for _ in range(reps):
    if which in ("ekf", "both"):
        B.ekf_fuse_batch(bt, out=o)
    if which in ("pipeline", "both"):
        B.fuse_pipeline_batch(bt, out=o)
torch.cuda.synchronize()
print("done", shape, which, reps)
