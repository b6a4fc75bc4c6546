"""Profiling driver: a few launches of the wave-per-trajectory kernels (trajectory-major) on one workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gps_optimize_slam_amd import batch as B
shape = sys.argv[1] if len(sys.argv) > 1 else "1000x271"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nb, n = (int(x) for x in shape.split("x"))
bt = B.TrajectoryBatch.synthetic(nb, n, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
o = B.FusedPoses(bt.layout, nb, n, "cuda")
for _ in range(reps):
    B.ekf_fuse_batch(bt, out=o)
    B.fuse_pipeline_batch(bt, out=o)
torch.cuda.synchronize()
print("done", shape, reps)
