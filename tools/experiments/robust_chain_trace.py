"""The robust chain (early exit on) and the whole-run chain at the C2 shape, a few dozen calls each: run under
   rocprofv3 --kernel-trace --stats -- python3 tools/experiments/robust_chain_trace.py   for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B
nb, N = 1000, 271
bt = B.TrajectoryBatch.synthetic(nb, N, layout=0, seed=20250523)
st0 = B.mt19937_seed(np.arange(nb))
o = B.FusedPoses(0, nb, N, "cuda")
for _ in range(40):
    B.fuse_pipeline_robust_batch(bt, st0.clone(), out=o, want_mask=False)
torch.cuda.synchronize()
gb = B.GeodeticBatch.synthetic(nb, N, seed=20250523)
st1 = B.mt19937_seed(np.arange(nb) + 1)
for _ in range(20):
    B.run_fusion_batch(gb, st1.clone(), want_mask=False)
torch.cuda.synchronize()
print("done")
