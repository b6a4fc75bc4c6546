"""Shader-clock stamps of the phases of eval_errors_lds_kernel (diagnostic build: make -C gps_optimize_slam_amd/csrc eval_timing;
GSF_LIBRARY=gps_optimize_slam_amd/libgsf_eval_timing.so python tools/experiments/eval_timing.py [B])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
b = B.TrajectoryBatch.synthetic(nb, 271, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
for _ in range(3):
    stats, err = B.eval_errors_batch(b.ts, b.pos, b.gps, b.valid, 5.0)
torch.cuda.synchronize()
full = err.cpu().numpy().reshape(nb, -1)
a, w = full[:, :8], full[:, 8:10]
names = ["start", "compaction", "tile set up", "-", "nearest fix (loads, pair loop, group min, errors)", "sums + barrier", "rank pass", "stats"]
prev = np.zeros(nb)
for k in (1, 2, 4, 5, 6, 7):
    print(f"{names[k]:50s} +{(a[:, k] - prev).mean():9.0f} cycles (at {a[:, k].mean():9.0f})")
    prev = a[:, k]
start, end = w[:, 0] - w[:, 0].min(), w[:, 1] - w[:, 0].min()
print(f"wall clock (10 ns ticks): kernel span {end.max() * 0.01:.1f} us; block starts: median {np.median(start) * 0.01:.1f} us, 90 % {np.percentile(start, 90) * 0.01:.1f} us, max {start.max() * 0.01:.1f} us; "
      f"block duration mean {(w[:, 1] - w[:, 0]).mean() * 0.01:.1f} us, max {(w[:, 1] - w[:, 0]).max() * 0.01:.1f} us; shader clock {a[:, 7].mean() / ((w[:, 1] - w[:, 0]).mean() * 10):.2f} GHz")
