"""What does the reference's step-6 metric (nearest-fix distance over ALL fixes + mean / median / RMSE, EKFGPSSLAM.py:1013-1033) cost at the
C3 track length, next to the fusion it grades?  10 000 x 1 000: the fused pipeline, then gsf_eval_errors_batch_dev on its output.
usage (GPU box): python tools/experiments/eval_long_tracks.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gps_optimize_slam_amd import batch as B


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for nb, N in ((10000, 1000), (10000, 271)):
    bt = B.TrajectoryBatch.synthetic(nb, N, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
    out = B.FusedPoses(bt.layout, nb, N, "cuda")
    ms_f = timed(lambda: B.fuse_pipeline_batch(bt, out=out), 5)
    ms_e = timed(lambda: B.eval_errors_batch(bt.ts, out.pos, bt.gps, bt.valid, 5.0), 3)
    print(f"{nb} x {N}: fused pipeline {ms_f:.3f} ms, error metric of one track set {ms_e:.3f} ms ({ms_e / ms_f:.1f} x the fusion; {nb * N * N / ms_e / 1e9:.2f} T pair distances / s)")
    del bt, out
