"""The whole-run chain (steps 1-6) on 1 000-pose tracks: nothing in it may be quadratic in the track length by accident.
usage (GPU box): python tools/experiments/run_chain_long_tracks.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for nb, N in ((1000, 271), (1000, 640), (1000, 1000), (4000, 1000)):
    gb = B.GeodeticBatch.synthetic(nb, N, seed=20250523)
    st0 = B.mt19937_seed(np.arange(nb) + 1)
    ms = timed(lambda: B.run_fusion_batch(gb, st0.clone(), want_mask=False), 5)
    r = B.run_fusion_batch(gb, st0.clone(), want_mask=False)
    print(f"{nb} x {N}: {ms:.3f} ms = {nb * N / ms / 1e3:.0f} M poses/s; runs the reference would abort {int((r.run_status != 0).sum())}, "
          f"EKF RMSE {float(r.err_stats[2, :, 3].nanmean()):.2f} m, fixes {gb.gps_t.numel()}")
    del gb, r
