"""Finds the tracks of synthetic batches whose fit took the Jacobi-SVD fallback (status bit 16 << 8) and saves their rows for a CPU analysis
(tests/host_harness: which test of umeyama_rotation_polar declined them).  usage: python tools/dump_fallbacks.py out.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B
rows = {}
tot = 0
for variant in (0, 1):
    for seed in (20250523, 1, 2, 3):
        for N in (271, 1000):
            bt = B.TrajectoryBatch.synthetic(4000, N, layout=0, seed=seed, variant=variant)
            out, R, t, s = B.fuse_pipeline_batch(bt)
            st = out.status.cpu().numpy()
            fb = np.where((st >> 8) & 16)[0]
            tot += 4000
            print(f"variant {variant} seed {seed} N {N}: {len(fb)} fallbacks of 4000", flush=True)
            for b in fb[:40]:
                rows[f"v{variant}_s{seed}_n{N}_b{b}_ts"] = bt.ts[b].cpu().numpy()
                rows[f"v{variant}_s{seed}_n{N}_b{b}_pos"] = bt.pos[b].cpu().numpy()
                rows[f"v{variant}_s{seed}_n{N}_b{b}_gps"] = bt.gps[b].cpu().numpy()
                rows[f"v{variant}_s{seed}_n{N}_b{b}_valid"] = bt.valid[b].cpu().numpy()
np.savez_compressed(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fallbacks.npz", **rows)
print("saved", len(rows) // 4, "tracks")
