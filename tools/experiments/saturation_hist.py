"""How early does the robust Sim3 fit of the bench batch saturate (a trial that counts every row, EKFGPSSLAM.py:413)?  For T = 1, 2, 4, ... the
best inlier count over the first T drawn trials of every track (K2b on the first T sample sets of the full draw) against the row count."""
import sys
import torch
from gps_optimize_slam_amd import batch as B

nb, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 271
bt = B.TrajectoryBatch.synthetic(nb, N, layout=0, seed=20250523)
mask, n_rows, st = B.sim3_fit_rows_batch(bt.ts, bt.gps, bt.valid)
seeds = torch.arange(nb, dtype=torch.int64) + 1
state = B.mt19937_seed(seeds)
idx = B.mt19937_choice_batch(state, n_rows.clamp(min=0), 1000, 4)
# compacted rows per track, as the robust chain forms them
sel = mask.bool()
counts = sel.sum(1)
offs = torch.zeros(nb + 1, dtype=torch.int64, device="cuda"); offs[1:] = torch.cumsum(counts, 0)
src, dst = bt.pos[sel].contiguous(), bt.gps[sel].contiguous()
prev = torch.zeros(nb, dtype=torch.bool, device="cuda")
for T in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256, 512, 1000):
    R, t, s, stt, m, nin = B.sim3_ransac_batch(src, dst, offs, idx[:, :T].contiguous(), 4.0, 4)
    sat = nin.to(torch.int64) == counts
    print(f"T={T:5d}: saturated {int(sat.sum())}/{nb}  (new {int((sat & ~prev).sum())})  min best/n = {float((nin.double() / counts.double()).min()):.4f}")
    prev = sat
