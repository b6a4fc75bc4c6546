import sys, time; sys.path.insert(0, '.')
import torch
from gps_optimize_slam_amd import batch as B
def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for nb, N in ((10000, 1000), (3000, 640)):
    bt = B.TrajectoryBatch.synthetic(nb, N, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
    fo = B.ekf_fuse_batch(bt)
    far = fo.pos + torch.tensor([4.0e5, 5.4e6, 100.0], dtype=torch.float64, device="cuda")
    print(nb, N, "near", round(timed(lambda: B.eval_errors_batch(bt.ts, fo.pos, bt.gps, bt.valid, 5.0), 5), 3), "ms; far (a track in another frame)", round(timed(lambda: B.eval_errors_batch(bt.ts, far, bt.gps, bt.valid, 5.0), 5), 3), "ms")
