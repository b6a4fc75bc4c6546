import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
nt, npts, trials, thr = 1000, 271, 1000, 4.0
_r = np.random.default_rng(0)
bt = B.TrajectoryBatch.synthetic(nt, npts, layout=B.LAYOUT_TRAJ_MAJOR, seed=3)
src = bt.pos.reshape(nt * npts, 3).contiguous(); g3 = bt.gps.reshape(nt * npts, 3)
plaus = torch.where(torch.isnan(g3), src + torch.nanmean(g3 - src, dim=0, keepdim=True), g3).contiguous()
offs = torch.arange(0, nt * npts + 1, npts, dtype=torch.int64, device="cuda")
idx = torch.as_tensor(np.stack([np.stack([_r.permutation(npts)[:4] for _ in range(trials)]) for _ in range(8)]).astype(np.int32)).cuda().repeat(nt // 8, 1, 1).contiguous()
nanmask = torch.isnan(g3).any(dim=1).reshape(nt, npts)
def run(name, dst):
    t = timed(lambda: B.sim3_ransac_batch(src, dst, offs, idx, thr, 4))
    print(f"{name:60s} {t*1e3:8.1f} us", flush=True)
run("plausible", plaus)
zf = torch.nan_to_num(g3, nan=0.0).contiguous(); run("zero-filled (5 % of rows)", zf)
z1 = zf.clone().reshape(nt, npts, 3); z1[:, 0] = plaus.reshape(nt, npts, 3)[:, 0]; run("zero-filled, first row of every set plausible", z1.reshape(-1, 3).contiguous())
one = plaus.clone().reshape(nt, npts, 3); one[:, 100] = 0.0; run("ONE zero row per set (row 100)", one.reshape(-1, 3).contiguous())
# zero rows never sampled: move sample indices off the zero rows
nm = nanmask.cpu().numpy(); ix = idx.cpu().numpy().copy()
for b in range(nt):
    bad = nm[b][ix[b]]
    good_rows = np.where(~nm[b])[0]
    ix[b][bad] = good_rows[_r.integers(0, len(good_rows), size=int(bad.sum()))]
idx2 = torch.as_tensor(ix).cuda()
t = timed(lambda: B.sim3_ransac_batch(src, z1.reshape(-1, 3).contiguous(), offs, idx2, thr, 4)); print(f"{'zero-filled, first row ok, zero rows never SAMPLED':60s} {t*1e3:8.1f} us")
