"""K2b (RANSAC hypotheses) with the single-precision screen on and off: same counts / masks / fits, and the time of 1 000 sets x 271
points x 1 000 fed trials (C2 shape), of sets with rows planted INSIDE the rounding band, and of one set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


dev = "cuda"
_r = np.random.default_rng(0)


def case(nt, npts, trials, thr, band_rows=0, seed=3, zero_fill=False):
    bt = B.TrajectoryBatch.synthetic(nt, npts, layout=B.LAYOUT_TRAJ_MAJOR, seed=seed)
    src = bt.pos.reshape(nt * npts, 3).contiguous(); g3 = bt.gps.reshape(nt * npts, 3)
    dst = torch.where(torch.isnan(g3), bt.pos.reshape(nt * npts, 3) + g3[0:1].nan_to_num(0.0) * 0.0 + torch.nanmean(g3 - bt.pos.reshape(nt * npts, 3), dim=0, keepdim=True), g3).contiguous()   # missing fixes: a plausible point (zeros would be 5 000 km away)
    if zero_fill:
        dst = torch.nan_to_num(g3, nan=0.0).contiguous()                  # the aux bench's input up to round 3: missing fixes 5 000 km away
    if band_rows:
        # rows pushed to within micrometres .. millimetres of the threshold sphere of the TRUE alignment: the screen must hand them on
        d = dst.reshape(nt, npts, 3)
        for b in range(nt):
            rows = _r.choice(npts, size=band_rows, replace=False)
            u = _r.normal(size=(band_rows, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
            d[b, rows] += torch.as_tensor(u * (thr + _r.normal(size=(band_rows, 1)) * 10.0 ** _r.uniform(-7, -2, size=(band_rows, 1))), device=dev)
    offs = torch.arange(0, nt * npts + 1, npts, dtype=torch.int64, device=dev)
    k = min(nt, 8)
    idx = torch.as_tensor(np.stack([np.stack([_r.permutation(npts)[:4] for _ in range(trials)]) for _ in range(k)]).astype(np.int32)).to(dev)
    idx = idx.repeat((nt + k - 1) // k, 1, 1)[:nt].contiguous()
    res = {}
    for scr in (1, 0):
        B.context().set_option("k2b_screen", scr)
        out = B.sim3_ransac_batch(src, dst, offs, idx, thr, 4)
        res[scr] = [o.cpu().numpy() for o in out] + [timed(lambda: B.sim3_ransac_batch(src, dst, offs, idx, thr, 4), reps=3)]
    B.context().set_option("k2b_screen", 1)
    zf = ' zero-filled' if zero_fill else ''
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[1][:-1], res[0][:-1]))
    print(f"{nt:5d} sets x {npts:4d} rows x {trials:4d} trials thr {thr:g} band rows {band_rows:3d}{zf}: screened {res[1][-1] * 1e3:8.1f} us  double {res[0][-1] * 1e3:8.1f} us  identical {same}", flush=True)
    return same


ok = True
ok &= case(1000, 271, 1000, 4.0)
ok &= case(1000, 271, 1000, 4.0, zero_fill=True)
ok &= case(1000, 271, 1000, 4.0, band_rows=40)
ok &= case(1000, 271, 1000, 0.05)                 # threshold inside the noise: many rows near it
ok &= case(1000, 271, 1000, 1e-4)                 # band wider than the threshold: everything re-checked
ok &= case(300, 1000, 500, 4.0, band_rows=100)
ok &= case(1, 271, 1000, 4.0, band_rows=40)
ok &= case(5, 1500, 300, 4.0)                     # above the LDS cap: double path
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
