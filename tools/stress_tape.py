"""Randomised campaign for the chip-wide draws (csrc/gsf_rng_tape.hip) and the K2b screen (csrc/gsf_sim3.hip), GPU box.
   part 1: CASES random (population n, trials, k, entry position, streams per call) against NumPy's np.random.choice: sample sets and
           final generator states, chip-wide route;
   part 2: random K2b problems (set sizes, thresholds from 1e-4 to 50 m, scales of the track, wild / NaN rows, rows planted at the
           threshold) with the single-precision screen on and off: every output identical.
usage: stress_tape.py [CASES] [SEED]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); bad = 0; routes = 0
for c in range(cases):
    nb = int(rng.integers(1, 5))
    ns = [int(v) for v in np.exp(rng.uniform(np.log(2), np.log(2040), size=nb)).astype(int)]
    ns = [max(2, min(2040, v)) for v in ns]
    k = int(rng.integers(1, min(min(ns), 64) + 1))
    trials = int(np.exp(rng.uniform(np.log(20), np.log(3000))))
    skip = int(rng.integers(0, 5000))
    seeds = [int(rng.integers(0, 2**31)) for _ in range(nb)]
    sts, refs = [], []
    for s_, n in zip(seeds, ns):
        np.random.seed(s_); np.random.random(skip)
        sts.append(B.mt19937_from_numpy())
        ref = np.stack([np.random.choice(n, k, replace=False) for _ in range(trials)])
        key, pos = np.random.get_state()[1:3]
        refs.append((ref, key.copy(), int(pos)))
    st = torch.cat(sts, dim=0).contiguous()
    idx = B.mt19937_choice_batch(st, ns, trials, k).cpu().numpy()
    got = st.cpu().numpy().view(np.uint32)
    routes += int(trials * max(ns) >= 16384)
    for b_, (ref, key, pos) in enumerate(refs):
        if not ((idx[b_] == ref).all() and (got[b_, :624] == key).all() and int(got[b_, 624]) == pos):
            bad += 1; print("MISMATCH", c, ns, k, trials, skip, seeds, flush=True)
print(f"draws: {cases} calls ({routes} of them large enough for the chip-wide route), {bad} mismatching streams, {time.time() - t0:.0f} s", flush=True)

t0 = time.time(); bad2 = 0
for c in range(max(20, cases // 4)):
    nt = int(rng.integers(1, 40)); npts = int(rng.integers(8, 700)); trials = int(rng.integers(64, 400))
    thr = float(10.0 ** rng.uniform(-4, 1.7)); scale = float(10.0 ** rng.uniform(-1, 2))
    bt = B.TrajectoryBatch.synthetic(nt, npts, layout=0, seed=int(rng.integers(0, 10**6)))
    src = (bt.pos.reshape(nt * npts, 3) * scale).contiguous()
    g3 = bt.gps.reshape(nt * npts, 3) * scale
    dst = torch.where(torch.isnan(g3), src + torch.nanmean(g3 - src, dim=0, keepdim=True), g3).contiguous()
    d = dst.reshape(nt, npts, 3)
    for b_ in range(nt):
        m = int(rng.integers(0, npts // 2))
        if m:
            rows = rng.choice(npts, size=m, replace=False)
            u = rng.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
            d[b_, rows] += torch.as_tensor(u * (thr + rng.normal(size=(m, 1)) * 10.0 ** rng.uniform(-10, -1, size=(m, 1))), device="cuda")
        for _ in range(int(rng.integers(0, 4))):
            r_ = int(rng.integers(0, npts)); kind = int(rng.integers(0, 4))
            if kind == 0: d[b_, r_] = 0.0
            elif kind == 1: d[b_, r_, int(rng.integers(0, 3))] = float("nan")
            elif kind == 2: d[b_, r_] *= 1e6
            else: d[b_, r_, 0] = float("inf")
    offs = torch.arange(0, nt * npts + 1, npts, dtype=torch.int64, device="cuda")
    idx = torch.as_tensor(np.stack([np.stack([rng.permutation(npts)[:4] for _ in range(trials)]) for _ in range(nt)]).astype(np.int32)).cuda()
    res = {}
    for scr in (1, 0):
        B.context().set_option("k2b_screen", scr)
        res[scr] = [o.cpu().numpy() for o in B.sim3_ransac_batch(src, dst, offs, idx, thr, 4)]
    B.context().set_option("k2b_screen", 1)
    if not all(np.array_equal(a, b_, equal_nan=True) for a, b_ in zip(res[1], res[0])):
        bad2 += 1; print("K2b MISMATCH", c, nt, npts, trials, thr, scale, flush=True)
print(f"K2b screen: {max(20, cases // 4)} random problems, {bad2} with any differing output, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad or bad2 else 0)
