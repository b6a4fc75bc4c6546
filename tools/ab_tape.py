"""Chip-wide draws (csrc/gsf_rng_tape.hip) against NumPy and against the one-wave route: sample sets and final generator states for a
spread of (seed, n, trials, k, streams), then timings of the C1 shape (ONE stream, 1 000 trials of permutation(271)[:4]).
usage: ab_tape.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def numpy_draws(seed, n, trials, k, skip=0):
    np.random.seed(seed)
    if skip: np.random.random(skip)
    ref = np.stack([np.random.choice(n, k, replace=False) for _ in range(trials)])
    key, pos = np.random.get_state()[1:3]
    return ref, key.copy(), int(pos)


def state_after_skip(seed, skip):
    np.random.seed(seed)
    if skip: np.random.random(skip)
    return B.mt19937_from_numpy()


def check(cases, tape):
    B.context().set_option("tape_draws", -1 if tape else 0)
    bad = 0
    for seeds, ns, trials, k, skip in cases:
        st = torch.cat([state_after_skip(s, skip) for s in seeds], dim=0).contiguous()
        idx = B.mt19937_choice_batch(st, list(ns), trials, k).cpu().numpy()
        got = st.cpu().numpy().view(np.uint32)
        for b, (s, n) in enumerate(zip(seeds, ns)):
            if n < k:
                continue
            ref, key, pos = numpy_draws(s, n, trials, k, skip)
            ok = (idx[b] == ref).all() and (got[b, :624] == key).all() and int(got[b, 624]) == pos
            if not ok:
                bad += 1
                first = int(np.nonzero((idx[b] != ref).any(axis=1))[0][0]) if (idx[b] != ref).any() else -1
                print(f"  MISMATCH tape={tape} seed {s} n {n} trials {trials} k {k} skip {skip}: first bad trial {first}, pos {int(got[b, 624])} vs {pos}")
    B.context().set_option("tape_draws", -1)
    return bad


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
cases = [((7,), (271,), 1000, 4, 0), ((11,), (271,), 1000, 4, 311),           # pos = 624 on entry / mid-block entry
         ((3, 4, 5), (271, 150, 64), 400, 6, 17),                           # several streams, different populations
         ((21,), (2,), 20000, 1, 5), ((22,), (3,), 9000, 2, 0),              # tiny populations: a trial ends every output or two
         ((23,), (65,), 700, 4, 1), ((24,), (129,), 300, 64, 2),             # power-of-two borders, k = 64
         ((25,), (2040,), 40, 4, 0), ((26,), (1730,), 120, 8, 623),          # the largest tables
         ((27, 28), (300, 2041), 100, 4, 0)]                                 # a stream above the bound -> one-wave route inside the same call (n_max = 2041: whole call serial)
if not quick:
    cases += [((100 + i,), (int(n),), int(t), int(k), int(sk)) for i, (n, t, k, sk) in enumerate(
        zip(np.random.default_rng(5).integers(2, 2041, 24), np.random.default_rng(6).integers(30, 1500, 24),
            np.random.default_rng(7).integers(1, 9, 24), np.random.default_rng(8).integers(0, 2000, 24))) if n >= k]
t0 = time.time()
for tape in (True, False):
    print(f"tape={tape}: {check(cases, tape)} mismatching streams of {sum(len(c[0]) for c in cases)}  ({time.time() - t0:.1f} s)", flush=True)

st1 = B.mt19937_seed([7])
for tape in (0, -1):
    B.context().set_option("tape_draws", tape)
    one = timed(lambda: B.mt19937_choice_batch(st1, [271], 1000, 4))
    four = timed(lambda: B.mt19937_choice_batch(B.mt19937_seed([1, 2, 3, 4]), [271] * 4, 1000, 4))
    big = timed(lambda: B.mt19937_choice_batch(st1, [1000], 1000, 4), reps=3)
    print(f"tape_draws={tape}: 1 stream x 1000 trials of permutation(271)[:4] {one * 1e3:.1f} us;  4 streams {four * 1e3:.1f} us;  permutation(1000) {big * 1e3:.1f} us", flush=True)
B.context().set_option("tape_draws", -1)
