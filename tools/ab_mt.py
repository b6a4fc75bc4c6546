"""A/B of the device-side np.random.choice draws: one stream (C1 shape) and 1 000 streams, 1 000 trials of permutation(271)[:4]."""
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
st1 = B.mt19937_seed([7]); st = B.mt19937_seed(np.arange(1000))
one = timed(lambda: B.mt19937_choice_batch(st1, [271], 1000, 4))
many = timed(lambda: B.mt19937_choice_batch(st, [271] * 1000, 1000, 4))
six = timed(lambda: B.mt19937_choice_batch(st1, [150], 600, 6))
np.random.seed(7); ref = np.stack([np.random.choice(271, 4, replace=False) for _ in range(50)])
got = B.mt19937_choice_batch(B.mt19937_seed([7]), [271], 50, 4).cpu().numpy()[0]
print(sys.argv[1] if len(sys.argv) > 1 else "?", f"1 stream x1000 trials {one:.3f} ms   1000 streams {many:.3f} ms   600 trials of permutation(150)[:6] {six:.3f} ms   exact {bool((got == ref).all())}")
