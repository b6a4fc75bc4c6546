"""A/B of the C4 kernels (1M windows x 50 pairs): one fused launch (default) vs moments + finalize launches (ekf_variant 9); results compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gps_optimize_slam_amd import batch as B
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ctx = B.context()
for nw, W in ((1_000_000, 50), (1_000_000, 20), (200_000, 271), (4096, 50)):
    src, dst, Rp, tp, sp = bench.planted_windows(torch, nw, W, 7)
    res = {}
    for var in (9, 0):
        ctx.set_option("ekf_variant", var)
        res[var] = B.sim3_umeyama_batch(src, dst)
        ms = timed(lambda: B.sim3_umeyama_batch(src, dst), 10)
        alg = nw * (48 * W + 104)
        print(f"nw={nw} W={W} variant {var}: {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s  frac {alg / ms / 1e6 / 8000:.3f}", flush=True)
    same = all(torch.equal(a, b) for a, b in zip(res[9], res[0]))
    print("   identical results:", same, flush=True)
    del src, dst
ctx.set_option("ekf_variant", 0)
