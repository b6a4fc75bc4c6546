"""A/B of the fed-sample polynomial RANSAC (30 000 problems x 150 rows x 50 trials of 6) for the library named by GSF_LIBRARY."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B
dev = "cuda"; g = torch.Generator(device=dev); g.manual_seed(1); _r = np.random.default_rng(0)
P, n, trials, ms = 30_000, 150, 50, 6
t = (torch.arange(n, dtype=torch.float64, device=dev) * 0.1).repeat(P) + 0.01 * torch.rand(P * n, dtype=torch.float64, device=dev, generator=g)
y = 5.4e6 + 3.0 * t + 0.2 * t * t + 0.5 * torch.randn(P * n, dtype=torch.float64, device=dev, generator=g)
y = y + (torch.rand(P * n, dtype=torch.float64, device=dev, generator=g) < 0.1) * 100.0
offs = torch.arange(0, (P + 1) * n, n, dtype=torch.int64, device=dev)
idx = torch.as_tensor(np.argsort(_r.random((300, trials, n)), axis=2)[:, :, :ms].astype(np.int32)).to(dev).repeat(P // 300, 1, 1).contiguous()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms_t = timed(lambda: B.ransac_poly_batch(t, y, offs, idx, 2, 10.0))
m, ntr, nin, st = B.ransac_poly_batch(t, y, offs, idx, 2, 10.0)
print(sys.argv[1] if len(sys.argv) > 1 else "?", f"{ms_t:.3f} ms  checks {int(ntr.sum())} {int(nin.sum())} {int(st.sum())}")
