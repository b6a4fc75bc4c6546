"""A/B of K2 (C4 windows) and K3 for the library named by GSF_LIBRARY."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gps_optimize_slam_amd import batch as B
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
dev = "cuda"; g = torch.Generator(device=dev); g.manual_seed(1)
nw, W = 1_000_000, 50
src = torch.cumsum(torch.randn(nw, W, 3, dtype=torch.float64, device=dev, generator=g) * torch.tensor([0.05, 0.03, 1.4], dtype=torch.float64, device=dev), dim=1)
dst = 1.05 * src + torch.tensor([4.5e5, 5.4e6, 100.0], dtype=torch.float64, device=dev) + 0.45 * torch.randn(nw, W, 3, dtype=torch.float64, device=dev, generator=g)
k2 = timed(lambda: B.sim3_umeyama_batch(src, dst))
R0 = B.sim3_umeyama_batch(src, dst)[0][:4].cpu()
del src, dst
nb, n = 100_000, 1000
offs = torch.arange(0, nb * n + 1, n, dtype=torch.int64, device=dev)
pos = torch.randn(nb * n, 3, dtype=torch.float64, device=dev, generator=g); quat = torch.randn(nb * n, 4, dtype=torch.float64, device=dev, generator=g)
R = torch.eye(3, dtype=torch.float64, device=dev).reshape(1, 9).repeat(nb, 1).contiguous(); t = torch.zeros(nb, 3, dtype=torch.float64, device=dev); s = torch.ones(nb, dtype=torch.float64, device=dev)
k3 = timed(lambda: B.apply_sim3_batch(pos, quat, offs, R, t, s))
print(f"{sys.argv[1] if len(sys.argv) > 1 else '?':6s} K2_C4 {k2*1e3:8.1f} us ({nw*(48*W+104)/k2/1e6:7.0f} GB/s)   K3 {k3*1e3:8.1f} us ({nb*n*112/k3/1e6:7.0f} GB/s)   R0 checksum {float(R0.sum()):.15f}")
