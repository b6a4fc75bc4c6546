"""A/B of K1 (UTM forward, 1e8 points) for the library named by GSF_LIBRARY."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gps_optimize_slam_amd import batch as B
dev = "cuda"; g = torch.Generator(device=dev); g.manual_seed(1)
nb, n = 100_000, 1000
lat = 49.03 + 0.02 * (torch.rand(nb * n, dtype=torch.float64, device=dev, generator=g) - 0.5)
lon = 8.39 + 0.02 * (torch.rand(nb * n, dtype=torch.float64, device=dev, generator=g) - 0.5)
offs = torch.arange(0, nb * n + 1, n, dtype=torch.int64, device=dev)
e, nn, zone, south = B.utm_forward_batch(lat, lon, offs)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print(sys.argv[1] if len(sys.argv) > 1 else "?", "K1 fwd ms", round(timed(lambda: B.utm_forward_batch(lat, lon, offs, zone, south)), 4), "checksum", float(e.sum() + nn.sum()))
