"""A/B of the geodesy kernels (K1 forward / inverse, WGS84 -> ENU; 1e8 points) for the library named by GSF_LIBRARY."""
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
tag = sys.argv[1] if len(sys.argv) > 1 else "?"
print(tag, "K1 fwd ms", round(timed(lambda: B.utm_forward_batch(lat, lon, offs, zone, south)), 4), "checksum", float(e.sum() + nn.sum()))
la2, lo2 = B.utm_inverse_batch(e, nn, offs, zone, south)
print(tag, "K1 inv ms", round(timed(lambda: B.utm_inverse_batch(e, nn, offs, zone, south)), 4), "round trip max |d| deg", float(torch.maximum((la2 - lat).abs().max(), (lo2 - lon).abs().max())))
alt = 110.0 + 5.0 * torch.rand(nb * n, dtype=torch.float64, device=dev, generator=g)
ref = torch.stack([lat[offs[:-1]], lon[offs[:-1]], alt[offs[:-1]]], dim=1).contiguous()
ee, en, eu = B.geodetic_to_enu_batch(lat, lon, alt, offs, ref)
print(tag, "ENU ms", round(timed(lambda: B.geodetic_to_enu_batch(lat, lon, alt, offs, ref)), 4), "checksum", float(ee.sum() + en.sum() + eu.sum()))

import ctypes as C
from gps_optimize_slam_amd import _lib
llh = torch.stack([lat, lon, alt], dim=1).contiguous()
utm = torch.empty_like(llh); zz = torch.empty(nb, dtype=torch.int32, device=dev); ss = torch.empty(nb, dtype=torch.int32, device=dev)
L, h = _lib.load(), B.context().handle
run = lambda: _lib.check(L.gsf_gps_rows_to_utm_batch_dev(h, llh.data_ptr(), offs.data_ptr(), nb, utm.data_ptr(), zz.data_ptr(), ss.data_ptr()))
print(tag, "geodesy slice ms", round(timed(run), 4), "checksum", float(utm.sum()))
