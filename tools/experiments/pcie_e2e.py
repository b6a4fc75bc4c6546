"""Why is the PCIe-inclusive step of bench.py's extras at 7 GB/s when the link gives 50?  Pieces timed on their own."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gps_optimize_slam_amd import batch as B
bt = B.TrajectoryBatch.synthetic(1000, 271, layout=0, seed=1)
out = B.FusedPoses(0, 1000, 271, "cuda")
names = ("ts", "pos", "quat", "gps", "valid")
hin = {k: getattr(bt, k).cpu().pin_memory() for k in names}
hout = torch.empty_like(out.buf, device="cpu").pin_memory()
print({k: (tuple(v.shape), v.dtype, v.is_pinned(), v.is_contiguous()) for k, v in hin.items()}, hout.is_pinned(), out.buf.shape)
def T(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for k in names:
    print("H2D", k, round(T(lambda: getattr(bt, k).copy_(hin[k], non_blocking=True)), 3), "ms", hin[k].numel() * hin[k].element_size() / 1e6, "MB")
print("D2H out", round(T(lambda: hout.copy_(out.buf, non_blocking=True)), 3), "ms", hout.numel() * 8 / 1e6, "MB")
print("kernel", round(T(lambda: B.fuse_pipeline_batch(bt, out=out)), 3), "ms")
def e2e():
    for k in names: getattr(bt, k).copy_(hin[k], non_blocking=True)
    B.fuse_pipeline_batch(bt, out=out)
    hout.copy_(out.buf, non_blocking=True)
print("e2e", round(T(e2e), 3), "ms")
