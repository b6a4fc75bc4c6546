"""A/B helper: kernel times of the trajectory-major K4 / fused pipeline for the library named by GSF_LIBRARY (same box, same
process layout).  usage: [GSF_LIBRARY=...] python tools/ab_bench.py [tag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gps_optimize_slam_amd import batch as B  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("GSF_LIBRARY", "libgsf.so"))
ctx = B.context()
res = []
bt = B.TrajectoryBatch.synthetic(1000, 271, layout=0, seed=20250523)
o = B.FusedPoses(0, 1000, 271, "cuda")
res.append(("c2_ekf", timed(lambda: B.ekf_fuse_batch(bt, out=o), 500)))
for duo in (0, 1, -1):
    try:
        ctx.set_option("duo_kernel", duo)
    except Exception:
        continue
    res.append((f"c2_pipe_duo{duo}", timed(lambda: B.fuse_pipeline_batch(bt, out=o), 500)))
ctx.set_option("duo_kernel", -1)
del bt, o
bt = B.TrajectoryBatch.synthetic(100_000, 1000, layout=0, seed=1)
o = B.FusedPoses(0, 100_000, 1000, "cuda")
res.append(("c3_ekf", timed(lambda: B.ekf_fuse_batch(bt, out=o), 10)))
res.append(("c3_pipe", timed(lambda: B.fuse_pipeline_batch(bt, out=o), 10)))
print(f"{tag:28s} " + "  ".join(f"{k}={v:8.2f}us" for k, v in res))
