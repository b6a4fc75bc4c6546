"""One-wave vs two-wave build of the fused pipeline over batch sizes and track lengths (the data behind the automatic choice in
gsf_ekf_wave.hip: launch_ekf_wave).  usage (GPU box): python tools/duo_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gps_optimize_slam_amd import batch as B  # noqa: E402


def timed(fn, reps=300):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


ctx = B.context()
for N in (271, 100, 600):
    for nb in (64, 256, 512, 768, 1000, 1024, 1500, 2048):
        bt = B.TrajectoryBatch.synthetic(nb, N, layout=0, seed=20250523)
        o = B.FusedPoses(0, nb, N, "cuda")
        r = []
        for duo in (0, 1):
            ctx.set_option("duo_kernel", duo)
            r.append(timed(lambda: B.fuse_pipeline_batch(bt, out=o)))
        ctx.set_option("duo_kernel", -1)
        print(f"N={N:4d} B={nb:5d}  one wave {r[0]:7.2f}  two waves {r[1]:7.2f} us   automatic {timed(lambda: B.fuse_pipeline_batch(bt, out=o)):7.2f}")
