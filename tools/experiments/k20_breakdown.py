"""Where do the microseconds of a K = 20 timed region go (the driver runs `bench.py --gpus 1 --steps 20 --warmup 5`)?  Host-side pieces of the
region around one graph replay of 20 launches of the C2 pipeline kernel, and the kernel span as a function of how long the GPU has been kept
busy right before.  usage: python tools/experiments/k20_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B
dev = torch.device("cuda:0")
bt = B.TrajectoryBatch.synthetic(1000, 271, layout=0, seed=20250523)
out = B.FusedPoses(0, 1000, 271, dev)
K = 20
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    B.context().set_sim3_rows("reference", B.CONFIG)
    B.fuse_pipeline_batch(bt, out=out)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
    for _ in range(K):
        B.fuse_pipeline_batch(bt, out=out)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()

def region(idle_ms, spin_replays, events=3):
    time.sleep(idle_ms / 1e3)
    for _ in range(spin_replays):
        g.replay()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    ev[0].record()
    t1 = time.perf_counter()
    g.replay()
    t2 = time.perf_counter()
    ev[1].record()
    if events == 3: ev[2].record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    return (t4 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, ev[0].elapsed_time(ev[1]) * 1e3

print("idle_ms spin  | region us | rec0  replay  rec1(+2)  sync | kernel span us (per launch)")
for idle, spin in ((0, 0), (50, 0), (500, 0), (500, 5), (500, 50), (500, 500), (0, 500), (0, 2000)):
    rs = np.array([region(idle, spin) for _ in range(7)])
    m = np.median(rs, axis=0)
    print(f"{idle:6d} {spin:5d} | {m[0]:8.1f} | {m[1]:5.1f} {m[2]:6.1f} {m[3]:6.1f} {m[4]:7.1f} | {m[5]:7.1f} ({m[5] / K:.2f})  -> region/K {m[0] / K:.2f} us, overhead {m[0] - m[5]:.1f} us")
