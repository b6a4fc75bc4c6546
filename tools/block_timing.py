"""Per-phase timeline of the workgroup-per-trajectory kernel from the `make block_timing` build (GSF_LIBRARY=.../libgsf_block_timing.so):
every wave stamps (100 MHz clock, shader clock) at: 0 entry, 1 chunk arrived, 2 before barrier 1, 3 after it, 4 carries done,
5 before barrier 2 (wave 0: fit done), 6 after it, 7 before barrier 3, 8 after it, 9 before the stores.
usage: GSF_LIBRARY=... python tools/block_timing.py B N [pipe|ekf]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gps_optimize_slam_amd import batch as B  # noqa: E402

Bn, N = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "pipe"
ctx = B.context()
ctx.set_option("block_kernel", 1)
bt = B.TrajectoryBatch.synthetic(Bn, N, layout=0, seed=20250523)
o = B.FusedPoses(0, Bn, N, "cuda")
fn = (lambda: B.fuse_pipeline_batch(bt, out=o)) if which == "pipe" else (lambda: B.ekf_fuse_batch(bt, out=o))
for _ in range(5):
    fn()
torch.cuda.synchronize()
W = (N + 63) // 64
raw = o.pos.cpu().numpy().reshape(Bn, N * 3).view(np.int64)[:, : W * 24].reshape(Bn, W, 2, 12)
wall, clk = raw[:, :, 0, :].astype(np.float64), raw[:, :, 1, :].astype(np.float64)
t0 = wall[:, :, 0].min()
w_us = (wall - t0) / 100.0            # 100 MHz -> us
names = ["entry", "arrived", "preB1", "postB1", "carries", "preB2", "postB2", "preB3", "postB3", "prestore"]
print(f"B={Bn} N={N} W={W} {which}: stamps in us since the first wave's entry (min / median / max over all waves)")
for k, nm in enumerate(names):
    v = w_us[:, :, k]
    v = v[wall[:, :, k] != 0]
    if v.size:
        print(f"  {k} {nm:9s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")
print("per-wave phase durations, shader cycles (median over trajectories), by wave index:")
for k in range(1, 10):
    dur = clk[:, :, k] - clk[:, :, k - 1]
    ok = (clk[:, :, k] != 0) & (clk[:, :, k - 1] != 0)
    row = [np.median(dur[:, w][ok[:, w]]) if ok[:, w].any() else float("nan") for w in range(W)]
    print(f"  {names[k - 1]:>8s}->{names[k]:9s} " + " ".join(f"{x:7.0f}" for x in row))
blk_end = w_us[:, :, 9].max(axis=1)
blk_start = w_us[:, :, 0].min(axis=1)
print(f"block latency (entry of first wave -> last prestore): median {np.median(blk_end - blk_start):.2f} us, max {np.max(blk_end - blk_start):.2f}; last block ends at {blk_end.max():.2f} us; block starts: median {np.median(blk_start):.2f} max {blk_start.max():.2f}")
hw = raw[:, :, 0, 10].astype(np.int64); xcc = raw[:, :, 1, 10].astype(np.int64)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = (xcc << 12) | (se << 8) | (sh_ << 4) | cu
late = blk_start > 2.0
print(f"blocks starting later than 2 us: {int(late.sum())} of {Bn}")
import collections
early_cu = collections.Counter(cuid[~late, 0].tolist())
print("blocks per CU among the early ones (count: CUs):", sorted(collections.Counter(early_cu.values()).items()), "distinct CUs", len(early_cu))
per_simd = collections.Counter(((cuid[~late] << 2) | simd[~late]).reshape(-1).tolist())
print("waves per SIMD among the early blocks (count: SIMDs):", sorted(collections.Counter(per_simd.values()).items()))
print("per-XCC early blocks:", sorted(collections.Counter(xcc[~late, 0].tolist()).items()))
