"""Per-phase shader-clock totals of the GPS pre-filter chain for 1 000 logs (diagnostic build: make -C gps_optimize_slam_amd/csrc pf_timing;
GSF_LIBRARY=gps_optimize_slam_amd/libgsf_pf_timing.so python tools/experiments/prefilter_timing.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B, _lib
nb, N = 1000, 271
gb = B.GeodeticBatch.synthetic(nb, N, seed=20250523)
L, ctx = _lib.load(), B.context()
utm = torch.empty_like(gb.gps_llh); zone = torch.empty(nb, dtype=torch.int32, device="cuda"); south = torch.empty_like(zone)
_lib.check(L.gsf_gps_rows_to_utm_batch_dev(ctx.handle, B._p(gb.gps_llh), B._p(gb.gps_offsets), nb, B._p(utm), B._p(zone), B._p(south)))
pc = _lib.PrefilterConfig.from_config(B.CONFIG["gps_filtering_ransac"])
keep = torch.empty(gb.gps_t.numel(), dtype=torch.uint8, device="cuda"); ls = torch.empty(nb, dtype=torch.int32, device="cuda")
info = torch.zeros((nb, 20), dtype=torch.int32, device="cuda")
st = B.mt19937_seed(np.arange(nb) + 1)
for _ in range(3):
    _lib.check(L.gsf_gps_prefilter_auto_dev(ctx.handle, B._p(gb.gps_t), B._p(utm), B._p(gb.gps_offsets), nb, int(gb.max_fixes), C.byref(pc), B._p(st.clone()), B._p(keep), B._p(ls), B._p(info)))
torch.cuda.synchronize()
a = info.cpu().numpy().astype(np.float64)
names = ["snapshot", "draw", "fit", "score", "walk", "rewind", "final model + mask", "speculative: masks, rewind", "-", "problems", "whole kernel", "-", "speculative: snapshot, draw", "speculative: fits", "speculative: counts", "speculative: stopping rule", "prologue (state, keep = 0)", "window set-up (loads, keep |= 2)", "window fold", "-"]
prob = a[:, 9].mean()
print(f"problems per log: {prob:.1f}; whole kernel {a[:, 10].mean():.0f} cycles mean, {a[:, 10].max():.0f} max")
print(f"speculative passes per log {a[:, 11].mean():.1f}, axes they finished {a[:, 8].mean():.1f}; sequential problems per log {prob:.1f}")
prob = max(prob, 1e-9)
for k in (16, 17, 18, 0, 1, 2, 3, 4, 5, 6, 12, 13, 14, 15, 7):
    print(f"  {names[k]:28s} {a[:, k].mean():10.0f} cycles per log = {a[:, k].mean() / prob:8.0f} per problem ({a[:, k].mean() / a[:, 10].mean():.1%})")
order = np.argsort(a[:, 10])
print("whole-kernel cycles per log: " + ", ".join(f"p{q} {np.percentile(a[:, 10], q):.0f}" for q in (50, 90, 99, 100)))
for b in order[-5:]:
    print(f"  log {b}: {a[b, 10]:.0f} cycles, speculative passes {a[b, 11]:.0f} finishing {a[b, 8]:.0f} axes ({a[b, 7]:.0f} cycles), sequential problems {a[b, 9]:.0f}: "
          + ", ".join(f"{names[k]} {a[b, k]:.0f}" for k in (1, 2, 3, 4, 5, 6)))
