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
info = torch.zeros((nb, 16), dtype=torch.int32, device="cuda")
st = B.mt19937_seed(np.arange(nb) + 1)
for _ in range(3):
    _lib.check(L.gsf_gps_prefilter_auto_dev(ctx.handle, B._p(gb.gps_t), B._p(utm), B._p(gb.gps_offsets), nb, int(gb.max_fixes), C.byref(pc), B._p(st.clone()), B._p(keep), B._p(ls), B._p(info)))
torch.cuda.synchronize()
a = info.cpu().numpy().astype(np.float64)
names = ["snapshot", "draw", "fit", "score", "walk", "rewind", "final model + mask", "-", "-", "problems", "whole kernel", "-"]
prob = a[:, 9].mean()
print(f"problems per log: {prob:.1f}; whole kernel {a[:, 10].mean():.0f} cycles mean, {a[:, 10].max():.0f} max")
for k in (0, 1, 2, 3, 4, 5, 6):
    print(f"  {names[k]:20s} {a[:, k].mean():10.0f} cycles per log = {a[:, k].mean() / prob:8.0f} per problem ({a[:, k].mean() / a[:, 10].mean():.1%})")
