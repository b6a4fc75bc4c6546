"""How many trials should the GPS pre-filter chain draw and score before its first look at scikit-learn's stopping rule?  (gsf_set_option
"prefilter_first_batch"; the batches then double.)  Times gsf_gps_prefilter_auto_dev for 1 000 logs with first batches of 1, 2, 4, 8 -- and with the speculative pass ("prefilter_speculate": three axes' first trials at once) -- on
(a) the clean synthetic logs of the bench and (b)-(d) the same logs with 2 % / 5 % / 15 % of the fixes pushed 40 m sideways, and checks that the
kept-row masks and the generator states are the same words whatever the batch size (the walk over the trials is scikit-learn's, in order).
usage (GPU box): python tools/experiments/prefilter_first_batch.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B, _lib

nb, N = 1000, 271
gb = B.GeodeticBatch.synthetic(nb, N, seed=20250523)
L, ctx = _lib.load(), B.context()
dev = "cuda"
zone = torch.empty(nb, dtype=torch.int32, device=dev); south = torch.empty_like(zone)
pc = _lib.PrefilterConfig.from_config(B.CONFIG["gps_filtering_ransac"])
total = gb.gps_t.numel()
st0 = B.mt19937_seed(np.arange(nb) + 1)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {}
for share in (0.0, 0.02, 0.05, 0.15):
    llh = gb.gps_llh.clone()
    if share > 0:
        g = torch.Generator(device="cpu"); g.manual_seed(7)
        hit = (torch.rand(total, generator=g) < share).to(dev)
        llh[:, 0] += hit.double() * (40.0 / 111320.0)                       # 40 m north
    utm = torch.empty_like(llh)
    _lib.check(L.gsf_gps_rows_to_utm_batch_dev(ctx.handle, B._p(llh), B._p(gb.gps_offsets), nb, B._p(utm), B._p(zone), B._p(south)))
    ref = None; row = {}
    for fb, spec, mb in ((1, 0, 4), (2, 0, 4), (4, 0, 4), (8, 0, 4), (1, 1, 1), (1, 1, 2), (1, 1, 3), (1, 1, 4), (1, 1, 8)):
        ctx.set_option("prefilter_first_batch", fb); ctx.set_option("prefilter_speculate", spec); ctx.set_option("prefilter_miss_batch", mb)
        keep = torch.empty(total, dtype=torch.uint8, device=dev); ls = torch.empty(nb, dtype=torch.int32, device=dev)
        info = torch.zeros((nb, 2), dtype=torch.int32, device=dev)
        st = [None]

        def run():
            st[0] = st0.clone()
            _lib.check(L.gsf_gps_prefilter_auto_dev(ctx.handle, B._p(gb.gps_t), B._p(utm), B._p(gb.gps_offsets), nb, int(gb.max_fixes), C.byref(pc), B._p(st[0]),
                                                    B._p(keep), B._p(ls), B._p(info)))
        ms = timed(run, 10)
        cur = (keep.clone(), st[0].clone(), ls.clone())
        if ref is None: ref = cur
        same = all(bool((a == b).all().item()) for a, b in zip(ref, cur))
        row[f"first_batch_{fb}" + (f"_speculative_miss_batch_{mb}" if spec else "")] = {"ms": round(ms, 4), "same_words_as_first_batch_1": same}
    row["kept_share"] = float(ref[0].double().mean().item())
    row["windows_processed_mean"] = float(info[:, 0].double().mean().item())
    out[f"outlier_share_{share}"] = row
ctx.set_option("prefilter_first_batch", 1); ctx.set_option("prefilter_speculate", 1); ctx.set_option("prefilter_miss_batch", 4)
print(json.dumps(out, indent=1))
