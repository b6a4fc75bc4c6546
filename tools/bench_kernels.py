"""Times the non-EKF kernels on BASELINE-shaped workloads (GPU box): K1 UTM fwd/inv, K2 Umeyama windows (C4), K2b RANSAC,
K3 apply-Sim3, time alignment.  Prints one JSON object per kernel with algorithmic GB/s (SURVEY 8d byte counts)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B, _lib

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

out = {}
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(1)
# ---- K1: 1e8 points in 1e5 trajectories of 1000
nb, n = 100_000, 1000
lat = 49.03 + 0.02 * (torch.rand(nb * n, dtype=torch.float64, device=dev, generator=g) - 0.5)
lon = 8.39 + 0.02 * (torch.rand(nb * n, dtype=torch.float64, device=dev, generator=g) - 0.5)
offs = torch.arange(0, nb * n + 1, n, dtype=torch.int64, device=dev)
e, nn, zone, south = B.utm_forward_batch(lat, lon, offs)
ms = timed(lambda: B.utm_forward_batch(lat, lon, offs, zone, south))
out["K1_utm_forward_1e8pts"] = {"ms": ms, "Gpts_s": nb * n / ms / 1e6, "alg_GBps": nb * n * 32 / ms / 1e6}
ms = timed(lambda: B.utm_inverse_batch(e, nn, offs, zone, south))
out["K1_utm_inverse_1e8pts"] = {"ms": ms, "Gpts_s": nb * n / ms / 1e6, "alg_GBps": nb * n * 32 / ms / 1e6}
ms = timed(lambda: B.utm_forward_batch(lat, lon, offs))
out["K1_zone_pick_plus_forward"] = {"ms": ms}
alt = 110.0 + torch.rand(nb * n, dtype=torch.float64, device=dev, generator=g)
ref = torch.tensor([[49.03, 8.39, 110.0]], dtype=torch.float64, device=dev).repeat(nb, 1)
ms = timed(lambda: B.geodetic_to_enu_batch(lat, lon, alt, offs, ref))
out["K1b_geodetic_to_enu_1e8pts"] = {"ms": ms, "Gpts_s": nb * n / ms / 1e6, "alg_GBps": nb * n * 48 / ms / 1e6}
del lat, lon, e, nn, alt
# ---- K2: C4 = 1M windows x 50 pairs
nw, W = 1_000_000, 50
src = torch.cumsum(torch.randn(nw, W, 3, dtype=torch.float64, device=dev, generator=g) * torch.tensor([0.05, 0.03, 1.4], dtype=torch.float64, device=dev), dim=1)
dst = 1.05 * src + torch.tensor([4.5e5, 5.4e6, 100.0], dtype=torch.float64, device=dev) + 0.45 * torch.randn(nw, W, 3, dtype=torch.float64, device=dev, generator=g)
ms = timed(lambda: B.sim3_umeyama_batch(src, dst))
out["K2_umeyama_C4_1Mx50"] = {"ms": ms, "Mwindows_s": nw / ms / 1e3, "alg_GBps": nw * (48 * W + 104) / ms / 1e6}
# ---- K3: apply sim3 to 1e8 poses (1e5 x 1000)
del src, dst
pos = torch.randn(nb * n, 3, dtype=torch.float64, device=dev, generator=g); quat = torch.randn(nb * n, 4, dtype=torch.float64, device=dev, generator=g)
R = torch.eye(3, dtype=torch.float64, device=dev).reshape(1, 9).repeat(nb, 1).contiguous(); t = torch.zeros(nb, 3, dtype=torch.float64, device=dev); s = torch.ones(nb, dtype=torch.float64, device=dev)
ms = timed(lambda: B.apply_sim3_batch(pos, quat, offs, R, t, s))
out["K3_apply_sim3_1e8poses"] = {"ms": ms, "Gposes_s": nb * n / ms / 1e6, "alg_GBps": nb * n * 112 / ms / 1e6}
del pos, quat
# ---- K2b: RANSAC, 1000 trajectories x 271 points x 1000 trials (C2-shaped)
nt, npts, trials = 1000, 271, 1000
bt = B.TrajectoryBatch.synthetic(nt, npts, layout=B.LAYOUT_TRAJ_MAJOR, seed=3)
srcp = bt.pos.reshape(nt * npts, 3).contiguous(); dstp = torch.nan_to_num(bt.gps.reshape(nt * npts, 3), nan=0.0).contiguous()   # missing fixes zero-filled (as in rounds 1-2)
offr = torch.arange(0, nt * npts + 1, npts, dtype=torch.int64, device=dev)
_r = np.random.default_rng(0)                                            # (host-made index sets: torch.randperm crashes under rocprofv3 --pmc)
idx = torch.as_tensor(np.stack([np.stack([_r.permutation(npts)[:4] for _ in range(trials)]) for _ in range(8)]).astype(np.int32)).to(dev)
idx = idx.repeat(nt // 8, 1, 1).contiguous()
ms = timed(lambda: B.sim3_ransac_batch(srcp, dstp, offr, idx, 4.0, 4), reps=3)
out["K2b_ransac_1000traj_271pts_1000trials"] = {"ms": ms, "traj_per_s": nt / ms * 1e3, "hypothesis_scores_per_s": nt * trials * npts / ms * 1e3}
# ---- time alignment: 1000 trajectories, 271 SLAM stamps, 279 fixes
st = (torch.arange(npts, dtype=torch.float64, device=dev) * 0.104).repeat(nt)
gtn = 279
gt = (torch.arange(gtn, dtype=torch.float64, device=dev) * 0.1047 + 0.001).repeat(nt)
gp = torch.randn(nt * gtn, 3, dtype=torch.float64, device=dev, generator=g).cumsum(0)
so = torch.arange(0, nt * npts + 1, npts, dtype=torch.int64, device=dev); go = torch.arange(0, nt * gtn + 1, gtn, dtype=torch.int64, device=dev)
al = torch.empty(nt * npts, 3, dtype=torch.float64, device=dev); va = torch.empty(nt * npts, dtype=torch.uint8, device=dev)
p = lambda x: C.c_void_p(x.data_ptr())
L, h = _lib.load(), B.context().handle
ms = timed(lambda: _lib.check(L.gsf_time_align_batch_dev(h, p(st), p(so), p(gt), p(gp), p(go), nt, 512, 5.0, p(al), p(va), None)))
out["align_1000traj_271x279"] = {"ms": ms, "Mposes_s": nt * npts / ms / 1e3}
# ---- next-3: polynomial RANSAC of the GPS pre-filter: 30 000 problems (10 000 windows x 3 axes) of 150 rows, 50 fed trials of 6 samples
P, n, trials, ms = 30_000, 150, 50, 6
t = (torch.arange(n, dtype=torch.float64, device=dev) * 0.1).repeat(P) + 0.01 * torch.rand(P * n, dtype=torch.float64, device=dev, generator=g)
y = 5.4e6 + 3.0 * t + 0.2 * t * t + 0.5 * torch.randn(P * n, dtype=torch.float64, device=dev, generator=g)
spike = torch.rand(P * n, dtype=torch.float64, device=dev, generator=g) < 0.1
y = y + spike * 100.0
offs = torch.arange(0, (P + 1) * n, n, dtype=torch.int64, device=dev)
idx = torch.as_tensor(np.argsort(_r.random((300, trials, n)), axis=2)[:, :, :ms].astype(np.int32)).to(dev).repeat(P // 300, 1, 1).contiguous()
ms_t = timed(lambda: B.ransac_poly_batch(t, y, offs, idx, 2, 10.0))
out["next3_ransac_poly_30k_problems_150rows_50trials"] = {"ms": ms_t, "problems_per_s": P / ms_t * 1e3, "residual_evals_per_s": P * trials * n * 2 / ms_t * 1e3}
del t, y, offs, idx, spike
# ---- next-4: error evaluation (nearest-fix distances, mean / median / RMSE) on 1 000 fused 271-pose tracks
bt = B.TrajectoryBatch.synthetic(1000, 271, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
fo = B.ekf_fuse_batch(bt)
try:
    ms_e = timed(lambda: B.eval_errors_batch(bt.ts, fo.pos, bt.gps, bt.valid, 0.0))
    out["next4_eval_errors_1000traj_271"] = {"ms": ms_e, "Mpairs_s": 1000 * 271 * 271 / ms_e / 1e3}
except Exception as ex:                                                  # signature differences must not hide the other timings
    out["next4_eval_errors_1000traj_271"] = {"error": str(ex)}
# ---- the same metric at the C3 track length (10 000 x 1 000): more than 400 evaluated poses -> the pruned nearest-fix search + sorted median
del bt, fo
try:
    bl = B.TrajectoryBatch.synthetic(10000, 1000, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
    fl = B.ekf_fuse_batch(bl)
    ms_l = timed(lambda: B.eval_errors_batch(bl.ts, fl.pos, bl.gps, bl.valid, 5.0), reps=5)
    out["next4_eval_errors_10000traj_1000"] = {"ms": ms_l, "poses_per_s": 1e7 / ms_l * 1e3}
    del bl, fl
except Exception as ex:
    out["next4_eval_errors_10000traj_1000"] = {"error": str(ex)}
# ---- the reference's draws on the device: 1 000 streams x 1 000 trials of permutation(271)[:4]
st = B.mt19937_seed(np.arange(1000))
ms_c = timed(lambda: B.mt19937_choice_batch(st, [271] * 1000, 1000, 4), reps=3)
out["mt19937_choice_1000streams_1000trials_n271"] = {"ms": ms_c, "trials_per_s": 1e6 / ms_c * 1e3, "raw_outputs_per_s_approx": 1e6 * 385 / ms_c * 1e3}
# ---- robust chain and the chain from the geodetic log, C2 shape
bt = B.TrajectoryBatch.synthetic(1000, 271, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
o = B.FusedPoses(bt.layout, 1000, 271, dev)
ms_r = timed(lambda: B.fuse_pipeline_robust_batch(bt, st, out=o, want_mask=False), reps=3)
out["robust_chain_1000x271"] = {"ms": ms_r, "Mposes_s": 271e3 / ms_r / 1e3}
gb = B.GeodeticBatch.synthetic(1000, 271, seed=1)
ms_g = timed(lambda: B.fuse_from_geodetic(gb, out=o), reps=20)
out["geodetic_chain_1000x271"] = {"ms": ms_g, "Mposes_s": 271e3 / ms_g / 1e3}
gb = B.GeodeticBatch.synthetic(100_000, 1000, seed=1)
utm = torch.empty_like(gb.gps_llh); zone = torch.empty(100_000, dtype=torch.int32, device=dev); south = torch.empty_like(zone)
ms_k = timed(lambda: _lib.check(L.gsf_gps_rows_to_utm_batch_dev(h, p(gb.gps_llh), p(gb.gps_offsets), gb.B, p(utm), p(zone), p(south))))
out["geodesy_slice_rows_1e8"] = {"ms": ms_k, "Gpts_s": gb.gps_t.numel() / ms_k / 1e6, "alg_GBps": gb.gps_t.numel() * 48 / ms_k / 1e6}
print(json.dumps(out, indent=1))
