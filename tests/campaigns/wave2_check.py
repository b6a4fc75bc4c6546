"""Experiment driver (GPU box): the two-poses-per-lane chunk loop (csrc/exp/gsf_wave2.hpp, `make wave2`, GSF_LIBRARY=.../libgsf_wave2.so)
against the oracle on random outage / sharp-turn / NaN-fix tracks of many lengths, K4 and fused pipeline, then its kernel time next to the
shipped kernel's.  usage: GSF_LIBRARY=... python tests/campaigns/wave2_check.py [check|time|both] [NB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
ctx = B.context()
ctx.set_option("duo_kernel", 0)
print("library", os.environ.get("GSF_LIBRARY", "libgsf.so"), flush=True)

if mode in ("check", "both"):
    from oracle import oracle as orc
    from test_gpu_parity import _random_outage_batch
    worst = 0.0
    fails = 0
    for N in (2, 3, 4, 5, 63, 64, 65, 126, 127, 128, 129, 130, 131, 255, 256, 257, 258, 271, 300, 383, 384, 385, 777, 1000):
        ts, pos, quat, gps, valid, ip, iq = _random_outage_batch(nb, N, 100 + N)
        po, qo, sto = orc.fuse_batch(ts, pos, quat, gps, valid, ip, iq)
        batch = B.TrajectoryBatch.from_host(ts, pos, quat, gps, valid, ip, iq, layout=0)
        p, q, st = B.ekf_fuse_batch(batch).host_traj_major()
        bad = np.nonzero(st != sto)[0]
        dp, dq = np.nanmax(np.abs(p - po)), np.nanmax(np.abs(q - qo))
        nanbad = int((np.isnan(p) != np.isnan(po)).sum())
        line = f"K4 N={N:5d}: status mismatches {len(bad)} {bad[:4].tolist()}  max|dp| {dp:.3e}  max|dq| {dq:.3e}  nan-mismatch {nanbad}"
        for rule in ("all", "reference"):
            pr, qr, str_, Rr, tr, sr = orc.fuse_pipeline_batch(ts, pos, quat, gps, valid, fit_rows=rule)
            ok = np.isfinite(pr).all(axis=(1, 2))
            out, R, t, s = B.fuse_pipeline_batch(batch, fit_rows=rule)
            p2, q2, st2 = out.host_traj_major()
            okg = np.isfinite(p2).all(axis=(1, 2))
            badp = int(((st2[ok] & 0xff) != (str_[ok] & 0xff)).sum()) + int((okg != ok).sum())
            dpp = np.abs(p2[ok & okg] - pr[ok & okg]).max() if (ok & okg).any() else 0.0
            line += f" | pipe[{rule[:3]}] bad {badp} dp {dpp:.2e}"
            if badp or not dpp < 1e-6: fails += 1
        print(line, flush=True)
        if len(bad) or nanbad or not dp < 1e-6 or not dq < 1e-8:
            fails += 1
            for bb in bad[:3]:
                print(f"    track {bb}: gpu status {st[bb]:#x} oracle {sto[bb]:#x}; first differing pose {int(np.argmax(np.abs(p[bb] - po[bb]).max(axis=1) > 1e-6))}")
            if not dp < 1e-6:
                bb = int(np.argmax(np.nanmax(np.abs(p - po), axis=(1, 2))))
                k = int(np.argmax(np.abs(p[bb] - po[bb]).max(axis=1) > 1e-6))
                print(f"    worst track {bb} (status {sto[bb]:#x}): first pose off by > 1e-6 at {k}; valid around: {valid[bb, max(0, k - 4):k + 5].tolist()}")
        worst = max(worst, dp)
    print("CHECK", "FAILED" if fails else "OK", "worst", worst, flush=True)

if mode in ("time", "both"):
    def timed(fn, reps):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps * 1e3)
        return best
    res = []
    for (T, N, variant) in ((1000, 271, 0), (1000, 256, 0), (1000, 1000, 0), (256, 271, 0), (2048, 271, 0)):
        bt = B.TrajectoryBatch.synthetic(T, N, layout=0, seed=20250523)
        o = B.FusedPoses(0, T, N, "cuda")
        res.append((f"{T}x{N} ekf", timed(lambda: B.ekf_fuse_batch(bt, out=o), 300)))
        res.append((f"{T}x{N} pipe[ref]", timed(lambda: B.fuse_pipeline_batch(bt, out=o, fit_rows="reference"), 300)))
        res.append((f"{T}x{N} pipe[all]", timed(lambda: B.fuse_pipeline_batch(bt, out=o, fit_rows="all"), 300)))
    print("TIME " + "  ".join(f"{k}={v:.2f}us" for k, v in res), flush=True)
