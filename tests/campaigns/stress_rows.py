"""Randomised campaign for the reference's Sim3 row rule (EKFGPSSLAM.py:973-998) on the device (GPU box): random gap structures, stamp jumps at /
next to the threshold, NaN fixes, random CONFIG values (min_samples, max_gps_gap_threshold, max_initial_duration) and track lengths --
gsf_sim3_fit_rows_batch_dev's mask / count / branch against the oracle's rule row for row, and the fused pipeline's row bits and poses against the
oracle's pipeline on a sample of every batch.  usage: stress_rows.py [ROUNDS] [SEED]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B, ekfgpsslam as E
from oracle import oracle as orc
from test_sim3_rows import _random_gap_tracks

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
tracks = cut = few = br1 = br2 = 0
worst = 0.0
t0 = time.time()
for r in range(rounds):
    N = int(rng.choice([4, 7, 33, 64, 65, 100, 128, 129, 271, 300, 511, 640, 1000, 1500]))
    ms = int(rng.choice([3, 4, 4, 6, 10, 25]))
    gap = float(rng.choice([0.5, 1.0, 5.0, 5.0, 20.0]))
    dur = float(rng.choice([0.5 * gap, 3.0, 12.0, 180.0, 180.0, 1e9]))
    nb = 4096 if N <= 300 else 2304
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in E.CONFIG.items()}
    cfg["sim3_ransac"]["min_samples"] = ms; cfg["time_alignment"]["max_gps_gap_threshold"] = gap; cfg["sim3_ransac"]["max_initial_duration"] = dur
    ts, pos, quat, gps, valid = _random_gap_tracks(nb, N, int(rng.integers(1 << 30)), gap, dur)
    usable = (valid != 0) & np.isfinite(gps).all(axis=2)
    mask, nr, st = B.sim3_fit_rows_batch(dev(ts), dev(gps), dev(valid), cfg)
    mask, nr, st = mask.cpu().numpy(), nr.cpu().numpy(), st.cpu().numpy()
    for b in range(nb):
        idx, br = orc.pick_sim3_rows(ts[b], usable[b], ms, gap, dur, return_branch=True)
        if idx is None:
            assert nr[b] == -1 and st[b] == 32 and not mask[b].any(), (r, b)
            few += 1
            continue
        want = np.zeros(N, np.uint8); want[idx] = 1
        assert (mask[b] == want).all() and nr[b] == len(idx) and st[b] == {0: 0, 1: 128, 2: 64}[br], (r, N, ms, gap, dur, b)
        cut += len(idx) < usable[b].sum(); br1 += br == 1; br2 += br == 2
    ip = np.zeros((nb, 3)); iq = np.tile([0.0, 0.0, 0.0, 1.0], (nb, 1))
    out, R, t, s = B.fuse_pipeline_batch(B.TrajectoryBatch.from_host(ts, pos, quat, gps, valid, ip, iq, layout=0), config=cfg)
    p, q, stp = out.host_traj_major()
    sel = np.unique(np.concatenate([np.arange(0, nb, 16), np.where(nr < 0)[0][:16], np.where(st != 0)[0][:64]]))
    po, qo, sto, Ro, to, so, nro = orc.fuse_pipeline_batch(ts[sel], pos[sel], quat[sel], gps[sel], valid[sel], cfg=cfg, fit_rows="reference", return_rows=True)
    assert (nro == nr[sel]).all() and ((stp[sel] & ~(16 << 8)) == (sto & ~(16 << 8))).all(), (r, N)
    ok = np.isfinite(po).all(axis=(1, 2))
    assert (np.isfinite(p[sel]).all(axis=(1, 2)) == ok).all()
    d = np.abs(p[sel][ok] - po[ok]).max(axis=(1, 2)) if ok.any() else np.zeros(1)
    assert np.median(d) < 1e-7 and d.max() < 1e-5, (r, N, d.max())        # (nearly collinear row sets have an ill-conditioned fit)
    worst = max(worst, float(np.median(d)))
    tracks += nb
    print(f"round {r}: {nb} x {N} poses, min_samples {ms}, max gap {gap} s, max duration {dur:g} s: ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{tracks} tracks: {few} with too few rows, {cut} cut by a gap or the duration limit, {br1} whole-first-segment and {br2} all-rows fall-backs; "
      f"every mask, count and branch equal to the oracle's; worst median |dp| of a batch sample {worst:.2e} m")
