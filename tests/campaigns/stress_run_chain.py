"""Randomised campaign for the whole-run chain (gsf_run_fusion_batch_dev: steps 1-6 of main_process_gui, EKFGPSSLAM.py:959-1033) on the GPU box:
random CONFIG values (pre-filter window length / step / degree / min_samples / threshold / trial cap, sliding or global; Sim3 min_samples /
threshold / trial cap / duration limit; gap threshold), random track lengths, logs with planted outliers, rows the loader drops, thinned logs --
every trajectory against the oracle's composition of the flow under ONE seeded generator (tests/test_run_chain.py:_oracle_run): loaded / kept
fixes, zone, alignment mask, n_inliers, status words, run_status and the final generator state exactly; R, t, s, fused poses and step-6 numbers to
the gates; then the same batch with the exact early exit: every output word equal.  usage: stress_run_chain.py [ROUNDS] [SEED] [TRACKS]"""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B, ekfgpsslam as E
from oracle import oracle as orc
from test_run_chain import _oracle_run, np_state

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 48
rng = np.random.default_rng(seed0)
orc.build()
tot = dict(tracks=0, ok=0, gps_empty=0, gps_few=0, unhandled=0, sim3_failed=0, filtered=0, saturated=0, rank_deficient_fit=0)
worst_p = worst_q = worst_s6 = 0.0
t0 = time.time()
for r in range(rounds):
    N = int(rng.choice([100, 150, 271, 271, 400, 640]))
    cfg = copy.deepcopy(E.CONFIG)
    f = cfg["gps_filtering_ransac"]
    f["use_sliding_window"] = bool(rng.random() < 0.8)
    f["window_duration_seconds"] = float(rng.choice([4.0, 8.0, 15.0, 15.0, 40.0]))
    f["window_step_factor"] = float(rng.choice([0.25, 0.5, 0.5, 1.0]))
    f["polynomial_degree"] = int(rng.choice([1, 2, 2, 3]))
    f["min_samples"] = int(rng.choice([4, 6, 6, 8]))
    f["residual_threshold_meters"] = float(rng.choice([3.0, 10.0, 10.0, 25.0]))
    f["max_trials"] = int(rng.choice([20, 50, 50, 120]))
    sc = cfg["sim3_ransac"]
    sc["min_samples"] = int(rng.choice([3, 4, 4, 6])); sc["residual_threshold"] = float(rng.choice([1.5, 4.0, 4.0, 8.0]))
    sc["max_trials"] = int(rng.choice([60, 200, 400])); sc["min_inliers_needed"] = int(rng.choice([4, 4, 30]))
    sc["max_initial_duration"] = float(rng.choice([6.0, 20.0, 180.0, 180.0]))
    cfg["time_alignment"]["max_gps_gap_threshold"] = float(rng.choice([2.0, 5.0, 5.0, 9.0]))
    src = B.GeodeticBatch.synthetic(nb, N, seed=int(rng.integers(1 << 30)))
    offs = src.gps_offsets.cpu().numpy()
    gt, llh = src.gps_t.cpu().numpy(), src.gps_llh.cpu().numpy()
    ts, pos, quat = src.ts.cpu().numpy(), src.pos.cpu().numpy(), src.quat.cpu().numpy()
    logs = []
    for b in range(nb):
        log = np.column_stack((gt[offs[b]:offs[b + 1]], llh[offs[b]:offs[b + 1]]))
        n = len(log)
        u = rng.random()
        if u < 0.45 and n > 30:                                         # fixes thrown 15 .. 80 m off
            for r_ in rng.choice(n, size=int(rng.integers(1, 8)), replace=False):
                d = rng.uniform(15.0, 80.0)
                log[r_, 1] += d / 111200.0 * rng.choice([-1, 1]); log[r_, 2] += d / 73000.0 * rng.choice([-1, 1])
        if rng.random() < 0.2 and n > 30:                               # rows the loader removes (ref :259)
            rr = rng.choice(n, size=3, replace=False)
            log[rr[0], 1] = 0.0; log[rr[1], 2] = 0.0; log[rr[2], 1] = -95.0
        if rng.random() < 0.04: log = log[np.sort(rng.choice(n, size=int(rng.integers(1, 6)), replace=False))]   # a thinned log
        if rng.random() < 0.02: log[:, 2] = 0.0                         # nothing survives the range mask
        logs.append(log)
    gb = B.GeodeticBatch.from_host(ts, pos, quat, logs)
    seeds = rng.integers(1, 1 << 31, size=nb)
    st = B.mt19937_seed(seeds)
    res = B.run_fusion_batch(gb, st, cfg, early_exit=False)
    p, q, status = res.fused.host_traj_major()
    o2 = gb.gps_offsets.cpu().numpy()
    keep, utm = res.gps_keep.cpu().numpy().astype(bool), res.gps_utm.cpu().numpy()
    va, stats, rs = res.valid.cpu().numpy().astype(bool), res.err_stats.cpu().numpy(), res.run_status.cpu().numpy()
    for b in range(nb):
        tot["tracks"] += 1
        if rs[b] & 4:                                                   # the device pre-filter does not cover this log: the caller's host route (not compared)
            tot["unhandled"] += 1
            continue
        o = _oracle_run(orc, ts[b], pos[b], quat[b], logs[b], cfg, int(seeds[b]))
        key, ppos = np.random.get_state()[1:3]
        gk, gp = np_state(st[b])
        ctx_ = (r, b, N, {k: f[k] for k in ("use_sliding_window", "window_duration_seconds", "window_step_factor", "min_samples", "max_trials")})
        assert (gk == key).all() and gp == int(ppos), ("generator", ctx_)
        assert rs[b] == o["status"], ("run_status", ctx_, rs[b], o["status"])
        u = utm[o2[b]:o2[b + 1]]
        assert (~(np.isnan(u[:, 0]) & np.isnan(u[:, 1])) == o["loaded"]).all(), ("loaded", ctx_)
        if o["status"] == 1:
            tot["gps_empty"] += 1; continue
        assert (keep[o2[b]:o2[b + 1]] == o["keep"]).all(), ("kept fixes", ctx_)
        tot["filtered"] += int(o["keep"].sum() < o["loaded"].sum())
        if o["status"] == 2:
            tot["gps_few"] += 1; continue
        assert (va[b] == o["valid"]).all(), ("valid", ctx_)
        if o["status"] == 8:
            tot["sim3_failed"] += 1
            assert np.isnan(p[b]).all(); continue
        assert int(res.n_inliers[b]) == o["n_inliers"], ("n_inliers", ctx_)
        # A fit whose cross-covariance is rank-deficient to rounding (a thinned log: two fixes -> the alignment interpolates LINEARLY, every
        # aligned point of the segment lies on one line to 1e-12) leaves the rotation about that line to the last bits of the inputs: the
        # reference's own answer moves by radians under a 1e-9 m change of the aligned fixes (gpurun_out/stress_run_chain/fail_r0_b4.npz: the
        # oracle fed the DEVICE's aligned rows returns the device's R to 2e-6).  What such a fit does determine is compared: the scale, and the
        # image of the fitted rows up to their own extent off the line.
        fr = o["fit_rows"]
        Hc = (pos[b][fr] - pos[b][fr].mean(0)).T @ (o["aligned"][fr] - o["aligned"][fr].mean(0))
        sv = np.linalg.svd(Hc, compute_uv=False)
        if sv[1] < 1e-7 * sv[0]:
            Rd, td, sd = res.R[b].cpu().numpy().reshape(3, 3), res.t[b].cpu().numpy(), float(res.s[b])
            img_d, img_o = sd * pos[b][fr] @ Rd.T + td, o["s"] * pos[b][fr] @ o["R"].T + o["t"]
            lateral = np.linalg.svd(pos[b][fr] - pos[b][fr].mean(0), compute_uv=False)[1]       # what a turn about the line can move: the rows' extent off it
            assert abs(sd - o["s"]) < 1e-8 and np.abs(img_d - img_o).max() <= 2.5 * lateral + 1e-6, ("rank-deficient fit", ctx_, sd, o["s"], float(np.abs(img_d - img_o).max()), lateral)
            tot["rank_deficient_fit"] += 1
            continue
        dp, dq = float(np.abs(p[b] - o["pos"]).max()), float(np.abs(q[b] - o["quat"]).max())
        # The orientation gate follows the conditioning of the FINAL fit (ref :420-421, on the inliers): a cross-covariance with singular values
        # s1 >> s2 ~ s3 (a handful of inliers almost on one line) turns by |dH| / (s2 + s3) under a change dH of its entries, and the aligned fixes
        # of the two sides differ by ~2e-9 m (the spline).  gpurun_out/stress_run_chain/fail_r25_b24.npz: 15 inliers, s = (1013, 0.048, 0.045),
        # |dR| 3.2e-8, dq 1.5e-8, positions 4e-8 m -- NumPy's SVD-based Umeyama on the device's aligned rows returns the device's R to 6e-13 and on the
        # oracle's rows the oracle's R to 5e-13.  Gate: 1e-8 up to s1 / (s2 + s3) = 1e3, proportional beyond.
        mi = res.inlier_mask[b].cpu().numpy().astype(bool)
        svi = np.linalg.svd((pos[b][mi] - pos[b][mi].mean(0)).T @ (o["aligned"][mi] - o["aligned"][mi].mean(0)), compute_uv=False) if mi.sum() >= 3 else np.ones(3)
        amp = svi[0] / max(svi[1] + svi[2], 1e-300)
        # (positions follow with the lever arm of the track: fail_r13_b35.npz -- 58 inliers on a straight stretch of a 940 m track, s1 / (s2 + s3) =
        # 1.3e5, |dR| 3.6e-8 on both sides' own rows, poses 3.6e-6 m apart)
        q_gate, p_gate = 1e-8 * max(1.0, amp / 1e3), 1e-6 * max(1.0, amp / 1e4)
        if not (dp < p_gate and dq < q_gate and (status[b] & 0xff) == o["st"]):
            import json
            os.makedirs(os.path.join(ROOT, "gpurun_out", "stress_run_chain"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "stress_run_chain", f"fail_r{r}_b{b}.npz"), ts=ts[b], pos=pos[b], quat=quat[b], log=logs[b], seed=int(seeds[b]),
                     cfg=json.dumps(cfg), dev_R=res.R[b].cpu().numpy(), dev_t=res.t[b].cpu().numpy(), dev_s=float(res.s[b]), dev_mask=res.inlier_mask[b].cpu().numpy(),
                     dev_pos=p[b], dev_aligned=res.aligned[b].cpu().numpy(), dev_valid=va[b], orc_R=o["R"], orc_t=o["t"], orc_s=o["s"], orc_pos=o["pos"],
                     orc_aligned=o["aligned"], dev_info=res.trial_info[b].cpu().numpy())
        assert dp < p_gate and dq < q_gate and (status[b] & 0xff) == o["st"], ("poses", ctx_, dp, dq, q_gate, p_gate, "R", float(np.abs(res.R[b].cpu().numpy().reshape(3, 3) - o["R"]).max()),
                                                                             "s", float(res.s[b]), o["s"], "n_inliers", o["n_inliers"])
        worst_p, worst_q = max(worst_p, dp), max(worst_q, dq)
        for row in range(3):
            e = o["errs"][row]
            assert int(stats[row, b, 0]) == e["count"], ("step-6 count", ctx_)
            if e["count"]:
                d6 = float(np.abs(stats[row, b, 1:] - np.array([e["mean"], e["median"], e["rmse"]])).max()) if row else 0.0
                assert d6 < p_gate, ("step 6", ctx_, row, d6)
                worst_s6 = max(worst_s6, d6)
        tot["ok"] += 1
    st_e = B.mt19937_seed(seeds)
    re_ = B.run_fusion_batch(gb, st_e, cfg, early_exit=True)
    for a, b_ in ((res.fused.pos, re_.fused.pos), (res.fused.quat, re_.fused.quat), (res.R, re_.R), (res.t, re_.t), (res.s, re_.s), (res.err_stats, re_.err_stats)):
        assert torch.equal(torch.nan_to_num(a, nan=-1.0).view(torch.int64), torch.nan_to_num(b_, nan=-1.0).view(torch.int64)), ("early exit", r)
    sat = ((re_.fused.status >> 8) & 256) != 0
    ee_ok = (torch.equal(res.fused.status, re_.fused.status & ~(256 << 8)), torch.equal(res.n_inliers, re_.n_inliers), torch.equal(st[~sat], st_e[~sat]),
             torch.equal(res.trial_info[:, 0], re_.trial_info[:, 0]))
    if not all(ee_ok):
        bad = torch.nonzero((res.n_inliers != re_.n_inliers) | (res.trial_info[:, 0] != re_.trial_info[:, 0]) | (res.fused.status != (re_.fused.status & ~(256 << 8)))).ravel().tolist()
        print("early exit differs:", ee_ok, "tracks", bad, "sim3 cfg", sc, [(int(res.n_inliers[b]), int(re_.n_inliers[b]), res.trial_info[b].tolist(), re_.trial_info[b].tolist()) for b in bad[:6]], flush=True)
    assert all(ee_ok), ("early exit", r, ee_ok)
    tot["saturated"] += int(sat.sum())
    print(f"round {r}: N={N} sliding={f['use_sliding_window']} win={f['window_duration_seconds']}x{f['window_step_factor']} deg={f['polynomial_degree']} ms={f['min_samples']} "
          f"trials={f['max_trials']} | sim3 ms={sc['min_samples']} thr={sc['residual_threshold']} trials={sc['max_trials']} | {tot} worst dp {worst_p:.2e} dq {worst_q:.2e} step6 {worst_s6:.2e} "
          f"({time.time() - t0:.0f} s)", flush=True)
print("stress_run_chain: no mismatch;", tot)
