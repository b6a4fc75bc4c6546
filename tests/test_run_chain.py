"""Steps 1-6 of main_process_gui (EKFGPSSLAM.py:959-1033) for B trajectories as ONE device chain: gsf_run_fusion_batch_dev / batch.run_fusion_batch.

Geodesy slice -> GPS RANSAC pre-filter (windows walked on the device) -> time alignment -> row choice -> robust Sim3 -> apply -> EKF + RTS ->
error metric, every trajectory's legacy MT19937 stream used by the pre-filter first and the fit second, in the reference's order.

Against: the two bundled C1 pipelines (c1_*.npz: the reference's functions on the bundled KITTI-04 files) as stacked copies, and the single-track
drop-in run_fusion on the same files row for row; the 38 headless runs of the reference's OWN main_process_gui (sim3_rows_*.npz: R, t, s, fused
track, the step-6 numbers of its own distance matrices, where it leaves np.random) through the chain from the pre-filter on; the oracle's
composition of the whole flow on synthetic logs with planted outliers, fixes the loader drops and logs the filter thins out."""
import copy
import ctypes as C

import numpy as np
import pytest

from test_sim3_rows import case_cfg, cases

pytestmark = pytest.mark.gpu
POS_TOL = 1e-7
Q_TOL = 1e-9


@pytest.fixture(scope="module")
def B():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gps_optimize_slam_amd import batch
    return batch


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def np_state(st_row):
    a = st_row.cpu().numpy().view(np.uint32)
    return a[:624].copy(), int(a[624])


@pytest.mark.parametrize("tag", ["kitti04gps", "combined"])
def test_stacked_c1_goldens_and_the_drop_in(B, golden, tmp_path, tag):
    """B stacked copies of a bundled C1 run (KITTI-04 SLAM track + its GNSS file as the loader reads it), every copy's generator seeded like
    the golden's pre-filter: zone, UTM rows, the fixes the seeded pre-filter keeps, step 2's alignment, n_inliers, R / t / s, the fused track
    (<= 1e-7 m) and the step-6 mean / median / RMSE of the golden on every copy -- and everything equal to the single-track drop-in
    run_fusion on the same two files (masks, counts and the final generator state exactly, poses to 1e-9 m)."""
    import torch
    from gps_optimize_slam_amd import ekfgpsslam as E
    g, k = golden(f"c1_{tag}.npz"), golden("kat_bundled.npz")
    copies, N = 6, len(k["ts"])
    log = np.column_stack((g["gps_t_raw"], g["lat"], g["lon"], g["alt"]))
    rep = lambda a: np.repeat(np.asarray(a)[None], copies, axis=0)
    gb = B.GeodeticBatch.from_host(rep(k["ts"]), rep(k["pos"]), rep(k["quat"]), [log] * copies)
    st = B.mt19937_seed([0] * copies)
    r = B.run_fusion_batch(gb, st, early_exit=False)
    p, q, status = r.fused.host_traj_major()
    assert (r.run_status == 0).all()
    n = len(log)
    utm, keep = r.gps_utm.cpu().numpy().reshape(copies, n, 3), r.gps_keep.cpu().numpy().reshape(copies, n).astype(bool)
    al, va, stats = r.aligned.cpu().numpy(), r.valid.cpu().numpy().astype(bool), r.err_stats.cpu().numpy()
    for c in range(copies):
        assert int(r.zone[c]) == int(g["zone"]) and int(r.south[c]) == int(g["south"])
        np.testing.assert_allclose(utm[c], g["utm"], atol=5e-9, rtol=0)
        np.testing.assert_array_equal(g["gps_t_raw"][keep[c]], g["gps_t"])                      # the fixes load_gps_data returns (seeded pre-filter)
        np.testing.assert_array_equal(va[c], g["valid"])
        np.testing.assert_allclose(al[c][va[c]], g["aligned"][g["valid"]], atol=5e-7, rtol=0)    # (scipy's cubic interp1d loses ~1e-7 m at |y| ~ 5e6: DESIGN section 2)
        assert int(r.n_inliers[c]) == int(g["n_inliers"])
        np.testing.assert_allclose(r.R[c].cpu().numpy().reshape(3, 3), g["R"], atol=2e-9, rtol=0)
        assert abs(float(r.s[c]) - float(g["s"])) < 1e-11
        assert np.abs(p[c] - g["ekf_pos"]).max() < POS_TOL and np.abs(q[c] - g["ekf_quat"]).max() < Q_TOL
        assert np.abs(r.sim3_pos[c].cpu().numpy() - g["sim3_pos"]).max() < POS_TOL
        np.testing.assert_allclose(stats[1, c, 1:], g["err_sim3"], atol=2e-7, rtol=0)
        np.testing.assert_allclose(stats[2, c, 1:], g["err_ekf"], atol=2e-7, rtol=0)
    assert (p == p[0]).all() and (stats == stats[:, :1]).all()                                  # copies of one run: the same bits
    # ---- the single-track drop-in on the same files, np.random seeded the same way: one continuous stream through pre-filter and fit
    slam_path, gps_path = tmp_path / "slam.txt", tmp_path / "gps.txt"
    np.savetxt(slam_path, np.column_stack((k["ts"], k["pos"], k["quat"])), fmt="%.18e")
    np.savetxt(gps_path, log, fmt="%.18e")
    np.random.seed(0)
    d = E.run_fusion(str(slam_path), str(gps_path))
    key, pos = np.random.get_state()[1:3]
    for c in (0, copies - 1):
        gk, gp = np_state(st[c])
        np.testing.assert_array_equal(gk, key); assert gp == int(pos)                             # every draw of the run, pre-filter and fit, in the same order
        np.testing.assert_array_equal(g["gps_t_raw"][keep[c]], d["gps"]["timestamps"])
        np.testing.assert_array_equal(va[c], d["valid"])
        np.testing.assert_allclose(al[c][va[c]], d["aligned"][d["valid"]], atol=1e-9, rtol=0)
        if int(r.n_inliers[c]) == len(d["sim3_idx"]):
            np.testing.assert_array_equal(np.where(r.inlier_mask[c].cpu().numpy())[0], d["sim3_idx"])
        np.testing.assert_allclose(r.R[c].cpu().numpy().reshape(3, 3), d["R"], atol=1e-12, rtol=0)
        assert np.abs(p[c] - d["pos"]).max() < 1e-9 and np.abs(q[c] - d["quat"]).max() < 1e-12
        for row, lab in ((0, "raw_slam"), (1, "sim3"), (2, "ekf")):
            e = d["errors"]["primary"][lab]
            assert int(stats[row, c, 0]) == e["count"]
            np.testing.assert_allclose(stats[row, c, 1:], [e["mean"], e["median"], e["rmse"]], rtol=1e-12, atol=1e-9)
    # ---- the host-pointer form of the chain (gsf_run_fusion_batch: what a C / cgo / JNI caller with host arrays calls): the same bits
    from gps_optimize_slam_amd import _lib
    L, ctx, hp = _lib.load(), B.context(), _lib.hptr
    ctx.set_option("ransac_early_exit", 0)
    rc = _lib.RunConfig.from_config(E.CONFIG)
    a = lambda x, dt=np.float64: np.ascontiguousarray(x, dtype=dt)
    h_ts, h_pos, h_quat = a(rep(k["ts"])), a(rep(k["pos"])), a(rep(k["quat"]))
    h_gt, h_llh = a(np.tile(log[:, 0], copies)), a(np.tile(log[:, 1:4], (copies, 1)))
    h_off = np.arange(copies + 1, dtype=np.int64) * n
    h_st = B.mt19937_seed([0] * copies).cpu().numpy().copy()
    o = dict(R=np.empty((copies, 9)), t=np.empty((copies, 3)), s=np.empty(copies), po=np.empty((copies, N, 3)), qo=np.empty((copies, N, 4)),
             st=np.empty(copies, np.int32), ni=np.empty(copies, np.int32), zone=np.empty(copies, np.int32), south=np.empty(copies, np.int32),
             utm=np.empty((copies * n, 3)), keep=np.empty(copies * n, np.uint8), al=np.empty((copies, N, 3)), va=np.empty((copies, N), np.uint8),
             sp=np.empty((copies, N, 3)), err=np.empty((3, copies, 4)), rs=np.empty(copies, np.int32), mask=np.empty((copies, N), np.uint8),
             info=np.empty((copies, 2), np.int32))
    _lib.check(L.gsf_run_fusion_batch(ctx.handle, hp(h_ts), hp(h_pos), hp(h_quat), copies, N, hp(h_gt), hp(h_llh), hp(h_off), C.byref(rc), hp(h_st),
                                      *[hp(o[k_]) for k_ in ("R", "t", "s", "po", "qo", "st", "ni", "zone", "south", "utm", "keep", "al", "va", "sp", "err", "rs", "mask", "info")]))
    for got, dev in ((o["po"], p), (o["qo"], q), (o["st"], status), (o["R"], r.R.cpu().numpy()), (o["s"], r.s.cpu().numpy()), (o["ni"], r.n_inliers.cpu().numpy()),
                     (o["utm"], r.gps_utm.cpu().numpy()), (o["keep"], r.gps_keep.cpu().numpy()), (o["al"], np.nan_to_num(al, nan=-1.0)), (o["va"], r.valid.cpu().numpy()),
                     (o["sp"], r.sim3_pos.cpu().numpy()), (o["err"], stats), (o["rs"], r.run_status.cpu().numpy()), (o["mask"], r.inlier_mask.cpu().numpy()),
                     (o["info"], r.trial_info.cpu().numpy()), (h_st.view(np.int32), st.cpu().numpy())):
        np.testing.assert_array_equal(np.nan_to_num(got, nan=-1.0) if got.dtype == np.float64 else got, dev)


def test_headless_main_process_gui_runs_through_the_chain(B, golden):
    """The 38 runs of the reference's own main_process_gui (14 crafted + 24 random SLAM / GNSS pairs, HeadlessGui: its loaders replaced, so no
    pre-filter) through the chain from the projected log on (pre-filter disabled: the log passes as it is, :139-141): step 2's mask, the
    fit, the fused track, the step-6 rows of the reference's own distance matrices, and -- drawing all max_trials -- the generator where
    the reference leaves np.random; a run that raises leaves NaN rows, its run_status bit and an untouched generator."""
    from gps_optimize_slam_amd import ekfgpsslam as E
    from gps_optimize_slam_amd import _lib
    g, names = cases(golden)
    copies = 3
    seen_fail = 0
    for n in names:
        cfg = case_cfg(E.CONFIG, g[f"{n}_par"])
        cfg["gps_filtering_ransac"] = dict(cfg["gps_filtering_ransac"], enabled=False)
        ts, pos, quat = g[f"{n}_ts"], g[f"{n}_pos"], g[f"{n}_quat"]
        log = np.column_stack((g[f"{n}_gps_t"], g[f"{n}_gps_p"]))
        rep = lambda a: np.repeat(np.asarray(a)[None], copies, axis=0)
        gb = B.GeodeticBatch.from_host(rep(ts), rep(pos), rep(quat), [log] * copies)
        seed = int(g[f"{n}_seed"])
        st = B.mt19937_seed([seed] * copies)
        st0 = st.clone()
        r = B.run_fusion_batch(gb, st, cfg, early_exit=False, projected=True)
        p, q, status = r.fused.host_traj_major()
        va, stats = r.valid.cpu().numpy().astype(bool), r.err_stats.cpu().numpy()
        if bool(g[f"{n}_failed"]) or bool(g[f"{n}_fit_none"]):
            seen_fail += 1
            assert ((r.run_status & _lib.RUN_SIM3_FAILED) != 0).all() and np.isnan(p).all() and (stats[:, :, 0] == 0).all(), n
            if bool(g[f"{n}_failed"]):
                assert (st == st0).all(), n                                                  # the reference raised before it drew (:975)
            continue
        assert (r.run_status == 0).all(), (n, r.run_status.tolist())
        for c in range(copies):
            np.testing.assert_array_equal(va[c], g[f"{n}_valid"], err_msg=n)
            # (the fit sees the chain's OWN alignment here, up to 5e-7 m from scipy's interp1d at |y| ~ 5e6 -- DESIGN section 2 --, and some cases
            # fit four rows: 2e-10 in the scale of first_segment_exactly_min; with the golden's aligned fixes fed in, tests/test_sim3_rows.py holds 1e-11)
            np.testing.assert_allclose(r.R[c].cpu().numpy().reshape(3, 3), g[f"{n}_R"], atol=5e-9, rtol=0, err_msg=n)
            assert abs(float(r.s[c]) - float(g[f"{n}_s"])) < 1e-9, n
            assert np.abs(p[c] - g[f"{n}_ekf_pos"]).max() < POS_TOL and np.abs(q[c] - g[f"{n}_ekf_quat"]).max() < Q_TOL, (n, np.abs(p[c] - g[f"{n}_ekf_pos"]).max())
            ref6 = g[f"{n}_step6"]
            if np.isnan(ref6).all():
                assert (stats[:, c, 0] == 0).all(), n                                        # no point past the first 5 s: the reference printed nothing
            else:
                np.testing.assert_array_equal(stats[:, c, 0], ref6[:, 0], err_msg=n)
                np.testing.assert_allclose(stats[:, c, 1:], ref6[:, 1:], rtol=1e-13, atol=2e-7, err_msg=n)
            gk, gp = np_state(st[c])
            np.testing.assert_array_equal(gk, g[f"{n}_rng_end"][:624], err_msg=n); assert gp == int(g[f"{n}_rng_end"][624]), n
    assert seen_fail >= 1


def _oracle_run(orc, ts, pos, quat, log, cfg, seed):
    """main_process_gui steps 1-6 composed from the oracle's restatements, np.random seeded once (the reference's single global stream)."""
    np.random.seed(seed)
    out = {"status": 0}
    t_raw, lat, lon, alt = log[:, 0], log[:, 1], log[:, 2], log[:, 3]
    m = orc.valid_latlon_mask(lat, lon)
    out["loaded"] = m
    if not m.any():
        out["status"] = 1; return out
    zone, hemi = orc.auto_utm_projection(lon[m], lat[m])
    south = "south" in hemi
    e, n = orc.utm_forward(lat[m], lon[m], zone, south)
    utm = np.column_stack((e, n, alt[m]))
    out.update(zone=zone, south=south, utm=utm)
    ft, fp = orc.filter_gps_outliers_ransac(t_raw[m], utm, cfg["gps_filtering_ransac"])
    keep = np.zeros(len(t_raw), bool); keep[np.where(m)[0][np.isin(t_raw[m], ft)]] = True        # (stamps are unique in these logs)
    out["keep"] = keep
    if len(ft) < 2:
        out["status"] = 2; return out
    al, va = orc.dynamic_time_alignment(ts, ft, fp, max_gap=cfg["time_alignment"]["max_gps_gap_threshold"])
    out.update(aligned=al, valid=va)
    sc = cfg["sim3_ransac"]
    rows = orc.pick_sim3_rows(ts, va, sc["min_samples"], cfg["time_alignment"]["max_gps_gap_threshold"], sc["max_initial_duration"])
    if rows is None:
        out["status"] = 8; return out
    res = orc.compute_sim3_transform_robust(pos[rows], al[rows], sc["min_samples"], sc["residual_threshold"], sc["max_trials"], sc["min_inliers_needed"], return_mask=True)
    if res[0] is None:
        out["status"] = 8; return out
    R, t, s, mask = res
    sp, sq = orc.transform_trajectory(pos, quat, R, t, s)
    po, qo, sto = orc.apply_ekf_correction_aligned(ts, pos, quat, al, va, sp[0], sq[0], cfg, return_status=True)
    out.update(R=R, t=t, s=s, n_inliers=int(mask.sum()), fit_rows=rows[mask], pos=po, quat=qo, st=sto, sim3_pos=sp,
               errs=[orc.evaluate_trajectory_errors(ts, tr, al, va) for tr in (pos, sp, po)])
    return out


@pytest.mark.parametrize("sliding", [True, False])
def test_chain_vs_the_oracles_composition_on_synthetic_logs(B, orc, sliding):
    """64 synthetic geodetic logs (KITTI-04-shaped tracks, fixes at their own stamps, outages = missing fixes) with what a real log holds:
    fixes 60 m off (the pre-filter must drop them -- and with them the draws it makes depend on the data), rows with lat = 0 or out of range
    (the loader drops them before the zone pick), a log thinned to three fixes (pre-filter skipped, :144-146) and one to a single fix
    (ValueError :283).  Against the oracle's composition of the whole flow under ONE seeded generator per trajectory: loaded / kept fixes,
    zone, alignment mask, n_inliers, status words and the final generator state exactly; R, t, s, fused poses and step-6 numbers to the
    gates.  sliding=False: the one-window form of the pre-filter (:148-182)."""
    import torch
    from gps_optimize_slam_amd import ekfgpsslam as E
    nb, N = 64, 271
    src = B.GeodeticBatch.synthetic(nb, N, seed=77)
    offs = src.gps_offsets.cpu().numpy()
    gt, llh = src.gps_t.cpu().numpy(), src.gps_llh.cpu().numpy()
    ts, pos, quat = src.ts.cpu().numpy(), src.pos.cpu().numpy(), src.quat.cpu().numpy()
    rng = np.random.default_rng(5)
    logs = []
    for b in range(nb):
        log = np.column_stack((gt[offs[b]:offs[b + 1]], llh[offs[b]:offs[b + 1]]))
        n = len(log)
        if b % 3 == 0 and n > 40:                                         # fixes thrown 60 m east / north (metres -> degrees at 49 N)
            for r_ in rng.choice(n, size=int(rng.integers(1, 6)), replace=False):
                log[r_, 1] += 60.0 / 111200.0 * rng.choice([-1, 1]); log[r_, 2] += 60.0 / 73000.0 * rng.choice([-1, 1])
        if b % 5 == 1 and n > 40:                                         # rows the loader removes (:259)
            rr = rng.choice(n, size=4, replace=False)
            log[rr[0], 1] = 0.0; log[rr[1], 2] = 0.0; log[rr[2], 1] = 91.0; log[rr[3], 2] = -181.0
        if b == 10: log = log[[0, n // 2, n - 1]]                         # three fixes: fewer than min_samples -> unfiltered
        if b == 11: log = log[[n // 2]]                                   # one fix: "fewer than 2 points"
        if b == 12: log[:, 1] = 0.0                                       # nothing survives the range mask
        logs.append(log)
    cfg = copy.deepcopy(E.CONFIG)
    cfg["gps_filtering_ransac"]["use_sliding_window"] = sliding
    gb = B.GeodeticBatch.from_host(ts, pos, quat, logs)
    seeds = np.arange(nb) + 500
    st = B.mt19937_seed(seeds)
    r = B.run_fusion_batch(gb, st, cfg, early_exit=False)
    p, q, status = r.fused.host_traj_major()
    o2 = gb.gps_offsets.cpu().numpy()
    keep, utm = r.gps_keep.cpu().numpy().astype(bool), r.gps_utm.cpu().numpy()
    va, stats, rs = r.valid.cpu().numpy().astype(bool), r.err_stats.cpu().numpy(), r.run_status.cpu().numpy()
    dropped_any = 0
    for b in range(nb):
        o = _oracle_run(orc, ts[b], pos[b], quat[b], logs[b], cfg, int(seeds[b]))
        key, ppos = np.random.get_state()[1:3]
        gk, gp = np_state(st[b])
        np.testing.assert_array_equal(gk, key, err_msg=str(b)); assert gp == int(ppos), b
        assert rs[b] == o["status"], (b, rs[b], o["status"])
        u = utm[o2[b]:o2[b + 1]]
        np.testing.assert_array_equal(~(np.isnan(u[:, 0]) & np.isnan(u[:, 1])), o["loaded"], err_msg=str(b))
        if o["status"] == 1:
            assert np.isnan(p[b]).all(); continue
        assert int(r.zone[b]) == o["zone"] and bool(r.south[b]) == bool(o["south"])
        np.testing.assert_allclose(u[o["loaded"]], o["utm"], atol=5e-9, rtol=0)
        np.testing.assert_array_equal(keep[o2[b]:o2[b + 1]], o["keep"], err_msg=str(b))
        dropped_any += int(o["keep"].sum() < o["loaded"].sum())
        if o["status"] != 0:
            assert np.isnan(p[b]).all() and (stats[:, b, 0] == 0).all(); continue
        np.testing.assert_array_equal(va[b], o["valid"], err_msg=str(b))
        assert int(r.n_inliers[b]) == o["n_inliers"], b
        np.testing.assert_allclose(r.R[b].cpu().numpy().reshape(3, 3), o["R"], atol=2e-9, rtol=0)
        assert abs(float(r.s[b]) - o["s"]) < 1e-11
        assert np.abs(p[b] - o["pos"]).max() < 1e-6 and np.abs(q[b] - o["quat"]).max() < 1e-8, (b, np.abs(p[b] - o["pos"]).max())
        assert (status[b] & 0xff) == o["st"]
        for row in range(3):
            e = o["errs"][row]
            assert int(stats[row, b, 0]) == e["count"]
            if e["count"]:
                np.testing.assert_allclose(stats[row, b, 1:], [e["mean"], e["median"], e["rmse"]], rtol=1e-12, atol=1e-6)
    assert dropped_any >= 10                                                # the planted fixes really were removed by the filter
    # the same chain stopping early: every output word equal, generators equal where no trial counted every row
    st_e = B.mt19937_seed(seeds)
    re_ = B.run_fusion_batch(gb, st_e, cfg, early_exit=True)
    for a, b_ in ((r.fused.pos, re_.fused.pos), (r.fused.quat, re_.fused.quat), (r.R, re_.R), (r.t, re_.t), (r.s, re_.s), (r.err_stats, re_.err_stats), (r.sim3_pos, re_.sim3_pos)):
        assert torch.equal(torch.nan_to_num(a, nan=-1.0).view(torch.int64), torch.nan_to_num(b_, nan=-1.0).view(torch.int64))
    assert torch.equal(r.n_inliers, re_.n_inliers) and torch.equal(r.gps_keep, re_.gps_keep) and torch.equal(r.run_status, re_.run_status)
    sat = ((re_.fused.status >> 8) & 256) != 0
    assert torch.equal(r.fused.status, re_.fused.status & ~(256 << 8)) and sat.any() and torch.equal(st[~sat], st_e[~sat])


def test_device_window_walk_equals_the_host_walk(B):
    """The windows of the sliding pre-filter found on the device (gsf_gps_prefilter_auto_dev) against the host's walk of the same stamps
    (ekfgpsslam._prefilter_windows, the drop-in's restatement of ref :199-234) fed to gsf_gps_prefilter_chain_dev: same kept rows, same
    generator afterwards -- 400 random logs: 6 ... 400 fixes, durations around and below one window, duplicated last stamps, a step factor
    that makes the tail window matter, window_step <= 1e-6 (the stamp-to-stamp walk of :230-232)."""
    import torch
    from gps_optimize_slam_amd import _lib
    from gps_optimize_slam_amd import ekfgpsslam as E
    L, h = _lib.load(), B.context().handle
    rng = np.random.default_rng(11)
    for width, factor in ((15.0, 0.5), (4.0, 0.7), (6.0, 1.0), (3.0, 0.0)):
        f = dict(E.CONFIG["gps_filtering_ransac"], window_duration_seconds=width, window_step_factor=factor, max_trials=20)
        logs, wins, wo = [], [], [0]
        nl = 100 if factor > 0 else 20
        for b in range(nl):
            n = int(rng.integers(6, 400 if factor > 0 else 60))
            t = np.cumsum(rng.uniform(0.02, 0.25, size=n)) + rng.uniform(0, 100)
            if b % 7 == 0: t[-1] = t[-2]
            pth = np.column_stack((3.0 * t + rng.normal(size=n) * 0.3, -2.0 * t + 0.05 * t * t + rng.normal(size=n) * 0.3, 100 + rng.normal(size=n) * 0.3))
            out = rng.choice(n, size=max(1, n // 25), replace=False); pth[out] += rng.normal(size=(len(out), 3)) * 40
            ranges, _ = E._prefilter_windows(t, f, f["min_samples"])
            assert ranges is not None
            logs.append((t, pth)); wins += ranges; wo.append(len(wins))
        offs = np.zeros(nl + 1, dtype=np.int64); offs[1:] = np.cumsum([len(t) for t, _ in logs])
        T = torch.as_tensor(np.concatenate([t for t, _ in logs])).cuda(); Pp = torch.as_tensor(np.concatenate([p_ for _, p_ in logs])).cuda()
        O = torch.as_tensor(offs).cuda()
        mx = int(max(len(t) for t, _ in logs))
        seeds = np.arange(nl) + 9
        # host-walked windows
        st_a = B.mt19937_seed(seeds)
        keep_a = torch.empty(int(offs[-1]), dtype=torch.uint8, device="cuda")
        WR = torch.as_tensor(np.array(wins, dtype=np.int32).reshape(-1, 2)).cuda() if wins else torch.zeros((1, 2), dtype=torch.int32, device="cuda")
        WO = torch.as_tensor(np.array(wo, dtype=np.int64)).cuda()
        ws, ls_a = torch.empty(max(1, len(wins)), dtype=torch.int32, device="cuda"), torch.empty(nl, dtype=torch.int32, device="cuda")
        _lib.check(L.gsf_gps_prefilter_chain_dev(h, B._p(T), B._p(Pp), B._p(O), nl, B._p(WR), B._p(WO), mx, int(f["max_trials"]), int(f["min_samples"]),
                                                 int(f["polynomial_degree"]), float(f["residual_threshold_meters"]), 0.99, B._p(st_a), B._p(keep_a), B._p(ws), B._p(ls_a)))
        # device-walked windows
        st_b = B.mt19937_seed(seeds)
        keep_b = torch.empty_like(keep_a); ls_b = torch.empty_like(ls_a); info = torch.empty((nl, 2), dtype=torch.int32, device="cuda")
        pc = _lib.PrefilterConfig.from_config(f)
        _lib.check(L.gsf_gps_prefilter_auto_dev(h, B._p(T), B._p(Pp), B._p(O), nl, mx, C.byref(pc), B._p(st_b), B._p(keep_b), B._p(ls_b), B._p(info)))
        assert torch.equal(ls_a, ls_b), (width, factor)
        ok = (ls_a == 0)
        # (the stamp-to-stamp walk ends every log in windows of exactly min_samples rows: min_samples / n = 1 is outside the permutation range
        # of scikit-learn's sampler, both routes flag the log at that window -- and must have walked the same windows up to it)
        assert ok.sum() >= nl // 2 or factor == 0.0
        assert torch.equal(keep_a, keep_b), (width, factor)
        assert torch.equal(st_a, st_b), (width, factor)
        nwin = torch.as_tensor(np.diff(np.array(wo))).cuda().to(torch.int32)
        assert torch.equal(info[ok, 0], nwin[ok]), (width, factor)
        assert (st_a != B.mt19937_seed(seeds)).any(dim=1).sum() >= nl // 2                      # the logs really drew


@pytest.mark.gpu
def test_prefilter_speculative_pass_and_batch_size_change_no_word(B):
    """gsf_set_option "prefilter_speculate" / "prefilter_first_batch" / "prefilter_miss_batch" only change how many trials the pre-filter chain draws AHEAD of
    scikit-learn's walk (three axes' first trials at once; a first batch of 1 / 4 / 16 trials): the walk over the trials is the sequential
    one, in order, so kept rows, log status, windows processed / succeeded and the generator state are the same words -- on logs where
    most windows stop after one trial (clean stretches), on logs where a third of the windows need several (5 % of the fixes 40 m off on
    one axis -- east, north or up from log to log, so the speculative pass misses at the first, second or third axis and re-speculates for what is left), and on logs where most do (15 %), with
    windows longer than the register tile (20 Hz: 300 rows) in the mix."""
    import torch
    from gps_optimize_slam_amd import _lib
    from gps_optimize_slam_amd import ekfgpsslam as E
    L, ctx = _lib.load(), B.context()
    rng = np.random.default_rng(5)
    f = dict(E.CONFIG["gps_filtering_ransac"])
    logs = []
    for b in range(96):
        rate = 20.0 if b % 8 == 0 else 10.0
        n = int(rng.integers(40, 700 if rate > 10 else 400))
        t = np.arange(n) / rate + rng.uniform(0, 50)
        p = np.column_stack((3.0 * t + rng.normal(size=n) * 0.2, -2.0 * t + 0.02 * t * t + rng.normal(size=n) * 0.2, 100 + rng.normal(size=n) * 0.2))
        share = (0.0, 0.05, 0.15)[b % 3]
        hit = rng.random(n) < share
        p[hit, (b // 3) % 3] += 40.0                                      # the axis that misses differs from log to log: every re-speculation path runs
        logs.append((t, p))
    offs = np.zeros(len(logs) + 1, dtype=np.int64); offs[1:] = np.cumsum([len(t) for t, _ in logs])
    T = torch.as_tensor(np.concatenate([t for t, _ in logs])).cuda(); P = torch.as_tensor(np.concatenate([p for _, p in logs])).cuda()
    O = torch.as_tensor(offs).cuda()
    mx, nl = int(max(len(t) for t, _ in logs)), len(logs)
    pc = _lib.PrefilterConfig.from_config(f)
    outs = {}
    try:
        for spec, fb, mb in ((0, 1, 4), (0, 4, 4), (0, 16, 4), (1, 1, 1), (1, 1, 4), (1, 1, 16), (1, 4, 2)):
            ctx.set_option("prefilter_speculate", spec); ctx.set_option("prefilter_first_batch", fb); ctx.set_option("prefilter_miss_batch", mb)
            st = B.mt19937_seed(np.arange(nl) + 77)
            keep = torch.empty(int(offs[-1]), dtype=torch.uint8, device="cuda"); ls = torch.empty(nl, dtype=torch.int32, device="cuda")
            info = torch.empty((nl, 2), dtype=torch.int32, device="cuda")
            _lib.check(L.gsf_gps_prefilter_auto_dev(ctx.handle, B._p(T), B._p(P), B._p(O), nl, mx, C.byref(pc), B._p(st), B._p(keep), B._p(ls), B._p(info)))
            outs[(spec, fb, mb)] = (keep, ls, info, st)
    finally:
        ctx.set_option("prefilter_speculate", 1); ctx.set_option("prefilter_first_batch", 1); ctx.set_option("prefilter_miss_batch", 4)
    ref = outs[(0, 1, 4)]
    assert (ref[1] == 0).all() and (ref[2][:, 0] > 0).all()
    kept = ref[0].double().mean().item()
    assert 0.85 < kept < 0.99                                              # fixes were dropped, most were kept
    for key, cur in outs.items():
        for a, b_ in zip(ref, cur):
            assert torch.equal(a, b_), key
