#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the resulting *.npz
files are pure data (inputs + the reference's outputs) and are what travels.
The reference module is imported with placeholder modules for the two packages
that are absent offline (pyproj, tkinter) -- SURVEY.md 8(c).  Nothing from the
reference's source is written anywhere.

Library versions the goldens are pinned to are stored in each file's `meta`.
UTM inputs are produced with the oracle's Krueger series (pyproj is absent:
"parity unpinned" for that one stage) and are STORED in the goldens, so every
downstream golden is self-contained.

    python tests/golden/gen_golden.py
"""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
os.environ["MPLBACKEND"] = "Agg"
sys.dont_write_bytecode = True


def _import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class Proj:  # placeholder: the real projection is unavailable offline
        def __init__(self, *a, **k):
            raise RuntimeError("pyproj is not installed")

    class CRSError(Exception):
        pass

    stub("pyproj", Proj=Proj)
    stub("pyproj.exceptions", CRSError=CRSError)
    tk = stub("tkinter")
    tk.filedialog = stub("tkinter.filedialog")
    tk.messagebox = stub("tkinter.messagebox")
    sys.path.insert(0, REF)
    import EKFGPSSLAM as ref  # noqa
    return ref


ref = _import_reference()
from oracle import oracle as orc  # noqa: E402  (only for the UTM stage inputs)
import scipy  # noqa: E402
import sklearn  # noqa: E402

META = json.dumps({"numpy": np.__version__, "scipy": scipy.__version__, "sklearn": sklearn.__version__,
                   "python": sys.version.split()[0], "reference": "A2ureeE/GPS-optimize-SLAM @ 2025-05-23"})


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, meta=np.array(META), **arrs)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB, {len(arrs)} arrays")


def cfg_copy(**over):
    import copy
    c = copy.deepcopy(ref.CONFIG)
    for k, v in over.items():
        sec, key = k.split("__")
        c[sec][key] = v
    return c


class AlignRecorder:
    """Wrap/replace ref.dynamic_time_alignment to record (or inject) its outputs."""

    def __init__(self, inject=None):
        self.inject, self.calls, self._orig = inject, [], ref.dynamic_time_alignment

    def __enter__(self):
        def wrapped(slam, gps, tcfg):
            out = self.inject if self.inject is not None else self._orig(slam, gps, tcfg)
            out = (np.array(out[0], dtype=float), np.array(out[1], dtype=bool))
            self.calls.append(out)
            return out[0].copy(), out[1].copy()
        ref.dynamic_time_alignment = wrapped
        return self

    def __exit__(self, *a):
        ref.dynamic_time_alignment = self._orig


def run_ekf(ts, pos, quat, aligned, valid, sim3_pos, sim3_quat, cfg):
    """apply_ekf_correction with a crafted alignment result."""
    slam = {"timestamps": ts, "positions": pos, "quaternions": quat}
    with quiet(), AlignRecorder(inject=(aligned, valid)):
        p, q = ref.apply_ekf_correction(slam, {"timestamps": ts, "positions": aligned}, sim3_pos, sim3_quat, cfg)
    return np.array(p), np.array(q)


# --------------------------------------------------------------------------
def gen_kat_bundled():
    with quiet():
        slam = ref.load_slam_trajectory(f"{REF}/yolotum04.txt")
    gt = np.loadtxt(f"{REF}/04.txt")[:, [3, 7, 11]]
    ts = slam["timestamps"]
    with quiet():
        R, t, s = ref.compute_sim3_transform(slam["positions"], gt)                     # KAT-1
        sp, sq = ref.transform_trajectory(slam["positions"], slam["quaternions"], R, t, s)  # KAT-2
    out = dict(ts=ts, pos=slam["positions"], quat=slam["quaternions"], gt=gt, kat1_R=R, kat1_t=t,
               kat1_s=np.float64(s), kat2_pos=sp, kat2_quat=sq)
    # KAT-3: all valid
    with quiet(), AlignRecorder() as rec:
        p3, q3 = ref.apply_ekf_correction(slam, {"timestamps": ts, "positions": gt}, sp, sq, ref.CONFIG)
    out.update(kat3_aligned=rec.calls[0][0], kat3_valid=rec.calls[0][1], kat3_pos=p3, kat3_quat=q3)
    # KAT-4: GPS rows 100..160 deleted
    keep = np.ones(len(ts), bool); keep[100:161] = False
    with quiet(), AlignRecorder() as rec:
        p4, q4 = ref.apply_ekf_correction(slam, {"timestamps": ts[keep], "positions": gt[keep]}, sp, sq, ref.CONFIG)
    out.update(kat4_gps_t=ts[keep], kat4_gps_p=gt[keep], kat4_aligned=rec.calls[0][0], kat4_valid=rec.calls[0][1],
               kat4_pos=p4, kat4_quat=q4)
    with quiet():
        out["kat5_offset"] = np.float64(ref.estimate_time_offset(ts, ts[keep] + 0.37, 500))
    save("kat_bundled.npz", **out)
    return slam


# --------------------------------------------------------------------------
def gen_c1(slam, gps_file, tag):
    """Config C1: steps 1(geodesy)..5 of main_process_gui (ref :959-1010) on bundled files."""
    try:
        raw = np.loadtxt(gps_file, delimiter=" ")
    except ValueError:
        raw = np.loadtxt(gps_file, delimiter=",")
    t_raw, lat, lon, alt = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]        # ref :258 (Q1)
    m = (np.abs(lat) <= 90) & (np.abs(lon) <= 180) & (lat != 0) & (lon != 0)  # ref :259
    t_raw, lat, lon, alt = t_raw[m], lat[m], lon[m], alt[m]
    zone, hemi = ref.auto_utm_projection(lon, lat)                            # ref :266 (the reference's own)
    e, n = orc.utm_forward(lat, lon, zone, "south" in hemi)                   # stands in for ref :270
    utm = np.column_stack((e, n, alt))
    np.random.seed(0)
    with quiet():
        ft, fp = ref.filter_gps_outliers_ransac(t_raw, utm, ref.CONFIG["gps_filtering_ransac"])  # ref :275
    gps = {"timestamps": ft, "positions": fp}
    with quiet():
        aligned, valid = ref.dynamic_time_alignment(slam, gps, ref.CONFIG["time_alignment"])   # ref :971
    vi = np.where(valid)[0]
    # Sim3 subset pick, ref :977-998 (restated from the driver; all bundled cases take the "timed" branch)
    vt = slam["timestamps"][vi]
    gaps = np.where(np.diff(vt) > ref.CONFIG["time_alignment"]["max_gps_gap_threshold"])[0]
    end = gaps[0] if len(gaps) > 0 else len(vi)
    first = vi[:end]
    ms = ref.CONFIG["sim3_ransac"]["min_samples"]
    if len(first) < ms:
        idx = vi
    else:
        lim = slam["timestamps"][first] <= slam["timestamps"][first[0]] + ref.CONFIG["sim3_ransac"]["max_initial_duration"]
        idx = first[lim] if lim.sum() >= ms else first
    src, dst = slam["positions"][idx], aligned[idx]
    # record the RNG draws of ref :405
    draws, orig_choice = [], np.random.choice

    def rec_choice(*a, **k):
        r = orig_choice(*a, **k); draws.append(np.array(r)); return r
    final_fit = {}
    orig_fit = ref.compute_sim3_transform

    def rec_fit(s_, d_):
        final_fit["n"] = s_.shape[0]
        return orig_fit(s_, d_)
    np.random.seed(0)
    np.random.choice = rec_choice; ref.compute_sim3_transform = rec_fit
    try:
        sc = ref.CONFIG["sim3_ransac"]
        with quiet():
            R, t, s = ref.compute_sim3_transform_robust(src, dst, sc["min_samples"], sc["residual_threshold"],
                                                        sc["max_trials"], sc["min_inliers_needed"])
    finally:
        np.random.choice = orig_choice; ref.compute_sim3_transform = orig_fit
    with quiet():
        sp, sq = ref.transform_trajectory(slam["positions"], slam["quaternions"], R, t, s)     # ref :1006
        with AlignRecorder() as rec:
            cp, cq = ref.apply_ekf_correction(slam, gps, sp, sq, ref.CONFIG)                   # ref :1010
    # reference-style error metric (Q15, ref :1017-1033) of the fused track vs primary GPS
    from scipy.spatial import distance
    post = vi[slam["timestamps"][vi] > slam["timestamps"][0] + 5.0]
    cand = aligned[post]
    errs = {}
    for lab, tr in (("sim3", sp), ("ekf", cp)):
        d = distance.cdist(tr[post], cand, "euclidean").min(axis=1)
        errs[lab] = np.array([d.mean(), np.median(d), np.sqrt((d ** 2).mean())])
    save(f"c1_{tag}.npz", gps_t_raw=t_raw, lat=lat, lon=lon, alt=alt, zone=np.int32(zone),
         south=np.int32("south" in hemi), utm=utm, gps_t=ft, gps_p=fp, aligned=aligned, valid=valid,
         sim3_idx=idx.astype(np.int32), sample_idx=np.stack(draws).astype(np.int32), n_inliers=np.int32(final_fit["n"]),
         R=R, t=t, s=np.float64(s), sim3_pos=sp, sim3_quat=sq, ekf_aligned=rec.calls[0][0], ekf_valid=rec.calls[0][1],
         ekf_pos=cp, ekf_quat=cq, err_sim3=errs["sim3"], err_ekf=errs["ekf"])
    print(f"   {tag}: zone {zone}{hemi!r} valid {valid.sum()}/{len(valid)} inliers {final_fit['n']}/{len(idx)} "
          f"scale {s:.6f} rmse sim3 {errs['sim3'][2]:.4f} ekf {errs['ekf'][2]:.4f}")


# --------------------------------------------------------------------------
def rand_quat(rng, n=None):
    q = rng.normal(size=(4,) if n is None else (n, 4))
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def rand_rot(rng):
    from scipy.spatial.transform import Rotation
    return Rotation.from_quat(rand_quat(rng)).as_matrix()


def gen_sim3_cases():
    rng = np.random.default_rng(20250523)
    out, names = {}, []

    def add(name, src, dst):
        with quiet():
            R, t, s = ref.compute_sim3_transform(src, dst)
        out[f"{name}_src"], out[f"{name}_dst"] = src, dst
        out[f"{name}_none"] = np.bool_(R is None)
        if R is not None:
            out[f"{name}_R"], out[f"{name}_t"], out[f"{name}_s"] = R, t, np.float64(s)
        names.append(name)

    for i, n in enumerate([3, 4, 5, 50, 271, 1000]):
        src = rng.normal(size=(n, 3)) * [30, 5, 120]
        R0, t0, s0 = rand_rot(rng), rng.normal(size=3) * 1e3 + [4.5e5, 5.4e6, 100], rng.uniform(0.5, 2.0)
        dst = s0 * src @ R0.T + t0 + rng.normal(size=(n, 3)) * 0.3
        add(f"planted{i}", src, dst)
    src = rng.normal(size=(40, 3)) * 10
    add("reflect", src, src * [1, 1, -1] + rng.normal(size=(40, 3)) * 0.01)      # det<0 branch :441
    add("n2", src[:2], src[:2] + 1.0)                                              # :430
    add("n0", np.empty((0, 3)), np.empty((0, 3)))
    add("zerovar", np.tile([[1.0, 2.0, 3.0]], (6, 1)), rng.normal(size=(6, 3)))   # :445
    add("tinyscale", src, src * 1e-8)                                              # :450
    add("planar", src * [1, 1, 0], (src * [1, 1, 0]) @ rand_rot(rng).T * 1.3 + 5)  # rank-2 H
    add("kitti4", src[:4] * [0.02, 0.02, 1.0], (src[:4] * [0.02, 0.02, 1.0]) * 0.99 + rng.normal(size=(4, 3)) * 0.4)
    out["names"] = np.array(names)

    # transform_trajectory: hit every branch of Rotation.from_matrix
    from scipy.spatial.transform import Rotation
    tn = []
    pos, quat = rng.normal(size=(17, 3)) * 50, rng.normal(size=(17, 4)) * rng.uniform(0.1, 5, size=(17, 1))
    mats = {"tr": Rotation.from_euler("z", 0.3).as_matrix(),
            "m00": Rotation.from_euler("x", 3.0).as_matrix(),
            "m11": Rotation.from_euler("y", 3.0).as_matrix(),
            "m22": Rotation.from_euler("z", 3.0).as_matrix(),
            "rnd": rand_rot(rng)}
    for k, Rm in mats.items():
        t, s = rng.normal(size=3) * 100, rng.uniform(0.5, 2)
        with quiet():
            p, q = ref.transform_trajectory(pos, quat, Rm, t, s)
        out[f"tt_{k}_R"], out[f"tt_{k}_t"], out[f"tt_{k}_s"], out[f"tt_{k}_pos"], out[f"tt_{k}_quat"] = Rm, t, np.float64(s), p, q
        tn.append(k)
    out["tt_in_pos"], out["tt_in_quat"], out["tt_names"] = pos, quat, np.array(tn)

    # RANSAC (ref :389-426), seeded, indices recorded
    rn = []

    def add_ransac(name, src, dst, ms=4, thr=4.0, trials=200, need=4, seed=1):
        draws, orig_choice = [], np.random.choice

        def rec_choice(*a, **k):
            r = orig_choice(*a, **k); draws.append(np.array(r)); return r
        fin, orig_fit = {}, ref.compute_sim3_transform

        def rec_fit(s_, d_):
            fin["src"] = s_.copy(); return orig_fit(s_, d_)
        np.random.seed(seed); np.random.choice = rec_choice; ref.compute_sim3_transform = rec_fit
        try:
            with quiet():
                R, t, s = ref.compute_sim3_transform_robust(src, dst, ms, thr, trials, need)
        finally:
            np.random.choice = orig_choice; ref.compute_sim3_transform = orig_fit
        out[f"rs_{name}_src"], out[f"rs_{name}_dst"] = src, dst
        out[f"rs_{name}_par"] = np.array([ms, thr, trials, need], dtype=float)
        out[f"rs_{name}_idx"] = (np.stack(draws) if draws else np.empty((0, ms))).astype(np.int32)
        out[f"rs_{name}_none"] = np.bool_(R is None)
        if R is not None:
            out[f"rs_{name}_R"], out[f"rs_{name}_t"], out[f"rs_{name}_s"] = R, t, np.float64(s)
            # inlier mask = rows of src that went into the final fit (positions are unique)
            fs = fin["src"]
            mask = np.array([any((row == f).all() for f in fs) for row in src])
            out[f"rs_{name}_mask"] = mask
        rn.append(name)

    n = 120
    src = np.cumsum(rng.normal(size=(n, 3)) * [0.05, 0.03, 1.4], axis=0)
    R0, t0, s0 = rand_rot(rng), np.array([4.5e5, 5.4e6, 110.0]), 1.07
    dst = s0 * src @ R0.T + t0 + rng.normal(size=(n, 3)) * 0.45
    bad = rng.choice(n, 25, replace=False)
    dst_o = dst.copy(); dst_o[bad] += rng.normal(size=(25, 3)) * 30
    add_ransac("outliers", src, dst_o)
    add_ransac("clean", src, dst, trials=50, seed=2)
    add_ransac("fail", src, dst_o, thr=1e-3, need=30, trials=30, seed=3)      # best < min_inliers_needed -> None
    add_ransac("short", src[:3], dst[:3])                                     # n < min_samples -> None
    add_ransac("ms5", src, dst_o, ms=5, thr=2.0, trials=80, need=10, seed=4)
    out["rs_names"] = np.array(rn)
    save("sim3_cases.npz", **out)


# --------------------------------------------------------------------------
def synth_traj(rng, n, yaw_rate_deg=3.0, burst=None, big=True):
    """Small KITTI-like camera-frame trajectory (z forward, y down) + its "GPS" (metric frame)."""
    from scipy.spatial.transform import Rotation
    ts = np.arange(n) * 0.104 + rng.uniform(-0.002, 0.002, size=n)
    ts[0] = 0.0
    yaw = np.deg2rad(yaw_rate_deg) * ts          # heading: about camera y
    zrot = np.deg2rad(2.0) * np.sin(ts)          # what the reference calls "yaw": euler 'zyx'[0], about z
    if burst is not None:
        a, b, rate = burst
        extra = np.zeros(n)
        extra[a:b] = np.deg2rad(rate) * np.diff(ts, prepend=ts[0])[a:b]
        yaw = yaw + np.cumsum(extra)
        zrot = zrot + np.cumsum(extra)
    pos = np.zeros((n, 3))
    for i in range(1, n):
        d = 1.45 * np.array([np.sin(yaw[i]), 0.002, np.cos(yaw[i])])
        pos[i] = pos[i - 1] + d
    quat = (Rotation.from_euler("z", zrot) * Rotation.from_euler("y", yaw)).as_quat() * rng.choice([-1, 1], size=(n, 1))
    quat = quat * rng.uniform(0.98, 1.02, size=(n, 1))            # not exactly unit, like real files
    slam_pos = pos * 0.93 + rng.normal(size=(n, 3)) * 0.01
    Rg, tg = rand_rot(rng), (np.array([4.58e5, 5.43e6, 112.0]) if big else np.array([10.0, -5.0, 2.0]))
    gps = pos @ Rg.T + tg + rng.normal(size=(n, 3)) * 0.45
    return ts, slam_pos, quat, gps, Rg, tg


def gen_ekf_cases():
    rng = np.random.default_rng(7)
    out, names = {}, []

    def add(name, ts, pos, quat, aligned, valid, cfg=None, cfg_over=None):
        cfg = cfg or ref.CONFIG
        vi = np.where(valid & ~np.isnan(aligned).any(axis=1))[0]
        with quiet():
            R, t, s = ref.compute_sim3_transform(pos[vi], aligned[vi])
            if R is None:
                R, t, s = np.eye(3), np.zeros(3), 1.0
            try:
                sp, sq = ref.transform_trajectory(pos, quat, R, t, s)
            except ValueError:   # zero quaternion in the file: init from an identity-rotated copy
                q2 = quat.copy(); q2[np.linalg.norm(q2, axis=1) == 0] = [0, 0, 0, 1]
                sp, sq = ref.transform_trajectory(pos, q2, R, t, s)
        p, q = run_ekf(ts, pos, quat, aligned, valid, sp, sq, cfg)
        for k, v in dict(ts=ts, pos=pos, quat=quat, aligned=aligned, valid=valid, sp0=sp[0], sq0=sq[0], out_pos=p,
                         out_quat=q).items():
            out[f"{name}_{k}"] = v
        out[f"{name}_cfg"] = np.array(json.dumps(cfg_over or {}))
        names.append(name)

    n = 90
    ts, pos, quat, gps, _, _ = synth_traj(rng, n)
    allv = np.ones(n, bool)
    add("allvalid", ts, pos, quat, gps, allv)
    v = allv.copy(); v[30:55] = False
    g = gps.copy(); g[30:55] = np.nan
    add("outage_rts", ts, pos, quat, g, v)
    add("outage_valid_but_nan", ts, pos, quat, g, allv)                  # Q10: mask True, NaN meas
    v2 = allv.copy(); v2[:12] = False
    add("start_in_outage", ts, pos, quat, gps, v2)                       # Q11
    v3 = allv.copy(); v3[70:] = False
    add("end_in_outage", ts, pos, quat, gps, v3)                         # Q11 trailing
    v4 = allv.copy(); v4[10:20] = False; v4[40] = False; v4[60:75] = False
    add("three_outages_one_single", ts, pos, quat, gps, v4)              # L<2 -> default RTS (:893)
    v5 = np.zeros(n, bool)
    add("never_valid", ts, pos, quat, gps, v5)                           # pure dead reckoning
    v6 = np.zeros(n, bool); v6[45] = True
    add("single_fix", ts, pos, quat, gps, v6)
    # sharp turn inside the outage
    ts_b, pos_b, quat_b, gps_b, _, _ = synth_traj(rng, n, burst=(35, 45, 70.0))
    vb = allv.copy(); vb[30:55] = False
    add("sharp_steps0", ts_b, pos_b, quat_b, gps_b, vb)                  # no RTS, hard update
    c5 = cfg_copy(rts_decision__default_ekf_transition_steps_on_sharp_turn=5)
    add("sharp_steps5", ts_b, pos_b, quat_b, gps_b, vb, cfg=c5,
        cfg_over={"rts_decision": {"default_ekf_transition_steps_on_sharp_turn": 5}})     # blend w=0.2 (:762-767)
    c1 = cfg_copy(rts_decision__default_ekf_transition_steps_on_sharp_turn=1)
    add("sharp_steps1", ts_b, pos_b, quat_b, gps_b, vb, cfg=c1,
        cfg_over={"rts_decision": {"default_ekf_transition_steps_on_sharp_turn": 1}})
    clow = cfg_copy(rts_decision__sharp_turn_yaw_rate_threshold_deg_per_sec=1.0)
    add("lowthr_everything_sharp", ts, pos, quat, g, v, cfg=clow,
        cfg_over={"rts_decision": {"sharp_turn_yaw_rate_threshold_deg_per_sec": 1.0}})
    # non-monotonic / repeated stamps (Q9)
    ts_n = ts.copy(); ts_n[20] = ts_n[19]; ts_n[50] = ts_n[48]
    add("nonmonotonic", ts_n, pos, quat, g, v)
    # zero quaternion in SLAM (relative-pose fallback :84-86; inside outage -> sharp-turn True :821)
    qz = quat.copy(); qz[15] = 0.0; qz[40] = 0.0
    add("zero_quat", ts, pos, qz, g, v)
    # custom noise
    cn = cfg_copy(ekf__initial_cov_diag=[1.0, 2.0, 3.0, 0.1, 0.2, 0.3, 0.4],
                  ekf__process_noise_diag=[0.5, 0.05, 1.5, 0.02, 0.03, 0.04, 0.05],
                  ekf__meas_noise_diag=[0.05, 1.0, 4.0])
    add("custom_noise", ts, pos, quat, g, v, cfg=cn,
        cfg_over={"ekf": {"initial_cov_diag": [1.0, 2.0, 3.0, 0.1, 0.2, 0.3, 0.4],
                          "process_noise_diag": [0.5, 0.05, 1.5, 0.02, 0.03, 0.04, 0.05],
                          "meas_noise_diag": [0.05, 1.0, 4.0]}})
    # tiny trajectories
    add("n1", ts[:1], pos[:1], quat[:1], gps[:1], allv[:1])
    add("n2", ts[:2], pos[:2], quat[:2], gps[:2], allv[:2])
    add("n2_first_invalid", ts[:2], pos[:2], quat[:2], gps[:2], np.array([False, True]))
    add("n3_mid_invalid", ts[:3], pos[:3], quat[:3], gps[:3], np.array([True, False, True]))
    # long 1k-pose trajectory with an outage (C3 shape)
    ts_l, pos_l, quat_l, gps_l, _, _ = synth_traj(rng, 1000, yaw_rate_deg=1.0)
    vl = np.ones(1000, bool); vl[400:480] = False
    add("n1000_outage", ts_l, pos_l, quat_l, gps_l, vl)
    out["names"] = np.array(names)
    save("ekf_cases.npz", **out)


# --------------------------------------------------------------------------
def gen_helper_cases():
    rng = np.random.default_rng(11)
    out = {}
    # calculate_relative_pose
    P1, Q1, P2, Q2 = rng.normal(size=(12, 3)) * 10, rng.normal(size=(12, 4)), rng.normal(size=(12, 3)) * 10, rng.normal(size=(12, 4))
    Q1[3] = 0.0; Q2[5] = 0.0
    dps, dqs = [], []
    for a, b, c, d in zip(P1, Q1, P2, Q2):
        with quiet():
            dp, dq = ref.calculate_relative_pose(a, b, c, d)
        dps.append(dp); dqs.append(dq)
    out.update(rp_p1=P1, rp_q1=Q1, rp_p2=P2, rp_q2=Q2, rp_dp=np.array(dps), rp_dq=np.array(dqs))
    # quaternion_nlerp
    A, B = rand_quat(rng, 10), rand_quat(rng, 10)
    W = np.array([0.0, 0.2, 0.5, 0.8, 1.0, -0.3, 1.7, 0.5, 0.3, 0.7])
    B[7] = -A[7]            # dot<0 flip then identical
    A[8] = [1, 0, 0, 0]; B[8] = [-1, 0, 0, 0]
    out.update(nl_a=A, nl_b=B, nl_w=W, nl_out=np.array([ref.quaternion_nlerp(a, b, w) for a, b, w in zip(A, B, W)]))
    # is_sharp_turn_in_segment
    from scipy.spatial.transform import Rotation
    sh_names = []
    def add_sh(name, quats, stamps, thr):
        with quiet():
            r = ref.is_sharp_turn_in_segment(list(quats), list(stamps), thr)
        out[f"sh_{name}_q"], out[f"sh_{name}_t"], out[f"sh_{name}_thr"], out[f"sh_{name}_r"] = quats, stamps, np.float64(thr), np.bool_(r)
        sh_names.append(name)
    st = np.arange(20) * 0.1
    add_sh("gentle", Rotation.from_euler("zyx", np.c_[np.deg2rad(20) * st, 0.1 * np.ones(20), 0.05 * st]).as_quat(), st, np.deg2rad(45))
    add_sh("sharp", Rotation.from_euler("zyx", np.c_[np.deg2rad(60) * st, 0.1 * np.ones(20), 0.05 * st]).as_quat(), st, np.deg2rad(45))
    add_sh("pitch_only", Rotation.from_euler("y", np.deg2rad(80) * st).as_quat(), st, np.deg2rad(45))
    add_sh("wrap", Rotation.from_euler("z", np.deg2rad(175) + np.deg2rad(30) * st).as_quat(), st, np.deg2rad(45))
    add_sh("single", rand_quat(rng, 1), st[:1], 0.1)
    st2 = st.copy(); st2[5] = st2[4]; st2[9] = st2[7]
    add_sh("nonmono", Rotation.from_euler("z", np.deg2rad(50) * st).as_quat(), st2, np.deg2rad(45))
    qbad = rand_quat(rng, 6); qbad[3] = 0
    add_sh("badquat", qbad, st[:6], 10.0)
    add_sh("random", rand_quat(rng, 30) * 1.7, np.sort(rng.uniform(0, 3, 30)), 2.0)
    out["sh_names"] = np.array(sh_names)
    # rts_smoother_segment: diagonal (as in the pipeline) and dense SPD covariances
    for tag, dense in (("diag", False), ("dense", True)):
        L = 9
        xf, xp = rng.normal(size=(L, 7)), rng.normal(size=(L, 7))
        def spd():
            if dense:
                M = rng.normal(size=(7, 7)); return M @ M.T + np.eye(7) * 0.5
            return np.diag(rng.uniform(0.05, 2.0, size=7))
        Pf, Pp = np.array([spd() for _ in range(L)]), np.array([spd() for _ in range(L)])
        with quiet():
            xs, Ps = ref.rts_smoother_segment(list(xf), list(Pf), list(xp), list(Pp))
        out.update({f"rts_{tag}_xf": xf, f"rts_{tag}_Pf": Pf, f"rts_{tag}_xp": xp, f"rts_{tag}_Pp": Pp,
                    f"rts_{tag}_xs": np.array(xs), f"rts_{tag}_Ps": np.array(Ps)})
    # ExtendedKalmanFilter.process_step sequences (class surface, ref :679-772), with blending
    for tag, steps, ovr in (("hard", 0, None), ("blend4", 4, None), ("override3", 10, 3)):
        ekf_cfg = dict(ref.CONFIG["ekf"])
        f = ref.ExtendedKalmanFilter(np.array([1.0, 2.0, 3.0]), np.array([0.1, 0.2, 0.3, 0.9]) * 2, ekf_cfg)
        f.current_transition_steps = steps
        f.gnss_available_prev = False
        avail_seq = [False, False, True, True, True, True, True, False, True, True]
        rec = {k: [] for k in ("dp", "dq", "z", "avail", "dt", "state", "cov", "ps", "pc", "w")}
        for av in avail_seq:
            dp, dq = rng.normal(size=3), rand_quat(rng) * rng.uniform(0.5, 2)
            z, dt = f.state[:3] + rng.normal(size=3) * 0.5, rng.uniform(0.05, 0.2)
            with quiet():
                s_, c_, ps, pc = f.process_step((dp, dq), z if av else None, av, dt, override_transition_steps=ovr)
            for k, vv in zip(rec, (dp, dq, z, av, dt, s_.copy(), c_.copy(), ps.copy(), pc.copy(), f.gnss_update_weight)):
                rec[k].append(vv)
        out.update({f"ps_{tag}_{k}": np.array(vv) for k, vv in rec.items()})
        out[f"ps_{tag}_par"] = np.array([steps, -1 if ovr is None else ovr])
    save("helper_cases.npz", **out)


# --------------------------------------------------------------------------
def gen_align_cases(slam):
    rng = np.random.default_rng(3)
    out, names = {}, []

    def add(name, st, gt, gp, gap=5.0):
        tc = {"max_samples_for_corr": 500, "max_gps_gap_threshold": gap}
        with quiet():
            al, va = ref.dynamic_time_alignment({"timestamps": st}, {"timestamps": gt, "positions": gp}, tc)
        out.update({f"{name}_st": st, f"{name}_gt": gt, f"{name}_gp": gp, f"{name}_gap": np.float64(gap),
                    f"{name}_al": al, f"{name}_va": va})
        names.append(name)

    st = slam["timestamps"]
    raw = np.loadtxt(f"{REF}/combined_output.txt")
    e, n = orc.utm_forward(raw[:, 1], raw[:, 2], 32, False)
    add("combined", st, raw[:, 0], np.column_stack((e, n, raw[:, 3])))
    gt = np.sort(rng.uniform(0, 30, 200)); gp = np.c_[np.sin(gt) * 50 + 4.5e5, gt * 13 + 5.4e6, np.cos(gt / 3) + 100]
    add("random_knots", st, gt, gp)
    keep = (gt < 8) | (gt > 14.5) & (gt < 20) | (gt > 26)
    add("two_gaps", st, gt[keep], gp[keep])
    # exact duplicate stamps carry IDENTICAL positions: which duplicate np.unique keeps after the reference's
    # unstable argsort (:339) is implementation-defined, so goldens must not depend on it
    perm = rng.permutation(len(gt)); gtd, gpd = np.r_[gt[perm], gt[:10]], np.r_[gp[perm], gp[:10]]
    add("unsorted_dups", st, gtd, gpd)
    # short segments: 1 point (skipped), 2 and 3 points (linear), 4 points (cubic)
    gts = np.array([0.0, 6.0, 6.5, 13.0, 13.4, 13.9, 20.0, 20.3, 20.9, 21.2, 27.0])
    add("short_segments", st, gts, np.c_[gts ** 2, -gts * 3 + 7, np.sin(gts)])
    add("slam_outside", st + 100.0, gt, gp)
    add("one_gps", st, gt[:1], gp[:1])
    add("exact_knots", gt[5:60].copy(), gt, gp)
    add("small_gap_thr", st, gt, gp, gap=0.2)
    out["names"] = np.array(names)
    save("align_cases.npz", **out)


# --------------------------------------------------------------------------
def gen_filter_cases():
    """filter_gps_outliers_ransac (ref :136-247, scikit-learn's RANSACRegressor on the global legacy RNG): inputs, config, seed ->
    the kept rows and one np.random.random() drawn right after the call (pins how much of the RNG stream was consumed)."""
    rng = np.random.default_rng(11)
    out, names = {}, []
    base = {"enabled": True, "use_sliding_window": True, "window_duration_seconds": 15.0, "window_step_factor": 0.5,
            "polynomial_degree": 2, "min_samples": 6, "residual_threshold_meters": 10.0, "max_trials": 50}

    def add(name, t, p, seed, **over):
        cfg = dict(base); cfg.update(over)
        np.random.seed(seed)
        with quiet():
            ft, fp = ref.filter_gps_outliers_ransac(t.copy(), p.copy(), cfg)
        after = np.random.random()
        out.update({f"{name}_t": t, f"{name}_p": p, f"{name}_seed": np.int64(seed), f"{name}_ft": np.array(ft), f"{name}_fp": np.array(fp),
                    f"{name}_after": np.float64(after),
                    f"{name}_cfg": np.array([cfg["use_sliding_window"], cfg["window_duration_seconds"], cfg["window_step_factor"], cfg["polynomial_degree"],
                                             cfg["min_samples"], cfg["residual_threshold_meters"], cfg["max_trials"]], dtype=np.float64)})
        names.append(name)

    for tag, f in (("kitti04gps", f"{REF}/5.1Kitti04gps"), ("combined", f"{REF}/combined_output.txt")):
        raw = np.loadtxt(f)
        e, n = orc.utm_forward(raw[:, 1], raw[:, 2], 32, False)
        add(f"bundled_{tag}", raw[:, 0], np.column_stack((e, n, raw[:, 3])), 0)

    def track(n, n_out, spike=(30.0, 200.0), noise=0.5):
        t = np.arange(n) * 0.1 + rng.uniform(-0.01, 0.01, n); t[0] = 0.0
        p = np.c_[4.5e5 + 12.0 * t + 30 * np.sin(t / 7), 5.4e6 + 3.0 * t + 0.2 * t * t, 110 + np.cos(t / 5)] + rng.normal(0, noise, (n, 3))
        bad = rng.choice(n, n_out, replace=False)
        p[bad] += rng.choice([-1.0, 1.0], (n_out, 3)) * rng.uniform(spike[0], spike[1], (n_out, 3)) * (rng.random((n_out, 3)) < 0.6)
        return t, p

    t, p = track(300, 15)
    add("spikes_s1", t, p, 1); add("spikes_s2", t, p, 2)
    add("spikes_global", t, p, 3, use_sliding_window=False)
    add("spikes_deg1", t, p, 4, polynomial_degree=1, residual_threshold_meters=25.0)
    add("spikes_deg3", t, p, 5, polynomial_degree=3, min_samples=8)
    add("tight_thr", t, p, 6, residual_threshold_meters=1.2, max_trials=100)
    add("few_trials", t, p, 7, max_trials=5, min_samples=4)
    t2, p2 = track(400, 160, spike=(15.0, 60.0))
    add("heavy_outliers", t2, p2, 8)
    add("heavy_outliers_global", t2, p2, 9, use_sliding_window=False, max_trials=100)
    add("too_few_points", t[:5], p[:5], 10)
    add("short_windows", t[::12], p[::12], 11, window_duration_seconds=4.0)          # windows with fewer than min_samples rows
    add("zero_stride", t[:80], p[:80], 12, window_step_factor=0.0, window_duration_seconds=3.0)
    perm = rng.permutation(300)
    add("unsorted", t[perm], p[perm], 13)
    add("disabled", t, p, 14, enabled=False)
    out["names"] = np.array(names)
    save("gpsfilter_cases.npz", **out)


# --------------------------------------------------------------------------
def gen_step6_with_ground_truth(slam):
    """Step 6 of main_process_gui with the optional second GNSS file (ref :949-953, :963-966, :1013-1075): primary GPS =
    combined_output.txt (the C1 'combined' case above, same seed), ground truth = 5.1Kitti04gps through load_gps_data's path with
    CONFIG['ground_truth_gps_filtering'].  The metric of :1027-1033 / :1049-1056 for the raw SLAM, the Sim3 and the EKF track against
    both, and which of the two the plot would use (:1064-1075).  (The two bundled files are read per Q1 and land in different UTM
    zones: the numbers are huge and meaningless as accuracy, which makes them a good arithmetic check.)"""
    from scipy.spatial import distance
    c1 = np.load(os.path.join(HERE, "c1_combined.npz"))
    gps = {"timestamps": c1["gps_t"], "positions": c1["gps_p"]}
    sim3_pos, ekf_pos = c1["sim3_pos"], c1["ekf_pos"]
    raw = np.loadtxt(f"{REF}/5.1Kitti04gps", delimiter=" ")
    t_raw, lat, lon, alt = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]        # ref :258 (Q1)
    m = (np.abs(lat) <= 90) & (np.abs(lon) <= 180) & (lat != 0) & (lon != 0)  # ref :259
    t_raw, lat, lon, alt = t_raw[m], lat[m], lon[m], alt[m]
    zone, hemi = ref.auto_utm_projection(lon, lat)
    e, n = orc.utm_forward(lat, lon, zone, "south" in hemi)                   # stands in for ref :270
    np.random.seed(0)
    with quiet():
        gt_t, gt_p = ref.filter_gps_outliers_ransac(t_raw, np.column_stack((e, n, alt)), ref.CONFIG["ground_truth_gps_filtering"])  # ref :964
        al_p, va_p = ref.dynamic_time_alignment(slam, gps, ref.CONFIG["time_alignment"])                       # ref :1014
        al_g, va_g = ref.dynamic_time_alignment(slam, {"timestamps": gt_t, "positions": gt_p}, ref.CONFIG["time_alignment"])   # ref :1037
    out = dict(gt_t_raw=t_raw, gt_lat=lat, gt_lon=lon, gt_alt=alt, gt_zone=np.int32(zone), gt_south=np.int32("south" in hemi),
               gt_t=gt_t, gt_p=gt_p, aligned_primary=al_p, valid_primary=va_p, aligned_gt=al_g, valid_gt=va_g)
    ekf_err = {}
    for tag, al, va in (("primary", al_p, va_p), ("gt", al_g, va_g)):
        vi = np.where(va)[0]
        post = vi[slam["timestamps"][vi] > slam["timestamps"][0] + 5.0]      # ref :1020-1023 / :1041-1044
        cand = al[post]
        rows = []
        for tr in (slam["positions"], sim3_pos, ekf_pos):                     # ref :1027 / :1049: raw SLAM, Sim3, EKF
            d = distance.cdist(tr[post], cand, "euclidean").min(axis=1)       # ref :1030-1031
            rows.append([len(d), d.mean(), np.median(d), np.sqrt((d ** 2).mean())])
            last = d
        ekf_err[tag] = last
        out[f"err_{tag}"] = np.array(rows)                                    # rows: raw SLAM, Sim3, EKF; cols: count, mean, median, rmse
        out[f"post_idx_{tag}"] = post.astype(np.int32)
    # ref :1064-1075: ground truth first, then the primary GPS
    out["plot_ref"] = np.array("gt" if len(gt_p) >= 2 and len(ekf_err["gt"]) > 0 else ("primary" if len(ekf_err["primary"]) > 0 else "none"))
    save("step6_gt.npz", **out)
    print(f"   step 6: GT zone {zone}{hemi!r}, {len(out['post_idx_gt'])} GT points / {len(out['post_idx_primary'])} primary points, plot ref {out['plot_ref']}")


# --------------------------------------------------------------------------
class HeadlessGui:
    """Runs the reference's OWN main_process_gui (ref :940-1123) without a display: the tk dialogs, the two file loaders and the
    plot are replaced on the imported module; everything between them -- the time alignment of step 2, the Sim3 row selection
    :973-998, the robust fit :1002, steps 4-5 -- is the reference's code, and what it hands from step to step is recorded."""

    def __init__(self, slam, gps):
        self.slam, self.gps, self.rec = slam, gps, {}

    def run(self, seed):
        names = ("select_slam_file", "select_gps_file", "load_slam_trajectory", "load_gps_data", "plot_results",
                 "compute_sim3_transform_robust", "transform_trajectory", "apply_ekf_correction")
        saved = {k: getattr(ref, k) for k in names}
        rec = self.rec
        ref.select_slam_file = lambda: "slam.txt"
        ref.select_gps_file = lambda *a, **k: "gps.txt"
        ref.load_slam_trajectory = lambda path: {k: v.copy() for k, v in self.slam.items()}
        ref.load_gps_data = lambda path, data_label="GPS", filter_config_override=None: {k: (None if v is None else v.copy()) for k, v in self.gps.items()}
        ref.plot_results = lambda *a, **k: None
        ref.messagebox.askyesno = lambda *a, **k: False
        ref.messagebox.showerror = lambda title, msg: rec.__setitem__("error", str(msg).splitlines()[0])

        def robust(src, dst, *a, **k):
            rec["src"], rec["dst"] = np.array(src), np.array(dst)
            out = saved["compute_sim3_transform_robust"](src, dst, *a, **k)
            rec["fit"] = out
            return out

        def transform(*a, **k):
            out = saved["transform_trajectory"](*a, **k)
            rec["sim3_pos"], rec["sim3_quat"] = np.array(out[0]), np.array(out[1])
            return out

        def ekf(*a, **k):
            with AlignRecorder() as ar:
                out = saved["apply_ekf_correction"](*a, **k)
            rec["ekf_aligned"], rec["ekf_valid"] = ar.calls[0]
            rec["ekf_pos"], rec["ekf_quat"] = np.array(out[0]), np.array(out[1])
            return out
        ref.compute_sim3_transform_robust, ref.transform_trajectory, ref.apply_ekf_correction = robust, transform, ekf
        # step 6 (ref :1027-1033): the distance matrices the reference's own loop forms for raw SLAM / Sim3 / EKF against the primary GPS, in its
        # order; the printed mean / median / RMSE are :1033's expressions of their row minima (:1031)
        orig_cdist = ref.distance.cdist
        mats = []

        def cdist_rec(a, b, *aa, **kk):
            d = orig_cdist(a, b, *aa, **kk); mats.append(np.array(d)); return d
        ref.distance.cdist = cdist_rec
        np.random.seed(seed)
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()), AlignRecorder() as ar:
                ref.main_process_gui()
            if ar.calls:
                rec["aligned"], rec["valid"] = ar.calls[0]                   # step 2's alignment (ref :971)
        finally:
            ref.distance.cdist = orig_cdist
            for k, v in saved.items():
                setattr(ref, k, v)
        if len(mats) >= 3:
            e = [np.min(m, axis=1) for m in mats[:3]]                        # :1031
            rec["step6"] = np.array([[len(x), np.mean(x), np.median(x), np.sqrt(np.mean(x ** 2))] for x in e])   # :1033
        rec["stdout"] = buf.getvalue()
        return rec


def _rows_case_adder(out, names, base, sec):
    """add(name, ...): one headless run of main_process_gui on a crafted SLAM / GNSS pair, its rows, branch, fit and fused poses into `out`"""
    def add(name, ts, pos, quat, gps_rows, seed, max_dur=None, gap=None, min_samples=None, gps_t=None, gps_p=None):
        for k in sec:
            ref.CONFIG[k].clear(); ref.CONFIG[k].update(base[k])
        if max_dur is not None: ref.CONFIG["sim3_ransac"]["max_initial_duration"] = max_dur
        if gap is not None: ref.CONFIG["time_alignment"]["max_gps_gap_threshold"] = gap
        if min_samples is not None: ref.CONFIG["sim3_ransac"]["min_samples"] = min_samples
        slam = {"timestamps": ts, "positions": pos, "quaternions": quat}
        gps = {"timestamps": gps_t[gps_rows], "positions": gps_p[gps_rows], "projector": None}
        r = HeadlessGui(slam, gps).run(seed)
        failed = "src" not in r
        idx = np.empty(0, np.int32)
        if not failed:
            idx = np.array([int(np.where((pos == row).all(axis=1))[0][0]) for row in r["src"]], dtype=np.int32)
            assert np.array_equal(r["dst"], r["aligned"][idx])
        # which branch of :982-996 the reference took, read off its own printed description (:999)
        so = r["stdout"]
        branch = 2 if "第一段太短" in so else (1 if "移除了时间阈值" in so else 0)
        o = dict(ts=ts, pos=pos, quat=quat, gps_t=gps["timestamps"], gps_p=gps["positions"], seed=np.int64(seed),
                 par=np.array([ref.CONFIG["time_alignment"]["max_gps_gap_threshold"], ref.CONFIG["sim3_ransac"]["max_initial_duration"],
                               ref.CONFIG["sim3_ransac"]["min_samples"]], dtype=np.float64),
                 aligned=r.get("aligned", np.empty((0, 3))), valid=r.get("valid", np.empty(0, bool)), sim3_idx=idx,
                 branch=np.int32(branch), failed=np.bool_(failed), error=np.array(r.get("error", "")))
        fit_none = failed or r["fit"][0] is None
        o["fit_none"] = np.bool_(fit_none)
        if not fit_none:
            o.update(R=r["fit"][0], t=r["fit"][1], s=np.float64(r["fit"][2]), sim3_pos0=r["sim3_pos"][0], sim3_quat0=r["sim3_quat"][0],
                     ekf_pos=r["ekf_pos"], ekf_quat=r["ekf_quat"], ekf_aligned=r["ekf_aligned"], ekf_valid=r["ekf_valid"])
            # step 6 against the primary GPS (rows raw SLAM / Sim3 / EKF x count, mean, median, RMSE); absent when no point lies past the first 5 s
            o["step6"] = r.get("step6", np.full((3, 4), np.nan))
            # the final state of np.random: where the run's draws (ref :405) left the global generator
            st = np.random.get_state()
            o["rng_end"] = np.concatenate([st[1].astype(np.uint32), np.array([st[2]], dtype=np.uint32)])
        for k, v in o.items():
            out[f"{name}_{k}"] = v
        names.append(name)
        print(f"   {name}: valid {int(np.sum(o['valid']))}/{len(ts)} -> {len(idx)} rows, branch {branch}, failed {failed}"
              + ("" if failed else f", rows {idx[0]}..{idx[-1]}") + (f" [{o['error']}]" if failed else ""))

    return add


def gen_sim3_rows_cases():
    """Which time-synchronised rows feed the global Sim3 (ref :973-998) -- decided by the reference's own main_process_gui, run
    headless (HeadlessGui) on crafted SLAM / GNSS pairs; plus what steps 3-5 make of them (seeded draws).  The rows are recovered
    from the (src, dst) arrays main_process_gui hands to compute_sim3_transform_robust (SLAM positions are unique)."""
    rng = np.random.default_rng(973)
    out, names = {}, []
    sec = ("time_alignment", "sim3_ransac")
    base = {k: dict(ref.CONFIG[k]) for k in sec}

    add = _rows_case_adder(out, names, base, sec)

    n = 300
    ts, pos, quat, gps, _, _ = synth_traj(rng, n)
    allr = np.arange(n)
    keep = lambda *cut: np.setdiff1d(allr, np.concatenate([np.arange(a, b) for a, b in cut]))
    kw = dict(gps_t=ts, gps_p=gps)
    add("all_valid", ts, pos, quat, allr, 1, **kw)
    add("gap_in_first_180s", ts, pos, quat, keep((100, 161)), 2, **kw)                 # first = vi[:99]: the row before the gap is dropped too
    add("first_segment_3_rows", ts, pos, quat, keep((3, 61)), 3, **kw)                 # :984-986 -> all valid rows
    add("first_segment_exactly_min", ts, pos, quat, keep((5, 70)), 4, **kw)            # vi[:4]: 4 rows == min_samples, used
    add("starts_in_outage", ts, pos, quat, keep((0, 40)), 5, **kw)
    add("two_gaps", ts, pos, quat, keep((80, 140), (200, 260)), 6, **kw)
    add("timed_too_short", ts, pos, quat, keep((150, 215)), 7, max_dur=0.25, **kw)     # :993-995 -> whole first segment
    add("short_duration_limit", ts, pos, quat, keep((150, 215)), 8, max_dur=6.0, **kw)
    add("too_few_valid", ts, pos, quat, np.array([10, 11, 12]), 9, **kw)               # :975 ValueError
    add("ends_in_outage", ts, pos, quat, keep((220, 300)), 10, **kw)
    add("min_samples_6", ts, pos, quat, keep((5, 70)), 11, min_samples=6, **kw)        # first segment of 4 rows < 6 -> all valid rows
    # a SLAM stamp that jumps ahead: a "gap" that comes from the SLAM stamps alone (np.diff of the valid rows' stamps, :979)
    tj = ts.copy(); tj[60] += 7.0
    add("slam_stamp_jump", tj, pos, quat, allr, 12, **kw)
    # a stamp out of order inside the first segment: the duration limit is a per-row mask, not a prefix (:990)
    tn = ts.copy(); tn[20] += 4.0
    add("limit_mask_not_prefix", tn, pos, quat, allr, 13, max_dur=3.0, **kw)
    # > 180 s: the default duration limit bites (2 000 poses = 208 s)
    ts_l, pos_l, quat_l, gps_l, _, _ = synth_traj(rng, 2000, yaw_rate_deg=0.5)
    add("longer_than_180s", ts_l, pos_l, quat_l, np.arange(2000), 14, gps_t=ts_l, gps_p=gps_l)
    for k in sec:
        ref.CONFIG[k].clear(); ref.CONFIG[k].update(base[k])
    out["names"] = np.array(names)
    save("sim3_rows_cases.npz", **out)


# --------------------------------------------------------------------------
def gen_random_ekf_tracks():
    """64 random 120-pose tracks through the REFERENCE's apply_ekf_correction (stacked arrays, one file): several outages per track (1 pose
    to a third of the track, at the start, at the end), fixes that are NaN with the mask set, yaw bursts inside outages (sharp-turn
    recoveries), repeated stamps -- the kind of input tests/test_gpu_parity.py's stress generator makes, but with the reference's own outputs,
    so that the oracle AND the kernels are compared with the reference itself on inputs nobody picked by hand."""
    rng = np.random.default_rng(20251004)
    nb, n = 64, 120
    T, P, Q, A, V, SP, SQ, OP, OQ = [], [], [], [], [], [], [], [], []
    for b in range(nb):
        burst = None
        valid = np.ones(n, bool)
        for _ in range(int(rng.integers(0, 4))):
            L = int(rng.choice([1, 2, 3, 8, 25, 40]))
            s0 = int(rng.integers(0, n - L))
            valid[s0:s0 + L] = False
            if L >= 8 and burst is None and rng.random() < 0.6:
                a = s0 + 1 + int(rng.integers(0, L - 4))
                burst = (a, a + 3, float(rng.choice([-1, 1]) * rng.uniform(60.0, 200.0)))
        if rng.random() < 0.15: valid[:int(rng.integers(1, 30))] = False
        if rng.random() < 0.15: valid[n - int(rng.integers(1, 30)):] = False
        ts, pos, quat, gps, _, _ = synth_traj(rng, n, yaw_rate_deg=float(rng.uniform(-4, 4)), burst=burst)
        if rng.random() < 0.3:
            k = int(rng.integers(2, n - 2)); ts[k] = ts[k - 1]                     # repeated stamp (Q9)
        aligned = gps.copy(); aligned[~valid] = np.nan
        nanfix = (rng.random(n) < 0.02) & valid
        aligned[nanfix, int(rng.integers(0, 3))] = np.nan                           # NaN fix, mask still set (Q10)
        vi = np.where(valid & ~np.isnan(aligned).any(axis=1))[0]
        with quiet():
            R, t, s = (None, None, None) if len(vi) < 3 else ref.compute_sim3_transform(pos[vi], aligned[vi])
            if R is None:
                R, t, s = np.eye(3), np.zeros(3), 1.0
            sp, sq = ref.transform_trajectory(pos, quat, R, t, s)
        p, q = run_ekf(ts, pos, quat, aligned, valid, sp, sq, ref.CONFIG)
        for lst, v in zip((T, P, Q, A, V, SP, SQ, OP, OQ), (ts, pos, quat, aligned, valid, sp[0], sq[0], p, q)):
            lst.append(np.asarray(v))
    save("ekf_random_tracks.npz", ts=np.stack(T), pos=np.stack(P), quat=np.stack(Q), aligned=np.stack(A), valid=np.stack(V), sp0=np.stack(SP), sq0=np.stack(SQ),
         out_pos=np.stack(OP), out_quat=np.stack(OQ))


def gen_sim3_rows_random():
    """24 RANDOM SLAM / GNSS pairs through the reference's own main_process_gui (headless): random outages (some longer than the gap
    threshold, some not), SLAM stamp jumps at / next to the threshold, random duration limits and min_samples -- rows, branch, robust fit
    (seeded draws) and fused poses as the reference produces them.  Same keys as sim3_rows_cases.npz."""
    rng = np.random.default_rng(97398)
    out, names = {}, []
    sec = ("time_alignment", "sim3_ransac")
    base = {k: dict(ref.CONFIG[k]) for k in sec}
    add = _rows_case_adder(out, names, base, sec)
    for c in range(24):
        n = int(rng.integers(100, 161))
        ts, pos, quat, gps, _, _ = synth_traj(rng, n, yaw_rate_deg=float(rng.uniform(-3, 3)))
        rows = np.ones(n, bool)
        for _ in range(int(rng.integers(0, 4))):
            L = int(rng.choice([1, 3, 20, 48, 49, 50, 60, 70]))
            a = int(rng.integers(0, max(1, n - L)))
            rows[a:a + L] = False
        if rng.random() < 0.2: rows[:int(rng.integers(1, 40))] = False
        if rng.random() < 0.2: rows[n - int(rng.integers(1, 40)):] = False
        if rows.sum() < 8: rows[:: max(1, n // 10)] = True
        tsj = ts.copy()
        if rng.random() < 0.3:
            k = int(rng.integers(5, n - 5)); tsj[k:] += float(rng.choice([4.9, 5.05, 7.0]))          # a gap made by the SLAM stamps alone
        kw = {}
        if rng.random() < 0.4: kw["max_dur"] = float(rng.choice([0.25, 3.0, 6.0]))
        if rng.random() < 0.25: kw["min_samples"] = 6
        add(f"rnd{c:02d}", tsj, pos, quat, np.where(rows)[0], 100 + c, gps_t=tsj, gps_p=gps, **kw)
    for k in sec:
        ref.CONFIG[k].clear(); ref.CONFIG[k].update(base[k])
    out["names"] = np.array(names)
    save("sim3_rows_random.npz", **out)


if __name__ == "__main__":
    if "--only-step6" in sys.argv:
        with quiet():
            slam_ = ref.load_slam_trajectory(f"{REF}/yolotum04.txt")
        gen_step6_with_ground_truth(slam_)
        sys.exit(0)
    if "--only-sim3-rows" in sys.argv:
        gen_sim3_rows_cases()
        sys.exit(0)
    if "--only-random-ekf" in sys.argv:
        gen_random_ekf_tracks()
        sys.exit(0)
    if "--only-sim3-rows-random" in sys.argv:
        gen_sim3_rows_random()
        sys.exit(0)
    if "--only-filter" in sys.argv:                  # later additions regenerate alone: the other files stay byte-identical
        gen_filter_cases()
        sys.exit(0)
    slam = gen_kat_bundled()
    gen_c1(slam, f"{REF}/5.1Kitti04gps", "kitti04gps")
    gen_c1(slam, f"{REF}/combined_output.txt", "combined")
    gen_sim3_cases()
    gen_ekf_cases()
    gen_helper_cases()
    gen_align_cases(slam)
    gen_filter_cases()
    gen_step6_with_ground_truth(slam)
    gen_sim3_rows_cases()
    gen_random_ekf_tracks()
    gen_sim3_rows_random()
