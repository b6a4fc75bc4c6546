#!/usr/bin/env python3
"""Time the REFERENCE ITSELF next to the oracle -- build container only (needs /root/reference; SURVEY 8(d), last sentence).

The reference module is imported the way tests/golden/gen_golden.py imports it (placeholder modules for pyproj / tkinter, which are
absent offline); its own functions run on one core (it is single-threaded Python):
  * apply_ekf_correction (EKFGPSSLAM.py:831-935)       -> fused poses per second
  * compute_sim3_transform_robust (:389-426)           -> ms per call (1 000 trials)
  * steps 2-5 of main_process_gui (:971-1010)          -> poses per second end to end
on (i) the bundled KITTI-04 track against its GNSS file (config C1), (ii) a 271-pose and (iii) a 1 000-pose synthetic track with a
mid-track outage (the shape of the bench batches).  The oracle (oracle/gsf_oracle.c, the bench's cpu_baseline "port") runs on the
same inputs, so the ratio port / reference is on record.  Writes profiles/r04_reference_timing.json, which bench.py quotes beside
cpu_baseline.value and BASELINE.md section 2 tabulates.

    python tests/campaigns/time_reference.py
"""
import contextlib
import datetime
import io
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference"


def best_of(fn, repeats):
    best = 1e99
    for _ in range(repeats):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    if not os.path.isdir(REF):
        sys.exit("tests/campaigns/time_reference.py runs in the build container only: /root/reference is not here")
    with contextlib.redirect_stdout(io.StringIO()):
        import gen_golden as G                                            # imports the reference (stubbed pyproj / tkinter), nothing else
    ref = G.ref
    from oracle import oracle as orc
    cfg, sc = ref.CONFIG, ref.CONFIG["sim3_ransac"]
    rng = np.random.default_rng(5)
    out = {"where": "build container (no GPU)", "date": datetime.date.today().isoformat(), "cpu": platform.processor() or platform.machine(),
           "cores_used": 1, "python": sys.version.split()[0], "numpy": np.__version__, "cases": {}}
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu"] = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass

    def case(name, ts, pos, quat, gps_t, gps_p, repeats):
        slam = {"timestamps": ts, "positions": pos, "quaternions": quat}
        gps = {"timestamps": gps_t, "positions": gps_p}
        with contextlib.redirect_stdout(io.StringIO()):
            aligned, valid = ref.dynamic_time_alignment(slam, gps, cfg["time_alignment"])
        idx = orc.pick_sim3_rows(ts, valid, sc["min_samples"], cfg["time_alignment"]["max_gps_gap_threshold"], sc["max_initial_duration"])
        src, dst = pos[idx], aligned[idx]
        np.random.seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            R, t, s = ref.compute_sim3_transform_robust(src, dst, sc["min_samples"], sc["residual_threshold"], sc["max_trials"], sc["min_inliers_needed"])
            sp, sq = ref.transform_trajectory(pos, quat, R, t, s)
        n = len(ts)

        def ref_steps_2_to_5():
            a, v = ref.dynamic_time_alignment(slam, gps, cfg["time_alignment"])
            np.random.seed(0)
            R_, t_, s_ = ref.compute_sim3_transform_robust(pos[idx], a[idx], sc["min_samples"], sc["residual_threshold"], sc["max_trials"], sc["min_inliers_needed"])
            p_, q_ = ref.transform_trajectory(pos, quat, R_, t_, s_)
            ref.apply_ekf_correction(slam, gps, p_, q_, cfg)

        def orc_steps_2_to_5():
            a, v = orc.dynamic_time_alignment(ts, gps_t, gps_p)
            np.random.seed(0)
            R_, t_, s_ = orc.compute_sim3_transform_robust(pos[idx], a[idx], sc["min_samples"], sc["residual_threshold"], sc["max_trials"], sc["min_inliers_needed"])
            p_, q_ = orc.transform_trajectory(pos, quat, R_, t_, s_)
            orc.apply_ekf_correction_aligned(ts, pos, quat, a, v, p_[0], q_[0])

        t_ref_ekf = best_of(lambda: ref.apply_ekf_correction(slam, gps, sp, sq, cfg), repeats)
        t_orc_ekf = best_of(lambda: orc.apply_ekf_correction_aligned(ts, pos, quat, aligned, valid, sp[0], sq[0]), max(repeats, 20))
        t_ref_rs = best_of(lambda: ref.compute_sim3_transform_robust(src, dst, sc["min_samples"], sc["residual_threshold"], sc["max_trials"], sc["min_inliers_needed"]), repeats)
        t_orc_rs = best_of(lambda: orc.compute_sim3_transform_robust(src, dst, sc["min_samples"], sc["residual_threshold"], sc["max_trials"], sc["min_inliers_needed"]), repeats)
        t_ref_all = best_of(ref_steps_2_to_5, repeats)
        t_orc_all = best_of(orc_steps_2_to_5, repeats)
        # the two agree on what they computed (the timing is of like for like)
        with contextlib.redirect_stdout(io.StringIO()):
            pr, _ = ref.apply_ekf_correction(slam, gps, sp, sq, cfg)
        po, _ = orc.apply_ekf_correction_aligned(ts, pos, quat, aligned, valid, sp[0], sq[0])
        out["cases"][name] = {
            "poses": n, "valid_rows": int(valid.sum()), "sim3_rows": int(len(idx)),
            "apply_ekf_correction": {"reference_ms": t_ref_ekf * 1e3, "reference_poses_per_s": n / t_ref_ekf, "oracle_ms": t_orc_ekf * 1e3,
                                     "oracle_poses_per_s": n / t_orc_ekf, "oracle_over_reference": t_ref_ekf / t_orc_ekf,
                                     "max_abs_pos_diff_m": float(np.abs(np.asarray(pr) - po).max())},
            "compute_sim3_transform_robust": {"reference_ms": t_ref_rs * 1e3, "oracle_ms_incl_numpy_draws": t_orc_rs * 1e3, "trials": sc["max_trials"]},
            "steps_2_to_5": {"reference_ms": t_ref_all * 1e3, "reference_poses_per_s": n / t_ref_all, "oracle_ms": t_orc_all * 1e3,
                             "oracle_poses_per_s": n / t_orc_all},
        }
        c = out["cases"][name]
        print(f"{name:28s} {n:5d} poses | EKF ref {c['apply_ekf_correction']['reference_poses_per_s']:9.0f} poses/s, oracle {c['apply_ekf_correction']['oracle_poses_per_s']:11.0f} "
              f"({c['apply_ekf_correction']['oracle_over_reference']:.0f}x) | robust fit ref {t_ref_rs * 1e3:6.1f} ms | steps 2-5 ref {c['steps_2_to_5']['reference_poses_per_s']:7.0f} poses/s")

    # (i) bundled KITTI-04: SLAM track vs the GT positions as metric GNSS (KAT-3's inputs) and vs the raw GNSS log of C1
    k = np.load(os.path.join(ROOT, "tests", "golden", "kat_bundled.npz"))
    c1 = np.load(os.path.join(ROOT, "tests", "golden", "c1_combined.npz"))
    case("kitti04_vs_gt_positions", k["ts"], k["pos"], k["quat"], k["ts"], k["gt"], 5)
    case("kitti04_vs_combined_gnss", k["ts"], k["pos"], k["quat"], c1["gps_t"], c1["gps_p"], 5)
    # (ii), (iii) synthetic tracks of the bench shapes, one mid-track outage each
    for n, cut in ((271, (120, 190)), (1000, (400, 480))):
        ts, pos, quat, gps, _, _ = G.synth_traj(rng, n, yaw_rate_deg=1.0)
        keep = np.ones(n, bool); keep[cut[0]:cut[1]] = False
        case(f"synthetic_{n}_with_outage", ts, pos, quat, ts[keep], gps[keep], 3)
    c271 = out["cases"]["synthetic_271_with_outage"]
    out["headline"] = {"reference_python_poses_per_s": c271["apply_ekf_correction"]["reference_poses_per_s"],
                       "oracle_poses_per_s_same_inputs": c271["apply_ekf_correction"]["oracle_poses_per_s"],
                       "oracle_over_reference": c271["apply_ekf_correction"]["oracle_over_reference"],
                       "what": "apply_ekf_correction (EKFGPSSLAM.py:831-935) on a 271-pose synthetic track with one outage, best of 3, one core"}
    path = os.path.join(ROOT, "profiles", "r04_reference_timing.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
