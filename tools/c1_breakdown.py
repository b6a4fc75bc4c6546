"""Where the C1 drop-in's wall time goes (271-pose KITTI-04 track, warm): every host-level call of steps 1-6 timed on its own."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gps_optimize_slam_amd import ekfgpsslam as E

gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
k, g = np.load(os.path.join(gold, "kat_bundled.npz")), np.load(os.path.join(gold, "c1_combined.npz"))
slam = {"timestamps": k["ts"].copy(), "positions": k["pos"].copy(), "quaternions": k["quat"].copy()}
ts, lats, lons, alts = g["gps_t_raw"].copy(), g["lat"].copy(), g["lon"].copy(), g["alt"].copy()
config = copy.deepcopy(E.CONFIG); sc = config["sim3_ransac"]
np.random.seed(0)


def timed(fn, reps=20):
    fn()
    best, tot, out = 1e9, 0.0, None
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); dt = time.perf_counter() - t0
        best, tot = min(best, dt), tot + dt
    return out, best * 1e3, tot / reps * 1e3


rows = []
zone, hemi = E.auto_utm_projection(lons, lats)
projector = E.UtmProjector(zone, "south" in hemi)
(x, y), b, m = timed(lambda: projector(lons, lats)); rows.append(("UTM forward (279 fixes)", b, m))
(ft, fp), b, m = timed(lambda: E.filter_gps_outliers_ransac(ts, np.column_stack((x, y, alts)), config["gps_filtering_ransac"])); rows.append(("GPS pre-filter", b, m))
gps = {"timestamps": ft, "positions": fp, "projector": projector}
(aligned, valid), b, m = timed(lambda: E.dynamic_time_alignment(slam, gps, config["time_alignment"])); rows.append(("time alignment", b, m))
idx = E.pick_sim3_indices(slam, valid, config)
(R, t, s), b, m = timed(lambda: E.compute_sim3_transform_robust(slam["positions"][idx], aligned[idx], sc["min_samples"], sc["residual_threshold"],
                                                               sc["max_trials"], sc["min_inliers_needed"])); rows.append(("robust Sim3 (1 000 trials)", b, m))
(sp, sq), b, m = timed(lambda: E.transform_trajectory(slam["positions"], slam["quaternions"], R, t, s)); rows.append(("apply Sim3", b, m))
(pos, quat), b, m = timed(lambda: E.apply_ekf_correction(slam, gps, sp, sq, config)); rows.append(("EKF + RTS (incl. its own alignment)", b, m))
_, b, m = timed(lambda: E.evaluate_trajectory_errors(slam["timestamps"], pos, aligned, valid)); rows.append(("error metric", b, m))
for name, b, m in rows:
    print(f"{name:40s} best {b:8.3f} ms   mean {m:8.3f} ms")
print(f"{'sum':40s} best {sum(r[1] for r in rows):8.3f} ms   mean {sum(r[2] for r in rows):8.3f} ms")
