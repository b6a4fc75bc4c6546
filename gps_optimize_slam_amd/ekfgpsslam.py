"""Drop-in surface of the reference's EKFGPSSLAM.py hot path, backed by the gfx950 kernels.

Same function names, argument order, return conventions and error behaviour as the reference
(/root/reference/EKFGPSSLAM.py, lines cited per function); arrays are C-order float64 NumPy,
quaternions scalar-last.  Every numerical stage runs on the GPU through libgsf.so -- there is no
CPU fallback: without the library / a device the first call raises GsfError.

What stays host Python, as in the reference: text I/O (np.loadtxt/np.savetxt) and the walk over the window stamps of the GPS
pre-filter.  The reference's random draws (np.random.choice of the Sim3 RANSAC, scikit-learn's sampler in the pre-filter) are
made ON THE DEVICE from NumPy's global legacy generator state, which is handed back advanced exactly as the reference leaves
it, so a seeded run reproduces the reference's rows.  Time alignment, the pre-filter's polynomial RANSAC and the error metric
are device kernels too.
"""
import copy
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EkfConfig, GsfError, check, f64, hptr  # noqa: F401

# EKFGPSSLAM.py:22-71
CONFIG = {
    "ekf": {
        "initial_cov_diag": [0.1, 0.1, 0.1, 0.01, 0.01, 0.01, 0.01],
        "process_noise_diag": [0.1, 0.1, 0.7, 0.01, 0.01, 0.01, 0.01],
        "meas_noise_diag": [0.2, 0.2, 0.2],
        "transition_steps": 10,
    },
    "sim3_ransac": {
        "min_samples": 4,
        "residual_threshold": 4.0,
        "max_trials": 1000,
        "min_inliers_needed": 4,
        "max_initial_duration": 180.0,
    },
    "gps_filtering_ransac": {
        "enabled": True, "use_sliding_window": True, "window_duration_seconds": 15.0, "window_step_factor": 0.5,
        "polynomial_degree": 2, "min_samples": 6, "residual_threshold_meters": 10.0, "max_trials": 50,
    },
    "time_alignment": {"max_samples_for_corr": 500, "max_gps_gap_threshold": 5.0},
    "ground_truth_gps_filtering": {
        "enabled": False, "use_sliding_window": True, "window_duration_seconds": 15.0, "window_step_factor": 0.5,
        "polynomial_degree": 2, "min_samples": 6, "residual_threshold_meters": 5.0, "max_trials": 50,
    },
    "rts_decision": {"sharp_turn_yaw_rate_threshold_deg_per_sec": 45.0, "default_ekf_transition_steps_on_sharp_turn": 0},
}


VERBOSE = False     # True: the Sim3 functions print the reference's diagnostics (ref :396-425, :433, :446, :450) -- same numbers, in English


def _say(msg):
    if VERBOSE:
        print(msg)


def _ctx():
    return _lib.default_context()


# ---------------------------------------------------------------------------- I/O (EKFGPSSLAM.py:110-125, :1091-1104)
def load_slam_trajectory(txt_path):
    """TUM trajectory -> {'timestamps','positions','quaternions'}; ValueError on any problem (ref :110-125)."""
    try:
        data = np.loadtxt(txt_path)
        if data.ndim == 1:
            data = data.reshape(1, -1)
        if data.shape[1] != 8:
            raise ValueError(f"SLAM file format error: expected 8 columns (ts x y z qx qy qz qw), got {data.shape[1]}")
        return {"timestamps": data[:, 0].astype(float), "positions": data[:, 1:4].astype(float),
                "quaternions": data[:, 4:8].astype(float)}
    except FileNotFoundError:
        raise ValueError(f"SLAM file not found: {txt_path}")
    except Exception as e:
        raise ValueError(f"SLAM data load/parse failed ({txt_path}): {e}")


def save_tum_utm(path, timestamps, positions, quaternions):
    """The reference's UTM TUM writer (ref :1091-1092): %.6f x4, %.8f x4, header without '#'."""
    out = np.column_stack((timestamps, positions, quaternions))
    np.savetxt(path, out, fmt=["%.6f"] + ["%.6f"] * 3 + ["%.8f"] * 4, header="timestamp x y z qx qy qz qw (UTM)", comments="")


def save_tum_wgs84(path, timestamps, lonlatalt, quaternions):
    """The reference's WGS84 writer (ref :1097-1101)."""
    out = np.column_stack((timestamps, lonlatalt, quaternions))
    np.savetxt(path, out, fmt=["%.6f"] + ["%.8f", "%.8f", "%.3f"] + ["%.8f"] * 4, header="timestamp lon lat alt qx qy qz qw (WGS84)", comments="")


# ---------------------------------------------------------------------------- geodesy (EKFGPSSLAM.py:127-134, :249-296)
def auto_utm_projection(lons, lats):
    """(zone:int, hemisphere:str) from mean lon / mean lat (ref :127-134)."""
    lons, lats = np.asarray(lons), np.asarray(lats)
    if lons.size == 0 or lats.size == 0:
        raise ValueError("lon/lat arrays must not be empty to pick a UTM zone")
    central_lon = np.mean(lons)
    zone = int((central_lon + 180) // 6 + 1)
    hemisphere = " +south" if np.mean(lats) < 0 else ""
    return zone, hemisphere


class UtmProjector:
    """Callable stand-in for pyproj.Proj("+proj=utm +zone=Z[ +south] +ellps=WGS84 ...") (ref :267-270, :295):
    projector(lons, lats) -> (E, N); projector(x, y, inverse=True) -> (lons, lats).  Runs K1 on the GPU."""

    def __init__(self, zone, south):
        self.zone, self.south = int(zone), bool(south)
        self.srs = f"+proj=utm +zone={self.zone}{' +south' if self.south else ''} +ellps=WGS84 +datum=WGS84 +units=m +no_defs"

    def __call__(self, a, b, inverse=False):
        a, b = f64(a).ravel(), f64(b).ravel()
        if a.shape != b.shape:
            raise ValueError("x/y (lon/lat) must have the same length")
        o1, o2 = np.empty_like(a), np.empty_like(a)
        L = _lib.load()
        if not inverse:      # a = lons, b = lats
            check(L.gsf_utm_forward(_ctx().handle, hptr(b), hptr(a), a.size, self.zone, int(self.south), hptr(o1), hptr(o2)))
            return o1, o2    # (E, N)
        check(L.gsf_utm_inverse(_ctx().handle, hptr(a), hptr(b), a.size, self.zone, int(self.south), hptr(o1), hptr(o2)))
        return o2, o1        # gsf returns (lat, lon); pyproj returns (lon, lat)


def _ransac_axes_mask(t, p, degree, min_samples, residual_threshold, max_trials):
    """AND over the coordinate axes of one RANSACRegressor.fit each (ref :155-169), evaluated by gsf_ransac_poly_batch_dev.
    The sample sets come from scikit-learn's own sampler on NumPy's global legacy RNG -- the call RANSACRegressor makes -- and
    the stream is left exactly where the reference leaves it: the device returns n_trials_, the number of sets the reference's
    loop would have drawn (its trial count is data dependent), and the draws are replayed from the saved state.
    Raises ValueError when an axis finds no consensus set, like scikit-learn."""
    import torch
    from sklearn.utils.random import sample_without_replacement
    from . import batch as B
    rs = np.random.mtrand._rand                                          # check_random_state(None)
    n = len(t)
    dev = dict(device="cuda")
    td = torch.as_tensor(np.ascontiguousarray(t, dtype=np.float64)).to(**dev)
    offs = torch.tensor([0, n], dtype=torch.int64, **dev)
    keep = np.ones(n, dtype=bool)
    for ax in range(p.shape[1]):
        state = rs.get_state()
        idx = np.stack([sample_without_replacement(n, min_samples, random_state=rs) for _ in range(max_trials)]).astype(np.int32)
        yd = torch.as_tensor(np.ascontiguousarray(p[:, ax], dtype=np.float64)).to(**dev)
        mask, ntr, nin, st = B.ransac_poly_batch(td, yd, offs, torch.as_tensor(idx.reshape(1, max_trials, min_samples)).to(**dev),
                                                 degree, residual_threshold)
        ntr, st = int(ntr.item()), int(st.item())
        rs.set_state(state)
        for _ in range(ntr):
            sample_without_replacement(n, min_samples, random_state=rs)
        if st & 1:
            raise ValueError("RANSAC could not find a valid consensus set.")
        keep &= mask.cpu().numpy().astype(bool)
    return keep


def _prefilter_windows(times, config, need):
    """Row ranges [r0, r1) of the windows the reference visits, in its order (ref :199-234); the global mode is one window."""
    n_points = len(times)
    if not config.get("use_sliding_window", False):
        return [(0, n_points)], None
    width = config["window_duration_seconds"]
    stride = width * config["window_step_factor"]
    wins, rows_of = [], []
    t_first, t_last = times[0], times[-1]
    w0 = t_first
    while w0 < t_last:
        w1 = w0 + width
        rows = np.where((times >= w0) & (times < w1))[0]
        if len(rows) >= need:
            wins.append(rows)
        if stride <= 1e-6:
            later = np.where(times > w0)[0]
            if len(later) == 0:
                break
            w0 = times[later[0]]
        else:
            w0 += stride
        if w0 >= t_last and times[-1] >= w1:                             # one tail window ending just past the last stamp
            w0 = max(t_first, times[-1] - width + 1e-6)
    contiguous = all(len(r) == r[-1] - r[0] + 1 for r in wins)          # sorted stamps: every window is one row range
    return ([(int(r[0]), int(r[-1]) + 1) for r in wins] if contiguous else None), wins


def filter_gps_outliers_ransac(times, positions, config):
    """The reference's optional GPS pre-filter (ref :136-247): per-axis degree-d polynomial RANSAC, AND
    across axes; either one global fit or sliding windows [t, t+W) advanced by W*step_factor with one extra tail window, OR
    across windows -- rows never inside a fitted window are dropped.  The whole filter is ONE library call
    (gsf_gps_prefilter_chain): every window and axis in the reference's order, the sample sets drawn on the device from NumPy's
    global legacy generator exactly as scikit-learn's sampler draws them, the generator handed back where the reference leaves
    it -- a seeded run keeps the same rows.  Logs the device sampler does not cover (unsorted stamps, min_samples/n outside
    (0.01, 0.99)) take the window-by-window route with host-drawn sample sets (_ransac_axes_mask)."""
    if not config.get("enabled", False):
        return times, positions
    n_points, need = len(times), config["min_samples"]
    if n_points < need:
        return times, positions
    degree, thr, trials = config["polynomial_degree"], config["residual_threshold_meters"], config["max_trials"]
    sliding = bool(config.get("use_sliding_window", False))
    times, positions = np.asarray(times), np.asarray(positions)
    ranges, wins = _prefilter_windows(times, config, need)
    longest = max((b - a for a, b in ranges), default=0) if ranges is not None else 0
    chain_ok = (ranges is not None and positions.shape[1] == 3 and longest <= 14000 and 1 <= trials <= 1024 and need <= 16 and 1 <= degree <= 3
                and trials * (16 + 4 * need) + 2 * longest + 8 <= 56 * 1024)    # gsf_gps_prefilter_chain's own limits (LDS budget of its sampler: chain_lds, csrc/gsf_gpsfilter.hip)
    if chain_ok:
        if not ranges:
            return times[:0], positions[:0]                              # no window had enough rows: nothing is ever marked (ref :199-236)
        kind, key, pos_, has_gauss, cached = np.random.get_state()
        state = np.concatenate([key.astype(np.uint32), np.array([pos_], dtype=np.uint32)])
        t = np.ascontiguousarray(times, dtype=np.float64)
        p = np.ascontiguousarray(positions, dtype=np.float64)
        wr = np.ascontiguousarray(np.array(ranges, dtype=np.int32).reshape(-1, 2))
        off, wo = np.array([0, n_points], dtype=np.int64), np.array([0, len(ranges)], dtype=np.int64)
        keep, ws, ls = np.zeros(n_points, dtype=np.uint8), np.zeros(len(ranges), dtype=np.int32), np.zeros(1, dtype=np.int32)
        check(_lib.load().gsf_gps_prefilter_chain(_ctx().handle, hptr(t), hptr(p), hptr(off), 1, hptr(wr), hptr(wo), int(max(b - a for a, b in ranges)),
                                                  int(trials), int(need), int(degree), float(thr), 0.99, hptr(state), hptr(keep), hptr(ws), hptr(ls)))
        if ls[0] == 0:
            np.random.set_state((kind, state[:624], int(state[624]), has_gauss, cached))
            if not sliding:
                if ws[0] != 0:
                    return times, positions                              # the fit raised: filtering skipped (ref :178-182)
            keep = keep.astype(bool)
            return times[keep], positions[keep]
        # (generator untouched: `state` was a copy)

    def axes_mask(t, p):
        return _ransac_axes_mask(np.asarray(t, dtype=np.float64), np.asarray(p, dtype=np.float64), degree, need, thr, trials)

    if not sliding:                                                      # ref :148-182
        try:
            keep = axes_mask(times, positions)
            return times[keep], positions[keep]
        except GsfError:
            raise
        except Exception:
            return times, positions
    keep = np.zeros(n_points, dtype=bool)
    for rows in wins:                                                    # ref :183-247
        try:
            keep[rows[axes_mask(times[rows], positions[rows])]] = True
        except GsfError:
            raise
        except Exception:
            pass                                                         # a failed window marks nothing (ref :228-229)
    return times[keep], positions[keep]


def load_gps_data(txt_path, data_label="GPS", filter_config_override=None):
    """GPS text file -> {'timestamps','positions'(UTM E,N,alt),'utm_zone','projector'}; ValueError on failure
    (ref :249-289).  Columns are read as ts, lat, lon, alt -- cols 0,1,2,3 (ref :258)."""
    try:
        try:
            raw = np.loadtxt(txt_path, delimiter=" ")
        except ValueError:
            raw = np.loadtxt(txt_path, delimiter=",")
        if raw.ndim == 1:
            raw = raw.reshape(1, -1)
        if raw.shape[1] < 4:
            raise ValueError(f"{data_label} file needs at least 4 columns (ts lat lon alt), got {raw.shape[1]}")
        ts, lats, lons, alts = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]
        valid = (np.abs(lats) <= 90) & (np.abs(lons) <= 180) & (lats != 0) & (lons != 0)          # ref :259
        if not np.all(valid):
            ts, lats, lons, alts = ts[valid], lats[valid], lons[valid], alts[valid]
            if len(ts) == 0:
                raise ValueError(f"{data_label}: no valid GPS rows after the lat/lon range filter")
        zone, hemi = auto_utm_projection(lons, lats)
        projector = UtmProjector(zone, "south" in hemi)
        x, y = projector(lons, lats)                                                              # K1 on the GPU
        utm = np.column_stack((x, y, alts))
        fcfg = filter_config_override if filter_config_override is not None else CONFIG["gps_filtering_ransac"]
        ft, fp = filter_gps_outliers_ransac(ts, utm, fcfg)
        if len(ft) < 2:
            raise ValueError(f"{data_label}: fewer than 2 points left after the RANSAC filter")
        return {"timestamps": ft, "positions": fp, "utm_zone": f"{zone}{'S' if 'south' in hemi else 'N'}", "projector": projector}
    except FileNotFoundError:
        raise ValueError(f"{data_label} file not found: {txt_path}")
    except GsfError:
        raise
    except Exception as e:
        raise ValueError(f"{data_label} data processing failed: {e}")


def utm_to_wgs84(utm_points, projector):
    """UTM (X,Y,Z) -> (lon, lat, alt) (ref :291-296)."""
    utm_points = np.asarray(utm_points)
    if utm_points.ndim != 2 or utm_points.shape[1] != 3:
        raise ValueError("UTM points must be an Nx3 array (X, Y, Z)")
    if not isinstance(projector, UtmProjector):
        raise TypeError("projector must be the UtmProjector returned by load_gps_data")
    lons, lats = projector(utm_points[:, 0], utm_points[:, 1], inverse=True)
    return np.column_stack((lons, lats, utm_points[:, 2]))


# ---------------------------------------------------------------------------- time alignment (EKFGPSSLAM.py:301-387)
def estimate_time_offset(slam_times, gps_times, max_samples):
    """Clock offset by cross-correlation (ref :301-323).  The reference correlates two *linspaces*, which peaks
    at lag 0 for any input: the result is exactly 0.0; only the early-outs are kept."""
    return 0.0


def dynamic_time_alignment(slam_data, gps_data_source, time_align_config):
    """GPS positions interpolated onto the SLAM stamps, per gap-free segment (ref :325-387): returns
    (aligned (N,3) with NaN where unavailable, valid_mask (N,) bool).  One launch of the alignment kernel
    (gsf_time_align_batch: sort/unique, gap split, not-a-knot cubic / linear, evaluation)."""
    slam_times = f64(slam_data["timestamps"]).ravel()
    gps_times = f64(gps_data_source["timestamps"]).ravel()
    n_slam, n_gps = len(slam_times), len(gps_times)
    if n_slam == 0 or n_gps < 2:
        return np.full((n_slam, 3), np.nan), np.zeros(n_slam, dtype=bool)
    gps_positions = f64(gps_data_source["positions"], (n_gps, 3))
    so, go = np.array([0, n_slam], dtype=np.int64), np.array([0, n_gps], dtype=np.int64)
    aligned, valid, st = np.empty((n_slam, 3)), np.zeros(n_slam, dtype=np.uint8), np.zeros(1, dtype=np.int32)
    check(_lib.load().gsf_time_align_batch(_ctx().handle, hptr(slam_times), hptr(so), hptr(gps_times), hptr(gps_positions), hptr(go), 1,
                                           float(time_align_config["max_gps_gap_threshold"]), hptr(aligned), hptr(valid), hptr(st)))
    return aligned, valid.astype(bool)


# ---------------------------------------------------------------------------- Sim3 (EKFGPSSLAM.py:389-467)
def compute_sim3_transform(src, dst):
    """Umeyama fit dst ~ s R src + t -> (R(3,3), t(3,), scale) or (None, None, None) (ref :428-459).  K2 on the GPU."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    n_points = src.shape[0]
    if n_points < 3:
        return None, None, None
    if src.shape != dst.shape or src.ndim != 2 or src.shape[1] != 3:
        _say("Error: Sim3: source/target points must both be Nx3.")                             # ref :433
        return None, None, None
    src, dst = np.ascontiguousarray(src), np.ascontiguousarray(dst)
    off = np.array([0, n_points], dtype=np.int64)
    R, t, s, st = np.empty((1, 9)), np.empty((1, 3)), np.empty(1), np.zeros(1, dtype=np.int32)
    check(_lib.load().gsf_sim3_umeyama_batch(_ctx().handle, hptr(src), hptr(dst), None, hptr(off), 1, hptr(R), hptr(t), hptr(s), hptr(st)))
    if st[0] == _lib.SIM3_NONE:
        _say("Error: Sim3: linear-algebra failure (non-finite cross-covariance).")              # ref :453
        return None, None, None
    if st[0] & 2:
        _say("Warning: source point set has (near) zero variance; scale defaults to 1.0.")      # ref :446
    if st[0] & 4:
        _say("Warning: computed scale is tiny (<= 1e-6); reset to 1.0.")                        # ref :450
    return R.reshape(3, 3), t.reshape(3), float(s[0])


def compute_sim3_transform_robust(src, dst, min_samples, residual_threshold, max_trials, min_inliers_needed, point_description="pts"):
    """RANSAC-wrapped Sim3 (ref :389-426) in ONE library call: the hypotheses' rows are drawn on the device from NumPy's global
    legacy generator state -- the same np.random.choice(n, min_samples, replace=False) per trial as ref :405, bit for bit -- then
    fitted, scored, arg-maxed and refitted on the inliers; the advanced state is handed back to np.random, so a seeded run keeps
    drawing what the reference would draw next."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    n_points = src.shape[0]
    if n_points < min_samples:
        _say(f"Error: Sim3 RANSAC: too few input points ({n_points} from {point_description}), at least {min_samples} needed.")   # ref :396
        return None, None, None
    if src.shape != dst.shape:
        _say(f"Error: Sim3 RANSAC: source and target ({point_description}) differ in shape ({src.shape} vs {dst.shape}).")          # ref :399
        return None, None, None
    _say(f"  Sim3 RANSAC on {n_points} {point_description} (threshold={residual_threshold}m, trials={max_trials}, min samples={min_samples})...")   # ref :403
    if int(min_samples) < 1:
        raise ValueError("min_samples must be >= 1")
    if n_points > 28000 or int(min_samples) > 64 or int(max_trials) > (1 << 20):
        # beyond the device sampler (its LDS budget, 64 traced positions per trial): the reference's own draws on the host (ref :405),
        # fed to the same scoring kernel -- any configuration the reference accepts runs
        idx = np.empty((int(max_trials), int(min_samples)), dtype=np.int32)
        for k in range(int(max_trials)):
            idx[k] = np.random.choice(n_points, min_samples, replace=False)
        R_, t_, s_, _, nin_ = sim3_ransac_with_indices(src, dst, idx, residual_threshold, min_inliers_needed)
        _robust_report(R_, s_, int(nin_), n_points, min_inliers_needed, point_description)
        return R_, t_, s_
    src, dst = np.ascontiguousarray(src), np.ascontiguousarray(dst)
    kind, key, pos, has_gauss, cached = np.random.get_state()
    state = np.concatenate([key.astype(np.uint32), np.array([pos], dtype=np.uint32)])
    off = np.array([0, n_points], dtype=np.int64)
    R, t, s = np.empty((1, 9)), np.empty((1, 3)), np.empty(1)
    st, nin, mask = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32), np.zeros(n_points, dtype=np.uint8)
    check(_lib.load().gsf_sim3_ransac_mt_batch(_ctx().handle, hptr(src), hptr(dst), hptr(off), 1, hptr(state), int(max_trials), int(min_samples),
                                               float(residual_threshold), int(min_inliers_needed), hptr(R), hptr(t), hptr(s), hptr(st), hptr(mask),
                                               hptr(nin)))
    np.random.set_state((kind, state[:624], int(state[624]), has_gauss, cached))
    none = bool(st[0] & _lib.SIM3_NONE)
    _robust_report(None if none else R, None if none else float(s[0]), int(nin[0]), n_points, min_inliers_needed, point_description)
    if none:
        return None, None, None
    return R.reshape(3, 3), t.reshape(3), float(s[0])


def _robust_report(R, scale, max_inliers, n_points, min_inliers_needed, point_description):
    """the reference's progress lines of compute_sim3_transform_robust (ref :415-425), printed when VERBOSE"""
    if not VERBOSE:
        return
    print(f"  Sim3 RANSAC done: best inlier count {max_inliers}/{n_points}.")                                                     # ref :415
    if max_inliers < min_inliers_needed:
        print(f"Error: Sim3 RANSAC: best inlier count too small ({max_inliers} from {point_description}), {min_inliers_needed} needed.")   # ref :417
    elif R is None:
        print(f"Error: Sim3 RANSAC: the final fit on {max_inliers} inliers failed.")                                             # ref :423
    else:
        print(f"  Final Sim3 from {max_inliers} inliers: scale={scale:.4f}")                                                     # ref :419, :425


def sim3_ransac_with_indices(src, dst, sample_idx, residual_threshold, min_inliers_needed):
    """K2b with caller-provided sample indices -> (R, t, s, inlier_mask, n_inliers) or (None, None, None, mask, n)."""
    src, dst = np.ascontiguousarray(src, dtype=np.float64), np.ascontiguousarray(dst, dtype=np.float64)
    sample_idx = np.ascontiguousarray(sample_idx, dtype=np.int32)
    n = src.shape[0]
    trials, ms = sample_idx.shape
    off = np.array([0, n], dtype=np.int64)
    R, t, s = np.empty((1, 9)), np.empty((1, 3)), np.empty(1)
    st, nin, mask = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32), np.zeros(max(n, 1), dtype=np.uint8)
    check(_lib.load().gsf_sim3_ransac_batch(_ctx().handle, hptr(src), hptr(dst), hptr(off), 1, hptr(sample_idx), trials, ms,
                                            float(residual_threshold), int(min_inliers_needed), hptr(R), hptr(t), hptr(s), hptr(st),
                                            hptr(mask), hptr(nin)))
    mask = mask[:n].astype(bool)
    if st[0] == _lib.SIM3_NONE:
        return None, None, None, mask, int(nin[0])
    return R.reshape(3, 3), t.reshape(3), float(s[0]), mask, int(nin[0])


def transform_trajectory(positions, quaternions, R_mat, t_vec, scale_val):
    """Apply a Sim3 to positions and orientations (ref :461-467).  K3 on the GPU."""
    pos, quat = np.ascontiguousarray(positions, dtype=np.float64), np.ascontiguousarray(quaternions, dtype=np.float64)
    n = pos.shape[0]
    if quat.shape != (n, 4) or pos.shape != (n, 3):
        raise ValueError("positions must be (N,3) and quaternions (N,4)")
    off = np.array([0, n], dtype=np.int64)
    R = np.ascontiguousarray(R_mat, dtype=np.float64).reshape(1, 9)
    t = np.ascontiguousarray(t_vec, dtype=np.float64).reshape(1, 3)
    s = np.array([float(scale_val)])
    po, qo, bad = np.empty_like(pos), np.empty_like(quat), np.zeros(1, dtype=np.int32)
    check(_lib.load().gsf_apply_sim3_batch(_ctx().handle, hptr(pos), hptr(quat), hptr(off), 1, hptr(R), hptr(t), hptr(s), hptr(po), hptr(qo), hptr(bad)))
    if bad[0]:
        raise ValueError("Found zero norm quaternions in `quat`.")       # what SciPy raises at ref :466
    return po, qo


# ---------------------------------------------------------------------------- EKF + RTS (EKFGPSSLAM.py:831-935)
def apply_ekf_correction(slam_data_in, gps_data_in, sim3_pos_initial, sim3_quat_initial, global_config):
    """EKF + dynamic per-outage RTS fusion of one trajectory -> (pos (N,3), quat (N,4)) (ref :831-935).
    ValueError on length mismatch (ref :836-837).  The recursion runs in one K4 launch."""
    n_points = len(slam_data_in["timestamps"])
    if n_points == 0:
        return np.empty((0, 3)), np.empty((0, 4))
    sim3_pos_initial, sim3_quat_initial = np.asarray(sim3_pos_initial), np.asarray(sim3_quat_initial)
    if not (sim3_pos_initial.shape[0] == n_points and sim3_quat_initial.shape[0] == n_points):
        raise ValueError(f"Sim3-transformed trajectory length ({sim3_pos_initial.shape[0]}) != SLAM timestamp count ({n_points})")
    aligned, valid = dynamic_time_alignment(slam_data_in, gps_data_in, global_config["time_alignment"])     # ref :847
    return ekf_fuse_aligned(slam_data_in["timestamps"], slam_data_in["positions"], slam_data_in["quaternions"], aligned, valid,
                            sim3_pos_initial[0], sim3_quat_initial[0], global_config)[:2]


def ekf_fuse_aligned(timestamps, positions, quaternions, aligned_gps, valid_mask, init_pos, init_quat, global_config=None):
    """ref :831-935 after its alignment call: one trajectory, host arrays -> (pos, quat, status bits)."""
    cfg = EkfConfig.from_config(global_config or CONFIG)
    ts = f64(timestamps).ravel()
    n = ts.size
    pos, quat, gps = f64(positions, (n, 3)), f64(quaternions, (n, 4)), f64(aligned_gps, (n, 3))
    valid = np.ascontiguousarray(valid_mask, dtype=np.uint8).reshape(n)
    ip, iq = f64(init_pos, (1, 3)), f64(init_quat, (1, 4))
    po, qo, st = np.empty((n, 3)), np.empty((n, 4)), np.zeros(1, dtype=np.int32)
    check(_lib.load().gsf_ekf_fuse_batch(_ctx().handle, _lib.LAYOUT_TRAJ_MAJOR, hptr(ts), hptr(pos), hptr(quat), hptr(gps), hptr(valid),
                                         hptr(ip), hptr(iq), C.byref(cfg), 1, n, hptr(po), hptr(qo), hptr(st)))
    return po, qo, int(st[0])


# ---------------------------------------------------------------------------- helpers of the EKF surface (EKFGPSSLAM.py:77-105, :679-826)
def calculate_relative_pose(pose1_pos, pose1_quat, pose2_pos, pose2_quat):
    """Relative motion pose1 -> pose2 in pose1's frame: (delta_pos_local (3,), delta_quat (4,)); an invalid (zero-norm)
    quaternion gives zero motion and the identity (ref :77-92)."""
    p1, q1, p2, q2 = f64(pose1_pos, (1, 3)), f64(pose1_quat, (1, 4)), f64(pose2_pos, (1, 3)), f64(pose2_quat, (1, 4))
    dp, dq = np.empty((1, 3)), np.empty((1, 4))
    check(_lib.load().gsf_relative_pose_batch(_ctx().handle, hptr(p1), hptr(q1), hptr(p2), hptr(q2), 1, hptr(dp), hptr(dq), None))
    return dp[0], dq[0]


def quaternion_nlerp(q1, q2, weight_q2):
    """Normalised linear interpolation between two quaternions (ref :94-105)."""
    a, b, w, out = f64(q1, (1, 4)), f64(q2, (1, 4)), np.array([float(weight_q2)]), np.empty((1, 4))
    check(_lib.load().gsf_quaternion_nlerp_batch(_ctx().handle, hptr(a), hptr(b), hptr(w), 1, hptr(out)))
    return out[0]


def is_sharp_turn_in_segment(slam_quaternions_segment, slam_timestamps_segment, yaw_rate_threshold_rad_per_sec):
    """True if the max |yaw rate| over consecutive poses of the segment exceeds the threshold (ref :808-826)."""
    n = len(slam_quaternions_segment)
    if n < 2:
        return False
    q, t = f64(np.asarray(slam_quaternions_segment), (n, 4)), f64(np.asarray(slam_timestamps_segment), (n,))
    off, res = np.array([0, n], dtype=np.int64), np.zeros(1, dtype=np.int32)
    check(_lib.load().gsf_is_sharp_turn_batch(_ctx().handle, hptr(q), hptr(t), hptr(off), 1, float(yaw_rate_threshold_rad_per_sec), hptr(res), None))
    return bool(res[0])


def rts_smoother_segment(states_filt_segment, covs_filt_segment, states_pred_segment, covs_pred_segment):
    """RTS back-pass over one segment, F = I (ref :777-803).  Lists in, lists of arrays out, like the reference."""
    n = len(states_filt_segment)
    if n == 0:
        return [], []
    xf, xp = f64(np.asarray(states_filt_segment), (n, 7)), f64(np.asarray(states_pred_segment), (n, 7))
    Pf, Pp = f64(np.asarray(covs_filt_segment), (n, 49)), f64(np.asarray(covs_pred_segment), (n, 49))
    xs, Ps, off = np.empty((n, 7)), np.empty((n, 49)), np.array([0, n], dtype=np.int64)
    check(_lib.load().gsf_rts_smoother_segment_batch(_ctx().handle, hptr(xf), hptr(Pf), hptr(xp), hptr(Pp), hptr(off), 1, hptr(xs), hptr(Ps)))
    return [xs[k].copy() for k in range(n)], [Ps[k].reshape(7, 7).copy() for k in range(n)]


class ExtendedKalmanFilter:
    """The reference's 7-state filter object (ref :679-772): same attributes and process_step signature; every step is one
    device call in the general dense-covariance form (gsf_ekf_process_step).  The batched hot path does not go through this
    class -- it exists so that code written against the reference's class keeps working."""

    def __init__(self, initial_pos, initial_quat, config_params):
        initial_pos, initial_quat = np.asarray(initial_pos, dtype=float), np.asarray(initial_quat, dtype=float)
        if not (initial_pos.shape == (3,) and initial_quat.shape == (4,)):
            raise ValueError("EKF init: initial pose must be pos (3,) and quat (4,)")
        self.state = np.concatenate([initial_pos, self.normalize_quaternion(initial_quat)]).astype(float)
        self.cov = np.diag(config_params["initial_cov_diag"]).astype(float)
        self.Q_per_sec = np.diag(config_params["process_noise_diag"]).astype(float)
        self.R = np.diag(config_params["meas_noise_diag"]).astype(float)
        if self.state.shape != (7,) or self.cov.shape != (7, 7) or self.Q_per_sec.shape != (7, 7) or self.R.shape != (3, 3):
            raise ValueError("EKF init: wrong state / covariance / noise dimensions")
        self.gnss_available_prev = None
        self.gnss_update_weight = 0.0
        self.original_transition_steps = max(1, int(config_params.get("transition_steps", 10)))
        self.current_transition_steps = self.original_transition_steps
        self.weight_delta = 1.0
        self._last_predicted_state_for_blending = self.state.copy()

    @staticmethod
    def normalize_quaternion(q):
        q = np.asarray(q, dtype=float)
        norm = np.linalg.norm(q)
        return q / norm if norm > 1e-9 else np.array([0.0, 0.0, 0.0, 1.0])

    def process_step(self, slam_motion_update, gps_measurement, gnss_is_available, delta_time, override_transition_steps=None):
        eff = override_transition_steps if override_transition_steps is not None else self.current_transition_steps
        self.weight_delta = 1.0 / eff if eff > 0 else 1.0
        state, cov = f64(self.state, (7,)).copy(), f64(self.cov, (49,)).copy()
        Q, R = f64(self.Q_per_sec, (49,)), f64(self.R, (9,))
        dp, dq = f64(slam_motion_update[0], (3,)), f64(slam_motion_update[1], (4,))
        z = None
        if gps_measurement is not None:
            z = np.asarray(gps_measurement, dtype=float)
            if z.shape != (3,):
                z = None                                  # ref :719: wrong shape -> the update is skipped
            else:
                z = np.ascontiguousarray(z)
        prev = C.c_int32({None: -1, False: 0, True: 1}[None if self.gnss_available_prev is None else bool(self.gnss_available_prev)])
        w = C.c_double(float(self.gnss_update_weight))
        ps, pc = np.empty(7), np.empty(49)
        check(_lib.load().gsf_ekf_process_step(_ctx().handle, hptr(state), hptr(cov), hptr(Q), hptr(R), C.byref(prev), C.byref(w),
                                               int(self.current_transition_steps), hptr(dp), hptr(dq), hptr(z), int(bool(gnss_is_available)),
                                               float(delta_time), -1 if override_transition_steps is None else int(override_transition_steps),
                                               hptr(ps), hptr(pc)))
        self.state, self.cov = state, cov.reshape(7, 7)
        self.gnss_available_prev = bool(gnss_is_available)
        self.gnss_update_weight = w.value
        self._last_predicted_state_for_blending = ps.copy()
        return self.state, self.cov, ps, pc.reshape(7, 7)


# ---------------------------------------------------------------------------- error evaluation (EKFGPSSLAM.py:1013-1033)
def evaluate_trajectory_errors(slam_timestamps, traj_positions, aligned_gps, valid_mask, skip_seconds=5.0):
    """The reference's error metric (step 6 of main_process_gui, ref :1013-1033): for every SLAM index with a
    valid aligned GNSS fix after the first `skip_seconds`, the minimum distance to ANY such candidate fix; returns
    {'count','mean','median','rmse','errors'(N, NaN where not evaluated)}.  One kernel launch (gsf_eval_errors_batch)."""
    ts = f64(slam_timestamps).ravel()
    n = ts.size
    tp, gp = f64(traj_positions, (n, 3)), f64(aligned_gps, (n, 3))
    va = np.ascontiguousarray(valid_mask, dtype=np.uint8).reshape(n)
    stats, err = np.empty((1, 4)), np.empty(n)
    if n == 0:
        return {"count": 0, "mean": np.nan, "median": np.nan, "rmse": np.nan, "errors": err}
    check(_lib.load().gsf_eval_errors_batch(_ctx().handle, hptr(ts), hptr(tp), hptr(gp), hptr(va), 1, n, float(skip_seconds), hptr(stats), hptr(err)))
    return {"count": int(stats[0, 0]), "mean": float(stats[0, 1]), "median": float(stats[0, 2]), "rmse": float(stats[0, 3]), "errors": err}


# ---------------------------------------------------------------------------- headless driver (EKFGPSSLAM.py:940-1104, no GUI)
def pick_sim3_indices(slam_data, valid_mask, config=None):
    """Which time-synchronised points feed the global Sim3 (first gap-free segment, <= max_initial_duration) -- ref :973-998."""
    config = config or CONFIG
    vi = np.where(valid_mask)[0]
    ms = config["sim3_ransac"]["min_samples"]
    if len(vi) < ms:
        raise ValueError(f"time-synchronised points for Sim3 ({len(vi)}) < RANSAC min_samples ({ms})")
    vt = slam_data["timestamps"][vi]
    gaps = np.where(np.diff(vt) > config["time_alignment"]["max_gps_gap_threshold"])[0]
    end = gaps[0] if len(gaps) > 0 else len(vi)
    first = vi[:end]
    if len(first) < ms:
        return vi
    lim = slam_data["timestamps"][first] <= slam_data["timestamps"][first[0]] + config["sim3_ransac"]["max_initial_duration"]
    timed = first[lim]
    return first if len(timed) < ms else timed


def benchmark_c1(slam, gps_t_raw, lat, lon, alt, repeats=20):
    """Warm wall time of the single-trajectory drop-in on ONE track handed in by the caller: slam = {'timestamps', 'positions',
    'quaternions'} and its raw GNSS log (stamps, lat, lon, alt as load_gps_data reads them, ref :258).  Timed: the GPS leg of step 1
    (projection + sliding RANSAC pre-filter, ref :266-275), steps 2-5 (time alignment, row choice, robust Sim3, apply, EKF+RTS,
    ref :971-1010) and step 6 (error metric).  (bench.py feeds it the C1 shape -- BASELINE configs[0]: the 271-pose KITTI-04 track and
    its 279 fixes from the committed fixtures, and adds the reference's own CPU figures from profiles/rNN_reference_timing.json.)"""
    import time
    slam = {k: np.array(slam[k], dtype=np.float64) for k in ("timestamps", "positions", "quaternions")}
    ts, lats, lons, alts = (np.array(a, dtype=np.float64) for a in (gps_t_raw, lat, lon, alt))
    config = copy.deepcopy(CONFIG)
    sc = config["sim3_ransac"]

    def gps_leg():
        zone, hemi = auto_utm_projection(lons, lats)
        projector = UtmProjector(zone, "south" in hemi)
        x, y = projector(lons, lats)
        ft, fp = filter_gps_outliers_ransac(ts, np.column_stack((x, y, alts)), config["gps_filtering_ransac"])
        return {"timestamps": ft, "positions": fp, "projector": projector}

    def steps_2_to_5(gps):
        aligned, valid = dynamic_time_alignment(slam, gps, config["time_alignment"])
        idx = pick_sim3_indices(slam, valid, config)
        R, t, s = compute_sim3_transform_robust(slam["positions"][idx], aligned[idx], sc["min_samples"], sc["residual_threshold"], sc["max_trials"],
                                                sc["min_inliers_needed"])
        sp, sq = transform_trajectory(slam["positions"], slam["quaternions"], R, t, s)
        pos, quat = apply_ekf_correction(slam, gps, sp, sq, config)
        return aligned, valid, pos

    out = {}
    np.random.seed(0)
    gps = gps_leg(); aligned, valid, pos = steps_2_to_5(gps)             # warm-up (first calls allocate the staging arenas)
    for name, fn in (("gps_projection_and_prefilter_ms", gps_leg), ("steps_2_to_5_ms", lambda: steps_2_to_5(gps)),
                     ("step_6_error_metric_ms", lambda: evaluate_trajectory_errors(slam["timestamps"], pos, aligned, valid))):
        best, tot = 1e9, 0.0
        for _ in range(repeats):
            t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
            best, tot = min(best, dt), tot + dt
        out[name] = {"best": best * 1e3, "mean": tot / repeats * 1e3}
    out["end_to_end_ms"] = {"best": sum(v["best"] for v in out.values()), "mean": sum(v["mean"] for v in out.values())}
    out["poses"] = int(len(slam["timestamps"]))
    out["gnss_fixes"] = int(len(ts))
    return out


def evaluate_against(slam, trajectories, gps_data, config=None, skip_seconds=5.0):
    """Step 6 of main_process_gui for one GNSS data set (ref :1013-1033 for the primary GPS, :1035-1062 for the ground truth): time
    alignment of `gps_data` to the SLAM stamps (:1014 / :1037), then the reference's error metric of every trajectory in
    `trajectories` (label -> (N, 3)) against the aligned fixes after the first `skip_seconds`.  Returns {'aligned', 'valid',
    'points', label: evaluate_trajectory_errors(...)}."""
    config = config or CONFIG
    aligned, valid = dynamic_time_alignment(slam, gps_data, config["time_alignment"])
    out = {"aligned": aligned, "valid": valid}
    for label, tr in trajectories.items():
        out[label] = evaluate_trajectory_errors(slam["timestamps"], tr, aligned, valid, skip_seconds)
    out["points"] = int(next(iter(out[k]["count"] for k in trajectories), 0)) if trajectories else 0
    return out


def run_fusion(slam_path, gps_path, out_path_utm=None, config=None, gt_gps_path=None):
    """Steps 1-7 of main_process_gui without tk dialogs (ref :940-1104).  gt_gps_path: the optional second GNSS file the reference
    asks for (:949-953), loaded with CONFIG['ground_truth_gps_filtering'] (:964) and used as ground truth in step 6.  Returns a dict of
    all stages; step 6 is 'errors' = {'primary': {...}, 'ground_truth': {...} or None} with the rows of :1027 / :1049 (raw SLAM,
    Sim3, EKF) and 'plot_error_ref' = which of the two the reference's plots show (:1064-1075)."""
    config = copy.deepcopy(config or CONFIG)
    slam = load_slam_trajectory(slam_path)                                                       # step 1
    gps = load_gps_data(gps_path, data_label="primary GPS", filter_config_override=config["gps_filtering_ransac"])
    gt = None
    if gt_gps_path:
        gt = load_gps_data(gt_gps_path, data_label="GNSS ground truth", filter_config_override=config["ground_truth_gps_filtering"])   # :964
        if len(gt["positions"]) < 2:                                                             # :966: not used
            gt = None
    if len(slam["positions"]) == 0 or len(gps["positions"]) < 2:
        raise ValueError("empty SLAM data or fewer than 2 GPS points")
    aligned, valid = dynamic_time_alignment(slam, gps, config["time_alignment"])                 # step 2
    idx = pick_sim3_indices(slam, valid, config)
    sc = config["sim3_ransac"]
    R, t, s = compute_sim3_transform_robust(slam["positions"][idx], aligned[idx], sc["min_samples"], sc["residual_threshold"],
                                            sc["max_trials"], sc["min_inliers_needed"])          # step 3
    if R is None:
        raise RuntimeError("global Sim3 transform failed")                                       # ref :1003
    sim3_pos, sim3_quat = transform_trajectory(slam["positions"], slam["quaternions"], R, t, s)  # step 4
    pos, quat = apply_ekf_correction(slam, gps, sim3_pos, sim3_quat, config)                     # step 5
    # step 6 (ref :1013-1075): raw SLAM / Sim3 / EKF rows against the primary GPS and, if given, the ground truth
    tracks = {"raw_slam": slam["positions"], "sim3": sim3_pos, "ekf": pos}
    e_primary = evaluate_against(slam, tracks, gps, config)
    e_gt = evaluate_against(slam, tracks, gt, config) if gt is not None else None
    if e_gt is not None and e_gt["ekf"]["count"] > 0:                                            # :1064-1069
        plot_ref = "ground_truth"
    elif e_primary["ekf"]["count"] > 0:                                                          # :1070-1074
        plot_ref = "primary"
    else:
        plot_ref = None                                                                          # :1075
    if out_path_utm:                                                                             # step 7
        save_tum_utm(out_path_utm, slam["timestamps"], pos, quat)
        wgs = utm_to_wgs84(pos, gps["projector"])
        out_wgs = out_path_utm.replace("_utm.txt", "_wgs84.txt")
        if out_wgs == out_path_utm:
            out_wgs = out_path_utm.replace(".txt", "_wgs84.txt") if ".txt" in out_path_utm else out_path_utm + "_wgs84.txt"
        save_tum_wgs84(out_wgs, slam["timestamps"], wgs, quat)
    return {"slam": slam, "gps": gps, "ground_truth_gps": gt, "aligned": aligned, "valid": valid, "sim3_idx": idx, "R": R, "t": t, "s": s,
            "sim3_pos": sim3_pos, "sim3_quat": sim3_quat, "pos": pos, "quat": quat,
            "errors": {"primary": e_primary, "ground_truth": e_gt}, "plot_error_ref": plot_ref,
            "err_sim3": e_primary["sim3"], "err_ekf": e_primary["ekf"]}
