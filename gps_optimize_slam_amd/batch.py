"""Batched device entry points: B independent trajectories resident in HBM as torch tensors.

torch is plumbing here (device memory, the current HIP stream, torch.distributed for the N>1
collect); every numerical stage is a libgsf.so kernel launched on torch's current stream.
Additions to the reference's surface (which is single-trajectory): `*_batch` functions taking
(B, N, .) trajectory-major or (N, ., B) time-major tensors -- SURVEY 8(b).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import LAYOUT_TIME_MAJOR, LAYOUT_TRAJ_MAJOR, EkfConfig, GsfError, check  # noqa: F401
from .ekfgpsslam import CONFIG

_ctxs = {}


def context(device=None):
    """gsf context bound to torch's CURRENT stream on `device` (cached per device/stream)."""
    if not torch.cuda.is_available():
        raise GsfError("torch sees no GPU: the batched fusion path needs an MI355X (no CPU fallback)")
    dev = torch.cuda.current_device() if device is None else torch.device(device).index or 0
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, stream)
    if key not in _ctxs:
        _ctxs[key] = _lib.Context(dev, stream)
    return _ctxs[key]


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype, shape, name):
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"{name}: expected contiguous cuda {dtype} tensor of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")


def shapes(layout, B, N):
    """tensor shapes of (ts, pos, quat, gps, valid) for a layout"""
    if layout == LAYOUT_TIME_MAJOR:
        return (N, B), (N, 3, B), (N, 4, B), (N, 3, B), (N, B)
    return (B, N), (B, N, 3), (B, N, 4), (B, N, 3), (B, N)


class TrajectoryBatch:
    """B trajectories x N poses on one GPU: original SLAM track + time-aligned GNSS (+ Sim3-aligned first pose)."""

    def __init__(self, layout, B, N, device="cuda"):
        self.layout, self.B, self.N = int(layout), int(B), int(N)
        s_ts, s_pos, s_quat, s_gps, s_val = shapes(layout, B, N)
        f = dict(dtype=torch.float64, device=device)
        self.ts, self.pos, self.quat, self.gps = torch.empty(s_ts, **f), torch.empty(s_pos, **f), torch.empty(s_quat, **f), torch.empty(s_gps, **f)
        self.valid = torch.empty(s_val, dtype=torch.uint8, device=device)
        self.init_pos, self.init_quat = torch.empty((B, 3), **f), torch.empty((B, 4), **f)

    @classmethod
    def synthetic(cls, B, N, layout=LAYOUT_TIME_MAJOR, seed=20250523, traj0=0, device="cuda", variant=0):
        """Deterministic KITTI-04-shaped batch generated on the device (gsf_synth_batch_dev).  variant 0: white 2 cm noise on
        the SLAM positions (the default workload); variant 1: SURVEY 8d to the letter -- a random-walk drift of 2 cm per pose and the
        sharp-turn burst on 5 % of the tracks."""
        b = cls(layout, B, N, device)
        ctx = context()
        previous = ctx.options.get("synth_variant", 0)
        ctx.set_option("synth_variant", int(variant))
        try:
            check(_lib.load().gsf_synth_batch_dev(ctx.handle, b.layout, C.c_uint64(seed), int(traj0), b.B, b.N, _p(b.ts), _p(b.pos),
                                                  _p(b.quat), _p(b.gps), _p(b.valid), _p(b.init_pos), _p(b.init_quat)))
        finally:
            ctx.set_option("synth_variant", previous)
        return b

    @classmethod
    def replicated(cls, ts, pos, quat, gps, valid, B, gnss_sigma=0.45, seed=0, layout=LAYOUT_TRAJ_MAJOR, device="cuda"):
        """B copies of ONE real track (ts (N,), pos (N,3), quat (N,4), time-aligned gps (N,3) with NaN where invalid, valid (N,)) with
        independent white GNSS noise of gnss_sigma metres per copy (copy 0 keeps the fixes as they are) -- e.g. the bundled KITTI-04
        track of config C1 as a batch.  init_pos / init_quat are pose 0 of the track (placeholders: the fused pipeline fits its own)."""
        import numpy as np
        ts, pos, quat, gps = (np.asarray(a, dtype=np.float64) for a in (ts, pos, quat, gps))
        N = ts.shape[0]
        rng = np.random.default_rng(seed)
        g = np.repeat(gps[None], B, axis=0)
        noise = rng.normal(scale=gnss_sigma, size=(B, N, 3)); noise[0] = 0.0
        g = g + noise
        rep = lambda a: np.repeat(np.asarray(a)[None], B, axis=0)
        return cls.from_host(rep(ts), rep(pos), rep(quat), g, rep(np.asarray(valid).astype(np.uint8)), rep(pos[0]), rep(quat[0]), layout=layout, device=device)

    @classmethod
    def from_host(cls, ts, pos, quat, gps, valid, init_pos, init_quat, layout=LAYOUT_TRAJ_MAJOR, device="cuda"):
        """(B,N,.) trajectory-major host arrays -> device batch in `layout`."""
        B, N = ts.shape
        tm = cls(LAYOUT_TRAJ_MAJOR, B, N, device)
        for name, arr in (("ts", ts), ("pos", pos), ("quat", quat), ("gps", gps)):
            getattr(tm, name).copy_(torch.as_tensor(arr, dtype=torch.float64).reshape(getattr(tm, name).shape))
        tm.valid.copy_(torch.as_tensor(valid).to(torch.uint8).reshape(B, N))
        tm.init_pos.copy_(torch.as_tensor(init_pos, dtype=torch.float64).reshape(B, 3))
        tm.init_quat.copy_(torch.as_tensor(init_quat, dtype=torch.float64).reshape(B, 4))
        return tm if layout == LAYOUT_TRAJ_MAJOR else tm.to_layout(layout)

    def to_layout(self, layout):
        if layout == self.layout:
            return self
        o = TrajectoryBatch(layout, self.B, self.N, self.ts.device)
        L, h = _lib.load(), context().handle
        fn = L.gsf_transpose_to_time_major_dev if layout == LAYOUT_TIME_MAJOR else L.gsf_transpose_to_traj_major_dev
        for name, Cc, eb in (("ts", 1, 8), ("pos", 3, 8), ("quat", 4, 8), ("gps", 3, 8), ("valid", 1, 1)):
            check(fn(h, _p(getattr(self, name)), _p(getattr(o, name)), self.B, self.N, Cc, eb))
        o.init_pos.copy_(self.init_pos); o.init_quat.copy_(self.init_quat)
        return o

    def host_traj_major(self):
        """-> dict of (B,N,.) numpy arrays (for the oracle / file output)"""
        t = self.to_layout(LAYOUT_TRAJ_MAJOR)
        torch.cuda.synchronize()
        return {k: getattr(t, k).cpu().numpy() for k in ("ts", "pos", "quat", "gps", "valid", "init_pos", "init_quat")}


class GeodeticBatch:
    """B trajectories x N poses whose GNSS side is still what the reference's loader reads (load_gps_data, EKFGPSSLAM.py:258):
    a ragged log of fixes with their own stamps, rows (lat deg, lon deg, alt m).  Input of fuse_from_geodetic()."""

    def __init__(self, B, N, ts, pos, quat, gps_offsets, gps_t, gps_llh, max_fixes):
        self.B, self.N, self.ts, self.pos, self.quat = int(B), int(N), ts, pos, quat
        self.gps_offsets, self.gps_t, self.gps_llh, self.max_fixes = gps_offsets, gps_t, gps_llh, int(max_fixes)
        self.slam_offsets = torch.arange(0, (self.B + 1) * self.N, self.N, dtype=torch.int64, device=ts.device)

    @classmethod
    def from_host(cls, ts, pos, quat, logs, device="cuda"):
        """B equal-length SLAM tracks (ts (B,N), pos (B,N,3), quat (B,N,4)) and their GNSS logs as the reference's loader reads them:
        logs = list of B arrays (n_b, >= 4) with columns stamp, col 1, col 2, col 3 of the text file (read as lat, lon, alt; ref :258)."""
        import numpy as np
        ts = np.asarray(ts, dtype=np.float64)
        Bn, N = ts.shape
        counts = np.array([len(l) for l in logs], dtype=np.int64)
        offs = np.zeros(Bn + 1, dtype=np.int64); offs[1:] = np.cumsum(counts)
        allr = np.concatenate([np.asarray(l, dtype=np.float64)[:, :4] for l in logs]) if counts.sum() else np.zeros((0, 4))
        f = dict(dtype=torch.float64, device=device)
        return cls(Bn, N, torch.as_tensor(ts, **f).contiguous(), torch.as_tensor(np.asarray(pos, dtype=np.float64), **f).contiguous(),
                   torch.as_tensor(np.asarray(quat, dtype=np.float64), **f).contiguous(), torch.as_tensor(offs, device=device),
                   torch.as_tensor(np.ascontiguousarray(allr[:, 0]), **f), torch.as_tensor(np.ascontiguousarray(allr[:, 1:4]), **f), int(counts.max(initial=0)))

    @classmethod
    def synthetic(cls, B, N, seed=20250523, traj0=0, device="cuda"):
        """Deterministic KITTI-04-shaped trajectories with a geodetic GNSS log around (49.0336 N, 8.3950 E)."""
        L, h = _lib.load(), context().handle
        f = dict(dtype=torch.float64, device=device)
        ts, pos, quat = torch.empty((B, N), **f), torch.empty((B, N, 3), **f), torch.empty((B, N, 4), **f)
        counts = torch.empty((B,), dtype=torch.int64, device=device)
        check(L.gsf_synth_geodetic_batch_dev(h, C.c_uint64(seed), int(traj0), B, N, None, None, None, _p(counts), None, None, None))
        offs = torch.zeros((B + 1,), dtype=torch.int64, device=device)
        offs[1:] = torch.cumsum(counts, 0)
        total, mx = int(offs[-1].item()), int(counts.max().item())      # sizing of the ragged log (host values, once per batch)
        gps_t, gps_llh = torch.empty((total,), **f), torch.empty((total, 3), **f)
        check(L.gsf_synth_geodetic_batch_dev(h, C.c_uint64(seed), int(traj0), B, N, _p(ts), _p(pos), _p(quat), None, _p(offs), _p(gps_t), _p(gps_llh)))
        return cls(B, N, ts, pos, quat, offs, gps_t, gps_llh, mx)


    def with_outliers(self, share, metres=40.0, seed=7):
        """A copy of the batch whose GNSS log has `share` of its fixes pushed `metres` north (multipath-like jumps the pre-filter is there
        for, ref :136-247); deterministic in `seed`.  SLAM side shared, log copied."""
        g = torch.Generator(device="cpu"); g.manual_seed(int(seed))
        hit = (torch.rand(self.gps_t.numel(), generator=g) < float(share)).to(self.gps_llh.device)
        llh = self.gps_llh.clone()
        llh[:, 0] += hit.double() * (float(metres) / 111320.0)
        return GeodeticBatch(self.B, self.N, self.ts, self.pos, self.quat, self.gps_offsets, self.gps_t, llh, self.max_fixes)


FIT_ROWS_DEFAULT = "reference"


def fuse_from_geodetic(gb, config=None, out=None, fit_rows=FIT_ROWS_DEFAULT):
    """The whole path from the geodetic GNSS log on the device, no host round trip: geodesy slice (mask, zone pick, UTM forward,
    [E, N, alt] rows; ref :258-271) -> dynamic_time_alignment to the SLAM stamps (ref :325-387, :971) -> Umeyama on the rows
    main_process_gui picks (ref :973-998; fit_rows="all": on every valid row) -> Sim3 of pose 0 -> EKF+RTS (ref :1002-1010, plain
    fit).  Returns (FusedPoses, R, t, s, aux) with aux = dict(zone, south, utm_rows, aligned, valid)."""
    g = config or CONFIG
    context().set_sim3_rows(fit_rows, g)
    cfg = EkfConfig.from_config(g)
    L, h, dev = _lib.load(), context().handle, gb.ts.device
    f = dict(dtype=torch.float64, device=dev)
    utm = torch.empty_like(gb.gps_llh)
    zone, south = torch.empty((gb.B,), dtype=torch.int32, device=dev), torch.empty((gb.B,), dtype=torch.int32, device=dev)
    check(L.gsf_gps_rows_to_utm_batch_dev(h, _p(gb.gps_llh), _p(gb.gps_offsets), gb.B, _p(utm), _p(zone), _p(south)))
    aligned = torch.empty((gb.B, gb.N, 3), **f)
    valid = torch.empty((gb.B, gb.N), dtype=torch.uint8, device=dev)
    # (rows the loader removes -- lat/lon zero or out of range, ref :259-264 -- come out of the geodesy slice as NaN rows and are dropped
    # when the log is staged: the alignment sees the fixes load_gps_data would have returned)
    check(L.gsf_time_align_loaded_rows_batch_dev(h, _p(gb.ts), _p(gb.slam_offsets), _p(gb.gps_t), _p(utm), _p(gb.gps_offsets), gb.B, max(2, gb.max_fixes),
                                                 float(g["time_alignment"]["max_gps_gap_threshold"]), _p(aligned), _p(valid), None))
    out = out or FusedPoses(LAYOUT_TRAJ_MAJOR, gb.B, gb.N, dev)
    R, t, s = torch.empty((gb.B, 9), **f), torch.empty((gb.B, 3), **f), torch.empty((gb.B,), **f)
    check(L.gsf_fuse_pipeline_batch_dev(h, LAYOUT_TRAJ_MAJOR, _p(gb.ts), _p(gb.pos), _p(gb.quat), _p(aligned), _p(valid), C.byref(cfg), gb.B, gb.N,
                                        _p(R), _p(t), _p(s), _p(out.pos), _p(out.quat), _p(out.status)))
    return out, R, t, s, {"zone": zone, "south": south, "utm_rows": utm, "aligned": aligned, "valid": valid}


class RunResult:
    """What steps 1-6 of main_process_gui leave behind for B trajectories (run_fusion_batch): fused = FusedPoses, R / t / s, n_inliers,
    inlier_mask, trial_info (deciding trial, trials drawn), zone / south, gps_utm (NaN rows where the loader drops the fix), gps_keep (fixes
    that survive loader and pre-filter), aligned / valid (step 2), sim3_pos (step 4), err_stats (3, B, 4) = count / mean / median / RMSE of
    raw SLAM, Sim3, EKF against the primary GPS (step 6), run_status (B,) = RUN_* bits (0 = the reference's run completes)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def run_fusion_batch(gb, mt_state, config=None, early_exit=True, skip_seconds=5.0, max_windows=0, want_mask=True, projected=False):
    """Steps 1-6 of main_process_gui (EKFGPSSLAM.py:959-1033) for the B trajectories of a GeodeticBatch as ONE device chain on torch's
    current stream: load-side geodesy (:258-271) -> GPS RANSAC pre-filter with its windows walked on the device (:275, :136-247) ->
    time alignment (:971) -> row choice (:973-998) -> robust Sim3 (:1002) -> apply (:1006) -> EKF + RTS (:1010) -> error metric (:1013-1033).
    mt_state (B, 625): every trajectory's NumPy legacy generator (mt19937_seed / mt19937_from_numpy), advanced by the pre-filter's and
    the fit's draws in the reference's order; early_exit as in fuse_pipeline_robust_batch (the pre-filter's draws are unaffected).
    projected=True: gb.gps_llh already holds (E, N, alt) rows -- what load_gps_data's projection returns -- and the chain starts at the
    pre-filter.  Returns a RunResult."""
    g = config or CONFIG
    ctx = context()
    ctx.set_option("ransac_early_exit", 1 if early_exit else 0)
    rc = _lib.RunConfig.from_config(g, skip_seconds, max_windows)
    B, N, dev = gb.B, gb.N, gb.ts.device
    f = dict(dtype=torch.float64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    total = int(gb.gps_t.numel())
    out = FusedPoses(LAYOUT_TRAJ_MAJOR, B, N, dev)
    r = RunResult(fused=out, R=torch.empty((B, 9), **f), t=torch.empty((B, 3), **f), s=torch.empty((B,), **f), n_inliers=torch.empty((B,), **i32),
                  zone=torch.empty((B,), **i32), south=torch.empty((B,), **i32), gps_utm=gb.gps_llh.clone() if projected else torch.empty((total, 3), **f),
                  gps_keep=torch.empty((total,), dtype=torch.uint8, device=dev), aligned=torch.empty((B, N, 3), **f),
                  valid=torch.empty((B, N), dtype=torch.uint8, device=dev), sim3_pos=torch.empty((B, N, 3), **f), err_stats=torch.empty((3, B, 4), **f),
                  run_status=torch.empty((B,), **i32), inlier_mask=torch.empty((B, N), dtype=torch.uint8, device=dev) if want_mask else None,
                  trial_info=torch.empty((B, 2), **i32))
    check(_lib.load().gsf_run_fusion_batch_dev(ctx.handle, _p(gb.ts), _p(gb.pos), _p(gb.quat), B, N, _p(gb.gps_t), None if projected else _p(gb.gps_llh), _p(gb.gps_offsets), total,
                                               int(gb.max_fixes), C.byref(rc), _p(mt_state), _p(r.R), _p(r.t), _p(r.s), _p(out.pos), _p(out.quat), _p(out.status),
                                               _p(r.n_inliers), _p(r.zone), _p(r.south), _p(r.gps_utm), _p(r.gps_keep), _p(r.aligned), _p(r.valid), _p(r.sim3_pos),
                                               _p(r.err_stats), _p(r.run_status), _p(r.inlier_mask), _p(r.trial_info)))
    return r


class FusedPoses:
    """Fused poses of a batch.  pos and quat are views of ONE allocation `buf` = [pos | quat] (7 doubles per pose), so the
    multi-GPU collect is a single all-gather of `buf`.  `buf` may be a caller-provided slice of a larger arena."""

    def __init__(self, layout, B, N, device="cuda", buf=None):
        _, s_pos, s_quat, _, _ = shapes(layout, B, N)
        self.layout, self.B, self.N = layout, B, N
        P = B * N
        if buf is None:
            buf = torch.empty((P * 7,), dtype=torch.float64, device=device)
        elif buf.dtype != torch.float64 or buf.numel() != P * 7 or not buf.is_contiguous():
            raise ValueError(f"FusedPoses: buf must be a contiguous float64 tensor of {P * 7} elements")
        self.buf = buf.view(-1)
        self.pos = self.buf[: P * 3].view(s_pos)
        self.quat = self.buf[P * 3:].view(s_quat)
        self.status = torch.empty((B,), dtype=torch.int32, device=self.buf.device)

    def host_traj_major(self):
        """-> (pos (B,N,3), quat (B,N,4), status (B,)) numpy"""
        if self.layout == LAYOUT_TRAJ_MAJOR:
            pos, quat = self.pos, self.quat
        else:
            pos = torch.empty((self.B, self.N, 3), dtype=torch.float64, device=self.pos.device)
            quat = torch.empty((self.B, self.N, 4), dtype=torch.float64, device=self.pos.device)
            L, h = _lib.load(), context().handle
            check(L.gsf_transpose_to_traj_major_dev(h, _p(self.pos), _p(pos), self.B, self.N, 3, 8))
            check(L.gsf_transpose_to_traj_major_dev(h, _p(self.quat), _p(quat), self.B, self.N, 4, 8))
        torch.cuda.synchronize()
        return pos.cpu().numpy(), quat.cpu().numpy(), self.status.cpu().numpy()


def ekf_fuse_batch(batch, config=None, out=None):
    """K4 over a device batch: apply_ekf_correction (EKFGPSSLAM.py:831-935, after its :847 alignment) per trajectory.
    Asynchronous on torch's current stream; returns FusedPoses in the batch's layout."""
    cfg = EkfConfig.from_config(config or CONFIG)
    out = out or FusedPoses(batch.layout, batch.B, batch.N, batch.ts.device)
    check(_lib.load().gsf_ekf_fuse_batch_dev(context().handle, batch.layout, _p(batch.ts), _p(batch.pos), _p(batch.quat), _p(batch.gps),
                                             _p(batch.valid), _p(batch.init_pos), _p(batch.init_quat), C.byref(cfg), batch.B, batch.N,
                                             _p(out.pos), _p(out.quat), _p(out.status)))
    return out


def fuse_pipeline_batch(batch, config=None, out=None, fit_rows=FIT_ROWS_DEFAULT):
    """Umeyama -> Sim3 of pose 0 -> EKF+RTS in one launch (steps 3-5 of EKFGPSSLAM.py:1002-1010, plain fit).  fit_rows="reference":
    the fit sees the rows main_process_gui hands to its fit (first gap-free segment of the valid rows, <= max_initial_duration, two
    fall-backs; ref :973-998); "all": every row with valid finite GNSS.  status >> 8 carries the GSF_SIM3_* bits (FEW_ROWS = the
    reference's ValueError).  Returns (FusedPoses, R (B,9), t (B,3), s (B,))."""
    cfg = EkfConfig.from_config(config or CONFIG)
    context().set_sim3_rows(fit_rows, config or CONFIG)
    out = out or FusedPoses(batch.layout, batch.B, batch.N, batch.ts.device)
    f = dict(dtype=torch.float64, device=batch.ts.device)
    R, t, s = torch.empty((batch.B, 9), **f), torch.empty((batch.B, 3), **f), torch.empty((batch.B,), **f)
    check(_lib.load().gsf_fuse_pipeline_batch_dev(context().handle, batch.layout, _p(batch.ts), _p(batch.pos), _p(batch.quat), _p(batch.gps),
                                                  _p(batch.valid), C.byref(cfg), batch.B, batch.N, _p(R), _p(t), _p(s), _p(out.pos),
                                                  _p(out.quat), _p(out.status)))
    return out, R, t, s


def mt19937_seed(seeds, device="cuda"):
    """np.random.seed(seeds[b]) for B independent legacy-MT19937 streams -> state (B, 625) uint32 on the device (key[624] + pos)."""
    seeds = torch.as_tensor(seeds, dtype=torch.int64).to(device)
    st = torch.empty((seeds.numel(), 625), dtype=torch.int32, device=device)
    check(_lib.load().gsf_mt19937_seed_batch_dev(context().handle, _p(seeds.to(torch.int32)), seeds.numel(), _p(st)))
    return st


def mt19937_from_numpy(device="cuda"):
    """NumPy's GLOBAL legacy generator as a one-stream device state (1, 625): what a seeded reference run would draw from next."""
    import numpy as np
    _, key, pos = np.random.get_state()[:3]
    st = np.concatenate([key.astype(np.uint32), np.array([pos], dtype=np.uint32)]).view(np.int32)
    return torch.from_numpy(st.copy()).reshape(1, 625).to(device)


def mt19937_choice_batch(state, n_population, trials, k):
    """sample_idx (B, trials, k) int32 = np.random.choice(n_population[b], k, replace=False) drawn `trials` times from stream b;
    `state` (B, 625) is advanced in place exactly as NumPy's generator would be."""
    B = state.shape[0]
    # populations given on the host: their maximum lets the library take the chip-wide route for a few streams (gsf.h); a device
    # tensor is passed as it is (no synchronising read-back)
    n_max = 0 if (torch.is_tensor(n_population) and n_population.is_cuda) else int(max((int(v) for v in torch.as_tensor(n_population).reshape(-1)), default=0))
    n = torch.as_tensor(n_population, dtype=torch.int32).to(state.device).contiguous()
    idx = torch.empty((B, trials, k), dtype=torch.int32, device=state.device)
    check(_lib.load().gsf_mt19937_choice_bounded_batch_dev(context().handle, _p(state), _p(n), n_max, B, int(trials), int(k), _p(idx)))
    return idx


def fuse_pipeline_robust_batch(batch, mt_state, config=None, out=None, want_mask=True, fit_rows=FIT_ROWS_DEFAULT, early_exit=True, return_info=False):
    """Steps 3-5 of main_process_gui with the reference's robust fit (EKFGPSSLAM.py:1002-1010): the rows main_process_gui picks
    (ref :973-998; fit_rows="all": every valid row) -> RANSAC hypotheses drawn on the device from each trajectory's legacy MT19937
    stream -> inlier refit -> Sim3 of pose 0 -> EKF+RTS, one chain on torch's current stream.  Trajectory-major batches.
    early_exit (default on): a trajectory stops drawing at the first trial that counts every row of its fit -- the reference keeps a trial
    only on a strictly larger count (ref :413), so R, t, s, mask, n_inliers and the fused poses are those of all max_trials, bit for bit;
    only where `mt_state` is left differs (after fewer trials).  early_exit=False: every generator ends where np.random ends in the
    reference.  Returns (FusedPoses, R, t, s, n_inliers (B,), inlier_mask (B, N) uint8 or None[, trial_info (B, 2) int32 = deciding trial,
    trials drawn])."""
    if batch.layout != LAYOUT_TRAJ_MAJOR:
        raise ValueError("fuse_pipeline_robust_batch: trajectory-major batches only")
    g = config or CONFIG
    ctx = context()
    ctx.set_sim3_rows(fit_rows, g)
    ctx.set_option("ransac_early_exit", 1 if early_exit else 0)
    cfg, r = EkfConfig.from_config(g), g["sim3_ransac"]
    out = out or FusedPoses(batch.layout, batch.B, batch.N, batch.ts.device)
    f = dict(dtype=torch.float64, device=batch.ts.device)
    R, t, s = torch.empty((batch.B, 9), **f), torch.empty((batch.B, 3), **f), torch.empty((batch.B,), **f)
    nin = torch.empty((batch.B,), dtype=torch.int32, device=batch.ts.device)
    mask = torch.empty((batch.B, batch.N), dtype=torch.uint8, device=batch.ts.device) if want_mask else None
    info = torch.empty((batch.B, 2), dtype=torch.int32, device=batch.ts.device) if return_info else None
    check(_lib.load().gsf_fuse_pipeline_robust_info_batch_dev(ctx.handle, _p(batch.ts), _p(batch.pos), _p(batch.quat), _p(batch.gps), _p(batch.valid),
                                                              C.byref(cfg), batch.B, batch.N, int(r["min_samples"]), float(r["residual_threshold"]),
                                                              int(r["max_trials"]), int(r["min_inliers_needed"]), _p(mt_state), _p(R), _p(t), _p(s),
                                                              _p(out.pos), _p(out.quat), _p(out.status), _p(nin), _p(mask), _p(info)))
    return (out, R, t, s, nin, mask, info) if return_info else (out, R, t, s, nin, mask)


def sim3_fit_rows_batch(ts, gps, valid, config=None, offsets=None):
    """main_process_gui's choice of the rows that feed the global Sim3 (EKFGPSSLAM.py:973-998) for B trajectories on the device: ts (B,N),
    gps (B,N,3) or None, valid (B,N) uint8 -- or flat tensors with int64 offsets (B+1,).  Returns row_mask (uint8, shape of valid),
    n_rows (B,) int32 (-1 where the reference raises ValueError), status (B,) int32 (GSF_SIM3_FLAG_FEW_ROWS / _ROWS_ALL / _ROWS_SEGMENT)."""
    g = config or CONFIG
    Bn = (offsets.numel() - 1) if offsets is not None else ts.shape[0]
    N = 0 if offsets is not None else ts.shape[1]
    mask = torch.empty(valid.shape, dtype=torch.uint8, device=ts.device)
    n_rows, st = torch.empty((Bn,), dtype=torch.int32, device=ts.device), torch.empty((Bn,), dtype=torch.int32, device=ts.device)
    check(_lib.load().gsf_sim3_fit_rows_batch_dev(context().handle, _p(ts), _p(gps), _p(valid), _p(offsets), Bn, N, int(g["sim3_ransac"]["min_samples"]),
                                                  float(g["time_alignment"]["max_gps_gap_threshold"]), float(g["sim3_ransac"]["max_initial_duration"]),
                                                  _p(mask), _p(n_rows), _p(st)))
    return mask, n_rows, st


def sim3_umeyama_batch(src, dst, offsets=None, mask=None):
    """K2 over device tensors.  src/dst: (total,3) with int64 offsets (B+1,), or (B,W,3) equal-size windows (config C4; mask (B,W)).
    Returns R (B,9), t (B,3), s (B,), status (B,) int32."""
    f = dict(dtype=torch.float64, device=src.device)
    if src.dim() == 3:
        B, W, _ = src.shape
        _chk(src, torch.float64, (B, W, 3), "src"); _chk(dst, torch.float64, (B, W, 3), "dst")
        R, t, s = torch.empty((B, 9), **f), torch.empty((B, 3), **f), torch.empty((B,), **f)
        st = torch.empty((B,), dtype=torch.int32, device=src.device)
        check(_lib.load().gsf_sim3_umeyama_windows_dev(context().handle, _p(src), _p(dst), _p(mask), B, W, _p(R), _p(t), _p(s), _p(st)))
        return R, t, s, st
    B = offsets.numel() - 1
    _chk(src, torch.float64, src.shape, "src"); _chk(dst, torch.float64, src.shape, "dst")
    R, t, s = torch.empty((B, 9), **f), torch.empty((B, 3), **f), torch.empty((B,), **f)
    st = torch.empty((B,), dtype=torch.int32, device=src.device)
    check(_lib.load().gsf_sim3_umeyama_batch_dev(context().handle, _p(src), _p(dst), _p(mask), _p(offsets), B, _p(R), _p(t), _p(s), _p(st)))
    return R, t, s, st


def sim3_ransac_batch(src, dst, offsets, sample_idx, residual_threshold, min_inliers_needed):
    """K2b over device tensors; sample_idx (B,trials,ms) int32 drawn by the caller with the reference's RNG call."""
    B, trials, ms = sample_idx.shape
    f = dict(dtype=torch.float64, device=src.device)
    R, t, s = torch.empty((B, 9), **f), torch.empty((B, 3), **f), torch.empty((B,), **f)
    st, nin = torch.empty((B,), dtype=torch.int32, device=src.device), torch.empty((B,), dtype=torch.int32, device=src.device)
    mask = torch.empty((src.shape[0],), dtype=torch.uint8, device=src.device)
    check(_lib.load().gsf_sim3_ransac_batch_rows_dev(context().handle, _p(src), _p(dst), _p(offsets), int(src.shape[0]), B, _p(sample_idx), trials, ms,
                                                     float(residual_threshold), int(min_inliers_needed), _p(R), _p(t), _p(s), _p(st), _p(mask), _p(nin)))
    return R, t, s, st, mask, nin


def apply_sim3_batch(pos, quat, offsets, R, t, s):
    """K3 over device tensors: pos (total,3), quat (total,4) -> transformed copies (+ bad_quat flags (B,))."""
    B = offsets.numel() - 1
    po, qo = torch.empty_like(pos), torch.empty_like(quat)
    bad = torch.empty((B,), dtype=torch.int32, device=pos.device)
    check(_lib.load().gsf_apply_sim3_batch_dev(context().handle, _p(pos), _p(quat), _p(offsets), B, _p(R), _p(t), _p(s), _p(po), _p(qo), _p(bad)))
    return po, qo, bad


def utm_forward_batch(lat, lon, offsets, zone=None, south=None):
    """K1 over device tensors (ragged trajectories).  zone/south None -> picked per trajectory like auto_utm_projection."""
    B = offsets.numel() - 1
    L, h = _lib.load(), context().handle
    if zone is None:
        zone = torch.empty((B,), dtype=torch.int32, device=lat.device)
        south = torch.empty((B,), dtype=torch.int32, device=lat.device)
        check(L.gsf_utm_zone_batch_dev(h, _p(lat), _p(lon), _p(offsets), B, _p(zone), _p(south)))
    e, n = torch.empty_like(lat), torch.empty_like(lat)
    check(L.gsf_utm_forward_batch_dev(h, _p(lat), _p(lon), _p(offsets), _p(zone), _p(south), B, _p(e), _p(n)))
    return e, n, zone, south


def utm_inverse_batch(e, n, offsets, zone, south):
    B = offsets.numel() - 1
    lat, lon = torch.empty_like(e), torch.empty_like(e)
    check(_lib.load().gsf_utm_inverse_batch_dev(context().handle, _p(e), _p(n), _p(offsets), _p(zone), _p(south), B, _p(lat), _p(lon)))
    return lat, lon


def fuse_pipeline_ragged(ts, pos, quat, gps, valid, offsets, config=None, fit_rows=FIT_ROWS_DEFAULT):
    """Fused Umeyama -> Sim3(pose 0) -> EKF+RTS for trajectories of DIFFERENT lengths: flat device tensors ts (T,), pos (T,3),
    quat (T,4), gps (T,3), valid (T,) uint8 and int64 offsets (B+1,); fit_rows as in fuse_pipeline_batch.  Returns pos_out, quat_out,
    status, R, t, s."""
    cfg = EkfConfig.from_config(config or CONFIG)
    context().set_sim3_rows(fit_rows, config or CONFIG)
    B = offsets.numel() - 1
    f = dict(dtype=torch.float64, device=ts.device)
    po, qo = torch.empty_like(pos), torch.empty_like(quat)
    R, t, s = torch.empty((B, 9), **f), torch.empty((B, 3), **f), torch.empty((B,), **f)
    st = torch.empty((B,), dtype=torch.int32, device=ts.device)
    check(_lib.load().gsf_fuse_pipeline_ragged_dev(context().handle, _p(ts), _p(pos), _p(quat), _p(gps), _p(valid), _p(offsets), C.byref(cfg), B,
                                                   _p(R), _p(t), _p(s), _p(po), _p(qo), _p(st)))
    return po, qo, st, R, t, s


def ekf_fuse_ragged(ts, pos, quat, gps, valid, offsets, init_pos, init_quat, config=None):
    """K4 for trajectories of different lengths (flat tensors + offsets, see fuse_pipeline_ragged)."""
    cfg = EkfConfig.from_config(config or CONFIG)
    B = offsets.numel() - 1
    po, qo = torch.empty_like(pos), torch.empty_like(quat)
    st = torch.empty((B,), dtype=torch.int32, device=ts.device)
    check(_lib.load().gsf_ekf_fuse_ragged_dev(context().handle, _p(ts), _p(pos), _p(quat), _p(gps), _p(valid), _p(offsets), _p(init_pos),
                                              _p(init_quat), C.byref(cfg), B, _p(po), _p(qo), _p(st)))
    return po, qo, st


def geodetic_to_enu_batch(lat, lon, alt, offsets, ref_llh):
    """WGS84 geodetic -> local ENU about ref_llh (B,3) per trajectory (device tensors).  Offered in addition to UTM."""
    B = offsets.numel() - 1
    e, n, u = torch.empty_like(lat), torch.empty_like(lat), torch.empty_like(lat)
    check(_lib.load().gsf_geodetic_to_enu_batch_dev(context().handle, _p(lat), _p(lon), _p(alt), _p(offsets), _p(ref_llh), B, _p(e), _p(n), _p(u)))
    return e, n, u


def ransac_poly_batch(t, y, offsets, sample_idx, degree, residual_threshold, stop_probability=0.99):
    """next-3 kernel over device tensors: P polynomial-RANSAC problems (rows offsets[p]..offsets[p+1] of t, y), sample sets
    sample_idx (P, max_trials, min_samples) int32 drawn by the caller.  Returns inlier_mask (rows,) uint8, n_trials, n_inliers,
    status (P,) int32 -- see gsf_ransac_poly_batch_dev."""
    P, trials, ms = sample_idx.shape
    mask = torch.empty((t.shape[0],), dtype=torch.uint8, device=t.device)
    ntr, nin, st = (torch.empty((P,), dtype=torch.int32, device=t.device) for _ in range(3))
    check(_lib.load().gsf_ransac_poly_batch_dev(context().handle, _p(t), _p(y), _p(offsets), P, _p(sample_idx), trials, ms, int(degree),
                                                float(residual_threshold), float(stop_probability), _p(mask), _p(ntr), _p(nin), _p(st)))
    return mask, ntr, nin, st


def eval_errors_batch(ts, traj_pos, gps, valid, skip_seconds=0.0):
    """next-4 over device tensors (trajectory-major (B,N) / (B,N,3)): nearest-fix error of every fused pose (ref :1013-1033).
    Returns stats (B,4) = count, mean, median, RMSE and errors (B,N) (NaN where not evaluated)."""
    Bn, N = ts.shape
    stats = torch.empty((Bn, 4), dtype=torch.float64, device=ts.device)
    err = torch.empty((Bn, N), dtype=torch.float64, device=ts.device)
    check(_lib.load().gsf_eval_errors_batch_dev(context().handle, _p(ts), _p(traj_pos), _p(gps), _p(valid), Bn, N, float(skip_seconds), _p(stats), _p(err)))
    return stats, err
