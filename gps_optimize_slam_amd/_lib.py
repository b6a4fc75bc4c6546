"""ctypes binding of libgsf.so (C ABI: include/gsf.h).  Thin: argument marshalling only."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("GSF_LIBRARY") or os.path.join(_HERE, "libgsf.so")      # GSF_LIBRARY: A/B a differently built libgsf.so

LAYOUT_TRAJ_MAJOR = 0
LAYOUT_TIME_MAJOR = 1

SIM3_NONE = 1
SIM3_FLAG_FEW_ROWS, SIM3_FLAG_ROWS_ALL, SIM3_FLAG_ROWS_SEGMENT = 32, 64, 128      # row choice of the fused chains (gsf_set_sim3_rows mode 1)
ST_HAD_OUTAGE, ST_RTS_APPLIED, ST_SHARP_TURN, ST_ENDED_IN_OUTAGE, ST_BAD_QUAT = 1, 2, 4, 8, 16


class GsfError(RuntimeError):
    pass


class EkfConfig(C.Structure):
    """gsf_ekf_config (include/gsf.h) <- CONFIG['ekf'] + CONFIG['rts_decision'] (EKFGPSSLAM.py:24-29, :67-70)."""
    _fields_ = [("initial_cov_diag", C.c_double * 7), ("process_noise_diag", C.c_double * 7),
                ("meas_noise_diag", C.c_double * 3), ("sharp_turn_yaw_rate_threshold_deg_per_sec", C.c_double),
                ("default_ekf_transition_steps_on_sharp_turn", C.c_int32), ("reserved", C.c_int32)]

    @classmethod
    def from_config(cls, global_config):
        e, r = global_config["ekf"], global_config["rts_decision"]
        c = cls()
        for name, n in (("initial_cov_diag", 7), ("process_noise_diag", 7), ("meas_noise_diag", 3)):
            v = [float(x) for x in e[name]]
            if len(v) != n:
                raise ValueError(f"EKF初始化: {name} must have {n} entries")       # EKFGPSSLAM.py:687-688
            getattr(c, name)[:] = v
        c.sharp_turn_yaw_rate_threshold_deg_per_sec = float(r["sharp_turn_yaw_rate_threshold_deg_per_sec"])
        c.default_ekf_transition_steps_on_sharp_turn = int(r["default_ekf_transition_steps_on_sharp_turn"])
        return c


class PrefilterConfig(C.Structure):
    """gsf_prefilter_config <- CONFIG['gps_filtering_ransac'] / CONFIG['ground_truth_gps_filtering'] (EKFGPSSLAM.py:39-48, :55-64)."""
    _fields_ = [("enabled", C.c_int32), ("use_sliding_window", C.c_int32), ("window_duration_seconds", C.c_double), ("window_step_factor", C.c_double),
                ("polynomial_degree", C.c_int32), ("min_samples", C.c_int32), ("residual_threshold_meters", C.c_double), ("max_trials", C.c_int32),
                ("max_windows", C.c_int32), ("stop_probability", C.c_double)]

    @classmethod
    def from_config(cls, f, max_windows=0):
        c = cls()
        c.enabled, c.use_sliding_window = int(bool(f.get("enabled", False))), int(bool(f.get("use_sliding_window", False)))
        c.window_duration_seconds, c.window_step_factor = float(f.get("window_duration_seconds", 0.0)), float(f.get("window_step_factor", 0.0))
        c.polynomial_degree, c.min_samples = int(f["polynomial_degree"]), int(f["min_samples"])
        c.residual_threshold_meters, c.max_trials = float(f["residual_threshold_meters"]), int(f["max_trials"])
        c.max_windows, c.stop_probability = int(max_windows), 0.99
        return c


class RunConfig(C.Structure):
    """gsf_run_config <- the whole CONFIG dict (EKFGPSSLAM.py:22-71): what main_process_gui reads between its step 1 and its step 6."""
    _fields_ = [("ekf", EkfConfig), ("gps_filter", PrefilterConfig), ("sim3_residual_threshold", C.c_double), ("sim3_max_initial_duration", C.c_double),
                ("max_gps_gap_threshold", C.c_double), ("eval_skip_seconds", C.c_double), ("sim3_min_samples", C.c_int32), ("sim3_max_trials", C.c_int32),
                ("sim3_min_inliers_needed", C.c_int32), ("reserved", C.c_int32)]

    @classmethod
    def from_config(cls, g, skip_seconds=5.0, max_windows=0):
        c = cls()
        c.ekf, c.gps_filter = EkfConfig.from_config(g), PrefilterConfig.from_config(g["gps_filtering_ransac"], max_windows)
        r = g["sim3_ransac"]
        c.sim3_residual_threshold, c.sim3_max_initial_duration = float(r["residual_threshold"]), float(r["max_initial_duration"])
        c.max_gps_gap_threshold, c.eval_skip_seconds = float(g["time_alignment"]["max_gps_gap_threshold"]), float(skip_seconds)
        c.sim3_min_samples, c.sim3_max_trials, c.sim3_min_inliers_needed = int(r["min_samples"]), int(r["max_trials"]), int(r["min_inliers_needed"])
        return c


RUN_GPS_EMPTY, RUN_GPS_FEW, RUN_PREFILTER_UNHANDLED, RUN_SIM3_FAILED, RUN_BAD_QUAT = 1, 2, 4, 8, 16      # run_status bits of gsf_run_fusion_batch_dev
SIM3_FLAG_SATURATED = 256


def library_path():
    return _SO


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 build of libgsf.so (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return _SO


_vp, _i64, _i32, _f64 = C.c_void_p, C.c_int64, C.c_int32, C.c_double

# name -> (restype, argtypes); must list EVERY function include/gsf.h declares (tests/test_capi_symbols.py checks)
SIGNATURES = {
    "gsf_version": (C.c_char_p, []),
    "gsf_abi_version": (C.c_int, []),
    "gsf_last_error": (C.c_int, [C.c_char_p, C.c_int]),
    "gsf_device_count": (C.c_int, []),
    "gsf_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "gsf_create_on_stream": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "gsf_destroy": (None, [_vp]),
    "gsf_synchronize": (C.c_int, [_vp]),
    "gsf_trim": (C.c_int, [_vp]),
    "gsf_set_option": (C.c_int, [_vp, C.c_char_p, _i64]),
    "gsf_set_sim3_rows": (C.c_int, [_vp, _i32, _i32, _f64, _f64]),
    "gsf_sim3_fit_rows_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _f64, _f64, _vp, _vp, _vp]),
    "gsf_sim3_fit_rows_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _f64, _f64, _vp, _vp, _vp]),
    "gsf_timer_start": (C.c_int, [_vp]),
    "gsf_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "gsf_utm_zone_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gsf_utm_forward_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gsf_utm_inverse_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gsf_utm_forward": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "gsf_utm_inverse": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "gsf_gps_rows_to_utm_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gsf_gps_rows_to_utm_batch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gsf_geodetic_to_enu_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gsf_geodetic_to_enu_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gsf_ransac_poly_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, _vp, _vp, _vp, _vp]),
    "gsf_ransac_poly_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, _vp, _vp, _vp, _vp]),
    "gsf_gps_prefilter_chain_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _f64, _f64, _vp, _vp, _vp, _vp]),
    "gsf_gps_prefilter_chain": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _f64, _f64, _vp, _vp, _vp, _vp]),
    "gsf_gps_prefilter_auto_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, C.POINTER(PrefilterConfig), _vp, _vp, _vp, _vp]),
    "gsf_run_fusion_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i32, C.POINTER(RunConfig), _vp] + [_vp] * 18),
    "gsf_run_fusion_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, C.POINTER(RunConfig), _vp] + [_vp] * 18),
    "gsf_sim3_umeyama_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "gsf_sim3_umeyama_windows_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "gsf_sim3_umeyama_windows": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "gsf_sim3_umeyama_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "gsf_sim3_ransac_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _f64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_sim3_ransac_batch_rows_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i32, _i32, _f64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_sim3_ransac_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _f64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_sim3_ransac_mt_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _f64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_mt19937_seed_batch_dev": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gsf_mt19937_choice_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "gsf_mt19937_choice_bounded_batch_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "gsf_fuse_pipeline_robust_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _i64, _i32, _f64, _i32, _i32, _vp,
                                                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_fuse_pipeline_robust_info_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _i64, _i32, _f64, _i32, _i32, _vp,
                                                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_fuse_pipeline_robust_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _i64, _i32, _f64, _i32, _i32, _vp,
                                                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_apply_sim3_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_apply_sim3_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_ekf_fuse_batch_dev": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _i64, _vp, _vp, _vp]),
    "gsf_ekf_fuse_batch": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _i64, _vp, _vp, _vp]),
    "gsf_fuse_pipeline_batch_dev": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_fuse_pipeline_batch": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_ekf_fuse_ragged_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _vp, _vp, _vp]),
    "gsf_ekf_fuse_ragged": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _vp, _vp, _vp]),
    "gsf_fuse_pipeline_ragged_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_fuse_pipeline_ragged": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(EkfConfig), _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_time_align_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _vp, _vp, _vp]),
    "gsf_time_align_loaded_rows_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _vp, _vp, _vp]),
    "gsf_time_align_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp]),
    "gsf_eval_errors_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f64, _vp, _vp]),
    "gsf_eval_errors_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f64, _vp, _vp]),
    "gsf_relative_pose_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gsf_quaternion_nlerp_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "gsf_is_sharp_turn_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp]),
    "gsf_ekf_process_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), C.POINTER(_f64), _i32, _vp, _vp, _vp, _i32, _f64, _i32, _vp, _vp]),
    "gsf_rts_smoother_segment_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gsf_comm_unique_id": (C.c_int, [_vp]),
    "gsf_comm_rccl_version": (C.c_int, [C.POINTER(C.c_int32)]),
    "gsf_comm_init_rank": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "gsf_comm_destroy": (C.c_int, [_vp]),
    "gsf_allgather_poses": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i64]),
    "gsf_transpose_to_time_major_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32]),
    "gsf_transpose_to_traj_major_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32]),
    "gsf_synth_geodetic_batch_dev": (C.c_int, [_vp, C.c_uint64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsf_synth_batch_dev": (C.c_int, [_vp, _i32, C.c_uint64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, same as /opt/rocm's).  Two HIP
    runtimes in one process each see 0 devices once the other owns the KFD queue, so when torch is installed its copy is
    mapped FIRST (without importing torch); libgsf.so's DT_NEEDED then binds to it by soname and the process has a single
    runtime whichever of torch / libgsf is touched first.  Without torch, libgsf.so uses /opt/rocm's runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


_hip_rt = None


def load():
    """dlopen libgsf.so and type every entry point.  Raises GsfError if the library is not built."""
    global _lib, _hip_rt
    with _lock:
        if _lib is None:
            if not os.path.exists(_SO):
                raise GsfError(f"{_SO} is missing: build it with gps_optimize_slam_amd.build_library() "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
            _hip_rt = _preload_torch_hip_runtime()
            L = C.CDLL(_SO)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
            if L.gsf_abi_version() != 1:
                raise GsfError("libgsf.so ABI version mismatch")
            _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    load().gsf_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc):
    if rc != 0:
        raise GsfError(f"libgsf error {rc}: {last_error()}")


class Context:
    """Owns a gsf_ctx (device + stream).  `stream` = a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None):
        L = load()
        if L.gsf_device_count() <= 0:
            raise GsfError("no HIP device visible: the fusion kernels need an MI355X (no CPU fallback)")
        h = _vp()
        if stream is None:
            check(L.gsf_create(int(device), C.byref(h)))
        else:
            check(L.gsf_create_on_stream(int(device), _vp(int(stream)), C.byref(h)))
        self._h, self.device, self._L = h, int(device), L
        self.options = {}
        # kernel-choice override for A/B runs of the whole test suite (the library itself reads no environment variable)
        v = os.environ.get("GSF_BLOCK_KERNEL")
        if v is not None and v.strip() in ("-1", "0", "1"):
            self.set_option("block_kernel", int(v))

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        check(self._L.gsf_synchronize(self._h))

    def set_option(self, key, value):
        check(self._L.gsf_set_option(self._h, key.encode(), int(value)))
        self.options[key] = int(value)

    def trim(self):
        """release the context's grow-only workspaces (gsf_trim)"""
        check(self._L.gsf_trim(self._h))

    def set_sim3_rows(self, fit_rows, global_config=None):
        """Which rows feed the Sim3 fit of the fused chains on this context: "reference" (main_process_gui's choice, EKFGPSSLAM.py:973-998,
        with min_samples / max_gps_gap_threshold / max_initial_duration from the CONFIG dict) or "all" (every valid row)."""
        if fit_rows in ("all", 0, False, None):
            check(self._L.gsf_set_sim3_rows(self._h, 0, 0, 0.0, 0.0))
        elif fit_rows in ("reference", 1, True):
            r, t = global_config["sim3_ransac"], global_config["time_alignment"]
            check(self._L.gsf_set_sim3_rows(self._h, 1, int(r["min_samples"]), float(t["max_gps_gap_threshold"]), float(r["max_initial_duration"])))
        else:
            raise ValueError(f"fit_rows must be 'reference' or 'all', got {fit_rows!r}")

    def timer_start(self):
        check(self._L.gsf_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        check(self._L.gsf_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """Lazily created context on device 0 with its own stream (used by the drop-in host-array functions)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("GSF_DEVICE", "0")))
    return _default_ctx


def hptr(a):
    """host pointer of a C-contiguous numpy array (or None)"""
    return None if a is None else _vp(a.ctypes.data)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)
