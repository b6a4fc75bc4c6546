/*
 * gsf.h -- C ABI of libgsf.so: the MI355X (gfx950) GPS<->SLAM trajectory-fusion hot path.
 *
 * The reference (A2ureeE/GPS-optimize-SLAM) has no FFI: its hot path sits behind
 * module-level Python functions of EKFGPSSLAM.py.  Each entry point below names the
 * reference function (file:line) whose arithmetic it replaces; the Python module
 * gps_optimize_slam_amd.ekfgpsslam binds them with ctypes and re-exports the reference's
 * own function names (INTEGRATION.md shows the stub a maintainer would add).
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is caller-owned, the library never
 *     retains or frees it.  All floating-point arrays are IEEE float64 and results are those of
 *     float64 arithmetic (gsf_set_option "k2b_screen" describes the one internal exception).  Quaternions are
 *     scalar-last [x,y,z,w] (SciPy convention, as in the reference's TUM files).
 *   - `*_dev` entry points take DEVICE pointers and are asynchronous on the context's
 *     HIP stream; the matching host-pointer entry points copy in/out and synchronise.
 *   - return value: 0 = GSF_OK, otherwise a gsf_error; gsf_last_error() gives the
 *     thread-local message.  Per-item results (the reference's `None` returns, outage
 *     bookkeeping) come back in `status` arrays, never as failures of the call.
 *   - there is NO CPU fallback: without a HIP device gsf_create fails.
 */
#ifndef GSF_H
#define GSF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSF_ABI_VERSION 1

#if defined(__GNUC__)
#define GSF_API __attribute__((visibility("default")))
#else
#define GSF_API
#endif

typedef enum {
    GSF_OK = 0,
    GSF_ERR_INVALID_ARG = 1,
    GSF_ERR_HIP = 2,
    GSF_ERR_NO_DEVICE = 3,
    GSF_ERR_UNSUPPORTED = 4
} gsf_error;

/* batch memory layouts of the fusion kernels */
typedef enum {
    /* trajectory-major AoS (what B stacked TUM files look like):
       ts[B][N] pos[B][N][3] quat[B][N][4] gps[B][N][3] valid[B][N], outputs alike */
    GSF_LAYOUT_TRAJ_MAJOR = 0,
    /* time-major SoA (trajectory index fastest: every wave access is one contiguous 512-B row):
       ts[N][B] pos[N][3][B] quat[N][4][B] gps[N][3][B] valid[N][B], outputs alike */
    GSF_LAYOUT_TIME_MAJOR = 1
} gsf_layout;

/* CONFIG['ekf'] + CONFIG['rts_decision'] of the reference (EKFGPSSLAM.py:24-29, :67-70) */
typedef struct {
    double initial_cov_diag[7];        /* :25 */
    double process_noise_diag[7];      /* :26  per second, used as variances */
    double meas_noise_diag[3];         /* :27  used as variances (Q4) */
    double sharp_turn_yaw_rate_threshold_deg_per_sec;   /* :68 */
    int32_t default_ekf_transition_steps_on_sharp_turn; /* :69 */
    int32_t reserved;
} gsf_ekf_config;

/* CONFIG['gps_filtering_ransac'] / CONFIG['ground_truth_gps_filtering'] of the reference (EKFGPSSLAM.py:39-48, :55-64) */
typedef struct {
    int32_t enabled;                   /* :40  0: the log passes unfiltered and nothing is drawn (:139-141) */
    int32_t use_sliding_window;        /* :41  0: one fit over the whole log (:148-182) */
    double window_duration_seconds;    /* :42 */
    double window_step_factor;         /* :43 */
    int32_t polynomial_degree;         /* :44 */
    int32_t min_samples;               /* :45 */
    double residual_threshold_meters;  /* :46 */
    int32_t max_trials;                /* :47 */
    int32_t max_windows;               /* not a reference value: cap on the windows the device loop visits per log (0 = 4096) */
    double stop_probability;           /* scikit-learn's RANSACRegressor default 0.99 (0 = that) */
} gsf_prefilter_config;

/* everything main_process_gui reads from CONFIG between its step 1 and its step 6 (EKFGPSSLAM.py:22-71) */
typedef struct {
    gsf_ekf_config ekf;                /* :24-29, :67-70 */
    gsf_prefilter_config gps_filter;   /* :39-48 */
    double sim3_residual_threshold;    /* :34 */
    double sim3_max_initial_duration;  /* :37 */
    double max_gps_gap_threshold;      /* :53 */
    double eval_skip_seconds;          /* 5.0, literal in :1018 */
    int32_t sim3_min_samples;          /* :33 */
    int32_t sim3_max_trials;           /* :35 */
    int32_t sim3_min_inliers_needed;   /* :36 */
    int32_t reserved;
} gsf_run_config;

/* per-trajectory outcome of gsf_run_fusion_batch_dev: where the reference's run would have stopped */
#define GSF_RUN_OK 0
#define GSF_RUN_GPS_EMPTY 1            /* no GNSS row passes the lat/lon range mask: load_gps_data raises ValueError (EKFGPSSLAM.py:264) */
#define GSF_RUN_GPS_FEW 2              /* fewer than 2 fixes left after the pre-filter: ValueError (:283, :967) */
#define GSF_RUN_PREFILTER_UNHANDLED 4  /* the device pre-filter does not cover this log (unsorted stamps, scikit-learn's sampler outside its
                                          permutation range 0.01 < min_samples / n < 0.99, more than max_windows windows): NaN outputs; the
                                          caller runs that track through the host route from its saved generator state */
#define GSF_RUN_SIM3_FAILED 8          /* ValueError of the row choice (:975, :997) or RuntimeError of the failed fit (:1003): see status >> 8 */
#define GSF_RUN_BAD_QUAT 16            /* a SLAM quaternion of the track cannot be normalised: SciPy raises in transform_trajectory (:466) */

/* status of a Sim3 fit: GSF_SIM3_NONE <=> the reference returned (None, None, None) */
#define GSF_SIM3_OK 0
#define GSF_SIM3_NONE 1
#define GSF_SIM3_FLAG_VAR0 2          /* var_src < 1e-12 -> scale := 1   (EKFGPSSLAM.py:445-447) */
#define GSF_SIM3_FLAG_SMALL_SCALE 4   /* scale <= 1e-6  -> scale := 1   (EKFGPSSLAM.py:450) */
#define GSF_SIM3_FLAG_BAD_INDEX 8     /* a caller-fed sample set named a row outside [0, n): that trial was skipped */
#define GSF_SIM3_FLAG_SVD_FALLBACK 16 /* informational (fused pipeline): the closed form's rotation came from the Jacobi SVD because the
                                         polar iteration declined this cross-covariance (rank-deficient / weakly separated reflection);
                                         same result to ~1e-14, only slower */
/* row choice of the fused chains under gsf_set_sim3_rows mode 1 (main_process_gui, EKFGPSSLAM.py:973-998) */
#define GSF_SIM3_FLAG_FEW_ROWS 32     /* fewer than min_samples time-synchronised rows: the reference raises ValueError (:975, :997);
                                         comes with GSF_SIM3_NONE, NaN outputs, and (robust chain) an untouched generator */
#define GSF_SIM3_FLAG_ROWS_ALL 64     /* informational: the first gap-free segment is shorter than min_samples -> all valid rows (:984-986) */
#define GSF_SIM3_FLAG_ROWS_SEGMENT 128 /* informational: fewer than min_samples rows inside max_initial_duration -> the whole first segment (:993-995) */
/* robust chain with gsf_set_option "ransac_early_exit" 1 (compute_sim3_transform_robust, EKFGPSSLAM.py:404-414) */
#define GSF_SIM3_FLAG_SATURATED 256   /* informational: a trial counted EVERY row of the fit, so by the strict '>' of :413 no later trial could
                                         have replaced it and the trajectory stopped drawing there: R, t, s, inlier mask, n_inliers and
                                         the fused poses are the all-max_trials result bit for bit; only its generator was advanced by
                                         fewer trials (trial_info of gsf_fuse_pipeline_robust_info_batch_dev says how many) */

/* status bits of a fused trajectory */
#define GSF_ST_HAD_OUTAGE 1
#define GSF_ST_RTS_APPLIED 2
#define GSF_ST_SHARP_TURN 4
#define GSF_ST_ENDED_IN_OUTAGE 8
#define GSF_ST_BAD_QUAT 16            /* a zero/NaN SLAM quaternion took the zero-motion branch (:84-86) */

typedef struct gsf_ctx gsf_ctx;       /* opaque: device id + HIP stream + scratch */

/* ---- context ------------------------------------------------------------------------- */
GSF_API const char *gsf_version(void);
GSF_API int gsf_abi_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf; returns its length */
GSF_API int gsf_last_error(char *buf, int n);
/* number of HIP devices visible (0 if none / no driver) */
GSF_API int gsf_device_count(void);
/* creates a context on `device_id` with its own non-blocking stream */
GSF_API int gsf_create(int device_id, gsf_ctx **out);
/* same, but launches on a caller-owned hipStream_t (e.g. torch's current stream); not destroyed by gsf_destroy */
GSF_API int gsf_create_on_stream(int device_id, void *hip_stream, gsf_ctx **out);
GSF_API void gsf_destroy(gsf_ctx *ctx);
GSF_API int gsf_synchronize(gsf_ctx *ctx);
/* Workspaces.  A context owns grow-only device arenas that are kept between calls (no call pays hipMalloc / hipFree once they have
   grown): the kernel workspace (time alignment slabs, the transposed copy of a small time-major batch: 145 B/pose; the robust chain:
   ~54 B/pose + 4 B x max_trials x min_samples per trajectory), the staging arena + pinned mirror of the host-pointer entry points
   (the size of the largest call, up to 32 MB pinned), the float rows of the K2b screen (28 B per row + 25 %) and -- chosen
   automatically for <= 16 MT19937 streams of <= 2 040 rows -- the tape and transition tables of the chip-wide draws (up to 768 MB,
   960 MB with the growth margin; a few MB at the reference's own shape of one stream of 271 rows).  gsf_trim synchronises the stream
   and releases all of them (they grow again on demand); call it before sizing a large allocation to what the device has free.
   A hipGraph captured from calls on this context holds workspace addresses as kernel arguments: after gsf_trim -- and after any call
   that made a workspace grow -- such a graph must be captured again before it is replayed. */
GSF_API int gsf_trim(gsf_ctx *ctx);
/* tuning knobs; keys:
     "duo_kernel"     -1 automatic (default) / 0 never / 1 always: two-wave build of the fused pipeline for small batches of short tracks
     "lane_min_traj"  time-major batches with fewer trajectories than this (default 32768) are transposed and run by the
                      wave-per-trajectory kernel; 0 = always the lane-per-trajectory kernel
     "block_kernel"   -1 / 0 never (what -1 means today) / 1 whenever it applies: workgroup-per-trajectory EKF kernel for 64 < N <= 1024
                      (under gsf_set_sim3_rows mode 1 its launcher marks the rows of the fit with a launch of their own first)
     "tape_draws"     -1 automatic (default): up to 16 MT19937 streams of <= 2040 rows are drawn chip-wide (csrc/gsf_rng_tape.hip) /
                      0 always one wave per stream / 2 tests only (a tape cut short: the one-wave kernel must take over)
     "k2b_screen"     1 (default): the residual counts of the RANSAC hypotheses are screened in packed single precision and re-checked in
                      double inside the rounding band (identical counts) / 0: double throughout.  The band carries 1e-6 m of slack for
                      the double path's own rounding, which covers coordinates up to ~1e9 m in magnitude (UTM: 1e7); rows beyond
                      that should be fitted with the screen off
     "ransac_early_exit"  0 (default): every trajectory of the robust chain draws and scores all max_trials hypotheses, so its generator
                      ends where np.random ends after compute_sim3_transform_robust (EKFGPSSLAM.py:404-405) / 1: a trajectory stops at
                      the first trial that counts every row of its fit -- :413 keeps a trial only on a STRICTLY larger count, so no later
                      trial can change the mask, the count or the final fit; outputs identical bit for bit, GSF_SIM3_FLAG_SATURATED in the
                      status word, the generator left after the round of eight trials that held the deciding one.  A trajectory that
                      never saturates (one GNSS outlier beyond the threshold among its rows is enough) runs all max_trials as before.
                      The Python batch binding switches it ON, the single-track drop-in never uses it
     "ransac_probe_trials"  (default 64) how many trials the early-exit probe draws and scores per trajectory (one wave each) before
                      the wide kernels take the rest of an undecided trajectory's trials
     "prefilter_speculate"  1 (default): the GPS pre-filter chain (gsf_gps_prefilter_chain_*, gsf_gps_prefilter_auto_dev, the whole-run entries)
                      first tries "every axis of the window stops after its first trial" -- three consecutive trials drawn, fitted and
                      counted at once; an axis whose first trial does not end RANSACRegressor's loop goes through the sequential walk
                      from the stream position where it starts.  0: the sequential walk only.  Same kept rows, same generator state.
     "prefilter_miss_batch"  (default 4, 1..64) after the speculative pass has missed at an axis, its first trial says how many trials
                      RANSACRegressor will want unless a later one counts more rows: the sequential walk opens with a batch of that many,
                      at most this value (a poor first sample asks for many; a better one among the next few shrinks that).  Same words.
     "prefilter_first_batch"  (default 1, 1..64) trials the sequential walk draws and scores before its first look at scikit-learn's
                      stopping rule (the batches then double); any value gives the same words, clean logs are fastest with 1
     "synth_variant"  workload of gsf_synth_batch: 0 white SLAM noise (default), 1 random-walk drift
     "ekf_variant"    reserved (0) */
GSF_API int gsf_set_option(gsf_ctx *ctx, const char *key, int64_t value);
/* Which rows feed the Sim3 fit of the fused chains (gsf_fuse_pipeline_*, gsf_fuse_pipeline_robust_*, gsf_run_fusion_batch_*) on this context
   from now on.  A context STARTS in mode 1 with the reference's CONFIG defaults (min_samples 4, max_gps_gap_threshold 5.0 s,
   max_initial_duration 180.0 s; EKFGPSSLAM.py:34, :53, :37), so a caller that sets nothing gets steps 3-5 as main_process_gui runs them;
   the Python binding (batch.py, fit_rows="reference") sets the same mode from its CONFIG dict on every call.
     mode 0            every row with valid, finite GNSS (the operator SURVEY 8(b)/(d) defined for the batch configs)
     mode 1 (default)  what main_process_gui does between its alignment and its fit (EKFGPSSLAM.py:973-998): of the valid rows V (in row
                       order), the rows before the first k with ts[V[k+1]] - ts[V[k]] > max_gps_gap_threshold -- V[k] itself is left out,
                       :981-982 -- or all of V when there is no such k; if those are fewer than min_samples: all of V (:984-986); else
                       the rows of that segment with ts <= ts[V[0]] + max_initial_duration, or the whole segment when fewer than
                       min_samples of them remain (:988-996).  Fewer than min_samples valid rows: the reference raises ValueError
                       (:975, :997) -> GSF_SIM3_NONE | GSF_SIM3_FLAG_FEW_ROWS in the status word.
   min_samples = CONFIG['sim3_ransac']['min_samples'], the other two CONFIG['time_alignment'] / CONFIG['sim3_ransac'] values. */
GSF_API int gsf_set_sim3_rows(gsf_ctx *ctx, int32_t mode, int32_t min_samples, double max_gps_gap_threshold, double max_initial_duration);
/* The same choice on its own, for B trajectories of N rows (offsets == NULL) or ragged ones (rows offsets[b]..offsets[b+1], N ignored):
   row_mask[total] = 1 on the chosen rows, n_rows[b] their number (-1 where the reference raises ValueError), status[b] (may be NULL) the
   GSF_SIM3_FLAG_FEW_ROWS / _ROWS_ALL / _ROWS_SEGMENT bit.  A row is "valid" when valid[i] != 0 and -- if gps != NULL -- its fix is
   free of NaN (what the fused chains use).  Device pointers, asynchronous. */
GSF_API int gsf_sim3_fit_rows_batch_dev(gsf_ctx *ctx, const double *ts, const double *gps, const uint8_t *valid, const int64_t *offsets,
                                        int64_t B, int64_t N, int32_t min_samples, double max_gps_gap_threshold,
                                        double max_initial_duration, uint8_t *row_mask, int32_t *n_rows, int32_t *status);
GSF_API int gsf_sim3_fit_rows_batch(gsf_ctx *ctx, const double *ts, const double *gps, const uint8_t *valid, const int64_t *offsets,
                                    int64_t B, int64_t N, int32_t min_samples, double max_gps_gap_threshold, double max_initial_duration,
                                    uint8_t *row_mask, int32_t *n_rows, int32_t *status);                     /* host arrays */
/* opens / closes a HIP-event bracket on the context's stream; gsf_timer_stop returns the elapsed ms */
GSF_API int gsf_timer_start(gsf_ctx *ctx);
GSF_API int gsf_timer_stop(gsf_ctx *ctx, float *elapsed_ms);

/* ---- K1: WGS84 -> UTM  (replaces pyproj Proj(...)(lons, lats), EKFGPSSLAM.py:266-271) ------ */
/* zone/hemisphere pick of auto_utm_projection (EKFGPSSLAM.py:127-134), one per trajectory:
   zone = int((mean(lon)+180)//6+1), south = mean(lat) < 0.  offsets: int64[B+1] into lat/lon. */
GSF_API int gsf_utm_zone_batch_dev(gsf_ctx *ctx, const double *lat_deg, const double *lon_deg, const int64_t *offsets,
                           int64_t B, int32_t *zone, int32_t *south);
/* forward projection of B ragged trajectories, each in its own zone; invalid input rows
   (EKFGPSSLAM.py:259: |lat|>90, |lon|>180, lat==0, lon==0) produce NaN */
GSF_API int gsf_utm_forward_batch_dev(gsf_ctx *ctx, const double *lat_deg, const double *lon_deg, const int64_t *offsets,
                              const int32_t *zone, const int32_t *south, int64_t B, double *easting, double *northing);
/* inverse projection (utm_to_wgs84, EKFGPSSLAM.py:291-296) */
GSF_API int gsf_utm_inverse_batch_dev(gsf_ctx *ctx, const double *easting, const double *northing, const int64_t *offsets,
                              const int32_t *zone, const int32_t *south, int64_t B, double *lat_deg, double *lon_deg);
/* host-pointer, single-zone convenience forms used by the drop-in load_gps_data / utm_to_wgs84 */
GSF_API int gsf_utm_forward(gsf_ctx *ctx, const double *lat_deg, const double *lon_deg, int64_t n, int32_t zone, int32_t south,
                    double *easting, double *northing);
GSF_API int gsf_utm_inverse(gsf_ctx *ctx, const double *easting, const double *northing, int64_t n, int32_t zone, int32_t south,
                    double *lat_deg, double *lon_deg);

/* The whole geodesy slice of load_gps_data (EKFGPSSLAM.py:258-271) for B ragged GNSS logs in one launch: rows llh[total][3] =
   (lat deg, lon deg, alt m) -- columns 1,2,3 of the reference's text file -- are masked (:259-264), each log's zone / hemisphere
   is picked from the means over its valid rows (:131-133), and rows utm_rows[total][3] = (E, N, alt) come back (:271).  Rows the
   reference drops are NaN rows here (a device array cannot shrink); zone[b] = 0 for a log without a valid row. */
GSF_API int gsf_gps_rows_to_utm_batch_dev(gsf_ctx *ctx, const double *llh, const int64_t *offsets, int64_t B, double *utm_rows, int32_t *zone,
                                          int32_t *south);
GSF_API int gsf_gps_rows_to_utm_batch(gsf_ctx *ctx, const double *llh, const int64_t *offsets, int64_t B, double *utm_rows, int32_t *zone,
                                      int32_t *south);                                   /* the same with host arrays */

/* WGS84 geodetic -> local East-North-Up about a per-trajectory origin ref_llh[B][3] = (lat0 deg, lon0 deg, h0 m).  Offered in
   addition to UTM: the reference's pipeline projects with UTM (EKFGPSSLAM.py:266-271); BASELINE.json words the step as
   "WGS84 -> local ENU". */
GSF_API int gsf_geodetic_to_enu_batch_dev(gsf_ctx *ctx, const double *lat_deg, const double *lon_deg, const double *alt, const int64_t *offsets,
                                          const double *ref_llh, int64_t B, double *east, double *north, double *up);
GSF_API int gsf_geodetic_to_enu_batch(gsf_ctx *ctx, const double *lat_deg, const double *lon_deg, const double *alt, const int64_t *offsets,
                                      const double *ref_llh, int64_t B, double *east, double *north, double *up);   /* host arrays */

/* ---- next-3: GPS outlier pre-filter (filter_gps_outliers_ransac, EKFGPSSLAM.py:136-247) ------------------------------------
   One problem = one RANSACRegressor.fit of the reference (one window x one coordinate axis): rows offsets[p]..offsets[p+1] of
   (t, y).  sample_idx[P][max_trials][min_samples] are the sample sets of every trial, drawn by the host with scikit-learn's
   sample_without_replacement on NumPy's legacy RNG (the call of sklearn/linear_model/_ransac.py), so a seeded run reproduces the
   reference.  Per problem: inlier_mask over its rows (|y - poly(t)| <= residual_threshold for the accepted model), n_trials (the
   number of sample sets scikit-learn's loop would have consumed: acceptance rule + dynamic trial count with stop_probability),
   n_inliers, status (bit 0: no consensus set -- the reference's ValueError; bit 1: a fed sample set named a row outside the problem
   and was skipped).  degree 1..8, min_samples <= 64, max_trials <= 2^20; inside degree <= 3, min_samples <= 16, max_trials <= 1024
   one thread scores one trial in one pass, beyond that a slower kernel strides the trials (per-trial results in the context's
   workspace). */
GSF_API int gsf_ransac_poly_batch_dev(gsf_ctx *ctx, const double *t, const double *y, const int64_t *offsets, int64_t P,
                                      const int32_t *sample_idx, int32_t max_trials, int32_t min_samples, int32_t degree,
                                      double residual_threshold, double stop_probability, uint8_t *inlier_mask, int32_t *n_trials,
                                      int32_t *n_inliers, int32_t *status);
GSF_API int gsf_ransac_poly_batch(gsf_ctx *ctx, const double *t, const double *y, const int64_t *offsets, int64_t P,
                                  const int32_t *sample_idx, int32_t max_trials, int32_t min_samples, int32_t degree,
                                  double residual_threshold, double stop_probability, uint8_t *inlier_mask, int32_t *n_trials,
                                  int32_t *n_inliers, int32_t *status);                  /* the same with host arrays */

/* The WHOLE pre-filter of B GNSS logs as one device chain, draws included.  Log b = rows offsets[b]..offsets[b+1] of t[] and
   pos[][3] (UTM E, N, alt); its windows are rows win_rows[2w], win_rows[2w+1] (relative to the log's first row; found by the host
   from the stamps, ref :205-206; one window = the whole log for the global mode, ref :148) for w in win_offsets[b]..win_offsets[b+1].
   Per window, per axis in the reference's order: max_trials sample sets from the log's legacy MT19937 stream (mt_state[B][625],
   in/out) exactly as scikit-learn's sample_without_replacement draws them, RANSACRegressor's acceptance walk, the stream rewound to
   where n_trials_ draws leave it; a failing axis drops its window and consumes nothing further for it (ref :228-229).
   keep[total] = OR over successful windows of the AND over axes; win_status[w] = 0 ok / 1 no consensus / 2 fewer rows than
   min_samples (not processed) / 3 not handled; log_status[b] = 2 when a window's min_samples/n is 0.01 or less -- scikit-learn then
   samples by tracking selection, which is not restated here -- and the caller has to take the fed-sample route
   (gsf_ransac_poly_batch_dev) for that log from its saved generator state.  (A window of exactly min_samples rows, ratio 1, is handled:
   scikit-learn's reservoir sampling then returns rows 0 .. min_samples-1 without a draw.)  max_trials <= 1024, min_samples <= 16, degree <= 3. */
GSF_API int gsf_gps_prefilter_chain_dev(gsf_ctx *ctx, const double *t, const double *pos, const int64_t *offsets, int64_t B,
                                        const int32_t *win_rows, const int64_t *win_offsets, int32_t max_window_rows, int32_t max_trials,
                                        int32_t min_samples, int32_t degree, double residual_threshold, double stop_probability,
                                        uint32_t *mt_state, uint8_t *keep, int32_t *win_status, int32_t *log_status);
GSF_API int gsf_gps_prefilter_chain(gsf_ctx *ctx, const double *t, const double *pos, const int64_t *offsets, int64_t B,
                                    const int32_t *win_rows, const int64_t *win_offsets, int32_t max_window_rows, int32_t max_trials,
                                    int32_t min_samples, int32_t degree, double residual_threshold, double stop_probability,
                                    uint32_t *mt_state, uint8_t *keep, int32_t *win_status, int32_t *log_status);
/* filter_gps_outliers_ransac AS A WHOLE for B logs (EKFGPSSLAM.py:136-247), no host step: the windows are found on the device from the
   stamps the way the reference walks them (:199-234: [w0, w0 + duration) advanced by duration x step_factor, one extra tail window, windows
   with fewer than min_samples rows skipped), or one global window (:148-182; a failing fit passes the log unfiltered), plus the
   reference's early-outs (disabled, fewer rows than min_samples: everything kept, nothing drawn).  Sorted stamps only (log_status 3
   otherwise); max_log_rows >= the longest log.  log_info[B][2] (may be NULL) = { windows fitted, windows that found a consensus }.
   Draws, acceptance walk, generator handling: as gsf_gps_prefilter_chain_dev. */
GSF_API int gsf_gps_prefilter_auto_dev(gsf_ctx *ctx, const double *t, const double *pos, const int64_t *offsets, int64_t B, int32_t max_log_rows,
                                       const gsf_prefilter_config *filter, uint32_t *mt_state, uint8_t *keep, int32_t *log_status,
                                       int32_t *log_info);

/* ---- K2: Sim3 / Umeyama (compute_sim3_transform, EKFGPSSLAM.py:428-459) ------------------- */
/* B ragged point sets: src/dst are [total][3]; offsets int64[B+1].  Optional `mask` (uint8[total], may be NULL)
   selects the rows that take part (used for "valid GNSS only" fits).  Outputs R[B][9] (row-major), t[B][3],
   s[B], status[B]. */
GSF_API int gsf_sim3_umeyama_batch_dev(gsf_ctx *ctx, const double *src, const double *dst, const uint8_t *mask,
                               const int64_t *offsets, int64_t B, double *R, double *t, double *s, int32_t *status);
/* B equal-size windows of W point pairs each (BASELINE config C4: sliding-window re-alignment, 1 M windows of 50 pairs):
   src/dst are [B][W][3], mask (may be NULL) [B][W].  Same results as gsf_sim3_umeyama_batch_dev on offsets b*W; the streaming
   moments pass and the per-window 3x3 SVD run as two launches (a 192-byte record per window in the context's workspace). */
GSF_API int gsf_sim3_umeyama_windows_dev(gsf_ctx *ctx, const double *src, const double *dst, const uint8_t *mask, int64_t B, int32_t W,
                                         double *R, double *t, double *s, int32_t *status);
GSF_API int gsf_sim3_umeyama_windows(gsf_ctx *ctx, const double *src, const double *dst, const uint8_t *mask, int64_t B, int32_t W,
                                     double *R, double *t, double *s, int32_t *status);               /* host arrays */
GSF_API int gsf_sim3_umeyama_batch(gsf_ctx *ctx, const double *src, const double *dst, const uint8_t *mask,
                           const int64_t *offsets, int64_t B, double *R, double *t, double *s, int32_t *status);

/* ---- K2b: RANSAC wrapper (compute_sim3_transform_robust, EKFGPSSLAM.py:389-426) ------------ */
/* sample_idx: int32[B][trials][min_samples], drawn BY THE HOST with the reference's RNG call
   (np.random.choice(n, min_samples, replace=False), :405) so results are reproducible against it.
   inlier_mask: uint8[total] out; n_inliers: int32[B] out (best count, -1 if no trial succeeded). */
GSF_API int gsf_sim3_ransac_batch_dev(gsf_ctx *ctx, const double *src, const double *dst, const int64_t *offsets, int64_t B,
                              const int32_t *sample_idx, int32_t trials, int32_t min_samples, double residual_threshold,
                              int32_t min_inliers_needed, double *R, double *t, double *s, int32_t *status,
                              uint8_t *inlier_mask, int32_t *n_inliers);
/* the same with the number of rows of src / dst (= offsets[B]) known to the HOST: the library can then keep the rows a second time
   as floats in its workspace and SCREEN the residual counts of the hypotheses in packed single precision, re-checking in double
   every row inside the rounding band -- the counts, masks and fits are those of the double count, in ~0.6 of the time
   (gsf_set_option "k2b_screen" 0 switches the screen off).  The host-pointer forms use it by themselves. */
GSF_API int gsf_sim3_ransac_batch_rows_dev(gsf_ctx *ctx, const double *src, const double *dst, const int64_t *offsets, int64_t total_rows,
                                           int64_t B, const int32_t *sample_idx, int32_t trials, int32_t min_samples,
                                           double residual_threshold, int32_t min_inliers_needed, double *R, double *t, double *s,
                                           int32_t *status, uint8_t *inlier_mask, int32_t *n_inliers);
GSF_API int gsf_sim3_ransac_batch(gsf_ctx *ctx, const double *src, const double *dst, const int64_t *offsets, int64_t B,
                          const int32_t *sample_idx, int32_t trials, int32_t min_samples, double residual_threshold,
                          int32_t min_inliers_needed, double *R, double *t, double *s, int32_t *status,
                          uint8_t *inlier_mask, int32_t *n_inliers);

/* The same with the draws made ON THE DEVICE from NumPy's legacy generator state (host pointers; mt_state[B][625] uint32 in/out =
   np.random.get_state()[1:3] per set, see gsf_mt19937_*): one call instead of `trials` host-side np.random.choice calls, and the
   caller's generator ends where the reference leaves it. */
GSF_API int gsf_sim3_ransac_mt_batch(gsf_ctx *ctx, const double *src, const double *dst, const int64_t *offsets, int64_t B, uint32_t *mt_state,
                                     int32_t trials, int32_t min_samples, double residual_threshold, int32_t min_inliers_needed, double *R,
                                     double *t, double *s, int32_t *status, uint8_t *inlier_mask, int32_t *n_inliers);

/* ---- the reference's random draws on the device (np.random.choice(n, k, replace=False), EKFGPSSLAM.py:405) ---------- */
/* state: uint32[B][625] = NumPy's legacy MT19937 state per stream, key[624] then pos -- exactly np.random.get_state()[1:3], so a
   host can hand over its global generator (B = 1) or seed one stream per trajectory.
   gsf_mt19937_seed_batch_dev: np.random.seed(seeds[b]) for every stream.
   gsf_mt19937_choice_batch_dev: sample_idx[b][trial][0..k) = RandomState.permutation(n_population[b])[:k] for `trials`
   consecutive trials of stream b (what np.random.choice(n, k, replace=False) draws; also scikit-learn's
   sample_without_replacement for 0.01 < k/n < 0.99), the state advanced exactly as NumPy advances it.  Streams with
   n_population[b] < k are left untouched (the reference returns before drawing, :395-397).  n_population[b] <= 28000, k <= 64.
   gsf_mt19937_choice_bounded_batch_dev: the same, with n_max >= every n_population[b] known to the HOST (0 = unknown).  The
   populations live in device memory, so only with this bound can the library size the workspace of the chip-wide route
   (csrc/gsf_rng_tape.hip): up to 16 streams of n_max <= 2040 are then drawn by thousands of waves instead of one wave per stream
   (the reference's own case is ONE stream); results and final states are identical either way.  A stream whose population
   exceeds n_max is drawn by the one-wave route. */
GSF_API int gsf_mt19937_seed_batch_dev(gsf_ctx *ctx, const uint32_t *seeds, int64_t B, uint32_t *state);
GSF_API int gsf_mt19937_choice_batch_dev(gsf_ctx *ctx, uint32_t *state, const int32_t *n_population, int64_t B, int32_t trials,
                                         int32_t k, int32_t *sample_idx);
GSF_API int gsf_mt19937_choice_bounded_batch_dev(gsf_ctx *ctx, uint32_t *state, const int32_t *n_population, int32_t n_max, int64_t B,
                                                 int32_t trials, int32_t k, int32_t *sample_idx);

/* ---- K3: apply Sim3 (transform_trajectory, EKFGPSSLAM.py:461-467) -------------------------- */
/* pos[total][3], quat[total][4]; per-trajectory R[B][9], t[B][3], s[B].  A zero-norm quaternion (SciPy raises
   ValueError there) yields NaN quaternion output and sets bad_quat[b] (int32[B], may be NULL). */
GSF_API int gsf_apply_sim3_batch_dev(gsf_ctx *ctx, const double *pos, const double *quat, const int64_t *offsets, int64_t B,
                             const double *R, const double *t, const double *s, double *pos_out, double *quat_out,
                             int32_t *bad_quat);
GSF_API int gsf_apply_sim3_batch(gsf_ctx *ctx, const double *pos, const double *quat, const int64_t *offsets, int64_t B,
                         const double *R, const double *t, const double *s, double *pos_out, double *quat_out,
                         int32_t *bad_quat);

/* ---- K4: EKF + per-outage RTS (apply_ekf_correction, EKFGPSSLAM.py:831-935, after its :847 alignment) -- */
/* B trajectories of N poses each.  ts/pos/quat = ORIGINAL SLAM track, gps/valid = GNSS time-aligned to the SLAM
   stamps (NaN allowed), init_pos[B][3]/init_quat[B][4] = row 0 of the Sim3-aligned track.
   Outputs pos_out/quat_out in the same layout, status[B] (GSF_ST_* bits). */
GSF_API int gsf_ekf_fuse_batch_dev(gsf_ctx *ctx, int32_t layout, const double *ts, const double *pos, const double *quat,
                           const double *gps, const uint8_t *valid, const double *init_pos, const double *init_quat,
                           const gsf_ekf_config *cfg, int64_t B, int64_t N, double *pos_out, double *quat_out,
                           int32_t *status);
GSF_API int gsf_ekf_fuse_batch(gsf_ctx *ctx, int32_t layout, const double *ts, const double *pos, const double *quat,
                       const double *gps, const uint8_t *valid, const double *init_pos, const double *init_quat,
                       const gsf_ekf_config *cfg, int64_t B, int64_t N, double *pos_out, double *quat_out,
                       int32_t *status);

/* ---- fused pipeline: Umeyama on the chosen rows -> Sim3 of pose 0 -> EKF+RTS, one launch chain --------- */
/* (steps 3-5 of main_process_gui, EKFGPSSLAM.py:1002-1010, with the plain fit of :428 instead of RANSAC; the rows of the fit are
   every valid row or the reference's choice of :973-998 -- gsf_set_sim3_rows).
   Trajectories whose fit is None get status GSF_ST_* | (GSF_SIM3_NONE << 8) and NaN outputs. */
GSF_API int gsf_fuse_pipeline_batch_dev(gsf_ctx *ctx, int32_t layout, const double *ts, const double *pos, const double *quat,
                                const double *gps, const uint8_t *valid, const gsf_ekf_config *cfg, int64_t B, int64_t N,
                                double *R, double *t, double *s, double *pos_out, double *quat_out, int32_t *status);
GSF_API int gsf_fuse_pipeline_batch(gsf_ctx *ctx, int32_t layout, const double *ts, const double *pos, const double *quat,
                            const double *gps, const uint8_t *valid, const gsf_ekf_config *cfg, int64_t B, int64_t N,
                            double *R, double *t, double *s, double *pos_out, double *quat_out, int32_t *status);   /* host arrays */

/* ---- the same steps with the reference's ROBUST fit (compute_sim3_transform_robust, EKFGPSSLAM.py:1002, :389-426) ---------- */
/* One device chain, no host round trip: rows with valid finite GNSS (all of them, or the reference's choice of :973-998:
   gsf_set_sim3_rows) -> max_trials hypotheses drawn from each trajectory's
   legacy MT19937 stream (mt_state[B][625], in/out: see gsf_mt19937_*) -> first-best inlier set, final Umeyama on the inliers ->
   Sim3 of pose 0 -> EKF+RTS.  Trajectory-major layout.  Outputs as gsf_fuse_pipeline_batch_dev plus n_inliers[B] (best count,
   -1 if no hypothesis succeeded) and inlier_mask[B][N] (uint8, original row order, may be NULL).  The workspace (about
   53 B/pose + 4 B x max_trials x min_samples per trajectory) is the context's, grown on demand. */
GSF_API int gsf_fuse_pipeline_robust_batch_dev(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, const double *gps,
                                               const uint8_t *valid, const gsf_ekf_config *cfg, int64_t B, int64_t N, int32_t min_samples,
                                               double residual_threshold, int32_t max_trials, int32_t min_inliers_needed, uint32_t *mt_state,
                                               double *R, double *t, double *s, double *pos_out, double *quat_out, int32_t *status,
                                               int32_t *n_inliers, uint8_t *inlier_mask);
/* the same with trial_info[B][2] (int32, may be NULL): { the trial whose inlier set was kept -- the first one with the best count, :413 --
   or -1 when no trial was usable, the number of trials drawn from the trajectory's generator } (max_trials unless
   gsf_set_option "ransac_early_exit" stopped it early; 0 where the reference returns before drawing, :395-397) */
GSF_API int gsf_fuse_pipeline_robust_info_batch_dev(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, const double *gps,
                                                    const uint8_t *valid, const gsf_ekf_config *cfg, int64_t B, int64_t N, int32_t min_samples,
                                                    double residual_threshold, int32_t max_trials, int32_t min_inliers_needed,
                                                    uint32_t *mt_state, double *R, double *t, double *s, double *pos_out, double *quat_out,
                                                    int32_t *status, int32_t *n_inliers, uint8_t *inlier_mask, int32_t *trial_info);
/* host arrays; mt_state[B][625] in/out as in gsf_sim3_ransac_mt_batch (np.random.get_state(): key[624] + pos) */
GSF_API int gsf_fuse_pipeline_robust_batch(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, const double *gps,
                                           const uint8_t *valid, const gsf_ekf_config *cfg, int64_t B, int64_t N, int32_t min_samples,
                                           double residual_threshold, int32_t max_trials, int32_t min_inliers_needed, uint32_t *mt_state,
                                           double *R, double *t, double *s, double *pos_out, double *quat_out, int32_t *status,
                                           int32_t *n_inliers, uint8_t *inlier_mask);

/* ---- steps 1-6 of main_process_gui for B trajectories as ONE device chain (EKFGPSSLAM.py:959-1033) -------------------------------
   In: B SLAM tracks of N poses (ts[B][N], pos[B][N][3], quat[B][N][4]: what load_slam_trajectory returns, :959) and their GNSS logs as
   load_gps_data reads them (:258): fixes gps_offsets[b]..gps_offsets[b+1] of gps_t[total] and gps_llh[total][3] = (col 1, col 2, col 3 of
   the text file: read as lat deg, lon deg, alt m).  total_fixes = gps_offsets[B], max_fixes >= the longest log (both known to the HOST:
   they size the workspace).  mt_state[B][625] in/out: each trajectory's NumPy legacy generator (np.random.get_state(): key[624] + pos);
   the pre-filter of step 1 and the robust fit of step 3 draw from it in the reference's order.
   Chain: lat/lon range mask, zone pick, UTM forward (:259-271) -> sliding-window RANSAC pre-filter (:275, :136-247) -> dynamic_time_alignment
   to the SLAM stamps (:971) -> the rows of the global fit (:973-998; this entry always applies the reference's choice with cfg's values,
   whatever gsf_set_sim3_rows says) -> compute_sim3_transform_robust (:1002; gsf_set_option "ransac_early_exit" applies) ->
   transform_trajectory (:1006) -> apply_ekf_correction (:1010) -> the error metric of step 6 against the primary GPS (:1013-1033).
   Out: R[B][9], t[B][3], s[B], pos_out[B][N][3], quat_out[B][N][4], status[B] (GSF_ST_* | GSF_SIM3_* << 8), n_inliers[B];
   zone[B] / south[B]; gps_utm[total][3] = (E, N, alt), NaN rows where the loader drops the fix; gps_keep[total] = 1 on the fixes that
   survive loader and pre-filter (what load_gps_data returns); aligned[B][N][3] / valid[B][N] = step 2's output; sim3_pos[B][N][3] (may be
   NULL) = step 4's positions; err_stats[3][B][4] = { count, mean, median, RMSE } of raw SLAM / Sim3 / EKF (:1027-1033);
   run_status[B] = GSF_RUN_* (a trajectory on which the reference raises has NaN outputs and its generator where the reference left it);
   gps_llh == NULL: gps_utm is an INPUT -- the logs as load_gps_data's projection left them, (E, N, alt) per fix, NaN easting and northing
   on a fix the loader drops -- and the chain starts at the pre-filter (zone / south are not written and may be NULL);
   inlier_mask[B][N] and trial_info[B][2] as in gsf_fuse_pipeline_robust_info_batch_dev (may be NULL). */
GSF_API int gsf_run_fusion_batch_dev(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, int64_t B, int64_t N,
                                     const double *gps_t, const double *gps_llh, const int64_t *gps_offsets, int64_t total_fixes,
                                     int32_t max_fixes, const gsf_run_config *cfg, uint32_t *mt_state, double *R, double *t, double *s,
                                     double *pos_out, double *quat_out, int32_t *status, int32_t *n_inliers, int32_t *zone, int32_t *south,
                                     double *gps_utm, uint8_t *gps_keep, double *aligned, uint8_t *valid, double *sim3_pos,
                                     double *err_stats, int32_t *run_status, uint8_t *inlier_mask, int32_t *trial_info);
/* the same with host arrays (gps_llh must be given; total_fixes and max_fixes are read from gps_offsets) */
GSF_API int gsf_run_fusion_batch(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, int64_t B, int64_t N, const double *gps_t,
                                 const double *gps_llh, const int64_t *gps_offsets, const gsf_run_config *cfg, uint32_t *mt_state, double *R,
                                 double *t, double *s, double *pos_out, double *quat_out, int32_t *status, int32_t *n_inliers, int32_t *zone,
                                 int32_t *south, double *gps_utm, uint8_t *gps_keep, double *aligned, uint8_t *valid, double *sim3_pos,
                                 double *err_stats, int32_t *run_status, uint8_t *inlier_mask, int32_t *trial_info);

/* ragged forms (trajectories of different lengths): flat [total][C] arrays, trajectory b = rows offsets[b]..offsets[b+1] */
GSF_API int gsf_ekf_fuse_ragged_dev(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, const double *gps,
                                    const uint8_t *valid, const int64_t *offsets, const double *init_pos, const double *init_quat,
                                    const gsf_ekf_config *cfg, int64_t B, double *pos_out, double *quat_out, int32_t *status);
GSF_API int gsf_fuse_pipeline_ragged_dev(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, const double *gps,
                                         const uint8_t *valid, const int64_t *offsets, const gsf_ekf_config *cfg, int64_t B, double *R,
                                         double *t, double *s, double *pos_out, double *quat_out, int32_t *status);
/* the same with host arrays */
GSF_API int gsf_ekf_fuse_ragged(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, const double *gps,
                                const uint8_t *valid, const int64_t *offsets, const double *init_pos, const double *init_quat,
                                const gsf_ekf_config *cfg, int64_t B, double *pos_out, double *quat_out, int32_t *status);
GSF_API int gsf_fuse_pipeline_ragged(gsf_ctx *ctx, const double *ts, const double *pos, const double *quat, const double *gps,
                                     const uint8_t *valid, const int64_t *offsets, const gsf_ekf_config *cfg, int64_t B, double *R,
                                     double *t, double *s, double *pos_out, double *quat_out, int32_t *status);

/* ---- time alignment (dynamic_time_alignment, EKFGPSSLAM.py:325-387) ------------------------------ */
/* B trajectories: SLAM stamps slam_t[slam_offsets[b]..), GNSS fixes gps_t / gps_p[.][3] at gps_offsets (any order, duplicates
   allowed: stable sort + first-of-equal-stamps).  Per gap-free segment (gap > max_gps_gap_threshold splits): not-a-knot cubic
   spline (>= 4 fixes) or linear (2-3 fixes) evaluated at the SLAM stamps inside the segment; aligned[total_slam][3] gets NaN
   elsewhere, valid[total_slam] = 1 where all three components are finite.  max_gps_per_trajectory sizes the per-trajectory staging
   (LDS up to 2560 fixes, a global scratch slab beyond); status[b] (may be NULL) = 1 if trajectory b has more fixes than that and
   was left unaligned. */
GSF_API int gsf_time_align_batch_dev(gsf_ctx *ctx, const double *slam_t, const int64_t *slam_offsets, const double *gps_t,
                                     const double *gps_p, const int64_t *gps_offsets, int64_t B, int32_t max_gps_per_trajectory,
                                     double max_gps_gap_threshold, double *aligned, uint8_t *valid, int32_t *status);
GSF_API int gsf_time_align_batch(gsf_ctx *ctx, const double *slam_t, const int64_t *slam_offsets, const double *gps_t, const double *gps_p,
                                 const int64_t *gps_offsets, int64_t B, double max_gps_gap_threshold, double *aligned, uint8_t *valid,
                                 int32_t *status);
/* The same for a GNSS log that comes straight from gsf_gps_rows_to_utm_batch_dev: rows whose easting AND northing are NaN are the
   rows load_gps_data removes before the projection (EKFGPSSLAM.py:259-264: lat/lon zero or out of range); they are dropped when the
   log is staged, so the alignment sees exactly the fixes the reference's loader hands to dynamic_time_alignment. */
GSF_API int gsf_time_align_loaded_rows_batch_dev(gsf_ctx *ctx, const double *slam_t, const int64_t *slam_offsets, const double *gps_t,
                                                 const double *gps_p, const int64_t *gps_offsets, int64_t B,
                                                 int32_t max_gps_per_trajectory, double max_gps_gap_threshold, double *aligned,
                                                 uint8_t *valid, int32_t *status);

/* ---- error evaluation (main_process_gui step 6, EKFGPSSLAM.py:1013-1033) -------------------- */
/* B trajectories x N poses, trajectory-major.  For every index with valid finite aligned GNSS and ts > ts[0] + skip_seconds:
   error = min over all such candidate fixes of |traj_pos[i] - gps[j]| (cdist + min, :1030-1031).
   stats[B][4] = {count, mean, median, rmse} (:1033); errors[B][N] = per-pose error, NaN where not evaluated. */
GSF_API int gsf_eval_errors_batch_dev(gsf_ctx *ctx, const double *ts, const double *traj_pos, const double *aligned_gps,
                                      const uint8_t *valid, int64_t B, int64_t N, double skip_seconds, double *stats, double *errors);
GSF_API int gsf_eval_errors_batch(gsf_ctx *ctx, const double *ts, const double *traj_pos, const double *aligned_gps, const uint8_t *valid,
                                  int64_t B, int64_t N, double skip_seconds, double *stats, double *errors);

/* ---- helper functions of the reference's EKF surface (API completeness; host pointers, synchronous) ------------ */
/* calculate_relative_pose (EKFGPSSLAM.py:77-92) for n pose pairs; bad[i]=1 where the zero-motion branch (:84-86) was taken */
GSF_API int gsf_relative_pose_batch(gsf_ctx *ctx, const double *p1, const double *q1, const double *p2, const double *q2, int64_t n,
                                    double *delta_pos_local, double *delta_quat, int32_t *bad);
/* quaternion_nlerp (EKFGPSSLAM.py:94-105) for n pairs */
GSF_API int gsf_quaternion_nlerp_batch(gsf_ctx *ctx, const double *q1, const double *q2, const double *weight_q2, int64_t n, double *out);
/* is_sharp_turn_in_segment (EKFGPSSLAM.py:808-826) for B ragged segments: result[b] in {0,1}, max_rate[b] rad/s (may be NULL) */
GSF_API int gsf_is_sharp_turn_batch(gsf_ctx *ctx, const double *quats, const double *stamps, const int64_t *offsets, int64_t B,
                                    double yaw_rate_threshold_rad_per_sec, int32_t *result, double *max_rate);
/* one ExtendedKalmanFilter.process_step (EKFGPSSLAM.py:736-772) in the reference's general dense form: state[7], cov[49],
   Q_per_sec[49], R[9]; gnss_available_prev in {-1 None, 0, 1}; gps_meas NULL = None; override_transition_steps < 0 = None */
GSF_API int gsf_ekf_process_step(gsf_ctx *ctx, double *state, double *cov, const double *process_noise_per_sec, const double *meas_noise,
                                 int32_t *gnss_available_prev, double *gnss_update_weight, int32_t current_transition_steps,
                                 const double *delta_pos_local, const double *delta_quat, const double *gps_meas,
                                 int32_t gnss_is_available, double delta_time, int32_t override_transition_steps, double *pred_state,
                                 double *pred_cov);
/* rts_smoother_segment (EKFGPSSLAM.py:777-803), dense 7x7, for B ragged segments (rows offsets[b]..offsets[b+1]) */
GSF_API int gsf_rts_smoother_segment_batch(gsf_ctx *ctx, const double *states_filt, const double *covs_filt, const double *states_pred,
                                           const double *covs_pred, const int64_t *offsets, int64_t B, double *states_smooth,
                                           double *covs_smooth);

/* ---- multi-GPU collect: RCCL all-gather of the fused poses over xGMI ------------------------------- */
/* The reference is single-process (no collective exists in it: filter state is per ExtendedKalmanFilter instance,
   EKFGPSSLAM.py:842); trajectories shard by contiguous id blocks, one process per GPU, and this is the one exchange: every rank
   receives every rank's fused poses.  RCCL is resolved at run time (dlopen).
   gsf_comm_unique_id: 128-byte ncclUniqueId (rank 0 creates it, the host ships it to the other ranks by any side channel);
   gsf_comm_init_rank: collective over all `world` processes, device = the context's; gsf_comm_destroy frees the communicator. */
/* ncclGetVersion of the librccl.so the library resolved (MAJOR*10000 + MINOR*100 + PATCH; 0 = the symbol is missing) */
GSF_API int gsf_comm_rccl_version(int32_t *version);
GSF_API int gsf_comm_unique_id(uint8_t *id128);
GSF_API int gsf_comm_init_rank(gsf_ctx *ctx, const uint8_t *id128, int32_t world, int32_t rank, void **comm);
GSF_API int gsf_comm_destroy(void *comm);
/* All-gather `count` doubles per rank into recv[world][count], asynchronously on the context's stream, on a communicator from
   gsf_comm_init_rank (or any caller-provided ncclComm_t).  mode 0 = one ncclAllGather; mode 1 = direct exchange (grouped
   ncclSend/ncclRecv with every peer, `chunk_count` doubles at a time; 0 = all at once) so all seven point-to-point xGMI links of
   a GPU carry traffic at once. */
GSF_API int gsf_allgather_poses(gsf_ctx *ctx, void *nccl_comm, const double *send, double *recv, int64_t count, int32_t mode,
                                int64_t chunk_count);

/* ---- layout helpers + synthetic workload (bench / tests) ------------------------------------------ */
/* [B][N][C] <-> [N][C][B] transposes of float64 (C = 1,3,4) and uint8 (C = 1) arrays, on device */
GSF_API int gsf_transpose_to_time_major_dev(gsf_ctx *ctx, const void *src, void *dst, int64_t B, int64_t N, int32_t C, int32_t elem_bytes);
GSF_API int gsf_transpose_to_traj_major_dev(gsf_ctx *ctx, const void *src, void *dst, int64_t B, int64_t N, int32_t C, int32_t elem_bytes);
/* deterministic KITTI-04-shaped synthetic batch, generated on device straight into `layout`;
   trajectory ids [traj0, traj0+B).  Integer counter-based RNG + polynomial / rational curves only (no libm call), so the values do
   not depend on a math library and a shard generated with traj0 = k equals rows k.. of the full batch bit for bit. */
/* the same trajectories with the GNSS side as a ragged GEODETIC log (fixes at their own stamps, rows (lat, lon, alt) about
   49.0336 N 8.3950 E, outages = missing fixes): the input of the device chain K1 -> time alignment -> fit -> EKF.  Two passes:
   counts != NULL sizes (counts[B] int64 fixes per trajectory), gps_offsets != NULL writes rows at gps_offsets[b]. */
GSF_API int gsf_synth_geodetic_batch_dev(gsf_ctx *ctx, uint64_t seed, int64_t traj0, int64_t B, int64_t N, double *ts, double *pos,
                                         double *quat, int64_t *counts, const int64_t *gps_offsets, double *gps_t, double *gps_llh);
GSF_API int gsf_synth_batch_dev(gsf_ctx *ctx, int32_t layout, uint64_t seed, int64_t traj0, int64_t B, int64_t N, double *ts,
                        double *pos, double *quat, double *gps, uint8_t *valid, double *init_pos, double *init_quat);

#ifdef __cplusplus
}
#endif
#endif /* GSF_H */
