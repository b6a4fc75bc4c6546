/* examples/run_fusion.c -- steps 1-6 of the reference's main_process_gui (EKFGPSSLAM.py:959-1033) for B trajectories from plain C99: what a
 * cgo / JNI / FFI binding of gsf_run_fusion_batch looks like, without Python or torch in the process.
 *
 *   run_fusion IN OUT [seed]
 *
 * IN:  int64 B, int64 N, int64 total; ts[B*N], pos[B*N*3], quat[B*N*4] f64 (what load_slam_trajectory returns, :959);
 *      gps_offsets[B+1] i64, gps_t[total] f64, gps_llh[total*3] f64 (stamp and columns 1, 2, 3 of the GNSS text file, read as lat, lon, alt, :258).
 * OUT: R[B*9], t[B*3], s[B], pos_out[B*N*3], quat_out[B*N*4], err_stats[3*B*4] f64; status[B], n_inliers[B], zone[B], run_status[B] i32; gps_keep[total] u8.
 * Every trajectory's generator is np.random.seed(seed + b) (default seed 0): the reference's CONFIG (:22-71) throughout, all max_trials drawn.
 * Exit code 0, or 1 with the library's message on stderr.  tests/test_c_consumer.py builds it and compares with the Python route.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsf.h"

static int fail(const char *what)
{
    char msg[512];
    gsf_last_error(msg, (int)sizeof msg);
    fprintf(stderr, "run_fusion: %s: %s\n", what, msg);
    return 1;
}

static void *xread(FILE *f, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "run_fusion: short input\n"); exit(2); }
    return p;
}

/* np.random.seed(s): Knuth's LCG over the 624 words of MT19937, position 624 (numpy/random/src/mt19937/mt19937.c: mt19937_seed) */
static void np_seed(uint32_t s, uint32_t *state)
{
    for (int i = 0; i < 624; ++i) { state[i] = s; s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1); }
    state[624] = 624;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: run_fusion IN OUT [seed]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int64_t hdr[3];
    if (fread(hdr, sizeof hdr[0], 3, f) != 3 || hdr[0] < 0 || hdr[1] < 0 || hdr[2] < 0) { fprintf(stderr, "run_fusion: bad header\n"); return 2; }
    const int64_t B = hdr[0], N = hdr[1], total = hdr[2];
    const size_t P = (size_t)B * (size_t)N, nb = (size_t)B, T = (size_t)total;
    double *ts = xread(f, P * 8), *pos = xread(f, P * 24), *quat = xread(f, P * 32);
    int64_t *off = xread(f, (nb + 1) * 8);
    double *gt = xread(f, T * 8), *llh = xread(f, T * 24);
    fclose(f);
    const uint32_t seed = argc > 3 ? (uint32_t)strtoul(argv[3], NULL, 10) : 0u;

    gsf_run_config cfg;                                    /* EKFGPSSLAM.py:22-71 */
    memset(&cfg, 0, sizeof cfg);
    {
        const double p0[7] = { 0.1, 0.1, 0.1, 0.01, 0.01, 0.01, 0.01 }, q[7] = { 0.1, 0.1, 0.7, 0.01, 0.01, 0.01, 0.01 }, r[3] = { 0.2, 0.2, 0.2 };
        memcpy(cfg.ekf.initial_cov_diag, p0, sizeof p0); memcpy(cfg.ekf.process_noise_diag, q, sizeof q); memcpy(cfg.ekf.meas_noise_diag, r, sizeof r);
        cfg.ekf.sharp_turn_yaw_rate_threshold_deg_per_sec = 45.0; cfg.ekf.default_ekf_transition_steps_on_sharp_turn = 0;
        cfg.gps_filter.enabled = 1; cfg.gps_filter.use_sliding_window = 1; cfg.gps_filter.window_duration_seconds = 15.0; cfg.gps_filter.window_step_factor = 0.5;
        cfg.gps_filter.polynomial_degree = 2; cfg.gps_filter.min_samples = 6; cfg.gps_filter.residual_threshold_meters = 10.0; cfg.gps_filter.max_trials = 50;
        cfg.sim3_min_samples = 4; cfg.sim3_residual_threshold = 4.0; cfg.sim3_max_trials = 1000; cfg.sim3_min_inliers_needed = 4; cfg.sim3_max_initial_duration = 180.0;
        cfg.max_gps_gap_threshold = 5.0; cfg.eval_skip_seconds = 5.0;
    }
    if (gsf_device_count() < 1) { fprintf(stderr, "run_fusion: no HIP device (there is no CPU fallback)\n"); return 1; }
    gsf_ctx *ctx = NULL;
    if (gsf_create(0, &ctx)) return fail("gsf_create");

    uint32_t *mt = malloc(nb * 625 * 4 + 4);
    double *R = malloc(nb * 72 + 8), *t = malloc(nb * 24 + 8), *s = malloc(nb * 8 + 8), *po = malloc(P * 24 + 8), *qo = malloc(P * 32 + 8);
    double *utm = malloc(T * 24 + 8), *al = malloc(P * 24 + 8), *err = malloc(nb * 96 + 8);
    int32_t *st = malloc(nb * 4 + 4), *ni = malloc(nb * 4 + 4), *zone = malloc(nb * 4 + 4), *south = malloc(nb * 4 + 4), *rs = malloc(nb * 4 + 4);
    uint8_t *keep = malloc(T + 1), *va = malloc(P + 1);
    if (!mt || !R || !t || !s || !po || !qo || !utm || !al || !err || !st || !ni || !zone || !south || !rs || !keep || !va) { fprintf(stderr, "run_fusion: out of memory\n"); return 2; }
    for (size_t b = 0; b < nb; ++b) np_seed(seed + (uint32_t)b, mt + b * 625);
    if (gsf_run_fusion_batch(ctx, ts, pos, quat, B, N, gt, llh, off, &cfg, mt, R, t, s, po, qo, st, ni, zone, south, utm, keep, al, va, NULL, err, rs, NULL, NULL))
        return fail("gsf_run_fusion_batch");
    gsf_destroy(ctx);

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    fwrite(R, 8, nb * 9, f); fwrite(t, 8, nb * 3, f); fwrite(s, 8, nb, f); fwrite(po, 8, P * 3, f); fwrite(qo, 8, P * 4, f); fwrite(err, 8, nb * 12, f);
    fwrite(st, 4, nb, f); fwrite(ni, 4, nb, f); fwrite(zone, 4, nb, f); fwrite(rs, 4, nb, f); fwrite(keep, 1, T, f);
    fclose(f);
    int64_t failed = 0, kept = 0;
    for (size_t b = 0; b < nb; ++b) failed += rs[b] != 0;
    for (size_t i = 0; i < T; ++i) kept += keep[i];
    printf("%s\nran %lld trajectories x %lld poses, %lld GNSS fixes (%lld kept); %lld runs the reference would have aborted\n", gsf_version(), (long long)B, (long long)N,
           (long long)total, (long long)kept, (long long)failed);
    free(ts); free(pos); free(quat); free(off); free(gt); free(llh); free(mt); free(R); free(t); free(s); free(po); free(qo); free(utm); free(al); free(err);
    free(st); free(ni); free(zone); free(south); free(rs); free(keep); free(va);
    return 0;
}
