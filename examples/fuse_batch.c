/* examples/fuse_batch.c -- a plain C99 consumer of the drop-in boundary (include/gsf.h, libgsf.so): what a cgo / JNI / FFI binding of
 * another host language does, without Python or torch in the process.
 *
 *   fuse_batch IN OUT [all]
 *
 * IN:  int64 B, int64 N, then ts[B*N] f64, pos[B*N*3] f64, quat[B*N*4] f64 (x y z w), gps[B*N*3] f64 (NaN = no fix), valid[B*N] u8
 *      -- B equal-length trajectories as main_process_gui holds them after its time alignment (EKFGPSSLAM.py:971-972).
 * OUT: R[B*9], t[B*3], s[B], pos_out[B*N*3], quat_out[B*N*4] f64, status[B] i32 -- steps 3-5 of main_process_gui (:1002-1010) with the
 *      plain fit on the rows :973-998 picks (third argument "all": every valid row), default CONFIG (:22-71).
 * Exit code 0, or 1 with the library's message on stderr.  tests/test_c_consumer.py builds it with gcc and compares with the Python route.
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
    fprintf(stderr, "fuse_batch: %s: %s\n", what, msg);
    return 1;
}

static void *xread(FILE *f, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "fuse_batch: short input\n"); exit(2); }
    return p;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: fuse_batch IN OUT [all]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int64_t hdr[2];
    if (fread(hdr, sizeof hdr[0], 2, f) != 2 || hdr[0] < 0 || hdr[1] < 0) { fprintf(stderr, "fuse_batch: bad header\n"); return 2; }
    const int64_t B = hdr[0], N = hdr[1];
    const size_t P = (size_t)B * (size_t)N;
    double *ts = xread(f, P * 8), *pos = xread(f, P * 24), *quat = xread(f, P * 32), *gps = xread(f, P * 24);
    uint8_t *valid = xread(f, P);
    fclose(f);

    gsf_ekf_config cfg;                                    /* EKFGPSSLAM.py:24-29, :68-69 */
    memset(&cfg, 0, sizeof cfg);
    {
        const double p0[7] = { 0.1, 0.1, 0.1, 0.01, 0.01, 0.01, 0.01 }, q[7] = { 0.1, 0.1, 0.7, 0.01, 0.01, 0.01, 0.01 }, r[3] = { 0.2, 0.2, 0.2 };
        memcpy(cfg.initial_cov_diag, p0, sizeof p0); memcpy(cfg.process_noise_diag, q, sizeof q); memcpy(cfg.meas_noise_diag, r, sizeof r);
        cfg.sharp_turn_yaw_rate_threshold_deg_per_sec = 45.0;
        cfg.default_ekf_transition_steps_on_sharp_turn = 0;
    }
    if (gsf_device_count() < 1) { fprintf(stderr, "fuse_batch: no HIP device (there is no CPU fallback)\n"); return 1; }
    gsf_ctx *ctx = NULL;
    if (gsf_create(0, &ctx)) return fail("gsf_create");
    const int all = argc > 3 && strcmp(argv[3], "all") == 0;
    /* a fresh context fits the rows main_process_gui picks (:973-998) under the reference's CONFIG defaults (:34, :53, :37): nothing to
       set for the reference's flow; "all" switches to every valid row */
    if (all && gsf_set_sim3_rows(ctx, 0, 0, 0.0, 0.0)) return fail("gsf_set_sim3_rows");

    double *R = malloc(B * 9 * 8 + 8), *t = malloc(B * 3 * 8 + 8), *s = malloc(B * 8 + 8), *po = malloc(P * 24 + 8), *qo = malloc(P * 32 + 8);
    int32_t *st = malloc(B * 4 + 4);
    if (!R || !t || !s || !po || !qo || !st) { fprintf(stderr, "fuse_batch: out of memory\n"); return 2; }
    if (gsf_fuse_pipeline_batch(ctx, 0 /* trajectory-major */, ts, pos, quat, gps, valid, &cfg, B, N, R, t, s, po, qo, st)) return fail("gsf_fuse_pipeline_batch");
    gsf_destroy(ctx);

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    fwrite(R, 8, (size_t)B * 9, f); fwrite(t, 8, (size_t)B * 3, f); fwrite(s, 8, (size_t)B, f);
    fwrite(po, 8, P * 3, f); fwrite(qo, 8, P * 4, f); fwrite(st, 4, (size_t)B, f);
    fclose(f);
    int64_t none = 0;
    for (int64_t b = 0; b < B; ++b) none += ((st[b] >> 8) & GSF_SIM3_NONE) != 0;
    printf("%s\nfused %lld trajectories x %lld poses; %lld without a fit\n", gsf_version(), (long long)B, (long long)N, (long long)none);
    free(ts); free(pos); free(quat); free(gps); free(valid); free(R); free(t); free(s); free(po); free(qo); free(st);
    return 0;
}
