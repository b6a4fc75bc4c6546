// gsf_capi.hip -- context / error plumbing of the C ABI (include/gsf.h) and the host-pointer
// convenience entry points (copy in, launch the *_dev form, copy out, synchronise).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include <vector>

#include "gsf_internal.hpp"

namespace gsf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int fail_hip(hipError_t e, const char* what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return GSF_ERR_HIP;
}

int ensure_scratch(gsf_ctx* ctx, size_t bytes)
{
    if (ctx->scratch_bytes >= bytes) return GSF_OK;
    if (ctx->scratch) {
        GSF_HIP(hipStreamSynchronize(ctx->stream));
        GSF_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr; ctx->scratch_bytes = 0;
    }
    GSF_HIP(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return GSF_OK;
}

int ensure_rng_scratch(gsf_ctx* ctx, size_t bytes)
{
    if (ctx->rng_scratch_bytes >= bytes) return GSF_OK;
    if (ctx->rng_scratch) {
        GSF_HIP(hipStreamSynchronize(ctx->stream));
        GSF_HIP(hipFree(ctx->rng_scratch));
        ctx->rng_scratch = nullptr; ctx->rng_scratch_bytes = 0;
    }
    GSF_HIP(hipMalloc(&ctx->rng_scratch, bytes + bytes / 4));
    ctx->rng_scratch_bytes = bytes + bytes / 4;
    return GSF_OK;
}

int ensure_k2b_scratch(gsf_ctx* ctx, size_t bytes)
{
    if (ctx->k2b_scratch_bytes >= bytes) return GSF_OK;
    if (ctx->k2b_scratch) {
        GSF_HIP(hipStreamSynchronize(ctx->stream));
        GSF_HIP(hipFree(ctx->k2b_scratch));
        ctx->k2b_scratch = nullptr; ctx->k2b_scratch_bytes = 0;
    }
    GSF_HIP(hipMalloc(&ctx->k2b_scratch, bytes + bytes / 4));
    ctx->k2b_scratch_bytes = bytes + bytes / 4;
    return GSF_OK;
}

int ensure_run_scratch(gsf_ctx* ctx, size_t bytes)
{
    if (ctx->run_scratch_bytes >= bytes) return GSF_OK;
    if (ctx->run_scratch) {
        GSF_HIP(hipStreamSynchronize(ctx->stream));
        GSF_HIP(hipFree(ctx->run_scratch));
        ctx->run_scratch = nullptr; ctx->run_scratch_bytes = 0;
    }
    GSF_HIP(hipMalloc(&ctx->run_scratch, bytes + bytes / 4));
    ctx->run_scratch_bytes = bytes + bytes / 4;
    return GSF_OK;
}

int ensure_rows_scratch(gsf_ctx* ctx, size_t bytes)
{
    if (ctx->rows_scratch_bytes >= bytes) return GSF_OK;
    if (ctx->rows_scratch) {
        GSF_HIP(hipStreamSynchronize(ctx->stream));
        GSF_HIP(hipFree(ctx->rows_scratch));
        ctx->rows_scratch = nullptr; ctx->rows_scratch_bytes = 0;
    }
    GSF_HIP(hipMalloc(&ctx->rows_scratch, bytes + bytes / 4));
    ctx->rows_scratch_bytes = bytes + bytes / 4;
    return GSF_OK;
}

static int ensure_arena(void** p, size_t* have, size_t bytes, bool pinned, hipStream_t stream)
{
    if (*have >= bytes) return GSF_OK;
    if (*p) {
        GSF_HIP(hipStreamSynchronize(stream));
        GSF_HIP(pinned ? hipHostFree(*p) : hipFree(*p));
        *p = nullptr; *have = 0;
    }
    size_t want = bytes + bytes / 2;                                      // grow-only with headroom: repeated calls stop allocating
    if (want < ((size_t)1 << 20)) want = (size_t)1 << 20;
    hipError_t e = pinned ? hipHostMalloc(p, want, hipHostMallocDefault) : hipMalloc(p, want);
    if (e != hipSuccess && want != bytes) { want = bytes; e = pinned ? hipHostMalloc(p, want, hipHostMallocDefault) : hipMalloc(p, want); }
    if (e != hipSuccess) { *p = nullptr; return fail_hip(e, pinned ? "hipHostMalloc(staging)" : "hipMalloc(staging)"); }
    *have = want;
    return GSF_OK;
}

Staging::Staging(gsf_ctx* ctx, size_t payload_bytes, int n_arrays) : ctx_(ctx)
{
    cap_ = payload_bytes + (size_t)256 * (size_t)(n_arrays + 1);
    direct_ = cap_ > PINNED_MAX;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { rc_ = fail_hip(e, "hipSetDevice"); return; }
    rc_ = ensure_arena(&ctx->stage, &ctx->stage_bytes, cap_, false, ctx->stream);
    if (rc_ == GSF_OK && !direct_) rc_ = ensure_arena(&ctx->pinned, &ctx->pinned_bytes, cap_, true, ctx->stream);
    d_ = (char*)ctx->stage; h_ = (char*)ctx->pinned;
}

void* Staging::take(size_t bytes, size_t& at)
{
    at = (off_ + 255) & ~(size_t)255;
    if (rc_ != GSF_OK || at + bytes > cap_) {
        if (rc_ == GSF_OK) { set_error("staging arena overrun (internal sizing error)"); rc_ = GSF_ERR_INVALID_ARG; }
        return nullptr;
    }
    off_ = at + bytes;
    return d_ + at;
}

void* Staging::in_bytes(const void* host, size_t bytes)
{
    size_t at; void* p = take(bytes, at);
    if (!p) return nullptr;
    if (has_out_) { set_error("staging: in() after out()"); rc_ = GSF_ERR_INVALID_ARG; return nullptr; }
    if (bytes && host) {
        if (direct_) { hipError_t e = hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, ctx_->stream); if (e != hipSuccess) rc_ = fail_hip(e, "hipMemcpyAsync(H2D)"); }
        else memcpy(h_ + at, host, bytes);
    }
    in_end_ = off_;
    return p;
}

void* Staging::out_bytes(void* host, size_t bytes)
{
    size_t at; void* p = take(bytes, at);
    if (!p) return nullptr;
    if (!has_out_) { has_out_ = true; out_lo_ = at; }
    if (host && bytes) {
        if (n_out_ >= MAX_OUT) { set_error("staging: too many outputs"); rc_ = GSF_ERR_INVALID_ARG; return nullptr; }
        outs_[n_out_++] = Out{ host, at, bytes };
    }
    return p;
}

int Staging::upload()
{
    if (rc_ != GSF_OK) return rc_;
    if (!direct_ && in_end_) GSF_HIP(hipMemcpyAsync(d_, h_, in_end_, hipMemcpyHostToDevice, ctx_->stream));
    return GSF_OK;
}

int Staging::finish()
{
    if (rc_ != GSF_OK) return rc_;
    if (direct_) {
        for (int k = 0; k < n_out_; ++k) GSF_HIP(hipMemcpyAsync(outs_[k].host, d_ + outs_[k].off, outs_[k].bytes, hipMemcpyDeviceToHost, ctx_->stream));
        GSF_HIP(hipStreamSynchronize(ctx_->stream));
        return GSF_OK;
    }
    if (n_out_) {
        const size_t hi = outs_[n_out_ - 1].off + outs_[n_out_ - 1].bytes;
        GSF_HIP(hipMemcpyAsync(h_ + out_lo_, d_ + out_lo_, hi - out_lo_, hipMemcpyDeviceToHost, ctx_->stream));
    }
    GSF_HIP(hipStreamSynchronize(ctx_->stream));
    for (int k = 0; k < n_out_; ++k) memcpy(outs_[k].host, h_ + outs_[k].off, outs_[k].bytes);
    return GSF_OK;
}

}  // namespace gsf

using namespace gsf;

extern "C" {

// The wave-per-trajectory kernels live in two translation units that are compiled with different instruction schedulers (Makefile); which
// scheduler and which floating-point contraction mode actually produced each object is part of the version string, so a build that took
// the Makefile's fall-back path cannot ship unnoticed (bench.py prints this string).
const char* gsf_version(void)
{
    // a function-local static is initialised once, also under concurrent first calls (C++11)
    static const std::string v = std::string("gsf 0.5.0 (gfx950, fp64) | ") + gsf::wave_small_build_info() + " | " + gsf::wave_big_build_info() + " | " +
                                 gsf::wave_block_build_info();
    return v.c_str();
}
int gsf_abi_version(void) { return GSF_ABI_VERSION; }

int gsf_last_error(char* buf, int n)
{
    int len = (int)strlen(g_err);
    if (buf && n > 0) { strncpy(buf, g_err, (size_t)n - 1); buf[n - 1] = '\0'; }
    return len;
}

int gsf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int create_common(int device_id, hipStream_t stream, bool owns, gsf_ctx** out)
{
    if (!out) { set_error("gsf_create: out is NULL"); return GSF_ERR_INVALID_ARG; }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("gsf_create: no HIP device is visible (this library has no CPU fallback)");
        return GSF_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) { set_error("gsf_create: device %d out of range [0,%d)", device_id, n); return GSF_ERR_INVALID_ARG; }
    GSF_HIP(hipSetDevice(device_id));
    gsf_ctx* c = new gsf_ctx();
    c->device = device_id; c->stream = stream; c->owns_stream = owns; c->scratch = nullptr; c->scratch_bytes = 0; c->stage = nullptr; c->stage_bytes = 0; c->pinned = nullptr; c->pinned_bytes = 0;
    c->rng_scratch = nullptr; c->rng_scratch_bytes = 0; c->rows_scratch = nullptr; c->rows_scratch_bytes = 0; c->run_scratch = nullptr; c->run_scratch_bytes = 0; c->tape_draws = -1; c->small_scratch = nullptr; c->k2b_screen = 1; c->k2b_scratch = nullptr; c->k2b_scratch_bytes = 0;
    c->ekf_variant = 0; c->synth_variant = 0; c->block_kernel = -1; c->duo_kernel = -1; c->lane_min_traj = 32768;
    c->ransac_early_exit = 0; c->ransac_probe_trials = 64; c->prefilter_first_batch = 1; c->prefilter_speculate = 1; c->prefilter_miss_batch = 4;
    // the fused chains fit the rows main_process_gui hands to its fit (ref :973-998) under the reference's CONFIG defaults (:34, :53, :37)
    // unless the caller says otherwise (gsf_set_sim3_rows): a raw C caller of gsf_fuse_pipeline_* gets steps 3-5 as the reference runs them
    c->fit_rows = gsf::FitRows{ 1, 4, 5.0, 180.0 };
    if (owns) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail_hip(e, "hipStreamCreateWithFlags"); }
    }
    // on a failure everything created so far is released again, and the message names the call that failed
    bool have_ev0 = false, have_ev1 = false;
    const char* what = "hipEventCreate";
    hipError_t e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) { have_ev0 = true; e = hipEventCreate(&c->ev1); }
    if (e == hipSuccess) { have_ev1 = true; what = "hipMalloc(small_scratch)"; e = hipMalloc(&c->small_scratch, 512); }
    if (e != hipSuccess) {
        if (have_ev0) (void)hipEventDestroy(c->ev0);
        if (have_ev1) (void)hipEventDestroy(c->ev1);
        if (owns) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail_hip(e, what);
    }
    *out = c;
    return GSF_OK;
}

int gsf_create(int device_id, gsf_ctx** out) { return create_common(device_id, nullptr, true, out); }
int gsf_create_on_stream(int device_id, void* hip_stream, gsf_ctx** out) { return create_common(device_id, (hipStream_t)hip_stream, false, out); }

void gsf_destroy(gsf_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->rng_scratch) (void)hipFree(ctx->rng_scratch);
    if (ctx->rows_scratch) (void)hipFree(ctx->rows_scratch);
    if (ctx->run_scratch) (void)hipFree(ctx->run_scratch);
    if (ctx->small_scratch) (void)hipFree(ctx->small_scratch);
    if (ctx->k2b_scratch) (void)hipFree(ctx->k2b_scratch);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int gsf_trim(gsf_ctx* ctx)
{
    GSF_REQUIRE(ctx, "ctx is NULL");
    GSF_HIP(hipSetDevice(ctx->device));
    GSF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) { GSF_HIP(hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    if (ctx->rng_scratch) { GSF_HIP(hipFree(ctx->rng_scratch)); ctx->rng_scratch = nullptr; ctx->rng_scratch_bytes = 0; }
    if (ctx->k2b_scratch) { GSF_HIP(hipFree(ctx->k2b_scratch)); ctx->k2b_scratch = nullptr; ctx->k2b_scratch_bytes = 0; }
    if (ctx->rows_scratch) { GSF_HIP(hipFree(ctx->rows_scratch)); ctx->rows_scratch = nullptr; ctx->rows_scratch_bytes = 0; }
    if (ctx->run_scratch) { GSF_HIP(hipFree(ctx->run_scratch)); ctx->run_scratch = nullptr; ctx->run_scratch_bytes = 0; }
    if (ctx->stage) { GSF_HIP(hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
    if (ctx->pinned) { GSF_HIP(hipHostFree(ctx->pinned)); ctx->pinned = nullptr; ctx->pinned_bytes = 0; }
    return GSF_OK;
}

int gsf_synchronize(gsf_ctx* ctx)
{
    GSF_REQUIRE(ctx, "ctx is NULL");
    GSF_HIP(hipStreamSynchronize(ctx->stream));
    return GSF_OK;
}

int gsf_set_option(gsf_ctx* ctx, const char* key, int64_t value)
{
    GSF_REQUIRE(ctx && key, "NULL argument");
    if (strcmp(key, "ekf_variant") == 0) { ctx->ekf_variant = (int)value; return GSF_OK; }
    if (strcmp(key, "synth_variant") == 0) {
        if (value < 0 || value > 1) { set_error("gsf_set_option: synth_variant must be 0 (white SLAM noise) or 1 (random-walk drift, SURVEY 8d)"); return GSF_ERR_INVALID_ARG; }
        ctx->synth_variant = (int)value; return GSF_OK;
    }
    if (strcmp(key, "block_kernel") == 0) {
        if (value < -1 || value > 1) { set_error("gsf_set_option: block_kernel must be -1 (automatic), 0 (never) or 1 (whenever it applies)"); return GSF_ERR_INVALID_ARG; }
        ctx->block_kernel = (int)value; return GSF_OK;
    }
    if (strcmp(key, "k2b_screen") == 0) {
        if (value < 0 || value > 1) { set_error("gsf_set_option: k2b_screen must be 1 (single-precision screen + exact re-check, default) or 0 (double throughout)"); return GSF_ERR_INVALID_ARG; }
        ctx->k2b_screen = (int)value; return GSF_OK;
    }
    if (strcmp(key, "tape_draws") == 0) {
        if (value < -1 || value > 2 || value == 1) { set_error("gsf_set_option: tape_draws must be -1 (automatic: a few streams are drawn chip-wide), 0 (always one wave per stream) or 2 (tests: a tape cut short, so that the one-wave kernel takes over)"); return GSF_ERR_INVALID_ARG; }
        ctx->tape_draws = (int)value; return GSF_OK;
    }
    if (strcmp(key, "ransac_early_exit") == 0) {
        if (value < 0 || value > 1) { set_error("gsf_set_option: ransac_early_exit must be 0 (every trajectory draws all max_trials: the generator ends where the reference leaves it) or 1 (a trajectory stops at the first trial that counts every row: same R, t, s, mask and poses)"); return GSF_ERR_INVALID_ARG; }
        ctx->ransac_early_exit = (int)value; return GSF_OK;
    }
    if (strcmp(key, "prefilter_speculate") == 0) {
        if (value != 0 && value != 1) { set_error("gsf_set_option: prefilter_speculate must be 0 or 1"); return GSF_ERR_INVALID_ARG; }
        ctx->prefilter_speculate = (int)value; return GSF_OK;
    }
    if (strcmp(key, "prefilter_miss_batch") == 0) {
        if (value < 1 || value > 64) { set_error("gsf_set_option: prefilter_miss_batch must be in [1, 64]"); return GSF_ERR_INVALID_ARG; }
        ctx->prefilter_miss_batch = (int)value; return GSF_OK;
    }
    if (strcmp(key, "prefilter_first_batch") == 0) {
        if (value < 1 || value > 64) { set_error("gsf_set_option: prefilter_first_batch must be in [1, 64]"); return GSF_ERR_INVALID_ARG; }
        ctx->prefilter_first_batch = (int)value; return GSF_OK;
    }
    if (strcmp(key, "ransac_probe_trials") == 0) {
        if (value < 1 || value > (1 << 20)) { set_error("gsf_set_option: ransac_probe_trials must be in [1, 2^20]"); return GSF_ERR_INVALID_ARG; }
        ctx->ransac_probe_trials = (int)value; return GSF_OK;
    }
    if (strcmp(key, "duo_kernel") == 0) {
        if (value < -1 || value > 1) { set_error("gsf_set_option: duo_kernel must be -1 (automatic), 0 (one wave) or 1 (two-wave blocks)"); return GSF_ERR_INVALID_ARG; }
        ctx->duo_kernel = (int)value; return GSF_OK;
    }
    if (strcmp(key, "lane_min_traj") == 0) {
        if (value < 0) { set_error("gsf_set_option: lane_min_traj must be >= 0"); return GSF_ERR_INVALID_ARG; }
        ctx->lane_min_traj = value; return GSF_OK;
    }
    set_error("gsf_set_option: unknown key '%s'", key);
    return GSF_ERR_INVALID_ARG;
}

int gsf_set_sim3_rows(gsf_ctx* ctx, int32_t mode, int32_t min_samples, double max_gps_gap_threshold, double max_initial_duration)
{
    GSF_REQUIRE(ctx, "ctx is NULL");
    GSF_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (all valid rows) or 1 (the reference's choice, EKFGPSSLAM.py:973-998)");
    GSF_REQUIRE(mode == 0 || min_samples >= 0, "min_samples must be >= 0");
    ctx->fit_rows = gsf::FitRows{ mode, min_samples, max_gps_gap_threshold, max_initial_duration };
    return GSF_OK;
}

int gsf_timer_start(gsf_ctx* ctx)
{
    GSF_REQUIRE(ctx, "ctx is NULL");
    GSF_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return GSF_OK;
}

int gsf_timer_stop(gsf_ctx* ctx, float* elapsed_ms)
{
    GSF_REQUIRE(ctx && elapsed_ms, "NULL argument");
    GSF_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    GSF_HIP(hipEventSynchronize(ctx->ev1));
    GSF_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return GSF_OK;
}

// ------------------------------------------------------------------------------------------
// host-pointer forms: pack -> one H2D -> *_dev launch -> one D2H -> synchronise (gsf::Staging)
// ------------------------------------------------------------------------------------------
#define ST_BEGIN(bytes, n) Staging st(ctx, (bytes), (n)); if (st.rc()) return st.rc()
#define ST_RUN(call) do { int rc__ = st.upload(); if (rc__) return rc__; rc__ = (call); if (rc__) return rc__; return st.finish(); } while (0)

static int utm_host(gsf_ctx* ctx, bool inverse, const double* a, const double* b, int64_t n, int32_t zone, int32_t south, double* oa, double* ob)
{
    GSF_REQUIRE(ctx && (n == 0 || (a && b && oa && ob)) && n >= 0, "bad arguments");
    if (n == 0) return GSF_OK;
    ST_BEGIN((size_t)n * 32 + 24, 6);
    const int64_t off[2] = { 0, n }; const int32_t zs[2] = { zone, south };
    const double* da = st.in(a, (size_t)n); const double* db = st.in(b, (size_t)n);
    const int64_t* doff = st.in(off, 2); const int32_t* dzs = st.in(zs, 2);
    double* doa = st.out(oa, (size_t)n); double* dob = st.out(ob, (size_t)n);
    if (inverse) ST_RUN(gsf_utm_inverse_batch_dev(ctx, da, db, doff, dzs, dzs + 1, 1, doa, dob));
    ST_RUN(gsf_utm_forward_batch_dev(ctx, da, db, doff, dzs, dzs + 1, 1, doa, dob));
}

int gsf_utm_forward(gsf_ctx* ctx, const double* lat, const double* lon, int64_t n, int32_t zone, int32_t south, double* e, double* nn)
{
    return utm_host(ctx, false, lat, lon, n, zone, south, e, nn);
}

int gsf_utm_inverse(gsf_ctx* ctx, const double* e, const double* nn, int64_t n, int32_t zone, int32_t south, double* lat, double* lon)
{
    return utm_host(ctx, true, e, nn, n, zone, south, lat, lon);
}

int gsf_sim3_umeyama_batch(gsf_ctx* ctx, const double* src, const double* dst, const uint8_t* mask, const int64_t* offsets,
                           int64_t B, double* R, double* t, double* s, int32_t* status)
{
    GSF_REQUIRE(ctx && offsets && B >= 0 && R && t && s && status, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (src && dst)), "bad offsets / NULL points");
    ST_BEGIN((size_t)total * 49 + (size_t)(B + 1) * 8 + (size_t)B * 108, 8);
    const double* dsrc = st.in(src, (size_t)total * 3); const double* ddst = st.in(dst, (size_t)total * 3);
    const int64_t* doff = st.in(offsets, (size_t)B + 1);
    const uint8_t* dmask = mask ? st.in(mask, (size_t)total) : nullptr;
    double* dR = st.out(R, (size_t)B * 9); double* dt = st.out(t, (size_t)B * 3); double* ds = st.out(s, (size_t)B);
    int32_t* dst_ = st.out(status, (size_t)B);
    ST_RUN(gsf_sim3_umeyama_batch_dev(ctx, dsrc, ddst, dmask, doff, B, dR, dt, ds, dst_));
}

int gsf_sim3_ransac_batch(gsf_ctx* ctx, const double* src, const double* dst, const int64_t* offsets, int64_t B,
                          const int32_t* sample_idx, int32_t trials, int32_t min_samples, double thr, int32_t min_inliers,
                          double* R, double* t, double* s, int32_t* status, uint8_t* inlier_mask, int32_t* n_inliers)
{
    GSF_REQUIRE(ctx && offsets && B >= 0 && R && t && s && status && inlier_mask && n_inliers, "bad arguments");
    GSF_REQUIRE(trials >= 0 && min_samples >= 1 && (trials == 0 || sample_idx), "bad trials/min_samples/sample_idx");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (src && dst)), "bad offsets / NULL points");
    const size_t nidx = (size_t)B * (size_t)trials * (size_t)min_samples;
    ST_BEGIN((size_t)total * 49 + (size_t)(B + 1) * 8 + nidx * 4 + (size_t)B * 112, 10);
    const double* dsrc = st.in(src, (size_t)total * 3); const double* ddst = st.in(dst, (size_t)total * 3);
    const int64_t* doff = st.in(offsets, (size_t)B + 1);
    const int32_t* didx = st.in(sample_idx, nidx);
    double* dR = st.out(R, (size_t)B * 9); double* dt = st.out(t, (size_t)B * 3); double* ds = st.out(s, (size_t)B);
    int32_t* dst_ = st.out(status, (size_t)B); int32_t* dni = st.out(n_inliers, (size_t)B);
    uint8_t* dmask = st.out(inlier_mask, (size_t)total);
    ST_RUN(gsf_sim3_ransac_batch_rows_dev(ctx, dsrc, ddst, doff, total, B, didx, trials, min_samples, thr, min_inliers, dR, dt, ds, dst_, dmask, dni));
}

// compute_sim3_transform_robust with the draws made on the device: mt_state[B][625] (host, in/out) is NumPy's legacy generator state
int gsf_sim3_ransac_mt_batch(gsf_ctx* ctx, const double* src, const double* dst, const int64_t* offsets, int64_t B, uint32_t* mt_state,
                             int32_t trials, int32_t min_samples, double thr, int32_t min_inliers, double* R, double* t, double* s,
                             int32_t* status, uint8_t* inlier_mask, int32_t* n_inliers)
{
    GSF_REQUIRE(ctx && offsets && B >= 0 && B <= 0x7fffffff && mt_state && R && t && s && status && inlier_mask && n_inliers, "bad arguments");
    GSF_REQUIRE(trials >= 0 && trials <= (1 << 20) && min_samples >= 1 && min_samples <= 64, "bad trials / min_samples (1..64: the device sampler traces up to 64 positions per trial)");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (src && dst)), "bad offsets / NULL points");
    std::vector<int32_t> counts((size_t)B);
    int32_t n_max = 1;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t n = offsets[b + 1] - offsets[b];
        GSF_REQUIRE(n >= 0 && n <= 28000, "a point set has more than 28000 rows (device-side draws) or negative length");
        counts[(size_t)b] = (int32_t)n;
        if ((int32_t)n > n_max) n_max = (int32_t)n;
    }
    const size_t nidx = (size_t)B * (size_t)trials * (size_t)min_samples;
    ST_BEGIN((size_t)total * 49 + (size_t)(B + 1) * 8 + (size_t)B * (625 * 8 + 4 + 112) + nidx * 4, 14);
    const double* dsrc = st.in(src, (size_t)total * 3); const double* ddst = st.in(dst, (size_t)total * 3);
    const int64_t* doff = st.in(offsets, (size_t)B + 1);
    const int32_t* dcnt = st.in(counts.data(), (size_t)B);
    const uint32_t* dst_in = st.in(mt_state, (size_t)B * 625);
    uint32_t* dstate = st.out(mt_state, (size_t)B * 625);
    double* dR = st.out(R, (size_t)B * 9); double* dt = st.out(t, (size_t)B * 3); double* ds = st.out(s, (size_t)B);
    int32_t* dst_ = st.out(status, (size_t)B); int32_t* dni = st.out(n_inliers, (size_t)B);
    uint8_t* dmask = st.out(inlier_mask, (size_t)total);
    int32_t* didx = st.tmp<int32_t>(nidx + 1);
    int rc = st.upload();
    if (rc) return rc;
    GSF_HIP(hipMemcpyAsync(dstate, dst_in, (size_t)B * 625 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (trials > 0 && (rc = launch_mt_choice(ctx, dstate, dcnt, B, trials, min_samples, didx, n_max))) return rc;
    if ((rc = launch_sim3_ransac(ctx, dsrc, ddst, doff, nullptr, B, didx, trials, min_samples, thr, min_inliers, dR, dt, ds, dst_, dmask, dni, total))) return rc;
    return st.finish();
}

int gsf_apply_sim3_batch(gsf_ctx* ctx, const double* pos, const double* quat, const int64_t* offsets, int64_t B, const double* R,
                         const double* t, const double* s, double* pos_out, double* quat_out, int32_t* bad_quat)
{
    GSF_REQUIRE(ctx && offsets && B >= 0 && R && t && s, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (pos && quat && pos_out && quat_out)), "bad offsets / NULL poses");
    ST_BEGIN((size_t)total * 112 + (size_t)(B + 1) * 8 + (size_t)B * 108, 9);
    const double* dpos = st.in(pos, (size_t)total * 3); const double* dquat = st.in(quat, (size_t)total * 4);
    const int64_t* doff = st.in(offsets, (size_t)B + 1);
    const double* dR = st.in(R, (size_t)B * 9); const double* dt = st.in(t, (size_t)B * 3); const double* ds = st.in(s, (size_t)B);
    double* dpo = st.out(pos_out, (size_t)total * 3); double* dqo = st.out(quat_out, (size_t)total * 4);
    int32_t* dbad = st.out(bad_quat, (size_t)B);
    ST_RUN(gsf_apply_sim3_batch_dev(ctx, dpos, dquat, doff, B, dR, dt, ds, dpo, dqo, dbad));
}

int gsf_ekf_fuse_batch(gsf_ctx* ctx, int32_t layout, const double* ts, const double* pos, const double* quat, const double* gps,
                       const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                       int64_t N, double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(ctx && cfg && B >= 0 && N >= 0, "bad arguments");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps && valid && init_pos && init_quat && pos_out && quat_out, "NULL array");
    const size_t P = (size_t)B * (size_t)N;
    ST_BEGIN(P * 145 + (size_t)B * 60, 10);
    const double* dts = st.in(ts, P); const double* dpos = st.in(pos, P * 3); const double* dquat = st.in(quat, P * 4);
    const double* dgps = st.in(gps, P * 3); const uint8_t* dval = st.in(valid, P);
    const double* dip = st.in(init_pos, (size_t)B * 3); const double* diq = st.in(init_quat, (size_t)B * 4);
    double* dpo = st.out(pos_out, P * 3); double* dqo = st.out(quat_out, P * 4); int32_t* dst_ = st.out(status, (size_t)B);
    ST_RUN(gsf_ekf_fuse_batch_dev(ctx, layout, dts, dpos, dquat, dgps, dval, dip, diq, cfg, B, N, dpo, dqo, dst_));
}

// steps 3-5 of main_process_gui with the plain fit, host arrays in / out (one upload, one launch, one download)
int gsf_fuse_pipeline_batch(gsf_ctx* ctx, int32_t layout, const double* ts, const double* pos, const double* quat, const double* gps,
                            const uint8_t* valid, const gsf_ekf_config* cfg, int64_t B, int64_t N, double* R, double* t, double* s,
                            double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(ctx && cfg && B >= 0 && N >= 0, "bad arguments");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps && valid && R && t && s && pos_out && quat_out && status, "NULL array");
    const size_t P = (size_t)B * (size_t)N;
    ST_BEGIN(P * 145 + (size_t)B * 108, 11);
    const double* dts = st.in(ts, P); const double* dpos = st.in(pos, P * 3); const double* dquat = st.in(quat, P * 4);
    const double* dgps = st.in(gps, P * 3); const uint8_t* dval = st.in(valid, P);
    double* dR = st.out(R, (size_t)B * 9); double* dt = st.out(t, (size_t)B * 3); double* ds = st.out(s, (size_t)B);
    double* dpo = st.out(pos_out, P * 3); double* dqo = st.out(quat_out, P * 4); int32_t* dst_ = st.out(status, (size_t)B);
    ST_RUN(gsf_fuse_pipeline_batch_dev(ctx, layout, dts, dpos, dquat, dgps, dval, cfg, B, N, dR, dt, ds, dpo, dqo, dst_));
}

// main_process_gui's row choice (ref :973-998), host arrays
int gsf_sim3_fit_rows_batch(gsf_ctx* ctx, const double* ts, const double* gps, const uint8_t* valid, const int64_t* offsets, int64_t B, int64_t N,
                            int32_t min_samples, double max_gps_gap_threshold, double max_initial_duration, uint8_t* row_mask, int32_t* n_rows,
                            int32_t* status)
{
    GSF_REQUIRE(ctx && B >= 0 && (offsets || N >= 0), "bad arguments");
    if (B == 0 || (!offsets && N == 0)) return GSF_OK;
    GSF_REQUIRE(ts && valid && row_mask && n_rows, "NULL array");
    const int64_t total = offsets ? offsets[B] : B * N;
    GSF_REQUIRE(total >= 0, "bad offsets");
    if (total == 0) { for (int64_t b = 0; b < B; ++b) { n_rows[b] = -1; if (status) status[b] = GSF_SIM3_FLAG_FEW_ROWS; } return GSF_OK; }
    const size_t P = (size_t)total;
    ST_BEGIN(P * 34 + (size_t)B * 16 + 64, 8);
    const double* dts = st.in(ts, P); const double* dgps = gps ? st.in(gps, P * 3) : nullptr; const uint8_t* dval = st.in(valid, P);
    const int64_t* doff = offsets ? st.in(offsets, (size_t)B + 1) : nullptr;
    uint8_t* dmask = st.out(row_mask, P); int32_t* dn = st.out(n_rows, (size_t)B); int32_t* dst_ = status ? st.out(status, (size_t)B) : nullptr;
    ST_RUN(gsf_sim3_fit_rows_batch_dev(ctx, dts, dgps, dval, doff, B, N, min_samples, max_gps_gap_threshold, max_initial_duration, dmask, dn, dst_));
}

// the same steps with the reference's robust fit; mt_state[B][625] (host, in/out) is each trajectory's NumPy legacy generator state
int gsf_fuse_pipeline_robust_batch(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps,
                                   const uint8_t* valid, const gsf_ekf_config* cfg, int64_t B, int64_t N, int32_t min_samples,
                                   double residual_threshold, int32_t max_trials, int32_t min_inliers_needed, uint32_t* mt_state, double* R,
                                   double* t, double* s, double* pos_out, double* quat_out, int32_t* status, int32_t* n_inliers,
                                   uint8_t* inlier_mask)
{
    GSF_REQUIRE(ctx && cfg && B >= 0 && N >= 0 && mt_state, "bad arguments");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps && valid && R && t && s && pos_out && quat_out && status && n_inliers, "NULL array");
    const size_t P = (size_t)B * (size_t)N;
    ST_BEGIN(P * 146 + (size_t)B * (112 + 625 * 8), 15);
    const double* dts = st.in(ts, P); const double* dpos = st.in(pos, P * 3); const double* dquat = st.in(quat, P * 4);
    const double* dgps = st.in(gps, P * 3); const uint8_t* dval = st.in(valid, P);
    const uint32_t* dst_in = st.in(mt_state, (size_t)B * 625);
    uint32_t* dstate = st.out(mt_state, (size_t)B * 625);
    double* dR = st.out(R, (size_t)B * 9); double* dt = st.out(t, (size_t)B * 3); double* ds = st.out(s, (size_t)B);
    double* dpo = st.out(pos_out, P * 3); double* dqo = st.out(quat_out, P * 4); int32_t* dst_ = st.out(status, (size_t)B);
    int32_t* dni = st.out(n_inliers, (size_t)B);
    uint8_t* dmask = inlier_mask ? st.out(inlier_mask, P) : nullptr;
    int rc = st.upload();
    if (rc) return rc;
    GSF_HIP(hipMemcpyAsync(dstate, dst_in, (size_t)B * 625 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    rc = gsf_fuse_pipeline_robust_batch_dev(ctx, dts, dpos, dquat, dgps, dval, cfg, B, N, min_samples, residual_threshold, max_trials,
                                            min_inliers_needed, dstate, dR, dt, ds, dpo, dqo, dst_, dni, dmask);
    if (rc) return rc;
    return st.finish();
}

// tracks of different lengths (flat [total][C] host arrays, trajectory b = rows offsets[b]..offsets[b+1])
int gsf_ekf_fuse_ragged(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps, const uint8_t* valid,
                        const int64_t* offsets, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                        double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(ctx && cfg && offsets && B >= 0 && init_pos && init_quat && status, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (ts && pos && quat && gps && valid && pos_out && quat_out)), "bad offsets / NULL arrays");
    const size_t P = (size_t)total;
    ST_BEGIN(P * 145 + (size_t)(B + 1) * 8 + (size_t)B * 60, 11);
    const double* dts = st.in(ts, P); const double* dpos = st.in(pos, P * 3); const double* dquat = st.in(quat, P * 4);
    const double* dgps = st.in(gps, P * 3); const uint8_t* dval = st.in(valid, P); const int64_t* doff = st.in(offsets, (size_t)B + 1);
    const double* dip = st.in(init_pos, (size_t)B * 3); const double* diq = st.in(init_quat, (size_t)B * 4);
    double* dpo = st.out(pos_out, P * 3); double* dqo = st.out(quat_out, P * 4); int32_t* dst_ = st.out(status, (size_t)B);
    ST_RUN(gsf_ekf_fuse_ragged_dev(ctx, dts, dpos, dquat, dgps, dval, doff, dip, diq, cfg, B, dpo, dqo, dst_));
}
int gsf_fuse_pipeline_ragged(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps, const uint8_t* valid,
                             const int64_t* offsets, const gsf_ekf_config* cfg, int64_t B, double* R, double* t, double* s, double* pos_out,
                             double* quat_out, int32_t* status)
{
    GSF_REQUIRE(ctx && cfg && offsets && B >= 0 && R && t && s && status, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (ts && pos && quat && gps && valid && pos_out && quat_out)), "bad offsets / NULL arrays");
    const size_t P = (size_t)total;
    ST_BEGIN(P * 145 + (size_t)(B + 1) * 8 + (size_t)B * 108, 12);
    const double* dts = st.in(ts, P); const double* dpos = st.in(pos, P * 3); const double* dquat = st.in(quat, P * 4);
    const double* dgps = st.in(gps, P * 3); const uint8_t* dval = st.in(valid, P); const int64_t* doff = st.in(offsets, (size_t)B + 1);
    double* dR = st.out(R, (size_t)B * 9); double* dt = st.out(t, (size_t)B * 3); double* ds = st.out(s, (size_t)B);
    double* dpo = st.out(pos_out, P * 3); double* dqo = st.out(quat_out, P * 4); int32_t* dst_ = st.out(status, (size_t)B);
    ST_RUN(gsf_fuse_pipeline_ragged_dev(ctx, dts, dpos, dquat, dgps, dval, doff, cfg, B, dR, dt, ds, dpo, dqo, dst_));
}

// B equal-size windows of W point pairs (sliding-window re-alignment) held by the host
int gsf_sim3_umeyama_windows(gsf_ctx* ctx, const double* src, const double* dst, const uint8_t* mask, int64_t B, int32_t W, double* R, double* t,
                             double* s, int32_t* status)
{
    GSF_REQUIRE(ctx && B >= 0 && W >= 0 && R && t && s && status, "bad arguments");
    if (B == 0) return GSF_OK;
    GSF_REQUIRE(W == 0 || (src && dst), "NULL points");
    const size_t P = (size_t)B * (size_t)W;
    ST_BEGIN(P * 49 + (size_t)B * 108, 7);
    const double* dsrc = st.in(src, P * 3); const double* ddst = st.in(dst, P * 3);
    const uint8_t* dmask = mask ? st.in(mask, P) : nullptr;
    double* dR = st.out(R, (size_t)B * 9); double* dt = st.out(t, (size_t)B * 3); double* ds = st.out(s, (size_t)B); int32_t* dst_ = st.out(status, (size_t)B);
    ST_RUN(gsf_sim3_umeyama_windows_dev(ctx, dsrc, ddst, dmask, B, W, dR, dt, ds, dst_));
}

// WGS84 -> local ENU about per-trajectory origins, host arrays
int gsf_geodetic_to_enu_batch(gsf_ctx* ctx, const double* lat_deg, const double* lon_deg, const double* alt, const int64_t* offsets,
                              const double* ref_llh, int64_t B, double* east, double* north, double* up)
{
    GSF_REQUIRE(ctx && offsets && ref_llh && B >= 0, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (lat_deg && lon_deg && alt && east && north && up)), "bad offsets / NULL arrays");
    const size_t P = (size_t)total;
    ST_BEGIN(P * 48 + (size_t)(B + 1) * 8 + (size_t)B * 24, 8);
    const double* dla = st.in(lat_deg, P); const double* dlo = st.in(lon_deg, P); const double* dal = st.in(alt, P);
    const int64_t* doff = st.in(offsets, (size_t)B + 1); const double* dref = st.in(ref_llh, (size_t)B * 3);
    double* de = st.out(east, P); double* dn = st.out(north, P); double* du = st.out(up, P);
    ST_RUN(gsf_geodetic_to_enu_batch_dev(ctx, dla, dlo, dal, doff, dref, B, de, dn, du));
}

// load_gps_data's geodesy slice for B ragged logs held by the host
int gsf_gps_rows_to_utm_batch(gsf_ctx* ctx, const double* llh, const int64_t* offsets, int64_t B, double* utm_rows, int32_t* zone, int32_t* south)
{
    GSF_REQUIRE(ctx && offsets && B >= 0 && zone && south, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (llh && utm_rows)), "bad offsets / NULL rows");
    ST_BEGIN((size_t)total * 48 + (size_t)(B + 1) * 8 + (size_t)B * 8, 5);
    const double* dllh = st.in(llh, (size_t)total * 3); const int64_t* doff = st.in(offsets, (size_t)B + 1);
    double* dutm = st.out(utm_rows, (size_t)total * 3); int32_t* dz = st.out(zone, (size_t)B); int32_t* dso = st.out(south, (size_t)B);
    ST_RUN(gsf_gps_rows_to_utm_batch_dev(ctx, dllh, doff, B, dutm, dz, dso));
}

// one RANSACRegressor.fit per problem with host-drawn sample sets, host arrays in / out
int gsf_ransac_poly_batch(gsf_ctx* ctx, const double* t, const double* y, const int64_t* offsets, int64_t P, const int32_t* sample_idx,
                          int32_t max_trials, int32_t min_samples, int32_t degree, double residual_threshold, double stop_probability,
                          uint8_t* inlier_mask, int32_t* n_trials, int32_t* n_inliers, int32_t* status)
{
    GSF_REQUIRE(ctx && offsets && P >= 0 && n_trials && n_inliers && status, "bad arguments");
    GSF_REQUIRE(max_trials >= 1 && min_samples >= 1, "bad max_trials / min_samples");
    if (P == 0) return GSF_OK;
    const int64_t total = offsets[P];
    GSF_REQUIRE(total >= 0 && (total == 0 || (t && y && inlier_mask)) && sample_idx, "bad offsets / NULL arrays");
    const size_t nidx = (size_t)P * (size_t)max_trials * (size_t)min_samples;
    ST_BEGIN((size_t)total * 17 + (size_t)(P + 1) * 8 + nidx * 4 + (size_t)P * 12, 8);
    const double* dt = st.in(t, (size_t)total); const double* dy = st.in(y, (size_t)total); const int64_t* doff = st.in(offsets, (size_t)P + 1);
    const int32_t* didx = st.in(sample_idx, nidx);
    uint8_t* dmask = st.out(inlier_mask, (size_t)total); int32_t* dnt = st.out(n_trials, (size_t)P); int32_t* dni = st.out(n_inliers, (size_t)P);
    int32_t* dst_ = st.out(status, (size_t)P);
    ST_RUN(gsf_ransac_poly_batch_dev(ctx, dt, dy, doff, P, didx, max_trials, min_samples, degree, residual_threshold, stop_probability, dmask, dnt, dni, dst_));
}

}  // extern "C"
