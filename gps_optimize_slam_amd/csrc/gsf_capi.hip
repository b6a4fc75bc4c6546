// gsf_capi.hip -- context / error plumbing of the C ABI (include/gsf.h) and the host-pointer
// convenience entry points (copy in, launch the *_dev form, copy out, synchronise).
#include <stdarg.h>
#include <string.h>

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

// RAII device staging for the host-pointer entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() { return (T*)p; }
};

}  // namespace gsf

using namespace gsf;

extern "C" {

const char* gsf_version(void) { return "gsf 0.1.0 (gfx950, fp64)"; }
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
    c->device = device_id; c->stream = stream; c->owns_stream = owns; c->scratch = nullptr; c->scratch_bytes = 0; c->ekf_variant = 0; c->wave_ppl = 0; c->seg_kernel = 0; c->duo_kernel = -1;
    if (owns) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail_hip(e, "hipStreamCreateWithFlags"); }
    }
    hipError_t e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) { delete c; return fail_hip(e, "hipEventCreate"); }
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
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
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
    if (strcmp(key, "duo_kernel") == 0) {
        if (value < -1 || value > 1) { set_error("gsf_set_option: duo_kernel must be -1 (automatic), 0 or 1"); return GSF_ERR_INVALID_ARG; }
        ctx->duo_kernel = (int)value; return GSF_OK;
    }
    if (strcmp(key, "seg_kernel") == 0) {
        if (value < 0 || value > 1) { set_error("gsf_set_option: seg_kernel must be 0 or 1"); return GSF_ERR_INVALID_ARG; }
        ctx->seg_kernel = (int)value; return GSF_OK;
    }
    if (strcmp(key, "wave_ppl") == 0) {
        if (value < 0 || value > 5) { set_error("gsf_set_option: wave_ppl must be 0 (automatic) .. 5"); return GSF_ERR_INVALID_ARG; }
        ctx->wave_ppl = (int)value; return GSF_OK;
    }
    set_error("gsf_set_option: unknown key '%s'", key);
    return GSF_ERR_INVALID_ARG;
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
// host-pointer forms
// ------------------------------------------------------------------------------------------
#define H2D(dst, src, bytes) GSF_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, ctx->stream))
#define D2H(dst, src, bytes) GSF_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, ctx->stream))

int gsf_utm_forward(gsf_ctx* ctx, const double* lat, const double* lon, int64_t n, int32_t zone, int32_t south, double* e, double* nn)
{
    GSF_REQUIRE(ctx && (n == 0 || (lat && lon && e && nn)) && n >= 0, "bad arguments");
    if (n == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    DevBuf b; GSF_HIP(b.alloc((size_t)n * 4 * sizeof(double) + 2 * sizeof(int64_t) + 2 * sizeof(int32_t)));
    double* d = b.as<double>();
    int64_t off[2] = { 0, n }; int32_t zs[2] = { zone, south };
    int64_t* doff = (int64_t*)(d + 4 * n); int32_t* dzs = (int32_t*)(doff + 2);
    H2D(d, lat, (size_t)n * 8); H2D(d + n, lon, (size_t)n * 8); H2D(doff, off, sizeof off); H2D(dzs, zs, sizeof zs);
    int rc = gsf_utm_forward_batch_dev(ctx, d, d + n, doff, dzs, dzs + 1, 1, d + 2 * n, d + 3 * n);
    if (rc) return rc;
    D2H(e, d + 2 * n, (size_t)n * 8); D2H(nn, d + 3 * n, (size_t)n * 8);
    GSF_HIP(hipStreamSynchronize(ctx->stream));
    return GSF_OK;
}

int gsf_utm_inverse(gsf_ctx* ctx, const double* e, const double* nn, int64_t n, int32_t zone, int32_t south, double* lat, double* lon)
{
    GSF_REQUIRE(ctx && (n == 0 || (lat && lon && e && nn)) && n >= 0, "bad arguments");
    if (n == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    DevBuf b; GSF_HIP(b.alloc((size_t)n * 4 * sizeof(double) + 2 * sizeof(int64_t) + 2 * sizeof(int32_t)));
    double* d = b.as<double>();
    int64_t off[2] = { 0, n }; int32_t zs[2] = { zone, south };
    int64_t* doff = (int64_t*)(d + 4 * n); int32_t* dzs = (int32_t*)(doff + 2);
    H2D(d, e, (size_t)n * 8); H2D(d + n, nn, (size_t)n * 8); H2D(doff, off, sizeof off); H2D(dzs, zs, sizeof zs);
    int rc = gsf_utm_inverse_batch_dev(ctx, d, d + n, doff, dzs, dzs + 1, 1, d + 2 * n, d + 3 * n);
    if (rc) return rc;
    D2H(lat, d + 2 * n, (size_t)n * 8); D2H(lon, d + 3 * n, (size_t)n * 8);
    GSF_HIP(hipStreamSynchronize(ctx->stream));
    return GSF_OK;
}

int gsf_sim3_umeyama_batch(gsf_ctx* ctx, const double* src, const double* dst, const uint8_t* mask, const int64_t* offsets,
                           int64_t B, double* R, double* t, double* s, int32_t* status)
{
    GSF_REQUIRE(ctx && offsets && B >= 0 && R && t && s && status, "bad arguments");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (src && dst)), "bad offsets / NULL points");
    DevBuf pts, msk, off, outb;
    GSF_HIP(pts.alloc((size_t)total * 6 * 8)); GSF_HIP(off.alloc((size_t)(B + 1) * 8));
    GSF_HIP(outb.alloc((size_t)B * (13 * 8 + 4)));
    double* dsrc = pts.as<double>(); double* ddst = dsrc + total * 3;
    H2D(dsrc, src, (size_t)total * 24); H2D(ddst, dst, (size_t)total * 24); H2D(off.p, offsets, (size_t)(B + 1) * 8);
    uint8_t* dmask = nullptr;
    if (mask) { GSF_HIP(msk.alloc((size_t)total)); dmask = msk.as<uint8_t>(); H2D(dmask, mask, (size_t)total); }
    double* dR = outb.as<double>(); double* dt = dR + 9 * B; double* ds = dt + 3 * B; int32_t* dst_ = (int32_t*)(ds + B);
    int rc = gsf_sim3_umeyama_batch_dev(ctx, dsrc, ddst, dmask, off.as<int64_t>(), B, dR, dt, ds, dst_);
    if (rc) return rc;
    D2H(R, dR, (size_t)B * 72); D2H(t, dt, (size_t)B * 24); D2H(s, ds, (size_t)B * 8); D2H(status, dst_, (size_t)B * 4);
    GSF_HIP(hipStreamSynchronize(ctx->stream));
    return GSF_OK;
}

int gsf_sim3_ransac_batch(gsf_ctx* ctx, const double* src, const double* dst, const int64_t* offsets, int64_t B,
                          const int32_t* sample_idx, int32_t trials, int32_t min_samples, double thr, int32_t min_inliers,
                          double* R, double* t, double* s, int32_t* status, uint8_t* inlier_mask, int32_t* n_inliers)
{
    GSF_REQUIRE(ctx && offsets && B >= 0 && R && t && s && status && inlier_mask && n_inliers, "bad arguments");
    GSF_REQUIRE(trials >= 0 && min_samples >= 1 && (trials == 0 || sample_idx), "bad trials/min_samples/sample_idx");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (src && dst)), "bad offsets / NULL points");
    DevBuf pts, off, outb, idx, msk;
    size_t nidx = (size_t)B * (size_t)trials * (size_t)min_samples;
    GSF_HIP(pts.alloc((size_t)total * 6 * 8)); GSF_HIP(off.alloc((size_t)(B + 1) * 8)); GSF_HIP(idx.alloc(nidx * 4));
    GSF_HIP(outb.alloc((size_t)B * (13 * 8 + 8))); GSF_HIP(msk.alloc((size_t)total));
    double* dsrc = pts.as<double>(); double* ddst = dsrc + total * 3;
    H2D(dsrc, src, (size_t)total * 24); H2D(ddst, dst, (size_t)total * 24); H2D(off.p, offsets, (size_t)(B + 1) * 8);
    if (nidx) H2D(idx.p, sample_idx, nidx * 4);
    double* dR = outb.as<double>(); double* dt = dR + 9 * B; double* ds = dt + 3 * B; int32_t* dst_ = (int32_t*)(ds + B); int32_t* dni = dst_ + B;
    int rc = gsf_sim3_ransac_batch_dev(ctx, dsrc, ddst, off.as<int64_t>(), B, idx.as<int32_t>(), trials, min_samples, thr, min_inliers,
                                       dR, dt, ds, dst_, msk.as<uint8_t>(), dni);
    if (rc) return rc;
    D2H(R, dR, (size_t)B * 72); D2H(t, dt, (size_t)B * 24); D2H(s, ds, (size_t)B * 8); D2H(status, dst_, (size_t)B * 4);
    D2H(n_inliers, dni, (size_t)B * 4);
    if (total) D2H(inlier_mask, msk.p, (size_t)total);
    GSF_HIP(hipStreamSynchronize(ctx->stream));
    return GSF_OK;
}

int gsf_apply_sim3_batch(gsf_ctx* ctx, const double* pos, const double* quat, const int64_t* offsets, int64_t B, const double* R,
                         const double* t, const double* s, double* pos_out, double* quat_out, int32_t* bad_quat)
{
    GSF_REQUIRE(ctx && offsets && B >= 0 && R && t && s, "bad arguments");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (pos && quat && pos_out && quat_out)), "bad offsets / NULL poses");
    DevBuf io, off, par;
    GSF_HIP(io.alloc((size_t)total * 14 * 8)); GSF_HIP(off.alloc((size_t)(B + 1) * 8)); GSF_HIP(par.alloc((size_t)B * (13 * 8 + 4)));
    double* dpos = io.as<double>(); double* dquat = dpos + 3 * total; double* dpo = dquat + 4 * total; double* dqo = dpo + 3 * total;
    double* dR = par.as<double>(); double* dt = dR + 9 * B; double* ds = dt + 3 * B; int32_t* dbad = (int32_t*)(ds + B);
    H2D(dpos, pos, (size_t)total * 24); H2D(dquat, quat, (size_t)total * 32); H2D(off.p, offsets, (size_t)(B + 1) * 8);
    H2D(dR, R, (size_t)B * 72); H2D(dt, t, (size_t)B * 24); H2D(ds, s, (size_t)B * 8);
    int rc = gsf_apply_sim3_batch_dev(ctx, dpos, dquat, off.as<int64_t>(), B, dR, dt, ds, dpo, dqo, dbad);
    if (rc) return rc;
    if (total) { D2H(pos_out, dpo, (size_t)total * 24); D2H(quat_out, dqo, (size_t)total * 32); }
    if (bad_quat) D2H(bad_quat, dbad, (size_t)B * 4);
    GSF_HIP(hipStreamSynchronize(ctx->stream));
    return GSF_OK;
}

int gsf_ekf_fuse_batch(gsf_ctx* ctx, int32_t layout, const double* ts, const double* pos, const double* quat, const double* gps,
                       const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                       int64_t N, double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(ctx && cfg && B >= 0 && N >= 0, "bad arguments");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps && valid && init_pos && init_quat && pos_out && quat_out, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    size_t P = (size_t)B * (size_t)N;
    DevBuf in, outb;
    GSF_HIP(in.alloc(P * (11 * 8 + 1) + (size_t)B * 7 * 8 + 64));
    GSF_HIP(outb.alloc(P * 7 * 8 + (size_t)B * 4));
    double* dts = in.as<double>(); double* dpos = dts + P; double* dquat = dpos + 3 * P; double* dgps = dquat + 4 * P;
    double* dip = dgps + 3 * P; double* diq = dip + 3 * B; uint8_t* dval = (uint8_t*)(diq + 4 * B);
    double* dpo = outb.as<double>(); double* dqo = dpo + 3 * P; int32_t* dst_ = (int32_t*)(dqo + 4 * P);
    H2D(dts, ts, P * 8); H2D(dpos, pos, P * 24); H2D(dquat, quat, P * 32); H2D(dgps, gps, P * 24); H2D(dval, valid, P);
    H2D(dip, init_pos, (size_t)B * 24); H2D(diq, init_quat, (size_t)B * 32);
    int rc = gsf_ekf_fuse_batch_dev(ctx, layout, dts, dpos, dquat, dgps, dval, dip, diq, cfg, B, N, dpo, dqo, dst_);
    if (rc) return rc;
    D2H(pos_out, dpo, P * 24); D2H(quat_out, dqo, P * 32);
    if (status) D2H(status, dst_, (size_t)B * 4);
    GSF_HIP(hipStreamSynchronize(ctx->stream));
    return GSF_OK;
}

}  // extern "C"
