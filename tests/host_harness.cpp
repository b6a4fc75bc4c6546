// TEST-ONLY host harness: compiles the PRODUCT's per-lane math headers (gsf_math.hpp,
// gsf_ekf_core.hpp) with g++ so that the exact code the HIP kernels inline can be checked
// against the oracle and the goldens in the CPU-only test tier (no GPU in the build
// container).  Never shipped, never loaded by gps_optimize_slam_amd.
#include <cstdint>
#include <cstring>
#include "../gps_optimize_slam_amd/csrc/gsf_ekf_core.hpp"

using namespace gsf;

namespace {
struct AosOut {
    const double* ts_; double* pos; double* quat;
    void store(int64_t i, const Vec3& p, const Quat& q) {
        pos[i * 3] = p.x; pos[i * 3 + 1] = p.y; pos[i * 3 + 2] = p.z;
        quat[i * 4] = q.x; quat[i * 4 + 1] = q.y; quat[i * 4 + 2] = q.z; quat[i * 4 + 3] = q.w;
    }
    void load(int64_t i, Vec3& p, Quat& q) const {
        p = Vec3{ pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2] };
        q = Quat{ quat[i * 4], quat[i * 4 + 1], quat[i * 4 + 2], quat[i * 4 + 3] };
    }
    double stamp(int64_t i) const { return ts_[i]; }
};
StepIn load_step(const double* ts, const double* pos, const double* quat, const double* gps, const uint8_t* valid, int64_t i) {
    StepIn s;
    s.t = ts[i];
    s.p = Vec3{ pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2] };
    s.q = Quat{ quat[i * 4], quat[i * 4 + 1], quat[i * 4 + 2], quat[i * 4 + 3] };
    s.z = Vec3{ gps[i * 3], gps[i * 3 + 1], gps[i * 3 + 2] };
    s.valid = valid[i];
    return s;
}
}  // namespace

extern "C" {
int hh_ekf_config_size() { return (int)sizeof(EkfConfig); }

int hh_ekf_fuse(const double* ts, const double* pos, const double* quat, const double* gps, const uint8_t* valid,
                int64_t n, const double* p0, const double* q0, const EkfConfig* cfg, double* pos_out, double* quat_out)
{
    if (n <= 0) return 0;
    AosOut out{ ts, pos_out, quat_out };
    EkfTraj<AosOut> f;
    f.init(*cfg, Vec3{ p0[0], p0[1], p0[2] }, Quat{ q0[0], q0[1], q0[2], q0[3] }, load_step(ts, pos, quat, gps, valid, 0), out);
    for (int64_t i = 1; i < n; ++i) f.step(*cfg, i, load_step(ts, pos, quat, gps, valid, i), out);
    return f.finish();
}

static int umeyama_impl(const double* src, const double* dst, int64_t n, double* R, double* t, double* s, bool polar)
{
    if (n < 3) return SIM3_NONE;
    double sc[3] = { 0, 0, 0 }, dc[3] = { 0, 0, 0 };
    for (int64_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { sc[k] += src[i * 3 + k]; dc[k] += dst[i * 3 + k]; }
    for (int k = 0; k < 3; ++k) { sc[k] /= (double)n; dc[k] /= (double)n; }
    double H[9] = { 0 }, ssq = 0;
    for (int64_t i = 0; i < n; ++i) {
        double a[3], b[3];
        for (int k = 0; k < 3; ++k) { a[k] = src[i * 3 + k] - sc[k]; b[k] = dst[i * 3 + k] - dc[k]; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r * 3 + c] += a[r] * b[c];
        ssq += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    }
    return polar ? umeyama_finalize<true>(H, ssq, sc, dc, (double)n, R, t, *s) : umeyama_finalize(H, ssq, sc, dc, (double)n, R, t, *s);
}
int hh_umeyama(const double* src, const double* dst, int64_t n, double* R, double* t, double* s) { return umeyama_impl(src, dst, n, R, t, s, false); }
// the fused pipeline's route (Newton polar iteration instead of the SVD where it applies)
int hh_umeyama_polar(const double* src, const double* dst, int64_t n, double* R, double* t, double* s) { return umeyama_impl(src, dst, n, R, t, s, true); }
// 1 if umeyama_rotation_polar accepts H (else the SVD route is taken)
int hh_polar_applies(const double* H) { double R[9], tr; return umeyama_rotation_polar(H, R, tr) ? 1 : 0; }

void hh_sincos(const double* x, int64_t n, double* s, double* c) { for (int64_t i = 0; i < n; ++i) gsf_sincos(x[i], s[i], c[i]); }

void hh_utm_forward(const double* lat, const double* lon, int64_t n, int zone, int south, double* e, double* nn)
{
    TmConsts c = tm_consts();
    for (int64_t i = 0; i < n; ++i) utm_forward_point(c, lat[i], lon[i], 6.0 * zone - 183.0, south ? 1e7 : 0.0, e[i], nn[i]);
}
void hh_utm_inverse(const double* e, const double* nn, int64_t n, int zone, int south, double* lat, double* lon)
{
    TmConsts c = tm_consts();
    for (int64_t i = 0; i < n; ++i) utm_inverse_point(c, e[i], nn[i], 6.0 * zone - 183.0, south ? 1e7 : 0.0, lat[i], lon[i]);
}
void hh_quat_from_matrix(const double* M, double* q) { Quat r = quat_from_matrix(M); q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
}
