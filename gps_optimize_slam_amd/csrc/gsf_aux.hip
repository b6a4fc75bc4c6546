// gsf_aux.hip -- the small helper functions of the reference's EKF surface as device kernels, so that every function of
// SURVEY 8(a)/(b) has a gfx950 implementation behind the C ABI:
//   calculate_relative_pose      EKFGPSSLAM.py:77-92     lane per pose pair
//   quaternion_nlerp             EKFGPSSLAM.py:94-105    lane per pair
//   is_sharp_turn_in_segment     EKFGPSSLAM.py:808-826   wave per segment (max-reduction of the yaw rates)
//   ExtendedKalmanFilter.process_step  :736-772          general DENSE 7x7 form (a caller may hand in any covariance)
//   rts_smoother_segment         EKFGPSSLAM.py:777-803   general dense form, lane per segment
// These are API-completeness paths (the batched hot path is gsf_ekf*.hip); they favour fidelity over speed and keep the
// dense algebra of the reference: 7x7 products and Gauss-Jordan inverses in per-lane arrays.
#include <string.h>

#include "gsf_internal.hpp"

using namespace gsf;

namespace {

__device__ void mm(const double* A, const double* B, double* C, int n, int k, int m)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * m + j];
            C[i * m + j] = s;
        }
}
__device__ void symmetrize(double* A, int n)
{
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) { double v = (A[i * n + j] + A[j * n + i]) / 2.0; A[i * n + j] = v; A[j * n + i] = v; }
}
// Gauss-Jordan with partial pivoting (np.linalg.inv); false if singular
__device__ bool inv_n(const double* A, double* X, int n)
{
    double W[7 * 14];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { W[i * 2 * n + j] = A[i * n + j]; W[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < n; ++c) {
        int p = c; double best = fabs(W[c * 2 * n + c]);
        for (int r = c + 1; r < n; ++r) { double v = fabs(W[r * 2 * n + c]); if (v > best) { best = v; p = r; } }
        if (!(best > 0.0)) return false;
        if (p != c) for (int j = 0; j < 2 * n; ++j) { double t = W[c * 2 * n + j]; W[c * 2 * n + j] = W[p * 2 * n + j]; W[p * 2 * n + j] = t; }
        const double piv = W[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) W[c * 2 * n + j] /= piv;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = W[r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) W[r * 2 * n + j] -= f * W[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) X[i * n + j] = W[i * 2 * n + n + j];
    return true;
}
// SciPy-exact unit quaternion (sqrt + divide: these paths are compared at 1e-15 against the reference's helpers)
__device__ bool unit_exact(const double* q, Quat& o)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(n > 0.0) || !(n < INFINITY)) return false;
    o = Quat{ q[0] / n, q[1] / n, q[2] / n, q[3] / n };
    return true;
}
__device__ Vec3 rotate_matrix_form(const Quat& q, const Vec3& v)        // Rotation.apply via as_matrix (ref :89, :707)
{
    const double x2 = q.x * q.x, y2 = q.y * q.y, z2 = q.z * q.z, w2 = q.w * q.w;
    const double xy = q.x * q.y, zw = q.z * q.w, xz = q.x * q.z, yw = q.y * q.w, yz = q.y * q.z, xw = q.x * q.w;
    return Vec3{ (x2 - y2 - z2 + w2) * v.x + 2.0 * (xy - zw) * v.y + 2.0 * (xz + yw) * v.z,
                 2.0 * (xy + zw) * v.x + (-x2 + y2 - z2 + w2) * v.y + 2.0 * (yz - xw) * v.z,
                 2.0 * (xz - yw) * v.x + 2.0 * (yz + xw) * v.y + (-x2 - y2 + z2 + w2) * v.z };
}
__device__ void normalize_exact(double* q)                               // ref :697-700
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 1e-9) { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; } else { q[0] = 0; q[1] = 0; q[2] = 0; q[3] = 1; }
}

__global__ void relative_pose_kernel(const double* p1, const double* q1, const double* p2, const double* q2, int64_t n, double* dp,
                                     double* dq, int32_t* bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Quat r1, r2;
    const bool ok1 = unit_exact(q1 + i * 4, r1), ok2 = unit_exact(q2 + i * 4, r2);
    const bool ok = ok1 && ok2;
    if (!ok) {                                                           // ref :84-86
        dp[i * 3] = dp[i * 3 + 1] = dp[i * 3 + 2] = 0.0; dq[i * 4] = dq[i * 4 + 1] = dq[i * 4 + 2] = 0.0; dq[i * 4 + 3] = 1.0;
        if (bad) bad[i] = 1;
        return;
    }
    const Quat r1i = quat_conj(r1);
    const Vec3 d = rotate_matrix_form(r1i, Vec3{ p2[i * 3] - p1[i * 3], p2[i * 3 + 1] - p1[i * 3 + 1], p2[i * 3 + 2] - p1[i * 3 + 2] });
    const Quat o = quat_mul(r1i, r2);
    dp[i * 3] = d.x; dp[i * 3 + 1] = d.y; dp[i * 3 + 2] = d.z;
    dq[i * 4] = o.x; dq[i * 4 + 1] = o.y; dq[i * 4 + 2] = o.z; dq[i * 4 + 3] = o.w;
    if (bad) bad[i] = 0;
}

__device__ void nlerp_exact(const double* q1, const double* q2in, double wq2, double* out)   // ref :94-105
{
    double q2[4] = { q2in[0], q2in[1], q2in[2], q2in[3] };
    const double dot = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
    if (dot < 0.0) for (int k = 0; k < 4; ++k) q2[k] = -q2[k];
    const double w = fmin(fmax(wq2, 0.0), 1.0);
    double qi[4];
    for (int k = 0; k < 4; ++k) qi[k] = (1.0 - w) * q1[k] + w * q2[k];
    const double n = sqrt(qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3]);
    if (n < 1e-9) { for (int k = 0; k < 4; ++k) out[k] = (wq2 < 0.5) ? q1[k] : q2[k]; return; }
    for (int k = 0; k < 4; ++k) out[k] = qi[k] / n;
}
__global__ void nlerp_kernel(const double* q1, const double* q2, const double* w, int64_t n, double* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) nlerp_exact(q1 + i * 4, q2 + i * 4, w[i], out + i * 4);
}

// one wave per segment; lanes stride over the pairs, max-reduce the yaw rates (faithful atan2 form, ref :819-824)
__global__ __launch_bounds__(64) void sharp_turn_kernel(const double* quats, const double* stamps, const int64_t* offsets, double thr,
                                                        int32_t* result, double* max_rate)
{
    const int64_t b = blockIdx.x;
    const int64_t i0 = offsets[b], i1 = offsets[b + 1];
    double mx = 0.0; int badq = 0;
    for (int64_t i = i0 + 1 + threadIdx.x; i < i1; i += 64) {
        const double t1 = stamps[i - 1], t2 = stamps[i];
        if (t2 <= t1) continue;                                          // ref :817
        Quat a, c;
        const bool oka = unit_exact(quats + (i - 1) * 4, a), okc = unit_exact(quats + i * 4, c);
        if (!(oka && okc)) { badq = 1; continue; }                      // ref :821
        const double y1 = quat_yaw_zyx(a), y2 = quat_yaw_zyx(c);
        const double dy = atan2(sin(y2 - y1), cos(y2 - y1));             // ref :822
        mx = fmax(mx, fabs(dy / (t2 - t1)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o, 64)); badq |= __shfl_xor(badq, o, 64); }
    if (threadIdx.x == 0) {
        const bool r = (i1 - i0 >= 2) && (badq || mx > thr);              // ref :812, :826
        result[b] = r ? 1 : 0;
        if (max_rate) max_rate[b] = mx;
    }
}

struct StepIO {            // ExtendedKalmanFilter state around one process_step (all dense)
    double state[7], cov[49], Qps[49], R[9];
    int32_t gnss_prev;     // -1 None, 0 False, 1 True
    double weight;
    int32_t current_steps;
};

__global__ void process_step_kernel(StepIO* io, const double* dpl, const double* dquat, const double* z, int has_meas, int avail, double dt,
                                    int override_steps, double* pred_state, double* pred_cov)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    StepIO& f = *io;
    const int eff = override_steps >= 0 ? override_steps : f.current_steps;          // ref :742
    const double wdelta = eff > 0 ? 1.0 / (double)eff : 1.0;                         // ref :743
    // _predict, ref :702-715
    Quat qn, dqn;
    unit_exact(f.state + 3, qn); unit_exact(dquat, dqn);
    const Vec3 rp = rotate_matrix_form(qn, Vec3{ dpl[0], dpl[1], dpl[2] });
    double ps[7] = { f.state[0] + rp.x, f.state[1] + rp.y, f.state[2] + rp.z, 0, 0, 0, 0 };
    const Quat pq = quat_mul(qn, dqn);
    ps[3] = pq.x; ps[4] = pq.y; ps[5] = pq.z; ps[6] = pq.w;
    normalize_exact(ps + 3);
    const double dta = fmax(fabs(dt), 1e-6);
    double pc[49];
    for (int i = 0; i < 49; ++i) pc[i] = f.cov[i] + f.Qps[i] * dta;
    symmetrize(pc, 7);
    for (int i = 0; i < 7; ++i) pred_state[i] = ps[i];
    for (int i = 0; i < 49; ++i) pred_cov[i] = pc[i];
    // _update, ref :717-734
    double us[7], uc[49]; bool ok = false;
    if (avail && has_meas && !(isnan(z[0]) || isnan(z[1]) || isnan(z[2]))) {
        double H[21] = { 0 }, HT[21];
        H[0] = 1; H[8] = 1; H[16] = 1;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 7; ++j) HT[j * 3 + i] = H[i * 7 + j];
        const double y[3] = { z[0] - ps[0], z[1] - ps[1], z[2] - ps[2] };
        double HP[21], S[9], Si[9];
        mm(H, pc, HP, 3, 7, 7); mm(HP, HT, S, 3, 7, 3);
        for (int i = 0; i < 9; ++i) S[i] += f.R[i];
        symmetrize(S, 3);
        if (inv_n(S, Si, 3)) {
            double PHT[21], K[21];
            mm(pc, HT, PHT, 7, 7, 3); mm(PHT, Si, K, 7, 3, 3);
            for (int i = 0; i < 7; ++i) us[i] = ps[i] + (K[i * 3] * y[0] + K[i * 3 + 1] * y[1] + K[i * 3 + 2] * y[2]);
            normalize_exact(us + 3);
            double KH[49], IKH[49], IKHT[49], T1[49], T2[49], KR[21], KT[21], KRK[49];
            mm(K, H, KH, 7, 3, 7);
            for (int i = 0; i < 49; ++i) IKH[i] = ((i % 8 == 0) ? 1.0 : 0.0) - KH[i];
            for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) IKHT[j * 7 + i] = IKH[i * 7 + j];
            mm(IKH, pc, T1, 7, 7, 7); mm(T1, IKHT, T2, 7, 7, 7);
            mm(K, f.R, KR, 7, 3, 3);
            for (int i = 0; i < 7; ++i) for (int j = 0; j < 3; ++j) KT[j * 7 + i] = K[i * 3 + j];
            mm(KR, KT, KRK, 7, 3, 7);
            for (int i = 0; i < 49; ++i) uc[i] = T2[i] + KRK[i];
            symmetrize(uc, 7);
            ok = true;
        }
    }
    // weight state machine + fuse, ref :752-768
    const bool just_rec = avail && (f.gnss_prev == 0);
    if (avail) {
        if (just_rec || eff == 0) f.weight = (eff == 0) ? 1.0 : wdelta;
        else if (f.weight < 1.0) f.weight = fmin(1.0, f.weight + wdelta);
    } else f.weight = 0.0;
    if (avail && ok) {
        if (f.weight < 1.0 && eff > 0) {
            const double w = f.weight;
            for (int i = 0; i < 3; ++i) f.state[i] = (1.0 - w) * ps[i] + w * us[i];
            nlerp_exact(ps + 3, us + 3, w, f.state + 3);
        } else for (int i = 0; i < 7; ++i) f.state[i] = us[i];
        for (int i = 0; i < 49; ++i) f.cov[i] = uc[i];
    } else {
        for (int i = 0; i < 7; ++i) f.state[i] = ps[i];
        for (int i = 0; i < 49; ++i) f.cov[i] = pc[i];
    }
    f.gnss_prev = avail ? 1 : 0;
}

// lane per segment: segments are offsets[b]..offsets[b+1] rows of xf/xp (x7) and Pf/Pp (x49)
__global__ void rts_segment_kernel(const double* xf, const double* Pf, const double* xp, const double* Pp, const int64_t* offsets, int64_t B,
                                   double* xs, double* Ps)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t i0 = offsets[b], L = offsets[b + 1] - i0;
    if (L <= 0) return;
    const int64_t last = i0 + L - 1;
    for (int k = 0; k < 7; ++k) xs[last * 7 + k] = xf[last * 7 + k];                  // ref :782
    for (int k = 0; k < 49; ++k) Ps[last * 49 + k] = Pf[last * 49 + k];
    for (int64_t k = last - 1; k >= i0; --k) {                                        // ref :784
        const double* Pk = Pf + k * 49; const double* Pp1 = Pp + (k + 1) * 49;
        double inv[49], A[49];
        if (!inv_n(Pp1, inv, 7)) {                                                    // ref :787-796
            for (int c = 0; c < 7; ++c) xs[k * 7 + c] = xf[k * 7 + c];
            for (int c = 0; c < 49; ++c) Ps[k * 49 + c] = Pk[c];
            continue;
        }
        mm(Pk, inv, A, 7, 7, 7);                                                      // ref :789
        double d[7];
        for (int c = 0; c < 7; ++c) d[c] = xs[(k + 1) * 7 + c] - xp[(k + 1) * 7 + c];
        for (int r = 0; r < 7; ++r) {
            double acc = 0.0;
            for (int c = 0; c < 7; ++c) acc += A[r * 7 + c] * d[c];
            xs[k * 7 + r] = xf[k * 7 + r] + acc;                                      // ref :798
        }
        normalize_exact(xs + k * 7 + 3);                                              // ref :799
        double Dm[49], T1[49], T2[49], AT[49];
        for (int c = 0; c < 49; ++c) Dm[c] = Ps[(k + 1) * 49 + c] - Pp1[c];
        for (int r = 0; r < 7; ++r) for (int c = 0; c < 7; ++c) AT[c * 7 + r] = A[r * 7 + c];
        mm(A, Dm, T1, 7, 7, 7); mm(T1, AT, T2, 7, 7, 7);
        for (int c = 0; c < 49; ++c) Ps[k * 49 + c] = Pk[c] + T2[c];                  // ref :801
        symmetrize(Ps + k * 49, 7);                                                   // ref :802
    }
}

}  // namespace

#define ST_BEGIN(bytes, n) Staging st(ctx, (bytes), (n)); if (st.rc()) return st.rc()
#define ST_UPLOAD() do { int rc__ = st.upload(); if (rc__) return rc__; } while (0)

extern "C" {

int gsf_relative_pose_batch(gsf_ctx* ctx, const double* p1, const double* q1, const double* p2, const double* q2, int64_t n, double* dp,
                            double* dq, int32_t* bad)
{
    GSF_REQUIRE(ctx && n >= 0 && (n == 0 || (p1 && q1 && p2 && q2 && dp && dq)), "bad arguments");
    if (n == 0) return GSF_OK;
    ST_BEGIN((size_t)n * (21 * 8 + 4), 7);
    const double* a = st.in(p1, (size_t)n * 3); const double* dq1 = st.in(q1, (size_t)n * 4);
    const double* dp2 = st.in(p2, (size_t)n * 3); const double* dq2 = st.in(q2, (size_t)n * 4);
    double* ddp = st.out(dp, (size_t)n * 3); double* ddq = st.out(dq, (size_t)n * 4); int32_t* dbad = st.out(bad, (size_t)n);
    ST_UPLOAD();
    hipLaunchKernelGGL(relative_pose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, dq1, dp2, dq2, n, ddp, ddq, dbad);
    GSF_HIP(hipGetLastError());
    return st.finish();
}

int gsf_quaternion_nlerp_batch(gsf_ctx* ctx, const double* q1, const double* q2, const double* w, int64_t n, double* out)
{
    GSF_REQUIRE(ctx && n >= 0 && (n == 0 || (q1 && q2 && w && out)), "bad arguments");
    if (n == 0) return GSF_OK;
    ST_BEGIN((size_t)n * 13 * 8, 4);
    const double* a = st.in(q1, (size_t)n * 4); const double* b = st.in(q2, (size_t)n * 4); const double* dw = st.in(w, (size_t)n);
    double* o = st.out(out, (size_t)n * 4);
    ST_UPLOAD();
    hipLaunchKernelGGL(nlerp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, b, dw, n, o);
    GSF_HIP(hipGetLastError());
    return st.finish();
}

int gsf_is_sharp_turn_batch(gsf_ctx* ctx, const double* quats, const double* stamps, const int64_t* offsets, int64_t B,
                            double yaw_rate_threshold_rad_per_sec, int32_t* result, double* max_rate)
{
    GSF_REQUIRE(ctx && offsets && result && B >= 0 && B <= 0x7fffffff, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (quats && stamps)), "bad offsets / NULL arrays");
    ST_BEGIN((size_t)total * 5 * 8 + (size_t)(B + 1) * 8 + (size_t)B * 12, 5);
    const double* dq = st.in(quats, (size_t)total * 4); const double* dt = st.in(stamps, (size_t)total);
    const int64_t* doff = st.in(offsets, (size_t)B + 1);
    int32_t* dres = st.out(result, (size_t)B);
    double* dmr = max_rate ? st.out(max_rate, (size_t)B) : st.tmp<double>((size_t)B);
    ST_UPLOAD();
    hipLaunchKernelGGL(sharp_turn_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, dq, dt, doff, yaw_rate_threshold_rad_per_sec, dres, dmr);
    GSF_HIP(hipGetLastError());
    return st.finish();
}

int gsf_ekf_process_step(gsf_ctx* ctx, double* state, double* cov, const double* process_noise_per_sec, const double* meas_noise,
                         int32_t* gnss_available_prev, double* gnss_update_weight, int32_t current_transition_steps,
                         const double* delta_pos_local, const double* delta_quat, const double* gps_meas, int32_t gnss_is_available,
                         double delta_time, int32_t override_transition_steps, double* pred_state, double* pred_cov)
{
    GSF_REQUIRE(ctx && state && cov && process_noise_per_sec && meas_noise && gnss_available_prev && gnss_update_weight && delta_pos_local &&
                delta_quat && pred_state && pred_cov, "NULL argument");
    StepIO h;
    memcpy(h.state, state, sizeof h.state); memcpy(h.cov, cov, sizeof h.cov);
    memcpy(h.Qps, process_noise_per_sec, sizeof h.Qps); memcpy(h.R, meas_noise, sizeof h.R);
    h.gnss_prev = *gnss_available_prev; h.weight = *gnss_update_weight; h.current_steps = current_transition_steps;
    ST_BEGIN(2 * sizeof(StepIO) + (3 + 4 + 3 + 7 + 49) * 8, 8);
    const double znan[3] = { NAN, NAN, NAN };
    const StepIO* din = st.in(&h, 1);
    const double* ddp = st.in(delta_pos_local, 3); const double* ddq = st.in(delta_quat, 4); const double* dz = st.in(gps_meas ? gps_meas : znan, 3);
    StepIO* dio = st.out(&h, 1); double* dps = st.out(pred_state, 7); double* dpc = st.out(pred_cov, 49);
    ST_UPLOAD();
    GSF_HIP(hipMemcpyAsync(dio, din, sizeof(StepIO), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(process_step_kernel, dim3(1), dim3(64), 0, ctx->stream, dio, ddp, ddq, dz, gps_meas ? 1 : 0, gnss_is_available ? 1 : 0,
                       delta_time, override_transition_steps, dps, dpc);
    GSF_HIP(hipGetLastError());
    int rc = st.finish();
    if (rc) return rc;
    memcpy(state, h.state, sizeof h.state); memcpy(cov, h.cov, sizeof h.cov);
    *gnss_available_prev = h.gnss_prev; *gnss_update_weight = h.weight;
    return GSF_OK;
}

int gsf_rts_smoother_segment_batch(gsf_ctx* ctx, const double* states_filt, const double* covs_filt, const double* states_pred,
                                   const double* covs_pred, const int64_t* offsets, int64_t B, double* states_smooth, double* covs_smooth)
{
    GSF_REQUIRE(ctx && offsets && B >= 0, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (states_filt && covs_filt && states_pred && covs_pred && states_smooth && covs_smooth)), "NULL arrays");
    if (total == 0) return GSF_OK;
    ST_BEGIN((size_t)total * (3 * 7 + 3 * 49) * 8 + (size_t)(B + 1) * 8, 7);
    const double* xf = st.in(states_filt, (size_t)total * 7); const double* xp = st.in(states_pred, (size_t)total * 7);
    const double* Pf = st.in(covs_filt, (size_t)total * 49); const double* Pp = st.in(covs_pred, (size_t)total * 49);
    const int64_t* doff = st.in(offsets, (size_t)B + 1);
    double* xs = st.out(states_smooth, (size_t)total * 7); double* Ps = st.out(covs_smooth, (size_t)total * 49);
    ST_UPLOAD();
    hipLaunchKernelGGL(rts_segment_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, xf, Pf, xp, Pp, doff, B, xs, Ps);
    GSF_HIP(hipGetLastError());
    return st.finish();
}

}  // extern "C"
