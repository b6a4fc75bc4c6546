// gsf_robust.hip -- steps 3-5 of main_process_gui (EKFGPSSLAM.py:1002-1010) with the reference's ROBUST fit, as one device chain
// without a host round trip:
//   compact      rows with valid finite GNSS -> (src, dst) point sets in fixed-stride slots            (ref :977-998: the valid indices)
//   draw         max_trials x np.random.choice(n, min_samples, replace=False) per trajectory          (ref :405; gsf_rng.hip)
//   K2b          hypotheses, first-best inlier set, final Umeyama on the inliers                      (ref :404-421; gsf_sim3.hip)
//   init pose    Sim3 of pose 0 (the only pose of the aligned track the filter consumes)    (ref :1006, :842)
//   K4           EKF + per-outage RTS, wave per trajectory                                              (ref :1010; gsf_ekf_wave.hip)
//   finish       fit status into the status word, inlier mask back to original rows, NaN rows for a fit that is None
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

// one wave per trajectory: stable compaction of the rows with valid, finite GNSS into slot [b*N, b*N + n_b)
__global__ __launch_bounds__(64) void compact_valid_kernel(const double* __restrict__ pos, const double* __restrict__ gps, const uint8_t* __restrict__ valid,
                                                           int64_t B, int64_t N, double* __restrict__ src, double* __restrict__ dst,
                                                           int32_t* __restrict__ rowmap, int32_t* __restrict__ counts, int64_t* __restrict__ offsets)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, base = b * N;
    int n = 0;
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        const int64_t i = c0 + lane;
        bool ok = false;
        double z0 = 0, z1 = 0, z2 = 0;
        if (i < N) {
            z0 = gps[(base + i) * 3]; z1 = gps[(base + i) * 3 + 1]; z2 = gps[(base + i) * 3 + 2];
            ok = valid[base + i] != 0 && !(isnan(z0) || isnan(z1) || isnan(z2));
        }
        const u64 m = __ballot(ok);
        if (ok) {
            const int k = n + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            const int64_t o = base + k;
            src[o * 3] = pos[(base + i) * 3]; src[o * 3 + 1] = pos[(base + i) * 3 + 1]; src[o * 3 + 2] = pos[(base + i) * 3 + 2];
            dst[o * 3] = z0; dst[o * 3 + 1] = z1; dst[o * 3 + 2] = z2;
            rowmap[o] = (int32_t)i;
        }
        n += __popcll(m);
    }
    if (lane == 0) { counts[b] = n; offsets[b] = base; if (b == B - 1) offsets[B] = B * N; }
}

// lane per trajectory: Sim3 of pose 0 (transform_trajectory row 0, ref :464-466)
__global__ __launch_bounds__(64) void robust_init_pose_kernel(const double* __restrict__ pos, const double* __restrict__ quat, int64_t B, int64_t N,
                                                              const double* __restrict__ R, const double* __restrict__ t, const double* __restrict__ s,
                                                              const int32_t* __restrict__ fit, double* __restrict__ init_pos, double* __restrict__ init_quat,
                                                              int32_t* __restrict__ fail)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    Quat qn; const bool qok = quat_unit(Quat{ quat[b * N * 4], quat[b * N * 4 + 1], quat[b * N * 4 + 2], quat[b * N * 4 + 3] }, qn);
    const bool none = (fit[b] & 1) != 0;
    double Rb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rb[k] = R[b * 9 + k];
    const double x = pos[b * N * 3], y = pos[b * N * 3 + 1], z = pos[b * N * 3 + 2], sb = s[b];
    Vec3 p0{ sb * (x * Rb[0] + y * Rb[1] + z * Rb[2]) + t[b * 3], sb * (x * Rb[3] + y * Rb[4] + z * Rb[5]) + t[b * 3 + 1],
             sb * (x * Rb[6] + y * Rb[7] + z * Rb[8]) + t[b * 3 + 2] };
    Quat q0 = quat_mul(quat_from_matrix(Rb), qn);
    if (none || !qok) { p0 = Vec3{ 0.0, 0.0, 0.0 }; q0 = Quat{ 0.0, 0.0, 0.0, 1.0 }; }      // the filter still runs; finish() blanks the rows
    init_pos[b * 3] = p0.x; init_pos[b * 3 + 1] = p0.y; init_pos[b * 3 + 2] = p0.z;
    init_quat[b * 4] = q0.x; init_quat[b * 4 + 1] = q0.y; init_quat[b * 4 + 2] = q0.z; init_quat[b * 4 + 3] = q0.w;
    fail[b] = (none ? 1 : 0) | (qok ? 0 : 2);
}

// one wave per trajectory: status word, inlier mask in original row order, NaN rows when the fit is None / pose 0 is invalid
__global__ __launch_bounds__(64) void robust_finish_kernel(int64_t N, const int32_t* __restrict__ fit, const int32_t* __restrict__ fail,
                                                           const int32_t* __restrict__ counts, const int32_t* __restrict__ rowmap,
                                                           const uint8_t* __restrict__ mask_c, uint8_t* __restrict__ inlier_mask,
                                                           double* __restrict__ pos_out, double* __restrict__ quat_out, int32_t* __restrict__ status)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, base = b * N;
    const int f = fail[b];
    if (inlier_mask) {
        for (int64_t i = lane; i < N; i += 64) inlier_mask[base + i] = 0;
        __syncthreads();
        const int n = counts[b];
        for (int k = lane; k < n; k += 64) if (mask_c[base + k]) inlier_mask[base + rowmap[base + k]] = 1;
    }
    if (f != 0) {
        for (int64_t i = lane; i < N; i += 64) {
            pos_out[(base + i) * 3] = NAN; pos_out[(base + i) * 3 + 1] = NAN; pos_out[(base + i) * 3 + 2] = NAN;
            quat_out[(base + i) * 4] = NAN; quat_out[(base + i) * 4 + 1] = NAN; quat_out[(base + i) * 4 + 2] = NAN; quat_out[(base + i) * 4 + 3] = NAN;
        }
        if (lane == 0) status[b] = ((f & 1) ? (SIM3_NONE << 8) : 0) | ((f & 2) ? ST_BAD_QUAT : 0);
    } else if (lane == 0) {
        status[b] = (status[b] & 0xff) | (fit[b] << 8);
    }
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int gsf_fuse_pipeline_robust_batch_dev(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps,
                                                  const uint8_t* valid, const gsf_ekf_config* cfg, int64_t B, int64_t N, int32_t min_samples,
                                                  double residual_threshold, int32_t max_trials, int32_t min_inliers_needed, uint32_t* mt_state,
                                                  double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status,
                                                  int32_t* n_inliers, uint8_t* inlier_mask)
{
    GSF_REQUIRE(ctx && cfg, "ctx/cfg is NULL");
    GSF_REQUIRE(B >= 0 && N >= 0 && B <= 0x7fffffff, "bad B or N");
    GSF_REQUIRE(min_samples >= 1 && min_samples <= 64 && max_trials >= 0 && max_trials <= (1 << 20), "min_samples must be in [1,64], max_trials in [0, 2^20]");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(N <= 28000, "N too large for the device-side draws (<= 28000 poses per trajectory)");
    GSF_REQUIRE(ts && pos && quat && gps && valid && mt_state && R && t && s && pos_out && quat_out && status && n_inliers, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    const size_t P = (size_t)B * (size_t)N, nb = (size_t)B;
    // workspace (context scratch, grow-only): point sets, row map, compact mask, counts/offsets, sample sets, fit status, initial poses
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off = align_up(off + bytes); return at; };
    const size_t o_src = take(P * 24), o_dst = take(P * 24), o_map = take(P * 4), o_mask = take(P), o_cnt = take(nb * 4), o_off = take((nb + 1) * 8),
                 o_idx = take(nb * (size_t)max_trials * (size_t)min_samples * 4 + 4), o_fit = take(nb * 4), o_fail = take(nb * 4), o_ip = take(nb * 24),
                 o_iq = take(nb * 32);
    int rc = ensure_scratch(ctx, off);
    if (rc) return rc;
    char* w = (char*)ctx->scratch;
    double* src = (double*)(w + o_src); double* dst = (double*)(w + o_dst); int32_t* rowmap = (int32_t*)(w + o_map); uint8_t* mask_c = (uint8_t*)(w + o_mask);
    int32_t* counts = (int32_t*)(w + o_cnt); int64_t* offsets = (int64_t*)(w + o_off); int32_t* idx = (int32_t*)(w + o_idx);
    int32_t* fit = (int32_t*)(w + o_fit); int32_t* fail = (int32_t*)(w + o_fail); double* ip = (double*)(w + o_ip); double* iq = (double*)(w + o_iq);
    hipLaunchKernelGGL(compact_valid_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, pos, gps, valid, B, N, src, dst, rowmap, counts, offsets);
    GSF_HIP(hipGetLastError());
    if (max_trials > 0 && (rc = launch_mt_choice(ctx, mt_state, counts, B, max_trials, min_samples, idx, (int32_t)N))) return rc;
    if ((rc = launch_sim3_ransac(ctx, src, dst, offsets, counts, B, idx, max_trials, min_samples, residual_threshold, min_inliers_needed, R, t, s, fit,
                                 mask_c, n_inliers, (int64_t)P))) return rc;
    hipLaunchKernelGGL(robust_init_pose_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, pos, quat, B, N, R, t, s, fit, ip, iq, fail);
    GSF_HIP(hipGetLastError());
    if ((rc = launch_ekf_wave(ctx, false, ts, pos, quat, gps, valid, ip, iq, cfg, B, N, nullptr, nullptr, nullptr, pos_out, quat_out, status))) return rc;
    hipLaunchKernelGGL(robust_finish_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, N, fit, fail, counts, rowmap, mask_c, inlier_mask, pos_out, quat_out, status);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
