// gsf_eval.hip -- the reference's trajectory error metric (main_process_gui step 6, EKFGPSSLAM.py:1013-1033; SURVEY Q15 /
// 8f "next-4"): for every SLAM index with a valid aligned GNSS fix and t > t[0] + skip (5 s), the MINIMUM Euclidean
// distance from the evaluated trajectory point to ANY such candidate fix (cdist + min, :1030-1031), then mean / median /
// RMSE (:1033).  O(M^2) per trajectory: one 256-thread block per trajectory, candidates re-read from L2 (they are the same
// M x 24 B for all queries of the block); the median comes from a rank count over the M errors (no sort).
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

constexpr int EVAL_THREADS = 256;

__device__ __forceinline__ double block_reduce_sum(double v, double* sh, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(EVAL_THREADS) void eval_errors_kernel(const double* __restrict__ ts, const double* __restrict__ traj,
                                                                   const double* __restrict__ gps, const uint8_t* __restrict__ valid,
                                                                   int64_t N, double skip, double* __restrict__ stats,
                                                                   double* __restrict__ errors)
{
    __shared__ double sh[EVAL_THREADS / 64];
    __shared__ double sh_med[2];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const double* t = ts + b * N; const double* p = traj + b * N * 3; const double* g = gps + b * N * 3;
    const uint8_t* v = valid + b * N;
    double* e = errors + b * N;
    const double thr = t[0] + skip;                                      // :1018
    // candidate / query set: valid, finite fix, after the first `skip` seconds (:1016-1021)
    auto in_set = [&](int64_t i) { return v[i] != 0 && t[i] > thr && !(isnan(g[i * 3]) || isnan(g[i * 3 + 1]) || isnan(g[i * 3 + 2])); };
    double cnt = 0.0, sum = 0.0, sum2 = 0.0;
    for (int64_t i = tid; i < N; i += EVAL_THREADS) {
        double err = NAN;
        if (in_set(i)) {
            const double x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
            double best = INFINITY;
            // the candidate rows are wave-uniform scalar loads: branch-free and unrolled so that several are in flight at once
#pragma unroll 8
            for (int64_t j = 0; j < N; ++j) {                            // :1030-1031
                const double dx = x - g[j * 3], dy = y - g[j * 3 + 1], dz = z - g[j * 3 + 2];
                const double d2 = dx * dx + dy * dy + dz * dz;
                best = fmin(best, in_set(j) ? d2 : INFINITY);
            }
            err = sqrt(best);
            cnt += 1.0; sum += err; sum2 += err * err;
        }
        e[i] = err;
    }
    const double M = block_reduce_sum(cnt, sh, tid);
    const double S = block_reduce_sum(sum, sh, tid);
    const double S2 = block_reduce_sum(sum2, sh, tid);
    __syncthreads();                                                     // errors[] of this block visible to the rank pass (same CU)
    if (tid == 0) { sh_med[0] = NAN; sh_med[1] = NAN; }
    __syncthreads();
    const int64_t Mi = (int64_t)M;
    if (Mi > 0) {
        const int64_t k_lo = (Mi - 1) / 2, k_hi = Mi / 2;               // np.median: mean of the two middle order statistics
        for (int64_t i = tid; i < N; i += EVAL_THREADS) {
            const double ei = e[i];
            if (isnan(ei)) continue;
            int64_t rank = 0;
#pragma unroll 8
            for (int64_t j = 0; j < N; ++j) { const double ej = e[j]; rank += (ej < ei || (ej == ei && j < i)) ? 1 : 0; }
            if (rank == k_lo) sh_med[0] = ei;
            if (rank == k_hi) sh_med[1] = ei;
        }
    }
    __syncthreads();
    if (tid == 0) {
        stats[b * 4] = M;
        stats[b * 4 + 1] = Mi > 0 ? S / M : NAN;
        stats[b * 4 + 2] = Mi > 0 ? 0.5 * (sh_med[0] + sh_med[1]) : NAN;
        stats[b * 4 + 3] = Mi > 0 ? sqrt(S2 / M) : NAN;
    }
}

}  // namespace

extern "C" {

int gsf_eval_errors_batch_dev(gsf_ctx* ctx, const double* ts, const double* traj_pos, const double* aligned_gps, const uint8_t* valid,
                              int64_t B, int64_t N, double skip_seconds, double* stats, double* errors)
{
    GSF_REQUIRE(ctx && B >= 0 && N >= 0 && B <= 0x7fffffff, "bad arguments");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && traj_pos && aligned_gps && valid && stats && errors, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(eval_errors_kernel, dim3((unsigned)B), dim3(EVAL_THREADS), 0, ctx->stream, ts, traj_pos, aligned_gps, valid, N, skip_seconds, stats, errors);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_eval_errors_batch(gsf_ctx* ctx, const double* ts, const double* traj_pos, const double* aligned_gps, const uint8_t* valid,
                          int64_t B, int64_t N, double skip_seconds, double* stats, double* errors)
{
    GSF_REQUIRE(ctx && B >= 0 && N >= 0, "bad arguments");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && traj_pos && aligned_gps && valid && stats, "NULL array");
    const size_t P = (size_t)B * (size_t)N;
    Staging st(ctx, P * 65 + (size_t)B * 32, 6);
    if (st.rc()) return st.rc();
    const double* dts = st.in(ts, P); const double* dp = st.in(traj_pos, P * 3); const double* dg = st.in(aligned_gps, P * 3);
    const uint8_t* dv = st.in(valid, P);
    double* dst = st.out(stats, (size_t)B * 4); double* de = st.out(errors, P);
    int rc = st.upload();
    if (rc) return rc;
    rc = gsf_eval_errors_batch_dev(ctx, dts, dp, dg, dv, B, N, skip_seconds, dst, de);
    if (rc) return rc;
    return st.finish();
}

}  // extern "C"
