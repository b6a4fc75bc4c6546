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

// The same metric for tracks of up to EVAL_LDS_MAX_N poses (every BASELINE config): the candidate set is compacted into LDS once
// (coordinates, original index, later the errors), and the M x M pair work -- nearest fix, then the rank count of the median -- is
// spread evenly: S adjacent lanes share a query (S = a power of two with M*S ~ 2 000 work items for the 256 threads), each walking
// every S-th candidate, and meet in a DPP-free xor butterfly.  min / counts are order-independent, so the results are the
// one-thread-per-query kernel's bit for bit; the sums of the mean / RMSE keep that kernel's order of partial sums only up to the
// last digits (gate 1e-9 m in the tests).  271 poses: 111 -> ~25 us for one track, 139 -> ~60 us for 1 000.
constexpr int EVAL_LDS_MAX_N = 1536;
// up to three trajectories per track against the SAME fixes in one launch (step 6 prints raw SLAM / Sim3 / EKF, ref :1027): blockIdx.y picks
// the set; stats / errors of set k start at k * B * 4 / k * B * N
struct EvalSets { const double* traj[3]; };
__global__ __launch_bounds__(EVAL_THREADS) void eval_errors_lds_kernel(const double* __restrict__ ts, EvalSets sets,
                                                                       const double* __restrict__ gps, const uint8_t* __restrict__ valid,
                                                                       int64_t N, double skip, double* __restrict__ stats,
                                                                       double* __restrict__ errors)
{
    const double* __restrict__ traj = sets.traj[blockIdx.y];
    stats += (int64_t)blockIdx.y * gridDim.x * 4; errors += (int64_t)blockIdx.y * gridDim.x * N;
    extern __shared__ double dynl[];                                     // cx[N], cy[N], cz[N], err[N], then int32 qidx[N]
    __shared__ double sh[EVAL_THREADS / 64];
    __shared__ double sh_med[2];
    __shared__ int sh_cnt[EVAL_THREADS / 64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = (int)N;
    const int64_t b = blockIdx.x;
    const double* t = ts + b * N; const double* p = traj + b * N * 3; const double* g = gps + b * N * 3;
    const uint8_t* v = valid + b * N;
    double* e = errors + b * N;
    double* cx = dynl; double* cy = cx + n; double* cz = cy + n; double* cerr = cz + n;
    int32_t* qidx = (int32_t*)(cerr + n);
    const double thr = t[0] + skip;                                      // :1018
    // ---- candidate / query set (:1016-1021), compacted in row order
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += EVAL_THREADS) {
        const int i = i0 + tid;
        double gx = 0.0, gy = 0.0, gz = 0.0;
        bool in = false;
        if (i < n) {
            gx = g[(int64_t)i * 3]; gy = g[(int64_t)i * 3 + 1]; gz = g[(int64_t)i * 3 + 2];
            in = v[i] != 0 && t[i] > thr && !(isnan(gx) || isnan(gy) || isnan(gz));
            if (!in) e[i] = NAN;
        }
        const unsigned long long m = __ballot(in);
        if (lane == 0) sh_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += sh_cnt[w];
        const int at = off + __popcll(m & ((1ull << lane) - 1ull));
        if (in) { cx[at] = gx; cy[at] = gy; cz[at] = gz; qidx[at] = i; }
        base += sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
        __syncthreads();
    }
    const int M = base;                                                  // block-uniform
    int S = 1;
    while (S < 64 && M * (S * 2) <= 2048) S *= 2;
    const int part = tid & (S - 1);
    // ---- nearest fix of every query (:1030-1031)
    double cnt = 0.0, sum = 0.0, sum2 = 0.0;
    for (int it0 = 0; it0 < M * S; it0 += EVAL_THREADS) {
        const int q = (it0 + tid) / S;                                   // S divides 64 and 256: the S lanes of a query sit side by side in one wave
        const bool live = q < M;
        const int row = live ? qidx[q] : 0;
        const double x = p[(int64_t)row * 3], y = p[(int64_t)row * 3 + 1], z = p[(int64_t)row * 3 + 2];
        double best = INFINITY;
        if (live) {
#pragma unroll 4
            for (int k = part; k < M; k += S) {
                const double dx = x - cx[k], dy = y - cy[k], dz = z - cz[k];
                best = fmin(best, dx * dx + dy * dy + dz * dz);
            }
        }
        for (int o = 1; o < S; o <<= 1) best = fmin(best, __shfl_xor(best, o, 64));
        if (live && part == 0) {
            const double err = sqrt(best);
            cerr[q] = err; e[row] = err;
            cnt += 1.0; sum += err; sum2 += err * err;
        }
    }
    const double Mf = block_reduce_sum(cnt, sh, tid);
    const double S1 = block_reduce_sum(sum, sh, tid);
    const double S2 = block_reduce_sum(sum2, sh, tid);
    if (tid == 0) { sh_med[0] = NAN; sh_med[1] = NAN; }
    __syncthreads();                                                     // cerr[] complete
    // ---- np.median: the two middle order statistics by rank count (ties broken by row order)
    if (M > 0) {
        const int k_lo = (M - 1) / 2, k_hi = M / 2;
        for (int it0 = 0; it0 < M * S; it0 += EVAL_THREADS) {
            const int q = (it0 + tid) / S;
            const bool live = q < M;
            const double ei = live ? cerr[q] : 0.0;
            int rank = 0;
            if (live) {
#pragma unroll 4
                for (int k = part; k < M; k += S) { const double ej = cerr[k]; rank += (ej < ei || (ej == ei && k < q)) ? 1 : 0; }
            }
            for (int o = 1; o < S; o <<= 1) rank += __shfl_xor(rank, o, 64);
            if (live && part == 0 && !isnan(ei)) {
                if (rank == k_lo) sh_med[0] = ei;
                if (rank == k_hi) sh_med[1] = ei;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        stats[b * 4] = Mf;
        stats[b * 4 + 1] = M > 0 ? S1 / Mf : NAN;
        stats[b * 4 + 2] = M > 0 ? 0.5 * (sh_med[0] + sh_med[1]) : NAN;
        stats[b * 4 + 3] = M > 0 ? sqrt(S2 / Mf) : NAN;
    }
}

}  // namespace

namespace gsf {
// step 6 for the three tracks main_process_gui prints (raw SLAM, Sim3, EKF; ref :1027) against the same aligned fixes: ONE launch for tracks
// up to EVAL_LDS_MAX_N poses; stats[3][B][4], errors[3][B][N]
int launch_eval_errors3(gsf_ctx* ctx, const double* ts, const double* traj0, const double* traj1, const double* traj2, const double* aligned_gps,
                        const uint8_t* valid, int64_t B, int64_t N, double skip_seconds, double* stats, double* errors)
{
    if (N <= EVAL_LDS_MAX_N) {
        hipLaunchKernelGGL(eval_errors_lds_kernel, dim3((unsigned)B, 3), dim3(EVAL_THREADS), (size_t)N * 36, ctx->stream, ts, EvalSets{ { traj0, traj1, traj2 } },
                           aligned_gps, valid, N, skip_seconds, stats, errors);
        GSF_HIP(hipGetLastError());
        return GSF_OK;
    }
    const double* tr[3] = { traj0, traj1, traj2 };
    for (int k = 0; k < 3; ++k) {
        hipLaunchKernelGGL(eval_errors_kernel, dim3((unsigned)B), dim3(EVAL_THREADS), 0, ctx->stream, ts, tr[k], aligned_gps, valid, N, skip_seconds,
                           stats + (size_t)k * (size_t)B * 4, errors + (size_t)k * (size_t)B * (size_t)N);
        GSF_HIP(hipGetLastError());
    }
    return GSF_OK;
}
}  // namespace gsf

extern "C" {

int gsf_eval_errors_batch_dev(gsf_ctx* ctx, const double* ts, const double* traj_pos, const double* aligned_gps, const uint8_t* valid,
                              int64_t B, int64_t N, double skip_seconds, double* stats, double* errors)
{
    GSF_REQUIRE(ctx && B >= 0 && N >= 0 && B <= 0x7fffffff, "bad arguments");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && traj_pos && aligned_gps && valid && stats && errors, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    if (N <= EVAL_LDS_MAX_N)
        hipLaunchKernelGGL(eval_errors_lds_kernel, dim3((unsigned)B), dim3(EVAL_THREADS), (size_t)N * 36, ctx->stream, ts, EvalSets{ { traj_pos, nullptr, nullptr } },
                           aligned_gps, valid, N, skip_seconds, stats, errors);
    else
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
