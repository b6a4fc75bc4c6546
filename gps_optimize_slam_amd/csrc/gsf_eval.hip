// gsf_eval.hip -- the reference's trajectory error metric (main_process_gui step 6, EKFGPSSLAM.py:1013-1033; SURVEY Q15 /
// 8f "next-4"): for every SLAM index with a valid aligned GNSS fix and t > t[0] + skip (5 s), the MINIMUM Euclidean
// distance from the evaluated trajectory point to ANY such candidate fix (cdist + min, :1030-1031), then mean / median /
// RMSE (:1033).  O(M^2) per trajectory: one 256-thread block per trajectory, candidates re-read from L2 (they are the same
// M x 24 B for all queries of the block); the median comes from a rank count over the M errors (no sort).
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

constexpr int EVAL_THREADS = 256;

// squared distance of a pose to a fix, the three products in ONE fixed association (explicit fma: the brute-force kernels and the pruned search
// below must form the same bits whatever the compiler would contract).  Two facts the pruned search leans on: the result is >= fl(d_a * d_a)
// for each axis a (non-negative terms added under a monotone rounding).
__device__ __forceinline__ double pair_d2(const double dx, const double dy, const double dz) { return fma(dz, dz, fma(dy, dy, dx * dx)); }


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
                const double d2 = pair_d2(x - g[j * 3], y - g[j * 3 + 1], z - g[j * 3 + 2]);
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

// min / sum over the S adjacent lanes that share a query (S = 1 .. 64, wave-uniform): DPP inside a 16-lane row, ds_bpermute beyond
__device__ __forceinline__ double group_min(double v, const int S)
{
    if (S > 1) v = fmin(v, dpp_quad<0xB1>(v));
    if (S > 2) v = fmin(v, dpp_quad<0x4E>(v));
    if (S > 4) v = fmin(v, dpp_row_xor<4>(v));
    if (S > 8) v = fmin(v, dpp_row_xor<8>(v));
    if (S > 16) v = fmin(v, __shfl_xor(v, 16, 64));
    if (S > 32) v = fmin(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ int group_sum(int v, const int S)
{
    if (S > 1) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    if (S > 2) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    if (S > 4) { int o = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0x5, false); o = __builtin_amdgcn_update_dpp(o, v, 0x114, 0xf, 0xa, false); v += o; }
    if (S > 8) { int o = __builtin_amdgcn_update_dpp(0, v, 0x108, 0xf, 0x3, false); o = __builtin_amdgcn_update_dpp(o, v, 0x118, 0xf, 0xc, false); v += o; }
    if (S > 16) v += __shfl_xor(v, 16, 64);
    if (S > 32) v += __shfl_xor(v, 32, 64);
    return v;
}

// nearest fix of the TQ queries q0 + u * G of this thread over its candidates part, part + S, ... (:1030-1031); queries beyond M compute on
// row 0 and are dropped
template <int TQ>
__device__ __forceinline__ void eval_nearest(const double* __restrict__ p, const double* cx, const double* cy, const double* cz, double* cerr,
                                             const int32_t* qidx, double* __restrict__ e, const int M, const int S, const int part, const int q0, const int G,
                                             double& cnt, double& sum, double& sum2)
{
    double qx[TQ], qy[TQ], qz[TQ], best[TQ];
    int rows[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int q = q0 + u * G;
        rows[u] = q < M ? qidx[q] : 0;
        qx[u] = p[(int64_t)rows[u] * 3]; qy[u] = p[(int64_t)rows[u] * 3 + 1]; qz[u] = p[(int64_t)rows[u] * 3 + 2];
        best[u] = INFINITY;
    }
#pragma unroll 2
    for (int k = part; k < M; k += S) {
        const double gx = cx[k], gy = cy[k], gz = cz[k];
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            best[u] = fmin(best[u], pair_d2(qx[u] - gx, qy[u] - gy, qz[u] - gz));
        }
    }
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const double bu = group_min(best[u], S);
        const int q = q0 + u * G;
        if (q < M && part == 0) {
            const double err = sqrt(bu);
            cerr[q] = err; e[rows[u]] = err;
            cnt += 1.0; sum += err; sum2 += err * err;
        }
    }
}

// ranks of the same queries' errors among all M errors (ties by row order); the two middle order statistics go to sh_med
template <int TQ>
__device__ __forceinline__ void eval_median(const double* cerr, const int M, const int S, const int part, const int q0, const int G, double* sh_med)
{
    const int k_lo = (M - 1) / 2, k_hi = M / 2;
    double ei[TQ];
    int rank[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) { const int q = q0 + u * G; ei[u] = q < M ? cerr[q] : 0.0; rank[u] = 0; }
#pragma unroll 2
    for (int k = part; k < M; k += S) {
        const double ej = cerr[k];
#pragma unroll
        for (int u = 0; u < TQ; ++u) rank[u] += (int)(ej < ei[u]) | ((int)(ej == ei[u]) & (int)(k < q0 + u * G));
    }
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int r = group_sum(rank[u], S);
        if (q0 + u * G < M && part == 0 && !isnan(ei[u])) {
            if (r == k_lo) sh_med[0] = ei[u];
            if (r == k_hi) sh_med[1] = ei[u];
        }
    }
}

// ---- long tracks: the nearest fix by an exact pruned search instead of all M x M pairs (round 5: at 1 000 poses the metric cost ten times the
// fusion it grades: 3.5 ms per 10 000 tracks).  The fixes are sorted along the axis of their largest extent; a query starts at its place in
// that order and walks outwards on both sides, always to the side with the smaller axis gap, until that gap squared is no smaller than the best
// squared distance so far: pair_d2 >= fl(gap^2) for every fix further out on that side, so none of them can lower the minimum -- the minimum is
// the brute-force one, formed from the same pair_d2 bits.  A track that runs along its long axis examines a handful of fixes per pose; one
// that does not (a vertical shaft) degrades towards all pairs, still exact.  The median's order statistics come from a bitonic sort of the
// errors (M (log2 P)^2 / 4 compare-exchanges instead of M^2 compares); an error that is NaN makes the median NaN, as np.median does.
constexpr int EVAL_PRUNE_MIN_M = 400;                                     // up to here the all-pairs tile is faster (36 dependent sorting steps cost more)
constexpr int EVAL_PRUNE_QMAX = 6;                                        // queries per thread: EVAL_LDS_MAX_N / EVAL_THREADS
template <bool WITH_IDX>
__device__ __forceinline__ void eval_bitonic(double* a, uint16_t* ix, const int P, const int tid)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += EVAL_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;   // the pair (i, i ^ j), bit j of i clear
                const double x = a[i], y = a[l];
                if (((i & k) == 0) ? (x > y) : (x < y)) {
                    a[i] = y; a[l] = x;
                    if (WITH_IDX) { const uint16_t u = ix[i]; ix[i] = ix[l]; ix[l] = u; }
                }
            }
            __syncthreads();
        }
    }
}
__device__ __forceinline__ bool eval_pruned(const double* __restrict__ p, const double* cx, const double* cy, const double* cz, double* skey,
                                            const int32_t* qidx, uint16_t* sidx, double* __restrict__ e, const int M, const int tid, double* sh,
                                            double* sh_med, double* __restrict__ stats_b)
{
    __shared__ double sh_ext[EVAL_THREADS / 64][6];
    const int lane = tid & 63, wave = tid >> 6;
    int P = 1;
    while (P < M) P <<= 1;
    // the axis of the largest extent
    double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = tid; i < M; i += EVAL_THREADS) {
        lo[0] = fmin(lo[0], cx[i]); hi[0] = fmax(hi[0], cx[i]); lo[1] = fmin(lo[1], cy[i]); hi[1] = fmax(hi[1], cy[i]);
        lo[2] = fmin(lo[2], cz[i]); hi[2] = fmax(hi[2], cz[i]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int o = 1; o < 64; o <<= 1) { lo[c] = fmin(lo[c], __shfl_xor(lo[c], o, 64)); hi[c] = fmax(hi[c], __shfl_xor(hi[c], o, 64)); }
    if (lane == 0) { for (int c = 0; c < 3; ++c) { sh_ext[wave][c] = lo[c]; sh_ext[wave][3 + c] = hi[c]; } }
    __syncthreads();
    double ext[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double l = sh_ext[0][c], h = sh_ext[0][3 + c];
        for (int w = 1; w < EVAL_THREADS / 64; ++w) { l = fmin(l, sh_ext[w][c]); h = fmax(h, sh_ext[w][3 + c]); }
        ext[c] = h - l;
    }
    const int axis = (ext[0] >= ext[1] && ext[0] >= ext[2]) ? 0 : (ext[1] >= ext[2] ? 1 : 2);   // block-uniform
    // Is there anything to prune?  The walk stops when the axis gap reaches the best distance, and a gap never exceeds the fixes' extent along the
    // axis while the pose lies inside it: a track that is FAR from its fixes -- the raw SLAM track of step 6 sits in another frame, millions of
    // metres away -- would visit every fix from every pose, one at a time (9.7 ms per 10 000 x 1 000 against the all-pairs tile's 3).  Probe: the
    // middle pose's nearest fix by the whole block; farther than twice the extent -> the caller takes the all-pairs tile (same minima either way).
    {
        const int row = qidx[M / 2];
        const double x = p[(int64_t)row * 3], y = p[(int64_t)row * 3 + 1], z = p[(int64_t)row * 3 + 2];
        double b0 = INFINITY;
        for (int i = tid; i < M; i += EVAL_THREADS) b0 = fmin(b0, pair_d2(x - cx[i], y - cy[i], z - cz[i]));
        for (int o = 1; o < 64; o <<= 1) b0 = fmin(b0, __shfl_xor(b0, o, 64));
        __syncthreads();                                                 // sh_ext has been read by every thread
        if (lane == 0) sh_ext[wave][0] = b0;
        __syncthreads();
        double best0 = sh_ext[0][0];
        for (int w = 1; w < EVAL_THREADS / 64; ++w) best0 = fmin(best0, sh_ext[w][0]);
        if (!(best0 <= 4.0 * ext[axis] * ext[axis])) return false;       // block-uniform (NaN poses: the tile treats them as before)
    }
    const double* ca = axis == 0 ? cx : (axis == 1 ? cy : cz);
    for (int i = tid; i < P; i += EVAL_THREADS) { skey[i] = i < M ? ca[i] : INFINITY; sidx[i] = (uint16_t)(i < M ? i : 0); }
    __syncthreads();
    eval_bitonic<true>(skey, sidx, P, tid);
    // the queries: thread tid takes q = tid, tid + 256, ...
    double err_r[EVAL_PRUNE_QMAX];
    double cnt = 0.0, sum = 0.0, sum2 = 0.0;
#pragma unroll
    for (int u = 0; u < EVAL_PRUNE_QMAX; ++u) {
        const int q = tid + u * EVAL_THREADS;
        err_r[u] = INFINITY;
        if (q < M) {
            const int row = qidx[q];
            const double x = p[(int64_t)row * 3], y = p[(int64_t)row * 3 + 1], z = p[(int64_t)row * 3 + 2];
            const double qa = axis == 0 ? x : (axis == 1 ? y : z);
            int a = 0, c = M;                                             // first sorted fix with key >= qa
            while (a < c) { const int m = (a + c) >> 1; if (skey[m] < qa) a = m + 1; else c = m; }
            int l = a - 1, r = a;
            double best = INFINITY;
            for (;;) {
                const double dl = l >= 0 ? qa - skey[l] : INFINITY, dr = r < M ? skey[r] - qa : INFINITY;
                const bool left = dl <= dr;
                const double gap = left ? dl : dr;
                if (!(gap * gap < best)) break;                           // also ends when both sides are used up (inf) or the keys are NaN
                const int j = sidx[left ? l : r];
                best = fmin(best, pair_d2(x - cx[j], y - cy[j], z - cz[j]));
                if (left) --l; else ++r;
            }
            const double err = sqrt(best);
            err_r[u] = err; e[row] = err;
            cnt += 1.0; sum += err; sum2 += err * err;
        }
    }
    const double Mf = block_reduce_sum(cnt, sh, tid);
    const double S1 = block_reduce_sum(sum, sh, tid);
    const double S2 = block_reduce_sum(sum2, sh, tid);
    __syncthreads();                                                     // every search has finished with the keys
    bool bad = false;
#pragma unroll
    for (int u = 0; u < EVAL_PRUNE_QMAX; ++u) { const int q = tid + u * EVAL_THREADS; if (q < P) { skey[q] = err_r[u]; bad = bad || isnan(err_r[u]); } }
    for (int q = tid + EVAL_PRUNE_QMAX * EVAL_THREADS; q < P; q += EVAL_THREADS) skey[q] = INFINITY;
    const bool any_nan = __syncthreads_or(bad ? 1 : 0) != 0;
    if (!any_nan) eval_bitonic<false>(skey, sidx, P, tid);
    if (tid == 0) {
        stats_b[0] = Mf;
        stats_b[1] = S1 / Mf;
        stats_b[2] = any_nan ? NAN : 0.5 * (skey[(M - 1) / 2] + skey[M / 2]);
        stats_b[3] = sqrt(S2 / Mf);
    }
    return true;
}

// The same metric for tracks of up to EVAL_LDS_MAX_N poses (every BASELINE config): the candidate set is compacted into LDS once
// (coordinates, original index, later the errors), and the M x M pair work -- nearest fix, then the rank count of the median -- is
// spread evenly: S adjacent lanes share a query (S = a power of two with M*S ~ 2 000 work items for the 256 threads), each walking
// every S-th candidate, and meet in a DPP-free xor butterfly.  min / counts are order-independent, so the results are the
// one-thread-per-query kernel's bit for bit; the sums of the mean / RMSE keep that kernel's order of partial sums only up to the
// last digits (gate 1e-9 m in the tests).  271 poses: 111 -> ~25 us for one track, 139 -> ~60 us for 1 000.
constexpr int EVAL_LDS_MAX_N = 1536;
#ifdef GSF_EVAL_TIMING   // diagnostic build (make eval_timing): shader-clock stamps of the phases land in errors[b * N + 0 .. 9] instead of the errors
#define EV_T(k) do { if (tid == 0) ev_t[k] = clock64(); } while (0)
#else
#define EV_T(k) do { } while (0)
#endif
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
    // long tracks (n > EVAL_PRUNE_MIN_M): the error array doubles as the sort buffer of P = next power of two >= n doubles, and a uint16 index
    // array of P entries follows qidx (pruned search below); else cx[n], cy[n], cz[n], err[n], int32 qidx[n]
    const bool long_layout = n > EVAL_PRUNE_MIN_M;
    int Pn = 1;
    while (Pn < n) Pn <<= 1;
    double* cx = dynl; double* cy = cx + n; double* cz = cy + n; double* cerr = cz + n;
    int32_t* qidx = (int32_t*)(cerr + (long_layout ? Pn : n));
    uint16_t* sidx = (uint16_t*)(qidx + n);
#ifdef GSF_EVAL_TIMING
    long long ev_t[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const long long ev_w0 = wall_clock64();                              // 100 MHz, common to the chip
#endif
    EV_T(0);
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
    EV_T(1);
    const int M = base;                                                  // block-uniform
    const bool long_track = long_layout && M > EVAL_PRUNE_MIN_M;           // block-uniform
    if (long_track && eval_pruned(p, cx, cy, cz, cerr, qidx, sidx, e, M, tid, sh, sh_med, stats + b * 4)) return;
    // (a long track far from its fixes comes back: all-pairs nearest fix below, then the sorted median)
    int S = 1;
    while (S < 64 && M * (S * 2) <= 2048) S *= 2;
    const int part = tid & (S - 1);
    // Query tile: a thread keeps ALL its queries (one per pass of the old form: q = tid / S + u * 256 / S, at most 8 since M * S <= 2 048)
    // in registers and walks its candidates once, so a candidate's LDS reads serve the whole tile, and the inner loops are straight-line code (the
    // old form's short-circuit tie rule compiled to exec-mask branches: 140 cycles per candidate of the rank pass, in-kernel clocks of the
    // eval_timing build).  min and the rank counts are order-independent and the per-thread sums run over the passes in the old order: the same bits.
    const int G = EVAL_THREADS / S, q0 = tid / S;                        // S divides 64 and 256: the S lanes of a query sit side by side in one wave
    const int passes = (M * S + EVAL_THREADS - 1) / EVAL_THREADS;         // block-uniform, <= 8
    // ---- nearest fix of every query (:1030-1031)
    double cnt = 0.0, sum = 0.0, sum2 = 0.0;
    EV_T(2);
    switch (passes) {                                                    // block-uniform; the tile is exactly as long as the passes
    case 0: case 1: eval_nearest<1>(p, cx, cy, cz, cerr, qidx, e, M, S, part, q0, G, cnt, sum, sum2); break;
#define GSF_EVAL_CASE(T) case T: eval_nearest<T>(p, cx, cy, cz, cerr, qidx, e, M, S, part, q0, G, cnt, sum, sum2); break;
    GSF_EVAL_CASE(2) GSF_EVAL_CASE(3) GSF_EVAL_CASE(4) GSF_EVAL_CASE(5) GSF_EVAL_CASE(6) GSF_EVAL_CASE(7)
#undef GSF_EVAL_CASE
    default: eval_nearest<8>(p, cx, cy, cz, cerr, qidx, e, M, S, part, q0, G, cnt, sum, sum2); break;
    }
    EV_T(4);
    const double Mf = block_reduce_sum(cnt, sh, tid);
    const double S1 = block_reduce_sum(sum, sh, tid);
    const double S2 = block_reduce_sum(sum2, sh, tid);
    if (tid == 0) { sh_med[0] = NAN; sh_med[1] = NAN; }
    __syncthreads();                                                     // cerr[] complete
    EV_T(5);
    // ---- np.median: the two middle order statistics -- long tracks: from a bitonic sort of the errors (the error array has P >= M slots there);
    // else by rank count (ties broken by row order), the same query tile
    if (long_track) {
        int P = 1;
        while (P < M) P <<= 1;
        bool bad = false;
        for (int i = tid; i < P; i += EVAL_THREADS) { if (i >= M) cerr[i] = INFINITY; else bad = bad || isnan(cerr[i]); }
        const bool any_nan = __syncthreads_or(bad ? 1 : 0) != 0;
        if (!any_nan) {
            eval_bitonic<false>(cerr, sidx, P, tid);
            if (tid == 0) { sh_med[0] = cerr[(M - 1) / 2]; sh_med[1] = cerr[M / 2]; }
        }
    } else if (M > 0) {
        switch (passes) {
        case 0: case 1: eval_median<1>(cerr, M, S, part, q0, G, sh_med); break;
#define GSF_EVAL_CASE(T) case T: eval_median<T>(cerr, M, S, part, q0, G, sh_med); break;
        GSF_EVAL_CASE(2) GSF_EVAL_CASE(3) GSF_EVAL_CASE(4) GSF_EVAL_CASE(5) GSF_EVAL_CASE(6) GSF_EVAL_CASE(7)
#undef GSF_EVAL_CASE
        default: eval_median<8>(cerr, M, S, part, q0, G, sh_med); break;
        }
    }
    __syncthreads();
    EV_T(6);
    if (tid == 0) {
        stats[b * 4] = Mf;
        stats[b * 4 + 1] = M > 0 ? S1 / Mf : NAN;
        stats[b * 4 + 2] = M > 0 ? 0.5 * (sh_med[0] + sh_med[1]) : NAN;
        stats[b * 4 + 3] = M > 0 ? sqrt(S2 / Mf) : NAN;
    }
#ifdef GSF_EVAL_TIMING
    EV_T(7);
    __syncthreads();
    if (tid == 0) { for (int k = 0; k < 8; ++k) e[k] = (double)(ev_t[k] - ev_t[0]); e[8] = (double)ev_w0; e[9] = (double)wall_clock64(); }
#endif
}

}  // namespace

namespace gsf {
// dynamic LDS of eval_errors_lds_kernel (its two layouts)
static size_t eval_lds_bytes(int64_t N)
{
    if (N <= EVAL_PRUNE_MIN_M) return (size_t)N * 36;
    size_t P = 1;
    while ((int64_t)P < N) P <<= 1;
    return (size_t)N * 28 + P * 10;
}
// step 6 for the three tracks main_process_gui prints (raw SLAM, Sim3, EKF; ref :1027) against the same aligned fixes: ONE launch for tracks
// up to EVAL_LDS_MAX_N poses; stats[3][B][4], errors[3][B][N]
int launch_eval_errors3(gsf_ctx* ctx, const double* ts, const double* traj0, const double* traj1, const double* traj2, const double* aligned_gps,
                        const uint8_t* valid, int64_t B, int64_t N, double skip_seconds, double* stats, double* errors)
{
    if (N <= EVAL_LDS_MAX_N) {
        hipLaunchKernelGGL(eval_errors_lds_kernel, dim3((unsigned)B, 3), dim3(EVAL_THREADS), eval_lds_bytes(N), ctx->stream, ts, EvalSets{ { traj0, traj1, traj2 } },
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
        hipLaunchKernelGGL(eval_errors_lds_kernel, dim3((unsigned)B), dim3(EVAL_THREADS), eval_lds_bytes(N), ctx->stream, ts, EvalSets{ { traj_pos, nullptr, nullptr } },
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
