// tools/experiments/median_bitonic.hip -- what would the step-6 kernel's MEDIAN cost as a bitonic sort in LDS instead of the M x M rank count?
// (eval_errors_lds_kernel, gsf_eval.hip: after the nearest-fix pass the two middle order statistics of the M errors are found by counting, for
// every error, the errors below it: M^2 compares, 12.7 k of the block's 47 k cycles at M = 220.)  Standalone: B blocks of 256 threads, M errors
// each in LDS; (a) the rank count as the kernel does it (register tile of the queries, S lanes per query), (b) a bitonic sort of the padded
// array, then the two middle elements.  Same medians required; times by HIP events.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/median_bitonic tools/experiments/median_bitonic.hip && /tmp/median_bitonic [B] [M]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 256;

template <int TQ>
__device__ __forceinline__ void rank_tile(const double* cerr, const int M, const int S, const int part, const int q0, const int G, double* med)
{
    const int k_lo = (M - 1) / 2, k_hi = M / 2;
    double ei[TQ]; int rank[TQ];
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
        int r = rank[u];
        for (int o = 1; o < S; o <<= 1) r += __shfl_xor(r, o, 64);
        if (q0 + u * G < M && part == 0) { if (r == k_lo) med[0] = ei[u]; if (r == k_hi) med[1] = ei[u]; }
    }
}

__global__ __launch_bounds__(THREADS) void median_rank_kernel(const double* __restrict__ err, int M, double* __restrict__ out)
{
    extern __shared__ double cerr[];
    __shared__ double med[2];
    const int tid = threadIdx.x;
    for (int i = tid; i < M; i += THREADS) cerr[i] = err[(size_t)blockIdx.x * M + i];
    __syncthreads();
    int S = 1;
    while (S < 64 && M * (S * 2) <= 2048) S *= 2;
    const int part = tid & (S - 1), G = THREADS / S, q0 = tid / S, passes = (M * S + THREADS - 1) / THREADS;
    switch (passes) {
    case 0: case 1: rank_tile<1>(cerr, M, S, part, q0, G, med); break;
    case 2: rank_tile<2>(cerr, M, S, part, q0, G, med); break;
    case 3: rank_tile<3>(cerr, M, S, part, q0, G, med); break;
    case 4: rank_tile<4>(cerr, M, S, part, q0, G, med); break;
    case 5: rank_tile<5>(cerr, M, S, part, q0, G, med); break;
    case 6: rank_tile<6>(cerr, M, S, part, q0, G, med); break;
    case 7: rank_tile<7>(cerr, M, S, part, q0, G, med); break;
    default: rank_tile<8>(cerr, M, S, part, q0, G, med); break;
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = 0.5 * (med[0] + med[1]);
}

// bitonic sort of P = next power of two >= M doubles (padding +inf), 256 threads, compare-exchanges of stage (k, j) on element pairs (i, i ^ j)
__global__ __launch_bounds__(THREADS) void median_bitonic_kernel(const double* __restrict__ err, int M, int P, double* __restrict__ out)
{
    extern __shared__ double a[];
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += THREADS) a[i] = i < M ? err[(size_t)blockIdx.x * M + i] : INFINITY;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));                 // index with bit j clear
                const int l = i | j;
                const bool up = (i & k) == 0;
                const double x = a[i], y = a[l];
                const bool sw = up ? (x > y) : (x < y);
                if (sw) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
    }
    if (tid == 0) out[blockIdx.x] = 0.5 * (a[(M - 1) / 2] + a[M / 2]);
}

// the same network, block barriers only around the steps whose pairs cross the 128-element slab a wave owns (j >= 128): comparator t works on
// elements inside slab t / 64 while j <= 64, and one wave's LDS instructions complete in issue order -- the compiler is told not to move them
__global__ __launch_bounds__(THREADS) void median_bitonic_wave_kernel(const double* __restrict__ err, int M, int P, double* __restrict__ out)
{
    extern __shared__ double a[];
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += THREADS) a[i] = i < M ? err[(size_t)blockIdx.x * M + i] : INFINITY;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = (i & k) == 0;
                const double x = a[i], y = a[l];
                const bool sw = up ? (x > y) : (x < y);
                if (sw) { a[i] = y; a[l] = x; }
            }
            if (j >= 128 || (j == 1 && k >= 128)) __syncthreads();
            else asm volatile("" ::: "memory");
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = 0.5 * (a[(M - 1) / 2] + a[M / 2]);
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 3000, M = argc > 2 ? atoi(argv[2]) : 220;
    int P = 1; while (P < M) P <<= 1;
    std::vector<double> h((size_t)B * M);
    srand(7);
    for (auto& v : h) v = (rand() % 100000) * 1e-4;                                  // ties included
    double *d, *o1, *o2, *o3;
    CK(hipMalloc(&d, h.size() * 8)); CK(hipMalloc(&o1, B * 8)); CK(hipMalloc(&o2, B * 8)); CK(hipMalloc(&o3, B * 8));
    CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms1 = 0, ms2 = 0, ms3 = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(median_rank_kernel, dim3(B), dim3(THREADS), (size_t)M * 8, 0, d, M, o1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
        CK(hipEventRecord(e0));
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(median_bitonic_kernel, dim3(B), dim3(THREADS), (size_t)P * 8, 0, d, M, P, o2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
        CK(hipEventRecord(e0));
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(median_bitonic_wave_kernel, dim3(B), dim3(THREADS), (size_t)P * 8, 0, d, M, P, o3);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms3, e0, e1));
    }
    std::vector<double> r1(B), r2(B), r3(B);
    CK(hipMemcpy(r1.data(), o1, B * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, B * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r3.data(), o3, B * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int b = 0; b < B; ++b) {
        std::vector<double> s(h.begin() + (size_t)b * M, h.begin() + (size_t)(b + 1) * M);
        std::sort(s.begin(), s.end());
        const double ref = 0.5 * (s[(M - 1) / 2] + s[M / 2]);
        if (r1[b] != ref || r2[b] != ref || r3[b] != ref) ++bad;
    }
    printf("B = %d blocks, M = %d errors (sorted array padded to %d): rank count %.1f us per launch, bitonic sort %.1f us, with wave-local steps %.1f us; medians wrong: %d\n",
           B, M, P, ms1 / 20 * 1e3, ms2 / 20 * 1e3, ms3 / 20 * 1e3, bad);
    return bad != 0;
}
