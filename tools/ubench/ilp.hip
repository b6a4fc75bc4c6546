// How fast does ONE wave on a SIMD issue dependent FP64 work, and how much do independent chains in the same wave help?
// Each test runs K instructions in CH independent chains (CH = 1: fully dependent) and reports shader cycles per instruction.
// build: hipcc -O2 --offload-arch=gfx950 ilp.hip -o ilp ; run: ./ilp [waves_per_block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CH>
__global__ void fma_chain(double* out, long long* cyc, int iters)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x + c;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = fma(x[c], 1.0000001, 0.5);
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// a scan-like chain: DPP row_shr move of a double (two 32-bit moves) followed by an FMA on the moved value
template <int CH>
__global__ void dpp_chain(double* out, long long* cyc, int iters)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x + c;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const long long b = __double_as_longlong(x[c]);
                const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x111, 0xf, 0xf, true);
                const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x111, 0xf, 0xf, true);
                const double o = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
                x[c] = fma(x[c], 0.5, o);
            }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// lane broadcast of a double through SGPRs (v_readlane x2, then used as a scalar operand) -- the carry pattern of the chunk loop
__global__ void readlane_chain(double* out, long long* cyc, int iters)
{
    double x = threadIdx.x + 1.0, y = 0.5;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long b = __double_as_longlong(x);
            const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
            const double s = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
            x = fma(y, s, x);                       // 3 instructions per step: readlane, readlane, fma with a scalar operand
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// compare -> lane mask in SGPRs -> select: the mask pattern of the outage logic
__global__ void ballot_chain(double* out, long long* cyc, int iters)
{
    double x = threadIdx.x + 1.0;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(x > 3.0);
            x = ((m >> (threadIdx.x & 63)) & 1ull) ? x * 0.999 : x + 1.0;      // v_cmp, s_lshr/s_and or v ops, v_cndmask ...
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// strided row loads like the chunk loop's (24-byte rows, one row per lane), dependent through the address
__global__ void load_chain(const double* in, double* out, long long* cyc, int iters)
{
    const double* p = in + (size_t)blockIdx.x * 64 * 3 * 64;
    double acc = 0;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        const double* q = p + (size_t)(i & 63) * 64 * 3 + threadIdx.x * 3;
        acc += q[0] + q[1] + q[2];
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// the cross-row DPP controls of the scans: row_bcast:15 (0x142), row_bcast:31 (0x143), wave_shr:1 (0x138)
template <int CTRL, int RM>
__global__ void dppx_chain(double* out, long long* cyc, int iters)
{
    double x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = threadIdx.x + c;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long long b = __double_as_longlong(x[c]);
                const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, RM, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, RM, 0xf, false);
                x[c] = fma(x[c], 0.5, __longlong_as_double(((long long)hi << 32) | (unsigned)lo));
            }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// gfx950 only: v_permlane16_swap_b32 / v_permlane32_swap_b32 (whole 16- / 32-lane rows exchanged lane for lane) as the cross-row move of a
// scan stage: two swaps per double + an FMA on the moved value, four independent chains -- to be compared with row_bcast:15/31 above
template <int WIDE>
__global__ void permswap_chain(double* out, long long* cyc, int iters)
{
    double x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = threadIdx.x + c;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long long b = __double_as_longlong(x[c]);
                const unsigned blo = (unsigned)b, bhi = (unsigned)(b >> 32);
                unsigned lo, hi;
                if (WIDE == 16) { lo = __builtin_amdgcn_permlane16_swap(blo, blo, false, false)[0]; hi = __builtin_amdgcn_permlane16_swap(bhi, bhi, false, false)[0]; }
                else { lo = __builtin_amdgcn_permlane32_swap(blo, blo, false, false)[0]; hi = __builtin_amdgcn_permlane32_swap(bhi, bhi, false, false)[0]; }
                x[c] = fma(x[c], 0.5, __longlong_as_double(((long long)hi << 32) | lo));
            }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// ds_bpermute of a double (two 32-bit LDS-crossbar moves) + FMA, four chains: the __shfl route
__global__ void bpermute_chain(double* out, long long* cyc, int iters)
{
    double x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = threadIdx.x + c;
    const int src = ((threadIdx.x & 63) ^ 16) << 2;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long long b = __double_as_longlong(x[c]);
                const int lo = __builtin_amdgcn_ds_bpermute(src, (int)b), hi = __builtin_amdgcn_ds_bpermute(src, (int)(b >> 32));
                x[c] = fma(x[c], 0.5, __longlong_as_double(((long long)hi << 32) | (unsigned)lo));
            }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// scalar ALU: dependent 64-bit mask arithmetic on a wave-uniform value (s_lshr_b64 / s_and_b64 / s_xor_b64 / s_add_u32 ...)
__global__ void salu_chain(double* out, long long* cyc, int iters, unsigned long long seed)
{
    unsigned long long m = __builtin_amdgcn_readfirstlane((int)seed) | 0x123456789abcdefull;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            m = (m >> 1) ^ (m & 0x5555555555555555ull);
            m = m + (unsigned long long)__builtin_popcountll(m);
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)m;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// wave-uniform branches: a chain of data-dependent (but uniform) conditional blocks
__global__ void branch_chain(double* out, long long* cyc, int iters, unsigned long long seed)
{
    unsigned long long m = __builtin_amdgcn_readfirstlane((int)seed) | 0x9e3779b97f4a7c15ull;
    double x = threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if ((m >> u) & 1ull) x = fma(x, 1.0000001, 0.5); else x = fma(x, 0.9999999, 0.25);
            m = m * 6364136223846793005ull + 1442695040888963407ull;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + (double)m;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// transcendental unit: v_rcp_f64 / v_rsq_f64 seeds, CH independent chains
template <int CH>
__global__ void trans_chain(double* out, long long* cyc, int iters)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x + 1.5 + c;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = __builtin_amdgcn_rcp(x[c]) + 1.25;     // v_rcp_f64 + v_add_f64
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// compare to a lane mask + select, independent of each other (4 per step)
__global__ void cmp_sel(double* out, long long* cyc, int iters)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x0 = (x0 > 3.0) ? x0 * 0.99 : x0 + 1.0; x1 = (x1 > 3.0) ? x1 * 0.99 : x1 + 1.0;
            x2 = (x2 > 3.0) ? x2 * 0.99 : x2 + 1.0; x3 = (x3 > 3.0) ? x3 * 0.99 : x3 + 1.0;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <class K> void run(const char* name, K kern, int ch, int per_iter, int blocks, int threads)
{
    double* d; long long* c; hipMalloc(&d, (size_t)blocks * threads * 8); hipMalloc(&c, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, c, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, c, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    // (the launch is ~0.1-1 ms of pure loop: the event time over the loop's clock64() count gives the rate of that counter)
    printf("%-28s chains %d: %.2f cycles per instruction (%.2f per chain step)   [clock64 ticks at >= %.0f MHz: %lld ticks in %.1f us]\n", name, ch,
           (double)h / ((double)iters * per_iter * ch), (double)h / ((double)iters * per_iter), (double)h / (ms * 1e3), h, ms * 1e3);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(d); hipFree(c);
}
int main(int argc, char** argv)
{
    const int wpb = argc > 1 ? atoi(argv[1]) : 1;
    const int blocks = argc > 2 ? atoi(argv[2]) : 1024 / wpb, threads = 64 * wpb;   // default: ~one wave per SIMD chip-wide; 256 blocks of one wave: one wave per CU
    printf("blocks %d x %d threads\n", blocks, threads);
    run("v_fma_f64 dependent", fma_chain<1>, 1, 16, blocks, threads);
    run("v_fma_f64", fma_chain<2>, 2, 16, blocks, threads);
    run("v_fma_f64", fma_chain<4>, 4, 16, blocks, threads);
    run("dpp mov x2 + fma (3 instr)", dpp_chain<1>, 1, 8 * 3, blocks, threads);
    run("dpp mov x2 + fma (3 instr)", dpp_chain<2>, 2, 8 * 3, blocks, threads);
    run("dpp mov x2 + fma (3 instr)", dpp_chain<4>, 4, 8 * 3, blocks, threads);
    run("row_bcast:15 x2 + fma, 4 chains", dppx_chain<0x142, 0xa>, 4, 8 * 3, blocks, threads);
    run("row_bcast:31 x2 + fma, 4 chains", dppx_chain<0x143, 0xc>, 4, 8 * 3, blocks, threads);
    run("wave_shr:1 x2 + fma, 4 chains", dppx_chain<0x138, 0xf>, 4, 8 * 3, blocks, threads);
    run("permlane16_swap x2 + fma, 4 chains", permswap_chain<16>, 4, 8 * 3, blocks, threads);
    run("permlane32_swap x2 + fma, 4 chains", permswap_chain<32>, 4, 8 * 3, blocks, threads);
    run("ds_bpermute x2 + fma, 4 chains", bpermute_chain, 4, 8 * 3, blocks, threads);
    run("v_rcp_f64 + add (2 instr)", trans_chain<1>, 1, 8 * 2, blocks, threads);
    run("v_rcp_f64 + add (2 instr)", trans_chain<4>, 4, 8 * 2, blocks, threads);
    run("cmp + mul + add + 2 cndmask (x4)", cmp_sel, 1, 8 * 4, blocks, threads);
    {
        double* d; long long* c; hipMalloc(&d, (size_t)blocks * threads * 8); hipMalloc(&c, 8);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(salu_chain, dim3(blocks), dim3(threads), 0, 0, d, c, 2000, 12345ull);
        hipDeviceSynchronize(); long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.2f cycles per step (shift, and, xor, popcount, add on 64-bit scalars)\n", "SALU dependent chain", (double)h / (2000.0 * 16));
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(branch_chain, dim3(blocks), dim3(threads), 0, 0, d, c, 2000, 12345ull);
        hipDeviceSynchronize(); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.2f cycles per uniform branch + fma + 64-bit scalar multiply-add\n", "uniform branches", (double)h / (2000.0 * 16));
    }
    run("readlane x2 + fma (3 instr)", readlane_chain, 1, 8 * 3, blocks, threads);
    run("ballot + select (per step)", ballot_chain, 1, 8, blocks, threads);
    {
        double* in; double* d; long long* c; hipMalloc(&in, (size_t)blocks * 64 * 3 * 64 * 8); hipMalloc(&d, (size_t)blocks * threads * 8); hipMalloc(&c, 8);
        hipMemset(in, 0, (size_t)blocks * 64 * 3 * 64 * 8);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(load_chain, dim3(blocks), dim3(threads), 0, 0, in, d, c, 2000);
        hipDeviceSynchronize();
        long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.1f cycles per dependent row load (L2/L1 resident)\n", "24-byte row loads", (double)h / 2000.0);
    }
    return 0;
}
