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
template <class K> void run(const char* name, K kern, int ch, int per_iter, int blocks, int threads)
{
    double* d; long long* c; hipMalloc(&d, (size_t)blocks * threads * 8); hipMalloc(&c, 8);
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, c, iters);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s chains %d: %.2f cycles per instruction (%.2f per chain step)\n", name, ch, (double)h / ((double)iters * per_iter * ch), (double)h / ((double)iters * per_iter));
    hipFree(d); hipFree(c);
}
int main(int argc, char** argv)
{
    const int wpb = argc > 1 ? atoi(argv[1]) : 1;
    const int blocks = 1024 / wpb * 1, threads = 64 * wpb;        // ~one wave per SIMD chip-wide at wpb = 1 (blocks of one wave)
    printf("blocks %d x %d threads\n", blocks, threads);
    run("v_fma_f64 dependent", fma_chain<1>, 1, 16, blocks, threads);
    run("v_fma_f64", fma_chain<2>, 2, 16, blocks, threads);
    run("v_fma_f64", fma_chain<4>, 4, 16, blocks, threads);
    run("dpp mov x2 + fma (3 instr)", dpp_chain<1>, 1, 8 * 3, blocks, threads);
    run("dpp mov x2 + fma (3 instr)", dpp_chain<2>, 2, 8 * 3, blocks, threads);
    run("dpp mov x2 + fma (3 instr)", dpp_chain<4>, 4, 8 * 3, blocks, threads);
    return 0;
}
