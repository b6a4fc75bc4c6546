// Micro-benchmark (diagnostic, not part of libgsf.so): issue cost of the pieces of one Moebius scan stage with ONE wave per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/scan_stage tools/ubench/scan_stage.hip && /tmp/scan_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL, int RM>
__device__ __forceinline__ double dppd(double old, double v)
{
    double oo = old;
    asm volatile("" : "+v"(oo));
    const long long o = __double_as_longlong(oo), x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)x, CTRL, RM, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), CTRL, RM, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// MODE 0: full stage (4 dpp doubles + 8 fp64 ops); 1: only the 8 fp64 ops (operands from registers); 2: only the 4 dpp doubles;
// 3: 8 INDEPENDENT fma chains (pure FMA issue rate); 4: one DEPENDENT fma chain (latency); 5: v_rcp_f64 chain; 6: v_readlane + use
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int iters)
{
    double A = 1.0 + threadIdx.x * 1e-3, B = 0.5, C = 0.25, D = 1.0 - threadIdx.x * 1e-4;
    double e0 = 1.0001, e1 = 0.9999, e2 = 1.0002, e3 = 0.9998, e4 = 1.0003, e5 = 0.9997, e6 = 1.0004, e7 = 0.9996;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            const double oA = dppd<0x111, 0xf>(1.0, A), oB = dppd<0x111, 0xf>(0.0, B), oC = dppd<0x111, 0xf>(0.0, C), oD = dppd<0x111, 0xf>(1.0, D);
            const double nA = A * oA + B * oC, nB = A * oB + B * oD, nC = C * oA + D * oC, nD = C * oB + D * oD;
            A = nA; B = nB; C = nC; D = nD;
        } else if (MODE == 1) {
            const double nA = A * e0 + B * e1, nB = A * e2 + B * e3, nC = C * e0 + D * e1, nD = C * e2 + D * e3;
            A = nA; B = nB; C = nC; D = nD;
        } else if (MODE == 2) {
            A = dppd<0x111, 0xf>(1.0, A); B = dppd<0x111, 0xf>(0.0, B); C = dppd<0x111, 0xf>(0.0, C); D = dppd<0x111, 0xf>(1.0, D);
        } else if (MODE == 3) {
            e0 = fma(e0, A, B); e1 = fma(e1, A, B); e2 = fma(e2, A, B); e3 = fma(e3, A, B); e4 = fma(e4, A, B); e5 = fma(e5, A, B); e6 = fma(e6, A, B); e7 = fma(e7, A, B);
        } else if (MODE == 4) {
            A = fma(A, e0, B); A = fma(A, e1, C); A = fma(A, e2, B); A = fma(A, e3, C); A = fma(A, e4, B); A = fma(A, e5, C); A = fma(A, e6, B); A = fma(A, e7, C);
        } else if (MODE == 5) {
            A = __builtin_amdgcn_rcp(A) + 1.0; A = __builtin_amdgcn_rcp(A) + 1.0; A = __builtin_amdgcn_rcp(A) + 1.0; A = __builtin_amdgcn_rcp(A) + 1.0;
        } else if (MODE == 6) {
            const long long x = __double_as_longlong(A);
            const int lo = __builtin_amdgcn_readlane((int)x, 63), hi = __builtin_amdgcn_readlane((int)(x >> 32), 63);
            const double c = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
            A = A * 0.5 + c * 0.25; B = B + c;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = A + B + C + D + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int ops_per_iter, int blocks)
{
    double* out; long long* cyc; const int iters = 4000;
    hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&cyc, blocks * sizeof(long long));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-44s blocks %5d: %7.1f cycles / iteration  (%d instruction(s) of interest -> %5.2f cycles each)\n", name, blocks, s / blocks / iters, ops_per_iter, s / blocks / iters / ops_per_iter);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int blocks : { 1, 1024, 3072 }) {
        run<0>("full Moebius stage (8 dpp + 4 mov + 8 fp64)", 20, blocks);
        run<1>("8 fp64 mul/fma, 4 independent pairs", 8, blocks);
        run<2>("4 dpp doubles (8 v_mov_dpp + 4 v_mov_b64)", 12, blocks);
        run<3>("8 independent v_fma_f64", 8, blocks);
        run<4>("8 dependent v_fma_f64", 8, blocks);
        run<5>("4 dependent (v_rcp_f64 + v_add_f64)", 8, blocks);
        run<6>("2 v_readlane + 3 fp64 using the SGPR pair", 5, blocks);
    }
    return 0;
}
