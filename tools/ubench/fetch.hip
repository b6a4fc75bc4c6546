// Does a lone wave's issue rate depend on how many other waves of the CU (or of the CU pair sharing an instruction cache) run the
// same LARGE straight-line loop?  4 independent FMA chains, body unrolled to ~N instructions (8 bytes each), optional skipped blocks.
// build: hipcc -O2 --offload-arch=gfx950 fetch.hip -o fetch ; run: ./fetch <blocks>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int UNROLL, bool BRANCHY>
__global__ void big_loop(double* out, long long* cyc, int iters, int never)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    if (never <= -2) {                          // de-phase the waves: a block-dependent spin before the loop, so that the waves of a CU sit at different PCs
        const int spin = (int)((blockIdx.x * 37u) % 61u) * 40;
        for (int k = 0; k < spin; ++k) x3 = fma(x3, 1.0000001, 1e-9);
    }
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            x0 = fma(x0, 1.0000001, 0.5 + u); x1 = fma(x1, 1.0000002, 0.25 + u); x2 = fma(x2, 1.0000003, 0.125 + u); x3 = fma(x3, 1.0000004, 0.75 + u);
            if (BRANCHY && (u & 3) == 3) {
                if (never == u) {               // wave-uniform, never true: a skipped cold block (taken branch over it) every 16 instructions
                    x0 = sqrt(x0) + x1; x1 = sqrt(x1) + x2; x2 = sqrt(x2) + x3; x3 = sqrt(x3) + x0;
                }
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <class K> void run(const char* name, K kern, int instr_per_iter, int blocks, int never = -1)
{
    double* d; long long* c; hipMalloc(&d, (size_t)blocks * 64 * 8); hipMalloc(&c, 8);
    const int iters = 200;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, c, iters, never);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("  %-46s %.2f cycles per FMA\n", name, (double)h / ((double)iters * instr_per_iter));
    hipFree(d); hipFree(c);
}
int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    printf("%d one-wave blocks\n", blocks);
    run("64 FMAs per iteration (0.5 KB loop)", big_loop<16, false>, 64, blocks);
    run("2048 FMAs per iteration (16 KB loop)", big_loop<512, false>, 2048, blocks);
    run("2048 FMAs + a skipped block every 16 (branchy)", big_loop<512, true>, 2048, blocks);
    run("16 KB loop, waves de-phased", big_loop<512, false>, 2048, blocks, -2);
    run("branchy 16 KB loop, waves de-phased", big_loop<512, true>, 2048, blocks, -2);
    return 0;
}
