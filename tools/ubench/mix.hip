// Does the issue rate of a wave that alternates vector and scalar instructions depend on other waves in its CU?
// Per step: 4 independent v_fma_f64 (4 chains) interleaved with NS independent scalar ALU operations.
// build: hipcc -O2 --offload-arch=gfx950 mix.hip -o mix ; run: ./mix <blocks>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int NS>
__global__ void mix(double* out, long long* cyc, int iters, unsigned seed)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned s0 = __builtin_amdgcn_readfirstlane(seed), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x0));
            if (NS > 0) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc");
            asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x1));
            if (NS > 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) :: "scc");
            asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x2));
            if (NS > 2) asm volatile("s_add_u32 %0, %0, 5" : "+s"(s2) :: "scc");
            asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x3));
            if (NS > 3) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s3) :: "scc");
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (double)(s0 + s1 + s2 + s3);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <class K> void run(const char* name, K kern, int nv, int ns, int blocks)
{
    double* d; long long* c; hipMalloc(&d, (size_t)blocks * 64 * 8); hipMalloc(&c, 8);
    const int iters = 1000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, c, iters, 7u);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("  %-34s %.2f cycles per group of %d VALU + %d SALU  (%.2f per instruction)\n", name, (double)h / (iters * 16.0), nv, ns, (double)h / (iters * 16.0 * (nv + ns)));
    hipFree(d); hipFree(c);
}
int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    printf("%d one-wave blocks\n", blocks);
    run("4 fma", mix<0>, 4, 0, blocks);
    run("4 fma + 2 scalar adds", mix<2>, 4, 2, blocks);
    run("4 fma + 4 scalar adds", mix<4>, 4, 4, blocks);
    return 0;
}
