// How many workgroups of T threads does a CU admit at a given VGPR allocation and LDS size?  Every block records its CU and its start time;
// blocks that start late were not resident with the first batch.
// build: hipcc -O2 --offload-arch=gfx950 residency.hip -o residency ; run: ./residency <blocks> <threads>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
template <int VREG, int LDSB>
__global__ void probe(unsigned* out, int spin)
{
    __shared__ char lds[LDSB > 0 ? LDSB : 4];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (VREG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (VREG == 88) asm volatile("v_mov_b32 v87, 0" ::: "v87");
    if (VREG == 84) asm volatile("v_mov_b32 v83, 0" ::: "v83");
    if (VREG == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if (VREG == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
    if (VREG == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (VREG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VREG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    const long long t0 = wall_clock64();
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 0.5);
    if (LDSB > 0) lds[threadIdx.x] = (char)x;
    __syncthreads();
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        out[w * 4 + 0] = hw; out[w * 4 + 1] = xcc; out[w * 4 + 2] = (unsigned)t0; out[w * 4 + 3] = (unsigned)(t1 - t0) + (x == 1.25 ? 1 : 0) + (LDSB > 0 ? lds[0] & 0 : 0);
    }
}
template <int VREG, int LDSB>
void run(int blocks, int threads)
{
    const int wpb = threads / 64, nw = blocks * wpb;
    unsigned* d; hipMalloc(&d, nw * 16);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((probe<VREG, LDSB>), dim3(blocks), dim3(threads), 0, 0, d, 3000); }
    hipDeviceSynchronize();
    std::vector<unsigned> h(nw * 4); hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost);
    unsigned tmin = ~0u;
    for (int w = 0; w < nw; ++w) tmin = h[w * 4 + 2] < tmin ? h[w * 4 + 2] : tmin;
    int late = 0; std::map<unsigned, int> per_cu; std::map<unsigned, int> per_simd;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[b * wpb * 4], xcc = h[b * wpb * 4 + 1] & 0xf;
        const unsigned cuid = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
        if (h[b * wpb * 4 + 2] - tmin > 300) { late++; continue; }   // 3 us
        per_cu[cuid]++;
        for (int k = 0; k < wpb; ++k) per_simd[(cuid << 2) | ((h[(b * wpb + k) * 4] >> 4) & 3)]++;
    }
    std::map<int, int> hist, hs;
    for (auto& kv : per_cu) hist[kv.second]++;
    for (auto& kv : per_simd) hs[kv.second]++;
    printf("VGPR %3d LDS %6d B, %d blocks x %d threads: late %d; early blocks per CU:", VREG, LDSB, blocks, threads, late);
    for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
    printf("; waves per SIMD:"); for (auto& kv : hs) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
    hipFree(d);
}
int main(int argc, char** argv)
{
    const int blocks = atoi(argv[1]), threads = atoi(argv[2]);
    run<64, 0>(blocks, threads); run<72, 0>(blocks, threads); run<80, 0>(blocks, threads); run<84, 0>(blocks, threads); run<88, 0>(blocks, threads);
    run<96, 0>(blocks, threads); run<104, 0>(blocks, threads); run<128, 0>(blocks, threads); run<80, 23552>(blocks, threads);
    return 0;
}
