// Where do the waves of a launch land?  Every wave records (xcc, se, cu, simd, slot) from the HW_ID registers plus start/end clocks.
// build: hipcc -O2 --offload-arch=gfx950 placement.hip -o placement ; run: ./placement <blocks> <threads>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 0.5);
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        out[w * 4 + 0] = hw; out[w * 4 + 1] = xcc; out[w * 4 + 2] = (unsigned)t0; out[w * 4 + 3] = (unsigned)(t1 - t0) + (x == 1.25 ? 1 : 0);
    }
}
int main(int argc, char** argv)
{
    const int blocks = atoi(argv[1]), threads = atoi(argv[2]), wpb = threads / 64, nw = blocks * wpb;
    unsigned* d; hipMalloc(&d, nw * 16);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, 2000); }
    hipDeviceSynchronize();
    std::vector<unsigned> h(nw * 4); hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu, per_simd; std::map<int, int> simd_of_wave[8];
    int same_cu_blocks = 0;
    for (int b = 0; b < blocks; ++b) {
        for (int k = 0; k < wpb; ++k) {
            const unsigned hw = h[(b * wpb + k) * 4], xcc = h[(b * wpb + k) * 4 + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_cu[cuid]++; per_simd[(cuid << 2) | simd]++;
            if (k < 8) simd_of_wave[k][simd]++;
        }
    }
    printf("blocks %d x %d threads: %zu distinct CUs, %zu distinct SIMDs used\n", blocks, threads, per_cu.size(), per_simd.size());
    std::map<int, int> hist_cu, hist_simd;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    for (auto& kv : per_simd) hist_simd[kv.second]++;
    printf("waves per CU histogram:"); for (auto& kv : hist_cu) printf(" %d:%d", kv.first, kv.second); printf("\n");
    printf("waves per SIMD histogram:"); for (auto& kv : hist_simd) printf(" %d:%d", kv.first, kv.second); printf("\n");
    for (int k = 0; k < wpb && k < 8; ++k) { printf("wave %d of a block -> SIMD", k); for (auto& kv : simd_of_wave[k]) printf(" %d:%d", kv.first, kv.second); printf("\n"); }
    printf("first 12 blocks (xcc se sh cu | simd of each wave):\n");
    for (int b = 0; b < 12 && b < blocks; ++b) {
        const unsigned hw = h[b * wpb * 4], xcc = h[b * wpb * 4 + 1] & 0xf;
        printf("  b%-3d xcc%u se%u sh%u cu%-2u |", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf);
        for (int k = 0; k < wpb; ++k) printf(" %u", (h[(b * wpb + k) * 4] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
