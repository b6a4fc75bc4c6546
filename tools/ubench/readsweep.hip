// What does the ACCESS SHAPE of the C4 window kernel cost against a plain read sweep of the same bytes?
// Two arrays of nw x W x 3 doubles (src, dst: 1 200 B per window at W = 50) are read once and summed, three ways:
//   pieces   every lane reads 16-byte pieces of the contiguous arrays, lane after lane (a plain read sweep, two streams)
//   rows     the kernel's shape: a 16-lane row of the wave per window, lane j reads rows j, j+16, j+32, j+48 of its window as three doubles
//            at 24-byte stride, four windows per wave trip (windows_fused_kernel, gsf_sim3.hip)
//   rows+fma the same with the kernel's sixteen moment accumulations per row (is it the arithmetic, not the loads?)
// build: hipcc -O3 --offload-arch=gfx950 readsweep.hip -o readsweep ; run: ./readsweep [windows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void pieces(const d2* __restrict__ a, const d2* __restrict__ b, size_t n, double* out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const d2 x = a[i], y = b[i];
        s += x.x + x.y + y.x + y.y;
    }
    if (s == 123.456) out[0] = s;
}
template <bool FMA>
__global__ __launch_bounds__(128) void rows(const double* __restrict__ src, const double* __restrict__ dst, long long B, int W, double* out)
{
    const int lane = threadIdx.x & 63, j0 = lane & 15, wv = threadIdx.x >> 6;
    const long long nsuper = (B + 63) / 64;
    double acc[16] = { 0 };
    for (long long sp = (long long)blockIdx.x * 2 + wv; sp < nsuper; sp += (long long)gridDim.x * 2) {
        for (int trip = 0; trip < 16; ++trip) {
            const long long w = sp * 64 + trip * 4 + (lane >> 4);
            const long long i0 = (w < B ? w : B - 1) * (long long)W;
            double pa[4][3], pc[4][3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long i = i0 + j0 + 16 * k, ic = (j0 + 16 * k < W) ? i : i0;
                pa[k][0] = src[ic * 3]; pa[k][1] = src[ic * 3 + 1]; pa[k][2] = src[ic * 3 + 2];
                pc[k][0] = dst[ic * 3]; pc[k][1] = dst[ic * 3 + 1]; pc[k][2] = dst[ic * 3 + 2];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (FMA) {
                    const double a0 = pa[k][0], a1 = pa[k][1], a2 = pa[k][2], c0 = pc[k][0], c1 = pc[k][1], c2 = pc[k][2];
                    acc[0] += a0; acc[1] += a1; acc[2] += a2; acc[3] += c0; acc[4] += c1; acc[5] += c2; acc[6] += a0 * a0 + a1 * a1 + a2 * a2;
                    acc[7] += a0 * c0; acc[8] += a0 * c1; acc[9] += a0 * c2; acc[10] += a1 * c0; acc[11] += a1 * c1; acc[12] += a1 * c2;
                    acc[13] += a2 * c0; acc[14] += a2 * c1; acc[15] += a2 * c2;
                } else {
                    acc[0] += pa[k][0] + pa[k][1] + pa[k][2] + pc[k][0] + pc[k][1] + pc[k][2];
                }
            }
        }
    }
    double s = 0;
    for (int k = 0; k < 16; ++k) s += acc[k];
    if (s == 123.456) out[0] = s;
}
int main(int argc, char** argv)
{
    const long long nw = argc > 1 ? atoll(argv[1]) : 1000000; const int W = 50;
    const size_t bytes = (size_t)nw * W * 24;
    double *a, *b, *o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int r = 0; r < 3; ++r) launch();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6);
    };
    for (int blocks : { 2048, 8192, 32768 })
        time(blocks == 2048 ? "pieces (16 B per lane), 2 048 x 256" : blocks == 8192 ? "pieces, 8 192 x 256" : "pieces, 32 768 x 256",
             [&] { hipLaunchKernelGGL(pieces, dim3(blocks), dim3(256), 0, 0, (const d2*)a, (const d2*)b, bytes / 16, o); });
    const int nblk = (int)(((nw + 63) / 64 + 1) / 2);
    time("rows (the kernel's shape), sums only", [&] { hipLaunchKernelGGL(rows<false>, dim3(nblk), dim3(128), 0, 0, a, b, nw, W, o); });
    time("rows + the sixteen moment accumulations", [&] { hipLaunchKernelGGL(rows<true>, dim3(nblk), dim3(128), 0, 0, a, b, nw, W, o); });
    return 0;
}
