// gsf_ekf_wave_big.hip -- the wave-per-trajectory kernels (gsf_ekf_wave.hip, gsf_wave_common.hpp) built for BIG batches: more than two
// waves per SIMD (B > 2 048 trajectories; C3, the C5 shard).  Same template, same arithmetic, same bits as the small-batch builds; what
// differs is how a chunk's rows travel: with several waves per SIMD the launch sits on the memory system under its mixed read + write
// stream (the C3 timing probes), so the strided 8-byte loads / stores of a lane's own row are replaced by whole
// 16-byte pieces of the chunk's contiguous slabs, lane after lane, transposed through a few KB of LDS (GSF_WIDE(SMALLBATCH) and
// GSF_WIDE_STORES in gsf_wave_common.hpp).  A separate translation unit because the two regimes want different instruction schedulers
// (Makefile): iterative-ilp for the lone wave of the small batches, max-ilp here.
#define GSF_ROWS_ROUND 4                                                  // rounds of the row-choice pass: six chunks in flight spill 17 registers at 168 per lane
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

#ifndef GSF_BIG_OCC
#define GSF_BIG_OCC 3                                                     // three waves per SIMD: 168 registers (max-ilp left alone takes 216 and halves the occupancy)
#endif
template <bool PIPELINE, int AXMODE>
__global__ __launch_bounds__(64, GSF_BIG_OCC) void ekf_wave_big_kernel(WaveArgs a, EkfConfig cfg)
{
    wave_serial_body<PIPELINE, false, false, 1, AXMODE>(a, cfg, (int64_t)blockIdx.x, (int)threadIdx.x);
}

}  // namespace

namespace gsf {

int launch_ekf_wave_big(gsf_ctx* ctx, bool pipeline, bool xy, const double* ts, const double* pos, const double* quat, const double* gps,
                        const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B, int64_t N,
                        double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status, const int64_t* offsets)
{
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, offsets, pipeline ? ctx->fit_rows : FitRows{ 0, 0, 0.0, 0.0 } };
    const EkfConfig k = to_core(cfg);
#define GSF_LAUNCH_BIG(P_, X_) hipLaunchKernelGGL((ekf_wave_big_kernel<P_, X_>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k)
    if (pipeline) { if (xy) GSF_LAUNCH_BIG(true, 1); else GSF_LAUNCH_BIG(true, 0); }
    else { if (xy) GSF_LAUNCH_BIG(false, 1); else GSF_LAUNCH_BIG(false, 0); }
#undef GSF_LAUNCH_BIG
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace gsf

namespace gsf { const char* wave_big_build_info() { return GSF_TU_BUILD_INFO("gsf_ekf_wave_big.hip"); } }
