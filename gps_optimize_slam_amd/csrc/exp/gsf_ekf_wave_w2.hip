// exp/gsf_ekf_wave_w2.hip -- EXPERIMENT, not part of libgsf.so (`make wave2` builds ../libgsf_wave2.so with this unit in place of
// gsf_ekf_wave.o): the small-batch wave-per-trajectory route with TWO poses per lane (exp/gsf_wave2.hpp).  Parity-green against the
// oracle on every track length tried (tests/campaigns/wave2_check.py) and SLOWER than the shipped kernel: DESIGN.md section 5,
// "two poses per lane".  The shipped translation unit is included unchanged, its launcher under another name.
#include "../gsf_wave_common.hpp"
#define launch_ekf_wave launch_ekf_wave_shipped
#include "../gsf_ekf_wave.hip"
#undef launch_ekf_wave
#include "gsf_wave2.hpp"

namespace {
template <bool PIPELINE>
__global__ __launch_bounds__(64, 1) void ekf_wave2_kernel(WaveArgs a, EkfConfig cfg)
{
    wave2_serial_body<PIPELINE>(a, cfg, (int64_t)blockIdx.x, (int)threadIdx.x);
}
}  // namespace

namespace gsf {
int launch_ekf_wave(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                    const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                    int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status,
                    const int64_t* offsets)
{
    const EkfConfig k = to_core(cfg);
    const bool xy = k.P0[1] == k.P0[0] && k.Qps[1] == k.Qps[0] && k.Rm[1] == k.Rm[0] &&
                    !(k.P0[2] == k.P0[0] && k.Qps[2] == k.Qps[0] && k.Rm[2] == k.Rm[0]);
    const bool duo = pipeline && !offsets && ctx->duo_kernel != 0 && N > 64 && N <= 640 && (ctx->duo_kernel == 1 || (ctx->duo_kernel == -1 && B <= 256));
    if (!(xy && !offsets && N >= 2 && B <= 2048 && !duo && ctx->block_kernel != 1))
        return launch_ekf_wave_shipped(ctx, pipeline, ts, pos, quat, gps, valid, init_pos, init_quat, cfg, B, N, R, t, s, pos_out, quat_out, status, offsets);
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, offsets, pipeline ? ctx->fit_rows : FitRows{ 0, 0, 0.0, 0.0 } };
    if (pipeline) hipLaunchKernelGGL((ekf_wave2_kernel<true>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k);
    else hipLaunchKernelGGL((ekf_wave2_kernel<false>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
}  // namespace gsf
