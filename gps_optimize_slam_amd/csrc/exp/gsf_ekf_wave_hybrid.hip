// exp/gsf_ekf_wave_hybrid.hip -- EXPERIMENT (`make hybrid` -> ../libgsf_hybrid.so): the fused pipeline at ONE wave per SIMD with a helper wave
// ONLY for the tracks that need it.  A launch of 1 000 tracks ends with its slowest wave, and the slowest waves are the tracks with a GNSS
// outage (RTS patch, sharp-turn test, a fit on the first segment: +2.4 us).  The two-wave build (a helper wave
// computes every chunk's variances while the main wave is in its fit: -28 % of a chunk's instructions) loses at 1 000 tracks because every
// SIMD then holds two waves.  Here every block has two waves, but the helper first looks at the track's validity bytes and LEAVES when all
// are set (nine tracks in ten); only an outage track keeps its helper.  The main wave learns what the helper did from a word in LDS after its
// prelude (bounded wait, falls back to computing the variances itself): no barrier that a diverging decision could deadlock.
// Same functions, same operands, same order: the bits of the one-wave kernel (51 parity tests green through this build).
//
// RESULT (round 4, same box, graph replay, 1 000 x 271; tools/experiments/hybrid_ab.sh, GSF_HYBRID_MODE): shipped one-wave kernel 18.4 us;
// this kernel with the helper leaving at once (mode 0) or after its scan (mode 2) 19.2-19.5 us -- the two-wave blocks and the run-time choice
// of the variance source cost every track a microsecond; with the helper working for the 134 outage tracks (mode 1) 24.0-24.6 us -- there is no
// idle SIMD at 1 000 tracks, the helper shares one with some other track's main wave, finishes late and its own main wave waits for it; with
// s_setprio(3) in the helper (mode 3) 20.1 us.  Two chunk loops inlined side by side instead of the run-time choice: 222 registers, 115 scalars
// spilled into VGPR lanes, 24.1 us; the rare loop behind a call: 270 registers = one wave per SIMD, 34.3 us.  Not shipped.
//
// The run-time switch it needs in gsf_wave_common.hpp (`int32_t use_pv` as last member of WaveArgs, `PREVAR && a.use_pv` where
// wave_serial_chunks chooses between the LDS variances and variance_chunk()) sits behind GSF_EXP_HYBRID, which only this file defines: the
// shipped translation units compile the profiled sources unchanged.
#define GSF_EXP_HYBRID 1
#include <cstdlib>
#include "../gsf_wave_common.hpp"
#define launch_ekf_wave launch_ekf_wave_shipped
#include "../gsf_ekf_wave.hip"
#undef launch_ekf_wave

namespace {

template <bool PIPELINE, int AXMODE>
__global__ __launch_bounds__(128) void ekf_wave_hybrid_kernel(WaveArgs a, EkfConfig cfg, int pv_stride, int mode)
{
    extern __shared__ double gsf_pv[];
    __shared__ int hy_flag;                                              // 0 undecided, 1 the helper left (all rows valid), 2 variances written
    const int lane = threadIdx.x & 63;
    const int64_t b = blockIdx.x;
    const bool helper = ((threadIdx.x >> 6) ^ ((blockIdx.x >> GSF_DUO_ROLE_SHIFT) & 1u)) != 0u;
    if (threadIdx.x == 0) hy_flag = 0;
    __syncthreads();                                                     // (the only barrier: both waves are here before either has decided anything)
    const int64_t N = a.N, base = b * N;                                 // equal-length tracks only (the launcher checks)
    if (N <= 0) { if (!helper && lane == 0 && a.status) a.status[b] = 0; return; }
    volatile int* flag = &hy_flag;
    if (helper) {
        if (mode == 0) { if (lane == 0) *flag = 1; return; }          // probe: no helper work at all
        const uint8_t* __restrict__ valb = a.valid + base;
        bool inv = false;
        for (int64_t i = lane; i < N; i += 64) inv = inv || (valb[i] == 0);
        if (__ballot(inv) == 0ull || mode == 2) { if (lane == 0) *flag = 1; return; }   // (mode 2, probe: the scan, but never any help)
        if (mode == 3) __builtin_amdgcn_s_setprio(3);                    // (probe: the helper wins the issue arbitration against the wave it shares a SIMD with)
        wave_variance_helper(a, cfg, b, lane, gsf_pv, pv_stride);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *flag = 2;
        return;
    }
    // ---- main wave: wave_serial_body with the choice of the chunk loop taken at run time
    const double* __restrict__ tsb = a.ts + base;
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ quatb = a.quat + base * 4;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;
    ChunkIn nxt = load_chunk(tsb, posb, quatb, gpsb, valb, lane, N);
    __builtin_amdgcn_sched_barrier(0);
    Vec3 p0; Quat q0; int32_t fit = 0;
    if (!wave_prelude<PIPELINE>(a, b, base, N, lane, p0, q0, fit)) return;
    int f = *flag;
    for (int spins = 0; f == 0 && spins < 4096; ++spins) { __builtin_amdgcn_s_sleep(2); f = *flag; }   // (the helper decides within 2 us and finishes within ~4: this wave arrives after 6)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    f = __builtin_amdgcn_readfirstlane(f);
    // ONE chunk loop, the source of the variances chosen per track at run time (two inlined loops: 222 registers, 115 scalars spilled into lanes,
    // 24.1 us against 18.3 at 1 000 tracks; the rare loop as a call: 270 registers = one wave per SIMD, 34 us)
    WaveArgs a2 = a; a2.use_pv = (f == 2) ? 1 : 0;
    wave_serial_chunks<PIPELINE, true, true, 1, AXMODE>(a2, cfg, b, lane, base, N, p0, q0, fit, nxt, gsf_pv, pv_stride, 0);
}

}  // namespace

namespace gsf {
int launch_ekf_wave(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                    const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                    int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status,
                    const int64_t* offsets)
{
    const bool duo = pipeline && !offsets && ctx->duo_kernel != 0 && N > 64 && N <= 640 && (ctx->duo_kernel == 1 || (ctx->duo_kernel == -1 && B <= 256));
    const bool hybrid = pipeline && !offsets && !duo && ctx->duo_kernel == -1 && ctx->block_kernel != 1 && N > 64 && N <= 640 && B > 256 && B <= 2048;
    if (!hybrid)
        return launch_ekf_wave_shipped(ctx, pipeline, ts, pos, quat, gps, valid, init_pos, init_quat, cfg, B, N, R, t, s, pos_out, quat_out, status, offsets);
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, offsets, ctx->fit_rows };
    const EkfConfig k = to_core(cfg);
    const bool xy = k.P0[1] == k.P0[0] && k.Qps[1] == k.Qps[0] && k.Rm[1] == k.Rm[0] &&
                    !(k.P0[2] == k.P0[0] && k.Qps[2] == k.Qps[0] && k.Rm[2] == k.Rm[0]);
    const int stride = (int)((N + 1) & ~(int64_t)1);
    const char* hm_ = getenv("GSF_HYBRID_MODE"); const int hy_mode = hm_ ? atoi(hm_) : 1;
    if (xy) hipLaunchKernelGGL((ekf_wave_hybrid_kernel<true, 1>), dim3((unsigned)B), dim3(128), (size_t)stride * 9 * sizeof(double), ctx->stream, a, k, stride, hy_mode);
    else hipLaunchKernelGGL((ekf_wave_hybrid_kernel<true, 0>), dim3((unsigned)B), dim3(128), (size_t)stride * 9 * sizeof(double), ctx->stream, a, k, stride, hy_mode);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
}  // namespace gsf
