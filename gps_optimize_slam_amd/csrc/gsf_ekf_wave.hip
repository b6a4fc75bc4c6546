// gsf_ekf_wave.hip -- K4 (and the fused K2+K3+K4 pipeline) with ONE WAVEFRONT PER TRAJECTORY.
//
// apply_ekf_correction (EKFGPSSLAM.py:831-935) is a serial recursion over the poses of one track, but every
// piece of it is an associative scan once the covariance is known to stay diagonal:
//   * orientation      q_i = q_{i-1} * dq_i                 -> prefix PRODUCT of quaternions
//   * variances        P_i = r(P+qdt)/((P+qdt)+r) or P+qdt   -> prefix composition of 2x2 Moebius maps
//   * positions        p_i = (1-k_i)(p_{i-1}+u_i) + k_i z_i  -> prefix composition of affine maps
//   * outage structure (start / recovery / sharp-turn gate)  -> 64-bit ballots + bit scans
//   * per-outage RTS   x_s[k] = x_f[k] + (P_f[k]/P_p[r]) (x_f[r]-x_p[r])   (the product of the gains A_j telescopes)
// so a wave takes 64 consecutive poses per iteration (lane = pose), runs log2(64) = 6 shuffle stages per scan and
// carries ~30 scalars to the next 64 poses.  All loads/stores of a chunk are contiguous (the natural
// trajectory-major layout of stacked TUM files), B trajectories give B independent waves, and a 271-pose track costs
// 5 iterations instead of 270 dependent steps: this is the low-latency / small-batch path (configs C1, C2); the
// lane-per-trajectory kernel of gsf_ekf.hip is the streaming path for huge batches.
//
// Arithmetic differs from the serial form only in rounding order (quaternion renormalisation once per chunk instead of
// every step, Moebius instead of Joseph variance update, local coordinates per chunk): observed |dp| ~1e-9 m against
// the 1e-6 m gate; the tests compare against the dense-7x7 CPU oracle.
#include "gsf_wave_chunk.hpp"

using namespace gsf;

namespace {

template <bool PIPELINE, bool SMALLBATCH>
__global__ __launch_bounds__(64) void ekf_wave_kernel(WaveArgs a, EkfConfig cfg)
{
    wave_serial_body<PIPELINE, false, SMALLBATCH>(a, cfg, (int64_t)blockIdx.x, (int)threadIdx.x);
}


// Two waves per trajectory for SMALL batches of the fused pipeline (one wave per SIMD, every wave in the same phase at the same
// time): while wave 0 is busy with the fit -- a memory burst followed by a latency-bound 3x3 Jacobi chain that leave the SIMD
// mostly idle -- wave 1 computes the variances of EVERY chunk (they depend on stamps and availability only, not on the fit) into
// LDS; after one block barrier wave 0 runs the chunk loop without its two Moebius scans (-28 % instructions per chunk).
// Same functions, same operands, same order as the one-wave kernel: bit-identical results.
#ifndef GSF_DUO_ROLE_SHIFT
#define GSF_DUO_ROLE_SHIFT 2        // measured best of 0..3 at 1 000 tracks (22.6 vs 23.1-23.2 us; 23.6 us without the helper)
#endif
template <bool PIPELINE>
__global__ __launch_bounds__(128) void ekf_wave_duo_kernel(WaveArgs a, EkfConfig cfg, int pv_stride)
{
    extern __shared__ double gsf_pv[];
    const int lane = threadIdx.x & 63;
    const int64_t b = blockIdx.x;
    // which wave of the block helps alternates with the block index: when two blocks share a pair of SIMDs, each SIMD then holds
    // one main and one helper wave (complementary phases) instead of two of a kind
    const bool helper = ((threadIdx.x >> 6) ^ ((blockIdx.x >> GSF_DUO_ROLE_SHIFT) & 1u)) != 0u;
    if (a.N <= 0) { if (!helper && lane == 0 && a.status) a.status[b] = 0; return; }   // empty tracks: both waves leave before any barrier
    if (helper) {
        wave_variance_helper(a, cfg, b, lane, gsf_pv, pv_stride);
        __syncthreads();
        return;
    }
    wave_serial_body<PIPELINE, true, true>(a, cfg, b, lane, gsf_pv, pv_stride);
}

// Several poses per lane: an iteration takes 64 * P consecutive poses, P = min(PPLMAX, ceil(remaining / 64)), lane l owning the P
// consecutive poses l*P .. l*P+P-1 (process_chunk<P>: in-lane composition, ONE set of DPP scans per iteration, in-lane
// application).  A track of up to 64 * PPLMAX poses is a single iteration with no chunk-to-chunk carry at all.  The wave executes
// fewer scan stages per pose the larger P is, at the price of registers and per-sub-pose bookkeeping (no next-chunk prefetch
// here).  Opt-in only: measured slower than one pose per lane on MI355X (see launch_ekf_wave).
template <int P>
__device__ __forceinline__ void load_and_process(const TrajPtrs& T, const EkfConfig& cfg, WaveCarry& C, const int64_t c0, const int lane)
{
    ChunkIn in[P];
#pragma unroll
    for (int j = 0; j < P; ++j) in[j] = load_chunk(T.ts, T.pos, T.quat, T.gps, T.valid, c0 + (int64_t)lane * P + j, T.N);
    process_chunk<P>(T, cfg, C, c0, in, lane);
}

template <bool PIPELINE, int PPLMAX>
__global__ __launch_bounds__(64) void ekf_wavep_kernel(WaveArgs a, EkfConfig cfg)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    int64_t base, N; traj_span(a, b, base, N);
    if (N <= 0) { if (lane == 0 && a.status) a.status[b] = 0; return; }              // empty track (ref :835)
    TrajPtrs T{ a.ts + base, a.pos + base * 3, a.quat + base * 4, a.gps + base * 3, a.valid + base, a.pos_out + base * 3, a.quat_out + base * 4, N };
    Vec3 p0; Quat q0; int32_t fit = 0;
    if (!wave_prelude<PIPELINE>(a, b, base, N, lane, p0, q0, fit)) return;
    WaveCarry C;
    C.q = ekf_normalize(q0); C.p = p0;                                   // ref :842, :683
    C.P[0] = cfg.P0[0]; C.P[1] = cfg.P0[1]; C.P[2] = cfg.P0[2];
    C.prev_avail = T.valid[0] != 0;                                      // ref :848 (raw mask)
    C.ostart = 0; C.seg_sharp = false;                                   // ref :861-862
    C.Pos[0] = C.P[0]; C.Pos[1] = C.P[1]; C.Pos[2] = C.P[2];
    C.po = Vec3{ T.pos[0], T.pos[1], T.pos[2] };
    C.ok = quat_unit(Quat{ T.quat[0], T.quat[1], T.quat[2], T.quat[3] }, C.r);
    C.t = T.ts[0];
    C.status = C.prev_avail ? 0 : ST_HAD_OUTAGE;
    C.same_axis[0] = C.same_axis[1] = C.same_axis[2] = -1;               // axes with identical (P0, Q, R) share the variance scan
    if (cfg.P0[1] == cfg.P0[0] && cfg.Qps[1] == cfg.Qps[0] && cfg.Rm[1] == cfg.Rm[0]) C.same_axis[1] = 0;
    if (cfg.P0[2] == cfg.P0[0] && cfg.Qps[2] == cfg.Qps[0] && cfg.Rm[2] == cfg.Rm[0]) C.same_axis[2] = 0;
    else if (cfg.P0[2] == cfg.P0[1] && cfg.Qps[2] == cfg.Qps[1] && cfg.Rm[2] == cfg.Rm[1]) C.same_axis[2] = 1;
    for (int64_t c0 = 0; c0 < N;) {
        const int64_t rem = N - c0;
        const int p = rem >= 64 * PPLMAX ? PPLMAX : (int)((rem + 63) / 64);          // wave-uniform
        if (PPLMAX >= 5 && p == 5) load_and_process<(PPLMAX >= 5 ? 5 : 1)>(T, cfg, C, c0, lane);
        else if (PPLMAX >= 4 && p == 4) load_and_process<(PPLMAX >= 4 ? 4 : 1)>(T, cfg, C, c0, lane);
        else if (PPLMAX >= 3 && p == 3) load_and_process<(PPLMAX >= 3 ? 3 : 1)>(T, cfg, C, c0, lane);
        else if (PPLMAX >= 2 && p == 2) load_and_process<(PPLMAX >= 2 ? 2 : 1)>(T, cfg, C, c0, lane);
        else load_and_process<1>(T, cfg, C, c0, lane);
        c0 += 64 * p;
    }
    if (lane == 0 && a.status) a.status[b] = (C.status | (C.prev_avail ? 0 : ST_ENDED_IN_OUTAGE)) | (PIPELINE ? (fit << 8) : 0);
}

EkfConfig to_core(const gsf_ekf_config* c)
{
    EkfConfig k;
    for (int i = 0; i < 7; ++i) { k.P0[i] = c->initial_cov_diag[i]; k.Qps[i] = c->process_noise_diag[i]; }
    for (int i = 0; i < 3; ++i) k.Rm[i] = c->meas_noise_diag[i];
    k.yaw_thr_rad = c->sharp_turn_yaw_rate_threshold_deg_per_sec * (M_PI / 180.0);
    k.sharp_turn_steps = c->default_ekf_transition_steps_on_sharp_turn;
    k._pad = 0;
    return k;
}

}  // namespace

namespace gsf {

// trajectory-major launches (called from gsf_ekf.hip's C entry points)
int launch_ekf_wave(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                    const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                    int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status,
                    const int64_t* offsets)
{
    GSF_REQUIRE(B <= 0x7fffffff, "B too large for one launch");
    // short equal-length tracks CAN take the single-shot kernel (gsf_ekf_seg.hip: the whole trajectory in one pass, scans paid once
    // per track) -- opt-in (gsf_set_option "seg_kernel" 1): it executes ~45 % fewer instructions per track but measured slower than
    // the chunked kernel from ~500 tracks up (C2 K4 23.8 vs 17.6 us, 400k x 271 7.5 vs 3.5 ms: 240 registers and 28 KB of LDS per
    // wave leave 5 waves per CU, and its 60 per-pose lane masks spill SGPRs); faster only below ~300 tracks (12 vs 15 us).
    if (!offsets && ctx->wave_ppl == 0 && ctx->ekf_variant == 0 && ctx->seg_kernel == 1 && N <= 64 * SEG_MAX_P)
        return launch_ekf_seg(ctx, pipeline, ts, pos, quat, gps, valid, init_pos, init_quat, cfg, B, N, R, t, s, pos_out, quat_out, status);
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, offsets };
    const EkfConfig k = to_core(cfg);
    // small batches of the fused pipeline: two waves per trajectory (see ekf_wave_duo_kernel).  Bit-identical to the one-wave
    // kernel, so choosing by batch size does not break shard invariance.  gsf_set_option "duo_kernel": -1 automatic, 0 never, 1 always.
    // Measured (pipeline, N = 271): 17.5 vs 20.8 us at 250 tracks, 20.8 vs 21.9 us at 500; at 1 000 tracks within +-3 % of the one-wave
    // kernel depending on the batch (22.4 vs 23.3 us on one, 24.6 vs 24.0 us on the bench's), slower from 2 000 on (every SIMD
    // then holds several waves anyway) -- automatic = up to 768 tracks.
    if (pipeline && !offsets && ctx->wave_ppl == 0 && ctx->ekf_variant == 0 && ctx->duo_kernel != 0 && N > 64 && N <= 640 &&
        (ctx->duo_kernel == 1 || B <= 768)) {
        const int stride = (int)((N + 1) & ~(int64_t)1);
        hipLaunchKernelGGL(ekf_wave_duo_kernel<true>, dim3((unsigned)B), dim3(128), (size_t)stride * 9 * sizeof(double), ctx->stream, a, k, stride);
        GSF_HIP(hipGetLastError());
        return GSF_OK;
    }
    // Poses per lane (gsf_set_option "wave_ppl": 0 = automatic, 1..5 forced).  Automatic is ONE pose per lane with a register
    // prefetch at every batch size.  The multi-pose builds (process_chunk<P>) pass the same parity tests but measured slower on
    // MI355X everywhere: C2 K4 18.1 / 21.1 / 21.6 / 25.1 / 30.0 us for P = 1..5, and worse at large batches (2 waves or fewer per
    // SIMD): their per-pose bookkeeping (one ballot set per sub-pose, position <-> lane arithmetic, 300-500 registers with AGPR
    // traffic) costs more than the scan stages they save.  They stay opt-in.
    int ppl = ctx->wave_ppl;
    if (ctx->ekf_variant == 5) ppl = 2;                                  // historical name of the two-pose build
    if (ppl == 0) ppl = 1;
#define GSF_LAUNCH_WAVEP(P) do { if (pipeline) hipLaunchKernelGGL((ekf_wavep_kernel<true, P>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k); \
                                 else hipLaunchKernelGGL((ekf_wavep_kernel<false, P>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k); } while (0)
    switch (ppl) {
    case 2: GSF_LAUNCH_WAVEP(2); break;
    case 3: GSF_LAUNCH_WAVEP(3); break;
    case 4: GSF_LAUNCH_WAVEP(4); break;
    case 5: GSF_LAUNCH_WAVEP(5); break;
    default: {
        // up to 2 048 waves (two per SIMD) the build with inlined cold blocks costs no occupancy; same arithmetic, same bits
        const bool small = B <= 2048;
        if (pipeline) { if (small) hipLaunchKernelGGL((ekf_wave_kernel<true, true>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k);
                        else hipLaunchKernelGGL((ekf_wave_kernel<true, false>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k); }
        else { if (small) hipLaunchKernelGGL((ekf_wave_kernel<false, true>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k);
               else hipLaunchKernelGGL((ekf_wave_kernel<false, false>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k); }
    }
    }
#undef GSF_LAUNCH_WAVEP
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace gsf
