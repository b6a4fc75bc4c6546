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

template <bool PIPELINE>
__global__ __launch_bounds__(64) void ekf_wave_kernel(WaveArgs a, EkfConfig cfg)
{
    wave_serial_body<PIPELINE>(a, cfg, (int64_t)blockIdx.x, (int)threadIdx.x);
}


// Experimental form (ekf_variant 5): 128 poses per iteration (two per lane) while more than 64 poses remain, a 64-pose
// iteration for the tail.  The prelude (fit + Sim3 of pose 0) is shared with the PPL = 1 kernel above via wave_prelude().
template <bool PIPELINE>
__global__ __launch_bounds__(64) void ekf_wave2_kernel(WaveArgs a, EkfConfig cfg)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, N = a.N;
    TrajPtrs T{ a.ts + b * N, a.pos + b * N * 3, a.quat + b * N * 4, a.gps + b * N * 3, a.valid + b * N, a.pos_out + b * N * 3, a.quat_out + b * N * 4, N };
    Vec3 p0; Quat q0; int32_t fit = 0;
    if (!wave_prelude<PIPELINE>(a, b, b * N, N, lane, p0, q0, fit)) return;
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
    // The poses of a chunk are loaded one iteration ahead (before the previous chunk's scans and stores are issued), so
    // their latency is covered and the in-order vmcnt never has to drain the stores to reach them.
    ChunkIn nx0, nx1;
    if (N > 64) {
        nx0 = load_chunk(T.ts, T.pos, T.quat, T.gps, T.valid, 2 * (int64_t)lane, N);
        nx1 = load_chunk(T.ts, T.pos, T.quat, T.gps, T.valid, 2 * (int64_t)lane + 1, N);
    } else {
        nx0 = load_chunk(T.ts, T.pos, T.quat, T.gps, T.valid, lane, N);
        nx1 = nx0;
    }
    chunk_arrived(nx0); chunk_arrived(nx1);
    int64_t c0 = 0;
    while (N - c0 > 64) {
        ChunkIn in[2] = { nx0, nx1 };
        const int64_t n0 = c0 + 128;
        if (N - n0 > 64) {
            nx0 = load_chunk(T.ts, T.pos, T.quat, T.gps, T.valid, n0 + 2 * (int64_t)lane, N);
            nx1 = load_chunk(T.ts, T.pos, T.quat, T.gps, T.valid, n0 + 2 * (int64_t)lane + 1, N);
        } else if (n0 < N) {
            nx0 = load_chunk(T.ts, T.pos, T.quat, T.gps, T.valid, n0 + lane, N);
        }
        process_chunk<2>(T, cfg, C, c0, in, lane, nx0, nx1);
        c0 = n0;
    }
    if (c0 < N) {
        ChunkIn in[1] = { nx0 };
        process_chunk<1>(T, cfg, C, c0, in, lane, nx0, nx0);
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
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, offsets };
    const EkfConfig k = to_core(cfg);
    // Default: one pose per lane.  The two-poses-per-lane build (ekf_variant 5) executes ~29 % fewer VALU instructions but
    // measured slower on MI355X (C3 K4 3.18 vs 2.84 ms, C2 44 vs 21 us: 177 vs 155 VGPRs -> 2 instead of 3 waves/SIMD, and
    // 58 % of its wave time in s_waitcnt); it stays opt-in until that is understood.
    const bool one_per_lane = ctx->ekf_variant != 5 || offsets != nullptr;
    if (pipeline) {
        if (one_per_lane) hipLaunchKernelGGL(ekf_wave_kernel<true>, dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k);
        else hipLaunchKernelGGL(ekf_wave2_kernel<true>, dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k);
    } else {
        if (one_per_lane) hipLaunchKernelGGL(ekf_wave_kernel<false>, dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k);
        else hipLaunchKernelGGL(ekf_wave2_kernel<false>, dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k);
    }
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace gsf
