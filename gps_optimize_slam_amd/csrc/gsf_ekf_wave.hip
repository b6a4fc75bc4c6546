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
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

// AXMODE 1: x and y share (P0, Q, R), z does not (checked by the launcher; the default CONFIG) -- see wave_serial_chunks.
// This file holds the SMALL-batch builds (inlined cold blocks, direct loads and stores; scheduled with iterative-ilp for the lone wave);
// the big-batch builds of the same template live in gsf_ekf_wave_big.hip (slab loads / stores through LDS, scheduled with max-ilp:
// iterative-ilp crashes clang's register allocator on them, and max-ilp is the faster of the two at many waves per SIMD anyway).
template <bool PIPELINE, bool SMALLBATCH, int AXMODE>
__global__ __launch_bounds__(64, 1) void ekf_wave_kernel(WaveArgs a, EkfConfig cfg)
{
    static_assert(SMALLBATCH, "big-batch instantiations belong to gsf_ekf_wave_big.hip");
    wave_serial_body<PIPELINE, false, SMALLBATCH, 1, AXMODE>(a, cfg, (int64_t)blockIdx.x, (int)threadIdx.x);
}


// Two waves per trajectory for SMALL batches of the fused pipeline (one wave per SIMD, every wave in the same phase at the same
// time): while wave 0 is busy with the fit -- a memory burst followed by a latency-bound 3x3 Jacobi chain that leave the SIMD
// mostly idle -- wave 1 computes the variances of EVERY chunk (they depend on stamps and availability only, not on the fit) into
// LDS; after one block barrier wave 0 runs the chunk loop without its two Moebius scans (-28 % instructions per chunk).
// Same functions, same operands, same order as the one-wave kernel: bit-identical results.
#ifndef GSF_DUO_ROLE_SHIFT
#define GSF_DUO_ROLE_SHIFT 2        // measured best of 0..3 at 1 000 tracks (22.6 vs 23.1-23.2 us; 23.6 us without the helper)
#endif
template <bool PIPELINE, int AXMODE>
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
    wave_serial_body<PIPELINE, true, true, 1, AXMODE>(a, cfg, b, lane, gsf_pv, pv_stride);
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
    // tracks of 65..1024 poses: one workgroup per trajectory, one wave per chunk (gsf_ekf_block.hip).  The choice depends on N and the
    // layout only, never on B, so a shard of a batch produces the same bits as the whole batch.
    // (under the reference's row choice its launcher marks the rows of the fit with a launch of sim3_rows_kernel first)
    if (ctx->block_kernel == 1 && ekf_block_applies(N, offsets))
        return launch_ekf_block(ctx, pipeline, ts, pos, quat, gps, valid, init_pos, init_quat, cfg, B, N, R, t, s, pos_out, quat_out, status);
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, offsets, pipeline ? ctx->fit_rows : FitRows{ 0, 0, 0.0, 0.0 } };
    const EkfConfig k = to_core(cfg);
    // x and y share their (P0, Q, R) and z does not (the default CONFIG): the build with that choice of scans compiled in
    const bool xy = k.P0[1] == k.P0[0] && k.Qps[1] == k.Qps[0] && k.Rm[1] == k.Rm[0] &&
                    !(k.P0[2] == k.P0[0] && k.Qps[2] == k.Qps[0] && k.Rm[2] == k.Rm[0]);
    // small batches of the fused pipeline: two waves per trajectory (see ekf_wave_duo_kernel).  Bit-identical to the one-wave
    // kernel, so choosing by batch size does not break shard invariance.  gsf_set_option "duo_kernel": -1 automatic, 0 never, 1 always.
    // Measured with the polar-iteration fit (pipeline, N = 271; tools/duo_sweep.py): 15.5 vs 17.8 us at 256 tracks,
    // 19.4 vs 18.9 us at 512, 20.9 vs 19.5 us at 1 000 (every SIMD then holds a main wave and the helper only competes with it)
    // -- automatic = up to 256 tracks.  The four-trajectory-per-block form of round 2 (main and helper of a trajectory forced onto
    // one SIMD) lost its edge with the shorter fit (20.2 vs 19.5 us at 1 000) and lives in tools/experiments/ now.
    if (pipeline && !offsets && ctx->duo_kernel != 0 && N > 64 && N <= 640 && (ctx->duo_kernel == 1 || (ctx->duo_kernel == -1 && B <= 256))) {
        const int stride = (int)((N + 1) & ~(int64_t)1);
        if (xy) hipLaunchKernelGGL((ekf_wave_duo_kernel<true, 1>), dim3((unsigned)B), dim3(128), (size_t)stride * 9 * sizeof(double), ctx->stream, a, k, stride);
        else hipLaunchKernelGGL((ekf_wave_duo_kernel<true, 0>), dim3((unsigned)B), dim3(128), (size_t)stride * 9 * sizeof(double), ctx->stream, a, k, stride);
        GSF_HIP(hipGetLastError());
        return GSF_OK;
    }
    {
        // up to 2 048 waves (two per SIMD) the build with inlined cold blocks costs no occupancy; same arithmetic, same bits
        const bool small = B <= 2048;
        if (!small)
            return launch_ekf_wave_big(ctx, pipeline, xy, ts, pos, quat, gps, valid, init_pos, init_quat, cfg, B, N, R, t, s, pos_out, quat_out, status, offsets);
#define GSF_LAUNCH_WAVE(P_, X_) hipLaunchKernelGGL((ekf_wave_kernel<P_, true, X_>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k)
        if (pipeline) { if (xy) GSF_LAUNCH_WAVE(true, 1); else GSF_LAUNCH_WAVE(true, 0); }
        else { if (xy) GSF_LAUNCH_WAVE(false, 1); else GSF_LAUNCH_WAVE(false, 0); }
#undef GSF_LAUNCH_WAVE
    }
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace gsf

namespace gsf { const char* wave_small_build_info() { return GSF_TU_BUILD_INFO("gsf_ekf_wave.hip"); } }
