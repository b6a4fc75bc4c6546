// gsf_ekf.hip -- K4: batched EKF predict/update + per-outage RTS (apply_ekf_correction,
// EKFGPSSLAM.py:831-935) for B independent trajectories of N poses.
//
// Mapping: ONE LANE PER TRAJECTORY.  The recursion over poses is inherently serial, so the
// parallel axis is the batch: a wave carries 64 filters, each with its whole state (7-vector,
// 7 variances, weight/outage flags) in VGPRs -- no LDS, no cross-lane traffic, and per-lane
// predication for the data-dependent outage / RTS / sharp-turn control flow.
// In the time-major layout (trajectory index fastest) every load/store of a wave is one
// contiguous 512-B row, i.e. perfectly coalesced; a PF-deep register ring keeps PF future
// poses per lane in flight so that HBM latency is covered even at ~1.5 waves per SIMD
// (B = 100k gives only 1563 waves on 1024 SIMDs).
// The trajectory-major variant runs the same per-lane code on strided addresses (functional
// drop-in for stacked TUM arrays; the bandwidth path is the time-major one).
//
// Algorithmic HBM traffic: 89 B read + 56 B written per pose.
#include "gsf_internal.hpp"

using namespace gsf;

namespace {

// Per-lane accessors.  In the time-major layout the address splits into a wave-UNIFORM row base
// ((i*C + c)*B, scalar registers / SALU) plus the constant per-lane offset b, so the recursion spends
// no VALU work on addressing; the trajectory-major layout has a per-lane row base instead.
template <int LAYOUT>
struct LaneIO {
    int64_t B, N, b;
    const double* __restrict__ ts; const double* __restrict__ pos; const double* __restrict__ quat;
    const double* __restrict__ gps; const uint8_t* __restrict__ valid;
    double* __restrict__ pos_out; double* __restrict__ quat_out;

    template <typename T>
    __device__ __forceinline__ T* at(T* base, int64_t i, int c, int C) const
    {
        // uniform 64-bit row base (SGPRs) + 32-bit zero-extended lane BYTE offset -> `global_load v, v_off, s[base]` addressing
        if (LAYOUT == GSF_LAYOUT_TIME_MAJOR)
            return (T*)((const char*)(base + (i * C + c) * B) + (uint32_t)((uint32_t)b * (uint32_t)sizeof(T)));
        return base + ((b * N + i) * C + c);
    }
    __device__ __forceinline__ StepIn load_step(int64_t i) const
    {
        StepIn s;
        s.t = *at(ts, i, 0, 1);
        s.p = Vec3{ *at(pos, i, 0, 3), *at(pos, i, 1, 3), *at(pos, i, 2, 3) };
        s.q = Quat{ *at(quat, i, 0, 4), *at(quat, i, 1, 4), *at(quat, i, 2, 4), *at(quat, i, 3, 4) };
        s.z = Vec3{ *at(gps, i, 0, 3), *at(gps, i, 1, 3), *at(gps, i, 2, 3) };
        s.valid = *at(valid, i, 0, 1);
        return s;
    }
    __device__ __forceinline__ void store(int64_t i, const Vec3& p, const Quat& q)
    {
        *at(pos_out, i, 0, 3) = p.x; *at(pos_out, i, 1, 3) = p.y; *at(pos_out, i, 2, 3) = p.z;
        *at(quat_out, i, 0, 4) = q.x; *at(quat_out, i, 1, 4) = q.y; *at(quat_out, i, 2, 4) = q.z; *at(quat_out, i, 3, 4) = q.w;
    }
    __device__ __forceinline__ void load(int64_t i, Vec3& p, Quat& q) const
    {
        p = Vec3{ *at(pos_out, i, 0, 3), *at(pos_out, i, 1, 3), *at(pos_out, i, 2, 3) };
        q = Quat{ *at(quat_out, i, 0, 4), *at(quat_out, i, 1, 4), *at(quat_out, i, 2, 4), *at(quat_out, i, 3, 4) };
    }
    __device__ __forceinline__ double stamp(int64_t i) const { return *at(ts, i, 0, 1); }
};

// PF = depth of the per-lane prefetch ring (poses in flight per trajectory)
template <int LAYOUT, int PF, int OCC>
__global__ __launch_bounds__(64, OCC) void ekf_fuse_kernel(const double* __restrict__ ts, const double* __restrict__ pos,
                                                      const double* __restrict__ quat, const double* __restrict__ gps,
                                                      const uint8_t* __restrict__ valid, const double* __restrict__ init_pos,
                                                      const double* __restrict__ init_quat, EkfConfig cfg, int64_t B, int64_t N,
                                                      double* __restrict__ pos_out, double* __restrict__ quat_out,
                                                      int32_t* __restrict__ status)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    LaneIO<LAYOUT> io{ B, N, b, ts, pos, quat, gps, valid, pos_out, quat_out };
    const Vec3 p0{ init_pos[b * 3], init_pos[b * 3 + 1], init_pos[b * 3 + 2] };
    const Quat q0{ init_quat[b * 4], init_quat[b * 4 + 1], init_quat[b * 4 + 2], init_quat[b * 4 + 3] };
    EkfTraj<LaneIO<LAYOUT>> f;
    f.init(cfg, p0, q0, io.load_step(0), io);
    // PF statically-named register buffers (the loop is unrolled PF times, so no rotation moves -- a move would force
    // a wait on the newest load): the load for pose i+PF is issued right before pose i is consumed, which keeps PF-1
    // whole steps of compute between a load and its first use.
    StepIn buf[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) buf[d] = io.load_step((1 + d < N) ? 1 + d : N - 1);
    for (int64_t i = 1; i < N; i += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int64_t ii = i + d;
            if (ii < N) {                                       // wave-uniform
                const StepIn cur = buf[d];
                const int64_t nx = ii + PF;
                buf[d] = io.load_step(nx < N ? nx : N - 1);    // clamped re-read at the tail
                f.step(cfg, ii, cur, io);
            }
        }
    }
    if (status) status[b] = f.finish();
}

// Fused steps 3-5 of main_process_gui (ref :1002-1010) in ONE launch, lane per trajectory:
//   pass A  Umeyama moments of (SLAM position, aligned GNSS) over the valid rows -- single pass on data shifted
//           by the first valid row (well conditioned at UTM magnitudes), 49 B/pose;
//   lane    3x3 Jacobi SVD + closed form (64 fits per wave at once), Sim3 of pose 0 only;
//   pass B  the EKF+RTS recursion above, 89 B read + 56 B written per pose.
template <int LAYOUT, int PF, int OCC>
__global__ __launch_bounds__(64, OCC) void fuse_pipeline_kernel(const double* __restrict__ ts, const double* __restrict__ pos,
                                                           const double* __restrict__ quat, const double* __restrict__ gps,
                                                           const uint8_t* __restrict__ valid, EkfConfig cfg, FitRows rows, int64_t B, int64_t N,
                                                           double* __restrict__ Rout, double* __restrict__ tout, double* __restrict__ sout,
                                                           double* __restrict__ pos_out, double* __restrict__ quat_out,
                                                           int32_t* __restrict__ status)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    LaneIO<LAYOUT> io{ B, N, b, ts, pos, quat, gps, valid, pos_out, quat_out };
    const Idx<LAYOUT> ix{ B, N };
    // ---- pass A.  Under gsf_set_sim3_rows mode 1 the rows are main_process_gui's choice (ref :973-998): serial per lane, so the row in
    // front of the gap (left out by V[:k], :981-982) is simply held back -- a valid row is accumulated when the NEXT valid row has been
    // seen not to open a gap, or at the end of the track; the two fall-backs repeat the pass for the lanes that need them.
    double n = 0.0, as[3] = { 0, 0, 0 }, bs[3] = { 0, 0, 0 }, Sa[3] = { 0, 0, 0 }, Sb[3] = { 0, 0, 0 }, Saa = 0.0;
    double Sab[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int32_t rows_flag = 0;
    const bool ref_rows = rows.mode != 0;
    int state = ref_rows ? 0 : 2;                                        // 0 the timed subset, 1 the whole first segment, 2 all valid rows
    bool detect = ref_rows, use_tlim = ref_rows;
    int64_t row_end = N;
    for (int attempt = 0; attempt < 2; ++attempt) {
        n = 0.0; Saa = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { Sa[k] = 0.0; Sb[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < 9; ++k) Sab[k] = 0.0;
        bool have_first = false, have_pend = false, pend_in = false, gap_found = false;
        double tlim = 0.0, t_pend = 0.0, pa[3] = { 0, 0, 0 }, pc[3] = { 0, 0, 0 };
        int64_t i_pend = 0;
        int nvalid = 0;
        auto accumulate = [&](const double a0, const double a1, const double a2, const double c0, const double c1, const double c2) __attribute__((always_inline)) {
            Sa[0] += a0; Sa[1] += a1; Sa[2] += a2; Sb[0] += c0; Sb[1] += c1; Sb[2] += c2;
            Saa += a0 * a0 + a1 * a1 + a2 * a2;
            Sab[0] += a0 * c0; Sab[1] += a0 * c1; Sab[2] += a0 * c2;
            Sab[3] += a1 * c0; Sab[4] += a1 * c1; Sab[5] += a1 * c2;
            Sab[6] += a2 * c0; Sab[7] += a2 * c1; Sab[8] += a2 * c2;
            n += 1.0;
        };
        for (int64_t i = 0; i < N && i < row_end; ++i) {
            const double z0 = gps[ix.at(b, i, 0, 3)], z1 = gps[ix.at(b, i, 1, 3)], z2 = gps[ix.at(b, i, 2, 3)];
            const bool ok = valid[ix.at(b, i, 0, 1)] != 0 && !(isnan(z0) || isnan(z1) || isnan(z2));
            const double p0 = pos[ix.at(b, i, 0, 3)], p1 = pos[ix.at(b, i, 1, 3)], p2 = pos[ix.at(b, i, 2, 3)];
            if (!ok) continue;
            if (!have_first) { as[0] = p0; as[1] = p1; as[2] = p2; bs[0] = z0; bs[1] = z1; bs[2] = z2; have_first = true; }
            const double a0 = p0 - as[0], a1 = p1 - as[1], a2 = p2 - as[2];
            const double c0 = z0 - bs[0], c1 = z1 - bs[1], c2 = z2 - bs[2];
            if (!ref_rows) { accumulate(a0, a1, a2, c0, c1, c2); continue; }
            const double t = ts[ix.at(b, i, 0, 1)];
            if (nvalid == 0) tlim = t + rows.max_dur;                    // segment_start_time + max_dur (:988-990)
            if (have_pend) {
                if (detect && t - t_pend > rows.max_gap) { gap_found = true; row_end = i_pend; break; }   // np.diff > threshold (:979-982): V[:k] drops the held-back row too
                if (pend_in) accumulate(pa[0], pa[1], pa[2], pc[0], pc[1], pc[2]);
            }
            pa[0] = a0; pa[1] = a1; pa[2] = a2; pc[0] = c0; pc[1] = c1; pc[2] = c2;
            t_pend = t; i_pend = i; have_pend = true; pend_in = !use_tlim || t <= tlim; ++nvalid;
        }
        if (!ref_rows) break;
        if (have_pend && !gap_found && pend_in) accumulate(pa[0], pa[1], pa[2], pc[0], pc[1], pc[2]);
        if (state != 0) { if (state == 2 && nvalid < rows.min_samples) rows_flag = SIM3_FLAG_FEW_ROWS; break; }   // :975
        const int nF = gap_found ? nvalid - 1 : nvalid;                  // valid rows in front of the held-back one
        detect = false;
        if (nF < rows.min_samples) {                                     // :983
            if (!gap_found) { rows_flag = SIM3_FLAG_FEW_ROWS; break; }   // V itself is that short (:975)
            row_end = N; use_tlim = false; state = 2; rows_flag = SIM3_FLAG_ROWS_ALL;      // :984
        } else if ((int)n < rows.min_samples) { use_tlim = false; state = 1; rows_flag = SIM3_FLAG_ROWS_SEGMENT; }   // :993-995
        else break;                                                      // :996
    }
    double Rb[9], tb[3], sb = NAN; int32_t fit = SIM3_NONE;
    if (n >= 3.0 && !(rows_flag & SIM3_FLAG_FEW_ROWS)) {
        const double rn = 1.0 / n;
        const double ma[3] = { Sa[0] * rn, Sa[1] * rn, Sa[2] * rn }, mb[3] = { Sb[0] * rn, Sb[1] * rn, Sb[2] * rn };
        double H[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) H[r * 3 + c] = Sab[r * 3 + c] - n * ma[r] * mb[c];
        const double ssq = fmax(0.0, Saa - n * (ma[0] * ma[0] + ma[1] * ma[1] + ma[2] * ma[2]));
        const double sc[3] = { as[0] + ma[0], as[1] + ma[1], as[2] + ma[2] }, dc[3] = { bs[0] + mb[0], bs[1] + mb[1], bs[2] + mb[2] };
        fit = umeyama_finalize(H, ssq, sc, dc, n, Rb, tb, sb);
    }
    const StepIn first = io.load_step(0);
    Quat q0n; const bool q0ok = quat_unit(first.q, q0n);
    if (fit == SIM3_NONE || !q0ok) {
        for (int k = 0; k < 9; ++k) Rout[b * 9 + k] = NAN;
        tout[b * 3] = tout[b * 3 + 1] = tout[b * 3 + 2] = NAN; sout[b] = NAN;
        const Vec3 pn{ NAN, NAN, NAN }; const Quat qn{ NAN, NAN, NAN, NAN };
        for (int64_t i = 0; i < N; ++i) io.store(i, pn, qn);
        if (status) status[b] = (fit == SIM3_NONE ? ((SIM3_NONE | (rows_flag & SIM3_FLAG_FEW_ROWS)) << 8) : 0) | (q0ok ? 0 : ST_BAD_QUAT);
        return;
    }
    fit |= rows_flag;
#pragma unroll
    for (int k = 0; k < 9; ++k) Rout[b * 9 + k] = Rb[k];
    tout[b * 3] = tb[0]; tout[b * 3 + 1] = tb[1]; tout[b * 3 + 2] = tb[2]; sout[b] = sb;
    // ---- Sim3 of pose 0 (transform_trajectory row 0, ref :464-466)
    const Vec3 p0{ sb * (first.p.x * Rb[0] + first.p.y * Rb[1] + first.p.z * Rb[2]) + tb[0],
                   sb * (first.p.x * Rb[3] + first.p.y * Rb[4] + first.p.z * Rb[5]) + tb[1],
                   sb * (first.p.x * Rb[6] + first.p.y * Rb[7] + first.p.z * Rb[8]) + tb[2] };
    const Quat q0 = quat_mul(quat_from_matrix(Rb), q0n);
    // ---- pass B
    EkfTraj<LaneIO<LAYOUT>> f;
    f.init(cfg, p0, q0, first, io);
    // PF statically-named register buffers (the loop is unrolled PF times, so no rotation moves -- a move would force
    // a wait on the newest load): the load for pose i+PF is issued right before pose i is consumed, which keeps PF-1
    // whole steps of compute between a load and its first use.
    StepIn buf[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) buf[d] = io.load_step((1 + d < N) ? 1 + d : N - 1);
    for (int64_t i = 1; i < N; i += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int64_t ii = i + d;
            if (ii < N) {                                       // wave-uniform
                const StepIn cur = buf[d];
                const int64_t nx = ii + PF;
                buf[d] = io.load_step(nx < N ? nx : N - 1);    // clamped re-read at the tail
                f.step(cfg, ii, cur, io);
            }
        }
    }
    if (status) status[b] = f.finish() | (fit << 8);
}

EkfConfig to_core(const gsf_ekf_config* c)
{
    EkfConfig k;
    for (int i = 0; i < 7; ++i) { k.P0[i] = c->initial_cov_diag[i]; k.Qps[i] = c->process_noise_diag[i]; }
    for (int i = 0; i < 3; ++i) k.Rm[i] = c->meas_noise_diag[i];
    k.yaw_thr_rad = c->sharp_turn_yaw_rate_threshold_deg_per_sec * (M_PI / 180.0);     // np.deg2rad, ref :886
    k.sharp_turn_steps = c->default_ekf_transition_steps_on_sharp_turn;
    k._pad = 0;
    return k;
}

}  // namespace

// A time-major batch with fewer trajectories than ctx->lane_min_traj cannot fill the chip with one lane per trajectory (64 tracks per
// wave: 1 000 tracks are 16 waves on 1 024 SIMDs, 0.9 ms instead of 25 us at C2).  Such batches are transposed into the context's
// workspace, fused by the wave-per-trajectory kernel and transposed back: 3 x 145 B/pose of traffic instead of 145, which still wins
// below ~32 k trajectories (measured crossover).  Workspace: 145 B/pose, capped at 8 GB.
static bool route_time_major_through_wave(const gsf_ctx* ctx, int64_t B, int64_t N)
{
    return B < ctx->lane_min_traj && (double)B * (double)N * 145.0 <= 8.0e9;
}

static int fuse_time_major_via_wave(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                                    const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                                    int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status)
{
    const size_t P = (size_t)B * (size_t)N;
    int rc = ensure_scratch(ctx, P * (18 * 8) + ((P + 7) & ~(size_t)7) + 64);
    if (rc) return rc;
    double* wts = (double*)ctx->scratch; double* wpos = wts + P; double* wquat = wpos + 3 * P; double* wgps = wquat + 4 * P;
    double* wpo = wgps + 3 * P; double* wqo = wpo + 3 * P; uint8_t* wval = (uint8_t*)(wqo + 4 * P);
    // three launches: the five inputs in one transpose, the wave kernel, the two outputs in one transpose
    const void* isrc[5] = { ts, pos, quat, gps, valid }; void* idst[5] = { wts, wpos, wquat, wgps, wval };
    const int iC[5] = { 1, 3, 4, 3, 1 }, ib[5] = { 8, 8, 8, 8, 1 };
    if ((rc = launch_transpose_set(ctx, false, 5, isrc, idst, iC, ib, B, N))) return rc;
    if ((rc = launch_ekf_wave(ctx, pipeline, wts, wpos, wquat, wgps, wval, init_pos, init_quat, cfg, B, N, R, t, s, wpo, wqo, status))) return rc;
    const void* osrc[2] = { wpo, wqo }; void* odst[2] = { pos_out, quat_out };
    const int oC[2] = { 3, 4 }, ob[2] = { 8, 8 };
    return launch_transpose_set(ctx, true, 2, osrc, odst, oC, ob, B, N);
}

extern "C" int gsf_ekf_fuse_batch_dev(gsf_ctx* ctx, int32_t layout, const double* ts, const double* pos, const double* quat,
                                      const double* gps, const uint8_t* valid, const double* init_pos, const double* init_quat,
                                      const gsf_ekf_config* cfg, int64_t B, int64_t N, double* pos_out, double* quat_out,
                                      int32_t* status)
{
    GSF_REQUIRE(ctx && cfg, "ctx/cfg is NULL");
    GSF_REQUIRE(B >= 0 && N >= 0, "negative B or N");
    GSF_REQUIRE(layout == GSF_LAYOUT_TRAJ_MAJOR || layout == GSF_LAYOUT_TIME_MAJOR, "unknown layout");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps && valid && init_pos && init_quat && pos_out && quat_out, "NULL array");
    GSF_REQUIRE(B <= (int64_t)0x7fffffff * 64, "B too large for one launch");
    for (int i = 0; i < 7; ++i) GSF_REQUIRE(cfg->initial_cov_diag[i] == cfg->initial_cov_diag[i], "NaN in config");
    GSF_HIP(hipSetDevice(ctx->device));
    if (layout == GSF_LAYOUT_TRAJ_MAJOR)                               // wave-per-trajectory scans (gsf_ekf_wave.hip)
        return launch_ekf_wave(ctx, false, ts, pos, quat, gps, valid, init_pos, init_quat, cfg, B, N, nullptr, nullptr, nullptr, pos_out, quat_out, status);
    if (route_time_major_through_wave(ctx, B, N))
        return fuse_time_major_via_wave(ctx, false, ts, pos, quat, gps, valid, init_pos, init_quat, cfg, B, N, nullptr, nullptr, nullptr, pos_out, quat_out, status);
    const EkfConfig k = to_core(cfg);
    const dim3 block(64), grid((unsigned)((B + 63) / 64));
    hipLaunchKernelGGL((ekf_fuse_kernel<GSF_LAYOUT_TIME_MAJOR, 2, 2>), grid, block, 0, ctx->stream, ts, pos, quat, gps, valid, init_pos, init_quat, k, B, N,
                       pos_out, quat_out, status);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

extern "C" int gsf_fuse_pipeline_batch_dev(gsf_ctx* ctx, int32_t layout, const double* ts, const double* pos, const double* quat,
                                           const double* gps, const uint8_t* valid, const gsf_ekf_config* cfg, int64_t B, int64_t N,
                                           double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(ctx && cfg, "ctx/cfg is NULL");
    GSF_REQUIRE(B >= 0 && N >= 0, "negative B or N");
    GSF_REQUIRE(layout == GSF_LAYOUT_TRAJ_MAJOR || layout == GSF_LAYOUT_TIME_MAJOR, "unknown layout");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps && valid && R && t && s && pos_out && quat_out, "NULL array");
    GSF_REQUIRE(B <= (int64_t)0x7fffffff * 64, "B too large for one launch");
    GSF_HIP(hipSetDevice(ctx->device));
    if (layout == GSF_LAYOUT_TRAJ_MAJOR)
        return launch_ekf_wave(ctx, true, ts, pos, quat, gps, valid, nullptr, nullptr, cfg, B, N, R, t, s, pos_out, quat_out, status);
    if (route_time_major_through_wave(ctx, B, N))
        return fuse_time_major_via_wave(ctx, true, ts, pos, quat, gps, valid, nullptr, nullptr, cfg, B, N, R, t, s, pos_out, quat_out, status);
    const EkfConfig k = to_core(cfg);
    const dim3 block(64), grid((unsigned)((B + 63) / 64));
    hipLaunchKernelGGL((fuse_pipeline_kernel<GSF_LAYOUT_TIME_MAJOR, 2, 2>), grid, block, 0, ctx->stream, ts, pos, quat, gps, valid, k, ctx->fit_rows, B, N,
                       R, t, s, pos_out, quat_out, status);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

// ---- ragged batches: B trajectories of different lengths, rows offsets[b]..offsets[b+1] of flat [total][C] arrays -------------
extern "C" int gsf_ekf_fuse_ragged_dev(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps,
                                       const uint8_t* valid, const int64_t* offsets, const double* init_pos, const double* init_quat,
                                       const gsf_ekf_config* cfg, int64_t B, double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(ctx && cfg && offsets, "ctx/cfg/offsets is NULL");
    GSF_REQUIRE(B >= 0, "negative B");
    if (B == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps && valid && init_pos && init_quat && pos_out && quat_out, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_ekf_wave(ctx, false, ts, pos, quat, gps, valid, init_pos, init_quat, cfg, B, 0, nullptr, nullptr, nullptr, pos_out, quat_out,
                           status, offsets);
}

extern "C" int gsf_fuse_pipeline_ragged_dev(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps,
                                            const uint8_t* valid, const int64_t* offsets, const gsf_ekf_config* cfg, int64_t B, double* R,
                                            double* t, double* s, double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(ctx && cfg && offsets, "ctx/cfg/offsets is NULL");
    GSF_REQUIRE(B >= 0, "negative B");
    if (B == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps && valid && R && t && s && pos_out && quat_out, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_ekf_wave(ctx, true, ts, pos, quat, gps, valid, nullptr, nullptr, cfg, B, 0, R, t, s, pos_out, quat_out, status, offsets);
}
