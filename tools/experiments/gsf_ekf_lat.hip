// gsf_ekf_lat.hip -- the LATENCY build of the fused pipeline (Umeyama on the valid rows -> Sim3 of pose 0 -> EKF + per-outage RTS,
// EKFGPSSLAM.py:1002-1010) for small batches of short tracks (config C2: 1 000 x 271 poses).
//
// With one wave per SIMD a launch lasts as long as its slowest wave, and a wave spends ~9 us in the fit (a memory burst, 17
// reductions and a 3x3 Jacobi SVD on dependent FP64 chains) before it can start the filter, whose ~550 instructions per 64 poses
// are all downstream of the fit's R, t, s.  But the filter is LINEAR in what the fit delivers: with the gains known (they depend on
// stamps and availability only) the position recursion  x_i = a_i (x_{i-1} + R dp_i) + b_i z_i  (a_i = 1 - k_i w_i, b_i = k_i w_i
// with a fix; a_i = 1, b_i = 0 without) unrolls to
//        x_i - O  =  A_i (x_0 - O)  +  Z_i  +  R W_i                                   per axis, O = a fit-independent origin,
//        A_i = prod a,     Z_i = a_i Z_{i-1} + b_i (z_i - O),     W_i = a_i (W_{i-1} + dp_i)   (one W per axis group and dp component),
// and the orientation telescopes to q_i = Cq r_i (gsf_wave_common.hpp).  So a HELPER wave runs every scan of the track -- variances,
// outage structure, A, Z, W, the RTS coefficients P_f[k]/P_p[r] and their recovery indices -- into LDS WHILE the main wave fits; after
// one barrier the main wave only evaluates the line above (12 FMAs per pose), the RTS correction x_s[k] = x_f[k] + g_k (x_f[r] - x_p[r])
// and q_i, and streams the rows out.  Four trajectories per 512-thread block: main wave k and helper wave k + 4 share SIMD k
// (tools/ubench/placement.hip), one block per CU for up to 1 024 trajectories.
//
// Same filter, other evaluation order than the chunked kernel (rounding differs at the 1e-10 m level; gate 1e-6 m), so the choice
// between the two must not depend on how a batch is sharded: it is made from the TRACK LENGTH and the batch-size CLASS only
// (launch_ekf_wave), and tracks the linear form does not cover (an invalid SLAM quaternion, per-axis noise that differs between x
// and y) are finished by the chunked code inside the same kernel.
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

// LDS traffic of ONE wave is ordered by the hardware; this pins the compiler's order and drains the queue before lanes read what other
// lanes of the same wave wrote
#define GSF_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

constexpr int LAT_MAX_N = 320;            // 5 chunks: 14 doubles + 1 int per pose and trajectory in LDS, four trajectories per block
constexpr int LAT_ARRAYS = 14;            // Axy Az | Z0 Z1 Z2 | Wxy0..2 | Wz0..2 | G0 G1 G2
constexpr int LAT_MISC = 16;              // O[3], flags

struct LatSlot {
    double* Axy; double* Az; double* Z[3]; double* Wxy[3]; double* Wz[3]; double* G[3]; int32_t* ridx; double* misc;
};
__device__ __forceinline__ LatSlot lat_slot(double* base, int S)
{
    LatSlot s;
    s.Axy = base; s.Az = base + S;
    for (int c = 0; c < 3; ++c) { s.Z[c] = base + (2 + c) * S; s.Wxy[c] = base + (5 + c) * S; s.Wz[c] = base + (8 + c) * S; s.G[c] = base + (11 + c) * S; }
    s.misc = base + LAT_ARRAYS * S;
    s.ridx = (int32_t*)(s.misc + LAT_MISC);
    return s;
}
__host__ __device__ inline size_t lat_slot_doubles(int S) { return (size_t)LAT_ARRAYS * S + LAT_MISC + (size_t)(S + 1) / 2; }

// misc layout
enum { M_O0 = 0, M_O1, M_O2, M_GENERIC, M_STATUS, M_HAVE_O, M_CHUNKS_DONE };

// ---- helper wave 1: variances, GNSS gate, outage structure (ref :864-930 minus everything that needs the Sim3 fit) -> per pose the
// blended gains k_xy, k_z (0 without a fix), the RTS coefficient P_f[k] / P_p[r] with its recovery pose r, and per chunk a "done" flag
__device__ __forceinline__ void lat_helper_gains(const WaveArgs& a, const EkfConfig& cfg, const int64_t b, const int lane, const LatSlot& L)
{
    const int64_t base = b * a.N, N = a.N;
    const double* __restrict__ tsb = a.ts + base;
    const double* __restrict__ quatb = a.quat + base * 4;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;
    struct In { double t; Quat q; Vec3 z; uint32_t v; };
    auto load = [&](const int64_t i) __attribute__((always_inline)) {
        const int64_t il = i < N ? i : N - 1;
        return In{ tsb[il], Quat{ quatb[il * 4], quatb[il * 4 + 1], quatb[il * 4 + 2], quatb[il * 4 + 3] }, Vec3{ gpsb[il * 3], gpsb[il * 3 + 1], gpsb[il * 3 + 2] }, valb[il] };
    };
    GSF_STAMP(19);
    In nxt = load(lane);
    double cP[3] = { cfg.P0[0], cfg.P0[1], cfg.P0[2] };
    int64_t c_ostart = 0;
    bool c_seg_sharp = false, have_O = false, generic = false;
    const int same2 = (cfg.P0[2] == cfg.P0[0] && cfg.Qps[2] == cfg.Qps[0] && cfg.Rm[2] == cfg.Rm[0]) ? 0 : -1;
    bool c_prev_avail = __builtin_amdgcn_readlane((int)nxt.v, 0) != 0;   // ref :848 (raw mask)
    Quat c_r; bool c_ok = quat_unit(lane_bcast(nxt.q, 0), c_r);
    double c_t = lane_bcast(nxt.t, 0);
    int32_t status = c_prev_avail ? 0 : ST_HAD_OUTAGE;
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        const int64_t i = c0 + lane;
        const bool active = i < N, is_init = (i == 0), stepping = active && !is_init;
        const int Ll = (int)((N - c0 < 64) ? (N - c0 - 1) : 63);
        const In in = nxt;
        if (c0 + 64 < N) nxt = load(c0 + 64 + lane);
        const double t = in.t;
        const Vec3 z = in.z;
        const bool vraw = in.v != 0;
        Quat r; const bool ok = quat_unit(in.q, r);
        const double t_pr = prev_lane(c_t, t);
        const u64 ok_mask = __ballot(ok), act_mask = __ballot(active);
        if (!c_ok || (ok_mask & act_mask) != act_mask) generic = true;    // an invalid quaternion: the linear form does not apply (ref :84-86)
        const bool ok_pr = (lane == 0) ? c_ok : (((ok_mask >> (lane - 1)) & 1ull) != 0ull);
        const double dt = fmax(1e-6, t - t_pr);                          // ref :865
        const bool both_ok = ok_pr && ok;
        // ---- GNSS gate (ref :867-869) and the outage structure of the chunk as ballots (same rules as wave_serial_chunks)
        const bool zfin = !(isnan(z.x) || isnan(z.y) || isnan(z.z));
        const bool avail = stepping && vraw && zfin;
        const bool av = is_init ? vraw : avail;
        const u64 a_mask = __ballot(active && av);
        const bool ap = (lane == 0) ? (is_init ? true : c_prev_avail) : (((a_mask >> (lane - 1)) & 1ull) != 0ull);
        const bool starts = active && !av && ap;
        const bool recovers = stepping && av && !ap;
        const bool outpair = stepping && !av && !ap;
        const u64 start_mask = __ballot(starts), rec_mask = __ballot(recovers), pair_mask = __ballot(outpair);
        if (start_mask != 0ull) status |= ST_HAD_OUTAGE;
        u64 f_mask = 0ull;
        if (pair_mask != 0ull) {
            const Quat r_pr = prev_lane(c_r, r);
            bool f = false;
            if (outpair && t > t_pr) f = !both_ok || yaw_rate_exceeds_body(r_pr, r, t - t_pr, cfg.yaw_thr_rad);
            f_mask = __ballot(f);
        }
        bool sharp = false;
        if (recovers) {
            const u64 sm = start_mask & bits(0, lane - 1);
            int64_t s_glob; bool seg;
            if (sm != 0ull) {
                const int s = 63 - __clzll((long long)sm);
                s_glob = c0 + s;
                seg = (f_mask & bits(s + 1, lane - 1)) != 0ull;
            } else {
                s_glob = c_ostart;
                seg = c_seg_sharp || (f_mask & bits(0, lane - 1)) != 0ull;
            }
            sharp = (i - s_glob >= 2) && seg;
        }
        const u64 sharp_mask = __ballot(sharp);
        const u64 rts_mask = rec_mask & ~sharp_mask;
        if (sharp_mask != 0ull) status |= ST_SHARP_TURN;
        if (rts_mask != 0ull) status |= ST_RTS_APPLIED;
        double wgt = 1.0;
        if (sharp && cfg.sharp_turn_steps > 1) wgt = 1.0 / (double)cfg.sharp_turn_steps;
        // ---- origin of the additive part: the first row with a usable fix (pose 0 included), fit-independent
        if (!have_O) {
            const u64 m = __ballot(active && vraw && zfin);
            if (m != 0ull) {
                const int f = __ffsll((long long)m) - 1;
                const double o0 = lane_bcast(z.x, f), o1 = lane_bcast(z.y, f), o2 = lane_bcast(z.z, f);
                if (lane == 0) { L.misc[M_O0] = o0; L.misc[M_O1] = o1; L.misc[M_O2] = o2; }
                have_O = true;
            }
        }
        // ---- variances (ref :712-713, :723-731); x and y share the recursion (checked by the launcher)
        AxisVar v0, v1, v2;
        variance_chunk(cfg, 0, same2, dt, stepping, avail, cP[0], cP[1], cP[2], v0, v1, v2);
        const double Pf[3] = { v0.Pf, v1.Pf, v2.Pf }, Pm[3] = { v0.Pm, v1.Pm, v2.Pm };
        if (active) {
            L.Axy[i] = avail ? v0.kg * wgt : 0.0; L.Az[i] = avail ? v2.kg * wgt : 0.0;      // the gains, until helper 2 replaces them by A_xy, A_z
            L.G[0][i] = Pf[0]; L.G[1][i] = Pf[1]; L.G[2][i] = Pf[2];
            L.ridx[i] = -1;
        }
        // ---- per-outage RTS (ref :906-922): coefficient P_f[k] / P_p[r] and the recovery pose r for every pose of a smoothed run
        if (rts_mask != 0ull) {
            const u64 later = rec_mask & ~bits(0, lane);
            const int rl = later != 0ull ? __ffsll((long long)later) - 1 : 0;
            const bool in_run = active && !av && later != 0ull && (((rts_mask >> rl) & 1ull) != 0ull);
            const double pr0 = shidx(Pm[0], rl), pr1 = shidx(Pm[1], rl), pr2 = shidx(Pm[2], rl);
            if (in_run) {
                L.G[0][i] = Pf[0] * fast_rcp(pr0); L.G[1][i] = Pf[1] * fast_rcp(pr1); L.G[2][i] = Pf[2] * fast_rcp(pr2);
                L.ridx[i] = (int32_t)(c0 + rl);
            }
            if (!c_prev_avail) {                                         // a run carried in from earlier chunks, closed by the first recovery here
                const int r1 = __ffsll((long long)rec_mask) - 1;
                if ((rts_mask >> r1) & 1ull) {
                    const double ipr0 = fast_rcp(lane_bcast(Pm[0], r1)), ipr1 = fast_rcp(lane_bcast(Pm[1], r1)), ipr2 = fast_rcp(lane_bcast(Pm[2], r1));
                    GSF_LDS_FENCE();
                    for (int64_t k = c_ostart + lane; k < c0; k += 64) {
                        L.G[0][k] *= ipr0; L.G[1][k] *= ipr1; L.G[2][k] *= ipr2;   // rows of the run written by earlier chunks still hold P_f[k]
                        L.ridx[k] = (int32_t)(c0 + r1);
                    }
                }
            }
        }
        // ---- carries
        const bool open = ((a_mask >> Ll) & 1ull) == 0ull;
        if (open) {
            const u64 sm = start_mask & bits(0, Ll);
            if (sm != 0ull) {
                const int s = 63 - __clzll((long long)sm);
                c_ostart = c0 + s;
                c_seg_sharp = (f_mask & bits(s + 1, Ll)) != 0ull;
            } else {
                c_seg_sharp = c_seg_sharp || (f_mask & bits(0, Ll)) != 0ull;
            }
        }
        c_prev_avail = !open;
        cP[0] = lane_bcast(Pf[0], Ll); cP[1] = lane_bcast(Pf[1], Ll); cP[2] = lane_bcast(Pf[2], Ll);
        c_r = lane_bcast(r, Ll); c_ok = ((ok_mask >> Ll) & 1ull) != 0ull; c_t = lane_bcast(t, Ll);
        // chunk done: its gains (and the origin) are visible to helper 2
        GSF_LDS_FENCE();
        if (lane == 0) L.misc[M_CHUNKS_DONE] = (double)(c0 / 64 + 1);
        GSF_STAMP(20 + (int)(c0 / 64));
    }
    if (lane == 0) {
        L.misc[M_GENERIC] = generic ? 1.0 : 0.0;
        L.misc[M_STATUS] = (double)(status | (c_prev_avail ? 0 : ST_ENDED_IN_OUTAGE));
        L.misc[M_HAVE_O] = have_O ? 1.0 : 0.0;
    }
}

// ---- helper wave 2: the scans of the linear form, chunk by chunk behind helper 1
__device__ __forceinline__ void lat_helper_scans(const WaveArgs& a, const int64_t b, const int lane, const LatSlot& L)
{
    const int64_t base = b * a.N, N = a.N;
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ gpsb = a.gps + base * 3;
    struct In { Vec3 p, z; };
    auto load = [&](const int64_t i) __attribute__((always_inline)) {
        const int64_t il = i < N ? i : N - 1;
        return In{ Vec3{ posb[il * 3], posb[il * 3 + 1], posb[il * 3 + 2] }, Vec3{ gpsb[il * 3], gpsb[il * 3 + 1], gpsb[il * 3 + 2] } };
    };
    In nxt = load(lane);
    double cA[2] = { 1.0, 1.0 }, cZ[3] = { 0.0, 0.0, 0.0 }, cWxy[3] = { 0.0, 0.0, 0.0 }, cWz[3] = { 0.0, 0.0, 0.0 };
    Vec3 c_po = lane_bcast(nxt.p, 0);                                    // "previous original pose" of pose 0 is pose 0 itself (ref :858)
    const volatile double* flag = L.misc + M_CHUNKS_DONE;
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        const int64_t i = c0 + lane;
        const bool active = i < N, stepping = active && i != 0;
        const int Ll = (int)((N - c0 < 64) ? (N - c0 - 1) : 63);
        const In in = nxt;
        if (c0 + 64 < N) nxt = load(c0 + 64 + lane);
        const Vec3 p = in.p, z = in.z;
        const Vec3 p_pr{ prev_lane(c_po.x, p.x), prev_lane(c_po.y, p.y), prev_lane(c_po.z, p.z) };
        const double dx = stepping ? p.x - p_pr.x : 0.0, dy = stepping ? p.y - p_pr.y : 0.0, dz = stepping ? p.z - p_pr.z : 0.0;
        // helper 1 has to be done with this chunk (same SIMD, always resident: it never waits for this wave)
        const int need = (int)(c0 / 64) + 1;
        while ((int)*flag < need) __builtin_amdgcn_s_sleep(1);
        GSF_LDS_FENCE();
        const int64_t il = active ? i : N - 1;
        const double kxy = active ? L.Axy[il] : 0.0, kz = active ? L.Az[il] : 0.0;
        const double O0 = L.misc[M_O0], O1 = L.misc[M_O1], O2 = L.misc[M_O2];
        double al0 = 1.0 - kxy, al1 = 1.0 - kz;
        double b0 = kxy != 0.0 ? kxy * (z.x - O0) : 0.0, b1 = kxy != 0.0 ? kxy * (z.y - O1) : 0.0, b2 = kz != 0.0 ? kz * (z.z - O2) : 0.0;
        double w0 = al0 * dx, w1 = al0 * dy, w2 = al0 * dz, w3 = al1 * dx, w4 = al1 * dy, w5 = al1 * dz;
#define GSF_LSTAGE(CTRL, RM) {                                                                                                        \
        const double oa0 = dpp<CTRL, RM>(1.0, al0), oa1 = dpp<CTRL, RM>(1.0, al1);                                                    \
        const double ob0 = dpp0<CTRL, RM>(b0), ob1 = dpp0<CTRL, RM>(b1), ob2 = dpp0<CTRL, RM>(b2);                                    \
        const double ow0 = dpp0<CTRL, RM>(w0), ow1 = dpp0<CTRL, RM>(w1), ow2 = dpp0<CTRL, RM>(w2);                                    \
        const double ow3 = dpp0<CTRL, RM>(w3), ow4 = dpp0<CTRL, RM>(w4), ow5 = dpp0<CTRL, RM>(w5);                                    \
        b0 = al0 * ob0 + b0; b1 = al0 * ob1 + b1; w0 = al0 * ow0 + w0; w1 = al0 * ow1 + w1; w2 = al0 * ow2 + w2;                      \
        b2 = al1 * ob2 + b2; w3 = al1 * ow3 + w3; w4 = al1 * ow4 + w4; w5 = al1 * ow5 + w5;                                           \
        al0 = al0 * oa0; al1 = al1 * oa1; }
        GSF_SCAN_STAGES(GSF_LSTAGE)
#undef GSF_LSTAGE
        // carry of the previous poses: value_i = al_i * carry + local_i
        const double Axy = cA[0] * al0, Az = cA[1] * al1;
        const double Z0 = al0 * cZ[0] + b0, Z1 = al0 * cZ[1] + b1, Z2 = al1 * cZ[2] + b2;
        const double X0 = al0 * cWxy[0] + w0, X1 = al0 * cWxy[1] + w1, X2 = al0 * cWxy[2] + w2;
        const double Y0 = al1 * cWz[0] + w3, Y1 = al1 * cWz[1] + w4, Y2 = al1 * cWz[2] + w5;
        if (active) {
            L.Axy[i] = Axy; L.Az[i] = Az; L.Z[0][i] = Z0; L.Z[1][i] = Z1; L.Z[2][i] = Z2;
            L.Wxy[0][i] = X0; L.Wxy[1][i] = X1; L.Wxy[2][i] = X2; L.Wz[0][i] = Y0; L.Wz[1][i] = Y1; L.Wz[2][i] = Y2;
        }
        cA[0] = lane_bcast(Axy, Ll); cA[1] = lane_bcast(Az, Ll);
        cZ[0] = lane_bcast(Z0, Ll); cZ[1] = lane_bcast(Z1, Ll); cZ[2] = lane_bcast(Z2, Ll);
        cWxy[0] = lane_bcast(X0, Ll); cWxy[1] = lane_bcast(X1, Ll); cWxy[2] = lane_bcast(X2, Ll);
        cWz[0] = lane_bcast(Y0, Ll); cWz[1] = lane_bcast(Y1, Ll); cWz[2] = lane_bcast(Y2, Ll);
        c_po = lane_bcast(p, Ll);
        GSF_STAMP(26 + (int)(c0 / 64));
    }
}

// ---- the main wave after the fit: evaluate the linear form, apply the RTS corrections, write the rows (one pass: a pose of a smoothed
// run evaluates the two poses around its recovery itself)
__device__ __forceinline__ void lat_finish(const WaveArgs& a, const int64_t b, const int lane, const LatSlot& L, const Vec3& p0, const Quat& q0, const int32_t fit)
{
    const int64_t base = b * a.N, N = a.N;
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ quatb = a.quat + base * 4;
    double* __restrict__ pob = a.pos_out + base * 3;
    double* __restrict__ qob = a.quat_out + base * 4;
    // the SLAM quaternions of every chunk are requested up front (<= 5 chunks): one memory round trip for the whole pass
    constexpr int MAXC = LAT_MAX_N / 64;
    Quat qraw[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int64_t i = (int64_t)c * 64 + lane, il = i < N ? i : N - 1;
        qraw[c] = Quat{ quatb[il * 4], quatb[il * 4 + 1], quatb[il * 4 + 2], quatb[il * 4 + 3] };
    }
    const double O0 = L.misc[M_O0], O1 = L.misc[M_O1], O2 = L.misc[M_O2];
    const double x00 = p0.x - O0, x01 = p0.y - O1, x02 = p0.z - O2;
    const Quat cq0 = ekf_normalize(q0);                                   // ref :842, :683
    Quat r0; quat_unit(lane_bcast(qraw[0], 0), r0);
    const Quat Cq = quat_mul(cq0, quat_conj(r0));                         // q_i = Cq r_i (telescoped increments)
    const Mat3 M = quat_matrix(Cq);
    auto eval = [&](const int64_t k, double& x, double& y, double& z) __attribute__((always_inline)) {
        const double wx = L.Wxy[0][k], wy = L.Wxy[1][k], wz = L.Wxy[2][k], vx = L.Wz[0][k], vy = L.Wz[1][k], vz = L.Wz[2][k];
        const double Axy = L.Axy[k], Az = L.Az[k];
        x = Axy * x00 + L.Z[0][k] + (M.m00 * wx + M.m01 * wy + M.m02 * wz);
        y = Axy * x01 + L.Z[1][k] + (M.m10 * wx + M.m11 * wy + M.m12 * wz);
        z = Az * x02 + L.Z[2][k] + (M.m20 * vx + M.m21 * vy + M.m22 * vz);
    };
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int64_t i = (int64_t)c * 64 + lane;
        if ((int64_t)c * 64 >= N) break;                                  // wave-uniform
        const bool active = i < N;
        const int64_t il = active ? i : N - 1;
        double x, y, z;
        eval(il, x, y, z);
        const int32_t r = L.ridx[il];
        if (r >= 1) {
            // x_f[r] - x_p[r],  x_p[r] = x_{r-1} + R (p_r - p_{r-1})                                       (ref :795-797 telescoped)
            double xr, yr, zr, xq, yq, zq;
            eval(r, xr, yr, zr); eval(r - 1, xq, yq, zq);
            const double dx = posb[(int64_t)r * 3] - posb[(int64_t)(r - 1) * 3], dy = posb[(int64_t)r * 3 + 1] - posb[(int64_t)(r - 1) * 3 + 1],
                         dz = posb[(int64_t)r * 3 + 2] - posb[(int64_t)(r - 1) * 3 + 2];
            x += L.G[0][il] * (xr - (xq + (M.m00 * dx + M.m01 * dy + M.m02 * dz)));
            y += L.G[1][il] * (yr - (yq + (M.m10 * dx + M.m11 * dy + M.m12 * dz)));
            z += L.G[2][il] * (zr - (zq + (M.m20 * dx + M.m21 * dy + M.m22 * dz)));
        }
        Quat ri; quat_unit(qraw[c], ri);
        Quat qi = quat_mul(Cq, ri);
        const bool first = (i == 0);                                     // pose 0 keeps the initial state (ref :842)
        qi.x = first ? cq0.x : qi.x; qi.y = first ? cq0.y : qi.y; qi.z = first ? cq0.z : qi.z; qi.w = first ? cq0.w : qi.w;
        if (active) {
            __builtin_nontemporal_store(x + O0, &pob[i * 3]); __builtin_nontemporal_store(y + O1, &pob[i * 3 + 1]); __builtin_nontemporal_store(z + O2, &pob[i * 3 + 2]);
            __builtin_nontemporal_store(qi.x, &qob[i * 4]); __builtin_nontemporal_store(qi.y, &qob[i * 4 + 1]);
            __builtin_nontemporal_store(qi.z, &qob[i * 4 + 2]); __builtin_nontemporal_store(qi.w, &qob[i * 4 + 3]);
        }
    }
    if (lane == 0 && GSF_STATUS_PTR(a)) a.status[b] = (int32_t)L.misc[M_STATUS] | (fit << 8);
}

__global__ __launch_bounds__(768) void ekf_lat_kernel(WaveArgs a, EkfConfig cfg, int S)
{
    extern __shared__ double gsf_lat[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, slot = w & 3, role = w >> 2;      // role 0 main, 1 gains, 2 scans: all three on SIMD `slot`
    const int64_t b = (int64_t)blockIdx.x * 4 + slot;
    const LatSlot L = lat_slot(gsf_lat + (size_t)slot * lat_slot_doubles(S), S);
    if (lane == 0 && role == 1) L.misc[M_CHUNKS_DONE] = 0.0;            // (slots of idle triples too: harmless)
    __syncthreads();                                                     // the chunk counters are zero before helper 2 polls them
    if (b >= a.B) { __syncthreads(); return; }                           // idle triple of the last block
    if (role != 0) {
        if (role == 1) lat_helper_gains(a, cfg, b, lane, L); else lat_helper_scans(a, b, lane, L);
        __syncthreads();
        return;
    }
    const int64_t base = b * a.N, N = a.N;
    Vec3 p0; Quat q0; int32_t fit = 0;
    GSF_STAMP(0);
    const bool ok = wave_prelude<true>(a, b, base, N, lane, p0, q0, fit);  // the fit; on failure the rows are already NaN and the status written
    GSF_STAMP(6);
    __syncthreads();                                                     // both helper waves have finished the track
    GSF_STAMP(7);
    if (!ok) return;
    if (L.misc[M_GENERIC] != 0.0) {                                      // an invalid quaternion somewhere: the chunked filter finishes this track
        ChunkIn nxt = load_chunk(a.ts + base, a.pos + base * 3, a.quat + base * 4, a.gps + base * 3, a.valid + base, lane, N);
        wave_serial_chunks<true, false, true, 0>(a, cfg, b, lane, base, N, p0, q0, fit, nxt, nullptr, 0, 0, L.Axy);   // (the slot's LDS is free now: its ring)
        return;
    }
    lat_finish(a, b, lane, L, p0, q0, fit);
    GSF_STAMP(8);
}

}  // namespace

namespace gsf {

// x and y share the variance recursion (default CONFIG) and the track fits the LDS plan?
bool lat_kernel_applies(const gsf_ekf_config* c, int64_t N)
{
    return N >= 1 && N <= LAT_MAX_N && c->initial_cov_diag[1] == c->initial_cov_diag[0] && c->process_noise_diag[1] == c->process_noise_diag[0] &&
           c->meas_noise_diag[1] == c->meas_noise_diag[0];
}

int launch_ekf_lat(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps, const uint8_t* valid,
                   const gsf_ekf_config* cfg, int64_t B, int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status)
{
    WaveArgs a{ ts, pos, quat, gps, valid, nullptr, nullptr, R, t, s, pos_out, quat_out, status, B, N, nullptr };
    EkfConfig k;
    for (int i = 0; i < 7; ++i) { k.P0[i] = cfg->initial_cov_diag[i]; k.Qps[i] = cfg->process_noise_diag[i]; }
    for (int i = 0; i < 3; ++i) k.Rm[i] = cfg->meas_noise_diag[i];
    k.yaw_thr_rad = cfg->sharp_turn_yaw_rate_threshold_deg_per_sec * (M_PI / 180.0);
    k.sharp_turn_steps = cfg->default_ekf_transition_steps_on_sharp_turn;
    k._pad = 0;
    const int S = (int)(((N + 63) / 64) * 64);
    const size_t lds = lat_slot_doubles(S) * 4 * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) { GSF_HIP(hipFuncSetAttribute((const void*)ekf_lat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)); attr_set = true; }
    hipLaunchKernelGGL(ekf_lat_kernel, dim3((unsigned)((B + 3) / 4)), dim3(768), lds, ctx->stream, a, k, S);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace gsf
