// gsf_ekf_seg.hip -- K4 (and the fused K2+K3+K4 pipeline) for SHORT tracks: the whole trajectory in ONE pass of one wave.
//
// A track of N <= 64*P poses is cut into 64 segments of P consecutive poses, one per lane (lane l owns poses l*P .. l*P+P-1, all
// of them in registers).  apply_ekf_correction (EKFGPSSLAM.py:831-935) then runs as
//   pass A  (in-lane, serial over the P poses)  availability / outage flags, variance maps composed into one Moebius map per lane
//   scan 1  (DPP, once per TRACK)               carry-in variance of every lane
//   pass B  (in-lane)                           the reference's own variance recursion, gains, affine position maps -> lane total
//   scan 2  (DPP, once per track)               carry-in position of every lane
//   pass C  (in-lane)                           positions, orientations, per-outage RTS correction, stores
// i.e. the same mathematics as gsf_ekf_wave.hip (see there for the scan formulation), but the six-stage scans are paid once per
// track instead of once per 64 poses and there is no chunk-to-chunk carry: a 271-pose track costs ~1/3 of the instructions.
// The outage bookkeeping (start / recovery / sharp-turn decision, ref :861-894) is a little state machine that every lane runs
// over its own poses; its state at the first pose of a lane comes from three ballots and one ds_bpermute.
// Used for 64 < N <= 64*SEG_MAX_P with equal-length batches; a track with an invalid quaternion (the reference's zero-motion
// branch, ref :84-86) cannot telescope its orientation chain and is handed to the generic chunked body, wave-uniformly.
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

template <bool PIPELINE, int P>
__global__ __launch_bounds__(64) void ekf_seg_kernel(WaveArgs a, EkfConfig cfg)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, N = a.N, base = b * N;
    const double* __restrict__ tsb = a.ts + base;
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ quatb = a.quat + base * 4;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;
    double* __restrict__ pob = a.pos_out + base * 3;
    double* __restrict__ qob = a.quat_out + base * 4;

    // ------------------------------------------------------------------ rows: global -> LDS coalesced, LDS -> the lane's P poses
    // (a lane reading its own P consecutive rows straight from memory costs 64 separate cache-line requests per load instruction:
    // measured 2x slower than the chunked kernel at 1 000 tracks; through LDS every global access is one contiguous 1-KB run)
    constexpr int ROWS = 64 * P;
    __shared__ __attribute__((aligned(16))) double sh[ROWS * 11];
    __shared__ uint8_t shv[ROWS];
    double* const sh_ts = sh; double* const sh_pos = sh + ROWS; double* const sh_quat = sh + ROWS * 4; double* const sh_gps = sh + ROWS * 8;
    {
        auto stage = [&](const double* __restrict__ g, double* sdst, const int total) __attribute__((always_inline)) {
            for (int k = 2 * lane; k < total; k += 128) {
                const double v0 = __builtin_nontemporal_load(&g[k]);
                const double v1 = (k + 1 < total) ? __builtin_nontemporal_load(&g[k + 1]) : 0.0;
                sdst[k] = v0; if (k + 1 < total) sdst[k + 1] = v1;
            }
        };
        stage(tsb, sh_ts, (int)N); stage(posb, sh_pos, (int)N * 3); stage(quatb, sh_quat, (int)N * 4); stage(gpsb, sh_gps, (int)N * 3);
        for (int k = lane; k < (int)N; k += 64) shv[k] = valb[k];
    }
    __syncthreads();
    ChunkIn in[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int i = lane * P + j, il = i < (int)N ? i : (int)N - 1;     // idle lanes re-read the last pose
        in[j].t = sh_ts[il];
        in[j].p = Vec3{ sh_pos[il * 3], sh_pos[il * 3 + 1], sh_pos[il * 3 + 2] };
        in[j].q = Quat{ sh_quat[il * 4], sh_quat[il * 4 + 1], sh_quat[il * 4 + 2], sh_quat[il * 4 + 3] };
        in[j].z = Vec3{ sh_gps[il * 3], sh_gps[il * 3 + 1], sh_gps[il * 3 + 2] };
        in[j].v = shv[il];
    }
    bool active[P], is_init[P], stepping[P];
    Quat r[P];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int64_t i = (int64_t)lane * P + j;
        active[j] = i < N; is_init[j] = (i == 0); stepping[j] = active[j] && !is_init[j];
        const bool okq = quat_unit(in[j].q, r[j]);                       // Rotation.from_quat, ref :80-81
        bad = bad || (active[j] && !okq);
    }
    if (__ballot(bad) != 0ull) { wave_serial_body<PIPELINE>(a, cfg, b, lane); return; }   // generic path (block-uniform: one wave)

    // ------------------------------------------------------------------ initial pose (ref :1002-1006 fused, or the caller's)
    Vec3 p0; Quat q0; int32_t fit = 0;
    if (PIPELINE) {
        // K2 on the rows with valid finite GNSS: shifted raw moments straight from the registers (see wave_prelude)
        const Vec3 as = lane_bcast(in[0].p, 0);
        bool okz[P];
        double m0 = 0.0, m1 = 0.0, m2 = 0.0; bool mine = false;           // this lane's first usable fix
#pragma unroll
        for (int j = P - 1; j >= 0; --j) {
            okz[j] = active[j] && in[j].v != 0 && !(isnan(in[j].z.x) || isnan(in[j].z.y) || isnan(in[j].z.z));
            m0 = okz[j] ? in[j].z.x : m0; m1 = okz[j] ? in[j].z.y : m1; m2 = okz[j] ? in[j].z.z : m2; mine = mine || okz[j];
        }
        const u64 mm = __ballot(mine);
        const int f = mm != 0ull ? __ffsll((long long)mm) - 1 : 0;
        const double bs0 = lane_bcast(m0, f), bs1 = lane_bcast(m1, f), bs2 = lane_bcast(m2, f);
        double sums[17];
#pragma unroll
        for (int k = 0; k < 17; ++k) sums[k] = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const bool o = okz[j];
            const double a0 = o ? in[j].p.x - as.x : 0.0, a1 = o ? in[j].p.y - as.y : 0.0, a2 = o ? in[j].p.z - as.z : 0.0;
            const double b0 = o ? in[j].z.x - bs0 : 0.0, b1 = o ? in[j].z.y - bs1 : 0.0, b2 = o ? in[j].z.z - bs2 : 0.0;
            sums[0] += o ? 1.0 : 0.0; sums[1] += a0; sums[2] += a1; sums[3] += a2; sums[4] += b0; sums[5] += b1; sums[6] += b2;
            sums[7] += a0 * a0 + a1 * a1 + a2 * a2;
            sums[8] += a0 * b0; sums[9] += a0 * b1; sums[10] += a0 * b2;
            sums[11] += a1 * b0; sums[12] += a1 * b1; sums[13] += a1 * b2;
            sums[14] += a2 * b0; sums[15] += a2 * b1; sums[16] += a2 * b2;
        }
        const double as_[3] = { as.x, as.y, as.z }, bs_[3] = { bs0, bs1, bs2 };
        if (!fit_from_partials(a, b, base, N, lane, sums, as_, bs_, lane_bcast(in[0].q, 0), p0, q0, fit)) return;
    } else {
        p0 = Vec3{ a.init_pos[b * 3], a.init_pos[b * 3 + 1], a.init_pos[b * 3 + 2] };
        q0 = Quat{ a.init_quat[b * 4], a.init_quat[b * 4 + 1], a.init_quat[b * 4 + 2], a.init_quat[b * 4 + 3] };
    }
    const Quat cq = ekf_normalize(q0);                                   // ref :842, :683
    // orientation: q_i = q_{i-1} dq_i with dq_i = r_{i-1}^-1 r_i telescopes to (cq r_0^-1) r_i; the predicted displacement
    // R(q_{i-1}) R(r_{i-1})^-1 (p_i - p_{i-1}) to one rotation by the wave-uniform Cq (ref :77-92, :707-709)
    const Quat Cq = quat_mul(cq, quat_conj(lane_bcast(r[0], 0)));

    // ------------------------------------------------------------------ pass A: time steps, GNSS gate, outage flags
    // previous pose of sub-pose 0 = last sub-pose of the previous lane (lane 0: pose 0 itself, ref :858)
    double t_pr[P]; Vec3 p_pr[P];
    t_pr[0] = prev_lane(lane_bcast(in[0].t, 0), in[P - 1].t);
    { const Vec3 c = lane_bcast(in[0].p, 0);
      p_pr[0] = Vec3{ prev_lane(c.x, in[P - 1].p.x), prev_lane(c.y, in[P - 1].p.y), prev_lane(c.z, in[P - 1].p.z) }; }
#pragma unroll
    for (int j = 1; j < P; ++j) { t_pr[j] = in[j - 1].t; p_pr[j] = in[j - 1].p; }
    double dt[P];
    bool avail[P], av[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        dt[j] = fmax(1e-6, in[j].t - t_pr[j]);                           // ref :865
        avail[j] = stepping[j] && in[j].v != 0 && !(isnan(in[j].z.x) || isnan(in[j].z.y) || isnan(in[j].z.z));   // ref :867-869
        av[j] = is_init[j] ? (in[j].v != 0) : avail[j];                  // pose 0: the raw mask (ref :848)
    }
    const u64 last_av_m = __ballot(active[P - 1] && av[P - 1]);
    bool ap[P];                                                          // "GNSS available" flag of the previous pose
    ap[0] = (lane == 0) ? true : (((last_av_m >> (lane - 1)) & 1ull) != 0ull);
#pragma unroll
    for (int j = 1; j < P; ++j) ap[j] = av[j - 1];
    bool starts[P], recovers[P], outpair[P];
    bool any_out_l = false, any_start_l = false, any_rec_l = false;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        starts[j] = active[j] && !av[j] && ap[j];                        // ref :875-877 (pose 0: :861)
        recovers[j] = stepping[j] && av[j] && !ap[j];                    // ref :879
        outpair[j] = stepping[j] && !av[j] && !ap[j];                    // poses i-1 and i both inside the outage
        any_out_l = any_out_l || (active[j] && !av[j]); any_start_l = any_start_l || starts[j]; any_rec_l = any_rec_l || recovers[j];
    }
    const u64 start_lanes = __ballot(any_start_l), rec_lanes = __ballot(any_rec_l);
    int32_t status = start_lanes != 0ull ? ST_HAD_OUTAGE : 0;

    // recovery decisions (ref :879-894): sharp[j] only where recovers[j]
    bool sharp[P];
#pragma unroll
    for (int j = 0; j < P; ++j) sharp[j] = false;
    if (rec_lanes != 0ull) {                                             // wave-uniform: some outage ends inside the track
        // is_sharp_turn_in_segment pairs (ref :808-826): pair (i-1, i) of an outage exceeds the yaw-rate threshold
        bool f[P];
        bool f_any = false;
#pragma unroll
        for (int j = 0; j < P; ++j) f[j] = false;
        bool any_pair_l = false;
#pragma unroll
        for (int j = 0; j < P; ++j) any_pair_l = any_pair_l || outpair[j];
        if (__ballot(any_pair_l) != 0ull) {
            const Quat rp0 = prev_lane(lane_bcast(r[0], 0), r[P - 1]);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const Quat rp = (j == 0) ? rp0 : r[j > 0 ? j - 1 : 0];
                if (outpair[j] && in[j].t > t_pr[j]) f[j] = yaw_rate_exceeds(rp, r[j], in[j].t - t_pr[j], cfg.yaw_thr_rad);
                f_any = f_any || f[j];
            }
        }
        // state at the lane's first pose: index of the open outage's first pose, "sharp pair seen since then"
        int last_start = -1; bool f_after = false;                       // within this lane: last start, sharp pairs after it
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if (starts[j]) { last_start = j; f_after = false; }
            f_after = f_after || f[j];
        }
        const u64 f_after_m = __ballot(f_after), f_any_m = __ballot(f_any);
        int ostart = 0; bool seg_sharp = false;                          // ref :861-862
        {   // a lane that begins inside an outage: the outage started in lane m, the last lane before it that holds a start
            // (an outage always begins with a start).  The ds_bpermute runs with every lane enabled (a disabled source reads 0).
            const u64 below = start_lanes & bits(0, lane - 1);
            const int m = below != 0ull ? 63 - __clzll((long long)below) : 0;
            const int ls = __shfl(last_start, m, 64);
            if (!ap[0] && lane > 0) {
                ostart = m * P + ls;
                seg_sharp = (((f_after_m >> m) & 1ull) != 0ull) || ((f_any_m & bits(m + 1, lane - 1)) != 0ull);
            }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int i = lane * P + j;
            if (starts[j]) { ostart = i; seg_sharp = false; }
            seg_sharp = seg_sharp || f[j];
            sharp[j] = recovers[j] && (i - ostart >= 2) && seg_sharp;
        }
    }
    bool any_sharp_l = false, any_rts_l = false;
#pragma unroll
    for (int j = 0; j < P; ++j) { any_sharp_l = any_sharp_l || sharp[j]; any_rts_l = any_rts_l || (recovers[j] && !sharp[j]); }
    const bool any_rts = __ballot(any_rts_l) != 0ull;
    if (__ballot(any_sharp_l) != 0ull) status |= ST_SHARP_TURN;
    if (any_rts) status |= ST_RTS_APPLIED;

    // ------------------------------------------------------------------ variances (ref :712-713, :723-731)
    // axes with identical (P0, Q, R) have identical recursions (default CONFIG: x == y)
    const bool y_is_x = cfg.P0[1] == cfg.P0[0] && cfg.Qps[1] == cfg.Qps[0] && cfg.Rm[1] == cfg.Rm[0];
    const bool z_is_x = cfg.P0[2] == cfg.P0[0] && cfg.Qps[2] == cfg.Qps[0] && cfg.Rm[2] == cfg.Rm[0];
    const bool z_is_y = !z_is_x && cfg.P0[2] == cfg.P0[1] && cfg.Qps[2] == cfg.Qps[1] && cfg.Rm[2] == cfg.Rm[1];
    double kg[P][3], Pm[P][3];                                           // Kalman gain if the fix is used, predicted variance
    auto variance_axis = [&](const int c) __attribute__((always_inline)) {
        const double q = cfg.Qps[c], rr = cfg.Rm[c], Pinit = cfg.P0[c];
        // lane total of the Moebius maps P -> (A P + B)/(C P + D): predict P + b, then (fix used) r (P+b) / ((P+b) + r)
        double A = 1.0, Bm = 0.0, Cm = 0.0, Dm = 1.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double b0 = stepping[j] ? q * dt[j] : 0.0;
            const double A1 = A + b0 * Cm, B1 = Bm + b0 * Dm;
            const double uA = rr * A1, uB = rr * B1, uC = A1 + rr * Cm, uD = B1 + rr * Dm;
            A = avail[j] ? uA : A1; Bm = avail[j] ? uB : B1; Cm = avail[j] ? uC : Cm; Dm = avail[j] ? uD : Dm;
        }
        // the maps are projective: normalise the lane total to D = 1 (D > 0), so that 64 of them can be multiplied without the
        // r^n factors of a long run of fixes underflowing (A -> 0 is benign: the initial variance is simply forgotten)
        { const double iD = fast_rcp(Dm); A *= iD; Bm *= iD; Cm *= iD; Dm = 1.0; }
#define GSF_MSTAGE(CTRL, RM) {                                                                                              \
        const double oA = dpp<CTRL, RM>(1.0, A), oB = dpp0<CTRL, RM>(Bm), oC = dpp0<CTRL, RM>(Cm), oD = dpp<CTRL, RM>(1.0, Dm); \
        const double nA = A * oA + Bm * oC, nB = A * oB + Bm * oD, nC = Cm * oA + Dm * oC, nD = Cm * oB + Dm * oD;            \
        A = nA; Bm = nB; Cm = nC; Dm = nD; }
        GSF_SCAN_STAGES(GSF_MSTAGE)
#undef GSF_MSTAGE
        const double Plast = (A * Pinit + Bm) * fast_rcp(Cm * Pinit + Dm);   // P_f after the lane's last pose
        double Pc = prev_lane(Pinit, Plast);                             // ... and before its first
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double Pp = Pc + (stepping[j] ? q * dt[j] : 0.0);      // P_p
            const double k = Pp * fast_rcp(Pp + rr);
            Pm[j][c] = Pp; kg[j][c] = k;
            Pc = avail[j] ? rr * k : Pp;                                 // r P/(P+r) == (1-k)^2 P + k^2 r (Joseph form, diagonal case)
        }
    };
    variance_axis(0);
    if (y_is_x) {
#pragma unroll
        for (int j = 0; j < P; ++j) { Pm[j][1] = Pm[j][0]; kg[j][1] = kg[j][0]; }
    } else variance_axis(1);
    if (z_is_x) {
#pragma unroll
        for (int j = 0; j < P; ++j) { Pm[j][2] = Pm[j][0]; kg[j][2] = kg[j][0]; }
    } else if (z_is_y) {
#pragma unroll
        for (int j = 0; j < P; ++j) { Pm[j][2] = Pm[j][1]; kg[j][2] = kg[j][1]; }
    } else variance_axis(2);

    // ------------------------------------------------------------------ positions (ref :707, :728, :762-763): affine maps
    // x -> ea x + eb in coordinates relative to the initial position (x = p - p0, x before pose 0 = 0)
    double ea[P][3], eb[P][3], uu[P][3];
    double al[3] = { 1.0, 1.0, 1.0 }, be[3] = { 0.0, 0.0, 0.0 };
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const Vec3 d = quat_rotate(Cq, Vec3{ in[j].p.x - p_pr[j].x, in[j].p.y - p_pr[j].y, in[j].p.z - p_pr[j].z });
        uu[j][0] = stepping[j] ? d.x : 0.0; uu[j][1] = stepping[j] ? d.y : 0.0; uu[j][2] = stepping[j] ? d.z : 0.0;
        // one-step blend weight on a sharp-turn recovery (ref :752-768, Q7): 1/eff if eff > 1, else a hard update
        const double wgt = (sharp[j] && cfg.sharp_turn_steps > 1) ? 1.0 / (double)cfg.sharp_turn_steps : 1.0;
        const double zl[3] = { in[j].z.x - p0.x, in[j].z.y - p0.y, in[j].z.z - p0.z };
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double kw = kg[j][c] * wgt;
            ea[j][c] = avail[j] ? (1.0 - kw) : 1.0;
            eb[j][c] = avail[j] ? ((1.0 - kw) * uu[j][c] + kw * zl[c]) : uu[j][c];
            be[c] = ea[j][c] * be[c] + eb[j][c]; al[c] = ea[j][c] * al[c];
        }
    }
#define GSF_ASTAGE(c, CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al[c]), ob = dpp0<CTRL, RM>(be[c]); be[c] = al[c] * ob + be[c]; al[c] = al[c] * oa; }
#define GSF_ASTAGE_X(CTRL, RM) GSF_ASTAGE(0, CTRL, RM)
#define GSF_ASTAGE_Y(CTRL, RM) GSF_ASTAGE(1, CTRL, RM)
#define GSF_ASTAGE_Z(CTRL, RM) GSF_ASTAGE(2, CTRL, RM)
#define GSF_ASTAGE_XY(CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al[0]), ob0 = dpp0<CTRL, RM>(be[0]), ob1 = dpp0<CTRL, RM>(be[1]); \
                                  be[0] = al[0] * ob0 + be[0]; be[1] = al[0] * ob1 + be[1]; al[0] = al[0] * oa; }
    if (y_is_x) { GSF_SCAN_STAGES(GSF_ASTAGE_XY) }                       // x and y share the gain, hence the multiplicative part
    else { GSF_SCAN_STAGES(GSF_ASTAGE_X) GSF_SCAN_STAGES(GSF_ASTAGE_Y) }
    GSF_SCAN_STAGES(GSF_ASTAGE_Z)
#undef GSF_ASTAGE_XY
#undef GSF_ASTAGE_Z
#undef GSF_ASTAGE_Y
#undef GSF_ASTAGE_X
#undef GSF_ASTAGE
    double xl[P][3], xo[P][3];
    {
        double x[3] = { prev_lane(0.0, be[0]), prev_lane(0.0, be[1]), prev_lane(0.0, be[2]) };   // x after the previous lane's last pose
#pragma unroll
        for (int j = 0; j < P; ++j) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { x[c] = ea[j][c] * x[c] + eb[j][c]; xl[j][c] = x[c]; xo[j][c] = x[c]; }
        }
    }

    // ------------------------------------------------------------------ per-outage RTS (ref :906-922, :777-803).  Inside an outage
    // x_f = x_p and P_f = P_p, so the gain product telescopes: x_s[k] = x_f[k] + (P_p[k] / P_p[r]) (x_f[r] - x_p[r]) for the poses
    // k of the run, r = the recovery pose that closes it (only if that recovery takes the RTS branch, i.e. is not "sharp").
    if (any_rts) {
        // this lane's FIRST recovery, for the lanes before it: innovation-side correction, 1 / P_p, "takes the RTS branch"
        double fd[3] = { 0.0, 0.0, 0.0 }, fi[3] = { 0.0, 0.0, 0.0 }; int frts = 0;
        const double xprev0[3] = { prev_lane(0.0, be[0]), prev_lane(0.0, be[1]), prev_lane(0.0, be[2]) };
#pragma unroll
        for (int j = P - 1; j >= 0; --j) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double xp = (j == 0) ? xprev0[c] : xl[j > 0 ? j - 1 : 0][c];
                const double dcorr = xl[j][c] - (xp + uu[j][c]);         // x_f[r] - x_p[r]
                fd[c] = recovers[j] ? dcorr : fd[c];
                fi[c] = recovers[j] ? fast_rcp(Pm[j][c]) : fi[c];
            }
            frts = recovers[j] ? (sharp[j] ? 0 : 1) : frts;
        }
        // the first recovery AFTER this lane
        const u64 later = (lane < 63) ? (rec_lanes & ~bits(0, lane)) : 0ull;
        const int m2 = later != 0ull ? __ffsll((long long)later) - 1 : lane;
        double cd[3], ci[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { cd[c] = shidx(fd[c], m2); ci[c] = shidx(fi[c], m2); }
        bool crts = (__shfl(frts, m2, 64) != 0) && later != 0ull;
        // walk the lane's poses backwards; a recovery replaces the "closing recovery" for the poses before it
#pragma unroll
        for (int j = P - 1; j >= 0; --j) {
            if (recovers[j]) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double xp = (j == 0) ? xprev0[c] : xl[j > 0 ? j - 1 : 0][c];
                    cd[c] = xl[j][c] - (xp + uu[j][c]); ci[c] = fast_rcp(Pm[j][c]);
                }
                crts = !sharp[j];
            } else if (active[j] && !av[j] && crts) {
#pragma unroll
                for (int c = 0; c < 3; ++c) xo[j][c] = xl[j][c] + Pm[j][c] * ci[c] * cd[c];
            }
        }
    }

    // ------------------------------------------------------------------ outputs: registers -> LDS (the input area is free now) ->
    // coalesced streaming stores (the rows are not read again)
    __syncthreads();
    double* const so_pos = sh; double* const so_quat = sh + ROWS * 3;
    bool ended_out_l = false;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int i = lane * P + j;
        if (active[j]) {
            const Quat qi = is_init[j] ? cq : ekf_normalize(quat_mul(Cq, r[j]));
            so_pos[i * 3] = p0.x + xo[j][0]; so_pos[i * 3 + 1] = p0.y + xo[j][1]; so_pos[i * 3 + 2] = p0.z + xo[j][2];
            so_quat[i * 4] = qi.x; so_quat[i * 4 + 1] = qi.y; so_quat[i * 4 + 2] = qi.z; so_quat[i * 4 + 3] = qi.w;
        }
        ended_out_l = ended_out_l || (i == (int)N - 1 && !av[j]);        // ref :932
    }
    __syncthreads();
    {
        auto drain = [&](const double* ssrc, double* __restrict__ g, const int total) __attribute__((always_inline)) {
            for (int k = 2 * lane; k < total; k += 128) {
                __builtin_nontemporal_store(ssrc[k], &g[k]);
                if (k + 1 < total) __builtin_nontemporal_store(ssrc[k + 1], &g[k + 1]);
            }
        };
        drain(so_pos, pob, (int)N * 3); drain(so_quat, qob, (int)N * 4);
    }
    if (__ballot(ended_out_l) != 0ull) status |= ST_ENDED_IN_OUTAGE;
    if (lane == 0 && a.status) a.status[b] = status | (PIPELINE ? (fit << 8) : 0);
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

// single-shot launches for equal-length batches with 64 < N <= 64 * SEG_MAX_P (called from launch_ekf_wave)
int launch_ekf_seg(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                   const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                   int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(B <= 0x7fffffff && N >= 1 && N <= 64 * SEG_MAX_P, "launch_ekf_seg: needs N <= 64 * SEG_MAX_P");
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, nullptr };
    const EkfConfig k = to_core(cfg);
    const int p = (int)((N + 63) / 64);
#define GSF_LAUNCH_SEG(PP) do { if (pipeline) hipLaunchKernelGGL((ekf_seg_kernel<true, PP>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k); \
                                else hipLaunchKernelGGL((ekf_seg_kernel<false, PP>), dim3((unsigned)B), dim3(64), 0, ctx->stream, a, k); } while (0)
    switch (p) {
    case 1: GSF_LAUNCH_SEG(1); break;
    case 2: GSF_LAUNCH_SEG(2); break;
    case 3: GSF_LAUNCH_SEG(3); break;
    case 4: GSF_LAUNCH_SEG(4); break;
    default: GSF_LAUNCH_SEG(5); break;
    }
#undef GSF_LAUNCH_SEG
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace gsf
