// gsf_ekf_block.hip -- K4 / fused pipeline with ONE WAVE PER 64-POSE CHUNK and one workgroup per trajectory.
//
// Same scans as gsf_ekf_wave.hip, but the chunks of a trajectory run CONCURRENTLY: a 271-pose track is 5 waves instead of
// 5 dependent iterations of one wave, so a small batch (config C2: 1 000 tracks) puts 5 000 waves on the chip instead of
// 1 000 and hides the scan latencies behind each other.  The chunks exchange only their scan TOTALS through LDS:
//   barrier 1: ballots (availability / outage starts / sharp-turn pairs), Moebius totals, (pipeline) chunk moments
//   barrier 2: affine totals       barrier 3: first-recovery records (for RTS runs that end in a later chunk)
// and each wave rebuilds its carry-in by composing the <= 15 totals of its predecessors (wave-uniform work).
// Preconditions: N <= 1024 (<= 16 waves) -- longer tracks and huge batches take the serial-wave kernel.  A trajectory with an
// invalid quaternion (zero-motion branch, ref :84-86) cannot telescope its orientation chain and takes the generic path:
// one more scan (quaternion prefix product) and one more barrier, block-uniformly.
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

constexpr int MAXW = 16;

// the 3x3 Jacobi SVD + closed form as a real CALL: it runs once per wave, and inlining it would make its ~100 registers
// count against the whole kernel (2 waves/SIMD instead of 4)
__device__ __attribute__((noinline)) int32_t umeyama_finalize_call(const double* H, double ssq, const double* sc, const double* dc, double n,
                                                                   double* R, double* t, double* scale)
{
    double s; const int32_t f = umeyama_finalize(H, ssq, sc, dc, n, R, t, s);
    *scale = s;
    return f;
}

struct Shared {
    u64 a_mask[MAXW], start_mask[MAXW], f_mask[MAXW];
    double Tm[MAXW][3][4];            // Moebius totals (A,B,C,D) per axis
    double Ta[MAXW][3][2];            // affine totals (alpha, beta) per axis
    double mom[MAXW][20];             // pipeline: n, mean_a[3], mean_b[3], Caa, Cab[9] of the chunk's valid rows
    double rec_d[MAXW][3], rec_pm[MAXW][3];
    double Tq[MAXW][4];               // quaternion-increment totals (generic path only)
    int rec_lane[MAXW], rec_rts[MAXW];
    int any_bad, status;
};

template <bool PIPELINE, int MAXT>
__global__ __launch_bounds__(MAXT) void ekf_block_kernel(WaveArgs a, EkfConfig cfg)
{
    __shared__ Shared sh;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int64_t b = blockIdx.x, N = a.N;
    const double* __restrict__ tsb = a.ts + b * N;
    const double* __restrict__ posb = a.pos + b * N * 3;
    const double* __restrict__ quatb = a.quat + b * N * 4;
    const double* __restrict__ gpsb = a.gps + b * N * 3;
    const uint8_t* __restrict__ valb = a.valid + b * N;
    double* __restrict__ pob = a.pos_out + b * N * 3;
    double* __restrict__ qob = a.quat_out + b * N * 4;
    if (threadIdx.x == 0) { sh.any_bad = 0; sh.status = 0; }
    __syncthreads();

    // ------------------------------------------------------------------ phase 0: own chunk + the pose before it
    const int64_t c0 = (int64_t)w * 64, i = c0 + lane;
    const bool active = i < N, is_init = (i == 0), stepping = active && !is_init;
    const int L = (int)((N - c0 < 64) ? (N - c0 - 1) : 63);
    const ChunkIn in = load_chunk(tsb, posb, quatb, gpsb, valb, i, N);
    const int64_t ip = c0 > 0 ? c0 - 1 : 0;                              // predecessor of lane 0 (wave-uniform address)
    const double c_t = tsb[ip];
    const Vec3 c_po{ posb[ip * 3], posb[ip * 3 + 1], posb[ip * 3 + 2] };
    Quat c_r; const bool c_ok = quat_unit(Quat{ quatb[ip * 4], quatb[ip * 4 + 1], quatb[ip * 4 + 2], quatb[ip * 4 + 3] }, c_r);
    const double cz0 = gpsb[ip * 3], cz1 = gpsb[ip * 3 + 1], cz2 = gpsb[ip * 3 + 2];
    const bool c_vraw = valb[ip] != 0;
    // "gnss available" flag of pose c0-1: pose 0 keeps the raw mask (ref :848), every other pose is NaN-gated (ref :867-869)
    const bool c_prev_avail = (ip == 0) ? c_vraw : (c_vraw && !(isnan(cz0) || isnan(cz1) || isnan(cz2)));
    Quat r0; const bool r0ok = quat_unit(Quat{ quatb[0], quatb[1], quatb[2], quatb[3] }, r0);

    const double t = in.t;
    const Vec3 p = in.p; const Vec3 z = in.z;
    const bool vraw = in.v != 0;
    Quat r; const bool ok = quat_unit(in.q, r);
    const double t_pr = prev_lane(c_t, t);
    const Vec3 p_pr{ prev_lane(c_po.x, p.x), prev_lane(c_po.y, p.y), prev_lane(c_po.z, p.z) };
    const Quat r_pr = prev_lane(c_r, r);
    const u64 act_mask = __ballot(active);
    const u64 ok_mask = __ballot(ok);
    const bool ok_pr = (lane == 0) ? c_ok : (((ok_mask >> (lane - 1)) & 1ull) != 0ull);
    if (lane == 0 && (!c_ok || (ok_mask & act_mask) != act_mask)) atomicOr(&sh.any_bad, 1);
    const double dt = fmax(1e-6, t - t_pr);                              // ref :865
    const bool avail = stepping && vraw && !(isnan(z.x) || isnan(z.y) || isnan(z.z));
    const bool av = is_init ? vraw : avail;
    const u64 a_mask = __ballot(active && av);
    const bool ap = (lane == 0) ? (is_init ? true : c_prev_avail) : (((a_mask >> (lane - 1)) & 1ull) != 0ull);
    const bool starts = active && !av && ap;
    const bool recovers = stepping && av && !ap;
    const bool outpair = stepping && !av && !ap;
    const u64 start_mask = __ballot(starts), rec_mask = __ballot(recovers), pair_mask = __ballot(outpair);
    u64 f_mask = 0ull;
    if (pair_mask != 0ull) {
        bool f = false;
        if (outpair && t > t_pr) f = !(ok_pr && ok) || yaw_rate_exceeds(r_pr, r, t - t_pr, cfg.yaw_thr_rad);
        f_mask = __ballot(f);
    }
    if (lane == 0) { sh.a_mask[w] = a_mask; sh.start_mask[w] = start_mask; sh.f_mask[w] = f_mask; }

    // variance maps of the chunk (identity carry): prefix composition of Moebius maps per axis
    int same_axis[3] = { -1, -1, -1 };
    if (cfg.P0[1] == cfg.P0[0] && cfg.Qps[1] == cfg.Qps[0] && cfg.Rm[1] == cfg.Rm[0]) same_axis[1] = 0;
    if (cfg.P0[2] == cfg.P0[0] && cfg.Qps[2] == cfg.Qps[0] && cfg.Rm[2] == cfg.Rm[0]) same_axis[2] = 0;
    else if (cfg.P0[2] == cfg.P0[1] && cfg.Qps[2] == cfg.Qps[1] && cfg.Rm[2] == cfg.Rm[1]) same_axis[2] = 1;
    double MA[3], MB[3], MC[3], MD[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c > 0 && same_axis[c] >= 0) { const int o = same_axis[c]; MA[c] = MA[o]; MB[c] = MB[o]; MC[c] = MC[o]; MD[c] = MD[o]; }
        else {
            const double b0 = cfg.Qps[c] * dt, rr = cfg.Rm[c];
            double A = 1.0, Bm = stepping ? b0 : 0.0, Cm = 0.0, Dm = 1.0;
            if (avail) { A = rr; Bm = rr * b0; Cm = 1.0; Dm = b0 + rr; }
#define GSF_MSTAGE(CTRL, RM) {                                                                                              \
            const double oA = dpp<CTRL, RM>(1.0, A), oB = dpp<CTRL, RM>(0.0, Bm), oC = dpp<CTRL, RM>(0.0, Cm), oD = dpp<CTRL, RM>(1.0, Dm); \
            const double nA = A * oA + Bm * oC, nB = A * oB + Bm * oD, nC = Cm * oA + Dm * oC, nD = Cm * oB + Dm * oD;                  \
            A = nA; Bm = nB; Cm = nC; Dm = nD; }
            GSF_SCAN_STAGES(GSF_MSTAGE)
#undef GSF_MSTAGE
            MA[c] = A; MB[c] = Bm; MC[c] = Cm; MD[c] = Dm;
        }
        if (lane == 63) { sh.Tm[w][c][0] = MA[c]; sh.Tm[w][c][1] = MB[c]; sh.Tm[w][c][2] = MC[c]; sh.Tm[w][c][3] = MD[c]; }
    }
    if (PIPELINE) {
        // chunk moments of the rows with valid finite GNSS: count / means, then centred sums (data is in registers)
        const bool okf = active && vraw && !(isnan(z.x) || isnan(z.y) || isnan(z.z));
        const double n = wave_sum(okf ? 1.0 : 0.0);
        const double rn = n > 0.0 ? 1.0 / n : 0.0;
        const double ma0 = wave_sum(okf ? p.x : 0.0) * rn, ma1 = wave_sum(okf ? p.y : 0.0) * rn, ma2 = wave_sum(okf ? p.z : 0.0) * rn;
        const double mb0 = wave_sum(okf ? z.x : 0.0) * rn, mb1 = wave_sum(okf ? z.y : 0.0) * rn, mb2 = wave_sum(okf ? z.z : 0.0) * rn;
        const double a0 = okf ? p.x - ma0 : 0.0, a1 = okf ? p.y - ma1 : 0.0, a2 = okf ? p.z - ma2 : 0.0;
        const double b0 = okf ? z.x - mb0 : 0.0, b1 = okf ? z.y - mb1 : 0.0, b2 = okf ? z.z - mb2 : 0.0;
        const double caa = wave_sum(a0 * a0 + a1 * a1 + a2 * a2);
        const double cab[9] = { wave_sum(a0 * b0), wave_sum(a0 * b1), wave_sum(a0 * b2), wave_sum(a1 * b0), wave_sum(a1 * b1),
                                wave_sum(a1 * b2), wave_sum(a2 * b0), wave_sum(a2 * b1), wave_sum(a2 * b2) };
        if (lane == 0) {
            double* m = sh.mom[w];
            m[0] = n; m[1] = ma0; m[2] = ma1; m[3] = ma2; m[4] = mb0; m[5] = mb1; m[6] = mb2; m[7] = caa;
            for (int k = 0; k < 9; ++k) m[8 + k] = cab[k];
        }
    }
    __syncthreads();                                                     // ---- barrier 1

    const bool generic = sh.any_bad != 0;                                // block-uniform: some quaternion of the track is invalid

    // ------------------------------------------------------------------ phase 1: initial pose, carry-in variances, gains
    Vec3 p0; Quat q0; int32_t fit = 0;
    if (PIPELINE) {
        // merge the chunk moments pairwise (Chan et al.): exact, and well conditioned at UTM magnitudes
        double n = 0.0, ma[3] = { 0, 0, 0 }, mb[3] = { 0, 0, 0 }, caa = 0.0, cab[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int v = 0; v < W; ++v) {
            const double* m = sh.mom[v];
            const double n2 = m[0];
            if (n2 <= 0.0) continue;
            const double nn = n + n2, f2 = n2 / nn, g = n * n2 / nn;
            const double da[3] = { m[1] - ma[0], m[2] - ma[1], m[3] - ma[2] }, db[3] = { m[4] - mb[0], m[5] - mb[1], m[6] - mb[2] };
            caa += m[7] + g * (da[0] * da[0] + da[1] * da[1] + da[2] * da[2]);
#pragma unroll
            for (int k = 0; k < 9; ++k) cab[k] += m[8 + k] + g * da[k / 3] * db[k % 3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { ma[k] += da[k] * f2; mb[k] += db[k] * f2; }
            n = nn;
        }
        double Rb[9], tb[3], sb = NAN;
        fit = SIM3_NONE;
        if (n >= 3.0) fit = umeyama_finalize_call(cab, caa, ma, mb, n, Rb, tb, &sb);      // ref :430-451
        if (fit == SIM3_NONE || !r0ok) {                                 // no fit, or SciPy would raise on pose 0's quaternion (ref :466)
            if (active) {
                pob[i * 3] = NAN; pob[i * 3 + 1] = NAN; pob[i * 3 + 2] = NAN;
                qob[i * 4] = NAN; qob[i * 4 + 1] = NAN; qob[i * 4 + 2] = NAN; qob[i * 4 + 3] = NAN;
            }
            if (threadIdx.x == 0) {
                for (int k = 0; k < 9; ++k) a.R[b * 9 + k] = NAN;
                a.t[b * 3] = a.t[b * 3 + 1] = a.t[b * 3 + 2] = NAN; a.s[b] = NAN;
                if (a.status) a.status[b] = (fit == SIM3_NONE ? (SIM3_NONE << 8) : 0) | (r0ok ? 0 : ST_BAD_QUAT);
            }
            return;
        }
        if (threadIdx.x == 0) {
            for (int k = 0; k < 9; ++k) a.R[b * 9 + k] = Rb[k];
            a.t[b * 3] = tb[0]; a.t[b * 3 + 1] = tb[1]; a.t[b * 3 + 2] = tb[2]; a.s[b] = sb;
        }
        const double x = posb[0], y = posb[1], zz = posb[2];
        p0 = Vec3{ sb * (x * Rb[0] + y * Rb[1] + zz * Rb[2]) + tb[0], sb * (x * Rb[3] + y * Rb[4] + zz * Rb[5]) + tb[1],
                   sb * (x * Rb[6] + y * Rb[7] + zz * Rb[8]) + tb[2] };   // ref :464
        q0 = quat_mul(quat_from_matrix(Rb), r0);                         // ref :465-466
    } else {
        p0 = Vec3{ a.init_pos[b * 3], a.init_pos[b * 3 + 1], a.init_pos[b * 3 + 2] };
        q0 = Quat{ a.init_quat[b * 4], a.init_quat[b * 4 + 1], a.init_quat[b * 4 + 2], a.init_quat[b * 4 + 3] };
    }
    const Quat cq0 = ekf_normalize(q0);                                  // ref :842, :683
    int32_t status = 0;
    Quat qi; Vec3 u;
    if (!generic) {
        // orientation / predicted displacement: telescoped (all quaternions valid, see gsf_ekf_wave.hip)
        const Quat Cq = quat_mul(cq0, quat_conj(r0));
        qi = is_init ? cq0 : ekf_normalize(quat_mul(Cq, r));
        u = quat_rotate(Cq, Vec3{ p.x - p_pr.x, p.y - p_pr.y, p.z - p_pr.z });
        u.x = stepping ? u.x : 0.0; u.y = stepping ? u.y : 0.0; u.z = stepping ? u.z : 0.0;
    } else {
        // generic: calculate_relative_pose with the zero-motion branch (ref :77-92), prefix product of the increments,
        // chunk totals exchanged through LDS, carry-in orientation = q0 * T_0 * ... * T_{w-1}
        const bool move = stepping && ok_pr && ok;
        const Quat r1i = quat_conj(r_pr);
        Vec3 dpl = quat_rotate(r1i, Vec3{ p.x - p_pr.x, p.y - p_pr.y, p.z - p_pr.z });
        Quat D = quat_mul(r1i, r);
        dpl.x = move ? dpl.x : 0.0; dpl.y = move ? dpl.y : 0.0; dpl.z = move ? dpl.z : 0.0;
        D.x = move ? D.x : 0.0; D.y = move ? D.y : 0.0; D.z = move ? D.z : 0.0; D.w = move ? D.w : 1.0;
        if (__ballot(stepping && !(ok_pr && ok)) != 0ull) status |= ST_BAD_QUAT;
        const Quat QID{ 0.0, 0.0, 0.0, 1.0 };
#define GSF_QSTAGE(CTRL, RM) { const Quat o = dpp<CTRL, RM>(QID, D); D = quat_mul(o, D); }
        GSF_SCAN_STAGES(GSF_QSTAGE)
#undef GSF_QSTAGE
        if (lane == 63) { sh.Tq[w][0] = D.x; sh.Tq[w][1] = D.y; sh.Tq[w][2] = D.z; sh.Tq[w][3] = D.w; }
        __syncthreads();                                                 // ---- barrier 1b (generic path only, block-uniform)
        Quat cq = cq0;
        for (int v = 0; v < w; ++v) cq = quat_mul(cq, Quat{ sh.Tq[v][0], sh.Tq[v][1], sh.Tq[v][2], sh.Tq[v][3] });
        qi = ekf_normalize(quat_mul(cq, D));
        const Quat q_prev = prev_lane(ekf_normalize(cq), qi);
        u = quat_rotate(q_prev, dpl);
    }

    double Pf[3], Pm[3], kg[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c > 0 && same_axis[c] >= 0) { const int o = same_axis[c]; Pf[c] = Pf[o]; Pm[c] = Pm[o]; kg[c] = kg[o]; continue; }
        // carry-in variance: the predecessors' maps applied in order to P0
        double Pc = cfg.P0[c];
        for (int v = 0; v < w; ++v) {
            const double* T = sh.Tm[v][c];
            Pc = (T[0] * Pc + T[1]) * fast_rcp(T[2] * Pc + T[3]);
        }
        Pf[c] = (MA[c] * Pc + MB[c]) * fast_rcp(MC[c] * Pc + MD[c]);     // P_f[i]
        Pm[c] = prev_lane(Pc, Pf[c]) + cfg.Qps[c] * dt;                  // P_p[i]
        kg[c] = Pm[c] * fast_rcp(Pm[c] + cfg.Rm[c]);
    }
    // outage state carried into this chunk, rebuilt from the predecessors' ballots
    int64_t c_ostart = 0; bool c_seg_sharp = false;
    if (w > 0 && !c_prev_avail) {
        for (int v = w - 1; v >= 0; --v) {
            const u64 sm = sh.start_mask[v];
            if (sm != 0ull) {
                const int s = 63 - __clzll((long long)sm);
                c_ostart = (int64_t)v * 64 + s;
                c_seg_sharp = c_seg_sharp || (sh.f_mask[v] & bits(s + 1, 63)) != 0ull;
                break;
            }
            c_seg_sharp = c_seg_sharp || sh.f_mask[v] != 0ull;
        }
    }
    if (start_mask != 0ull) status |= ST_HAD_OUTAGE;
    bool sharp = false;
    if (recovers) {                                                      // ref :879-894
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

    // positions: affine maps in coordinates relative to the initial position p0 (x = p - p0; x_0 = 0)
    const double uu[3] = { u.x, u.y, u.z }, zl[3] = { z.x - p0.x, z.y - p0.y, z.z - p0.z };
    double al[3], be[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double kw = kg[c] * wgt;
        al[c] = avail ? (1.0 - kw) : 1.0;
        be[c] = avail ? ((1.0 - kw) * uu[c] + kw * zl[c]) : uu[c];
#define GSF_ASTAGE(CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al[c]), ob = dpp<CTRL, RM>(0.0, be[c]); be[c] = al[c] * ob + be[c]; al[c] = al[c] * oa; }
        GSF_SCAN_STAGES(GSF_ASTAGE)
#undef GSF_ASTAGE
        if (lane == 63) { sh.Ta[w][c][0] = al[c]; sh.Ta[w][c][1] = be[c]; }
    }
    __syncthreads();                                                     // ---- barrier 2

    // ------------------------------------------------------------------ phase 2: carry-in position, filtered positions
    double xl[3], dcorr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double xc = 0.0;
        for (int v = 0; v < w; ++v) xc = sh.Ta[v][c][0] * xc + sh.Ta[v][c][1];
        xl[c] = al[c] * xc + be[c];
        dcorr[c] = xl[c] - (prev_lane(xc, xl[c]) + uu[c]);               // x_f[i] - x_p[i]
    }
    {
        const int r1 = rec_mask != 0ull ? __ffsll((long long)rec_mask) - 1 : -1;
        const int rr = r1 >= 0 ? r1 : 0;
        const double d0 = lane_bcast(dcorr[0], rr), d1 = lane_bcast(dcorr[1], rr), d2 = lane_bcast(dcorr[2], rr);
        const double m0 = lane_bcast(Pm[0], rr), m1 = lane_bcast(Pm[1], rr), m2 = lane_bcast(Pm[2], rr);
        if (lane == 0) {
            sh.rec_lane[w] = r1; sh.rec_rts[w] = (r1 >= 0) ? (int)((rts_mask >> r1) & 1ull) : 0;
            sh.rec_d[w][0] = d0; sh.rec_d[w][1] = d1; sh.rec_d[w][2] = d2;
            sh.rec_pm[w][0] = m0; sh.rec_pm[w][1] = m1; sh.rec_pm[w][2] = m2;
        }
    }
    __syncthreads();                                                     // ---- barrier 3

    // ------------------------------------------------------------------ phase 3: per-outage RTS (ref :906-922), stores
    double xo[3] = { xl[0], xl[1], xl[2] };
    const bool open_end = ((a_mask >> L) & 1ull) == 0ull;                // the chunk ends inside an outage
    if (rts_mask != 0ull || (open_end && w + 1 < W)) {
        const u64 later = rec_mask & ~bits(0, lane);
        const int rl = later != 0ull ? __ffsll((long long)later) - 1 : 0;
        const bool in_local = active && !av && later != 0ull && (((rts_mask >> rl) & 1ull) != 0ull);
        // a run that is still open at the end of this chunk is closed by the first recovery of a later chunk
        int v2 = -1;
        if (open_end) for (int v = w + 1; v < W; ++v) if (sh.rec_lane[v] >= 0) { v2 = v; break; }
        const bool in_later = active && !av && later == 0ull && v2 >= 0 && sh.rec_rts[v2] != 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double dr_l = shidx(dcorr[c], rl), pr_l = shidx(Pm[c], rl);
            if (in_local) xo[c] = xl[c] + Pf[c] * fast_rcp(pr_l) * dr_l;
            else if (in_later) xo[c] = xl[c] + Pf[c] * fast_rcp(sh.rec_pm[v2][c]) * sh.rec_d[v2][c];
        }
        if (v2 >= 0 && sh.rec_rts[v2] != 0) status |= ST_RTS_APPLIED;
    }
    if (active) {
        pob[i * 3] = p0.x + xo[0]; pob[i * 3 + 1] = p0.y + xo[1]; pob[i * 3 + 2] = p0.z + xo[2];
        qob[i * 4] = qi.x; qob[i * 4 + 1] = qi.y; qob[i * 4 + 2] = qi.z; qob[i * 4 + 3] = qi.w;
    }
    if (w == W - 1 && open_end) status |= ST_ENDED_IN_OUTAGE;            // ref :932
    if (lane == 0 && status) atomicOr(&sh.status, status);
    __syncthreads();
    if (threadIdx.x == 0 && a.status) a.status[b] = sh.status | (PIPELINE ? (fit << 8) : 0);
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

int launch_ekf_block(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                     const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                     int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(B <= 0x7fffffff && N >= 1 && N <= 64 * MAXW, "launch_ekf_block: needs N <= 1024");
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, nullptr };
    const EkfConfig k = to_core(cfg);
    const int W = (int)((N + 63) / 64);
    const dim3 grid((unsigned)B), block((unsigned)(W * 64));
#define GSF_LAUNCH_BLOCK(P, T) hipLaunchKernelGGL((ekf_block_kernel<P, T>), grid, block, 0, ctx->stream, a, k)
    // pipeline: the 128-register build (4 waves/SIMD) also for short tracks unless ekf_variant 6 asks for the 256-register one
    if (pipeline) { if (W <= 8 && ctx->ekf_variant == 6) GSF_LAUNCH_BLOCK(true, 512); else GSF_LAUNCH_BLOCK(true, 1024); }
    else { if (W <= 8) GSF_LAUNCH_BLOCK(false, 512); else GSF_LAUNCH_BLOCK(false, 1024); }
#undef GSF_LAUNCH_BLOCK
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace gsf
