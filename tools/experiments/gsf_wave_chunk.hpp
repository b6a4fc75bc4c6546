// gsf_wave_chunk.hpp -- one iteration of the wave-per-trajectory K4: a wave takes 64*PPL consecutive poses, PPL per lane.
//
// PPL = 1 is the plain form (lane = pose).  PPL = 2 halves the cost of everything that is paid per WAVE-INSTRUCTION and not
// per pose: each lane first composes its own two scan elements, the six DPP stages then run once per 128 poses, and the
// wave-uniform work (carries, ballots, outage bookkeeping) is amortised over twice the poses.  Per-lane pose pairs are also
// contiguous in memory (48/64-byte runs per lane).  Positions inside a chunk are numbered pos = lane*PPL + j; every ballot
// is kept per sub-pose j (bit = lane), and the helpers below translate position ranges into lane ranges.
#pragma once
#include "gsf_wave_common.hpp"

namespace {

struct TrajPtrs {
    const double* __restrict__ ts; const double* __restrict__ pos; const double* __restrict__ quat; const double* __restrict__ gps;
    const uint8_t* __restrict__ valid;
    double* __restrict__ pos_out; double* __restrict__ quat_out;
    int64_t N;
};

// wave-uniform state carried from one chunk to the next (replicated in every lane / held in SGPRs)
struct WaveCarry {
    Quat q; Vec3 p; double P[3];            // filter state after the last pose of the previous chunk (ref :770)
    bool prev_avail; int64_t ostart; bool seg_sharp; double Pos[3];   // open outage: first pose, sharp flag so far, P_f at its start
    Vec3 po; Quat r; bool ok; double t;     // previous ORIGINAL pose (ref :858, :930)
    int32_t status;
    int same_axis[3];
};

// One ballot per sub-pose.  Scalar members with value-level selects on purpose: an array member read through a select chain
// is folded by LLVM into a dynamically indexed load, i.e. a table in scratch (seen with four poses per lane).
template <int PPL>
struct Masks {
    u64 m0, m1, m2, m3, m4;
    // always-inlined constructor: an out-of-line implicit one would keep the object's address alive through the early
    // optimisation passes, and the select chain of get() is then folded into a dynamically indexed load from scratch
    __device__ __forceinline__ Masks() : m0(0ull), m1(0ull), m2(0ull), m3(0ull), m4(0ull) {}
    // pure arithmetic (no select between member loads: hipcc optimises every helper stand-alone BEFORE inlining, and would fold
    // such a select into one dynamically indexed load -- the object then stays in scratch for good)
    __device__ __forceinline__ u64 get(int j) const
    {
        return (m0 & (0ull - (u64)(j == 0))) | (m1 & (0ull - (u64)(j == 1))) | (m2 & (0ull - (u64)(j == 2))) | (m3 & (0ull - (u64)(j == 3))) |
               (m4 & (0ull - (u64)(j == 4)));
    }
    __device__ __forceinline__ void set(int j, u64 v) { m0 = (j == 0) ? v : m0; m1 = (j == 1) ? v : m1; m2 = (j == 2) ? v : m2; m3 = (j == 3) ? v : m3; m4 = (j == 4) ? v : m4; }
};

template <int PPL> __device__ __forceinline__ bool any(const Masks<PPL>& a) { u64 o = 0; for (int j = 0; j < PPL; ++j) o |= a.get(j); return o != 0ull; }

// highest position < pos whose bit is set, or -1
template <int PPL>
__device__ __forceinline__ int last_before(const Masks<PPL>& a, int pos)
{
    int best = -1;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int d = pos - k - 1;                           // positions lane*PPL + k <= pos-1  <=>  lane <= d / PPL
        if (d < 0) continue;
        const u64 m = a.get(k) & bits(0, d / PPL);
        if (m != 0ull) { const int c = (63 - __clzll((long long)m)) * PPL + k; best = c > best ? c : best; }
    }
    return best;
}
// lowest position > pos whose bit is set, or -1
template <int PPL>
__device__ __forceinline__ int first_after(const Masks<PPL>& a, int pos)
{
    int best = -1;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int d = pos - k;                               // lane*PPL + k > pos  <=>  lane > d / PPL (d >= 0), any lane if d < 0
        const int lmin = d < 0 ? 0 : d / PPL + 1;
        if (lmin > 63) continue;
        const u64 m = a.get(k) & ~bits(0, lmin - 1);
        if (m != 0ull) { const int c = (__ffsll((long long)m) - 1) * PPL + k; best = (best < 0 || c < best) ? c : best; }
    }
    return best;
}
// any bit set at a position in [lo, hi]
template <int PPL>
__device__ __forceinline__ bool any_between(const Masks<PPL>& a, int lo, int hi)
{
    if (lo > hi) return false;
    u64 o = 0ull;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int dl = lo - k, dh = hi - k;
        if (dh < 0) continue;
        const int l0 = dl <= 0 ? 0 : (dl + PPL - 1) / PPL, l1 = dh / PPL;
        o |= a.get(k) & bits(l0, l1);
    }
    return o != 0ull;
}
template <int PPL> __device__ __forceinline__ bool bit_at(const Masks<PPL>& a, int pos)
{
    return ((a.get(pos % PPL) >> (pos / PPL)) & 1ull) != 0ull;
}

// The pick helpers take the sub-pose values as SEPARATE SCALAR ARGUMENTS, never as a pointer / reference / struct: hipcc
// optimises each helper stand-alone before it is inlined, and there folds select(c, load a[1], load a[0]) into load a[c] -- the
// caller's array then lives in scratch (or LDS) for good.
constexpr int PPL_MAX = 5;
// GSF_SUBS(E): the value of expression E (written in terms of `j`) for j = 0 .. 4 as five arguments; unused slots repeat j = 0
#define GSF_SUB1(E, K) [&]() __attribute__((always_inline)) { constexpr int j = (PPL > (K)) ? (K) : 0; return (double)(E); }()
#define GSF_SUBS(E) GSF_SUB1(E, 0), GSF_SUB1(E, 1), GSF_SUB1(E, 2), GSF_SUB1(E, 3), GSF_SUB1(E, 4)
template <int PPL> __device__ __forceinline__ double pick_sub(double v0, double v1, double v2, double v3, double v4, int sub)
{
    static_assert(PPL <= PPL_MAX, "at most PPL_MAX poses per lane");
    double x = v0;
    if (PPL > 1) x = (sub == 1) ? v1 : x;
    if (PPL > 2) x = (sub == 2) ? v2 : x;
    if (PPL > 3) x = (sub == 3) ? v3 : x;
    if (PPL > 4) x = (sub == 4) ? v4 : x;
    return x;
}
// value at the wave-uniform position `pos` (lane pos / PPL, sub-pose pos % PPL), broadcast
template <int PPL> __device__ __forceinline__ double pick_bcast(double v0, double v1, double v2, double v3, double v4, int pos)
{
    return lane_bcast(pick_sub<PPL>(v0, v1, v2, v3, v4, pos % PPL), pos / PPL);
}
// value at a per-lane position (lane and sub-pose differ per lane): one bpermute per sub-pose + select
template <int PPL> __device__ __forceinline__ double pick_shfl(double v0, double v1, double v2, double v3, double v4, int pos)
{
    const int src = pos / PPL, sub = pos % PPL;
    double x = shidx(v0, src);
    if (PPL > 1) { const double y = shidx(v1, src); x = (sub == 1) ? y : x; }
    if (PPL > 2) { const double y = shidx(v2, src); x = (sub == 2) ? y : x; }
    if (PPL > 3) { const double y = shidx(v3, src); x = (sub == 3) ? y : x; }
    if (PPL > 4) { const double y = shidx(v4, src); x = (sub == 4) ? y : x; }
    return x;
}

// One chunk.  c0 = index of the chunk's first pose.  `in` = the poses of this lane (already loaded).
template <int PPL>
__device__ __forceinline__ void process_chunk(const TrajPtrs& T, const EkfConfig& cfg, WaveCarry& C, const int64_t c0, const ChunkIn* in,
                                              const int lane)
{
    const int64_t N = T.N;
    const int CH = 64 * PPL;
    const int Lp = (int)((N - c0 < CH) ? (N - c0 - 1) : CH - 1);        // last active position of the chunk
    bool active[PPL], is_init[PPL], stepping[PPL], ok[PPL], vraw[PPL];
    Quat r[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int64_t i = c0 + (int64_t)lane * PPL + j;
        active[j] = i < N; is_init[j] = (i == 0); stepping[j] = active[j] && !is_init[j];
        ok[j] = quat_unit(in[j].q, r[j]);
        vraw[j] = in[j].v != 0;
    }
    Masks<PPL> act_m, ok_m;
#pragma unroll
    for (int j = 0; j < PPL; ++j) { act_m.set(j, __ballot(active[j])); ok_m.set(j, __ballot(ok[j])); }
    // ---- previous pose of every sub-pose: sub 0 <- last sub-pose of the previous lane (lane 0: the carry), sub j <- sub j-1
    double t_pr[PPL]; Vec3 p_pr[PPL]; Quat r_pr[PPL]; bool ok_pr[PPL];
    t_pr[0] = prev_lane(C.t, in[PPL - 1].t);
    p_pr[0] = Vec3{ prev_lane(C.po.x, in[PPL - 1].p.x), prev_lane(C.po.y, in[PPL - 1].p.y), prev_lane(C.po.z, in[PPL - 1].p.z) };
    r_pr[0] = prev_lane(C.r, r[PPL - 1]);
    ok_pr[0] = (lane == 0) ? C.ok : (((ok_m.get(PPL - 1) >> (lane - 1)) & 1ull) != 0ull);
#pragma unroll
    for (int j = 1; j < PPL; ++j) { t_pr[j] = in[j - 1].t; p_pr[j] = in[j - 1].p; r_pr[j] = r[j - 1]; ok_pr[j] = ok[j - 1]; }
    double dt[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) dt[j] = fmax(1e-6, in[j].t - t_pr[j]);                           // ref :865
    bool all_ok = C.ok;
#pragma unroll
    for (int j = 0; j < PPL; ++j) all_ok = all_ok && ((ok_m.get(j) & act_m.get(j)) == act_m.get(j));
    const bool telescope = all_ok;                                        // see gsf_ekf_wave.hip: no invalid quaternion in sight

    // ---- GNSS gate (ref :867-869) and the outage structure as ballots
    bool avail[PPL], av[PPL], ap[PPL], recovers[PPL], outpair[PPL];
    Masks<PPL> a_m;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        avail[j] = stepping[j] && vraw[j] && !(isnan(in[j].z.x) || isnan(in[j].z.y) || isnan(in[j].z.z));
        av[j] = is_init[j] ? vraw[j] : avail[j];                          // pose 0: raw mask (ref :848)
        a_m.set(j, __ballot(active[j] && av[j]));
    }
    ap[0] = (lane == 0) ? (is_init[0] ? true : C.prev_avail) : (((a_m.get(PPL - 1) >> (lane - 1)) & 1ull) != 0ull);
#pragma unroll
    for (int j = 1; j < PPL; ++j) ap[j] = av[j - 1];
    Masks<PPL> start_m, rec_m, pair_m, f_m;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const bool starts = active[j] && !av[j] && ap[j];                 // ref :875-877 (pose 0: :861)
        recovers[j] = stepping[j] && av[j] && !ap[j];                     // ref :879
        outpair[j] = stepping[j] && !av[j] && !ap[j];
        start_m.set(j, __ballot(starts)); rec_m.set(j, __ballot(recovers[j])); pair_m.set(j, __ballot(outpair[j]));
        f_m.set(j, 0ull);
    }
    if (any(start_m)) C.status |= ST_HAD_OUTAGE;
    if (any(pair_m)) {                                                    // is_sharp_turn_in_segment pairs, ref :808-826
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            bool f = false;
            if (outpair[j] && in[j].t > t_pr[j]) f = !(ok_pr[j] && ok[j]) || yaw_rate_exceeds(r_pr[j], r[j], in[j].t - t_pr[j], cfg.yaw_thr_rad);
            f_m.set(j, __ballot(f));
        }
    }
    // recovery decision (ref :879-894)
    bool sharp[PPL];
    Masks<PPL> sharp_m, rts_m;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        sharp[j] = false;
        if (recovers[j]) {
            const int pos = lane * PPL + j;
            const int s = last_before<PPL>(start_m, pos);
            int64_t s_glob; bool seg;
            if (s >= 0) { s_glob = c0 + s; seg = any_between<PPL>(f_m, s + 1, pos - 1); }
            else { s_glob = C.ostart; seg = C.seg_sharp || any_between<PPL>(f_m, 0, pos - 1); }
            sharp[j] = (c0 + pos - s_glob >= 2) && seg;
        }
        sharp_m.set(j, __ballot(sharp[j]));
        rts_m.set(j, rec_m.get(j) & ~sharp_m.get(j));
    }
    if (any(sharp_m)) C.status |= ST_SHARP_TURN;
    if (any(rts_m)) C.status |= ST_RTS_APPLIED;

    // ---- orientation (ref :708-709) and predicted displacement (ref :707)
    Quat qi[PPL]; Vec3 u[PPL];
    if (telescope) {
        const Quat Cq = quat_mul(C.q, quat_conj(C.r));
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            qi[j] = is_init[j] ? C.q : ekf_normalize(quat_mul(Cq, r[j]));
            const Vec3 d = quat_rotate(Cq, Vec3{ in[j].p.x - p_pr[j].x, in[j].p.y - p_pr[j].y, in[j].p.z - p_pr[j].z });
            u[j] = Vec3{ stepping[j] ? d.x : 0.0, stepping[j] ? d.y : 0.0, stepping[j] ? d.z : 0.0 };
        }
    } else {
        // generic: calculate_relative_pose with its zero-motion branch (ref :77-92), prefix product of the increments
        Vec3 dpl[PPL]; Quat dq[PPL];
        bool anybad = false;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const bool move = stepping[j] && ok_pr[j] && ok[j];
            const Quat r1i = quat_conj(r_pr[j]);
            const Vec3 d = quat_rotate(r1i, Vec3{ in[j].p.x - p_pr[j].x, in[j].p.y - p_pr[j].y, in[j].p.z - p_pr[j].z });
            const Quat m = quat_mul(r1i, r[j]);
            dpl[j] = Vec3{ move ? d.x : 0.0, move ? d.y : 0.0, move ? d.z : 0.0 };
            dq[j] = Quat{ move ? m.x : 0.0, move ? m.y : 0.0, move ? m.z : 0.0, move ? m.w : 1.0 };
            anybad = anybad || (__ballot(stepping[j] && !(ok_pr[j] && ok[j])) != 0ull);
        }
        if (anybad) C.status |= ST_BAD_QUAT;
        Quat loc[PPL];                                                    // in-lane prefix products
        loc[0] = dq[0];
#pragma unroll
        for (int j = 1; j < PPL; ++j) loc[j] = quat_mul(loc[j - 1], dq[j]);
        Quat S = loc[PPL - 1];                                            // lane total, then inclusive scan over lanes
        const Quat QID{ 0.0, 0.0, 0.0, 1.0 };
#define GSF_QSTAGE(CTRL, RM) { const Quat o = dpp<CTRL, RM>(QID, S); S = quat_mul(o, S); }
        GSF_SCAN_STAGES(GSF_QSTAGE)
#undef GSF_QSTAGE
        const Quat X = prev_lane(QID, S);                                 // product of all earlier lanes
        const Quat base = quat_mul(C.q, X);
        Quat qprev = ekf_normalize(base);
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            qi[j] = ekf_normalize(quat_mul(base, loc[j]));
            u[j] = quat_rotate(qprev, dpl[j]);
            qprev = qi[j];
        }
    }

    // ---- variances: Moebius maps P -> (A P + B)/(Cm P + D) per axis (ref :712-713, :723-731).  Lane-local composition of
    // the PPL elements, one inclusive scan of the lane totals, then the in-lane values by sequential application.
    double Pf[PPL][3], Pm[PPL][3], kg[PPL][3];
    auto variance_axis = [&](const int c) {
        const double rr = cfg.Rm[c];
        double b0[PPL];
        double A = 1.0, Bm = 0.0, Cm = 0.0, Dm = 1.0;                     // lane total = E_{PPL-1} o ... o E_0
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            b0[j] = stepping[j] ? cfg.Qps[c] * dt[j] : 0.0;
            double eA = 1.0, eB = b0[j], eC = 0.0, eD = 1.0;
            if (avail[j]) { eA = rr; eB = rr * b0[j]; eC = 1.0; eD = b0[j] + rr; }
            if (j == 0) { A = eA; Bm = eB; Cm = eC; Dm = eD; }
            else { const double nA = eA * A + eB * Cm, nB = eA * Bm + eB * Dm, nC = eC * A + eD * Cm, nD = eC * Bm + eD * Dm; A = nA; Bm = nB; Cm = nC; Dm = nD; }
        }
#define GSF_MSTAGE(CTRL, RM) {                                                                                              \
        const double oA = dpp<CTRL, RM>(1.0, A), oB = dpp<CTRL, RM>(0.0, Bm), oC = dpp<CTRL, RM>(0.0, Cm), oD = dpp<CTRL, RM>(1.0, Dm); \
        const double nA = A * oA + Bm * oC, nB = A * oB + Bm * oD, nC = Cm * oA + Dm * oC, nD = Cm * oB + Dm * oD;                  \
        A = nA; Bm = nB; Cm = nC; Dm = nD; }
        GSF_SCAN_STAGES(GSF_MSTAGE)
#undef GSF_MSTAGE
        const double Plast = (A * C.P[c] + Bm) * fast_rcp(Cm * C.P[c] + Dm);          // P_f of the lane's last sub-pose
        double Pprev = prev_lane(C.P[c], Plast);                          // P_f of the pose before this lane's first sub-pose
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            Pm[j][c] = Pprev + b0[j];                                     // P_p
            kg[j][c] = Pm[j][c] * fast_rcp(Pm[j][c] + rr);
            Pf[j][c] = (j == PPL - 1) ? Plast : (avail[j] ? rr * kg[j][c] : Pm[j][c]);   // r P/(P+r) == (1-k)^2 P + k^2 r
            Pprev = Pf[j][c];
        }
    };
    // axes with identical (P0, Q, R) have identical variance recursions (default CONFIG: x == y): compute once, copy with
    // compile-time indices (wave-uniform branches)
    variance_axis(0);
    if (C.same_axis[1] == 0) { for (int j = 0; j < PPL; ++j) { Pf[j][1] = Pf[j][0]; Pm[j][1] = Pm[j][0]; kg[j][1] = kg[j][0]; } }
    else variance_axis(1);
    if (C.same_axis[2] == 0) { for (int j = 0; j < PPL; ++j) { Pf[j][2] = Pf[j][0]; Pm[j][2] = Pm[j][0]; kg[j][2] = kg[j][0]; } }
    else if (C.same_axis[2] == 1) { for (int j = 0; j < PPL; ++j) { Pf[j][2] = Pf[j][1]; Pm[j][2] = Pm[j][1]; kg[j][2] = kg[j][1]; } }
    else variance_axis(2);

    // ---- positions: affine maps x -> al x + be in chunk-local coordinates (x = p - p_carry), ref :707, :728, :762-763
    double xl[PPL][3], dcorr[PPL][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double ea[PPL], eb[PPL];
        double al = 1.0, be = 0.0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const double uu = (c == 0) ? u[j].x : (c == 1 ? u[j].y : u[j].z);
            const double zz = ((c == 0) ? in[j].z.x : (c == 1 ? in[j].z.y : in[j].z.z)) - ((c == 0) ? C.p.x : (c == 1 ? C.p.y : C.p.z));
            double wgt = 1.0;                                             // one-step blend on a sharp-turn recovery (Q7)
            if (sharp[j] && cfg.sharp_turn_steps > 1) wgt = 1.0 / (double)cfg.sharp_turn_steps;
            const double kw = kg[j][c] * wgt;
            ea[j] = avail[j] ? (1.0 - kw) : 1.0;
            eb[j] = avail[j] ? ((1.0 - kw) * uu + kw * zz) : uu;
            if (j == 0) { al = ea[0]; be = eb[0]; } else { be = ea[j] * be + eb[j]; al = ea[j] * al; }
        }
#define GSF_ASTAGE(CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al), ob = dpp<CTRL, RM>(0.0, be); be = al * ob + be; al = al * oa; }
        GSF_SCAN_STAGES(GSF_ASTAGE)
#undef GSF_ASTAGE
        double xprev = prev_lane(0.0, be);                                // x of the pose before this lane's first sub-pose
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const double uu = (c == 0) ? u[j].x : (c == 1 ? u[j].y : u[j].z);
            xl[j][c] = (j == PPL - 1) ? be : (ea[j] * xprev + eb[j]);
            dcorr[j][c] = xl[j][c] - (xprev + uu);                        // x_f - x_p (non-zero only where a fix was used)
            xprev = xl[j][c];
        }
    }

    // ---- per-outage RTS (ref :906-922, :777-803): x_s[k] = x_f[k] + (P_f[k] / P_p[r]) (x_f[r] - x_p[r]), r = recovery pose
    double xo[PPL][3];
#pragma unroll
    for (int j = 0; j < PPL; ++j) { xo[j][0] = xl[j][0]; xo[j][1] = xl[j][1]; xo[j][2] = xl[j][2]; }
    if (any(rts_m)) {
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int pos = lane * PPL + j;
            const int nr = first_after<PPL>(rec_m, pos);
            const int nrc = nr >= 0 ? nr : 0;
            const bool in_run = active[j] && !av[j] && nr >= 0 && bit_at<PPL>(rts_m, nrc);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double dr = pick_shfl<PPL>(GSF_SUBS(dcorr[j][c]), nrc), pr = pick_shfl<PPL>(GSF_SUBS(Pm[j][c]), nrc);
                if (in_run) xo[j][c] = xl[j][c] + Pf[j][c] * fast_rcp(pr) * dr;
            }
        }
        // outage carried in from earlier chunks and closed here: fix the rows that are already in memory
        if (!C.prev_avail) {
            const int r1 = first_after<PPL>(rec_m, -1);                   // first recovery of the chunk closes the carried run
            if (r1 >= 0 && bit_at<PPL>(rts_m, r1)) {
                double dr[3], ipr[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { dr[c] = pick_bcast<PPL>(GSF_SUBS(dcorr[j][c]), r1); ipr[c] = fast_rcp(pick_bcast<PPL>(GSF_SUBS(Pm[j][c]), r1)); }
                double acc = 0.0;                                         // sum of dt over (ostart, k]
                for (int64_t k0 = (C.ostart / 64) * 64; k0 < c0; k0 += 64) {
                    const int64_t k = k0 + lane;
                    const double tk = T.ts[k];
                    const double tkp = prev_lane((k0 > 0) ? T.ts[k0 - 1] : tk, tk);
                    double dsum = (k > C.ostart) ? fmax(1e-6, tk - tkp) : 0.0;
#define GSF_SSTAGE(CTRL, RM) { dsum += dpp<CTRL, RM>(0.0, dsum); }
                    GSF_SCAN_STAGES(GSF_SSTAGE)
#undef GSF_SSTAGE
                    const double tot = lane_bcast(dsum, 63);
                    if (k >= C.ostart) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const double Pk = C.Pos[c] + cfg.Qps[c] * (acc + dsum);      // P_f[k] inside the outage
                            T.pos_out[k * 3 + c] += Pk * ipr[c] * dr[c];
                        }
                    }
                    acc += tot;
                }
            }
        }
    }

    // ---- output rows
    double orow[PPL][3];
#pragma unroll
    for (int j = 0; j < PPL; ++j) { orow[j][0] = C.p.x + xo[j][0]; orow[j][1] = C.p.y + xo[j][1]; orow[j][2] = C.p.z + xo[j][2]; }

    // ---- carry to the next chunk, from the last active position Lp
    const bool open = !bit_at<PPL>(a_m, Lp);                              // the chunk ends inside an outage
    if (open) {
        const int s = last_before<PPL>(start_m, Lp + 1);
        if (s >= 0) {
            C.ostart = c0 + s;
            C.seg_sharp = any_between<PPL>(f_m, s + 1, Lp);
#pragma unroll
            for (int c = 0; c < 3; ++c) C.Pos[c] = pick_bcast<PPL>(GSF_SUBS(Pf[j][c]), s);
        } else {
            C.seg_sharp = C.seg_sharp || any_between<PPL>(f_m, 0, Lp);
        }
    }
    C.prev_avail = !open;
    {
#define GSF_CARRY(dst, expr) { dst = pick_bcast<PPL>(GSF_SUBS(expr), Lp); }
        double x0, x1, x2;
        GSF_CARRY(x0, xl[j][0]) GSF_CARRY(x1, xl[j][1]) GSF_CARRY(x2, xl[j][2])
        Quat nq; GSF_CARRY(nq.x, qi[j].x) GSF_CARRY(nq.y, qi[j].y) GSF_CARRY(nq.z, qi[j].z) GSF_CARRY(nq.w, qi[j].w)
        C.q = nq;
        C.p = Vec3{ C.p.x + x0, C.p.y + x1, C.p.z + x2 };
        GSF_CARRY(C.P[0], Pf[j][0]) GSF_CARRY(C.P[1], Pf[j][1]) GSF_CARRY(C.P[2], Pf[j][2])
        GSF_CARRY(C.po.x, in[j].p.x) GSF_CARRY(C.po.y, in[j].p.y) GSF_CARRY(C.po.z, in[j].p.z)
        GSF_CARRY(C.r.x, r[j].x) GSF_CARRY(C.r.y, r[j].y) GSF_CARRY(C.r.z, r[j].z) GSF_CARRY(C.r.w, r[j].w)
        GSF_CARRY(C.t, in[j].t)
#undef GSF_CARRY
        C.ok = bit_at<PPL>(ok_m, Lp);
    }
#pragma unroll
    for (int j = 0; j < PPL; ++j) {                                       // PPL consecutive poses per lane
        if (active[j]) {
            const int64_t i = c0 + (int64_t)lane * PPL + j;
            T.pos_out[i * 3] = orow[j][0]; T.pos_out[i * 3 + 1] = orow[j][1]; T.pos_out[i * 3 + 2] = orow[j][2];
            T.quat_out[i * 4] = qi[j].x; T.quat_out[i * 4 + 1] = qi[j].y; T.quat_out[i * 4 + 2] = qi[j].z; T.quat_out[i * 4 + 3] = qi[j].w;
        }
    }
}

}  // namespace
