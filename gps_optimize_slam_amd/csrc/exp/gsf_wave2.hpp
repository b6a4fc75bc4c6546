// gsf_wave2.hpp -- the wave-per-trajectory filter with TWO CONSECUTIVE POSES PER LANE: chunks of 128 poses, lane l holds pose 2l
// ("slot A") and pose 2l + 1 ("slot B") of the chunk.
//
// Why: a 64-pose chunk of the one-pose-per-lane loop (gsf_wave_common.hpp) executes 623 vector
// instructions of which 282 are its four prefix scans (two Moebius scans for the variances, two affine scans for the positions; six
// shuffle stages each, 218 of them cross-lane moves that cost as much as the FMA they feed or twice that).  A scan over 64 lanes costs the
// same whether a lane carries one pose or the composition of two: with two consecutive poses per lane the pair is composed locally
// (8 flops for a Moebius pair, 3 per axis for an affine pair), ONE scan runs over the 64 pair totals, and the second pose of a lane
// follows from the first by one serial step.  Per pose that is half the scan stages, half the carries and chunk bookkeeping, and a
// lane's two rows are 48 / 64 / 48 contiguous bytes: fetched and stored as whole 16-byte pieces without the LDS transposition of the
// big-batch build.
//
// Same recursion, same reference lines (apply_ekf_correction, EKFGPSSLAM.py:831-935) as wave_serial_chunks; what changes is the
// bookkeeping: every lane predicate exists twice (slot A, slot B) and a position inside the chunk is p = 2 * lane + slot, so the
// outage structure is worked out on PAIRS of 64-bit masks (M2).
#pragma once
#include "../gsf_wave_common.hpp"

namespace {

// ---- pairs of lane masks: a = the poses 2l (slot A), b = the poses 2l + 1 (slot B); position p = 2l + slot, 0 <= p < 128
struct M2 { u64 a, b; };
__device__ __forceinline__ M2 operator&(const M2 x, const M2 y) { return M2{ x.a & y.a, x.b & y.b }; }
__device__ __forceinline__ M2 operator|(const M2 x, const M2 y) { return M2{ x.a | y.a, x.b | y.b }; }
__device__ __forceinline__ M2 operator~(const M2 x) { return M2{ ~x.a, ~x.b }; }
__device__ __forceinline__ bool m2_any(const M2 x) { return (x.a | x.b) != 0ull; }
__device__ __forceinline__ M2 m2_first(const int n) { return M2{ mask_first((n + 1) >> 1), mask_first(n >> 1) }; }   // positions 0..n-1
__device__ __forceinline__ M2 m2_range(const int lo, const int hi) { const M2 u = m2_first(hi + 1), d = m2_first(lo); return M2{ u.a & ~d.a, u.b & ~d.b }; }   // lo..hi inclusive (empty if lo > hi)
// the flag of position p - 1 seen at position p (position 0 sees `carry`)
__device__ __forceinline__ M2 m2_prev(const M2 x, const bool carry) { return M2{ (x.b << 1) | (carry ? 1ull : 0ull), x.a }; }
__device__ __forceinline__ int m2_last(const M2 x)                        // highest position set, -1 if none
{
    const int pa = x.a != 0ull ? 2 * (63 - __clzll((long long)x.a)) : -1, pb = x.b != 0ull ? 2 * (63 - __clzll((long long)x.b)) + 1 : -1;
    return pa > pb ? pa : pb;
}
__device__ __forceinline__ int m2_lowest(const M2 x)                      // lowest position set, 128 if none
{
    const int pa = x.a != 0ull ? 2 * (__ffsll((long long)x.a) - 1) : 128, pb = x.b != 0ull ? 2 * (__ffsll((long long)x.b) - 1) + 1 : 128;
    return pa < pb ? pa : pb;
}
__device__ __forceinline__ bool m2_bit(const M2 x, const int p) { return (((p & 1) ? x.b : x.a) >> (p >> 1)) & 1ull; }
__device__ __forceinline__ int m2_count(const M2 x) { return __popcll(x.a) + __popcll(x.b); }

// value of position p (wave-uniform) of a per-slot pair of lane values
__device__ __forceinline__ double pos_bcast(const double va, const double vb, const int p) { return (p & 1) ? lane_bcast(vb, p >> 1) : lane_bcast(va, p >> 1); }
// ... of a per-lane position (rare paths)
__device__ __forceinline__ double pos_gather(const double va, const double vb, const int p) { const double x = shidx(va, p >> 1), y = shidx(vb, p >> 1); return (p & 1) ? y : x; }

typedef double w2_v2 __attribute__((ext_vector_type(2), aligned(8)));
// the two rows of a lane, as loaded: ts[2], pos[6], quat[8], gps[6] in row order, the two mask bytes
typedef uint16_t w2_u16 __attribute__((aligned(1)));
struct Rows2 { w2_v2 T; w2_v2 P[3]; w2_v2 Q[4]; w2_v2 Z[3]; uint32_t vv; };   // vv: the two mask bytes as loaded (split at use time, never at load time)
// rows r0, r0 + 1 of the track (r0 <= N - 2): every array as whole 16-byte pieces
__device__ __forceinline__ Rows2 load_rows2(const double* __restrict__ tsb, const double* __restrict__ posb, const double* __restrict__ quatb,
                                            const double* __restrict__ gpsb, const uint8_t* __restrict__ valb, const int64_t r0)
{
    Rows2 w;
#define GSF_NT2(p) __builtin_nontemporal_load((const w2_v2*)(p))
    w.T = GSF_NT2(tsb + r0);
    w.P[0] = GSF_NT2(posb + r0 * 3); w.P[1] = GSF_NT2(posb + r0 * 3 + 2); w.P[2] = GSF_NT2(posb + r0 * 3 + 4);
    w.Q[0] = GSF_NT2(quatb + r0 * 4); w.Q[1] = GSF_NT2(quatb + r0 * 4 + 2); w.Q[2] = GSF_NT2(quatb + r0 * 4 + 4); w.Q[3] = GSF_NT2(quatb + r0 * 4 + 6);
    w.Z[0] = GSF_NT2(gpsb + r0 * 3); w.Z[1] = GSF_NT2(gpsb + r0 * 3 + 2); w.Z[2] = GSF_NT2(gpsb + r0 * 3 + 4);
#undef GSF_NT2
    w.vv = __builtin_nontemporal_load((const w2_u16*)(valb + r0));
    return w;
}
__device__ __forceinline__ void rows2_arrived(const Rows2& w)
{
    asm volatile("" :: "v"(w.T), "v"(w.P[0]), "v"(w.P[1]), "v"(w.P[2]), "v"(w.Q[0]), "v"(w.Q[1]), "v"(w.Q[2]), "v"(w.Q[3]), "v"(w.Z[0]), "v"(w.Z[1]), "v"(w.Z[2]),
                 "v"(w.vv) : "memory");
}

// One Moebius element of the variance recursion (ref :712-713, :723-731): P -> (A P + B) / (C P + D)
__device__ __forceinline__ Moebius var_elem(const double b0, const double rr, const bool stepping, const bool avail)
{
    Moebius m{ 1.0, stepping ? b0 : 0.0, 0.0, 1.0 };
    if (avail) { m.A = rr; m.B = rr * b0; m.C = 1.0; m.D = b0 + rr; }
    return m;
}
// later o earlier
__device__ __forceinline__ Moebius moebius_mul(const Moebius& l, const Moebius& e)
{
    return Moebius{ l.A * e.A + l.B * e.C, l.A * e.B + l.B * e.D, l.C * e.A + l.D * e.C, l.C * e.B + l.D * e.D };
}
struct AxisVar2 { double PfA, PmA, kgA, PfB, PmB, kgB; };
// variances of the 128 poses of a chunk for one axis: pair totals scanned over the lanes, the first pose of a lane stepped from the last
// pose of the lane before, the second from the first
__device__ __forceinline__ AxisVar2 variance_axis2(const double q, const double rr, const double dtA, const double dtB, const bool stepA, const bool stepB,
                                                   const bool availA, const bool availB, const double cPc)
{
    const double bA = q * dtA, bB = q * dtB;
    const Moebius eA = var_elem(bA, rr, stepA, availA), eB = var_elem(bB, rr, stepB, availB);
    Moebius m = moebius_mul(eB, eA);
    double A = m.A, Bm = m.B, Cm = m.C, Dm = m.D;
#define GSF_MSTAGE2(CTRL, RM) {                                                                                             \
        const double oA = dpp<CTRL, RM>(1.0, A), oB = dpp0<CTRL, RM>(Bm), oC = dpp0<CTRL, RM>(Cm), oD = dpp<CTRL, RM>(1.0, Dm); \
        const double nA = A * oA + Bm * oC, nB = A * oB + Bm * oD, nC = Cm * oA + Dm * oC, nD = Cm * oB + Dm * oD;                  \
        A = nA; Bm = nB; Cm = nC; Dm = nD; }
    GSF_SCAN_STAGES(GSF_MSTAGE2)
#undef GSF_MSTAGE2
    AxisVar2 v;
    v.PfB = moebius_apply(Moebius{ A, Bm, Cm, Dm }, cPc);                // P_f of the lane's second pose
    const double Pin = prev_lane(cPc, v.PfB);                            // P_f of the pose in front of the lane's first one
    v.PmA = stepA ? Pin + bA : Pin;
    v.kgA = v.PmA * fast_rcp(v.PmA + rr);
    v.PfA = availA ? rr * v.kgA : v.PmA;
    v.PmB = stepB ? v.PfA + bB : v.PfA;
    v.kgB = v.PmB * fast_rcp(v.PmB + rr);
    return v;
}

// The chunk loop (two poses per lane).  Default noise layout only (x and y share (P0, Q, R), z does not: checked by the launcher).
#define W2_STAMP(k) do { if (c0 == 128) GSF_STAMP(k); } while (0)
template <bool PIPELINE>
__device__ __forceinline__ void wave2_serial_chunks(const WaveArgs& a, const EkfConfig& cfg, const int64_t b, const int lane, const int64_t base,
                                                    const int64_t N, const Vec3& p0, const Quat& q0, const int32_t fit)
{
    const double* __restrict__ tsb = a.ts + base;
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ quatb = a.quat + base * 4;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;
    double* __restrict__ pob = a.pos_out + base * 3;
    double* __restrict__ qob = a.quat_out + base * 4;
    __shared__ double ring2[2][6][2][64];                                // rows + P_f of the last two open-outage chunks: [chunk parity][value][slot][lane]

    // ---- carry (wave-uniform)
    Quat cq = ekf_normalize(q0);                                         // ref :842, :683
    Vec3 cp = p0;
    double cP[3] = { cfg.P0[0], cfg.P0[1], cfg.P0[2] };
    int64_t c_ostart = 0;                                                // ref :861-862
    bool c_seg_sharp = false;
    double cPos[3] = { cP[0], cP[1], cP[2] };
    // rows of the first chunk
    auto row0_of = [&](const int64_t c0) __attribute__((always_inline)) { const int64_t r = c0 + 2 * lane; return r <= N - 2 ? r : N - 2; };
    Rows2 nxt = load_rows2(tsb, posb, quatb, gpsb, valb, row0_of(0));
    rows2_arrived(nxt);
    bool c_prev_avail = (__builtin_amdgcn_readlane((int)nxt.vv, 0) & 0xff) != 0;  // ref :848 (raw mask of pose 0)
    Vec3 c_po{ lane_bcast(nxt.P[0].x, 0), lane_bcast(nxt.P[0].y, 0), lane_bcast(nxt.P[1].x, 0) };
    const Quat c_q0{ lane_bcast(nxt.Q[0].x, 0), lane_bcast(nxt.Q[0].y, 0), lane_bcast(nxt.Q[1].x, 0), lane_bcast(nxt.Q[1].y, 0) };
    Quat c_r; bool c_ok = quat_unit(c_q0, c_r);
    double c_t = lane_bcast(nxt.T.x, 0);
    int32_t status = c_prev_avail ? 0 : ST_HAD_OUTAGE;
    const Quat cq0 = cq;
    Quat Cq = lane_bcast(quat_mul(cq, quat_conj(c_r)), 0);
    bool cq_fresh = true;

    for (int64_t c0 = 0; c0 < N; c0 += 128) {
        const int Lp = (int)((N - c0 < 128) ? (N - c0 - 1) : 127);       // last active position of the chunk
        const M2 act = m2_first(Lp + 1);
        W2_STAMP(8);
        const M2 init{ (c0 == 0) ? 1ull : 0ull, 0ull };
        const M2 step = act & ~init;
        const bool actA = __builtin_amdgcn_inverse_ballot_w64(act.a), actB = __builtin_amdgcn_inverse_ballot_w64(act.b);
        const bool is_initA = __builtin_amdgcn_inverse_ballot_w64(init.a);
        const bool stepA = __builtin_amdgcn_inverse_ballot_w64(step.a), stepB = actB;
        // ---- this chunk's rows; the next chunk's are requested now
        const Rows2 w = nxt;
        const int64_t r0 = row0_of(c0);
        if (c0 + 128 < N) nxt = load_rows2(tsb, posb, quatb, gpsb, valb, row0_of(c0 + 128));
        // a lane whose first pose is the LAST row of an odd-length track loaded rows N-2, N-1: its slot A is the second row
        const bool shifted = (c0 + 2 * lane == N - 1) && (r0 != c0 + 2 * lane);
        double tA = w.T.x, tB = w.T.y;
        Vec3 pA{ w.P[0].x, w.P[0].y, w.P[1].x }, pB{ w.P[1].y, w.P[2].x, w.P[2].y };
        Quat qA{ w.Q[0].x, w.Q[0].y, w.Q[1].x, w.Q[1].y }, qB{ w.Q[2].x, w.Q[2].y, w.Q[3].x, w.Q[3].y };
        Vec3 zA{ w.Z[0].x, w.Z[0].y, w.Z[1].x }, zB{ w.Z[1].y, w.Z[2].x, w.Z[2].y };
        uint32_t vA = w.vv & 0xffu, vB = w.vv >> 8;
        if (__ballot(shifted) != 0ull) {                                 // wave-uniform, only the last chunk of an odd-length track
            if (shifted) { tA = tB; pA = pB; qA = qB; zA = zB; vA = vB; }
        }
        const M2 vraw{ mask_nonzero(vA), mask_nonzero(vB) };
        // ---- calculate_relative_pose (ref :77-92): pose B against pose A of the same lane, pose A against pose B of the lane before
        Quat rA, rB; const bool okA = quat_unit(qA, rA), okB = quat_unit(qB, rB);
        const double tpA = prev_lane(c_t, tB);
        const Vec3 ppA{ prev_lane(c_po.x, pB.x), prev_lane(c_po.y, pB.y), prev_lane(c_po.z, pB.z) };
        const M2 okm{ __ballot(okA), __ballot(okB) };
        const M2 okp = m2_prev(okm, c_ok);
        const M2 bothm = okm & okp;
        const bool bothA = __builtin_amdgcn_inverse_ballot_w64(bothm.a), bothB = __builtin_amdgcn_inverse_ballot_w64(bothm.b);
        const double dtA = fmax(1e-6, tA - tpA), dtB = fmax(1e-6, tB - tA);   // ref :865
        const bool telescope = c_ok && ((okm.a & act.a) == act.a) && ((okm.b & act.b) == act.b);
        // ---- GNSS gate (ref :867-869) and the outage structure, as mask pairs
        const M2 zfin{ mask_not_nan(zA.x) & mask_not_nan(zA.y) & mask_not_nan(zA.z), mask_not_nan(zB.x) & mask_not_nan(zB.y) & mask_not_nan(zB.z) };
        const M2 availm = step & vraw & zfin;
        const bool availA = __builtin_amdgcn_inverse_ballot_w64(availm.a), availB = __builtin_amdgcn_inverse_ballot_w64(availm.b);
        const M2 avm = (init & vraw) | availm;                           // "gnss available" flag of a pose (pose 0: raw mask, :848)
        const M2 a_mask = act & avm;
        const M2 apm = m2_prev(a_mask, c0 == 0 || c_prev_avail);        // the flag of the pose in front (pose 0: true)
        const M2 start_m = act & ~avm & apm;                             // outage begins (ref :875-877; pose 0: :861)
        const M2 rec_m = step & avm & ~apm;                              // ref :879
        const M2 pair_m = step & ~avm & ~apm;                            // poses i-1 and i both inside the outage
        status |= m2_any(start_m) ? ST_HAD_OUTAGE : 0;
        // is_sharp_turn_in_segment (ref :808-826): pair (i-1, i) exceeds the yaw-rate threshold (or has a bad quaternion)
        M2 f_m{ 0ull, 0ull };
        if (m2_any(pair_m)) {
            const Quat rpA = prev_lane(c_r, rB);
            const bool pairA = __builtin_amdgcn_inverse_ballot_w64(pair_m.a), pairB = __builtin_amdgcn_inverse_ballot_w64(pair_m.b);
            bool fA = false, fB = false;
            if (pairA && tA > tpA) fA = !bothA || yaw_rate_exceeds_body(rpA, rA, tA - tpA, cfg.yaw_thr_rad);
            if (pairB && tB > tA) fB = !bothB || yaw_rate_exceeds_body(rA, rB, tB - tA, cfg.yaw_thr_rad);
            f_m = M2{ __ballot(fA), __ballot(fB) };
        }
        // recovery decision per recovering pose (ref :879-894)
        bool sharpA = false, sharpB = false;
        if (m2_any(rec_m)) {
            auto decide = [&](const int p) __attribute__((always_inline)) {
                const int s = m2_last(start_m & m2_first(p));            // the outage's first pose, if it lies in this chunk
                int64_t s_glob; bool seg;
                if (s >= 0) { s_glob = c0 + s; seg = m2_any(f_m & m2_range(s + 1, p - 1)); }
                else { s_glob = c_ostart; seg = c_seg_sharp || m2_any(f_m & m2_first(p)); }
                return (c0 + p - s_glob >= 2) && seg;
            };
            if (__builtin_amdgcn_inverse_ballot_w64(rec_m.a)) sharpA = decide(2 * lane);
            if (__builtin_amdgcn_inverse_ballot_w64(rec_m.b)) sharpB = decide(2 * lane + 1);
        }
        const M2 sharp_m{ __ballot(sharpA), __ballot(sharpB) };
        const M2 rts_m = rec_m & ~sharp_m;
        status |= (m2_any(sharp_m) ? ST_SHARP_TURN : 0) | (m2_any(rts_m) ? ST_RTS_APPLIED : 0);
        const double wgt_sharp = (cfg.sharp_turn_steps > 1) ? 1.0 / (double)cfg.sharp_turn_steps : 1.0;
        const double wgtA = sharpA ? wgt_sharp : 1.0, wgtB = sharpB ? wgt_sharp : 1.0;
        W2_STAMP(9);

        // ---- orientation (ref :708-709) and predicted displacement (ref :707), telescoped form
        Quat qiA = quat_mul(Cq, rA), qiB = quat_mul(Cq, rB);
        qiA.x = is_initA ? cq0.x : qiA.x; qiA.y = is_initA ? cq0.y : qiA.y; qiA.z = is_initA ? cq0.z : qiA.z; qiA.w = is_initA ? cq0.w : qiA.w;
        Vec3 uA = quat_rotate(Cq, Vec3{ pA.x - ppA.x, pA.y - ppA.y, pA.z - ppA.z });
        Vec3 uB = quat_rotate(Cq, Vec3{ pB.x - pA.x, pB.y - pA.y, pB.z - pA.z });
        uA.x = stepA ? uA.x : 0.0; uA.y = stepA ? uA.y : 0.0; uA.z = stepA ? uA.z : 0.0;
        uB.x = stepB ? uB.x : 0.0; uB.y = stepB ? uB.y : 0.0; uB.z = stepB ? uB.z : 0.0;

        // ---- variances (x == y share the scan, z has its own)
        const AxisVar2 vx = variance_axis2(cfg.Qps[0], cfg.Rm[0], dtA, dtB, stepA, stepB, availA, availB, cP[0]);
        const AxisVar2 vz = variance_axis2(cfg.Qps[2], cfg.Rm[2], dtA, dtB, stepA, stepB, availA, availB, cP[2]);
        W2_STAMP(10);

        if (telescope) cq_fresh = false;
        else {
            // calculate_relative_pose, ref :77-92 -- generic path: a quaternion prefix product over the pairs
            if (!cq_fresh) { cq = quat_mul(Cq, c_r); cq_fresh = true; }
            const Quat rpA = prev_lane(c_r, rB);
            const Quat r1A = quat_conj(rpA), r1B = quat_conj(rA);
            Vec3 dplA = quat_rotate(r1A, Vec3{ pA.x - ppA.x, pA.y - ppA.y, pA.z - ppA.z }), dplB = quat_rotate(r1B, Vec3{ pB.x - pA.x, pB.y - pA.y, pB.z - pA.z });
            Quat dqA = quat_mul(r1A, rA), dqB = quat_mul(r1B, rB);
            const bool moveA = stepA && bothA, moveB = stepB && bothB;
            dplA.x = moveA ? dplA.x : 0.0; dplA.y = moveA ? dplA.y : 0.0; dplA.z = moveA ? dplA.z : 0.0;
            dplB.x = moveB ? dplB.x : 0.0; dplB.y = moveB ? dplB.y : 0.0; dplB.z = moveB ? dplB.z : 0.0;
            dqA.x = moveA ? dqA.x : 0.0; dqA.y = moveA ? dqA.y : 0.0; dqA.z = moveA ? dqA.z : 0.0; dqA.w = moveA ? dqA.w : 1.0;
            dqB.x = moveB ? dqB.x : 0.0; dqB.y = moveB ? dqB.y : 0.0; dqB.z = moveB ? dqB.z : 0.0; dqB.w = moveB ? dqB.w : 1.0;
            if ((__ballot(stepA && !bothA) | __ballot(stepB && !bothB)) != 0ull) status |= ST_BAD_QUAT;
            Quat D = quat_mul(dqA, dqB);                                 // the pair's increment; inclusive prefix product over the lanes
            const Quat QID{ 0.0, 0.0, 0.0, 1.0 };
#define GSF_QSTAGE2(CTRL, RM) { const Quat o = dpp<CTRL, RM>(QID, D); D = quat_mul(o, D); }
            GSF_SCAN_STAGES(GSF_QSTAGE2)
#undef GSF_QSTAGE2
            const Quat Dprev = prev_lane(QID, D);                        // product of the pairs in front of this lane
            const Quat qinA_raw = quat_mul(cq, Dprev);                   // state quaternion in front of pose A (not normalised: one normalisation per pose below)
            qiA = ekf_normalize(quat_mul(qinA_raw, dqA));
            qiB = ekf_normalize(quat_mul(cq, D));
            qiA.x = is_initA ? cq0.x : qiA.x; qiA.y = is_initA ? cq0.y : qiA.y; qiA.z = is_initA ? cq0.z : qiA.z; qiA.w = is_initA ? cq0.w : qiA.w;
            const Quat q_prevA = prev_lane(cq, qiB);
            uA = quat_rotate(q_prevA, dplA);
            uB = quat_rotate(qiA, dplB);
        }

        // ---- positions: affine maps x -> al x + be in chunk-local coordinates (x = p - p_carry), pairs composed, one scan over the lanes
        const double uuA[3] = { uA.x, uA.y, uA.z }, uuB[3] = { uB.x, uB.y, uB.z };
        W2_STAMP(11);
        const double zlA[3] = { zA.x - cp.x, zA.y - cp.y, zA.z - cp.z }, zlB[3] = { zB.x - cp.x, zB.y - cp.y, zB.z - cp.z };
        const double kgA[3] = { vx.kgA, vx.kgA, vz.kgA }, kgB[3] = { vx.kgB, vx.kgB, vz.kgB };
        double alA[3], beA[3], alB[3], beB[3], AL[3], BE[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double kwA = kgA[c] * wgtA, kwB = kgB[c] * wgtB;
            alA[c] = availA ? (1.0 - kwA) : 1.0; beA[c] = availA ? ((1.0 - kwA) * uuA[c] + kwA * zlA[c]) : uuA[c];
            alB[c] = availB ? (1.0 - kwB) : 1.0; beB[c] = availB ? ((1.0 - kwB) * uuB[c] + kwB * zlB[c]) : uuB[c];
            AL[c] = alB[c] * alA[c]; BE[c] = alB[c] * beA[c] + beB[c];
        }
        // x and y share the gain, hence the multiplicative part: one joint scan of (AL; BE_x, BE_y), one of (AL_z; BE_z)
#define GSF_ASTAGE2_XY(CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, AL[0]), ob0 = dpp0<CTRL, RM>(BE[0]), ob1 = dpp0<CTRL, RM>(BE[1]); \
                                   BE[0] = AL[0] * ob0 + BE[0]; BE[1] = AL[0] * ob1 + BE[1]; AL[0] = AL[0] * oa; }
#define GSF_ASTAGE2_Z(CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, AL[2]), ob = dpp0<CTRL, RM>(BE[2]); BE[2] = AL[2] * ob + BE[2]; AL[2] = AL[2] * oa; }
        GSF_SCAN_STAGES(GSF_ASTAGE2_XY)
        GSF_SCAN_STAGES(GSF_ASTAGE2_Z)
#undef GSF_ASTAGE2_Z
#undef GSF_ASTAGE2_XY
        double xA[3], xB[3], xin[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xB[c] = BE[c];                                              // x of the lane's second pose (the carry is x = 0)
            xin[c] = prev_lane(0.0, xB[c]);                             // x in front of the lane's first pose
            xA[c] = alA[c] * xin[c] + beA[c];
        }

        // ---- per-outage RTS (ref :906-922, :777-803): x_s[k] = x_f[k] + (P_f[k] / P_p[r]) (x_f[r] - x_p[r]) for k in [start, r-1]
        const double PfA[3] = { vx.PfA, vx.PfA, vz.PfA }, PfB[3] = { vx.PfB, vx.PfB, vz.PfB };
        const double PmA[3] = { vx.PmA, vx.PmA, vz.PmA }, PmB[3] = { vx.PmB, vx.PmB, vz.PmB };
        double xoA[3] = { xA[0], xA[1], xA[2] }, xoB[3] = { xB[0], xB[1], xB[2] };
        W2_STAMP(12);
        if (m2_any(rts_m)) {
            double dcA[3], dcB[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { dcA[c] = xA[c] - (xin[c] + uuA[c]); dcB[c] = xB[c] - (xA[c] + uuB[c]); }   // x_f - x_p (non-zero only where a fix was used)
            auto smooth = [&](const int p, const bool active, const bool av, const double* xl, const double* Pf, double* xo) __attribute__((always_inline)) {
                const M2 later = rec_m & ~m2_first(p + 1);               // recoveries after this pose
                const int rl = m2_lowest(later);
                const bool in_run = active && !av && rl < 128 && m2_bit(rts_m, rl < 128 ? rl : 0);
                const int rr_ = rl < 128 ? rl : 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double dr = pos_gather(dcA[c], dcB[c], rr_), pr = pos_gather(PmA[c], PmB[c], rr_);
                    if (in_run) xo[c] = xl[c] + Pf[c] * fast_rcp(pr) * dr;
                }
            };
            smooth(2 * lane, actA, __builtin_amdgcn_inverse_ballot_w64(avm.a), xA, PfA, xoA);
            smooth(2 * lane + 1, actB, __builtin_amdgcn_inverse_ballot_w64(avm.b), xB, PfB, xoB);
            // an outage carried in from earlier chunks and closed here by an RTS recovery: fix the rows already written
            if (!c_prev_avail) {
                const int r1 = m2_lowest(rec_m);                         // the first recovery of the chunk closes the carried run
                if (m2_bit(rts_m, r1)) {
                    const double dr[3] = { pos_bcast(dcA[0], dcB[0], r1), pos_bcast(dcA[1], dcB[1], r1), pos_bcast(dcA[2], dcB[2], r1) };
                    const double ipr[3] = { fast_rcp(pos_bcast(PmA[0], PmB[0], r1)), fast_rcp(pos_bcast(PmA[1], PmB[1], r1)), fast_rcp(pos_bcast(PmA[2], PmB[2], r1)) };
                    // the last two chunks of the run come from the lane-private ring; older ones from memory (stamps re-scanned)
                    const int64_t kfirst = (c_ostart / 128) * 128, kring = (c0 - 256 > kfirst) ? c0 - 256 : kfirst;
                    double acc = 0.0;                                    // sum of dt over (ostart, k]
                    for (int64_t k0 = kfirst; k0 < kring; k0 += 64) {
                        const int64_t k = k0 + lane;
                        const double tk = tsb[k];
                        const double tkp = prev_lane((k0 > 0) ? tsb[k0 - 1] : tk, tk);
                        double dsum = (k > c_ostart) ? fmax(1e-6, tk - tkp) : 0.0;
#define GSF_SSTAGE2(CTRL, RM) { dsum += dpp0<CTRL, RM>(dsum); }
                        GSF_SCAN_STAGES(GSF_SSTAGE2)
#undef GSF_SSTAGE2
                        const double tot = lane_bcast(dsum, 63);
                        if (k >= c_ostart) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) pob[k * 3 + c] += (cPos[c] + cfg.Qps[c] * (acc + dsum)) * ipr[c] * dr[c];
                        }
                        acc += tot;
                    }
                    for (int64_t k0 = kring; k0 < c0; k0 += 128) {
                        const int par = (int)(k0 >> 7) & 1;
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl) {
                            const int64_t k = k0 + 2 * lane + sl;
                            if (k >= c_ostart) {
#pragma unroll
                                for (int c = 0; c < 3; ++c)
                                    __builtin_nontemporal_store(ring2[par][c][sl][lane] + ring2[par][3 + c][sl][lane] * ipr[c] * dr[c], &pob[k * 3 + c]);
                            }
                        }
                    }
                }
            }
        }

        // ---- output rows and the carry to the next 128 poses (from the last active position Lp)
        const double oA[3] = { cp.x + xoA[0], cp.y + xoA[1], cp.z + xoA[2] }, oB[3] = { cp.x + xoB[0], cp.y + xoB[1], cp.z + xoB[2] };
        const bool open = !m2_bit(a_mask, Lp);                           // the chunk ends inside an outage
        W2_STAMP(13);
        if (open) {
            const int par = (int)(c0 >> 7) & 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ring2[par][c][0][lane] = oA[c]; ring2[par][c][1][lane] = oB[c];
                ring2[par][3 + c][0][lane] = PfA[c]; ring2[par][3 + c][1][lane] = PfB[c];
            }
            const int s = m2_last(start_m & m2_first(Lp + 1));
            if (s >= 0) {
                c_ostart = c0 + s;
                c_seg_sharp = m2_any(f_m & m2_range(s + 1, Lp));
                cPos[0] = pos_bcast(PfA[0], PfB[0], s); cPos[1] = pos_bcast(PfA[1], PfB[1], s); cPos[2] = pos_bcast(PfA[2], PfB[2], s);
            } else {
                c_seg_sharp = c_seg_sharp || m2_any(f_m & m2_first(Lp + 1));
            }
        }
        c_prev_avail = !open;
        if (!telescope) {
            cq = Quat{ pos_bcast(qiA.x, qiB.x, Lp), pos_bcast(qiA.y, qiB.y, Lp), pos_bcast(qiA.z, qiB.z, Lp), pos_bcast(qiA.w, qiB.w, Lp) };
            const Quat rL{ pos_bcast(rA.x, rB.x, Lp), pos_bcast(rA.y, rB.y, Lp), pos_bcast(rA.z, rB.z, Lp), pos_bcast(rA.w, rB.w, Lp) };
            if (m2_bit(okm, Lp)) Cq = lane_bcast(quat_mul(cq, quat_conj(rL)), 0);
        }
        cp = Vec3{ cp.x + pos_bcast(xA[0], xB[0], Lp), cp.y + pos_bcast(xA[1], xB[1], Lp), cp.z + pos_bcast(xA[2], xB[2], Lp) };
        cP[0] = pos_bcast(PfA[0], PfB[0], Lp); cP[1] = cP[0]; cP[2] = pos_bcast(PfA[2], PfB[2], Lp);
        c_po = Vec3{ pos_bcast(pA.x, pB.x, Lp), pos_bcast(pA.y, pB.y, Lp), pos_bcast(pA.z, pB.z, Lp) };
        c_r = Quat{ pos_bcast(rA.x, rB.x, Lp), pos_bcast(rA.y, rB.y, Lp), pos_bcast(rA.z, rB.z, Lp), pos_bcast(rA.w, rB.w, Lp) };
        c_ok = m2_bit(okm, Lp); c_t = pos_bcast(tA, tB, Lp);
        rows2_arrived(nxt);                                              // the next chunk's rows; then this chunk's stores
        W2_STAMP(14);
        // a lane's two rows are contiguous: 48 bytes of positions, 64 of quaternions
        const int64_t i0 = c0 + 2 * lane;
        if (actB) {
            w2_v2* ps = (w2_v2*)(pob + i0 * 3); w2_v2* qs = (w2_v2*)(qob + i0 * 4);
            __builtin_nontemporal_store(w2_v2{ oA[0], oA[1] }, ps); __builtin_nontemporal_store(w2_v2{ oA[2], oB[0] }, ps + 1); __builtin_nontemporal_store(w2_v2{ oB[1], oB[2] }, ps + 2);
            __builtin_nontemporal_store(w2_v2{ qiA.x, qiA.y }, qs); __builtin_nontemporal_store(w2_v2{ qiA.z, qiA.w }, qs + 1);
            __builtin_nontemporal_store(w2_v2{ qiB.x, qiB.y }, qs + 2); __builtin_nontemporal_store(w2_v2{ qiB.z, qiB.w }, qs + 3);
        } else if (actA) {
            __builtin_nontemporal_store(oA[0], &pob[i0 * 3]); __builtin_nontemporal_store(oA[1], &pob[i0 * 3 + 1]); __builtin_nontemporal_store(oA[2], &pob[i0 * 3 + 2]);
            __builtin_nontemporal_store(qiA.x, &qob[i0 * 4]); __builtin_nontemporal_store(qiA.y, &qob[i0 * 4 + 1]);
            __builtin_nontemporal_store(qiA.z, &qob[i0 * 4 + 2]); __builtin_nontemporal_store(qiA.w, &qob[i0 * 4 + 3]);
        }
        W2_STAMP(15);
    }
    GSF_STAMP_FLUSH();
    if (lane == 0 && GSF_STATUS_PTR(a)) a.status[b] = (status | (c_prev_avail ? 0 : ST_ENDED_IN_OUTAGE)) | (PIPELINE ? (fit << 8) : 0);
}

template <bool PIPELINE>
__device__ __forceinline__ void wave2_serial_body(const WaveArgs& a, const EkfConfig& cfg, const int64_t b, const int lane)
{
    GSF_STAMP(0);
    int64_t base, N; traj_span(a, b, base, N);
    if (N <= 0) { if (lane == 0 && GSF_STATUS_PTR(a)) a.status[b] = 0; return; }
    Vec3 p0; Quat q0; int32_t fit = 0;
    if (!wave_prelude<PIPELINE>(a, b, base, N, lane, p0, q0, fit)) return;
    GSF_STAMP(6);
    wave2_serial_chunks<PIPELINE>(a, cfg, b, lane, base, N, p0, q0, fit);
}

}  // namespace
