// gsf_ekf_block.hip -- K4 and the fused K2+K3+K4 pipeline with ONE WORKGROUP PER TRAJECTORY, one wave per 64-pose chunk.
//
// The wave-per-trajectory kernel (gsf_ekf_wave.hip) walks the chunks of a track one after the other and, in the fused pipeline,
// reads the positions / fixes / mask twice (fit pass, then filter pass).  Here every chunk of the track is owned by a wave and
// stays in that wave's REGISTERS from the first load to the last store: each input byte is read once (145 B/pose moved for
// 145 B/pose of algorithmic traffic) and the chunks of a track run concurrently.  apply_ekf_correction (EKFGPSSLAM.py:831-935)
// becomes a two-level scan:
//   level 1 (inside a wave, DPP)   the same prefix scans as gsf_wave_common.hpp, each started from the IDENTITY carry
//   level 2 (across chunks, LDS)   every chunk publishes its TOTAL (Moebius matrices of the variance recursion, affine maps of the
//                                  position recursion, outage ballots, first-recovery record); a chunk rebuilds its carry-in by
//                                  applying the totals of its predecessors in order -- the same operations in the same order as the
//                                  serial chunk loop, so variances and gains are the ones that kernel computes
// and compute_sim3_transform (:428-459) is one cross-wave reduction: per-chunk moment sums, transposed inside 16-lane rows, added up
// by wave 0, which runs the closed form once and publishes the initial pose.
//
//   barrier 0 (rare)   the GNSS-side shift of the moments when pose 0 has no usable fix (first usable fix of the track)
//   barrier 1          ballots, Moebius totals, moment sums        -> carry-in variances, gains, recovery decisions; wave 0: the fit
//   barrier 2          initial pose (pipeline only)                -> orientation, predicted displacements, affine scan
//   barrier 2b (rare)  quaternion-increment totals when the track holds an invalid quaternion (ref :84-86: no telescoping)
//   barrier 3          affine totals                               -> carry-in position, filtered positions, recovery records
//   barrier 4 (rare)   first-recovery records when an outage crosses a chunk boundary (per-outage RTS, ref :906-922) -> stores
//
// Residency (tools/ubench/residency.hip): a CU admits four 5-wave workgroups only at <= 80 registers per lane; at the 96 this kernel
// needs it holds three, i.e. 768 tracks of 271 poses at once -- the batch size up to which this kernel beats the wave-per-trajectory
// one.  A build in which one wave took the short last chunk as a second pass (four waves per 271-pose track, all
// 1 000 workgroups resident) was measured and is slower: the three other waves wait at every barrier for the double pass.
//
// Preconditions (checked by the launcher): trajectory-major layout, one N for the batch, 64 < N <= 1024.  Positions are carried
// relative to the initial position p0 (|x| <= track length), outputs are p0 + x.
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

constexpr int BLK_MAXC = 16;          // chunks per track

struct BlockShared {
    double mom[BLK_MAXC][64];         // per-chunk moment sums, one per lane (row_sums16_transposed)
    double Tm[BLK_MAXC][3][4];        // Moebius totals (A, B, C, D) per axis
    double Ta[BLK_MAXC][3][2];        // affine totals (alpha, beta) per axis
    double Tq[BLK_MAXC][4];           // quaternion-increment totals (generic orientation path only)
    double rec_d[BLK_MAXC][3], rec_pm[BLK_MAXC][3];
    double first_fix[BLK_MAXC][3];
    double fitv[16];                  // p0[3], cq0[4], Cq[4]; [12..14] the GNSS-side shift of the moments
    double stash[20][64];             // wave 0's per-lane state while it runs the fit (the closed form needs the registers)
    u64 a_mask[BLK_MAXC], start_mask[BLK_MAXC], f_mask[BLK_MAXC];
    int cnt[BLK_MAXC], has_fix[BLK_MAXC], bad[BLK_MAXC], rec_lane[BLK_MAXC], rec_rts[BLK_MAXC], status[BLK_MAXC];
    int fit, fit_ok;
};

// LDS-only barrier: the waves exchange LDS words only, and the streaming stores of a wave must not be waited for here
__device__ __forceinline__ void block_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ u64 uniform_u64(u64 v) { return (u64)uniform64((int64_t)v); }
// quat_unit()'s acceptance test alone
__device__ __forceinline__ bool quat_norm_ok(const Quat& q)
{
    const double n2 = quat_norm2(q);                                      // the same fused chain as quat_unit: the two cannot disagree at the thresholds
    return (n2 >= 1e-280) && (n2 <= 1e280);
}

#ifdef GSF_BLOCK_TIMING
// diagnostic build only (tools/block_timing.py): every wave keeps (100 MHz clock, shader clock) stamps of its phases and writes them
// over its trajectory's position rows at the end -- the outputs of this build are meaningless
#define BSTAMP(k) do { st_w[k] = wall_clock64(); st_c[k] = clock64(); } while (0)
#else
#define BSTAMP(k) do { } while (0)
#endif

// what every chunk of the block shares
struct BlockCtx {
    const double* __restrict__ tsb; const double* __restrict__ posb; const double* __restrict__ quatb; const double* __restrict__ gpsb;
    const uint8_t* __restrict__ valb;
    const uint8_t* __restrict__ rselb; // pipeline under the reference's row choice (gsf_set_sim3_rows mode 1): the rows of the fit, marked by sim3_rows_kernel; else NULL
    double* __restrict__ pob; double* __restrict__ qob;
    int64_t N; int lane, C;            // C = chunks of the track
    int same1, same2;                 // axis c repeats axis same_c (-1: scans of its own)
};

// One 64-pose chunk of the track: its rows and everything derived from them, from the loads to the stores.  All members live in
// registers (the object never has its address taken).
template <bool PIPELINE, bool INLINE_COLD>
struct Chunk {
    int vc;                           // chunk index (wave-uniform)
    int64_t c0, i, ip;
    bool active, is_init, stepping;
    int L;
    ChunkIn in;
    double c_t; Vec3 c_po; Quat c_qraw; double cz0, cz1, cz2; uint32_t c_vraw;
    // phase A results
    Vec3 d, z; Quat r; double dt;
    bool avail, av, recovers, both_ok, c_prev_avail, open_end, okf, fitrow;
    u64 a_mask, start_mask, rec_mask, f_mask, rts_mask, okf_mask;
    uint32_t rsel;
    Moebius M0, M1, M2;
    // after barrier 1
    double Pf[3], Pm[3], kg[3], wgt;
    int32_t status;
    // after barrier 2
    Quat qi; double uu[3], al[3], be[3];
    // after barrier 3
    double xl[3], dcorr[3];

    __device__ __forceinline__ void load(const BlockCtx& k, int vchunk)
    {
        vc = vchunk;
        c0 = (int64_t)vc * 64; i = c0 + k.lane;
        active = i < k.N; is_init = (i == 0); stepping = active && !is_init;
        L = (int)((k.N - c0 < 64) ? (k.N - c0 - 1) : 63);
        in = load_chunk(k.tsb, k.posb, k.quatb, k.gpsb, k.valb, i, k.N);
        ip = c0 > 0 ? c0 - 1 : 0;                                        // predecessor of lane 0 (wave-uniform address)
        c_t = k.tsb[ip];
        c_po = Vec3{ k.posb[ip * 3], k.posb[ip * 3 + 1], k.posb[ip * 3 + 2] };
        c_qraw = Quat{ k.quatb[ip * 4], k.quatb[ip * 4 + 1], k.quatb[ip * 4 + 2], k.quatb[ip * 4 + 3] };
        cz0 = k.gpsb[ip * 3]; cz1 = k.gpsb[ip * 3 + 1]; cz2 = k.gpsb[ip * 3 + 2];
        c_vraw = k.valb[ip];
        rsel = (PIPELINE && k.rselb) ? k.rselb[active ? i : 0] : 1u;    // ref :973-998, decided by the launcher's pre-pass
    }

    // everything that needs no other chunk: ballots of the outage structure, variance maps with the identity carry
    __device__ __forceinline__ void phase_a(const BlockCtx& k, const EkfConfig& cfg, BlockShared& sh, const bool r0ok)
    {
        const int lane = k.lane;
        const double t = in.t;
        const Vec3 p = in.p;
        z = in.z;
        const bool vraw = in.v != 0;
        const bool ok = quat_unit(in.q, r);
        // (the unit quaternions of pose c0-1 and of pose 0 are formed where they are used -- cold paths / the fit wave -- so that they
        // do not occupy registers in between; here only whether Rotation.from_quat would accept them)
        const bool c_ok = quat_norm_ok(c_qraw);
        // "gnss available" flag of pose c0-1: pose 0 keeps the raw mask (ref :848), every other pose is NaN-gated (ref :867-869)
        c_prev_avail = (ip == 0) ? (c_vraw != 0) : ((c_vraw != 0) && !(isnan(cz0) || isnan(cz1) || isnan(cz2)));
        const double t_pr = prev_lane(c_t, t);
        d = Vec3{ p.x - prev_lane(c_po.x, p.x), p.y - prev_lane(c_po.y, p.y), p.z - prev_lane(c_po.z, p.z) };   // p_i - p_{i-1}
        const u64 act_mask = __ballot(active);
        const u64 ok_mask = __ballot(ok);
        const bool ok_pr = (lane == 0) ? c_ok : (((ok_mask >> (lane - 1)) & 1ull) != 0ull);
        both_ok = ok_pr && ok;
        const bool wave_bad = !c_ok || !r0ok || ((ok_mask & act_mask) != act_mask);
        dt = fmax(1e-6, t - t_pr);                                       // ref :865
        const bool zfin = !(isnan(z.x) || isnan(z.y) || isnan(z.z));
        avail = stepping && vraw && zfin;                                // ref :867-869
        av = is_init ? vraw : avail;                                     // pose 0: raw mask, ref :848
        okf = active && vraw && zfin;                                    // rows with a usable fix
        okf_mask = __ballot(okf);
        fitrow = okf && rsel != 0;                                       // rows of the fit (ref :430-438; under mode 1 the reference's choice of them)
        a_mask = __ballot(active && av);
        const bool ap = (lane == 0) ? (is_init ? true : c_prev_avail) : (((a_mask >> (lane - 1)) & 1ull) != 0ull);
        const bool starts = active && !av && ap;                         // ref :875-877 (pose 0: :861)
        recovers = stepping && av && !ap;                                // ref :879
        const bool outpair = stepping && !av && !ap;
        start_mask = __ballot(starts); rec_mask = __ballot(recovers);
        const u64 pair_mask = __ballot(outpair);
        f_mask = 0ull;                                                   // is_sharp_turn_in_segment pairs, ref :808-826
        if (pair_mask != 0ull) {
            Quat c_r; quat_unit(c_qraw, c_r);
            const Quat r_pr = prev_lane(c_r, r);
            bool f = false;
            if (outpair && t > t_pr) f = !both_ok || (INLINE_COLD ? yaw_rate_exceeds_body(r_pr, r, t - t_pr, cfg.yaw_thr_rad) : yaw_rate_exceeds_poly(r_pr, r, t - t_pr, cfg.yaw_thr_rad));
            f_mask = __ballot(f);
        }
        open_end = ((a_mask >> L) & 1ull) == 0ull;                       // the chunk ends inside an outage
        // variance maps of the chunk, identity carry (ref :712-713, :723-731); axes with identical (P0, Q, R) share the scan
        M0 = variance_scan(cfg.Qps[0], cfg.Rm[0], dt, stepping, avail);
        M1 = M0; M2 = M0;
        if (k.same1 != 0) M1 = variance_scan(cfg.Qps[1], cfg.Rm[1], dt, stepping, avail);
        if (k.same2 == 1) M2 = M1; else if (k.same2 != 0) M2 = variance_scan(cfg.Qps[2], cfg.Rm[2], dt, stepping, avail);
        if (lane == 63) {
            sh.Tm[vc][0][0] = M0.A; sh.Tm[vc][0][1] = M0.B; sh.Tm[vc][0][2] = M0.C; sh.Tm[vc][0][3] = M0.D;
            if (k.same1 != 0) { sh.Tm[vc][1][0] = M1.A; sh.Tm[vc][1][1] = M1.B; sh.Tm[vc][1][2] = M1.C; sh.Tm[vc][1][3] = M1.D; }
            if (k.same2 < 0) { sh.Tm[vc][2][0] = M2.A; sh.Tm[vc][2][1] = M2.B; sh.Tm[vc][2][2] = M2.C; sh.Tm[vc][2][3] = M2.D; }
        }
        if (lane == 0) { sh.a_mask[vc] = a_mask; sh.start_mask[vc] = start_mask; sh.f_mask[vc] = f_mask; sh.bad[vc] = wave_bad ? 1 : 0; }
    }

    // pipeline, rare: this chunk's first usable fix as a candidate for the GNSS-side shift of the moments
    __device__ __forceinline__ void publish_first_fix(const BlockCtx& k, BlockShared& sh)
    {
        const int f = okf_mask != 0ull ? __ffsll((long long)okf_mask) - 1 : 0;
        const double f0 = lane_bcast(z.x, f), f1 = lane_bcast(z.y, f), f2 = lane_bcast(z.z, f);
        if (k.lane == 0) { sh.has_fix[vc] = okf_mask != 0ull; sh.first_fix[vc][0] = f0; sh.first_fix[vc][1] = f1; sh.first_fix[vc][2] = f2; }
    }
    // pipeline: K2 moments of the rows with valid finite GNSS (ref :430-438), shifted by pose 0 / a usable fix of the track
    __device__ __forceinline__ void moments(const BlockCtx& k, BlockShared& sh, const double as0, const double as1, const double as2,
                                            const double bs0, const double bs1, const double bs2)
    {
        const Vec3 p = in.p;
        const double a0 = fitrow ? p.x - as0 : 0.0, a1 = fitrow ? p.y - as1 : 0.0, a2 = fitrow ? p.z - as2 : 0.0;
        const double b0 = fitrow ? z.x - bs0 : 0.0, b1 = fitrow ? z.y - bs1 : 0.0, b2 = fitrow ? z.z - bs2 : 0.0;
        const double zrow = row_sums16_transposed(a0, a1, a2, b0, b1, b2, a0 * a0 + a1 * a1 + a2 * a2, a0 * b0, a0 * b1, a0 * b2,
                                                  a1 * b0, a1 * b1, a1 * b2, a2 * b0, a2 * b1, a2 * b2, k.lane);
        sh.mom[vc][k.lane] = zrow;
        const u64 fit_mask = __ballot(fitrow);
        if (k.lane == 0) sh.cnt[vc] = __popcll(fit_mask);
    }

    // after barrier 1: outage state carried in, recovery decisions, carry-in variances, gains
    __device__ __forceinline__ void gains(const BlockCtx& k, const EkfConfig& cfg, BlockShared& sh)
    {
        const int lane = k.lane;
        // outage state carried into this chunk, rebuilt from the predecessors' ballots (ref :859-862, :875-877)
        int64_t c_ostart = 0; bool c_seg_sharp = false;
        if (vc > 0 && !c_prev_avail) {
            for (int v = vc - 1; v >= 0; --v) {
                const u64 sm = uniform_u64(sh.start_mask[v]), fm = uniform_u64(sh.f_mask[v]);
                if (sm != 0ull) {
                    const int s = 63 - __clzll((long long)sm);
                    c_ostart = (int64_t)v * 64 + s;
                    c_seg_sharp = c_seg_sharp || (fm & bits(s + 1, 63)) != 0ull;
                    break;
                }
                c_seg_sharp = c_seg_sharp || fm != 0ull;
            }
        }
        status = (start_mask != 0ull) ? ST_HAD_OUTAGE : 0;
        bool sharp = false;
        if (rec_mask != 0ull && recovers) {                              // ref :879-894
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
        rts_mask = rec_mask & ~sharp_mask;                               // recoveries that run the RTS back-pass
        status |= ((sharp_mask != 0ull) ? ST_SHARP_TURN : 0) | ((rts_mask != 0ull) ? ST_RTS_APPLIED : 0);
        if (vc == k.C - 1 && open_end) status |= ST_ENDED_IN_OUTAGE;     // ref :932
        const double wgt_sharp = (cfg.sharp_turn_steps > 1) ? 1.0 / (double)cfg.sharp_turn_steps : 1.0;   // ref :752-768, Q7
        wgt = sharp ? wgt_sharp : 1.0;

        // carry-in variances: the predecessors' maps applied in order to P0, then P_f / P_p / gain of every pose
        double Pc0 = cfg.P0[0], Pc1 = cfg.P0[1], Pc2 = cfg.P0[2];
        for (int v = 0; v < vc; ++v) {
            Pc0 = moebius_apply(Moebius{ sh.Tm[v][0][0], sh.Tm[v][0][1], sh.Tm[v][0][2], sh.Tm[v][0][3] }, Pc0);
            if (k.same1 != 0) Pc1 = moebius_apply(Moebius{ sh.Tm[v][1][0], sh.Tm[v][1][1], sh.Tm[v][1][2], sh.Tm[v][1][3] }, Pc1);
            if (k.same2 < 0) Pc2 = moebius_apply(Moebius{ sh.Tm[v][2][0], sh.Tm[v][2][1], sh.Tm[v][2][2], sh.Tm[v][2][3] }, Pc2);
        }
        const AxisVar v0 = variance_finish(M0, cfg.Qps[0], cfg.Rm[0], dt, Pc0);
        AxisVar v1 = v0, v2 = v0;
        if (k.same1 != 0) v1 = variance_finish(M1, cfg.Qps[1], cfg.Rm[1], dt, Pc1);
        if (k.same2 == 1) v2 = v1; else if (k.same2 != 0) v2 = variance_finish(M2, cfg.Qps[2], cfg.Rm[2], dt, Pc2);
        Pf[0] = v0.Pf; Pf[1] = v1.Pf; Pf[2] = v2.Pf; Pm[0] = v0.Pm; Pm[1] = v1.Pm; Pm[2] = v2.Pm; kg[0] = v0.kg; kg[1] = v1.kg; kg[2] = v2.kg;
    }

    // generic orientation path, part 1 (block-uniform, rare): calculate_relative_pose with the zero-motion branch (ref :77-92), prefix
    // product of the increments; the chunk total goes through LDS
    Vec3 g_dpl; Quat g_D;
    __device__ __forceinline__ void generic_a(const BlockCtx& k, BlockShared& sh)
    {
        const bool move = stepping && both_ok;
        Quat c_r; quat_unit(Quat{ k.quatb[ip * 4], k.quatb[ip * 4 + 1], k.quatb[ip * 4 + 2], k.quatb[ip * 4 + 3] }, c_r);
        const Quat r1i = quat_conj(prev_lane(c_r, r));
        Vec3 dpl = quat_rotate(r1i, d);
        Quat D = quat_mul(r1i, r);
        dpl.x = move ? dpl.x : 0.0; dpl.y = move ? dpl.y : 0.0; dpl.z = move ? dpl.z : 0.0;
        D.x = move ? D.x : 0.0; D.y = move ? D.y : 0.0; D.z = move ? D.z : 0.0; D.w = move ? D.w : 1.0;
        if (__ballot(stepping && !both_ok) != 0ull) status |= ST_BAD_QUAT;
        const Quat QID{ 0.0, 0.0, 0.0, 1.0 };
#define GSF_QSTAGE(CTRL, RM) { const Quat o = dpp<CTRL, RM>(QID, D); D = quat_mul(o, D); }
        GSF_SCAN_STAGES(GSF_QSTAGE)
#undef GSF_QSTAGE
        if (k.lane == 63) { sh.Tq[vc][0] = D.x; sh.Tq[vc][1] = D.y; sh.Tq[vc][2] = D.z; sh.Tq[vc][3] = D.w; }
        g_dpl = dpl; g_D = D;
    }
    // part 2: carry-in orientation = q0 * T_0 * ... * T_{vc-1}
    __device__ __forceinline__ void generic_b(const BlockShared& sh, const Quat& cq0, Vec3& u)
    {
        Quat cq = cq0;
        for (int v = 0; v < vc; ++v) cq = quat_mul(cq, Quat{ sh.Tq[v][0], sh.Tq[v][1], sh.Tq[v][2], sh.Tq[v][3] });
        qi = ekf_normalize(quat_mul(cq, g_D));
        const Quat q_prev = prev_lane(ekf_normalize(cq), qi);
        u = quat_rotate(q_prev, g_dpl);
    }
    // telescoped orientation: every quaternion of the track is valid, so the increments telescope (see wave_serial_chunks) -- q_i = Cq r_i,
    // predicted displacement = R(Cq) (p_i - p_{i-1}), with ONE rotation Cq = q_0 conj(r_0) for the whole track
    __device__ __forceinline__ void telescoped(const Quat& cq0, const Quat& Cq, Vec3& u)
    {
        qi = quat_mul(Cq, r);
        qi.x = is_init ? cq0.x : qi.x; qi.y = is_init ? cq0.y : qi.y; qi.z = is_init ? cq0.z : qi.z; qi.w = is_init ? cq0.w : qi.w;
        u = quat_rotate(Cq, d);
        u.x = stepping ? u.x : 0.0; u.y = stepping ? u.y : 0.0; u.z = stepping ? u.z : 0.0;
    }

    // positions: prefix composition of the affine maps x -> al x + be in coordinates relative to p0 (ref :707, :727-728)
    __device__ __forceinline__ void affine(const BlockCtx& k, BlockShared& sh, const Vec3& p0, const Vec3& u)
    {
        uu[0] = u.x; uu[1] = u.y; uu[2] = u.z;
        const double zl[3] = { z.x - p0.x, z.y - p0.y, z.z - p0.z };
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double kw = kg[c] * wgt;
            al[c] = avail ? (1.0 - kw) : 1.0;
            be[c] = avail ? ((1.0 - kw) * uu[c] + kw * zl[c]) : uu[c];
        }
#define GSF_ASTAGE(c, CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al[c]), ob = dpp0<CTRL, RM>(be[c]); be[c] = al[c] * ob + be[c]; al[c] = al[c] * oa; }
#define GSF_ASTAGE_X(CTRL, RM) GSF_ASTAGE(0, CTRL, RM)
#define GSF_ASTAGE_Y(CTRL, RM) GSF_ASTAGE(1, CTRL, RM)
#define GSF_ASTAGE_Z(CTRL, RM) GSF_ASTAGE(2, CTRL, RM)
#define GSF_ASTAGE_XY(CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al[0]), ob0 = dpp0<CTRL, RM>(be[0]), ob1 = dpp0<CTRL, RM>(be[1]); \
                                  be[0] = al[0] * ob0 + be[0]; be[1] = al[0] * ob1 + be[1]; al[0] = al[0] * oa; }
        if (k.same1 == 0) { GSF_SCAN_STAGES(GSF_ASTAGE_XY) al[1] = al[0]; }
        else { GSF_SCAN_STAGES(GSF_ASTAGE_X) GSF_SCAN_STAGES(GSF_ASTAGE_Y) }
        GSF_SCAN_STAGES(GSF_ASTAGE_Z)
#undef GSF_ASTAGE_XY
#undef GSF_ASTAGE_Z
#undef GSF_ASTAGE_Y
#undef GSF_ASTAGE_X
#undef GSF_ASTAGE
        if (k.lane == 63) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { sh.Ta[vc][c][0] = al[c]; sh.Ta[vc][c][1] = be[c]; }
        }
        if (k.lane == 0) sh.status[vc] = status;
    }

    // after barrier 3: carry-in position, filtered positions, this chunk's first-recovery record
    __device__ __forceinline__ void positions(const BlockCtx& k, BlockShared& sh, const bool cross_rts)
    {
        double xc[3] = { 0.0, 0.0, 0.0 };                                // x_0 - p0 = 0
        for (int v = 0; v < vc; ++v) {
#pragma unroll
            for (int c = 0; c < 3; ++c) xc[c] = sh.Ta[v][c][0] * xc[c] + sh.Ta[v][c][1];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xl[c] = al[c] * xc[c] + be[c];
            dcorr[c] = xl[c] - (prev_lane(xc[c], xl[c]) + uu[c]);        // x_f[i] - x_p[i] (non-zero only where a fix was used)
        }
        if (cross_rts) {                                                 // block-uniform
            const int r1 = rec_mask != 0ull ? __ffsll((long long)rec_mask) - 1 : -1;
            const int rr = r1 >= 0 ? r1 : 0;
            const double d0 = lane_bcast(dcorr[0], rr), d1 = lane_bcast(dcorr[1], rr), d2 = lane_bcast(dcorr[2], rr);
            const double m0 = lane_bcast(Pm[0], rr), m1 = lane_bcast(Pm[1], rr), m2 = lane_bcast(Pm[2], rr);
            if (k.lane == 0) {
                sh.rec_lane[vc] = r1; sh.rec_rts[vc] = (r1 >= 0) ? (int)((rts_mask >> r1) & 1ull) : 0;
                sh.rec_d[vc][0] = d0; sh.rec_d[vc][1] = d1; sh.rec_d[vc][2] = d2;
                sh.rec_pm[vc][0] = m0; sh.rec_pm[vc][1] = m1; sh.rec_pm[vc][2] = m2;
            }
        }
    }

    // per-outage RTS (ref :906-922, :777-803) and the stores.  Inside an outage x_f = x_p and P_f = P_p, so the gain product telescopes:
    // x_s[k] = x_f[k] + (P_f[k] / P_p[r]) (x_f[r] - x_p[r]) for k in [start, r-1], r = the recovery pose (this chunk's, or the first
    // recovery of a later chunk for a run still open at the end).
    __device__ __forceinline__ void finish(const BlockCtx& k, const BlockShared& sh, const Vec3& p0)
    {
        const int lane = k.lane;
        double xo[3] = { xl[0], xl[1], xl[2] };                          // what is written out
        if (rts_mask != 0ull || (open_end && vc + 1 < k.C)) {
            const u64 later = rec_mask & ~bits(0, lane);                 // recoveries after this lane
            const int rl = later != 0ull ? __ffsll((long long)later) - 1 : 0;
            const bool in_local = active && !av && later != 0ull && (((rts_mask >> rl) & 1ull) != 0ull);
            int v2 = -1;
            if (open_end) for (int v = vc + 1; v < k.C; ++v) if (sh.rec_lane[v] >= 0) { v2 = v; break; }
            const int v2c = v2 >= 0 ? v2 : 0;
            const bool in_later = active && !av && later == 0ull && v2 >= 0 && sh.rec_rts[v2c] != 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double dr_l = shidx(dcorr[c], rl), pr_l = shidx(Pm[c], rl);
                if (in_local) xo[c] = xl[c] + Pf[c] * fast_rcp(pr_l) * dr_l;
                else if (in_later) xo[c] = xl[c] + Pf[c] * fast_rcp(sh.rec_pm[v2c][c]) * sh.rec_d[v2c][c];
            }
        }
        if (active) {
            __builtin_nontemporal_store(p0.x + xo[0], &k.pob[i * 3]); __builtin_nontemporal_store(p0.y + xo[1], &k.pob[i * 3 + 1]);
            __builtin_nontemporal_store(p0.z + xo[2], &k.pob[i * 3 + 2]);
            __builtin_nontemporal_store(qi.x, &k.qob[i * 4]); __builtin_nontemporal_store(qi.y, &k.qob[i * 4 + 1]);
            __builtin_nontemporal_store(qi.z, &k.qob[i * 4 + 2]); __builtin_nontemporal_store(qi.w, &k.qob[i * 4 + 3]);
        }
    }
};

// MAXT threads at most; OCC = waves per SIMD the register allocation must allow
template <bool PIPELINE, int AXMODE, int MAXT, int OCC, bool INLINE_COLD>
__global__ __launch_bounds__(MAXT, OCC) void ekf_block_kernel(WaveArgs a, EkfConfig cfg, const uint8_t* __restrict__ rowsel,
                                                              const int32_t* __restrict__ rows_status)
{
    __shared__ BlockShared sh;
#ifdef GSF_BLOCK_TIMING
    long long st_w[12], st_c[12];
    for (int k = 0; k < 12; ++k) { st_w[k] = 0; st_c[k] = 0; }
    { unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      st_w[10] = (long long)hw; st_c[10] = (long long)(xcc & 0xf); }
#endif
    BSTAMP(0);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const int64_t b = blockIdx.x, N = a.N, base = b * N;
    BlockCtx k;
    k.tsb = a.ts + base; k.posb = a.pos + base * 3; k.quatb = a.quat + base * 4; k.gpsb = a.gps + base * 3; k.valb = a.valid + base;
    k.pob = a.pos_out + base * 3; k.qob = a.quat_out + base * 4;
    k.rselb = (PIPELINE && rowsel) ? rowsel + base : nullptr;
    const int32_t rows_flag = (PIPELINE && rows_status) ? rows_status[b] : 0;   // SIM3_FLAG_FEW_ROWS / _ROWS_ALL / _ROWS_SEGMENT
    k.N = N; k.lane = lane; k.C = W;
    k.same1 = -1; k.same2 = -1;
    if (AXMODE == 1) k.same1 = 0;
    else {
        if (cfg.P0[1] == cfg.P0[0] && cfg.Qps[1] == cfg.Qps[0] && cfg.Rm[1] == cfg.Rm[0]) k.same1 = 0;
        if (cfg.P0[2] == cfg.P0[0] && cfg.Qps[2] == cfg.Qps[0] && cfg.Rm[2] == cfg.Rm[0]) k.same2 = 0;
        else if (cfg.P0[2] == cfg.P0[1] && cfg.Qps[2] == cfg.Qps[1] && cfg.Rm[2] == cfg.Rm[1]) k.same2 = 1;
    }

    // ------------------------------------------------------------------ loads: own chunk, the pose before it, pose 0
    Chunk<PIPELINE, INLINE_COLD> c1;
    c1.load(k, w);
    const Quat qraw0{ k.quatb[0], k.quatb[1], k.quatb[2], k.quatb[3] };
    const double as0 = k.posb[0], as1 = k.posb[1], as2 = k.posb[2];      // pipeline: source-side shift of the moments = pose 0
    const double z00 = k.gpsb[0], z01 = k.gpsb[1], z02 = k.gpsb[2];
    const uint32_t v00 = k.valb[0];
    Vec3 p0; Quat q0;
    if (!PIPELINE) {
        p0 = Vec3{ a.init_pos[b * 3], a.init_pos[b * 3 + 1], a.init_pos[b * 3 + 2] };
        q0 = Quat{ a.init_quat[b * 4], a.init_quat[b * 4 + 1], a.init_quat[b * 4 + 2], a.init_quat[b * 4 + 3] };
    }

    // ------------------------------------------------------------------ phase A
#ifdef GSF_BLOCK_TIMING
    chunk_arrived(c1.in);
#endif
    BSTAMP(1);
    const bool r0ok = quat_norm_ok(qraw0);
    c1.phase_a(k, cfg, sh, r0ok);
    if (PIPELINE) {
        double bs0 = z00, bs1 = z01, bs2 = z02;                          // GNSS-side shift of the moments (the same in every wave)
        const bool okf0 = (v00 != 0) && !(isnan(z00) || isnan(z01) || isnan(z02));
        if (!okf0) {                                                     // block-uniform, rare: the first usable fix of the track
            c1.publish_first_fix(k, sh);
            block_barrier();                                             // ---- barrier 0
            bs0 = 0.0; bs1 = 0.0; bs2 = 0.0;
            for (int v = k.C - 1; v >= 0; --v)
                if (sh.has_fix[v]) { bs0 = sh.first_fix[v][0]; bs1 = sh.first_fix[v][1]; bs2 = sh.first_fix[v][2]; }
        }
        c1.moments(k, sh, as0, as1, as2, bs0, bs1, bs2);
        if (w == 0 && lane == 0) { sh.fitv[12] = bs0; sh.fitv[13] = bs1; sh.fitv[14] = bs2; }
    }
    BSTAMP(2);
    block_barrier();                                                     // ---- barrier 1
    BSTAMP(3);

    // ------------------------------------------------------------------ after barrier 1: block-uniform flags from the ballots
    int anybad_l = 0; u64 am_l = ~0ull;
    if (lane < k.C) { anybad_l = sh.bad[lane]; am_l = sh.a_mask[lane]; }
    const bool generic = __ballot(anybad_l != 0) != 0ull;                // some quaternion of the track is invalid
    // an outage crosses a chunk boundary: the last pose of some chunk but the last is unavailable
    const bool cross_rts = __ballot(lane < k.C - 1 && ((am_l >> 63) & 1ull) == 0ull) != 0ull;
    c1.gains(k, cfg, sh);
    BSTAMP(4);

    // ------------------------------------------------------------------ the fit (wave 0), ref :439-451, :464-466
    int32_t fit = 0;
    Quat cq0, Cq;
    if (PIPELINE) {
        if (w == 0) {
            // per-lane state that must survive the fit goes to LDS and comes back afterwards: the closed form alone takes ~90-120
            // registers, and several waves per SIMD have to fit
            constexpr bool S1 = (AXMODE != 1);                           // axis 1 has values of its own
            {
                double* st = &sh.stash[0][lane];
                st[0 * 64] = c1.d.x; st[1 * 64] = c1.d.y; st[2 * 64] = c1.d.z; st[3 * 64] = c1.r.x; st[4 * 64] = c1.r.y; st[5 * 64] = c1.r.z; st[6 * 64] = c1.r.w;
                st[7 * 64] = c1.z.x; st[8 * 64] = c1.z.y; st[9 * 64] = c1.z.z; st[10 * 64] = c1.wgt;
                st[11 * 64] = c1.Pf[0]; st[12 * 64] = c1.Pm[0]; st[13 * 64] = c1.kg[0]; st[14 * 64] = c1.Pf[2]; st[15 * 64] = c1.Pm[2]; st[16 * 64] = c1.kg[2];
                if (S1) { st[17 * 64] = c1.Pf[1]; st[18 * 64] = c1.Pm[1]; st[19 * 64] = c1.kg[1]; }
            }
            asm volatile("" ::: "memory");
            double zs = 0.0; int cn = 0;
            for (int v = 0; v < k.C; ++v) { zs += sh.mom[v][lane]; cn += sh.cnt[v]; }
            const double n = (double)cn;
            const double bsx = sh.fitv[12], bsy = sh.fitv[13], bsz = sh.fitv[14];
            const double a0 = k.posb[0], a1 = k.posb[1], a2 = k.posb[2];  // (read again: cheaper than keeping them in registers)
            double Rb[9], tb[3], sb = NAN;
            fit = SIM3_NONE;
            if (n >= 3.0 && !(rows_flag & SIM3_FLAG_FEW_ROWS)) {          // ref :430 (and :975 / :997: the reference raised before it got here)
                const Sums16 S = sums16_finish(zs);
                const double rn = fast_rcp(n);
                const double ma[3] = { S.v[0] * rn, S.v[1] * rn, S.v[2] * rn };
                const double mb[3] = { S.v[3] * rn, S.v[4] * rn, S.v[5] * rn };
                const double ssq = fmax(0.0, S.v[6] - n * (ma[0] * ma[0] + ma[1] * ma[1] + ma[2] * ma[2]));
                double H[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) H[j] = S.v[7 + j] - n * ma[j / 3] * mb[j % 3];
                const double sc[3] = { a0 + ma[0], a1 + ma[1], a2 + ma[2] }, dc[3] = { bsx + mb[0], bsy + mb[1], bsz + mb[2] };
                fit = umeyama_finalize<true, true>(H, ssq, sc, dc, n, Rb, tb, sb);
            }
            const bool good = fit != SIM3_NONE && r0ok;
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 9; ++j) a.R[b * 9 + j] = good ? Rb[j] : NAN;
                a.t[b * 3] = good ? tb[0] : NAN; a.t[b * 3 + 1] = good ? tb[1] : NAN; a.t[b * 3 + 2] = good ? tb[2] : NAN;
                a.s[b] = good ? sb : NAN;
            }
            const Vec3 pp{ sb * (a0 * Rb[0] + a1 * Rb[1] + a2 * Rb[2]) + tb[0], sb * (a0 * Rb[3] + a1 * Rb[4] + a2 * Rb[5]) + tb[1],
                           sb * (a0 * Rb[6] + a1 * Rb[7] + a2 * Rb[8]) + tb[2] };                          // ref :464
            Quat r0; quat_unit(Quat{ k.quatb[0], k.quatb[1], k.quatb[2], k.quatb[3] }, r0);
            const Quat qq = quat_mul(quat_from_matrix(Rb), r0);                                            // ref :465-466
            const Quat c0q = ekf_normalize(qq);                                                            // ref :842, :683
            const Quat Cqq = quat_mul(c0q, quat_conj(r0));
            if (lane == 0) {
                sh.fitv[0] = pp.x; sh.fitv[1] = pp.y; sh.fitv[2] = pp.z;
                sh.fitv[3] = c0q.x; sh.fitv[4] = c0q.y; sh.fitv[5] = c0q.z; sh.fitv[6] = c0q.w;
                sh.fitv[7] = Cqq.x; sh.fitv[8] = Cqq.y; sh.fitv[9] = Cqq.z; sh.fitv[10] = Cqq.w;
                sh.fit = fit; sh.fit_ok = good ? 1 : 0;
            }
            asm volatile("" ::: "memory");
            {
                const double* st = &sh.stash[0][lane];
                c1.d = Vec3{ st[0 * 64], st[1 * 64], st[2 * 64] }; c1.r = Quat{ st[3 * 64], st[4 * 64], st[5 * 64], st[6 * 64] };
                c1.z = Vec3{ st[7 * 64], st[8 * 64], st[9 * 64] }; c1.wgt = st[10 * 64];
                c1.Pf[0] = st[11 * 64]; c1.Pm[0] = st[12 * 64]; c1.kg[0] = st[13 * 64]; c1.Pf[2] = st[14 * 64]; c1.Pm[2] = st[15 * 64]; c1.kg[2] = st[16 * 64];
                if (S1) { c1.Pf[1] = st[17 * 64]; c1.Pm[1] = st[18 * 64]; c1.kg[1] = st[19 * 64]; }
                else { c1.Pf[1] = c1.Pf[0]; c1.Pm[1] = c1.Pm[0]; c1.kg[1] = c1.kg[0]; }
            }
        }
        BSTAMP(5);
        block_barrier();                                                 // ---- barrier 2
        BSTAMP(6);
        fit = sh.fit;
        if (!sh.fit_ok) {                                                // block-uniform: no fit, or SciPy would raise on pose 0's quaternion
            if (c1.active) {
                k.pob[c1.i * 3] = NAN; k.pob[c1.i * 3 + 1] = NAN; k.pob[c1.i * 3 + 2] = NAN;
                k.qob[c1.i * 4] = NAN; k.qob[c1.i * 4 + 1] = NAN; k.qob[c1.i * 4 + 2] = NAN; k.qob[c1.i * 4 + 3] = NAN;
            }
            if (threadIdx.x == 0 && a.status) a.status[b] = (fit == SIM3_NONE ? ((SIM3_NONE | (rows_flag & SIM3_FLAG_FEW_ROWS)) << 8) : 0) | (r0ok ? 0 : ST_BAD_QUAT);
            return;
        }
        p0 = Vec3{ sh.fitv[0], sh.fitv[1], sh.fitv[2] };
        cq0 = Quat{ sh.fitv[3], sh.fitv[4], sh.fitv[5], sh.fitv[6] };
        Cq = Quat{ sh.fitv[7], sh.fitv[8], sh.fitv[9], sh.fitv[10] };
    } else {
        Quat r0; quat_unit(qraw0, r0);
        cq0 = ekf_normalize(q0);                                         // ref :842, :683
        Cq = quat_mul(cq0, quat_conj(r0));
    }

    // ------------------------------------------------------------------ phase B: orientation, displacements, affine scan
    Vec3 u1;
    if (!generic) {
        c1.telescoped(cq0, Cq, u1);
    } else {
        c1.generic_a(k, sh);
        block_barrier();                                                 // ---- barrier 2b (block-uniform)
        c1.generic_b(sh, cq0, u1);
    }
    c1.affine(k, sh, p0, u1);
    BSTAMP(7);
    block_barrier();                                                     // ---- barrier 3
    BSTAMP(8);

    // ------------------------------------------------------------------ phase C: carry-in position, filtered positions
    c1.positions(k, sh, cross_rts);
    if (w == 0 && lane == 0 && a.status) {
        int32_t st = 0;
        for (int v = 0; v < k.C; ++v) st |= sh.status[v];
        a.status[b] = st | (PIPELINE ? ((fit | rows_flag) << 8) : 0);
    }
    if (cross_rts) block_barrier();                                      // ---- barrier 4 (block-uniform)
    BSTAMP(9);
#ifdef GSF_BLOCK_TIMING
    if (lane == 0) {
        long long* dbg = (long long*)k.pob + (int64_t)w * 24;
        for (int j = 0; j < 12; ++j) { dbg[j] = st_w[j]; dbg[12 + j] = st_c[j]; }
    }
    return;
#endif
    // ------------------------------------------------------------------ phase D: per-outage RTS, stores
    c1.finish(k, sh, p0);
}

EkfConfig to_core_block(const gsf_ekf_config* c)
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

bool ekf_block_applies(int64_t N, const int64_t* offsets) { return !offsets && N > 64 && N <= 1024; }

int launch_ekf_block(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                     const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                     int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status)
{
    GSF_REQUIRE(B <= 0x7fffffff && ekf_block_applies(N, nullptr), "launch_ekf_block: needs 64 < N <= 1024");
    WaveArgs a{ ts, pos, quat, gps, valid, init_pos, init_quat, R, t, s, pos_out, quat_out, status, B, N, nullptr, FitRows{ 0, 0, 0.0, 0.0 } };
    const EkfConfig k = to_core_block(cfg);
    const bool xy = k.P0[1] == k.P0[0] && k.Qps[1] == k.Qps[0] && k.Rm[1] == k.Rm[0] &&
                    !(k.P0[2] == k.P0[0] && k.Qps[2] == k.Qps[0] && k.Rm[2] == k.Rm[0]);
    const int W = (int)((N + 63) / 64);
    const dim3 grid((unsigned)B), block((unsigned)(W * 64));
    // the fit under the reference's row choice (gsf_set_sim3_rows mode 1, ref :973-998): the rows are marked by a launch of their own (the
    // rule is a serial walk over the valid rows of a track; here every chunk has its own wave) and the moments take the marked rows only
    const uint8_t* rowsel = nullptr; const int32_t* rows_status = nullptr;
    if (pipeline && ctx->fit_rows.mode != 0) {
        const size_t P = (size_t)B * (size_t)N, o_st = (P + 255) & ~(size_t)255, o_n = o_st + (((size_t)B * 4 + 255) & ~(size_t)255);
        int rc = ensure_rows_scratch(ctx, o_n + (size_t)B * 4);
        if (rc) return rc;
        char* w = (char*)ctx->rows_scratch;
        if ((rc = launch_sim3_rows(ctx, ts, gps, valid, nullptr, B, N, ctx->fit_rows, (uint8_t*)w, (int32_t*)(w + o_n), (int32_t*)(w + o_st)))) return rc;
        rowsel = (const uint8_t*)w; rows_status = (const int32_t*)(w + o_st);
    }
#define GSF_LAUNCH_BLOCK(P_, X_, T_, O_, I_) hipLaunchKernelGGL((ekf_block_kernel<P_, X_, T_, O_, I_>), grid, block, 0, ctx->stream, a, k, rowsel, rows_status)
#define GSF_LAUNCH_BLOCK_T(P_, X_) do { \
        if (W <= 5) GSF_LAUNCH_BLOCK(P_, X_, 320, 5, true); \
        else if (W <= 8) GSF_LAUNCH_BLOCK(P_, X_, 512, 4, true); \
        else GSF_LAUNCH_BLOCK(P_, X_, 1024, 4, false); } while (0)
    if (pipeline) { if (xy) GSF_LAUNCH_BLOCK_T(true, 1); else GSF_LAUNCH_BLOCK_T(true, 0); }
    else { if (xy) GSF_LAUNCH_BLOCK_T(false, 1); else GSF_LAUNCH_BLOCK_T(false, 0); }
#undef GSF_LAUNCH_BLOCK_T
#undef GSF_LAUNCH_BLOCK
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace gsf

namespace gsf { const char* wave_block_build_info() { return GSF_TU_BUILD_INFO("gsf_ekf_block.hip"); } }
