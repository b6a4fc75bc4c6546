// gsf_sim3.hip -- K2 Umeyama fit, K2b RANSAC wrapper, K3 apply-Sim3.
//   compute_sim3_transform        EKFGPSSLAM.py:428-459
//   compute_sim3_transform_robust EKFGPSSLAM.py:389-426
//   transform_trajectory          EKFGPSSLAM.py:461-467
//
// K2: the fit is a reduction (centroids, then the 3x3 cross-covariance of the CENTRED points --
// raw moments would cancel catastrophically at UTM magnitudes ~5e6 m) followed by a 3x3 SVD.
// One wavefront reduces one point set with DPP-routed scans (no LDS, no atomics); the SVD +
// closed form then runs LANE-PARALLEL: a 64-thread block owns 64 point sets, wave-reduces them one
// after another and parks the 19 moments of set k in lane k's registers, so the ~500-instruction
// Jacobi SVD is paid once per 64 sets instead of once per set.
// K2b: one 256-thread block per point set; one hypothesis per thread (4-point fit entirely in
// registers), all hypotheses score the same points so the reads are wave-broadcasts out of L1/L2.
#include "gsf_wave_common.hpp"      // DPP-routed wave_sum()
#include "gsf_ransac.hpp"           // the hypothesis fit, residual and threshold test (shared with gsf_robust.hip)

using namespace gsf;

namespace {

struct Moments { double n, sc[3], dc[3], H[9], ssq; };

// Wave-cooperative two-pass moments of rows [i0,i1) (optionally masked).  All lanes return the sums.
__device__ __forceinline__ Moments wave_moments(const double* __restrict__ src, const double* __restrict__ dst,
                                                const uint8_t* __restrict__ mask, int64_t i0, int64_t i1, int lane)
{
    Moments m;
    double cnt = 0.0, s0 = 0, s1 = 0, s2 = 0, d0 = 0, d1 = 0, d2 = 0;
    for (int64_t i = i0 + lane; i < i1; i += 64) {
        if (mask && !mask[i]) continue;
        cnt += 1.0;
        s0 += src[i * 3]; s1 += src[i * 3 + 1]; s2 += src[i * 3 + 2];
        d0 += dst[i * 3]; d1 += dst[i * 3 + 1]; d2 += dst[i * 3 + 2];
    }
    m.n = wave_sum(cnt);
    const double rn = 1.0 / m.n;
    m.sc[0] = wave_sum(s0) * rn; m.sc[1] = wave_sum(s1) * rn; m.sc[2] = wave_sum(s2) * rn;
    m.dc[0] = wave_sum(d0) * rn; m.dc[1] = wave_sum(d1) * rn; m.dc[2] = wave_sum(d2) * rn;
    double H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, ssq = 0.0;
    if (m.n >= 1.0) {
        for (int64_t i = i0 + lane; i < i1; i += 64) {
            if (mask && !mask[i]) continue;
            const double a0 = src[i * 3] - m.sc[0], a1 = src[i * 3 + 1] - m.sc[1], a2 = src[i * 3 + 2] - m.sc[2];
            const double b0 = dst[i * 3] - m.dc[0], b1 = dst[i * 3 + 1] - m.dc[1], b2 = dst[i * 3 + 2] - m.dc[2];
            H[0] += a0 * b0; H[1] += a0 * b1; H[2] += a0 * b2;
            H[3] += a1 * b0; H[4] += a1 * b1; H[5] += a1 * b2;
            H[6] += a2 * b0; H[7] += a2 * b1; H[8] += a2 * b2;
            ssq += a0 * a0 + a1 * a1 + a2 * a2;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) m.H[k] = wave_sum(H[k]);
    m.ssq = wave_sum(ssq);
    return m;
}

// Shifted raw moments of one point set, as accumulated by the short-set path: n, sum a, sum b, sum |a|^2, sum a b^T with
// a = src - src_shift, b = dst - dst_shift (the shift is a row of the set itself, so |a|, |b| <= the set's extent and
// H = Sab - n ma mb^T loses nothing at UTM magnitudes).
struct RawMoments { double n, Sa[3], Sb[3], Saa, Sab[9], as[3], bs[3]; };

__device__ __forceinline__ int32_t finalize_raw(const RawMoments& m, double* R, double* t, double& s)
{
    if (!(m.n >= 3.0)) return SIM3_NONE;                                     // ref :430
    const double rn = 1.0 / m.n;
    const double ma[3] = { m.Sa[0] * rn, m.Sa[1] * rn, m.Sa[2] * rn }, mb[3] = { m.Sb[0] * rn, m.Sb[1] * rn, m.Sb[2] * rn };
    double H[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = m.Sab[k] - m.n * ma[k / 3] * mb[k % 3];
    const double ssq = fmax(0.0, m.Saa - m.n * (ma[0] * ma[0] + ma[1] * ma[1] + ma[2] * ma[2]));
    const double sc[3] = { m.as[0] + ma[0], m.as[1] + ma[1], m.as[2] + ma[2] }, dc[3] = { m.bs[0] + mb[0], m.bs[1] + mb[1], m.bs[2] + mb[2] };
    return umeyama_finalize(H, ssq, sc, dc, m.n, R, t, s);
}

// sum over the 16 lanes of a DPP row (inclusive scan, total in lane 15 of each row); bound_ctrl zero-fills: no identity moves
__device__ __forceinline__ double row16_scan_sum(double v)
{
    v += dpp0<DPP_ROW_SHR1, 0xf>(v); v += dpp0<DPP_ROW_SHR2, 0xf>(v); v += dpp0<DPP_ROW_SHR4, 0xf>(v); v += dpp0<DPP_ROW_SHR8, 0xf>(v);
    return v;
}
__device__ __forceinline__ int64_t shfl64(int64_t v, int src)
{
    const int lo = __shfl((int)v, src, 64), hi = __shfl((int)(v >> 32), src, 64);
    return (int64_t)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// 64 point sets per block, SVD + closed form lane-parallel (lane k finishes set k).  The moments are reduced in one of two ways:
//   * short sets (every set of the block <= 256 rows -- config C4's 50-pair windows): FOUR sets per pass, one per 16-lane DPP
//     row; the 17 sums need only the four row_shr stages (no identity moves, no cross-row stages) and the rows of a set are
//     read once (shifted raw moments).  ~120 wave-instructions per set instead of ~560.
//   * long sets: the whole wave reduces one set after the other (two-pass centred moments).
__global__ __launch_bounds__(64) void umeyama_batch_kernel(const double* __restrict__ src, const double* __restrict__ dst,
                                                           const uint8_t* __restrict__ mask, const int64_t* __restrict__ offsets,
                                                           int64_t B, double* __restrict__ R, double* __restrict__ t,
                                                           double* __restrict__ s, int32_t* __restrict__ status)
{
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int nsets = (int)((B - b0 < 64) ? (B - b0) : 64);
    // lane k holds the bounds of set k
    const int64_t my_i0 = offsets[b0 + (lane < nsets ? lane : nsets - 1)], my_i1 = offsets[b0 + (lane < nsets ? lane : nsets - 1) + 1];
    const int64_t my_len = lane < nsets ? my_i1 - my_i0 : 0;
    const bool all_short = __ballot(my_len > 256) == 0ull;
    double Rb[9], tb[3], sb = NAN; int32_t st;
    if (all_short) {
        RawMoments mine;
        mine.n = 0.0; mine.Saa = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { mine.Sa[k] = mine.Sb[k] = mine.as[k] = mine.bs[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < 9; ++k) mine.Sab[k] = 0.0;
        const int grp = lane >> 4, j0 = lane & 15, gbase = lane & 48;
        for (int p = 0; 4 * p < nsets; ++p) {
            const int k = 4 * p + grp;                                       // the set of my row group (may be >= nsets: empty)
            const int ks = k < nsets ? k : nsets - 1;
            const int64_t i0 = shfl64(my_i0, ks), i1 = k < nsets ? shfl64(my_i1, ks) : i0;
            // ---- shift: the first usable row of the set (normally found in the first 16 rows)
            double as0 = 0, as1 = 0, as2 = 0, bs0 = 0, bs1 = 0, bs2 = 0;
            bool have = !(i0 < i1);
            for (int64_t off = 0; __ballot(!have && i0 + off < i1) != 0ull; off += 16) {
                const int64_t i = i0 + off + j0;
                bool ok = i < i1 && !have;
                double a0 = 0, a1 = 0, a2 = 0, c0 = 0, c1 = 0, c2 = 0;
                if (ok) {
                    ok = !mask || mask[i] != 0;
                    a0 = src[i * 3]; a1 = src[i * 3 + 1]; a2 = src[i * 3 + 2]; c0 = dst[i * 3]; c1 = dst[i * 3 + 1]; c2 = dst[i * 3 + 2];
                    ok = ok && (fabs(a0) < INFINITY) && (fabs(a1) < INFINITY) && (fabs(a2) < INFINITY) && (fabs(c0) < INFINITY) &&
                         (fabs(c1) < INFINITY) && (fabs(c2) < INFINITY);
                }
                const u64 m = __ballot(ok);
                const unsigned field = (unsigned)(m >> gbase) & 0xffffu;
                const int srcl = gbase + (field ? __ffs((int)field) - 1 : 0);
                const double f0 = shidx(a0, srcl), f1 = shidx(a1, srcl), f2 = shidx(a2, srcl), g0 = shidx(c0, srcl), g1 = shidx(c1, srcl), g2 = shidx(c2, srcl);
                if (!have && field) { as0 = f0; as1 = f1; as2 = f2; bs0 = g0; bs1 = g1; bs2 = g2; have = true; }
            }
            // ---- shifted raw moments, 16 rows of each of the four sets per iteration
            double cnt = 0, Sa0 = 0, Sa1 = 0, Sa2 = 0, Sb0 = 0, Sb1 = 0, Sb2 = 0, Saa = 0;
            double Sab[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            for (int64_t off = 0; __ballot(i0 + off < i1) != 0ull; off += 16) {
                const int64_t i = i0 + off + j0;
                if (i < i1 && (!mask || mask[i] != 0)) {
                    const double a0 = src[i * 3] - as0, a1 = src[i * 3 + 1] - as1, a2 = src[i * 3 + 2] - as2;
                    const double c0 = dst[i * 3] - bs0, c1 = dst[i * 3 + 1] - bs1, c2 = dst[i * 3 + 2] - bs2;
                    cnt += 1.0; Sa0 += a0; Sa1 += a1; Sa2 += a2; Sb0 += c0; Sb1 += c1; Sb2 += c2;
                    Saa += a0 * a0 + a1 * a1 + a2 * a2;
                    Sab[0] += a0 * c0; Sab[1] += a0 * c1; Sab[2] += a0 * c2;
                    Sab[3] += a1 * c0; Sab[4] += a1 * c1; Sab[5] += a1 * c2;
                    Sab[6] += a2 * c0; Sab[7] += a2 * c1; Sab[8] += a2 * c2;
                }
            }
            // ---- row totals (lane 15 of each row), parked in the lane that finishes the set: lane 4p+g <- lane 16g+15
            const int from = 16 * (lane & 3) + 15;
            const bool take = (lane >> 2) == p;
#define GSF_PARK(dst_, v_) { const double tot_ = shidx(row16_scan_sum(v_), from); dst_ = take ? tot_ : dst_; }
            GSF_PARK(mine.n, cnt) GSF_PARK(mine.Sa[0], Sa0) GSF_PARK(mine.Sa[1], Sa1) GSF_PARK(mine.Sa[2], Sa2)
            GSF_PARK(mine.Sb[0], Sb0) GSF_PARK(mine.Sb[1], Sb1) GSF_PARK(mine.Sb[2], Sb2) GSF_PARK(mine.Saa, Saa)
#pragma unroll
            for (int q = 0; q < 9; ++q) GSF_PARK(mine.Sab[q], Sab[q])
#undef GSF_PARK
            // the shifts are row-group uniform: any lane of the group can hand them over
            { const double v0 = shidx(as0, from), v1 = shidx(as1, from), v2 = shidx(as2, from), w0 = shidx(bs0, from), w1 = shidx(bs1, from), w2 = shidx(bs2, from);
              if (take) { mine.as[0] = v0; mine.as[1] = v1; mine.as[2] = v2; mine.bs[0] = w0; mine.bs[1] = w1; mine.bs[2] = w2; } }
        }
        if (lane >= nsets) return;
        st = finalize_raw(mine, Rb, tb, sb);
    } else {
        Moments mine; mine.n = 0.0;
        for (int k = 0; k < nsets; ++k) {
            const int64_t i0 = offsets[b0 + k], i1 = offsets[b0 + k + 1];
            Moments m = wave_moments(src, dst, mask, i0, i1, lane);
            if (lane == k) mine = m;
        }
        if (lane >= nsets) return;
        if (mine.n < 3.0) st = SIM3_NONE;                                    // ref :430
        else st = umeyama_finalize(mine.H, mine.ssq, mine.sc, mine.dc, mine.n, Rb, tb, sb);
    }
    const int64_t b = b0 + lane;
    if (st == SIM3_NONE) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rb[k] = NAN;
        tb[0] = tb[1] = tb[2] = NAN; sb = NAN;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) R[b * 9 + k] = Rb[k];
    t[b * 3] = tb[0]; t[b * 3 + 1] = tb[1]; t[b * 3 + 2] = tb[2];
    s[b] = sb; status[b] = st;
}

// ---- equal-size windows (config C4: 1 M windows of 50 point pairs), split in two launches so that the STREAMING part runs at
// high occupancy without the SVD's registers:
//   windows_moments_kernel   one 16-lane DPP row per window: shifted raw moments of its W rows (rows read once), the 16 sums of a row
//                            reduced by a transposing butterfly (15 exchanges instead of 16 four-stage scans), one 192-byte record
//                            per window written by the row's lanes;
//   windows_finalize_kernel  lane per window: H, centroids, 3x3 Jacobi SVD, R / t / s (the ~900-instruction tail, paid once per
//                            64 windows and overlapping nothing it could stall).
constexpr int WIN_REC = 24;            // doubles per record: 16 sums (Sa3 Sb3 Saa Sab9) | as3 bs3 | n | pad

// sixteen row-local sums at once: after the four exchange stages lane l of a 16-lane row holds the row total of value index
// bitrev4(l & 15) (the same butterfly as wave_sum16, without the cross-row adds)
__device__ __forceinline__ double row_sum16(double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7,
                                            double a8, double a9, double a10, double a11, double a12, double a13, double a14, double a15, int lane)
{
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0;
#define GSF_BFLY(bit, lo_, hi_, XCHG) ((bit ? hi_ : lo_) + XCHG(bit ? lo_ : hi_))
    const double w0 = GSF_BFLY(b0, a0, a8, dpp_quad<0xB1>), w1 = GSF_BFLY(b0, a1, a9, dpp_quad<0xB1>), w2 = GSF_BFLY(b0, a2, a10, dpp_quad<0xB1>),
                 w3 = GSF_BFLY(b0, a3, a11, dpp_quad<0xB1>), w4 = GSF_BFLY(b0, a4, a12, dpp_quad<0xB1>), w5 = GSF_BFLY(b0, a5, a13, dpp_quad<0xB1>),
                 w6 = GSF_BFLY(b0, a6, a14, dpp_quad<0xB1>), w7 = GSF_BFLY(b0, a7, a15, dpp_quad<0xB1>);
    const double x0 = GSF_BFLY(b1, w0, w4, dpp_quad<0x4E>), x1 = GSF_BFLY(b1, w1, w5, dpp_quad<0x4E>), x2 = GSF_BFLY(b1, w2, w6, dpp_quad<0x4E>),
                 x3 = GSF_BFLY(b1, w3, w7, dpp_quad<0x4E>);
    const double y0 = GSF_BFLY(b2, x0, x2, dpp_row_xor<4>), y1 = GSF_BFLY(b2, x1, x3, dpp_row_xor<4>);
    return GSF_BFLY(b3, y0, y1, dpp_row_xor<8>);
#undef GSF_BFLY
}

__global__ void window_offsets_kernel(int64_t* off, int64_t B, int64_t W)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= B) off[b] = b * W;
}

__global__ __launch_bounds__(256) void windows_moments_kernel(const double* __restrict__ src, const double* __restrict__ dst,
                                                              const uint8_t* __restrict__ mask, int64_t B, int W, double* __restrict__ rec)
{
    const int lane = threadIdx.x & 63, j0 = lane & 15;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngroups = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int brev = ((j0 & 1) << 3) | ((j0 & 2) << 1) | ((j0 & 4) >> 1) | ((j0 & 8) >> 3);
    for (int64_t w = group; w < ((B + 3) & ~(int64_t)3); w += ngroups) {   // whole waves iterate together (four windows per wave and trip)
        const bool live = w < B;
        const int64_t i0 = (live ? w : B - 1) * (int64_t)W, i1 = live ? i0 + W : i0;
        // shift = the first usable row of the window: row 0 unless it is masked out or not finite
        double as0 = src[i0 * 3], as1 = src[i0 * 3 + 1], as2 = src[i0 * 3 + 2], bs0 = dst[i0 * 3], bs1 = dst[i0 * 3 + 1], bs2 = dst[i0 * 3 + 2];
        bool have = live && (!mask || mask[i0] != 0) && (fabs(as0) < INFINITY) && (fabs(as1) < INFINITY) && (fabs(as2) < INFINITY) &&
                    (fabs(bs0) < INFINITY) && (fabs(bs1) < INFINITY) && (fabs(bs2) < INFINITY);
        if (__ballot(live && !have) != 0ull) {                            // rare: search the window for its first usable row
            const int gbase = lane & 48;
            for (int64_t off = 0; __ballot(live && !have && i0 + off < i1) != 0ull; off += 16) {
                const int64_t i = i0 + off + j0;
                bool ok = live && i < i1 && !have;
                double a0 = 0, a1 = 0, a2 = 0, c0 = 0, c1 = 0, c2 = 0;
                if (ok) {
                    ok = !mask || mask[i] != 0;
                    a0 = src[i * 3]; a1 = src[i * 3 + 1]; a2 = src[i * 3 + 2]; c0 = dst[i * 3]; c1 = dst[i * 3 + 1]; c2 = dst[i * 3 + 2];
                    ok = ok && (fabs(a0) < INFINITY) && (fabs(a1) < INFINITY) && (fabs(a2) < INFINITY) && (fabs(c0) < INFINITY) &&
                         (fabs(c1) < INFINITY) && (fabs(c2) < INFINITY);
                }
                const u64 m = __ballot(ok);
                const unsigned field = (unsigned)(m >> gbase) & 0xffffu;
                const int srcl = gbase + (field ? __ffs((int)field) - 1 : 0);
                const double f0 = shidx(a0, srcl), f1 = shidx(a1, srcl), f2 = shidx(a2, srcl), g0 = shidx(c0, srcl), g1 = shidx(c1, srcl), g2 = shidx(c2, srcl);
                if (!have && field) { as0 = f0; as1 = f1; as2 = f2; bs0 = g0; bs1 = g1; bs2 = g2; have = true; }
            }
        }
        double cnt = 0, Sa0 = 0, Sa1 = 0, Sa2 = 0, Sb0 = 0, Sb1 = 0, Sb2 = 0, Saa = 0;
        double Sab[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int64_t i = i0 + j0; i < i1; i += 16) {                      // (two slices per trip with the loads up front: no faster, measured)
            if (!mask || mask[i] != 0) {
                const double a0 = src[i * 3] - as0, a1 = src[i * 3 + 1] - as1, a2 = src[i * 3 + 2] - as2;
                const double c0 = dst[i * 3] - bs0, c1 = dst[i * 3 + 1] - bs1, c2 = dst[i * 3 + 2] - bs2;
                cnt += 1.0; Sa0 += a0; Sa1 += a1; Sa2 += a2; Sb0 += c0; Sb1 += c1; Sb2 += c2;
                Saa += a0 * a0 + a1 * a1 + a2 * a2;
                Sab[0] += a0 * c0; Sab[1] += a0 * c1; Sab[2] += a0 * c2;
                Sab[3] += a1 * c0; Sab[4] += a1 * c1; Sab[5] += a1 * c2;
                Sab[6] += a2 * c0; Sab[7] += a2 * c1; Sab[8] += a2 * c2;
            }
        }
        const double tot = row_sum16(Sa0, Sa1, Sa2, Sb0, Sb1, Sb2, Saa, Sab[0], Sab[1], Sab[2], Sab[3], Sab[4], Sab[5], Sab[6], Sab[7], Sab[8], lane);
        const double n = row16_scan_sum(cnt);                             // total in lane 15 of the row
        if (live) {
            double* r = rec + w * WIN_REC;
            r[brev] = tot;                                                // lane l holds sum index bitrev4(l)
            const double tail = (j0 == 0) ? as0 : (j0 == 1) ? as1 : (j0 == 2) ? as2 : (j0 == 3) ? bs0 : (j0 == 4) ? bs1 : bs2;
            if (j0 < 6) r[16 + j0] = tail;
            if (j0 == 15) r[22] = have ? n : 0.0;
        }
    }
}

// The two steps in ONE launch (gsf_set_option "ekf_variant" 9 selects the two-launch form above for A/B): a wave owns 64 consecutive
// windows, runs their moments four windows at a time exactly as windows_moments_kernel does, parks the 64 records in LDS (never in
// HBM: 192 B per window less written and read again, 15 % of the traffic at W = 50) and then finishes all 64 with one lane per
// window -- the Jacobi SVD is paid once per 64 windows with every lane busy, and its registers only cap the occupancy at three
// waves per SIMD, which the up-front loads of a trip (all slices of the four windows requested before the first use) make up for.
constexpr int WINF_WAVES = 2, WINF_STRIDE = WIN_REC + 1;   // records padded to 25 doubles: lane-per-window reads spread over the LDS banks
__global__ __launch_bounds__(64 * WINF_WAVES) void windows_fused_kernel(const double* __restrict__ src, const double* __restrict__ dst,
                                                                         const uint8_t* __restrict__ mask, int64_t B, int W, double* __restrict__ R,
                                                                         double* __restrict__ t, double* __restrict__ s, int32_t* __restrict__ status)
{
    __shared__ double recs[WINF_WAVES][64 * WINF_STRIDE];
    const int lane = threadIdx.x & 63, j0 = lane & 15, wv = threadIdx.x >> 6;
    const int brev = ((j0 & 1) << 3) | ((j0 & 2) << 1) | ((j0 & 4) >> 1) | ((j0 & 8) >> 3);
    double* rw = recs[wv];
    const int64_t nsuper = (B + 63) / 64;
    for (int64_t sp = (int64_t)blockIdx.x * WINF_WAVES + wv; sp < nsuper; sp += (int64_t)gridDim.x * WINF_WAVES) {
        const int64_t w0 = sp * 64;
        for (int trip = 0; trip < 16; ++trip) {
            const int lw = trip * 4 + (lane >> 4);                        // window of my 16-lane row inside the wave's 64
            const int64_t w = w0 + lw;
            const bool live = w < B;
            const int64_t i0 = (live ? w : B - 1) * (int64_t)W, i1 = live ? i0 + W : i0;
            double as0 = src[i0 * 3], as1 = src[i0 * 3 + 1], as2 = src[i0 * 3 + 2], bs0 = dst[i0 * 3], bs1 = dst[i0 * 3 + 1], bs2 = dst[i0 * 3 + 2];
            // up to four slices of the window requested before anything is used (W <= 64: the whole window; longer windows loop below)
            double pa[4][3], pc[4][3]; bool pm[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = i0 + j0 + 16 * k;
                const bool in = i < i1;
                const int64_t ic = in ? i : i0;
                pa[k][0] = src[ic * 3]; pa[k][1] = src[ic * 3 + 1]; pa[k][2] = src[ic * 3 + 2];
                pc[k][0] = dst[ic * 3]; pc[k][1] = dst[ic * 3 + 1]; pc[k][2] = dst[ic * 3 + 2];
                pm[k] = in && (!mask || mask[ic] != 0);
            }
            bool have = live && (!mask || mask[i0] != 0) && (fabs(as0) < INFINITY) && (fabs(as1) < INFINITY) && (fabs(as2) < INFINITY) &&
                        (fabs(bs0) < INFINITY) && (fabs(bs1) < INFINITY) && (fabs(bs2) < INFINITY);
            if (__ballot(live && !have) != 0ull) {                        // rare: search the window for its first usable row
                const int gbase = lane & 48;
                for (int64_t off = 0; __ballot(live && !have && i0 + off < i1) != 0ull; off += 16) {
                    const int64_t i = i0 + off + j0;
                    bool ok = live && i < i1 && !have;
                    double a0 = 0, a1 = 0, a2 = 0, c0 = 0, c1 = 0, c2 = 0;
                    if (ok) {
                        ok = !mask || mask[i] != 0;
                        a0 = src[i * 3]; a1 = src[i * 3 + 1]; a2 = src[i * 3 + 2]; c0 = dst[i * 3]; c1 = dst[i * 3 + 1]; c2 = dst[i * 3 + 2];
                        ok = ok && (fabs(a0) < INFINITY) && (fabs(a1) < INFINITY) && (fabs(a2) < INFINITY) && (fabs(c0) < INFINITY) &&
                             (fabs(c1) < INFINITY) && (fabs(c2) < INFINITY);
                    }
                    const u64 m = __ballot(ok);
                    const unsigned field = (unsigned)(m >> gbase) & 0xffffu;
                    const int srcl = gbase + (field ? __ffs((int)field) - 1 : 0);
                    const double f0 = shidx(a0, srcl), f1 = shidx(a1, srcl), f2 = shidx(a2, srcl), g0 = shidx(c0, srcl), g1 = shidx(c1, srcl), g2 = shidx(c2, srcl);
                    if (!have && field) { as0 = f0; as1 = f1; as2 = f2; bs0 = g0; bs1 = g1; bs2 = g2; have = true; }
                }
            }
            double cnt = 0, Sa0 = 0, Sa1 = 0, Sa2 = 0, Sb0 = 0, Sb1 = 0, Sb2 = 0, Saa = 0;
            double Sab[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#define GSF_WIN_ACC(A0, A1, A2, C0, C1, C2) {                                                                     \
                const double a0 = (A0) - as0, a1 = (A1) - as1, a2 = (A2) - as2, c0 = (C0) - bs0, c1 = (C1) - bs1, c2 = (C2) - bs2; \
                cnt += 1.0; Sa0 += a0; Sa1 += a1; Sa2 += a2; Sb0 += c0; Sb1 += c1; Sb2 += c2;                        \
                Saa += a0 * a0 + a1 * a1 + a2 * a2;                                                                   \
                Sab[0] += a0 * c0; Sab[1] += a0 * c1; Sab[2] += a0 * c2;                                              \
                Sab[3] += a1 * c0; Sab[4] += a1 * c1; Sab[5] += a1 * c2;                                              \
                Sab[6] += a2 * c0; Sab[7] += a2 * c1; Sab[8] += a2 * c2; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (pm[k]) GSF_WIN_ACC(pa[k][0], pa[k][1], pa[k][2], pc[k][0], pc[k][1], pc[k][2])
            for (int64_t i = i0 + j0 + 64; i < i1; i += 16)               // windows of more than 64 rows: the rest, slice by slice (same order)
                if (!mask || mask[i] != 0) GSF_WIN_ACC(src[i * 3], src[i * 3 + 1], src[i * 3 + 2], dst[i * 3], dst[i * 3 + 1], dst[i * 3 + 2])
#undef GSF_WIN_ACC
            const double tot = row_sum16(Sa0, Sa1, Sa2, Sb0, Sb1, Sb2, Saa, Sab[0], Sab[1], Sab[2], Sab[3], Sab[4], Sab[5], Sab[6], Sab[7], Sab[8], lane);
            const double n = row16_scan_sum(cnt);                         // total in lane 15 of the row
            double* r = rw + lw * WINF_STRIDE;
            r[brev] = tot;                                                // lane l holds sum index bitrev4(l)
            const double tail = (j0 == 0) ? as0 : (j0 == 1) ? as1 : (j0 == 2) ? as2 : (j0 == 3) ? bs0 : (j0 == 4) ? bs1 : bs2;
            if (j0 < 6) r[16 + j0] = tail;
            if (j0 == 15) r[22] = (live && have) ? n : 0.0;
        }
        // ---- lane per window: H, centroids, Jacobi SVD, R / t / s of the wave's 64 windows (only this wave reads its records)
        __builtin_amdgcn_s_waitcnt(0xc07f);                               // lgkmcnt(0): the record stores above have landed
        __builtin_amdgcn_wave_barrier();
        const int64_t b = w0 + lane;
        if (b < B) {
            const double* r = rw + lane * WINF_STRIDE;
            RawMoments m;
            m.Sa[0] = r[0]; m.Sa[1] = r[1]; m.Sa[2] = r[2]; m.Sb[0] = r[3]; m.Sb[1] = r[4]; m.Sb[2] = r[5]; m.Saa = r[6];
#pragma unroll
            for (int k = 0; k < 9; ++k) m.Sab[k] = r[7 + k];
            m.as[0] = r[16]; m.as[1] = r[17]; m.as[2] = r[18]; m.bs[0] = r[19]; m.bs[1] = r[20]; m.bs[2] = r[21]; m.n = r[22];
            double Rb[9], tb[3], sb = NAN;
            const int32_t st = finalize_raw(m, Rb, tb, sb);
            if (st == SIM3_NONE) {
#pragma unroll
                for (int k = 0; k < 9; ++k) Rb[k] = NAN;
                tb[0] = tb[1] = tb[2] = NAN; sb = NAN;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) R[b * 9 + k] = Rb[k];
            t[b * 3] = tb[0]; t[b * 3 + 1] = tb[1]; t[b * 3 + 2] = tb[2];
            s[b] = sb; status[b] = st;
        }
        __builtin_amdgcn_wave_barrier();                                  // the next super-trip overwrites the records
    }
}

__global__ __launch_bounds__(64) void windows_finalize_kernel(const double* __restrict__ rec, int64_t B, double* __restrict__ R, double* __restrict__ t,
                                                              double* __restrict__ s, int32_t* __restrict__ status)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const double* r = rec + b * WIN_REC;
    RawMoments m;
    m.Sa[0] = r[0]; m.Sa[1] = r[1]; m.Sa[2] = r[2]; m.Sb[0] = r[3]; m.Sb[1] = r[4]; m.Sb[2] = r[5]; m.Saa = r[6];
#pragma unroll
    for (int k = 0; k < 9; ++k) m.Sab[k] = r[7 + k];
    m.as[0] = r[16]; m.as[1] = r[17]; m.as[2] = r[18]; m.bs[0] = r[19]; m.bs[1] = r[20]; m.bs[2] = r[21]; m.n = r[22];
    double Rb[9], tb[3], sb = NAN;
    const int32_t st = finalize_raw(m, Rb, tb, sb);
    if (st == SIM3_NONE) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rb[k] = NAN;
        tb[0] = tb[1] = tb[2] = NAN; sb = NAN;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) R[b * 9 + k] = Rb[k];
    t[b * 3] = tb[0]; t[b * 3 + 1] = tb[1]; t[b * 3 + 2] = tb[2];
    s[b] = sb; status[b] = st;
}

// ------------------------------------------------------------------------------------------------
constexpr int RANSAC_THREADS = RANSAC_FINAL_THREADS;     // (gsf_ransac.hpp: the early-exit probe reproduces this block's order of summation)
constexpr int RANSAC_SPLIT_MAX_SETS = 32;   // up to this many sets, a set's hypotheses go to many single-wave blocks (ransac_scan_kernel)
constexpr int RANSAC_MAX_SAMPLES = 4096;    // the kernel walks a sample set row by row: no structural limit (a sanity bound)

__device__ __forceinline__ double block_sum(double v, double* sh, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < RANSAC_THREADS / 64; ++w) r += sh[w];
    return r;
}

// hypotheses tr = first, first + step, ... < last of set b (ref :404-414): this thread's best as a key, highest count first, then the
// LOWEST trial (strict > keeps the first, :413); key 0 = no usable hypothesis
__device__ __forceinline__ unsigned long long ransac_scan_trials(const double* __restrict__ src, const double* __restrict__ dst, int64_t i0, int64_t n,
                                                                 const int32_t* __restrict__ my_idx, int first, int last, int step, int ms, double thr,
                                                                 bool& bad_index)
{
    const int64_t i1 = i0 + n;
    long long best_cnt = -1; int best_trial = 0x7fffffff;
    for (int tr = first; tr < last; tr += step) {
        double R[9], t[3], s;
        // caller-fed row indices are validated: a sample naming a row outside [0, n) is skipped like a degenerate one and flagged
        bool in_range = true;
        for (int k = 0; k < ms; ++k) { const int32_t ix = my_idx[(size_t)tr * ms + k]; in_range = in_range && ix >= 0 && (int64_t)ix < n; }
        if (!in_range) { bad_index = true; continue; }
        if (fit_sample(src, dst, i0, my_idx + (size_t)tr * ms, ms, R, t, s) == SIM3_NONE) continue;   // :408
        long long cnt = 0;
        // the rows are wave-uniform scalar loads: eight rows' residuals are formed before the first decision so that their loads are
        // in flight together (one row per round trip otherwise: the loop was latency-bound)
        int64_t r = i0;
        for (; r + 8 <= i1; r += 8) {
            double d2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d2[u] = resid2(src, dst, r + u, R, t, s);
#pragma unroll
            for (int u = 0; u < 8; ++u) cnt += within(d2[u], thr) ? 1 : 0;
        }
        for (; r < i1; ++r) cnt += is_inlier(src, dst, r, R, t, s, thr) ? 1 : 0;
        if (cnt > best_cnt) { best_cnt = cnt; best_trial = tr; }            // strict > keeps the first (:413)
    }
    return ransac_key(best_cnt, best_trial);
}
// ---- the same count with the residuals SCREENED in packed single precision.  ransac_rows_kernel writes every set's rows once as floats
// relative to the set's first row (x' = x - x0 and y' = y - y0, differences taken in double; six component arrays in the context's
// workspace, read back through the SCALAR cache like the double rows -- a lane is a hypothesis, so a row is wave-uniform; the same
// floats in LDS cost more than they saved: a 64-lane read of ONE 8- or 16-byte LDS word is serialised), a hypothesis becomes M = sR and
// t' = M x0 + t - y0, and d = M x' + t' - y' is evaluated two rows per v_pk_fma_f32.  Every term of d carries a relative error of a few
// 2^-24, bounded per hypothesis by e = 2^-24 * 8 * (|M|_rowsum * L_src + |t'|_max + L_dst) + 1e-6 m (L = largest |x'|, |y'| of the set's
// NEAR rows; rows farther than 65 km from the reference point, or not finite, are set apart by ransac_rows_kernel and screened with a
// band of their own -- wide, but a fix 5 000 km off is still an outlier for certain),
// so | |d|_f32 - |d| | <= sqrt(3) e:  a row is an inlier for certain below (thr - sqrt(3) e)^2 and an outlier for certain above
// (thr + sqrt(3) e)^2; in between -- a few millimetres around a 4 m threshold, or a NaN -- the decision is taken in double by the code
// above.  The count, hence the arg-max and everything after it, is the double-precision kernel's.
typedef float rfloat2 __attribute__((ext_vector_type(2)));
typedef float rfloat4 __attribute__((ext_vector_type(4)));
struct RansacRows { const float* x[3]; const float* y[3]; const int32_t* ridx; double x0[3], y0[3]; double Lsrc, Ldst, Fsrc, Fdst; int n_near; };   // x / y / ridx: at the set's first slot

__device__ __forceinline__ unsigned long long ransac_scan_trials_screened(const double* __restrict__ src, const double* __restrict__ dst, int64_t i0,
                                                                          int64_t n, const RansacRows& rows, const int32_t* __restrict__ my_idx,
                                                                          int first, int last, int step, int ms, double thr, bool& bad_index)
{
    long long best_cnt = -1; int best_trial = 0x7fffffff;
    const int nn = rows.n_near, nall = (int)n;                             // slots 0 .. nn-1: near rows (screened); nn .. nall-1: far rows (double)
    for (int tr = first; tr < last; tr += step) {
        double R[9], t[3], s;
        bool in_range = true;
        for (int k = 0; k < ms; ++k) { const int32_t ix = my_idx[(size_t)tr * ms + k]; in_range = in_range && ix >= 0 && (int64_t)ix < n; }
        if (!in_range) { bad_index = true; continue; }
        if (fit_sample(src, dst, i0, my_idx + (size_t)tr * ms, ms, R, t, s) == SIM3_NONE) continue;   // :408
        double M[9], tp[3], rowsum = 0.0, tmax = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            M[c * 3] = s * R[c * 3]; M[c * 3 + 1] = s * R[c * 3 + 1]; M[c * 3 + 2] = s * R[c * 3 + 2];
            tp[c] = M[c * 3] * rows.x0[0] + M[c * 3 + 1] * rows.x0[1] + M[c * 3 + 2] * rows.x0[2] + t[c] - rows.y0[c];
            rowsum = fmax(rowsum, fabs(M[c * 3]) + fabs(M[c * 3 + 1]) + fabs(M[c * 3 + 2]));
            tmax = fmax(tmax, fabs(tp[c]));
        }
        // band of a group of rows whose largest |x'| / |y'| are Ls / Ld: lo = 0 and hi = inf (every row to the double path) when it cannot be
        // trusted -- NaN / inf model, thr <= 0, overflow in float
        auto band = [&](const double Ls, const double Ld, float& lo, float& hi) {
            const double e3 = 1.7320508075688772 * (4.76837158203125e-07 * (rowsum * Ls + tmax + Ld) + 1e-6);   // sqrt(3) * e, 8 * 2^-24
            const double lo_d = thr - e3 > 0.0 ? (thr - e3) * (thr - e3) * (1.0 - 1e-6) : 0.0, hi_d = (thr + e3) * (thr + e3) * (1.0 + 1e-6);
            const bool usable = thr > 0.0 && hi_d < 1e30 && e3 == e3;
            lo = usable ? (float)lo_d * (1.0f - 2e-7f) : 0.0f; hi = usable ? (float)hi_d * (1.0f + 2e-7f) : INFINITY;
        };
        float lo_n, hi_n, lo_f, hi_f;
        band(rows.Lsrc, rows.Ldst, lo_n, hi_n);
        band(rows.Fsrc, rows.Fdst, lo_f, hi_f);
        const float m00 = (float)M[0], m01 = (float)M[1], m02 = (float)M[2], m10 = (float)M[3], m11 = (float)M[4], m12 = (float)M[5],
                    m20 = (float)M[6], m21 = (float)M[7], m22 = (float)M[8], t0 = (float)tp[0], t1 = (float)tp[1], t2 = (float)tp[2];
        long long cnt = 0;
        // slots r0 .. r1-1, eight rows per round: four packed pairs, ONE test whether any of the eight fell into the band -- then, and only
        // for the lanes it happened to, the eight are counted again in double; the rows left over (< 8) always are
        auto count_rows = [&](const int r0, const int r1, const float lo, const float hi) {
            int r = r0;
            for (; r + 8 <= r1; r += 8) {
                int cg = 0; bool unc = false;
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    const rfloat2 X = { rows.x[0][r + u], rows.x[0][r + u + 1] }, Y = { rows.x[1][r + u], rows.x[1][r + u + 1] }, Z = { rows.x[2][r + u], rows.x[2][r + u + 1] };
                    const rfloat2 A = { rows.y[0][r + u], rows.y[0][r + u + 1] }, Bv = { rows.y[1][r + u], rows.y[1][r + u + 1] }, C = { rows.y[2][r + u], rows.y[2][r + u + 1] };
                    const rfloat2 dx = __builtin_elementwise_fma(Z, (rfloat2)m02, __builtin_elementwise_fma(Y, (rfloat2)m01, __builtin_elementwise_fma(X, (rfloat2)m00, (rfloat2)t0))) - A;
                    const rfloat2 dy = __builtin_elementwise_fma(Z, (rfloat2)m12, __builtin_elementwise_fma(Y, (rfloat2)m11, __builtin_elementwise_fma(X, (rfloat2)m10, (rfloat2)t1))) - Bv;
                    const rfloat2 dz = __builtin_elementwise_fma(Z, (rfloat2)m22, __builtin_elementwise_fma(Y, (rfloat2)m21, __builtin_elementwise_fma(X, (rfloat2)m20, (rfloat2)t2))) - C;
                    const rfloat2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    const bool in0 = d2.x < lo, in1 = d2.y < lo;
                    cg += (in0 ? 1 : 0) + (in1 ? 1 : 0);
                    unc = unc || !(in0 || d2.x > hi) || !(in1 || d2.y > hi);       // inside the band, or NaN
                }
                if (unc) {                                                 // rare
                    cg = 0;
                    for (int u = 0; u < 8; ++u) cg += is_inlier(src, dst, i0 + rows.ridx[r + u], R, t, s, thr) ? 1 : 0;
                }
                cnt += cg;
            }
            if (r < r1) {                                                  // eight row numbers, then their rows, in flight together
                int id[8]; double d2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) id[u] = rows.ridx[r + u < r1 ? r + u : r1 - 1];
#pragma unroll
                for (int u = 0; u < 8; ++u) d2[u] = resid2(src, dst, i0 + id[u], R, t, s);
#pragma unroll
                for (int u = 0; u < 8; ++u) cnt += (r + u < r1 && within(d2[u], thr)) ? 1 : 0;
            }
        };
        count_rows(0, nn, lo_n, hi_n);                                     // near rows
        count_rows(nn, nall, lo_f, hi_f);                                  // far rows: their own, much wider band -- a fix 5 000 km off is an outlier for certain
        if (cnt > best_cnt) { best_cnt = cnt; best_trial = tr; }            // strict > keeps the first (:413)
    }
    return ransac_key(best_cnt, best_trial);
}
// rows of every set as floats relative to the set's first row, NEAR rows first: frows = six arrays of `total` floats (x', y', z' of src,
// then of dst), ridx = the row each slot came from (offset inside the set), fhdr[b] = { x0[3], y0[3], L_src, L_dst, n_near } (16 doubles
// per set; [9], [10] = extents of the far rows).  A row is FAR when it lies more than RANSAC_NEAR_M from the reference point in any
// coordinate, or holds a NaN / inf: far rows are listed after the near ones and screened against their own extent, so that one wild fix (a
// zero row is 5 000 km from a UTM track) cannot inflate the rounding bound of the whole set -- L is taken over the near rows only.
constexpr double RANSAC_NEAR_M = 65536.0;
constexpr int RANSAC_HDR = 16;
__global__ __launch_bounds__(256) void ransac_rows_kernel(const double* __restrict__ src, const double* __restrict__ dst, const int64_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ counts, int64_t total, float* __restrict__ frows,
                                                          int32_t* __restrict__ ridx, double* __restrict__ fhdr, const int32_t* __restrict__ decided)
{
    __shared__ double sh_max[16];
    __shared__ int sh_cnt[4];
    __shared__ int sh_near;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    const int64_t i0 = offsets[b], n = counts ? (int64_t)counts[b] : offsets[b + 1] - offsets[b];
    // (a set the early-exit probe decided is not scanned again: nothing to stage)
    if (n <= 0 || i0 < 0 || i0 + n > total || (decided && decided[b])) { if (tid == 0) fhdr[b * RANSAC_HDR + 6] = NAN; return; }   // NaN extent: the set is counted in double
    // reference point: the component-wise median of five probe rows (first, quartiles, last) -- one or two wild rows among them (a set
    // that STARTS with a missing fix is ordinary) do not drag it away from the track.  It need not be a row; NaN sorts last.
    double x0[3], y0[3];
    {
        const int64_t pr[5] = { 0, n / 4, n / 2, (3 * n) / 4, n - 1 };
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double va[5], vb[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) { va[k] = src[(i0 + pr[k]) * 3 + c]; vb[k] = dst[(i0 + pr[k]) * 3 + c]; }
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2)
#pragma unroll
                for (int k = 0; k + 1 < 5 - p2; ++k) {
                    if (!(va[k] <= va[k + 1])) { const double tt = va[k]; va[k] = va[k + 1]; va[k + 1] = tt; }     // NaN bubbles to the end
                    if (!(vb[k] <= vb[k + 1])) { const double tt = vb[k]; vb[k] = vb[k + 1]; vb[k + 1] = tt; }
                }
            x0[c] = va[2]; y0[c] = vb[2];
        }
    }
    auto row = [&](int64_t i, double* a, double* bb) {
        double g = 0.0; bool fin = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = src[(i0 + i) * 3 + c] - x0[c]; bb[c] = dst[(i0 + i) * 3 + c] - y0[c];
            fin = fin && fabs(a[c]) <= RANSAC_NEAR_M && fabs(bb[c]) <= RANSAC_NEAR_M;       // false for NaN / inf as well
            g = fmax(g, fmax(fabs(a[c]), fabs(bb[c])));
        }
        return fin;
    };
    // pass 1: how many near rows
    int mine = 0;
    for (int64_t i = tid; i < n; i += 256) { double a[3], bb[3]; mine += row(i, a, bb) ? 1 : 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) sh_cnt[wave] = mine;
    __syncthreads();
    const int n_near = sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
    __syncthreads();
    // pass 2: near rows to slots 0 .. n_near-1, far rows behind them, both in row order
    double ls = 0.0, ld = 0.0, fs = 0.0, fd = 0.0;                       // extents of the near rows / of the finite far rows
    int base_near = 0, base_far = n_near;
    for (int64_t it = 0; it < n; it += 256) {
        const int64_t i = it + tid;
        double a[3] = { 0, 0, 0 }, bb[3] = { 0, 0, 0 };
        const bool live = i < n, near = live && row(i, a, bb);
        const unsigned long long mn = __ballot(near), mf = __ballot(live && !near);
        if (lane == 0) sh_cnt[wave] = __popcll(mn);
        __syncthreads();
        int before_near = 0, all_near = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) before_near += sh_cnt[w]; all_near += sh_cnt[w]; }
        const int rows_before = (int)(it < n ? ((n - it < 256 ? n - it : 256)) : 0);
        const int live_before_wave = wave * 64 < rows_before ? wave * 64 : rows_before;     // live rows in the waves before this one
        const int before_far = live_before_wave - before_near;
        const unsigned long long below = (1ull << lane) - 1ull;
        if (live) {
            const int slot = near ? base_near + before_near + __popcll(mn & below) : base_far + before_far + __popcll(mf & below);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                frows[(int64_t)c * total + i0 + slot] = (float)a[c];     // far rows too: screened against THEIR extent (NaN / inf stay what they are)
                frows[(int64_t)(3 + c) * total + i0 + slot] = (float)bb[c];
                if (near) { ls = fmax(ls, fabs(a[c])); ld = fmax(ld, fabs(bb[c])); }
                else { if (fabs(a[c]) < 1e300) fs = fmax(fs, fabs(a[c])); if (fabs(bb[c]) < 1e300) fd = fmax(fd, fabs(bb[c])); }
            }
            ridx[i0 + slot] = (int32_t)i;
        }
        base_near += all_near; base_far += rows_before - all_near;
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ls = fmax(ls, __shfl_xor(ls, o, 64)); ld = fmax(ld, __shfl_xor(ld, o, 64));
        fs = fmax(fs, __shfl_xor(fs, o, 64)); fd = fmax(fd, __shfl_xor(fd, o, 64));
    }
    if (lane == 0) { sh_max[wave * 4] = ls; sh_max[wave * 4 + 1] = ld; sh_max[wave * 4 + 2] = fs; sh_max[wave * 4 + 3] = fd; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 0; w < 4; ++w) { ls = fmax(ls, sh_max[w * 4]); ld = fmax(ld, sh_max[w * 4 + 1]); fs = fmax(fs, sh_max[w * 4 + 2]); fd = fmax(fd, sh_max[w * 4 + 3]); }
        for (int c = 0; c < 3; ++c) { fhdr[b * RANSAC_HDR + c] = x0[c]; fhdr[b * RANSAC_HDR + 3 + c] = y0[c]; }
        fhdr[b * RANSAC_HDR + 6] = ls; fhdr[b * RANSAC_HDR + 7] = ld; fhdr[b * RANSAC_HDR + 8] = (double)n_near;
        fhdr[b * RANSAC_HDR + 9] = fs; fhdr[b * RANSAC_HDR + 10] = fd;
    }
    (void)sh_near;
}
__device__ __forceinline__ bool ransac_rows_of_set(const float* __restrict__ frows, const int32_t* __restrict__ ridx, const double* __restrict__ fhdr,
                                                   int64_t total, int64_t b, int64_t i0, RansacRows& rows)
{
    if (!frows) return false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rows.x[c] = frows + (int64_t)c * total + i0; rows.y[c] = frows + (int64_t)(3 + c) * total + i0;
        rows.x0[c] = fhdr[b * RANSAC_HDR + c]; rows.y0[c] = fhdr[b * RANSAC_HDR + 3 + c];
    }
    rows.ridx = ridx + i0;
    rows.Lsrc = fhdr[b * RANSAC_HDR + 6]; rows.Ldst = fhdr[b * RANSAC_HDR + 7]; rows.n_near = (int)fhdr[b * RANSAC_HDR + 8];
    rows.Fsrc = fhdr[b * RANSAC_HDR + 9]; rows.Fdst = fhdr[b * RANSAC_HDR + 10];
    return rows.Lsrc == rows.Lsrc;                                           // NaN: not staged
}

__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long key)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other > key ? other : key;
    }
    return key;
}
__device__ __forceinline__ void ransac_write_none(int64_t b, int32_t best, int tid, double* __restrict__ Rout, double* __restrict__ tout,
                                                  double* __restrict__ sout, int32_t* __restrict__ status, int32_t* __restrict__ n_inliers)
{
    if (tid == 0) {
        for (int k = 0; k < 9; ++k) Rout[b * 9 + k] = NAN;
        tout[b * 3] = tout[b * 3 + 1] = tout[b * 3 + 2] = NAN; sout[b] = NAN;
        status[b] = SIM3_NONE; n_inliers[b] = best;
    }
}
// the winner's mask and the final fit on its inliers (ref :415-421), by a whole RANSAC_THREADS block; key = block-uniform arg-max
__device__ __forceinline__ void ransac_finish(int64_t b, unsigned long long key, bool any_bad, const double* __restrict__ src,
                                              const double* __restrict__ dst, int64_t i0, int64_t n, const int32_t* __restrict__ my_idx, int ms,
                                              double thr, int min_inliers, double* __restrict__ Rout, double* __restrict__ tout,
                                              double* __restrict__ sout, int32_t* __restrict__ status, uint8_t* __restrict__ inlier_mask,
                                              int32_t* __restrict__ n_inliers, double* sh_fit, double* sh_red)
{
    const int tid = threadIdx.x;
    const int64_t i1 = i0 + n;
    const long long win_cnt = (long long)(key >> 32) - 1;
    const int win_trial = 0x7fffffff - (int)(key & 0xffffffffull);
    if (win_cnt < 0) {                                                       // every trial was degenerate
        for (int64_t i = i0 + tid; i < i1; i += RANSAC_THREADS) inlier_mask[i] = 0;
        ransac_write_none(b, -1, tid, Rout, tout, sout, status, n_inliers);
        return;
    }
    // ---- winner's mask
    if (tid == 0) {
        double R[9], t[3], s;
        fit_sample(src, dst, i0, my_idx + (size_t)win_trial * ms, ms, R, t, s);
        for (int k = 0; k < 9; ++k) sh_fit[k] = R[k];
        sh_fit[9] = t[0]; sh_fit[10] = t[1]; sh_fit[11] = t[2]; sh_fit[12] = s;
    }
    __syncthreads();
    double Rw[9], tw[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rw[k] = sh_fit[k];
    tw[0] = sh_fit[9]; tw[1] = sh_fit[10]; tw[2] = sh_fit[11];
    const double sw = sh_fit[12];
    for (int64_t i = i0 + tid; i < i1; i += RANSAC_THREADS) inlier_mask[i] = is_inlier(src, dst, i, Rw, tw, sw, thr) ? 1 : 0;
    if (win_cnt < (long long)min_inliers) { ransac_write_none(b, (int32_t)win_cnt, tid, Rout, tout, sout, status, n_inliers); return; }   // ref :416-418
    __syncthreads();                                                         // mask visible block-wide (same CU)
    // ---- final fit on the inliers (ref :420-421): block-wide two-pass moments
    double acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
    for (int64_t i = i0 + tid; i < i1; i += RANSAC_THREADS) {
        if (!inlier_mask[i]) continue;
        final_moments1(acc, src[i * 3], src[i * 3 + 1], src[i * 3 + 2], dst[i * 3], dst[i * 3 + 1], dst[i * 3 + 2]);
    }
    double tot[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) tot[k] = block_sum(acc[k], sh_red, tid);
    const double cntf = tot[0];
    double sc[3] = { tot[1] / cntf, tot[2] / cntf, tot[3] / cntf }, dc[3] = { tot[4] / cntf, tot[5] / cntf, tot[6] / cntf };
    double h[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int64_t i = i0 + tid; i < i1; i += RANSAC_THREADS) {
        if (!inlier_mask[i]) continue;
        final_moments2(h, src[i * 3], src[i * 3 + 1], src[i * 3 + 2], dst[i * 3], dst[i * 3 + 1], dst[i * 3 + 2], sc, dc);
    }
    double H[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) H[k] = block_sum(h[k], sh_red, tid);
    if (tid == 0) {
        double R[9], t[3], s = NAN; int32_t st;
        if (cntf < 3.0) st = SIM3_NONE;
        else st = umeyama_finalize(H, H[9], sc, dc, cntf, R, t, s);
        if (st == SIM3_NONE) { for (int k = 0; k < 9; ++k) R[k] = NAN; t[0] = t[1] = t[2] = NAN; s = NAN; }
        for (int k = 0; k < 9; ++k) Rout[b * 9 + k] = R[k];
        tout[b * 3] = t[0]; tout[b * 3 + 1] = t[1]; tout[b * 3 + 2] = t[2]; sout[b] = s;
        status[b] = st | ((any_bad && st != SIM3_NONE) ? SIM3_FLAG_BAD_INDEX : 0); n_inliers[b] = (int32_t)win_cnt;
    }
}

// One block per set: hypotheses strided over the threads, block arg-max, finish.
__global__ __launch_bounds__(RANSAC_THREADS) void ransac_batch_kernel(
    const double* __restrict__ src, const double* __restrict__ dst, const int64_t* __restrict__ offsets, const int32_t* __restrict__ counts,
    const int32_t* __restrict__ sample_idx, int trials, int ms, double thr, int min_inliers, double* __restrict__ Rout,
    double* __restrict__ tout, double* __restrict__ sout, int32_t* __restrict__ status, uint8_t* __restrict__ inlier_mask,
    int32_t* __restrict__ n_inliers, const float* __restrict__ frows, const int32_t* __restrict__ ridx, const double* __restrict__ fhdr, int64_t total,
    int trial0, unsigned long long* __restrict__ keys_io, const int32_t* __restrict__ decided)
{
    __shared__ unsigned long long sh_key[RANSAC_THREADS / 64];
    __shared__ double sh_fit[13];
    __shared__ double sh_red[RANSAC_THREADS / 64];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    if (decided && decided[b] == 2) return;                                  // the early-exit probe also formed the final fit: R, t, s, mask, counts are written
    // set b = rows offsets[b] .. offsets[b] + n; n = counts[b] when the sets sit in fixed-stride slots (robust pipeline), else the gap
    const int64_t i0 = offsets[b], n = counts ? (int64_t)counts[b] : offsets[b + 1] - offsets[b], i1 = i0 + n;
    if (n < ms) {                                                            // ref :395-397
        for (int64_t i = i0 + tid; i < i1; i += RANSAC_THREADS) inlier_mask[i] = 0;
        ransac_write_none(b, -1, tid, Rout, tout, sout, status, n_inliers);
        return;
    }
    const int32_t* my_idx = sample_idx + (size_t)b * (size_t)trials * (size_t)ms;
    bool bad_index = false;
    unsigned long long key = 0ull;
    RansacRows rows;
    // hypotheses trial0 .. trials - 1 (trial0 > 0: the ones before were scored by the early-exit probe of the robust chain, whose best comes
    // in through keys_io; a set it DECIDED -- some trial counted every row, so no later one can replace it, ref :413 -- is not scanned)
    if (decided && decided[b]) {                                             // block-uniform
    } else if (ransac_rows_of_set(frows, ridx, fhdr, total, b, i0, rows)) {  // block-uniform
        key = wave_max_key(ransac_scan_trials_screened(src, dst, i0, n, rows, my_idx, trial0 + tid, trials, RANSAC_THREADS, ms, thr, bad_index));
    } else {
        key = wave_max_key(ransac_scan_trials(src, dst, i0, n, my_idx, trial0 + tid, trials, RANSAC_THREADS, ms, thr, bad_index));
    }
    const bool any_bad = __syncthreads_or(bad_index ? 1 : 0) != 0;
    if ((tid & 63) == 0) sh_key[tid >> 6] = key;
    __syncthreads();
    key = sh_key[0];
#pragma unroll
    for (int w = 1; w < RANSAC_THREADS / 64; ++w) key = sh_key[w] > key ? sh_key[w] : key;
    if (keys_io) {                                                           // the same ordering across both parts: count first, then the lowest trial
        const unsigned long long before = keys_io[b * 2];
        key = before > key ? before : key;
        if (tid == 0) keys_io[b * 2] = key;
    }
    ransac_finish(b, key, any_bad, src, dst, i0, n, my_idx, ms, thr, min_inliers, Rout, tout, sout, status, inlier_mask, n_inliers, sh_fit, sh_red);
}

// FEW sets (the reference's own case is one): the hypotheses of a set are spread over many single-wave blocks, a hypothesis per lane,
// and meet in keys[b] = { arg-max key, bad-index flag } (zeroed by the launcher); ransac_finish_kernel then does what the tail of
// ransac_batch_kernel does.  One set x 1 000 trials: 130 us on one CU -> two launches of ~25 and ~8 us.
__global__ __launch_bounds__(64) void ransac_scan_kernel(const double* __restrict__ src, const double* __restrict__ dst, const int64_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ counts, const int32_t* __restrict__ sample_idx, int trials, int ms,
                                                         double thr, unsigned long long* __restrict__ keys, const float* __restrict__ frows,
                                                         const int32_t* __restrict__ ridx, const double* __restrict__ fhdr, int64_t total, int trial0,
                                                         const int32_t* __restrict__ decided)
{
    const int64_t b = blockIdx.y;
    const int64_t i0 = offsets[b], n = counts ? (int64_t)counts[b] : offsets[b + 1] - offsets[b];
    if (n < ms || (decided && decided[b])) return;
    const int tr = trial0 + blockIdx.x * 64 + threadIdx.x;
    bool bad_index = false;
    const int32_t* my_idx = sample_idx + (size_t)b * (size_t)trials * (size_t)ms;
    unsigned long long key;
    RansacRows rows;
    if (ransac_rows_of_set(frows, ridx, fhdr, total, b, i0, rows)) {
        key = wave_max_key(ransac_scan_trials_screened(src, dst, i0, n, rows, my_idx, tr, tr < trials ? tr + 1 : tr, 1, ms, thr, bad_index));
    } else {
        key = wave_max_key(ransac_scan_trials(src, dst, i0, n, my_idx, tr, tr < trials ? tr + 1 : tr, 1, ms, thr, bad_index));
    }
    const bool any_bad = __ballot(bad_index) != 0ull;
    if (threadIdx.x == 0) {
        if (key) atomicMax(&keys[b * 2], key);
        if (any_bad) atomicOr(&keys[b * 2 + 1], 1ull);
    }
}
__global__ __launch_bounds__(RANSAC_THREADS) void ransac_finish_kernel(
    const double* __restrict__ src, const double* __restrict__ dst, const int64_t* __restrict__ offsets, const int32_t* __restrict__ counts,
    const int32_t* __restrict__ sample_idx, int trials, int ms, double thr, int min_inliers, const unsigned long long* __restrict__ keys,
    double* __restrict__ Rout, double* __restrict__ tout, double* __restrict__ sout, int32_t* __restrict__ status,
    uint8_t* __restrict__ inlier_mask, int32_t* __restrict__ n_inliers, const int32_t* __restrict__ decided)
{
    __shared__ double sh_fit[13];
    __shared__ double sh_red[RANSAC_THREADS / 64];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    if (decided && decided[b] == 2) return;                                  // finished by the early-exit probe
    const int64_t i0 = offsets[b], n = counts ? (int64_t)counts[b] : offsets[b + 1] - offsets[b], i1 = i0 + n;
    if (n < ms) {                                                            // ref :395-397
        for (int64_t i = i0 + tid; i < i1; i += RANSAC_THREADS) inlier_mask[i] = 0;
        ransac_write_none(b, -1, tid, Rout, tout, sout, status, n_inliers);
        return;
    }
    // a key of 0 (no usable hypothesis anywhere) reads as count -1, like the one-block kernel's
    ransac_finish(b, keys[b * 2], keys[b * 2 + 1] != 0ull, src, dst, i0, n, sample_idx + (size_t)b * (size_t)trials * (size_t)ms, ms, thr, min_inliers, Rout,
                  tout, sout, status, inlier_mask, n_inliers, sh_fit, sh_red);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void apply_sim3_kernel(const double* __restrict__ pos, const double* __restrict__ quat,
                                                         const int64_t* __restrict__ offsets, const double* __restrict__ R,
                                                         const double* __restrict__ t, const double* __restrict__ s,
                                                         double* __restrict__ pos_out, double* __restrict__ quat_out,
                                                         int32_t* __restrict__ bad_quat)
{
    const int64_t b = blockIdx.x;
    const int64_t i0 = offsets[b], i1 = offsets[b + 1];
    double Rb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rb[k] = R[b * 9 + k];
    const double t0 = t[b * 3], t1 = t[b * 3 + 1], t2 = t[b * 3 + 2], sb = s[b];
    const Quat qR = quat_from_matrix(Rb);                                    // ref :465
    int bad = 0;
    for (int64_t i = i0 + blockIdx.y * blockDim.x + threadIdx.x; i < i1; i += (int64_t)blockDim.x * gridDim.y) {
        const double x = pos[i * 3], y = pos[i * 3 + 1], z = pos[i * 3 + 2];
        pos_out[i * 3] = sb * (x * Rb[0] + y * Rb[1] + z * Rb[2]) + t0;      // ref :464
        pos_out[i * 3 + 1] = sb * (x * Rb[3] + y * Rb[4] + z * Rb[5]) + t1;
        pos_out[i * 3 + 2] = sb * (x * Rb[6] + y * Rb[7] + z * Rb[8]) + t2;
        Quat qn; const bool ok = quat_unit(Quat{ quat[i * 4], quat[i * 4 + 1], quat[i * 4 + 2], quat[i * 4 + 3] }, qn);
        Quat o = quat_mul(qR, qn);                                           // ref :466
        if (!ok) { o = Quat{ NAN, NAN, NAN, NAN }; bad = 1; }
        quat_out[i * 4] = o.x; quat_out[i * 4 + 1] = o.y; quat_out[i * 4 + 2] = o.z; quat_out[i * 4 + 3] = o.w;
    }
    if (bad_quat && bad) atomicOr(&bad_quat[b], 1);
}

// K3 with the rows moved as SLABS (the big-batch EKF kernel's scheme, gsf_wave_common.hpp): a wave takes 64 consecutive poses of a
// trajectory, fetches their pos / quat slabs (1 536 + 2 048 contiguous bytes) as 16-byte pieces lane after lane, transposes through its own
// 3.5 KB of LDS so that every lane holds one pose, transforms it, and sends the results back the same way -- four loads and four stores of a
// kilobyte per 64 poses instead of seven + seven 8-byte accesses at 24 / 32-byte stride.  The next chunk's pieces are requested before the
// current chunk is transformed.  Same arithmetic per pose as apply_sim3_kernel: same bits.
typedef double k3_v2 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ int k3_piece_off(const int piece, const int slab_bytes) { const int o = piece * 16; return o < slab_bytes - 16 ? o : slab_bytes - 16; }
__global__ __launch_bounds__(256) void apply_sim3_slab_kernel(const double* __restrict__ pos, const double* __restrict__ quat,
                                                              const int64_t* __restrict__ offsets, const double* __restrict__ R,
                                                              const double* __restrict__ t, const double* __restrict__ s,
                                                              double* __restrict__ pos_out, double* __restrict__ quat_out,
                                                              int32_t* __restrict__ bad_quat)
{
    __shared__ double stage_all[4][64 * 7];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* stage = stage_all[wv];
    char* sp = (char*)stage; char* sq = sp + 64 * 24;
    const int64_t b = blockIdx.x;
    const int64_t i0 = offsets[b], i1 = offsets[b + 1];
    double Rb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rb[k] = R[b * 9 + k];
    const double t0 = t[b * 3], t1 = t[b * 3 + 1], t2 = t[b * 3 + 2], sb = s[b];
    const Quat qR = quat_from_matrix(Rb);                                    // ref :465
    int bad = 0;
    const int64_t nchunks = (i1 - i0 + 63) / 64, cstep = 4 * (int64_t)gridDim.y;
    auto fetch = [&](const int64_t c, k3_v2* P, k3_v2* Q) {
        const int64_t r0 = i0 + (c < nchunks ? c : nchunks - 1) * 64;        // past the end: any chunk of the track, never used
        const int rows = (int)(i1 - r0 < 64 ? i1 - r0 : 64);
        const char* pb = (const char*)(pos + r0 * 3); const char* qb = (const char*)(quat + r0 * 4);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            P[k] = __builtin_nontemporal_load((const k3_v2*)(pb + k3_piece_off(lane + 64 * k, rows * 24)));
            Q[k] = __builtin_nontemporal_load((const k3_v2*)(qb + k3_piece_off(lane + 64 * k, rows * 32)));
        }
    };
    k3_v2 P[2], Q[2];
    int64_t c = (int64_t)blockIdx.y * 4 + wv;
    if (c < nchunks) fetch(c, P, Q);
    for (; c < nchunks; c += cstep) {
        const int64_t r0 = i0 + c * 64;
        const int rows = (int)(i1 - r0 < 64 ? i1 - r0 : 64);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            *(k3_v2*)(sp + k3_piece_off(lane + 64 * k, rows * 24)) = P[k];
            *(k3_v2*)(sq + k3_piece_off(lane + 64 * k, rows * 32)) = Q[k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // a wave's LDS operations complete in order: no barrier between its own phases
        if (c + cstep < nchunks) fetch(c + cstep, P, Q);                     // (wave-uniform) the next chunk travels while this one is transformed
        const int r = lane < rows ? lane : rows - 1;
        const double* dp = (const double*)sp + r * 3; const double* dq = (const double*)sq + r * 4;
        const double x = dp[0], y = dp[1], z = dp[2];
        const Quat qin{ dq[0], dq[1], dq[2], dq[3] };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const double ox = sb * (x * Rb[0] + y * Rb[1] + z * Rb[2]) + t0;     // ref :464
        const double oy = sb * (x * Rb[3] + y * Rb[4] + z * Rb[5]) + t1;
        const double oz = sb * (x * Rb[6] + y * Rb[7] + z * Rb[8]) + t2;
        Quat qn; const bool ok = quat_unit(qin, qn);
        Quat o = quat_mul(qR, qn);                                           // ref :466
        if (!ok) { o = Quat{ NAN, NAN, NAN, NAN }; if (lane < rows) bad = 1; }
        double* wp = (double*)sp + lane * 3; double* wq = (double*)sq + lane * 4;
        wp[0] = ox; wp[1] = oy; wp[2] = oz; wq[0] = o.x; wq[1] = o.y; wq[2] = o.z; wq[3] = o.w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int ppieces = (rows * 3) / 2, qpieces = rows * 2;
        k3_v2* ps = (k3_v2*)(pos_out + r0 * 3); k3_v2* qs = (k3_v2*)(quat_out + r0 * 4);
        const k3_v2* svp = (const k3_v2*)sp; const k3_v2* svq = (const k3_v2*)sq;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pc = lane + 64 * k;
            if (pc < ppieces) __builtin_nontemporal_store(svp[pc], &ps[pc]);
            if (pc < qpieces) __builtin_nontemporal_store(svq[pc], &qs[pc]);
        }
        if ((rows & 1) && lane == 0) __builtin_nontemporal_store(((const double*)sp)[rows * 3 - 1], &pos_out[(r0 + rows) * 3 - 1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the pieces are read before the next chunk's land in the stage
    }
    if (bad_quat && __ballot(bad != 0) != 0ull && lane == 0) atomicOr(&bad_quat[b], 1);
}

}  // namespace

namespace gsf {
int launch_sim3_ransac(gsf_ctx* ctx, const double* src, const double* dst, const int64_t* offsets, const int32_t* counts, int64_t B,
                       const int32_t* sample_idx, int32_t trials, int32_t min_samples, double thr, int32_t min_inliers, double* R, double* t,
                       double* s, int32_t* status, uint8_t* inlier_mask, int32_t* n_inliers, int64_t total_rows, int32_t trial0,
                       unsigned long long* keys_io, const int32_t* decided)
{
    // residual counts screened in packed single precision with an exact double re-check inside the error band (the counts are the
    // double kernel's); needs the rows as floats in the workspace, hence their total number (0 = unknown to the host: double
    // throughout, as with gsf_set_option "k2b_screen" 0)
    const float* frows = nullptr; const double* fhdr = nullptr; const int32_t* ridx = nullptr;
    if (ctx->k2b_screen != 0 && total_rows > 0 && total_rows < ((int64_t)1 << 31) && trials - trial0 >= 64) {   // (slots and row indices of the screen are int32)
        const size_t hdr_bytes = ((size_t)B * RANSAC_HDR * 8 + 255) & ~(size_t)255;
        const int rc = ensure_k2b_scratch(ctx, hdr_bytes + (size_t)total_rows * 28);
        if (rc) return rc;
        double* h = (double*)ctx->k2b_scratch; float* f = (float*)((char*)ctx->k2b_scratch + hdr_bytes); int32_t* ri = (int32_t*)(f + (size_t)total_rows * 6);
        hipLaunchKernelGGL(ransac_rows_kernel, dim3((unsigned)B), dim3(256), 0, ctx->stream, src, dst, offsets, counts, total_rows, f, ri, h, decided);
        GSF_HIP(hipGetLastError());                                       // a failed staging launch must not leave the scoring kernels on unstaged rows
        frows = f; fhdr = h; ridx = ri;
    }
    if (B <= RANSAC_SPLIT_MAX_SETS && trials - trial0 >= 256) {
        // few sets: hypotheses spread over the chip, then one finishing block per set
        unsigned long long* keys = keys_io ? keys_io : (unsigned long long*)ctx->small_scratch;
        if (!keys_io) GSF_HIP(hipMemsetAsync(keys, 0, (size_t)B * 16, ctx->stream));
        hipLaunchKernelGGL(ransac_scan_kernel, dim3((unsigned)((trials - trial0 + 63) / 64), (unsigned)B), dim3(64), 0, ctx->stream, src, dst, offsets, counts,
                           sample_idx, (int)trials, (int)min_samples, thr, keys, frows, ridx, fhdr, total_rows, (int)trial0, decided);
        hipLaunchKernelGGL(ransac_finish_kernel, dim3((unsigned)B), dim3(RANSAC_THREADS), 0, ctx->stream, src, dst, offsets, counts, sample_idx,
                           (int)trials, (int)min_samples, thr, (int)min_inliers, keys, R, t, s, status, inlier_mask, n_inliers, decided);
        GSF_HIP(hipGetLastError());
        return GSF_OK;
    }
    hipLaunchKernelGGL(ransac_batch_kernel, dim3((unsigned)B), dim3(RANSAC_THREADS), 0, ctx->stream, src, dst, offsets, counts, sample_idx,
                       (int)trials, (int)min_samples, thr, (int)min_inliers, R, t, s, status, inlier_mask, n_inliers, frows, ridx, fhdr, total_rows,
                       (int)trial0, keys_io, decided);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
// K3 (ref :428-459); bad_quat (may be NULL): OR of "a quaternion could not be normalised" per trajectory -- zeroed here unless the caller's
// chain has done so already (the whole-run chain: one fill launch fewer)
int launch_apply_sim3(gsf_ctx* ctx, const double* pos, const double* quat, const int64_t* offsets, int64_t B, const double* R, const double* t,
                      const double* s, double* pos_out, double* quat_out, int32_t* bad_quat, bool bad_quat_zeroed)
{
    if (bad_quat && !bad_quat_zeroed) GSF_HIP(hipMemsetAsync(bad_quat, 0, (size_t)B * 4, ctx->stream));
    // few trajectories -> several blocks per trajectory so a single long track still fills the chip
    const unsigned gy = B >= 2048 ? 1u : (B >= 256 ? 4u : 64u);
    if (ctx->ekf_variant == 11)                                              // A/B: the per-pose accesses of rounds 1-2
        hipLaunchKernelGGL(apply_sim3_kernel, dim3((unsigned)B, gy), dim3(256), 0, ctx->stream, pos, quat, offsets, R, t, s, pos_out, quat_out, bad_quat);
    else
        hipLaunchKernelGGL(apply_sim3_slab_kernel, dim3((unsigned)B, gy), dim3(256), 0, ctx->stream, pos, quat, offsets, R, t, s, pos_out, quat_out, bad_quat);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
}  // namespace gsf

extern "C" {

int gsf_sim3_umeyama_batch_dev(gsf_ctx* ctx, const double* src, const double* dst, const uint8_t* mask, const int64_t* offsets,
                               int64_t B, double* R, double* t, double* s, int32_t* status)
{
    GSF_REQUIRE(ctx && offsets && R && t && s && status, "NULL argument");
    GSF_REQUIRE(B >= 0, "negative B");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(umeyama_batch_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, src, dst, mask, offsets, B, R, t, s, status);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_sim3_umeyama_windows_dev(gsf_ctx* ctx, const double* src, const double* dst, const uint8_t* mask, int64_t B, int32_t W, double* R, double* t,
                                 double* s, int32_t* status)
{
    GSF_REQUIRE(ctx && R && t && s && status, "NULL argument");
    GSF_REQUIRE(B >= 0 && W >= 0, "negative B or W");
    if (B == 0) return GSF_OK;
    GSF_REQUIRE(W == 0 || (src && dst), "NULL points");
    GSF_HIP(hipSetDevice(ctx->device));
    if (W == 0 || W > 4096) {                                             // empty windows (all None) / long windows: the ragged kernel on generated offsets
        int rc = ensure_scratch(ctx, (size_t)(B + 1) * 8);
        if (rc) return rc;
        int64_t* off = (int64_t*)ctx->scratch;
        hipLaunchKernelGGL(window_offsets_kernel, dim3((unsigned)((B + 256) / 256)), dim3(256), 0, ctx->stream, off, B, (int64_t)W);
        GSF_HIP(hipGetLastError());
        return gsf_sim3_umeyama_batch_dev(ctx, src, dst, mask, off, B, R, t, s, status);
    }
    // one launch (moments in LDS, lane-per-window finish per 64 windows) where it is the faster form -- windows of up to 64 rows (a
    // trip requests the whole window up front) in batches that fill the chip: 0.50 vs 0.58 ms at 1 M x 50; 0.57 vs 0.53 ms at 200 k x
    // 271 and 27 vs 23 us at 4 096 x 50 the other way round.  Both forms produce the same bits (tools/ab_c4.py), so the choice by size
    // changes no result.  gsf_set_option "ekf_variant": 9 = always two launches, 10 = always one.
    if (ctx->ekf_variant == 10 || (ctx->ekf_variant != 9 && W <= 64 && B >= 16384)) {
        int64_t nblk = ((B + 63) / 64 + WINF_WAVES - 1) / WINF_WAVES;
        if (nblk > 65536) nblk = 65536;
        hipLaunchKernelGGL(windows_fused_kernel, dim3((unsigned)nblk), dim3(64 * WINF_WAVES), 0, ctx->stream, src, dst, mask, B, (int)W, R, t, s, status);
        GSF_HIP(hipGetLastError());
        return GSF_OK;
    }
    int rc = ensure_scratch(ctx, (size_t)B * WIN_REC * 8);
    if (rc) return rc;
    double* rec = (double*)ctx->scratch;
    const int64_t groups = (B + 3) / 4 * 4;                               // one 16-lane row per window and trip; grid-stride beyond 64 k blocks
    int64_t blocks = (groups * 16 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(windows_moments_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, src, dst, mask, B, (int)W, rec);
    GSF_HIP(hipGetLastError());
    hipLaunchKernelGGL(windows_finalize_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, rec, B, R, t, s, status);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_sim3_ransac_batch_dev(gsf_ctx* ctx, const double* src, const double* dst, const int64_t* offsets, int64_t B,
                              const int32_t* sample_idx, int32_t trials, int32_t min_samples, double thr, int32_t min_inliers,
                              double* R, double* t, double* s, int32_t* status, uint8_t* inlier_mask, int32_t* n_inliers)
{
    GSF_REQUIRE(ctx && offsets && R && t && s && status && inlier_mask && n_inliers, "NULL argument");
    GSF_REQUIRE(B >= 0 && trials >= 0, "negative B or trials");
    GSF_REQUIRE(min_samples >= 1 && min_samples <= RANSAC_MAX_SAMPLES, "min_samples must be in [1,4096]");
    GSF_REQUIRE(trials == 0 || sample_idx, "sample_idx is NULL");
    GSF_REQUIRE(B <= 0x7fffffff, "B too large for one launch");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_sim3_ransac(ctx, src, dst, offsets, nullptr, B, sample_idx, trials, min_samples, thr, min_inliers, R, t, s, status, inlier_mask, n_inliers);
}

int gsf_sim3_ransac_batch_rows_dev(gsf_ctx* ctx, const double* src, const double* dst, const int64_t* offsets, int64_t total_rows, int64_t B,
                                   const int32_t* sample_idx, int32_t trials, int32_t min_samples, double thr, int32_t min_inliers,
                                   double* R, double* t, double* s, int32_t* status, uint8_t* inlier_mask, int32_t* n_inliers)
{
    GSF_REQUIRE(ctx && offsets && R && t && s && status && inlier_mask && n_inliers, "NULL argument");
    GSF_REQUIRE(B >= 0 && trials >= 0 && total_rows >= 0, "negative B, trials or total_rows");
    GSF_REQUIRE(min_samples >= 1 && min_samples <= RANSAC_MAX_SAMPLES, "min_samples must be in [1,4096]");
    GSF_REQUIRE(trials == 0 || sample_idx, "sample_idx is NULL");
    GSF_REQUIRE(B <= 0x7fffffff, "B too large for one launch");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_sim3_ransac(ctx, src, dst, offsets, nullptr, B, sample_idx, trials, min_samples, thr, min_inliers, R, t, s, status, inlier_mask, n_inliers,
                              total_rows);
}

int gsf_apply_sim3_batch_dev(gsf_ctx* ctx, const double* pos, const double* quat, const int64_t* offsets, int64_t B, const double* R,
                             const double* t, const double* s, double* pos_out, double* quat_out, int32_t* bad_quat)
{
    GSF_REQUIRE(ctx && offsets && R && t && s, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_apply_sim3(ctx, pos, quat, offsets, B, R, t, s, pos_out, quat_out, bad_quat, false);
}

}  // extern "C"
