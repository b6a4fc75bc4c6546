// gsf_align.hip -- GNSS positions interpolated onto the SLAM stamps: dynamic_time_alignment (EKFGPSSLAM.py:325-387), the
// step right before both the Sim3 fit and the EKF (SURVEY 8f "next-1").
//
// One 64-lane workgroup per trajectory, the GNSS track staged in LDS (t, y[3], second derivatives M[3], c': 64 B per fix;
// tracks of more than 2560 fixes are staged in a global scratch slab instead):
//   1. sort by stamp if needed (rank sort, stable by input order) and keep the first of equal stamps (np.unique, :339-346);
//   2. split at gaps > max_gps_gap_threshold (:348-352); a segment needs >= 2 fixes and stamps increasing by > 1e-9 (:364);
//   3. >= 4 fixes: not-a-knot cubic spline (scipy interp1d(kind='cubic'), :362/:368) -- the tridiagonal system for the knot
//      second derivatives is solved as three prefix scans over the rows (a Moebius scan for c', an affine scan for d', an affine
//      scan in reverse for the back substitution; all 64 lanes); 2-3 fixes: linear;
//   4. every SLAM stamp inside [t0-1e-9, t1+1e-9] (:372-373) is evaluated by its own lane (binary search of the interval);
//      stamps outside [t0, t1] get NaN like interp1d(bounds_error=False, fill_value=nan); valid = all three finite (:377-379).
// The clock-offset estimate of :336 is identically 0 and is not computed.
// NaN stamps sort last (np.argsort), which leaves the last segment without strictly increasing stamps: it is skipped (:364), as in
// the reference.  drop_masked != 0 (the chain from the geodetic log): fixes whose easting AND northing are NaN -- the mark
// gsf_gps_rows_to_utm_batch_dev leaves on rows that load_gps_data removes before the projection (:259-264) -- never reach the
// spline, exactly as if the loader had removed them.
#include "gsf_internal.hpp"
#include "gsf_ekf_core.hpp"
#include "gsf_wave_common.hpp"   // DPP scan stages, lane broadcasts

using namespace gsf;

namespace {

constexpr int ALIGN_THREADS = 64;

// The staging area is LDS when the track fits (<= 2 560 fixes), else a slab of global scratch.  The body is a template over the pointer
// type: with one generic pointer for both, every access of the serial spline sweeps was a FLAT instruction (hundreds of cycles of
// latency per dependent step instead of an LDS access) -- 93 of the kernel's 115 us.
template <class StagePtr>
__device__ __forceinline__ void time_align_body(StagePtr lds, const double* __restrict__ slam_t, const int64_t* __restrict__ slam_off,
                                                const double* __restrict__ gps_t, const double* __restrict__ gps_p,
                                                const int64_t* __restrict__ gps_off, double max_gap, int max_g, int drop_masked,
                                                double* __restrict__ aligned, uint8_t* __restrict__ valid, int32_t* __restrict__ status);

__global__ __launch_bounds__(ALIGN_THREADS) void time_align_kernel(const double* __restrict__ slam_t, const int64_t* __restrict__ slam_off,
                                                                    const double* __restrict__ gps_t, const double* __restrict__ gps_p,
                                                                    const int64_t* __restrict__ gps_off, double max_gap, int max_g, int drop_masked,
                                                                    double* __restrict__ gscratch, double* __restrict__ aligned,
                                                                    uint8_t* __restrict__ valid, int32_t* __restrict__ status)
{
    extern __shared__ double lds_[];
    typedef __attribute__((address_space(3))) double* LdsPtr;
    if (gscratch) time_align_body<double*>(gscratch + (size_t)blockIdx.x * 8 * (size_t)max_g, slam_t, slam_off, gps_t, gps_p, gps_off, max_gap, max_g, drop_masked, aligned, valid, status);
    else time_align_body<LdsPtr>((LdsPtr)lds_, slam_t, slam_off, gps_t, gps_p, gps_off, max_gap, max_g, drop_masked, aligned, valid, status);
}

template <class StagePtr>
__device__ __forceinline__ void time_align_body(StagePtr lds, const double* __restrict__ slam_t, const int64_t* __restrict__ slam_off,
                                                const double* __restrict__ gps_t, const double* __restrict__ gps_p,
                                                const int64_t* __restrict__ gps_off, double max_gap, int max_g, int drop_masked,
                                                double* __restrict__ aligned, uint8_t* __restrict__ valid, int32_t* __restrict__ status)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t s0 = slam_off[b], ns = slam_off[b + 1] - s0;
    const int64_t g0 = gps_off[b];
    int ng = (int)(gps_off[b + 1] - g0);
    const double* st = slam_t + s0;
    double* al = aligned + s0 * 3;
    uint8_t* va = valid + s0;
    for (int64_t i = lane; i < ns; i += ALIGN_THREADS) { al[i * 3] = NAN; al[i * 3 + 1] = NAN; al[i * 3 + 2] = NAN; va[i] = 0; }   // :331
    if (status && lane == 0) status[b] = 0;
    if (ns == 0 || ng < 2) return;                                       // :332-334
    if (ng > max_g) { if (status && lane == 0) status[b] = 1; return; }  // does not fit the LDS staging: reported, not computed
    StagePtr T = lds;                   // [ng] stamps
    StagePtr Y = T + max_g;             // [ng][3] positions
    StagePtr M = Y + 3 * (size_t)max_g; // [ng][3] second derivatives (cubic segments)
    StagePtr W = M + 3 * (size_t)max_g; // [ng] scratch: c' of the Thomas sweep / sort keys
    // ---- rows the loader would have removed (drop_masked): the kept fixes' input indices, compacted in input order into M (free
    // until the spline runs)
    if (drop_masked) {
        int kept = 0;
        for (int k0 = 0; k0 < ng; k0 += ALIGN_THREADS) {
            const int k = k0 + lane;
            const bool keep = k < ng && !(isnan(gps_p[(g0 + k) * 3]) && isnan(gps_p[(g0 + k) * 3 + 1]));
            const unsigned long long m = __ballot(keep);
            if (keep) M[kept + __popcll(m & ((1ull << lane) - 1ull))] = (double)k;
            kept += __popcll(m);
        }
        __syncthreads();
        ng = kept;
        if (ng < 2) return;                                              // :332-334 on the rows that survive the loader
    }
#define GSF_AL_SRC(k) (g0 + (drop_masked ? (int)M[k] : (k)))
    // ---- stage + order: rank of fix k = #{j : t_j < t_k or (t_j == t_k and j < k)} (stable argsort, :339); NaN stamps rank last
    bool sorted = true;
    for (int k = lane; k < ng; k += ALIGN_THREADS) {
        const double tk = gps_t[GSF_AL_SRC(k)];
        W[k] = tk;
        if (k > 0 && !(gps_t[GSF_AL_SRC(k - 1)] < tk)) sorted = false;
    }
    __syncthreads();
    sorted = (__ballot(!sorted) == 0ull);
    if (sorted) {
        for (int k = lane; k < ng; k += ALIGN_THREADS) {
            const int64_t src = GSF_AL_SRC(k);
            T[k] = W[k];
            Y[k * 3] = gps_p[src * 3]; Y[k * 3 + 1] = gps_p[src * 3 + 1]; Y[k * 3 + 2] = gps_p[src * 3 + 2];
        }
    } else {
        for (int k = lane; k < ng; k += ALIGN_THREADS) {
            const double tk = W[k];
            const bool nk = isnan(tk);
            int rank = 0;
            for (int j = 0; j < ng; ++j) {
                const double tj = W[j];
                const bool nj = isnan(tj);
                rank += (tj < tk || (!nj && nk) || ((tj == tk || (nj && nk)) && j < k)) ? 1 : 0;
            }
            const int64_t src = GSF_AL_SRC(k);
            T[rank] = tk;
            Y[rank * 3] = gps_p[src * 3]; Y[rank * 3 + 1] = gps_p[src * 3 + 1]; Y[rank * 3 + 2] = gps_p[src * 3 + 2];
        }
    }
#undef GSF_AL_SRC
    __syncthreads();
    // ---- np.unique(return_index=True): keep the first fix of every run of equal stamps (:341-346).  Compaction in place by
    // one lane -- duplicates are rare and ng is a few hundred.
    int nu = ng;
    if (!sorted) {
        if (lane == 0) {
            int w = 0;
            for (int k = 0; k < ng; ++k) {
                if (w > 0 && T[k] == T[w - 1]) continue;
                if (w != k) { T[w] = T[k]; Y[w * 3] = Y[k * 3]; Y[w * 3 + 1] = Y[k * 3 + 1]; Y[w * 3 + 2] = Y[k * 3 + 2]; }
                ++w;
            }
            W[0] = (double)w;
        }
        __syncthreads();
        nu = (int)W[0];
        __syncthreads();
    }
    if (nu < 2) return;                                                  // :343-345
    // ---- segments (:348-352), processed one after the other (usually 1-3 per track)
    int seg_s = 0;
    while (seg_s < nu) {
        // end of the segment = first k >= seg_s with T[k+1] - T[k] > max_gap (or the last fix): 64 gaps per step, first set bit of the ballot
        int seg_e = nu - 1;
        for (int k0 = seg_s; k0 + 1 < nu; k0 += ALIGN_THREADS) {
            const int k = k0 + lane;
            const unsigned long long gaps = __ballot(k + 1 < nu && (T[k + 1] - T[k] > max_gap));
            if (gaps != 0ull) { seg_e = k0 + __ffsll((long long)gaps) - 1; break; }
        }
        const int m = seg_e - seg_s + 1;
        if (m >= 2) {                                                    // :360
            bool inc = true;
            for (int k = seg_s + lane; k < seg_e; k += ALIGN_THREADS) if (!(T[k + 1] - T[k] > 1e-9)) inc = false;     // :364
            inc = (__ballot(!inc) == 0ull);
            if (inc) {
                const StagePtr x = T + seg_s;
                const StagePtr y = Y + (size_t)seg_s * 3;
                StagePtr Ms = M + (size_t)seg_s * 3;
                const bool cubic = m >= 4;                               // :362
                if (cubic) {
                    // knot second derivatives M_0..M_{m-1}:  h[i-1] M[i-1] + 2(h[i-1]+h[i]) M[i] + h[i] M[i+1] = 6 (d[i]-d[i-1]),
                    // not-a-knot ends folded into the first / last interior row.  Lanes 0..2 = components; c' is shared.
                    // right-hand sides 6 (d[i+1] - d[i]), d = divided differences: independent per row, all lanes (their divisions were
                    // most of the serial sweep's time); parked where the sweep leaves d'_i
                    const int kk = m - 2;
                    for (int i = lane; i < kk; i += ALIGN_THREADS) {
                        const double hl = x[i + 1] - x[i], hr = x[i + 2] - x[i + 1];
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            Ms[(i + 1) * 3 + c] = 6.0 * ((y[(i + 2) * 3 + c] - y[(i + 1) * 3 + c]) / hr - (y[(i + 1) * 3 + c] - y[i * 3 + c]) / hl);
                    }
                    __syncthreads();
                    // The tridiagonal solve as three prefix scans over the rows (64 per step, all lanes), instead of two serial sweeps on
                    // three lanes: with rows scaled by 1/b_i, the forward elimination's c'_i = c_i / (1 - a_i c'_{i-1}) is a Moebius
                    // recurrence (2x2 matrix prefix products), d'_i = (r_i - a_i d'_{i-1}) / (b_i - a_i c'_{i-1}) is affine in d'_{i-1}
                    // once the c' are known (one multiplier for the three components), and the back substitution
                    // M_i = d'_i - c'_i M_{i+1} is affine in reverse order.  Rows are diagonally dominant (|a| + |c| <= b / 2 inside, the
                    // not-a-knot ends included for increasing stamps), so the products stay O(1).
                    const double r0 = (x[1] - x[0]) / (x[2] - x[1]);
                    const double r1 = (x[m - 1] - x[m - 2]) / (x[m - 2] - x[m - 3]);
                    StagePtr cp = W + seg_s;                             // c'_i
                    double c_in = 0.0, d_in0 = 0.0, d_in1 = 0.0, d_in2 = 0.0;
                    for (int c0 = 0; c0 < kk; c0 += ALIGN_THREADS) {
                        const int i = c0 + lane;
                        const bool act = i < kk;
                        const int ic = act ? i : kk - 1;
                        const double hl = x[ic + 1] - x[ic], hr = x[ic + 2] - x[ic + 1];
                        double aa = hl, bb = 2.0 * (hl + hr), cc = hr;
                        if (ic == 0) { bb += aa * (1.0 + r0); cc -= aa * r0; aa = 0.0; }
                        if (ic == kk - 1) { bb += cc * (1.0 + r1); aa -= cc * r1; cc = 0.0; }
                        const double rb = fast_rcp(bb);
                        // f_i(c) = (0 c + cc/bb) / (-(aa/bb) c + 1); idle lanes carry the identity map
                        double A = act ? 0.0 : 1.0, Bm = act ? cc * rb : 0.0, Cm = act ? -(aa * rb) : 0.0, Dm = 1.0;
#define GSF_AL_MSTAGE(CTRL, RM) {                                                                                              \
                        const double oA = dpp<CTRL, RM>(1.0, A), oB = dpp0<CTRL, RM>(Bm), oC = dpp0<CTRL, RM>(Cm), oD = dpp<CTRL, RM>(1.0, Dm); \
                        const double nA = A * oA + Bm * oC, nB = A * oB + Bm * oD, nC = Cm * oA + Dm * oC, nD = Cm * oB + Dm * oD;                  \
                        A = nA; Bm = nB; Cm = nC; Dm = nD; }
                        GSF_SCAN_STAGES(GSF_AL_MSTAGE)
#undef GSF_AL_MSTAGE
                        const double cpi = (A * c_in + Bm) * fast_rcp(Cm * c_in + Dm);
                        const double cprev = prev_lane(c_in, cpi);
                        const double rden = fast_rcp(bb - aa * cprev);
                        double al = act ? -(aa * rden) : 1.0;
                        double be0 = act ? Ms[(ic + 1) * 3] * rden : 0.0, be1 = act ? Ms[(ic + 1) * 3 + 1] * rden : 0.0, be2 = act ? Ms[(ic + 1) * 3 + 2] * rden : 0.0;
#define GSF_AL_ASTAGE(CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al), o0 = dpp0<CTRL, RM>(be0), o1 = dpp0<CTRL, RM>(be1), o2 = dpp0<CTRL, RM>(be2); \
                                  be0 = al * o0 + be0; be1 = al * o1 + be1; be2 = al * o2 + be2; al = al * oa; }
                        GSF_SCAN_STAGES(GSF_AL_ASTAGE)
                        const double dp0 = al * d_in0 + be0, dp1 = al * d_in1 + be1, dp2 = al * d_in2 + be2;
                        if (act) { cp[i] = cpi; Ms[(i + 1) * 3] = dp0; Ms[(i + 1) * 3 + 1] = dp1; Ms[(i + 1) * 3 + 2] = dp2; }
                        const int L = (kk - c0 < ALIGN_THREADS) ? (kk - c0 - 1) : ALIGN_THREADS - 1;
                        c_in = lane_bcast(cpi, L); d_in0 = lane_bcast(dp0, L); d_in1 = lane_bcast(dp1, L); d_in2 = lane_bcast(dp2, L);
                    }
                    __syncthreads();
                    double x_in0 = 0.0, x_in1 = 0.0, x_in2 = 0.0;
                    for (int c0 = 0; c0 < kk; c0 += ALIGN_THREADS) {         // back substitution, rows in reverse: j = kk-1-i
                        const int j = c0 + lane;
                        const bool act = j < kk;
                        const int i = act ? kk - 1 - j : 0;
                        double al = act ? -cp[i] : 1.0;
                        double be0 = act ? Ms[(i + 1) * 3] : 0.0, be1 = act ? Ms[(i + 1) * 3 + 1] : 0.0, be2 = act ? Ms[(i + 1) * 3 + 2] : 0.0;
                        GSF_SCAN_STAGES(GSF_AL_ASTAGE)
#undef GSF_AL_ASTAGE
                        const double m0 = al * x_in0 + be0, m1 = al * x_in1 + be1, m2 = al * x_in2 + be2;
                        if (act) { Ms[(i + 1) * 3] = m0; Ms[(i + 1) * 3 + 1] = m1; Ms[(i + 1) * 3 + 2] = m2; }
                        const int L = (kk - c0 < ALIGN_THREADS) ? (kk - c0 - 1) : ALIGN_THREADS - 1;
                        x_in0 = lane_bcast(m0, L); x_in1 = lane_bcast(m1, L); x_in2 = lane_bcast(m2, L);
                    }
                    __syncthreads();
                    if (lane < 3) {
                        const int c = lane;
                        Ms[c] = Ms[3 + c] * (1.0 + r0) - Ms[6 + c] * r0;
                        Ms[(m - 1) * 3 + c] = Ms[(m - 2) * 3 + c] * (1.0 + r1) - Ms[(m - 3) * 3 + c] * r1;
                    }
                }
                __syncthreads();
                const double t0 = x[0], t1 = x[m - 1];
                for (int64_t i = lane; i < ns; i += ALIGN_THREADS) {
                    const double t = st[i];
                    if (!(t >= t0 - 1e-9 && t <= t1 + 1e-9)) continue;   // :372-373
                    double v0 = NAN, v1 = NAN, v2 = NAN;
                    if (t >= t0 && t <= t1) {
                        if (cubic) {
                            int lo = 0, hi = m - 1;                      // interval [x[lo], x[lo+1]] containing t
                            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (x[mid] <= t) lo = mid; else hi = mid; }
                            const double hh = x[lo + 1] - x[lo], A = (x[lo + 1] - t) / hh, Bc = (t - x[lo]) / hh;
                            const double ca = (A * A * A - A) * hh * hh / 6.0, cb = (Bc * Bc * Bc - Bc) * hh * hh / 6.0;
                            v0 = A * y[lo * 3] + Bc * y[(lo + 1) * 3] + ca * Ms[lo * 3] + cb * Ms[(lo + 1) * 3];
                            v1 = A * y[lo * 3 + 1] + Bc * y[(lo + 1) * 3 + 1] + ca * Ms[lo * 3 + 1] + cb * Ms[(lo + 1) * 3 + 1];
                            v2 = A * y[lo * 3 + 2] + Bc * y[(lo + 1) * 3 + 2] + ca * Ms[lo * 3 + 2] + cb * Ms[(lo + 1) * 3 + 2];
                        } else {
                            // scipy _call_linear: hi = clip(searchsorted(x, t, 'left'), 1, m-1); lo = hi-1
                            int hi = 0;
                            while (hi < m && x[hi] < t) ++hi;
                            hi = hi < 1 ? 1 : (hi > m - 1 ? m - 1 : hi);
                            const int lo = hi - 1;
                            const double dx = x[hi] - x[lo], dtq = t - x[lo];
                            v0 = (y[hi * 3] - y[lo * 3]) / dx * dtq + y[lo * 3];
                            v1 = (y[hi * 3 + 1] - y[lo * 3 + 1]) / dx * dtq + y[lo * 3 + 1];
                            v2 = (y[hi * 3 + 2] - y[lo * 3 + 2]) / dx * dtq + y[lo * 3 + 2];
                        }
                    }
                    al[i * 3] = v0; al[i * 3 + 1] = v1; al[i * 3 + 2] = v2;                       // :375
                    if (!(isnan(v0) || isnan(v1) || isnan(v2))) va[i] = 1;                          // :377-379 (only ever set, never cleared)
                }
                __syncthreads();
            }
        }
        seg_s = seg_e + 1;
    }
}

}  // namespace

extern "C" {

static int time_align_launch(gsf_ctx* ctx, const double* slam_t, const int64_t* slam_offsets, const double* gps_t, const double* gps_p,
                             const int64_t* gps_offsets, int64_t B, int32_t max_gps_per_trajectory, double max_gps_gap_threshold,
                             int drop_masked, double* aligned, uint8_t* valid, int32_t* status)
{
    GSF_REQUIRE(ctx && slam_offsets && gps_offsets && aligned && valid, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    GSF_REQUIRE(max_gps_per_trajectory >= 2 && max_gps_per_trajectory <= (1 << 24), "max_gps_per_trajectory out of range");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    size_t lds = (size_t)max_gps_per_trajectory * 8 * sizeof(double);                // T + Y(3) + M(3) + W
    double* gscratch = nullptr;
    if (max_gps_per_trajectory > 2560) {                                             // does not fit 160 KB of LDS: stage in HBM scratch
        int rc = ensure_scratch(ctx, lds * (size_t)B);
        if (rc) return rc;
        gscratch = (double*)ctx->scratch; lds = 0;
    } else if (lds > 64 * 1024) {
        GSF_HIP(hipFuncSetAttribute((const void*)time_align_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(time_align_kernel, dim3((unsigned)B), dim3(ALIGN_THREADS), lds, ctx->stream, slam_t, slam_offsets, gps_t, gps_p, gps_offsets,
                       max_gps_gap_threshold, (int)max_gps_per_trajectory, drop_masked, gscratch, aligned, valid, status);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_time_align_batch_dev(gsf_ctx* ctx, const double* slam_t, const int64_t* slam_offsets, const double* gps_t, const double* gps_p,
                             const int64_t* gps_offsets, int64_t B, int32_t max_gps_per_trajectory, double max_gps_gap_threshold,
                             double* aligned, uint8_t* valid, int32_t* status)
{
    return time_align_launch(ctx, slam_t, slam_offsets, gps_t, gps_p, gps_offsets, B, max_gps_per_trajectory, max_gps_gap_threshold, 0, aligned, valid, status);
}

int gsf_time_align_loaded_rows_batch_dev(gsf_ctx* ctx, const double* slam_t, const int64_t* slam_offsets, const double* gps_t, const double* gps_p,
                                         const int64_t* gps_offsets, int64_t B, int32_t max_gps_per_trajectory, double max_gps_gap_threshold,
                                         double* aligned, uint8_t* valid, int32_t* status)
{
    return time_align_launch(ctx, slam_t, slam_offsets, gps_t, gps_p, gps_offsets, B, max_gps_per_trajectory, max_gps_gap_threshold, 1, aligned, valid, status);
}

int gsf_time_align_batch(gsf_ctx* ctx, const double* slam_t, const int64_t* slam_offsets, const double* gps_t, const double* gps_p,
                         const int64_t* gps_offsets, int64_t B, double max_gps_gap_threshold, double* aligned, uint8_t* valid, int32_t* status)
{
    GSF_REQUIRE(ctx && slam_offsets && gps_offsets && B >= 0, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t ns = slam_offsets[B], ng = gps_offsets[B];
    GSF_REQUIRE(ns >= 0 && ng >= 0 && (ns == 0 || (slam_t && aligned && valid)) && (ng == 0 || (gps_t && gps_p)), "bad offsets / NULL arrays");
    int64_t maxg = 2;
    for (int64_t b = 0; b < B; ++b) { const int64_t g = gps_offsets[b + 1] - gps_offsets[b]; if (g > maxg) maxg = g; }
    Staging st(ctx, (size_t)ns * 33 + (size_t)ng * 32 + (size_t)(B + 1) * 16 + (size_t)B * 4, 8);
    if (st.rc()) return st.rc();
    const double* dst = st.in(slam_t, (size_t)ns); const double* dgt = st.in(gps_t, (size_t)ng); const double* dgp = st.in(gps_p, (size_t)ng * 3);
    const int64_t* dso = st.in(slam_offsets, (size_t)B + 1); const int64_t* dgo = st.in(gps_offsets, (size_t)B + 1);
    double* dal = st.out(aligned, (size_t)ns * 3); uint8_t* dva = st.out(valid, (size_t)ns); int32_t* dstat = st.out(status, (size_t)B);
    int rc = st.upload();
    if (rc) return rc;
    rc = gsf_time_align_batch_dev(ctx, dst, dso, dgt, dgp, dgo, B, (int32_t)maxg, max_gps_gap_threshold, dal, dva, dstat);
    if (rc) return rc;
    return st.finish();
}

}  // extern "C"
