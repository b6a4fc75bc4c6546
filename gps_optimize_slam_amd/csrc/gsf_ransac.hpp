// gsf_ransac.hpp -- what one hypothesis of compute_sim3_transform_robust (EKFGPSSLAM.py:404-414) is made of, shared by K2b (gsf_sim3.hip) and the
// early-exit probe of the robust chain (gsf_robust.hip): the 4-point fit, the residual of a row, the threshold test, the arg-max key.  ONE
// definition of each, so that a count formed by either kernel is the same number.
#pragma once
#include "gsf_math.hpp"

namespace {

// Fit on the `ms` sampled rows, sequential sums like the reference's np.mean / matmul on 4 rows.
__device__ __forceinline__ int32_t fit_sample(const double* __restrict__ src, const double* __restrict__ dst, int64_t i0,
                                              const int32_t* __restrict__ idx, int ms, double* R, double* t, double& s)
{
    double sc[3] = { 0, 0, 0 }, dc[3] = { 0, 0, 0 };
    for (int k = 0; k < ms; ++k) {
        const int64_t r = i0 + idx[k];
        sc[0] += src[r * 3]; sc[1] += src[r * 3 + 1]; sc[2] += src[r * 3 + 2];
        dc[0] += dst[r * 3]; dc[1] += dst[r * 3 + 1]; dc[2] += dst[r * 3 + 2];
    }
    const double n = (double)ms;
#pragma unroll
    for (int c = 0; c < 3; ++c) { sc[c] /= n; dc[c] /= n; }
    double H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, ssq = 0.0;
    for (int k = 0; k < ms; ++k) {
        const int64_t r = i0 + idx[k];
        const double a0 = src[r * 3] - sc[0], a1 = src[r * 3 + 1] - sc[1], a2 = src[r * 3 + 2] - sc[2];
        const double b0 = dst[r * 3] - dc[0], b1 = dst[r * 3 + 1] - dc[1], b2 = dst[r * 3 + 2] - dc[2];
        H[0] += a0 * b0; H[1] += a0 * b1; H[2] += a0 * b2;
        H[3] += a1 * b0; H[4] += a1 * b1; H[5] += a1 * b2;
        H[6] += a2 * b0; H[7] += a2 * b1; H[8] += a2 * b2;
        ssq += a0 * a0 + a1 * a1 + a2 * a2;
    }
    if (ms < 3) return SIM3_NONE;
    return umeyama_finalize(H, ssq, sc, dc, n, R, t, s);
}

// squared residual of a row (x, y, z) -> (d0, d1, d2) under (R, t, s): pure arithmetic.  ONE definition behind the memory-reading form below
// and the callers that hold their rows in registers (the early-exit probe), so that both form the same number.
__device__ __forceinline__ double resid2_vals(const double x, const double y, const double z, const double d0, const double d1, const double d2,
                                              const double* R, const double* t, double s)
{
    const double dx = s * (x * R[0] + y * R[1] + z * R[2]) + t[0] - d0;
    const double dy = s * (x * R[3] + y * R[4] + z * R[5]) + t[1] - d1;
    const double dz = s * (x * R[6] + y * R[7] + z * R[8]) + t[2] - d2;
    return dx * dx + dy * dy + dz * dz;
}
// ... of row r of (src, dst): several rows' loads can be in flight together
__device__ __forceinline__ double resid2(const double* __restrict__ src, const double* __restrict__ dst, int64_t r,
                                         const double* R, const double* t, double s)
{
    return resid2_vals(src[r * 3], src[r * 3 + 1], src[r * 3 + 2], dst[r * 3], dst[r * 3 + 1], dst[r * 3 + 2], R, t, s);
}
// ref :410-411 tests norm < thr, i.e. sqrt(d2) < thr.  The correctly rounded sqrt is only needed within a few ulp of the boundary:
// d2 clearly below / above thr^2 decides without it (the band is ~50x wider than the rounding of d2 and thr^2).
__device__ __forceinline__ bool within(double d2, double thr)
{
    const double t2 = thr * thr;
    if (d2 < t2 * (1.0 - 1e-14)) return thr > 0.0;
    if (!(d2 <= t2 * (1.0 + 1e-14))) return false;                          // also NaN -> false, like the comparison with sqrt(NaN)
    return sqrt(d2) < thr;
}
__device__ __forceinline__ bool is_inlier(const double* __restrict__ src, const double* __restrict__ dst, int64_t r,
                                          const double* R, const double* t, double s, double thr)
{
    return within(resid2(src, dst, r, R, t, s), thr);
}

// arg-max key of a hypothesis: highest count first, then the LOWEST trial (strict > keeps the first, ref :413); 0 = no usable hypothesis
__device__ __forceinline__ unsigned long long ransac_key(const long long count, const int trial)
{
    return ((unsigned long long)(count + 1) << 32) | (unsigned long long)(0x7fffffff - trial);
}

// The final fit on the inliers (ref :420-421) sums two passes of moments.  Both routes that form it -- K2b's block of RANSAC_FINAL_THREADS
// threads (gsf_sim3.hip: ransac_finish) and the early-exit probe's single wave for a set whose every row is an inlier (gsf_robust.hip) -- add
// one row at a time with these two functions, thread by thread in the same order, so the sums carry the same bits.
constexpr int RANSAC_FINAL_THREADS = 256;
__device__ __forceinline__ void final_moments1(double* acc, const double sx, const double sy, const double sz, const double dx, const double dy, const double dz)
{
    acc[0] += 1.0;
    acc[1] += sx; acc[2] += sy; acc[3] += sz;
    acc[4] += dx; acc[5] += dy; acc[6] += dz;
}
__device__ __forceinline__ void final_moments2(double* h, const double sx, const double sy, const double sz, const double dx, const double dy, const double dz,
                                               const double* sc, const double* dc)
{
    const double a0 = sx - sc[0], a1 = sy - sc[1], a2 = sz - sc[2];
    const double b0 = dx - dc[0], b1 = dy - dc[1], b2 = dz - dc[2];
    h[0] += a0 * b0; h[1] += a0 * b1; h[2] += a0 * b2;
    h[3] += a1 * b0; h[4] += a1 * b1; h[5] += a1 * b2;
    h[6] += a2 * b0; h[7] += a2 * b1; h[8] += a2 * b2;
    h[9] += a0 * a0 + a1 * a1 + a2 * a2;
}

}  // namespace
