// gsf_gpsfilter.hip -- next-3: the polynomial RANSAC of filter_gps_outliers_ransac (EKFGPSSLAM.py:136-247).
//
// The reference fits, per window and per coordinate axis, make_pipeline(PolynomialFeatures(d), RANSACRegressor(min_samples,
// residual_threshold, max_trials)) on (t, coordinate) and keeps the rows that are inliers on every axis.  One "problem" here is
// one such RANSACRegressor.fit.  The random sample sets are FED by the host (drawn with scikit-learn's own sampler on NumPy's
// legacy global RNG, so a seeded run consumes the stream exactly like the reference -- the same construction as K2b); the kernel
//   1. scores every fed trial in parallel, one thread per trial: LinearRegression on the sampled rows (features t..t^d and the
//      target centred by the subset means, least squares by modified Gram-Schmidt), |y - prediction| <= threshold over all rows,
//      inlier count and the R^2 of the subset model on its inliers (two-pass, like sklearn.metrics.r2_score);
//   2. walks the trials in order with scikit-learn's acceptance rule (fewer inliers than the best -> skip; equal count and lower
//      score -> skip; NaN scores compare false, as in Python) and its dynamic trial count
//      ceil(log(1 - p) / log(1 - (n_in / n)^min_samples)), which also yields n_trials_ -- the number of sample sets the reference
//      would have drawn, which the host needs to put the RNG where the reference leaves it;
//   3. writes the inlier mask of the accepted model.
// One 128-thread block per problem; rows of a problem are contiguous ([total] arrays + int64 offsets[P+1]).
#include "gsf_internal.hpp"

using namespace gsf;

namespace {

constexpr int RP_THREADS = 128;        // >= max_trials
constexpr int RP_MAX_SAMPLES = 16;
constexpr int RP_MAX_DEGREE = 3;

struct PolyModel { double coef[RP_MAX_DEGREE], intercept; };

__device__ __forceinline__ double poly_predict(const PolyModel& m, int degree, double t)
{
    double acc = 0.0, tk = t;
    for (int k = 0; k < degree; ++k) { acc += m.coef[k] * tk; tk *= t; }
    return acc + m.intercept;
}

// LinearRegression(fit_intercept=True) on PolynomialFeatures(degree)(t): centre the columns t^k and y by their subset means, solve
// the least squares by modified Gram-Schmidt (the constant column is identically zero after centring: coefficient 0),
// intercept = mean(y) - sum coef_k mean(t^k).
__device__ __forceinline__ PolyModel fit_subset(const double* __restrict__ t, const double* __restrict__ y, const int32_t* __restrict__ idx,
                                                int ms, int degree)
{
    double c[RP_MAX_DEGREE][RP_MAX_SAMPLES], yy[RP_MAX_SAMPLES], mean[RP_MAX_DEGREE] = { 0, 0, 0 }, ymean = 0.0;
    for (int i = 0; i < ms; ++i) {
        const double ti = t[idx[i]];
        double tk = ti;
        for (int k = 0; k < degree; ++k) { c[k][i] = tk; mean[k] += tk; tk *= ti; }
        yy[i] = y[idx[i]]; ymean += yy[i];
    }
    const double rn = 1.0 / (double)ms;
    ymean *= rn;
    for (int k = 0; k < degree; ++k) mean[k] *= rn;
    for (int i = 0; i < ms; ++i) { yy[i] -= ymean; for (int k = 0; k < degree; ++k) c[k][i] -= mean[k]; }
    // MGS: c_k = sum_{j<=k} R[j][k] q_j, q_j stored in place of c_j;  z_j = q_j . y
    double Rm[RP_MAX_DEGREE][RP_MAX_DEGREE] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } }, z[RP_MAX_DEGREE] = { 0, 0, 0 };
    for (int k = 0; k < degree; ++k) {
        for (int j = 0; j < k; ++j) {
            double d = 0.0;
            for (int i = 0; i < ms; ++i) d += c[j][i] * c[k][i];
            Rm[j][k] = d;
            for (int i = 0; i < ms; ++i) c[k][i] -= d * c[j][i];
        }
        double nn = 0.0;
        for (int i = 0; i < ms; ++i) nn += c[k][i] * c[k][i];
        nn = sqrt(nn);
        Rm[k][k] = nn;
        const double inv = nn > 0.0 ? 1.0 / nn : 0.0;                     // a dependent column gets coefficient 0 (minimum-norm spirit)
        double d = 0.0;
        for (int i = 0; i < ms; ++i) { c[k][i] *= inv; d += c[k][i] * yy[i]; }
        z[k] = d;
    }
    PolyModel m;
    for (int k = degree - 1; k >= 0; --k) {                               // back substitution R coef = z
        double v = z[k];
        for (int j = k + 1; j < degree; ++j) v -= Rm[k][j] * m.coef[j];
        m.coef[k] = Rm[k][k] > 0.0 ? v / Rm[k][k] : 0.0;
    }
    for (int k = degree; k < RP_MAX_DEGREE; ++k) m.coef[k] = 0.0;
    double off = 0.0;
    for (int k = 0; k < degree; ++k) off += mean[k] * m.coef[k];
    m.intercept = ymean - off;
    return m;
}

// sklearn.linear_model._ransac._dynamic_max_trials
__device__ __forceinline__ double dynamic_max_trials(int n_inliers, int n_samples, int min_samples, double probability)
{
    const double EPS = 2.220446049250313e-16;                             // np.spacing(1)
    const double ratio = (double)n_inliers / (double)n_samples;
    const double nom = fmax(EPS, 1.0 - probability);
    const double denom = fmax(EPS, 1.0 - pow(ratio, (double)min_samples));
    if (nom == 1.0) return 0.0;
    if (denom == 1.0) return INFINITY;
    return fabs(ceil(log(nom) / log(denom)));
}

__global__ __launch_bounds__(RP_THREADS) void ransac_poly_kernel(const double* __restrict__ t, const double* __restrict__ y,
                                                                  const int64_t* __restrict__ offsets, const int32_t* __restrict__ sample_idx,
                                                                  int max_trials, int ms, int degree, double thr, double stop_prob,
                                                                  uint8_t* __restrict__ inlier_mask, int32_t* __restrict__ n_trials,
                                                                  int32_t* __restrict__ n_inliers, int32_t* __restrict__ status)
{
    __shared__ int sh_nin[RP_THREADS];
    __shared__ double sh_score[RP_THREADS];
    __shared__ int sh_best;
    const int64_t p = blockIdx.x;
    const int64_t i0 = offsets[p], i1 = offsets[p + 1];
    const int n = (int)(i1 - i0);
    const double* tp = t + i0; const double* yp = y + i0;
    const int tau = threadIdx.x;
    PolyModel m;
    m.coef[0] = m.coef[1] = m.coef[2] = 0.0; m.intercept = 0.0;
    if (tau < max_trials && n > 0) {
        m = fit_subset(tp, yp, sample_idx + ((int64_t)p * max_trials + tau) * ms, ms, degree);
        // |y - y_pred| <= threshold over all rows (ref loss "absolute_error"), then r2_score of the model on its inliers
        int cnt = 0; double sy = 0.0;
#pragma unroll 8                                                          // rows are wave-uniform scalar loads: keep several in flight
        for (int i = 0; i < n; ++i) {
            const double res = fabs(yp[i] - poly_predict(m, degree, tp[i]));
            if (res <= thr) { ++cnt; sy += yp[i]; }
        }
        double score = NAN;                                               // fewer than two samples: sklearn returns nan (and warns)
        if (cnt >= 2) {
            const double ym = sy / (double)cnt;
            double ss_res = 0.0, ss_tot = 0.0;
#pragma unroll 8
            for (int i = 0; i < n; ++i) {
                const double pr = poly_predict(m, degree, tp[i]);
                if (fabs(yp[i] - pr) <= thr) { ss_res += (yp[i] - pr) * (yp[i] - pr); ss_tot += (yp[i] - ym) * (yp[i] - ym); }
            }
            score = ss_tot != 0.0 ? 1.0 - ss_res / ss_tot : (ss_res == 0.0 ? 1.0 : 0.0);   // force_finite
        }
        sh_nin[tau] = cnt; sh_score[tau] = score;
    }
    __syncthreads();
    if (tau == 0) {
        // RANSACRegressor.fit's loop over the trials, in order (sklearn/linear_model/_ransac.py)
        int best = -1, best_n = 1, ntr = 0;
        double best_score = -INFINITY, max_tr = (double)max_trials;
        if (n > 0) {
            while ((double)ntr < max_tr) {
                const int k = ntr++;
                const int c = sh_nin[k];
                if (c < best_n) continue;                                 // less inliers -> skip
                const double sc = sh_score[k];
                if (c == best_n && sc < best_score) continue;             // same number of inliers but worse score -> skip (nan: false)
                best = k; best_n = c; best_score = sc;
                max_tr = fmin(max_tr, dynamic_max_trials(best_n, n, ms, stop_prob));
            }
        }
        sh_best = best;
        n_trials[p] = ntr; n_inliers[p] = best >= 0 ? best_n : 0; status[p] = best >= 0 ? 0 : 1;
    }
    __syncthreads();
    const int best = sh_best;
    if (best < 0) { for (int i = tau; i < n; i += RP_THREADS) inlier_mask[i0 + i] = 0; return; }
    // every thread re-fits the accepted sample set (6 rows) and marks its share of the rows
    const PolyModel mb = fit_subset(tp, yp, sample_idx + ((int64_t)p * max_trials + best) * ms, ms, degree);
    for (int i = tau; i < n; i += RP_THREADS) inlier_mask[i0 + i] = fabs(yp[i] - poly_predict(mb, degree, tp[i])) <= thr ? 1 : 0;
}

}  // namespace

extern "C" {

int gsf_ransac_poly_batch_dev(gsf_ctx* ctx, const double* t, const double* y, const int64_t* offsets, int64_t P, const int32_t* sample_idx,
                              int32_t max_trials, int32_t min_samples, int32_t degree, double residual_threshold, double stop_probability,
                              uint8_t* inlier_mask, int32_t* n_trials, int32_t* n_inliers, int32_t* status)
{
    GSF_REQUIRE(ctx && offsets && inlier_mask && n_trials && n_inliers && status, "NULL argument");
    GSF_REQUIRE(P >= 0 && P <= 0x7fffffff, "bad P");
    GSF_REQUIRE(max_trials >= 1 && max_trials <= RP_THREADS, "max_trials must be in [1,128]");
    GSF_REQUIRE(min_samples >= 1 && min_samples <= RP_MAX_SAMPLES, "min_samples must be in [1,16]");
    GSF_REQUIRE(degree >= 1 && degree <= RP_MAX_DEGREE, "polynomial degree must be in [1,3]");
    GSF_REQUIRE(sample_idx, "sample_idx is NULL");
    if (P == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(ransac_poly_kernel, dim3((unsigned)P), dim3(RP_THREADS), 0, ctx->stream, t, y, offsets, sample_idx, (int)max_trials,
                       (int)min_samples, (int)degree, residual_threshold, stop_probability, inlier_mask, n_trials, n_inliers, status);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // extern "C"
