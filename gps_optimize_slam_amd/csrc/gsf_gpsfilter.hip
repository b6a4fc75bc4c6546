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
#include "gsf_mt19937.hpp"
#include "gsf_wave_common.hpp"   // wave_sum on DPP (no LDS round trips)

using namespace gsf;

namespace {

constexpr int RP_MAX_SAMPLES = 16;
constexpr int RP_MAX_DEGREE = 3;

template <int MAXD> struct PolyModelT { double coef[MAXD], intercept; };
typedef PolyModelT<RP_MAX_DEGREE> PolyModel;

template <int MAXD>
__device__ __forceinline__ double poly_predict(const PolyModelT<MAXD>& m, int degree, double t)
{
    // (compile-time bounds: a loop to the run-time degree indexes m.coef dynamically, which put every model of the chain kernel into scratch
    // memory -- each prediction then started with dependent trips to it.  The terms k < degree are formed exactly as that loop formed them.)
    double acc = 0.0, tk = t;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) { acc = (k < degree) ? (acc + m.coef[k] * tk) : acc; tk *= t; }
    return acc + m.intercept;
}

// LinearRegression(fit_intercept=True) on PolynomialFeatures(degree)(t): centre the columns t^k and y by their subset means, solve
// the least squares by modified Gram-Schmidt (the constant column is identically zero after centring: coefficient 0),
// intercept = mean(y) - sum coef_k mean(t^k).
// MAXD / MAXS size the per-thread arrays (3 / 16 for the kernels of the hot path, 8 / 64 for the wide kernel); REORTH: a second
// Gram-Schmidt sweep per column ("twice is enough"), for the higher degrees whose centred power columns are nearly dependent.
template <int MAXD, int MAXS, bool REORTH>
__device__ __forceinline__ PolyModelT<MAXD> fit_subset_t(const double* __restrict__ t, const double* __restrict__ y, const int32_t* __restrict__ idx,
                                                         int ms, int degree, int ystride = 1)
{
    double c[MAXD][MAXS], yy[MAXS], mean[MAXD], ymean = 0.0;
    for (int k = 0; k < MAXD; ++k) mean[k] = 0.0;
    for (int i = 0; i < ms; ++i) {
        const double ti = t[idx[i]];
        double tk = ti;
        for (int k = 0; k < degree; ++k) { c[k][i] = tk; mean[k] += tk; tk *= ti; }
        yy[i] = y[(int64_t)idx[i] * ystride]; ymean += yy[i];
    }
    const double rn = 1.0 / (double)ms;
    ymean *= rn;
    for (int k = 0; k < degree; ++k) mean[k] *= rn;
    for (int i = 0; i < ms; ++i) { yy[i] -= ymean; for (int k = 0; k < degree; ++k) c[k][i] -= mean[k]; }
    // MGS: c_k = sum_{j<=k} R[j][k] q_j, q_j stored in place of c_j;  z_j = q_j . y
    double Rm[MAXD][MAXD], z[MAXD];
    for (int k = 0; k < MAXD; ++k) { z[k] = 0.0; for (int j = 0; j < MAXD; ++j) Rm[k][j] = 0.0; }
    for (int k = 0; k < degree; ++k) {
        for (int j = 0; j < k; ++j) {
            double d = 0.0;
            for (int i = 0; i < ms; ++i) d += c[j][i] * c[k][i];
            Rm[j][k] = d;
            for (int i = 0; i < ms; ++i) c[k][i] -= d * c[j][i];
        }
        if (REORTH) {
            for (int j = 0; j < k; ++j) {
                double d = 0.0;
                for (int i = 0; i < ms; ++i) d += c[j][i] * c[k][i];
                Rm[j][k] += d;
                for (int i = 0; i < ms; ++i) c[k][i] -= d * c[j][i];
            }
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
    PolyModelT<MAXD> m;
    for (int k = degree - 1; k >= 0; --k) {                               // back substitution R coef = z
        double v = z[k];
        for (int j = k + 1; j < degree; ++j) v -= Rm[k][j] * m.coef[j];
        m.coef[k] = Rm[k][k] > 0.0 ? v / Rm[k][k] : 0.0;
    }
    for (int k = degree; k < MAXD; ++k) m.coef[k] = 0.0;
    double off = 0.0;
    for (int k = 0; k < degree; ++k) off += mean[k] * m.coef[k];
    m.intercept = ymean - off;
    return m;
}
// The same fit with every array in REGISTERS: fit_subset_t's c[MAXD][MAXS] / yy[MAXS] are indexed by run-time loop bounds (ms, degree), which
// puts them in scratch memory -- 656 bytes per lane in the chain kernel, and the two fits of a window-axis problem (the batch's models, the
// winner's model again) were 51 % of its time (23 k + 27 k of 98 k cycles, in-kernel clocks, gpurun_out/r5h/pf_timing.log).  Here every loop
// has a compile-time bound and is unrolled; rows >= ms and columns >= degree hold zeros, which the sums take in as exact no-ops (x + 0 * 0 = x),
// so the operations that matter happen in fit_subset_t's order: the same bits.
template <int MAXD, int MAXS>
__device__ __forceinline__ PolyModelT<MAXD> fit_subset_regs(const double* __restrict__ t, const double* __restrict__ y, const int32_t* __restrict__ idx,
                                                            const int ms, const int degree, const int ystride)
{
    double c[MAXD][MAXS], yy[MAXS], mean[MAXD], ymean = 0.0;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) mean[k] = 0.0;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const bool on = i < ms;
        const int32_t ix = idx[on ? i : 0];
        const double ti = t[ix], yv = y[(int64_t)ix * ystride];
        double tk = ti;
#pragma unroll
        for (int k = 0; k < MAXD; ++k) {
            const bool use = on && k < degree;
            c[k][i] = use ? tk : 0.0;
            mean[k] += use ? tk : 0.0;
            tk *= ti;
        }
        yy[i] = on ? yv : 0.0; ymean += on ? yv : 0.0;
    }
    const double rn = 1.0 / (double)ms;
    ymean *= rn;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) mean[k] *= rn;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const bool on = i < ms;
        yy[i] = on ? yy[i] - ymean : 0.0;
#pragma unroll
        for (int k = 0; k < MAXD; ++k) c[k][i] = (on && k < degree) ? c[k][i] - mean[k] : 0.0;
    }
    double Rm[MAXD][MAXD], z[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
        z[k] = 0.0;
#pragma unroll
        for (int j = 0; j < MAXD; ++j) Rm[k][j] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
        if (k < degree) {                                                 // (uniform over the wave: degree is a launch parameter)
#pragma unroll
            for (int j = 0; j < k; ++j) {
                double d = 0.0;
#pragma unroll
                for (int i = 0; i < MAXS; ++i) d += c[j][i] * c[k][i];
                Rm[j][k] = d;
#pragma unroll
                for (int i = 0; i < MAXS; ++i) c[k][i] -= d * c[j][i];
            }
            double nn = 0.0;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) nn += c[k][i] * c[k][i];
            nn = sqrt(nn);
            Rm[k][k] = nn;
            const double inv = nn > 0.0 ? 1.0 / nn : 0.0;
            double d = 0.0;
#pragma unroll
            for (int i = 0; i < MAXS; ++i) { c[k][i] *= inv; d += c[k][i] * yy[i]; }
            z[k] = d;
        }
    }
    PolyModelT<MAXD> m;
#pragma unroll
    for (int k = MAXD - 1; k >= 0; --k) {
        double v = z[k];
#pragma unroll
        for (int j = k + 1; j < MAXD; ++j) if (j < degree) v -= Rm[k][j] * m.coef[j];
        m.coef[k] = (k < degree && Rm[k][k] > 0.0) ? v / Rm[k][k] : 0.0;
    }
    double off = 0.0;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) if (k < degree) off += mean[k] * m.coef[k];
    m.intercept = ymean - off;
    return m;
}
// REGS: the register form -- for the chain kernel, one wave per SIMD with registers to spare; the thread-per-trial kernels of the fed-sample
// route run several waves per SIMD and keep the compact scratch form (90 instead of 216 registers) unless GSF_POLY_FIT_REGS says otherwise
#ifndef GSF_POLY_FIT_REGS
#define GSF_POLY_FIT_REGS 0
#endif
template <bool REGS = (GSF_POLY_FIT_REGS != 0)>
__device__ __forceinline__ PolyModel fit_subset(const double* __restrict__ t, const double* __restrict__ y, const int32_t* __restrict__ idx,
                                                int ms, int degree, int ystride = 1)
{
    if (!REGS) return fit_subset_t<RP_MAX_DEGREE, RP_MAX_SAMPLES, false>(t, y, idx, ms, degree, ystride);
    if (ms <= 8) return fit_subset_regs<RP_MAX_DEGREE, 8>(t, y, idx, ms, degree, ystride);
    return fit_subset_regs<RP_MAX_DEGREE, RP_MAX_SAMPLES>(t, y, idx, ms, degree, ystride);
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

constexpr int RP_MAX_TRIALS = 1024;
// THREADS >= max_trials: one thread per trial.  (Trials strided over a fixed block in a loop -- even a template-unrolled one -- made
// the two inlined fit_subset copies spill twice as much to scratch: 1.26 -> 2.15 ms per 30 000 problems.  A wider block for more than
// 128 trials keeps the single-pass shape.)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void ransac_poly_kernel(const double* __restrict__ t, const double* __restrict__ y,
                                                               const int64_t* __restrict__ offsets, const int32_t* __restrict__ sample_idx,
                                                               int max_trials, int ms, int degree, double thr, double stop_prob,
                                                               uint8_t* __restrict__ inlier_mask, int32_t* __restrict__ n_trials,
                                                               int32_t* __restrict__ n_inliers, int32_t* __restrict__ status)
{
    __shared__ int sh_nin[THREADS];
    __shared__ double sh_score[THREADS];
    __shared__ int sh_best;
    const int64_t p = blockIdx.x;
    const int64_t i0 = offsets[p], i1 = offsets[p + 1];
    const int n = (int)(i1 - i0);
    const double* tp = t + i0; const double* yp = y + i0;
    const int tau = threadIdx.x;
    PolyModel m;
    m.coef[0] = m.coef[1] = m.coef[2] = 0.0; m.intercept = 0.0;
    // a caller-fed sample set naming a row outside [0, n) is never accepted and is flagged (status bit 1)
    bool in_range = true;
    if (tau < max_trials && n > 0) {
        const int32_t* ix = sample_idx + ((int64_t)p * max_trials + tau) * ms;
        for (int k = 0; k < ms; ++k) in_range = in_range && ix[k] >= 0 && ix[k] < n;
    }
    if (tau < max_trials && n > 0 && in_range) {
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
    } else if (tau < max_trials && n > 0) { sh_nin[tau] = -1; sh_score[tau] = NAN; }
    const bool any_bad = __syncthreads_or(in_range ? 0 : 1) != 0;
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
        n_trials[p] = ntr; n_inliers[p] = best >= 0 ? best_n : 0; status[p] = (best >= 0 ? 0 : 1) | (any_bad ? 2 : 0);
    }
    __syncthreads();
    const int best = sh_best;
    if (best < 0) { for (int i = tau; i < n; i += THREADS) inlier_mask[i0 + i] = 0; return; }
    // every thread re-fits the accepted sample set (6 rows) and marks its share of the rows
    const PolyModel mb = fit_subset(tp, yp, sample_idx + ((int64_t)p * max_trials + best) * ms, ms, degree);
    for (int i = tau; i < n; i += THREADS) inlier_mask[i0 + i] = fabs(yp[i] - poly_predict(mb, degree, tp[i])) <= thr ? 1 : 0;
}

// The same problem for configurations beyond the fast kernel's per-thread arrays and one-pass block (polynomial_degree up to 8,
// min_samples up to 64, any max_trials): trials strided over a 256-thread block, per-trial (inlier count, score) parked in a global
// slab, the same acceptance walk.  Any CONFIG scikit-learn accepts in practice runs; speed is secondary here.
constexpr int RPW_MAX_DEGREE = 8, RPW_MAX_SAMPLES = 64, RPW_THREADS = 256;
__global__ __launch_bounds__(RPW_THREADS) void ransac_poly_wide_kernel(const double* __restrict__ t, const double* __restrict__ y,
                                                                        const int64_t* __restrict__ offsets, const int32_t* __restrict__ sample_idx,
                                                                        int max_trials, int ms, int degree, double thr, double stop_prob,
                                                                        int32_t* __restrict__ tr_cnt, double* __restrict__ tr_score,
                                                                        uint8_t* __restrict__ inlier_mask, int32_t* __restrict__ n_trials,
                                                                        int32_t* __restrict__ n_inliers, int32_t* __restrict__ status)
{
    __shared__ int sh_best;
    const int64_t p = blockIdx.x;
    const int64_t i0 = offsets[p], i1 = offsets[p + 1];
    const int n = (int)(i1 - i0);
    const double* tp = t + i0; const double* yp = y + i0;
    int32_t* cntp = tr_cnt + p * (int64_t)max_trials; double* scp = tr_score + p * (int64_t)max_trials;
    bool bad = false;
    for (int tau = threadIdx.x; tau < max_trials && n > 0; tau += RPW_THREADS) {
        const int32_t* ix = sample_idx + ((int64_t)p * max_trials + tau) * ms;
        bool in_range = true;
        for (int k = 0; k < ms; ++k) in_range = in_range && ix[k] >= 0 && ix[k] < n;
        if (!in_range) { bad = true; cntp[tau] = -1; scp[tau] = NAN; continue; }
        const PolyModelT<RPW_MAX_DEGREE> m = fit_subset_t<RPW_MAX_DEGREE, RPW_MAX_SAMPLES, true>(tp, yp, ix, ms, degree);
        int cnt = 0; double sy = 0.0;
        for (int i = 0; i < n; ++i) {
            const double res = fabs(yp[i] - poly_predict(m, degree, tp[i]));
            if (res <= thr) { ++cnt; sy += yp[i]; }
        }
        double score = NAN;
        if (cnt >= 2) {
            const double ym = sy / (double)cnt;
            double ss_res = 0.0, ss_tot = 0.0;
            for (int i = 0; i < n; ++i) {
                const double pr = poly_predict(m, degree, tp[i]);
                if (fabs(yp[i] - pr) <= thr) { ss_res += (yp[i] - pr) * (yp[i] - pr); ss_tot += (yp[i] - ym) * (yp[i] - ym); }
            }
            score = ss_tot != 0.0 ? 1.0 - ss_res / ss_tot : (ss_res == 0.0 ? 1.0 : 0.0);
        }
        cntp[tau] = cnt; scp[tau] = score;
    }
    const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;             // (also orders the slab writes before the walk below)
    if (threadIdx.x == 0) {
        int best = -1, best_n = 1, ntr = 0;
        double best_score = -INFINITY, max_tr = (double)max_trials;
        if (n > 0) {
            while ((double)ntr < max_tr) {
                const int k = ntr++;
                const int c = cntp[k];
                if (c < best_n) continue;
                const double sc = scp[k];
                if (c == best_n && sc < best_score) continue;
                best = k; best_n = c; best_score = sc;
                max_tr = fmin(max_tr, dynamic_max_trials(best_n, n, ms, stop_prob));
            }
        }
        sh_best = best;
        n_trials[p] = ntr; n_inliers[p] = best >= 0 ? best_n : 0; status[p] = (best >= 0 ? 0 : 1) | (any_bad ? 2 : 0);
    }
    __syncthreads();
    const int best = sh_best;
    if (best < 0) { for (int i = threadIdx.x; i < n; i += RPW_THREADS) inlier_mask[i0 + i] = 0; return; }
    const PolyModelT<RPW_MAX_DEGREE> mb = fit_subset_t<RPW_MAX_DEGREE, RPW_MAX_SAMPLES, true>(tp, yp, sample_idx + ((int64_t)p * max_trials + best) * ms, ms, degree);
    for (int i = threadIdx.x; i < n; i += RPW_THREADS) inlier_mask[i0 + i] = fabs(yp[i] - poly_predict(mb, degree, tp[i])) <= thr ? 1 : 0;
}

// score of one fed trial (steps 1 of the header comment) for rows with a stride (AoS position rows)
__device__ __forceinline__ void score_trial(const double* __restrict__ tp, const double* __restrict__ yp, int ystride, int n, const int32_t* idx, int ms,
                                            int degree, double thr, int& cnt_out, double& score_out)
{
    const PolyModel m = fit_subset(tp, yp, idx, ms, degree, ystride);
    int cnt = 0; double sy = 0.0;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
        const double yi = yp[(int64_t)i * ystride];
        const double res = fabs(yi - poly_predict(m, degree, tp[i]));
        if (res <= thr) { ++cnt; sy += yi; }
    }
    double score = NAN;
    if (cnt >= 2) {
        const double ym = sy / (double)cnt;
        double ss_res = 0.0, ss_tot = 0.0;
#pragma unroll 4
        for (int i = 0; i < n; ++i) {
            const double yi = yp[(int64_t)i * ystride];
            const double pr = poly_predict(m, degree, tp[i]);
            if (fabs(yi - pr) <= thr) { ss_res += (yi - pr) * (yi - pr); ss_tot += (yi - ym) * (yi - ym); }
        }
        score = ss_tot != 0.0 ? 1.0 - ss_res / ss_tot : (ss_res == 0.0 ? 1.0 : 0.0);
    }
    cnt_out = cnt; score_out = score;
}

// The WHOLE pre-filter of one GNSS log as one device chain (ref :183-247 sliding windows, :148-182 the single global window): for
// every window (row range, found by the host from the stamps) and every coordinate axis in the reference's order -- draw max_trials
// sample sets from the log's legacy MT19937 stream exactly like scikit-learn's sample_without_replacement (permutation(n)[:k] for
// 0.01 < k/n < 0.99), score them, walk scikit-learn's acceptance rule, REWIND the stream to where n_trials_ draws leave it, and
// either AND the axis mask into the window mask or -- no consensus set: the reference's exception -- drop the window and skip its
// remaining axes (they consume nothing).  keep[] = OR over the successful windows.  One wave per log.
constexpr int CH_MAX_TRIALS = 1024;
constexpr int PF_TILE = 4;            // row iterations of a window held in registers by the chain kernel (64 rows each)
#ifdef GSF_PF_TIMING
// diagnostic build only (make pf_timing, tools/experiments/prefilter_timing.py): shader-clock totals per phase of the chain, written by lane 0 into
// log_info[b * 16 + k] (the caller passes 16 ints per log): 0 snapshot, 1 draw, 2 fit, 3 score, 4 walk, 5 rewind, 6 final model + mask, 7 fold,
// 8 window search, 9 problems, 10 whole kernel
#define PF_T0() long long pf_t_ = clock64()
#define PF_ADD(k) do { const long long n_ = clock64(); pf_acc[k] += n_ - pf_t_; pf_t_ = n_; } while (0)
#else
#define PF_T0() do { } while (0)
#define PF_ADD(k) do { } while (0)
#endif
// Where the windows of a log come from.  mode 0: the caller lists them (win_rows / win_offsets: the host walked the stamps).  mode 1: the
// kernel walks the stamps itself the way the reference does (ref :199-234: windows [w0, w0 + width) advanced by `stride`, one extra tail
// window ending just past the last stamp; a window with fewer than `need` rows is skipped) -- sorted stamps only, so that every window is
// one row range (an unsorted log is flagged 3 and left to the host route).  mode 2: one window = the whole log (ref :148-182).
// Modes 1 / 2 also take the reference's early-outs: filtering disabled -> every row kept, nothing drawn (:139-141); fewer than `need` rows
// -> the same (:144-146).
struct WinGen { int32_t mode, need; double width, stride; int32_t enabled, max_windows, first_batch, speculate, miss_batch; };
__global__ __launch_bounds__(64) void gps_prefilter_chain_kernel(const double* __restrict__ t, const double* __restrict__ pos, const int64_t* __restrict__ offsets,
                                                                 const int32_t* __restrict__ counts,
                                                                 const int32_t* __restrict__ win_rows, const int64_t* __restrict__ win_offsets,
                                                                 int max_trials, int ms, int degree, double thr, double stop_prob, int jseq_elems,
                                                                 uint32_t* __restrict__ state, uint8_t* __restrict__ keep, int32_t* __restrict__ win_status,
                                                                 int32_t* __restrict__ log_status, WinGen gen, int32_t* __restrict__ log_info)
{
    __shared__ uint32_t mt[MT_N + 1];
    __shared__ uint32_t snap[MT_N + 1];
    __shared__ PolyModel sh_model[64];                                    // models of the batch being scored (batches are 4, 8, 16, 32, then 64 trials)
    // dynamic LDS, sized by max_trials (round 5: the per-trial arrays were static arrays of CH_MAX_TRIALS entries, 16 KB a block whatever the
    // CONFIG's 50 trials, and with a 40 KB sampler buffer only TWO one-wave blocks fitted a CU -- 1 000 logs ran as two rounds):
    // score[max_trials] (double), idx[max_trials * ms], end[max_trials], nin[max_trials] (int32), then the swap partners [jseq_elems] (uint16)
    extern __shared__ double dyn[];
    double* sh_score = dyn;
    int32_t* sh_idx = (int32_t*)(sh_score + max_trials);
    int32_t* sh_end = sh_idx + (size_t)max_trials * ms;
    int* sh_nin = (int*)(sh_end + max_trials);
    uint16_t* jseq = (uint16_t*)(sh_nin + max_trials);
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
#ifdef GSF_PF_TIMING
    long long pf_acc[20] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const long long pf_start = clock64();
#endif
    const int64_t r_base = offsets[b];
    const int n_log = counts ? counts[b] : (int)(offsets[b + 1] - r_base);   // (counts: logs in fixed-stride slots, e.g. after the loader's rows were compacted)
    uint32_t* st = state + b * MT_STATE_WORDS;
    if (gen.mode != 0 && (gen.enabled == 0 || n_log < gen.need)) {        // ref :139-146: the log passes unfiltered, the generator is not touched
        for (int i = lane; i < n_log; i += 64) keep[r_base + i] = 1;
        if (lane == 0) { log_status[b] = 0; if (log_info) { log_info[b * 2] = 0; log_info[b * 2 + 1] = 0; } }
        return;
    }
    for (int i = lane; i < MT_N; i += 64) mt[i] = st[i];
    int pos_mt = (int)st[MT_N];
    uint8_t* kp = keep + r_base;                                           // the log's keep bytes
    for (int i = lane; i < n_log; i += 64) kp[i] = 0;
    __syncthreads();
#ifdef GSF_PF_TIMING
    pf_acc[16] = clock64() - pf_start;
#endif
    int lstat = 0;
    // one window (rows r0 .. r1-1 of the log): 0 ok / 1 no consensus set / 2 fewer rows than min_samples (not processed) / 3 not handled
    auto run_window = [&](const int r0, const int r1) -> int {
        const int n = r1 - r0;
        int wstat = 0;
        if (n < ms || r0 < 0 || r1 > n_log) return 2;                     // ref :209: too few rows, window not processed
        // scikit-learn's sampler takes the permutation route only for 0.01 < k/n < 0.99.  Above that range it samples by reservoir: rows
        // 0 .. k-1, then one randint per FURTHER row -- and with min_samples <= 16 a ratio >= 0.99 means n == k: no further row, no draw,
        // every trial's sample is rows 0 .. k-1 (a window of exactly min_samples fixes: the tail of a log, a thinned log).  Below the range
        // (tracking selection, n > 100 k) the log is flagged and left to the host-drawn path.
        const double ratio = (double)ms / (double)n;
        const bool identity = n == ms;
        if (!identity && (!(ratio > 0.01 && ratio < 0.99) || n > jseq_elems)) { lstat = 2; return 3; }
#ifdef GSF_PF_TIMING
        const long long pf_w0 = clock64();
#endif
        const double* tp = t + r_base + r0;
        // window mask lives in keep[] itself as bit 1 (AND over axes), folded into bit 0 (OR over windows) when the window succeeds
        for (int i = lane; i < n; i += 64) kp[r0 + i] |= 2;
        // the window's rows in registers when they fit (up to 64 x PF_TILE = 512 rows; lane l holds rows l, l + 64, ...): the stamps once per
        // window, an axis's values once per axis -- every trial's two scoring passes and the final mask read them from there instead of memory
        const bool in_regs = n <= 64 * PF_TILE;
        double tv[PF_TILE], yv[PF_TILE], y3[3][PF_TILE];
        {
            // one batch of loads per window: the stamps and all three axes (clamped rows, no branch: every load in flight at once)
            const double* p3 = pos + (r_base + r0) * 3;
#pragma unroll
            for (int k = 0; k < PF_TILE; ++k) {
                const int i = k * 64 + lane, ic = in_regs ? (i < n ? i : n - 1) : 0;
                tv[k] = tp[ic]; y3[0][k] = p3[(int64_t)ic * 3]; y3[1][k] = p3[(int64_t)ic * 3 + 1]; y3[2][k] = p3[(int64_t)ic * 3 + 2]; yv[k] = 0.0;
            }
        }
        // ---- speculative pass (round 5).  On a clean window every axis's RANSACRegressor stops after its FIRST trial (the sample's model
        // counts enough rows for _dynamic_max_trials <= 1), so the three axes consume three CONSECUTIVE trials of the stream.  Draw those three
        // in one go, fit the three models on three lanes, count their inliers, and evaluate the stopping rule for all three at once: an axis
        // whose first trial ends its loop is finished exactly as the sequential walk would finish it (one trial considered: the R^2 score
        // cannot matter, the accepted model is that trial's).  The first axis that needs more trials -- and every axis after it -- goes
        // through the sequential loop below, from the stream position where that axis starts.  Same words out, ~1/4 of the time per clean window.
#ifdef GSF_PF_TIMING
        pf_acc[17] += clock64() - pf_w0 + (long long)(tv[0] == 1.25 ? 1 : 0) + (long long)(y3[2][0] == 1.25 ? 1 : 0);   // (the loads have landed)
#endif
        const bool spec_ok = gen.speculate != 0 && !identity && in_regs && max_trials >= 3;
        int miss_batch = 0;                                               // > 0: the speculative pass has just missed at this axis
        for (int ax = 0; ax < 3 && wstat == 0; ++ax) {
          if (spec_ok && ax != 2) {                                       // (a pass over the last axis alone would save nothing)
            const int na = 3 - ax;                                        // axes ax .. 2: na consecutive trials
            PF_T0();
            for (int i = lane; i <= MT_N; i += 64) snap[i] = (i < MT_N) ? mt[i] : (uint32_t)pos_mt;
            __syncthreads();
            mt_draw_choice(mt, pos_mt, n, na, ms, jseq, jseq_elems, sh_idx, sh_end, lane);
            PF_ADD(12);
            const double* y0 = pos + (r_base + r0) * 3 + ax;
            if (lane < na) sh_model[lane] = fit_subset<true>(tp, y0 + lane, sh_idx + (size_t)lane * ms, ms, degree, 3);
            __syncthreads();
            PF_ADD(13);
            int cnt3[3] = { 0, 0, 0 };                                    // by ABSOLUTE axis (constant register indices)
            unsigned inl[PF_TILE];                                        // bit A: the row is an inlier of axis A's model
#pragma unroll
            for (int k = 0; k < PF_TILE; ++k) inl[k] = 0u;
#pragma unroll
            for (int A = 0; A < 3; ++A) {
                if (A >= ax) {                                            // wave-uniform
                    const PolyModel m = sh_model[A - ax];
#pragma unroll
                    for (int k = 0; k < PF_TILE; ++k) {
                        if (k * 64 < n) {                                 // wave-uniform
                            const bool in = (k * 64 + lane < n) && fabs(y3[A][k] - poly_predict(m, degree, tv[k])) <= thr;
                            cnt3[A] += __popcll(__ballot(in));
                            inl[k] |= in ? (1u << A) : 0u;
                        }
                    }
                }
            }
            PF_ADD(14);
            // the walk over ONE trial: skipped if it counts no row (c < 1: the loop would go on), else accepted and max_trials shrinks
            const int ax_l = ax + lane;                                   // lane l speaks for axis ax + l
            const int c_l = ax_l == 0 ? cnt3[0] : (ax_l == 1 ? cnt3[1] : cnt3[2]);
            const double mt_l = c_l >= 1 ? fmin((double)max_trials, dynamic_max_trials(c_l >= 1 ? c_l : 1, n, ms, stop_prob)) : INFINITY;
            const unsigned long long okm = __ballot(mt_l <= 1.0) & ((1ull << na) - 1ull);   // n_trials_ = 1 >= max_trials: fit() leaves its loop
            const int n_ok = (okm & 1ull) ? ((okm & 2ull) ? ((okm & 4ull) ? 3 : 2) : 1) : 0;   // leading axes that stop after their first trial
            PF_ADD(15);
            const unsigned need_bits = ((1u << n_ok) - 1u) << ax;
#pragma unroll
            for (int k = 0; k < PF_TILE; ++k) {
                const int i = k * 64 + lane;
                if (k * 64 < n && i < n && (~inl[k] & need_bits) != 0u) kp[r0 + i] &= (uint8_t)~2u;
            }
            if (n_ok < na) {                                              // back to where axis ax + n_ok starts
                // its first trial is known now, and with it how many trials RANSACRegressor will want unless a later one counts more rows:
                // the sequential walk below opens with a batch of that many instead of growing to it by rounds of 1, 2, 4
                const double want = lane_bcast(mt_l, n_ok);
                const int cap = gen.miss_batch;                           // (a poor first sample asks for many: a better one among the next few shrinks that)
                miss_batch = want < (double)cap ? (want > 2.0 ? (int)want : (cap < 2 ? cap : 2)) : cap;
                const int skip = n_ok > 0 ? sh_end[n_ok - 1] : 0;
                __syncthreads();
                for (int i = lane; i < MT_N; i += 64) mt[i] = snap[i];
                pos_mt = (int)snap[MT_N];
                __syncthreads();
                mt_skip(mt, pos_mt, skip, lane);
            }
            ax += n_ok;
            PF_ADD(7);
#ifdef GSF_PF_TIMING
            pf_acc[8] += n_ok; pf_acc[11] += 1;
#endif
            if (ax >= 3) break;
          }
          {
            const double* yp = pos + (r_base + r0) * 3 + ax;
#pragma unroll
            for (int k = 0; k < PF_TILE; ++k) yv[k] = ax == 0 ? y3[0][k] : (ax == 1 ? y3[1][k] : y3[2][k]);
            PF_T0();
            for (int i = lane; i <= MT_N; i += 64) snap[i] = (i < MT_N) ? mt[i] : (uint32_t)pos_mt;
            __syncthreads();
            PF_ADD(0);
            // RANSACRegressor.fit's loop over the trials, in order, with the trials drawn and scored in growing batches (8, 16, 32, ...):
            // the loop shortens max_trials as soon as a consensus set is found (typically to 4-15 of the 50), and every trial costs
            // ~600 dependent instructions of the stream walk.  The walk itself runs redundantly on every lane (wave-uniform state).
            int best = -1, best_n = 1, ntr = 0, drawn = 0, raw_base = 0, last_nb = 0;
            double best_score = -INFINITY, max_tr = (double)max_trials;
            int tbn0 = gen.first_batch;
            if (miss_batch > tbn0) tbn0 = miss_batch;
            miss_batch = 0;
            for (int tbn = tbn0; (double)ntr < max_tr; tbn = tbn < 64 ? tbn * 2 : 64) {
                const int nb = (max_trials - drawn < tbn) ? (max_trials - drawn) : tbn;
                if (nb <= 0) break;
                if (identity) {
                    for (int e = lane; e < nb * ms; e += 64) sh_idx[(size_t)drawn * ms + e] = e % ms;
                    for (int tau = lane; tau < nb; tau += 64) sh_end[drawn + tau] = 0;      // nothing consumed
                    __syncthreads();
                } else {
                    mt_draw_choice(mt, pos_mt, n, nb, ms, jseq, jseq_elems, sh_idx + (size_t)drawn * ms, sh_end + drawn, lane);
                }
                PF_ADD(1);
                // models: a lane per trial (at most 32 of them work); scores: the whole wave over the ROWS of one trial at a time -- a batch
                // is 4-64 trials of ~150 rows, and a lane walking all rows of its trial alone was 44 of a window-axis's 85 us
                for (int tau = lane; tau < nb; tau += 64) {
                    sh_model[tau] = fit_subset<true>(tp, yp, sh_idx + (size_t)(drawn + tau) * ms, ms, degree, 3);
                    sh_end[drawn + tau] += raw_base;                          // outputs consumed since the window-axis start
                }
                __syncthreads();
                PF_ADD(2);
                // (the R^2 score below is summed per lane and then across the wave; score_trial / ransac_poly_kernel of the fed-sample route
                // sum row after row on one thread.  Equal inlier counts are ordered by the score with an exact compare, so a near-tie in the
                // last ulp may resolve differently on the two routes -- scikit-learn's own order of summation is a third one; the sixteen
                // reference runs and sixty live scikit-learn cases of the tests hold no such tie.)
                for (int tau = 0; tau < nb; ++tau) {
                    const PolyModel m = sh_model[tau];
                    int cnt = 0; double sy = 0.0;
                    if (in_regs) {
#pragma unroll
                        for (int k = 0; k < PF_TILE; ++k) {
                            if (k * 64 < n) {                             // wave-uniform
                                const bool in = (k * 64 + lane < n) && fabs(yv[k] - poly_predict(m, degree, tv[k])) <= thr;
                                cnt += __popcll(__ballot(in));
                                sy += in ? yv[k] : 0.0;
                            }
                        }
                    } else {
                        for (int i0 = 0; i0 < n; i0 += 64) {
                            const int i = i0 + lane;
                            const double yi = i < n ? yp[(int64_t)i * 3] : 0.0;
                            const bool in = i < n && fabs(yi - poly_predict(m, degree, tp[i < n ? i : 0])) <= thr;
                            cnt += __popcll(__ballot(in));
                            sy += in ? yi : 0.0;
                        }
                    }
                    double score = NAN;
                    if (cnt >= 2) {                                           // wave-uniform
                        const double ym = wave_sum(sy) / (double)cnt;
                        double ss_res = 0.0, ss_tot = 0.0;
                        if (in_regs) {
#pragma unroll
                            for (int k = 0; k < PF_TILE; ++k) {
                                if (k * 64 < n) {
                                    const double pr = poly_predict(m, degree, tv[k]);
                                    const bool in = (k * 64 + lane < n) && fabs(yv[k] - pr) <= thr;
                                    ss_res += in ? (yv[k] - pr) * (yv[k] - pr) : 0.0;
                                    ss_tot += in ? (yv[k] - ym) * (yv[k] - ym) : 0.0;
                                }
                            }
                        } else {
                            for (int i0 = 0; i0 < n; i0 += 64) {
                                const int i = i0 + lane;
                                const double yi = i < n ? yp[(int64_t)i * 3] : 0.0;
                                const double pr = poly_predict(m, degree, tp[i < n ? i : 0]);
                                const bool in = i < n && fabs(yi - pr) <= thr;
                                ss_res += in ? (yi - pr) * (yi - pr) : 0.0;
                                ss_tot += in ? (yi - ym) * (yi - ym) : 0.0;
                            }
                        }
                        ss_res = wave_sum(ss_res); ss_tot = wave_sum(ss_tot);
                        score = ss_tot != 0.0 ? 1.0 - ss_res / ss_tot : (ss_res == 0.0 ? 1.0 : 0.0);
                    }
                    if (lane == 0) { sh_nin[drawn + tau] = cnt; sh_score[drawn + tau] = score; }
                }
                __syncthreads();
                PF_ADD(3);
                drawn += nb; last_nb = nb;
                raw_base = sh_end[drawn - 1];
                while ((double)ntr < max_tr && ntr < drawn) {
                    const int k = ntr++;
                    const int c = sh_nin[k];
                    if (c < best_n) continue;
                    const double sc = sh_score[k];
                    if (c == best_n && sc < best_score) continue;
                    best = k; best_n = c; best_score = sc;
                    max_tr = fmin(max_tr, dynamic_max_trials(best_n, n, ms, stop_prob));
                }
                __syncthreads();
                PF_ADD(4);
            }
            const int skip = ntr > 0 ? sh_end[ntr - 1] : 0;
            // the stream goes back to the window-axis start and forward by what n_trials_ draws consume
            for (int i = lane; i < MT_N; i += 64) mt[i] = snap[i];
            pos_mt = (int)snap[MT_N];
            __syncthreads();
            mt_skip(mt, pos_mt, skip, lane);
            PF_ADD(5);
#ifdef GSF_PF_TIMING
            pf_acc[9] += 1;
#endif
            if (best < 0) { wstat = 1; break; }                           // "RANSAC could not find a valid consensus set": the window fails (ref :228-229)
            // the winner's model: still in LDS when the winner belongs to the batch scored last (the usual case: the first batch decides), else fitted
            // again -- same function, same rows, same bits either way
            const int batch0 = drawn - last_nb;
            const PolyModel mb = best >= batch0 ? sh_model[best - batch0] : fit_subset<true>(tp, yp, sh_idx + (size_t)best * ms, ms, degree, 3);
            if (in_regs) {
#pragma unroll
                for (int k = 0; k < PF_TILE; ++k) {
                    const int i = k * 64 + lane;
                    if (k * 64 < n && i < n && !(fabs(yv[k] - poly_predict(mb, degree, tv[k])) <= thr)) kp[r0 + i] &= (uint8_t)~2u;
                }
            } else {
                for (int i = lane; i < n; i += 64) {
                    const bool in = fabs(yp[(int64_t)i * 3] - poly_predict(mb, degree, tp[i])) <= thr;
                    if (!in) kp[r0 + i] &= (uint8_t)~2u;
                }
            }
            __syncthreads();
            PF_ADD(6);
          }
        }
#ifdef GSF_PF_TIMING
        const long long pf_f0 = clock64();
#endif
        for (int i = lane; i < n; i += 64) {
            const uint8_t v = kp[r0 + i];
            kp[r0 + i] = (uint8_t)((v & 1u) | ((wstat == 0 && (v & 2u)) ? 1u : 0u));
        }
        __syncthreads();
#ifdef GSF_PF_TIMING
        pf_acc[18] += clock64() - pf_f0;
#endif
        return wstat;
    };
    int processed = 0, succeeded = 0;
    // ONE call site of run_window for the three ways the windows come (the lambda is ~70 KB of code: inlined once, not three times):
    // mode 0 = row ranges fed by the host, mode 2 = the whole log (ref :148-182), mode 1 = the sliding windows of ref :196-234 found here.
    const double* tl = t + r_base;
    double t_first = 0.0, t_last = 0.0, w0 = 0.0, w1 = 0.0;
    const bool log_regs = n_log <= 64 * PF_TILE;
    double tlv[PF_TILE];                                                  // mode 1, logs of up to 64 x PF_TILE fixes: the stamps sit in registers for the walk
#pragma unroll
    for (int k = 0; k < PF_TILE; ++k) tlv[k] = 0.0;
    if (gen.mode == 1) {
        // stamps must be sorted (every window one row range): checked first
        bool sorted = true;
        for (int i = lane; i + 1 < n_log; i += 64) sorted = sorted && tl[i] <= tl[i + 1];
        if (__ballot(!sorted) != 0ull) lstat = 3;
        else {
            t_first = tl[0]; t_last = tl[n_log - 1]; w0 = t_first;
#pragma unroll
            for (int k = 0; k < PF_TILE; ++k) { const int i = k * 64 + lane; tlv[k] = (log_regs && k * 64 < n_log) ? tl[i < n_log ? i : n_log - 1] : 0.0; }
        }
    }
    // first row with stamp >= x / > x (wave-parallel count of the rows below: the stamps are sorted)
    auto rows_below = [&](const double x, const bool strict) -> int {
        int c = 0;
        if (log_regs) {
#pragma unroll
            for (int k = 0; k < PF_TILE; ++k) {
                if (k * 64 < n_log) {                                     // wave-uniform
                    const bool below = (k * 64 + lane < n_log) && (strict ? tlv[k] < x : tlv[k] <= x);
                    c += __popcll(__ballot(below));
                }
            }
            return c;
        }
        for (int i0 = 0; i0 < n_log; i0 += 64) {
            const int i = i0 + lane;
            const bool below = i < n_log && (strict ? tl[i] < x : tl[i] <= x);
            const int k = __popcll(__ballot(below));
            c += k;
            if (k < 64) break;
        }
        return c;
    };
    int64_t w = gen.mode == 0 ? win_offsets[b] : 0;
    const int64_t w_end = gen.mode == 0 ? win_offsets[b + 1] : 1;
    int visited = 0;
    for (;;) {
        int r0 = 0, r1 = 0;
        bool run = true;
        if (gen.mode == 1) {
            if (lstat == 3 || !(w0 < t_last)) break;                      // :200
            if (++visited > gen.max_windows) { lstat = 3; break; }
            w1 = w0 + gen.width;                                          // :201
            r0 = rows_below(w0, true); r1 = rows_below(w1, true);         // rows with w0 <= t < w1 (:202)
            run = r1 - r0 >= gen.need;                                    // :204
        } else {
            if (w >= w_end) break;
            if (gen.mode == 0) { r0 = win_rows[w * 2]; r1 = win_rows[w * 2 + 1]; } else { r0 = 0; r1 = n_log; }
        }
        int ws = -1;
        if (run) ws = run_window(r0, r1);
        if (gen.mode == 0) {
            if (lane == 0) win_status[w] = ws;
            if (ws == 3) break;
            processed += ws != 2; succeeded += ws == 0;
            ++w;
        } else if (gen.mode == 2) {                                       // if the one fit raises, the log passes unfiltered
            processed = 1; succeeded = ws == 0;
            if (ws == 1) { for (int i = lane; i < n_log; i += 64) kp[i] = 1; }
            ++w;
        } else {
            if (run) {
                if (ws == 3) break;
                processed += 1; succeeded += ws == 0;
            }
            if (gen.stride <= 1e-6) {                                     // :230-232
                const int nx = rows_below(w0, false);                     // first row with t > w0
                if (nx < n_log) w0 = tl[nx]; else break;
            } else w0 += gen.stride;                                      // :233
            if (w0 >= t_last && t_last >= w1) w0 = fmax(t_first, t_last - gen.width + 1e-6);   // :234-235
        }
    }
    for (int i = lane; i < MT_N; i += 64) st[i] = mt[i];
#ifdef GSF_PF_TIMING
    pf_acc[10] = clock64() - pf_start;
    if (lane == 0 && log_info) { for (int k = 0; k < 20; ++k) log_info[b * 20 + k] = (int32_t)(pf_acc[k] > 0x7fffffff ? 0x7fffffff : pf_acc[k]); st[MT_N] = (uint32_t)pos_mt; log_status[b] = lstat; }
    (void)processed; (void)succeeded;
#else
    if (lane == 0) { st[MT_N] = (uint32_t)pos_mt; log_status[b] = lstat; if (log_info) { log_info[b * 2] = processed; log_info[b * 2 + 1] = succeeded; } }
#endif
}

}  // namespace

extern "C" {

int gsf_ransac_poly_batch_dev(gsf_ctx* ctx, const double* t, const double* y, const int64_t* offsets, int64_t P, const int32_t* sample_idx,
                              int32_t max_trials, int32_t min_samples, int32_t degree, double residual_threshold, double stop_probability,
                              uint8_t* inlier_mask, int32_t* n_trials, int32_t* n_inliers, int32_t* status)
{
    GSF_REQUIRE(ctx && offsets && inlier_mask && n_trials && n_inliers && status, "NULL argument");
    GSF_REQUIRE(P >= 0 && P <= 0x7fffffff, "bad P");
    GSF_REQUIRE(max_trials >= 1 && max_trials <= (1 << 20), "max_trials must be in [1, 2^20]");
    GSF_REQUIRE(min_samples >= 1 && min_samples <= RPW_MAX_SAMPLES, "min_samples must be in [1,64]");
    GSF_REQUIRE(degree >= 1 && degree <= RPW_MAX_DEGREE, "polynomial degree must be in [1,8]");
    GSF_REQUIRE(sample_idx, "sample_idx is NULL");
    if (P == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    if (max_trials > RP_MAX_TRIALS || min_samples > RP_MAX_SAMPLES || degree > RP_MAX_DEGREE) {
        // beyond the one-pass kernel: the wide kernel with its per-trial slab in the context's workspace
        const size_t slots = (size_t)P * (size_t)max_trials;
        int rc = ensure_scratch(ctx, slots * 12 + 64);
        if (rc) return rc;
        double* sc = (double*)ctx->scratch; int32_t* cn = (int32_t*)(sc + slots);
        hipLaunchKernelGGL(ransac_poly_wide_kernel, dim3((unsigned)P), dim3(RPW_THREADS), 0, ctx->stream, t, y, offsets, sample_idx, (int)max_trials,
                           (int)min_samples, (int)degree, residual_threshold, stop_probability, cn, sc, inlier_mask, n_trials, n_inliers, status);
        GSF_HIP(hipGetLastError());
        return GSF_OK;
    }
#define GSF_LAUNCH_RP(THREADS_) hipLaunchKernelGGL(ransac_poly_kernel<THREADS_>, dim3((unsigned)P), dim3(THREADS_), 0, ctx->stream, t, y, offsets, sample_idx, (int)max_trials, \
                       (int)min_samples, (int)degree, residual_threshold, stop_probability, inlier_mask, n_trials, n_inliers, status)
    if (max_trials <= 128) GSF_LAUNCH_RP(128); else if (max_trials <= 256) GSF_LAUNCH_RP(256); else if (max_trials <= 512) GSF_LAUNCH_RP(512); else GSF_LAUNCH_RP(1024);
#undef GSF_LAUNCH_RP
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

// LDS of the chain kernel (40 KB dynamic next to 23 KB of static arrays): the sample sets of one window-axis, then the swap partners of as
// many trials as the rest allows (at least one trial's worth)
static int chain_lds(int32_t max_trials, int32_t min_samples, int32_t max_window_rows, int64_t B, int& jseq_elems, size_t& lds)
{
    // per trial: score 8 + end 4 + nin 4 + idx 4 * min_samples bytes; then the swap partners of as many buffered trials as the budget allows
    // (at least one trial's worth).  Many logs: 16 buffered trials are enough for the batches scikit-learn's walk usually needs (4, 8, 16)
    // and keep a block under 20 KB, i.e. every SIMD of a CU busy with its own log; few logs: up to 64 trials / 40 KB as before.
    const size_t fixed = (size_t)max_trials * (16 + 4 * (size_t)min_samples);
    const size_t budget = 56 * 1024;
    if (fixed + 2 * (size_t)max_window_rows + 8 > budget) return 1;
    int tb = (int)((budget - fixed - 8) / 2 / (size_t)max_window_rows);
    const int cap = B > 256 ? 16 : 64;
    if (tb > cap) tb = cap;
    jseq_elems = tb * max_window_rows;
    lds = fixed + (size_t)((jseq_elems + 3) & ~3) * 2;
    return 0;
}

int gsf_gps_prefilter_chain_dev(gsf_ctx* ctx, const double* t, const double* pos, const int64_t* offsets, int64_t B, const int32_t* win_rows,
                                const int64_t* win_offsets, int32_t max_window_rows, int32_t max_trials, int32_t min_samples, int32_t degree,
                                double residual_threshold, double stop_probability, uint32_t* mt_state, uint8_t* keep, int32_t* win_status,
                                int32_t* log_status)
{
    GSF_REQUIRE(ctx && offsets && win_rows && win_offsets && mt_state && keep && win_status && log_status, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    GSF_REQUIRE(max_trials >= 1 && max_trials <= CH_MAX_TRIALS, "max_trials must be in [1,1024]");
    GSF_REQUIRE(min_samples >= 1 && min_samples <= RP_MAX_SAMPLES, "min_samples must be in [1,16]");
    GSF_REQUIRE(degree >= 1 && degree <= RP_MAX_DEGREE, "polynomial degree must be in [1,3]");
    GSF_REQUIRE(max_window_rows >= 1 && max_window_rows <= 14000, "max_window_rows must be in [1,14000]");
    if (B == 0) return GSF_OK;
    GSF_REQUIRE(t && pos, "NULL rows");
    GSF_HIP(hipSetDevice(ctx->device));
    int jseq_elems = 0; size_t lds = 0;
    GSF_REQUIRE(chain_lds(max_trials, min_samples, max_window_rows, B, jseq_elems, lds) == 0, "max_trials x min_samples / window length exceed the device sampler's LDS budget");
    hipLaunchKernelGGL(gps_prefilter_chain_kernel, dim3((unsigned)B), dim3(64), lds, ctx->stream, t, pos, offsets, (const int32_t*)nullptr, win_rows, win_offsets,
                       (int)max_trials, (int)min_samples, (int)degree, residual_threshold, stop_probability, jseq_elems, mt_state, keep, win_status, log_status,
                       WinGen{ 0, min_samples, 0.0, 0.0, 1, 0, ctx->prefilter_first_batch, ctx->prefilter_speculate, ctx->prefilter_miss_batch }, (int32_t*)nullptr);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // extern "C"

namespace gsf {
// the chain with the windows found ON THE DEVICE from the stamps (filter_gps_outliers_ransac as a whole, ref :136-247); counts (may be
// NULL): log b = rows offsets[b] .. offsets[b] + counts[b]
int launch_gps_prefilter_auto(gsf_ctx* ctx, const double* t, const double* pos, const int64_t* offsets, const int32_t* counts, int64_t B,
                              int32_t max_log_rows, const gsf_prefilter_config* f, uint32_t* mt_state, uint8_t* keep, int32_t* log_status, int32_t* log_info)
{
    GSF_REQUIRE(f->max_trials >= 1 && f->max_trials <= CH_MAX_TRIALS, "pre-filter max_trials must be in [1,1024]");
    GSF_REQUIRE(f->min_samples >= 1 && f->min_samples <= RP_MAX_SAMPLES, "pre-filter min_samples must be in [1,16]");
    GSF_REQUIRE(f->polynomial_degree >= 1 && f->polynomial_degree <= RP_MAX_DEGREE, "pre-filter polynomial degree must be in [1,3]");
    GSF_REQUIRE(max_log_rows >= 1 && max_log_rows <= 14000, "a log must hold 1 .. 14000 fixes for the device pre-filter");
    int jseq_elems = 0; size_t lds = 0;
    GSF_REQUIRE(chain_lds(f->max_trials, f->min_samples, max_log_rows, B, jseq_elems, lds) == 0, "max_trials x min_samples / log length exceed the device sampler's LDS budget");
    const WinGen gen{ f->use_sliding_window ? 1 : 2, f->min_samples, f->window_duration_seconds, f->window_duration_seconds * f->window_step_factor,
                      f->enabled ? 1 : 0, f->max_windows > 0 ? f->max_windows : 4096, ctx->prefilter_first_batch, ctx->prefilter_speculate, ctx->prefilter_miss_batch };
    hipLaunchKernelGGL(gps_prefilter_chain_kernel, dim3((unsigned)B), dim3(64), lds, ctx->stream, t, pos, offsets, counts, (const int32_t*)nullptr,
                       (const int64_t*)nullptr, (int)f->max_trials, (int)f->min_samples, (int)f->polynomial_degree, f->residual_threshold_meters,
                       f->stop_probability > 0.0 ? f->stop_probability : 0.99, jseq_elems, mt_state, keep, (int32_t*)nullptr, log_status, gen, log_info);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
}  // namespace gsf

extern "C" {

int gsf_gps_prefilter_auto_dev(gsf_ctx* ctx, const double* t, const double* pos, const int64_t* offsets, int64_t B, int32_t max_log_rows,
                               const gsf_prefilter_config* filter, uint32_t* mt_state, uint8_t* keep, int32_t* log_status, int32_t* log_info)
{
    GSF_REQUIRE(ctx && offsets && filter && mt_state && keep && log_status, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    if (B == 0) return GSF_OK;
    GSF_REQUIRE(t && pos, "NULL rows");
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_gps_prefilter_auto(ctx, t, pos, offsets, nullptr, B, max_log_rows, filter, mt_state, keep, log_status, log_info);
}

int gsf_gps_prefilter_chain(gsf_ctx* ctx, const double* t, const double* pos, const int64_t* offsets, int64_t B, const int32_t* win_rows,
                            const int64_t* win_offsets, int32_t max_window_rows, int32_t max_trials, int32_t min_samples, int32_t degree,
                            double residual_threshold, double stop_probability, uint32_t* mt_state, uint8_t* keep, int32_t* win_status, int32_t* log_status)
{
    GSF_REQUIRE(ctx && offsets && win_offsets && mt_state && keep && log_status && B >= 0, "bad arguments");
    if (B == 0) return GSF_OK;
    const int64_t total = offsets[B], nw = win_offsets[B];
    GSF_REQUIRE(total >= 0 && nw >= 0 && (total == 0 || (t && pos)) && (nw == 0 || (win_rows && win_status)), "bad offsets / NULL arrays");
    Staging st(ctx, (size_t)total * 33 + (size_t)(B + 1) * 16 + (size_t)nw * 12 + (size_t)B * (625 * 8 + 4), 12);
    if (st.rc()) return st.rc();
    const double* dt = st.in(t, (size_t)total); const double* dp = st.in(pos, (size_t)total * 3);
    const int64_t* doff = st.in(offsets, (size_t)B + 1); const int32_t* dwr = st.in(win_rows, (size_t)nw * 2);
    const int64_t* dwo = st.in(win_offsets, (size_t)B + 1);
    const uint32_t* dst_in = st.in(mt_state, (size_t)B * 625);
    uint32_t* dstate = st.out(mt_state, (size_t)B * 625);
    uint8_t* dkeep = st.out(keep, (size_t)total); int32_t* dws = st.out(win_status, (size_t)nw); int32_t* dls = st.out(log_status, (size_t)B);
    int rc = st.upload();
    if (rc) return rc;
    GSF_HIP(hipMemcpyAsync(dstate, dst_in, (size_t)B * 625 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    rc = gsf_gps_prefilter_chain_dev(ctx, dt, dp, doff, B, dwr, dwo, max_window_rows, max_trials, min_samples, degree, residual_threshold, stop_probability,
                                     dstate, dkeep, dws, dls);
    if (rc) return rc;
    return st.finish();
}

}  // extern "C"
