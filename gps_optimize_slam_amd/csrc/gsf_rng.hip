// gsf_rng.hip -- the reference's random draws, generated on the device.
//
// compute_sim3_transform_robust draws every hypothesis with np.random.choice(n, min_samples, replace=False) (EKFGPSSLAM.py:405) on
// NumPy's GLOBAL LEGACY generator.  That call is RandomState.permutation(n)[:k]: a Fisher-Yates shuffle of arange(n) from the END
// (for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j])) where random_interval(max) draws 32-bit MT19937 outputs, masks them
// with the smallest 2^m - 1 >= max and rejects values > max (numpy/random/_legacy ... legacy-distributions.c, mt19937.c).  The
// number of raw outputs a trial consumes is therefore data dependent (~1.39 n), and trial t starts where trial t-1 stopped: the
// stream is inherently sequential.  scikit-learn's sample_without_replacement (the GPS pre-filter's RANSACRegressor, ref :157-165)
// takes the same permutation(n)[:k] route whenever 0.01 < k/n < 0.99.
//
// Mapping: ONE WAVE PER STREAM (= per trajectory).  The 624-word state lives in LDS; a regeneration is lane-parallel (three
// dependency-free chunks).  Consumption is 64 raw outputs at a time: lane k decides "accepted" from the number of acceptances in
// the lanes before it, A_k -- the value is masked and compared with i - A_k -- which is solved by fixed-point iteration on the
// ballot (each pass makes at least one more leading lane exact; the mask of a lane follows ITS i, so level changes need no cut).
// Accepted values are scattered to jseq[i] in LDS; once a batch of trials is drawn, lane (trial, p) traces position p backwards
// through the swaps (n steps, no array is shuffled) and writes sample_idx[trial][p].
#include "gsf_mt19937.hpp"

using namespace gsf;

namespace {

// np.random.seed(int) = mt19937_seed: Knuth's LCG over the 624 words, pos = 624 (a regeneration precedes the first output)
__global__ __launch_bounds__(64) void mt_seed_kernel(const uint32_t* __restrict__ seeds, int64_t B, uint32_t* __restrict__ state)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    uint32_t* st = state + b * MT_STATE_WORDS;
    uint32_t s = seeds[b];
    for (int i = 0; i < MT_N; ++i) {
        st[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1);
    }
    st[MT_N] = MT_N;
}

// sample_idx[b][trial][0..k) = permutation(n_b)[:k] for `trials` consecutive trials of stream b; state advanced exactly as NumPy's.
// idx_stride / trial0: stream b's sets live at sample_idx[b][idx_stride][k] and this launch fills trials trial0 .. trial0 + trials - 1 of them
// (the continuation of the robust chain's early-exit probe, gsf_robust.hip); done_flags: streams to leave alone.
__global__ __launch_bounds__(64) void mt_choice_kernel(uint32_t* __restrict__ state, const int32_t* __restrict__ counts, int trials, int kk,
                                                       int32_t* __restrict__ sample_idx, int jseq_bytes, const int32_t* __restrict__ done_flags,
                                                       int done_stride, int idx_stride, int trial0)
{
    __shared__ uint32_t mt[MT_N + 1];
    extern __shared__ uint16_t jseq[];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    if (done_flags && done_flags[b * done_stride]) return;               // drawn by the chip-wide route (gsf_rng_tape.hip) / decided by the probe
    const int n = counts[b];
    int32_t* out = sample_idx + ((size_t)b * (size_t)idx_stride + (size_t)trial0) * (size_t)kk;
    if (n < kk || n < 1 || n > CHOICE_MAX_N) {                           // the reference returns before drawing (ref :395-397): stream untouched
        for (int i = lane; i < trials * kk; i += 64) out[i] = 0;
        return;
    }
    uint32_t* st = state + b * MT_STATE_WORDS;
    for (int i = lane; i < MT_N; i += 64) mt[i] = st[i];
    int pos = (int)st[MT_N];
    __syncthreads();
    mt_draw_choice(mt, pos, n, trials, kk, jseq, jseq_bytes / 2, out, nullptr, lane);
    for (int i = lane; i < MT_N; i += 64) st[i] = mt[i];
    if (lane == 0) st[MT_N] = (uint32_t)pos;
}

}  // namespace

namespace gsf {
int launch_mt_choice(gsf_ctx* ctx, uint32_t* state, const int32_t* counts, int64_t B, int32_t trials, int32_t k, int32_t* sample_idx, int32_t n_max)
{
    // a few streams of moderate size: the chip-wide route draws them; this kernel then only serves the streams that route left (its
    // header says which: sets outside its range, or a tape that ran short)
    const int32_t* done_flags = nullptr; int done_stride = 0;
    if (n_max > 0 && mt_tape_applies(ctx, B, trials, k, n_max)) {
        const int rc = launch_mt_tape(ctx, state, counts, B, trials, k, sample_idx, n_max, &done_flags, &done_stride);
        if (rc) return rc;
    }
    const int bytes = choice_lds_bytes(n_max);
    hipLaunchKernelGGL(mt_choice_kernel, dim3((unsigned)B), dim3(64), (size_t)bytes, ctx->stream, state, counts, (int)trials, (int)k, sample_idx, bytes,
                       done_flags, done_stride, (int)trials, 0);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

// trials trial0 .. total_trials - 1 of the streams whose skip[b] is 0, one wave per stream (the robust chain after its early-exit probe)
int launch_mt_choice_rest(gsf_ctx* ctx, uint32_t* state, const int32_t* counts, int64_t B, int32_t total_trials, int32_t trial0, int32_t k,
                          int32_t* sample_idx, int32_t n_max, const int32_t* skip)
{
    if (trial0 >= total_trials) return GSF_OK;
    const int bytes = choice_lds_bytes(n_max);
    hipLaunchKernelGGL(mt_choice_kernel, dim3((unsigned)B), dim3(64), (size_t)bytes, ctx->stream, state, counts, (int)(total_trials - trial0), (int)k,
                       sample_idx, bytes, skip, 1, (int)total_trials, (int)trial0);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
}  // namespace gsf

extern "C" {

int gsf_mt19937_seed_batch_dev(gsf_ctx* ctx, const uint32_t* seeds, int64_t B, uint32_t* state)
{
    GSF_REQUIRE(ctx && B >= 0 && (B == 0 || (seeds && state)), "bad arguments");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(mt_seed_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, seeds, B, state);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_mt19937_choice_batch_dev(gsf_ctx* ctx, uint32_t* state, const int32_t* n_population, int64_t B, int32_t trials, int32_t k,
                                 int32_t* sample_idx)
{
    GSF_REQUIRE(ctx && B >= 0 && B <= 0x7fffffff && trials >= 0 && k >= 1 && k <= 64, "bad arguments");
    if (B == 0 || trials == 0) return GSF_OK;
    GSF_REQUIRE(state && n_population && sample_idx, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_mt_choice(ctx, state, n_population, B, trials, k, sample_idx, 0);
}

int gsf_mt19937_choice_bounded_batch_dev(gsf_ctx* ctx, uint32_t* state, const int32_t* n_population, int32_t n_max, int64_t B, int32_t trials,
                                         int32_t k, int32_t* sample_idx)
{
    GSF_REQUIRE(ctx && B >= 0 && B <= 0x7fffffff && trials >= 0 && k >= 1 && k <= 64 && n_max >= 0, "bad arguments");
    if (B == 0 || trials == 0) return GSF_OK;
    GSF_REQUIRE(state && n_population && sample_idx, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_mt_choice(ctx, state, n_population, B, trials, k, sample_idx, n_max);
}

}  // extern "C"
