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
#include "gsf_internal.hpp"

using namespace gsf;

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_UPPER = 0x80000000u, MT_LOWER = 0x7fffffffu, MT_MATRIX_A = 0x9908b0dfu;
constexpr int MT_STATE_WORDS = 625;                       // key[624] + pos
constexpr int CHOICE_LDS_JSEQ_BYTES = 56 * 1024;          // trials buffered between draw and trace phases
constexpr int CHOICE_MAX_N = 28000;                       // one trial's jseq (uint16 per step) must fit the buffer

__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far)
{
    const uint32_t y = (cur & MT_UPPER) | (nxt & MT_LOWER);
    return far ^ (y >> 1) ^ ((nxt & 1u) ? MT_MATRIX_A : 0u);
}
// mt19937_gen (numpy/random/src/mt19937/mt19937.c): regenerate all 624 words.  mt[k] <- mt[k+397] ^ f(mt[k], mt[k+1]) for k < 227
// (old far words), mt[k] <- mt[k-227] ^ f(mt[k], mt[k+1]) for 227 <= k < 623 (NEW far words, distance 227 > 64 lanes), and the
// last word pairs with the NEW mt[0].  Within a 64-lane step every read happens before any write (lock-step wave).
__device__ __forceinline__ void mt_regenerate(uint32_t* mt, int lane)
{
    for (int k0 = 0; k0 < MT_N - MT_M; k0 += 64) {                       // 0 .. 226
        const int k = k0 + lane;
        uint32_t v = 0;
        const bool on = k < MT_N - MT_M;
        if (on) v = mt_twist(mt[k], mt[k + 1], mt[k + MT_M]);
        __syncthreads();
        if (on) mt[k] = v;
        __syncthreads();
    }
    for (int k0 = MT_N - MT_M; k0 < MT_N - 1; k0 += 64) {                // 227 .. 622
        const int k = k0 + lane;
        uint32_t v = 0;
        const bool on = k < MT_N - 1;
        if (on) v = mt_twist(mt[k], mt[k + 1], mt[k + (MT_M - MT_N)]);
        __syncthreads();
        if (on) mt[k] = v;
        __syncthreads();
    }
    if (lane == 0) mt[MT_N - 1] = mt_twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
    __syncthreads();
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
// smallest 2^m - 1 >= x (x >= 1)
__device__ __forceinline__ uint32_t mask_for(uint32_t x) { return 0xffffffffu >> __clz((int)x); }

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
__global__ __launch_bounds__(64) void mt_choice_kernel(uint32_t* __restrict__ state, const int32_t* __restrict__ counts, int trials, int kk,
                                                       int32_t* __restrict__ sample_idx)
{
    __shared__ uint32_t mt[MT_N + 1];
    __shared__ uint16_t jseq[CHOICE_LDS_JSEQ_BYTES / 2];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int n = counts[b];
    int32_t* out = sample_idx + (size_t)b * (size_t)trials * (size_t)kk;
    if (n < kk || n < 1 || n > CHOICE_MAX_N) {                           // the reference returns before drawing (ref :395-397): stream untouched
        for (int i = lane; i < trials * kk; i += 64) out[i] = 0;
        return;
    }
    uint32_t* st = state + b * MT_STATE_WORDS;
    for (int i = lane; i < MT_N; i += 64) mt[i] = st[i];
    int pos = (int)st[MT_N];
    __syncthreads();
    const int tbatch_max = CHOICE_LDS_JSEQ_BYTES / 2 / n;                // trials whose swap partners fit the LDS buffer
    const int tbatch = tbatch_max < 64 ? (tbatch_max < 1 ? 1 : tbatch_max) : 64;
    for (int t0 = 0; t0 < trials; t0 += tbatch) {
        const int nt = (trials - t0 < tbatch) ? (trials - t0) : tbatch;
        // ---- draw phase: swap partners j_i (i = n-1 .. 1) of nt trials
        for (int tb = 0; tb < nt; ++tb) {
            uint16_t* js = jseq + (size_t)tb * n;
            int i = n - 1;                                               // wave-uniform
            while (i >= 1) {
                if (pos >= MT_N) { mt_regenerate(mt, lane); pos = 0; }
                const int avail = (MT_N - pos < 64) ? (MT_N - pos) : 64;
                const bool have = lane < avail;
                const uint32_t y = have ? mt_temper(mt[pos + lane]) : 0u;
                const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                // fixed point of  a_k = [ i_k >= 1  and  (y_k & mask(i_k)) <= i_k ],  i_k = i - popcount(a & lanes below k)
                unsigned long long acc = __ballot(have);
                int ik = 0; uint32_t u = 0;
                for (int it = 0; it < 65; ++it) {
                    ik = i - __popcll(acc & below);
                    u = (ik >= 1) ? (y & mask_for((uint32_t)ik)) : 0u;
                    const unsigned long long nxt = __ballot(have && ik >= 1 && u <= (uint32_t)ik);
                    if (nxt == acc) break;
                    acc = nxt;
                }
                // the trial ends with the acceptance that takes i to 0: outputs after it belong to the next trial
                const int taken = __popcll(acc);
                int consumed = avail;
                if (taken >= i) {
                    const unsigned long long last = __ballot(((acc >> lane) & 1ull) != 0ull && ik == 1);   // the lane whose acceptance was made at i = 1
                    consumed = __ffsll((long long)last);                 // its index + 1
                    acc &= (consumed >= 64) ? ~0ull : ((1ull << consumed) - 1ull);
                }
                if (((acc >> lane) & 1ull) != 0ull) js[ik] = (uint16_t)u;
                i -= __popcll(acc);
                pos += consumed;
            }
        }
        __syncthreads();
        // ---- trace phase: x[p] after the shuffle = the start position reached by undoing the swaps i = 1 .. n-1 from position p
        for (int task = lane; task < nt * kk; task += 64) {
            const int tb = task / kk, p = task - tb * kk;
            const uint16_t* js = jseq + (size_t)tb * n;
            int at = p;
            for (int i = 1; i < n; ++i) {
                const int j = js[i];
                at = (at == i) ? j : ((at == j) ? i : at);
            }
            out[(size_t)(t0 + tb) * kk + p] = at;
        }
        __syncthreads();
    }
    for (int i = lane; i < MT_N; i += 64) st[i] = mt[i];
    if (lane == 0) st[MT_N] = (uint32_t)pos;
}

}  // namespace

namespace gsf {
int launch_mt_choice(gsf_ctx* ctx, uint32_t* state, const int32_t* counts, int64_t B, int32_t trials, int32_t k, int32_t* sample_idx)
{
    hipLaunchKernelGGL(mt_choice_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, state, counts, (int)trials, (int)k, sample_idx);
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
    return launch_mt_choice(ctx, state, n_population, B, trials, k, sample_idx);
}

}  // extern "C"
