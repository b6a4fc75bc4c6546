// gsf_mt19937.hpp -- NumPy's legacy MT19937 stream and RandomState.permutation(n)[:k] for ONE WAVE (64-thread block), state in LDS.
// Shared by gsf_rng.hip (sample sets of the robust Sim3 fit, ref :405) and gsf_gpsfilter.hip (scikit-learn's sampler in the GPS
// pre-filter, ref :157-165).  See gsf_rng.hip for the algorithm notes.
#pragma once
#include "gsf_internal.hpp"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_UPPER = 0x80000000u, MT_LOWER = 0x7fffffffu, MT_MATRIX_A = 0x9908b0dfu;
constexpr int MT_STATE_WORDS = 625;                       // key[624] + pos
constexpr int CHOICE_LDS_JSEQ_MAX = 56 * 1024;            // swap partners of the trials buffered between draw and trace phases (dynamic LDS)
constexpr int CHOICE_LDS_JSEQ_AIM = 36 * 1024;            // ... sized to this when the sets allow: four blocks per CU instead of two
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

// `trials` consecutive trials of permutation(n)[:kk] from the stream (mt, pos): out[trial * kk + p] (LDS or global), the stream
// advanced exactly as NumPy advances it.  end_raw (may be null): raw 32-bit outputs consumed since the call started, after each
// trial.  jseq / jseq_elems: LDS buffer for the swap partners of the trials processed together.  One wave; uses block barriers.
__device__ __forceinline__ void mt_draw_choice(uint32_t* mt, int& pos, const int n, const int trials, const int kk, uint16_t* jseq,
                                               const int jseq_elems, int32_t* out, int32_t* end_raw, const int lane)
{
    const int tbatch_max = jseq_elems / n;                               // trials whose swap partners fit the LDS buffer
    const int tbatch = tbatch_max < 64 ? (tbatch_max < 1 ? 1 : tbatch_max) : 64;
    int raw = 0;
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
                // fixed point of  a_k = [ i_k >= 1  and  (y_k & mask(i_k)) <= i_k ],  i_k = i - (acceptances in the lanes below k)
                unsigned long long acc = __ballot(have);
                int ik = 0; uint32_t u = 0;
                for (int it = 0; it < 65; ++it) {
                    // two refinement passes per convergence test (a pass is ~10 VALU instructions, the test a VALU -> SALU -> branch
                    // round trip): the second pass of a converged pattern reproduces it
                    int i1 = i - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(acc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)acc, 0u));
                    uint32_t u1 = (i1 >= 1) ? (y & mask_for((uint32_t)i1)) : 0u;
                    const unsigned long long mid = __ballot(have && i1 >= 1 && u1 <= (uint32_t)i1);
                    ik = i - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mid >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mid, 0u));
                    u = (ik >= 1) ? (y & mask_for((uint32_t)ik)) : 0u;
                    const unsigned long long nxt = __ballot(have && ik >= 1 && u <= (uint32_t)ik);
                    if (nxt == mid) { acc = nxt; break; }
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
                pos += consumed; raw += consumed;
            }
            if (end_raw && lane == 0) end_raw[t0 + tb] = raw;
        }
        __syncthreads();
        // ---- trace phase: x[p] after the shuffle = the start position reached by undoing the swaps i = 1 .. n-1 from position p
        if (kk <= 4 && nt * 2 <= 64 + 63) {
            // up to four heads: lane (trial, half) traces two positions through one read of every swap partner
            for (int task = lane; task < nt * 2; task += 64) {
                const int tb = task >> 1, p0 = (task & 1) * 2;
                const uint16_t* js = jseq + (size_t)tb * n;
                int a0 = p0, a1 = p0 + 1;
                for (int i = 1; i < n; ++i) {
                    const int j = js[i];
                    a0 = (a0 == i) ? j : ((a0 == j) ? i : a0);
                    a1 = (a1 == i) ? j : ((a1 == j) ? i : a1);
                }
                if (p0 < kk) out[(size_t)(t0 + tb) * kk + p0] = a0;
                if (p0 + 1 < kk) out[(size_t)(t0 + tb) * kk + p0 + 1] = a1;
            }
        } else {
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
        }
        __syncthreads();
    }
}
// discard `k` raw outputs of the stream
__device__ __forceinline__ void mt_skip(uint32_t* mt, int& pos, int k, const int lane)
{
    while (k > 0) {
        if (pos >= MT_N) { mt_regenerate(mt, lane); pos = 0; }
        const int step = (k < MT_N - pos) ? k : (MT_N - pos);
        pos += step; k -= step;
    }
}

}  // namespace
