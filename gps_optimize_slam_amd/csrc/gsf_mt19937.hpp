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
// LDS for the swap partners of up to 64 buffered trials of the LARGEST set (n_max rows; 0 = unknown): as little as the sets need, so that
// four streams share a CU when they can
inline int choice_lds_bytes(const int64_t n_max)
{
    if (n_max > 0 && n_max * 2 * 64 <= CHOICE_LDS_JSEQ_AIM) return (int)(n_max * 2 * 64);
    if (n_max > 0 && n_max * 2 * 8 <= CHOICE_LDS_JSEQ_AIM) return CHOICE_LDS_JSEQ_AIM;
    return CHOICE_LDS_JSEQ_MAX;
}

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
    // Three dependency phases instead of eleven 64-word steps: words 0..226 need old words only, 227..453 need the new 0..226, and
    // 454..622 the new 227..395.  Within a phase every read is issued before any write (one wave, lock-step), so the "old" neighbours
    // mt[k + 1] are still old when they are read and the LDS round trips of a phase overlap.
    auto phase = [&](const int lo, const int hi, const int far) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = lo + u * 64 + lane, kc = k < hi ? k : hi - 1;     // clamped: unconditional loads, all in flight together
            v[u] = mt_twist(mt[kc], mt[kc + 1], mt[kc + far]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = lo + u * 64 + lane;
            if (k < hi) mt[k] = v[u];
        }
        __syncthreads();
    };
    phase(0, MT_N - MT_M, MT_M);                                           // 0 .. 226      (227 words)
    phase(MT_N - MT_M, 2 * (MT_N - MT_M), MT_M - MT_N);                    // 227 .. 453    (227 words)
    phase(2 * (MT_N - MT_M), MT_N - 1, MT_M - MT_N);                       // 454 .. 622    (169 words)
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
__device__ __forceinline__ uint32_t mask_for(uint32_t x) { return 0xffffffffu >> (__builtin_clz(x) & 31); }   // x = 0: unspecified (callers discard it)

// Trace phase for kk <= 4: a lane follows H of a trial's first four positions backwards through the swaps, eight swap partners read
// ahead per step (one LDS round trip per eight swaps instead of one per swap).
template <int H>
__device__ __forceinline__ void mt_trace_heads(const uint16_t* jseq, const int n, const int nt, const int kk, const int t0, int32_t* out,
                                               const int lane)
{
    constexpr int PER = 4 / H;                                           // lanes per trial
    for (int task = lane; task < nt * PER; task += 64) {
        const int tb = task / PER, p0 = (task % PER) * H;
        const uint16_t* js = jseq + (size_t)tb * n;
        int a[H];
#pragma unroll
        for (int h = 0; h < H; ++h) a[h] = p0 + h;
        // undoing swap (i, j_i), j_i <= i:  a -> j_i if a == i,  a -> i if a == j_i.  After step i the position is <= max(p, i), so from
        // i = 4 on (p <= 3) "a == i" cannot hold and a step is one compare and one select.
        int i = 1;
        for (; i < 4 && i < n; ++i) {
            const int j = js[i];
#pragma unroll
            for (int h = 0; h < H; ++h) a[h] = (a[h] == i) ? j : ((a[h] == j) ? i : a[h]);
        }
        for (; i + 8 <= n; i += 8) {
            int j[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) j[u] = js[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int h = 0; h < H; ++h) a[h] = (a[h] == j[u]) ? i + u : a[h];
        }
        for (; i < n; ++i) {
            const int j = js[i];
#pragma unroll
            for (int h = 0; h < H; ++h) a[h] = (a[h] == j) ? i : a[h];
        }
#pragma unroll
        for (int h = 0; h < H; ++h)
            if (p0 + h < kk) out[(size_t)(t0 + tb) * kk + p0 + h] = a[h];
    }
}

// `trials` consecutive trials of permutation(n)[:kk] from the stream (mt, pos): out[trial * kk + p] (LDS or global), the stream
// advanced exactly as NumPy advances it.  end_raw (may be null): raw 32-bit outputs consumed since the call started, after each
// trial.  jseq / jseq_elems: LDS buffer for the swap partners of the trials processed together.  One wave; uses block barriers.
__device__ __forceinline__ void mt_draw_choice(uint32_t* mt, int& pos, const int n, const int trials, const int kk, uint16_t* jseq,
                                               const int jseq_elems, int32_t* out, int32_t* end_raw, const int lane)
{
    const int tbatch_max = jseq_elems / n;                               // trials whose swap partners fit the LDS buffer
    const int tbatch = tbatch_max < 64 ? (tbatch_max < 1 ? 1 : tbatch_max) : 64;
    int raw = 0;
    uint32_t pref = 0; int pref_pos = -1;                                // block words fetched ahead (valid for position pref_pos)
    for (int t0 = 0; t0 < trials; t0 += tbatch) {
        const int nt = (trials - t0 < tbatch) ? (trials - t0) : tbatch;
        // ---- draw phase: swap partners j_i (i = n-1 .. 1) of nt trials
        for (int tb = 0; tb < nt; ++tb) {
            uint16_t* js = jseq + (size_t)tb * n;
            int i = n - 1;                                               // wave-uniform
            while (i >= 1) {
                if (pos >= MT_N) { mt_regenerate(mt, lane); pos = 0; pref_pos = -1; }
                // fixed point of  a_k = [ i_k >= 1  and  (y_k & mask(i_k)) <= i_k ],  i_k = i - (acceptances in the lanes below k).
                // First guess: i falls at the chunk-start acceptance rate (i + 1) / (mask(i) + 1) per lane -- 1.7 refinement rounds per chunk
                // on average instead of 2.7 from "every lane accepts".  Two refinement passes per convergence test (a pass is ~7 VALU
                // instructions, the test a VALU -> SALU -> branch round trip): the second pass of a converged pattern reproduces it.
                // Lane k is final after pass k + 1, so the loops end within 65 passes.
                const int lvl = 32 - __clz(i);                                                   // mask(i) + 1 = 2^lvl   (wave-uniform)
                const uint32_t rate16 = ((uint32_t)(i + 1) << 16) >> lvl;                        // <= 65536
                int ik = i - (int)(__umul24((unsigned)lane, rate16) >> 16);
                if (((i - 65) | (MT_N - 64 - pos)) >= 0) {
                    // i > 64 and 64 outputs left in the block: every lane holds an output, no lane can reach i = 0, the trial cannot end
                    // here.  A pass is count, mask, compare; the next chunk's words are fetched while this one is resolved.
                    const uint32_t word = (pref_pos == pos) ? pref : mt[pos + lane];
                    const int nx = pos + 64 + lane;
                    pref = mt[nx < MT_N ? nx : MT_N - 1]; pref_pos = pos + 64;
                    const uint32_t y = mt_temper(word);
                    uint32_t u = y & mask_for((uint32_t)ik);
                    unsigned long long acc = __builtin_amdgcn_ballot_w64(u <= (uint32_t)ik);
                    for (;;) {
                        const int i1 = i - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(acc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)acc, 0u));
                        const unsigned long long mid = __builtin_amdgcn_ballot_w64((y & mask_for((uint32_t)i1)) <= (uint32_t)i1);
                        ik = i - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mid >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mid, 0u));
                        u = y & mask_for((uint32_t)ik);
                        acc = __builtin_amdgcn_ballot_w64(u <= (uint32_t)ik);
                        if (acc == mid) break;
                    }
                    js[((acc >> lane) & 1ull) != 0ull ? ik : 0] = (uint16_t)u;   // rejected lanes write the unused slot 0
                    i -= __popcll(acc);
                    pos += 64; raw += 64;
                    continue;
                }
                const int avail = (MT_N - pos < 64) ? (MT_N - pos) : 64;
                const bool have = lane < avail;
                const uint32_t y = have ? mt_temper(mt[pos + lane]) : 0u;
                ik = ik < 1 ? 1 : ik;
                uint32_t u = y & mask_for((uint32_t)ik);
                const unsigned long long have_mask = (avail >= 64) ? ~0ull : ((1ull << avail) - 1ull);
                unsigned long long acc = __builtin_amdgcn_ballot_w64(u <= (uint32_t)ik) & have_mask;
                for (;;) {
                    const int i1 = i - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(acc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)acc, 0u));
                    const unsigned long long mid = __builtin_amdgcn_ballot_w64((y & mask_for((uint32_t)i1)) <= (uint32_t)i1) &
                                                   __builtin_amdgcn_ballot_w64(i1 >= 1) & have_mask;
                    ik = i - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mid >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mid, 0u));
                    u = y & mask_for((uint32_t)ik);
                    acc = __builtin_amdgcn_ballot_w64(u <= (uint32_t)ik) & __builtin_amdgcn_ballot_w64(ik >= 1) & have_mask;
                    if (acc == mid) break;
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
        if (kk <= 4) {
            // lanes per trial chosen so that one round covers the batch: 4 heads per lane above 32 trials, 2 above 16, else 1
            if (nt > 32) mt_trace_heads<4>(jseq, n, nt, kk, t0, out, lane);
            else if (nt > 16) mt_trace_heads<2>(jseq, n, nt, kk, t0, out, lane);
            else mt_trace_heads<1>(jseq, n, nt, kk, t0, out, lane);
        } else {
            for (int task = lane; task < nt * kk; task += 64) {
                const int tb = task / kk, p = task - tb * kk;
                const uint16_t* js = jseq + (size_t)tb * n;
                int at = p, i = 1;
                for (; i < kk && i < n; ++i) {                            // p < kk: "at == i" is possible only while i < kk (see mt_trace_heads)
                    const int j = js[i];
                    at = (at == i) ? j : ((at == j) ? i : at);
                }
                for (; i + 8 <= n; i += 8) {
                    int j[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) j[u] = js[i + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) at = (at == j[u]) ? i + u : at;
                }
                for (; i < n; ++i) {
                    const int j = js[i];
                    at = (at == j) ? i : at;
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
