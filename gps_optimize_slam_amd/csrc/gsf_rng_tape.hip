// gsf_rng_tape.hip -- the reference's random draws (np.random.choice(n, k, replace=False), EKFGPSSLAM.py:405) for a FEW streams,
// spread over the whole chip.  gsf_rng.hip gives every stream one wave and walks the stream in order (~2.3 us per trial of
// permutation(271)); with one stream -- the reference's own case, one trajectory -- that wave is all the chip does.  Here the same
// draws, bit for bit and with the same final generator state, come from a handful of short launches:
//
//   the generator is a finite automaton on the stream of raw 32-bit outputs: state i (the current random_interval(i) of the
//   Fisher-Yates shuffle, i = n-1 .. 1, back to n-1 when a trial ends) and an output y is accepted iff (y & mask(i)) <= i, which
//   moves i to i-1.  Which outputs are accepted depends on the state they meet, but the OUTPUTS do not depend on anything:
//
//   1. mt_tape_kernel       one workgroup per stream writes the MT19937 words the draws can need ("the tape": expected consumption
//                           + 8 sigma) -- the only sequential part, a 227-word step of the recurrence per barrier;
//   2. mt_transition_kernel the tape is cut into segments of 512 outputs; a wave runs 64 START STATES (one per lane) through a
//                           segment, output by output (the output is wave-uniform), and stores where each start state ends and how
//                           many trials it completes on the way: the segment's transition table.  n-1 states x all segments =
//                           ~1.4 n^2 * trials lane-steps of 8 instructions, thousands of independent waves;
//   3. mt_compose_kernel /  the state and the trial number every segment starts in: tables of K consecutive segments are composed into
//      mt_expand_kernel     one (a thread per start state, tables staged in LDS) until a level fits one workgroup, which is walked in
//                           order; the starts are then handed down level by level (see the kernels);
//   4. mt_resolve_kernel    a wave per segment replays it from its now known start (64 outputs per step, acceptance pattern by the
//                           same fixed point as gsf_mt19937.hpp) and scatters the accepted values j_i to jseq[trial][i];
//   5. mt_tape_trace_kernel traces positions 0..k-1 of every trial backwards through its swaps (as gsf_mt19937.hpp does) and stores the
//                           generator state at the output where the last trial ended.
//
// Should the tape be too short (8 sigma: never observed) nothing has been written; mt_choice_kernel then runs as before, told by
// the header's `done` flag which streams still need it.  The work grows with n^2 * trials * B, so gsf::launch_mt_choice takes this route
// only for a few streams of moderate n (mt_tape_applies) -- the C1 drop-in and small robust batches; 1 000 streams already fill the chip
// the other way.
#include <math.h>

#include "gsf_mt19937.hpp"

using namespace gsf;

namespace {

constexpr int TAPE_SEG = 512;                   // raw outputs per segment
constexpr int TAPE_MAX_N = 2040;                // transition tables are (n-1) x segments; the trace stages 8..16 rows of n uint16 in LDS
constexpr int TAPE_MAX_STREAMS = 16;
constexpr size_t TAPE_MAX_BYTES = (size_t)768 << 20;
constexpr int WALK_LDS_WORDS = 15 * 1024;       // 60 KB of transition tables staged by a workgroup of the compose / expand kernels

struct TapeHdr {
    int32_t n;        // population of the stream; 0 = not drawn here (the reference returns before drawing, or out of this route's range)
    int32_t g0;       // tape index of the first output the draws consume (= the state's pos on entry; tape block 0 = the state's words)
    int32_t nblk;     // 624-word blocks on the tape
    int32_t nseg;     // segments [g0 + s*SEG, g0 + (s+1)*SEG) that hold at least one output of the tape
    int32_t done;     // set by the segment in which the last trial ended
    int32_t g_end;    // tape index after the last consumed output
    int32_t pad0, pad1;
};

// mean and variance of the outputs ONE trial of permutation(n) consumes: random_interval(i) is geometric with p = (i+1) / (mask(i)+1)
__host__ __device__ inline void trial_cost_term(int i, double& e, double& v)
{
    unsigned m = (unsigned)i; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
    const double r = ((double)m + 1.0) / ((double)i + 1.0);              // 1 / p
    e += r; v += r * (r - 1.0);                                          // (1 - p) / p^2
}
// short != 0 (gsf_set_option "tape_draws" = 2, tests only): a tape that ends before the draws do, so that the hand-over to
// mt_choice_kernel is exercised
__host__ __device__ inline double tape_outputs_needed(double e, double v, int trials, int cut_short)
{
    if (cut_short) return 0.9 * e * (double)trials;
    return e * (double)trials + 8.0 * sqrt(v * (double)trials) + 256.0;
}

// x_g = word g of the stream counted from the state's word 0.  mt19937_gen is the sliding recurrence x_g = x_{g-227} ^ f(x_{g-624}, x_{g-623})
// (the block-wise in-place form reads exactly these words); substituting it into itself once gives
// x_g = x_{g-454} ^ f(x_{g-851}, x_{g-850}) ^ f(x_{g-624}, x_{g-623}): 454 new words per barrier instead of 227.
constexpr int TAPE_THREADS = 512, TAPE_LAG = 2 * (MT_N - MT_M), TAPE_HIST = MT_N + (MT_N - MT_M), TAPE_PAD = 512;
// LDS-only barrier: the waves exchange ring words only; __syncthreads() would also wait for the tape stores of every step to land
__device__ __forceinline__ void tape_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Ring of three 454-word slots: step j (words 851 + 454 j + t, t < 454) writes slot j % 3 and reads slots (j-1) % 3 and (j-2) % 3 only
// (851 = 454 + 397 words back at most).  Unrolled three steps deep, every slot base is a compile-time constant, so a step computes NO
// address: the five reads use per-thread offsets set up once (which of the two older slots a look-back falls in depends on t alone).
__global__ __launch_bounds__(TAPE_THREADS) void mt_tape_kernel(const uint32_t* __restrict__ state, const int32_t* __restrict__ counts, int trials,
                                                               int kk, TapeHdr* __restrict__ hdr, uint32_t* __restrict__ tape, int64_t tape_stride,
                                                               int nblk_alloc, int nseg_alloc, int n_cap, int cut_short)
{
    __shared__ uint32_t ring[3 * TAPE_LAG];
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int n = counts[b];
    TapeHdr* h = hdr + b;
    if (n < 2 || n < kk || n > n_cap) {                                  // n_cap: the bound the workspace was sized for (<= TAPE_MAX_N)
        if (tid == 0) { TapeHdr z = { 0, 0, 0, 0, 0, 0, 0, 0 }; *h = z; }
        return;
    }
    const uint32_t* st = state + b * MT_STATE_WORDS;
    uint32_t* tp = tape + (size_t)b * (size_t)tape_stride;
    double e = 0.0, v = 0.0;
    for (int i = 1 + tid; i < n; i += TAPE_THREADS) trial_cost_term(i, e, v);
    for (int off = 32; off >= 1; off >>= 1) { e += __shfl_down(e, off); v += __shfl_down(v, off); }
    if ((tid & 63) == 0) { red[tid >> 6] = e; red[8 + (tid >> 6)] = v; }
    // ring position of word g < 851: "step -1" (words 397 .. 850) is slot 2, "step -2" (words -57 .. 396) slot 1
    auto hist = [](int g) { return g >= MT_M ? 2 * TAPE_LAG + g - MT_M : TAPE_LAG + g + (TAPE_LAG - MT_M); };
    for (int k = tid; k < MT_N; k += TAPE_THREADS) { const uint32_t w = st[k]; ring[hist(k)] = w; tp[k] = w; }
    __syncthreads();
    e = 0.0; v = 0.0;
    for (int k = 0; k < 8; ++k) { e += red[k]; v += red[8 + k]; }
    const int g0 = (int)st[MT_N];
    const double nb = ((double)g0 + tape_outputs_needed(e, v, trials, cut_short)) / (double)MT_N + 2.0;
    const int nblk = nb < (double)nblk_alloc ? (int)nb : nblk_alloc;
    int nseg = (nblk * MT_N - g0 + TAPE_SEG - 1) / TAPE_SEG;
    if (nseg > nseg_alloc) nseg = nseg_alloc;
    if (tid == 0) { TapeHdr z = { n, g0, nblk, nseg, 0, 0, 0, 0 }; *h = z; }
    const int total = nblk * MT_N;
    // words 624 .. 850 by the plain recurrence (the doubled one needs 851 words of history)
    if (tid < MT_N - MT_M) {
        const int g = MT_N + tid;
        const uint32_t x = mt_twist(ring[hist(g - MT_N)], ring[hist(g - MT_N + 1)], ring[hist(g - (MT_N - MT_M))]);
        ring[hist(g)] = x;
        tp[g] = x;
    }
    tape_barrier();
    // look-backs of thread t at a step whose slot is u:  g-454 -> (u+2)%3, t;   g-851 -> t < 397 ? ((u+1)%3, t+57) : ((u+2)%3, t-397);   ...
    const int t = tid < TAPE_LAG ? tid : 0;
    int a851[3], a850[3], a624[3], a623[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int s1 = ((u + 2) % 3) * TAPE_LAG, s2 = ((u + 1) % 3) * TAPE_LAG;
        a851[u] = t < 397 ? s2 + t + 57 : s1 + t - 397;
        a850[u] = t < 396 ? s2 + t + 58 : s1 + t - 396;
        a624[u] = t < 170 ? s2 + t + 284 : s1 + t - 170;
        a623[u] = t < 169 ? s2 + t + 285 : s1 + t - 169;
    }
    uint32_t* out = tp + TAPE_HIST;                                      // the tape has TAPE_PAD words of slack: the last step stores unguarded
    const bool act = tid < TAPE_LAG;
    auto step = [&](const int u) {
        if (act) {
            // mt_twist(cur, nxt, far) = far ^ f(cur, nxt): the inner call supplies x_{g-227} = x_{g-454} ^ f(x_{g-851}, x_{g-850})
            const uint32_t x = mt_twist(ring[a624[u]], ring[a623[u]], mt_twist(ring[a851[u]], ring[a850[u]], ring[((u + 2) % 3) * TAPE_LAG + t]));
            ring[u * TAPE_LAG + t] = x;
            out[t] = x;
        }
        out += TAPE_LAG;
        tape_barrier();
    };
    for (int G = TAPE_HIST; G < total; G += 3 * TAPE_LAG) {
        step(0);
        if (G + TAPE_LAG >= total) break;
        step(1);
        if (G + 2 * TAPE_LAG >= total) break;
        step(2);
    }
}

// one wave: 64 start states through one segment
__global__ __launch_bounds__(64) void mt_transition_kernel(const TapeHdr* __restrict__ hdr, const uint32_t* __restrict__ tape, int64_t tape_stride,
                                                           uint32_t* __restrict__ F, int nseg_alloc, int fstride)
{
    const int seg = blockIdx.x, grp = blockIdx.y, lane = threadIdx.x;
    const int64_t b = blockIdx.z;
    const TapeHdr h = hdr[b];
    if (h.n == 0 || seg >= h.nseg || grp * 64 >= h.n - 1) return;
    const uint32_t nm1 = (uint32_t)(h.n - 1);
    const uint32_t s0 = (uint32_t)(grp * 64 + lane + 1);
    uint32_t i = s0 <= nm1 ? s0 : nm1, w = 0;
    const uint32_t* tp = tape + (size_t)b * (size_t)tape_stride;
    const int G = h.nblk * MT_N;
    int g = h.g0 + seg * TAPE_SEG;
    const int ge = (g + TAPE_SEG < G) ? g + TAPE_SEG : G;
    auto step = [&](const uint32_t y) {
        const uint32_t u = y & (0xffffffffu >> __builtin_clz(i));
        i -= (u <= i) ? 1u : 0u;
        const bool wrap = (i == 0u);
        i = wrap ? nm1 : i;
        w += wrap ? 1u : 0u;
    };
    for (; g + 64 <= ge; g += 64) {
        const uint32_t yv = mt_temper(tp[g + lane]);
#pragma unroll
        for (int j = 0; j < 64; ++j) step((uint32_t)__builtin_amdgcn_readlane((int)yv, j));
    }
    if (g < ge) {
        const uint32_t yv = (g + lane < ge) ? mt_temper(tp[g + lane]) : 0u;
        const int m = ge - g;
        for (int j = 0; j < m; ++j) step((uint32_t)__builtin_amdgcn_readlane((int)yv, j));
    }
    if (s0 <= nm1) F[((size_t)b * (size_t)nseg_alloc + (size_t)seg) * (size_t)fstride + (s0 - 1)] = i | (w << 11);
}

// Transition tables pack (end state, trials completed) as state | trials << 11 (state <= 2039, trials <= 2^21 per table).
// Finding every segment's start from them is a walk in order; it is made short by composing tables first:
//   mt_compose_kernel  level l+1 table g = the K level-l tables g*K .. g*K+K-1 applied one after the other (K = what 60 KB of LDS hold),
//                      a workgroup per g, a thread per start state -- repeated until one workgroup's LDS holds a whole level;
//   mt_expand_kernel   a workgroup per level-(l+1) table: from ITS start (known from the level above; the top level starts in
//                      (n-1, trial 0)) one thread steps through the K level-l tables in LDS and stores where each of them starts.
__device__ __forceinline__ int tape_level_count(int nseg, int level, int K)
{
    int c = nseg;
    for (int l = 0; l < level; ++l) c = (c + K - 1) / K;
    return c;
}
__device__ __forceinline__ void tape_stage_tables(uint32_t* tab, const uint32_t* Tin, int first, int nb, int nst, int fstride, int tid)
{
    if (nst >= 48) {                                                     // a wave per table, rows read in order
        for (int sg = tid >> 6; sg < nb; sg += 4)
#pragma unroll 4
            for (int e = tid & 63; e < nst; e += 64) tab[sg * nst + e] = Tin[(size_t)(first + sg) * (size_t)fstride + e];
    } else {                                                             // short rows: lanes spread over (table, entry)
        for (int k = tid; k < nb * nst; k += 256) { const int sg = k / nst; tab[k] = Tin[(size_t)(first + sg) * (size_t)fstride + (k - sg * nst)]; }
    }
}
__global__ __launch_bounds__(256) void mt_compose_kernel(const TapeHdr* __restrict__ hdr, const uint32_t* __restrict__ Tin, int cap_in,
                                                         uint32_t* __restrict__ Tout, int cap_out, int K, int level_in, int fstride)
{
    extern __shared__ uint32_t tab[];
    const int tid = threadIdx.x, g = blockIdx.x;
    const int64_t b = blockIdx.y;
    const TapeHdr h = hdr[b];
    if (h.n == 0) return;
    const int cnt = tape_level_count(h.nseg, level_in, K), nst = h.n - 1;
    if (g * K >= cnt) return;
    const int nb = (cnt - g * K < K) ? (cnt - g * K) : K;
    tape_stage_tables(tab, Tin + (size_t)b * (size_t)cap_in * (size_t)fstride, g * K, nb, nst, fstride, tid);
    __syncthreads();
    uint32_t* out = Tout + ((size_t)b * (size_t)cap_out + (size_t)g) * (size_t)fstride;
    for (int s = tid; s < nst; s += 256) {
        uint32_t x = (uint32_t)s, w = 0;                                 // x = state - 1
        for (int r = 0; r < nb; ++r) { const uint32_t e = tab[r * nst + x]; x = (e & 0x7ffu) - 1u; w += e >> 11; }
        out[s] = (x + 1u) | (w << 11);
    }
}
// start_out[b][child] = (state, trial) the level-l table `child` begins in; state 0 = nothing left to draw from there on
__global__ __launch_bounds__(256) void mt_expand_kernel(const TapeHdr* __restrict__ hdr, const uint32_t* __restrict__ Tin, int cap_in,
                                                        const int2* __restrict__ start_up, int cap_up, int2* __restrict__ start_out, int K,
                                                        int level_in, int fstride, int trials)
{
    extern __shared__ uint32_t tab[];
    const int tid = threadIdx.x, g = blockIdx.x;
    const int64_t b = blockIdx.y;
    const TapeHdr h = hdr[b];
    if (h.n == 0) return;
    const int cnt = tape_level_count(h.nseg, level_in, K), nst = h.n - 1;
    if (g * K >= cnt) return;
    const int nb = (cnt - g * K < K) ? (cnt - g * K) : K;
    const int2 s0 = start_up ? start_up[(size_t)b * (size_t)cap_up + g] : make_int2(nst, 0);
    int2* so = start_out + (size_t)b * (size_t)cap_in + (size_t)g * K;
    if (s0.x == 0 || s0.y >= trials) {
        for (int r = tid; r < nb; r += 256) so[r] = make_int2(0, trials);
        return;
    }
    tape_stage_tables(tab, Tin + (size_t)b * (size_t)cap_in * (size_t)fstride, g * K, nb, nst, fstride, tid);
    __syncthreads();
    if (tid == 0) {
        int state = s0.x, trial = s0.y;
        for (int r = 0; r < nb; ++r) {
            const bool live = trial < trials;
            so[r] = make_int2(live ? state : 0, live ? trial : trials);
            if (live) { const uint32_t e = tab[r * nst + state - 1]; state = (int)(e & 0x7ffu); trial += (int)(e >> 11); }
        }
    }
}

// one wave per segment: the accepted values of its outputs, from the known start
__global__ __launch_bounds__(64) void mt_resolve_kernel(TapeHdr* __restrict__ hdr, const uint32_t* __restrict__ tape, int64_t tape_stride,
                                                        const int2* __restrict__ start, int nseg_alloc, int trials, uint16_t* __restrict__ jseq,
                                                        int jrow)
{
    const int seg = blockIdx.x, lane = threadIdx.x;
    const int64_t b = blockIdx.y;
    const TapeHdr h = hdr[b];
    if (h.n == 0 || seg >= h.nseg) return;
    const int2 st = start[(size_t)b * (size_t)nseg_alloc + seg];
    int i = st.x, t = st.y;                                              // wave-uniform
    if (i == 0 || t >= trials) return;
    const int n = h.n, G = h.nblk * MT_N;
    const uint32_t* tp = tape + (size_t)b * (size_t)tape_stride;
    int p = h.g0 + seg * TAPE_SEG;
    const int pe = (p + TAPE_SEG < G) ? p + TAPE_SEG : G;
    uint16_t* js = jseq + ((size_t)b * (size_t)trials + (size_t)t) * (size_t)jrow;
    while (p < pe) {
        // as mt_draw_choice (gsf_mt19937.hpp): fixed point of a_k = [i_k >= 1 and (y_k & mask(i_k)) <= i_k], i_k = i - (acceptances below lane k)
        const int avail = (pe - p < 64) ? (pe - p) : 64;
        const uint32_t y = (lane < avail) ? mt_temper(tp[p + lane]) : 0u;
        const int lvl = 32 - __clz(i);
        const uint32_t rate16 = ((uint32_t)(i + 1) << 16) >> lvl;
        int ik = i - (int)(__umul24((unsigned)lane, rate16) >> 16);
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
        int consumed = avail;
        const bool ends = __popcll(acc) >= i;                            // the acceptance made at i = 1 ends the trial: later outputs belong to the next one
        if (ends) {
            const unsigned long long last = __ballot(((acc >> lane) & 1ull) != 0ull && ik == 1);
            consumed = __ffsll((long long)last);
            acc &= (consumed >= 64) ? ~0ull : ((1ull << consumed) - 1ull);
        }
        if (((acc >> lane) & 1ull) != 0ull) js[ik] = (uint16_t)u;
        i -= __popcll(acc);
        p += consumed;
        if (ends) {
            t += 1; i = n - 1; js += jrow;
            if (t >= trials) {
                if (lane == 0) { hdr[b].g_end = p; hdr[b].done = 1; }
                return;
            }
        }
    }
}

// positions 0..kk-1 of `tb` trials per wave traced backwards through the swaps; block 0 of a stream also stores the generator state
constexpr int TRACE_ROWS = 16;
__global__ __launch_bounds__(64) void mt_tape_trace_kernel(const TapeHdr* __restrict__ hdr, const uint32_t* __restrict__ tape, int64_t tape_stride,
                                                           const uint16_t* __restrict__ jseq, int jrow, int trials, int kk, int rows,
                                                           int32_t* __restrict__ sample_idx, uint32_t* __restrict__ state)
{
    extern __shared__ uint32_t rows_l[];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.y;
    const TapeHdr h = hdr[b];
    if (h.n == 0 || !h.done) return;                                    // mt_choice_kernel draws (or blanks) this stream
    const int t0 = blockIdx.x * rows;
    const int nt = (trials - t0 < rows) ? (trials - t0) : rows;
    const int n = h.n;
    const uint32_t* src = (const uint32_t*)(jseq + ((size_t)b * (size_t)trials + (size_t)t0) * (size_t)jrow);   // jrow is even: rows are 4-byte aligned
    for (int k = lane; k < nt * (jrow / 2); k += 64) rows_l[k] = src[k];
    __syncthreads();
    const uint16_t* jl = (const uint16_t*)rows_l;
    int32_t* out = sample_idx + (size_t)b * (size_t)trials * (size_t)kk;
    for (int task = lane; task < nt * kk; task += 64) {
        const int tb = task / kk, pp = task - tb * kk;
        const uint16_t* js = jl + (size_t)tb * jrow;
        int at = pp, i = 1;
        // undoing swap (i, j_i), j_i <= i:  at -> j_i if at == i,  at -> i if at == j_i; from i = kk on "at == i" cannot hold (at <= max(p, i-1))
        for (; i < kk && i < n; ++i) {
            const int j = js[i];
            at = (at == i) ? j : ((at == j) ? i : at);
        }
        for (; i + 8 <= n; i += 8) {
            int j[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) j[q] = js[i + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) at = (at == j[q]) ? i + q : at;
        }
        for (; i < n; ++i) {
            const int j = js[i];
            at = (at == j) ? i : at;
        }
        out[(size_t)(t0 + tb) * kk + pp] = at;
    }
    if (blockIdx.x == 0) {
        const int blk = (h.g_end - 1) / MT_N;                            // g_end >= 1: n >= 2 consumes at least one output per trial
        const uint32_t* tp = tape + (size_t)b * (size_t)tape_stride + (size_t)blk * MT_N;
        uint32_t* st = state + b * MT_STATE_WORDS;
        for (int k = lane; k < MT_N; k += 64) st[k] = tp[k];
        if (lane == 0) st[MT_N] = (uint32_t)(h.g_end - blk * MT_N);      // 1 .. 624, as mt19937's pos after that output
    }
}

constexpr int TAPE_MAX_LEVELS = 12;
struct TapePlan {
    int nblk, nseg, fstride, jrow, rows, K, levels; int64_t tape_stride;                      // levels: table levels 0 .. levels-1; the last one fits one workgroup's LDS
    int cap[TAPE_MAX_LEVELS];                                            // tables per stream at each level (level 0 = segments)
    size_t o_hdr, o_tape, o_T[TAPE_MAX_LEVELS], o_start[TAPE_MAX_LEVELS], o_jseq, bytes;
};

bool tape_plan(int64_t B, int32_t trials, int32_t kk, int32_t n_max, int cut_short, TapePlan& pl)
{
    if (B < 1 || B > TAPE_MAX_STREAMS || n_max < 2 || n_max > TAPE_MAX_N || kk > n_max || trials < 1) return false;
    if ((int64_t)trials * n_max < 16384) return false;                   // a short job: the launches would cost more than the walk
    double e = 0.0, v = 0.0;
    for (int i = 1; i < n_max; ++i) trial_cost_term(i, e, v);
    const double nb = ((double)MT_N + tape_outputs_needed(e, v, trials, cut_short)) / (double)MT_N + 2.0;
    if (nb > 1.0e6) return false;
    pl.nblk = (int)nb;
    pl.nseg = (pl.nblk * MT_N + TAPE_SEG - 1) / TAPE_SEG;
    pl.fstride = n_max - 1;
    pl.jrow = (n_max + 1) & ~1;
    pl.rows = pl.jrow > 1024 ? 8 : TRACE_ROWS;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; };
    const size_t nb_ = (size_t)B;
    pl.o_hdr = take(nb_ * sizeof(TapeHdr));
    pl.tape_stride = (int64_t)pl.nblk * MT_N + TAPE_PAD;
    pl.o_tape = take(nb_ * (size_t)pl.tape_stride * 4);
    // fan-in of a composed table: what the LDS holds at most, but about sqrt(segments) when two levels then suffice (compose, the top walk
    // and the hand-down each stage and step through ~K tables per workgroup: equal shares are fastest)
    pl.K = WALK_LDS_WORDS / pl.fstride;                                  // >= 7
    {
        int r = (int)ceil(sqrt((double)pl.nseg));
        if (r < 8) r = 8;
        if (r < pl.K) pl.K = r;
    }
    pl.levels = 1; pl.cap[0] = pl.nseg;
    while (pl.cap[pl.levels - 1] > pl.K) {
        if (pl.levels >= TAPE_MAX_LEVELS) return false;
        pl.cap[pl.levels] = (pl.cap[pl.levels - 1] + pl.K - 1) / pl.K;
        ++pl.levels;
    }
    for (int l = 0; l < pl.levels; ++l) {
        pl.o_T[l] = take(nb_ * (size_t)pl.cap[l] * (size_t)pl.fstride * 4);
        pl.o_start[l] = take(nb_ * (size_t)pl.cap[l] * sizeof(int2));
    }
    pl.o_jseq = take(nb_ * (size_t)trials * (size_t)pl.jrow * 2);
    pl.bytes = off;
    return off <= TAPE_MAX_BYTES;
}

}  // namespace

namespace gsf {

bool mt_tape_applies(const gsf_ctx* ctx, int64_t B, int32_t trials, int32_t k, int32_t n_max)
{
    if (ctx->tape_draws == 0) return false;
    TapePlan pl;
    return tape_plan(B, trials, k, n_max, ctx->tape_draws == 2, pl);
}

// Enqueues launches 1-5; *done_flags / *done_stride (int32 units) tell mt_choice_kernel which streams it must still draw.
int launch_mt_tape(gsf_ctx* ctx, uint32_t* state, const int32_t* counts, int64_t B, int32_t trials, int32_t k, int32_t* sample_idx, int32_t n_max,
                   const int32_t** done_flags, int* done_stride)
{
    TapePlan pl;
    if (!tape_plan(B, trials, k, n_max, ctx->tape_draws == 2, pl)) { set_error("launch_mt_tape: out of range"); return GSF_ERR_INVALID_ARG; }
    int rc = ensure_rng_scratch(ctx, pl.bytes);
    if (rc) return rc;
    char* w = (char*)ctx->rng_scratch;
    TapeHdr* hdr = (TapeHdr*)(w + pl.o_hdr);
    uint32_t* tape = (uint32_t*)(w + pl.o_tape);
    uint16_t* jseq = (uint16_t*)(w + pl.o_jseq);
    auto T = [&](int l) { return (uint32_t*)(w + pl.o_T[l]); };
    auto S = [&](int l) { return (int2*)(w + pl.o_start[l]); };
    const int groups = (n_max - 1 + 63) / 64;
    const size_t lds = (size_t)pl.K * (size_t)pl.fstride * 4;             // <= WALK_LDS_WORDS words
    hipLaunchKernelGGL(mt_tape_kernel, dim3((unsigned)B), dim3(TAPE_THREADS), 0, ctx->stream, state, counts, (int)trials, (int)k, hdr, tape,
                       pl.tape_stride, pl.nblk, pl.nseg, (int)n_max, ctx->tape_draws == 2 ? 1 : 0);
    hipLaunchKernelGGL(mt_transition_kernel, dim3((unsigned)pl.nseg, (unsigned)groups, (unsigned)B), dim3(64), 0, ctx->stream, hdr, tape, pl.tape_stride, T(0),
                       pl.nseg, pl.fstride);
    for (int l = 0; l + 1 < pl.levels; ++l)
        hipLaunchKernelGGL(mt_compose_kernel, dim3((unsigned)pl.cap[l + 1], (unsigned)B), dim3(256), lds, ctx->stream, hdr, T(l), pl.cap[l], T(l + 1),
                           pl.cap[l + 1], pl.K, l, pl.fstride);
    for (int l = pl.levels - 1; l >= 0; --l) {
        const bool top = (l == pl.levels - 1);
        hipLaunchKernelGGL(mt_expand_kernel, dim3((unsigned)(top ? 1 : pl.cap[l + 1]), (unsigned)B), dim3(256), lds, ctx->stream, hdr, T(l), pl.cap[l],
                           top ? (const int2*)nullptr : S(l + 1), top ? 0 : pl.cap[l + 1], S(l), pl.K, l, pl.fstride, (int)trials);
    }
    int2* start = S(0);
    hipLaunchKernelGGL(mt_resolve_kernel, dim3((unsigned)pl.nseg, (unsigned)B), dim3(64), 0, ctx->stream, hdr, tape, pl.tape_stride, start, pl.nseg, (int)trials,
                       jseq, pl.jrow);
    hipLaunchKernelGGL(mt_tape_trace_kernel, dim3((unsigned)((trials + pl.rows - 1) / pl.rows), (unsigned)B), dim3(64), (size_t)pl.rows * pl.jrow * 2,
                       ctx->stream, hdr, tape, pl.tape_stride, jseq, pl.jrow, (int)trials, (int)k, pl.rows, sample_idx, state);
    GSF_HIP(hipGetLastError());
    *done_flags = &hdr->done;
    *done_stride = (int)(sizeof(TapeHdr) / 4);
    return GSF_OK;
}

}  // namespace gsf
