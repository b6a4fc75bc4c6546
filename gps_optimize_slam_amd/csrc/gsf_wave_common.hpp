// gsf_wave_common.hpp -- cross-lane primitives and chunk loads shared by the wave-level kernels
// (gsf_ekf_wave.hip: one wave per trajectory; gsf_ekf_block.hip: one wave per 64-pose chunk, a block per trajectory).
#pragma once
#include "gsf_internal.hpp"

namespace {

using namespace gsf;

typedef unsigned long long u64;

// ---- cross-lane movement.  The scans run on DPP (data-parallel-primitive) operand routing -- a VALU move, no LDS round
// trip: row_shr:1/2/4/8 inside the four 16-lane rows, then row_bcast:15 (rows 1,3 <- lane 15 of the row before) and
// row_bcast:31 (rows 2,3 <- lane 31).  Lanes without a source keep `old`, which is passed as the IDENTITY of the scanned
// monoid, so every stage is an unconditional "other o mine".  ds_bpermute (__shfl) is kept only for per-lane indices.
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp(double old, double v)
{
    // `old` is tied to the destination: materialise it as ONE 64-bit register pair (a single v_mov_b64) instead of two
    // 32-bit constant moves -- the empty asm pins the value in a VGPR pair before it is split into halves
    double oo = old;
    asm volatile("" : "+v"(oo));
    const long long o = __double_as_longlong(oo), x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)x, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// the same for a monoid whose identity is 0.0: on the full-row-mask stages (row_shr) bound_ctrl zero-fills the lanes without a
// source, so no identity has to be materialised at all; the two row_bcast stages still need `old` = 0 for the masked-off rows
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp0(double v)
{
    if (ROW_MASK != 0xf) return dpp<CTRL, ROW_MASK>(0.0, v);
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Quat dpp(const Quat& old, const Quat& v)
{
    return Quat{ dpp<CTRL, ROW_MASK>(old.x, v.x), dpp<CTRL, ROW_MASK>(old.y, v.y), dpp<CTRL, ROW_MASK>(old.z, v.z), dpp<CTRL, ROW_MASK>(old.w, v.w) };
}
// value of the previous lane; lane 0 receives `carry`
__device__ __forceinline__ double prev_lane(double carry, double v) { return dpp<DPP_WAVE_SHR1, 0xf>(carry, v); }
__device__ __forceinline__ Quat prev_lane(const Quat& c, const Quat& v) { return dpp<DPP_WAVE_SHR1, 0xf>(c, v); }
// wave-uniform source lane -> v_readlane (result lives in SGPRs)
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)x, src), hi = __builtin_amdgcn_readlane((int)(x >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ Quat lane_bcast(const Quat& q, int s) { return Quat{ lane_bcast(q.x, s), lane_bcast(q.y, s), lane_bcast(q.z, s), lane_bcast(q.w, s) }; }
__device__ __forceinline__ Vec3 lane_bcast(const Vec3& v, int s) { return Vec3{ lane_bcast(v.x, s), lane_bcast(v.y, s), lane_bcast(v.z, s) }; }
__device__ __forceinline__ double shidx(double v, int src) { return __shfl(v, src, 64); }          // per-lane source index
#define GSF_SCAN_STAGES(STAGE) STAGE(DPP_ROW_SHR1, 0xf) STAGE(DPP_ROW_SHR2, 0xf) STAGE(DPP_ROW_SHR4, 0xf) STAGE(DPP_ROW_SHR8, 0xf) \
                               STAGE(DPP_ROW_BCAST15, 0xa) STAGE(DPP_ROW_BCAST31, 0xc)
// sum over the wave: inclusive DPP scan, total read from lane 63 (wave-uniform result)
__device__ __forceinline__ double wave_sum(double v)
{
#define GSF_SUMSTAGE(CTRL, RM) { v += dpp0<CTRL, RM>(v); }
    GSF_SCAN_STAGES(GSF_SUMSTAGE)
#undef GSF_SUMSTAGE
    return lane_bcast(v, 63);
}
// ---- SIXTEEN wave sums at once (the moments of a point set): a "transposing" butterfly.  Each exchange stage halves the number
// of values a lane carries -- lane pairs swap complementary halves and add -- so the sixteen values cost 8+4+2+1 exchanges inside
// a 16-lane row (instead of 16 x 4 scan stages), two more adds across the four rows, and one v_readlane pair per total.
// ~150 instructions instead of ~420 for sixteen wave_sum() calls.  Inputs are scalars and the result is a by-value struct written
// with constant indices on purpose (see the note on pick helpers in gsf_wave_chunk.hpp: no select between array elements here).
struct Sums16 { double v[16]; };
// partner exchange inside a quad (xor 1 / xor 2): DPP quad_perm, every lane has a source
template <int QP> __device__ __forceinline__ double dpp_quad(double v)
{
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)x, QP, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), QP, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// partner at distance D = 4 or 8 inside a 16-lane row (lane ^ D): row_shl:D into the banks whose bit is clear, row_shr:D into the others
template <int D> __device__ __forceinline__ double dpp_row_xor(double v)
{
    constexpr int SHL = 0x100 + D, SHR = 0x110 + D;
    constexpr int LOW_BANKS = (D == 4) ? 0x5 : 0x3, HIGH_BANKS = (D == 4) ? 0xa : 0xc;
    const long long x = __double_as_longlong(v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)x, SHL, 0xf, LOW_BANKS, false);
    lo = __builtin_amdgcn_update_dpp(lo, (int)x, SHR, 0xf, HIGH_BANKS, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(x >> 32), SHL, 0xf, LOW_BANKS, false);
    hi = __builtin_amdgcn_update_dpp(hi, (int)(x >> 32), SHR, 0xf, HIGH_BANKS, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// first half: the exchanges inside a 16-lane row.  Afterwards lane l holds the sum over ITS ROW of the value whose index is the bit
// reversal of l & 15 (the block-per-trajectory kernel adds these per-row values of all its waves through LDS before the second half)
__device__ __forceinline__ double row_sums16_transposed(double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7,
                                                        double a8, double a9, double a10, double a11, double a12, double a13, double a14, double a15, int lane)
{
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0;
    // keep the half selected by my bit, hand the other half to the partner, add what the partner hands over
#define GSF_BFLY(bit, lo_, hi_, XCHG) ((bit ? hi_ : lo_) + XCHG(bit ? lo_ : hi_))
    // stage 1 (lane ^ 1): 16 -> 8 values;  b0 = 0 keeps indices 0..7, b0 = 1 keeps 8..15
    const double w0 = GSF_BFLY(b0, a0, a8, dpp_quad<0xB1>), w1 = GSF_BFLY(b0, a1, a9, dpp_quad<0xB1>), w2 = GSF_BFLY(b0, a2, a10, dpp_quad<0xB1>),
                 w3 = GSF_BFLY(b0, a3, a11, dpp_quad<0xB1>), w4 = GSF_BFLY(b0, a4, a12, dpp_quad<0xB1>), w5 = GSF_BFLY(b0, a5, a13, dpp_quad<0xB1>),
                 w6 = GSF_BFLY(b0, a6, a14, dpp_quad<0xB1>), w7 = GSF_BFLY(b0, a7, a15, dpp_quad<0xB1>);
    // stage 2 (lane ^ 2): 8 -> 4
    const double x0 = GSF_BFLY(b1, w0, w4, dpp_quad<0x4E>), x1 = GSF_BFLY(b1, w1, w5, dpp_quad<0x4E>), x2 = GSF_BFLY(b1, w2, w6, dpp_quad<0x4E>),
                 x3 = GSF_BFLY(b1, w3, w7, dpp_quad<0x4E>);
    // stage 3 (lane ^ 4): 4 -> 2;  stage 4 (lane ^ 8): 2 -> 1
    const double y0 = GSF_BFLY(b2, x0, x2, dpp_row_xor<4>), y1 = GSF_BFLY(b2, x1, x3, dpp_row_xor<4>);
    const double z = GSF_BFLY(b3, y0, y1, dpp_row_xor<8>);
#undef GSF_BFLY
    return z;
}
// second half: add the four rows, broadcast the sixteen totals
__device__ __forceinline__ Sums16 sums16_finish(double z)
{
    // lane l holds the sum over its 16-lane row of value index 8 b0 + 4 b1 + 2 b2 + b3; add the four rows
    z += __shfl_xor(z, 16, 64);
    z += __shfl_xor(z, 32, 64);
    Sums16 r;
    // index j lives in the lane whose low four bits are the bit reversal of j
#define GSF_OUT(j) r.v[j] = lane_bcast(z, ((j >> 3) & 1) | (((j >> 2) & 1) << 1) | (((j >> 1) & 1) << 2) | ((j & 1) << 3));
    GSF_OUT(0) GSF_OUT(1) GSF_OUT(2) GSF_OUT(3) GSF_OUT(4) GSF_OUT(5) GSF_OUT(6) GSF_OUT(7)
    GSF_OUT(8) GSF_OUT(9) GSF_OUT(10) GSF_OUT(11) GSF_OUT(12) GSF_OUT(13) GSF_OUT(14) GSF_OUT(15)
#undef GSF_OUT
    return r;
}
__device__ __forceinline__ Sums16 wave_sum16(double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7,
                                            double a8, double a9, double a10, double a11, double a12, double a13, double a14, double a15, int lane)
{
    return sums16_finish(row_sums16_transposed(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, lane));
}

// bits lo..hi (inclusive) of a 64-bit mask; empty if lo > hi
__device__ __forceinline__ u64 bits(int lo, int hi)
{
    if (lo > hi) return 0ull;
    const u64 upto_hi = (hi >= 63) ? ~0ull : ((1ull << (hi + 1)) - 1ull);
    const u64 below_lo = (lo <= 0) ? 0ull : ((1ull << lo) - 1ull);
    return upto_hi & ~below_lo;
}

// Lane masks STRAIGHT from a compare (v_cmp into an SGPR pair).  __ballot(flag) of a flag the compiler already holds as a lane mask is
// re-materialised (v_cndmask 0/1 + v_cmp_ne against exec): two vector instructions and a VALU -> SGPR hop per mask, and the moments
// passes below are made of masks.  Ordered compares: false for NaN operands, like the C operators.
__device__ __forceinline__ u64 mask_not_nan(const double x) { return __builtin_amdgcn_fcmp(x, x, 7); }                 // FCMP_ORD
__device__ __forceinline__ u64 mask_le(const double a, const double b) { return __builtin_amdgcn_fcmp(a, b, 5); }     // FCMP_OLE
__device__ __forceinline__ u64 mask_gt(const double a, const double b) { return __builtin_amdgcn_fcmp(a, b, 2); }     // FCMP_OGT
__device__ __forceinline__ u64 mask_nonzero(const uint32_t v) { return __builtin_amdgcn_uicmp(v, 0u, 33); }          // ICMP_NE
__device__ __forceinline__ u64 mask_first(const int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull)); }   // lanes 0..n-1

struct WaveArgs {
    const double* ts; const double* pos; const double* quat; const double* gps; const uint8_t* valid;
    const double* init_pos; const double* init_quat;
    double* R; double* t; double* s;              // pipeline outputs (may be null)
    double* pos_out; double* quat_out; int32_t* status;
    int64_t B, N;
    const int64_t* offsets;                       // ragged batches: trajectory b = rows offsets[b]..offsets[b+1] (else b*N.., N rows)
    FitRows rows;                                 // which rows feed the pipeline's fit (gsf_set_sim3_rows)
#ifdef GSF_EXP_HYBRID
    int32_t use_pv;                               // exp/gsf_ekf_wave_hybrid.hip only (`make hybrid`): run-time choice of the variance source
#endif
};

__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}
// first row and length of trajectory b
__device__ __forceinline__ void traj_span(const WaveArgs& a, int64_t b, int64_t& base, int64_t& n)
{
    if (a.offsets) {
        // wave-uniform by construction (b = blockIdx.x): pin both words in SGPRs so that every address below stays scalar
        const int64_t o0 = a.offsets[b], o1 = a.offsets[b + 1];
        base = uniform64(o0); n = uniform64(o1) - base;
    } else { base = b * a.N; n = a.N; }
}

#ifdef GSF_CHUNK_TIMING
// diagnostic build only (tools/chunk_timing.py): wave 0 writes (shader clock, 100 MHz clock) stamps into status[2k], status[2k+1];
// no wave writes its real status word in this build
#ifndef GSF_TIMING_B
#define GSF_TIMING_B 0
#endif
// The stamps are parked in LDS and written out when the wave is done: a global store behind every stamp stalls the next VALU
// write to its source registers until the memory pipeline has taken the store, which under the input burst took microseconds and
// showed up as a phantom gap after the prelude.
__device__ __forceinline__ int32_t* gsf_stamp_buf() { __shared__ int32_t buf[64]; return buf; }
#define GSF_STAMP(k) do { if (b == GSF_TIMING_B && lane == 0) { int32_t* sb_ = gsf_stamp_buf(); sb_[2 * (k)] = (int32_t)(clock64() & 0x7fffffff); sb_[2 * (k) + 1] = (int32_t)(wall_clock64() & 0x7fffffff); } } while (0)
#define GSF_STAMP_FLUSH() do { if (b == GSF_TIMING_B && lane == 0 && a.status) { const int32_t* sb_ = gsf_stamp_buf(); for (int k_ = 0; k_ < 32; ++k_) a.status[k_] = sb_[k_]; } } while (0)
#define GSF_STATUS_PTR(a) ((int32_t*)nullptr)
#elif defined(GSF_WAVE_START_TIMING)
// diagnostic build only (tools/wave_start_timing.py): EVERY wave writes the 100 MHz clock at its entry (stamp 0) and at its last
// chunk (stamp 8 + last chunk) into status[b] / status[B + b]; no real status words in this build
#ifndef GSF_WT_FIRST
#define GSF_WT_FIRST 0                                                   // stamp index recorded into status[b] ...
#define GSF_WT_SECOND 8                                                  // ... and (every index >= this one) into status[B + b]
#endif
#define GSF_STAMP(k) do { if (lane == 0 && a.status) { if ((k) == GSF_WT_FIRST) a.status[b] = (int32_t)(wall_clock64() & 0x7fffffff); else if (((k) == GSF_WT_SECOND || (GSF_WT_SECOND == 8 && (k) > 8)) && (k) < 13) a.status[a.B + b] = (int32_t)(wall_clock64() & 0x7fffffff); } } while (0)
#define GSF_STAMP_FLUSH() do { } while (0)
#define GSF_STATUS_PTR(a) ((int32_t*)nullptr)
#else
#define GSF_STAMP(k) do { } while (0)
#define GSF_STAMP_FLUSH() do { } while (0)
#define GSF_STATUS_PTR(a) ((a).status)
#endif

#ifndef GSF_WIDE_STORES
#define GSF_WIDE_STORES 1                                                 // big-batch builds store whole output slabs through LDS (see the chunk loop's store site)
#endif

inline EkfConfig to_core(const gsf_ekf_config* c)
{
    EkfConfig k;
    for (int i = 0; i < 7; ++i) { k.P0[i] = c->initial_cov_diag[i]; k.Qps[i] = c->process_noise_diag[i]; }
    for (int i = 0; i < 3; ++i) k.Rm[i] = c->meas_noise_diag[i];
    k.yaw_thr_rad = c->sharp_turn_yaw_rate_threshold_deg_per_sec * (M_PI / 180.0);
    k.sharp_turn_steps = c->default_ekf_transition_steps_on_sharp_turn;
    k._pad = 0;
    return k;
}

struct ChunkIn { double t; Vec3 p; Quat q; Vec3 z; uint32_t v; };

// pose i of one trajectory (clamped to the last pose for the idle lanes of the final chunk): one contiguous slab per array
__device__ __forceinline__ ChunkIn load_chunk(const double* __restrict__ tsb, const double* __restrict__ posb, const double* __restrict__ quatb,
                                              const double* __restrict__ gpsb, const uint8_t* __restrict__ valb, int64_t i, int64_t N)
{
    const int64_t il = i < N ? i : N - 1;
    ChunkIn c;
    // last use of every input row (the pipeline's fit pass has already run): streaming loads, so L2 keeps rows still to be fitted
#define GSF_NT(p) __builtin_nontemporal_load(&(p))
    c.t = GSF_NT(tsb[il]);
    c.p = Vec3{ GSF_NT(posb[il * 3]), GSF_NT(posb[il * 3 + 1]), GSF_NT(posb[il * 3 + 2]) };
    c.q = Quat{ GSF_NT(quatb[il * 4]), GSF_NT(quatb[il * 4 + 1]), GSF_NT(quatb[il * 4 + 2]), GSF_NT(quatb[il * 4 + 3]) };
    c.z = Vec3{ GSF_NT(gpsb[il * 3]), GSF_NT(gpsb[il * 3 + 1]), GSF_NT(gpsb[il * 3 + 2]) };
    c.v = GSF_NT(valb[il]);
#undef GSF_NT
    return c;
}


// Pins the point where a prefetched chunk must have ARRIVED.  vmcnt counts loads and stores in issue order, and hipcc merges
// control-flow paths conservatively: left alone it waits with vmcnt(0) at the loop top, i.e. for the previous chunk's STORES
// to be acknowledged (~1 us per chunk on a latency-bound small batch).  Consuming the registers right BEFORE this chunk's
// stores are issued makes that vmcnt(0) harmless (only the prefetch, issued a whole chunk ago, is outstanding), and the loop
// top then has nothing pending on either edge: store latency is never waited for.
__device__ __forceinline__ void chunk_arrived(const ChunkIn& c)
{
    asm volatile("" :: "v"(c.t), "v"(c.p.x), "v"(c.p.y), "v"(c.p.z), "v"(c.q.x), "v"(c.q.y), "v"(c.q.z), "v"(c.q.w), "v"(c.z.x), "v"(c.z.y), "v"(c.z.z), "v"(c.v) : "memory");
}

// ---- the same rows fetched as WHOLE SLABS (big-batch builds).  pos / quat / gps rows are 24 / 32 / 24 bytes apart, so a lane's 8-byte
// loads of its own row are strided; the chunk's slab of each array is contiguous (rows x 24 / 32 / 24 bytes), and the wave fetches it as
// 16-byte pieces, lane after lane (2 + 2 + 2 loads of a kilobyte instead of 3 + 4 + 3 strided ones), parks the pieces in LDS when the
// chunk is consumed and every lane picks its row up from there.  Stamps and mask bytes are contiguous per lane already.  Pieces past the
// end of a short last chunk are clamped onto its last 16 bytes (same bytes written to the same LDS place twice), idle lanes read the
// last row -- what the per-lane loads' clamping did.  Same values in the same registers afterwards: bit-identical results.
#ifndef GSF_WIDE_LOADS
#define GSF_WIDE_LOADS 1
#endif
typedef double gsf_v2 __attribute__((ext_vector_type(2), aligned(8)));
struct ChunkWide { double t; gsf_v2 P[2], Q[2], Z[2]; uint32_t v; };
__device__ __forceinline__ int wide_piece_off(const int piece, const int slab_bytes) { const int o = piece * 16; return o < slab_bytes - 16 ? o : slab_bytes - 16; }
__device__ __forceinline__ ChunkWide load_chunk_wide(const double* __restrict__ tsb, const double* __restrict__ posb, const double* __restrict__ quatb,
                                                     const double* __restrict__ gpsb, const uint8_t* __restrict__ valb, int64_t c0n, const int lane,
                                                     const int64_t N)
{
    if (c0n >= N) c0n = (N - 1) & ~(int64_t)63;                          // the prefetch issued by the last chunk: any rows of the track, never used
    const int rows = (int)(N - c0n < 64 ? N - c0n : 64);
    const int64_t il = c0n + lane < N ? c0n + lane : N - 1;
    ChunkWide w;
    w.t = __builtin_nontemporal_load(&tsb[il]);
    w.v = __builtin_nontemporal_load(&valb[il]);
    const char* pb = (const char*)(posb + c0n * 3); const char* qb = (const char*)(quatb + c0n * 4); const char* zb = (const char*)(gpsb + c0n * 3);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pc = lane + 64 * k;
        w.P[k] = __builtin_nontemporal_load((const gsf_v2*)(pb + wide_piece_off(pc, rows * 24)));
        w.Q[k] = __builtin_nontemporal_load((const gsf_v2*)(qb + wide_piece_off(pc, rows * 32)));
        w.Z[k] = __builtin_nontemporal_load((const gsf_v2*)(zb + wide_piece_off(pc, rows * 24)));
    }
    return w;
}
__device__ __forceinline__ void chunk_arrived(const ChunkWide& c)
{
    asm volatile("" :: "v"(c.t), "v"(c.P[0]), "v"(c.P[1]), "v"(c.Q[0]), "v"(c.Q[1]), "v"(c.Z[0]), "v"(c.Z[1]), "v"(c.v) : "memory");
}
// One staging area per wave for BOTH transpositions of the big-batch build -- the chunk's input pieces at the top of an iteration (64 x 10
// doubles), its output rows at the bottom (64 x 7): each use ends on its own wait, a wave's LDS operations execute in order, and with one
// 5 KB buffer instead of 5 + 3.5 KB a wave's LDS (with the 6 KB outage ring) drops from 14.5 to 11 KB -- twelve waves per CU (three per SIMD,
// what the registers allow) instead of the eleven that 160 KB / 14.5 KB admitted.
__device__ __forceinline__ double* wave_stage() { __shared__ double gsf_stage[64 * 10]; return gsf_stage; }
// the pieces of the chunk whose last active lane is L, parked in `stage` (64 x 10 doubles of LDS, this wave's own) and picked up row by row
__device__ __forceinline__ ChunkIn unpack_chunk(const ChunkWide& w, double* stage, const int lane, const int L)
{
    const int rows = L + 1;
    char* sp = (char*)stage; char* sq = sp + 64 * 24; char* sz = sq + 64 * 32;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pc = lane + 64 * k;
        *(gsf_v2*)(sp + wide_piece_off(pc, rows * 24)) = w.P[k];
        *(gsf_v2*)(sq + wide_piece_off(pc, rows * 32)) = w.Q[k];
        *(gsf_v2*)(sz + wide_piece_off(pc, rows * 24)) = w.Z[k];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // one wave per block: its LDS operations complete in order, no barrier
    const int r = lane < L ? lane : L;
    const double* dp = (const double*)sp + r * 3; const double* dq = (const double*)sq + r * 4; const double* dz = (const double*)sz + r * 3;
    ChunkIn c;
    c.t = w.t; c.v = w.v;
    c.p = Vec3{ dp[0], dp[1], dp[2] };
    c.q = Quat{ dq[0], dq[1], dq[2], dq[3] };
    c.z = Vec3{ dz[0], dz[1], dz[2] };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // ... and the reads are back before the next chunk's pieces overwrite the stage
    return c;
}
template <bool WIDE> struct NextChunk { typedef ChunkIn type; };
template <> struct NextChunk<true> { typedef ChunkWide type; };
#define GSF_WIDE(SMALLBATCH_) (GSF_WIDE_LOADS && !(SMALLBATCH_))

// Second half of the fused fit: wave totals of the per-lane partial moments (shifted by as_ / bs_), Umeyama closed form, R/t/s
// outputs, Sim3 of pose 0 (ref :439-451, :464-466).  Returns false -- after writing NaN rows and the status word -- when the fit
// is None or pose 0's quaternion is invalid (wave-uniform).  sums = { n, Sa[3], Sb[3], Saa, Sab[9] } per lane.
__device__ __forceinline__ bool fit_from_partials(const WaveArgs& a, const int64_t b, const int64_t base, const int64_t N, const int lane,
                                                  const double* sums, const double* as_, const double* bs_, const Quat& qraw0,
                                                  Vec3& p0, Quat& q0, int32_t& fit, const int32_t rows_flag = 0, const bool n_is_total = false)
{
    double* __restrict__ pob = a.pos_out + base * 3;
    double* __restrict__ qob = a.quat_out + base * 4;
    const double n = n_is_total ? sums[0] : wave_sum(sums[0]);            // (the moments passes of wave_prelude count their rows from the ballots)
    double Rb[9], tb[3], sb = NAN;
    fit = SIM3_NONE;
    if (n >= 3.0 && !(rows_flag & SIM3_FLAG_FEW_ROWS)) {                  // ref :430 (and :975 / :997: the reference raised before it got here)
        const Sums16 S = wave_sum16(sums[1], sums[2], sums[3], sums[4], sums[5], sums[6], sums[7], sums[8], sums[9], sums[10], sums[11],
                                    sums[12], sums[13], sums[14], sums[15], sums[16], lane);
        const double rn = fast_rcp(n);                                    // n >= 3
        const double ma[3] = { S.v[0] * rn, S.v[1] * rn, S.v[2] * rn };
        const double mb[3] = { S.v[3] * rn, S.v[4] * rn, S.v[5] * rn };
        const double ssq = fmax(0.0, S.v[6] - n * (ma[0] * ma[0] + ma[1] * ma[1] + ma[2] * ma[2]));
        double H[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) H[k] = S.v[7 + k] - n * ma[k / 3] * mb[k % 3];
        const double sc[3] = { as_[0] + ma[0], as_[1] + ma[1], as_[2] + ma[2] }, dc[3] = { bs_[0] + mb[0], bs_[1] + mb[1], bs_[2] + mb[2] };
        GSF_STAMP(3);
#ifdef GSF_FIT_SVD                                                        // A/B build (make ab): the Jacobi SVD route
        fit = umeyama_finalize<false>(H, ssq, sc, dc, n, Rb, tb, sb);
#else
        fit = umeyama_finalize<true>(H, ssq, sc, dc, n, Rb, tb, sb);     // every lane redundantly (wave-uniform inputs)
#endif
        GSF_STAMP(4);
    }
    Quat qn0; const bool q0ok = quat_unit(qraw0, qn0);
    if (fit == SIM3_NONE || !q0ok) {                                      // wave-uniform
        for (int64_t i = lane; i < N; i += 64) {
            pob[i * 3] = NAN; pob[i * 3 + 1] = NAN; pob[i * 3 + 2] = NAN;
            qob[i * 4] = NAN; qob[i * 4 + 1] = NAN; qob[i * 4 + 2] = NAN; qob[i * 4 + 3] = NAN;
        }
        if (lane == 0) {
            for (int k = 0; k < 9; ++k) a.R[b * 9 + k] = NAN;
            a.t[b * 3] = a.t[b * 3 + 1] = a.t[b * 3 + 2] = NAN; a.s[b] = NAN;
            if (GSF_STATUS_PTR(a)) a.status[b] = (fit == SIM3_NONE ? ((SIM3_NONE | (rows_flag & SIM3_FLAG_FEW_ROWS)) << 8) : 0) | (q0ok ? 0 : ST_BAD_QUAT);
        }
        return false;
    }
    fit |= rows_flag;                                                     // informational: which of the reference's row sets was fitted
    if (lane == 0) {
        for (int k = 0; k < 9; ++k) a.R[b * 9 + k] = Rb[k];
        a.t[b * 3] = tb[0]; a.t[b * 3 + 1] = tb[1]; a.t[b * 3 + 2] = tb[2]; a.s[b] = sb;
    }
    const double x = as_[0], y = as_[1], z = as_[2];                      // the source-side shift IS pose 0's position
    p0 = Vec3{ sb * (x * Rb[0] + y * Rb[1] + z * Rb[2]) + tb[0], sb * (x * Rb[3] + y * Rb[4] + z * Rb[5]) + tb[1],
               sb * (x * Rb[6] + y * Rb[7] + z * Rb[8]) + tb[2] };       // ref :464
    q0 = quat_mul(quat_from_matrix(Rb), qn0);                            // ref :465-466
    GSF_STAMP(5);
    return true;
}

// ---- the Sim3 row choice of main_process_gui (ref :973-998) inside a wave.  The valid rows V are walked 64 at a time; what travels
// from chunk to chunk (wave-uniform) is the last valid row seen so far -- its stamp, its index, and how many valid rows there are up
// to and including it.
struct RowScan { bool have_prev; double t_prev; int i_prev; int nvalid; };   // (row indices are 32-bit: 2^31 poses x 145 B would not fit the 288 GB of HBM)
// One chunk of the walk: m = ballot of the valid rows of the chunk whose first row is c0, t = this lane's stamp.  A gap is a valid
// row whose stamp exceeds the PREVIOUS VALID row's by more than max_gap (np.diff of the valid rows' stamps, :979-980).  On the first
// gap: row_end = index of that previous valid row -- the segment is V[:k] with k the index of the diff, so the row in front of the gap
// is left out as well (:981-982) -- nF = the number of valid rows before it, in_chunk = whether it sits in this chunk (else it is the
// carried row).  Returns true on a gap; otherwise advances the carry.
__device__ __forceinline__ bool rows_gap_in_chunk(RowScan& rs, const u64 m, const double t, const bool ok, const int lane, const int c0,
                                                  const double max_gap, int& row_end, int& nF, bool& in_chunk)
{
    if (m == 0ull) return false;
    u64 g;
    if ((m & (m + 1ull)) == 0ull) {
        // the usual chunk: its valid rows are lanes 0..k-1, so the previous valid row is the previous lane (lane 0: the carried row) --
        // one DPP move instead of a per-lane bit search and a ds_bpermute round trip on the lone wave's critical path
        const double tp = prev_lane(rs.t_prev, t);
        g = __ballot(ok && (lane > 0 || rs.have_prev) && (t - tp > max_gap));
    } else {
        // (kept behind a REAL branch: left to itself the compiler computes both forms in every chunk and selects -- a per-lane bit search and two
        // ds_bpermute round trips on the lone wave's critical path, 740 cycles per chunk by the in-kernel stamps, gpurun_out/r4k/timing.log)
        asm volatile("" ::: "memory");
        const u64 lower = m & bits(0, lane - 1);                          // valid rows of the chunk in front of this lane
        const int pl = lower != 0ull ? 63 - __clzll((long long)lower) : 0;
        const double tp_in = shidx(t, pl);
        const double tp = lower != 0ull ? tp_in : rs.t_prev;
        g = __ballot(ok && (lower != 0ull || rs.have_prev) && (t - tp > max_gap));
    }
    if (g != 0ull) {
        const int gl = __ffsll((long long)g) - 1;
        const u64 lg = m & bits(0, gl - 1);
        if (lg != 0ull) {
            const int ipl = 63 - __clzll((long long)lg);
            row_end = c0 + ipl; nF = rs.nvalid + __popcll(m & bits(0, ipl - 1)); in_chunk = true;
        } else { row_end = rs.i_prev; nF = rs.nvalid - 1; in_chunk = false; }
        return true;
    }
    const int hl = 63 - __clzll((long long)m);
    rs.t_prev = lane_bcast(t, hl); rs.i_prev = c0 + hl; rs.have_prev = true; rs.nvalid += __popcll(m);
    return false;
}

// The pipeline's moments pass under gsf_set_sim3_rows mode 1: the same one-pass shifted moments as in wave_prelude below, over the rows
// main_process_gui would hand to its fit (ref :973-998).  The first attempt looks for the gap WHILE it accumulates the rows that pass
// the duration limit -- the rounds stop at the gap, so an outage track reads less than before, not more -- and only what the
// reference's two fall-backs need costs a second pass: all valid rows when the first segment has fewer than min_samples rows
// (:984-986), the whole segment when the limit leaves fewer than min_samples (:993-995).  (One more case repeats the pass: the gap is
// found in a later ROUND than the row in front of it, which has been accumulated by then and does not belong to V[:k].)
// sums = { n, Sa[3], Sb[3], Saa, Sab[9] } per lane, bs_ = the GNSS-side shift; rows_flag = SIM3_FLAG_FEW_ROWS / _ROWS_ALL / _ROWS_SEGMENT.
#ifndef GSF_ROWS_ROUND
#define GSF_ROWS_ROUND 6                                                  // chunks per round of this pass (the big-batch build takes fewer: registers)
#endif
template <int MOM_ROUND>
__device__ __forceinline__ void fit_moments_reference_rows(const WaveArgs& a, const int64_t b, const int64_t base, const int64_t N, const int lane,
                                                           const double as0, const double as1, const double as2, double* sums, double* bs_,
                                                           int32_t& rows_flag)
{
    const double* __restrict__ tsb = a.ts + base;
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;
    const int ms = a.rows.min_samples;
    const double max_gap = a.rows.max_gap, max_dur = a.rows.max_dur;
    bool detect = true, use_tlim = true, gap_found = false;
    const int Ni = (int)N;
    int row_end = Ni;
    int nF = 0, state = 0;                                               // state 0: the timed subset, 1: the whole first segment, 2: all valid rows
    rows_flag = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        double bs0 = 0.0, bs1 = 0.0, bs2 = 0.0, tlim = 0.0;
        bool have_shift = false, over = false;
        bool prev_row_ok = false; int i_last = -1, nvalid = 0; double t_before = 0.0;   // the gap check's carry: row c0 - 1 valid / its stamp, the last valid row so far, the valid rows so far
        int nT = 0;
        double Sa0 = 0, Sa1 = 0, Sa2 = 0, Sb0 = 0, Sb1 = 0, Sb2 = 0, Saa = 0;
        double Sab[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int c0 = 0; c0 < Ni && c0 < row_end; c0 += 64 * MOM_ROUND) {
            double pa[MOM_ROUND][3], pz[MOM_ROUND][3], pt[MOM_ROUND]; uint32_t pv[MOM_ROUND];
#pragma unroll
            for (int k = 0; k < MOM_ROUND; ++k) {
                if (c0 + 64 * k < Ni) {                                   // wave-uniform
                    const int i = c0 + 64 * k + lane; const int64_t il = i < Ni ? i : Ni - 1;
                    pa[k][0] = posb[il * 3]; pa[k][1] = posb[il * 3 + 1]; pa[k][2] = posb[il * 3 + 2];
                    pz[k][0] = gpsb[il * 3]; pz[k][1] = gpsb[il * 3 + 1]; pz[k][2] = gpsb[il * 3 + 2];
                    pt[k] = tsb[il]; pv[k] = valb[il];
                } else {
                    pa[k][0] = pa[k][1] = pa[k][2] = 0.0; pz[k][0] = pz[k][1] = pz[k][2] = 0.0; pt[k] = 0.0; pv[k] = 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < MOM_ROUND; ++k) asm volatile("" : "+v"(pv[k]));   // (see wave_prelude: one memory round trip per round)
            // validity of the round's rows as lane masks (rows of the track, mask byte set, fix free of NaN)
            u64 mok[MOM_ROUND];
#pragma unroll
            for (int k = 0; k < MOM_ROUND; ++k)
                mok[k] = mask_first(Ni - (c0 + 64 * k)) & mask_nonzero(pv[k]) & mask_not_nan(pz[k][0]) & mask_not_nan(pz[k][1]) & mask_not_nan(pz[k][2]);
            GSF_STAMP(1);                                                 // the round's rows have arrived
            if (!have_shift && (mok[0] & 1ull) != 0ull) {                 // the usual track: its very first row of the round is valid
                bs0 = lane_bcast(pz[0][0], 0); bs1 = lane_bcast(pz[0][1], 0); bs2 = lane_bcast(pz[0][2], 0);
                tlim = lane_bcast(pt[0], 0) + max_dur;                    // segment_start_time + max_dur (:989-990)
                have_shift = true;
            }
            if (!have_shift) {                                            // the first valid row: GNSS-side shift and segment_start_time (:988)
                asm volatile("" ::: "memory");                           // (behind a real branch: ~150 selects otherwise folded into every track)
                u64 msel = 0ull; int ksel = -1;
#pragma unroll
                for (int k = MOM_ROUND - 1; k >= 0; --k) { if (mok[k] != 0ull) { msel = mok[k]; ksel = k; } }
                if (ksel >= 0) {
                    double v0 = pz[0][0], v1 = pz[0][1], v2 = pz[0][2], vt = pt[0];
#pragma unroll
                    for (int k = 1; k < MOM_ROUND; ++k) { const bool pick = (ksel == k); v0 = pick ? pz[k][0] : v0; v1 = pick ? pz[k][1] : v1; v2 = pick ? pz[k][2] : v2; vt = pick ? pt[k] : vt; }
                    const int f = __ffsll((long long)msel) - 1;
                    bs0 = lane_bcast(v0, f); bs1 = lane_bcast(v1, f); bs2 = lane_bcast(v2, f);
                    tlim = lane_bcast(vt, f) + max_dur;                   // segment_start_time + max_dur (:989-990)
                    have_shift = true;
                }
            }
            if (detect && !gap_found) {
                // Where can the first gap be?  Between two ADJACENT valid rows (g: the stamp of row i - 1 comes from the previous lane -- lane
                // 0: lane 63 of the chunk before, a constant-lane read -- and its validity is the chunk's mask shifted by one), or in front of
                // a valid row whose predecessor row is invalid, a RUN START (st).  Both sets are lane masks formed without a branch and without
                // a chain from chunk to chunk.  The usual track -- its valid rows are one run -- has no candidate at all except the very first
                // valid row: the round is accepted on two scalar compares.  Otherwise the candidates (one or two per outage) are visited in
                // row order on the scalar unit; the stamp in front of a run start is the end of the run before, fetched through the scalar cache.
                // (History, in-kernel stamps on a 271-pose track, gpurun_out/r4k .. r4q: walking every chunk with ~5 scalar branches each, the
                // bit-search + ds_bpermute form folded into every chunk by if-conversion: 1.56 us on a clean track; a branch-free form that
                // still carried the last valid stamp from chunk to chunk through v_readlane / s_cselect: 1.20 us -- a lone wave pays 20-40
                // cycles for every VALU -> SGPR -> VALU hop; masks straight from the compares + this form: DESIGN.md section 5.)
                u64 gk[MOM_ROUND], sk[MOM_ROUND]; u64 g_any = 0ull; int nst = 0; bool top = prev_row_ok; double tb = t_before;
#pragma unroll
                for (int k = 0; k < MOM_ROUND; ++k) {                     // (chunks past the end of the track: m = 0, no effect)
                    const u64 m = mok[k];
                    const u64 mp = (m << 1) | (top ? 1ull : 0ull);        // row i - 1 is valid
                    const double tp = prev_lane(tb, pt[k]);               // stamp of row i - 1
                    tb = lane_bcast(pt[k], 63);
                    gk[k] = mask_gt(pt[k] - tp, max_gap) & m & mp;        // rows i - 1 and i both valid, more than max_gap apart
                    sk[k] = m & ~mp;                                      // run starts
                    g_any |= gk[k]; nst += __popcll(sk[k]); top = (m >> 63) != 0ull;
                }
                // (the first valid row of the track is a run start with nothing in front of it: not a candidate.  One compare for the whole
                // round -- the per-lane maximum of the differences -- instead of one per chunk measured 0.1-0.2 us slower, gpurun_out/r4s.)
                if (g_any != 0ull || nst > (i_last < 0 ? 1 : 0)) {
                    asm volatile("" ::: "memory");                       // (a real branch, not if-conversion)
                    int e = i_last, cnt = nvalid;                      // last valid row so far, number of valid rows up to and including it
#pragma unroll
                    for (int k = 0; k < MOM_ROUND; ++k) {
                        const u64 m = mok[k];
                        if (!gap_found && m != 0ull) {                    // wave-uniform
                            u64 cand = gk[k] | sk[k];
                            while (cand != 0ull && !gap_found) {
                                const int l = __ffsll((long long)cand) - 1; cand &= cand - 1ull;
                                const u64 lower = m & mask_first(l);      // valid rows of this chunk in front of the candidate
                                const int ep = lower != 0ull ? c0 + 64 * k + (63 - __clzll((long long)lower)) : e;   // the valid row in front of it ...
                                const int before = lower != 0ull ? cnt + __popcll(lower) - 1 : cnt - 1;              // ... and how many valid rows precede THAT one
                                if (ep < 0) continue;                     // the first valid row of the track
                                bool gap = ((gk[k] >> l) & 1ull) != 0ull;
                                if (!gap) gap = tsb[c0 + 64 * k + l] - tsb[ep] > max_gap;   // across a hole: np.diff of the valid rows' stamps (:979-980)
                                if (gap) { gap_found = true; row_end = ep; nF = before; over = ep < c0; }            // V[:k] leaves row ep out as well (:981-982)
                            }
                            e = c0 + 64 * k + (63 - __clzll((long long)m)); cnt += __popcll(m);
                        }
                    }
                    if (!gap_found) { i_last = e; nvalid = cnt; }
                } else {
#pragma unroll
                    for (int k = 0; k < MOM_ROUND; ++k) {
                        if (mok[k] != 0ull) { i_last = c0 + 64 * k + (63 - __clzll((long long)mok[k])); nvalid += __popcll(mok[k]); }
                    }
                }
                prev_row_ok = top; t_before = tb;
            }
            GSF_STAMP(15);                                                // gap check of the round done
            // which rows of the round are summed: valid AND in front of row_end AND inside the duration limit -- as 64-bit masks on the
            // scalar unit (the row bound is a shift, only the limit needs a per-lane compare, and those are issued together up front), handed
            // to the lanes as ready-made select masks; the number of rows summed is a popcount, not a per-lane counter
            u64 mo[MOM_ROUND];
#pragma unroll
            for (int k = 0; k < MOM_ROUND; ++k) mo[k] = use_tlim ? (mok[k] & mask_le(pt[k], tlim)) : mok[k];
#pragma unroll
            for (int k = 0; k < MOM_ROUND; ++k) {
                mo[k] &= mask_first(row_end - (c0 + 64 * k));             // rows of this chunk in front of row_end
                nT += __popcll(mo[k]);
            }
#pragma unroll
            for (int k = 0; k < MOM_ROUND; ++k) {
                if (mo[k] != 0ull) {                                      // wave-uniform (a track with a gap sums its first segment only)
                    const bool o = __builtin_amdgcn_inverse_ballot_w64(mo[k]);
                    const double a0 = o ? pa[k][0] - as0 : 0.0, a1 = o ? pa[k][1] - as1 : 0.0, a2 = o ? pa[k][2] - as2 : 0.0;
                    const double b0 = o ? pz[k][0] - bs0 : 0.0, b1 = o ? pz[k][1] - bs1 : 0.0, b2 = o ? pz[k][2] - bs2 : 0.0;
                    Sa0 += a0; Sa1 += a1; Sa2 += a2; Sb0 += b0; Sb1 += b1; Sb2 += b2;
                    Saa += a0 * a0 + a1 * a1 + a2 * a2;
                    Sab[0] += a0 * b0; Sab[1] += a0 * b1; Sab[2] += a0 * b2;
                    Sab[3] += a1 * b0; Sab[4] += a1 * b1; Sab[5] += a1 * b2;
                    Sab[6] += a2 * b0; Sab[7] += a2 * b1; Sab[8] += a2 * b2;
                }
            }
        }
        sums[0] = (double)nT;                                             // (wave-uniform: the count of the rows summed, from the ballots)
        sums[1] = Sa0; sums[2] = Sa1; sums[3] = Sa2; sums[4] = Sb0; sums[5] = Sb1; sums[6] = Sb2; sums[7] = Saa;
#pragma unroll
        for (int k = 0; k < 9; ++k) sums[8 + k] = Sab[k];
        bs_[0] = bs0; bs_[1] = bs1; bs_[2] = bs2;
        if (detect) {
            detect = false;
            if (!gap_found) nF = nvalid;                                  // no gap: the first segment is all of V (:981)
            if (over) continue;                                           // once more, with the bound known from the start
        }
        // the sums now belong to exactly the rows (row_end, use_tlim) describe
        if (state == 0) {
            if (nF < ms) {                                                // :983
                if (!gap_found) { rows_flag = SIM3_FLAG_FEW_ROWS; break; }           // V itself is that short: ValueError (:975)
                row_end = Ni; use_tlim = false; state = 2; rows_flag = SIM3_FLAG_ROWS_ALL; continue;       // :984
            }
            if (nT < ms) { use_tlim = false; state = 1; rows_flag = SIM3_FLAG_ROWS_SEGMENT; continue; }   // :993-995
            break;                                                        // :996
        }
        if (state == 2 && nT < ms) rows_flag = SIM3_FLAG_FEW_ROWS;        // fewer than min_samples valid rows in all: ValueError (:975)
        break;
    }
}

// Initial pose of trajectory b: either the caller's Sim3-aligned pose 0, or (PIPELINE) the Umeyama fit on the rows with valid
// finite GNSS + Sim3 of pose 0.  Returns false (after writing NaN outputs / status) when the fit is None or pose 0's quaternion
// is invalid -- wave-uniformly.
template <bool PIPELINE>
__device__ __forceinline__ bool wave_prelude(const WaveArgs& a, const int64_t b, const int64_t base, const int64_t N, const int lane,
                                             Vec3& p0_out, Quat& q0_out, int32_t& fit_out)
{
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ quatb = a.quat + base * 4;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;
    // ------------------------------------------------------------------ initial pose
    Vec3 p0; Quat q0;
    int32_t fit = 0;
    if (PIPELINE) {
        // K2 on the rows with valid, finite GNSS, ONE pass: moments of the data shifted by pose 0 / the first finite fix
        // (|shifted| <= track length, so the raw-moment form H = Sab - n ma mb^T loses nothing at UTM magnitudes), then K3
        // of pose 0.  Sums are per-lane partials + a DPP wave reduction.
        // The rows are read in rounds of MOM_ROUND chunks with every load of a round issued before the first use, so a 271-pose
        // track (the small-batch, latency-bound case) pays ONE memory round trip for the whole pass instead of one per chunk.
        // The GNSS-side shift (first valid finite fix, wave-uniform) is found in the same pass: nothing is accumulated before it.
        const double as0 = posb[0], as1 = posb[1], as2 = posb[2];
        const Quat qraw0{ quatb[0], quatb[1], quatb[2], quatb[3] };      // pose 0's quaternion: requested here, used after the fit
        constexpr int MOM_ROUND = 6;
        // ONE closed form behind either moments pass (the two row rules differ in which rows they sum, not in what happens to the sums)
        double sums[17], bs_[3]; int32_t rows_flag = 0;
#ifdef GSF_NO_ROWS_RULE                                                    // A/B build (make norows): what the row-choice pass costs the OTHER path by being there
        if (false) {
#else
        if (a.rows.mode != 0) {                                           // wave-uniform: the reference's row choice (ref :973-998)
#endif
            fit_moments_reference_rows<GSF_ROWS_ROUND>(a, b, base, N, lane, as0, as1, as2, sums, bs_, rows_flag);
        } else {
            double bs0 = 0.0, bs1 = 0.0, bs2 = 0.0;
            bool have_shift = false;
            int nrows = 0;                                                // (counted from the ballots: wave-uniform)
            double Sa0 = 0, Sa1 = 0, Sa2 = 0, Sb0 = 0, Sb1 = 0, Sb2 = 0, Saa = 0;
            double Sab[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            for (int64_t c0 = 0; c0 < N; c0 += 64 * MOM_ROUND) {
                double pa[MOM_ROUND][3], pz[MOM_ROUND][3]; uint32_t pv[MOM_ROUND];
#pragma unroll
                for (int k = 0; k < MOM_ROUND; ++k) {
                    if (c0 + 64 * k < N) {                                    // wave-uniform
                        const int64_t i = c0 + 64 * k + lane, il = i < N ? i : N - 1;
                        pa[k][0] = posb[il * 3]; pa[k][1] = posb[il * 3 + 1]; pa[k][2] = posb[il * 3 + 2];
                        pz[k][0] = gpsb[il * 3]; pz[k][1] = gpsb[il * 3 + 1]; pz[k][2] = gpsb[il * 3 + 2];
                        pv[k] = valb[il];
                    } else {
                        pa[k][0] = pa[k][1] = pa[k][2] = 0.0; pz[k][0] = pz[k][1] = pz[k][2] = 0.0; pv[k] = 0u;
                    }
                }
                // the mask bytes stay opaque until every load of the round is issued: left alone, the compiler turns each byte into a lane
                // mask right behind its load (one VGPR less) and thereby waits for memory once per 64 rows instead of once per round
#pragma unroll
                for (int k = 0; k < MOM_ROUND; ++k) asm volatile("" : "+v"(pv[k]));
                u64 mok[MOM_ROUND];                                           // validity of the round's rows as lane masks
#pragma unroll
                for (int k = 0; k < MOM_ROUND; ++k)
                    mok[k] = mask_first((int)(N - (c0 + 64 * k) > 64 ? 64 : N - (c0 + 64 * k))) & mask_nonzero(pv[k]) & mask_not_nan(pz[k][0]) & mask_not_nan(pz[k][1]) &
                             mask_not_nan(pz[k][2]);
                GSF_STAMP(1);                                             // the round's rows have arrived
                if (!have_shift && (mok[0] & 1ull) != 0ull) {             // the usual track: its very first row is valid
                    bs0 = lane_bcast(pz[0][0], 0); bs1 = lane_bcast(pz[0][1], 0); bs2 = lane_bcast(pz[0][2], 0);
                    have_shift = true;
                }
                if (!have_shift) {                                            // wave-uniform, normally only in the first round
                    asm volatile("" ::: "memory");                           // (behind a real branch: ~150 selects otherwise folded into every track)
                    u64 msel = 0ull; int ksel = -1;
#pragma unroll
                    for (int k = MOM_ROUND - 1; k >= 0; --k) { if (mok[k] != 0ull) { msel = mok[k]; ksel = k; } }
                    if (ksel >= 0) {
                        double v0 = pz[0][0], v1 = pz[0][1], v2 = pz[0][2];
#pragma unroll
                        for (int k = 1; k < MOM_ROUND; ++k) { const bool pick = (ksel == k); v0 = pick ? pz[k][0] : v0; v1 = pick ? pz[k][1] : v1; v2 = pick ? pz[k][2] : v2; }
                        const int f = __ffsll((long long)msel) - 1;
                        bs0 = lane_bcast(v0, f); bs1 = lane_bcast(v1, f); bs2 = lane_bcast(v2, f);
                        have_shift = true;
                    }
                }
#pragma unroll
                for (int k = 0; k < MOM_ROUND; ++k) {
                    if (c0 + 64 * k < N) {                                    // wave-uniform
                        const bool o = __builtin_amdgcn_inverse_ballot_w64(mok[k]);
                        const double a0 = o ? pa[k][0] - as0 : 0.0, a1 = o ? pa[k][1] - as1 : 0.0, a2 = o ? pa[k][2] - as2 : 0.0;
                        const double b0 = o ? pz[k][0] - bs0 : 0.0, b1 = o ? pz[k][1] - bs1 : 0.0, b2 = o ? pz[k][2] - bs2 : 0.0;
                        nrows += __popcll(mok[k]); Sa0 += a0; Sa1 += a1; Sa2 += a2; Sb0 += b0; Sb1 += b1; Sb2 += b2;
                        Saa += a0 * a0 + a1 * a1 + a2 * a2;
                        Sab[0] += a0 * b0; Sab[1] += a0 * b1; Sab[2] += a0 * b2;
                        Sab[3] += a1 * b0; Sab[4] += a1 * b1; Sab[5] += a1 * b2;
                        Sab[6] += a2 * b0; Sab[7] += a2 * b1; Sab[8] += a2 * b2;
                    }
                }
            }
            sums[0] = (double)nrows; sums[1] = Sa0; sums[2] = Sa1; sums[3] = Sa2; sums[4] = Sb0; sums[5] = Sb1; sums[6] = Sb2; sums[7] = Saa;
#pragma unroll
            for (int k = 0; k < 9; ++k) sums[8 + k] = Sab[k];
            bs_[0] = bs0; bs_[1] = bs1; bs_[2] = bs2;
        }
        GSF_STAMP(2);
        const double as_[3] = { as0, as1, as2 };
        if (!fit_from_partials(a, b, base, N, lane, sums, as_, bs_, qraw0, p0, q0, fit, rows_flag, true)) return false;
    } else {
        p0 = Vec3{ a.init_pos[b * 3], a.init_pos[b * 3 + 1], a.init_pos[b * 3 + 2] };
        q0 = Quat{ a.init_quat[b * 4], a.init_quat[b * 4 + 1], a.init_quat[b * 4 + 2], a.init_quat[b * 4 + 3] };
    }

    p0_out = p0; q0_out = q0; fit_out = fit;
    return true;
}

// Variances of one 64-pose chunk (ref :712-713, :723-731): prefix composition of the Moebius maps P -> (A P + Bm)/(Cm P + Dm) per
// axis, carry-in cP.  Axes with identical (P0, Q, R) have identical recursions and reuse the scan (default CONFIG: x == y).
// One function for the chunked body AND the helper wave of the two-wave kernel, so that both produce the same bits.
struct AxisVar { double Pf, Pm, kg; };
// the scan on its own (identity carry): lane i holds the composition of the maps of poses first..i of the chunk; lanes that do not
// step hold the identity, so lane 63 always holds the chunk's total
struct Moebius { double A, B, C, D; };
__device__ __forceinline__ Moebius variance_scan(const double q, const double rr, const double dt, const bool stepping, const bool avail)
{
    const double b0 = q * dt;
    double A = 1.0, Bm = stepping ? b0 : 0.0, Cm = 0.0, Dm = 1.0;
    if (avail) { A = rr; Bm = rr * b0; Cm = 1.0; Dm = b0 + rr; }
    // mine (later) o other (earlier); lanes without a source see the identity map (1,0;0,1)
#define GSF_MSTAGE(CTRL, RM) {                                                                                              \
        const double oA = dpp<CTRL, RM>(1.0, A), oB = dpp0<CTRL, RM>(Bm), oC = dpp0<CTRL, RM>(Cm), oD = dpp<CTRL, RM>(1.0, Dm); \
        const double nA = A * oA + Bm * oC, nB = A * oB + Bm * oD, nC = Cm * oA + Dm * oC, nD = Cm * oB + Dm * oD;                  \
        A = nA; Bm = nB; Cm = nC; Dm = nD; }
    GSF_SCAN_STAGES(GSF_MSTAGE)
#undef GSF_MSTAGE
    return Moebius{ A, Bm, Cm, Dm };
}
__device__ __forceinline__ double moebius_apply(const Moebius& m, const double P) { return (m.A * P + m.B) * fast_rcp(m.C * P + m.D); }
// carry-in cPc applied: filtered / predicted variance and the Kalman gain of every pose of the chunk
__device__ __forceinline__ AxisVar variance_finish(const Moebius& m, const double q, const double rr, const double dt, const double cPc)
{
    AxisVar v;
    v.Pf = moebius_apply(m, cPc);                                        // P_f[i]
    v.Pm = prev_lane(cPc, v.Pf) + q * dt;                                // P_p[i]
    v.kg = v.Pm * fast_rcp(v.Pm + rr);                                   // Kalman gain if the fix is used
    return v;
}
__device__ __forceinline__ AxisVar variance_axis(const double q, const double rr, const double dt, const bool stepping, const bool avail, const double cPc)
{
    return variance_finish(variance_scan(q, rr, dt, stepping, avail), q, rr, dt, cPc);
}
// (scalar arguments and constant indices only: a helper that indexes its caller's arrays dynamically puts them into scratch)
__device__ __forceinline__ void variance_chunk(const EkfConfig& cfg, const int same1, const int same2, const double dt, const bool stepping,
                                               const bool avail, const double cP0, const double cP1, const double cP2,
                                               AxisVar& v0, AxisVar& v1, AxisVar& v2)
{
    v0 = variance_axis(cfg.Qps[0], cfg.Rm[0], dt, stepping, avail, cP0);
    if (same1 == 0) v1 = v0; else v1 = variance_axis(cfg.Qps[1], cfg.Rm[1], dt, stepping, avail, cP1);
    if (same2 == 0) v2 = v0; else if (same2 == 1) v2 = v1; else v2 = variance_axis(cfg.Qps[2], cfg.Rm[2], dt, stepping, avail, cP2);
}

// The helper wave of the two-wave kernel: variances of EVERY chunk of the track, written to LDS (pv[(axis*3 + {Pf,Pm,kg}) * stride
// + pose]) while the main wave is busy with the fit.  It reads only what the variance recursion depends on: stamps, mask, NaN-ness
// of the fixes.  Same flags, same dt, same variance_chunk() as the chunked body.
__device__ __forceinline__ void wave_variance_helper(const WaveArgs& a, const EkfConfig& cfg, const int64_t b, const int lane, double* pv, const int pv_stride)
{
    int64_t base, N; traj_span(a, b, base, N);
    if (N <= 0) return;
    const double* __restrict__ tsb = a.ts + base;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;
    int same_axis[3] = { -1, -1, -1 };
    if (cfg.P0[1] == cfg.P0[0] && cfg.Qps[1] == cfg.Qps[0] && cfg.Rm[1] == cfg.Rm[0]) same_axis[1] = 0;
    if (cfg.P0[2] == cfg.P0[0] && cfg.Qps[2] == cfg.Qps[0] && cfg.Rm[2] == cfg.Rm[0]) same_axis[2] = 0;
    else if (cfg.P0[2] == cfg.P0[1] && cfg.Qps[2] == cfg.Qps[1] && cfg.Rm[2] == cfg.Rm[1]) same_axis[2] = 1;
    double cP[3] = { cfg.P0[0], cfg.P0[1], cfg.P0[2] };
    double c_t = tsb[0];
    // the next chunk's rows are requested before the current chunk's scans (this wave must finish inside the main wave's fit)
    struct HIn { double t, z0, z1, z2; uint32_t v; };
    auto hload = [&](const int64_t i) __attribute__((always_inline)) {
        const int64_t il = i < N ? i : N - 1;
        return HIn{ tsb[il], gpsb[il * 3], gpsb[il * 3 + 1], gpsb[il * 3 + 2], valb[il] };
    };
    HIn nx = hload(lane);
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        const int64_t i = c0 + lane;
        const bool active = i < N, stepping = active && i != 0;
        const int L = (int)((N - c0 < 64) ? (N - c0 - 1) : 63);
        const HIn in = nx;
        if (c0 + 64 < N) nx = hload(c0 + 64 + lane);
        const double t = in.t, z0 = in.z0, z1 = in.z1, z2 = in.z2;
        const bool vraw = in.v != 0;
        const double dt = fmax(1e-6, t - prev_lane(c_t, t));             // ref :865
        const bool avail = stepping && vraw && !(isnan(z0) || isnan(z1) || isnan(z2));   // ref :867-869
        AxisVar v0, v1, v2;
        variance_chunk(cfg, same_axis[1], same_axis[2], dt, stepping, avail, cP[0], cP[1], cP[2], v0, v1, v2);
        const double Pf[3] = { v0.Pf, v1.Pf, v2.Pf }, Pm[3] = { v0.Pm, v1.Pm, v2.Pm }, kg[3] = { v0.kg, v1.kg, v2.kg };
        if (active) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { pv[(c * 3 + 0) * pv_stride + i] = Pf[c]; pv[(c * 3 + 1) * pv_stride + i] = Pm[c]; pv[(c * 3 + 2) * pv_stride + i] = kg[c]; }
        }
        cP[0] = lane_bcast(Pf[0], L); cP[1] = lane_bcast(Pf[1], L); cP[2] = lane_bcast(Pf[2], L);
        c_t = lane_bcast(t, L);
    }
}

// One wave walks trajectory `b` 64 poses at a time (serial over chunks, scans inside a chunk).
// PREVAR (two-wave kernel): the variances come from LDS (pv, written by wave_variance_helper) once the block barrier after the
// fit has been passed; everything else is identical, so the two kernels produce the same bits.
// SMALLBATCH: the cold blocks are inlined (no far calls; ~27 more registers, which only matter when three waves per SIMD do).
// RINGS / ring_slot: main waves per block (each owns one slice of the lane-private LDS ring).
template <int RINGS> struct RingStore {
    static __device__ __forceinline__ double (*get(int slot, double*))[6][64]
    {
        __shared__ double ring[RINGS][2][6][64];
        return ring[RINGS > 1 ? slot : 0];
    }
};
template <> struct RingStore<0> {
    static __device__ __forceinline__ double (*get(int, double* ext))[6][64] { return (double (*)[6][64])ext; }
};

// The chunk loop of the wave-per-trajectory filter, entered with the initial pose (p0, q0) and the first 64 poses already requested
// (nxt).  Split from the prelude so that a kernel with its own fit / initial-pose logic can fall back to it (gsf_ekf_lat.hip).
// AXMODE 1: the caller has checked on the host that axes x and y share (P0, Q, R) and z does not -- the default CONFIG -- so the
// choice of scans is compiled in and the variance scans, the orientation and the x/y and z position scans sit in straight-line
// code that the scheduler can interleave (a lone wave issues dependent FP64 / DPP work every 6-7 cycles, independent work every 4.5).
template <bool PIPELINE, bool PREVAR = false, bool SMALLBATCH = false, int RINGS = 1, int AXMODE = 0>
__device__ __forceinline__ void wave_serial_chunks(const WaveArgs& a, const EkfConfig& cfg, const int64_t b, const int lane, const int64_t base,
                                                   const int64_t N, const Vec3& p0, const Quat& q0, const int32_t fit,
                                                   typename NextChunk<GSF_WIDE(SMALLBATCH)>::type nxt,
                                                   const double* pv = nullptr, const int pv_stride = 0, const int ring_slot = 0,
                                                   double* ext_ring = nullptr)
{
    const double* __restrict__ tsb = a.ts + base;
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ quatb = a.quat + base * 4;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;
    double* __restrict__ pob = a.pos_out + base * 3;
    double* __restrict__ qob = a.quat_out + base * 4;
    // lane-private ring: rows + P_f of the last two open-outage chunks (RINGS = 0: the caller provides 768 doubles of LDS)
    double (*gsf_ring)[6][64] = RingStore<RINGS>::get(ring_slot, ext_ring);
    // ------------------------------------------------------------------ carry (wave-uniform, replicated in every lane)
    Quat cq = ekf_normalize(q0);                                         // ref :842, :683
    Vec3 cp = p0;
    double cP[3] = { cfg.P0[0], cfg.P0[1], cfg.P0[2] };
    int64_t c_ostart = 0;                                                // ref :861-862
    bool c_seg_sharp = false;
    double cPos[3] = { cP[0], cP[1], cP[2] };                            // P_f at the first pose of the open outage
    int same_axis[3] = { -1, -1, -1 };                                   // wave-uniform: axis c repeats axis same_axis[c]
    if (AXMODE == 1) { same_axis[1] = 0; }                               // known at compile time (see above)
    else {
        if (cfg.P0[1] == cfg.P0[0] && cfg.Qps[1] == cfg.Qps[0] && cfg.Rm[1] == cfg.Rm[0]) same_axis[1] = 0;
        if (cfg.P0[2] == cfg.P0[0] && cfg.Qps[2] == cfg.Qps[0] && cfg.Rm[2] == cfg.Rm[0]) same_axis[2] = 0;
        else if (cfg.P0[2] == cfg.P0[1] && cfg.Qps[2] == cfg.Qps[1] && cfg.Rm[2] == cfg.Rm[1]) same_axis[2] = 1;
    }

    chunk_arrived(nxt);
    GSF_STAMP(7);
    // "previous original pose" of pose 0 is pose 0 itself (ref :858): taken from lane 0 of the chunk that has just arrived
    bool c_prev_avail = __builtin_amdgcn_readlane((int)nxt.v, 0) != 0;   // ref :848 (raw mask)
    Vec3 c_po; Quat c_q0;
    if constexpr (GSF_WIDE(SMALLBATCH)) {
        // (the slab pieces hold pose 0 across lanes 0 and 1; these wave-uniform rows come through the scalar cache instead, other waves
        // cover the round trip in the big-batch build)
        c_po = Vec3{ posb[0], posb[1], posb[2] };
        c_q0 = Quat{ a.quat[base * 4], a.quat[base * 4 + 1], a.quat[base * 4 + 2], a.quat[base * 4 + 3] };
    } else {
        c_po = lane_bcast(nxt.p, 0); c_q0 = lane_bcast(nxt.q, 0);
    }
    Quat c_r; bool c_ok = quat_unit(c_q0, c_r);
    double c_t = lane_bcast(nxt.t, 0);
    int32_t status = c_prev_avail ? 0 : ST_HAD_OUTAGE;
    // Orientation on the fast path.  With every quaternion valid the increments telescope to q_i = Cq r_i, Cq = q_carry conj(r_carry),
    // and Cq is the SAME rotation for the whole track (q_carry = Cq r_L, so the next chunk's q_carry conj(r_carry) = Cq r_L conj(r_L)
    // = Cq): it is formed once and pinned in SGPRs; the carried state quaternion is only materialised (cq_fresh) for a chunk that has
    // to take the generic path.
    const Quat cq0 = cq;
    Quat Cq = lane_bcast(quat_mul(cq, quat_conj(c_r)), 0);
    bool cq_fresh = true;
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        const int64_t i = c0 + lane;
        const int L = (int)((N - c0 < 64) ? (N - c0 - 1) : 63);          // last active lane of this chunk
        // Lane predicates that depend on the lane index alone are formed as 64-bit masks on the SCALAR unit and handed to the lanes with
        // inverse_ballot (a register copy): no compare / select / shift of the vector unit is spent on the outage structure.
        const u64 act_mask = (L >= 63) ? ~0ull : ((2ull << L) - 1ull);   // lanes 0..L
        const u64 init_m = (c0 == 0) ? 1ull : 0ull;                      // pose 0 sits in lane 0 of the first chunk
        const u64 step_m = act_mask & ~init_m;
        const bool active = __builtin_amdgcn_inverse_ballot_w64(act_mask);
        const bool is_init = __builtin_amdgcn_inverse_ballot_w64(init_m);
        const bool stepping = __builtin_amdgcn_inverse_ballot_w64(step_m);
        // ---- this chunk's poses were loaded one iteration ago; issue the loads of the NEXT 64 poses now so that their
        // latency overlaps the scans below (the mask byte is compared at use time, never at load time)
        ChunkIn in;
        if constexpr (GSF_WIDE(SMALLBATCH)) {
            in = unpack_chunk(nxt, wave_stage(), lane, L);
            nxt = load_chunk_wide(tsb, posb, quatb, gpsb, valb, c0 + 64, lane, N);
        } else {
            in = nxt;
            nxt = load_chunk(tsb, posb, quatb, gpsb, valb, c0 + 64 + lane, N);   // unconditional (clamped to the last row past the end): no branch between the loads and the arithmetic below
        }
        const double t = in.t;
        const Vec3 p = in.p; const Quat q = in.q; const Vec3 z = in.z;
        const u64 vraw_m = __ballot(in.v != 0);
        // ---- calculate_relative_pose (ref :77-92) against the previous lane / the carry
        Quat r; const bool ok = quat_unit(q, r);
        const double t_pr = prev_lane(c_t, t);
        const Vec3 p_pr{ prev_lane(c_po.x, p.x), prev_lane(c_po.y, p.y), prev_lane(c_po.z, p.z) };
        // (the previous pose's unit quaternion is only needed on the cold paths -- generic orientation, sharp-turn pairs -- and the
        // cross-lane move cannot be sunk there by the compiler: it is fetched inside those wave-uniform branches)
        const u64 ok_mask = __ballot(ok);
        const u64 both_m = ((ok_mask << 1) | (c_ok ? 1ull : 0ull)) & ok_mask;   // pose i-1 and pose i both have a valid quaternion
        const bool both_ok = __builtin_amdgcn_inverse_ballot_w64(both_m);
        const double dt = fmax(1e-6, t - t_pr);                          // ref :865
        // Fast path (every quaternion of the chunk and the carried one valid -- the normal case): the increments telescope,
        //   dq_first * ... * dq_i = conj(r_carry) * r_i   and   R(q_{i-1}) R(r_{i-1})^-1 = R(Cq),  Cq = q_carry * conj(r_carry),
        // so the orientation needs no scan and the predicted displacement is ONE rotation by the wave-uniform Cq
        // (identical up to rounding: r conj(r) = 1 to 1 ulp for unit r).  Any invalid quaternion -> generic path below.
        const bool telescope = c_ok && ((ok_mask & act_mask) == act_mask);
        // (the generic path -- calculate_relative_pose per pose + a quaternion prefix product -- sits in ONE block further down, so
        // that the usual chunk runs from the loads to the scans without a branch)
        // ---- GNSS gate (ref :867-869) and the outage structure of the chunk, as masks
        const u64 zfin_m = __ballot(!(isnan(z.x) || isnan(z.y) || isnan(z.z)));
        const u64 avail_m = step_m & vraw_m & zfin_m;                    // the fix of pose i is used
        const bool avail = __builtin_amdgcn_inverse_ballot_w64(avail_m);
        const u64 av_m = (init_m & vraw_m) | avail_m;                    // "gnss available" flag of pose i (pose 0: raw mask, :848)
        const bool av = __builtin_amdgcn_inverse_ballot_w64(av_m);
        const u64 a_mask = act_mask & av_m;
        const u64 ap_m = (a_mask << 1) | ((c0 == 0 || c_prev_avail) ? 1ull : 0ull);   // the flag of pose i-1 (lane 0: the carry; pose 0: true)
        const u64 start_mask = act_mask & ~av_m & ap_m;                  // outage begins at this pose (ref :875-877; pose 0: :861)
        const u64 rec_mask = step_m & av_m & ~ap_m;                      // ref :879
        const u64 pair_mask = step_m & ~av_m & ~ap_m;                    // poses i-1 and i both inside the outage
        const bool recovers = __builtin_amdgcn_inverse_ballot_w64(rec_mask);
        const bool outpair = __builtin_amdgcn_inverse_ballot_w64(pair_mask);
        status |= (start_mask != 0ull) ? ST_HAD_OUTAGE : 0;
        // is_sharp_turn_in_segment (ref :808-826): pair (i-1, i) exceeds the yaw-rate threshold (or has a bad quaternion)
        u64 f_mask = 0ull;
        if (pair_mask != 0ull) {
            const Quat r_pr = prev_lane(c_r, r);
            bool f = false;
            if (outpair && t > t_pr) f = !both_ok || (SMALLBATCH ? yaw_rate_exceeds_body(r_pr, r, t - t_pr, cfg.yaw_thr_rad) : yaw_rate_exceeds_poly(r_pr, r, t - t_pr, cfg.yaw_thr_rad));
            f_mask = __ballot(f);
        }
        // recovery decision per recovering lane (ref :879-894)
        bool sharp = false;
        if (rec_mask != 0ull && recovers) {                              // (wave-uniform test first: the usual chunk skips this with a scalar branch)
            const u64 sm = start_mask & bits(0, lane - 1);
            int64_t s_glob; bool seg;
            if (sm != 0ull) {
                const int s = 63 - __clzll((long long)sm);
                s_glob = c0 + s;
                seg = (f_mask & bits(s + 1, lane - 1)) != 0ull;
            } else {
                s_glob = c_ostart;
                seg = c_seg_sharp || (f_mask & bits(0, lane - 1)) != 0ull;
            }
            sharp = (i - s_glob >= 2) && seg;
        }
        const u64 sharp_mask = __ballot(sharp);
        const u64 rts_mask = rec_mask & ~sharp_mask;                     // recoveries that run the RTS back-pass
        status |= ((sharp_mask != 0ull) ? ST_SHARP_TURN : 0) | ((rts_mask != 0ull) ? ST_RTS_APPLIED : 0);
        // one-step blend weight on a sharp-turn recovery (ref :752-768, Q7): 1/eff if eff > 1, else a hard update
        const double wgt_sharp = (cfg.sharp_turn_steps > 1) ? 1.0 / (double)cfg.sharp_turn_steps : 1.0;    // wave-uniform
        const double wgt = sharp ? wgt_sharp : 1.0;

        // ---- orientation (ref :708-709) and predicted displacement (ref :707), telescoped form -- computed unconditionally, in the same
        // straight-line code as the variance scans below (independent work for the scheduler); a chunk that has to take the generic
        // path overwrites both afterwards.
        // normalize_quaternion (ref :697-700) is the identity here up to rounding: Cq and r are unit quaternions (every quaternion of
        // the chunk passed quat_unit), so |Cq r| = 1 +- 2e-16 and its "norm > 1e-9" guard cannot fire.  The product is written out as
        // it is; pose 0 keeps the initial state (ref :842).
        Quat qi = quat_mul(Cq, r);
        qi.x = is_init ? cq0.x : qi.x; qi.y = is_init ? cq0.y : qi.y; qi.z = is_init ? cq0.z : qi.z; qi.w = is_init ? cq0.w : qi.w;
        Vec3 u = quat_rotate(Cq, Vec3{ p.x - p_pr.x, p.y - p_pr.y, p.z - p_pr.z });
        u.x = stepping ? u.x : 0.0; u.y = stepping ? u.y : 0.0; u.z = stepping ? u.z : 0.0;

        // ---- variances (ref :712-713, :723-731): scanned here, or -- PREVAR -- already computed by the helper wave (LDS)
        double Pf[3], Pm[3], kg[3];
#ifdef GSF_EXP_HYBRID
        if (PREVAR && (a).use_pv != 0) {
#else
        if (PREVAR) {
#endif
            const int64_t il = active ? i : N - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) { Pf[c] = pv[(c * 3 + 0) * pv_stride + il]; Pm[c] = pv[(c * 3 + 1) * pv_stride + il]; kg[c] = pv[(c * 3 + 2) * pv_stride + il]; }
        } else {
            AxisVar v0, v1, v2;
            variance_chunk(cfg, same_axis[1], same_axis[2], dt, stepping, avail, cP[0], cP[1], cP[2], v0, v1, v2);
            Pf[0] = v0.Pf; Pf[1] = v1.Pf; Pf[2] = v2.Pf; Pm[0] = v0.Pm; Pm[1] = v1.Pm; Pm[2] = v2.Pm; kg[0] = v0.kg; kg[1] = v1.kg; kg[2] = v2.kg;
        }

        if (telescope) cq_fresh = false;
        else {
            // calculate_relative_pose, ref :77-92
            if (!cq_fresh) { cq = quat_mul(Cq, c_r); cq_fresh = true; }  // the state quaternion the telescoped chunks did not carry
            const Quat r1i = quat_conj(prev_lane(c_r, r));
            Vec3 dpl = quat_rotate(r1i, Vec3{ p.x - p_pr.x, p.y - p_pr.y, p.z - p_pr.z });
            Quat dq = quat_mul(r1i, r);
            const bool move = stepping && both_ok;
            dpl.x = move ? dpl.x : 0.0; dpl.y = move ? dpl.y : 0.0; dpl.z = move ? dpl.z : 0.0;
            dq.x = move ? dq.x : 0.0; dq.y = move ? dq.y : 0.0; dq.z = move ? dq.z : 0.0; dq.w = move ? dq.w : 1.0;
            if (__ballot(stepping && !both_ok) != 0ull) status |= ST_BAD_QUAT;
            // inclusive prefix product of the increments, q_i = normalize(q_carry * dq_first * ... * dq_i)
            Quat D = dq;
            const Quat QID{ 0.0, 0.0, 0.0, 1.0 };
#define GSF_QSTAGE(CTRL, RM) { const Quat o = dpp<CTRL, RM>(QID, D); D = quat_mul(o, D); }
            GSF_SCAN_STAGES(GSF_QSTAGE)
#undef GSF_QSTAGE
            qi = ekf_normalize(quat_mul(cq, D));                         // one normalisation per chunk
            const Quat q_prev = prev_lane(cq, qi);
            u = quat_rotate(q_prev, dpl);
        }

        // ---- positions: prefix composition of affine maps x -> al x + be in chunk-local coordinates (x = p - p_carry)
        const double uu[3] = { u.x, u.y, u.z }, zl[3] = { z.x - cp.x, z.y - cp.y, z.z - cp.z };
        double xl[3], dcorr[3];
        double al[3], be[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double kw = kg[c] * wgt;
            al[c] = avail ? (1.0 - kw) : 1.0;
            be[c] = avail ? ((1.0 - kw) * uu[c] + kw * zl[c]) : uu[c];
        }
#define GSF_ASTAGE(c, CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al[c]), ob = dpp0<CTRL, RM>(be[c]); be[c] = al[c] * ob + be[c]; al[c] = al[c] * oa; }
#define GSF_ASTAGE_X(CTRL, RM) GSF_ASTAGE(0, CTRL, RM)
#define GSF_ASTAGE_Y(CTRL, RM) GSF_ASTAGE(1, CTRL, RM)
#define GSF_ASTAGE_Z(CTRL, RM) GSF_ASTAGE(2, CTRL, RM)
        // x and y share the gain in the default CONFIG, hence the multiplicative part: one joint scan of (al; be_x, be_y)
#define GSF_ASTAGE_XY(CTRL, RM) { const double oa = dpp<CTRL, RM>(1.0, al[0]), ob0 = dpp0<CTRL, RM>(be[0]), ob1 = dpp0<CTRL, RM>(be[1]); \
                                  be[0] = al[0] * ob0 + be[0]; be[1] = al[0] * ob1 + be[1]; al[0] = al[0] * oa; }
        if (same_axis[1] == 0) { GSF_SCAN_STAGES(GSF_ASTAGE_XY) }
        else { GSF_SCAN_STAGES(GSF_ASTAGE_X) GSF_SCAN_STAGES(GSF_ASTAGE_Y) }
        GSF_SCAN_STAGES(GSF_ASTAGE_Z)
#undef GSF_ASTAGE_XY
#undef GSF_ASTAGE_Z
#undef GSF_ASTAGE_Y
#undef GSF_ASTAGE_X
#undef GSF_ASTAGE
#pragma unroll
        for (int c = 0; c < 3; ++c) xl[c] = be[c];                      // x_i (the carry is x = 0)

        // ---- per-outage RTS (ref :906-922, :777-803).  Inside an outage x_f = x_p and P_f = P_p, so the gain product
        // telescopes: x_s[k] = x_f[k] + (P_f[k] / P_p[r]) (x_f[r] - x_p[r]) for k in [start, r-1], r = the recovery pose.
        double xo[3] = { xl[0], xl[1], xl[2] };                          // what is written out (filter state stays xl)
        if (rts_mask != 0ull) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dcorr[c] = xl[c] - (prev_lane(0.0, be[c]) + uu[c]);   // x_f[i] - x_p[i] (non-zero only where a fix was used)
            const int r1 = __ffsll((long long)rec_mask) - 1;             // first recovery of the chunk (rts_mask != 0, so there is one)
            const bool one_rec = (rec_mask & (rec_mask - 1ull)) == 0ull; // ... and the only one: the usual chunk that closes an outage
            // the recovery's correction and predicted variance as wave-uniform values: what every smoothed pose of a one-recovery chunk needs,
            // and what the patch of the chunks already written needs (two v_readlane per value instead of a ds_bpermute pair per lane and value)
            const double dr[3] = { lane_bcast(dcorr[0], r1), lane_bcast(dcorr[1], r1), lane_bcast(dcorr[2], r1) };
            const double ipr[3] = { fast_rcp(lane_bcast(Pm[0], r1)), fast_rcp(lane_bcast(Pm[1], r1)), fast_rcp(lane_bcast(Pm[2], r1)) };
            if (one_rec) {
                const bool in_run = active && !av && lane < r1;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (in_run) xo[c] = xl[c] + Pf[c] * ipr[c] * dr[c];
            } else {
                const u64 later = rec_mask & ~bits(0, lane);             // recoveries after this lane
                const int rl = later != 0ull ? __ffsll((long long)later) - 1 : 0;
                const bool in_run = active && !av && later != 0ull && (((rts_mask >> rl) & 1ull) != 0ull);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double drl = shidx(dcorr[c], rl), pr = shidx(Pm[c], rl);
                    if (in_run) xo[c] = xl[c] + Pf[c] * fast_rcp(pr) * drl;
                }
            }
            // outage carried in from earlier chunks and closed here by an RTS recovery: fix the rows already written
            if (!c_prev_avail) {
                if ((rts_mask >> r1) & 1ull) {
                    // The last two chunks of the run are patched from a lane-private LDS ring (the rows as they were written and
                    // their P_f, kept by every chunk that ended inside the outage): no global read-modify-write, no stamps to
                    // re-scan.  Only a run that reaches further back than 128 poses takes the memory path for its older chunks.
                    const int64_t kfirst = (c_ostart / 64) * 64, kring = (c0 - 128 > kfirst) ? c0 - 128 : kfirst;
                    double acc = 0.0;                                    // sum of dt over (ostart, k]
                    for (int64_t k0 = kfirst; k0 < kring; k0 += 64) {
                        const int64_t k = k0 + lane;
                        const double tk = tsb[k];
                        const double tkp = prev_lane((k0 > 0) ? tsb[k0 - 1] : tk, tk);
                        double dsum = (k > c_ostart) ? fmax(1e-6, tk - tkp) : 0.0;
#define GSF_SSTAGE(CTRL, RM) { dsum += dpp0<CTRL, RM>(dsum); }
                        GSF_SCAN_STAGES(GSF_SSTAGE)
#undef GSF_SSTAGE
                        const double tot = lane_bcast(dsum, 63);
                        if (k >= c_ostart) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const double Pk = cPos[c] + cfg.Qps[c] * (acc + dsum);      // P_f[k] inside the outage
                                pob[k * 3 + c] += Pk * ipr[c] * dr[c];
                            }
                        }
                        acc += tot;
                    }
                    for (int64_t k0 = kring; k0 < c0; k0 += 64) {
                        const int64_t k = k0 + lane;
                        const int slot = (int)(k0 >> 6) & 1;
                        if (k >= c_ostart) {
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                __builtin_nontemporal_store(gsf_ring[slot][c][lane] + gsf_ring[slot][3 + c][lane] * ipr[c] * dr[c], &pob[k * 3 + c]);
                        }
                    }
                }
            }
        }

        // ---- output rows (stored at the bottom of the iteration, after the prefetch has been waited for)
        const double o0 = cp.x + xo[0], o1 = cp.y + xo[1], o2 = cp.z + xo[2];

        // ---- carry to the next 64 poses (from the last active lane L)
        const bool open = ((a_mask >> L) & 1ull) == 0ull;                // the chunk ends inside an outage
        if (open) {
            // keep this chunk's rows and variances for the recovery that will smooth them (see the per-outage RTS above)
            { const int slot = (int)(c0 >> 6) & 1;
              gsf_ring[slot][0][lane] = o0; gsf_ring[slot][1][lane] = o1; gsf_ring[slot][2][lane] = o2;
              gsf_ring[slot][3][lane] = Pf[0]; gsf_ring[slot][4][lane] = Pf[1]; gsf_ring[slot][5][lane] = Pf[2]; }
            const u64 sm = start_mask & bits(0, L);
            if (sm != 0ull) {
                const int s = 63 - __clzll((long long)sm);
                c_ostart = c0 + s;
                c_seg_sharp = (f_mask & bits(s + 1, L)) != 0ull;
                cPos[0] = lane_bcast(Pf[0], s); cPos[1] = lane_bcast(Pf[1], s); cPos[2] = lane_bcast(Pf[2], s);
            } else {
                c_seg_sharp = c_seg_sharp || (f_mask & bits(0, L)) != 0ull;
            }
        }
        c_prev_avail = !open;
        if (!telescope) {                                                // (the fast path carries no state quaternion: see Cq above)
            cq = lane_bcast(qi, L);
            const Quat rL = lane_bcast(r, L);
            if (((ok_mask >> L) & 1ull) != 0ull) Cq = lane_bcast(quat_mul(cq, quat_conj(rL)), 0);
        }
        cp = Vec3{ cp.x + lane_bcast(xl[0], L), cp.y + lane_bcast(xl[1], L), cp.z + lane_bcast(xl[2], L) };
        cP[0] = lane_bcast(Pf[0], L); cP[1] = lane_bcast(Pf[1], L); cP[2] = lane_bcast(Pf[2], L);
        c_po = lane_bcast(p, L); c_r = lane_bcast(r, L); c_ok = ((ok_mask >> L) & 1ull) != 0ull; c_t = lane_bcast(t, L);
        chunk_arrived(nxt);                                              // the next chunk's rows; then this chunk's stores
        if (GSF_WIDE_STORES && !SMALLBATCH) {
            // Many waves per SIMD (C3-sized batches): the launch sits on the memory system under its mixed read + write stream, and
            // 8-byte stores at 24 / 32-byte stride leave L2 as incomplete lines (counter writes 1.11x the bytes).  The chunk's two output
            // slabs are contiguous (rows x 24 and rows x 32 bytes): the wave lays its rows out in LDS and stores the slabs as whole
            // 16-byte pieces, lane after lane -- four stores of a kilobyte each instead of seven strided ones.  A lone wave (C2) would
            // pay the LDS round trip on its critical path, so the small-batch build keeps the direct stores.
            double* gsf_out_stage = wave_stage();                         // (the input stage of this iteration has been read back: unpack_chunk ends on a wait)
            typedef double gsf_d2 __attribute__((ext_vector_type(2), aligned(8)));
            gsf_out_stage[lane * 3] = o0; gsf_out_stage[lane * 3 + 1] = o1; gsf_out_stage[lane * 3 + 2] = o2;
            gsf_out_stage[192 + lane * 4] = qi.x; gsf_out_stage[192 + lane * 4 + 1] = qi.y; gsf_out_stage[192 + lane * 4 + 2] = qi.z; gsf_out_stage[192 + lane * 4 + 3] = qi.w;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // one wave per block: its LDS operations complete in order, no barrier
            const int rows = L + 1, ppieces = (rows * 3) / 2, qpieces = rows * 2;
            const gsf_d2* sv = (const gsf_d2*)gsf_out_stage;
            gsf_d2* ps = (gsf_d2*)(pob + c0 * 3); gsf_d2* qs = (gsf_d2*)(qob + c0 * 4);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int pc = lane + 64 * k;
                if (pc < ppieces) __builtin_nontemporal_store(sv[pc], &ps[pc]);
                if (pc < qpieces) __builtin_nontemporal_store(sv[96 + pc], &qs[pc]);
            }
            if ((rows & 1) && lane == 0) __builtin_nontemporal_store(gsf_out_stage[rows * 3 - 1], &pob[(c0 + rows) * 3 - 1]);   // odd row count: the slab ends on half a piece
        } else if (active) {
            // streaming stores: the fused rows are not read again (except by the rare carried-outage fix-up, which stays
            // coherent through L2), so they should not wait in L2 for the end-of-kernel write-back
            __builtin_nontemporal_store(o0, &pob[i * 3]); __builtin_nontemporal_store(o1, &pob[i * 3 + 1]); __builtin_nontemporal_store(o2, &pob[i * 3 + 2]);
            __builtin_nontemporal_store(qi.x, &qob[i * 4]); __builtin_nontemporal_store(qi.y, &qob[i * 4 + 1]);
            __builtin_nontemporal_store(qi.z, &qob[i * 4 + 2]); __builtin_nontemporal_store(qi.w, &qob[i * 4 + 3]);
        }
        GSF_STAMP(8 + (int)(c0 / 64));
    }
    GSF_STAMP_FLUSH();
    if (lane == 0 && GSF_STATUS_PTR(a)) a.status[b] = (status | (c_prev_avail ? 0 : ST_ENDED_IN_OUTAGE)) | (PIPELINE ? (fit << 8) : 0);
}

template <bool PIPELINE, bool PREVAR = false, bool SMALLBATCH = false, int RINGS = 1, int AXMODE = 0>
__device__ __forceinline__ void wave_serial_body(const WaveArgs& a, const EkfConfig& cfg, const int64_t b, const int lane,
                                                 const double* pv = nullptr, const int pv_stride = 0, const int ring_slot = 0)
{
    GSF_STAMP(0);
    int64_t base, N; traj_span(a, b, base, N);
    if (N <= 0) { if (lane == 0 && GSF_STATUS_PTR(a)) a.status[b] = 0; return; }              // empty track (ref :835)
    const double* __restrict__ tsb = a.ts + base;
    const double* __restrict__ posb = a.pos + base * 3;
    const double* __restrict__ quatb = a.quat + base * 4;
    const double* __restrict__ gpsb = a.gps + base * 3;
    const uint8_t* __restrict__ valb = a.valid + base;

    // the first 64 poses are requested before the prelude (fit / initial pose), whose latency then covers theirs
    typename NextChunk<GSF_WIDE(SMALLBATCH)>::type nxt;
    if constexpr (GSF_WIDE(SMALLBATCH)) nxt = load_chunk_wide(tsb, posb, quatb, gpsb, valb, 0, lane, N);
    else nxt = load_chunk(tsb, posb, quatb, gpsb, valb, lane, N);
    __builtin_amdgcn_sched_barrier(0);                                   // ... and stay requested HERE: nothing of the prelude is scheduled above them
    Vec3 p0; Quat q0; int32_t fit = 0;
    if (!wave_prelude<PIPELINE>(a, b, base, N, lane, p0, q0, fit)) {
        if (PREVAR) __syncthreads();                                     // meet the helper wave at its barrier before leaving
        return;
    }
    GSF_STAMP(6);
    if (PREVAR) __syncthreads();                                         // the helper wave has written every chunk's variances

    wave_serial_chunks<PIPELINE, PREVAR, SMALLBATCH, RINGS, AXMODE>(a, cfg, b, lane, base, N, p0, q0, fit, nxt, pv, pv_stride, ring_slot);
}


}  // namespace
