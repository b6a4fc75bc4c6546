// gsf_robust.hip -- steps 3-5 of main_process_gui (EKFGPSSLAM.py:1002-1010) with the reference's ROBUST fit, as one device chain
// without a host round trip:
//   compact      rows with valid finite GNSS -> (src, dst) point sets in fixed-stride slots            (ref :977-998: the valid indices)
//   draw         max_trials x np.random.choice(n, min_samples, replace=False) per trajectory          (ref :405; gsf_rng.hip)
//   K2b          hypotheses, first-best inlier set, final Umeyama on the inliers                      (ref :404-421; gsf_sim3.hip)
//   init pose    Sim3 of pose 0 (the only pose of the aligned track the filter consumes)    (ref :1006, :842)
//   K4           EKF + per-outage RTS, wave per trajectory                                              (ref :1010; gsf_ekf_wave.hip)
//   finish       fit status into the status word, inlier mask back to original rows, NaN rows for a fit that is None
#include "gsf_wave_common.hpp"
#include "gsf_mt19937.hpp"
#include "gsf_ransac.hpp"

using namespace gsf;

namespace {

// (optional, the robust chain: fixed-stride trajectories) the chosen rows compacted into slot [b*N, b*N + n_b) by the same pass that marks them --
// what compact_valid_kernel does from the mask, without the mask's trip through memory and without its launch
struct RowsCompact { const double* pos; double* src; double* dst; int32_t* rowmap; int32_t* counts; int64_t* offsets; int64_t B; };

// one wave per trajectory: which rows feed the Sim3 fit, main_process_gui's way (ref :973-998; gsf_set_sim3_rows mode 1,
// gsf_sim3_fit_rows_batch_dev).  Pass 1 walks the valid rows for the first gap and counts what the duration limit keeps; the
// reference's two fall-backs are decided from the counts; pass 2 writes the mask.
// TILE: the trajectory's rows (stamp, fix, valid byte) are fetched ONCE into registers -- up to 64 x ROWS_TILE = 512 poses, every load in flight
// together -- and both passes run from there; the loop form makes three to five dependent trips to memory per pass (a lone wave per
// trajectory: each trip is ~0.7 us).  Same chunks, same order, same helper: the same words.
constexpr int ROWS_TILE = 8;
template <bool TILE>
__global__ __launch_bounds__(64) void sim3_rows_kernel(const double* __restrict__ ts, const double* __restrict__ gps, const uint8_t* __restrict__ valid,
                                                       const int64_t* __restrict__ offsets, int64_t N, FitRows rule, uint8_t* __restrict__ row_mask,
                                                       int32_t* __restrict__ n_rows, int32_t* __restrict__ status, RowsCompact cp)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    int64_t base = b * N, n = N;
    if (offsets) { base = uniform64(offsets[b]); n = uniform64(offsets[b + 1]) - base; }
    const double* tsb = ts + base; const uint8_t* valb = valid + base; const double* gpsb = gps ? gps + base * 3 : nullptr;
    auto row_ok = [&](const int64_t i) __attribute__((always_inline)) {
        if (i >= n) return false;
        bool ok = valb[i] != 0;
        if (gpsb) ok = ok && !(isnan(gpsb[i * 3]) || isnan(gpsb[i * 3 + 1]) || isnan(gpsb[i * 3 + 2]));
        return ok;
    };
    // TILE (launcher: gps != NULL, n <= 64 * ROWS_TILE): row k * 64 + lane
    double tt[ROWS_TILE], gx[ROWS_TILE], gy[ROWS_TILE], gz[ROWS_TILE];
    bool okk[ROWS_TILE];
    if (TILE) {
#pragma unroll
        for (int k = 0; k < ROWS_TILE; ++k) {
            const int64_t i = (int64_t)k * 64 + lane, ic = i < n ? i : n - 1;
            tt[k] = tsb[ic]; gx[k] = gpsb[ic * 3]; gy[k] = gpsb[ic * 3 + 1]; gz[k] = gpsb[ic * 3 + 2];
            okk[k] = i < n && valb[ic] != 0 && !(isnan(gx[k]) || isnan(gy[k]) || isnan(gz[k]));
        }
    }
    RowScan rs{ false, 0.0, 0, 0 };
    bool gap_found = false, have_t0 = false, carry_in_T = false;
    int row_end = (int)n;
    int nF = 0, nT = 0;
    double tlim = 0.0;
    auto pass1 = [&](const int64_t c0, const bool ok, const double t) __attribute__((always_inline)) {
        const u64 m = __ballot(ok);
        if (m == 0ull) return;
        if (!have_t0) { tlim = lane_bcast(t, __ffsll((long long)m) - 1) + rule.max_dur; have_t0 = true; }   // segment_start_time + max_dur (:988-990)
        bool in_chunk = false;
        gap_found = rows_gap_in_chunk(rs, m, t, ok, lane, (int)c0, rule.max_gap, row_end, nF, in_chunk);
        const u64 tm = __ballot(ok && t <= tlim);
        if (gap_found) {
            if (in_chunk) nT += __popcll(tm & bits(0, row_end - (int)c0 - 1));
            else nT -= carry_in_T ? 1 : 0;                                // the row in front of the gap is the carried one: it was counted, and V[:k] leaves it out
        } else {
            nT += __popcll(tm);
            carry_in_T = ((tm >> (63 - __clzll((long long)m))) & 1ull) != 0ull;
        }
    };
    if (TILE) {
#pragma unroll
        for (int k = 0; k < ROWS_TILE; ++k)
            if ((int64_t)k * 64 < n && !gap_found) pass1((int64_t)k * 64, okk[k], tt[k]);
    } else {
        for (int64_t c0 = 0; c0 < n && !gap_found; c0 += 64) {
            const int64_t i = c0 + lane;
            pass1(c0, row_ok(i), tsb[i < n ? i : n - 1]);
        }
    }
    if (!gap_found) nF = rs.nvalid;
    bool use_tlim = true;
    int32_t flag = 0; int count = nT;
    if (nF < rule.min_samples) {                                          // :983
        if (!gap_found) { flag = SIM3_FLAG_FEW_ROWS; count = -1; }       // :975
        else { row_end = (int)n; use_tlim = false; flag = SIM3_FLAG_ROWS_ALL; count = -2; }   // :984 (counted in pass 2)
    } else if (nT < rule.min_samples) { use_tlim = false; flag = SIM3_FLAG_ROWS_SEGMENT; count = nF; }   // :993-995
    int total = 0;
    auto pass2 = [&](const int64_t c0, const bool ok, const double t, const double z0, const double z1, const double z2) __attribute__((always_inline)) {
        const int64_t i = c0 + lane;
        const bool sel = ok && flag != SIM3_FLAG_FEW_ROWS && i < row_end && (!use_tlim || t <= tlim);
        const u64 sm = __ballot(sel);
        if (cp.src && sel) {
            const int64_t o = base + total + __popcll(sm & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            cp.src[o * 3] = cp.pos[(base + i) * 3]; cp.src[o * 3 + 1] = cp.pos[(base + i) * 3 + 1]; cp.src[o * 3 + 2] = cp.pos[(base + i) * 3 + 2];
            cp.dst[o * 3] = z0; cp.dst[o * 3 + 1] = z1; cp.dst[o * 3 + 2] = z2;
            cp.rowmap[o] = (int32_t)i;
        }
        total += __popcll(sm);
        if (i < n) row_mask[base + i] = sel ? 1 : 0;
    };
    if (TILE) {
#pragma unroll
        for (int k = 0; k < ROWS_TILE; ++k)
            if ((int64_t)k * 64 < n) pass2((int64_t)k * 64, okk[k], tt[k], gx[k], gy[k], gz[k]);
    } else {
        for (int64_t c0 = 0; c0 < n; c0 += 64) {
            const int64_t i = c0 + lane;
            const bool ok = row_ok(i);
            double z0 = 0.0, z1 = 0.0, z2 = 0.0;
            if (cp.src && ok) { z0 = gpsb[i * 3]; z1 = gpsb[i * 3 + 1]; z2 = gpsb[i * 3 + 2]; }
            pass2(c0, ok, tsb[i < n ? i : n - 1], z0, z1, z2);
        }
    }
    if (count == -2) {                                                    // all valid rows: still fewer than min_samples -> ValueError (:975)
        count = total;
        if (total < rule.min_samples) {
            flag = SIM3_FLAG_FEW_ROWS; count = -1; total = 0;
            for (int64_t i = lane; i < n; i += 64) row_mask[base + i] = 0;
        }
    }
    if (lane == 0) {
        n_rows[b] = count; if (status) status[b] = flag;
        if (cp.src) { cp.counts[b] = total; cp.offsets[b] = base; if (b == cp.B - 1) cp.offsets[cp.B] = cp.B * N; }
    }
}

// one wave per trajectory: stable compaction of the rows with valid, finite GNSS -- or, under the reference's row choice, of the rows
// `rowsel` marks -- into slot [b*N, b*N + n_b)
__global__ __launch_bounds__(64) void compact_valid_kernel(const double* __restrict__ pos, const double* __restrict__ gps, const uint8_t* __restrict__ valid,
                                                           const uint8_t* __restrict__ rowsel,
                                                           int64_t B, int64_t N, double* __restrict__ src, double* __restrict__ dst,
                                                           int32_t* __restrict__ rowmap, int32_t* __restrict__ counts, int64_t* __restrict__ offsets)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, base = b * N;
    int n = 0;
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        const int64_t i = c0 + lane;
        bool ok = false;
        double z0 = 0, z1 = 0, z2 = 0;
        if (i < N) {
            z0 = gps[(base + i) * 3]; z1 = gps[(base + i) * 3 + 1]; z2 = gps[(base + i) * 3 + 2];
            ok = rowsel ? rowsel[base + i] != 0 : (valid[base + i] != 0 && !(isnan(z0) || isnan(z1) || isnan(z2)));
        }
        const u64 m = __ballot(ok);
        if (ok) {
            const int k = n + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            const int64_t o = base + k;
            src[o * 3] = pos[(base + i) * 3]; src[o * 3 + 1] = pos[(base + i) * 3 + 1]; src[o * 3 + 2] = pos[(base + i) * 3 + 2];
            dst[o * 3] = z0; dst[o * 3 + 1] = z1; dst[o * 3 + 2] = z2;
            rowmap[o] = (int32_t)i;
        }
        n += __popcll(m);
    }
    if (lane == 0) { counts[b] = n; offsets[b] = base; if (b == B - 1) offsets[B] = B * N; }
}


// ---- exact early termination of the robust fit (gsf_set_option "ransac_early_exit") -------------------------------------------------
// The reference's loop keeps a trial only if it counts STRICTLY more rows than the best so far (ref :413), so once a trial has counted all
// n rows no later trial can change the inlier mask, the count, or the final fit: the rest of the max_trials draws only move the generator.
// One wave per trajectory draws and scores its own trials in rounds of PROBE_ROUND = 8 and stops after the round that holds the first trial
// that counts every row.  Why 8: a round costs one lane-parallel 4-point fit whatever its size -- the Jacobi SVD of K2b's fit_sample, ~11 us
// for a lone wave --, a trial its draw (~2.3 us at 271 rows) and one row-parallel count (~0.3 us), and the launch ends with its SLOWEST
// trajectory.  Measured at 1 000 x 271 (rocprofv3, gpurun_out/r5e): rounds of 1, 1, 2, 4, 8 -- the kept trial is trial 0 on 75 % of the
// tracks, within the first four on 99.3 %, at most trial 6 -- 79 us (four fits on the slowest track); rounds of 8 from the start: one fit.  A trajectory that is still undecided after `probe_trials` is handed to the wide kernels for the REST of its trials
// (mt_choice_kernel + K2b from trial `drawn` on, the arg-max key carried over), so data that never saturates costs what it cost before.
// Counts are formed with the functions K2b forms them with (gsf_ransac.hpp): the decision is the one the full chain takes.
//   keys[b][2]      arg-max key of the trials scored here (0 = none usable), [1] = 0 (no caller-fed sample can be out of range)
//   decided[b]      1 when a trial counted every row: R, t, s, mask, n_inliers are final; the generator stops after that trial's ROUND;
//                   2 when the final fit was formed here as well (R, t, s, fit status, mask, n_inliers written): K2b skips the set
//   trial_info[b]   { deciding trial or -1, trials drawn here }
constexpr int PROBE_ROUND = 8;
constexpr int PROBE_TILE = 8;       // row iterations of a set held in registers by the probe (64 rows each)
__global__ __launch_bounds__(64) void robust_probe_kernel(uint32_t* __restrict__ state, const double* __restrict__ src, const double* __restrict__ dst,
                                                          const int64_t* __restrict__ offsets, const int32_t* __restrict__ counts, int max_trials,
                                                          int probe_trials, int ms, double thr, int32_t* sample_idx, int jseq_bytes,
                                                          unsigned long long* __restrict__ keys, int32_t* __restrict__ decided,
                                                          int32_t* __restrict__ trial_info, int min_inliers, double* __restrict__ Rout,
                                                          double* __restrict__ tout, double* __restrict__ sout, int32_t* __restrict__ fit_status,
                                                          uint8_t* __restrict__ mask_c, int32_t* __restrict__ n_inliers)
{
    __shared__ uint32_t mt[MT_N + 1];
    extern __shared__ uint16_t jseq[];
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int n = counts[b];
    const int64_t i0 = offsets[b];
    int32_t* my_idx = sample_idx + (size_t)b * (size_t)max_trials * (size_t)ms;
    long long best = -1; int best_trial = 0x7fffffff, drawn = 0; bool sat = false, finished = false;
    if (!(n < ms || n < 1 || n > CHOICE_MAX_N)) {                         // else: the reference returns before drawing (ref :395-397), stream untouched
        uint32_t* st = state + b * MT_STATE_WORDS;
        for (int i = lane; i < MT_N; i += 64) mt[i] = st[i];
        int pos = (int)st[MT_N];
        __syncthreads();
        const int limit = probe_trials < max_trials ? probe_trials : max_trials;
        // the rows of the set, held in registers when they fit (up to 64 x PROBE_TILE = 512 rows: lane l holds rows l, l + 64, ...): fetched once
        // -- the loads travel while the first round is drawn -- instead of once per hypothesis (five dependent memory round trips per count
        // were 4 us of a hypothesis's 5, gpurun_out/r5e)
        const bool in_regs = n <= 64 * PROBE_TILE;
        double rx[PROBE_TILE][6];
#pragma unroll
        for (int k = 0; k < PROBE_TILE; ++k) {
            if (in_regs && k * 64 < n) {                                   // wave-uniform
                const int64_t r = i0 + (k * 64 + lane < n ? k * 64 + lane : n - 1);
                rx[k][0] = src[r * 3]; rx[k][1] = src[r * 3 + 1]; rx[k][2] = src[r * 3 + 2];
                rx[k][3] = dst[r * 3]; rx[k][4] = dst[r * 3 + 1]; rx[k][5] = dst[r * 3 + 2];
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) rx[k][c] = 0.0;
            }
        }
        while (drawn < limit && !sat) {
            int T = PROBE_ROUND;
            if (T > limit - drawn) T = limit - drawn;
            mt_draw_choice(mt, pos, n, T, ms, jseq, jseq_bytes / 2, my_idx + (size_t)drawn * ms, nullptr, lane);   // (ends on a block barrier: the sets are visible)
            double R[9], t[3], s = 0.0;
            bool ok = false;
            if (lane < T) ok = fit_sample(src, dst, i0, my_idx + (size_t)(drawn + lane) * ms, ms, R, t, s) != SIM3_NONE;   // ref :407-408
            const u64 okm = __ballot(ok);
            for (int h = 0; h < T; ++h) {                                  // wave-uniform: one hypothesis at a time, its rows spread over the lanes
                if (((okm >> h) & 1ull) == 0ull) continue;
                double Rh[9], th[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) Rh[k] = lane_bcast(R[k], h);
                th[0] = lane_bcast(t[0], h); th[1] = lane_bcast(t[1], h); th[2] = lane_bcast(t[2], h);
                const double sh = lane_bcast(s, h);
                long long cnt = 0;
                if (in_regs) {                                             // ref :409-412
#pragma unroll
                    for (int k = 0; k < PROBE_TILE; ++k) {
                        if (k * 64 < n) {                                  // wave-uniform
                            const bool in = (k * 64 + lane < n) && within(resid2_vals(rx[k][0], rx[k][1], rx[k][2], rx[k][3], rx[k][4], rx[k][5], Rh, th, sh), thr);
                            cnt += __popcll(__ballot(in));
                        }
                    }
                } else {
                    for (int r0 = 0; r0 < n; r0 += 64) {
                        const int r = r0 + lane;
                        const bool in = r < n && is_inlier(src, dst, i0 + (r < n ? r : n - 1), Rh, th, sh, thr);
                        cnt += __popcll(__ballot(in));
                    }
                }
                if (cnt > best) { best = cnt; best_trial = drawn + h; }    // strict > keeps the first (:413)
                if (cnt == (long long)n) { sat = true; break; }            // every row counted: nothing after this trial can be kept
            }
            drawn += T;
        }
        for (int i = lane; i < MT_N; i += 64) st[i] = mt[i];
        if (lane == 0) st[MT_N] = (uint32_t)pos;
        // ---- a decided set whose rows sit in registers is finished here: every row is an inlier of the kept trial (mask = all ones, count = n),
        // and the final fit (ref :420-421) is Umeyama over ALL rows -- what K2b's finishing block would compute after fitting the kept sample
        // again, marking the rows, and seventeen block reductions (32 us of the 131 us chain at 1 000 x 271; here ~8 us inside a launch that is
        // waiting for its slowest wave anyway).  The sums follow that block's order: thread v of its 256 adds rows v, v + 256, ... (tiles
        // k = w, w + 4, ... for the lanes of its wave w), a wave_sum per wave, the four waves added in turn (block_sum, gsf_sim3.hip).
        if (sat && in_regs && Rout) {
            finished = true;
            for (int k = lane; k < n; k += 64) mask_c[i0 + k] = 1;
            double Ro[9], to[3], so = NAN; int32_t fs = SIM3_NONE;
            if (n >= min_inliers) {                                       // ref :416-418 (the count is n)
                constexpr int VW = RANSAC_FINAL_THREADS / 64;
                double tot[7], H[10];
                {
                    double acc[VW][7];
#pragma unroll
                    for (int w = 0; w < VW; ++w)
#pragma unroll
                        for (int c = 0; c < 7; ++c) acc[w][c] = 0.0;
#pragma unroll
                    for (int k = 0; k < PROBE_TILE; ++k)
                        if (k * 64 + lane < n) final_moments1(acc[k % VW], rx[k][0], rx[k][1], rx[k][2], rx[k][3], rx[k][4], rx[k][5]);
#pragma unroll
                    for (int c = 0; c < 7; ++c) {
                        double r = 0.0;
#pragma unroll
                        for (int w = 0; w < VW; ++w) r += wave_sum(acc[w][c]);
                        tot[c] = r;
                    }
                }
                const double cntf = tot[0];
                const double sc[3] = { tot[1] / cntf, tot[2] / cntf, tot[3] / cntf }, dc[3] = { tot[4] / cntf, tot[5] / cntf, tot[6] / cntf };
                {
                    double h[VW][10];
#pragma unroll
                    for (int w = 0; w < VW; ++w)
#pragma unroll
                        for (int c = 0; c < 10; ++c) h[w][c] = 0.0;
#pragma unroll
                    for (int k = 0; k < PROBE_TILE; ++k)
                        if (k * 64 + lane < n) final_moments2(h[k % VW], rx[k][0], rx[k][1], rx[k][2], rx[k][3], rx[k][4], rx[k][5], sc, dc);
#pragma unroll
                    for (int c = 0; c < 10; ++c) {
                        double r = 0.0;
#pragma unroll
                        for (int w = 0; w < VW; ++w) r += wave_sum(h[w][c]);
                        H[c] = r;
                    }
                }
                if (cntf >= 3.0) fs = umeyama_finalize(H, H[9], sc, dc, cntf, Ro, to, so);    // every lane, wave-uniform inputs
            }
            if (fs == SIM3_NONE) { for (int k = 0; k < 9; ++k) Ro[k] = NAN; to[0] = to[1] = to[2] = NAN; so = NAN; }
            if (lane == 0) {
                for (int k = 0; k < 9; ++k) Rout[b * 9 + k] = Ro[k];
                tout[b * 3] = to[0]; tout[b * 3 + 1] = to[1]; tout[b * 3 + 2] = to[2]; sout[b] = so;
                fit_status[b] = fs; n_inliers[b] = n;
            }
        }
    }
    if (lane == 0) {
        keys[b * 2] = best >= 0 ? ransac_key(best, best_trial) : 0ull; keys[b * 2 + 1] = 0ull;
        decided[b] = sat ? (finished ? 2 : 1) : 0;
        trial_info[b * 2] = sat ? best_trial : -1; trial_info[b * 2 + 1] = drawn;
    }
}

// lane per trajectory: Sim3 of pose 0 (transform_trajectory row 0, ref :464-466)
__global__ __launch_bounds__(64) void robust_init_pose_kernel(const double* __restrict__ pos, const double* __restrict__ quat, int64_t B, int64_t N,
                                                              const double* __restrict__ R, const double* __restrict__ t, const double* __restrict__ s,
                                                              const int32_t* __restrict__ fit, double* __restrict__ init_pos, double* __restrict__ init_quat,
                                                              int32_t* __restrict__ fail)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    Quat qn; const bool qok = quat_unit(Quat{ quat[b * N * 4], quat[b * N * 4 + 1], quat[b * N * 4 + 2], quat[b * N * 4 + 3] }, qn);
    const bool none = (fit[b] & 1) != 0;
    double Rb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rb[k] = R[b * 9 + k];
    const double x = pos[b * N * 3], y = pos[b * N * 3 + 1], z = pos[b * N * 3 + 2], sb = s[b];
    Vec3 p0{ sb * (x * Rb[0] + y * Rb[1] + z * Rb[2]) + t[b * 3], sb * (x * Rb[3] + y * Rb[4] + z * Rb[5]) + t[b * 3 + 1],
             sb * (x * Rb[6] + y * Rb[7] + z * Rb[8]) + t[b * 3 + 2] };
    Quat q0 = quat_mul(quat_from_matrix(Rb), qn);
    if (none || !qok) { p0 = Vec3{ 0.0, 0.0, 0.0 }; q0 = Quat{ 0.0, 0.0, 0.0, 1.0 }; }      // the filter still runs; finish() blanks the rows
    init_pos[b * 3] = p0.x; init_pos[b * 3 + 1] = p0.y; init_pos[b * 3 + 2] = p0.z;
    init_quat[b * 4] = q0.x; init_quat[b * 4 + 1] = q0.y; init_quat[b * 4 + 2] = q0.z; init_quat[b * 4 + 3] = q0.w;
    fail[b] = (none ? 1 : 0) | (qok ? 0 : 2);
}

// one wave per trajectory: status word, inlier mask in original row order, NaN rows when the fit is None / pose 0 is invalid
__global__ __launch_bounds__(64) void robust_finish_kernel(int64_t N, const int32_t* __restrict__ fit, const int32_t* __restrict__ fail,
                                                           const int32_t* __restrict__ rows_status, const int32_t* __restrict__ counts, const int32_t* __restrict__ rowmap,
                                                           const uint8_t* __restrict__ mask_c, uint8_t* __restrict__ inlier_mask,
                                                           double* __restrict__ pos_out, double* __restrict__ quat_out, int32_t* __restrict__ status,
                                                           const int32_t* __restrict__ decided, const unsigned long long* __restrict__ keys,
                                                           int32_t* __restrict__ trial_info, const int32_t* __restrict__ probe_info, int max_trials, int min_samples)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, base = b * N;
    const int f = fail[b];
    const int32_t sat = (decided && decided[b]) ? SIM3_FLAG_SATURATED : 0;
    if (trial_info && lane == 0) {
        // { the trial whose inlier set was kept (first one with the best count, ref :413) or -1, trials drawn from the generator }
        int win = -1, drawn = counts[b] < min_samples ? 0 : max_trials;
        if (keys) { const unsigned long long k = keys[b * 2]; if (k) win = 0x7fffffff - (int)(k & 0xffffffffull); }
        if (probe_info && decided && decided[b]) drawn = probe_info[b * 2 + 1];
        trial_info[b * 2] = win; trial_info[b * 2 + 1] = drawn;
    }
    if (inlier_mask) {
        for (int64_t i = lane; i < N; i += 64) inlier_mask[base + i] = 0;
        __syncthreads();
        const int n = counts[b];
        for (int k = lane; k < n; k += 64) if (mask_c[base + k]) inlier_mask[base + rowmap[base + k]] = 1;
    }
    if (f != 0) {
        for (int64_t i = lane; i < N; i += 64) {
            pos_out[(base + i) * 3] = NAN; pos_out[(base + i) * 3 + 1] = NAN; pos_out[(base + i) * 3 + 2] = NAN;
            quat_out[(base + i) * 4] = NAN; quat_out[(base + i) * 4 + 1] = NAN; quat_out[(base + i) * 4 + 2] = NAN; quat_out[(base + i) * 4 + 3] = NAN;
        }
        const int32_t few = rows_status ? (rows_status[b] & SIM3_FLAG_FEW_ROWS) : 0;      // the reference raised ValueError before the fit (:975, :997)
        // (a trajectory the probe decided keeps its SATURATED bit when the fit then fails -- every row counted, but fewer rows than
        // min_inliers_needed, or a final fit that is None: its generator did stop early, and the bit is what says so)
        if (lane == 0) status[b] = ((f & 1) ? ((SIM3_NONE | few | sat) << 8) : (sat << 8)) | ((f & 2) ? ST_BAD_QUAT : 0);
    } else if (lane == 0) {
        status[b] = (status[b] & 0xff) | ((fit[b] | (rows_status ? rows_status[b] : 0) | sat) << 8);
    }
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

namespace gsf {
int launch_sim3_rows(gsf_ctx* ctx, const double* ts, const double* gps, const uint8_t* valid, const int64_t* offsets, int64_t B, int64_t N,
                     const FitRows& rule, uint8_t* row_mask, int32_t* n_rows, int32_t* status)
{
    const RowsCompact none{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0 };
    if (!offsets && gps && N <= 64 * ROWS_TILE)
        hipLaunchKernelGGL(sim3_rows_kernel<true>, dim3((unsigned)B), dim3(64), 0, ctx->stream, ts, gps, valid, offsets, N, rule, row_mask, n_rows, status, none);
    else
        hipLaunchKernelGGL(sim3_rows_kernel<false>, dim3((unsigned)B), dim3(64), 0, ctx->stream, ts, gps, valid, offsets, N, rule, row_mask, n_rows, status, none);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
}  // namespace gsf

extern "C" int gsf_fuse_pipeline_robust_info_batch_dev(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps,
                                                       const uint8_t* valid, const gsf_ekf_config* cfg, int64_t B, int64_t N, int32_t min_samples,
                                                       double residual_threshold, int32_t max_trials, int32_t min_inliers_needed, uint32_t* mt_state,
                                                       double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status,
                                                       int32_t* n_inliers, uint8_t* inlier_mask, int32_t* trial_info)
{
    GSF_REQUIRE(ctx && cfg, "ctx/cfg is NULL");
    GSF_REQUIRE(B >= 0 && N >= 0 && B <= 0x7fffffff, "bad B or N");
    GSF_REQUIRE(min_samples >= 1 && min_samples <= 64 && max_trials >= 0 && max_trials <= (1 << 20), "min_samples must be in [1,64], max_trials in [0, 2^20]");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(N <= 28000, "N too large for the device-side draws (<= 28000 poses per trajectory)");
    GSF_REQUIRE(ts && pos && quat && gps && valid && mt_state && R && t && s && pos_out && quat_out && status && n_inliers, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    const size_t P = (size_t)B * (size_t)N, nb = (size_t)B;
    // workspace (context scratch, grow-only): point sets, row map, compact mask, counts/offsets, sample sets, fit status, initial poses
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off = align_up(off + bytes); return at; };
    const size_t o_src = take(P * 24), o_dst = take(P * 24), o_map = take(P * 4), o_mask = take(P), o_cnt = take(nb * 4), o_off = take((nb + 1) * 8),
                 o_idx = take(nb * (size_t)max_trials * (size_t)min_samples * 4 + 4), o_fit = take(nb * 4), o_fail = take(nb * 4), o_ip = take(nb * 24),
                 o_iq = take(nb * 32), o_sel = take(P), o_rst = take(nb * 4), o_rn = take(nb * 4), o_key = take(nb * 16), o_dec = take(nb * 4),
                 o_pinfo = take(nb * 8);
    int rc = ensure_scratch(ctx, off);
    if (rc) return rc;
    char* w = (char*)ctx->scratch;
    double* src = (double*)(w + o_src); double* dst = (double*)(w + o_dst); int32_t* rowmap = (int32_t*)(w + o_map); uint8_t* mask_c = (uint8_t*)(w + o_mask);
    int32_t* counts = (int32_t*)(w + o_cnt); int64_t* offsets = (int64_t*)(w + o_off); int32_t* idx = (int32_t*)(w + o_idx);
    int32_t* fit = (int32_t*)(w + o_fit); int32_t* fail = (int32_t*)(w + o_fail); double* ip = (double*)(w + o_ip); double* iq = (double*)(w + o_iq);
    // the rows the fit may draw from: every valid row, or the reference's choice (ref :973-998)
    uint8_t* rowsel = nullptr; int32_t* rows_status = nullptr;
    if (ctx->fit_rows.mode != 0) {                                        // row choice and compaction in one launch
        rowsel = (uint8_t*)(w + o_sel); rows_status = (int32_t*)(w + o_rst);
        const RowsCompact cp{ pos, src, dst, rowmap, counts, offsets, B };
        if (N <= 64 * ROWS_TILE)
            hipLaunchKernelGGL(sim3_rows_kernel<true>, dim3((unsigned)B), dim3(64), 0, ctx->stream, ts, gps, valid, (const int64_t*)nullptr, N, ctx->fit_rows, rowsel,
                               (int32_t*)(w + o_rn), rows_status, cp);
        else
            hipLaunchKernelGGL(sim3_rows_kernel<false>, dim3((unsigned)B), dim3(64), 0, ctx->stream, ts, gps, valid, (const int64_t*)nullptr, N, ctx->fit_rows, rowsel,
                               (int32_t*)(w + o_rn), rows_status, cp);
    } else {
        hipLaunchKernelGGL(compact_valid_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, pos, gps, valid, (const uint8_t*)rowsel, B, N, src, dst, rowmap, counts, offsets);
    }
    GSF_HIP(hipGetLastError());
    unsigned long long* keys = (unsigned long long*)(w + o_key);         // arg-max key per trajectory: K2b leaves the winner's trial there (trial_info)
    int32_t* decided = nullptr; int32_t* pinfo = nullptr; int32_t trial0 = 0;
    if (!(ctx->ransac_early_exit != 0 && max_trials > 0)) GSF_HIP(hipMemsetAsync(keys, 0, nb * 16, ctx->stream));
    if (ctx->ransac_early_exit != 0 && max_trials > 0) {
        // growing rounds of drawn-and-scored trials per trajectory until one counts every row (ref :413), at most probe_trials of them ...
        decided = (int32_t*)(w + o_dec); pinfo = (int32_t*)(w + o_pinfo);
        trial0 = ctx->ransac_probe_trials < max_trials ? ctx->ransac_probe_trials : max_trials;
        const int bytes = choice_lds_bytes(N);
        hipLaunchKernelGGL(robust_probe_kernel, dim3((unsigned)B), dim3(64), (size_t)bytes, ctx->stream, mt_state, (const double*)src, (const double*)dst,
                           (const int64_t*)offsets, (const int32_t*)counts, (int)max_trials, (int)trial0, (int)min_samples, residual_threshold, idx, bytes, keys,
                           decided, pinfo, (int)min_inliers_needed, R, t, s, fit, mask_c, n_inliers);
        GSF_HIP(hipGetLastError());
        // ... then the wide kernels for the rest of the trials of the trajectories that are still undecided
        if ((rc = launch_mt_choice_rest(ctx, mt_state, counts, B, max_trials, trial0, min_samples, idx, (int32_t)N, decided))) return rc;
    } else if (max_trials > 0 && (rc = launch_mt_choice(ctx, mt_state, counts, B, max_trials, min_samples, idx, (int32_t)N))) return rc;
    if ((rc = launch_sim3_ransac(ctx, src, dst, offsets, counts, B, idx, max_trials, min_samples, residual_threshold, min_inliers_needed, R, t, s, fit,
                                 mask_c, n_inliers, (int64_t)P, trial0, keys, decided))) return rc;
    hipLaunchKernelGGL(robust_init_pose_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, pos, quat, B, N, R, t, s, fit, ip, iq, fail);
    GSF_HIP(hipGetLastError());
    if ((rc = launch_ekf_wave(ctx, false, ts, pos, quat, gps, valid, ip, iq, cfg, B, N, nullptr, nullptr, nullptr, pos_out, quat_out, status))) return rc;
    hipLaunchKernelGGL(robust_finish_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, N, fit, fail, (const int32_t*)rows_status, counts, rowmap, mask_c, inlier_mask, pos_out, quat_out, status,
                       (const int32_t*)decided, (const unsigned long long*)keys, trial_info, (const int32_t*)pinfo, (int)max_trials, (int)min_samples);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

extern "C" int gsf_fuse_pipeline_robust_batch_dev(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, const double* gps,
                                                  const uint8_t* valid, const gsf_ekf_config* cfg, int64_t B, int64_t N, int32_t min_samples,
                                                  double residual_threshold, int32_t max_trials, int32_t min_inliers_needed, uint32_t* mt_state,
                                                  double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status,
                                                  int32_t* n_inliers, uint8_t* inlier_mask)
{
    return gsf_fuse_pipeline_robust_info_batch_dev(ctx, ts, pos, quat, gps, valid, cfg, B, N, min_samples, residual_threshold, max_trials, min_inliers_needed,
                                                   mt_state, R, t, s, pos_out, quat_out, status, n_inliers, inlier_mask, nullptr);
}

// main_process_gui's row choice on its own (ref :973-998)
extern "C" int gsf_sim3_fit_rows_batch_dev(gsf_ctx* ctx, const double* ts, const double* gps, const uint8_t* valid, const int64_t* offsets, int64_t B,
                                           int64_t N, int32_t min_samples, double max_gps_gap_threshold, double max_initial_duration,
                                           uint8_t* row_mask, int32_t* n_rows, int32_t* status)
{
    GSF_REQUIRE(ctx, "ctx is NULL");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff && (offsets || N >= 0) && min_samples >= 0, "bad B, N or min_samples");
    if (B == 0 || (!offsets && N == 0)) return GSF_OK;
    GSF_REQUIRE(ts && valid && row_mask && n_rows, "NULL array");
    GSF_HIP(hipSetDevice(ctx->device));
    return launch_sim3_rows(ctx, ts, gps, valid, offsets, B, N, FitRows{ 1, min_samples, max_gps_gap_threshold, max_initial_duration }, row_mask, n_rows, status);
}
