// gsf_run.hip -- steps 1-6 of main_process_gui (EKFGPSSLAM.py:959-1033) for B trajectories as ONE device chain, no host round trip:
//   geodesy slice    lat/lon range mask, zone pick, UTM forward, [E, N, alt] rows                (ref :258-271;  gsf_utm.hip)
//   loaded rows      the fixes the loader keeps, compacted per log                                (ref :259-264)
//   pre-filter       sliding-window polynomial RANSAC, windows walked on the device               (ref :275, :136-247;  gsf_gpsfilter.hip)
//   filtered log     rows the pre-filter drops blanked; "fewer than 2 fixes" flagged               (ref :283, :967)
//   alignment        dynamic_time_alignment to the SLAM stamps                                     (ref :971;  gsf_align.hip)
//   steps 3-5        row choice, robust fit, Sim3 of pose 0, EKF + RTS                             (ref :973-1010;  gsf_robust.hip)
//   step 4 in full   transform_trajectory of every pose (the metric's "Sim3" row)                  (ref :1006;  gsf_sim3.hip)
//   step 6           nearest-fix error of raw SLAM / Sim3 / EKF against the primary GPS            (ref :1013-1033;  gsf_eval.hip)
//   outcome          run_status per trajectory, NaN outputs where the reference raises
// Each trajectory's legacy MT19937 stream is used by the pre-filter first and the robust fit second, in the reference's order.
#include "gsf_wave_common.hpp"

using namespace gsf;

namespace {

// one wave per log: stable compaction of the fixes the loader keeps (ref :259-264: the geodesy slice marks a dropped fix by NaN easting AND
// northing) into slot [gps_offsets[b], +counts[b]); rowmap = the row of the log each slot came from
__global__ __launch_bounds__(64) void run_compact_rows_kernel(const double* __restrict__ gps_t, const double* __restrict__ utm, const int64_t* __restrict__ offsets,
                                                              double* __restrict__ ct, double* __restrict__ cp, int32_t* __restrict__ rowmap,
                                                              int32_t* __restrict__ counts)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, base = offsets[b], n_log = offsets[b + 1] - base;
    int n = 0;
    for (int64_t c0 = 0; c0 < n_log; c0 += 64) {
        const int64_t i = c0 + lane;
        double e = NAN, nn = NAN, a = NAN, tt = 0.0;
        if (i < n_log) { e = utm[(base + i) * 3]; nn = utm[(base + i) * 3 + 1]; a = utm[(base + i) * 3 + 2]; tt = gps_t[base + i]; }
        const bool ok = i < n_log && !(isnan(e) && isnan(nn));
        const u64 m = __ballot(ok);
        if (ok) {
            const int64_t o = base + n + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            ct[o] = tt; cp[o * 3] = e; cp[o * 3 + 1] = nn; cp[o * 3 + 2] = a; rowmap[o] = (int32_t)i;
        }
        n += __popcll(m);
    }
    if (lane == 0) counts[b] = n;
}

// one wave per log: what load_gps_data returns (ref :275-287) as a mask over the ORIGINAL rows and as a copy of the UTM rows in which every
// other row is blanked (NaN easting and northing: the alignment drops such rows when it stages a log); flags GPS_EMPTY / GPS_FEW /
// PREFILTER_UNHANDLED
__global__ __launch_bounds__(64) void run_filtered_rows_kernel(const double* __restrict__ utm, const int64_t* __restrict__ offsets, const int32_t* __restrict__ counts,
                                                               const int32_t* __restrict__ rowmap, const uint8_t* __restrict__ ckeep,
                                                               const int32_t* __restrict__ log_status, double* __restrict__ fut,
                                                               uint8_t* __restrict__ gps_keep, int32_t* __restrict__ run_status,
                                                               int64_t* __restrict__ slam_off, int32_t* __restrict__ bad_quat, int64_t B, int64_t N)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, base = offsets[b], n_log = offsets[b + 1] - base;
    // (two chores of the later steps ride along: the fixed-stride offsets of the SLAM tracks, and the zeroed flag K3 ORs into)
    if (lane == 0) { slam_off[b] = b * N; if (b == B - 1) slam_off[B] = B * N; bad_quat[b] = 0; }
    const int n = counts[b];
    const bool unhandled = log_status[b] != 0;
    for (int64_t i = lane; i < n_log; i += 64) {
        gps_keep[base + i] = 0;
        fut[(base + i) * 3] = NAN; fut[(base + i) * 3 + 1] = NAN; fut[(base + i) * 3 + 2] = utm[(base + i) * 3 + 2];
    }
    __syncthreads();
    int kept = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const bool keep = k < n && !unhandled && ckeep[base + k] != 0;
        if (keep) {
            const int64_t r = base + rowmap[base + k];
            gps_keep[r] = 1;
            fut[r * 3] = utm[r * 3]; fut[r * 3 + 1] = utm[r * 3 + 1];
        }
        kept += __popcll(__ballot(keep));
    }
    if (lane == 0) run_status[b] = (n == 0 ? GSF_RUN_GPS_EMPTY : 0) | (unhandled ? GSF_RUN_PREFILTER_UNHANDLED : ((n > 0 && kept < 2) ? GSF_RUN_GPS_FEW : 0));
}

// one wave per trajectory: the reference stopped before (or at) the fit -> every output of the later steps is NaN; Sim3 failures are flagged
__global__ __launch_bounds__(64) void run_outcome_kernel(int64_t B, int64_t N, const int32_t* __restrict__ status, int32_t* __restrict__ run_status,
                                                         double* __restrict__ R, double* __restrict__ t, double* __restrict__ s,
                                                         double* __restrict__ pos_out, double* __restrict__ quat_out, double* __restrict__ sim3_pos,
                                                         double* __restrict__ err_stats, int32_t* __restrict__ n_inliers, const int32_t* __restrict__ bad_quat)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, base = b * N;
    int32_t rs = run_status[b];
    // a SLAM quaternion that cannot be normalised: SciPy raises in transform_trajectory (ref :466), the run ends in step 4
    if (bad_quat[b] != 0 && (rs & (GSF_RUN_GPS_EMPTY | GSF_RUN_GPS_FEW | GSF_RUN_PREFILTER_UNHANDLED)) == 0 && ((status[b] >> 8) & SIM3_NONE) == 0) rs |= GSF_RUN_BAD_QUAT;
    if (((status[b] >> 8) & SIM3_NONE) != 0 && (rs & (GSF_RUN_GPS_EMPTY | GSF_RUN_GPS_FEW | GSF_RUN_PREFILTER_UNHANDLED)) == 0) rs |= GSF_RUN_SIM3_FAILED;
    if (rs != 0) {
        for (int64_t i = lane; i < N; i += 64) {
            for (int c = 0; c < 3; ++c) { pos_out[(base + i) * 3 + c] = NAN; if (sim3_pos) sim3_pos[(base + i) * 3 + c] = NAN; }
            for (int c = 0; c < 4; ++c) quat_out[(base + i) * 4 + c] = NAN;
        }
        if (lane < 9) R[b * 9 + lane] = NAN;
        if (lane < 3) t[b * 3 + lane] = NAN;
        if (lane == 0) s[b] = NAN;
        // the metric of a run that raised was never printed: count 0 and NaN rows (the raw-SLAM row too: the reference never got to step 6)
        if (lane < 12) err_stats[((int64_t)(lane / 4) * B + b) * 4 + (lane & 3)] = (lane & 3) == 0 ? 0.0 : NAN;
        if (lane == 0 && (rs & (GSF_RUN_GPS_EMPTY | GSF_RUN_GPS_FEW | GSF_RUN_PREFILTER_UNHANDLED)) != 0) n_inliers[b] = -1;
    }
    if (lane == 0) run_status[b] = rs;
}

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int gsf_run_fusion_batch_dev(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, int64_t B, int64_t N,
                                        const double* gps_t, const double* gps_llh, const int64_t* gps_offsets, int64_t total_fixes,
                                        int32_t max_fixes, const gsf_run_config* cfg, uint32_t* mt_state, double* R, double* t, double* s,
                                        double* pos_out, double* quat_out, int32_t* status, int32_t* n_inliers, int32_t* zone, int32_t* south,
                                        double* gps_utm, uint8_t* gps_keep, double* aligned, uint8_t* valid, double* sim3_pos,
                                        double* err_stats, int32_t* run_status, uint8_t* inlier_mask, int32_t* trial_info)
{
    GSF_REQUIRE(ctx && cfg, "ctx/cfg is NULL");
    GSF_REQUIRE(B >= 0 && N >= 0 && B <= 0x7fffffff && total_fixes >= 0 && max_fixes >= 0, "bad B, N, total_fixes or max_fixes");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && gps_offsets && mt_state && R && t && s && pos_out && quat_out && status && n_inliers && aligned && valid && err_stats &&
                run_status && (!gps_llh || (zone && south)), "NULL array");
    GSF_REQUIRE(total_fixes == 0 || (gps_t && gps_utm && gps_keep), "NULL GNSS array");
    GSF_REQUIRE(N <= 28000, "N too large for the device-side draws (<= 28000 poses per trajectory)");
    GSF_HIP(hipSetDevice(ctx->device));
    const size_t P = (size_t)B * (size_t)N, nb = (size_t)B, T = (size_t)(total_fixes > 0 ? total_fixes : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off = up256(off + bytes); return at; };
    const size_t o_ct = take(T * 8), o_cp = take(T * 24), o_map = take(T * 4), o_ck = take(T), o_cnt = take(nb * 4), o_ls = take(nb * 4), o_li = take(nb * 8),
                 o_fut = take(T * 24), o_so = take((nb + 1) * 8), o_as = take(nb * 4), o_sp = take(sim3_pos ? 0 : P * 24), o_sq = take(P * 32),
                 o_bq = take(nb * 4), o_err = take(P * 24);
    int rc = ensure_run_scratch(ctx, off);
    if (rc) return rc;
    char* w = (char*)ctx->run_scratch;
    double* ct = (double*)(w + o_ct); double* cp = (double*)(w + o_cp); int32_t* rowmap = (int32_t*)(w + o_map); uint8_t* ckeep = (uint8_t*)(w + o_ck);
    int32_t* counts = (int32_t*)(w + o_cnt); int32_t* log_status = (int32_t*)(w + o_ls); int32_t* log_info = (int32_t*)(w + o_li);
    double* fut = (double*)(w + o_fut); int64_t* slam_off = (int64_t*)(w + o_so); int32_t* align_status = (int32_t*)(w + o_as);
    double* sp = sim3_pos ? sim3_pos : (double*)(w + o_sp); double* sq = (double*)(w + o_sq); int32_t* badq = (int32_t*)(w + o_bq);
    double* errs = (double*)(w + o_err);
    // ---- step 1 (GPS side of load_gps_data)
    // (gps_llh == NULL: the caller's gps_utm rows are the projected log already)
    if (gps_llh && (rc = gsf_gps_rows_to_utm_batch_dev(ctx, gps_llh, gps_offsets, B, gps_utm, zone, south))) return rc;
    hipLaunchKernelGGL(run_compact_rows_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, gps_t, (const double*)gps_utm, gps_offsets, ct, cp, rowmap, counts);
    GSF_HIP(hipGetLastError());
    if ((rc = launch_gps_prefilter_auto(ctx, ct, cp, gps_offsets, counts, B, max_fixes > 0 ? max_fixes : 1, &cfg->gps_filter, mt_state, ckeep, log_status, log_info))) return rc;
    hipLaunchKernelGGL(run_filtered_rows_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, (const double*)gps_utm, gps_offsets, (const int32_t*)counts,
                       (const int32_t*)rowmap, (const uint8_t*)ckeep, (const int32_t*)log_status, fut, gps_keep, run_status, slam_off, badq, B, N);
    GSF_HIP(hipGetLastError());
    // ---- step 2
    if ((rc = gsf_time_align_loaded_rows_batch_dev(ctx, ts, slam_off, gps_t, fut, gps_offsets, B, max_fixes > 2 ? max_fixes : 2, cfg->max_gps_gap_threshold,
                                                   aligned, valid, align_status))) return rc;
    // ---- steps 3-5 on the rows main_process_gui picks (ref :973-998), whatever the context's own row rule is
    const FitRows saved = ctx->fit_rows;
    ctx->fit_rows = FitRows{ 1, cfg->sim3_min_samples, cfg->max_gps_gap_threshold, cfg->sim3_max_initial_duration };
    rc = gsf_fuse_pipeline_robust_info_batch_dev(ctx, ts, pos, quat, aligned, valid, &cfg->ekf, B, N, cfg->sim3_min_samples, cfg->sim3_residual_threshold,
                                                 cfg->sim3_max_trials, cfg->sim3_min_inliers_needed, mt_state, R, t, s, pos_out, quat_out, status, n_inliers,
                                                 inlier_mask, trial_info);
    ctx->fit_rows = saved;
    if (rc) return rc;
    // ---- step 4 for every pose, step 6
    if ((rc = launch_apply_sim3(ctx, pos, quat, slam_off, B, R, t, s, sp, sq, badq, true))) return rc;
    if ((rc = launch_eval_errors3(ctx, ts, pos, sp, pos_out, aligned, valid, B, N, cfg->eval_skip_seconds, err_stats, errs))) return rc;
    hipLaunchKernelGGL(run_outcome_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, B, N, (const int32_t*)status, run_status, R, t, s, pos_out, quat_out,
                       sim3_pos, err_stats, n_inliers, (const int32_t*)badq);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

// the same with host arrays (what a cgo / JNI / ctypes caller with its data in host memory calls): one staged upload, the chain, one download
extern "C" int gsf_run_fusion_batch(gsf_ctx* ctx, const double* ts, const double* pos, const double* quat, int64_t B, int64_t N, const double* gps_t,
                                    const double* gps_llh, const int64_t* gps_offsets, const gsf_run_config* cfg, uint32_t* mt_state, double* R,
                                    double* t, double* s, double* pos_out, double* quat_out, int32_t* status, int32_t* n_inliers, int32_t* zone,
                                    int32_t* south, double* gps_utm, uint8_t* gps_keep, double* aligned, uint8_t* valid, double* sim3_pos,
                                    double* err_stats, int32_t* run_status, uint8_t* inlier_mask, int32_t* trial_info)
{
    GSF_REQUIRE(ctx && cfg && B >= 0 && N >= 0 && gps_offsets && mt_state, "bad arguments");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE(ts && pos && quat && R && t && s && pos_out && quat_out && status && n_inliers && zone && south && aligned && valid && err_stats && run_status,
                "NULL array");
    const int64_t total = gps_offsets[B];
    GSF_REQUIRE(total >= 0 && (total == 0 || (gps_t && gps_llh && gps_utm && gps_keep)), "bad gps_offsets / NULL GNSS array");
    int64_t max_fixes = 0;
    for (int64_t b = 0; b < B; ++b) { const int64_t g = gps_offsets[b + 1] - gps_offsets[b]; GSF_REQUIRE(g >= 0, "gps_offsets must not decrease"); if (g > max_fixes) max_fixes = g; }
    GSF_REQUIRE(max_fixes <= 0x7fffffff, "a log is too long");
    const size_t P = (size_t)B * (size_t)N, nb = (size_t)B, T = (size_t)total;
    Staging st(ctx, P * (64 + 56 + 24 + 1 + 24 + 1) + T * (8 + 24 + 24 + 1) + nb * (8 + 625 * 8 + 13 * 8 + 5 * 4 + 96 + 8) + 4096, 24);
    if (st.rc()) return st.rc();
    const double* dts = st.in(ts, P); const double* dpos = st.in(pos, P * 3); const double* dquat = st.in(quat, P * 4);
    const double* dgt = st.in(gps_t, T); const double* dllh = st.in(gps_llh, T * 3); const int64_t* doff = st.in(gps_offsets, nb + 1);
    const uint32_t* dst_in = st.in(mt_state, nb * 625);
    uint32_t* dstate = st.out(mt_state, nb * 625);
    double* dR = st.out(R, nb * 9); double* dt = st.out(t, nb * 3); double* ds = st.out(s, nb);
    double* dpo = st.out(pos_out, P * 3); double* dqo = st.out(quat_out, P * 4); int32_t* dstat = st.out(status, nb); int32_t* dni = st.out(n_inliers, nb);
    int32_t* dzone = st.out(zone, nb); int32_t* dsouth = st.out(south, nb);
    double* dutm = st.out(gps_utm, T * 3); uint8_t* dkeep = st.out(gps_keep, T);
    double* dal = st.out(aligned, P * 3); uint8_t* dva = st.out(valid, P);
    double* dsp = sim3_pos ? st.out(sim3_pos, P * 3) : nullptr;
    double* derr = st.out(err_stats, nb * 12); int32_t* drs = st.out(run_status, nb);
    uint8_t* dmask = inlier_mask ? st.out(inlier_mask, P) : nullptr;
    int32_t* dinfo = trial_info ? st.out(trial_info, nb * 2) : nullptr;
    int rc = st.upload();
    if (rc) return rc;
    GSF_HIP(hipMemcpyAsync(dstate, dst_in, nb * 625 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    rc = gsf_run_fusion_batch_dev(ctx, dts, dpos, dquat, B, N, dgt, dllh, doff, total, (int32_t)max_fixes, cfg, dstate, dR, dt, ds, dpo, dqo, dstat, dni, dzone, dsouth,
                                  dutm, dkeep, dal, dva, dsp, derr, drs, dmask, dinfo);
    if (rc) return rc;
    return st.finish();
}
