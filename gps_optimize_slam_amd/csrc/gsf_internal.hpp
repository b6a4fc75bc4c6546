// gsf_internal.hpp -- context, error plumbing and layout indexing shared by the .hip units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gsf.h"
#include "gsf_ekf_core.hpp"

struct gsf_ctx {
    int device;
    hipStream_t stream;
    bool owns_stream;
    hipEvent_t ev0, ev1;
    // kernel workspace (time-alignment slabs, the transposed copy of a small time-major batch), grown on demand
    void* scratch;
    size_t scratch_bytes;
    // staging of the host-pointer entry points: one grow-only device arena + a pinned host mirror (see gsf::Staging)
    void* stage;
    size_t stage_bytes;
    void* pinned;
    size_t pinned_bytes;
    // tape, transition tables and swap partners of the chip-wide draws (gsf_rng_tape.hip); separate from `scratch`, whose layout the caller of
    // launch_mt_choice may hold pointers into
    void* small_scratch;   // 512 B: arg-max keys of the split K2b launch (gsf_sim3.hip: up to 32 sets x 16 B)
    void* k2b_scratch;     // K2b's rows as floats (screened residual counts, gsf_sim3.hip), grow-only
    size_t k2b_scratch_bytes;
    void* rng_scratch;
    size_t rng_scratch_bytes;
    void* run_scratch;     // temporaries of gsf_run_fusion_batch_dev (its stages use `scratch` themselves), grow-only
    size_t run_scratch_bytes;
    void* rows_scratch;    // row mask + flags of the Sim3 row choice for the kernels that take it as a pre-pass (gsf_ekf_block.hip), grow-only
    size_t rows_scratch_bytes;
    int k2b_screen;        // K2b residual counts screened in packed single precision, exact re-check in the band (gsf_set_option "k2b_screen"): 1 default, 0 all double
    int tape_draws;        // chip-wide draws for a few streams (gsf_set_option "tape_draws"): -1 automatic, 0 never, 2 tests (tape cut short)
    int ekf_variant;       // reserved tuning knob (gsf_set_option "ekf_variant"); 0 = default
    int synth_variant;     // synthetic workload (gsf_set_option "synth_variant"): 0 = white SLAM noise (default), 1 = SURVEY 8d's random-walk drift
    int block_kernel;      // workgroup-per-trajectory kernel for 64 < N <= 1024 (gsf_set_option "block_kernel"): -1 automatic, 0 never, 1 always
    int ransac_early_exit; // robust chain: stop a trajectory's trials at the first one that counts every row (gsf_set_option "ransac_early_exit"): 0 default, 1 on
    int prefilter_miss_batch; // ... and the largest batch the sequential walk opens with after the speculative pass has missed (gsf_set_option "prefilter_miss_batch")
    int prefilter_speculate;  // the pre-filter chain tries "every axis of the window stops after its first trial" first (gsf_set_option "prefilter_speculate")
    int prefilter_first_batch; // trials the pre-filter chain draws and scores before its first look at scikit-learn's stopping rule (gsf_set_option "prefilter_first_batch")
    int ransac_probe_trials; // ... trials the early-exit probe draws and scores itself before the wide kernels take the rest (gsf_set_option "ransac_probe_trials")
    int duo_kernel;        // two-wave pipeline kernel for small batches (gsf_set_option "duo_kernel"): -1 automatic, 0 never, 1 always
    gsf::FitRows fit_rows; // rows of the fused chains' Sim3 fit (gsf_set_sim3_rows); mode 0 = all valid rows
    int64_t lane_min_traj; // time-major batches with fewer trajectories are transposed and run by the wave kernel (gsf_set_option "lane_min_traj")
};

namespace gsf {

void set_error(const char* fmt, ...);
int fail_hip(hipError_t e, const char* what);

#define GSF_HIP(call)                                         \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return gsf::fail_hip(e__, #call); \
    } while (0)

#define GSF_REQUIRE(cond, msg)                                \
    do {                                                      \
        if (!(cond)) { gsf::set_error("%s: %s", __func__, msg); return GSF_ERR_INVALID_ARG; } \
    } while (0)

// Element index of component c (of C) of pose i of trajectory b.
template <int LAYOUT>
struct Idx {
    int64_t B, N;
    __host__ __device__ __forceinline__ int64_t at(int64_t b, int64_t i, int c, int C) const
    {
        if (LAYOUT == GSF_LAYOUT_TIME_MAJOR) return (i * C + c) * B + b;
        return (b * N + i) * C + c;
    }
};

// how each wave-level translation unit was compiled (scheduler, -ffp-contract mode, compiler): recorded by the Makefile's -D flags
#ifndef GSF_TU_SCHED
#define GSF_TU_SCHED "default"
#endif
#ifndef GSF_TU_CONTRACT
#define GSF_TU_CONTRACT "fast (hipcc default)"
#endif
#define GSF_TU_BUILD_INFO(file) file ": sched=" GSF_TU_SCHED ", fp-contract=" GSF_TU_CONTRACT ", clang " __clang_version__
const char* wave_small_build_info();
const char* wave_big_build_info();
const char* wave_block_build_info();

int ensure_scratch(gsf_ctx* ctx, size_t bytes);
int ensure_rng_scratch(gsf_ctx* ctx, size_t bytes);
int ensure_k2b_scratch(gsf_ctx* ctx, size_t bytes);
int ensure_rows_scratch(gsf_ctx* ctx, size_t bytes);
int ensure_run_scratch(gsf_ctx* ctx, size_t bytes);

// wave-per-trajectory K4 / fused pipeline for the trajectory-major layout (gsf_ekf_wave.hip)
int launch_ekf_wave(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                    const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                    int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status,
                    const int64_t* offsets = nullptr);

// the big-batch builds of the wave kernels (gsf_ekf_wave_big.hip): B > 2 048
int launch_ekf_wave_big(gsf_ctx* ctx, bool pipeline, bool xy, const double* ts, const double* pos, const double* quat, const double* gps,
                        const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B, int64_t N,
                        double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status, const int64_t* offsets);

// workgroup-per-trajectory K4 / fused pipeline (gsf_ekf_block.hip): one wave per 64-pose chunk, every input byte read once
bool ekf_block_applies(int64_t N, const int64_t* offsets);
int launch_ekf_block(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                     const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                     int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status);

// filter_gps_outliers_ransac as a whole, windows found on the device (gsf_gpsfilter.hip); counts (may be NULL): log b = rows offsets[b] .. +counts[b]
int launch_gps_prefilter_auto(gsf_ctx* ctx, const double* t, const double* pos, const int64_t* offsets, const int32_t* counts, int64_t B,
                              int32_t max_log_rows, const gsf_prefilter_config* f, uint32_t* mt_state, uint8_t* keep, int32_t* log_status, int32_t* log_info);

// step 6 for raw SLAM / Sim3 / EKF in one launch (gsf_eval.hip): stats[3][B][4], errors[3][B][N]
int launch_apply_sim3(gsf_ctx* ctx, const double* pos, const double* quat, const int64_t* offsets, int64_t B, const double* R, const double* t,
                      const double* s, double* pos_out, double* quat_out, int32_t* bad_quat, bool bad_quat_zeroed);
int launch_eval_errors3(gsf_ctx* ctx, const double* ts, const double* traj0, const double* traj1, const double* traj2, const double* aligned_gps,
                        const uint8_t* valid, int64_t B, int64_t N, double skip_seconds, double* stats, double* errors);

// main_process_gui's row choice as a launch of its own (gsf_robust.hip: sim3_rows_kernel; ref :973-998): row_mask[B*N] / ragged, n_rows[B], status[B]
int launch_sim3_rows(gsf_ctx* ctx, const double* ts, const double* gps, const uint8_t* valid, const int64_t* offsets, int64_t B, int64_t N,
                     const FitRows& rule, uint8_t* row_mask, int32_t* n_rows, int32_t* status);

// K2b launch with optional per-set row counts (sets in fixed-stride slots: rows offsets[b] .. offsets[b] + counts[b])
int launch_sim3_ransac(gsf_ctx* ctx, const double* src, const double* dst, const int64_t* offsets, const int32_t* counts, int64_t B,
                       const int32_t* sample_idx, int32_t trials, int32_t min_samples, double thr, int32_t min_inliers, double* R, double* t,
                       double* s, int32_t* status, uint8_t* inlier_mask, int32_t* n_inliers,
                       int64_t total_rows = 0 /* rows of src / dst if the host knows them: enables the single-precision screen */,
                       int32_t trial0 = 0 /* hypotheses below this one were scored by the early-exit probe (gsf_robust.hip) ... */,
                       unsigned long long* keys_io = nullptr /* ... whose arg-max key per set comes in here ([B][2]) and the merged one goes out */,
                       const int32_t* decided = nullptr /* ... and sets it decided (a trial counted every row, ref :413) are not scanned again */);
// sample sets of the reference's RNG call, generated on the device (gsf_rng.hip): permutation(n_b)[:k] per trial from each set's
// legacy MT19937 state; n_b = counts[b] (int32) -- asynchronous on the context's stream
int launch_mt_choice(gsf_ctx* ctx, uint32_t* state, const int32_t* counts, int64_t B, int32_t trials, int32_t k, int32_t* sample_idx,
                     int32_t n_max /* largest counts[b] if the host knows it, else 0 */);
int launch_mt_choice_rest(gsf_ctx* ctx, uint32_t* state, const int32_t* counts, int64_t B, int32_t total_trials, int32_t trial0, int32_t k,
                          int32_t* sample_idx, int32_t n_max, const int32_t* skip);
// the same draws for a few streams, spread over the chip (gsf_rng_tape.hip); launch_mt_choice picks it when mt_tape_applies
bool mt_tape_applies(const gsf_ctx* ctx, int64_t B, int32_t trials, int32_t k, int32_t n_max);
int launch_mt_tape(gsf_ctx* ctx, uint32_t* state, const int32_t* counts, int64_t B, int32_t trials, int32_t k, int32_t* sample_idx, int32_t n_max,
                   const int32_t** done_flags, int* done_stride);

// Staging of the host-pointer entry points.  ONE grow-only device arena per context plus a pinned host mirror of it: the
// inputs of a call are packed into the mirror and cross PCIe in one hipMemcpyAsync, the outputs come back in one, and no call
// pays hipMalloc/hipFree (which synchronise the device).  Calls whose arrays exceed PINNED_MAX copy straight from/to the caller's
// pageable arrays instead (a bulk transfer, where the extra host memcpy would cost more than the runtime's own staging).
// Order of use: every in() before the first out()/tmp(); upload(); launches; finish().
class Staging {
public:
    static constexpr size_t PINNED_MAX = (size_t)32 << 20;
    Staging(gsf_ctx* ctx, size_t payload_bytes, int n_arrays);
    int rc() const { return rc_; }
    template <class T> T* in(const T* host, size_t n) { return (T*)in_bytes(host, n * sizeof(T)); }
    template <class T> T* out(T* host, size_t n) { return (T*)out_bytes(host, n * sizeof(T)); }
    template <class T> T* tmp(size_t n) { return (T*)out_bytes(nullptr, n * sizeof(T)); }
    int upload();
    int finish();      // device -> caller arrays, then hipStreamSynchronize

private:
    struct Out { void* host; size_t off, bytes; };
    static constexpr int MAX_OUT = 24;
    void* take(size_t bytes, size_t& at);
    void* in_bytes(const void* host, size_t bytes);
    void* out_bytes(void* host, size_t bytes);
    gsf_ctx* ctx_; char* d_ = nullptr; char* h_ = nullptr; size_t cap_ = 0, off_ = 0, in_end_ = 0, out_lo_ = 0; bool direct_ = false, has_out_ = false;
    int rc_ = GSF_OK; Out outs_[MAX_OUT]; int n_out_ = 0;
};

int launch_transpose_set(gsf_ctx* ctx, bool to_time, int n, const void* const* src, void* const* dst, const int* C, const int* elem_bytes, int64_t B, int64_t N);
}  // namespace gsf
