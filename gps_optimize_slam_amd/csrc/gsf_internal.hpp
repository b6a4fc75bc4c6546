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
    // scratch for the fused pipeline (R,t,s,status per trajectory + init poses), grown on demand
    void* scratch;
    size_t scratch_bytes;
    int ekf_variant;       // tuning knob (gsf_set_option "ekf_variant")
    int wave_ppl;          // poses per lane of the wave-per-trajectory kernels (gsf_set_option "wave_ppl"; 0 = automatic)
    int duo_kernel;        // two-wave pipeline kernel for small batches (gsf_set_option "duo_kernel"): -1 automatic, 0 never, 1 always
    int seg_kernel;        // single-shot kernel for short tracks (gsf_set_option "seg_kernel"): 1 = whenever N fits, otherwise never (opt-in)
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

int ensure_scratch(gsf_ctx* ctx, size_t bytes);

// wave-per-trajectory K4 / fused pipeline for the trajectory-major layout (gsf_ekf_wave.hip)
int launch_ekf_wave(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                    const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                    int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status,
                    const int64_t* offsets = nullptr);

// single-shot kernel for short tracks: the whole trajectory in one wave pass, ceil(N/64) consecutive poses per lane
// (gsf_ekf_seg.hip), equal-length batches with N <= 64 * SEG_MAX_P
constexpr int SEG_MAX_P = 5;
int launch_ekf_seg(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                   const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                   int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status);

// wave-per-chunk / block-per-trajectory variant for small batches of short tracks (gsf_ekf_block.hip), N <= 1024
int launch_ekf_block(gsf_ctx* ctx, bool pipeline, const double* ts, const double* pos, const double* quat, const double* gps,
                     const uint8_t* valid, const double* init_pos, const double* init_quat, const gsf_ekf_config* cfg, int64_t B,
                     int64_t N, double* R, double* t, double* s, double* pos_out, double* quat_out, int32_t* status);
// trajectory-major dispatch.  Measured on MI355X: the chunk-parallel block kernel is SLOWER than the serial
// wave kernel even at C2 (1 000 x 271: 29 us vs 22 us for K4) -- 1 000 waves already occupy all 1 024 SIMDs and the serial kernel
// is ~60 % issue-bound, so extra waves only add carry-composition / barrier work.  It is therefore opt-in (ekf_variant 8 / 6).
inline bool use_block_kernel(const gsf_ctx* ctx, int64_t B, int64_t N)
{
    (void)B;
    return N <= 1024 && (ctx->ekf_variant == 8 || ctx->ekf_variant == 6);
}

}  // namespace gsf
