// gsf_util.hip -- layout transposes and the deterministic KITTI-04-shaped synthetic workload
//  used by bench.py and the GPU parity tests.  Not part of the reference's surface.
#include "gsf_internal.hpp"

using namespace gsf;

namespace {

// [B][N][C] -> [N][C][B] through a 64x64 LDS tile per (component) so both sides stay coalesced.
template <typename T, bool TO_TIME>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t B, int64_t N, int C)
{
    __shared__ T tile[64][65];
    const int64_t b0 = (int64_t)blockIdx.x * 64, i0 = (int64_t)blockIdx.y * 64;
    const int c = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 x 4
    if (TO_TIME) {
        // read rows b (fixed b, varying i): element (b, i, c) at (b*N + i)*C + c
        for (int r = ty; r < 64; r += 4) {
            const int64_t b = b0 + r, i = i0 + tx;
            if (b < B && i < N) tile[r][tx] = src[(b * N + i) * C + c];
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {
            const int64_t i = i0 + r, b = b0 + tx;
            if (b < B && i < N) dst[(i * C + c) * B + b] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 64; r += 4) {
            const int64_t i = i0 + r, b = b0 + tx;
            if (b < B && i < N) tile[r][tx] = src[(i * C + c) * B + b];
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {
            const int64_t b = b0 + r, i = i0 + tx;
            if (b < B && i < N) dst[(b * N + i) * C + c] = tile[tx][r];
        }
    }
}

// Several arrays of one batch in ONE launch (the time-major route through the wave kernel transposes five inputs and two outputs; at
// small batches each extra launch costs more than the copy itself).  blockIdx.z walks the components of all arrays: segment k holds
// comp[k] .. comp[k+1]-1.  8-byte elements, except a segment flagged as bytes.
struct TransposeSet { const void* src[5]; void* dst[5]; int C[5]; int comp0[6]; int bytes[5]; int n; };
template <bool TO_TIME>
__global__ __launch_bounds__(256) void transpose_set_kernel(TransposeSet ts, int64_t B, int64_t N)
{
    __shared__ double tile[64][65];
    int k = 0;
#pragma unroll
    for (int j = 1; j < 5; ++j) if (j < ts.n && (int)blockIdx.z >= ts.comp0[j]) k = j;
    const int c = (int)blockIdx.z - ts.comp0[k], C = ts.C[k];
    const int64_t b0 = (int64_t)blockIdx.x * 64, i0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const bool by = ts.bytes[k] != 0;
    const double* sd = (const double*)ts.src[k]; double* dd = (double*)ts.dst[k];
    const uint8_t* sb = (const uint8_t*)ts.src[k]; uint8_t* db = (uint8_t*)ts.dst[k];
    for (int r = ty; r < 64; r += 4) {
        const int64_t b = TO_TIME ? b0 + r : b0 + tx, i = TO_TIME ? i0 + tx : i0 + r;
        if (b < B && i < N) {
            const int64_t at = TO_TIME ? (b * N + i) * C + c : (i * C + c) * B + b;
            tile[r][tx] = by ? (double)sb[at] : sd[at];
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t i = TO_TIME ? i0 + r : i0 + tx, b = TO_TIME ? b0 + tx : b0 + r;
        if (b < B && i < N) {
            const int64_t at = TO_TIME ? (i * C + c) * B + b : (b * N + i) * C + c;
            if (by) db[at] = (uint8_t)tile[tx][r]; else dd[at] = tile[tx][r];
        }
    }
}

template <bool TO_TIME>
int launch_transpose(gsf_ctx* ctx, const void* src, void* dst, int64_t B, int64_t N, int32_t C, int32_t elem_bytes)
{
    GSF_REQUIRE(ctx && src && dst, "NULL argument");
    GSF_REQUIRE(B >= 0 && N >= 0 && C >= 1 && C <= 8, "bad shape");
    GSF_REQUIRE(elem_bytes == 8 || elem_bytes == 1, "elem_bytes must be 8 or 1");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_REQUIRE((N + 63) / 64 <= 65535, "N too large");
    GSF_HIP(hipSetDevice(ctx->device));
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)((N + 63) / 64), (unsigned)C), block(256);
    if (elem_bytes == 8)
        hipLaunchKernelGGL((transpose_kernel<double, TO_TIME>), grid, block, 0, ctx->stream, (const double*)src, (double*)dst, B, N, (int)C);
    else
        hipLaunchKernelGGL((transpose_kernel<uint8_t, TO_TIME>), grid, block, 0, ctx->stream, (const uint8_t*)src, (uint8_t*)dst, B, N, (int)C);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // namespace
namespace gsf {
// n arrays ([B][N][C_k] <-> [N][C_k][B]) in one launch; elem_bytes[k] is 8 or 1
int launch_transpose_set(gsf_ctx* ctx, bool to_time, int n, const void* const* src, void* const* dst, const int* C, const int* elem_bytes, int64_t B, int64_t N)
{
    if (B == 0 || N == 0 || n == 0) return GSF_OK;
    GSF_REQUIRE(n >= 1 && n <= 5 && (N + 63) / 64 <= 65535, "bad transpose set");
    TransposeSet ts{};
    ts.n = n; ts.comp0[0] = 0;
    for (int k = 0; k < n; ++k) { ts.src[k] = src[k]; ts.dst[k] = dst[k]; ts.C[k] = C[k]; ts.bytes[k] = elem_bytes[k] == 1; ts.comp0[k + 1] = ts.comp0[k] + C[k]; }
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)((N + 63) / 64), (unsigned)ts.comp0[n]), block(256);
    if (to_time) hipLaunchKernelGGL(transpose_set_kernel<true>, grid, block, 0, ctx->stream, ts, B, N);
    else hipLaunchKernelGGL(transpose_set_kernel<false>, grid, block, 0, ctx->stream, ts, B, N);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}
}  // namespace gsf
namespace {

// ---- counter-based RNG: integer hash -> uniform double; no libm, so values do not depend on a math library
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ double u01(uint64_t seed, uint64_t j, uint64_t i, uint64_t k)
{
    uint64_t h = mix64(mix64(mix64(seed ^ (j * 0xD1342543DE82EF95ull)) + i) + k * 0x2545F4914F6CDD1Dull);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double uni(double a, double b, uint64_t seed, uint64_t j, uint64_t i, uint64_t k) { return a + (b - a) * u01(seed, j, i, k); }
// Irwin-Hall(4) stand-in for a unit normal
__device__ __forceinline__ double nrm(uint64_t seed, uint64_t j, uint64_t i, uint64_t k)
{
    return (u01(seed, j, i, 4 * k) + u01(seed, j, i, 4 * k + 1) + u01(seed, j, i, 4 * k + 2) + u01(seed, j, i, 4 * k + 3) - 2.0) * 1.7320508075688772;
}

constexpr uint64_t TRAJ = 0xFFFFFFFFull;   // "step" slot used for per-trajectory draws

// variant 0: white 2 cm noise on the SLAM positions, the sharp-turn burst on a quarter of the mid-track outages (2.5 % of the tracks);
// variant 1 (SURVEY 8d to the letter): the SLAM error is a random-walk DRIFT of 2 cm per pose, and 5 % of ALL tracks get the burst
// (inside the outage where there is one, else from mid-track on)
template <int LAYOUT>
__global__ __launch_bounds__(64) void synth_kernel(int variant, uint64_t seed, int64_t traj0, int64_t B, int64_t N, double* __restrict__ ts,
                                                   double* __restrict__ pos, double* __restrict__ quat, double* __restrict__ gps,
                                                   uint8_t* __restrict__ valid, double* __restrict__ init_pos, double* __restrict__ init_quat)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Idx<LAYOUT> ix{ B, N };
    const uint64_t j = (uint64_t)(traj0 + b);
    // per-trajectory constants
    const double scale = uni(0.9, 1.1, seed, j, TRAJ, 0);
    Quat qg; quat_unit(Quat{ uni(-1, 1, seed, j, TRAJ, 1), uni(-1, 1, seed, j, TRAJ, 2), uni(-1, 1, seed, j, TRAJ, 3), uni(0.2, 1, seed, j, TRAJ, 4) }, qg);
    const Vec3 tg{ 458000.0 + uni(-2000, 2000, seed, j, TRAJ, 5), 5430000.0 + uni(-2000, 2000, seed, j, TRAJ, 6), 112.0 + uni(-5, 5, seed, j, TRAJ, 7) };
    const double a1 = uni(-0.15, 0.15, seed, j, TRAJ, 8), a2 = uni(-0.15, 0.15, seed, j, TRAJ, 9);
    const double T = 0.10411 * (double)(N > 1 ? N - 1 : 1);
    // outage plan (10 % mid outage of 5.5-12 s, 2 % start in outage, 2 % end in outage)
    const double kind = u01(seed, j, TRAJ, 10);
    int64_t o0 = -1, o1 = -1; bool burst = false; int64_t bstart = 0;
    if (kind < 0.10) {
        int64_t L = 53 + (int64_t)(u01(seed, j, TRAJ, 11) * 63.0);
        if (L > N / 3) L = N / 3;
        const int64_t room = N - L - 40;
        o0 = room > 0 ? 20 + (int64_t)(u01(seed, j, TRAJ, 12) * (double)room) : N / 3;
        o1 = o0 + L;
        burst = u01(seed, j, TRAJ, 13) < 0.25; bstart = o0 + L / 4;
    } else if (kind < 0.12) { o0 = 0; o1 = 10 + (int64_t)(u01(seed, j, TRAJ, 11) * 31.0); if (o1 > N / 2) o1 = N / 2; }
    else if (kind < 0.14) { int64_t K = 10 + (int64_t)(u01(seed, j, TRAJ, 11) * 31.0); if (K > N / 2) K = N / 2; o0 = N - K; o1 = N; }
    if (variant == 1) {
        burst = u01(seed, j, TRAJ, 13) < 0.05;
        if (!(kind < 0.10)) bstart = N / 2;
    }
    double t_prev = 0.0, t_burst = 0.0;
    Vec3 p{ 0.0, 0.0, 0.0 }, drift{ 0.0, 0.0, 0.0 };
    for (int64_t i = 0; i < N; ++i) {
        const double t = (i == 0) ? 0.0 : (double)i * 0.10411 + uni(-0.002, 0.002, seed, j, (uint64_t)i, 0);
        const double tau = t / T;
        const double u = a1 * tau + a2 * tau * tau;                     // tan(heading/2), rational unit quaternion
        double v = 0.0;                                                 // tan(roll/2) about z: what the reference's gate sees
        if (burst && i >= bstart) { if (i == bstart) t_burst = t; v = fmin(0.6 * (t - t_burst), 0.6); }
        const double ru = 1.0 / (1.0 + u * u);
        if (i > 0) {
            const double step = 14.0 * (t - t_prev);
            p.x += step * (2.0 * u * ru); p.y += step * 0.002; p.z += step * ((1.0 - u * u) * ru);
        }
        t_prev = t;
        Quat qy, qz; quat_unit(Quat{ 0.0, u, 0.0, 1.0 }, qy); quat_unit(Quat{ 0.0, 0.0, v, 1.0 }, qz);
        const Quat q = quat_mul(qz, qy);
        Vec3 e{ 0.02 * nrm(seed, j, (uint64_t)i, 1), 0.02 * nrm(seed, j, (uint64_t)i, 2), 0.02 * nrm(seed, j, (uint64_t)i, 3) };
        if (variant == 1) { drift.x += e.x; drift.y += e.y; drift.z += e.z; e = drift; }
        const Vec3 ps{ p.x / scale + e.x, p.y / scale + e.y, p.z / scale + e.z };
        const Vec3 rg = quat_rotate(qg, p);
        const bool ok = !(i >= o0 && i < o1);
        const Vec3 z{ ok ? rg.x + tg.x + 0.45 * nrm(seed, j, (uint64_t)i, 4) : NAN, ok ? rg.y + tg.y + 0.45 * nrm(seed, j, (uint64_t)i, 5) : NAN,
                      ok ? rg.z + tg.z + 0.45 * nrm(seed, j, (uint64_t)i, 6) : NAN };
        ts[ix.at(b, i, 0, 1)] = t;
        pos[ix.at(b, i, 0, 3)] = ps.x; pos[ix.at(b, i, 1, 3)] = ps.y; pos[ix.at(b, i, 2, 3)] = ps.z;
        quat[ix.at(b, i, 0, 4)] = q.x; quat[ix.at(b, i, 1, 4)] = q.y; quat[ix.at(b, i, 2, 4)] = q.z; quat[ix.at(b, i, 3, 4)] = q.w;
        gps[ix.at(b, i, 0, 3)] = z.x; gps[ix.at(b, i, 1, 3)] = z.y; gps[ix.at(b, i, 2, 3)] = z.z;
        valid[ix.at(b, i, 0, 1)] = ok ? 1 : 0;
        if (i == 0 && init_pos && init_quat) {                          // planted Sim3 of pose 0
            const Vec3 r0 = quat_rotate(qg, ps);
            init_pos[b * 3] = scale * r0.x + tg.x; init_pos[b * 3 + 1] = scale * r0.y + tg.y; init_pos[b * 3 + 2] = scale * r0.z + tg.z;
            const Quat q0 = quat_mul(qg, q);
            init_quat[b * 4] = q0.x; init_quat[b * 4 + 1] = q0.y; init_quat[b * 4 + 2] = q0.z; init_quat[b * 4 + 3] = q0.w;
        }
    }
}

// The same trajectories with the GNSS side as a RAGGED geodetic log (what load_gps_data reads, ref :258): fix k of trajectory j is
// taken at the SLAM stamp + a per-trajectory receiver phase in (0, 0.03) s on the true path (linear between the neighbouring
// poses), mapped to (lat, lon, alt) about (49.0336 N, 8.3950 E, 112 m) by the tangent-plane metric at that latitude -- two
// constants, no libm, so a host generator can reproduce the values -- plus the same 0.45 m noise; fixes inside the outage plan are
// ABSENT (the log just has a gap > max_gps_gap_threshold), the planted UTM frame is replaced by true geography.
// counts[j] fixes per trajectory (pass 1, WRITE = false), rows at gps_offsets[j] (pass 2).
constexpr double SYN_DEG_PER_M_NORTH = 8.991965674724778e-06, SYN_DEG_PER_M_EAST = 1.3675669825622133e-05;
template <bool WRITE>
__global__ __launch_bounds__(64) void synth_geodetic_kernel(uint64_t seed, int64_t traj0, int64_t B, int64_t N, double* __restrict__ ts,
                                                            double* __restrict__ pos, double* __restrict__ quat, int64_t* __restrict__ counts,
                                                            const int64_t* __restrict__ gps_offsets, double* __restrict__ gps_t,
                                                            double* __restrict__ gps_llh)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Idx<GSF_LAYOUT_TRAJ_MAJOR> ix{ B, N };
    const uint64_t j = (uint64_t)(traj0 + b);
    const double scale = uni(0.9, 1.1, seed, j, TRAJ, 0);
    const double heading = uni(-1.0, 1.0, seed, j, TRAJ, 1);            // tan(heading/2) of the track's course over ground
    const double ch = (1.0 - heading * heading) / (1.0 + heading * heading), sh = 2.0 * heading / (1.0 + heading * heading);
    const double a1 = uni(-0.15, 0.15, seed, j, TRAJ, 8), a2 = uni(-0.15, 0.15, seed, j, TRAJ, 9);
    const double phase = uni(0.004, 0.03, seed, j, TRAJ, 14);
    const double T = 0.10411 * (double)(N > 1 ? N - 1 : 1);
    const double kind = u01(seed, j, TRAJ, 10);
    int64_t o0 = -1, o1 = -1;
    if (kind < 0.10) {
        int64_t L = 53 + (int64_t)(u01(seed, j, TRAJ, 11) * 63.0);
        if (L > N / 3) L = N / 3;
        const int64_t room = N - L - 40;
        o0 = room > 0 ? 20 + (int64_t)(u01(seed, j, TRAJ, 12) * (double)room) : N / 3;
        o1 = o0 + L;
    } else if (kind < 0.12) { o0 = 0; o1 = 10 + (int64_t)(u01(seed, j, TRAJ, 11) * 31.0); if (o1 > N / 2) o1 = N / 2; }
    else if (kind < 0.14) { int64_t K = 10 + (int64_t)(u01(seed, j, TRAJ, 11) * 31.0); if (K > N / 2) K = N / 2; o0 = N - K; o1 = N; }
    if (!WRITE) { counts[b] = (N - 1) - ((o1 > o0) ? ((o1 < N - 1 ? o1 : N - 1) - o0) : 0); return; }
    int64_t w = gps_offsets[b];
    double t_prev = 0.0;
    Vec3 p{ 0.0, 0.0, 0.0 }, p_prev{ 0.0, 0.0, 0.0 };
    for (int64_t i = 0; i < N; ++i) {
        const double t = (i == 0) ? 0.0 : (double)i * 0.10411 + uni(-0.002, 0.002, seed, j, (uint64_t)i, 0);
        const double tau = t / T;
        const double u = a1 * tau + a2 * tau * tau;
        const double ru = 1.0 / (1.0 + u * u);
        p_prev = p;
        if (i > 0) {
            const double step = 14.0 * (t - t_prev);
            p.x += step * (2.0 * u * ru); p.y += step * 0.002; p.z += step * ((1.0 - u * u) * ru);
        }
        Quat qy; quat_unit(Quat{ 0.0, u, 0.0, 1.0 }, qy);
        const Vec3 ps{ p.x / scale + 0.02 * nrm(seed, j, (uint64_t)i, 1), p.y / scale + 0.02 * nrm(seed, j, (uint64_t)i, 2),
                       p.z / scale + 0.02 * nrm(seed, j, (uint64_t)i, 3) };
        ts[ix.at(b, i, 0, 1)] = t;
        pos[ix.at(b, i, 0, 3)] = ps.x; pos[ix.at(b, i, 1, 3)] = ps.y; pos[ix.at(b, i, 2, 3)] = ps.z;
        quat[ix.at(b, i, 0, 4)] = qy.x; quat[ix.at(b, i, 1, 4)] = qy.y; quat[ix.at(b, i, 2, 4)] = qy.z; quat[ix.at(b, i, 3, 4)] = qy.w;
        // fix i-1 sits between poses i-1 and i (camera axes: x right, y down, z forward -> east/north by the course, up = -y)
        if (i > 0 && !(i - 1 >= o0 && i - 1 < o1)) {
            const double f = phase / (t - t_prev);
            const Vec3 g{ p_prev.x + f * (p.x - p_prev.x), p_prev.y + f * (p.y - p_prev.y), p_prev.z + f * (p.z - p_prev.z) };
            const double east = ch * g.x + sh * g.z + 0.45 * nrm(seed, j, (uint64_t)i, 4), north = -sh * g.x + ch * g.z + 0.45 * nrm(seed, j, (uint64_t)i, 5);
            gps_t[w] = t_prev + phase;
            gps_llh[w * 3] = 49.0336 + north * SYN_DEG_PER_M_NORTH; gps_llh[w * 3 + 1] = 8.3950 + east * SYN_DEG_PER_M_EAST;
            gps_llh[w * 3 + 2] = 112.0 - g.y + 0.45 * nrm(seed, j, (uint64_t)i, 6);
            ++w;
        }
        t_prev = t;
    }
}

}  // namespace

extern "C" {

int gsf_transpose_to_time_major_dev(gsf_ctx* ctx, const void* src, void* dst, int64_t B, int64_t N, int32_t C, int32_t elem_bytes)
{
    return launch_transpose<true>(ctx, src, dst, B, N, C, elem_bytes);
}
int gsf_transpose_to_traj_major_dev(gsf_ctx* ctx, const void* src, void* dst, int64_t B, int64_t N, int32_t C, int32_t elem_bytes)
{
    return launch_transpose<false>(ctx, src, dst, B, N, C, elem_bytes);
}

int gsf_synth_geodetic_batch_dev(gsf_ctx* ctx, uint64_t seed, int64_t traj0, int64_t B, int64_t N, double* ts, double* pos, double* quat,
                                 int64_t* counts, const int64_t* gps_offsets, double* gps_t, double* gps_llh)
{
    GSF_REQUIRE(ctx && B >= 0 && N >= 0 && traj0 >= 0, "bad shape");
    GSF_REQUIRE((counts != nullptr) != (gps_offsets != nullptr), "pass counts (sizing pass) OR gps_offsets (writing pass)");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    const dim3 block(64), grid((unsigned)((B + 63) / 64));
    if (counts) {
        hipLaunchKernelGGL(synth_geodetic_kernel<false>, grid, block, 0, ctx->stream, seed, traj0, B, N, ts, pos, quat, counts, gps_offsets, gps_t, gps_llh);
    } else {
        GSF_REQUIRE(ts && pos && quat && gps_t && gps_llh, "NULL argument");
        hipLaunchKernelGGL(synth_geodetic_kernel<true>, grid, block, 0, ctx->stream, seed, traj0, B, N, ts, pos, quat, counts, gps_offsets, gps_t, gps_llh);
    }
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_synth_batch_dev(gsf_ctx* ctx, int32_t layout, uint64_t seed, int64_t traj0, int64_t B, int64_t N, double* ts, double* pos,
                        double* quat, double* gps, uint8_t* valid, double* init_pos, double* init_quat)
{
    GSF_REQUIRE(ctx && ts && pos && quat && gps && valid, "NULL argument");
    GSF_REQUIRE(B >= 0 && N >= 0 && traj0 >= 0, "bad shape");
    GSF_REQUIRE(layout == GSF_LAYOUT_TRAJ_MAJOR || layout == GSF_LAYOUT_TIME_MAJOR, "unknown layout");
    if (B == 0 || N == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    const dim3 block(64), grid((unsigned)((B + 63) / 64));
    if (layout == GSF_LAYOUT_TIME_MAJOR)
        hipLaunchKernelGGL(synth_kernel<GSF_LAYOUT_TIME_MAJOR>, grid, block, 0, ctx->stream, ctx->synth_variant, seed, traj0, B, N, ts, pos, quat, gps, valid, init_pos, init_quat);
    else
        hipLaunchKernelGGL(synth_kernel<GSF_LAYOUT_TRAJ_MAJOR>, grid, block, 0, ctx->stream, ctx->synth_variant, seed, traj0, B, N, ts, pos, quat, gps, valid, init_pos, init_quat);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // extern "C"
