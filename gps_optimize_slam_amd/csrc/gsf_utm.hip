// gsf_utm.hip -- K1: batched WGS84 <-> UTM (replaces the pyproj/PROJ calls of load_gps_data,
// EKFGPSSLAM.py:266-271, and utm_to_wgs84, :291-296) plus the zone pick of auto_utm_projection (:127-134).
//
// One lane per point, SoA lat[]/lon[] so a wave reads two contiguous 512-B rows and writes two.
// This kernel is FP64-VALU-bound, not HBM-bound (profiles/r02_pmc_aux.txt: ~420 VALU instructions per point, 82 % of the issue
// slots of the launch, against 32 B of traffic): three libm-class calls are left on the forward side (sincos of the latitude,
// sincos of the longitude difference, one atan2), everything else is short series and angle-addition recurrences
// (gsf_math.hpp: utm_forward_point, tm_series).
#include "gsf_internal.hpp"

using namespace gsf;

namespace {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// mean(lon), mean(lat) per trajectory -> zone / hemisphere (ref :131-133)
__global__ __launch_bounds__(256) void utm_zone_kernel(const double* __restrict__ lat, const double* __restrict__ lon,
                                                       const int64_t* __restrict__ offsets, int32_t* __restrict__ zone,
                                                       int32_t* __restrict__ south)
{
    __shared__ double sh[2][4];
    const int64_t b = blockIdx.x;
    const int64_t i0 = offsets[b], i1 = offsets[b + 1];
    double sl = 0.0, sp = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) { sl += lon[i]; sp += lat[i]; }
    sl = wave_sum(sl); sp = wave_sum(sp);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = sl; sh[1][threadIdx.x >> 6] = sp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = (double)(i1 - i0);
        const double ml = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / n, mp = (sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) / n;
        zone[b] = (i1 > i0) ? (int32_t)(floor((ml + 180.0) / 6.0) + 1.0) : 0;      // int((mean+180)//6+1)
        south[b] = (i1 > i0 && mp < 0.0) ? 1 : 0;
    }
}

#ifndef GSF_UTM_OCC
#define GSF_UTM_OCC 3
#endif
// (three waves per SIMD: 168 registers.  Left alone the compiler takes 170 for the geodesy slice -- allocated as 176, i.e. two waves)
template <bool INVERSE>
__global__ __launch_bounds__(256, GSF_UTM_OCC) void utm_kernel(const double* __restrict__ a, const double* __restrict__ bb,
                                                  const int64_t* __restrict__ offsets, const int32_t* __restrict__ zone,
                                                  const int32_t* __restrict__ south, double* __restrict__ o1, double* __restrict__ o2)
{
    const int64_t b = blockIdx.x;
    const int64_t i0 = offsets[b], i1 = offsets[b + 1];
    const double lon0 = 6.0 * (double)zone[b] - 183.0;
    const double fn = south[b] ? 10000000.0 : 0.0;
    const TmConsts c = tm_consts();
    for (int64_t i = i0 + blockIdx.y * blockDim.x + threadIdx.x; i < i1; i += (int64_t)blockDim.x * gridDim.y) {
        double r1, r2;
        if (!INVERSE) {
            const double la = a[i], lo = bb[i];
            // validity mask of ref :259 -- rows the reference drops come back as NaN
            const bool ok = (fabs(la) <= 90.0) && (fabs(lo) <= 180.0) && (la != 0.0) && (lo != 0.0);
            utm_forward_point(c, la, lo, lon0, fn, r1, r2);
            if (!ok) { r1 = NAN; r2 = NAN; }
        } else {
            utm_inverse_point(c, a[i], bb[i], lon0, fn, r1, r2);
        }
        o1[i] = r1; o2[i] = r2;
    }
}

// The geodesy slice of load_gps_data (ref :258-271) for B ragged GNSS logs in ONE launch, block per log: rows (lat, lon, alt) ->
// validity mask (:259-264; rows the reference DROPS come back as NaN rows -- a fixed-shape device array cannot shrink), zone and
// hemisphere from the means over the valid rows (:131-133), UTM forward, rows [E, N, alt] (:271).
__global__ __launch_bounds__(256, GSF_UTM_OCC) void gps_rows_to_utm_kernel(const double* __restrict__ llh, const int64_t* __restrict__ offsets,
                                                              double* __restrict__ enu, int32_t* __restrict__ zone, int32_t* __restrict__ south)
{
    __shared__ double sh[3][4];
    __shared__ double sh_lon0, sh_fn;
    const int64_t b = blockIdx.x;
    const int64_t i0 = offsets[b], i1 = offsets[b + 1];
    double sl = 0.0, sp = 0.0, cnt = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const double la = llh[i * 3], lo = llh[i * 3 + 1];
        const bool ok = (fabs(la) <= 90.0) && (fabs(lo) <= 180.0) && (la != 0.0) && (lo != 0.0);
        if (ok) { sl += lo; sp += la; cnt += 1.0; }
    }
    sl = wave_sum(sl); sp = wave_sum(sp); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = sl; sh[1][threadIdx.x >> 6] = sp; sh[2][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
        const double ml = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / n, mp = (sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) / n;
        const int32_t z = (n > 0.0) ? (int32_t)(floor((ml + 180.0) / 6.0) + 1.0) : 0;
        const int32_t so = (n > 0.0 && mp < 0.0) ? 1 : 0;
        zone[b] = z; south[b] = so;
        sh_lon0 = 6.0 * (double)z - 183.0; sh_fn = so ? 10000000.0 : 0.0;
    }
    __syncthreads();
    const double lon0 = sh_lon0, fn = sh_fn;
    const TmConsts c = tm_consts();
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const double la = llh[i * 3], lo = llh[i * 3 + 1], al = llh[i * 3 + 2];
        const bool ok = (fabs(la) <= 90.0) && (fabs(lo) <= 180.0) && (la != 0.0) && (lo != 0.0);
        double e, n;
        utm_forward_point(c, la, lo, lon0, fn, e, n);
        enu[i * 3] = ok ? e : NAN; enu[i * 3 + 1] = ok ? n : NAN; enu[i * 3 + 2] = ok ? al : NAN;
    }
}

// geodetic -> ENU about a per-trajectory origin (ref_llh[b] = lat0, lon0, h0), lane per point.
// Measured on 1e8 points in tracks of 1 000 (round 4, same box, gpurun_out/r4z): the origin's frame costs a wave as much arithmetic as a
// trip of 64 points, and with 256 lanes per track a wave only makes four trips -- the frame formed from TWO sincos (it was six libm calls:
// the ECEF of the origin and its four sines / cosines separately) 1.13 -> 1.01 ms; 128 lanes per block (eight trips per frame) 0.94 ms,
// 64 lanes 1.03 (too few waves per track in flight); several points per lane and trip with the loads up front is SLOWER (2: 1.03, 4: 1.10 ms
// at 256 lanes -- 170 registers, three waves per SIMD instead of eight): the kernel lives on occupancy, not on ILP.
#ifndef GSF_ENU_U
#define GSF_ENU_U 1
#endif
constexpr int ENU_U = GSF_ENU_U, ENU_BLOCK = 128;
__global__ __launch_bounds__(256) void enu_kernel(const double* __restrict__ lat, const double* __restrict__ lon, const double* __restrict__ alt,
                                                  const int64_t* __restrict__ offsets, const double* __restrict__ ref_llh,
                                                  double* __restrict__ e, double* __restrict__ n, double* __restrict__ u)
{
    const int64_t b = blockIdx.x;
    const int64_t i0 = offsets[b], i1 = offsets[b + 1];
    const double la0 = ref_llh[b * 3], lo0 = ref_llh[b * 3 + 1], h0 = ref_llh[b * 3 + 2];
    if (i1 <= i0) return;
    const int64_t lanes = (int64_t)blockDim.x * gridDim.y;
    int64_t i = i0 + (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    double la[ENU_U], lo[ENU_U], al[ENU_U];
#pragma unroll
    for (int k = 0; k < ENU_U; ++k) { const int64_t j = i + k * lanes, jc = j < i1 ? j : i1 - 1; la[k] = lat[jc]; lo[k] = lon[jc]; al[k] = alt[jc]; }
    const EnuFrame f = enu_frame(la0, lo0, h0);
    for (; i < i1; i += lanes * ENU_U) {
        double ee[ENU_U], nn[ENU_U], uu[ENU_U];
#pragma unroll
        for (int k = 0; k < ENU_U; ++k) geodetic_to_enu_point(f, la[k], lo[k], al[k], ee[k], nn[k], uu[k]);
        const int64_t inext = i + lanes * ENU_U;
        if (inext < i1) {                                                 // the next trip's rows, requested before this trip's stores
#pragma unroll
            for (int k = 0; k < ENU_U; ++k) { const int64_t j = inext + k * lanes, jc = j < i1 ? j : i1 - 1; la[k] = lat[jc]; lo[k] = lon[jc]; al[k] = alt[jc]; }
        }
#pragma unroll
        for (int k = 0; k < ENU_U; ++k) { const int64_t j = i + k * lanes; if (j < i1) { e[j] = ee[k]; n[j] = nn[k]; u[j] = uu[k]; } }
    }
}

}  // namespace

extern "C" {

int gsf_utm_zone_batch_dev(gsf_ctx* ctx, const double* lat, const double* lon, const int64_t* offsets, int64_t B, int32_t* zone,
                           int32_t* south)
{
    GSF_REQUIRE(ctx && offsets && zone && south, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(utm_zone_kernel, dim3((unsigned)B), dim3(256), 0, ctx->stream, lat, lon, offsets, zone, south);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

static unsigned blocks_per_traj(int64_t B) { return B >= 2048 ? 1u : (B >= 256 ? 4u : 64u); }

int gsf_utm_forward_batch_dev(gsf_ctx* ctx, const double* lat, const double* lon, const int64_t* offsets, const int32_t* zone,
                              const int32_t* south, int64_t B, double* easting, double* northing)
{
    GSF_REQUIRE(ctx && offsets && zone && south, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(utm_kernel<false>, dim3((unsigned)B, blocks_per_traj(B)), dim3(256), 0, ctx->stream, lat, lon, offsets, zone, south, easting, northing);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_utm_inverse_batch_dev(gsf_ctx* ctx, const double* easting, const double* northing, const int64_t* offsets, const int32_t* zone,
                              const int32_t* south, int64_t B, double* lat, double* lon)
{
    GSF_REQUIRE(ctx && offsets && zone && south, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(utm_kernel<true>, dim3((unsigned)B, blocks_per_traj(B)), dim3(256), 0, ctx->stream, easting, northing, offsets, zone, south, lat, lon);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_gps_rows_to_utm_batch_dev(gsf_ctx* ctx, const double* llh, const int64_t* offsets, int64_t B, double* utm_rows, int32_t* zone, int32_t* south)
{
    GSF_REQUIRE(ctx && offsets && zone && south, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    if (B == 0) return GSF_OK;
    GSF_REQUIRE(llh && utm_rows, "NULL rows");
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(gps_rows_to_utm_kernel, dim3((unsigned)B), dim3(256), 0, ctx->stream, llh, offsets, utm_rows, zone, south);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

int gsf_geodetic_to_enu_batch_dev(gsf_ctx* ctx, const double* lat, const double* lon, const double* alt, const int64_t* offsets,
                                  const double* ref_llh, int64_t B, double* east, double* north, double* up)
{
    GSF_REQUIRE(ctx && offsets && ref_llh, "NULL argument");
    GSF_REQUIRE(B >= 0 && B <= 0x7fffffff, "bad B");
    if (B == 0) return GSF_OK;
    GSF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(enu_kernel, dim3((unsigned)B, blocks_per_traj(B)), dim3(ENU_BLOCK), 0, ctx->stream, lat, lon, alt, offsets, ref_llh, east, north, up);
    GSF_HIP(hipGetLastError());
    return GSF_OK;
}

}  // extern "C"
