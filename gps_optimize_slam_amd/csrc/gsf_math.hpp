// gsf_math.hpp -- FP64 building blocks of the GPS<->SLAM fusion kernels (gfx950).
//
// Everything here is branch-light inline code meant to live in VGPRs of one lane:
// scalar-last quaternions [x,y,z,w] with SciPy's conventions, a 3x3
// one-sided Jacobi SVD, the Umeyama closed form, and the Krueger transverse-Mercator
// series.  The same header is compiled by g++ into a test-only host harness
// (tests/host_harness.cpp) so the math can be checked against the oracle without a GPU;
// the shipped library only ever runs it on the device.
//
// Reference lines ("ref :N") are /root/reference/EKFGPSSLAM.py.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GSF_HD __host__ __device__ __forceinline__
#define GSF_HD_COLD __host__ __device__ inline __attribute__((noinline))   // rare paths: keep them out of the hot loop's register budget
#else
#define GSF_HD inline
#define GSF_HD_COLD inline
#endif

namespace gsf {

struct Quat { double x, y, z, w; };
struct Vec3 { double x, y, z; };

GSF_HD double sq(double a) { return a * a; }

// 1/a and 1/sqrt(a) for normal, positive-magnitude a.  On the device: the hardware seed (v_rcp_f64 / v_rsq_f64) plus two
// Newton steps in FMA form -- ~5 / ~8 instructions and <= 1 ulp, instead of the ~12 / ~25-instruction IEEE expansions with
// their div_scale/div_fixup denormal handling (the operands here are |q|^2 ~ 1 and variances ~ 0.1: never denormal).
GSF_HD double fast_rcp(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double x = __builtin_amdgcn_rcp(a);
    double e = fma(-a, x, 1.0); x = fma(x, e, x);
    e = fma(-a, x, 1.0); x = fma(x, e, x);
    return x;
#else
    return 1.0 / a;
#endif
}
GSF_HD double fast_rsqrt(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5); g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5); h = fma(h, r, h);
    return h + h;
#else
    return 1.0 / sqrt(a);
#endif
}

// the bare hardware seeds (~2^-26 relative): for factors whose accuracy only steers an iteration
GSF_HD double seed_rcp(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(a);
#else
    return 1.0 / a;
#endif
}
GSF_HD double seed_rsqrt(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(a);
#else
    return 1.0 / sqrt(a);
#endif
}

// Rotation.from_quat: q/|q|; false if the norm is 0/NaN/inf (SciPy raises ValueError).
// x^2 + y^2 + z^2 + w^2 as ONE chain (see quat_mul: a sum of products has no fixed rounding under -ffp-contract=fast)
GSF_HD double quat_norm2(const Quat& q) { return fma(q.w, q.w, fma(q.z, q.z, fma(q.y, q.y, q.x * q.x))); }
GSF_HD bool quat_unit(const Quat& q, Quat& o)
{
    double n2 = quat_norm2(q);
    bool ok = (n2 >= 1e-280) && (n2 <= 1e280);     // sqrt(n2) > 0 and finite; the margins keep fast_rsqrt in its normal range
    double r = fast_rsqrt(ok ? n2 : 1.0);
    o.x = q.x * r; o.y = q.y * r; o.z = q.z * r; o.w = q.w * r;
    return ok;
}
GSF_HD Quat quat_conj(const Quat& q) { return Quat{ -q.x, -q.y, -q.z, q.w }; }
// Rotation.__mul__ (Hamilton product p*q, "q first")
GSF_HD Quat quat_mul(const Quat& p, const Quat& q)
{
    // Written as explicit fma chains: under -ffp-contract=fast the compiler is free to fuse "a*b + c*d + ..." in ANY association, and it
    // picked different ones for the scalar part in two builds of the same kernel (the big-batch build that fetches rows through LDS vs
    // the small-batch one: 1 ulp in q.w on ~0.01 % of the rows, which broke the bit-equality of a batch and its shards).  With the chain
    // spelled out there is one result.
    Quat o;
    o.x = fma(p.w, q.x, fma(q.w, p.x, fma(p.y, q.z, -(p.z * q.y))));
    o.y = fma(p.w, q.y, fma(q.w, p.y, fma(p.z, q.x, -(p.x * q.z))));
    o.z = fma(p.w, q.z, fma(q.w, p.z, fma(p.x, q.y, -(p.y * q.x))));
    o.w = fma(p.w, q.w, -fma(p.z, q.z, fma(p.y, q.y, p.x * q.x)));
    return o;
}
// Rotation.apply for a unit quaternion: as_matrix() @ v
GSF_HD Vec3 quat_rotate(const Quat& q, const Vec3& v)
{
    // M(q) v for a UNIT q in the cross-product form v + w t + u x t, t = 2 u x v (18 flops instead of 39 for the matrix)
    // (explicit chains, as in quat_mul)
    const double tx = 2.0 * fma(q.y, v.z, -(q.z * v.y)), ty = 2.0 * fma(q.z, v.x, -(q.x * v.z)), tz = 2.0 * fma(q.x, v.y, -(q.y * v.x));
    Vec3 o;
    o.x = fma(q.w, tx, v.x) + fma(q.y, tz, -(q.z * ty));
    o.y = fma(q.w, ty, v.y) + fma(q.z, tx, -(q.x * tz));
    o.z = fma(q.w, tz, v.z) + fma(q.x, ty, -(q.y * tx));
    return o;
}
// as_matrix() of a unit quaternion (SciPy's form: m00 = x^2 - y^2 - z^2 + w^2, m10 = 2(xy + zw), ...).  Returned by value with
// named members: an array filled through a pointer ends up in scratch (see the compiler notes in DESIGN.md).
struct Mat3 { double m00, m01, m02, m10, m11, m12, m20, m21, m22; };
GSF_HD Mat3 quat_matrix(const Quat& q)
{
    const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, ww = q.w * q.w;
    const double xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z, xw = q.x * q.w, yw = q.y * q.w, zw = q.z * q.w;
    return Mat3{ xx - yy - zz + ww, 2.0 * (xy - zw), 2.0 * (xz + yw), 2.0 * (xy + zw), -xx + yy - zz + ww, 2.0 * (yz - xw),
                 2.0 * (xz - yw), 2.0 * (yz + xw), -xx - yy + zz + ww };
}
// as_euler('zyx')[0] of a unit quaternion: atan2(-m01, m00)
GSF_HD double quat_yaw_zyx(const Quat& q)
{
    double m00 = q.x * q.x - q.y * q.y - q.z * q.z + q.w * q.w;
    double m01 = 2.0 * (q.x * q.y - q.z * q.w);
    return atan2(-m01, m00);
}
// ExtendedKalmanFilter.normalize_quaternion, ref :697-700
GSF_HD Quat ekf_normalize(const Quat& q)
{
    double n2 = quat_norm2(q);
    bool ok = (n2 > 1e-18) && (n2 <= 1e280);       // norm > 1e-9 (ref :699); NaN/inf fall through to the identity like NaN > 1e-9
    double r = fast_rsqrt(ok ? n2 : 1.0);
    return ok ? Quat{ q.x * r, q.y * r, q.z * r, q.w * r } : Quat{ 0.0, 0.0, 0.0, 1.0 };
}
// quaternion_nlerp, ref :94-105
GSF_HD Quat quat_nlerp(const Quat& q1, Quat q2, double weight_q2)
{
    double dot = fma(q1.w, q2.w, fma(q1.z, q2.z, fma(q1.y, q2.y, q1.x * q2.x)));
    if (dot < 0.0) { q2.x = -q2.x; q2.y = -q2.y; q2.z = -q2.z; q2.w = -q2.w; }
    double w = fmin(fmax(weight_q2, 0.0), 1.0);
    Quat qi{ (1.0 - w) * q1.x + w * q2.x, (1.0 - w) * q1.y + w * q2.y, (1.0 - w) * q1.z + w * q2.z,
             (1.0 - w) * q1.w + w * q2.w };
    double n = sqrt(qi.x * qi.x + qi.y * qi.y + qi.z * qi.z + qi.w * qi.w);
    if (n < 1e-9) return (weight_q2 < 0.5) ? q1 : q2;
    double r = 1.0 / n;
    return Quat{ qi.x * r, qi.y * r, qi.z * r, qi.w * r };
}
// Rotation.from_matrix (largest of m00,m11,m22,trace), R row-major
GSF_HD Quat quat_from_matrix(const double* M)
{
    double tr = M[0] + M[4] + M[8];
    double q[4];
    int c = 0; double best = M[0];
    if (M[4] > best) { best = M[4]; c = 1; }
    if (M[8] > best) { best = M[8]; c = 2; }
    if (tr > best) { c = 3; }
    if (c == 0)      { q[0] = 1.0 - tr + 2.0 * M[0]; q[1] = M[3] + M[1]; q[2] = M[6] + M[2]; q[3] = M[7] - M[5]; }
    else if (c == 1) { q[1] = 1.0 - tr + 2.0 * M[4]; q[2] = M[7] + M[5]; q[0] = M[1] + M[3]; q[3] = M[2] - M[6]; }
    else if (c == 2) { q[2] = 1.0 - tr + 2.0 * M[8]; q[0] = M[2] + M[6]; q[1] = M[5] + M[7]; q[3] = M[3] - M[1]; }
    else             { q[0] = M[7] - M[5]; q[1] = M[2] - M[6]; q[2] = M[3] - M[1]; q[3] = 1.0 + tr; }
    double r = fast_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);   // >= 1 for a rotation matrix: never denormal
    return Quat{ q[0] * r, q[1] * r, q[2] * r, q[3] * r };
}

GSF_HD double det3(const double* M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// ---------------------------------------------------------------------------------------
// 3x3 SVD by one-sided (Hestenes) Jacobi: columns of A = H are rotated until mutually
// orthogonal; V accumulates the rotations.  Relative accuracy even for sigma1 >> sigma3
// (straight-road tracks), which an eigen-solve of H^T H would lose.  Outputs the pieces the
// Umeyama closed form needs: U, V (column-major access via idx), S sorted descending.
// ---------------------------------------------------------------------------------------
struct Svd3 { double U[9], V[9], S[3]; };   // row-major 3x3; columns are the singular vectors

GSF_HD void jacobi_pair(double* A, double* W, int p, int q, bool& rotated)
{
    double a = A[p] * A[p] + A[3 + p] * A[3 + p] + A[6 + p] * A[6 + p];
    double b = A[q] * A[q] + A[3 + q] * A[3 + q] + A[6 + q] * A[6 + q];
    double g = A[p] * A[q] + A[3 + p] * A[3 + q] + A[6 + p] * A[6 + q];
    if (g == 0.0 || !(g * g > 5.29e-32 * (a * b))) return;           // |g| <= 2.3e-16 sqrt(a b): columns already orthogonal
    rotated = true;
    // rotation angle without divisions:  zeta = (b-a)/(2g),  t = tan = sign(zeta) / (|zeta| + sqrt(1 + zeta^2))
    //                                                         =  sign(d) h / (|d| + sqrt(d^2 + h^2)),  d = b - a, h = 2 g
    const double d = b - a, h = 2.0 * g;
    const double rad = d * d + h * h;
    // with w = |d| + sqrt(d^2 + h^2):  t = sign(d) h / w,  cos = w / sqrt(w^2 + h^2),  sin = sign(d) h / sqrt(w^2 + h^2)
    const double w = fabs(d) + rad * fast_rsqrt(rad);
    const double rs = fast_rsqrt(w * w + h * h);
    const double cs = w * rs, sn = ((d < 0.0) ? -h : h) * rs;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double x = A[i * 3 + p], y = A[i * 3 + q];
        A[i * 3 + p] = cs * x - sn * y; A[i * 3 + q] = sn * x + cs * y;
        x = W[i * 3 + p]; y = W[i * 3 + q];
        W[i * 3 + p] = cs * x - sn * y; W[i * 3 + q] = sn * x + cs * y;
    }
}

GSF_HD void svd3(const double* H, Svd3& out)
{
    double A[9], W[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = H[i];
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        jacobi_pair(A, W, 0, 1, rotated);
        jacobi_pair(A, W, 0, 2, rotated);
        jacobi_pair(A, W, 1, 2, rotated);
        if (!rotated) break;
    }
    double s0 = sqrt(A[0] * A[0] + A[3] * A[3] + A[6] * A[6]);
    double s1 = sqrt(A[1] * A[1] + A[4] * A[4] + A[7] * A[7]);
    double s2 = sqrt(A[2] * A[2] + A[5] * A[5] + A[8] * A[8]);
    // sort columns descending by singular value (3-element network on column indices)
    int i0 = 0, i1 = 1, i2 = 2; double t0 = s0, t1 = s1, t2 = s2;
    if (t1 > t0) { double t = t0; t0 = t1; t1 = t; int k = i0; i0 = i1; i1 = k; }
    if (t2 > t1) { double t = t1; t1 = t2; t2 = t; int k = i1; i1 = i2; i2 = k; }
    if (t1 > t0) { double t = t0; t0 = t1; t1 = t; int k = i0; i0 = i1; i1 = k; }
    out.S[0] = t0; out.S[1] = t1; out.S[2] = t2;
    double u0[3], u1[3], u2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        out.V[i * 3 + 0] = W[i * 3 + i0]; out.V[i * 3 + 1] = W[i * 3 + i1]; out.V[i * 3 + 2] = W[i * 3 + i2];
        u0[i] = A[i * 3 + i0]; u1[i] = A[i * 3 + i1]; u2[i] = A[i * 3 + i2];
    }
    // U: normalise col 0, Gram-Schmidt col 1, col 2 = +-(u0 x u1); orthonormal even when rank-deficient
    if (t0 > 0.0) { double r = 1.0 / t0; u0[0] *= r; u0[1] *= r; u0[2] *= r; }
    else { u0[0] = 1.0; u0[1] = 0.0; u0[2] = 0.0; }
    if (t1 > 1e-14 * t0 && t1 > 0.0) {
        double d = u1[0] * u0[0] + u1[1] * u0[1] + u1[2] * u0[2];
        u1[0] -= d * u0[0]; u1[1] -= d * u0[1]; u1[2] -= d * u0[2];
    } else {
        int m = 0; if (fabs(u0[1]) < fabs(u0[m])) m = 1; if (fabs(u0[2]) < fabs(u0[m])) m = 2;
        double d = u0[m];
        u1[0] = (m == 0 ? 1.0 : 0.0) - d * u0[0]; u1[1] = (m == 1 ? 1.0 : 0.0) - d * u0[1]; u1[2] = (m == 2 ? 1.0 : 0.0) - d * u0[2];
    }
    { double r = 1.0 / sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]); u1[0] *= r; u1[1] *= r; u1[2] *= r; }
    double c0 = u0[1] * u1[2] - u0[2] * u1[1], c1 = u0[2] * u1[0] - u0[0] * u1[2], c2 = u0[0] * u1[1] - u0[1] * u1[0];
    double sg = (c0 * u2[0] + c1 * u2[1] + c2 * u2[2] < 0.0) ? -1.0 : 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { out.U[i * 3 + 0] = u0[i]; out.U[i * 3 + 1] = u1[i]; }
    out.U[2] = sg * c0; out.U[5] = sg * c1; out.U[8] = sg * c2;
}

// ---------------------------------------------------------------------------------------
// The rotation and the singular-value sum of the Umeyama closed form WITHOUT an SVD, for the fused pipeline's latency-bound
// prelude (one trajectory per wave: the Jacobi SVD above is ~1 150 dependent instructions, this is ~400 / ~530).
//   Vt.T @ U.T (ref :440) is the transpose of the orthogonal polar factor Q = U V^T of H, and S0+S1+S2 = trace(Q^T H).
//   Q by Newton's iteration X <- (g X + (1/g) X^-T)/2 from X0 = H/|H|_F with X^-T = cof(X)/det(X) (3x3 cofactors), three steps
//   with Frobenius scaling g and four plain ones.  X and cof(X) share their singular vectors, so the accuracy of the two
//   coefficients only steers convergence (hardware seeds suffice); the last step takes a full-precision 1/det.  Measured against
//   a 40-digit SVD on 400 track-shaped H with sigma1/sigma3 from 3e3 to 2e14: |Q - Q_true| <= 2.9e-15 (NumPy's own SVD: 1e-13).
//   det(H) < 0 (ref :441-442: the last row of Vt is negated) is R = (I - 2 v3 v3^T) Q^T with v3 the right singular vector of
//   sigma3: cof(X0) = +-sum (sigma_j sigma_k) u_i v_i^T is dominated by u3 v3^T up to e = sigma3/sigma2.  For e < 3e-3 (the
//   usual case) three steps of x <- cof^T (cof x) from its largest row (e^7); up to e = 0.33, B = cof^T cof squared five times
//   (e^64).  A track whose vertical direction is noise has det(H) < 0 half of the time and e up to a few 0.1, so this case is not
//   rare and has to be as cheap as the other.  (Rayleigh-quotient steps on P = Q^T H were tried: with sigma2/sigma1 ~ 1e-6 --
//   straight tracks -- the adjugate of P - rho I loses the direction again: 4e-14 after the power steps, 7e-12 after two steps.)
// Returns false -- the caller then takes the SVD -- for H that is zero / non-finite, singular to working precision, whose
// reflection direction is not separated (sigma3/sigma2 > 0.33), or if the iteration did not land on an orthogonal matrix.
// ---------------------------------------------------------------------------------------
GSF_HD void cof3(const double* X, double* C)                               // C = det(X) X^-T
{
    C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
    C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
    C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
}
GSF_HD double sumsq9(const double* M)
{
    return M[0] * M[0] + M[1] * M[1] + M[2] * M[2] + M[3] * M[3] + M[4] * M[4] + M[5] * M[5] + M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}
// LOWREG: cof(X0) is not kept across the Newton steps but formed again (same operations, same bits) by the rare branch that needs it
template <bool LOWREG = false>
GSF_HD bool umeyama_rotation_polar(const double* H, double* R, double& sum_sigma)
{
    const double nh2 = sumsq9(H);
    if (!(nh2 > 1e-280 && nh2 < 1e280)) return false;
    const double rn = fast_rsqrt(nh2);
    double X[9], C[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) X[i] = H[i] * rn;
    // ---- step 1 (scaled) and the reflection direction, both from cof(X0)
    cof3(X, C);
    double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
    double nc2 = sumsq9(C);
    if (!(det * det > 1e-28 * nc2)) return false;                          // sigma3 < 1e-14 sigma1 (also rank <= 1, NaN)
    const bool reflect = det < 0.0;
    const double e0 = fabs(det), nc20 = nc2;                               // e0 / nc20 >= sigma3 / sigma2  (|X0|_F = 1: sigma1 <= 1)
    // e0 / nc20 >= e / (1 + e^2), e = sigma3 / sigma2 (equality for sigma2 << sigma1; larger for rounder clouds), so a bound on the quotient
    // bounds e: < 0.3 -> e < 0.33 (five squarings below: e^64 < 1e-30), < 0.47 -> e < 0.70 (eight: e^512).  Beyond that the reflection
    // direction is left to the SVD.  (Round 4: under the reference's row choice a track with a GNSS gap fits its first segment only --
    // fewer rows, noisier small singular values -- and those tracks are the slowest of a launch already; with the 0.3 limit alone 7 of
    // the 1 000 tracks of the bench batch took the Jacobi route there, 2 when fitting all valid rows.)
    if (reflect && !(e0 < 0.47 * nc20)) return false;
    // The well-separated reflection direction (the usual case): the largest row of cof(X0), three steps of x <- cof^T (cof x)
    // (e^7 < 1e-17 for e < 3e-3).  Computed unconditionally and in the same straight-line code as the Newton steps below: it depends on
    // cof(X0) only, and the Newton steps are a dependent chain that leaves issue slots free (a lone wave issues a dependent FP64
    // instruction every ~6 cycles, independent ones every ~4.5).  Not normalised yet.
    double C0[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) C0[i] = C[i];
    double v[3];
    {
        const double r0 = C0[0] * C0[0] + C0[1] * C0[1] + C0[2] * C0[2], r1 = C0[3] * C0[3] + C0[4] * C0[4] + C0[5] * C0[5],
                     r2 = C0[6] * C0[6] + C0[7] * C0[7] + C0[8] * C0[8];
        const bool p1 = r1 > r0 && r1 >= r2, p2 = r2 > r0 && r2 > r1;
        v[0] = p2 ? C0[6] : (p1 ? C0[3] : C0[0]); v[1] = p2 ? C0[7] : (p1 ? C0[4] : C0[1]); v[2] = p2 ? C0[8] : (p1 ? C0[5] : C0[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double w0 = C0[0] * v[0] + C0[1] * v[1] + C0[2] * v[2], w1 = C0[3] * v[0] + C0[4] * v[1] + C0[5] * v[2],
                         w2 = C0[6] * v[0] + C0[7] * v[1] + C0[8] * v[2];
            v[0] = C0[0] * w0 + C0[3] * w1 + C0[6] * w2; v[1] = C0[1] * w0 + C0[4] * w1 + C0[7] * w2; v[2] = C0[2] * w0 + C0[5] * w1 + C0[8] * w2;
        }
    }
#pragma unroll
    for (int it = 0; it < 7; ++it) {
        if (it > 0) { cof3(X, C); det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2]; }
        double a, b;
        if (it < 3) {
            // g^2 = sqrt(nc2 / nx2) / |det|:  a = g / 2,  b = 1 / (2 g det) = a sign(det) sqrt(nx2 / nc2)
            if (it > 0) nc2 = sumsq9(C);
            const double nx2 = sumsq9(X);
            const double q = nc2 * seed_rcp(nx2), rq = seed_rsqrt(q);       // rq = sqrt(nx2 / nc2)
            a = 0.5 * ((q * rq) * seed_rsqrt(q * rq)) * seed_rsqrt(fabs(det));   // sqrt(sqrt(q)) / sqrt|det| / 2
            b = reflect ? -(a * rq) : a * rq;
        } else {
            a = 0.5;
            b = 0.5 * (it == 6 ? fast_rcp(det) : seed_rcp(det));
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) X[i] = a * X[i] + b * C[i];
    }
    if (!(fabs(sumsq9(X) - 3.0) < 1e-12)) return false;
    sum_sigma = X[0] * H[0] + X[1] * H[1] + X[2] * H[2] + X[3] * H[3] + X[4] * H[4] + X[5] * H[5] + X[6] * H[6] + X[7] * H[7] + X[8] * H[8];
    if (!reflect) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = X[c * 3 + r];
    } else {
        if (!(e0 < 3e-3 * nc20)) {
            // weakly separated (e up to 0.70): B = cof^T cof = sum (sigma_j sigma_k)^2 v_i v_i^T squared five times (trace-normalised), so
            // that v3 v3^T dominates by e^64; v3 is its largest column
            if (LOWREG) {
                double X0[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) X0[i] = H[i] * rn;
                cof3(X0, C0);
            }
            double b00 = C0[0] * C0[0] + C0[3] * C0[3] + C0[6] * C0[6], b01 = C0[0] * C0[1] + C0[3] * C0[4] + C0[6] * C0[7],
                   b02 = C0[0] * C0[2] + C0[3] * C0[5] + C0[6] * C0[8], b11 = C0[1] * C0[1] + C0[4] * C0[4] + C0[7] * C0[7],
                   b12 = C0[1] * C0[2] + C0[4] * C0[5] + C0[7] * C0[8], b22 = C0[2] * C0[2] + C0[5] * C0[5] + C0[8] * C0[8];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const double rt = seed_rcp(b00 + b11 + b22);
                b00 *= rt; b01 *= rt; b02 *= rt; b11 *= rt; b12 *= rt; b22 *= rt;
                const double c00 = b00 * b00 + b01 * b01 + b02 * b02, c01 = b00 * b01 + b01 * b11 + b02 * b12,
                             c02 = b00 * b02 + b01 * b12 + b02 * b22, c11 = b01 * b01 + b11 * b11 + b12 * b12,
                             c12 = b01 * b02 + b11 * b12 + b12 * b22, c22 = b02 * b02 + b12 * b12 + b22 * b22;
                b00 = c00; b01 = c01; b02 = c02; b11 = c11; b12 = c12; b22 = c22;
            }
            if (!(e0 < 0.3 * nc20)) {                                      // e up to 0.70: three more squarings (e^512)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double rt = seed_rcp(b00 + b11 + b22);
                    b00 *= rt; b01 *= rt; b02 *= rt; b11 *= rt; b12 *= rt; b22 *= rt;
                    const double c00 = b00 * b00 + b01 * b01 + b02 * b02, c01 = b00 * b01 + b01 * b11 + b02 * b12,
                                 c02 = b00 * b02 + b01 * b12 + b02 * b22, c11 = b01 * b01 + b11 * b11 + b12 * b12,
                                 c12 = b01 * b02 + b11 * b12 + b12 * b22, c22 = b02 * b02 + b12 * b12 + b22 * b22;
                    b00 = c00; b01 = c01; b02 = c02; b11 = c11; b12 = c12; b22 = c22;
                }
            }
            const bool p1 = b11 > b00 && b11 >= b22, p2 = b22 > b00 && b22 > b11;   // the largest diagonal entry names the largest column
            v[0] = p2 ? b02 : (p1 ? b01 : b00); v[1] = p2 ? b12 : (p1 ? b11 : b01); v[2] = p2 ? b22 : (p1 ? b12 : b02);
        }
        const double vn2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        if (!(vn2 > 1e-290)) return false;
        const double rv = fast_rsqrt(vn2);
        v[0] *= rv; v[1] *= rv; v[2] *= rv;
        const double w0 = X[0] * v[0] + X[1] * v[1] + X[2] * v[2], w1 = X[3] * v[0] + X[4] * v[1] + X[5] * v[2],
                     w2 = X[6] * v[0] + X[7] * v[1] + X[8] * v[2];                  // Q v3 (= u3)
        const double w[3] = { w0, w1, w2 };
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = X[c * 3 + r] - 2.0 * v[r] * w[c];
    }
    return true;
}

// status codes/flags of a Sim3 fit (mirrored in include/gsf.h)
enum : int32_t { SIM3_OK = 0, SIM3_NONE = 1, SIM3_FLAG_VAR0 = 2, SIM3_FLAG_SMALL_SCALE = 4, SIM3_FLAG_BAD_INDEX = 8, SIM3_FLAG_SVD_FALLBACK = 16,
                 SIM3_FLAG_FEW_ROWS = 32, SIM3_FLAG_ROWS_ALL = 64, SIM3_FLAG_ROWS_SEGMENT = 128,     // the row choice of main_process_gui (ref :973-998)
                 SIM3_FLAG_SATURATED = 256 };                                                        // robust chain, early exit: a trial counted every row (ref :413)

// the SVD route of the closed form's rotation, ref :439-444: R = Vt.T @ U.T with the reflection fix, tr = S0 + S1 + S2 det(R)
GSF_HD void umeyama_rotation_svd(const double* H, double* R, double& tr)
{
    Svd3 s; svd3(H, s);
    // R = Vt.T @ U.T = V U^T                                               (:440)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            R[r * 3 + c] = s.V[r * 3 + 0] * s.U[c * 3 + 0] + s.V[r * 3 + 1] * s.U[c * 3 + 1] + s.V[r * 3 + 2] * s.U[c * 3 + 2];
    if (det3(R) < 0.0) {                                                   // (:441-442) flip last row of Vt
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                R[r * 3 + c] = s.V[r * 3 + 0] * s.U[c * 3 + 0] + s.V[r * 3 + 1] * s.U[c * 3 + 1] - s.V[r * 3 + 2] * s.U[c * 3 + 2];
    }
    tr = s.S[0] + s.S[1] + s.S[2] * det3(R);                               // (:444, Q12)
}
// the same as a real CALL (arguments through memory): for kernels whose register budget must not pay for the rare fallback
struct SvdRot { double R[9], tr; };
GSF_HD_COLD void umeyama_rotation_svd_cold(const double* H, SvdRot* out)
{
    double Hc[9], R[9], tr;
    for (int i = 0; i < 9; ++i) Hc[i] = H[i];
    umeyama_rotation_svd(Hc, R, tr);
    for (int i = 0; i < 9; ++i) out->R[i] = R[i];
    out->tr = tr;
}

// Umeyama closed form from the reduced moments, ref :439-451.
//   H = sum src_c dst_c^T (row-major), ssq = sum |src_c|^2, sc/dc centroids, n points.
// POLAR: take the rotation from umeyama_rotation_polar when it applies (the fused pipeline; identical to the SVD route to ~1e-14).
// COLD_SVD: the SVD fallback of the POLAR route is a call instead of inline code (same arithmetic).
template <bool POLAR = false, bool COLD_SVD = false>
GSF_HD int32_t umeyama_finalize(const double* H, double ssq, const double* sc, const double* dc, double n,
                                double* R, double* t, double& scale)
{
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 9; ++i) finite = finite && (fabs(H[i]) < INFINITY);   // NaN/inf -> LinAlgError -> None (:452)
    if (!finite) return SIM3_NONE;
    double tr = 0.0;
    int32_t flags = SIM3_OK;
    if (!(POLAR && umeyama_rotation_polar<COLD_SVD>(H, R, tr))) {                        // tr: S0+S1+S2 (det(R) = 1 to rounding, Q12)
        if (POLAR) flags |= SIM3_FLAG_SVD_FALLBACK;                                      // informational: the polar route declined this H
        if (COLD_SVD) {
            double Hm[9]; SvdRot o;
#pragma unroll
            for (int i = 0; i < 9; ++i) Hm[i] = H[i];
            umeyama_rotation_svd_cold(Hm, &o);
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = o.R[i];
            tr = o.tr;
        } else {
            umeyama_rotation_svd(H, R, tr);
        }
    }
    const double rcp_n = fast_rcp(n);                                          // n >= 3
    double var_src = ssq * rcp_n;                                              // (:443)
    if (var_src < 1e-12) { scale = 1.0; flags |= SIM3_FLAG_VAR0; }             // (:445-447)
    else {
        scale = tr * fast_rcp(n * var_src);                                    // (:449)  n var_src >= 3e-12: normal range
        if (scale <= 1e-6) { scale = 1.0; flags |= SIM3_FLAG_SMALL_SCALE; }    // (:450)
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
        t[r] = dc[r] - scale * (R[r * 3] * sc[0] + R[r * 3 + 1] * sc[1] + R[r * 3 + 2] * sc[2]);   // (:451)
    return flags;
}

// ---------------------------------------------------------------------------------------
// WGS84 UTM, Krueger n-series to n^6 (replaces pyproj Proj("+proj=utm ..."), ref :267-270, :295)
// ---------------------------------------------------------------------------------------
struct TmConsts {
    double e, k0A, alpha[6], beta[6], delta[6];
};
GSF_HD TmConsts tm_consts()
{
    const double f = 1.0 / 298.257223563, a = 6378137.0, k0 = 0.9996;
    const double n = f / (2.0 - f), n2 = n * n, n3 = n2 * n, n4 = n3 * n, n5 = n4 * n, n6 = n5 * n;
    TmConsts c;
    c.e = sqrt(f * (2.0 - f));
    c.k0A = k0 * (a / (1.0 + n) * (1.0 + n2 / 4.0 + n4 / 64.0 + n6 / 256.0));
    c.alpha[0] = n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800;
    c.alpha[1] = 13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360;
    c.alpha[2] = 61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440;
    c.alpha[3] = 49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600;
    c.alpha[4] = 34729 * n5 / 80640 - 3418889 * n6 / 1995840;
    c.alpha[5] = 212378941 * n6 / 319334400;
    c.beta[0] = n / 2 - 2 * n2 / 3 + 37 * n3 / 96 - n4 / 360 - 81 * n5 / 512 + 96199 * n6 / 604800;
    c.beta[1] = n2 / 48 + n3 / 15 - 437 * n4 / 1440 + 46 * n5 / 105 - 1118711 * n6 / 3870720;
    c.beta[2] = 17 * n3 / 480 - 37 * n4 / 840 - 209 * n5 / 4480 + 5569 * n6 / 90720;
    c.beta[3] = 4397 * n4 / 161280 - 11 * n5 / 504 - 830251 * n6 / 7257600;
    c.beta[4] = 4583 * n5 / 161280 - 108847 * n6 / 3991680;
    c.beta[5] = 20648693 * n6 / 638668800;
    // conformal (Gaussian) latitude chi -> geodetic latitude: phi = chi + sum_j delta_j sin(2 j chi), to n^6 (the series PROJ's extended
    // transverse Mercator takes on its way back; truncation 8e-18 rad against a 50-digit evaluation, tests/golden/gen_utm_mpmath.py)
    c.delta[0] = 2 * n - 2 * n2 / 3 - 2 * n3 + 116 * n4 / 45 + 26 * n5 / 45 - 2854 * n6 / 675;
    c.delta[1] = 7 * n2 / 3 - 8 * n3 / 5 - 227 * n4 / 45 + 2704 * n5 / 315 + 2323 * n6 / 945;
    c.delta[2] = 56 * n3 / 15 - 136 * n4 / 35 - 1262 * n5 / 105 + 73814 * n6 / 2835;
    c.delta[3] = 4279 * n4 / 630 - 332 * n5 / 35 - 399572 * n6 / 14175;
    c.delta[4] = 4174 * n5 / 315 - 144838 * n6 / 6237;
    c.delta[5] = 601676 * n6 / 22275;
    return c;
}

// Sum_{j=1..6} c_j * {sin,cos}(2j x) * {cosh,sinh}(2j y) via angle-addition recurrences from the j = 1 values
// (s2, c2) = (sin 2x, cos 2x) and (sh2, ch2) = (sinh 2y, cosh 2y): no transcendental call in here.
GSF_HD void tm_series(const double* c, double s2, double c2, double sh2, double ch2, double& s_sin_cosh, double& s_cos_sinh)
{
    double sj = s2, cj = c2, chj = ch2, shj = sh2;
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        a += c[j] * sj * chj;
        b += c[j] * cj * shj;
        double sn = sj * c2 + cj * s2, cn = cj * c2 - sj * s2;
        double chn = chj * ch2 + shj * sh2, shn = shj * ch2 + chj * sh2;
        sj = sn; cj = cn; chj = chn; shj = shn;
    }
    s_sin_cosh = a; s_cos_sinh = b;
}

// sin and cos of an angle in radians, for the arguments geodesy has: latitudes, longitudes and their differences, |x| <= 7 (401 degrees).
// One Cody-Waite step in FMA form (x - k pi/2 with pi/2 split into 33 + 33 + 53 bits: the first two products are exact for |k| <= 4, so
// the reduced argument carries the rounding of the last step only, also right next to a multiple of pi/2), then fdlibm's kernel polynomials
// on |r| <= pi/4 (degree 13 / 14; truncation below 2^-57) and the quadrant.  About 35 instructions for both values; absolute errors below
// 1.2e-16, i.e. under 1.3 units in the last place of each (tests/test_host_math.py measures them against 60-digit values and libm: the same
// code runs on the host; a device libm is allowed 4).  libm's general routine
// -- large-argument reduction included, which these arguments never need -- stays behind the range test.
GSF_HD void gsf_sincos(double x, double& s, double& c)
{
    if (!(fabs(x) <= 7.0)) {
#if defined(__HIP_DEVICE_COMPILE__)
        ::sincos(x, &s, &c);
#else
        s = sin(x); c = cos(x);
#endif
        return;
    }
    const double kd = rint(x * 6.36619772367581382433e-01);              // nearest multiple of pi/2
    double r = fma(-kd, 1.57079632673412561417e+00, x);                  // pi/2: first 33 bits,
    r = fma(-kd, 6.07710050630396597660e-11, r);                         //       next 33 bits,
    r = fma(-kd, 2.02226624879595063154e-21, r);                         //       the rest
    const double z = r * r;
    const double ps = -1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10));
    const double sn = fma(z * r, fma(z, fma(z, ps, 8.33333333332248946124e-03), -1.66666666666666324348e-01), r);
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double hz = 0.5 * z, w1 = 1.0 - hz;
    const double cs = w1 + fma(z * z, pc, (1.0 - w1) - hz);             // 1 - z/2 with its rounding error put back (the sum is exact: z/2 <= 0.31)
    const int k = (int)kd;
    const double a = (k & 1) ? cs : sn, b = (k & 1) ? sn : cs;          // k odd: the two swap
    s = (k & 2) ? -a : a;                                                // k = 2, 3: sin changes sign
    c = ((k + 1) & 2) ? -b : b;                                          // k = 1, 2: cos changes sign
}

// sigma(tau) = sinh(e atanh(e tau / sqrt(1 + tau^2))) = sinh(e atanh(e sin phi)) (Karney 2011 eq. 8-9).  |e sin phi| <= 0.082,
// so both functions are short odd series (remainders < 1e-17 relative) instead of two libm calls.
GSF_HD double tm_sigma(double e, double sin_phi)
{
    const double z = e * sin_phi, z2 = z * z;
    const double ath = z * (1.0 + z2 * (1.0 / 3 + z2 * (1.0 / 5 + z2 * (1.0 / 7 + z2 * (1.0 / 9 + z2 * (1.0 / 11 + z2 * (1.0 / 13 + z2 * (1.0 / 15))))))));
    const double y = e * ath, y2 = y * y;
    return y * (1.0 + y2 * (1.0 / 6 + y2 * (1.0 / 120 + y2 * (1.0 / 5040))));
}

// asinh for the small arguments of a UTM zone (|w| = |sinh eta'| <= sin 3 deg = 0.052 on the equator): alternating series to
// w^18 below 1/8 (remainder < 4e-17), libm beyond.
GSF_HD double asinh_small(double w)
{
    if (!(fabs(w) < 0.125)) return asinh(w);
    const double w2 = w * w;
    return w * (1.0 + w2 * (-1.0 / 6 + w2 * (3.0 / 40 + w2 * (-15.0 / 336 + w2 * (105.0 / 3456 + w2 * (-945.0 / 42240 + w2 * (10395.0 / 599040 +
               w2 * (-135135.0 / 9676800 + w2 * (2027025.0 / 175472640)))))))));
}

// forward: degrees -> metres.  lon0_deg = 6*zone-183; fn = 0 or 1e7.  Karney's exact-tau form of the Krueger series; the only
// libm-class calls left are sincos(phi), sincos(lambda) and atan2:
//   sin/cos(2 xi') come from (tau', cos lambda) / hypot, and with w = sin(lambda) / hypot = sinh(eta') exactly,
//   sinh(2 eta') = 2 w sqrt(1 + w^2), cosh(2 eta') = 1 + 2 w^2 -- no exp, no second sincos.
GSF_HD void utm_forward_point(const TmConsts& c, double lat_deg, double lon_deg, double lon0_deg, double fn,
                              double& easting, double& northing)
{
    const double d2r = 0.017453292519943295769;
    // (the difference is taken in DEGREES, where it is exact or nearly so -- lon0 is an integer --, and converted once: converting both
    // longitudes first leaves 2e-16 x |lon| rad of rounding in a lambda of a few hundredths, 2.7e-9 m at zone 60; round 5)
    const double phi = lat_deg * d2r, lam = (lon_deg - lon0_deg) * d2r;
    double sp, cp; gsf_sincos(phi, sp, cp);
    // (1/x and 1/sqrt(x) by hardware seed + Newton: every operand here is a normal number of moderate size)
    const double t1 = fast_rcp(cp), tau = sp * t1;            // sqrt(1 + tan^2), tan(phi)   (|phi| <= pi/2: cp > 0)
    const double sig = tm_sigma(c.e, sp);
    const double os2 = 1.0 + sig * sig;
    const double taup = tau * (os2 * fast_rsqrt(os2)) - sig * t1;
    double sl, cl; gsf_sincos(lam, sl, cl);
    const double ih = fast_rsqrt(taup * taup + cl * cl);
    const double xip = atan2(taup, cl);
    const double w = sl * ih;                                  // sinh(eta')
    const double etap = asinh_small(w);
    const double s1 = taup * ih, c1 = cl * ih;                 // sin(xi'), cos(xi')
    const double ow2 = 1.0 + w * w;
    const double ch1 = ow2 * fast_rsqrt(ow2);
    double a, b;
    tm_series(c.alpha, 2.0 * s1 * c1, (c1 - s1) * (c1 + s1), 2.0 * w * ch1, 1.0 + 2.0 * w * w, a, b);
    easting = 500000.0 + c.k0A * (etap + b);
    northing = fn + c.k0A * (xip + a);
}

// sinh / cosh of a small argument by series (|z| < 1/2: remainders < 1e-17), exp beyond
GSF_HD void sinh_cosh_small(double z, double& sh, double& ch)
{
    if (fabs(z) < 0.5) {
        // (Taylor coefficients as constants: the nested z2 / 6.0 * (1.0 + z2 / 20.0 * ...) form of rounds 1-4 cost fifteen float64 DIVISIONS per
        // call -- two calls per inverse projection, ~25 instructions each -- and was what held K1 inverse at 1.8 x the forward time)
        const double z2 = z * z;
        sh = z * (1.0 + z2 * (1.0 / 6 + z2 * (1.0 / 120 + z2 * (1.0 / 5040 + z2 * (1.0 / 362880 + z2 * (1.0 / 39916800 + z2 * (1.0 / 6227020800.0 +
                 z2 * (1.0 / 1307674368000.0))))))));
        ch = 1.0 + z2 * (1.0 / 2 + z2 * (1.0 / 24 + z2 * (1.0 / 720 + z2 * (1.0 / 40320 + z2 * (1.0 / 3628800 + z2 * (1.0 / 479001600.0 + z2 * (1.0 / 87178291200.0 +
                   z2 * (1.0 / 20922789888000.0))))))));
    } else {
        const double ez = exp(z), ezi = 1.0 / ez;
        sh = 0.5 * (ez - ezi); ch = 0.5 * (ez + ezi);
    }
}

// inverse: metres -> degrees
GSF_HD void utm_inverse_point(const TmConsts& c, double easting, double northing, double lon0_deg, double fn,
                              double& lat_deg, double& lon_deg)
{
    const double r2d = 57.295779513082320877;
    double xi = (northing - fn) / c.k0A, eta = (easting - 500000.0) / c.k0A;
    double s2, c2, sh2, ch2;
    gsf_sincos(2.0 * xi, s2, c2);
    sinh_cosh_small(2.0 * eta, sh2, ch2);
    double a, b;
    tm_series(c.beta, s2, c2, sh2, ch2, a, b);
    double xip = xi - a, etap = eta - b;
    double sh, chd; sinh_cosh_small(etap, sh, chd);
    double sx, cx; gsf_sincos(xip, sx, cx);
    const double lam = atan2(sh, cx);
    // conformal latitude chi of the point: sin chi = sin xi' / cosh eta', cos chi = hypot(sinh eta', cos xi') / cosh eta' (the three squares
    // add up to cosh^2 eta'), then the geodetic latitude by the direct series in n -- no iteration, no data-dependent exit inside a wave
    // (rounds 1-4 solved tau'(tau) = tau' by Newton: up to six rounds with a divergent break, 2.59 ms per 1e8 points against 1.48 ms forward)
    const double hyp2 = sh * sh + cx * cx;
    const double hyp = hyp2 > 0.0 ? hyp2 * fast_rsqrt(hyp2) : 0.0;
    const double ich = fast_rcp(chd);
    const double sc = sx * ich, cc = hyp * ich;
    const double chi = atan2(sx, hyp);
    const double s2c = 2.0 * sc * cc, c2c = (cc - sc) * (cc + sc);
    double sj = s2c, cj = c2c, dphi = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        dphi += c.delta[j] * sj;
        const double sn = sj * c2c + cj * s2c, cn = cj * c2c - sj * s2c;
        sj = sn; cj = cn;
    }
    lat_deg = (chi + dphi) * r2d;
    lon_deg = lam * r2d + lon0_deg;
}

// ---------------------------------------------------------------------------------------
// WGS84 geodetic (deg, deg, m) -> local East-North-Up about a reference point (offered in ADDITION to UTM: the reference's
// pipeline projects with UTM, BASELINE.json's north star words it as "WGS84 -> local ENU")
// ---------------------------------------------------------------------------------------
struct EnuFrame { double x0, y0, z0, sl, cl, sp, cp; };      // ECEF of the origin, sin/cos of its lon / lat
GSF_HD void geodetic_to_ecef(double lat_deg, double lon_deg, double h, double& x, double& y, double& z)
{
    const double a = 6378137.0, f = 1.0 / 298.257223563, e2 = f * (2.0 - f), d2r = 0.017453292519943295769;
    double sp, cp, sl, cl;
    gsf_sincos(lat_deg * d2r, sp, cp); gsf_sincos(lon_deg * d2r, sl, cl);
    const double Nn = a * fast_rsqrt(1.0 - e2 * sp * sp);                 // prime-vertical radius; the argument is in [0.9933, 1]
    x = (Nn + h) * cp * cl; y = (Nn + h) * cp * sl; z = (Nn * (1.0 - e2) + h) * sp;
}
GSF_HD EnuFrame enu_frame(double lat0_deg, double lon0_deg, double h0)
{
    const double a = 6378137.0, f0 = 1.0 / 298.257223563, e2 = f0 * (2.0 - f0), d2r = 0.017453292519943295769;
    EnuFrame f;
    gsf_sincos(lat0_deg * d2r, f.sp, f.cp); gsf_sincos(lon0_deg * d2r, f.sl, f.cl);   // the same four values geodetic_to_ecef forms for the origin
    const double Nn = a * fast_rsqrt(1.0 - e2 * f.sp * f.sp);
    f.x0 = (Nn + h0) * f.cp * f.cl; f.y0 = (Nn + h0) * f.cp * f.sl; f.z0 = (Nn * (1.0 - e2) + h0) * f.sp;
    return f;
}
GSF_HD void geodetic_to_enu_point(const EnuFrame& f, double lat_deg, double lon_deg, double h, double& e, double& n, double& u)
{
    double x, y, z; geodetic_to_ecef(lat_deg, lon_deg, h, x, y, z);
    const double dx = x - f.x0, dy = y - f.y0, dz = z - f.z0;
    e = -f.sl * dx + f.cl * dy;
    n = -f.sp * f.cl * dx - f.sp * f.sl * dy + f.cp * dz;
    u = f.cp * f.cl * dx + f.cp * f.sl * dy + f.sp * dz;
}

}  // namespace gsf
