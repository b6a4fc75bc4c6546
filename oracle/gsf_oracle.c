/*
 * gsf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded restatement of the GPS<->SLAM trajectory-fusion
 * hot path of the reference (A2ureeE/GPS-optimize-SLAM, EKFGPSSLAM.py).  It is
 * used ONLY as the checker by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py.  The product path (gps_optimize_slam_amd/)
 * never links, imports or calls anything in this directory.
 *
 * Every function cites the reference lines it follows ("ref :NNN" means
 * /root/reference/EKFGPSSLAM.py line NNN).  The restatement deliberately keeps
 * the reference's *dense* 7x7 covariance algebra (matrix products, explicit
 * inverses) instead of the diagonal shortcut the HIP kernels use, so that the
 * two implementations are independent.
 *
 * Parity status:
 *   - Sim3 / transform / EKF / RTS / sharp-turn / time-alignment: PINNED against
 *     golden vectors produced by importing the reference itself in the build
 *     container (tests/golden/gen_golden.py -> tests/golden/ *.npz).
 *   - UTM forward/inverse: "parity unpinned vs pyproj" (pyproj/PROJ are not
 *     installable offline).  Pinned instead against a 50-digit mpmath evaluation
 *     of the transverse-Mercator *definition* (tests/golden/gen_utm_mpmath.py)
 *     and closed-form identities.
 *
 * Third-party arithmetic restated here (library calls in the reference):
 *   numpy.linalg.svd/det/inv/norm, scipy Rotation (from_quat, inv, apply, *,
 *   as_quat, from_matrix, as_euler('zyx')), scipy interp1d(cubic|linear),
 *   pyproj Proj("+proj=utm ...") forward/inverse.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* small dense helpers                                                       */
/* ------------------------------------------------------------------------ */
static void mat_mul(const double *A, const double *B, double *C, int n, int k, int m)
{   /* C[n x m] = A[n x k] * B[k x m], row-major, plain triple loop */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * m + j];
            C[i * m + j] = s;
        }
}
static void mat_T(const double *A, double *AT, int n, int m)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) AT[j * n + i] = A[i * m + j];
}
static void mat_sym(double *A, int n)
{   /* (A + A^T)/2, as ref :715,:724,:732,:802 */
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            double v = (A[i * n + j] + A[j * n + i]) / 2.0;
            A[i * n + j] = v; A[j * n + i] = v;
        }
}
/* Gauss-Jordan inverse with partial pivoting (stands in for np.linalg.inv,
 * LAPACK dgetrf/dgetri).  Returns 0 on success, 1 if singular. */
static int mat_inv(const double *A, double *Ainv, int n)
{
    double W[7 * 14];
    if (n > 7) return 1;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            W[i * 2 * n + j] = A[i * n + j];
            W[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; ++c) {
        int p = c; double best = fabs(W[c * 2 * n + c]);
        for (int r = c + 1; r < n; ++r) {
            double v = fabs(W[r * 2 * n + c]);
            if (v > best) { best = v; p = r; }
        }
        if (!(best > 0.0) || isnan(best)) return 1;
        if (p != c)
            for (int j = 0; j < 2 * n; ++j) {
                double t = W[c * 2 * n + j]; W[c * 2 * n + j] = W[p * 2 * n + j]; W[p * 2 * n + j] = t;
            }
        double piv = W[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) W[c * 2 * n + j] /= piv;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = W[r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) W[r * 2 * n + j] -= f * W[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Ainv[i * n + j] = W[i * 2 * n + n + j];
    return 0;
}
static double det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* ------------------------------------------------------------------------ */
/* scipy.spatial.transform.Rotation conventions                 */
/* quaternions are scalar-last [x,y,z,w]                                     */
/* ------------------------------------------------------------------------ */
/* Rotation.from_quat: normalise; ValueError on zero (or NaN) norm -> return 1 */
static int quat_from(const double *q, double *out)
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(n > 0.0) || isinf(n)) return 1;
    out[0] = q[0] / n; out[1] = q[1] / n; out[2] = q[2] / n; out[3] = q[3] / n;
    return 0;
}
static void quat_inv(const double *q, double *o) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
/* Rotation.__mul__: p*q = Hamilton product, "q first then p", no renormalise */
static void quat_mul(const double *p, const double *q, double *o)
{
    double x = p[3] * q[0] + q[3] * p[0] + (p[1] * q[2] - p[2] * q[1]);
    double y = p[3] * q[1] + q[3] * p[1] + (p[2] * q[0] - p[0] * q[2]);
    double z = p[3] * q[2] + q[3] * p[2] + (p[0] * q[1] - p[1] * q[0]);
    double w = p[3] * q[3] - (p[0] * q[0] + p[1] * q[1] + p[2] * q[2]);
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
/* Rotation.as_matrix for a unit quaternion */
static void quat_to_mat(const double *q, double *M)
{
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    M[0] = x2 - y2 - z2 + w2; M[3] = 2 * (xy + zw);       M[6] = 2 * (xz - yw);
    M[1] = 2 * (xy - zw);     M[4] = -x2 + y2 - z2 + w2;  M[7] = 2 * (yz + xw);
    M[2] = 2 * (xz + yw);     M[5] = 2 * (yz - xw);       M[8] = -x2 - y2 + z2 + w2;
}
/* Rotation.apply(v) = as_matrix() @ v */
static void quat_apply(const double *q, const double *v, double *o)
{
    double M[9]; quat_to_mat(q, M);
    double a = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    double b = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    double c = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
/* Rotation.from_matrix (Markley: pick the largest of m00,m11,m22,trace) */
static void quat_from_mat(const double *M, double *q)
{
    double tr = M[0] + M[4] + M[8];
    double dec[4] = { M[0], M[4], M[8], tr };
    int c = 0;
    for (int i = 1; i < 4; ++i) if (dec[i] > dec[c]) c = i;
    if (c != 3) {
        int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1.0 - tr + 2.0 * M[i * 3 + i];
        q[j] = M[j * 3 + i] + M[i * 3 + j];
        q[k] = M[k * 3 + i] + M[i * 3 + k];
        q[3] = M[k * 3 + j] - M[j * 3 + k];
    } else {
        q[0] = M[7] - M[5]; q[1] = M[2] - M[6]; q[2] = M[3] - M[1]; q[3] = 1.0 + tr;
    }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
/* as_euler('zyx')[0]: extrinsic z-y-x, first angle (rotation about fixed z).
 * For R = Rx(c) Ry(b) Rz(a): a = atan2(-R01, R00). */
static double quat_yaw_zyx(const double *qn)
{
    double M[9]; quat_to_mat(qn, M);
    return atan2(-M[1], M[0]);
}

/* ref :697-700  ExtendedKalmanFilter.normalize_quaternion */
static void ekf_normalize_quat(double *q)
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 1e-9) { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
    else { q[0] = 0; q[1] = 0; q[2] = 0; q[3] = 1; }
}

/* ref :77-92  calculate_relative_pose.  Returns 1 if the invalid-quaternion
 * branch (:84-86) was taken. */
ORC_API int orc_relative_pose(const double *p1, const double *q1, const double *p2, const double *q2,
                              double *dpos_local, double *dquat)
{
    double r1[4], r2[4], r1i[4];
    if (quat_from(q1, r1) || quat_from(q2, r2)) {
        dpos_local[0] = dpos_local[1] = dpos_local[2] = 0.0;
        dquat[0] = dquat[1] = dquat[2] = 0.0; dquat[3] = 1.0;
        return 1;
    }
    quat_inv(r1, r1i);
    double dw[3] = { p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2] };
    quat_apply(r1i, dw, dpos_local);
    quat_mul(r1i, r2, dquat);
    return 0;
}

/* ref :94-105  quaternion_nlerp */
ORC_API void orc_quaternion_nlerp(const double *q1, const double *q2in, double weight_q2, double *out)
{
    double q2[4] = { q2in[0], q2in[1], q2in[2], q2in[3] };
    double dot = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
    if (dot < 0.0) { q2[0] = -q2[0]; q2[1] = -q2[1]; q2[2] = -q2[2]; q2[3] = -q2[3]; }
    double w = weight_q2 < 0.0 ? 0.0 : (weight_q2 > 1.0 ? 1.0 : weight_q2);
    double qi[4];
    for (int i = 0; i < 4; ++i) qi[i] = (1.0 - w) * q1[i] + w * q2[i];
    double n = sqrt(qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3]);
    if (n < 1e-9) {
        const double *src = (weight_q2 < 0.5) ? q1 : q2;
        for (int i = 0; i < 4; ++i) out[i] = src[i];
        return;
    }
    for (int i = 0; i < 4; ++i) out[i] = qi[i] / n;
}

/* ------------------------------------------------------------------------ */
/* UTM  (ref :127-134 zone pick, :266-271 forward, :291-296 inverse)         */
/* Kruger n-series, 6th order (Karney 2011, eqs 35-36), WGS84                */
/* ------------------------------------------------------------------------ */
#define WGS84_A 6378137.0
#define WGS84_F (1.0 / 298.257223563)
#define UTM_K0 0.9996
#define UTM_FE 500000.0
#define UTM_FN_SOUTH 10000000.0

typedef struct { double n, A, e, alpha[6], beta[6]; } tm_consts;

static void tm_init(tm_consts *c)
{
    double f = WGS84_F, n = f / (2.0 - f);
    double n2 = n * n, n3 = n2 * n, n4 = n3 * n, n5 = n4 * n, n6 = n5 * n;
    c->n = n;
    c->e = sqrt(f * (2.0 - f));
    c->A = WGS84_A / (1.0 + n) * (1.0 + n2 / 4.0 + n4 / 64.0 + n6 / 256.0);
    c->alpha[0] = n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800;
    c->alpha[1] = 13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360;
    c->alpha[2] = 61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440;
    c->alpha[3] = 49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600;
    c->alpha[4] = 34729 * n5 / 80640 - 3418889 * n6 / 1995840;
    c->alpha[5] = 212378941 * n6 / 319334400;
    c->beta[0] = n / 2 - 2 * n2 / 3 + 37 * n3 / 96 - n4 / 360 - 81 * n5 / 512 + 96199 * n6 / 604800;
    c->beta[1] = n2 / 48 + n3 / 15 - 437 * n4 / 1440 + 46 * n5 / 105 - 1118711 * n6 / 3870720;
    c->beta[2] = 17 * n3 / 480 - 37 * n4 / 840 - 209 * n5 / 4480 + 5569 * n6 / 90720;
    c->beta[3] = 4397 * n4 / 161280 - 11 * n5 / 504 - 830251 * n6 / 7257600;
    c->beta[4] = 4583 * n5 / 161280 - 108847 * n6 / 3991680;
    c->beta[5] = 20648693 * n6 / 638668800;
}

/* ref :127-134 auto_utm_projection: zone = int((mean(lon)+180)//6+1); south iff mean(lat)<0 */
ORC_API int orc_utm_zone(const double *lons, const double *lats, int64_t n, int *zone, int *south)
{
    if (n <= 0) return 1;
    double sl = 0.0, sp = 0.0;
    for (int64_t i = 0; i < n; ++i) { sl += lons[i]; sp += lats[i]; }
    double ml = sl / (double)n, mp = sp / (double)n;
    *zone = (int)(floor((ml + 180.0) / 6.0) + 1.0);
    *south = (mp < 0.0) ? 1 : 0;
    return 0;
}

/* ref :270 projector(lons, lats): degrees in, metres out */
ORC_API void orc_utm_forward(const double *lat_deg, const double *lon_deg, int64_t n, int zone, int south,
                             double *easting, double *northing)
{
    tm_consts c; tm_init(&c);
    const double d2r = M_PI / 180.0;
    double lon0 = ((double)zone * 6.0 - 183.0) * d2r;
    double fn = south ? UTM_FN_SOUTH : 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double phi = lat_deg[i] * d2r, lam = lon_deg[i] * d2r - lon0;
        double tau = tan(phi);
        double sig = sinh(c.e * atanh(c.e * tau / sqrt(1.0 + tau * tau)));
        double taup = tau * sqrt(1.0 + sig * sig) - sig * sqrt(1.0 + tau * tau);
        double cl = cos(lam), sl = sin(lam);
        double xip = atan2(taup, cl);
        double etap = asinh(sl / sqrt(taup * taup + cl * cl));
        double xi = xip, eta = etap;
        for (int j = 0; j < 6; ++j) {
            double k2 = 2.0 * (double)(j + 1);
            xi += c.alpha[j] * sin(k2 * xip) * cosh(k2 * etap);
            eta += c.alpha[j] * cos(k2 * xip) * sinh(k2 * etap);
        }
        easting[i] = UTM_FE + UTM_K0 * c.A * eta;
        northing[i] = fn + UTM_K0 * c.A * xi;
    }
}

/* ref :295 projector(x, y, inverse=True): metres in, degrees out */
ORC_API void orc_utm_inverse(const double *easting, const double *northing, int64_t n, int zone, int south,
                             double *lat_deg, double *lon_deg)
{
    tm_consts c; tm_init(&c);
    const double r2d = 180.0 / M_PI;
    double lon0_deg = (double)zone * 6.0 - 183.0;
    double fn = south ? UTM_FN_SOUTH : 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double xi = (northing[i] - fn) / (UTM_K0 * c.A), eta = (easting[i] - UTM_FE) / (UTM_K0 * c.A);
        double xip = xi, etap = eta;
        for (int j = 0; j < 6; ++j) {
            double k2 = 2.0 * (double)(j + 1);
            xip -= c.beta[j] * sin(k2 * xi) * cosh(k2 * eta);
            etap -= c.beta[j] * cos(k2 * xi) * sinh(k2 * eta);
        }
        double sh = sinh(etap), cx = cos(xip);
        double taup = sin(xip) / sqrt(sh * sh + cx * cx);
        double lam = atan2(sh, cx);
        /* Newton for tau from tau' (Karney eqs 19-21) */
        double tau = taup / (1.0 - c.e * c.e);  /* starting guess */
        for (int it = 0; it < 8; ++it) {
            double t1 = sqrt(1.0 + tau * tau);
            double sig = sinh(c.e * atanh(c.e * tau / t1));
            double tpi = tau * sqrt(1.0 + sig * sig) - sig * t1;
            double dtau = (taup - tpi) / sqrt(1.0 + tpi * tpi) * (1.0 + (1.0 - c.e * c.e) * tau * tau) /
                          ((1.0 - c.e * c.e) * t1);
            tau += dtau;
            if (fabs(dtau) < 1e-15 * (1.0 + fabs(tau))) break;
        }
        lat_deg[i] = atan(tau) * r2d;
        lon_deg[i] = lam * r2d + lon0_deg;
    }
}

/* ------------------------------------------------------------------------ */
/* Sim3 / Umeyama                                                            */
/* ------------------------------------------------------------------------ */
/* One-sided (Hestenes) Jacobi SVD of a 3x3, H = U diag(S) V^T, S descending.
 * Stands in for np.linalg.svd (LAPACK dgesdd).  Only R, t, s derived from it
 * are ever compared (U/V carry sign/ordering freedom). */
static void svd3(const double *H, double *U, double *S, double *V)
{
    double A[9], W[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(A, H, sizeof A);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < 3; ++i) {
                    a += A[i * 3 + p] * A[i * 3 + p];
                    b += A[i * 3 + q] * A[i * 3 + q];
                    g += A[i * 3 + p] * A[i * 3 + q];
                }
                if (g == 0.0) continue;
                if (fabs(g) <= 1e-300 || fabs(g) <= 2.3e-16 * sqrt(a * b)) continue;
                off = fmax(off, fabs(g) / sqrt(a * b));
                double zeta = (b - a) / (2.0 * g);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    double x = A[i * 3 + p], y = A[i * 3 + q];
                    A[i * 3 + p] = cs * x - sn * y; A[i * 3 + q] = sn * x + cs * y;
                    x = W[i * 3 + p]; y = W[i * 3 + q];
                    W[i * 3 + p] = cs * x - sn * y; W[i * 3 + q] = sn * x + cs * y;
                }
            }
        if (off == 0.0) break;
    }
    double s[3]; int ord[3] = { 0, 1, 2 };
    for (int j = 0; j < 3; ++j)
        s[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (s[ord[j]] > s[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Uc[3][3];
    for (int k = 0; k < 3; ++k) {
        int j = ord[k];
        S[k] = s[j];
        for (int i = 0; i < 3; ++i) { V[i * 3 + k] = W[i * 3 + j]; Uc[k][i] = A[i * 3 + j]; }
    }
    /* U: normalise column 1, Gram-Schmidt column 2, column 3 = +-(u1 x u2).  For (numerically)
     * null singular values the completion is arbitrary but orthonormal -- R is then not unique
     * in the reference either (LAPACK picks some completion). */
    const double rel = 1e-14;
    if (S[0] > 0.0) { for (int i = 0; i < 3; ++i) Uc[0][i] /= S[0]; }
    else { Uc[0][0] = 1; Uc[0][1] = 0; Uc[0][2] = 0; }
    if (S[1] > rel * S[0] && S[1] > 0.0) {
        double d = Uc[1][0] * Uc[0][0] + Uc[1][1] * Uc[0][1] + Uc[1][2] * Uc[0][2];
        for (int i = 0; i < 3; ++i) Uc[1][i] -= d * Uc[0][i];
    } else {
        int m = 0; if (fabs(Uc[0][1]) < fabs(Uc[0][m])) m = 1; if (fabs(Uc[0][2]) < fabs(Uc[0][m])) m = 2;
        double d = Uc[0][m];
        for (int i = 0; i < 3; ++i) Uc[1][i] = ((i == m) ? 1.0 : 0.0) - d * Uc[0][i];
    }
    {
        double nv = sqrt(Uc[1][0] * Uc[1][0] + Uc[1][1] * Uc[1][1] + Uc[1][2] * Uc[1][2]);
        for (int i = 0; i < 3; ++i) Uc[1][i] /= nv;
    }
    {
        double c[3] = { Uc[0][1] * Uc[1][2] - Uc[0][2] * Uc[1][1], Uc[0][2] * Uc[1][0] - Uc[0][0] * Uc[1][2],
                        Uc[0][0] * Uc[1][1] - Uc[0][1] * Uc[1][0] };
        double d = c[0] * Uc[2][0] + c[1] * Uc[2][1] + c[2] * Uc[2][2];
        double sg = (d < 0.0) ? -1.0 : 1.0;
        for (int i = 0; i < 3; ++i) Uc[2][i] = sg * c[i];
    }
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) U[i * 3 + k] = Uc[k][i];
}

/* status bits shared with the product C-ABI (include/gsf.h) */
#define ORC_SIM3_OK 0
#define ORC_SIM3_NONE 1        /* reference returned (None,None,None) */
#define ORC_SIM3_FLAG_VAR0 2   /* var_src < 1e-12 -> scale := 1 (ref :445-447) */
#define ORC_SIM3_FLAG_SMALL 4  /* scale <= 1e-6  -> scale := 1 (ref :449-450) */

/* ref :428-459 compute_sim3_transform.  src,dst: n x 3 row-major.
 * Returns ORC_SIM3_NONE for the reference's None-tuple, else 0 | flags. */
ORC_API int orc_umeyama(const double *src, const double *dst, int64_t n, double *R, double *t, double *scale)
{
    if (n < 3) return ORC_SIM3_NONE;                               /* :430 */
    double sc[3] = { 0, 0, 0 }, dc[3] = { 0, 0, 0 };
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { sc[k] += src[i * 3 + k]; dc[k] += dst[i * 3 + k]; }
    for (int k = 0; k < 3; ++k) { sc[k] /= (double)n; dc[k] /= (double)n; }     /* :436 */
    double H[9] = { 0 }, ssq = 0.0;
    for (int64_t i = 0; i < n; ++i) {                              /* :437-438, :443 */
        double a[3], b[3];
        for (int k = 0; k < 3; ++k) { a[k] = src[i * 3 + k] - sc[k]; b[k] = dst[i * 3 + k] - dc[k]; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r * 3 + c] += a[r] * b[c];
        ssq += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    }
    for (int k = 0; k < 9; ++k) if (isnan(H[k]) || isinf(H[k])) return ORC_SIM3_NONE; /* LinAlgError :452 */
    double U[9], S[3], V[9], UT[9];
    svd3(H, U, S, V);                                              /* :439 */
    mat_T(U, UT, 3, 3);
    mat_mul(V, UT, R, 3, 3, 3);                                    /* :440  R = Vt.T @ U.T */
    if (det3(R) < 0.0) {                                           /* :441-442 */
        V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8];                  /* Vt[-1,:] *= -1  == last column of V */
        mat_mul(V, UT, R, 3, 3, 3);
    }
    double var_src = ssq / (double)n;                              /* :443 */
    double tr = S[0] * 1.0 + S[1] * 1.0 + S[2] * det3(R);          /* :444 (Q12) */
    int flags = 0; double s;
    if (var_src < 1e-12) { s = 1.0; flags |= ORC_SIM3_FLAG_VAR0; } /* :445-447 */
    else {
        s = tr / ((double)n * var_src);                            /* :449 */
        if (s <= 1e-6) { s = 1.0; flags |= ORC_SIM3_FLAG_SMALL; }  /* :450 */
    }
    double Rs[3];
    for (int r = 0; r < 3; ++r) Rs[r] = R[r * 3] * sc[0] + R[r * 3 + 1] * sc[1] + R[r * 3 + 2] * sc[2];
    for (int r = 0; r < 3; ++r) t[r] = dc[r] - s * Rs[r];          /* :451 */
    *scale = s;
    return flags;
}

/* ref :389-426 compute_sim3_transform_robust.  sample_idx: trials x min_samples
 * indices drawn by the caller with the reference's RNG call (:405).
 * inlier_mask (n bytes) receives the best mask.  n_inliers_out may be NULL. */
ORC_API int orc_sim3_ransac(const double *src, const double *dst, int64_t n, const int32_t *sample_idx,
                            int trials, int min_samples, double thr, int min_inliers_needed,
                            double *R, double *t, double *scale, uint8_t *inlier_mask, int64_t *n_inliers_out)
{
    if (n_inliers_out) *n_inliers_out = -1;
    if (n < min_samples) return ORC_SIM3_NONE;                     /* :395-397 */
    if (min_samples > 64) return ORC_SIM3_NONE;
    int64_t best = -1;
    uint8_t *mask = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    double ss[64 * 3], ds[64 * 3];
    for (int tr = 0; tr < trials; ++tr) {                          /* :404 */
        const int32_t *idx = sample_idx + (size_t)tr * min_samples;
        for (int k = 0; k < min_samples; ++k)
            for (int c = 0; c < 3; ++c) { ss[k * 3 + c] = src[idx[k] * 3 + c]; ds[k * 3 + c] = dst[idx[k] * 3 + c]; }
        double Rt[9], tt[3], st;
        int st_code = orc_umeyama(ss, ds, min_samples, Rt, tt, &st);           /* :407 */
        if (st_code == ORC_SIM3_NONE) continue;                    /* :408 */
        int64_t cnt = 0;
        for (int64_t i = 0; i < n; ++i) {                          /* :409-412 */
            double y[3];
            for (int r = 0; r < 3; ++r)
                y[r] = st * (src[i * 3] * Rt[r * 3] + src[i * 3 + 1] * Rt[r * 3 + 1] + src[i * 3 + 2] * Rt[r * 3 + 2]) + tt[r];
            double dx = y[0] - dst[i * 3], dy = y[1] - dst[i * 3 + 1], dz = y[2] - dst[i * 3 + 2];
            double res = sqrt(dx * dx + dy * dy + dz * dz);
            mask[i] = (res < thr) ? 1 : 0;
            cnt += mask[i];
        }
        if (cnt > best) { best = cnt; memcpy(inlier_mask, mask, (size_t)n); }   /* :413-414 strict > */
    }
    free(mask);
    if (n_inliers_out) *n_inliers_out = best;
    if (best < min_inliers_needed) return ORC_SIM3_NONE;           /* :416-418 */
    double *si = (double *)malloc(sizeof(double) * 3 * (size_t)best);
    double *di = (double *)malloc(sizeof(double) * 3 * (size_t)best);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        if (inlier_mask[i]) {
            for (int c = 0; c < 3; ++c) { si[m * 3 + c] = src[i * 3 + c]; di[m * 3 + c] = dst[i * 3 + c]; }
            ++m;
        }
    int rc = orc_umeyama(si, di, m, R, t, scale);                  /* :420-421 */
    free(si); free(di);
    return rc;
}

/* ref :461-467 transform_trajectory.  Returns number of invalid (zero-norm)
 * quaternions met (the reference would raise ValueError on those). */
ORC_API int orc_transform_trajectory(const double *pos, const double *quat, int64_t n, const double *R,
                                     const double *t, double s, double *pos_out, double *quat_out)
{
    double qR[4]; int bad = 0;
    quat_from_mat(R, qR);                                          /* :465 */
    for (int64_t i = 0; i < n; ++i) {
        const double *p = pos + i * 3;
        for (int r = 0; r < 3; ++r)                                /* :464  s*(pos @ R.T)+t */
            pos_out[i * 3 + r] = s * (p[0] * R[r * 3] + p[1] * R[r * 3 + 1] + p[2] * R[r * 3 + 2]) + t[r];
        double qn[4];
        if (quat_from(quat + i * 4, qn)) { ++bad; for (int k = 0; k < 4; ++k) quat_out[i * 4 + k] = NAN; continue; }
        quat_mul(qR, qn, quat_out + i * 4);                        /* :466 */
    }
    return bad;
}

/* ------------------------------------------------------------------------ */
/* EKF (dense 7x7, as the reference)                                         */
/* ------------------------------------------------------------------------ */
typedef struct {
    double P0[7], Qps[7], Rm[3];          /* CONFIG['ekf'] diag lists, ref :25-27 */
    int32_t transition_steps;             /* ref :28 (dead under apply_ekf_correction, Q6) */
    double yaw_rate_thr_deg;              /* ref :68 */
    int32_t sharp_turn_steps;             /* ref :69 */
} orc_config;

typedef struct {
    double state[7], cov[49], Qps[49], R[9];
    int gnss_prev;                        /* -1 None, 0 False, 1 True */
    double weight, weight_delta;
    int original_steps, current_steps;
    double last_pred[7];
} orc_ekf;

/* ref :680-695 */
static void ekf_init(orc_ekf *f, const double *p0, const double *q0, const orc_config *c)
{
    memset(f, 0, sizeof *f);
    for (int i = 0; i < 3; ++i) f->state[i] = p0[i];
    for (int i = 0; i < 4; ++i) f->state[3 + i] = q0[i];
    ekf_normalize_quat(f->state + 3);
    for (int i = 0; i < 7; ++i) { f->cov[i * 8] = c->P0[i]; f->Qps[i * 8] = c->Qps[i]; }
    for (int i = 0; i < 3; ++i) f->R[i * 4] = c->Rm[i];
    f->gnss_prev = -1; f->weight = 0.0;
    f->original_steps = c->transition_steps > 1 ? c->transition_steps : 1;
    f->current_steps = f->original_steps;
    f->weight_delta = 1.0;
    memcpy(f->last_pred, f->state, sizeof f->last_pred);
}

/* ref :702-715 */
static void ekf_predict(const orc_ekf *f, const double *dpos_local, const double *dquat, double dt,
                        double *ps, double *pc)
{
    double qn[4], dqn[4], rp[3], pq[4];
    quat_from(f->state + 3, qn);                                   /* :704 */
    quat_from(dquat, dqn);                                         /* :706 */
    quat_apply(qn, dpos_local, rp);
    for (int i = 0; i < 3; ++i) ps[i] = f->state[i] + rp[i];       /* :707 */
    quat_mul(qn, dqn, pq);                                         /* :708 */
    ekf_normalize_quat(pq);                                        /* :709 */
    for (int i = 0; i < 4; ++i) ps[3 + i] = pq[i];
    double dta = fmax(fabs(dt), 1e-6);                             /* :711 */
    for (int i = 0; i < 49; ++i) pc[i] = f->cov[i] + f->Qps[i] * dta;          /* :712-713 */
    mat_sym(pc, 7);                                                /* :714 */
}

/* ref :717-734; returns 0 ok, 1 if (None,None) */
static int ekf_update(const orc_ekf *f, const double *ps, const double *pc, const double *z, double *us, double *uc)
{
    if (isnan(z[0]) || isnan(z[1]) || isnan(z[2])) return 1;       /* :719 */
    double H[21] = { 0 }; H[0] = 1; H[7 + 1] = 1; H[14 + 2] = 1;   /* :720 */
    double HT[21]; mat_T(H, HT, 3, 7);
    double y[3] = { z[0] - ps[0], z[1] - ps[1], z[2] - ps[2] };    /* :722 */
    double HP[21], S[9], Sinv[9];
    mat_mul(H, pc, HP, 3, 7, 7);
    mat_mul(HP, HT, S, 3, 7, 3);
    for (int i = 0; i < 9; ++i) S[i] += f->R[i];                   /* :723 */
    mat_sym(S, 3);                                                 /* :724 */
    if (mat_inv(S, Sinv, 3)) return 1;                             /* :725-726 (pinv fallback unreachable for SPD S) */
    double PHT[21], K[21];
    mat_mul(pc, HT, PHT, 7, 7, 3);
    mat_mul(PHT, Sinv, K, 7, 3, 3);                                /* :727 */
    for (int i = 0; i < 7; ++i)
        us[i] = ps[i] + (K[i * 3] * y[0] + K[i * 3 + 1] * y[1] + K[i * 3 + 2] * y[2]);   /* :728 */
    ekf_normalize_quat(us + 3);                                    /* :729 */
    double KH[49], IKH[49], IKHT[49], T1[49], T2[49], KR[21], KT[21], KRK[49];
    mat_mul(K, H, KH, 7, 3, 7);
    for (int i = 0; i < 49; ++i) IKH[i] = ((i % 8 == 0) ? 1.0 : 0.0) - KH[i];
    mat_T(IKH, IKHT, 7, 7);
    mat_mul(IKH, pc, T1, 7, 7, 7);
    mat_mul(T1, IKHT, T2, 7, 7, 7);
    mat_mul(K, f->R, KR, 7, 3, 3);
    mat_T(K, KT, 7, 3);
    mat_mul(KR, KT, KRK, 7, 3, 7);
    for (int i = 0; i < 49; ++i) uc[i] = T2[i] + KRK[i];           /* :731 */
    mat_sym(uc, 7);                                                /* :732 */
    return 0;
}

/* ref :736-772.  has_meas = (gps_measurement is not None). override_steps<0 -> None. */
static void ekf_process_step(orc_ekf *f, const double *dpos_local, const double *dquat, const double *z, int has_meas,
                             int avail, double dt, int override_steps, double *pred_s, double *pred_c)
{
    int eff = override_steps >= 0 ? override_steps : f->current_steps;           /* :742 */
    f->weight_delta = eff > 0 ? 1.0 / (double)eff : 1.0;                          /* :743 */
    ekf_predict(f, dpos_local, dquat, dt, pred_s, pred_c);                        /* :745 */
    memcpy(f->last_pred, pred_s, sizeof f->last_pred);                            /* :746 */
    double us[7], uc[49]; int ok = 0;
    if (avail && has_meas) ok = (ekf_update(f, pred_s, pred_c, z, us, uc) == 0);  /* :749-751 */
    int just_rec = avail && (f->gnss_prev == 0);                                  /* :753 */
    if (avail) {                                                                  /* :754-758 */
        if (just_rec || eff == 0) f->weight = (eff == 0) ? 1.0 : f->weight_delta;
        else if (f->weight < 1.0) f->weight = fmin(1.0, f->weight + f->weight_delta);
    } else f->weight = 0.0;
    if (avail && ok) {                                                            /* :761 */
        if (f->weight < 1.0 && eff > 0) {                                         /* :762-767 */
            double w = f->weight;
            for (int i = 0; i < 3; ++i) f->state[i] = (1.0 - w) * f->last_pred[i] + w * us[i];
            double qb[4]; orc_quaternion_nlerp(f->last_pred + 3, us + 3, w, qb);
            for (int i = 0; i < 4; ++i) f->state[3 + i] = qb[i];
            memcpy(f->cov, uc, sizeof f->cov);
        } else { memcpy(f->state, us, sizeof us); memcpy(f->cov, uc, sizeof uc); } /* :768 */
    } else { memcpy(f->state, pred_s, sizeof f->state); memcpy(f->cov, pred_c, sizeof f->cov); } /* :760 */
    f->gnss_prev = avail ? 1 : 0;                                                 /* :771 */
}

/* Python-surface wrapper around one ExtendedKalmanFilter.process_step call,
 * with explicit state in/out (for golden tests of the class, ref :679-772).
 * io_state[7], io_cov[49], io_gnss_prev (-1/0/1), io_weight, current_steps. */
ORC_API void orc_ekf_process_step(const orc_config *cfg, double *io_state, double *io_cov, int *io_gnss_prev,
                                  double *io_weight, int current_steps, const double *dpos_local,
                                  const double *dquat, const double *z, int has_meas, int avail, double dt,
                                  int override_steps, double *pred_state, double *pred_cov)
{
    orc_ekf f; double p0[3] = { 0, 0, 0 }, q0[4] = { 0, 0, 0, 1 };
    ekf_init(&f, p0, q0, cfg);
    memcpy(f.state, io_state, sizeof f.state); memcpy(f.cov, io_cov, sizeof f.cov);
    f.gnss_prev = *io_gnss_prev; f.weight = *io_weight; f.current_steps = current_steps;
    ekf_process_step(&f, dpos_local, dquat, z, has_meas, avail, dt, override_steps, pred_state, pred_cov);
    memcpy(io_state, f.state, sizeof f.state); memcpy(io_cov, f.cov, sizeof f.cov);
    *io_gnss_prev = f.gnss_prev; *io_weight = f.weight;
}

/* ref :777-803 rts_smoother_segment.  Arrays: L x 7 states, L x 49 covs. */
ORC_API void orc_rts_segment(const double *xf, const double *Pf, const double *xp, const double *Pp, int64_t L,
                             double *xs, double *Ps)
{
    if (L <= 0) return;
    memcpy(xs + (L - 1) * 7, xf + (L - 1) * 7, 7 * sizeof(double));               /* :782 */
    memcpy(Ps + (L - 1) * 49, Pf + (L - 1) * 49, 49 * sizeof(double));
    for (int64_t k = L - 2; k >= 0; --k) {                                        /* :784 */
        const double *Pk = Pf + k * 49, *Pp1 = Pp + (k + 1) * 49;
        double inv[49], A[49], AT[49];
        if (mat_inv(Pp1, inv, 7)) {                                               /* :787-796 */
            memcpy(xs + k * 7, xf + k * 7, 7 * sizeof(double));
            memcpy(Ps + k * 49, Pk, 49 * sizeof(double));
            continue;
        }
        mat_mul(Pk, inv, A, 7, 7, 7);                                             /* :789 (F = I) */
        double d[7];
        for (int i = 0; i < 7; ++i) d[i] = xs[(k + 1) * 7 + i] - xp[(k + 1) * 7 + i];
        for (int i = 0; i < 7; ++i) {                                             /* :798 */
            double a = 0.0;
            for (int j = 0; j < 7; ++j) a += A[i * 7 + j] * d[j];
            xs[k * 7 + i] = xf[k * 7 + i] + a;
        }
        ekf_normalize_quat(xs + k * 7 + 3);                                       /* :799 */
        double D[49], T1[49], T2[49];
        for (int i = 0; i < 49; ++i) D[i] = Ps[(k + 1) * 49 + i] - Pp1[i];
        mat_T(A, AT, 7, 7);
        mat_mul(A, D, T1, 7, 7, 7);
        mat_mul(T1, AT, T2, 7, 7, 7);
        for (int i = 0; i < 49; ++i) Ps[k * 49 + i] = Pk[i] + T2[i];              /* :801 */
        mat_sym(Ps + k * 49, 7);                                                  /* :802 */
    }
}

/* ref :808-826 is_sharp_turn_in_segment; max_rate_out may be NULL */
ORC_API int orc_is_sharp_turn(const double *quats, const double *stamps, int64_t L, double thr_rad_per_s,
                              double *max_rate_out)
{
    if (max_rate_out) *max_rate_out = 0.0;
    if (L < 2) return 0;                                                          /* :812 */
    double maxr = 0.0;
    for (int64_t i = 1; i < L; ++i) {
        double t1 = stamps[i - 1], t2 = stamps[i];
        if (t2 <= t1) continue;                                                   /* :817 */
        double a[4], b[4];
        if (quat_from(quats + (i - 1) * 4, a) || quat_from(quats + i * 4, b)) return 1;   /* :821 */
        double y1 = quat_yaw_zyx(a), y2 = quat_yaw_zyx(b);                        /* :819-820 */
        double dy = atan2(sin(y2 - y1), cos(y2 - y1));                            /* :822 */
        double r = fabs(dy / (t2 - t1));                                          /* :823 */
        if (r > maxr) maxr = r;
    }
    if (max_rate_out) *max_rate_out = maxr;
    return maxr > thr_rad_per_s;                                                  /* :826 */
}

/* status bits of the fused-trajectory result (shared with include/gsf.h) */
#define ORC_ST_HAD_OUTAGE 1
#define ORC_ST_RTS_APPLIED 2
#define ORC_ST_SHARP_TURN 4
#define ORC_ST_ENDED_IN_OUTAGE 8
#define ORC_ST_BAD_QUAT 16

/* ref :831-935 apply_ekf_correction, AFTER its dynamic_time_alignment call
 * (:847): `aligned` (n x 3, NaN allowed) and `valid` (n bytes) are that call's
 * outputs.  sim3_pos0/sim3_quat0 = row 0 of the Sim3-aligned trajectory (Q3).
 * Returns status bits.  */
ORC_API int orc_apply_ekf_correction(const double *ts, const double *pos, const double *quat,
                                     const double *aligned, const uint8_t *valid, int64_t n,
                                     const double *sim3_pos0, const double *sim3_quat0, const orc_config *cfg,
                                     double *pos_out, double *quat_out)
{
    if (n <= 0) return 0;                                                         /* :835 */
    int status = 0;
    orc_ekf f; ekf_init(&f, sim3_pos0, sim3_quat0, cfg);                          /* :842 */
    f.current_steps = 0;                                                          /* :845 */
    f.gnss_prev = valid[0] ? 1 : 0;                                               /* :848 */
    double *xf = (double *)malloc(sizeof(double) * 7 * (size_t)n), *xp = (double *)malloc(sizeof(double) * 7 * (size_t)n);
    double *Pf = (double *)malloc(sizeof(double) * 49 * (size_t)n), *Pp = (double *)malloc(sizeof(double) * 49 * (size_t)n);
    memcpy(xf, f.state, 7 * sizeof(double)); memcpy(xp, f.state, 7 * sizeof(double));   /* :852-853 */
    memcpy(Pf, f.cov, 49 * sizeof(double)); memcpy(Pp, f.cov, 49 * sizeof(double));
    for (int i = 0; i < 3; ++i) pos_out[i] = f.state[i];                          /* :856 */
    for (int i = 0; i < 4; ++i) quat_out[i] = f.state[3 + i];
    double last_time = ts[0];
    int in_outage = !f.gnss_prev;                                                 /* :861 */
    int64_t ostart = in_outage ? 0 : -1;                                          /* :862 */
    if (in_outage) status |= ORC_ST_HAD_OUTAGE;
    for (int64_t i = 1; i < n; ++i) {                                             /* :864 */
        double now = ts[i], dt = fmax(1e-6, now - last_time);                     /* :865 */
        double dpl[3], dq[4];
        if (orc_relative_pose(pos + (i - 1) * 3, quat + (i - 1) * 4, pos + i * 3, quat + i * 4, dpl, dq))
            status |= ORC_ST_BAD_QUAT;                                            /* :866 */
        int avail = valid[i] ? 1 : 0;                                             /* :867 */
        const double *z = aligned + i * 3;
        int has_meas = avail && !(isnan(z[0]) || isnan(z[1]) || isnan(z[2]));    /* :868 */
        if (!has_meas) avail = 0;                                                 /* :869 */
        int perform_rts = 1, iter_steps = 0;                                      /* :872-873 */
        if (!avail && !in_outage) {                                               /* :875-877 */
            in_outage = 1; ostart = i; status |= ORC_ST_HAD_OUTAGE;
        } else if (avail && in_outage) {                                          /* :879 */
            int64_t L = i - ostart;                                               /* range(ostart, i) */
            if (L >= 2) {                                                         /* :882 */
                double thr = cfg->yaw_rate_thr_deg * (M_PI / 180.0);              /* np.deg2rad :886 */
                if (orc_is_sharp_turn(quat + ostart * 4, ts + ostart, L, thr, NULL)) {
                    perform_rts = 0; iter_steps = cfg->sharp_turn_steps;          /* :888-889 */
                    status |= ORC_ST_SHARP_TURN;
                }
            }
        }
        int ovr = (avail && in_outage) ? iter_steps : f.current_steps;            /* :899 */
        ekf_process_step(&f, dpl, dq, z, has_meas, avail, dt, ovr, xp + i * 7, Pp + i * 49);   /* :897 */
        memcpy(xf + i * 7, f.state, 7 * sizeof(double));                          /* :902-903 */
        memcpy(Pf + i * 49, f.cov, 49 * sizeof(double));
        for (int k = 0; k < 3; ++k) pos_out[i * 3 + k] = f.state[k];              /* :904 */
        for (int k = 0; k < 4; ++k) quat_out[i * 4 + k] = f.state[3 + k];
        if (avail && in_outage) {                                                 /* :906 */
            if (perform_rts) {
                int64_t L = i - ostart + 1;                                       /* :909 */
                if (L > 1) {
                    double *xs = (double *)malloc(sizeof(double) * 7 * (size_t)L);
                    double *Ps = (double *)malloc(sizeof(double) * 49 * (size_t)L);
                    orc_rts_segment(xf + ostart * 7, Pf + ostart * 49, xp + ostart * 7, Pp + ostart * 49, L, xs, Ps);  /* :917 */
                    for (int64_t k = 0; k < L; ++k) {                             /* :918-922 */
                        int64_t o = ostart + k;
                        for (int c = 0; c < 3; ++c) pos_out[o * 3 + c] = xs[k * 7 + c];
                        for (int c = 0; c < 4; ++c) quat_out[o * 4 + c] = xs[k * 7 + 3 + c];
                        memcpy(xf + o * 7, xs + k * 7, 7 * sizeof(double));
                    }
                    free(xs); free(Ps);
                    status |= ORC_ST_RTS_APPLIED;
                }
            }
            in_outage = 0; ostart = -1; f.current_steps = 0;                      /* :926-928 */
        }
        last_time = now;                                                          /* :930 */
    }
    if (in_outage && ostart != -1) status |= ORC_ST_ENDED_IN_OUTAGE;              /* :932 */
    free(xf); free(xp); free(Pf); free(Pp);
    return status;
}

/* Batch driver used for the cpu_baseline timing and for batch parity tests.
 * Trajectory-major AoS: ts[B*n], pos[B*n*3], quat[B*n*4], aligned[B*n*3],
 * valid[B*n]; init_pos[B*3], init_quat[B*4]. */
ORC_API void orc_fuse_batch(const double *ts, const double *pos, const double *quat, const double *aligned,
                            const uint8_t *valid, int64_t B, int64_t n, const double *init_pos,
                            const double *init_quat, const orc_config *cfg, double *pos_out, double *quat_out,
                            int32_t *status)
{
    for (int64_t b = 0; b < B; ++b) {
        int st = orc_apply_ekf_correction(ts + b * n, pos + b * n * 3, quat + b * n * 4, aligned + b * n * 3,
                                          valid + b * n, n, init_pos + b * 3, init_quat + b * 4, cfg,
                                          pos_out + b * n * 3, quat_out + b * n * 4);
        if (status) status[b] = st;
    }
}


/* Which time-synchronised rows feed the global Sim3: main_process_gui, ref :973-998.
 * valid[i] != 0 marks the rows of valid_indices_all_sim3 (:973).  idx receives the chosen row indices (room for n), *branch
 * 0 = the timed subset (:996), 1 = the whole first segment (:993-995), 2 = all valid rows (:984-986).
 * Returns the number of rows, or -1 where the reference raises ValueError (:975, :997). */
ORC_API int64_t orc_pick_sim3_rows(const double *ts, const uint8_t *valid, int64_t n, int min_samples, double max_gap,
                                   double max_dur, int64_t *idx, int *branch)
{
    int64_t *vi = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t nv = 0, m = 0;
    if (branch) *branch = 2;
    for (int64_t i = 0; i < n; ++i) if (valid[i]) vi[nv++] = i;                      /* :973 np.where */
    if (nv < min_samples) { free(vi); return -1; }                                   /* :974-975 */
    if (nv > 0) {                                                                    /* :978 */
        int64_t end = nv;                                                            /* :981 no gap: len(valid_indices) */
        for (int64_t k = 0; k + 1 < nv; ++k)                                         /* :979-980 np.diff > threshold, first hit */
            if (ts[vi[k + 1]] - ts[vi[k]] > max_gap) { end = k; break; }             /* :981 first_gap_idx[0] -- NOT k + 1 */
        if (end < min_samples) {                                                     /* :983 first segment too short */
            for (int64_t k = 0; k < nv; ++k) idx[m++] = vi[k];                       /* :984 all valid rows */
            if (branch) *branch = 2;
        } else {
            const double t_lim = ts[vi[0]] + max_dur;                                /* :988-989 */
            int64_t nt = 0;
            for (int64_t k = 0; k < end; ++k) if (ts[vi[k]] <= t_lim) ++nt;          /* :990-992 */
            if (nt < min_samples) {                                                  /* :993-995 */
                for (int64_t k = 0; k < end; ++k) idx[m++] = vi[k];
                if (branch) *branch = 1;
            } else {
                for (int64_t k = 0; k < end; ++k) if (ts[vi[k]] <= t_lim) idx[m++] = vi[k];   /* :996 */
                if (branch) *branch = 0;
            }
        }
    }
    free(vi);
    if (m < min_samples) return -1;                                                  /* :997 */
    return m;
}

/* Steps 3-5 of main_process_gui (ref :1002-1010) with the plain fit of :428 for B equal-length trajectories:
 * compute_sim3_transform on the chosen rows -> transform_trajectory (only row 0 is consumed) ->
 * apply_ekf_correction.  fit_rows 0: every row with valid, non-NaN GNSS; 1: the reference's choice among those rows
 * (orc_pick_sim3_rows, ref :973-998).  status = ekf bits | (sim3 status << 8); a None fit leaves NaN outputs; where the
 * reference raises ValueError (:975, :997) the sim3 status is ORC_SIM3_NONE | ORC_SIM3_FLAG_FEW_ROWS. */
#define ORC_SIM3_FLAG_FEW_ROWS 32
#define ORC_SIM3_FLAG_ROWS_ALL 64
#define ORC_SIM3_FLAG_ROWS_SEGMENT 128
ORC_API void orc_fuse_pipeline_rows_batch(const double *ts, const double *pos, const double *quat, const double *aligned,
                                          const uint8_t *valid, int64_t B, int64_t n, const orc_config *cfg, int fit_rows,
                                          int min_samples, double max_gap, double max_dur, double *Rout, double *tout,
                                          double *sout, double *pos_out, double *quat_out, int32_t *status, int32_t *n_rows)
{
    double *src = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    double *dst = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    uint8_t *ok = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t b = 0; b < B; ++b) {
        const double *p = pos + b * n * 3, *z = aligned + b * n * 3, *q = quat + b * n * 4;
        const uint8_t *v = valid + b * n;
        int64_t m = 0;
        int rows_flag = 0;
        for (int64_t i = 0; i < n; ++i) ok[i] = (uint8_t)(v[i] && !(isnan(z[i * 3]) || isnan(z[i * 3 + 1]) || isnan(z[i * 3 + 2])));
        if (fit_rows) {
            int br = 0;
            m = orc_pick_sim3_rows(ts + b * n, ok, n, min_samples, max_gap, max_dur, idx, &br);
            rows_flag = m < 0 ? ORC_SIM3_FLAG_FEW_ROWS : (br == 2 ? ORC_SIM3_FLAG_ROWS_ALL : (br == 1 ? ORC_SIM3_FLAG_ROWS_SEGMENT : 0));
        } else {
            for (int64_t i = 0; i < n; ++i) if (ok[i]) idx[m++] = i;
        }
        if (n_rows) n_rows[b] = (int32_t)m;
        for (int64_t k = 0; k < m; ++k)
            for (int c = 0; c < 3; ++c) { src[k * 3 + c] = p[idx[k] * 3 + c]; dst[k * 3 + c] = z[idx[k] * 3 + c]; }
        double R[9], t[3], s = NAN;
        int fit = m < 0 ? ORC_SIM3_NONE : orc_umeyama(src, dst, m, R, t, &s);
        fit |= rows_flag;
        double p0[3], q0[4];
        int bad = 1;
        if (!(fit & ORC_SIM3_NONE) && n > 0) bad = orc_transform_trajectory(p, q, 1, R, t, s, p0, q0);
        if ((fit & ORC_SIM3_NONE) || bad) {
            for (int k = 0; k < 9; ++k) Rout[b * 9 + k] = NAN;
            for (int k = 0; k < 3; ++k) tout[b * 3 + k] = NAN;
            sout[b] = NAN;
            for (int64_t i = 0; i < n * 3; ++i) pos_out[b * n * 3 + i] = NAN;
            for (int64_t i = 0; i < n * 4; ++i) quat_out[b * n * 4 + i] = NAN;
            if (status) status[b] = ((fit & ORC_SIM3_NONE) ? ((ORC_SIM3_NONE | (fit & ORC_SIM3_FLAG_FEW_ROWS)) << 8) : 0) | (bad && !(fit & ORC_SIM3_NONE) ? ORC_ST_BAD_QUAT : 0);
            continue;
        }
        for (int k = 0; k < 9; ++k) Rout[b * 9 + k] = R[k];
        for (int k = 0; k < 3; ++k) tout[b * 3 + k] = t[k];
        sout[b] = s;
        int st = orc_apply_ekf_correction(ts + b * n, p, q, z, v, n, p0, q0, cfg, pos_out + b * n * 3, quat_out + b * n * 4);
        if (status) status[b] = st | (fit << 8);
    }
    free(src); free(dst); free(ok); free(idx);
}

/* the same on every row with valid, non-NaN GNSS (fit_rows = 0): the operator SURVEY 8(b)/(d) defined for the batch configs */
ORC_API void orc_fuse_pipeline_batch(const double *ts, const double *pos, const double *quat, const double *aligned,
                                     const uint8_t *valid, int64_t B, int64_t n, const orc_config *cfg, double *Rout,
                                     double *tout, double *sout, double *pos_out, double *quat_out, int32_t *status)
{
    orc_fuse_pipeline_rows_batch(ts, pos, quat, aligned, valid, B, n, cfg, 0, 0, 0.0, 0.0, Rout, tout, sout, pos_out, quat_out, status, NULL);
}

/* ------------------------------------------------------------------------ */
/* time alignment  (ref :301-387)  -- "next-1" row of SURVEY 8(f)            */
/* ------------------------------------------------------------------------ */
/* ref :301-323: correlates two linspaces -> exactly 0.0 (Q2 / KAT-5); we keep
 * only the early-outs that also return 0.0. */
ORC_API double orc_estimate_time_offset(const double *slam_t, int64_t ns, const double *gps_t, int64_t ng, int max_samples)
{
    (void)slam_t; (void)gps_t; (void)ns; (void)ng; (void)max_samples;
    return 0.0;
}

/* not-a-knot cubic spline through (x[0..m-1], y[0..m-1][3]) evaluated at xq,
 * == scipy interp1d(kind='cubic') == make_interp_spline(k=3) (ref :368).
 * Piecewise-polynomial form: solve for the knot second derivatives M. */
static void spline_notaknot_eval(const double *x, const double *y, int64_t m, const double *xq, int64_t nq,
                                 const int64_t *qidx, double *out)
{
    /* dense solve per component (m is small in tests; O(m) memory, tridiagonal + 2 corner terms) */
    double *h = (double *)malloc(sizeof(double) * (size_t)m);
    double *M = (double *)malloc(sizeof(double) * (size_t)m * 3);
    for (int64_t i = 0; i + 1 < m; ++i) h[i] = x[i + 1] - x[i];
    /* unknowns M_0..M_{m-1}; interior eqs i=1..m-2:
         h[i-1] M[i-1] + 2 (h[i-1]+h[i]) M[i] + h[i] M[i+1] = 6 (d[i] - d[i-1])
       not-a-knot: third derivative continuous at x[1] and x[m-2]:
         (M1-M0)/h0 = (M2-M1)/h1  ;  (M[m-2]-M[m-3])/h[m-3] = (M[m-1]-M[m-2])/h[m-2]
       eliminate M0 and M[m-1] into the first / last interior equations -> tridiagonal of size m-2. */
    int64_t k = m - 2;
    double *a = (double *)malloc(sizeof(double) * (size_t)k), *b = (double *)malloc(sizeof(double) * (size_t)k),
           *c = (double *)malloc(sizeof(double) * (size_t)k), *r = (double *)malloc(sizeof(double) * (size_t)k);
    for (int comp = 0; comp < 3; ++comp) {
        for (int64_t i = 1; i <= m - 2; ++i) {
            double di = (y[(i + 1) * 3 + comp] - y[i * 3 + comp]) / h[i];
            double dm = (y[i * 3 + comp] - y[(i - 1) * 3 + comp]) / h[i - 1];
            a[i - 1] = h[i - 1]; b[i - 1] = 2.0 * (h[i - 1] + h[i]); c[i - 1] = h[i]; r[i - 1] = 6.0 * (di - dm);
        }
        /* M0 = M1 (1 + h0/h1) - M2 h0/h1 */
        double r0 = h[0] / h[1];
        b[0] += a[0] * (1.0 + r0); c[0] -= a[0] * r0; a[0] = 0.0;
        /* M[m-1] = M[m-2] (1 + h[m-2]/h[m-3]) - M[m-3] h[m-2]/h[m-3] */
        double r1 = h[m - 2] / h[m - 3];
        b[k - 1] += c[k - 1] * (1.0 + r1); a[k - 1] -= c[k - 1] * r1; c[k - 1] = 0.0;
        /* m == 4 -> k == 2 : both corrections touch a 2x2 system; Thomas still applies */
        for (int64_t i = 1; i < k; ++i) {
            double w = a[i] / b[i - 1];
            b[i] -= w * c[i - 1]; r[i] -= w * r[i - 1];
        }
        double *Mi = M + (size_t)comp * (size_t)m;
        Mi[k] = r[k - 1] / b[k - 1];
        for (int64_t i = k - 2; i >= 0; --i) Mi[i + 1] = (r[i] - c[i] * Mi[i + 2]) / b[i];
        Mi[0] = Mi[1] * (1.0 + r0) - Mi[2] * r0;
        Mi[m - 1] = Mi[m - 2] * (1.0 + r1) - Mi[m - 3] * r1;
    }
    for (int64_t q = 0; q < nq; ++q) {
        double t = xq[q];
        int64_t lo = 0, hi = m - 1;           /* interval [x[lo], x[lo+1]] containing t */
        while (hi - lo > 1) { int64_t mid = (lo + hi) / 2; if (x[mid] <= t) lo = mid; else hi = mid; }
        double hh = h[lo], A = (x[lo + 1] - t) / hh, Bc = (t - x[lo]) / hh;
        for (int comp = 0; comp < 3; ++comp) {
            const double *Mi = M + (size_t)comp * (size_t)m;
            double v = A * y[lo * 3 + comp] + Bc * y[(lo + 1) * 3 + comp] +
                       ((A * A * A - A) * Mi[lo] + (Bc * Bc * Bc - Bc) * Mi[lo + 1]) * hh * hh / 6.0;
            out[qidx[q] * 3 + comp] = v;
        }
    }
    free(h); free(M); free(a); free(b); free(c); free(r);
}

static int cmp_idx_by_time(const void *pa, const void *pb, void *ctx)
{
    const double *t = (const double *)ctx;
    int64_t a = *(const int64_t *)pa, b = *(const int64_t *)pb;
    if (t[a] < t[b]) return -1;
    if (t[a] > t[b]) return 1;
    return (a < b) ? -1 : (a > b);       /* stable, like the merge the reference relies on for np.unique(return_index) */
}

/* ref :325-387 dynamic_time_alignment.  aligned: ns x 3 (NaN-filled), valid: ns bytes. */
ORC_API void orc_dynamic_time_alignment(const double *slam_t, int64_t ns, const double *gps_t, const double *gps_p,
                                        int64_t ng, int max_samples, double max_gap, double *aligned, uint8_t *valid)
{
    for (int64_t i = 0; i < ns * 3; ++i) aligned[i] = NAN;                        /* :331 */
    for (int64_t i = 0; i < ns; ++i) valid[i] = 0;
    if (ns == 0 || ng < 2) return;                                                /* :332-334 */
    double offset = orc_estimate_time_offset(slam_t, ns, gps_t, ng, max_samples); /* :336 */
    int64_t *ord = (int64_t *)malloc(sizeof(int64_t) * (size_t)ng);
    double *tt = (double *)malloc(sizeof(double) * (size_t)ng);
    for (int64_t i = 0; i < ng; ++i) { ord[i] = i; tt[i] = gps_t[i] + offset; }   /* :337 */
    qsort_r(ord, (size_t)ng, sizeof(int64_t), cmp_idx_by_time, tt);               /* :339 argsort */
    /* np.unique(..., return_index=True): first occurrence of each distinct stamp (:341) */
    double *ut = (double *)malloc(sizeof(double) * (size_t)ng);
    double *up = (double *)malloc(sizeof(double) * 3 * (size_t)ng);
    int64_t nu = 0;
    for (int64_t i = 0; i < ng; ++i) {
        double t = tt[ord[i]];
        if (nu > 0 && t == ut[nu - 1]) continue;
        if (isnan(t)) continue;
        ut[nu] = t;
        for (int c = 0; c < 3; ++c) up[nu * 3 + c] = gps_p[ord[i] * 3 + c];
        ++nu;
    }
    if (nu >= 2) {                                                                /* :343-345 */
        int64_t s0 = 0;
        int64_t *qidx = (int64_t *)malloc(sizeof(int64_t) * (size_t)ns);
        double *xq = (double *)malloc(sizeof(double) * (size_t)ns);
        while (s0 < nu) {                                                         /* :350-355 segments */
            int64_t s1 = s0;
            while (s1 + 1 < nu && !(ut[s1 + 1] - ut[s1] > max_gap)) ++s1;
            int64_t len = s1 - s0 + 1;
            if (len >= 2) {                                                       /* :360 */
                int strictly = 1;
                for (int64_t i = s0; i < s1; ++i) if (!(ut[i + 1] - ut[i] > 1e-9)) strictly = 0;   /* :364 */
                if (strictly) {
                    const double eps = 1e-9;                                      /* :372 */
                    int64_t nq = 0;
                    for (int64_t i = 0; i < ns; ++i)
                        if (slam_t[i] >= ut[s0] - eps && slam_t[i] <= ut[s1] + eps) { qidx[nq] = i; xq[nq] = slam_t[i]; ++nq; }   /* :373 */
                    /* bounds_error=False, fill_value=nan: queries outside [x0, x_last] give NaN (:368) */
                    int64_t nin = 0;
                    for (int64_t q = 0; q < nq; ++q) {
                        if (xq[q] < ut[s0] || xq[q] > ut[s1]) {
                            for (int c = 0; c < 3; ++c) aligned[qidx[q] * 3 + c] = NAN;
                        } else { qidx[nin] = qidx[q]; xq[nin] = xq[q]; ++nin; }
                    }
                    if (len >= 4) spline_notaknot_eval(ut + s0, up + s0 * 3, len, xq, nin, qidx, aligned);   /* :362 cubic */
                    else {
                        for (int64_t q = 0; q < nin; ++q) {                       /* linear */
                            /* scipy _call_linear: hi = clip(searchsorted(x, t, 'left'), 1, len-1); lo = hi-1 */
                            double t = xq[q]; int64_t hi = s0;
                            while (hi <= s1 && ut[hi] < t) ++hi;
                            if (hi < s0 + 1) hi = s0 + 1;
                            if (hi > s1) hi = s1;
                            int64_t lo = hi - 1;
                            double x0 = ut[lo], x1 = ut[lo + 1];
                            for (int c = 0; c < 3; ++c) {
                                double y0 = up[lo * 3 + c], y1 = up[(lo + 1) * 3 + c];
                                double slope = (y1 - y0) / (x1 - x0);
                                aligned[qidx[q] * 3 + c] = slope * (t - x0) + y0;
                            }
                        }
                    }
                    for (int64_t q = 0; q < nin; ++q) {                           /* :377-379 */
                        const double *v = aligned + qidx[q] * 3;
                        if (!(isnan(v[0]) || isnan(v[1]) || isnan(v[2]))) valid[qidx[q]] = 1;
                    }
                }
            }
            s0 = s1 + 1;
        }
        free(qidx); free(xq);
    }
    free(ord); free(tt); free(ut); free(up);
}


/* WGS84 geodetic -> local ENU about (lat0, lon0, h0): textbook ECEF difference rotated into the tangent frame.  Not a
 * reference function (the reference projects with UTM); checker for the product's additional ENU kernel. */
ORC_API void orc_geodetic_to_enu(const double *lat_deg, const double *lon_deg, const double *alt, int64_t n, double lat0, double lon0,
                                 double h0, double *e, double *nn, double *u)
{
    const long double a = 6378137.0L, f = 1.0L / 298.257223563L, e2 = f * (2.0L - f), d2r = 3.14159265358979323846264338327950288L / 180.0L;
    long double sp0 = sinl(lat0 * d2r), cp0 = cosl(lat0 * d2r), sl0 = sinl(lon0 * d2r), cl0 = cosl(lon0 * d2r);
    long double N0 = a / sqrtl(1.0L - e2 * sp0 * sp0);
    long double x0 = (N0 + h0) * cp0 * cl0, y0 = (N0 + h0) * cp0 * sl0, z0 = (N0 * (1.0L - e2) + h0) * sp0;
    for (int64_t i = 0; i < n; ++i) {
        long double sp = sinl(lat_deg[i] * d2r), cp = cosl(lat_deg[i] * d2r), sl = sinl(lon_deg[i] * d2r), cl = cosl(lon_deg[i] * d2r);
        long double Nn = a / sqrtl(1.0L - e2 * sp * sp);
        long double dx = (Nn + alt[i]) * cp * cl - x0, dy = (Nn + alt[i]) * cp * sl - y0, dz = (Nn * (1.0L - e2) + alt[i]) * sp - z0;
        e[i] = (double)(-sl0 * dx + cl0 * dy);
        nn[i] = (double)(-sp0 * cl0 * dx - sp0 * sl0 * dy + cp0 * dz);
        u[i] = (double)(cp0 * cl0 * dx + cp0 * sl0 * dy + sp0 * dz);
    }
}

/* layout probe so the ctypes mirror of orc_config can be checked */
ORC_API int orc_config_size(void) { return (int)sizeof(orc_config); }
