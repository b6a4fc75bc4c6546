// gsf_ekf_core.hpp -- per-trajectory EKF(+RTS) state machine, all state in registers.
//
// Restates apply_ekf_correction (ref :831-935) + ExtendedKalmanFilter (ref :679-772) +
// rts_smoother_segment (ref :777-803) + is_sharp_turn_in_segment (ref :808-826) for ONE
// trajectory, after the reference's time-alignment call (:847).
//
// MI355X-first choices (see DESIGN.md "K4"):
//  * The 7x7 covariance is diagonal for the whole run by construction (SURVEY F4/Q5:
//    P0,Q,R come from np.diag, H=[I3 0], Joseph update keeps it diagonal), so P is 7
//    doubles in VGPRs and the Kalman gain is 3 scalar reciprocals.  The oracle keeps the
//    dense 7x7 algebra; parity between the two is what the tests check.
//  * One lane = one trajectory, so 64 different outage histories share a wave.  The per-pose
//    body is therefore BRANCH-FREE (predict, update and the GNSS gate are computed for every
//    lane and blended with selects); everything that only happens around outages (sharp-turn
//    accumulation, recovery decision, one-step blend, RTS back-pass) sits behind ONE
//    wave-uniform vote, so a wave whose 64 tracks all have GNSS executes none of it.
//  * RTS needs no history buffer: inside an outage x_f[k]==x_p[k] and P_f[k]==P_p[k], so the
//    back-pass re-reads the already-written output rows (they hold x_f) and rebuilds
//    P_p[k] = P_p[k+1] - Q*dt[k+1] on the way down (<=1 ulp per step; gate 1e-6 m).
//  * The sharp-turn gate is accumulated forward during the outage.  max|dyaw|/dt > thr is
//    evaluated per pair as cos(dyaw) < cos(thr*dt) with cos(dyaw) from the two headings'
//    (cos,sin) -- one cos() per pair instead of three atan2 + sin + cos.
#pragma once
#include "gsf_math.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define GSF_WAVE_ANY(pred) (__builtin_amdgcn_ballot_w64(pred) != 0ull)
#else
#define GSF_WAVE_ANY(pred) (pred)
#endif

namespace gsf {

struct EkfConfig {                 // CONFIG['ekf'] + CONFIG['rts_decision'], ref :24-29, :67-70
    double P0[7], Qps[7], Rm[3];
    double yaw_thr_rad;            // deg2rad(sharp_turn_yaw_rate_threshold_deg_per_sec)
    int32_t sharp_turn_steps;      // default_ekf_transition_steps_on_sharp_turn
    int32_t _pad;
};

// Which time-synchronised rows feed the Sim3 fit of the fused chains (gsf_set_sim3_rows): mode 0 = every row with valid finite GNSS,
// mode 1 = the choice of main_process_gui (ref :973-998: first gap-free segment, <= max_initial_duration, two fall-backs)
struct FitRows {
    int32_t mode, min_samples;     // CONFIG['sim3_ransac']['min_samples'] (:34)
    double max_gap;                // CONFIG['time_alignment']['max_gps_gap_threshold'] (:53)
    double max_dur;                // CONFIG['sim3_ransac']['max_initial_duration'] (:37)
};

enum : int32_t {                   // per-trajectory status bits (include/gsf.h)
    ST_HAD_OUTAGE = 1, ST_RTS_APPLIED = 2, ST_SHARP_TURN = 4, ST_ENDED_IN_OUTAGE = 8, ST_BAD_QUAT = 16
};

struct StepIn {                    // one pose of the ORIGINAL SLAM track + its time-aligned GNSS fix
    double t;
    Vec3 p;
    Quat q;
    Vec3 z;
    uint32_t valid;               // raw mask byte, compared at USE time (a compare at load time would stall on the newest load)
};

// (cos, sin) * h of the reference's "yaw" (as_euler('zyx')[0] = atan2(-m01, m00)) of a unit quaternion
GSF_HD void yaw_vec(const Quat& q, double& a, double& b)
{
    a = q.x * q.x - q.y * q.y - q.z * q.z + q.w * q.w;       // m00
    b = -2.0 * (q.x * q.y - q.z * q.w);                      // -m01
}

// |wrap(yaw2 - yaw1)| / dt > thr  (ref :819-826), without forming the angles:
//   |wrap(d)| > c  <=>  cos(d) < cos(c)  for c in [0, pi);  never for c >= pi.
// atan2(0,0) = 0 in the reference, i.e. a degenerate heading vector counts as (1, 0).
GSF_HD_COLD bool yaw_rate_exceeds(Quat r1, Quat r2, double dt, double thr)   // by value: by-reference args of a noinline call live in scratch
{
    double a1, b1, a2, b2;
    yaw_vec(r1, a1, b1); yaw_vec(r2, a2, b2);
    double h1 = a1 * a1 + b1 * b1, h2 = a2 * a2 + b2 * b2;
    if (!(h1 > 0.0)) { a1 = 1.0; b1 = 0.0; h1 = 1.0; }
    if (!(h2 > 0.0)) { a2 = 1.0; b2 = 0.0; h2 = 1.0; }
    double c = thr * dt;
    if (!(c < 3.141592653589793)) return false;
    if (c < 0.0) return true;                                // any rate >= 0 exceeds a negative threshold
    double cosd = (a1 * a2 + b1 * b2) * fast_rsqrt(h1 * h2);
    return cosd < cos(c);
}

// The same test without a libm call on the usual range (c = thr dt <= pi/4: degree-14 kernel polynomial of cos) -- for the wave
// kernels, where the cold blocks of a launch start with a cold instruction cache: the call into yaw_rate_exceeds and on into
// libm's cos costs ~1 us of instruction-fetch misses per wave that meets an outage, and at small batches the slowest wave IS the
// launch time.  Two wrappers around one body: a CALL for big batches (inlined it costs the hot loop 27 registers, i.e. one wave
// per SIMD) and an INLINE form for small batches, where registers are free and even the one far call is worth avoiding.
GSF_HD bool yaw_rate_exceeds_body(const Quat& r1, const Quat& r2, double dt, double thr)
{
    double a1, b1, a2, b2;
    yaw_vec(r1, a1, b1); yaw_vec(r2, a2, b2);
    double h1 = a1 * a1 + b1 * b1, h2 = a2 * a2 + b2 * b2;
    if (!(h1 > 0.0)) { a1 = 1.0; b1 = 0.0; h1 = 1.0; }
    if (!(h2 > 0.0)) { a2 = 1.0; b2 = 0.0; h2 = 1.0; }
    const double c = thr * dt;
    const double cosd = (a1 * a2 + b1 * b2) * fast_rsqrt(h1 * h2);
    double cc;
    if (c <= 0.785) {
        const double z = c * c;
        const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                          z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
        cc = fma(z * z, pc, fma(-0.5, z, 1.0));
    } else {
        cc = cos(c);
    }
    const bool exceeds = cosd < cc;
    return (c < 0.0) ? true : ((c < 3.141592653589793) ? exceeds : false);
}
GSF_HD_COLD bool yaw_rate_exceeds_poly(Quat r1, Quat r2, double dt, double thr) { return yaw_rate_exceeds_body(r1, r2, dt, thr); }

// Out must provide:
//   void  store(int64_t i, const Vec3& p, const Quat& q);      fused pose i
//   void  load(int64_t i, Vec3& p, Quat& q) const;             read back a previously stored pose
//   double stamp(int64_t i) const;                               input stamp i (RTS dt rebuild)
template <class Out>
struct EkfTraj {
    // filter state
    Vec3 p; Quat q; double P[7];
    bool prev_avail;               // gnss_available_prev; after every step also == !in_gnss_outage
    int64_t ostart;                // first index of the open outage (meaningful while !prev_avail), ref :859-862
    // previous original pose
    Vec3 po_prev; Quat r_prev; bool ok_prev; double t_prev;
    bool seg_sharp;                // is_sharp_turn_in_segment over the open outage so far
    int32_t status;

    GSF_HD void init(const EkfConfig& cfg, const Vec3& p0, const Quat& q0, const StepIn& first, Out& out)
    {
        p = p0; q = ekf_normalize(q0);                                   // :842, :683
#pragma unroll
        for (int c = 0; c < 7; ++c) P[c] = cfg.P0[c];
        prev_avail = first.valid != 0;                                   // :848  (raw mask, not NaN-gated)
        ostart = 0;                                                      // :861-862 (in_outage = !prev_avail, start 0)
        status = prev_avail ? 0 : ST_HAD_OUTAGE;
        po_prev = first.p; ok_prev = quat_unit(first.q, r_prev); t_prev = first.t;
        seg_sharp = false;
        out.store(0, p, q);                                              // :856
    }

    GSF_HD void step(const EkfConfig& cfg, int64_t i, const StepIn& in, Out& out)
    {
        const double t = in.t;
        const double dt = fmax(1e-6, t - t_prev);                        // :865 (== the dt_adj of :711)
        // ---- calculate_relative_pose, ref :77-92 (zero motion / identity if either quaternion is invalid, :84-86)
        Quat r_cur; const bool ok_cur = quat_unit(in.q, r_cur);
        const bool both_ok = ok_prev && ok_cur;
        const Quat r1i = quat_conj(r_prev);
        Vec3 dpl = quat_rotate(r1i, Vec3{ in.p.x - po_prev.x, in.p.y - po_prev.y, in.p.z - po_prev.z });
        Quat dq = quat_mul(r1i, r_cur);
        dpl.x = both_ok ? dpl.x : 0.0; dpl.y = both_ok ? dpl.y : 0.0; dpl.z = both_ok ? dpl.z : 0.0;
        dq.x = both_ok ? dq.x : 0.0; dq.y = both_ok ? dq.y : 0.0; dq.z = both_ok ? dq.z : 0.0; dq.w = both_ok ? dq.w : 1.0;
        status |= both_ok ? 0 : ST_BAD_QUAT;
        // ---- measurement gate, ref :867-869
        const bool avail = (in.valid != 0) && !(isnan(in.z.x) || isnan(in.z.y) || isnan(in.z.z));
        const bool was_outage = !prev_avail;                             // in_gnss_outage before this step
        // ---- _predict, ref :702-715.  The reference re-normalises the state quaternion and the increment through
        // Rotation.from_quat; both are unit already (q leaves every step through ekf_normalize, dq is a product of two
        // unit quaternions), so those normalisations are identities to 1 ulp and are not re-done.
        const Vec3 rp = quat_rotate(q, dpl);
        const Vec3 pp{ p.x + rp.x, p.y + rp.y, p.z + rp.z };
        const Quat pq = ekf_normalize(quat_mul(q, dq));
        double Pp[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) Pp[c] = P[c] + cfg.Qps[c] * dt;
        // ---- _update, ref :717-734: three scalar Kalman filters (the quaternion block is untouched: K rows 3..6 == 0,
        // and :729 re-normalises an already unit quaternion).  Computed for every lane, selected by `avail`.
        const double pv[3] = { pp.x, pp.y, pp.z }, zz[3] = { in.z.x, in.z.y, in.z.z };
        double uv[3], Pu[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double k = Pp[c] * fast_rcp(Pp[c] + cfg.Rm[c]);        // K = P H^T S^-1
            uv[c] = pv[c] + k * (zz[c] - pv[c]);
            const double ik = 1.0 - k;
            Pu[c] = (ik * Pp[c]) * ik + (k * cfg.Rm[c]) * k;             // Joseph form, :731
        }
        // ---- fuse, ref :752-768 with effective transition steps 0 (hard update; the reference's driver always passes 0
        // except on a sharp-turn recovery, handled in the cold block below)
        p.x = avail ? uv[0] : pv[0]; p.y = avail ? uv[1] : pv[1]; p.z = avail ? uv[2] : pv[2];
        q = pq;
        P[0] = avail ? Pu[0] : Pp[0]; P[1] = avail ? Pu[1] : Pp[1]; P[2] = avail ? Pu[2] : Pp[2];
        P[3] = Pp[3]; P[4] = Pp[4]; P[5] = Pp[5]; P[6] = Pp[6];
        // ---- outage machinery: work only for lanes that are in / entering / leaving an outage
        const bool special = was_outage || !avail;
        if (GSF_WAVE_ANY(special)) {
            if (!avail && !was_outage) {                                 // outage starts here, ref :875-877
                ostart = i; seg_sharp = false; status |= ST_HAD_OUTAGE;
            } else if (!avail && was_outage) {
                // still inside: extend is_sharp_turn_in_segment (:808-826) by the pair (i-1, i); i-1 >= ostart holds
                if (t > t_prev && !seg_sharp)                            // :817
                    seg_sharp = !both_ok || yaw_rate_exceeds(r_prev, r_cur, t - t_prev, cfg.yaw_thr_rad);   // :821-824
            } else if (avail && was_outage) {                            // recovery, ref :879-928
                const bool sharp = (i - ostart >= 2) && seg_sharp;       // :882-894
                if (sharp) {
                    status |= ST_SHARP_TURN;
                    const int eff = cfg.sharp_turn_steps;                // :889
                    // gnss_update_weight (:752-758) on this step is 1/eff; it blends only if that is < 1 (Q7)
                    if (eff > 1) {
                        const double w = 1.0 / (double)eff;
                        p.x = (1.0 - w) * pv[0] + w * uv[0]; p.y = (1.0 - w) * pv[1] + w * uv[1]; p.z = (1.0 - w) * pv[2] + w * uv[2];
                        q = quat_nlerp(pq, pq, w);                       // :765 (the updated quaternion == the predicted one)
                    }
                } else {
                    rts_backpass(cfg, i, pp, pq, Pp, out);               // :906-922
                    status |= ST_RTS_APPLIED;
                }
            }
        }
        prev_avail = avail;                                              // :771, :926
        out.store(i, p, q);                                              // :904
        po_prev = in.p; r_prev = r_cur; ok_prev = ok_cur; t_prev = t;    // :930
    }

    // Smooth [ostart .. i-1] against x_s[i] = x_f[i] (:782), which is the current (p, q).
    GSF_HD void rts_backpass(const EkfConfig& cfg, int64_t i, const Vec3& pp_i, const Quat& pq_i, const double* Pp_i, Out& out)
    {
        double xs[7] = { p.x, p.y, p.z, q.x, q.y, q.z, q.w };            // x_s[k+1]
        double xp[7] = { pp_i.x, pp_i.y, pp_i.z, pq_i.x, pq_i.y, pq_i.z, pq_i.w };   // x_p[k+1]
        double Pn[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) Pn[c] = Pp_i[c];                     // P_p[k+1]
        double tk1 = out.stamp(i);
        for (int64_t k = i - 1; k >= ostart; --k) {                      // :784
            const double tk = out.stamp(k);
            const double dta = fmax(1e-6, tk1 - tk);                     // the forward pass's dt[k+1] (:865, :711)
            Vec3 fpk; Quat fqk; out.load(k, fpk, fqk);                   // x_f[k] (== x_p[k] inside the outage)
            const double xf[7] = { fpk.x, fpk.y, fpk.z, fqk.x, fqk.y, fqk.z, fqk.w };
            double s[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                double Pk = Pn[c] - cfg.Qps[c] * dta;                    // P_f[k] = P_p[k]
                double a = Pk * fast_rcp(Pn[c]);                         // A_k = P_f[k] inv(P_p[k+1]), F = I (:789)
                s[c] = xf[c] + a * (xs[c] - xp[c]);                      // :798
                Pn[c] = Pk;
            }
            Quat sq = ekf_normalize(Quat{ s[3], s[4], s[5], s[6] });     // :799
            s[3] = sq.x; s[4] = sq.y; s[5] = sq.z; s[6] = sq.w;
            out.store(k, Vec3{ s[0], s[1], s[2] }, sq);                  // :920-921
#pragma unroll
            for (int c = 0; c < 7; ++c) { xs[c] = s[c]; xp[c] = xf[c]; }
            tk1 = tk;
        }
    }

    GSF_HD int32_t finish()
    {
        if (!prev_avail) status |= ST_ENDED_IN_OUTAGE;                   // :932
        return status;
    }
};

}  // namespace gsf
