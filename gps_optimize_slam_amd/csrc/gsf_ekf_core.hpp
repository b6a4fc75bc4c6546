// gsf_ekf_core.hpp -- per-trajectory EKF(+RTS) state machine, all state in registers.
//
// Restates apply_ekf_correction (ref :831-935) + ExtendedKalmanFilter (ref :679-772) +
// rts_smoother_segment (ref :777-803) + is_sharp_turn_in_segment (ref :808-826) for ONE
// trajectory, after the reference's time-alignment call (:847).
//
// MI355X-first choices (see DESIGN.md "K4"):
//  * The 7x7 covariance is diagonal for the whole run by construction (SURVEY F4/Q5:
//    P0,Q,R come from np.diag, H=[I3 0], Joseph update keeps it diagonal), so P is 7
//    doubles in VGPRs and the Kalman gain is 3 scalar divisions.  The oracle keeps the
//    dense 7x7 algebra; parity between the two is what the tests check.
//  * RTS needs no history buffer: inside an outage x_f[k]==x_p[k] and P_f[k]==P_p[k], so the
//    back-pass re-reads the already-written output rows (they hold x_f) and rebuilds
//    P_p[k] = P_p[k+1] - Q*dt[k+1] on the way down (<=1 ulp per step; tolerance 1e-6 m).
//  * The sharp-turn gate is accumulated forward during the outage (running max yaw rate),
//    so the recovery step does no extra pass over the segment.
#pragma once
#include "gsf_math.hpp"

namespace gsf {

struct EkfConfig {                 // CONFIG['ekf'] + CONFIG['rts_decision'], ref :24-29, :67-70
    double P0[7], Qps[7], Rm[3];
    double yaw_thr_rad;            // deg2rad(sharp_turn_yaw_rate_threshold_deg_per_sec)
    int32_t sharp_turn_steps;      // default_ekf_transition_steps_on_sharp_turn
    int32_t _pad;
};

enum : int32_t {                   // per-trajectory status bits (include/gsf.h)
    ST_HAD_OUTAGE = 1, ST_RTS_APPLIED = 2, ST_SHARP_TURN = 4, ST_ENDED_IN_OUTAGE = 8, ST_BAD_QUAT = 16
};

struct StepIn {                    // one pose of the ORIGINAL SLAM track + its time-aligned GNSS fix
    double t;
    Vec3 p;
    Quat q;
    Vec3 z;
    bool valid;
};

// Out must provide:
//   void  store(int64_t i, const Vec3& p, const Quat& q);      fused pose i
//   void  load(int64_t i, Vec3& p, Quat& q) const;             read back a previously stored pose
//   double stamp(int64_t i) const;                               input stamp i (RTS dt rebuild)
// |wrapped yaw difference| / dt of two unit quaternions, ref :819-823
GSF_HD_COLD double yaw_rate_pair(const Quat& r1, const Quat& r2, double dt)
{
    double d = quat_yaw_zyx(r2) - quat_yaw_zyx(r1);
    double dy = atan2(sin(d), cos(d));                                   // :822
    return fabs(dy / dt);                                                // :823
}

template <class Out>
struct EkfTraj {
    // filter state
    Vec3 p; Quat q; double P[7];
    double weight; bool prev_avail;
    // driver state (ref :859-862)
    bool in_outage; int64_t ostart;
    // previous original pose
    Vec3 po_prev; Quat r_prev; bool ok_prev; double t_prev;
    // sharp-turn accumulators over the open outage
    double max_rate; bool seg_bad;
    int32_t status;

    GSF_HD void init(const EkfConfig& cfg, const Vec3& p0, const Quat& q0, const StepIn& first, Out& out)
    {
        p = p0; q = ekf_normalize(q0);                                   // :842, :683
#pragma unroll
        for (int c = 0; c < 7; ++c) P[c] = cfg.P0[c];
        weight = 0.0;
        prev_avail = first.valid;                                        // :848
        in_outage = !prev_avail; ostart = in_outage ? 0 : -1;            // :861-862
        status = in_outage ? ST_HAD_OUTAGE : 0;
        po_prev = first.p; ok_prev = quat_unit(first.q, r_prev); t_prev = first.t;
        max_rate = 0.0; seg_bad = false;
        out.store(0, p, q);                                              // :856
    }

    GSF_HD void step(const EkfConfig& cfg, int64_t i, const StepIn& in, Out& out)
    {
        const double t = in.t;
        const double dt = fmax(1e-6, t - t_prev);                        // :865
        // ---- calculate_relative_pose, ref :77-92
        Quat r_cur; const bool ok_cur = quat_unit(in.q, r_cur);
        Vec3 dpl{ 0.0, 0.0, 0.0 }; Quat dq{ 0.0, 0.0, 0.0, 1.0 };
        if (ok_prev && ok_cur) {
            Quat r1i = quat_conj(r_prev);
            dpl = quat_rotate(r1i, Vec3{ in.p.x - po_prev.x, in.p.y - po_prev.y, in.p.z - po_prev.z });
            dq = quat_mul(r1i, r_cur);
        } else status |= ST_BAD_QUAT;                                    // :84-86
        // ---- measurement gate, ref :867-869
        bool avail = in.valid && !(isnan(in.z.x) || isnan(in.z.y) || isnan(in.z.z));
        // ---- outage bookkeeping, ref :872-894
        bool perform_rts = true; int eff = 0;
        const bool recovering = avail && in_outage;
        if (!avail && !in_outage) {                                      // :875-877
            in_outage = true; ostart = i; status |= ST_HAD_OUTAGE;
            max_rate = 0.0; seg_bad = false;
        } else if (recovering) {                                         // :879-894
            if (i - ostart >= 2 && (seg_bad || max_rate > cfg.yaw_thr_rad)) {
                perform_rts = false; eff = cfg.sharp_turn_steps; status |= ST_SHARP_TURN;
            }
        } else if (!avail && i > ostart) {
            // still inside the outage: extend is_sharp_turn_in_segment (:808-826) by the pair (i-1, i)
            if (t > t_prev) {                                            // :817
                if (!(ok_prev && ok_cur)) seg_bad = true;                // :821
                else max_rate = fmax(max_rate, yaw_rate_pair(r_prev, r_cur, t - t_prev));   // :819-824
            }
        }
        // ---- ExtendedKalmanFilter.process_step, ref :736-772 (current_transition_steps == 0 always, Q6)
        const double weight_delta = eff > 0 ? 1.0 / (double)eff : 1.0;   // :743
        // _predict, ref :702-715
        Quat qn; quat_unit(q, qn);
        Quat dqn; quat_unit(dq, dqn);
        Vec3 rp = quat_rotate(qn, dpl);
        const Vec3 pp{ p.x + rp.x, p.y + rp.y, p.z + rp.z };
        const Quat pq = ekf_normalize(quat_mul(qn, dqn));
        const double dta = fmax(fabs(dt), 1e-6);
        double Pp[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) Pp[c] = P[c] + cfg.Qps[c] * dta;
        // _update, ref :717-734 (diagonal: 3 scalar Kalman filters; quaternion block untouched)
        Vec3 up = pp; Quat uq = pq; double Pu[3] = { Pp[0], Pp[1], Pp[2] };
        if (avail) {
            const double zz[3] = { in.z.x, in.z.y, in.z.z };
            const double pv[3] = { pp.x, pp.y, pp.z };
            double uv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double S = Pp[c] + cfg.Rm[c];
                double k = Pp[c] * (1.0 / S);
                uv[c] = pv[c] + k * (zz[c] - pv[c]);
                double ik = 1.0 - k;
                Pu[c] = (ik * Pp[c]) * ik + (k * cfg.Rm[c]) * k;         // Joseph form, :731
            }
            up = Vec3{ uv[0], uv[1], uv[2] };
            uq = ekf_normalize(pq);                                      // :729
        }
        // GNSS weight state machine, ref :752-758
        const bool just_rec = avail && !prev_avail;
        if (avail) {
            if (just_rec || eff == 0) weight = (eff == 0) ? 1.0 : weight_delta;
            else if (weight < 1.0) weight = fmin(1.0, weight + weight_delta);
        } else weight = 0.0;
        // fuse, ref :760-768
        Vec3 fp; Quat fq;
        if (avail) {
            if (weight < 1.0 && eff > 0) {                               // one-step blend (Q7)
                const double w = weight;
                fp = Vec3{ (1.0 - w) * pp.x + w * up.x, (1.0 - w) * pp.y + w * up.y, (1.0 - w) * pp.z + w * up.z };
                fq = quat_nlerp(pq, uq, w);
            } else { fp = up; fq = uq; }
            P[0] = Pu[0]; P[1] = Pu[1]; P[2] = Pu[2];
        } else { fp = pp; fq = pq; P[0] = Pp[0]; P[1] = Pp[1]; P[2] = Pp[2]; }
        P[3] = Pp[3]; P[4] = Pp[4]; P[5] = Pp[5]; P[6] = Pp[6];
        p = fp; q = fq;
        prev_avail = avail;                                              // :771
        out.store(i, p, q);                                              // :904
        // ---- per-outage RTS back-pass, ref :906-928 + :777-803
        if (recovering) {
            if (perform_rts) {
                rts_backpass(cfg, i, pp, pq, Pp, out);
                status |= ST_RTS_APPLIED;
            }
            in_outage = false; ostart = -1;                              // :926-928
        }
        po_prev = in.p; r_prev = r_cur; ok_prev = ok_cur; t_prev = t;    // :930
    }

    // Smooth [ostart .. i]; x_s[i] = x_f[i] (:782) is already stored.
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
                double a = Pk * (1.0 / Pn[c]);                           // A_k = P_f[k] inv(P_p[k+1]), F = I (:789)
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
        if (in_outage && ostart != -1) status |= ST_ENDED_IN_OUTAGE;     // :932
        return status;
    }
};

}  // namespace gsf
