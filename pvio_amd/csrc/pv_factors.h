// pv_factors.h -- per-factor residual / Jacobian evaluation, written for one GPU thread per factor.
//
// Replaces the reference's ceres cost functors on the hot path:
//   A2  ReprojectionErrorCost::Evaluate            estimation/ceres/reprojection_error_cost.h:40-120
//   A3  PreIntegrationErrorCost::Evaluate          estimation/ceres/preintegration_error_cost.h:40-160
//   A5  MarginalizationErrorCost::Evaluate         estimation/ceres/marginalization_error_cost.h:53-94
//   A6  AugmentedPlaneDistanceErrorCost::Evaluate  estimation/ceres/augmented_plane_distance_error_cost.h:53-136
// All Jacobians are in LOCAL coordinates (theta = right perturbation q (x) exp(theta)); the reference's
// QuaternionParameterization::ComputeJacobian is "identity on top" (quaternion_parameterization.h:33-36).
//
// Layout choices for the GPU: a frame is expanded ONCE per kernel into a 28-double record (rotation
// matrices instead of quaternions) that lives in LDS; factor code reads records, never quaternions.
#pragma once
#include "pv_math.h"

namespace pv {

constexpr double kGravity = 9.80665; // PVIO_GRAVITY_NOMINAL (common.h:62)

// ---- frame record -----------------------------------------------------------------------------------
// [0:9) Rb = R(q_body)   [9:12) pb   [12:21) Rc = R(q_cs)   [21:24) pc   [24:28) W (2x2 sqrt_inv_cov)
constexpr int kFrameRec = 28;
PV_HD void frame_record(double *rec, const double *state16, const double *cam7, const double *W4) {
    q_to_mat(rec, state16);
    rec[9] = state16[4], rec[10] = state16[5], rec[11] = state16[6];
    q_to_mat(rec + 12, cam7);
    rec[21] = cam7[4], rec[22] = cam7[5], rec[23] = cam7[6];
    rec[24] = W4[0], rec[25] = W4[1], rec[26] = W4[2], rec[27] = W4[3];
}

// ---- A2: reprojection ----------------------------------------------------------------------------------
// Ft / Fr: frame records of the target / reference (anchor) frame.  Outputs the whitened residual r[2] and,
// when JAC, Jt (2x6: theta_tgt, p_tgt), Jr (2x6: theta_ref, p_ref), Jd (2: inverse depth), row-major.
// Returns y_t.z (depth in the target camera) for validity checks.
template <bool JAC>
PV_HD double reproj_eval(const double *Ft, const double *Fr, double rho, double zr0, double zr1, double zt0, double zt1,
                         double *r, double *Jt, double *Jr, double *Jd) {
    const double inv = 1.0 / rho;
    const double y_ref[3] = {zr0 * inv, zr1 * inv, inv};                   // :58
    double y_rc[3], x[3], d[3], y_tc[3], e[3], y_t[3];
    m3_vec(y_rc, Fr + 12, y_ref);
    v3_add(y_rc, y_rc, Fr + 21);                                           // :59
    m3_vec(x, Fr, y_rc);
    v3_add(x, x, Fr + 9);                                                  // :60
    v3_sub(d, x, Ft + 9);
    m3_tvec(y_tc, Ft, d);                                                  // :61
    v3_sub(e, y_tc, Ft + 21);
    m3_tvec(y_t, Ft + 12, e);                                              // :62
    const double iz = 1.0 / y_t[2];
    const double u = y_t[0] * iz - zt0, v = y_t[1] * iz - zt1;            // :63
    const double *W = Ft + 24;
    r[0] = W[0] * u + W[1] * v;                                            // :116
    r[1] = W[2] * u + W[3] * v;
    if (JAC) {
        const double xz = -y_t[0] * iz * iz, yz = -y_t[1] * iz * iz;
        // Jpi = W * [iz 0 xz ; 0 iz yz]                                     :66-69
        const double P[6] = {W[0] * iz, W[1] * iz, W[0] * xz + W[1] * yz, W[2] * iz, W[3] * iz, W[2] * xz + W[3] * yz};
        double A[6], B[6], C[6];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // A = Jpi Rc_t^T ; B = A Rb_t^T ; C = B Rb_r                    :75,:79,:86
#pragma unroll
            for (int j = 0; j < 3; ++j) A[3 * i + j] = P[3 * i] * Ft[12 + 3 * j] + P[3 * i + 1] * Ft[12 + 3 * j + 1] + P[3 * i + 2] * Ft[12 + 3 * j + 2];
#pragma unroll
            for (int j = 0; j < 3; ++j) B[3 * i + j] = A[3 * i] * Ft[3 * j] + A[3 * i + 1] * Ft[3 * j + 1] + A[3 * i + 2] * Ft[3 * j + 2];
#pragma unroll
            for (int j = 0; j < 3; ++j) C[3 * i + j] = B[3 * i] * Fr[j] + B[3 * i + 1] * Fr[3 + j] + B[3 * i + 2] * Fr[6 + j];
        }
        double ry[3];
        m3_vec(ry, Fr + 12, y_ref);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double a0 = A[3 * i], a1 = A[3 * i + 1], a2 = A[3 * i + 2];
            const double c0 = C[3 * i], c1 = C[3 * i + 1], c2 = C[3 * i + 2];
            // A * hat(y_tc)                                                 :95
            Jt[6 * i + 0] = a1 * y_tc[2] - a2 * y_tc[1];
            Jt[6 * i + 1] = a2 * y_tc[0] - a0 * y_tc[2];
            Jt[6 * i + 2] = a0 * y_tc[1] - a1 * y_tc[0];
            Jt[6 * i + 3] = -B[3 * i], Jt[6 * i + 4] = -B[3 * i + 1], Jt[6 * i + 5] = -B[3 * i + 2]; // :100
            // -C * hat(y_rc)                                                :104
            Jr[6 * i + 0] = -(c1 * y_rc[2] - c2 * y_rc[1]);
            Jr[6 * i + 1] = -(c2 * y_rc[0] - c0 * y_rc[2]);
            Jr[6 * i + 2] = -(c0 * y_rc[1] - c1 * y_rc[0]);
            Jr[6 * i + 3] = B[3 * i], Jr[6 * i + 4] = B[3 * i + 1], Jr[6 * i + 5] = B[3 * i + 2];     // :109
            Jd[i] = -(c0 * ry[0] + c1 * ry[1] + c2 * ry[2]) * inv;                                     // :113
        }
    }
    return y_t[2];
}

// ---- A3: IMU pre-integration (un-whitened part; the caller multiplies by U = sqrt_inv_cov) ----------------
// si/sj: 16-double states; bias0: live frame_i->motion.bg/ba (6); delta: dt,dq,dp,dv (11); jac: 5 3x3 blocks (45);
// imu_i/imu_j: 7-double extrinsics.  raw[15]; G = 15x30 row-major (may be null), columns = error state i, j.
// G_is_zero: the caller has already cleared G (a workgroup does that with all its threads, one thread takes ~3 us).
PV_HD void preint_raw(const double *si, const double *sj, const double *bias0, const double *delta, const double *jac,
                      const double *imu_i, const double *imu_j, double *raw, double *G, bool G_is_zero = false) {
    const double g[3] = {0.0, 0.0, -kGravity};
    const double dt = delta[0];
    const double *dq = delta + 1, *dp = delta + 5, *dv = delta + 8;
    const double *dq_dbg = jac, *dp_dbg = jac + 9, *dp_dba = jac + 18, *dv_dbg = jac + 27, *dv_dba = jac + 36;
    double q_i[4], q_j[4], p_i[3], p_j[3], t3[3];
    q_mul(q_i, si, imu_i);                                              // :60
    q_mul(q_j, sj, imu_j);                                              // :62
    q_rot(t3, si, imu_i + 4);
    v3_add(p_i, si + 4, t3);                                            // :61
    q_rot(t3, sj, imu_j + 4);
    v3_add(p_j, sj + 4, t3);                                            // :63
    double dbg[3], dba[3];
    v3_sub(dbg, si + 10, bias0);                                        // :69
    v3_sub(dba, si + 13, bias0 + 3);                                    // :70
    // r_q = Log( (dq * Exp(dq_dbg dbg))^-1 * q_i^-1 * q_j )              :79
    double th[3], eq[4], dqc[4], a[4], b[4], c[4], rq[3];
    m3_vec(th, dq_dbg, dbg);
    q_expmap(eq, th);
    q_mul(a, dq, eq);
    q_conj(dqc, a);
    q_conj(b, q_i);
    q_mul(c, dqc, b);
    q_mul(a, c, q_j);
    q_logmap(rq, a);
    // r_p, r_v                                                           :80-81
    double w1[3], w2[3], corr[3], rp[3], rv[3];
    for (int k = 0; k < 3; ++k) w1[k] = p_j[k] - p_i[k] - dt * si[7 + k] - 0.5 * dt * dt * g[k];
    q_rot_inv(rp, q_i, w1);
    m3_vec(corr, dp_dbg, dbg);
    m3_vec(t3, dp_dba, dba);
    for (int k = 0; k < 3; ++k) rp[k] -= dp[k] + corr[k] + t3[k];
    for (int k = 0; k < 3; ++k) w2[k] = sj[7 + k] - si[7 + k] - dt * g[k];
    q_rot_inv(rv, q_i, w2);
    m3_vec(corr, dv_dbg, dbg);
    m3_vec(t3, dv_dba, dba);
    for (int k = 0; k < 3; ++k) rv[k] -= dv[k] + corr[k] + t3[k];
    for (int k = 0; k < 3; ++k) {
        raw[k] = rq[k], raw[3 + k] = rp[k], raw[6 + k] = rv[k];
        raw[9 + k] = sj[10 + k] - si[10 + k];                            // :82
        raw[12 + k] = sj[13 + k] - si[13 + k];                           // :83
    }
    if (!G) return;
    if (!G_is_zero)
        for (int k = 0; k < 450; ++k) G[k] = 0.0;
    auto put = [&](int row, int col, const double *m, double s) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) G[(row + i) * 30 + col + j] = s * m[3 * i + j];
    };
    double Jr[9], JrInv[9], Ri_T[9], Rci[9], Rcj[9], Rii_T[9], Rij_T[9], Rqj_T[9], M1[9], M2[9], qc[4], I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    so3_right_jacobian(Jr, rq);
    m3_inverse(JrInv, Jr);
    q_conj(qc, q_i);
    q_to_mat(Ri_T, qc);
    q_to_mat(Rci, si);
    q_to_mat(Rcj, sj);
    q_conj(qc, imu_i);
    q_to_mat(Rii_T, qc);
    q_conj(qc, imu_j);
    q_to_mat(Rij_T, qc);
    q_conj(qc, q_j);
    q_to_mat(Rqj_T, qc);
    // theta_i                                                            :86-92
    m3_mul(M1, Rqj_T, Rci);
    m3_mul(M2, JrInv, M1);
    put(0, 0, M2, -1.0);
    for (int k = 0; k < 3; ++k) w1[k] = p_j[k] - si[4 + k] - dt * si[7 + k] - 0.5 * dt * dt * g[k]; // p_center_i, :90
    q_rot_inv(t3, si, w1);
    m3_mul_hat(M1, Rii_T, t3);
    put(3, 0, M1, 1.0);
    q_rot_inv(t3, si, w2);
    m3_mul_hat(M1, Rii_T, t3);
    put(6, 0, M1, 1.0);
    put(3, 3, Ri_T, -1.0);                                               // p_i  :97
    put(3, 6, Ri_T, -dt);                                                // v_i  :103-104
    put(6, 6, Ri_T, -1.0);
    // bg_i                                                               :110-113
    double Rexp_T[9], Jr2[9];
    q_expmap(eq, rq);
    q_conj(qc, eq);
    q_to_mat(Rexp_T, qc);
    so3_right_jacobian(Jr2, th);
    m3_mul(M1, Jr2, dq_dbg);
    m3_mul(M2, Rexp_T, M1);
    m3_mul(M1, JrInv, M2);
    put(0, 9, M1, -1.0);
    put(3, 9, dp_dbg, -1.0);
    put(6, 9, dv_dbg, -1.0);
    put(9, 9, I3, -1.0);
    put(3, 12, dp_dba, -1.0);                                            // ba_i :119-121
    put(6, 12, dv_dba, -1.0);
    put(12, 12, I3, -1.0);
    // theta_j                                                            :127-128
    m3_mul(M1, JrInv, Rij_T);
    put(0, 15, M1, 1.0);
    m3_mul_hat(M1, Rcj, imu_j + 4);
    m3_mul(M2, Ri_T, M1);
    put(3, 15, M2, -1.0);
    put(3, 18, Ri_T, 1.0);                                               // p_j  :134
    put(6, 21, Ri_T, 1.0);                                               // v_j  :140
    put(9, 24, I3, 1.0);                                                 // bg_j :146
    put(12, 27, I3, 1.0);                                                // ba_j :152
}

// ---- A5: marginalization prior, per-frame error and its local Jacobian block --------------------------
// e[15] = [Log(q0^-1 q); p-p0; v-v0; bg-bg0; ba-ba0] (:65-69);  JrInv[9] = right_jacobian(e_q)^-1 (:77)
PV_HD void prior_frame_error(const double *x, const double *x0, double *e, double *JrInv) {
    double qc[4], dq[4], Jr[9];
    q_conj(qc, x0);
    q_mul(dq, qc, x);
    q_logmap(e, dq);
    for (int k = 0; k < 12; ++k) e[3 + k] = x[4 + k] - x0[4 + k];
    if (JrInv) {
        so3_right_jacobian(Jr, e);
        m3_inverse(JrInv, Jr);
    }
}

// ---- RotationPriorFactor: NO REFERENCE COUNTERPART (BASELINE.json names it; the class does not exist @ v0).  Defined as
// SURVEY.md section 8a prescribes -- the r_q rows of the marginalization factor (marginalization_error_cost.h:65,77) under
// a 3 x 3 sqrt-information W (row-major): r = W Log(q0^-1 q), J = W Jr^-1(Log(q0^-1 q)).  Outputs r[3], H = J^T J [9], g = J^T r [3].
PV_HD void rot_prior_eval(const double *q, const double *q0, const double *W, double *r, double *H, double *g) {
    double qc[4], dq[4], e[3], Jr[9], Ji[9], J[9];
    q_conj(qc, q0);
    q_mul(dq, qc, q);
    q_logmap(e, dq);
    so3_right_jacobian(Jr, e);
    m3_inverse(Ji, Jr);
    for (int i = 0; i < 3; ++i) {
        r[i] = W[3 * i] * e[0] + W[3 * i + 1] * e[1] + W[3 * i + 2] * e[2];
        for (int j = 0; j < 3; ++j) J[3 * i + j] = W[3 * i] * Ji[j] + W[3 * i + 1] * Ji[3 + j] + W[3 * i + 2] * Ji[6 + j];
    }
    for (int a = 0; a < 3; ++a) {
        g[a] = J[a] * r[0] + J[3 + a] * r[1] + J[6 + a] * r[2];
        for (int c = 0; c < 3; ++c) H[3 * a + c] = J[a] * J[c] + J[3 + a] * J[3 + c] + J[6 + a] * J[6 + c];
    }
}

// ---- A6: augmented plane-distance factor ------------------------------------------------------------------
// 3x3 symmetric pseudo-inverse via cyclic Jacobi, eigenvalues <= 1e-8 dropped (:90-92)
PV_HD void sym3_pinv(double *P, const double *Ain) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) A[k] = Ain[k];
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        const double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off <= 1e-300 || off <= 1e-34 * dg) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[3 * p + q];
                if (apq == 0.0) continue;
                const double theta = (A[4 * q] - A[4 * p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[3 * k + p], akq = A[3 * k + q];
                    A[3 * k + p] = c * akp - s * akq, A[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[3 * p + k], aqk = A[3 * q + k];
                    A[3 * p + k] = c * apk - s * aqk, A[3 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - s * vkq, V[3 * k + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int k = 0; k < 9; ++k) P[k] = 0.0;
    for (int e = 0; e < 3; ++e) {
        const double lam = A[4 * e];
        const double li = lam > 1.0e-8 ? 1.0 / lam : 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) P[3 * i + j] += V[3 * i + e] * li * V[3 * j + e];
    }
}

// rows of the triangulation system for one observation: Rsw (9), A rows a0,a1 (3 each), b0,b1 (:67-73)
PV_HD void plane_obs_rows(const double *rec, double u, double v, double *Rsw, double *a0, double *a1, double *b) {
    // Rsw = (R_b R_cs)^T ; Tsw = -Rsw p_b - R_cs^T p_cs
    double Rwc[9], Tsw[3], t[3];
    m3_mul(Rwc, rec, rec + 12);
    m3_transpose(Rsw, Rwc);
    m3_vec(Tsw, Rsw, rec + 9);
    m3_tvec(t, rec + 12, rec + 21);
    for (int k = 0; k < 3; ++k) Tsw[k] = -Tsw[k] - t[k];
    for (int j = 0; j < 3; ++j) {
        a0[j] = u * Rsw[6 + j] - Rsw[j];
        a1[j] = v * Rsw[6 + j] - Rsw[3 + j];
    }
    b[0] = u * Tsw[2] - Tsw[0];
    b[1] = v * Tsw[2] - Tsw[1];
}

// Two-pass evaluation without per-observation arrays.  `recs` = frame records (indexable by frame id),
// frames[k] / z[2k..] the K observations.  Writes r (whitened) and, when row != null, ADDS the 1x6 Jacobian of
// observation k (theta, p) into row[6*frames[k] .. +6) -- the dense 6N-wide row the tile accumulation consumes.
PV_HD void plane_eval_row(int K, const int *frames, const double *z, const double *recs, const double *normal, double distance,
                          double sqrt_inv_cov, double *r_out, double *row) {
    double ATA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ATb[3] = {0, 0, 0};
    auto acc = [&](const double *a, double b) {
        for (int i = 0; i < 3; ++i) {
            ATb[i] += a[i] * b;
            for (int j = 0; j < 3; ++j) ATA[3 * i + j] += a[i] * a[j];
        }
    };
    for (int k = 0; k < K; ++k) {
        double Rsw[9], a0[3], a1[3], b[2];
        plane_obs_rows(recs + kFrameRec * frames[k], z[2 * k], z[2 * k + 1], Rsw, a0, a1, b);
        acc(a0, b[0]);
        acc(a1, b[1]);
    }
    acc(normal, distance); // regularization row, weight 1 (:84-85)
    double P[9], x[3];
    sym3_pinv(P, ATA);
    m3_vec(x, P, ATb);
    x[0] = -x[0], x[1] = -x[1], x[2] = -x[2];                         // :94
    *r_out = (v3_dot(normal, x) - distance) * sqrt_inv_cov;           // :96,:133
    if (!row) return;
    for (int k = 0; k < K; ++k) {
        const double *rec = recs + kFrameRec * frames[k];
        const double u = z[2 * k], v = z[2 * k + 1];
        double Rsw[9], a0[3], a1[3], b[2];
        plane_obs_rows(rec, u, v, Rsw, a0, a1, b);
        const double Jb[6] = {-1, 0, u, 0, -1, v};                    // :100-102
        double dxdAdq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int rr = 0; rr < 2; ++rr) {
            const double *arow = rr ? a1 : a0;
            const double coef = b[rr] + v3_dot(arow, x);
            double aP[3], dxdA[9], cj[3], dAdq[9], prod[9];
            m3_tvec(aP, P, arow);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) dxdA[3 * i + j] = coef * P[3 * i + j] + aP[i] * x[j]; // :105-106
            m3_vec(cj, rec + 12, Jb + 3 * rr);                        // q_cs * Jb.row^T
            m3_mul_hat(dAdq, rec, cj);                                // R_b hat(.)        :107-108
            m3_mul(prod, dxdA, dAdq);
            for (int e = 0; e < 9; ++e) dxdAdq[e] += prod[e];         // :109
        }
        double PAJ[9]; // P * A_blk^T * Jb
        for (int i = 0; i < 3; ++i) {
            const double pa0 = P[3 * i] * a0[0] + P[3 * i + 1] * a0[1] + P[3 * i + 2] * a0[2];
            const double pa1 = P[3 * i] * a1[0] + P[3 * i + 1] * a1[1] + P[3 * i + 2] * a1[2];
            for (int j = 0; j < 3; ++j) PAJ[3 * i + j] = pa0 * Jb[j] + pa1 * Jb[3 + j];
        }
        double t[3], RcT_h[9], RcT[9], dxdbdq[9], tp[9];
        m3_tvec(t, rec, rec + 9);                                     // q_wc^-1 * p_wc
        m3_transpose(RcT, rec + 12);
        m3_mul_hat(RcT_h, RcT, t);
        m3_mul(dxdbdq, PAJ, RcT_h);                                   // :110
        m3_mul(tp, PAJ, Rsw);                                         // :117
        double *Jk = row + 6 * frames[k];
        for (int j = 0; j < 3; ++j) {
            Jk[j] += sqrt_inv_cov * (normal[0] * (dxdAdq[j] + dxdbdq[j]) + normal[1] * (dxdAdq[3 + j] + dxdbdq[3 + j]) +
                                     normal[2] * (dxdAdq[6 + j] + dxdbdq[6 + j]));              // :111-113
            Jk[3 + j] += sqrt_inv_cov * (normal[0] * tp[j] + normal[1] * tp[3 + j] + normal[2] * tp[6 + j]); // :117-118
        }
    }
}

} // namespace pv
