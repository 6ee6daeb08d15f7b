// oracle_factors.h -- CPU restatement of the reference's cost functors and pre-integrator.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Every function cites the reference lines it follows.
// PARITY UNPINNED: the reference ships no tests/golden vectors and cannot be built here (Eigen/Ceres
// absent); these restatements are validated by finite differences and invariants (tests/test_oracle_*.py).
#pragma once
#include "oracle_math.h"

namespace orc {

static const double kGravity = 9.80665; // PVIO_GRAVITY_NOMINAL, pvio/src/pvio/common.h:62

struct Ext {
    Q q;
    V3 p;
};
inline Ext ext_load(const double *e) { return Ext{qload(e), vload(e + 4)}; }

// ---------------------------------------------------------------------------------------------
// ReprojectionErrorCost::Evaluate -- estimation/ceres/reprojection_error_cost.h:40-120
// st_* = 16-double frame state (q xyzw, p, ...).  J is 2x13 row-major in LOCAL coordinates:
// [theta_tgt(3) p_tgt(3) theta_ref(3) p_ref(3) inv_depth(1)] (quaternion_parameterization.h:33-36
// makes the local Jacobian the first three columns of each 2x4 block).
// ---------------------------------------------------------------------------------------------
inline void eval_reprojection(const double *st_tgt, const double *st_ref, double inv_depth, const double *z_ref,
                              const double *z_tgt, const Ext &cam_ref, const Ext &cam_tgt, const double *W /*2x2*/,
                              double *r, double *J) {
    Q q_tgt = qload(st_tgt), q_ref = qload(st_ref);
    V3 p_tgt = vload(st_tgt + 4), p_ref = vload(st_ref + 4);
    V3 y_ref = mk(z_ref[0] / inv_depth, z_ref[1] / inv_depth, 1.0 / inv_depth);      // :58
    V3 y_rc = qrot(cam_ref.q, y_ref) + cam_ref.p;                                    // :59
    V3 x = qrot(q_ref, y_rc) + p_ref;                                                // :60
    V3 y_tc = qrot(qconj(q_tgt), x - p_tgt);                                         // :61
    V3 y_t = qrot(qconj(cam_tgt.q), y_tc - cam_tgt.p);                               // :62
    double r0 = y_t[0] / y_t[2] - z_tgt[0], r1 = y_t[1] / y_t[2] - z_tgt[1];         // :63
    if (J) {
        double iz = 1.0 / y_t[2], iz2 = 1.0 / (y_t[2] * y_t[2]);
        double d[2][3] = {{iz, 0, -y_t[0] * iz2}, {0, iz, -y_t[1] * iz2}};           // :66-68
        double Jp[2][3];
        for (int j = 0; j < 3; ++j) {                                                // :69  sqrt_inv_cov * dr_dy
            Jp[0][j] = W[0] * d[0][j] + W[1] * d[1][j];
            Jp[1][j] = W[2] * d[0][j] + W[3] * d[1][j];
        }
        M3 Rcs_t_T = qmat(qconj(cam_tgt.q)), Rt_T = qmat(qconj(q_tgt)), Rr = qmat(q_ref), Rcs_r = qmat(cam_ref.q);
        double A[2][3], B[2][3], C[2][3];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) A[i][j] = Jp[i][0] * Rcs_t_T.m[0][j] + Jp[i][1] * Rcs_t_T.m[1][j] + Jp[i][2] * Rcs_t_T.m[2][j]; // :75
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) B[i][j] = A[i][0] * Rt_T.m[0][j] + A[i][1] * Rt_T.m[1][j] + A[i][2] * Rt_T.m[2][j];             // :79
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) C[i][j] = B[i][0] * Rr.m[0][j] + B[i][1] * Rr.m[1][j] + B[i][2] * Rr.m[2][j];                   // :86
        M3 Hy_tc = hat(y_tc), Hy_rc = hat(y_rc);
        V3 ry = Rcs_r * y_ref;
        for (int i = 0; i < 2; ++i) {
            double *Ji = J + 13 * i;
            for (int j = 0; j < 3; ++j) {
                Ji[0 + j] = A[i][0] * Hy_tc.m[0][j] + A[i][1] * Hy_tc.m[1][j] + A[i][2] * Hy_tc.m[2][j];    // :95
                Ji[3 + j] = -B[i][j];                                                                        // :100
                Ji[6 + j] = -(C[i][0] * Hy_rc.m[0][j] + C[i][1] * Hy_rc.m[1][j] + C[i][2] * Hy_rc.m[2][j]); // :104
                Ji[9 + j] = B[i][j];                                                                         // :109
            }
            Ji[12] = -(C[i][0] * ry[0] + C[i][1] * ry[1] + C[i][2] * ry[2]) / inv_depth;                    // :113
        }
    }
    r[0] = W[0] * r0 + W[1] * r1; // :116
    r[1] = W[2] * r0 + W[3] * r1;
}

// ---------------------------------------------------------------------------------------------
// PreIntegrator -- estimation/preintegrator.{h,cpp}
// ---------------------------------------------------------------------------------------------
struct PreInt {
    double dt;
    Q dq;
    V3 dp, dv;
    double cov[15][15];
    double sqrt_inv_cov[15][15];
    M3 dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba;
};
struct ImuNoise {
    M3 cov_w, cov_a, cov_bg, cov_ba;
};
inline void preint_reset(PreInt &pi) { // preintegrator.cpp:24-37
    pi.dt = 0;
    pi.dq = Q{0, 0, 0, 1};
    pi.dp = pi.dv = mk(0, 0, 0);
    std::memset(pi.cov, 0, sizeof pi.cov);
    std::memset(pi.sqrt_inv_cov, 0, sizeof pi.sqrt_inv_cov);
    pi.dq_dbg = pi.dp_dbg = pi.dp_dba = pi.dv_dbg = pi.dv_dba = zero3();
}
inline void preint_increment(PreInt &pi, double dt, const V3 &dw, const V3 &da, const V3 &bg, const V3 &ba, const ImuNoise &nz) {
    // preintegrator.cpp:39-82
    V3 w = dw - bg, a = da - ba;
    M3 Rdq = qmat(pi.dq);
    M3 Rexp_T = qmat(qconj(expmap(dt * w)));
    M3 Jr = right_jacobian(dt * w);
    M3 Ra = Rdq * hat(a);
    {
        double A[9][9], B[9][6], Wn[6][6];
        std::memset(A, 0, sizeof A);
        std::memset(B, 0, sizeof B);
        std::memset(Wn, 0, sizeof Wn);
        for (int i = 0; i < 9; ++i) A[i][i] = 1;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                A[0 + i][0 + j] = Rexp_T.m[i][j];             // A(Q,Q)  :46
                A[6 + i][0 + j] = -dt * Ra.m[i][j];           // A(V,Q)  :47
                A[3 + i][0 + j] = -0.5 * dt * dt * Ra.m[i][j];// A(P,Q)  :48
                A[3 + i][6 + j] = (i == j) ? dt : 0.0;        // A(P,V)  :49
                B[0 + i][0 + j] = dt * Jr.m[i][j];            // :53
                B[6 + i][3 + j] = dt * Rdq.m[i][j];           // :54
                B[3 + i][3 + j] = 0.5 * dt * dt * Rdq.m[i][j];// :55
            }
        double inv_dt = 1.0 / std::fmax(dt, 1.0e-7);          // :58
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                Wn[i][j] = nz.cov_w.m[i][j] * inv_dt;
                Wn[3 + i][3 + j] = nz.cov_a.m[i][j] * inv_dt;
            }
        double AC[9][9], T[9][9], BW[9][6];
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) {
                double s = 0;
                for (int k = 0; k < 9; ++k) s += A[i][k] * pi.cov[k][j];
                AC[i][j] = s;
            }
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 6; ++j) {
                double s = 0;
                for (int k = 0; k < 6; ++k) s += B[i][k] * Wn[k][j];
                BW[i][j] = s;
            }
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) {
                double s = 0;
                for (int k = 0; k < 9; ++k) s += AC[i][k] * A[j][k];
                for (int k = 0; k < 6; ++k) s += BW[i][k] * B[j][k];
                T[i][j] = s;
            }
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) pi.cov[i][j] = T[i][j];                       // :63
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                pi.cov[9 + i][9 + j] += nz.cov_bg.m[i][j] * dt;                       // :64
                pi.cov[12 + i][12 + j] += nz.cov_ba.m[i][j] * dt;                     // :65
            }
    }
    // bias Jacobians, in the reference's statement order (:69-75)
    pi.dp_dbg = pi.dp_dbg + dt * pi.dv_dbg - (0.5 * dt * dt) * (Ra * pi.dq_dbg);
    pi.dp_dba = pi.dp_dba + dt * pi.dv_dba - (0.5 * dt * dt) * Rdq;
    pi.dv_dbg = pi.dv_dbg - dt * (Ra * pi.dq_dbg);
    pi.dv_dba = pi.dv_dba - dt * Rdq;
    pi.dq_dbg = Rexp_T * pi.dq_dbg - dt * Jr;
    // mean (:77-80)
    V3 Rda = qrot(pi.dq, a);
    pi.dt = pi.dt + dt;
    pi.dp = pi.dp + dt * pi.dv + (0.5 * dt * dt) * Rda;
    pi.dv = pi.dv + dt * Rda;
    pi.dq = qnormalized(qmul(pi.dq, expmap(dt * w)));
}
inline bool preint_sqrt_inv_cov(PreInt &pi) { // preintegrator.cpp:98-100: LLT(cov.inverse()).matrixL().transpose()
    double inv[225];
    if (!lu_inverse(&pi.cov[0][0], 15, inv)) return false;
    for (int i = 0; i < 15; ++i) // symmetrize like LLT does implicitly (reads the lower triangle only)
        for (int j = i + 1; j < 15; ++j) inv[i * 15 + j] = inv[j * 15 + i];
    if (!cholesky_lower(inv, 15, 15)) return false;
    for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) pi.sqrt_inv_cov[i][j] = (j >= i) ? inv[j * 15 + i] : 0.0;
    return true;
}
// PreIntegrator::integrate(t, bg, ba, true, true) -- preintegrator.cpp:84-96
inline bool preint_integrate(PreInt &pi, int n, const double *t, const double *w, const double *a, double t_end,
                             const V3 &bg, const V3 &ba, const ImuNoise &nz) {
    if (n == 0) return false;
    preint_reset(pi);
    for (int i = 0; i + 1 < n; ++i) preint_increment(pi, t[i + 1] - t[i], vload(w + 3 * i), vload(a + 3 * i), bg, ba, nz);
    preint_increment(pi, t_end - t[n - 1], vload(w + 3 * (n - 1)), vload(a + 3 * (n - 1)), bg, ba, nz);
    return preint_sqrt_inv_cov(pi);
}

// ---------------------------------------------------------------------------------------------
// PreIntegrationErrorCost::Evaluate -- estimation/ceres/preintegration_error_cost.h:40-160
// bg0/ba0 are the LIVE frame_i->motion.bg/ba the functor reads at evaluation time (:57-58).
// J is 15x30 row-major, local coordinates, columns = error state of frame i then frame j.
// ---------------------------------------------------------------------------------------------
struct PreIntFactor {
    double dt;
    Q dq;
    V3 dp, dv;
    const double *U; // 15x15 row-major sqrt_inv_cov
    M3 dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba;
};
inline M3 m3load(const double *p) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = p[3 * i + j];
    return r;
}
inline void eval_preintegration(const double *si, const double *sj, const V3 &bg0, const V3 &ba0, const PreIntFactor &pre,
                                const Ext &imu_i, const Ext &imu_j, double *r, double *J) {
    const V3 g = mk(0, 0, -kGravity);
    Q q_ci = qload(si), q_cj = qload(sj);
    V3 p_ci = vload(si + 4), v_i = vload(si + 7), bg_i = vload(si + 10), ba_i = vload(si + 13);
    V3 p_cj = vload(sj + 4), v_j = vload(sj + 7), bg_j = vload(sj + 10), ba_j = vload(sj + 13);
    Q q_i = qmul(q_ci, imu_i.q), q_j = qmul(q_cj, imu_j.q);                 // :60-63
    V3 p_i = p_ci + qrot(q_ci, imu_i.p), p_j = p_cj + qrot(q_cj, imu_j.p);
    double dt = pre.dt;
    V3 dbg = bg_i - bg0, dba = ba_i - ba0;                                   // :69-70
    double raw[15];
    V3 rq = logmap(qmul(qmul(qconj(qmul(pre.dq, expmap(pre.dq_dbg * dbg))), qconj(q_i)), q_j));           // :79
    V3 rp = qrot(qconj(q_i), p_j - p_i - dt * v_i - (0.5 * dt * dt) * g) - (pre.dp + pre.dp_dbg * dbg + pre.dp_dba * dba); // :80
    V3 rv = qrot(qconj(q_i), v_j - v_i - dt * g) - (pre.dv + pre.dv_dbg * dbg + pre.dv_dba * dba);        // :81
    V3 rbg = bg_j - bg_i, rba = ba_j - ba_i;                                                              // :82-83
    for (int k = 0; k < 3; ++k) raw[k] = rq[k], raw[3 + k] = rp[k], raw[6 + k] = rv[k], raw[9 + k] = rbg[k], raw[12 + k] = rba[k];
    if (J) {
        double G[15][30];
        std::memset(G, 0, sizeof G);
        auto put = [&](int row, int col, const M3 &m) {
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) G[row + i][col + j] = m.m[i][j];
        };
        M3 JrInv = inverse(right_jacobian(rq));
        M3 Ri_T = qmat(qconj(q_i)), Rci = qmat(q_ci), Rcj = qmat(q_cj);
        M3 Rimu_i_T = qmat(qconj(imu_i.q)), Rimu_j_T = qmat(qconj(imu_j.q));
        M3 I = eye3();
        // theta_i (:86-92)
        put(0, 0, -(JrInv * (qmat(qconj(q_j)) * Rci)));
        put(3, 0, Rimu_i_T * hat(qrot(qconj(q_ci), p_j - p_ci - dt * v_i - (0.5 * dt * dt) * g)));
        put(6, 0, Rimu_i_T * hat(qrot(qconj(q_ci), v_j - v_i - dt * g)));
        // p_i (:94-98)
        put(3, 3, -Ri_T);
        // v_i (:100-105)
        put(3, 6, (-dt) * Ri_T);
        put(6, 6, -Ri_T);
        // bg_i (:107-114)
        put(0, 9, -(JrInv * (qmat(qconj(expmap(rq))) * (right_jacobian(pre.dq_dbg * dbg) * pre.dq_dbg))));
        put(3, 9, -pre.dp_dbg);
        put(6, 9, -pre.dv_dbg);
        put(9, 9, -I);
        // ba_i (:116-122)
        put(3, 12, -pre.dp_dba);
        put(6, 12, -pre.dv_dba);
        put(12, 12, -I);
        // theta_j (:124-129)
        put(0, 15, JrInv * Rimu_j_T);
        put(3, 15, -(Ri_T * (Rcj * hat(imu_j.p))));
        // p_j, v_j, bg_j, ba_j (:131-153)
        put(3, 18, Ri_T);
        put(6, 21, Ri_T);
        put(9, 24, I);
        put(12, 27, I);
        for (int i = 0; i < 15; ++i)
            for (int j = 0; j < 30; ++j) {
                double s = 0;
                for (int k = 0; k < 15; ++k) s += pre.U[i * 15 + k] * G[k][j];
                J[i * 30 + j] = s;
            }
    }
    for (int i = 0; i < 15; ++i) { // :157
        double s = 0;
        for (int k = 0; k < 15; ++k) s += pre.U[i * 15 + k] * raw[k];
        r[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// MarginalizationErrorCost::Evaluate -- estimation/ceres/marginalization_error_cost.h:53-94
// states/lin: n pointers to 16-double frame states.  r: 15n.  J: 15n x 15n row-major (local coords).
// ---------------------------------------------------------------------------------------------
inline void eval_prior(int n, const double *const *states, const double *lin, const double *S, const double *s,
                       double *r, double *J) {
    int D = 15 * n;
    std::vector<double> e(D);
    for (int i = 0; i < n; ++i) {
        const double *x = states[i], *x0 = lin + 16 * i;
        V3 rq = logmap(qmul(qconj(qload(x0)), qload(x))); // :65
        for (int k = 0; k < 3; ++k) {
            e[15 * i + k] = rq[k];
            e[15 * i + 3 + k] = x[4 + k] - x0[4 + k];
            e[15 * i + 6 + k] = x[7 + k] - x0[7 + k];
            e[15 * i + 9 + k] = x[10 + k] - x0[10 + k];
            e[15 * i + 12 + k] = x[13 + k] - x0[13 + k];
        }
    }
    if (J) {
        for (int i = 0; i < n; ++i) {
            M3 JrInv = inverse(right_jacobian(mk(e[15 * i], e[15 * i + 1], e[15 * i + 2]))); // :77
            for (int row = 0; row < D; ++row) {
                const double *Srow = S + (size_t)row * D + 15 * i;
                double *Jrow = J + (size_t)row * D + 15 * i;
                for (int j = 0; j < 3; ++j) Jrow[j] = Srow[0] * JrInv.m[0][j] + Srow[1] * JrInv.m[1][j] + Srow[2] * JrInv.m[2][j];
                for (int j = 3; j < 15; ++j) Jrow[j] = Srow[j]; // :80-87
            }
        }
    }
    for (int row = 0; row < D; ++row) { // :90-91
        double acc = 0;
        for (int k = 0; k < D; ++k) acc += S[(size_t)row * D + k] * e[k];
        r[row] = acc + s[row];
    }
}

// ---------------------------------------------------------------------------------------------
// RotationPriorFactor -- NO REFERENCE COUNTERPART (the name is BASELINE.json's; the class does not exist @ v0).  Defined as
// SURVEY.md section 8a's name-mapping note prescribes: the r_q rows of MarginalizationErrorCost
// (marginalization_error_cost.h:65 residual, :77 Jacobian) under a 3 x 3 sqrt-information W (row-major):
//   r = W log(q0^-1 q),   J = W Jr^-1(log(q0^-1 q))   (3 x 3 row-major, columns = the frame's theta tangent)
// ---------------------------------------------------------------------------------------------
inline void eval_rot_prior(const double *state, const double *q0, const double *W, double *r, double *J) {
    V3 rq = logmap(qmul(qconj(qload(q0)), qload(state)));
    for (int i = 0; i < 3; ++i) r[i] = W[3 * i] * rq[0] + W[3 * i + 1] * rq[1] + W[3 * i + 2] * rq[2];
    if (J) {
        M3 JrInv = inverse(right_jacobian(rq));
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) J[3 * i + j] = W[3 * i] * JrInv.m[0][j] + W[3 * i + 1] * JrInv.m[1][j] + W[3 * i + 2] * JrInv.m[2][j];
    }
}

// ---------------------------------------------------------------------------------------------
// AugmentedPlaneDistanceErrorCost::Evaluate -- estimation/ceres/augmented_plane_distance_error_cost.h:53-136
// K observations; states[k] = 16-double state of the k-th observing frame; cams[k] its camera extrinsic.
// J: K x 6 row-major (theta, p per observing frame); plane blocks are constant (bundle_adjustor.cpp:108-109).
// ---------------------------------------------------------------------------------------------
inline void eval_plane(int K, const double *const *states, const Ext *cams, const double *z /*[K][2]*/, const double *normal,
                       double distance, double sqrt_inv_cov, double reg_w, double *r, double *J) {
    int R = 2 * K + 1;
    std::vector<double> A((size_t)R * 3), b(R);
    std::vector<M3> Rsw(K);
    for (int i = 0; i < K; ++i) {
        Q qwc = qload(states[i]);
        V3 pwc = vload(states[i] + 4);
        Rsw[i] = qmat(qmul(qconj(cams[i].q), qconj(qwc)));          // :67
        V3 Tsw = -(Rsw[i] * pwc) - qrot(qconj(cams[i].q), cams[i].p); // :68
        for (int j = 0; j < 3; ++j) {
            A[(2 * i) * 3 + j] = z[2 * i] * Rsw[i].m[2][j] - Rsw[i].m[0][j];         // :70
            A[(2 * i + 1) * 3 + j] = z[2 * i + 1] * Rsw[i].m[2][j] - Rsw[i].m[1][j]; // :71
        }
        b[2 * i] = z[2 * i] * Tsw[2] - Tsw[0];
        b[2 * i + 1] = z[2 * i + 1] * Tsw[2] - Tsw[1];
    }
    for (int j = 0; j < 3; ++j) A[(2 * K) * 3 + j] = reg_w * normal[j]; // :84
    b[2 * K] = reg_w * distance;                                       // :85
    double ATA[9] = {0}, ATb[3] = {0};
    for (int k = 0; k < R; ++k)
        for (int i = 0; i < 3; ++i) {
            ATb[i] += A[k * 3 + i] * b[k];
            for (int j = 0; j < 3; ++j) ATA[i * 3 + j] += A[k * 3 + i] * A[k * 3 + j];
        }
    double ev[3], V[9];
    sym_eig(ATA, 3, ev, V); // :90
    M3 P = zero3();
    for (int k = 0; k < 3; ++k) {
        double li = ev[k] > 1.0e-8 ? 1.0 / ev[k] : 0.0; // :91
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) P.m[i][j] += V[i * 3 + k] * li * V[j * 3 + k];
    }
    V3 x = -(P * mk(ATb[0], ATb[1], ATb[2])); // :94
    V3 nrm = mk(normal[0], normal[1], normal[2]);
    double res = dot(nrm, x) - distance; // :96
    if (J) {
        for (int i = 0; i < K; ++i) {
            Q qwc = qload(states[i]);
            V3 pwc = vload(states[i] + 4);
            double Jb[2][3] = {{-1, 0, z[2 * i]}, {0, -1, z[2 * i + 1]}}; // :100-102
            M3 dxdAdq = zero3();
            M3 Rwc = qmat(qwc);
            for (int rr = 0; rr < 2; ++rr) {
                V3 arow = mk(A[(2 * i + rr) * 3], A[(2 * i + rr) * 3 + 1], A[(2 * i + rr) * 3 + 2]);
                double coef = b[2 * i + rr] + dot(arow, x);
                // (x * arow * P)^T  -> M[i][j] = (arow P)[i] * x[j]
                V3 aP = transpose(P) * arow;
                M3 dxdA;
                for (int a2 = 0; a2 < 3; ++a2)
                    for (int c2 = 0; c2 < 3; ++c2) dxdA.m[a2][c2] = coef * P.m[a2][c2] + aP[a2] * x[c2]; // :105-106
                M3 dAdq = Rwc * hat(qrot(cams[i].q, mk(Jb[rr][0], Jb[rr][1], Jb[rr][2])));            // :107-108
                dxdAdq = dxdAdq + dxdA * dAdq;                                                          // :109
            }
            // P * A_blk^T * Jb  (3x3)
            M3 PAJ;
            for (int a2 = 0; a2 < 3; ++a2)
                for (int c2 = 0; c2 < 3; ++c2) {
                    double s2 = 0;
                    for (int rr = 0; rr < 2; ++rr) {
                        double pa = 0;
                        for (int k = 0; k < 3; ++k) pa += P.m[a2][k] * A[(2 * i + rr) * 3 + k];
                        s2 += pa * Jb[rr][c2];
                    }
                    PAJ.m[a2][c2] = s2;
                }
            M3 dxdbdq = PAJ * (transpose(qmat(cams[i].q)) * hat(qrot(qconj(qwc), pwc))); // :110
            M3 tq = dxdAdq + dxdbdq;
            M3 tp = PAJ * Rsw[i]; // :117
            for (int j = 0; j < 3; ++j) {
                J[i * 6 + j] = sqrt_inv_cov * (nrm[0] * tq.m[0][j] + nrm[1] * tq.m[1][j] + nrm[2] * tq.m[2][j]);     // :111-113
                J[i * 6 + 3 + j] = sqrt_inv_cov * (nrm[0] * tp.m[0][j] + nrm[1] * tp.m[1][j] + nrm[2] * tp.m[2][j]); // :117-118
            }
        }
    }
    r[0] = res * sqrt_inv_cov; // :133
}

} // namespace orc
