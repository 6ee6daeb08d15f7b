// oracle_pnp.cpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into or called by the product).
//
// visual_inertial_pnp, pvio/src/pvio/estimation/pnp.cpp:32-100: one free frame (q, p and -- when inertial -- v, bg, ba) against
//   PoseOnlyReprojectionErrorCost      estimation/ceres/reprojection_error_cost.h:128-157  (CauchyLoss(1); = the target-frame columns of
//                                      ReprojectionErrorCost with the anchor frame and the inverse depth held fixed)
//   PoseOnlyReprojectionXYZErrorCost   reprojection_error_cost.h:159-203                   (CauchyLoss(1); fixed world point, the plane branch :61-88)
//   PreIntegrationPriorCost            estimation/ceres/preintegration_error_cost.h:167-206 (no loss; the map's last frame held fixed)
// minimized by ceres::Solve as configured in solver_options.h:26-33.  Ceres (pinned 1.14.0, pvio/depends/CMakeLists.txt:31-35)
// is not in /root/reference: TrustRegionMinimizer + DoglegStrategy(TRADITIONAL_DOGLEG) + Jacobi scaling + the Cauchy corrector
// are restated from the published 1.14 algorithm, here for a DENSE problem with a single parameter group (no Schur elimination:
// there is no e-block) -- the same semantics as oracle_ba.cpp, written separately on plain row-major arrays.  PARITY UNPINNED.
//
// This is the C++ counterpart of the numpy loop tests/np_reference.solve_dense; tests/test_oracle_pnp.py holds the three
// (C++ oracle, numpy loop, the product's pvio_amd/host/pnp.cpp + dense_minimizer.h) against each other.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "oracle_factors.h"

using namespace orc;

namespace {

struct Pnp {
    Ext cam, imu, last_imu;
    const double *W;
    int n, n_pts, inertial;
    const double *anchor_states, *anchor_cams, *z_ref, *z_tgt, *rho, *points, *z_pts;
    const double *last_state;
    PreIntFactor pre;
    int ncols() const { return inertial ? 15 : 6; }
    int nrows() const { return 2 * n + 2 * n_pts + (inertial ? 15 : 0); }
};

// reprojection_error_cost.h:159-203
void eval_point(const double *st, const double *X, const double *z, const Ext &cam, const double *W, double *r, double *J /* 2 x 6 or null */) {
    const Q q = qload(st);
    const V3 p = vload(st + 4);
    const V3 yc = qrot(qconj(q), vload(X) - p), y = qrot(qconj(cam.q), yc - cam.p);
    const double e0 = y[0] / y[2] - z[0], e1 = y[1] / y[2] - z[1];
    r[0] = W[0] * e0 + W[1] * e1, r[1] = W[2] * e0 + W[3] * e1;
    if (!J) return;
    const double iz = 1.0 / y[2];
    const double d[2][3] = {{iz, 0.0, -y[0] * iz * iz}, {0.0, iz, -y[1] * iz * iz}};
    double Wd[2][3];
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 3; ++k) Wd[i][k] = W[2 * i] * d[0][k] + W[2 * i + 1] * d[1][k];
    const M3 Rc = transpose(qmat(cam.q)), A = Rc * hat(yc), B = Rc * transpose(qmat(q));
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 3; ++k) {
            double a = 0, b = 0;
            for (int m = 0; m < 3; ++m) a += Wd[i][m] * A.m[m][k], b += Wd[i][m] * B.m[m][k];
            J[6 * i + k] = a, J[6 * i + 3 + k] = -b;
        }
}

// cost = sum 0.5 rho(|r_b|^2); r / J are the Ceres-corrected rows (Corrector with rho'' <= 0 for Cauchy: rows scaled by sqrt(rho'))
double evaluate(const Pnp &P, const double *x, const double *user_bias, std::vector<double> *r_out, std::vector<double> *J_out) {
    const int nc = P.ncols(), nr = P.nrows();
    if (r_out) r_out->assign((size_t)nr, 0.0);
    if (J_out) J_out->assign((size_t)nr * nc, 0.0);
    double cost = 0;
    int row = 0;
    if (P.inertial) {
        double r[15], J[15 * 30];
        eval_preintegration(P.last_state, x, vload(user_bias), vload(user_bias + 3), P.pre, P.last_imu, P.imu, r, J_out ? J : nullptr);
        double s = 0;
        for (int i = 0; i < 15; ++i) s += r[i] * r[i];
        cost += 0.5 * s;
        for (int i = 0; i < 15; ++i) {
            if (r_out) (*r_out)[(size_t)row + i] = r[i];
            if (J_out)
                for (int k = 0; k < 15; ++k) (*J_out)[(size_t)(row + i) * nc + k] = J[30 * i + 15 + k];
        }
        row += 15;
    }
    auto robust_rows = [&](const double *r, const double *J6) {
        const double s = r[0] * r[0] + r[1] * r[1], w = std::sqrt(1.0 / (1.0 + s));
        cost += 0.5 * std::log1p(s);
        for (int i = 0; i < 2; ++i) {
            if (r_out) (*r_out)[(size_t)row + i] = w * r[i];
            if (J_out)
                for (int k = 0; k < 6; ++k) (*J_out)[(size_t)(row + i) * nc + k] = w * J6[6 * i + k];
        }
        row += 2;
    };
    for (int f = 0; f < P.n; ++f) {
        double r[2], J[2 * 13], J6[12] = {0};
        Ext ca;
        ca.q = qload(P.anchor_cams + 7 * f), ca.p = vload(P.anchor_cams + 7 * f + 4);
        eval_reprojection(x, P.anchor_states + 16 * f, P.rho[f], P.z_ref + 2 * f, P.z_tgt + 2 * f, ca, P.cam, P.W, r, J_out ? J : nullptr);
        if (J_out)
            for (int i = 0; i < 2; ++i)
                for (int k = 0; k < 6; ++k) J6[6 * i + k] = J[13 * i + k];
        robust_rows(r, J6);
    }
    for (int f = 0; f < P.n_pts; ++f) {
        double r[2], J6[12] = {0};
        eval_point(x, P.points + 3 * f, P.z_pts + 2 * f, P.cam, P.W, r, J_out ? J6 : nullptr);
        robust_rows(r, J6);
    }
    return cost;
}

void plus(const Pnp &P, const double *x, const double *delta, double *out) {
    Q q = qnormalized(qmul(qload(x), expmap(mk(delta[0], delta[1], delta[2]))));
    qstore(q, out);
    for (int k = 0; k < 3; ++k) out[4 + k] = x[4 + k] + delta[3 + k];
    for (int k = 7; k < 16; ++k) out[k] = x[k] + (P.inertial ? delta[6 + (k - 7)] : 0.0);
}
int n_ambient(const Pnp &P) { return P.inertial ? 16 : 7; }

} // namespace

extern "C" int32_t oracle_pnp_flat(const double *cam, const double *imu, const double *W, int32_t n, const double *anchor_states, const double *anchor_cams,
                                   const double *z_ref, const double *z_tgt, const double *rho, int32_t n_pts, const double *points, const double *z_pts,
                                   int32_t use_inertial, const double *last_state, const double *last_imu, const double *delta, const double *U, const double *jac,
                                   int32_t max_iter, double *state16, int32_t *iterations, int32_t *termination, double *costs2) {
    Pnp P;
    P.cam.q = qload(cam), P.cam.p = vload(cam + 4), P.imu.q = qload(imu), P.imu.p = vload(imu + 4);
    P.W = W, P.n = n, P.n_pts = n_pts, P.inertial = use_inertial ? 1 : 0;
    P.anchor_states = anchor_states, P.anchor_cams = anchor_cams, P.z_ref = z_ref, P.z_tgt = z_tgt, P.rho = rho, P.points = points, P.z_pts = z_pts;
    P.last_state = last_state;
    if (P.inertial) {
        P.last_imu.q = qload(last_imu), P.last_imu.p = vload(last_imu + 4);
        P.pre.dt = delta[0], P.pre.dq = qload(delta + 1), P.pre.dp = vload(delta + 5), P.pre.dv = vload(delta + 8);
        P.pre.U = U;
        P.pre.dq_dbg = m3load(jac), P.pre.dp_dbg = m3load(jac + 9), P.pre.dp_dba = m3load(jac + 18), P.pre.dv_dbg = m3load(jac + 27), P.pre.dv_dba = m3load(jac + 36);
    }
    const int nc = P.ncols(), nr = P.nrows(), na = n_ambient(P);
    *iterations = 0, *termination = 0;
    if (nr == 0) { // nothing to minimize
        costs2[0] = costs2[1] = 0.0;
        return 0;
    }
    // the bias the pre-integration was linearized at is read from the LAST frame (held fixed): no live-bias quirk here
    double bias0[6];
    for (int k = 0; k < 6; ++k) bias0[k] = P.inertial ? last_state[10 + k] : 0.0;

    std::vector<double> x(state16, state16 + 16), best(x), r, J, cand(16);
    double radius = 1e4, mu = 1e-8, x_cost = evaluate(P, x.data(), bias0, &r, &J);
    costs2[0] = x_cost;
    bool reuse = false, success = true;
    int invalid = 0, it = 0, term = 1;
    std::vector<double> scale((size_t)nc), g_unscaled((size_t)nc), diag((size_t)nc), ghat((size_t)nc), gn((size_t)nc), step((size_t)nc), deltav(15, 0.0);
    auto col_sq = [&](int k) {
        double s = 0;
        for (int i = 0; i < nr; ++i) s += J[(size_t)i * nc + k] * J[(size_t)i * nc + k];
        return s;
    };
    auto JTr = [&](std::vector<double> &out) {
        for (int k = 0; k < nc; ++k) {
            double s = 0;
            for (int i = 0; i < nr; ++i) s += J[(size_t)i * nc + k] * r[(size_t)i];
            out[(size_t)k] = s;
        }
    };
    JTr(g_unscaled);
    for (int k = 0; k < nc; ++k) scale[(size_t)k] = 1.0 / (1.0 + std::sqrt(col_sq(k))); // jacobi_scaling, once
    auto apply_scale = [&]() {
        for (int i = 0; i < nr; ++i)
            for (int k = 0; k < nc; ++k) J[(size_t)i * nc + k] *= scale[(size_t)k];
    };
    apply_scale();
    auto grad_max = [&](const std::vector<double> &xx, const std::vector<double> &g) { // max |x - Plus(x, -g)| over the ambient coordinates
        double d15[15] = {0}, out[16], m = 0;
        for (int k = 0; k < nc; ++k) d15[k] = -g[(size_t)k];
        plus(P, xx.data(), d15, out);
        for (int k = 0; k < na; ++k) m = std::max(m, std::fabs(xx[(size_t)k] - out[k]));
        return m;
    };
    auto amb_norm = [&](const std::vector<double> &xx) {
        double s = 0;
        for (int k = 0; k < na; ++k) s += xx[(size_t)k] * xx[(size_t)k];
        return std::sqrt(s);
    };
    double gmax = grad_max(x, g_unscaled), x_norm = amb_norm(x), min_cost = std::numeric_limits<double>::infinity(), alpha = 0, step_norm_dl = 0, model_change = 0;
    while (true) {
        if (success && x_cost < min_cost) min_cost = x_cost, best = x;
        if (it >= max_iter) {
            term = 1; // NO_CONVERGENCE
            break;
        }
        if (success && gmax <= 1e-10) {
            term = 0;
            break;
        }
        if (radius <= 1e-32) {
            term = 0;
            break;
        }
        ++it;
        success = false;
        bool ok = true;
        if (!reuse) {
            reuse = true;
            for (int k = 0; k < nc; ++k) diag[(size_t)k] = std::sqrt(std::min(std::max(col_sq(k), 1e-6), 1e32));
            std::vector<double> g((size_t)nc);
            JTr(g);
            for (int k = 0; k < nc; ++k) ghat[(size_t)k] = g[(size_t)k] / diag[(size_t)k];
            double gg = 0, jg2 = 0;
            for (int k = 0; k < nc; ++k) gg += ghat[(size_t)k] * ghat[(size_t)k];
            for (int i = 0; i < nr; ++i) {
                double s = 0;
                for (int k = 0; k < nc; ++k) s += J[(size_t)i * nc + k] * (ghat[(size_t)k] / diag[(size_t)k]);
                jg2 += s * s;
            }
            alpha = gg / jg2;
            ok = false;
            std::vector<double> H((size_t)nc * nc), A((size_t)nc * nc), y((size_t)nc);
            for (int a = 0; a < nc; ++a)
                for (int b = 0; b <= a; ++b) {
                    double s = 0;
                    for (int i = 0; i < nr; ++i) s += J[(size_t)i * nc + a] * J[(size_t)i * nc + b];
                    H[(size_t)a * nc + b] = H[(size_t)b * nc + a] = s;
                }
            while (mu < 1.0) {
                A = H;
                for (int k = 0; k < nc; ++k) A[(size_t)k * nc + k] += mu * diag[(size_t)k] * diag[(size_t)k];
                y = g;
                bool fin = false;
                if (cholesky_lower(A.data(), nc, nc)) {
                    cholesky_solve(A.data(), nc, nc, y.data());
                    fin = true;
                    for (int k = 0; k < nc; ++k) fin = fin && std::isfinite(y[(size_t)k]);
                }
                if (fin) {
                    ok = true;
                    break;
                }
                mu *= 10.0;
            }
            if (ok)
                for (int k = 0; k < nc; ++k) gn[(size_t)k] = -diag[(size_t)k] * y[(size_t)k];
        }
        if (ok) {
            double g2 = 0, gn2 = 0, gdot = 0;
            for (int k = 0; k < nc; ++k) g2 += ghat[(size_t)k] * ghat[(size_t)k], gn2 += gn[(size_t)k] * gn[(size_t)k], gdot += ghat[(size_t)k] * gn[(size_t)k];
            const double gnorm = std::sqrt(g2), gnn = std::sqrt(gn2);
            if (gnn <= radius) {
                step = gn, step_norm_dl = gnn;
            } else if (gnorm * alpha >= radius) {
                for (int k = 0; k < nc; ++k) step[(size_t)k] = -(radius / gnorm) * ghat[(size_t)k];
                step_norm_dl = radius;
            } else {
                const double b_dot_a = -alpha * gdot, a2 = (alpha * gnorm) * (alpha * gnorm), bma2 = a2 - 2 * b_dot_a + gn2, c = b_dot_a - a2;
                const double d = std::sqrt(c * c + bma2 * (radius * radius - a2));
                const double beta = c <= 0 ? (d - c) / bma2 : (radius * radius - a2) / (d + c);
                double s2 = 0;
                for (int k = 0; k < nc; ++k) step[(size_t)k] = (-alpha * (1 - beta)) * ghat[(size_t)k] + beta * gn[(size_t)k], s2 += step[(size_t)k] * step[(size_t)k];
                step_norm_dl = std::sqrt(s2);
            }
            for (int k = 0; k < nc; ++k) step[(size_t)k] /= diag[(size_t)k];
            model_change = 0;
            for (int i = 0; i < nr; ++i) {
                double mr = 0;
                for (int k = 0; k < nc; ++k) mr += J[(size_t)i * nc + k] * step[(size_t)k];
                model_change -= mr * (r[(size_t)i] + mr / 2.0);
            }
        }
        if (!ok || !(model_change > 0)) {
            if (++invalid >= 5) {
                term = 2; // FAILURE
                break;
            }
            mu *= 10.0, reuse = false;
            continue;
        }
        invalid = 0;
        for (int k = 0; k < nc; ++k) deltav[(size_t)k] = step[(size_t)k] * scale[(size_t)k];
        plus(P, x.data(), deltav.data(), cand.data());
        double cand_cost = evaluate(P, cand.data(), bias0, nullptr, nullptr);
        if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
        double sn = 0;
        for (int k = 0; k < na; ++k) sn += (x[(size_t)k] - cand[(size_t)k]) * (x[(size_t)k] - cand[(size_t)k]);
        if (std::sqrt(sn) <= 1e-8 * (x_norm + 1e-8)) { // parameter tolerance: the candidate is dropped
            term = 0;
            break;
        }
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= 1e-6 * x_cost) { // function tolerance: the candidate is dropped
            term = 0;
            break;
        }
        const double rel = cost_change / model_change;
        if (rel > 1e-3) {
            x = cand;
            x_norm = amb_norm(x);
            x_cost = evaluate(P, x.data(), bias0, &r, &J);
            JTr(g_unscaled);
            apply_scale();
            gmax = grad_max(x, g_unscaled);
            success = true;
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = std::max(radius, 3.0 * step_norm_dl);
            mu = std::max(1e-8, 2.0 * mu / 10.0);
            reuse = false;
        } else {
            radius *= 0.5, reuse = true;
        }
    }
    for (int k = 0; k < 16; ++k) state16[k] = best[(size_t)k];
    *iterations = it, *termination = term;
    costs2[1] = min_cost < std::numeric_limits<double>::infinity() ? min_cost : x_cost;
    return 0;
}
