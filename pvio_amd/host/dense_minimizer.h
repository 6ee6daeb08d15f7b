// dense_minimizer.h -- host trust-region minimizer for the small problems of the seam that are not worth a kernel
// (visual_inertial_pnp: one frame, <= 15 tangent dimensions, a few hundred residual rows).
//
// It restates what `ceres::Solve` does under the reference's options (pvio/src/pvio/estimation/ceres/solver_options.h:
// 26-33: DOGLEG, max_num_iterations, update_state_every_iteration; Ceres pinned 1.14.0 at
// pvio/depends/CMakeLists.txt:31-35, source not in the tree) for a problem without eliminated blocks, i.e. the linear
// solve is a dense Cholesky of J^T J + mu D^2.  SURVEY.md App. B lists the semantics: Jacobi scaling computed once from
// the first Jacobian (1 / (1 + ||col||)), TRADITIONAL_DOGLEG with radius 1e4, mu 1e-8 (x10 on a failed factorization or
// an invalid step, max(1e-8, 2 mu / 10) after an accepted step), min_relative_decrease 1e-3, function / parameter /
// gradient tolerances 1e-6 / 1e-8 / 1e-10 (the tolerance exits drop the candidate), five consecutive invalid steps =
// failure, the best accepted point is returned.
#pragma once
#include "host_namespace.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

namespace pvio {
namespace dense {

struct Summary {
    int iterations = 0, successful_steps = 0;
    int termination = 1; // 0 convergence, 1 no convergence (iteration limit), 2 failure
    double initial_cost = 0, final_cost = 0;
    bool usable() const { return termination != 2; }
};

// Problem concept:
//   int ambient() const, tangent() const;
//   bool evaluate(const double *x, double &cost, std::vector<double> *r, std::vector<double> *J /* rows x tangent, row-major */) const;
//       (r and J are the robustified residuals / Jacobian; cost = sum of 1/2 rho(s)); false = evaluation failed
//   void plus(const double *x, const double *delta, double *x_plus_delta) const;
template <class Problem>
Summary minimize(const Problem &pb, double *x, int max_iterations) {
    const int na = pb.ambient(), n = pb.tangent();
    Summary sum;
    std::vector<double> r, J, cand(na), best(x, x + na), scale(n), diag(n), ghat(n), gn(n), step(n), delta(n), g(n), tmpx(na);
    double x_cost = 0;
    if (n == 0 || !pb.evaluate(x, x_cost, &r, &J)) {
        sum.termination = n == 0 ? 0 : 2;
        return sum;
    }
    sum.initial_cost = sum.final_cost = x_cost;
    int rows = (int)r.size();
    // The sums over the rows of J walk J row by row (it is rows x n row-major, n <= 15: a walk down one column touches a cache line per
    // element -- with 1500 factors that was most of a PnP's 2.5 ms); every entry is still summed over the rows in ascending order.
    std::vector<double> cn2(n);
    auto col_norms2 = [&]() {
        std::fill(cn2.begin(), cn2.end(), 0.0);
        for (int i = 0; i < rows; ++i) {
            const double *Ji = &J[(size_t)i * n];
            for (int c = 0; c < n; ++c) cn2[c] += Ji[c] * Ji[c];
        }
    };
    auto apply_scale = [&]() {
        for (int i = 0; i < rows; ++i)
            for (int c = 0; c < n; ++c) J[(size_t)i * n + c] *= scale[c];
    };
    auto gradient = [&](std::vector<double> &out) { // J^T r with the current (possibly scaled) J
        std::fill(out.begin(), out.begin() + n, 0.0);
        for (int i = 0; i < rows; ++i) {
            const double *Ji = &J[(size_t)i * n];
            const double ri = r[i];
            for (int c = 0; c < n; ++c) out[c] += Ji[c] * ri;
        }
    };
    auto norm = [](const double *a, int m) {
        double s = 0;
        for (int i = 0; i < m; ++i) s += a[i] * a[i];
        return std::sqrt(s);
    };
    auto grad_max = [&](const double *xx) { // max | x - (x (+) -g) | with the UNscaled gradient
        std::vector<double> ng(n);
        for (int c = 0; c < n; ++c) ng[c] = -g[c];
        pb.plus(xx, ng.data(), tmpx.data());
        double m = 0;
        for (int i = 0; i < na; ++i) m = std::max(m, std::fabs(xx[i] - tmpx[i]));
        return m;
    };
    gradient(g); // unscaled
    col_norms2();
    for (int c = 0; c < n; ++c) scale[c] = 1.0 / (1.0 + std::sqrt(cn2[c]));
    apply_scale();
    double gmax = grad_max(x), x_norm = norm(x, na);
    double radius = 1e4, mu = 1e-8, min_cost = DBL_MAX, alpha = 0, dogleg_norm = 0;
    bool reuse = false, success = true;
    int invalid = 0, it = 0;
    std::vector<double> A((size_t)n * n), JtJ((size_t)n * n), rhs(n), y(n);
    while (true) {
        if (success && x_cost < min_cost) {
            min_cost = x_cost;
            std::copy(x, x + na, best.begin());
        }
        if (it >= max_iterations) {
            sum.termination = 1;
            break;
        }
        if (success && gmax <= 1e-10) {
            sum.termination = 0;
            break;
        }
        if (radius <= 1e-32) {
            sum.termination = 0;
            break;
        }
        ++it;
        success = false;
        bool ok = true;
        double model_change = 0;
        if (!reuse) {
            reuse = true;
            col_norms2();
            for (int c = 0; c < n; ++c) diag[c] = std::sqrt(std::min(std::max(cn2[c], 1e-6), 1e32));
            gradient(rhs); // scaled J^T r
            for (int c = 0; c < n; ++c) ghat[c] = rhs[c] / diag[c];
            double jg2 = 0; // |J (ghat / D)|^2
            for (int i = 0; i < rows; ++i) {
                double s = 0;
                for (int c = 0; c < n; ++c) s += J[(size_t)i * n + c] * (ghat[c] / diag[c]);
                jg2 += s * s;
            }
            double g2 = 0;
            for (int c = 0; c < n; ++c) g2 += ghat[c] * ghat[c];
            alpha = g2 / jg2;
            ok = false;
            std::fill(JtJ.begin(), JtJ.end(), 0.0); // lower triangle of J^T J, once per linearization (a retry only changes mu)
            for (int i = 0; i < rows; ++i) {
                const double *Ji = &J[(size_t)i * n];
                for (int a = 0; a < n; ++a) {
                    const double ja = Ji[a];
                    double *row = &JtJ[(size_t)a * n];
                    for (int b = 0; b <= a; ++b) row[b] += ja * Ji[b];
                }
            }
            while (mu < 1.0) {
                for (int a = 0; a < n; ++a)
                    for (int b = 0; b <= a; ++b) A[(size_t)a * n + b] = JtJ[(size_t)a * n + b] + (a == b ? mu * diag[a] * diag[a] : 0.0);
                bool spd = true; // in-place Cholesky, lower
                for (int a = 0; a < n && spd; ++a) {
                    for (int b = 0; b <= a; ++b) {
                        double s = A[(size_t)a * n + b];
                        for (int k = 0; k < b; ++k) s -= A[(size_t)a * n + k] * A[(size_t)b * n + k];
                        if (a == b) {
                            if (!(s > 0.0) || !std::isfinite(s)) spd = false;
                            else A[(size_t)a * n + a] = std::sqrt(s);
                        } else {
                            A[(size_t)a * n + b] = s / A[(size_t)b * n + b];
                        }
                    }
                }
                if (spd) {
                    for (int a = 0; a < n; ++a) { // L z = rhs
                        double s = rhs[a];
                        for (int k = 0; k < a; ++k) s -= A[(size_t)a * n + k] * y[k];
                        y[a] = s / A[(size_t)a * n + a];
                    }
                    for (int a = n - 1; a >= 0; --a) { // L^T y = z
                        double s = y[a];
                        for (int k = a + 1; k < n; ++k) s -= A[(size_t)k * n + a] * y[k];
                        y[a] = s / A[(size_t)a * n + a];
                    }
                    bool fin = true;
                    for (int a = 0; a < n; ++a) fin = fin && std::isfinite(y[a]);
                    if (fin) {
                        ok = true;
                        break;
                    }
                }
                mu *= 10.0;
            }
            if (ok)
                for (int c = 0; c < n; ++c) gn[c] = -diag[c] * y[c];
        }
        if (ok) {
            const double gnorm = norm(ghat.data(), n), gnn = norm(gn.data(), n);
            if (gnn <= radius) {
                step = gn, dogleg_norm = gnn;
            } else if (gnorm * alpha >= radius) {
                for (int c = 0; c < n; ++c) step[c] = -(radius / gnorm) * ghat[c];
                dogleg_norm = radius;
            } else {
                double dot = 0;
                for (int c = 0; c < n; ++c) dot += ghat[c] * gn[c];
                const double b_dot_a = -alpha * dot, a2 = (alpha * gnorm) * (alpha * gnorm);
                const double bma2 = a2 - 2 * b_dot_a + gnn * gnn, cc = b_dot_a - a2;
                const double dd = std::sqrt(cc * cc + bma2 * (radius * radius - a2));
                const double beta = cc <= 0 ? (dd - cc) / bma2 : (radius * radius - a2) / (dd + cc);
                for (int c = 0; c < n; ++c) step[c] = (-alpha * (1 - beta)) * ghat[c] + beta * gn[c];
                dogleg_norm = norm(step.data(), n);
            }
            for (int c = 0; c < n; ++c) step[c] /= diag[c];
            for (int i = 0; i < rows; ++i) { // -(J s)^T (r + J s / 2)
                double mr = 0;
                for (int c = 0; c < n; ++c) mr += J[(size_t)i * n + c] * step[c];
                model_change -= mr * (r[i] + 0.5 * mr);
            }
        }
        if (!ok || !(model_change > 0)) {
            if (++invalid >= 5) {
                sum.termination = 2;
                break;
            }
            mu *= 10.0;
            reuse = false;
            continue;
        }
        invalid = 0;
        for (int c = 0; c < n; ++c) delta[c] = step[c] * scale[c];
        pb.plus(x, delta.data(), cand.data());
        double cand_cost = 0;
        if (!pb.evaluate(cand.data(), cand_cost, nullptr, nullptr) || !std::isfinite(cand_cost)) cand_cost = DBL_MAX;
        double sn = 0;
        for (int i = 0; i < na; ++i) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
        sn = std::sqrt(sn);
        if (sn <= 1e-8 * (x_norm + 1e-8)) {
            sum.termination = 0;
            break;
        }
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= 1e-6 * x_cost) {
            sum.termination = 0;
            break;
        }
        const double rel = cost_change / model_change;
        if (rel > 1e-3) {
            std::copy(cand.begin(), cand.end(), x);
            x_norm = norm(x, na);
            if (!pb.evaluate(x, x_cost, &r, &J)) {
                sum.termination = 2;
                break;
            }
            rows = (int)r.size();
            gradient(g);
            apply_scale();
            gmax = grad_max(x);
            success = true;
            ++sum.successful_steps;
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = std::max(radius, 3.0 * dogleg_norm);
            mu = std::max(1e-8, 2.0 * mu / 10.0);
            reuse = false;
        } else {
            radius *= 0.5;
            reuse = true;
        }
    }
    sum.iterations = it;
    std::copy(best.begin(), best.end(), x);
    sum.final_cost = min_cost == DBL_MAX ? x_cost : min_cost;
    return sum;
}

} // namespace dense
} // namespace pvio
