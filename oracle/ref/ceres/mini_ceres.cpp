// mini_ceres.cpp -- implementation of oracle/ref/ceres/ceres/ceres.h (TEST INFRASTRUCTURE, see that header).
//
// ceres::Solve below RESTATES ceres-solver 1.14 (third party, pinned at pvio/depends/CMakeLists.txt:31-35, source not in
// /root/reference) as the reference configures it (estimation/ceres/solver_options.h:26-33 + Ceres defaults):
//   Program reduction            constant parameter blocks and residual blocks that touch only constant blocks are removed
//                                (their cost becomes fixed_cost), parameter blocks no remaining residual references are dropped
//   ResidualBlock::Evaluate      cost = rho(|r|^2) / 2; Corrector (for Cauchy: r and J scaled by sqrt(rho')); local
//                                Jacobian = J_global * LocalParameterization::ComputeJacobian
//   TrustRegionMinimizer         IterationZero, ComputeTrustRegionStep, invalid / successful / unsuccessful step handling,
//                                parameter / function / gradient tolerance, max iterations, min radius; Jacobi scaling
//                                1 / (1 + ||column||) computed once; parameters_ = minimum-cost iterate;
//                                StateUpdatingCallback (update_state_every_iteration) before the user's callbacks
//   DoglegStrategy               TRADITIONAL_DOGLEG, mu in [1e-8, 1], x10 on failure, x0.2 on success, reuse after a rejected step
// The linear solver is a dense Cholesky of the full scaled normal equations (SPARSE_SCHUR solves the same system).
// Residual blocks are evaluated on an INTERNAL state vector, like Ceres: user memory only changes through the state-updating
// callback and at the end of Solve -- which is what makes the reference's live-bias read (preintegration_error_cost.h:57-58)
// observable.
#include <ceres/ceres.h>

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

namespace ceres {

bool HomogeneousVectorParameterization::Plus(const double *, const double *, double *) const {
    std::fprintf(stderr, "mini-ceres: HomogeneousVectorParameterization::Plus on a free block is not restated\n");
    std::abort();
}
bool HomogeneousVectorParameterization::ComputeJacobian(const double *, double *) const {
    std::fprintf(stderr, "mini-ceres: HomogeneousVectorParameterization::ComputeJacobian on a free block is not restated\n");
    std::abort();
}

Problem::~Problem() {
    std::set<CostFunction *> costs;
    std::set<LossFunction *> losses;
    std::set<LocalParameterization *> locals;
    for (auto &rb : residuals_) {
        if (options_.cost_function_ownership == TAKE_OWNERSHIP && rb->cost) costs.insert(rb->cost);
        if (options_.loss_function_ownership == TAKE_OWNERSHIP && rb->loss) losses.insert(rb->loss);
    }
    if (options_.local_parameterization_ownership == TAKE_OWNERSHIP)
        for (auto &kv : blocks_)
            if (kv.second.local) locals.insert(kv.second.local);
    for (auto *c : costs) delete c;
    for (auto *l : losses) delete l;
    for (auto *l : locals) delete l;
}

void Problem::AddParameterBlock(double *values, int size) { AddParameterBlock(values, size, nullptr); }
void Problem::AddParameterBlock(double *values, int size, LocalParameterization *local) {
    auto it = blocks_.find(values);
    if (it != blocks_.end()) {
        assert(it->second.size == size);
        if (local) it->second.local = local;
        return;
    }
    ParameterBlock b;
    b.user = values, b.size = size, b.local = local, b.order = (int)blocks_.size();
    blocks_[values] = b;
}
void Problem::SetParameterBlockConstant(double *values) { blocks_.at(values).constant = true; }
void Problem::SetParameterBlockVariable(double *values) { blocks_.at(values).constant = false; }
void Problem::SetParameterization(double *values, LocalParameterization *local) { blocks_.at(values).local = local; }
Problem::ResidualBlockId Problem::AddResidualBlock(CostFunction *cost, LossFunction *loss, const std::vector<double *> &params) {
    const std::vector<int32> &sizes = cost->parameter_block_sizes();
    if (sizes.size() != params.size()) {
        std::fprintf(stderr, "mini-ceres: AddResidualBlock with %zu parameter blocks, cost function expects %zu\n", params.size(), sizes.size());
        std::abort();
    }
    for (size_t i = 0; i < params.size(); ++i) AddParameterBlock(params[i], sizes[i]); // implicit add (bundle_adjustor.cpp:165-179 relies on it)
    std::unique_ptr<ResidualBlock> rb(new ResidualBlock);
    rb->cost = cost, rb->loss = loss, rb->params = params;
    residuals_.emplace_back(std::move(rb));
    return residuals_.back().get();
}

std::string Solver::Summary::BriefReport() const {
    char buf[256];
    std::snprintf(buf, sizeof(buf), "mini-ceres: iterations %d, initial cost %.6e, final cost %.6e, termination %d", (int)iterations.size() - 1, initial_cost,
                  final_cost, (int)termination_type);
    return buf;
}

namespace mini {
static thread_local std::function<void(const IterationSummary &)> g_observer;
static thread_local Solver::Summary g_last;
static thread_local int g_fail_factorizations = 0, g_invalid_steps = 0;
void set_observer(std::function<void(const IterationSummary &)> fn) { g_observer = std::move(fn); }
const Solver::Summary &last_summary() { return g_last; }
void set_fault_injection(int fail_factorizations, int invalid_steps) { g_fail_factorizations = fail_factorizations, g_invalid_steps = invalid_steps; }
} // namespace mini

namespace {

struct FreeBlock {
    double *user;
    int size, local_size, off, loff; // offsets into the state vector / the tangent vector
    LocalParameterization *local;
};
struct ActiveResidual {
    const Problem::ResidualBlock *rb;
    int nres, roff;
    std::vector<int> pidx; // free-block index per parameter, -1 = constant
};
struct BlockJ { // local, corrected Jacobians of one residual block
    std::vector<std::vector<double>> J; // per parameter: nres x local_size row-major (empty for constant blocks)
    std::vector<double> r;
};

struct Program {
    std::vector<FreeBlock> blocks;
    std::vector<ActiveResidual> res;
    int n_state = 0, n_tangent = 0, n_res = 0;
    double fixed_cost = 0;

    void build(Problem *problem) {
        std::vector<const Problem::ParameterBlock *> ordered(problem->parameter_blocks().size());
        for (auto &kv : problem->parameter_blocks()) ordered[kv.second.order] = &kv.second;
        // residual blocks with at least one free parameter stay; the rest is fixed cost
        std::map<double *, int> used;
        std::vector<const Problem::ResidualBlock *> kept;
        for (auto &rbp : problem->residual_blocks()) {
            const Problem::ResidualBlock *rb = rbp.get();
            bool any_free = false;
            for (double *p : rb->params)
                if (!problem->parameter_blocks().at(p).constant) any_free = true;
            if (any_free) {
                kept.push_back(rb);
                for (double *p : rb->params) used[p]++;
            } else {
                std::vector<double> r(rb->cost->num_residuals());
                std::vector<const double *> ps(rb->params.begin(), rb->params.end());
                if (rb->cost->Evaluate(ps.data(), r.data(), nullptr)) {
                    double s = 0;
                    for (double v : r) s += v * v;
                    double rho[3] = {s, 1, 0};
                    if (rb->loss) rb->loss->Evaluate(s, rho);
                    fixed_cost += 0.5 * rho[0];
                }
            }
        }
        std::map<double *, int> index;
        for (const Problem::ParameterBlock *pb : ordered) {
            if (pb->constant || !used.count(pb->user)) continue;
            FreeBlock fb;
            fb.user = pb->user, fb.size = pb->size, fb.local = pb->local;
            fb.local_size = pb->local ? pb->local->LocalSize() : pb->size;
            fb.off = n_state, fb.loff = n_tangent;
            n_state += fb.size, n_tangent += fb.local_size;
            index[pb->user] = (int)blocks.size();
            blocks.push_back(fb);
        }
        for (const Problem::ResidualBlock *rb : kept) {
            ActiveResidual ar;
            ar.rb = rb, ar.nres = rb->cost->num_residuals(), ar.roff = n_res;
            n_res += ar.nres;
            for (double *p : rb->params) ar.pidx.push_back(index.count(p) ? index[p] : -1);
            res.push_back(std::move(ar));
        }
    }

    void read_user(std::vector<double> &x) const {
        x.resize(n_state);
        for (const FreeBlock &b : blocks) std::memcpy(&x[b.off], b.user, sizeof(double) * b.size);
    }
    void write_user(const std::vector<double> &x) const {
        for (const FreeBlock &b : blocks) std::memcpy(b.user, &x[b.off], sizeof(double) * b.size);
    }
    bool plus(const std::vector<double> &x, const std::vector<double> &delta, std::vector<double> &out) const {
        out.resize(n_state);
        for (const FreeBlock &b : blocks) {
            if (b.local) {
                if (!b.local->Plus(&x[b.off], &delta[b.loff], &out[b.off])) return false;
            } else {
                for (int k = 0; k < b.size; ++k) out[b.off + k] = x[b.off + k] + delta[b.loff + k];
            }
        }
        return true;
    }

    // cost (reduced program) and, when `jac` is given, the corrected local Jacobians and residuals of every block
    bool evaluate(const std::vector<double> &x, double *cost, std::vector<BlockJ> *jac) const {
        double total = 0;
        if (jac) jac->resize(res.size());
        std::vector<const double *> ps;
        std::vector<double *> jp;
        std::vector<std::vector<double>> jg;
        std::vector<double> r, lj;
        for (size_t bi = 0; bi < res.size(); ++bi) {
            const ActiveResidual &ar = res[bi];
            const size_t np = ar.pidx.size();
            ps.resize(np), jp.assign(np, nullptr), jg.resize(np);
            for (size_t i = 0; i < np; ++i) {
                ps[i] = ar.pidx[i] >= 0 ? &x[blocks[ar.pidx[i]].off] : ar.rb->params[i];
                if (jac && ar.pidx[i] >= 0) {
                    jg[i].assign((size_t)ar.nres * blocks[ar.pidx[i]].size, 0.0);
                    jp[i] = jg[i].data();
                }
            }
            r.assign(ar.nres, 0.0);
            if (!ar.rb->cost->Evaluate(ps.data(), r.data(), jac ? jp.data() : nullptr)) return false;
            double s = 0;
            for (double v : r) s += v * v;
            double rho[3] = {s, 1.0, 0.0};
            if (ar.rb->loss) ar.rb->loss->Evaluate(s, rho);
            total += 0.5 * rho[0];
            for (double v : r)
                if (!std::isfinite(v)) return false;
            if (!jac) continue;
            double scale = 1.0;
            if (ar.rb->loss) { // Corrector: only the rho'' <= 0 (or s == 0) branch occurs for the losses restated here
                if (!(s == 0.0 || rho[2] <= 0.0)) {
                    std::fprintf(stderr, "mini-ceres: Corrector branch rho'' > 0 is not restated\n");
                    std::abort();
                }
                scale = std::sqrt(rho[1]);
            }
            BlockJ &bj = (*jac)[bi];
            bj.J.resize(np);
            for (size_t i = 0; i < np; ++i) {
                bj.J[i].clear();
                if (ar.pidx[i] < 0) continue;
                const FreeBlock &fb = blocks[ar.pidx[i]];
                std::vector<double> &out = bj.J[i];
                out.assign((size_t)ar.nres * fb.local_size, 0.0);
                if (fb.local) {
                    lj.assign((size_t)fb.size * fb.local_size, 0.0);
                    if (!fb.local->ComputeJacobian(&x[fb.off], lj.data())) return false;
                    for (int a = 0; a < ar.nres; ++a)
                        for (int c = 0; c < fb.local_size; ++c) {
                            double acc = 0;
                            for (int k = 0; k < fb.size; ++k) acc += jg[i][(size_t)a * fb.size + k] * lj[(size_t)k * fb.local_size + c];
                            out[(size_t)a * fb.local_size + c] = acc;
                        }
                } else {
                    out = jg[i];
                }
                for (double &v : out) {
                    if (!std::isfinite(v)) return false;
                    v *= scale;
                }
            }
            bj.r = r;
            for (double &v : bj.r) v *= scale;
        }
        *cost = total;
        return std::isfinite(total);
    }
};

bool cholesky_lower(std::vector<double> &A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            const double *ri = &A[(size_t)i * n], *rj = &A[(size_t)j * n];
            for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    return true;
}
void cholesky_solve(const std::vector<double> &L, int n, std::vector<double> &b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * b[k];
        b[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k];
        b[i] = s / L[(size_t)i * n + i];
    }
}

struct Minimizer {
    const Solver::Options &opt;
    Program &prog;
    Solver::Summary *summary;
    const int n;
    std::vector<double> x, cand, best; // iterate, candidate, parameters_ (minimum-cost iterate)
    std::vector<double> H, g;          // unscaled J^T J (dense, symmetric), J^T r
    std::vector<double> c;             // Jacobi scaling
    std::vector<double> D, gh, gn, step, delta;
    double x_cost = 0, cand_cost = 0, min_cost = DBL_MAX, x_norm = 0;
    double radius, mu = 1e-8, alpha = 0, dogleg_step_norm = 0, model_cost_change = 0, grad_max = 0, grad_norm = 0;
    bool reuse = false;
    int invalid_steps = 0, fail_left = 0, invalid_left = 0;

    Minimizer(const Solver::Options &o, Program &p, Solver::Summary *s) : opt(o), prog(p), summary(s), n(p.n_tangent), radius(o.initial_trust_region_radius) {}

    bool evaluate_gradient_and_jacobian(bool iteration_zero) {
        std::vector<BlockJ> jac;
        if (!prog.evaluate(x, &x_cost, &jac)) return false;
        H.assign((size_t)n * n, 0.0), g.assign(n, 0.0);
        for (size_t bi = 0; bi < prog.res.size(); ++bi) {
            const ActiveResidual &ar = prog.res[bi];
            const BlockJ &bj = jac[bi];
            for (size_t i = 0; i < ar.pidx.size(); ++i) {
                if (ar.pidx[i] < 0) continue;
                const FreeBlock &bi_ = prog.blocks[ar.pidx[i]];
                for (int a = 0; a < bi_.local_size; ++a) {
                    double ga = 0;
                    for (int rr = 0; rr < ar.nres; ++rr) ga += bj.J[i][(size_t)rr * bi_.local_size + a] * bj.r[rr];
                    g[bi_.loff + a] += ga;
                }
                for (size_t j = 0; j < ar.pidx.size(); ++j) {
                    if (ar.pidx[j] < 0) continue;
                    const FreeBlock &bj_ = prog.blocks[ar.pidx[j]];
                    for (int a = 0; a < bi_.local_size; ++a)
                        for (int b = 0; b < bj_.local_size; ++b) {
                            double s = 0;
                            for (int rr = 0; rr < ar.nres; ++rr) s += bj.J[i][(size_t)rr * bi_.local_size + a] * bj.J[j][(size_t)rr * bj_.local_size + b];
                            H[(size_t)(bi_.loff + a) * n + bj_.loff + b] += s;
                        }
                }
            }
        }
        if (iteration_zero) {
            c.assign(n, 1.0);
            if (opt.jacobi_scaling)
                for (int a = 0; a < n; ++a) c[a] = 1.0 / (1.0 + std::sqrt(H[(size_t)a * n + a]));
        }
        // gradient_max_norm = || x - Plus(x, -g) ||_inf, gradient_norm likewise in 2-norm (unscaled gradient)
        std::vector<double> ng(n), xp;
        for (int a = 0; a < n; ++a) ng[a] = -g[a];
        if (!prog.plus(x, ng, xp)) return false;
        grad_max = 0, grad_norm = 0;
        for (int k = 0; k < prog.n_state; ++k) grad_max = std::max(grad_max, std::fabs(x[k] - xp[k])), grad_norm += (x[k] - xp[k]) * (x[k] - xp[k]);
        grad_norm = std::sqrt(grad_norm);
        return true;
    }

    double quad(const std::vector<double> &u, const std::vector<double> &w) const { // u^T (C H C) w
        double s = 0;
        for (int a = 0; a < n; ++a) {
            double row = 0;
            const double *Ha = &H[(size_t)a * n];
            for (int b = 0; b < n; ++b) row += Ha[b] * c[b] * w[b];
            s += u[a] * c[a] * row;
        }
        return s;
    }

    bool solve_gauss_newton() {
        std::vector<double> S((size_t)n * n), rhs(n);
        for (int a = 0; a < n; ++a) {
            for (int b = 0; b < n; ++b) S[(size_t)a * n + b] = c[a] * H[(size_t)a * n + b] * c[b];
            S[(size_t)a * n + a] += mu * D[a] * D[a];
            rhs[a] = c[a] * g[a];
        }
        if (n > 0 && !cholesky_lower(S, n)) return false;
        cholesky_solve(S, n, rhs);
        for (int a = 0; a < n; ++a) {
            if (!std::isfinite(rhs[a])) return false;
            gn[a] = -D[a] * rhs[a];
        }
        return true;
    }

    int compute_step() { // DoglegStrategy::ComputeStep; 1 = linear solver failure
        if (!reuse) {
            reuse = true;
            D.resize(n), gh.resize(n), gn.assign(n, 0.0);
            double g2 = 0;
            std::vector<double> v(n);
            for (int a = 0; a < n; ++a) {
                const double d2 = c[a] * c[a] * H[(size_t)a * n + a];
                D[a] = std::sqrt(std::min(std::max(d2, opt.min_lm_diagonal), opt.max_lm_diagonal));
                gh[a] = c[a] * g[a] / D[a];
                v[a] = gh[a] / D[a];
                g2 += gh[a] * gh[a];
            }
            alpha = g2 / quad(v, v);
            bool solved = false;
            while (mu < 1.0) {
                const bool injected = fail_left > 0;
                if (injected) --fail_left;
                if (solve_gauss_newton() && !injected) {
                    solved = true;
                    break;
                }
                mu *= 10.0;
            }
            if (!solved) return 1;
        }
        double gnorm2 = 0, gnn2 = 0, gdot = 0;
        for (int a = 0; a < n; ++a) gnorm2 += gh[a] * gh[a], gnn2 += gn[a] * gn[a], gdot += gh[a] * gn[a];
        const double gradient_norm = std::sqrt(gnorm2), gauss_newton_norm = std::sqrt(gnn2);
        double ca, cb;
        if (gauss_newton_norm <= radius) {
            ca = 0, cb = 1, dogleg_step_norm = gauss_newton_norm;
        } else if (gradient_norm * alpha >= radius) {
            ca = -(radius / gradient_norm), cb = 0, dogleg_step_norm = radius;
        } else {
            const double b_dot_a = -alpha * gdot;
            const double a_squared_norm = std::pow(alpha * gradient_norm, 2.0);
            const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gauss_newton_norm, 2);
            const double cc = b_dot_a - a_squared_norm;
            const double d = std::sqrt(cc * cc + b_minus_a_squared_norm * (std::pow(radius, 2.0) - a_squared_norm));
            const double beta = (cc <= 0) ? (d - cc) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + cc);
            ca = -alpha * (1.0 - beta), cb = beta;
            double n2 = 0;
            for (int a = 0; a < n; ++a) {
                const double v = ca * gh[a] + cb * gn[a];
                n2 += v * v;
            }
            dogleg_step_norm = std::sqrt(n2);
        }
        step.resize(n);
        for (int a = 0; a < n; ++a) step[a] = (ca * gh[a] + cb * gn[a]) / D[a];
        return 0;
    }

    void record(int it, bool valid, bool success, double cost, double cost_change, double step_norm, double rel) {
        IterationSummary s;
        s.iteration = it, s.step_is_valid = valid, s.step_is_successful = success;
        s.cost = cost + prog.fixed_cost, s.cost_change = cost_change, s.gradient_max_norm = grad_max, s.gradient_norm = grad_norm;
        s.step_norm = step_norm, s.relative_decrease = rel, s.trust_region_radius = radius, s.mu = mu;
        summary->iterations.push_back(s);
    }

    void run() {
        fail_left = mini::g_fail_factorizations, invalid_left = mini::g_invalid_steps;
        prog.read_user(x);
        best = x, cand = x;
        TerminationType termination = NO_CONVERGENCE;
        int iter = 0, num_success = 0;
        bool step_success = true, done = false, it_valid = true;
        auto sqnorm = [&](const std::vector<double> &a, const std::vector<double> *b) {
            double s = 0;
            for (int k = 0; k < prog.n_state; ++k) {
                const double d = b ? a[k] - (*b)[k] : a[k];
                s += d * d;
            }
            return s;
        };
        x_norm = std::sqrt(sqnorm(x, nullptr));
        if (!evaluate_gradient_and_jacobian(true)) termination = FAILURE, done = true;
        const double initial_cost = x_cost;
        double it_cost = x_cost, it_cost_change = 0, it_step_norm = 0, it_rel = 0;
        while (!done) {
            // FinalizeIterationAndCheckIfMinimizerCanContinue
            if (step_success) {
                ++num_success;
                if (x_cost < min_cost) min_cost = x_cost, best = x;
            }
            record(iter, it_valid, step_success, it_cost, it_cost_change, it_step_norm, it_rel);
            if (opt.update_state_every_iteration && step_success) prog.write_user(best); // StateUpdatingCallback
            for (IterationCallback *cb : opt.callbacks) (*cb)(summary->iterations.back());
            if (mini::g_observer) mini::g_observer(summary->iterations.back());
            if (iter >= opt.max_num_iterations) {
                termination = NO_CONVERGENCE;
                break;
            }
            if (step_success && grad_max <= opt.gradient_tolerance) {
                termination = CONVERGENCE;
                break;
            }
            if (radius <= opt.min_trust_region_radius) {
                termination = CONVERGENCE;
                break;
            }
            ++iter;
            step_success = false, it_valid = false;
            it_cost_change = 0, it_step_norm = 0, it_rel = 0, it_cost = x_cost;

            const int rc = compute_step();
            if (rc == 0) {
                double gs = 0;
                for (int a = 0; a < n; ++a) gs += c[a] * g[a] * step[a];
                model_cost_change = -(gs + 0.5 * quad(step, step)); // == -(J step)^T (r + J step / 2)
                it_valid = model_cost_change > 0.0;
                if (invalid_left > 0) --invalid_left, it_valid = false;
            }
            if (!it_valid) { // HandleInvalidStep
                if (++invalid_steps >= opt.max_num_consecutive_invalid_steps) {
                    termination = FAILURE;
                    break;
                }
                mu *= 10.0;
                reuse = false;
                continue;
            }
            invalid_steps = 0;
            delta.resize(n);
            for (int a = 0; a < n; ++a) delta[a] = step[a] * c[a];
            if (!prog.plus(x, delta, cand) || !prog.evaluate(cand, &cand_cost, nullptr)) cand_cost = DBL_MAX;
            it_step_norm = std::sqrt(sqnorm(x, &cand));
            if (it_step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
                termination = CONVERGENCE;
                break;
            }
            it_cost_change = x_cost - cand_cost;
            if (std::fabs(it_cost_change) <= opt.function_tolerance * x_cost) {
                termination = CONVERGENCE;
                break;
            }
            it_rel = it_cost_change / model_cost_change;
            if (it_rel > opt.min_relative_decrease) { // HandleSuccessfulStep
                x = cand;
                x_norm = std::sqrt(sqnorm(x, nullptr));
                if (!evaluate_gradient_and_jacobian(false)) {
                    termination = FAILURE;
                    break;
                }
                step_success = true;
                it_cost = x_cost;
                if (it_rel < 0.25) radius *= 0.5;
                if (it_rel > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
                mu = std::max(1e-8, 2.0 * mu / 10.0);
                reuse = false;
            } else { // HandleUnsuccessfulStep
                radius *= 0.5;
                reuse = true;
                it_cost = cand_cost;
            }
        }
        if (termination != FAILURE || num_success > 0) prog.write_user(best);
        summary->termination_type = termination;
        summary->initial_cost = initial_cost + prog.fixed_cost;
        summary->final_cost = (min_cost == DBL_MAX ? initial_cost : min_cost) + prog.fixed_cost;
        summary->fixed_cost = prog.fixed_cost;
        summary->num_successful_steps = num_success;
        summary->num_unsuccessful_steps = iter + 1 - num_success;
        summary->iterations_started = iter;
    }
};

} // namespace

void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary) {
    const auto t0 = std::chrono::steady_clock::now();
    *summary = Solver::Summary();
    if (options.minimizer_type != TRUST_REGION || options.trust_region_strategy_type != DOGLEG || options.use_nonmonotonic_steps) {
        std::fprintf(stderr, "mini-ceres: only TRUST_REGION + DOGLEG (monotonic) is restated\n");
        std::abort();
    }
    Program prog;
    prog.build(problem);
    summary->num_parameter_blocks_reduced = (int)prog.blocks.size();
    summary->num_parameters_reduced = prog.n_state;
    summary->num_effective_parameters_reduced = prog.n_tangent;
    summary->num_residual_blocks_reduced = (int)prog.res.size();
    summary->num_residuals_reduced = prog.n_res;
    if (prog.blocks.empty()) { // nothing to optimize: Ceres reports CONVERGENCE with the fixed cost
        summary->termination_type = CONVERGENCE;
        summary->initial_cost = summary->final_cost = summary->fixed_cost = prog.fixed_cost;
    } else {
        Minimizer m(options, prog, summary);
        m.run();
    }
    summary->total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    mini::g_last = *summary;
}

} // namespace ceres
