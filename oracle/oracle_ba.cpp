// oracle_ba.cpp -- CPU oracle for the bundle-adjustment hot path.
//
// TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
// the checker / reported baseline.  The product (libpvio_hip.so) never links or calls it.
//
// Restates:
//   BundleAdjustorSolver::solve        pvio/src/pvio/estimation/bundle_adjustor.cpp:63-299
//   BundleAdjustor::marginalize_frame  pvio/src/pvio/estimation/bundle_adjustor.cpp:348-599
//   compute_reprojection_error         pvio/src/pvio/estimation/bundle_adjustor.cpp:321-336
//   ceres::Solve with set_solver_options (solver_options.h:26-33) -- third-party ceres-solver, pinned
//     1.14.0 only at pvio/depends/CMakeLists.txt:31-35.  Ceres source is NOT in /root/reference, so the
//     trust-region/Dogleg loop below restates the published Ceres-1.14 algorithm (TrustRegionMinimizer,
//     DoglegStrategy(TRADITIONAL_DOGLEG), Corrector, Jacobi scaling, SchurComplementSolver) -- SURVEY.md App. B.
//
// PARITY UNPINNED: the reference has no tests or golden vectors for this path and cannot be built here.
// The oracle is pinned only by its own invariants (finite-difference Jacobians, Schur-vs-dense equality
// against an independent numpy implementation in tests/np_reference.py, zero-residual-at-truth, ...).
#include "../include/pvio_hip.h"
#include "oracle_factors.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace orc;

namespace {

struct Layout {
    int N = 0, M = 0, F = 0, P = 0;
    std::vector<int> pose_off, motion_off; // tangent column offset per frame or -1
    std::vector<char> lm_used;
    std::vector<char> pre_active;          // per frame j: preintegration factor (j-1 -> j) present
    std::vector<char> plane_active;
    bool prior_active = false;
    std::vector<int> obs_lm;               // landmark of each observation
};

Layout make_layout(const pvio_ba_problem &pb) {
    Layout L;
    L.N = pb.n_frames;
    L.M = pb.n_landmarks;
    L.F = pb.n_obs;
    L.pose_off.assign(L.N, -1);
    L.motion_off.assign(L.N, -1);
    L.lm_used.assign(L.M, 0);
    L.pre_active.assign(L.N, 0);
    L.plane_active.assign(pb.n_plane_factors, 0);
    L.obs_lm.assign(L.F, -1);
    std::vector<char> pose_used(L.N, 0), motion_used(L.N, 0);
    for (int l = 0; l < L.M; ++l) {
        int b = pb.lm_obs_ptr[l], e = pb.lm_obs_ptr[l + 1];
        if (e > b) {
            L.lm_used[l] = 1;
            pose_used[pb.lm_anchor_frame[l]] = 1;
        }
        for (int o = b; o < e; ++o) {
            L.obs_lm[o] = l;
            pose_used[pb.obs_frame[o]] = 1;
        }
    }
    if (pb.use_inertial)
        for (int j = 1; j < L.N; ++j)
            if (pb.preint_valid && pb.preint_valid[j]) {
                L.pre_active[j] = 1;
                pose_used[j - 1] = pose_used[j] = 1;
                motion_used[j - 1] = motion_used[j] = 1;
            }
    if (pb.prior_n > 0) {
        L.prior_active = true;
        for (int i = 0; i < pb.prior_n; ++i) pose_used[pb.prior_frames[i]] = motion_used[pb.prior_frames[i]] = 1;
    }
    for (int i = 0; i < pb.n_rot_priors; ++i) pose_used[pb.rot_prior_frame[i]] = 1;
    for (int f = 0; f < pb.n_plane_factors; ++f) {
        bool any_free = false;
        for (int o = pb.plane_obs_ptr[f]; o < pb.plane_obs_ptr[f + 1]; ++o)
            if (!pb.frame_fixed[pb.plane_obs_frame[o]]) any_free = true;
        L.plane_active[f] = any_free; // all-constant residual blocks are folded into fixed_cost by Ceres
        if (any_free)
            for (int o = pb.plane_obs_ptr[f]; o < pb.plane_obs_ptr[f + 1]; ++o) pose_used[pb.plane_obs_frame[o]] = 1;
    }
    int off = 0;
    for (int i = 0; i < L.N; ++i) {
        if (pose_used[i] && !pb.frame_fixed[i]) {
            L.pose_off[i] = off;
            off += 6;
        }
        if (motion_used[i]) {
            L.motion_off[i] = off;
            off += 9;
        }
    }
    L.P = off;
    return L;
}

struct Lin { // normal-equation pieces of one linearization (unscaled, robustified)
    std::vector<double> Hpp, gp; // P x P, P
    std::vector<double> Hll, bl; // M
    std::vector<double> Wt;      // F x 6  (rho row x target pose cols)
    std::vector<double> Wa;      // M x 6  (rho row x anchor pose cols)
};

// Second summation order (tests only): every sum over landmarks -- the pose blocks of J^T J and J^T r, the Schur
// complement, the dogleg scalars -- runs over the landmarks last to first (and over a landmark's observations last to first) instead of in
// the reference's order (bundle_adjustor.cpp:142-161: tracks in map order, observations in frame order).  Same algorithm, same factors,
// a different rounding of every sum: what two runs of it differ by is what ANY reordering of the sums (the kernels' included) is entitled to.
// Orders 2 and 3 are two more samples of the same thing: even landmarks first, then the odd ones; and that walk backwards.
static int g_sum_order = 0;
static inline int lm_at(int k, int M) {
    if (g_sum_order == 0) return k;
    if (g_sum_order == 1) return M - 1 - k;
    const int h = (M + 1) / 2, kk = g_sum_order == 3 ? M - 1 - k : k; // 2, 3: evens, then odds
    return kk < h ? 2 * kk : 2 * (kk - h) + 1;
}

struct Evaluator {
    const pvio_ba_problem &pb;
    const Layout &L;
    std::vector<Ext> cam, imu;
    Evaluator(const pvio_ba_problem &pb_, const Layout &L_) : pb(pb_), L(L_) {
        for (int i = 0; i < L.N; ++i) {
            cam.push_back(ext_load(pb.cam_extrinsic + 7 * i));
            imu.push_back(ext_load(pb.imu_extrinsic + 7 * i));
        }
    }

    static void add_block(std::vector<double> &H, int P, int ro, int co, const double *Ja, int lda, const double *Jb, int ldb,
                          int rows, int na, int nb) { // H[ro:ro+na, co:co+nb] += Ja^T Jb
        for (int a = 0; a < na; ++a)
            for (int b = 0; b < nb; ++b) {
                double s = 0;
                for (int r = 0; r < rows; ++r) s += Ja[r * lda + a] * Jb[r * ldb + b];
                H[(size_t)(ro + a) * P + co + b] += s;
            }
    }

    // Evaluates cost (and the linearization when lin != nullptr) at (fs, rho); `user` supplies the live
    // frame_i->motion.bg/ba read by the pre-integration functor.  Returns false on non-finite output.
    bool eval(const double *fs, const double *rho, const double *user, double *cost_out, Lin *lin) const {
        const int P = L.P;
        double cost = 0;
        bool ok = true;
        if (lin) {
            lin->Hpp.assign((size_t)P * P, 0.0);
            lin->gp.assign(P, 0.0);
            lin->Hll.assign(L.M, 0.0);
            lin->bl.assign(L.M, 0.0);
            lin->Wt.assign((size_t)L.F * 6, 0.0);
            lin->Wa.assign((size_t)L.M * 6, 0.0);
        }
        // marginalization prior: bundle_adjustor.cpp:126-139 (no loss)
        if (L.prior_active) {
            int n = pb.prior_n, D = 15 * n;
            std::vector<const double *> st(n);
            for (int i = 0; i < n; ++i) st[i] = fs + 16 * pb.prior_frames[i];
            std::vector<double> r(D), J(lin ? (size_t)D * D : 0);
            eval_prior(n, st.data(), pb.prior_lin_state, pb.prior_S, pb.prior_s, r.data(), lin ? J.data() : nullptr);
            double sq = 0;
            for (int k = 0; k < D; ++k) sq += r[k] * r[k];
            if (!std::isfinite(sq)) ok = false;
            cost += 0.5 * sq;
            if (lin) {
                std::vector<int> col(D, -1); // prior column -> tangent column
                for (int i = 0; i < n; ++i) {
                    int f = pb.prior_frames[i];
                    for (int k = 0; k < 6; ++k) col[15 * i + k] = L.pose_off[f] >= 0 ? L.pose_off[f] + k : -1;
                    for (int k = 0; k < 9; ++k) col[15 * i + 6 + k] = L.motion_off[f] >= 0 ? L.motion_off[f] + k : -1;
                }
                for (int a = 0; a < D; ++a) {
                    if (col[a] < 0) continue;
                    double g = 0;
                    for (int row = 0; row < D; ++row) g += J[(size_t)row * D + a] * r[row];
                    lin->gp[col[a]] += g;
                    for (int b = 0; b < D; ++b) {
                        if (col[b] < 0) continue;
                        double s = 0;
                        for (int row = 0; row < D; ++row) s += J[(size_t)row * D + a] * J[(size_t)row * D + b];
                        lin->Hpp[(size_t)col[a] * P + col[b]] += s;
                    }
                }
            }
        }
        // rotation priors (no reference counterpart, see eval_rot_prior): no loss; a prior on a constant pose block is
        // an all-constant residual block and is left out like Ceres does
        for (int i = 0; i < pb.n_rot_priors; ++i) {
            const int f = pb.rot_prior_frame[i], po = L.pose_off[f];
            if (po < 0) continue;
            double r[3], J[9];
            eval_rot_prior(fs + 16 * f, pb.rot_prior_q0 + 4 * i, pb.rot_prior_sqrt_info + 9 * i, r, lin ? J : nullptr);
            const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
            if (!std::isfinite(sq)) ok = false;
            cost += 0.5 * sq;
            if (lin) {
                add_block(lin->Hpp, P, po, po, J, 3, J, 3, 3, 3, 3);
                for (int k = 0; k < 3; ++k) lin->gp[po + k] += J[k] * r[0] + J[3 + k] * r[1] + J[6 + k] * r[2];
            }
        }
        // reprojection factors: bundle_adjustor.cpp:142-161, CauchyLoss(1.0)
        for (int lk = 0; lk < L.M; ++lk) {
            const int l = lm_at(lk, L.M);
            int a = pb.lm_anchor_frame[l];
            for (int ok_ = pb.lm_obs_ptr[l]; ok_ < pb.lm_obs_ptr[l + 1]; ++ok_) {
                const int o = (g_sum_order & 1) ? pb.lm_obs_ptr[l] + pb.lm_obs_ptr[l + 1] - 1 - ok_ : ok_;
                int t = pb.obs_frame[o];
                double r[2], J[26];
                eval_reprojection(fs + 16 * t, fs + 16 * a, rho[l], pb.lm_anchor_z + 2 * l, pb.obs_z + 2 * o, cam[a], cam[t],
                                  pb.sqrt_inv_cov + 4 * t, r, lin ? J : nullptr);
                double s = r[0] * r[0] + r[1] * r[1];
                if (!std::isfinite(s)) ok = false;
                // duplicate residual blocks (bundle_adjustor.cpp:165-179): Ceres sums m identical blocks, each robustified on its own
                const double mult = pb.lm_multiplicity ? (double)pb.lm_multiplicity[l] : 1.0;
                cost += mult * (0.5 * std::log(1.0 + s)); // rho[0] = b log(1 + s c), a = 1
                if (lin) {
                    for (int k = 0; k < 26; ++k)
                        if (!std::isfinite(J[k])) ok = false;
                    double sw = std::sqrt(std::max(DBL_MIN, 1.0 / (1.0 + s))); // Corrector: rho'' < 0 -> sqrt(rho')
                    if (mult != 1.0) sw *= std::sqrt(mult);                       // m (J^T J), m (J^T r)
                    for (int k = 0; k < 26; ++k) J[k] *= sw;
                    r[0] *= sw;
                    r[1] *= sw;
                    int to = L.pose_off[t], ao = L.pose_off[a];
                    const double *Jt = J, *Jr = J + 6, *Jd = J + 12;
                    if (to >= 0) {
                        add_block(lin->Hpp, P, to, to, Jt, 13, Jt, 13, 2, 6, 6);
                        for (int k = 0; k < 6; ++k) lin->gp[to + k] += Jt[k] * r[0] + Jt[13 + k] * r[1];
                        for (int k = 0; k < 6; ++k) lin->Wt[(size_t)o * 6 + k] = Jd[0] * Jt[k] + Jd[13] * Jt[13 + k];
                    }
                    if (ao >= 0) {
                        add_block(lin->Hpp, P, ao, ao, Jr, 13, Jr, 13, 2, 6, 6);
                        for (int k = 0; k < 6; ++k) lin->gp[ao + k] += Jr[k] * r[0] + Jr[13 + k] * r[1];
                        for (int k = 0; k < 6; ++k) lin->Wa[(size_t)l * 6 + k] += Jd[0] * Jr[k] + Jd[13] * Jr[13 + k];
                    }
                    if (to >= 0 && ao >= 0) {
                        add_block(lin->Hpp, P, to, ao, Jt, 13, Jr, 13, 2, 6, 6);
                        add_block(lin->Hpp, P, ao, to, Jr, 13, Jt, 13, 2, 6, 6);
                    }
                    lin->Hll[l] += Jd[0] * Jd[0] + Jd[13] * Jd[13];
                    lin->bl[l] += Jd[0] * r[0] + Jd[13] * r[1];
                }
            }
        }
        // plane-distance factors: bundle_adjustor.cpp:180-195, CauchyLoss(1.0)
        for (int f = 0; f < pb.n_plane_factors; ++f) {
            if (!L.plane_active[f]) continue;
            int b = pb.plane_obs_ptr[f], K = pb.plane_obs_ptr[f + 1] - b;
            std::vector<const double *> st(K);
            std::vector<Ext> cs(K);
            for (int k = 0; k < K; ++k) {
                st[k] = fs + 16 * pb.plane_obs_frame[b + k];
                cs[k] = cam[pb.plane_obs_frame[b + k]];
            }
            double r;
            std::vector<double> J(lin ? 6 * K : 0);
            eval_plane(K, st.data(), cs.data(), pb.plane_obs_z + 2 * b, pb.plane_normal + 3 * f, pb.plane_distance[f],
                       pb.plane_sqrt_inv_cov, 1.0, &r, lin ? J.data() : nullptr);
            double s = r * r;
            if (!std::isfinite(s)) ok = false;
            cost += 0.5 * std::log(1.0 + s);
            if (lin) {
                double sw = std::sqrt(std::max(DBL_MIN, 1.0 / (1.0 + s)));
                r *= sw;
                for (auto &v : J) v *= sw;
                for (int k1 = 0; k1 < K; ++k1) {
                    int o1 = L.pose_off[pb.plane_obs_frame[b + k1]];
                    if (o1 < 0) continue;
                    for (int c = 0; c < 6; ++c) lin->gp[o1 + c] += J[6 * k1 + c] * r;
                    for (int k2 = 0; k2 < K; ++k2) {
                        int o2 = L.pose_off[pb.plane_obs_frame[b + k2]];
                        if (o2 < 0) continue;
                        for (int c1 = 0; c1 < 6; ++c1)
                            for (int c2 = 0; c2 < 6; ++c2) lin->Hpp[(size_t)(o1 + c1) * P + o2 + c2] += J[6 * k1 + c1] * J[6 * k2 + c2];
                    }
                }
            }
        }
        // IMU pre-integration factors: bundle_adjustor.cpp:220-242 (no loss)
        for (int j = 1; j < L.N; ++j) {
            if (!L.pre_active[j]) continue;
            int i = j - 1;
            PreIntFactor pre;
            const double *d = pb.preint_delta + 11 * j;
            pre.dt = d[0];
            pre.dq = qload(d + 1);
            pre.dp = vload(d + 5);
            pre.dv = vload(d + 8);
            pre.U = pb.preint_sqrt_inv_cov + 225 * j;
            const double *pj = pb.preint_jacobian + 45 * j;
            pre.dq_dbg = m3load(pj);
            pre.dp_dbg = m3load(pj + 9);
            pre.dp_dba = m3load(pj + 18);
            pre.dv_dbg = m3load(pj + 27);
            pre.dv_dba = m3load(pj + 36);
            double r[15], J[450];
            // ORACLE_NO_LIVE_BIAS=1 (experiment only, tests/probe_rejections.py): what the factor would be WITHOUT the reference's read of the live
            // frame_i->motion.bg / ba (preintegration_error_cost.h:57-58) -- the bias correction taken at the evaluation point itself
            static const bool no_live_bias = getenv("ORACLE_NO_LIVE_BIAS") != nullptr;
            const double *bias_src = no_live_bias ? fs : user;
            eval_preintegration(fs + 16 * i, fs + 16 * j, vload(bias_src + 16 * i + 10), vload(bias_src + 16 * i + 13), pre, imu[i], imu[j], r,
                                lin ? J : nullptr);
            double sq = 0;
            for (int k = 0; k < 15; ++k) sq += r[k] * r[k];
            if (!std::isfinite(sq)) ok = false;
            cost += 0.5 * sq;
            if (lin) {
                int col[30];
                for (int k = 0; k < 6; ++k) {
                    col[k] = L.pose_off[i] >= 0 ? L.pose_off[i] + k : -1;
                    col[15 + k] = L.pose_off[j] >= 0 ? L.pose_off[j] + k : -1;
                }
                for (int k = 0; k < 9; ++k) {
                    col[6 + k] = L.motion_off[i] + k;
                    col[21 + k] = L.motion_off[j] + k;
                }
                for (int a = 0; a < 30; ++a) {
                    if (col[a] < 0) continue;
                    double g = 0;
                    for (int row = 0; row < 15; ++row) g += J[row * 30 + a] * r[row];
                    lin->gp[col[a]] += g;
                    for (int b2 = 0; b2 < 30; ++b2) {
                        if (col[b2] < 0) continue;
                        double s = 0;
                        for (int row = 0; row < 15; ++row) s += J[row * 30 + a] * J[row * 30 + b2];
                        lin->Hpp[(size_t)col[a] * P + col[b2]] += s;
                    }
                }
            }
        }
        *cost_out = cost;
        return ok;
    }
};

// x (+) delta: QuaternionParameterization::Plus (quaternion_parameterization.h:28-32) for q, addition elsewhere.
void plus(const Layout &L, const double *fs, const double *rho, const double *dp, const double *dl, double *fs_out, double *rho_out) {
    for (int i = 0; i < L.N; ++i) {
        const double *x = fs + 16 * i;
        double *y = fs_out + 16 * i;
        std::memcpy(y, x, 16 * sizeof(double));
        if (L.pose_off[i] >= 0) {
            const double *d = dp + L.pose_off[i];
            Q q = qnormalized(qmul(qload(x), expmap(mk(d[0], d[1], d[2]))));
            qstore(q, y);
            for (int k = 0; k < 3; ++k) y[4 + k] = x[4 + k] + d[3 + k];
        }
        if (L.motion_off[i] >= 0) {
            const double *d = dp + L.motion_off[i];
            for (int k = 0; k < 9; ++k) y[7 + k] = x[7 + k] + d[k];
        }
    }
    for (int l = 0; l < L.M; ++l) rho_out[l] = L.lm_used[l] ? rho[l] + dl[l] : rho[l];
}

// norms over the free ("reduced program") parameters in ambient coordinates
double ambient_sqnorm_diff(const Layout &L, const double *fa, const double *ra, const double *fb, const double *rb, double *maxabs) {
    double s = 0, mx = 0;
    auto acc = [&](double a, double b) {
        double d = a - b;
        s += d * d;
        mx = std::max(mx, std::fabs(d));
    };
    for (int i = 0; i < L.N; ++i) {
        if (L.pose_off[i] >= 0)
            for (int k = 0; k < 7; ++k) acc(fa[16 * i + k], fb ? fb[16 * i + k] : 0.0);
        if (L.motion_off[i] >= 0)
            for (int k = 7; k < 16; ++k) acc(fa[16 * i + k], fb ? fb[16 * i + k] : 0.0);
    }
    for (int l = 0; l < L.M; ++l)
        if (L.lm_used[l]) acc(ra[l], rb ? rb[l] : 0.0);
    if (maxabs) *maxabs = mx;
    return s;
}

// fault injection mirroring pvio_hip_opts::debug_* (tests of the rarely taken paths)
static int g_dbg_fail_factorizations = 0, g_dbg_invalid_steps = 0;
static int g_dbg_fail_left = 0, g_dbg_invalid_left = 0;

struct Solver {
    const pvio_ba_problem &pb;
    Layout L;
    Evaluator ev;
    int P, M;
    // current iterate / candidate / best ("parameters_") / user state
    std::vector<double> fs, rho, cfs, crho, bfs, brho, user;
    Lin lin;
    double x_cost = 0, cand_cost = 0, min_cost = 0, x_norm = 0;
    std::vector<double> cp, cl;                 // Jacobi scaling (computed once)
    std::vector<double> Dp, Dl, ghp, ghl;       // dogleg diagonal, scaled gradient / D
    std::vector<double> gnp, gnl;               // scaled Gauss-Newton step (D * step)
    std::vector<double> stp, stl;               // trust-region step in the scaled space (already / D)
    std::vector<double> dp, dl;                 // delta = step * jacobian_scaling
    double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_step_norm = 0;
    bool reuse = false;
    int invalid_steps = 0;
    double model_cost_change = 0;
    double grad_max_norm = 0;

    Solver(const pvio_ba_problem &pb_) : pb(pb_), L(make_layout(pb_)), ev(pb, L), P(L.P), M(L.M) {}

    // u^T (C H C) w over pose + landmark parts, all in the Jacobi-scaled space
    double quad(const double *up, const double *ul, const double *wp, const double *wl) const {
        double s = 0;
        for (int a = 0; a < P; ++a) {
            double row = 0;
            for (int b = 0; b < P; ++b) row += lin.Hpp[(size_t)a * P + b] * cp[b] * wp[b];
            s += up[a] * cp[a] * row;
        }
        for (int lk = 0; lk < M; ++lk) {
            const int l = lm_at(lk, M);
            if (!L.lm_used[l]) continue;
            double Wu = 0, Ww = 0;
            wdot(l, up, wp, &Wu, &Ww);
            s += cl[l] * (ul[l] * Ww + wl[l] * Wu) + cl[l] * cl[l] * lin.Hll[l] * ul[l] * wl[l];
        }
        return s;
    }
    // (W_l C_p) . u and . w
    void wdot(int l, const double *u, const double *w, double *Wu, double *Ww) const {
        double su = 0, sw = 0;
        int ao = L.pose_off[pb.lm_anchor_frame[l]];
        if (ao >= 0)
            for (int k = 0; k < 6; ++k) {
                double wv = lin.Wa[(size_t)l * 6 + k] * cp[ao + k];
                su += wv * u[ao + k];
                if (w) sw += wv * w[ao + k];
            }
        for (int o = pb.lm_obs_ptr[l]; o < pb.lm_obs_ptr[l + 1]; ++o) {
            int to = L.pose_off[pb.obs_frame[o]];
            if (to < 0) continue;
            for (int k = 0; k < 6; ++k) {
                double wv = lin.Wt[(size_t)o * 6 + k] * cp[to + k];
                su += wv * u[to + k];
                if (w) sw += wv * w[to + k];
            }
        }
        *Wu = su;
        if (Ww) *Ww = sw;
    }

    bool evaluate_gradient_and_jacobian(bool iteration_zero) {
        if (!ev.eval(fs.data(), rho.data(), user.data(), &x_cost, &lin)) return false;
        if (iteration_zero) { // jacobi_scaling: 1 / (1 + sqrt(squared column norm)), computed once
            cp.resize(P);
            cl.assign(M, 1.0);
            for (int a = 0; a < P; ++a) cp[a] = 1.0 / (1.0 + std::sqrt(lin.Hpp[(size_t)a * P + a]));
            for (int l = 0; l < M; ++l)
                if (L.lm_used[l]) cl[l] = 1.0 / (1.0 + std::sqrt(lin.Hll[l]));
        }
        // gradient_max_norm = || x - Plus(x, -g) ||_inf with the UNscaled gradient
        std::vector<double> ng(P), nl(M), tf(fs.size()), tr(M);
        for (int a = 0; a < P; ++a) ng[a] = -lin.gp[a];
        for (int l = 0; l < M; ++l) nl[l] = -lin.bl[l];
        plus(L, fs.data(), rho.data(), ng.data(), nl.data(), tf.data(), tr.data());
        ambient_sqnorm_diff(L, fs.data(), rho.data(), tf.data(), tr.data(), &grad_max_norm);
        return true;
    }

    // DoglegStrategy::ComputeStep.  Returns: 0 ok, 1 linear solver failure (step invalid)
    int compute_step() {
        if (!reuse) {
            reuse = true;
            Dp.resize(P), Dl.assign(M, 1.0), ghp.resize(P), ghl.assign(M, 0.0);
            for (int a = 0; a < P; ++a) {
                double d2 = cp[a] * cp[a] * lin.Hpp[(size_t)a * P + a];
                Dp[a] = std::sqrt(std::min(std::max(d2, 1e-6), 1e32));
                ghp[a] = cp[a] * lin.gp[a] / Dp[a];
            }
            for (int l = 0; l < M; ++l) {
                if (!L.lm_used[l]) continue;
                double d2 = cl[l] * cl[l] * lin.Hll[l];
                Dl[l] = std::sqrt(std::min(std::max(d2, 1e-6), 1e32));
                ghl[l] = cl[l] * lin.bl[l] / Dl[l];
            }
            // Cauchy point: alpha = |g^|^2 / |J (g^/D)|^2
            std::vector<double> vp(P), vl(M, 0.0);
            double g2 = 0;
            for (int a = 0; a < P; ++a) vp[a] = ghp[a] / Dp[a], g2 += ghp[a] * ghp[a];
            for (int lk = 0; lk < M; ++lk) {
                const int l = lm_at(lk, M);
                if (L.lm_used[l]) vl[l] = ghl[l] / Dl[l], g2 += ghl[l] * ghl[l];
            }
            alpha = g2 / quad(vp.data(), vl.data(), vp.data(), vl.data());
            // Gauss-Newton step with mu escalation
            bool solved = false;
            gnp.assign(P, 0.0), gnl.assign(M, 0.0);
            while (mu < 1.0) {
                const bool injected = g_dbg_fail_left > 0;
                if (injected) --g_dbg_fail_left;
                if (solve_gauss_newton() && !injected) {
                    solved = true;
                    break;
                }
                mu *= 10.0;
            }
            if (!solved) return 1;
        }
        // ComputeTraditionalDoglegStep
        double gnorm2 = 0, gnn2 = 0, gdot = 0;
        for (int a = 0; a < P; ++a) gnorm2 += ghp[a] * ghp[a], gnn2 += gnp[a] * gnp[a], gdot += ghp[a] * gnp[a];
        for (int lk = 0; lk < M; ++lk) {
            const int l = lm_at(lk, M);
            if (L.lm_used[l]) gnorm2 += ghl[l] * ghl[l], gnn2 += gnl[l] * gnl[l], gdot += ghl[l] * gnl[l];
        }
        double gradient_norm = std::sqrt(gnorm2), gauss_newton_norm = std::sqrt(gnn2);
        double ca, cb; // step = ca * g^ + cb * gn
        if (gauss_newton_norm <= radius) {
            ca = 0, cb = 1;
            dogleg_step_norm = gauss_newton_norm;
        } else if (gradient_norm * alpha >= radius) {
            ca = -(radius / gradient_norm), cb = 0;
            dogleg_step_norm = radius;
        } else {
            double b_dot_a = -alpha * gdot;
            double a_squared_norm = std::pow(alpha * gradient_norm, 2.0);
            double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gauss_newton_norm, 2);
            double c = b_dot_a - a_squared_norm;
            double d = std::sqrt(c * c + b_minus_a_squared_norm * (std::pow(radius, 2.0) - a_squared_norm));
            double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
            ca = -alpha * (1.0 - beta), cb = beta;
            double n2 = 0;
            for (int a = 0; a < P; ++a) {
                double v = ca * ghp[a] + cb * gnp[a];
                n2 += v * v;
            }
            for (int lk = 0; lk < M; ++lk) {
                const int l = lm_at(lk, M);
                if (L.lm_used[l]) {
                    double v = ca * ghl[l] + cb * gnl[l];
                    n2 += v * v;
                }
            }
            dogleg_step_norm = std::sqrt(n2);
        }
        stp.resize(P), stl.assign(M, 0.0);
        for (int a = 0; a < P; ++a) stp[a] = (ca * ghp[a] + cb * gnp[a]) / Dp[a];
        for (int l = 0; l < M; ++l)
            if (L.lm_used[l]) stl[l] = (ca * ghl[l] + cb * gnl[l]) / Dl[l];
        return 0;
    }

    // SchurComplementSolver on (J^T J + mu D^2) y = J^T r in the scaled space; e-blocks = inverse depths.
    bool solve_gauss_newton() {
        std::vector<double> S((size_t)P * P), rhs(P), A(M, 1.0);
        for (int a = 0; a < P; ++a) {
            for (int b = 0; b < P; ++b) S[(size_t)a * P + b] = cp[a] * lin.Hpp[(size_t)a * P + b] * cp[b];
            S[(size_t)a * P + a] += mu * Dp[a] * Dp[a];
            rhs[a] = cp[a] * lin.gp[a];
        }
        std::vector<int> offs;
        std::vector<const double *> ws;
        for (int lk = 0; lk < M; ++lk) {
            const int l = lm_at(lk, M);
            if (!L.lm_used[l]) continue;
            A[l] = cl[l] * cl[l] * lin.Hll[l] + mu * Dl[l] * Dl[l];
            double wgt = cl[l] * cl[l] / A[l];
            offs.clear(), ws.clear();
            int ao = L.pose_off[pb.lm_anchor_frame[l]];
            if (ao >= 0) offs.push_back(ao), ws.push_back(&lin.Wa[(size_t)l * 6]);
            for (int o = pb.lm_obs_ptr[l]; o < pb.lm_obs_ptr[l + 1]; ++o) {
                int to = L.pose_off[pb.obs_frame[o]];
                if (to >= 0) offs.push_back(to), ws.push_back(&lin.Wt[(size_t)o * 6]);
            }
            for (size_t i = 0; i < offs.size(); ++i) {
                for (int a = 0; a < 6; ++a) {
                    double wa = ws[i][a] * cp[offs[i] + a];
                    rhs[offs[i] + a] -= wgt * wa * lin.bl[l];
                    for (size_t j = 0; j < offs.size(); ++j)
                        for (int b = 0; b < 6; ++b) S[(size_t)(offs[i] + a) * P + offs[j] + b] -= wgt * wa * ws[j][b] * cp[offs[j] + b];
                }
            }
        }
        if (P > 0 && !cholesky_lower(S.data(), P, P)) return false;
        cholesky_solve(S.data(), P, P, rhs.data());
        bool fin = true;
        for (int a = 0; a < P; ++a) {
            if (!std::isfinite(rhs[a])) fin = false;
            gnp[a] = -Dp[a] * rhs[a];
        }
        for (int l = 0; l < M; ++l) {
            if (!L.lm_used[l]) continue;
            double Wy;
            wdot(l, rhs.data(), nullptr, &Wy, nullptr);
            double yl = (cl[l] * lin.bl[l] - cl[l] * Wy) / A[l];
            if (!std::isfinite(yl)) fin = false;
            gnl[l] = -Dl[l] * yl;
        }
        return fin;
    }
};

void record(pvio_ba_summary *sum, const Solver &S, int it, bool valid, bool success, double cost, double cost_change, double step_norm,
            double rel_dec) {
    if (!sum || !sum->trace || sum->trace_len >= sum->trace_capacity) return;
    pvio_ba_iteration &r = sum->trace[sum->trace_len];
    r.iteration = it;
    r.step_is_valid = valid;
    r.step_is_successful = success;
    r.reserved = 0;
    r.cost = cost;
    r.cost_change = cost_change;
    r.gradient_max_norm = S.grad_max_norm;
    r.step_norm = step_norm;
    r.relative_decrease = rel_dec;
    r.trust_region_radius = S.radius;
    r.mu = S.mu;
    if (sum->trace_states) {
        double *dst = sum->trace_states + (size_t)sum->trace_len * (S.L.N * 16 + S.M);
        std::memcpy(dst, S.fs.data(), sizeof(double) * S.L.N * 16);
        std::memcpy(dst + S.L.N * 16, S.rho.data(), sizeof(double) * S.M);
    }
    sum->trace_len++;
}

// bundle_adjustor.cpp:277-296
void quality_pass(const pvio_ba_problem &pb, const double *fs, const double *rho, double *quality, uint8_t *valid) {
    for (int l = 0; l < pb.n_landmarks; ++l) {
        int a = pb.lm_anchor_frame[l];
        Ext ca = ext_load(pb.cam_extrinsic + 7 * a);
        Q qa = qmul(qload(fs + 16 * a), ca.q);
        V3 pa = vload(fs + 16 * a + 4) + qrot(qload(fs + 16 * a), ca.p);
        const double *za = pb.lm_anchor_z + 2 * l;
        V3 x = qrot(qa, mk(za[0] / rho[l], za[1] / rho[l], 1.0 / rho[l])) + pa; // track.cpp:137-141
        double q = 0, qn = 0;
        bool ok = true;
        int nobs = pb.lm_obs_ptr[l + 1] - pb.lm_obs_ptr[l];
        for (int k = -1; k < nobs && ok; ++k) { // keypoint_map(): anchor first, then ascending frame id
            int f = k < 0 ? a : pb.obs_frame[pb.lm_obs_ptr[l] + k];
            const double *z = k < 0 ? za : pb.obs_z + 2 * (pb.lm_obs_ptr[l] + k);
            Ext c = ext_load(pb.cam_extrinsic + 7 * f);
            Q qc = qmul(qload(fs + 16 * f), c.q);
            V3 pc = vload(fs + 16 * f + 4) + qrot(qload(fs + 16 * f), c.p);
            V3 y = qrot(qconj(qc), x - pc);
            if (y[2] <= 1.0e-3 || y[2] > 50) {
                ok = false;
                break;
            }
            const double *Kf = pb.intrinsics + 4 * f;
            double du = (y[0] / y[2]) * Kf[0] + Kf[2] - (z[0] * Kf[0] + Kf[2]);
            double dv = (y[1] / y[2]) * Kf[1] + Kf[3] - (z[1] * Kf[1] + Kf[3]);
            q += std::sqrt(du * du + dv * dv);
            qn += 1.0;
        }
        if (valid) valid[l] = ok ? 1 : 0;
        if (ok && quality) quality[l] = q / std::max(qn, 1.0);
    }
}

} // namespace

extern "C" {

// ceres::Solve(...) as configured at bundle_adjustor.cpp:244-249 + post passes :277-296
void oracle_debug_sum_order(int32_t order) { g_sum_order = order >= 0 && order <= 3 ? order : 0; }
void oracle_debug_fault_injection(int32_t fail_factorizations, int32_t invalid_steps) { g_dbg_fail_factorizations = fail_factorizations, g_dbg_invalid_steps = invalid_steps; }

int32_t oracle_ba_solve(const pvio_ba_problem *pbp, pvio_ba_state *state, pvio_ba_summary *sum) {
    g_dbg_fail_left = g_dbg_fail_factorizations, g_dbg_invalid_left = g_dbg_invalid_steps;
    auto t0 = std::chrono::steady_clock::now();
    const pvio_ba_problem &pb = *pbp;
    Solver S(pb);
    const int N = pb.n_frames, M = pb.n_landmarks;
    S.fs.assign(state->frame_state, state->frame_state + 16 * N);
    S.rho.assign(state->lm_inv_depth, state->lm_inv_depth + M);
    S.user = S.fs;
    S.cfs = S.fs, S.crho = S.rho, S.bfs = S.fs, S.brho = S.rho;
    if (sum) sum->trace_len = 0;
    const int max_iter = pb.max_iterations;
    int termination = PVIO_TERM_NO_CONVERGENCE;
    int iter = 0, num_success = 0;
    double initial_cost = 0;
    bool step_success = true, done = false;

    // IterationZero
    S.x_norm = std::sqrt(ambient_sqnorm_diff(S.L, S.fs.data(), S.rho.data(), nullptr, nullptr, nullptr));
    if (!S.evaluate_gradient_and_jacobian(true)) {
        termination = PVIO_TERM_FAILURE;
        done = true;
    }
    initial_cost = S.x_cost;
    S.min_cost = DBL_MAX;
    double it_cost = S.x_cost, it_cost_change = 0, it_step_norm = 0, it_rel = 0;
    bool it_valid = true;

    while (!done) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (step_success) {
            ++num_success;
            if (S.x_cost < S.min_cost) {
                S.min_cost = S.x_cost;
                S.bfs = S.fs, S.brho = S.rho;
            }
        }
        record(sum, S, iter, it_valid, step_success, it_cost, it_cost_change, it_step_norm, it_rel);
        if (step_success && !getenv("ORACLE_NO_STATE_UPDATE")) S.user = S.bfs; // StateUpdatingCallback (update_state_every_iteration), then PVIO's no-op callback
        if (iter >= max_iter) {
            termination = PVIO_TERM_NO_CONVERGENCE;
            break;
        }
        if (step_success && S.grad_max_norm <= 1e-10) {
            termination = PVIO_TERM_CONVERGENCE;
            break;
        }
        if (S.radius <= 1e-32) {
            termination = PVIO_TERM_CONVERGENCE;
            break;
        }
        ++iter;
        step_success = false;
        it_valid = false;
        it_cost_change = 0, it_step_norm = 0, it_rel = 0;
        it_cost = S.x_cost;

        // ComputeTrustRegionStep
        int rc = S.compute_step();
        if (rc == 0) {
            double gs = 0;
            for (int a = 0; a < S.P; ++a) gs += S.cp[a] * S.lin.gp[a] * S.stp[a];
            for (int l = 0; l < M; ++l)
                if (S.L.lm_used[l]) gs += S.cl[l] * S.lin.bl[l] * S.stl[l];
            double q = S.quad(S.stp.data(), S.stl.data(), S.stp.data(), S.stl.data());
            S.model_cost_change = -(gs + 0.5 * q); // == -(J step)^T (r + J step / 2)
            it_valid = S.model_cost_change > 0.0;
            if (g_dbg_invalid_left > 0) --g_dbg_invalid_left, it_valid = false;
        }
        if (!it_valid) { // HandleInvalidStep
            if (++S.invalid_steps >= 5) {
                termination = PVIO_TERM_FAILURE;
                break;
            }
            S.mu *= 10.0; // DoglegStrategy::StepIsInvalid
            S.reuse = false;
            continue;
        }
        S.invalid_steps = 0;
        S.dp.resize(S.P), S.dl.assign(M, 0.0);
        for (int a = 0; a < S.P; ++a) S.dp[a] = S.stp[a] * S.cp[a];
        for (int l = 0; l < M; ++l) S.dl[l] = S.stl[l] * S.cl[l];
        // ComputeCandidatePointAndEvaluateCost
        plus(S.L, S.fs.data(), S.rho.data(), S.dp.data(), S.dl.data(), S.cfs.data(), S.crho.data());
        if (!S.ev.eval(S.cfs.data(), S.crho.data(), S.user.data(), &S.cand_cost, nullptr)) S.cand_cost = DBL_MAX;
        // ParameterToleranceReached
        it_step_norm = std::sqrt(ambient_sqnorm_diff(S.L, S.fs.data(), S.rho.data(), S.cfs.data(), S.crho.data(), nullptr));
        if (it_step_norm <= 1e-8 * (S.x_norm + 1e-8)) {
            termination = PVIO_TERM_CONVERGENCE;
            break;
        }
        // FunctionToleranceReached
        it_cost_change = S.x_cost - S.cand_cost;
        if (std::fabs(it_cost_change) <= 1e-6 * S.x_cost) {
            termination = PVIO_TERM_CONVERGENCE;
            break;
        }
        it_rel = it_cost_change / S.model_cost_change;
        if (it_rel > 1e-3) { // HandleSuccessfulStep
            S.fs = S.cfs, S.rho = S.crho;
            S.x_norm = std::sqrt(ambient_sqnorm_diff(S.L, S.fs.data(), S.rho.data(), nullptr, nullptr, nullptr));
            if (!S.evaluate_gradient_and_jacobian(false)) {
                termination = PVIO_TERM_FAILURE;
                break;
            }
            step_success = true;
            it_cost = S.x_cost;
            if (it_rel < 0.25) S.radius *= 0.5; // DoglegStrategy::StepAccepted
            if (it_rel > 0.75) S.radius = std::max(S.radius, 3.0 * S.dogleg_step_norm);
            S.mu = std::max(1e-8, 2.0 * S.mu / 10.0);
            S.reuse = false;
        } else { // HandleUnsuccessfulStep
            S.radius *= 0.5; // DoglegStrategy::StepRejected
            S.reuse = true;
            it_cost = S.cand_cost;
        }
    }
    // Ceres copies `parameters_` (the minimum-cost iterate) back to the user state on exit.
    if (termination != PVIO_TERM_FAILURE || num_success > 0) {
        std::memcpy(state->frame_state, S.bfs.data(), sizeof(double) * 16 * N);
        std::memcpy(state->lm_inv_depth, S.brho.data(), sizeof(double) * M);
    }
    quality_pass(pb, state->frame_state, state->lm_inv_depth, state->lm_quality, state->lm_valid);
    if (sum) {
        sum->termination = termination;
        sum->is_usable = termination != PVIO_TERM_FAILURE;
        sum->num_iterations = iter;
        sum->num_successful_steps = num_success;
        sum->initial_cost = initial_cost;
        sum->final_cost = S.min_cost;
        sum->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        sum->device_seconds = 0;
    }
    return PVIO_OK;
}

// Cost + dense normal equations at a state (tests: Schur/dense cross-check against numpy).
// Hpp: PxP, gp: P, Hll/bl: M, W: M x P dense (rho row x pose columns). Returns P.
int32_t oracle_ba_linearize(const pvio_ba_problem *pb, const double *fs, const double *rho, double *cost, double *Hpp, double *gp,
                            double *Hll, double *bl, double *W, int32_t *pose_off, int32_t *motion_off) {
    Layout L = make_layout(*pb);
    Evaluator ev(*pb, L);
    Lin lin;
    double c;
    ev.eval(fs, rho, fs, &c, &lin);
    if (cost) *cost = c;
    int P = L.P;
    if (Hpp) std::memcpy(Hpp, lin.Hpp.data(), sizeof(double) * P * P);
    if (gp) std::memcpy(gp, lin.gp.data(), sizeof(double) * P);
    if (Hll) std::memcpy(Hll, lin.Hll.data(), sizeof(double) * L.M);
    if (bl) std::memcpy(bl, lin.bl.data(), sizeof(double) * L.M);
    if (W) {
        std::memset(W, 0, sizeof(double) * (size_t)L.M * P);
        for (int l = 0; l < L.M; ++l) {
            int ao = L.pose_off[pb->lm_anchor_frame[l]];
            if (ao >= 0)
                for (int k = 0; k < 6; ++k) W[(size_t)l * P + ao + k] += lin.Wa[(size_t)l * 6 + k];
            for (int o = pb->lm_obs_ptr[l]; o < pb->lm_obs_ptr[l + 1]; ++o) {
                int to = L.pose_off[pb->obs_frame[o]];
                if (to >= 0)
                    for (int k = 0; k < 6; ++k) W[(size_t)l * P + to + k] += lin.Wt[(size_t)o * 6 + k];
            }
        }
    }
    if (pose_off) std::copy(L.pose_off.begin(), L.pose_off.end(), pose_off);
    if (motion_off) std::copy(L.motion_off.begin(), L.motion_off.end(), motion_off);
    return P;
}

int32_t oracle_ba_cost(const pvio_ba_problem *pb, const double *fs, const double *rho, const double *user, double *cost) {
    Layout L = make_layout(*pb);
    Evaluator ev(*pb, L);
    return ev.eval(fs, rho, user ? user : fs, cost, nullptr) ? 0 : 1;
}

// bundle_adjustor.cpp:321-336
int32_t oracle_ba_reprojection_error(const pvio_ba_problem *pb, const pvio_ba_state *st, double *out) {
    std::vector<double> q(pb->n_landmarks, 0.0);
    double sum = 0, num = 0;
    for (int l = 0; l < pb->n_landmarks; ++l) {
        int a = pb->lm_anchor_frame[l];
        const double *fs = st->frame_state;
        Ext ca = ext_load(pb->cam_extrinsic + 7 * a);
        Q qa = qmul(qload(fs + 16 * a), ca.q);
        V3 pa = vload(fs + 16 * a + 4) + qrot(qload(fs + 16 * a), ca.p);
        const double *za = pb->lm_anchor_z + 2 * l;
        double rho = st->lm_inv_depth[l];
        V3 x = qrot(qa, mk(za[0] / rho, za[1] / rho, 1.0 / rho)) + pa;
        int nobs = pb->lm_obs_ptr[l + 1] - pb->lm_obs_ptr[l];
        for (int k = -1; k < nobs; ++k) {
            int f = k < 0 ? a : pb->obs_frame[pb->lm_obs_ptr[l] + k];
            const double *z = k < 0 ? za : pb->obs_z + 2 * (pb->lm_obs_ptr[l] + k);
            Ext c = ext_load(pb->cam_extrinsic + 7 * f);
            Q qc = qmul(qload(fs + 16 * f), c.q);
            V3 pc = vload(fs + 16 * f + 4) + qrot(qload(fs + 16 * f), c.p);
            V3 y = qrot(qconj(qc), x - pc);
            const double *Kf = pb->intrinsics + 4 * f;
            double du = (y[0] / y[2]) * Kf[0] - z[0] * Kf[0], dv = (y[1] / y[2]) * Kf[1] - z[1] * Kf[1];
            sum += std::sqrt(du * du + dv * dv);
            num += 1.0;
        }
    }
    *out = sum / std::max(num, 1.0);
    return 0;
}

// BundleAdjustor::marginalize_frame -- bundle_adjustor.cpp:348-599
int32_t oracle_ba_marginalize(const pvio_ba_problem *pbp, const pvio_ba_state *st, int32_t victim, pvio_ba_prior *out) {
    const pvio_ba_problem &pb = *pbp;
    const int N = pb.n_frames, D = 15 * N;
    const double *fs = st->frame_state;
    std::vector<double> H((size_t)D * D, 0.0), b(D, 0.0);
    std::vector<Ext> cam, imu;
    for (int i = 0; i < N; ++i) cam.push_back(ext_load(pb.cam_extrinsic + 7 * i)), imu.push_back(ext_load(pb.imu_extrinsic + 7 * i));
    // (a) old prior (:369-413)
    if (pb.prior_n > 0) {
        int n = pb.prior_n, Dn = 15 * n;
        std::vector<const double *> sp(n);
        for (int i = 0; i < n; ++i) sp[i] = fs + 16 * pb.prior_frames[i];
        std::vector<double> r(Dn), J((size_t)Dn * Dn);
        eval_prior(n, sp.data(), pb.prior_lin_state, pb.prior_S, pb.prior_s, r.data(), J.data());
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 15; ++a) {
                int ga = 15 * pb.prior_frames[i] + a;
                double g = 0;
                for (int row = 0; row < Dn; ++row) g += J[(size_t)row * Dn + 15 * i + a] * r[row];
                b[ga] += g;
                for (int j = 0; j < n; ++j)
                    for (int c = 0; c < 15; ++c) {
                        double s = 0;
                        for (int row = 0; row < Dn; ++row) s += J[(size_t)row * Dn + 15 * i + a] * J[(size_t)row * Dn + 15 * j + c];
                        H[(size_t)ga * D + 15 * pb.prior_frames[j] + c] += s;
                    }
            }
    }
    // (a') the victim's rotation prior (no reference counterpart, see eval_rot_prior): a factor on the victim goes into the
    // new prior like every other factor on it
    for (int i = 0; i < pb.n_rot_priors; ++i) {
        if (pb.rot_prior_frame[i] != victim) continue;
        double r[3], J[9];
        eval_rot_prior(fs + 16 * victim, pb.rot_prior_q0 + 4 * i, pb.rot_prior_sqrt_info + 9 * i, r, J);
        for (int a = 0; a < 3; ++a) {
            b[15 * victim + a] += J[a] * r[0] + J[3 + a] * r[1] + J[6 + a] * r[2];
            for (int c = 0; c < 3; ++c) H[(size_t)(15 * victim + a) * D + 15 * victim + c] += J[a] * J[c] + J[3 + a] * J[3 + c] + J[6 + a] * J[6 + c];
        }
    }
    // (b) pre-integration factors touching the victim (:416-450); live biases == parameters -> dbg = 0
    for (int j = victim; j <= victim + 1; ++j) {
        if (j == 0 || j >= N) continue;
        if (!pb.preint_valid || !pb.preint_valid[j]) continue;
        int i = j - 1;
        PreIntFactor pre;
        const double *d = pb.preint_delta + 11 * j;
        pre.dt = d[0], pre.dq = qload(d + 1), pre.dp = vload(d + 5), pre.dv = vload(d + 8);
        pre.U = pb.preint_sqrt_inv_cov + 225 * j;
        const double *pj = pb.preint_jacobian + 45 * j;
        pre.dq_dbg = m3load(pj), pre.dp_dbg = m3load(pj + 9), pre.dp_dba = m3load(pj + 18), pre.dv_dbg = m3load(pj + 27), pre.dv_dba = m3load(pj + 36);
        double r[15], J[450];
        eval_preintegration(fs + 16 * i, fs + 16 * j, vload(fs + 16 * i + 10), vload(fs + 16 * i + 13), pre, imu[i], imu[j], r, J);
        for (int a = 0; a < 30; ++a) {
            double g = 0;
            for (int row = 0; row < 15; ++row) g += J[row * 30 + a] * r[row];
            b[15 * i + a] += g;
            for (int c = 0; c < 30; ++c) {
                double s = 0;
                for (int row = 0; row < 15; ++row) s += J[row * 30 + a] * J[row * 30 + c];
                H[(size_t)(15 * i + a) * D + 15 * i + c] += s;
            }
        }
    }
    // (c) reprojection factors of the landmarks the victim observes (:453-533), no robust loss
    struct LInfo {
        double mat = 0, vec = 0;
        std::vector<int> frames;
        std::vector<std::array<double, 6>> h;
    };
    std::vector<LInfo> infos;
    for (int l = 0; l < pb.n_landmarks; ++l) {
        int a = pb.lm_anchor_frame[l];
        bool sees = (a == victim);
        for (int o = pb.lm_obs_ptr[l]; o < pb.lm_obs_ptr[l + 1]; ++o)
            if (pb.obs_frame[o] == victim) sees = true;
        if (!sees || pb.lm_obs_ptr[l + 1] == pb.lm_obs_ptr[l]) continue;
        LInfo info;
        auto hrow = [&](int f) -> std::array<double, 6> & {
            for (size_t k = 0; k < info.frames.size(); ++k)
                if (info.frames[k] == f) return info.h[k];
            info.frames.push_back(f);
            info.h.push_back(std::array<double, 6>{{0, 0, 0, 0, 0, 0}});
            return info.h.back();
        };
        for (int o = pb.lm_obs_ptr[l]; o < pb.lm_obs_ptr[l + 1]; ++o) {
            int t = pb.obs_frame[o];
            double r[2], J[26];
            eval_reprojection(fs + 16 * t, fs + 16 * a, st->lm_inv_depth[l], pb.lm_anchor_z + 2 * l, pb.obs_z + 2 * o, cam[a], cam[t],
                              pb.sqrt_inv_cov + 4 * t, r, J);
            const double *Jt = J, *Jr = J + 6, *Jd = J + 12;
            auto blk = [&](int fa, const double *Ja, int fb, const double *Jb) {
                for (int x = 0; x < 6; ++x)
                    for (int y = 0; y < 6; ++y) H[(size_t)(15 * fa + x) * D + 15 * fb + y] += Ja[x] * Jb[y] + Ja[13 + x] * Jb[13 + y];
            };
            blk(t, Jt, t, Jt), blk(a, Jr, t, Jt), blk(t, Jt, a, Jr), blk(a, Jr, a, Jr);
            for (int x = 0; x < 6; ++x) {
                b[15 * t + x] += Jt[x] * r[0] + Jt[13 + x] * r[1];
                b[15 * a + x] += Jr[x] * r[0] + Jr[13 + x] * r[1];
            }
            info.mat += Jd[0] * Jd[0] + Jd[13] * Jd[13];
            info.vec += Jd[0] * r[0] + Jd[13] * r[1];
            auto &ht = hrow(t);
            for (int x = 0; x < 6; ++x) ht[x] += Jd[0] * Jt[x] + Jd[13] * Jt[13 + x];
            auto &ha = hrow(a);
            for (int x = 0; x < 6; ++x) ha[x] += Jd[0] * Jr[x] + Jd[13] * Jr[13 + x];
        }
        infos.push_back(std::move(info));
    }
    // marginalize landmarks (:536-545)
    for (const LInfo &info : infos) {
        double inv = 1.0 / info.mat;
        if (!std::isfinite(inv)) continue;
        for (size_t i = 0; i < info.frames.size(); ++i) {
            for (size_t j = 0; j < info.frames.size(); ++j)
                for (int x = 0; x < 6; ++x)
                    for (int y = 0; y < 6; ++y) H[(size_t)(15 * info.frames[i] + x) * D + 15 * info.frames[j] + y] -= info.h[i][x] * inv * info.h[j][y];
            for (int x = 0; x < 6; ++x) b[15 * info.frames[i] + x] -= info.h[i][x] * inv * info.vec;
        }
    }
    // marginalize the victim's 15x15 block (:547-581)
    const int R = D - 15;
    double Hvv[225], Hinv[225];
    for (int x = 0; x < 15; ++x)
        for (int y = 0; y < 15; ++y) Hvv[x * 15 + y] = H[(size_t)(15 * victim + x) * D + 15 * victim + y];
    if (!lu_inverse(Hvv, 15, Hinv)) return PVIO_ERR_INVALID_ARGUMENT;
    auto gidx = [&](int k) { return k < 15 * victim ? k : k + 15; }; // remaining index -> full index
    std::vector<double> C((size_t)R * R, 0.0), cv(R, 0.0), T((size_t)R * 15);
    for (int i = 0; i < R; ++i)
        for (int y = 0; y < 15; ++y) {
            double s = 0;
            for (int x = 0; x < 15; ++x) s += H[(size_t)gidx(i) * D + 15 * victim + x] * Hinv[x * 15 + y];
            T[(size_t)i * 15 + y] = s;
        }
    const int split = 15 * victim;
    for (int i = 0; i < R; ++i) {
        double s = 0;
        for (int y = 0; y < 15; ++y) s += T[(size_t)i * 15 + y] * b[15 * victim + y];
        cv[i] = b[gidx(i)] - s;
        for (int j = 0; j < R; ++j) {
            bool lower_left = (i >= split && j < split);
            if (lower_left) continue; // filled by transposing the upper-right block (:572-576)
            double s2 = 0;
            for (int y = 0; y < 15; ++y) s2 += T[(size_t)i * 15 + y] * H[(size_t)(15 * victim + y) * D + gidx(j)];
            C[(size_t)i * R + j] = H[(size_t)gidx(i) * D + gidx(j)] - s2;
        }
    }
    for (int i = split; i < R; ++i)
        for (int j = 0; j < split; ++j) C[(size_t)i * R + j] = C[(size_t)j * R + i];
    if (out->info_matrix) std::memcpy(out->info_matrix, C.data(), sizeof(double) * R * R);
    if (out->info_vector) std::memcpy(out->info_vector, cv.data(), sizeof(double) * R);
    // sqrt_infomat = sqrt(L) V^T, sqrt_infovec = L^{-1/2} V^T b, eigenvalues <= 1e-8 zeroed (:583-590)
    std::vector<double> w(R), V((size_t)R * R);
    { // SelfAdjointEigenSolver reads the lower triangle
        std::vector<double> Cs(C);
        for (int i = 0; i < R; ++i)
            for (int j = i + 1; j < R; ++j) Cs[(size_t)i * R + j] = Cs[(size_t)j * R + i];
        sym_eig(Cs.data(), R, w.data(), V.data());
    }
    out->n = N - 1;
    for (int k = 0; k < R; ++k) {
        double lam = w[k] > 1.0e-8 ? w[k] : 0.0, lam_inv = w[k] > 1.0e-8 ? 1.0 / w[k] : 0.0;
        double sl = std::sqrt(lam), sli = std::sqrt(lam_inv), acc = 0;
        for (int i = 0; i < R; ++i) {
            out->S[(size_t)k * R + i] = sl * V[(size_t)i * R + k];
            acc += V[(size_t)i * R + k] * cv[i];
        }
        out->s[k] = sli * acc;
    }
    return PVIO_OK;
}

int32_t oracle_preintegrate(int32_t n, const double *t, const double *w, const double *a, double t_end, const double *bg, const double *ba,
                            const pvio_imu_noise *nz, double *delta, double *cov, double *sqrt_inv_cov, double *jac) {
    ImuNoise noise{m3load(nz->cov_w), m3load(nz->cov_a), m3load(nz->cov_bg), m3load(nz->cov_ba)};
    PreInt pi;
    bool ok = preint_integrate(pi, n, t, w, a, t_end, vload(bg), vload(ba), noise);
    delta[0] = pi.dt;
    qstore(pi.dq, delta + 1);
    vstore(pi.dp, delta + 5);
    vstore(pi.dv, delta + 8);
    if (cov) std::memcpy(cov, pi.cov, sizeof pi.cov);
    if (sqrt_inv_cov) std::memcpy(sqrt_inv_cov, pi.sqrt_inv_cov, sizeof pi.sqrt_inv_cov);
    const M3 *js[5] = {&pi.dq_dbg, &pi.dp_dbg, &pi.dp_dba, &pi.dv_dbg, &pi.dv_dba};
    for (int k = 0; k < 5; ++k)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) jac[9 * k + 3 * i + j] = js[k]->m[i][j];
    return ok ? 0 : 1;
}

// ---- single-factor entry points (finite-difference tests, GPU factor-level parity) ----------------
void oracle_eval_reprojection(const double *st_tgt, const double *st_ref, double inv_depth, const double *z_ref, const double *z_tgt,
                              const double *cam_ref, const double *cam_tgt, const double *W, double *r, double *J) {
    eval_reprojection(st_tgt, st_ref, inv_depth, z_ref, z_tgt, ext_load(cam_ref), ext_load(cam_tgt), W, r, J);
}
void oracle_eval_preintegration(const double *si, const double *sj, const double *bias0, const double *delta, const double *U,
                                const double *jac, const double *imu_i, const double *imu_j, double *r, double *J) {
    PreIntFactor pre;
    pre.dt = delta[0], pre.dq = qload(delta + 1), pre.dp = vload(delta + 5), pre.dv = vload(delta + 8);
    pre.U = U;
    pre.dq_dbg = m3load(jac), pre.dp_dbg = m3load(jac + 9), pre.dp_dba = m3load(jac + 18), pre.dv_dbg = m3load(jac + 27), pre.dv_dba = m3load(jac + 36);
    eval_preintegration(si, sj, vload(bias0), vload(bias0 + 3), pre, ext_load(imu_i), ext_load(imu_j), r, J);
}
void oracle_eval_prior(int32_t n, const double *states, const double *lin, const double *S, const double *s, double *r, double *J) {
    std::vector<const double *> sp(n);
    for (int i = 0; i < n; ++i) sp[i] = states + 16 * i;
    eval_prior(n, sp.data(), lin, S, s, r, J);
}
void oracle_eval_rot_prior(const double *state, const double *q0, const double *W, double *r, double *J) { eval_rot_prior(state, q0, W, r, J); }
void oracle_eval_plane(int32_t K, const double *states, const double *cams, const double *z, const double *normal, double distance,
                       double sqrt_inv_cov, double *r, double *J) {
    std::vector<const double *> sp(K);
    std::vector<Ext> cs(K);
    for (int i = 0; i < K; ++i) sp[i] = states + 16 * i, cs[i] = ext_load(cams + 7 * i);
    eval_plane(K, sp.data(), cs.data(), z, normal, distance, sqrt_inv_cov, 1.0, r, J);
}
void oracle_plus(const double *state, const double *delta15, double *out) { // one frame: q (+) theta, rest additive
    Q q = qnormalized(qmul(qload(state), expmap(mk(delta15[0], delta15[1], delta15[2]))));
    qstore(q, out);
    for (int k = 0; k < 12; ++k) out[4 + k] = state[4 + k] + delta15[3 + k];
}
void oracle_expmap(const double *w, double *q) { qstore(expmap(vload(w)), q); }
void oracle_logmap(const double *q, double *w) { vstore(logmap(qload(q)), w); }
void oracle_right_jacobian(const double *w, double *J) {
    M3 m = right_jacobian(vload(w));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) J[3 * i + j] = m.m[i][j];
}
} // extern "C"
