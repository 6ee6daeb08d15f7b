// ref_capi.cpp -- extern "C" harness around the REFERENCE'S OWN code (TEST INFRASTRUCTURE, oracle/ref/Makefile).
//
// Everything numerical below the `extern "C"` line is the reference's: this file only (1) turns flat arrays into the
// reference's object graph -- pvio::Map / Frame / Track / Plane / Factor, built through their public interface
// (map/{map,frame,track,plane}.h) -- (2) calls the reference's functions
//     BundleAdjustor::solve / marginalize_frame / compute_reprojection_error   estimation/bundle_adjustor.cpp:63-599
//     ReprojectionErrorCost / PreIntegrationErrorCost / MarginalizationErrorCost / AugmentedPlaneDistanceErrorCost ::Evaluate
//     QuaternionParameterization::Plus / ComputeJacobian, PreIntegrator::integrate, expmap / logmap / right_jacobian
//     visual_inertial_pnp                                                      estimation/pnp.cpp:32-100
// and (3) copies the results back into flat arrays laid out like the oracle's entry points (oracle/oracle_ba.cpp), so
// tests can hold `oracle_*` against `ref_*` call by call.  <Eigen/Eigen> and <ceres/ceres.h> are the stand-ins of
// oracle/ref/{eigen,ceres}: ceres::Solve is a restatement (A8 stays unpinned), the rest of the arithmetic is the reference's
// source text compiled as it is.
#include <ceres/ceres.h>
#include "ref_window.h"
#include <pvio/estimation/ceres/augmented_plane_distance_error_cost.h>
#include <pvio/estimation/ceres/marginalization_error_cost.h>
#include <pvio/estimation/ceres/preintegration_error_cost.h>
#include <pvio/estimation/ceres/quaternion_parameterization.h>
#include <pvio/estimation/ceres/reprojection_error_cost.h>
#include <pvio/utility/poisson_disk_filter.h>

using namespace pvio;

using namespace ref_window;

namespace {

// local (tangent) Jacobian of a q block: J_global (rows x 4, row-major) * QuaternionParameterization::ComputeJacobian (4 x 3)
void q_local(const double *q, const double *Jg, int rows, double *out /* rows x 3 */, int out_stride) {
    QuaternionParameterization qp;
    double P[12];
    qp.ComputeJacobian(q, P);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += Jg[4 * r + k] * P[3 * k + c];
            out[(size_t)r * out_stride + c] = s;
        }
}
const double kId7[7] = {0, 0, 0, 1, 0, 0, 0};
const double kW2[4] = {1, 0, 0, 1};

} // namespace

extern "C" {

const char *ref_version(void) { return PVIO_VERSION_STRING " reference sources, mini-Eigen / mini-Ceres stand-ins"; }

void ref_expmap(const double *w, double *q) {
    quaternion r = expmap(vector<3>(w[0], w[1], w[2]));
    q[0] = r.x(), q[1] = r.y(), q[2] = r.z(), q[3] = r.w();
}
void ref_logmap(const double *q, double *w) {
    vector<3> r = logmap(quaternion(q[3], q[0], q[1], q[2]));
    w[0] = r(0), w[1] = r(1), w[2] = r(2);
}
void ref_right_jacobian(const double *w, double *J) {
    matrix<3> m = right_jacobian(vector<3>(w[0], w[1], w[2]));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) J[3 * i + j] = m(i, j);
}
// one frame: QuaternionParameterization::Plus on q, addition elsewhere (the reference adds p, v, bg, ba as plain blocks)
void ref_plus(const double *state, const double *delta15, double *out) {
    QuaternionParameterization qp;
    qp.Plus(state, delta15, out);
    for (int k = 0; k < 12; ++k) out[4 + k] = state[4 + k] + delta15[3 + k];
}

// PreIntegrator::integrate(t_end, bg, ba, true, true) -- preintegrator.cpp:84-100
int32_t ref_preintegrate(int32_t n, const double *t, const double *w, const double *a, double t_end, const double *bg, const double *ba,
                         const pvio_imu_noise *nz, double *delta, double *cov, double *sqrt_inv_cov, double *jac) {
    PreIntegrator pre;
    pre.cov_w = m3_rowmajor(nz->cov_w), pre.cov_a = m3_rowmajor(nz->cov_a), pre.cov_bg = m3_rowmajor(nz->cov_bg), pre.cov_ba = m3_rowmajor(nz->cov_ba);
    for (int k = 0; k < n; ++k) {
        ImuData d;
        d.t = t[k], d.w = vector<3>(w[3 * k], w[3 * k + 1], w[3 * k + 2]), d.a = vector<3>(a[3 * k], a[3 * k + 1], a[3 * k + 2]);
        pre.data.push_back(d);
    }
    const bool ok = pre.integrate(t_end, vector<3>(bg[0], bg[1], bg[2]), vector<3>(ba[0], ba[1], ba[2]), true, true);
    delta[0] = pre.delta.t;
    delta[1] = pre.delta.q.x(), delta[2] = pre.delta.q.y(), delta[3] = pre.delta.q.z(), delta[4] = pre.delta.q.w();
    for (int k = 0; k < 3; ++k) delta[5 + k] = pre.delta.p(k), delta[8 + k] = pre.delta.v(k);
    for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) {
            if (cov) cov[15 * i + j] = pre.delta.cov(i, j);
            if (sqrt_inv_cov) sqrt_inv_cov[15 * i + j] = pre.delta.sqrt_inv_cov(i, j);
        }
    const matrix<3> *js[5] = {&pre.jacobian.dq_dbg, &pre.jacobian.dp_dbg, &pre.jacobian.dp_dba, &pre.jacobian.dv_dbg, &pre.jacobian.dv_dba};
    for (int k = 0; k < 5; ++k)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) jac[9 * k + 3 * i + j] = (*js[k])(i, j);
    return ok ? 0 : 1;
}

// ReprojectionErrorCost::Evaluate (reprojection_error_cost.h:40-120).  J: 2 x 13 row-major, local coordinates
// [theta_tgt p_tgt theta_ref p_ref inv_depth] -- the layout of oracle_eval_reprojection.
void ref_eval_reprojection(const double *st_tgt, const double *st_ref, double inv_depth, const double *z_ref, const double *z_tgt,
                           const double *cam_ref, const double *cam_tgt, const double *Wm, double *r, double *J) {
    Window W;
    W.add_frame(st_ref, cam_ref, kId7, kW2, nullptr, 0.0, false);
    W.add_frame(st_tgt, cam_tgt, kId7, Wm, nullptr, 1.0, false);
    const int32_t of[2] = {0, 1};
    const double oz[4] = {z_ref[0], z_ref[1], z_tgt[0], z_tgt[1]};
    Track *t = W.add_track(2, of, oz);
    t->landmark.inv_depth = inv_depth;
    ReprojectionErrorCost *cost = W.frames[1]->get_reprojection_factor(0)->get_cost_function<ReprojectionErrorCost>();
    const double *params[5] = {st_tgt, st_tgt + 4, st_ref, st_ref + 4, &inv_depth};
    double jq_t[8], jp_t[6], jq_r[8], jp_r[6], jd[2];
    double *jac[5] = {jq_t, jp_t, jq_r, jp_r, jd};
    cost->Evaluate(params, r, J ? jac : nullptr);
    if (!J) return;
    q_local(st_tgt, jq_t, 2, J + 0, 13);
    q_local(st_ref, jq_r, 2, J + 6, 13);
    for (int a = 0; a < 2; ++a) {
        for (int c = 0; c < 3; ++c) J[13 * a + 3 + c] = jp_t[3 * a + c], J[13 * a + 9 + c] = jp_r[3 * a + c];
        J[13 * a + 12] = jd[a];
    }
}

// PreIntegrationErrorCost::Evaluate (preintegration_error_cost.h:40-160).  bias0 = the LIVE frame_i->motion.{bg,ba} the functor
// reads (:57-58); si / sj = the parameter blocks.  J: 15 x 30 row-major, local coordinates (frame i then frame j).
void ref_eval_preintegration(const double *si, const double *sj, const double *bias0, const double *delta, const double *U, const double *jacb,
                             const double *imu_i, const double *imu_j, double *r, double *J) {
    Window W;
    double live_i[16];
    std::memcpy(live_i, si, sizeof live_i);
    std::memcpy(live_i + 10, bias0, 6 * sizeof(double));
    W.add_frame(live_i, kId7, imu_i, kW2, nullptr, 0.0, false);
    W.add_frame(sj, kId7, imu_j, kW2, nullptr, 1.0, false);
    W.set_preintegration(1, delta, U, jacb);
    PreIntegrationErrorCost *cost = W.frames[1]->get_preintegration_factor()->get_cost_function<PreIntegrationErrorCost>();
    const double *params[10] = {si, si + 4, si + 7, si + 10, si + 13, sj, sj + 4, sj + 7, sj + 10, sj + 13};
    double jb[10][60];
    double *jac[10];
    for (int k = 0; k < 10; ++k) jac[k] = jb[k];
    cost->Evaluate(params, r, J ? jac : nullptr);
    if (!J) return;
    for (int f = 0; f < 2; ++f) {
        q_local(f == 0 ? si : sj, jb[5 * f], 15, J + 15 * f, 30);
        for (int blk = 1; blk < 5; ++blk)
            for (int a = 0; a < 15; ++a)
                for (int c = 0; c < 3; ++c) J[30 * a + 15 * f + 3 * blk + c] = jb[5 * f + blk][3 * a + c];
    }
}

// MarginalizationErrorCost::Evaluate (marginalization_error_cost.h:53-94).  J: 15n x 15n row-major, local coordinates.
void ref_eval_prior(int32_t n, const double *states, const double *lin, const double *S, const double *s, double *r, double *J) {
    Window W;
    std::vector<int32_t> pf(n);
    for (int i = 0; i < n; ++i) W.add_frame(states + 16 * i, kId7, kId7, kW2, nullptr, double(i), false), pf[i] = i;
    W.set_prior(n, pf.data(), S, s, lin);
    MarginalizationErrorCost *cost = W.map->get_marginalization_factor()->get_cost_function<MarginalizationErrorCost>();
    const int D = 15 * n;
    std::vector<const double *> params(5 * n);
    std::vector<std::vector<double>> jb(5 * n);
    std::vector<double *> jac(5 * n);
    for (int i = 0; i < n; ++i) {
        const double *st = states + 16 * i;
        const double *blk[5] = {st, st + 4, st + 7, st + 10, st + 13};
        for (int k = 0; k < 5; ++k) params[5 * i + k] = blk[k], jb[5 * i + k].assign((size_t)D * (k == 0 ? 4 : 3), 0.0), jac[5 * i + k] = jb[5 * i + k].data();
    }
    cost->Evaluate(params.data(), r, J ? jac.data() : nullptr);
    if (!J) return;
    for (int i = 0; i < n; ++i) {
        q_local(states + 16 * i, jb[5 * i].data(), D, J + 15 * i, D);
        for (int k = 1; k < 5; ++k)
            for (int a = 0; a < D; ++a)
                for (int c = 0; c < 3; ++c) J[(size_t)a * D + 15 * i + 3 * k + c] = jb[5 * i + k][3 * a + c];
    }
}

// AugmentedPlaneDistanceErrorCost::Evaluate (augmented_plane_distance_error_cost.h:53-136), regularization weight 1 as
// constructed at bundle_adjustor.cpp:183.  states / cams: K observing frames in ascending frame order.  J: K x 6 row-major
// (theta, p per frame; the plane blocks are constant in the reference and are not returned).
void ref_eval_plane(int32_t K, const double *states, const double *cams, const double *z, const double *normal, double distance, double sqrt_inv_cov,
                    double *r, double *J) {
    Window W;
    std::vector<int32_t> of(K);
    for (int i = 0; i < K; ++i) W.add_frame(states + 16 * i, cams + 7 * i, kId7, kW2, nullptr, double(i), false), of[i] = i;
    Track *t = W.add_track(K, of.data(), z);
    AugmentedPlaneDistanceErrorCost cost(t, sqrt_inv_cov);
    std::vector<const double *> params(2 * K + 2);
    std::vector<std::vector<double>> jb(2 * K + 2);
    std::vector<double *> jac(2 * K + 2);
    for (int i = 0; i < K; ++i) {
        params[2 * i] = states + 16 * i, params[2 * i + 1] = states + 16 * i + 4;
        jb[2 * i].assign(4, 0.0), jb[2 * i + 1].assign(3, 0.0);
        jac[2 * i] = jb[2 * i].data(), jac[2 * i + 1] = jb[2 * i + 1].data();
    }
    params[2 * K] = normal, params[2 * K + 1] = &distance;
    jac[2 * K] = nullptr, jac[2 * K + 1] = nullptr;
    cost.Evaluate(params.data(), r, J ? jac.data() : nullptr);
    if (!J) return;
    for (int i = 0; i < K; ++i) {
        q_local(states + 16 * i, jb[2 * i].data(), 1, J + 6 * i, 6);
        for (int c = 0; c < 3; ++c) J[6 * i + 3 + c] = jb[2 * i + 1][c];
    }
}

// BundleAdjustor::solve (bundle_adjustor.cpp:63-299) on the window the flat arrays describe; frame_state and the in/out
// fields of `trk` are updated like the reference updates its Map.  The landmark fields of `pb` are ignored: the tracks come
// from `trk`.  sum->trace / trace_states ([slot][16 N + T]: user state after every iteration, inverse depth of every track)
// are filled through mini-Ceres' observer hook.
int32_t ref_ba_solve(const pvio_ba_problem *pb, double *frame_state, ref_tracks *trk, const ref_imu *imu, pvio_ba_summary *sum) {
    Window W;
    if (int rc = build_window(W, pb, frame_state, trk, imu)) return rc;
    const int N = pb->n_frames, T = trk ? trk->n_tracks : 0;
    if (sum) sum->trace_len = 0;
    ceres::mini::set_observer([&](const ceres::IterationSummary &s) {
        if (!sum || !sum->trace || sum->trace_len >= sum->trace_capacity) return;
        pvio_ba_iteration &rec = sum->trace[sum->trace_len];
        rec.iteration = s.iteration, rec.step_is_valid = s.step_is_valid, rec.step_is_successful = s.step_is_successful, rec.reserved = 0;
        rec.cost = s.cost, rec.cost_change = s.cost_change, rec.gradient_max_norm = s.gradient_max_norm, rec.step_norm = s.step_norm;
        rec.relative_decrease = s.relative_decrease, rec.trust_region_radius = s.trust_region_radius, rec.mu = s.mu;
        if (sum->trace_states) {
            double *dst = sum->trace_states + (size_t)sum->trace_len * (16 * N + T);
            for (int i = 0; i < N; ++i) get_state(W.frames[i], dst + 16 * i);
            for (int t = 0; t < T; ++t) dst[16 * N + t] = W.tracks[t]->landmark.inv_depth;
        }
        sum->trace_len++;
    });
    const bool usable = BundleAdjustor().solve(W.map.get(), &W.config, pb->use_inertial != 0);
    ceres::mini::set_observer(nullptr);
    read_back(W, frame_state, trk);
    if (sum) {
        const ceres::Solver::Summary &cs = ceres::mini::last_summary();
        sum->termination = cs.termination_type == ceres::CONVERGENCE ? PVIO_TERM_CONVERGENCE
                                                                     : (cs.termination_type == ceres::NO_CONVERGENCE ? PVIO_TERM_NO_CONVERGENCE : PVIO_TERM_FAILURE);
        sum->is_usable = usable ? 1 : 0;
        sum->num_iterations = cs.iterations_started;
        sum->num_successful_steps = cs.num_successful_steps;
        sum->initial_cost = cs.initial_cost, sum->final_cost = cs.final_cost;
        sum->solve_seconds = cs.total_time_in_seconds, sum->device_seconds = 0;
    }
    return PVIO_OK;
}

// per-iteration records of the last ceres::Solve on this thread (any of the calls above); returns the count
int32_t ref_last_trace(pvio_ba_iteration *out, int32_t capacity, int32_t *termination) {
    const ceres::Solver::Summary &cs = ceres::mini::last_summary();
    int32_t n = 0;
    for (const ceres::IterationSummary &s : cs.iterations) {
        if (n >= capacity) break;
        pvio_ba_iteration &rec = out[n++];
        rec.iteration = s.iteration, rec.step_is_valid = s.step_is_valid, rec.step_is_successful = s.step_is_successful, rec.reserved = 0;
        rec.cost = s.cost, rec.cost_change = s.cost_change, rec.gradient_max_norm = s.gradient_max_norm, rec.step_norm = s.step_norm;
        rec.relative_decrease = s.relative_decrease, rec.trust_region_radius = s.trust_region_radius, rec.mu = s.mu;
    }
    if (termination) *termination = (int32_t)cs.termination_type;
    return n;
}

void ref_fault_injection(int32_t fail_factorizations, int32_t invalid_steps) { ceres::mini::set_fault_injection(fail_factorizations, invalid_steps); }

// BundleAdjustor::marginalize_frame (bundle_adjustor.cpp:348-599): the new prior of the remaining frames, in window order.
int32_t ref_ba_marginalize(const pvio_ba_problem *pb, const double *frame_state, ref_tracks *trk, const ref_imu *imu, int32_t victim, pvio_ba_prior *out) {
    Window W;
    std::vector<double> fs(frame_state, frame_state + 16 * pb->n_frames);
    if (int rc = build_window(W, pb, fs.data(), trk, nullptr)) return rc;
    if (imu) { // integrate like the last solve() did (:224): biases of frame j - 1
        for (int j = 1; j < pb->n_frames; ++j) {
            const int b = imu->ptr[j], e = imu->ptr[j + 1];
            if (e <= b) continue;
            W.set_imu(j, e - b, imu->t + b, imu->w + 3 * b, imu->a + 3 * b, imu->noise);
            W.frames[j]->preintegration.integrate(imu->frame_t[j], W.frames[j - 1]->motion.bg, W.frames[j - 1]->motion.ba, true, true);
        }
    }
    // (a frame without a pre-integration block keeps the factor Map::put_frame created, map.cpp:52-63, over a reset() block:
    // its sqrt_inv_cov is zero, so it contributes exactly nothing)
    BundleAdjustor().marginalize_frame(W.map.get(), (size_t)victim);
    Factor *f = W.map->get_marginalization_factor();
    if (!f) return PVIO_ERR_INVALID_ARGUMENT;
    MarginalizationErrorCost *cost = f->get_cost_function<MarginalizationErrorCost>();
    const int n = (int)cost->related_frames().size(), D = 15 * n;
    out->n = n;
    // S and s are private: recover them from Evaluate at the linearization point, where r = s and dr/d(p, v, bg, ba) = S columns,
    // dr/dtheta = S columns * Jr^-1(0) = S columns
    std::vector<const double *> params(5 * n);
    std::vector<std::vector<double>> jb(5 * n);
    std::vector<double *> jac(5 * n);
    std::vector<double> st((size_t)16 * n), r(D);
    for (int i = 0; i < n; ++i) {
        get_state(cost->related_frames()[i], &st[(size_t)16 * i]); // frames still hold the states the prior was linearized at
        const double *s0 = &st[(size_t)16 * i];
        const double *blk[5] = {s0, s0 + 4, s0 + 7, s0 + 10, s0 + 13};
        for (int k = 0; k < 5; ++k) params[5 * i + k] = blk[k], jb[5 * i + k].assign((size_t)D * (k == 0 ? 4 : 3), 0.0), jac[5 * i + k] = jb[5 * i + k].data();
    }
    cost->Evaluate(params.data(), r.data(), jac.data());
    for (int a = 0; a < D; ++a) {
        out->s[a] = r[a];
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 5; ++k)
                for (int c = 0; c < 3; ++c) out->S[(size_t)a * D + 15 * i + 3 * k + c] = jb[5 * i + k][(k == 0 ? 4 : 3) * a + c];
    }
    if (out->info_matrix || out->info_vector) { // S^T S and S^T s (what the eigendecomposition factored, up to the dropped eigenvalues)
        for (int a = 0; a < D; ++a) {
            if (out->info_vector) {
                double v = 0;
                for (int k = 0; k < D; ++k) v += out->S[(size_t)k * D + a] * out->s[k];
                out->info_vector[a] = v;
            }
            if (out->info_matrix)
                for (int b = 0; b < D; ++b) {
                    double v = 0;
                    for (int k = 0; k < D; ++k) v += out->S[(size_t)k * D + a] * out->S[(size_t)k * D + b];
                    out->info_matrix[(size_t)a * D + b] = v;
                }
        }
    }
    return PVIO_OK;
}

// the keyframe cycle (ref_window.h: Map::marginalize_frame + solve on one Map), the reference's code behind both calls
int32_t ref_marginalize_then_solve(const pvio_ba_problem *pb, const double *frame_state, ref_tracks *trk, const ref_imu *imu, int32_t victim, double *out_state,
                                   int32_t *usable) {
    return cycle_marginalize_then_solve(pb, frame_state, trk, imu, victim, out_state, usable);
}

// BundleAdjustor::compute_reprojection_error (bundle_adjustor.cpp:321-336)
int32_t ref_ba_reprojection_error(const pvio_ba_problem *pb, const double *frame_state, ref_tracks *trk, double *out) {
    Window W;
    std::vector<double> fs(frame_state, frame_state + 16 * pb->n_frames);
    if (int rc = build_window(W, pb, fs.data(), trk, nullptr)) return rc;
    *out = BundleAdjustor().compute_reprojection_error(W.map.get());
    return PVIO_OK;
}

// visual_inertial_pnp (pnp.cpp:32-100): the window + one NEW frame (state new_state, observing new_obs_track[k] at new_obs_z[k]);
// the pose (and, with use_inertial, nothing else: v / bg / ba are parameters of the IMU prior only) is updated in new_state.
int32_t ref_pnp(const pvio_ba_problem *pb, const double *frame_state, ref_tracks *trk, double *new_state, const double *new_cam, const double *new_imu,
                const double *new_W, const double *new_K, int32_t n_new_obs, const int32_t *new_obs_track, const double *new_obs_z,
                const double *new_delta, const double *new_U, const double *new_jac, int32_t use_inertial, int32_t *iterations) {
    Window W;
    std::vector<double> fs(frame_state, frame_state + 16 * pb->n_frames);
    if (int rc = build_window(W, pb, fs.data(), trk, nullptr)) return rc;
    // the new frame is NOT in the map (sliding_window_tracker.cpp:75-79 calls pnp before put_frame), but its tracks are
    std::unique_ptr<Frame> f = std::make_unique<Frame>();
    f->K = matrix<3>::Identity();
    f->K(0, 0) = new_K[0], f->K(1, 1) = new_K[1], f->K(0, 2) = new_K[2], f->K(1, 2) = new_K[3];
    f->sqrt_inv_cov(0, 0) = new_W[0], f->sqrt_inv_cov(0, 1) = new_W[1], f->sqrt_inv_cov(1, 0) = new_W[2], f->sqrt_inv_cov(1, 1) = new_W[3];
    set_state(f.get(), new_state);
    f->camera = ext(new_cam), f->imu = ext(new_imu);
    f->preintegration.reset();
    W.frames.push_back(f.get());
    if (use_inertial) W.set_preintegration((int)W.frames.size() - 1, new_delta, new_U, new_jac);
    for (int k = 0; k < n_new_obs; ++k) {
        const size_t idx = f->keypoint_num();
        f->append_keypoint(vector<2>(new_obs_z[2 * k], new_obs_z[2 * k + 1]));
        W.tracks[new_obs_track[k]]->add_keypoint(f.get(), idx);
    }
    visual_inertial_pnp(W.map.get(), f.get(), &W.config, use_inertial != 0);
    get_state(f.get(), new_state);
    if (iterations) *iterations = ceres::mini::last_summary().iterations_started;
    for (int k = 0; k < n_new_obs; ++k) W.tracks[new_obs_track[k]]->remove_keypoint(f.get(), false); // the frame dies before the map
    return PVIO_OK;
}


// ---- K3 / K5 of SURVEY.md section 8: Frame::track_keypoints (map/frame.cpp:89-139) and PoissonDiskFilter<2> (utility/poisson_disk_filter.h:25-130),
// the reference's own, behind the signatures of the oracle's restatement (oracle/oracle_front.cpp) and the product's (pvio_amd/host/feature_front.cpp,
// tests/host/front_shim.cpp) so that tests/test_host_frontend.py holds all three against each other. ----

// PoissonDiskFilter<2>: presets (preset_point, :40-44) then candidates in order (permit_point + preset_point = what frame.cpp:123-129 does;
// insert_point of :56-63 is the same two steps)
void ref_poisson_insert(double radius, int n_preset, const double *preset_xy, int n, const double *xy, uint8_t *accepted) {
    PoissonDiskFilter<2> filter(radius);
    for (int i = 0; i < n_preset; ++i) filter.preset_point(vector<2>(preset_xy[2 * i], preset_xy[2 * i + 1]));
    for (int i = 0; i < n; ++i) accepted[i] = filter.insert_point(vector<2>(xy[2 * i], xy[2 * i + 1])) ? 1 : 0;
}
}

namespace {
struct ScriptedImage : public DummyImage { // plays back a given LK result; remembers the initial guess it was handed
    std::vector<vector<2>> result, guess;
    std::vector<char> result_status;
    void track_keypoints(const Image *, const std::vector<vector<2>> &curr, std::vector<vector<2>> &next, std::vector<char> &status) const override {
        const_cast<ScriptedImage *>(this)->guess = next;
        next.resize(curr.size());
        status.assign(curr.size(), 0);
        for (size_t i = 0; i < curr.size() && i < result.size(); ++i) next[i] = result[i], status[i] = result_status[i];
    }
};
struct FrontConfig : public FlatConfig {
    double distance = 20.0;
    bool predict = false;
    double feature_tracker_min_keypoint_distance() const override { return distance; }
    bool feature_tracker_predict_keypoints() const override { return predict; }
};
// a feature-tracking map whose LAST frame has n keypoints, keypoint i on a track of track_length[i] observations (0: no track)
struct FrontWindow {
    Window W;
    Frame *curr = nullptr;
    std::shared_ptr<ScriptedImage> image = std::make_shared<ScriptedImage>();
    FrontWindow(int n, const double *kp_xy, const uint64_t *track_length, const double *K4, const double *cam, const double *imu) {
        uint64_t longest = 1;
        for (int i = 0; i < n; ++i) longest = std::max<uint64_t>(longest, track_length ? track_length[i] : 0);
        longest = std::min<uint64_t>(longest, 64); // the order only depends on the lengths' ORDER: lengths above 64 are capped by the caller
        const double st[16] = {0, 0, 0, 1};
        for (uint64_t f = 0; f < longest; ++f) W.add_frame(st, cam ? cam : kId7, imu ? imu : kId7, kW2, K4, double(f), false);
        curr = W.frames.back();
        curr->image = image;
        for (int i = 0; i < n; ++i) {
            const size_t idx = curr->keypoint_num();
            curr->append_keypoint(vector<2>(kp_xy[2 * i], kp_xy[2 * i + 1]));
            const uint64_t len = track_length ? track_length[i] : 1;
            if (len == 0) continue;
            Track *t = W.map->create_track();
            for (uint64_t f = longest - len; f + 1 < longest; ++f) { // earlier observations, oldest first
                Frame *fr = W.frames[f];
                const size_t k = fr->keypoint_num();
                fr->append_keypoint(vector<2>(0.0, 0.0));
                t->add_keypoint(fr, k);
            }
            t->add_keypoint(curr, idx);
        }
    }
};
} // namespace

extern "C" {
// Frame::track_keypoints :108-130 -- survivors of the LK call sorted by track length (std::sort), Poisson-disk acceptance in that order.
// next_xy: pixels (K = identity here); track_length[i] = 0: keypoint without a track; lengths must be <= 64 (ties and order are what matters).
void ref_select_tracked(int n, const double *next_xy, const uint64_t *track_length, double min_distance, uint8_t *status) {
    if (n <= 0) return;
    std::vector<double> zeros(2 * (size_t)n, 0.0);
    FrontWindow F(n, zeros.data(), track_length, nullptr, nullptr, nullptr);
    FrontConfig cfg;
    cfg.distance = min_distance, cfg.predict = false;
    F.image->result.resize((size_t)n), F.image->result_status.resize((size_t)n);
    for (int i = 0; i < n; ++i) F.image->result[(size_t)i] = vector<2>(next_xy[2 * i], next_xy[2 * i + 1]), F.image->result_status[(size_t)i] = (char)status[i];
    std::unique_ptr<Frame> next = std::make_unique<Frame>();
    next->K = matrix<3>::Identity();
    next->image = std::make_shared<DummyImage>();
    next->preintegration.reset();
    F.curr->track_keypoints(next.get(), &cfg);
    // which keypoints went on: those whose track (or, for a trackless keypoint, a new track) now has an observation in `next`
    for (int i = 0; i < n; ++i) {
        Track *t = F.curr->get_track((size_t)i);
        const bool kept = t && t->has_keypoint(next.get());
        // a trackless keypoint whose LK status was 1 is appended as well (:132-137 create_if_empty): oracle_select_tracked / select_tracked leave its status alone
        status[i] = kept ? 1 : 0;
    }
    for (int i = 0; i < n; ++i)
        if (Track *t = F.curr->get_track((size_t)i); t && t->has_keypoint(next.get())) t->remove_keypoint(next.get(), false); // `next` dies before the map
}

// Frame::track_keypoints :97-103 -- the gyro-only prediction handed to Image::track_keypoints as the initial guess (pixels of the next frame)
void ref_predict_keypoints(const double q_cam_i[4], const double q_imu_i[4], const double dq[4], const double q_imu_j[4], const double q_cam_j[4],
                           const double K_next[4], int n, const double *kp_xy, double *out_xy) {
    if (n <= 0) return;
    double cam_i[7] = {q_cam_i[0], q_cam_i[1], q_cam_i[2], q_cam_i[3], 0, 0, 0}, imu_i[7] = {q_imu_i[0], q_imu_i[1], q_imu_i[2], q_imu_i[3], 0, 0, 0};
    FrontWindow F(n, kp_xy, nullptr, K_next, cam_i, imu_i);
    FrontConfig cfg;
    cfg.predict = true;
    std::unique_ptr<Frame> next = std::make_unique<Frame>();
    next->K = matrix<3>::Identity();
    next->K(0, 0) = K_next[0], next->K(1, 1) = K_next[1], next->K(0, 2) = K_next[2], next->K(1, 2) = K_next[3];
    next->image = std::make_shared<DummyImage>();
    next->camera.q_cs = quaternion(q_cam_j[3], q_cam_j[0], q_cam_j[1], q_cam_j[2]), next->imu.q_cs = quaternion(q_imu_j[3], q_imu_j[0], q_imu_j[1], q_imu_j[2]);
    next->preintegration.reset();
    next->preintegration.delta.q = quaternion(dq[3], dq[0], dq[1], dq[2]);
    F.curr->track_keypoints(next.get(), &cfg); // the scripted image reports every track lost: nothing is appended
    for (int i = 0; i < n; ++i) out_xy[2 * i] = F.image->guess[(size_t)i][0], out_xy[2 * i + 1] = F.image->guess[(size_t)i][1];
}
} // extern "C"
