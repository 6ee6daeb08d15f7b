// chain_shim.cpp -- TEST INFRASTRUCTURE: runs a sequence that is already in memory through the headless driver (standin/headless.*)
// and writes the record stream of chain_log.h.  Compiled into BOTH chain libraries (tests/host/Makefile):
//   libpvio_chain_hip[_emu].so   the product: pvio_amd/host/*.cpp above libpvio_hip.so (or the emulated kernels)
//   libpvio_chain_oracle.so      -DCHAIN_ORACLE: the same driver, every arithmetic piece from the CPU oracle (oracle_chain.cpp)
// The product chain ALSO replays each of its window solves, marginalizations and PnP solves through the oracle on its own inputs
// (records 4, 5, 7 behind 2, 3, 6): the free-running chains drift apart by the rounding of the LK tracker, the replay does not.
// The window solves and marginalizations are recorded by wrapping the two C-ABI calls at link time (-Wl,--wrap=...): the adapter
// (bundle_adjustor.cpp) is not touched, and asks for no trace itself -- the wrapper adds the trace buffers.
#include <cstdint>
#include <cstring>
#include <memory>

#include "../../include/pvio_hip.h"
#include "chain_log.h"
#include "../../pvio_amd/host/pnp_problem.h"
#include "standin/headless.h"
#ifndef CHAIN_ORACLE
#include "../../pvio_amd/host/feature_front.h"
#endif

using namespace pvio;

#ifdef CHAIN_ORACLE
std::shared_ptr<Image> oracle_chain_make_image(const uint8_t *pixels, int w, int h, double t); // oracle_chain.cpp
static std::shared_ptr<Image> make_image(pvio_hip_ctx *, const uint8_t *px, int w, int h, double t) { return oracle_chain_make_image(px, w, h, t); }
#else
static std::shared_ptr<Image> make_image(pvio_hip_ctx *ctx, const uint8_t *px, int w, int h, double t) { return std::make_shared<HipImage>(ctx, px, w, h, w, t); }
#endif

#ifndef CHAIN_ORACLE
extern "C" { // liboracle.so: the REPLAY of every call of the product chain on the product chain's own inputs (records 4, 5, 7)
int32_t oracle_ba_solve(const pvio_ba_problem *, pvio_ba_state *, pvio_ba_summary *);
int32_t oracle_ba_marginalize(const pvio_ba_problem *, const pvio_ba_state *, int32_t, pvio_ba_prior *);
int32_t oracle_pnp_flat(const double *, const double *, const double *, int32_t, const double *, const double *, const double *, const double *, const double *, int32_t,
                        const double *, const double *, int32_t, const double *, const double *, const double *, const double *, const double *, int32_t, double *,
                        int32_t *, int32_t *, double *);
}
#endif

namespace {
constexpr int kTraceCap = 64;
struct SolveRun {
    std::vector<pvio_ba_iteration> trace = std::vector<pvio_ba_iteration>((size_t)kTraceCap);
    std::vector<double> states, fs, rho, quality;
    std::vector<uint8_t> valid;
    pvio_ba_summary sum;
    int32_t rc = 0;
};
template <class Fn>
void run_solve(Fn fn, const pvio_ba_problem *pb, const pvio_ba_state *in, const pvio_ba_summary *opts, SolveRun &R) {
    const int N = pb->n_frames, M = pb->n_landmarks;
    const size_t S = (size_t)16 * N + M;
    R.states.assign((size_t)kTraceCap * S, 0.0);
    R.fs.assign(in->frame_state, in->frame_state + 16 * N), R.rho.assign(in->lm_inv_depth, in->lm_inv_depth + M);
    R.quality.assign((size_t)M, 0.0), R.valid.assign((size_t)M, 1);
    R.sum = *opts;
    R.sum.trace_capacity = kTraceCap, R.sum.trace_len = 0, R.sum.trace = R.trace.data(), R.sum.trace_states = R.states.data();
    pvio_ba_state st{R.fs.data(), R.rho.data(), R.quality.data(), R.valid.data()};
    R.rc = fn(pb, &st, &R.sum);
}
void record_solve(int tag, const pvio_ba_problem *pb, const SolveRun &R) {
    const int N = pb->n_frames, M = pb->n_landmarks, len = R.sum.trace_len;
    const size_t S = (size_t)16 * N + M;
    std::vector<int64_t> I = {N, M, pb->n_obs, pb->use_inertial, pb->prior_n, R.sum.termination, R.sum.is_usable, R.sum.num_iterations, R.sum.num_successful_steps, len};
    std::vector<double> D = {R.sum.initial_cost, R.sum.final_cost};
    for (int k = 0; k < len; ++k) {
        const pvio_ba_iteration &t = R.trace[(size_t)k];
        I.push_back(t.iteration), I.push_back(t.step_is_valid), I.push_back(t.step_is_successful);
        D.insert(D.end(), {t.cost, t.cost_change, t.gradient_max_norm, t.step_norm, t.relative_decrease, t.trust_region_radius, t.mu});
    }
    for (int l = 0; l < M; ++l) I.push_back(R.valid[(size_t)l]);
    D.insert(D.end(), R.states.begin(), R.states.begin() + (std::ptrdiff_t)((size_t)len * S));
    D.insert(D.end(), R.fs.begin(), R.fs.end()), D.insert(D.end(), R.rho.begin(), R.rho.end()), D.insert(D.end(), R.quality.begin(), R.quality.end());
    chain_log::record(R.rc == 0 ? tag : -tag, I, D);
}
void record_marg(int tag, const pvio_ba_problem *pb, int32_t victim, int32_t rc, const double *S, const double *s) {
    const int n = pb->n_frames - 1;
    const size_t D15 = (size_t)15 * n;
    std::vector<double> D;
    if (rc == 0) D.insert(D.end(), S, S + D15 * D15), D.insert(D.end(), s, s + D15);
    chain_log::record(tag, {pb->n_frames, victim, n, rc}, D);
}
void record_pnp(int tag, const PnpProblem &pb, const double in[16], const double out[16], int iterations, int termination, double c0, double c1) {
    std::vector<double> D(in, in + 16);
    D.insert(D.end(), out, out + 16), D.push_back(c0), D.push_back(c1);
    chain_log::record(tag, {(int64_t)pb.factors.size(), (int64_t)pb.point_factors.size(), pb.use_inertial ? 1 : 0, iterations, termination}, D);
}
} // namespace

extern "C" {
int32_t __real_pvio_hip_ba_solve(pvio_hip_ctx *, const pvio_ba_problem *, pvio_ba_state *, pvio_ba_summary *);
int32_t __real_pvio_hip_ba_marginalize(pvio_hip_ctx *, const pvio_ba_problem *, const pvio_ba_state *, int32_t, pvio_ba_prior *);
dense::Summary __real__ZN4pvio9solve_pnpERKNS_10PnpProblemEPdi(const PnpProblem &, double *, int);

int32_t __wrap_pvio_hip_ba_solve(pvio_hip_ctx *ctx, const pvio_ba_problem *pb, pvio_ba_state *st, pvio_ba_summary *sum) {
    const int N = pb->n_frames, M = pb->n_landmarks;
    SolveRun R;
    run_solve([&](const pvio_ba_problem *p, pvio_ba_state *s, pvio_ba_summary *m) { return __real_pvio_hip_ba_solve(ctx, p, s, m); }, pb, st, sum, R);
    record_solve(2, pb, R);
#ifndef CHAIN_ORACLE
    { // the oracle on the very same window (states as they were BEFORE the solve): record 4
        SolveRun O;
        run_solve([&](const pvio_ba_problem *p, pvio_ba_state *s, pvio_ba_summary *m) { return oracle_ba_solve(p, s, m); }, pb, st, sum, O);
        record_solve(4, pb, O);
    }
#endif
    std::copy(R.fs.begin(), R.fs.end(), st->frame_state), std::copy(R.rho.begin(), R.rho.end(), st->lm_inv_depth);
    if (st->lm_quality) std::copy(R.quality.begin(), R.quality.end(), st->lm_quality);
    if (st->lm_valid) std::copy(R.valid.begin(), R.valid.end(), st->lm_valid);
    (void)N, (void)M;
    *sum = R.sum;
    sum->trace_capacity = 0, sum->trace_len = 0, sum->trace = nullptr, sum->trace_states = nullptr;
    return R.rc;
}

int32_t __wrap_pvio_hip_ba_marginalize(pvio_hip_ctx *ctx, const pvio_ba_problem *pb, const pvio_ba_state *st, int32_t victim, pvio_ba_prior *out) {
    const int32_t rc = __real_pvio_hip_ba_marginalize(ctx, pb, st, victim, out);
    record_marg(3, pb, victim, rc, out->S, out->s);
#ifndef CHAIN_ORACLE
    {
        const size_t D15 = (size_t)15 * (pb->n_frames - 1);
        std::vector<double> S(D15 * D15, 0.0), s(D15, 0.0);
        pvio_ba_prior o;
        std::memset(&o, 0, sizeof o);
        o.S = S.data(), o.s = s.data();
        const int32_t rc2 = oracle_ba_marginalize(pb, st, victim, &o);
        record_marg(5, pb, victim, rc2, S.data(), s.data());
    }
#endif
    return rc;
}

dense::Summary __wrap__ZN4pvio9solve_pnpERKNS_10PnpProblemEPdi(const PnpProblem &pb, double *state16, int max_iterations) {
    double in[16];
    std::memcpy(in, state16, sizeof in);
    const dense::Summary s = __real__ZN4pvio9solve_pnpERKNS_10PnpProblemEPdi(pb, state16, max_iterations);
    record_pnp(6, pb, in, state16, s.iterations, s.termination, s.initial_cost, s.final_cost);
#ifndef CHAIN_ORACLE
    {
        const size_t n = pb.factors.size(), m = pb.point_factors.size();
        std::vector<double> A(16 * n), Cm(7 * n), zr(2 * n), zt(2 * n), rho(n), pts(3 * m), zp(2 * m);
        for (size_t k = 0; k < n; ++k) {
            const PnpFactor &f = pb.factors[k];
            std::memcpy(&A[16 * k], f.anchor_state, 128), std::memcpy(&Cm[7 * k], f.anchor_cam, 56);
            zr[2 * k] = f.z_ref[0], zr[2 * k + 1] = f.z_ref[1], zt[2 * k] = f.z_tgt[0], zt[2 * k + 1] = f.z_tgt[1], rho[k] = f.inv_depth;
        }
        for (size_t k = 0; k < m; ++k) std::memcpy(&pts[3 * k], pb.point_factors[k].point, 24), zp[2 * k] = pb.point_factors[k].z_tgt[0], zp[2 * k + 1] = pb.point_factors[k].z_tgt[1];
        double x[16], costs[2] = {0, 0};
        std::memcpy(x, in, sizeof x);
        int32_t it = 0, term = 0;
        oracle_pnp_flat(pb.cam, pb.imu, pb.sqrt_inv_cov, (int32_t)n, A.data(), Cm.data(), zr.data(), zt.data(), rho.data(), (int32_t)m, pts.data(), zp.data(),
                        pb.use_inertial ? 1 : 0, pb.last_state, pb.last_imu, pb.delta, pb.sqrt_inv_cov_imu, pb.jac, max_iterations, x, &it, &term, costs);
        record_pnp(7, pb, in, x, it, term, costs[0], costs[1]);
    }
#endif
    return s;
}

int host_chain_run(int n_frames, int w, int h, const uint8_t *images, const double *image_t, int n_imu, const double *imu_t, const double *imu_w,
                   const double *imu_a, const double *K4, const double *q_bc, const double *p_bc, int n_gt, const double *gt /* [n_gt][8] t p q(xyzw) */,
                   int window, int keyframe_gap, double min_keypoint_distance, const char *log_path, double *out_pose /* [n_frames][8] */,
                   int32_t *stats /* [4] */, char *err, int err_len) {
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof opts);
    opts.world_size = 1, opts.use_graph = 1;
    if (pvio_hip_create(&opts, &ctx) != 0 || !ctx) {
        std::strncpy(err, "pvio_hip_create failed (no GPU?)", (size_t)err_len - 1);
        return -1;
    }
    chain_log::open(log_path);
    int rc = 0;
    try {
        auto cfg = HeadlessConfig::euroc();
        cfg->K.setZero();
        cfg->K(0, 0) = K4[0], cfg->K(1, 1) = K4[1], cfg->K(0, 2) = K4[2], cfg->K(1, 2) = K4[3], cfg->K(2, 2) = 1;
        cfg->q_bc = quaternion(q_bc[3], q_bc[0], q_bc[1], q_bc[2]);
        cfg->p_bc = vector<3>(p_bc[0], p_bc[1], p_bc[2]);
        cfg->window = (size_t)window, cfg->keyframe_gap = (size_t)keyframe_gap, cfg->min_keypoint_distance = min_keypoint_distance;
        HeadlessVio vio(cfg);
        std::vector<TimedPose> poses((size_t)n_gt);
        for (int i = 0; i < n_gt; ++i) {
            const double *g = gt + 8 * i;
            poses[(size_t)i].t = g[0];
            poses[(size_t)i].pose.p = vector<3>(g[1], g[2], g[3]);
            poses[(size_t)i].pose.q = quaternion(g[7], g[4], g[5], g[6]);
        }
        vio.set_bootstrap_trajectory(std::move(poses));
        int k = 0;
        for (int f = 0; f < n_frames; ++f) {
            while (k < n_imu && imu_t[k] <= image_t[f]) {
                vio.track_gyroscope(imu_t[k], imu_w[3 * k], imu_w[3 * k + 1], imu_w[3 * k + 2]);
                vio.track_accelerometer(imu_t[k], imu_a[3 * k], imu_a[3 * k + 1], imu_a[3 * k + 2]);
                ++k;
            }
            const OutputPose p = vio.track_camera(make_image(ctx, images + (size_t)f * w * h, w, h, image_t[f]));
            double *o = out_pose + 8 * f;
            o[0] = image_t[f];
            for (int c = 0; c < 3; ++c) o[1 + c] = p.p[c];
            o[4] = p.q.x(), o[5] = p.q.y(), o[6] = p.q.z(), o[7] = p.q.w();
            // the newest frame of the feature-tracking map: which tracks survived, which corners are new
            const Map *ft = vio.tracking_map();
            const Frame *last = ft && ft->frame_num() ? ft->get_frame(ft->frame_num() - 1) : nullptr;
            std::vector<int64_t> I = {f, last ? (int64_t)last->id() : -1, vio.initialized() ? 1 : 0, (int64_t)vio.window_frames(), last ? (int64_t)last->keypoint_num() : 0};
            std::vector<double> D;
            if (last)
                for (size_t i = 0; i < last->keypoint_num(); ++i) {
                    const Track *t = last->get_track(i);
                    I.push_back(t ? (int64_t)t->id() : 0), I.push_back(t ? (int64_t)t->keypoint_num() : 0);
                    D.push_back(last->get_keypoint(i)[0]), D.push_back(last->get_keypoint(i)[1]);
                }
            D.insert(D.end(), o, o + 8);
            chain_log::record(1, I, D);
            if (const Map *wm = vio.window()) { // the sliding window's tracks: which are valid, where they are
                std::vector<int64_t> WI = {f, (int64_t)wm->frame_num(), (int64_t)wm->track_num()};
                std::vector<double> WD;
                for (size_t i = 0; i < wm->track_num(); ++i) {
                    const Track *t = wm->get_track(i);
                    WI.push_back((int64_t)t->id()), WI.push_back(t->flag(TrackFlag::TF_VALID) ? 1 : 0), WI.push_back((int64_t)t->keypoint_num());
                    WD.push_back(t->landmark.inv_depth), WD.push_back(t->landmark.quality);
                }
                chain_log::record(8, WI, WD);
            }
        }
        stats[0] = vio.initialized() ? 1 : 0, stats[1] = (int32_t)vio.window_frames(), stats[2] = (int32_t)vio.keyframe_solves();
        stats[3] = vio.window() ? (int32_t)vio.window()->track_num() : 0;
    } catch (const std::exception &e) {
        std::strncpy(err, e.what(), (size_t)err_len - 1);
        rc = -1;
    }
    chain_log::close();
    pvio_hip_destroy(ctx);
    return rc;
}
}
