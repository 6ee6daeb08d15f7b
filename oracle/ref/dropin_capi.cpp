// dropin_capi.cpp -- TEST INFRASTRUCTURE: the drop-in A/B on the reference's REAL object graph (oracle/ref/Makefile, target dropin).
//
// libpvio_dropin[_emu].so is libpvio_ref.so with three objects swapped: the reference's estimation/bundle_adjustor.cpp and estimation/pnp.cpp
// (+ mini-Ceres below them) are left out, and the PRODUCT's pvio_amd/host/{bundle_adjustor,pnp,pnp_solve}.cpp are linked in their place,
// compiled with -DPVIO_HOST_USE_REFERENCE_TYPES against the reference's own headers -- what INTEGRATION.md tells a PVIO maintainer to do.
// Everything else is the same: the reference's map/{frame,track,map,plane}.cpp, estimation/{factor,preintegrator}.cpp,
// geometry/lie_algebra.cpp, core/plane_extractor.cpp, compiled unedited from /root/reference; the functional mini-Eigen; the window
// builder of ref_window.h.  So `pvio::BundleAdjustor().solve(map, config, use_inertial)` below is the call
// core/sliding_window_tracker.cpp:113 makes, on a pvio::Map built by the reference's own code, and it lands in the HIP back-end
// (libpvio_hip.so; the fiber emulator of tests/hipemu for the CPU suite).
//
// Same entry points as ref_capi.cpp with the prefix dropin_, same flat layout, so tests diff the two libraries call by call:
// every Frame::pose / motion, Track::landmark.inv_depth / quality, TF_VALID / TF_PLANE, Plane::tracks, and the new prior.
// The two C-ABI calls of the adapter are wrapped at link time (-Wl,--wrap) only to hand trace buffers to the solve the adapter issues:
// the adapter source is not touched and asks for no trace itself.
#include "ref_window.h"

#include <marginalization_error_cost.h> // the Ceres-free holder of pvio_amd/host/dropin (include path of the dropin objects)

using namespace pvio;
using namespace ref_window;

extern "C" int32_t __real_pvio_hip_ba_solve(pvio_hip_ctx *, const pvio_ba_problem *, pvio_ba_state *, pvio_ba_summary *);

namespace {
thread_local pvio_ba_summary *g_capture = nullptr; // where the next wrapped solve leaves its summary + trace
thread_local int g_capture_landmarks = 0;
thread_local std::vector<double> g_states;
} // namespace

extern "C" {

int32_t __wrap_pvio_hip_ba_solve(pvio_hip_ctx *ctx, const pvio_ba_problem *pb, pvio_ba_state *st, pvio_ba_summary *sum) {
    if (!g_capture) return __real_pvio_hip_ba_solve(ctx, pb, st, sum);
    pvio_ba_summary own = *sum;
    const size_t S = (size_t)16 * pb->n_frames + pb->n_landmarks;
    g_states.assign((size_t)g_capture->trace_capacity * S, 0.0);
    own.trace_capacity = g_capture->trace_capacity, own.trace = g_capture->trace, own.trace_states = g_states.data(), own.trace_len = 0;
    const int32_t rc = __real_pvio_hip_ba_solve(ctx, pb, st, &own);
    // frame part of the per-iteration states only: the landmark order of the flattened problem is the adapter's business
    if (g_capture->trace_states)
        for (int k = 0; k < own.trace_len; ++k)
            std::memcpy(g_capture->trace_states + (size_t)k * (16 * pb->n_frames + g_capture_landmarks), g_states.data() + (size_t)k * S, sizeof(double) * 16 * pb->n_frames);
    double *ts = g_capture->trace_states;
    pvio_ba_iteration *tr = g_capture->trace;
    const int32_t cap = g_capture->trace_capacity;
    *g_capture = own;
    g_capture->trace = tr, g_capture->trace_states = ts, g_capture->trace_capacity = cap;
    *sum = own;
    sum->trace = nullptr, sum->trace_states = nullptr, sum->trace_capacity = 0, sum->trace_len = 0;
    return rc;
}

const char *dropin_version(void) { return PVIO_VERSION_STRING " reference map layer + pvio_amd/host adapter above the HIP C ABI"; }

// pvio::BundleAdjustor::solve -- the PRODUCT's (pvio_amd/host/bundle_adjustor.cpp) on the reference's Map.  Layout as ref_ba_solve;
// sum->trace_states rows are [16 N + T] with only the frame part filled.
int32_t dropin_ba_solve(const pvio_ba_problem *pb, double *frame_state, ref_tracks *trk, const ref_imu *imu, pvio_ba_summary *sum) {
    Window W;
    if (int rc = build_window(W, pb, frame_state, trk, imu)) return rc;
    if (sum) sum->trace_len = 0;
    g_capture = sum, g_capture_landmarks = trk ? trk->n_tracks : 0;
    const bool usable = BundleAdjustor().solve(W.map.get(), &W.config, pb->use_inertial != 0);
    g_capture = nullptr;
    read_back(W, frame_state, trk);
    if (sum && (sum->is_usable != 0) != usable) return PVIO_ERR_INVALID_ARGUMENT; // the adapter returns IsSolutionUsable() (:298)
    return PVIO_OK;
}

// pvio::BundleAdjustor::marginalize_frame -- the PRODUCT's.  The new prior is read through the holder's accessors.
int32_t dropin_ba_marginalize(const pvio_ba_problem *pb, const double *frame_state, ref_tracks *trk, const ref_imu *imu, int32_t victim, pvio_ba_prior *out) {
    Window W;
    std::vector<double> fs(frame_state, frame_state + 16 * pb->n_frames);
    if (int rc = build_window(W, pb, fs.data(), trk, nullptr)) return rc;
    if (imu) {
        for (int j = 1; j < pb->n_frames; ++j) {
            const int b = imu->ptr[j], e = imu->ptr[j + 1];
            if (e <= b) continue;
            W.set_imu(j, e - b, imu->t + b, imu->w + 3 * b, imu->a + 3 * b, imu->noise);
            W.frames[j]->preintegration.integrate(imu->frame_t[j], W.frames[j - 1]->motion.bg, W.frames[j - 1]->motion.ba, true, true);
        }
    }
    BundleAdjustor().marginalize_frame(W.map.get(), (size_t)victim);
    Factor *f = W.map->get_marginalization_factor();
    if (!f) return PVIO_ERR_INVALID_ARGUMENT;
    const MarginalizationErrorCost *cost = f->get_cost_function<MarginalizationErrorCost>();
    const int n = (int)cost->related_frames().size(), D = 15 * n;
    out->n = n;
    const matrix<> &S = cost->sqrt_information();
    const vector<> &s = cost->information_vector();
    for (int a = 0; a < D; ++a) {
        out->s[a] = s[a];
        for (int b = 0; b < D; ++b) out->S[(size_t)a * D + b] = S(a, b);
    }
    for (int a = 0; a < D && (out->info_matrix || out->info_vector); ++a) {
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
    return PVIO_OK;
}

// the keyframe cycle (ref_window.h: Map::marginalize_frame + solve on one Map) with the product behind both calls
int32_t dropin_marginalize_then_solve(const pvio_ba_problem *pb, const double *frame_state, ref_tracks *trk, const ref_imu *imu, int32_t victim, double *out_state,
                                      int32_t *usable) {
    return cycle_marginalize_then_solve(pb, frame_state, trk, imu, victim, out_state, usable);
}

int32_t dropin_ba_reprojection_error(const pvio_ba_problem *pb, const double *frame_state, ref_tracks *trk, double *out) {
    Window W;
    std::vector<double> fs(frame_state, frame_state + 16 * pb->n_frames);
    if (int rc = build_window(W, pb, fs.data(), trk, nullptr)) return rc;
    *out = BundleAdjustor().compute_reprojection_error(W.map.get());
    return PVIO_OK;
}

// visual_inertial_pnp -- the PRODUCT's (pvio_amd/host/pnp.cpp) on the reference's Map.  Layout as ref_pnp; *iterations is not reported (-1).
int32_t dropin_pnp(const pvio_ba_problem *pb, const double *frame_state, ref_tracks *trk, double *new_state, const double *new_cam, const double *new_imu,
                   const double *new_W, const double *new_K, int32_t n_new_obs, const int32_t *new_obs_track, const double *new_obs_z,
                   const double *new_delta, const double *new_U, const double *new_jac, int32_t use_inertial, int32_t *iterations) {
    Window W;
    std::vector<double> fs(frame_state, frame_state + 16 * pb->n_frames);
    if (int rc = build_window(W, pb, fs.data(), trk, nullptr)) return rc;
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
    if (iterations) *iterations = -1;
    for (int k = 0; k < n_new_obs; ++k) W.tracks[new_obs_track[k]]->remove_keypoint(f.get(), false);
    return PVIO_OK;
}

} // extern "C"
