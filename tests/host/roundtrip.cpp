// roundtrip.cpp -- test harness for the host adapter (pvio_amd/host/bundle_adjustor.cpp): rebuilds a pvio::Map object
// graph from a flat window, calls pvio::BundleAdjustor exactly like the reference's SlidingWindowTracker does
// (`BundleAdjustor().solve(map, config, true)`, sliding_window_tracker.cpp:113) and copies the in-place results back.
#include <cstring>
#include <map>
#include <utility>

#include "../../include/pvio_hip.h"
#include "../../pvio_amd/host/pvio_min.h"

using namespace pvio;

namespace {
struct Cfg : Config {
    size_t iters;
    double plane_cov;
    size_t solver_iteration_limit() const override { return iters; }
    double plane_distance_cov() const override { return plane_cov; }
};

void build_map(const pvio_ba_problem *pb, const pvio_ba_state *st, Map &map, std::vector<Track *> &lm_tracks) {
    const int N = pb->n_frames;
    for (int i = 0; i < N; ++i) {
        auto f = std::make_unique<Frame>();
        f->id_ = i;
        f->flags[(size_t)FrameFlag::FF_FIX_POSE] = pb->frame_fixed[i] != 0;
        std::memcpy(f->camera.q_cs.c, pb->cam_extrinsic + 7 * i, 32);
        std::memcpy(f->imu.q_cs.c, pb->imu_extrinsic + 7 * i, 32);
        for (int k = 0; k < 3; ++k) f->camera.p_cs[k] = pb->cam_extrinsic[7 * i + 4 + k], f->imu.p_cs[k] = pb->imu_extrinsic[7 * i + 4 + k];
        f->sqrt_inv_cov(0, 0) = pb->sqrt_inv_cov[4 * i], f->sqrt_inv_cov(0, 1) = pb->sqrt_inv_cov[4 * i + 1];
        f->sqrt_inv_cov(1, 0) = pb->sqrt_inv_cov[4 * i + 2], f->sqrt_inv_cov(1, 1) = pb->sqrt_inv_cov[4 * i + 3];
        f->K(0, 0) = pb->intrinsics[4 * i], f->K(1, 1) = pb->intrinsics[4 * i + 1], f->K(0, 2) = pb->intrinsics[4 * i + 2], f->K(1, 2) = pb->intrinsics[4 * i + 3], f->K(2, 2) = 1;
        const double *s = st->frame_state + 16 * i;
        std::memcpy(f->pose.q.c, s, 32);
        for (int k = 0; k < 3; ++k) f->pose.p[k] = s[4 + k], f->motion.v[k] = s[7 + k], f->motion.bg[k] = s[10 + k], f->motion.ba[k] = s[13 + k];
        map.frames.push_back(std::move(f));
    }
    auto add_kp = [&](Track *t, int frame, const double *z) {
        Frame *f = map.get_frame(frame);
        vector<2> kp;
        kp[0] = z[0], kp[1] = z[1];
        f->keypoints.push_back(kp);
        f->tracks.push_back(t);
        t->keypoint_refs[f] = f->keypoints.size() - 1;
        t->life++;
    };
    for (int l = 0; l < pb->n_landmarks; ++l) {
        auto t = std::make_unique<Track>();
        t->id_ = l;
        t->set_flag(TrackFlag::TF_VALID, true);
        t->landmark.inv_depth = st->lm_inv_depth[l];
        add_kp(t.get(), pb->lm_anchor_frame[l], pb->lm_anchor_z + 2 * l);
        for (int o = pb->lm_obs_ptr[l]; o < pb->lm_obs_ptr[l + 1]; ++o) add_kp(t.get(), pb->obs_frame[o], pb->obs_z + 2 * o);
        lm_tracks.push_back(t.get());
        map.tracks.push_back(std::move(t));
    }
    // plane factors -> PLANE tracks grouped into planes by (normal, distance)
    std::map<std::pair<double, double>, Plane *> planes;
    for (int f = 0; f < pb->n_plane_factors; ++f) {
        auto key = std::make_pair(pb->plane_normal[3 * f] * 7 + pb->plane_normal[3 * f + 1] * 3 + pb->plane_normal[3 * f + 2], pb->plane_distance[f]);
        Plane *pl;
        if (!planes.count(key)) {
            auto p = std::make_unique<Plane>();
            p->id_ = map.planes.size();
            for (int k = 0; k < 3; ++k) p->parameter.normal[k] = pb->plane_normal[3 * f + k];
            p->parameter.distance = pb->plane_distance[f];
            planes[key] = pl = p.get();
            map.planes.push_back(std::move(p));
        } else {
            pl = planes[key];
        }
        auto t = std::make_unique<Track>();
        t->id_ = pb->n_landmarks + f;
        t->set_flag(TrackFlag::TF_PLANE, true);
        t->landmark.plane_id = pl->id();
        t->landmark.inv_depth = 1.0;
        for (int o = pb->plane_obs_ptr[f]; o < pb->plane_obs_ptr[f + 1]; ++o) add_kp(t.get(), pb->plane_obs_frame[o], pb->plane_obs_z + 2 * o);
        pl->tracks.insert(t.get());
        map.tracks.push_back(std::move(t));
    }
    if (pb->prior_n > 0) {
        auto pr = std::make_unique<MarginalizationPrior>();
        const size_t D = 15 * (size_t)pb->prior_n;
        pr->sqrt_infomat.assign(pb->prior_S, pb->prior_S + D * D);
        pr->sqrt_infovec.assign(pb->prior_s, pb->prior_s + D);
        for (int i = 0; i < pb->prior_n; ++i) {
            pr->frames.push_back(map.get_frame(pb->prior_frames[i]));
            PoseState p0;
            MotionState m0;
            const double *s = pb->prior_lin_state + 16 * i;
            std::memcpy(p0.q.c, s, 32);
            for (int k = 0; k < 3; ++k) p0.p[k] = s[4 + k], m0.v[k] = s[7 + k], m0.bg[k] = s[10 + k], m0.ba[k] = s[13 + k];
            pr->pose_0.push_back(p0), pr->motion_0.push_back(m0);
        }
        map.set_marginalization_factor(std::move(pr));
    }
}

void copy_back(const Map &map, const std::vector<Track *> &lm_tracks, pvio_ba_state *st) {
    for (size_t i = 0; i < map.frame_num(); ++i) {
        Frame *f = map.get_frame(i);
        double *s = st->frame_state + 16 * i;
        std::memcpy(s, f->pose.q.c, 32);
        for (int k = 0; k < 3; ++k) s[4 + k] = f->pose.p[k], s[7 + k] = f->motion.v[k], s[10 + k] = f->motion.bg[k], s[13 + k] = f->motion.ba[k];
    }
    for (size_t l = 0; l < lm_tracks.size(); ++l) {
        st->lm_inv_depth[l] = lm_tracks[l]->landmark.inv_depth;
        if (st->lm_quality) st->lm_quality[l] = lm_tracks[l]->landmark.quality;
        if (st->lm_valid) st->lm_valid[l] = lm_tracks[l]->flag(TrackFlag::TF_VALID) ? 1 : 0;
    }
}
} // namespace

extern "C" {

// imu_ptr[N+1] indexes the concatenated IMU samples of frame j (those between frame j-1 and j); imu_tend[j] = image time
int host_roundtrip_solve(const pvio_ba_problem *pb, pvio_ba_state *st, const int32_t *imu_ptr, const double *imu_t, const double *imu_w,
                         const double *imu_a, const double *imu_tend, const pvio_imu_noise *nz, double plane_cov, int32_t *usable) {
    Map map;
    std::vector<Track *> lm_tracks;
    build_map(pb, st, map, lm_tracks);
    if (pb->use_inertial && imu_ptr)
        for (int j = 1; j < pb->n_frames; ++j) {
            Frame *f = map.get_frame(j);
            f->image_t = imu_tend[j];
            std::memcpy(f->preintegration.cov_w, nz->cov_w, 72), std::memcpy(f->preintegration.cov_a, nz->cov_a, 72);
            std::memcpy(f->preintegration.cov_bg, nz->cov_bg, 72), std::memcpy(f->preintegration.cov_ba, nz->cov_ba, 72);
            for (int k = imu_ptr[j]; k < imu_ptr[j + 1]; ++k) {
                ImuData d;
                d.t = imu_t[k];
                for (int c = 0; c < 3; ++c) d.w[c] = imu_w[3 * k + c], d.a[c] = imu_a[3 * k + c];
                f->preintegration.data.push_back(d);
            }
            f->has_preintegration_factor = true;
        }
    Cfg cfg;
    cfg.iters = pb->max_iterations, cfg.plane_cov = plane_cov;
    const bool ok = BundleAdjustor().solve(&map, &cfg, pb->use_inertial != 0);
    if (usable) *usable = ok ? 1 : 0;
    copy_back(map, lm_tracks, st);
    return 0;
}

int host_roundtrip_marginalize(const pvio_ba_problem *pb, const pvio_ba_state *st, int32_t victim, double *S, double *s) {
    Map map;
    std::vector<Track *> lm_tracks;
    build_map(pb, st, map, lm_tracks);
    for (int j = 1; j < pb->n_frames; ++j) { // stored deltas are reused as they are (bundle_adjustor.cpp:416-450)
        if (!pb->preint_valid || !pb->preint_valid[j]) continue;
        Frame *f = map.get_frame(j);
        f->has_preintegration_factor = true;
        auto &d = f->preintegration.delta;
        const double *pd = pb->preint_delta + 11 * j;
        d.t = pd[0];
        std::memcpy(d.q.c, pd + 1, 32);
        for (int k = 0; k < 3; ++k) d.p[k] = pd[5 + k], d.v[k] = pd[8 + k];
        std::memcpy(d.sqrt_inv_cov, pb->preint_sqrt_inv_cov + 225 * j, sizeof d.sqrt_inv_cov);
        auto &jc = f->preintegration.jacobian;
        const double *pj = pb->preint_jacobian + 45 * j;
        std::memcpy(jc.dq_dbg, pj, 72), std::memcpy(jc.dp_dbg, pj + 9, 72), std::memcpy(jc.dp_dba, pj + 18, 72), std::memcpy(jc.dv_dbg, pj + 27, 72), std::memcpy(jc.dv_dba, pj + 36, 72);
    }
    BundleAdjustor().marginalize_frame(&map, (size_t)victim);
    MarginalizationPrior *pr = map.get_marginalization_factor();
    if (!pr || (int)pr->frames.size() != pb->n_frames - 1) return 1;
    std::memcpy(S, pr->sqrt_infomat.data(), pr->sqrt_infomat.size() * sizeof(double));
    std::memcpy(s, pr->sqrt_infovec.data(), pr->sqrt_infovec.size() * sizeof(double));
    return 0;
}

} // extern "C"

// ---- visual_inertial_pnp through the Map object graph: the window's last frame is taken out of the map (the reference
// solves it before put_frame) and refined against the rest ------------------------------------------------------------
#include "../../pvio_amd/host/pnp.h"

namespace pvio {
double flatten_seconds(Map *map, bool use_inertial, int reps);
}
extern "C" double host_flatten_seconds(const pvio_ba_problem *pb, const pvio_ba_state *st, int32_t use_inertial, int32_t reps) {
    Map map;
    std::vector<Track *> lm_tracks;
    build_map(pb, st, map, lm_tracks);
    return pvio::flatten_seconds(&map, use_inertial != 0, reps);
}

extern "C" int host_roundtrip_pnp(const pvio_ba_problem *pb, pvio_ba_state *st, int32_t use_inertial, int32_t max_iter, double *state_out) {
    Map map;
    std::vector<Track *> lm_tracks;
    build_map(pb, st, map, lm_tracks);
    const int N = pb->n_frames;
    std::unique_ptr<Frame> frame = std::move(map.frames.back());
    map.frames.pop_back();
    if (use_inertial) {
        auto &d = frame->preintegration.delta;
        const double *dl = pb->preint_delta + 11 * (N - 1);
        d.t = dl[0];
        std::memcpy(d.q.c, dl + 1, 32);
        for (int k = 0; k < 3; ++k) d.p[k] = dl[5 + k], d.v[k] = dl[8 + k];
        std::memcpy(d.sqrt_inv_cov, pb->preint_sqrt_inv_cov + 225 * (size_t)(N - 1), sizeof d.sqrt_inv_cov);
        auto &jc = frame->preintegration.jacobian;
        const double *j = pb->preint_jacobian + 45 * (size_t)(N - 1);
        std::memcpy(jc.dq_dbg, j, 72), std::memcpy(jc.dp_dbg, j + 9, 72), std::memcpy(jc.dp_dba, j + 18, 72), std::memcpy(jc.dv_dbg, j + 27, 72), std::memcpy(jc.dv_dba, j + 36, 72);
    }
    Cfg cfg;
    cfg.iters = (size_t)max_iter, cfg.plane_cov = 1e-4;
    visual_inertial_pnp(&map, frame.get(), &cfg, use_inertial != 0);
    std::memcpy(state_out, frame->pose.q.c, 32);
    for (int k = 0; k < 3; ++k) state_out[4 + k] = frame->pose.p[k], state_out[7 + k] = frame->motion.v[k], state_out[10 + k] = frame->motion.bg[k], state_out[13 + k] = frame->motion.ba[k];
    return 0;
}

// flat entry: world-point factors as well (PoseOnlyReprojectionXYZErrorCost)
extern "C" int host_pnp_flat(const double *cam, const double *imu, const double *W, int32_t n, const double *anchor_states, const double *anchor_cams,
                             const double *z_ref, const double *z_tgt, const double *rho, int32_t n_pts, const double *points, const double *z_pts,
                             int32_t use_inertial, const double *last_state, const double *last_imu, const double *delta, const double *U, const double *jac,
                             int32_t max_iter, double *state16, int32_t *iterations, int32_t *termination, double *costs2) {
    PnpProblem pb;
    std::memcpy(pb.cam, cam, 56), std::memcpy(pb.imu, imu, 56), std::memcpy(pb.sqrt_inv_cov, W, 32);
    for (int k = 0; k < n; ++k) {
        PnpFactor f;
        std::memcpy(f.anchor_state, anchor_states + 16 * k, 128), std::memcpy(f.anchor_cam, anchor_cams + 7 * k, 56);
        f.z_ref[0] = z_ref[2 * k], f.z_ref[1] = z_ref[2 * k + 1], f.z_tgt[0] = z_tgt[2 * k], f.z_tgt[1] = z_tgt[2 * k + 1], f.inv_depth = rho[k];
        pb.factors.push_back(f);
    }
    for (int k = 0; k < n_pts; ++k) {
        PnpPointFactor f;
        std::memcpy(f.point, points + 3 * k, 24);
        f.z_tgt[0] = z_pts[2 * k], f.z_tgt[1] = z_pts[2 * k + 1];
        pb.point_factors.push_back(f);
    }
    pb.use_inertial = use_inertial != 0;
    if (use_inertial) {
        std::memcpy(pb.last_state, last_state, 128), std::memcpy(pb.last_imu, last_imu, 56);
        std::memcpy(pb.delta, delta, 88), std::memcpy(pb.sqrt_inv_cov_imu, U, 1800), std::memcpy(pb.jac, jac, 360);
    }
    const dense::Summary s = solve_pnp(pb, state16, max_iter);
    *iterations = s.iterations, *termination = s.termination;
    costs2[0] = s.initial_cost, costs2[1] = s.final_cost;
    return 0;
}
