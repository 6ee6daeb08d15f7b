// headless.cpp -- see headless.h.
#include "headless.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace pvio {

namespace {
quaternion expmap(const vector<3> &w) { // geometry/lie_algebra.h:32-35
    const double th = w.norm();
    if (th == 0.0) return quaternion::Identity();
    const double s = std::sin(0.5 * th) / th;
    return quaternion(std::cos(0.5 * th), s * w[0], s * w[1], s * w[2]);
}
quaternion slerp(const quaternion &a, quaternion b, double u) {
    double d = a.x() * b.x() + a.y() * b.y() + a.z() * b.z() + a.w() * b.w();
    if (d < 0) b = quaternion(-b.w(), -b.x(), -b.y(), -b.z()), d = -d;
    const double th = std::acos(std::min(1.0, d));
    if (th < 1e-9) return a;
    const double sa = std::sin((1 - u) * th) / std::sin(th), sb = std::sin(u * th) / std::sin(th);
    return quaternion(sa * a.w() + sb * b.w(), sa * a.x() + sb * b.x(), sa * a.y() + sb * b.y(), sa * a.z() + sb * b.z()).normalized();
}
} // namespace

HeadlessVio::HeadlessVio(std::shared_ptr<HeadlessConfig> cfg) : config(std::move(cfg)) {
    latest_state = std::make_tuple(nil(), PoseState(), MotionState());
    feature_tracker = std::make_unique<HostFeatureTracker>(config);
    feature_tracker->latest_optimized_state = [this] { return latest_state; };
    feature_tracker->issue_frame = [this](Frame *f) { frontend_work(f->id()); };
}
HeadlessVio::~HeadlessVio() = default;

// ---- core/core.cpp:59-107: one ImuData per accelerometer sample, the gyroscope interpolated to its time ---------------
OutputPose HeadlessVio::track_gyroscope(const double &t, const double &x, const double &y, const double &z) {
    if (!accelerometers.empty()) {
        if (t < accelerometers.front().t) {
            gyroscopes.clear();
        } else {
            while (!accelerometers.empty() && t >= accelerometers.front().t) {
                const Acc &acc = accelerometers.front();
                const double lambda = (acc.t - gyroscopes[0].t) / (t - gyroscopes[0].t);
                const vector<3> w = gyroscopes[0].w + lambda * (vector<3>(x, y, z) - gyroscopes[0].w);
                track_imu(ImuData{acc.t, w, acc.a});
                accelerometers.pop_front();
            }
            if (!accelerometers.empty())
                while (!gyroscopes.empty() && gyroscopes.front().t < t) gyroscopes.pop_front();
        }
    }
    gyroscopes.push_back(Gyr{t, vector<3>(x, y, z)});
    return predict_pose(t);
}

OutputPose HeadlessVio::track_accelerometer(const double &t, const double &x, const double &y, const double &z) {
    if (!gyroscopes.empty() && t >= gyroscopes.front().t) {
        if (t > gyroscopes.back().t) {
            while (gyroscopes.size() > 1) gyroscopes.pop_front();
            accelerometers.push_back(Acc{t, vector<3>(x, y, z)});
        } else if (t == gyroscopes.back().t) {
            while (gyroscopes.size() > 1) gyroscopes.pop_front();
            track_imu(ImuData{t, gyroscopes.front().w, vector<3>(x, y, z)});
        } else {
            while (t >= gyroscopes[1].t) gyroscopes.pop_front();
            const double lambda = (t - gyroscopes[0].t) / (gyroscopes[1].t - gyroscopes[0].t);
            const vector<3> w = gyroscopes[0].w + lambda * (gyroscopes[1].w - gyroscopes[0].w);
            track_imu(ImuData{t, w, vector<3>(x, y, z)});
        }
    }
    return predict_pose(t);
}

OutputPose HeadlessVio::track_camera(std::shared_ptr<Image> image) { // core.cpp:109-125
    auto f = std::make_unique<Frame>();
    f->K = config->camera_intrinsic();
    f->image = image;
    f->sqrt_inv_cov.setZero();
    f->sqrt_inv_cov(0, 0) = f->K(0, 0) / std::sqrt(config->keypoint_noise_cov()(0, 0));
    f->sqrt_inv_cov(1, 1) = f->K(1, 1) / std::sqrt(config->keypoint_noise_cov()(1, 1));
    f->camera.q_cs = config->camera_to_body_rotation(), f->camera.p_cs = config->camera_to_body_translation();
    f->imu.q_cs = config->imu_to_body_rotation(), f->imu.p_cs = config->imu_to_body_translation();
    f->preintegration.cov_a = config->accelerometer_noise_cov(), f->preintegration.cov_w = config->gyroscope_noise_cov();
    f->preintegration.cov_ba = config->accelerometer_bias_noise_cov(), f->preintegration.cov_bg = config->gyroscope_bias_noise_cov();
    const double t = image->t;
    frames.emplace_back(std::move(f));
    return predict_pose(t);
}

void HeadlessVio::track_imu(const ImuData &imu) { // core.cpp:127-140: a frame is released once an IMU sample lies behind it
    frontal_imus.push_back(imu), imus.push_back(imu);
    while (!imus.empty() && !frames.empty()) {
        if (imus.front().t <= frames.front()->image->t) {
            frames.front()->preintegration.data.push_back(imus.front());
            imus.pop_front();
        } else {
            feature_tracker->track_frame(std::move(frames.front()));
            frames.pop_front();
        }
    }
}

OutputPose HeadlessVio::predict_pose(const double &t) { // core.cpp:32-40,142-163
    OutputPose out;
    out.q = quaternion(0, 0, 0, 0), out.p = vector<3>::Zero(); // "invalid": all-zero quaternion
    auto st = feature_tracker->get_latest_state();
    if (!st) return out;
    double time = std::get<0>(*st);
    PoseState pose = std::get<1>(*st);
    MotionState motion = std::get<2>(*st);
    while (!frontal_imus.empty() && frontal_imus.front().t <= time) frontal_imus.pop_front();
    const vector<3> gravity(0, 0, -9.80665);
    for (const ImuData &imu : frontal_imus)
        if (imu.t <= t) {
            const double dt = imu.t - time;
            const vector<3> acc = gravity + pose.q * (imu.a - motion.ba);
            pose.p = pose.p + dt * motion.v + 0.5 * dt * dt * acc;
            motion.v = motion.v + dt * acc;
            pose.q = (pose.q * expmap((imu.w - motion.bg) * dt)).normalized();
            time = imu.t;
        }
    out.q = pose.q * config->output_to_body_rotation();
    out.p = pose.p + pose.q * config->output_to_body_translation();
    return out;
}

// ---- core/frontend_worker.cpp:43-79 -----------------------------------------------------------------------------------
void HeadlessVio::frontend_work(size_t frame_id) {
    if (!window_map) {
        if (bootstrap_window(frame_id)) {
            const Frame *last = window_map->last_frame();
            latest_state = std::make_tuple(frame_id, last->pose, last->motion);
        }
        return;
    }
    mirror_frame(frame_id);
    if (frame && track()) {
        const Frame *last = window_map->last_frame();
        latest_state = std::make_tuple(frame_id, last->pose, last->motion);
    }
}

bool HeadlessVio::pose_at(double t, PoseState &out, vector<3> &velocity) const {
    if (bootstrap.size() < 2 || t < bootstrap.front().t || t > bootstrap.back().t) return false;
    size_t k = 1;
    while (k + 1 < bootstrap.size() && bootstrap[k].t < t) ++k;
    const TimedPose &a = bootstrap[k - 1], &b = bootstrap[k];
    const double u = (t - a.t) / (b.t - a.t);
    out.p = a.pose.p + u * (b.pose.p - a.pose.p);
    out.q = slerp(a.pose.q, b.pose.q, u);
    velocity = (b.pose.p - a.pose.p) / (b.t - a.t);
    return true;
}

// In place of Initializer::mirror_keyframe_map + initialize (core/initializer.cpp:40-100): same keyframe choice, same
// track mirroring and IMU concatenation; the SfM / IMU alignment is replaced by the supplied trajectory.
bool HeadlessVio::bootstrap_window(size_t frame_id) {
    Map *ft = feature_tracker->map.get();
    const size_t last = ft->frame_index_by_id(frame_id), gap = config->keyframe_gap, W = config->sliding_window_size();
    if (last == nil() || last < gap * (W - 1)) return false;
    std::vector<size_t> idx;
    for (size_t i = 0; i < W; ++i) idx.push_back(last - gap * (W - 1) + i * gap);
    auto map = std::make_unique<Map>();
    for (size_t i : idx) {
        auto f = ft->get_frame(i)->clone();
        vector<3> v;
        if (!pose_at(f->image->t, f->pose, v)) return false; // no ground truth for this time (yet)
        f->motion.v = v, f->motion.bg.setZero(), f->motion.ba.setZero();
        map->put_frame(std::move(f));
    }
    for (size_t j = 1; j < map->frame_num(); ++j) {
        Frame *oi = ft->get_frame(idx[j - 1]), *oj = ft->get_frame(idx[j]), *ni = map->get_frame(j - 1), *nj = map->get_frame(j);
        for (size_t ki = 0; ki < oi->keypoint_num(); ++ki)
            if (Track *t = oi->get_track(ki)) {
                const size_t kj = t->get_keypoint_index(oj);
                if (kj != nil()) ni->get_track(ki, create_if_empty)->add_keypoint(nj, kj);
            }
        nj->preintegration.data.clear();
        for (size_t f = idx[j - 1]; f < idx[j]; ++f) {
            const std::vector<ImuData> &d = ft->get_frame(f + 1)->preintegration.data;
            nj->preintegration.data.insert(nj->preintegration.data.end(), d.begin(), d.end());
        }
    }
    size_t valid = 0;
    for (size_t i = 0; i < map->track_num(); ++i)
        if (map->get_track(i)->keypoint_num() >= 2 && map->get_track(i)->triangulate()) ++valid;
    if (valid < 20) return false; // initializer_min_landmarks territory: wait for a better window
    map->get_frame(0)->flag(FrameFlag::FF_FIX_POSE) = true; // initializer.cpp:91-92
    BundleAdjustor().solve(map.get(), config.get(), true);
    ++solves;
    for (size_t i = 0; i < map->frame_num(); ++i) map->get_frame(i)->flag(FrameFlag::FF_KEYFRAME) = true;
    // SlidingWindowTracker's constructor re-integrates every interval at the solved biases (sliding_window_tracker.cpp:36-41)
    for (size_t j = 1; j < map->frame_num(); ++j) {
        Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
        fj->preintegration.integrate(fj->image->t, fi->motion.bg, fi->motion.ba, true, true);
    }
    window_map = std::move(map);
    skipped_frames = 0;
    return true;
}

void HeadlessVio::mirror_frame(size_t frame_id) { // sliding_window_tracker.cpp:52-74
    Map *ft = feature_tracker->map.get();
    Frame *new_i = window_map->last_frame();
    const size_t ii = ft->frame_index_by_id(new_i->id()), jj = ft->frame_index_by_id(frame_id);
    frame.reset();
    if (ii == nil() || jj == nil()) return;
    Frame *old_i = ft->get_frame(ii), *old_j = ft->get_frame(jj);
    frame = old_j->clone();
    for (size_t ki = 0; ki < old_i->keypoint_num(); ++ki)
        if (Track *t = old_i->get_track(ki)) {
            const size_t kj = t->get_keypoint_index(old_j);
            if (kj != nil()) new_i->get_track(ki, create_if_empty)->add_keypoint(frame.get(), kj);
        }
}

bool HeadlessVio::track() { // sliding_window_tracker.cpp:76-131
    Map *map = window_map.get();
    Frame *last = map->last_frame();
    frame->preintegration.integrate(frame->image->t, last->motion.bg, last->motion.ba, true, true);
    frame->preintegration.predict(last, frame.get());
    visual_inertial_pnp(map, frame.get(), config.get(), true);
    keyframe_check(frame.get());
    for (size_t i = 0; i < frame->keypoint_num(); ++i) {
        Track *t = frame->get_track(i);
        if (!t || t->flag(TrackFlag::TF_VALID)) continue;
        t->triangulate();
    }
    const bool last_is_keyframe = last->flag(FrameFlag::FF_KEYFRAME);
    if (last_is_keyframe) {
        while (map->frame_num() >= config->sliding_window_size() + 1) map->marginalize_frame(0);
        map->put_frame(std::move(frame));
        if (!map->get_marginalization_factor()) { // the gauge "prior": information handed over as sqrt-information (App. D item 2)
            std::vector<Frame *> init_frames;
            for (size_t i = 1; i < map->frame_num(); ++i) init_frames.push_back(map->get_frame(i - 1));
            const int D = ES_SIZE * (int)(map->frame_num() - 1);
            matrix<> S;
            vector<> s;
            S.resize(D, D), s.resize(D);
            S.setZero(), s.setZero();
            for (int k = 0; k < 3; ++k) S(ES_P + k, ES_P + k) = 1.0e15, S(ES_Q + k, ES_Q + k) = 1.0e15;
            map->set_marginalization_factor(Factor::create_marginalization_error(S, s, std::move(init_frames)));
        }
        BundleAdjustor().solve(map, config.get(), true);
        ++solves;
    } else { // the last frame was not a keyframe: it is replaced, its IMU interval is merged into the new frame's
        const std::vector<ImuData> &data = last->preintegration.data;
        frame->preintegration.data.insert(frame->preintegration.data.begin(), data.begin(), data.end());
        frame->preintegration.integrate(frame->image->t, last->motion.bg, last->motion.ba, true, true);
        map->erase_frame(map->frame_num() - 1);
        map->put_frame(std::move(frame));
    }
    map->prune_tracks([](const Track *t) {
        return (!t->flag(TrackFlag::TF_VALID) || t->landmark.quality > 3.0) && (!t->flag(TrackFlag::TF_PLANE) || t->landmark.quality > 3.0);
    });
    return true;
}

void HeadlessVio::keyframe_check(Frame *fj) { // sliding_window_tracker.cpp:258-296
    Map *map = window_map.get();
    Frame *fi = nullptr;
    for (size_t i = 0; i < map->frame_num(); ++i)
        if (map->get_frame(map->frame_num() - i - 1)->flag(FrameFlag::FF_KEYFRAME)) {
            fi = map->get_frame(map->frame_num() - i - 1);
            break;
        }
    if (!fi) {
        fj->flag(FrameFlag::FF_KEYFRAME) = true;
    } else {
        const quaternion qij = (fi->camera.q_cs.conjugate() * fi->imu.q_cs * fj->preintegration.delta.q * fj->imu.q_cs.conjugate() * fj->camera.q_cs).conjugate();
        std::vector<double> parallax;
        for (size_t kj = 0; kj < fj->keypoint_num(); ++kj) {
            Track *t = fj->get_track(kj);
            if (!t) continue;
            const size_t ki = t->get_keypoint_index(fi);
            if (ki == nil()) continue;
            const vector<3> r = qij * fi->get_keypoint(ki).homogeneous();
            const double ux = r[0] / r[2] * fi->K(0, 0) + fi->K(0, 2), uy = r[1] / r[2] * fi->K(1, 1) + fi->K(1, 2);
            const vector<2> &z = fj->get_keypoint(kj);
            const double vx = z[0] * fj->K(0, 0) + fj->K(0, 2), vy = z[1] * fj->K(1, 1) + fj->K(1, 2);
            parallax.push_back(std::sqrt((ux - vx) * (ux - vx) + (uy - vy) * (uy - vy)));
        }
        if (parallax.size() < 50) {
            fj->flag(FrameFlag::FF_KEYFRAME) = true;
        } else {
            std::sort(parallax.begin(), parallax.end());
            if (parallax[parallax.size() * 4 / 5] > 50) fj->flag(FrameFlag::FF_KEYFRAME) = true;
            else skipped_frames++;
        }
    }
    if (skipped_frames > 10) fj->flag(FrameFlag::FF_KEYFRAME) = true;
    if (fj->flag(FrameFlag::FF_KEYFRAME)) skipped_frames = 0;
}

} // namespace pvio
