// gt_initializer.cpp -- TEST INFRASTRUCTURE (oracle/ref/Makefile): the ONE piece of the reference's pipeline that is not the reference's.
//
// libpvio_ref.so / libpvio_dropin*.so compile the reference's whole control plane unedited -- pvio.cpp, core/{core,feature_tracker,
// frontend_worker,sliding_window_tracker,plane_extractor}.cpp, map/*.cpp -- so that pvio::PVIO itself can be driven over a sequence
// (seq_capi.cpp).  What they do NOT compile is core/initializer.cpp: its SfM + IMU alignment (essential / homography RANSAC with
// Eigen::EigenSolver, fullPivHouseholderQr; SURVEY.md section 2: out of scope) is beyond the mini-Eigen, and it is not on the hot path.
// This file defines the class the reference DECLARES (core/initializer.h:29-58, included unedited) with a bootstrap from supplied body
// poses instead: the keyframes are picked and mirrored the way FrontendWorker expects (window of sliding_window_size() frames,
// initializer_keyframe_gap() apart, tracks re-created on the copies, IMU samples concatenated), poses / velocities come from
// seq_bootstrap(), biases start at zero, tracks are triangulated by the reference's Track::triangulate, and the hand-over is the
// reference's own: frame 0 fixed, BundleAdjustor().solve(map, config, true), every frame a keyframe, SlidingWindowTracker(map, config)
// (core/initializer.cpp:86-100).  After that nothing reads the supplied poses.
#include <pvio/common.h>
#include <pvio/core/initializer.h>
#include <pvio/core/sliding_window_tracker.h>
#include <pvio/estimation/bundle_adjustor.h>
#include <pvio/map/frame.h>
#include <pvio/map/map.h>
#include <pvio/map/track.h>

#include "seq_bootstrap.h"

#include <algorithm>
#include <cmath>

namespace pvio {

std::vector<SeqTimedPose> &seq_bootstrap() {
    static std::vector<SeqTimedPose> poses;
    return poses;
}

namespace {
quaternion slerp(const quaternion &a, quaternion b, double u) {
    double d = a.x() * b.x() + a.y() * b.y() + a.z() * b.z() + a.w() * b.w();
    if (d < 0) b = quaternion(-b.w(), -b.x(), -b.y(), -b.z()), d = -d;
    const double th = std::acos(std::min(1.0, d));
    if (th < 1.0e-9) return a;
    const double sa = std::sin((1.0 - u) * th) / std::sin(th), sb = std::sin(u * th) / std::sin(th);
    quaternion r(sa * a.w() + sb * b.w(), sa * a.x() + sb * b.x(), sa * a.y() + sb * b.y(), sa * a.z() + sb * b.z());
    r.normalize();
    return r;
}
bool pose_at(double t, PoseState &out, vector<3> &velocity) {
    const std::vector<SeqTimedPose> &B = seq_bootstrap();
    if (B.size() < 2 || t < B.front().t || t > B.back().t) return false;
    size_t k = 1;
    while (k + 1 < B.size() && B[k].t < t) ++k;
    const SeqTimedPose &a = B[k - 1], &b = B[k];
    const double u = (t - a.t) / (b.t - a.t);
    out.p = a.p + u * (b.p - a.p);
    out.q = slerp(a.q, b.q, u);
    velocity = (b.p - a.p) / (b.t - a.t);
    return true;
}
} // namespace

Initializer::Initializer(std::shared_ptr<Config> config) : config(config) {}
Initializer::~Initializer() = default;

void Initializer::mirror_keyframe_map(Map *ft, size_t init_frame_id) {
    map.reset();
    const size_t last = ft->frame_index_by_id(init_frame_id), gap = config->initializer_keyframe_gap(), W = config->sliding_window_size();
    if (last == nil() || last < gap * (W - 1)) return;
    std::vector<size_t> idx;
    for (size_t i = 0; i < W; ++i) idx.push_back(last - gap * (W - 1) + i * gap);
    auto m = std::make_unique<Map>();
    for (size_t i : idx) m->put_frame(ft->get_frame(i)->clone());
    for (size_t j = 1; j < m->frame_num(); ++j) {
        Frame *oi = ft->get_frame(idx[j - 1]), *oj = ft->get_frame(idx[j]), *ni = m->get_frame(j - 1), *nj = m->get_frame(j);
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
    map = std::move(m);
}

std::unique_ptr<SlidingWindowTracker> Initializer::initialize() {
    if (!map) return nullptr;
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *f = map->get_frame(i);
        vector<3> v;
        if (!pose_at(f->image->t, f->pose, v)) return nullptr; // no supplied pose for this time (yet)
        f->motion.v = v, f->motion.bg.setZero(), f->motion.ba.setZero();
    }
    size_t valid = 0;
    for (size_t i = 0; i < map->track_num(); ++i)
        if (map->get_track(i)->keypoint_num() >= 2 && map->get_track(i)->triangulate()) ++valid;
    if (valid < 20) return nullptr;
    map->get_frame(0)->flag(FrameFlag::FF_FIX_POSE) = true;
    BundleAdjustor().solve(map.get(), config.get(), true);
    for (size_t i = 0; i < map->frame_num(); ++i) map->get_frame(i)->flag(FrameFlag::FF_KEYFRAME) = true;
    return std::make_unique<SlidingWindowTracker>(std::move(map), config);
}

} // namespace pvio
