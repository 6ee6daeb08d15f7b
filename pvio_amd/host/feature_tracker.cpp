// feature_tracker.cpp -- see feature_tracker.h.
#include "feature_tracker.h"

#include <cstdio>

namespace pvio {

HostFeatureTracker::HostFeatureTracker(std::shared_ptr<Config> config) : map(std::make_unique<Map>()), config(std::move(config)) {}
HostFeatureTracker::~HostFeatureTracker() = default;

// The reference's forensics timer of FeatureTracker::work (core/feature_tracker.cpp:38-46: the running average behind pvio-pc's "FT Time"
// graph, main.cpp:165-167): the reference's own `make_timer` / `critical_forensics` inside the PVIO tree, nothing in the standalone build.
#ifdef PVIO_HOST_USE_REFERENCE_TYPES
#define PVIO_HOST_FT_TIMER()                                                 \
    auto ft_timer = make_timer([](double t) {                                \
        critical_forensics(feature_tracker_time, time) {                     \
            static double avg_time = 0;                                      \
            static double avg_count = 0;                                     \
            avg_time = (avg_time * avg_count + t) / (avg_count + 1);         \
            avg_count += 1.0;                                                \
            time = avg_time;                                                 \
        }                                                                    \
    })
#else
#define PVIO_HOST_FT_TIMER() ((void)0)
#endif

void HostFeatureTracker::track_frame(std::unique_ptr<Frame> frame) {
    PVIO_HOST_FT_TIMER();
    frame->image->preprocess(); // HipImage: upload (+ undistortion), CLAHE, pyramid, Scharr on the device

    size_t optimized_id = nil();
    PoseState optimized_pose;
    MotionState optimized_motion;
    if (latest_optimized_state) std::tie(optimized_id, optimized_pose, optimized_motion) = latest_optimized_state();
    const bool is_initialized = optimized_id != nil();

    if (map->frame_num() > 0) {
        if (is_initialized) { // feature_tracker.cpp:58-79
            const size_t k = map->frame_index_by_id(optimized_id);
            if (k != nil()) {
                Frame *opt = map->get_frame(k);
                opt->pose = optimized_pose, opt->motion = optimized_motion;
                for (size_t j = k + 1; j < map->frame_num(); ++j) {
                    Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
                    fj->preintegration.integrate(fj->image->t, fi->motion.bg, fi->motion.ba, false, false);
                    fj->preintegration.predict(fi, fj);
                }
            } else {
                std::fprintf(stderr, "[pvio-hip] feature tracker: the optimized frame has slid out of the tracking map\n");
                latest_state.reset();
            }
        }
        Frame *last = map->last_frame();
        if (!last->preintegration.data.empty()) { // :81-87: the interval starts at the last image time
            if (frame->preintegration.data.empty() || (frame->preintegration.data.front().t - last->image->t > 1.0e-5)) {
                ImuData imu = last->preintegration.data.back();
                imu.t = last->image->t;
                frame->preintegration.data.insert(frame->preintegration.data.begin(), imu);
            }
        }
        frame->preintegration.integrate(frame->image->t, last->motion.bg, last->motion.ba, false, false);
        if (is_initialized) {
            frame->preintegration.predict(last, frame.get());
            latest_state = std::make_tuple(frame->image->t, frame->pose, frame->motion);
        }
        last->track_keypoints(frame.get(), config.get()); // gyro prediction, device LK, RANSAC, Poisson-disk survivors
    }

    frame->detect_keypoints(config.get()); // device Harris corners where there is room
    map->put_frame(std::move(frame));

    const size_t keep = is_initialized ? config->feature_tracker_max_frames() : config->feature_tracker_max_init_frames();
    while (map->frame_num() > keep) map->erase_frame(0);

    if (issue_frame) issue_frame(map->last_frame());
}

} // namespace pvio
