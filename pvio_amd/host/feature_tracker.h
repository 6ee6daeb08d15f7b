// feature_tracker.h -- the FeatureTracker orchestration above the pvio::Image seam.
//
// Reference: PVIO::Core::FeatureTracker (pvio/src/pvio/core/feature_tracker.h:27-50, feature_tracker.cpp:37-142) -- a class
// private to the reference's core that owns the feature-tracking Map and, per camera frame, runs
//   image->preprocess()  ->  catch up with the back end's latest state (re-integrate + predict the frames after it)  ->
//   carry the last IMU sample over, integrate, predict  ->  last_frame->track_keypoints(frame)  ->  frame->detect_keypoints()
//   ->  map->put_frame(frame)  ->  drop the oldest frames (feature_tracker_max_[init_]frames).
// Inside the PVIO tree that code stays the reference's own and reaches the GPU through pvio::Image (HipImage,
// feature_front.h).  This class is the same sequence for programs that do not link the reference's core (the headless
// driver, tests): same method names, same order of operations, no Worker thread (the reference's default build has
// PVIO_ENABLE_THREADING off and runs work() inline, utility/worker.h:58-65), the core's two callbacks as plain members.
#pragma once
#include <deque>
#include <functional>
#include <memory>
#include <optional>
#include <tuple>

#include "host_seam.h"

namespace pvio {

class HostFeatureTracker {
  public:
    explicit HostFeatureTracker(std::shared_ptr<Config> config);
    ~HostFeatureTracker();

    // FeatureTracker::track_frame + work(): the frame carries image, K, extrinsics, noise and its IMU samples (core.cpp:127-158)
    void track_frame(std::unique_ptr<Frame> frame);

    // what core->frontend->get_latest_state() hands to work(): (frame id, pose, motion) of the newest optimized frame, or
    // nil() while the system is not initialized
    std::function<std::tuple<size_t, PoseState, MotionState>()> latest_optimized_state;
    // what work() ends with: core->frontend->issue_frame(map->last_frame())
    std::function<void(Frame *)> issue_frame;

    std::optional<std::tuple<double, PoseState, MotionState>> get_latest_state() const { return latest_state; }

    std::unique_ptr<Map> map;

  private:
    std::shared_ptr<Config> config;
    std::optional<std::tuple<double, PoseState, MotionState>> latest_state;
};

} // namespace pvio
