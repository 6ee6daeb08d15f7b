// TEST SCAFFOLDING (tests/host/standin/, moved out of the product directory in round 3): NOT part of what a PVIO maintainer links.
// Inside the PVIO tree the reference's own Core / SlidingWindowTracker / pvio-pc stay; this file exists so that the hot path can be
// driven over a whole sequence in this repository, where the reference cannot be built.  The IMU pairing and the tracker steps
// below restate reference control-plane code (out of scope) and are kept only as labelled scaffolding.
//
// headless.h -- a headless stand-in for pvio::PVIO so that sequences can be run end to end --
// readers -> front end (GPU) -> PnP -> sliding-window BA (GPU) -> trajectory.tum -- without the reference's core, GUI,
// OpenCV, Ceres or yaml-cpp.
//
// Same entry points as pvio::PVIO (pvio/include/pvio/pvio.h:135-148): track_gyroscope / track_accelerometer / track_camera.
// What is behind them follows the reference step by step, each function citing the lines it restates:
//   IMU pairing                core/core.cpp:59-107,127-140 (gyroscope samples interpolated to accelerometer times)
//   frame setup                core/core.cpp:109-125
//   feature tracker            HostFeatureTracker (feature_tracker.h)
//   sliding-window tracker     core/sliding_window_tracker.cpp:52-131 (mirror_frame, track) and :258-296 (keyframe_check);
//                              no plane extractor (core/plane_extractor.cpp is outside the hot path): the window never holds
//                              planes, the plane branches of BA / PnP stay idle
// and ONE piece is different on purpose: the reference bootstraps with an SfM + IMU-alignment initializer
// (core/initializer.cpp:86-381: essential / homography RANSAC, PnP chains, gravity refinement -- SURVEY section 2, out of
// scope).  Here the first window is bootstrapped from externally supplied body poses (a dataset's ground truth):
// keyframes picked like Initializer::mirror_keyframe_map (:40-84), poses / velocities from the supplied trajectory, biases
// zero, tracks triangulated, then the same first solve (frame 0 fixed, :91-92).  After that nothing reads the supplied poses.
#pragma once
#include <deque>
#include <memory>
#include <vector>

#include "dataset_config.h"
#include "feature_tracker.h"
#include "host_seam.h"

namespace pvio {

// the dataset constants are product code now (pvio_amd/host/dataset_config.h, round 5): the headless driver needs them whatever control plane it drives
using HeadlessConfig = DatasetConfig;

struct TimedPose { // body pose in the world frame at time t
    double t;
    PoseState pose;
};

class HeadlessVio {
  public:
    explicit HeadlessVio(std::shared_ptr<HeadlessConfig> config);
    ~HeadlessVio();
    void set_bootstrap_trajectory(std::vector<TimedPose> poses) { bootstrap = std::move(poses); }

    OutputPose track_gyroscope(const double &t, const double &x, const double &y, const double &z);
    OutputPose track_accelerometer(const double &t, const double &x, const double &y, const double &z);
    OutputPose track_camera(std::shared_ptr<Image> image);

    bool initialized() const { return window_map != nullptr; }
    size_t window_frames() const { return window_map ? window_map->frame_num() : 0; }
    size_t keyframe_solves() const { return solves; }
    const Map *window() const { return window_map.get(); }
    const Map *tracking_map() const { return feature_tracker ? feature_tracker->map.get() : nullptr; } // the feature tracker's own map (tests)

  private:
    struct Gyr {
        double t;
        vector<3> w;
    };
    struct Acc {
        double t;
        vector<3> a;
    };
    void track_imu(const ImuData &imu);
    OutputPose predict_pose(const double &t);
    void frontend_work(size_t frame_id);     // FrontendWorker::work
    bool bootstrap_window(size_t frame_id);  // in place of Initializer::initialize
    void mirror_frame(size_t frame_id);      // SlidingWindowTracker::mirror_frame
    bool track();                            // SlidingWindowTracker::track
    void keyframe_check(Frame *frame);       // SlidingWindowTracker::keyframe_check
    bool pose_at(double t, PoseState &out, vector<3> &velocity) const;

    std::shared_ptr<HeadlessConfig> config;
    std::unique_ptr<HostFeatureTracker> feature_tracker;
    std::deque<Gyr> gyroscopes;
    std::deque<Acc> accelerometers;
    std::deque<ImuData> imus, frontal_imus;
    std::deque<std::unique_ptr<Frame>> frames;
    std::vector<TimedPose> bootstrap;
    std::unique_ptr<Map> window_map;
    std::unique_ptr<Frame> frame; // the frame being localized
    std::tuple<size_t, PoseState, MotionState> latest_state;
    size_t skipped_frames = 0, solves = 0;
};

} // namespace pvio
