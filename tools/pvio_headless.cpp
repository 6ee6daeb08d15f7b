// pvio_headless -- the sequence loop of pvio-pc (pvio-pc/src/main.cpp:207-258) without GUI, OpenCV, Ceres or yaml-cpp:
//   DatasetReader::next() -> read_gyroscope / read_accelerometer / read_image -> HeadlessVio::track_* -> trajectory.tum
// Usage: pvio_headless <euroc://DIR | tum://DIR> <ground_truth.tum> [trajectory.tum] [max_frames] [window] [keyframe_gap]
//   window / keyframe_gap: sliding_window_size and initializer_keyframe_gap (config yaml: 8 and 5); short test sequences pass smaller ones
//   ground_truth.tum  "t px py pz qx qy qz qw" lines (body poses): used ONLY to bootstrap the first window, in place of the
//                     reference's SfM initializer (see tests/host/standin/headless.h)
// Camera / IMU constants are those of config/euroc.yaml and config/tum-vi.yaml, chosen by the URI scheme.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "../pvio_amd/host/dataset_reader.h"
#include "../tests/host/standin/headless.h"

using namespace pvio;

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <euroc://DIR|tum://DIR> <ground_truth.tum> [trajectory.tum] [max_frames] [window] [keyframe_gap]\n", argv[0]);
        return 2;
    }
    const std::string uri = argv[1], out_path = argc > 3 ? argv[3] : "trajectory.tum";
    const long max_frames = argc > 4 ? std::atol(argv[4]) : -1;
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof opts);
    opts.world_size = 1, opts.use_graph = 1;
    if (pvio_hip_create(&opts, &ctx) != PVIO_OK) {
        std::fprintf(stderr, "no usable GPU\n"); // there is no CPU path
        return 1;
    }
    int rc = 0;
    try {
        auto reader = DatasetReader::create_reader(uri, ctx);
        if (!reader) throw std::runtime_error("unknown dataset scheme: " + uri);
        auto config = uri.rfind("euroc://", 0) == 0 ? HeadlessConfig::euroc() : HeadlessConfig::tum_vi();
        if (argc > 5 && std::atol(argv[5]) >= 2) config->window = (size_t)std::atol(argv[5]);
        if (argc > 6 && std::atol(argv[6]) >= 1) config->keyframe_gap = (size_t)std::atol(argv[6]);
        HeadlessVio vio(config);
        {
            std::ifstream gt(argv[2]);
            std::vector<TimedPose> poses;
            std::string line;
            while (std::getline(gt, line)) {
                if (line.empty() || line[0] == '#') continue;
                std::istringstream ss(line);
                TimedPose tp;
                double q[4];
                if (!(ss >> tp.t >> tp.pose.p[0] >> tp.pose.p[1] >> tp.pose.p[2] >> q[0] >> q[1] >> q[2] >> q[3])) continue;
                tp.pose.q = quaternion(q[3], q[0], q[1], q[2]);
                poses.push_back(tp);
            }
            if (poses.size() < 2) throw std::runtime_error("ground truth file holds fewer than two poses");
            vio.set_bootstrap_trajectory(std::move(poses));
        }
        TumOutputWriter writer(out_path);
        bool has_gyr = false, has_acc = false;
        long n_frames = 0, n_poses = 0;
        for (;;) {
            DatasetReader::NextDataType type;
            while ((type = reader->next()) == DatasetReader::AGAIN) {}
            if (type == DatasetReader::END) break;
            if (type == DatasetReader::GYROSCOPE) {
                auto tw = reader->read_gyroscope();
                vio.track_gyroscope(tw.first, tw.second[0], tw.second[1], tw.second[2]);
                has_gyr = true;
            } else if (type == DatasetReader::ACCELEROMETER) {
                auto ta = reader->read_accelerometer();
                vio.track_accelerometer(ta.first, ta.second[0], ta.second[1], ta.second[2]);
                has_acc = true;
            } else {
                auto image = reader->read_image();
                if (has_acc && has_gyr) {
                    const OutputPose pose = vio.track_camera(image);
                    const bool zero = pose.q.x() == 0 && pose.q.y() == 0 && pose.q.z() == 0 && pose.q.w() == 0; // main.cpp:231
                    if (!zero) writer.write_pose(image->t, pose), ++n_poses;
                }
                if (max_frames >= 0 && ++n_frames >= max_frames) break;
            }
        }
        std::fprintf(stderr, "%ld poses written to %s (%zu keyframe solves, window %zu frames)\n", n_poses, out_path.c_str(), vio.keyframe_solves(), vio.window_frames());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pvio_headless: %s\n", e.what());
        rc = 1;
    }
    pvio_hip_destroy(ctx);
    return rc;
}
