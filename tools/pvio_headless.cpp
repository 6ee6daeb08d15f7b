// pvio_headless -- the sequence loop of pvio-pc (pvio-pc/src/main.cpp:207-258) without GUI, OpenCV, Ceres or yaml-cpp:
//   DatasetReader::next() -> read_gyroscope / read_accelerometer / read_image -> PVIO::track_* -> trajectory.tum
// Usage: pvio_headless <euroc://DIR | tum://DIR> <ground_truth.tum | -> [trajectory.tum] [max_frames] [window] [keyframe_gap]
//   window / keyframe_gap: sliding_window_size and initializer_keyframe_gap (config yaml: 8 and 5); short test sequences pass smaller ones
// Camera / IMU constants are those of config/euroc.yaml and config/tum-vi.yaml (pvio_amd/host/dataset_config.h), chosen by the URI scheme.
//
// What it drives is `pvio::PVIO` through its public interface only (pvio/include/pvio/pvio.h:135-148: track_gyroscope, track_accelerometer,
// track_camera; pvio::Config; pvio::Image = the product's HipImage from the reader):
//   -DPVIO_HOST_USE_REFERENCE_TYPES   the reference's OWN pvio::PVIO -- inside the PVIO tree, or here against oracle/_ref's objects of the
//                                     reference's sources (make -C oracle/ref headless -> oracle/_ref/pvio_headless).  This is the binary a
//                                     PVIO maintainer ships: nothing under tests/ is compiled into it.
//   otherwise (tests/host/Makefile)   the stand-in control plane of tests/host/standin (HeadlessVio), for boxes where the reference's
//                                     sources are absent.  Same loop, same readers, same output.
// ground_truth.tum ("t px py pz qx qy qz qw" lines, body poses) is used ONLY where the SfM initializer is absent -- the stand-in, and
// oracle/_ref's gt_initializer.cpp (the reference's core/initializer.cpp needs SfM geometry beyond this container's Eigen miniature; SURVEY
// section 2: out of scope) -- to bootstrap the first window.  A build inside the real PVIO tree (-DPVIO_HEADLESS_NO_BOOTSTRAP, or "-" as the
// argument) runs the reference's own initializer and reads no ground truth.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../pvio_amd/host/dataset_config.h"
#include "../pvio_amd/host/dataset_reader.h"

#ifdef PVIO_HOST_USE_REFERENCE_TYPES
#include <pvio/pvio.h>
#ifndef PVIO_HEADLESS_NO_BOOTSTRAP
#include "seq_bootstrap.h" // oracle/ref: the poses gt_initializer.cpp takes the place of the SfM initializer with (test build of the reference only)
#endif
namespace {
typedef pvio::PVIO Vio;
void set_bootstrap(Vio &, const std::vector<std::vector<double>> &rows) {
#ifndef PVIO_HEADLESS_NO_BOOTSTRAP
    std::vector<pvio::SeqTimedPose> &B = pvio::seq_bootstrap();
    B.clear();
    for (const auto &r : rows) B.push_back(pvio::SeqTimedPose{r[0], pvio::quaternion(r[7], r[4], r[5], r[6]), pvio::vector<3>(r[1], r[2], r[3])});
#else
    (void)rows;
#endif
}
const char *kControlPlane = "the reference's pvio::PVIO";
} // namespace
#else
#include "../tests/host/standin/headless.h"
namespace {
typedef pvio::HeadlessVio Vio;
void set_bootstrap(Vio &vio, const std::vector<std::vector<double>> &rows) {
    std::vector<pvio::TimedPose> poses;
    for (const auto &r : rows) {
        pvio::TimedPose tp;
        tp.t = r[0];
        tp.pose.p = pvio::vector<3>(r[1], r[2], r[3]);
        tp.pose.q = pvio::quaternion(r[7], r[4], r[5], r[6]);
        poses.push_back(tp);
    }
    vio.set_bootstrap_trajectory(std::move(poses));
}
const char *kControlPlane = "stand-in control plane (tests/host/standin)";
} // namespace
#endif

using namespace pvio;

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <euroc://DIR|tum://DIR> <ground_truth.tum|-> [trajectory.tum] [max_frames] [window] [keyframe_gap]\n", argv[0]);
        return 2;
    }
    const std::string uri = argv[1], out_path = argc > 3 ? argv[3] : "trajectory.tum";
    const long max_frames = argc > 4 ? std::atol(argv[4]) : -1;
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof opts);
    opts.world_size = 1, opts.use_graph = 1;
    if (pvio_hip_abi_version() != PVIO_HIP_ABI_VERSION || pvio_hip_create(&opts, &ctx) != PVIO_OK) {
        std::fprintf(stderr, "no usable GPU\n"); // there is no CPU path
        return 1;
    }
    int rc = 0;
    try {
        auto reader = DatasetReader::create_reader(uri, ctx);
        if (!reader) throw std::runtime_error("unknown dataset scheme: " + uri);
        auto config = uri.rfind("euroc://", 0) == 0 ? DatasetConfig::euroc() : DatasetConfig::tum_vi();
        if (argc > 5 && std::atol(argv[5]) >= 2) config->window = (size_t)std::atol(argv[5]);
        if (argc > 6 && std::atol(argv[6]) >= 1) config->keyframe_gap = (size_t)std::atol(argv[6]);
        Vio vio(config);
        if (std::strcmp(argv[2], "-") != 0) {
            std::ifstream gt(argv[2]);
            std::vector<std::vector<double>> rows;
            std::string line;
            while (std::getline(gt, line)) {
                if (line.empty() || line[0] == '#') continue;
                std::istringstream ss(line);
                std::vector<double> r(8);
                bool ok = true;
                for (double &x : r) ok = ok && (bool)(ss >> x);
                if (ok) rows.push_back(r);
            }
            if (rows.size() < 2) throw std::runtime_error("ground truth file holds fewer than two poses");
            set_bootstrap(vio, rows);
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
        std::fprintf(stderr, "%ld poses written to %s (%s)\n", n_poses, out_path.c_str(), kControlPlane);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pvio_headless: %s\n", e.what());
        rc = 1;
    }
    pvio_hip_destroy(ctx);
    return rc;
}
