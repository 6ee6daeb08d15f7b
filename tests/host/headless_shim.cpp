// headless_shim.cpp -- extern "C" handle on the headless pipeline (tests/host/standin/headless.*) for the Python tests: feeds a
// sequence that is already in memory (pinhole images + IMU samples) the way pvio-pc's loop feeds a dataset
// (pvio-pc/src/main.cpp:207-258) and returns the poses the pipeline reports.
#include <cstdint>
#include <cstring>

#include "../../pvio_amd/host/feature_front.h"
#include "standin/headless.h"

using namespace pvio;

extern "C" int host_headless_run(int n_frames, int w, int h, const uint8_t *images, const double *image_t, int n_imu, const double *imu_t,
                                 const double *imu_w, const double *imu_a, const double *K4, const double *q_bc, const double *p_bc, int n_gt,
                                 const double *gt /* [n_gt][8] t p q(xyzw) */, int window, int keyframe_gap, double min_keypoint_distance,
                                 double *out_pose /* [n_frames][8] t p q, q = 0 while not initialized */, int32_t *stats /* [4] */, char *err, int err_len) {
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof opts);
    opts.world_size = 1, opts.use_graph = 1;
    if (pvio_hip_create(&opts, &ctx) != 0 || !ctx) {
        std::strncpy(err, "pvio_hip_create failed (no GPU?)", (size_t)err_len - 1);
        return -1;
    }
    int rc = 0;
    try {
        auto cfg = HeadlessConfig::euroc(); // noise constants; camera and extrinsics from the caller
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
        auto put = [&](int f, const OutputPose &p) {
            double *o = out_pose + 8 * f;
            o[0] = image_t[f];
            for (int c = 0; c < 3; ++c) o[1 + c] = p.p[c];
            o[4] = p.q.x(), o[5] = p.q.y(), o[6] = p.q.z(), o[7] = p.q.w();
        };
        for (int f = 0; f < n_frames; ++f) {
            while (k < n_imu && imu_t[k] <= image_t[f]) {
                vio.track_gyroscope(imu_t[k], imu_w[3 * k], imu_w[3 * k + 1], imu_w[3 * k + 2]);
                vio.track_accelerometer(imu_t[k], imu_a[3 * k], imu_a[3 * k + 1], imu_a[3 * k + 2]);
                ++k;
            }
            auto img = std::make_shared<HipImage>(ctx, images + (size_t)f * w * h, w, h, w, image_t[f]);
            put(f, vio.track_camera(img));
        }
        stats[0] = vio.initialized() ? 1 : 0, stats[1] = (int32_t)vio.window_frames(), stats[2] = (int32_t)vio.keyframe_solves();
        stats[3] = vio.window() ? (int32_t)vio.window()->track_num() : 0;
    } catch (const std::exception &e) {
        std::strncpy(err, e.what(), (size_t)err_len - 1);
        rc = -1;
    }
    pvio_hip_destroy(ctx);
    return rc;
}
