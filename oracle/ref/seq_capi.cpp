// seq_capi.cpp -- TEST INFRASTRUCTURE (oracle/ref/Makefile): the reference's own pvio::PVIO driven over a sequence that is already in memory.
//
// Compiled into BOTH libraries of oracle/ref/Makefile, above the reference's whole control plane compiled unedited (pvio.cpp,
// core/{core,feature_tracker,frontend_worker,sliding_window_tracker,plane_extractor}.cpp, map/*.cpp; only the SfM initializer is the
// bootstrap of gt_initializer.cpp):
//   libpvio_ref.so          + the reference's own BundleAdjustor / visual_inertial_pnp (mini-Ceres below them)
//   libpvio_dropin[_emu].so + the PRODUCT's pvio_amd/host/{bundle_adjustor,pnp,pnp_solve}.cpp above the HIP C ABI, and (image kind "hip")
//                             the product's pvio::HipImage (pvio_amd/host/feature_front.cpp) as the pvio::Image
// So the calls below -- pvio::PVIO::track_gyroscope / track_accelerometer / track_camera, pvio/include/pvio/pvio.h:135-148 -- are exactly
// what pvio-pc's main loop makes (pvio-pc/src/main.cpp:207-258), and everything between them and the hot path is the reference's:
// Core's IMU pairing and frame setup, FeatureTracker::work, Frame::track_keypoints / detect_keypoints + PoissonDiskFilter,
// FrontendWorker::work, SlidingWindowTracker::mirror_frame / track / keyframe_check, Map::marginalize_frame, PlaneExtractor.
// This is "drops into pvio-pc unchanged", run.
//
// Records: tests/host/chain_log.h tags 1 (camera frame: track id / length / position of every keypoint of the feature tracker's newest
// frame + the reported pose) and 8 (sliding window tracks), plus
//   tag 9  window frames   ints: frame index, N, then per frame (id, FF_KEYFRAME, FF_FIX_POSE, keypoints), planes, then per plane (id, tracks)
//                          doubles: per frame q(xyzw) p v bg ba, then per plane normal(3) distance
// Private members of PVIO / FrontendWorker are read for those records (this one file is compiled with -fno-access-control).
#include <pvio/common.h>
#include <pvio/core/core.h>
#include <pvio/core/feature_tracker.h>
#include <pvio/core/frontend_worker.h>
#include <pvio/core/sliding_window_tracker.h>
#include <pvio/map/frame.h>
#include <pvio/map/map.h>
#include <pvio/map/plane.h>
#include <pvio/map/track.h>
#include <pvio/pvio.h>

#include <pvio_hip.h>

#include <cstdlib>
#include <cstring>

#include "../../tests/host/chain_log.h"
#include "../../tests/host/oracle_image.h"
#include "seq_bootstrap.h"
#ifdef SEQ_WITH_HIP_IMAGE
#include "feature_front.h" // pvio::HipImage of pvio_amd/host (include path of the dropin objects)
#endif

using namespace pvio;

namespace {

struct SeqConfig : public Config { // constants of config/euroc.yaml:11-45 (the reference parses them with yaml-cpp, absent here); camera from the caller
    matrix<3> K;
    quaternion q_bc;
    vector<3> p_bc;
    size_t window = 8, gap = 5;
    double min_distance = 20.0;
    static matrix<3> diag3(double v) {
        matrix<3> m = matrix<3>::Zero();
        m(0, 0) = m(1, 1) = m(2, 2) = v;
        return m;
    }
    matrix<3> camera_intrinsic() const override { return K; }
    quaternion camera_to_body_rotation() const override { return q_bc; }
    vector<3> camera_to_body_translation() const override { return p_bc; }
    quaternion imu_to_body_rotation() const override { return quaternion::Identity(); }
    vector<3> imu_to_body_translation() const override { return vector<3>::Zero(); }
    matrix<2> keypoint_noise_cov() const override {
        matrix<2> m = matrix<2>::Zero();
        m(0, 0) = m(1, 1) = 0.5;
        return m;
    }
    matrix<3> gyroscope_noise_cov() const override { return diag3(2.8791302399999997e-08); }
    matrix<3> accelerometer_noise_cov() const override { return diag3(4.0e-6); }
    matrix<3> gyroscope_bias_noise_cov() const override { return diag3(3.7608844899999997e-10); }
    matrix<3> accelerometer_bias_noise_cov() const override { return diag3(9.0e-6); }
    size_t sliding_window_size() const override { return window; }
    size_t initializer_keyframe_gap() const override { return gap; }
    double feature_tracker_min_keypoint_distance() const override { return min_distance; }
};

} // namespace

extern "C" int host_chain_run(int n_frames, int w, int h, const uint8_t *images, const double *image_t, int n_imu, const double *imu_t, const double *imu_w,
                              const double *imu_a, const double *K4, const double *q_bc, const double *p_bc, int n_gt, const double *gt /* [n_gt][8] t p q(xyzw) */,
                              int window, int keyframe_gap, double min_keypoint_distance, const char *log_path, double *out_pose /* [n_frames][8] */,
                              int32_t *stats /* [4] */, char *err, int err_len) {
    const char *kind = std::getenv("PVIO_SEQ_IMAGE"); // "oracle" (default): the CPU oracle's front end; "hip": the product's HipImage (dropin libraries only)
    const bool hip_image = kind && std::strcmp(kind, "hip") == 0;
    pvio_hip_ctx *ctx = nullptr;
#ifdef SEQ_WITH_HIP_IMAGE
    if (hip_image) {
        pvio_hip_opts opts;
        std::memset(&opts, 0, sizeof opts);
        opts.world_size = 1, opts.use_graph = 1;
        if (pvio_hip_create(&opts, &ctx) != 0 || !ctx) {
            std::strncpy(err, "pvio_hip_create failed (no GPU?)", (size_t)err_len - 1);
            return -1;
        }
    }
#else
    if (hip_image) {
        std::strncpy(err, "this library has no HipImage (the reference side of the A/B)", (size_t)err_len - 1);
        return -1;
    }
#endif
    chain_log::open(log_path);
    int rc = 0;
    try {
        auto cfg = std::make_shared<SeqConfig>();
        cfg->K.setZero();
        cfg->K(0, 0) = K4[0], cfg->K(1, 1) = K4[1], cfg->K(0, 2) = K4[2], cfg->K(1, 2) = K4[3], cfg->K(2, 2) = 1;
        cfg->q_bc = quaternion(q_bc[3], q_bc[0], q_bc[1], q_bc[2]);
        cfg->q_bc.normalize();
        cfg->p_bc = vector<3>(p_bc[0], p_bc[1], p_bc[2]);
        cfg->window = (size_t)window, cfg->gap = (size_t)keyframe_gap, cfg->min_distance = min_keypoint_distance;
        std::vector<SeqTimedPose> &B = seq_bootstrap();
        B.resize((size_t)n_gt);
        for (int i = 0; i < n_gt; ++i) {
            const double *g = gt + 8 * i;
            B[(size_t)i].t = g[0], B[(size_t)i].p = vector<3>(g[1], g[2], g[3]), B[(size_t)i].q = quaternion(g[7], g[4], g[5], g[6]);
        }
        PVIO vio(cfg); // the reference's own top-level object (pvio.cpp)
        size_t solves_seen = 0;
        int k = 0;
        for (int f = 0; f < n_frames; ++f) {
            while (k < n_imu && imu_t[k] <= image_t[f]) {
                vio.track_gyroscope(imu_t[k], imu_w[3 * k], imu_w[3 * k + 1], imu_w[3 * k + 2]);
                vio.track_accelerometer(imu_t[k], imu_a[3 * k], imu_a[3 * k + 1], imu_a[3 * k + 2]);
                ++k;
            }
            std::shared_ptr<Image> img;
#ifdef SEQ_WITH_HIP_IMAGE
            if (hip_image) img = std::make_shared<HipImage>(ctx, images + (size_t)f * w * h, w, h, w, image_t[f]);
#endif
            if (!img) img = std::make_shared<OracleImage>(images + (size_t)f * w * h, w, h, image_t[f]);
            // Core::track_camera only queues the frame; it reaches the feature tracker with the first IMU sample behind it (core.cpp:127-140),
            // so the pose reported here is the prediction pvio-pc would write to trajectory.tum for this image (main.cpp:236-249)
            const OutputPose p = vio.track_camera(img);
            double *o = out_pose + 8 * f;
            o[0] = image_t[f];
            for (int c = 0; c < 3; ++c) o[1 + c] = p.p[c];
            o[4] = p.q.x(), o[5] = p.q.y(), o[6] = p.q.z(), o[7] = p.q.w();
            const Map *ft = vio.core->feature_tracker->map.get();
            const SlidingWindowTracker *swt = vio.core->frontend->sliding_window_tracker.get();
            const Map *wm = swt ? swt->map.get() : nullptr;
            const Frame *last = ft && ft->frame_num() ? ft->get_frame(ft->frame_num() - 1) : nullptr;
            std::vector<int64_t> I = {f, last ? (int64_t)last->id() : -1, wm ? 1 : 0, wm ? (int64_t)wm->frame_num() : 0, last ? (int64_t)last->keypoint_num() : 0};
            std::vector<double> D;
            if (last)
                for (size_t i = 0; i < last->keypoint_num(); ++i) {
                    const Track *t = last->get_track(i);
                    I.push_back(t ? (int64_t)t->id() : 0), I.push_back(t ? (int64_t)t->keypoint_num() : 0);
                    D.push_back(last->get_keypoint(i)[0]), D.push_back(last->get_keypoint(i)[1]);
                }
            D.insert(D.end(), o, o + 8);
            chain_log::record(1, I, D);
            if (wm) {
                std::vector<int64_t> WI = {f, (int64_t)wm->frame_num(), (int64_t)wm->track_num()};
                std::vector<double> WD;
                for (size_t i = 0; i < wm->track_num(); ++i) {
                    const Track *t = wm->get_track(i);
                    WI.push_back((int64_t)t->id()), WI.push_back((t->flag(TrackFlag::TF_VALID) ? 1 : 0) | (t->flag(TrackFlag::TF_PLANE) ? 2 : 0)), WI.push_back((int64_t)t->keypoint_num());
                    WD.push_back(t->landmark.inv_depth), WD.push_back(t->landmark.quality);
                }
                chain_log::record(8, WI, WD);
                std::vector<int64_t> FI = {f, (int64_t)wm->frame_num()};
                std::vector<double> FD;
                for (size_t i = 0; i < wm->frame_num(); ++i) {
                    const Frame *fr = wm->get_frame(i);
                    FI.push_back((int64_t)fr->id()), FI.push_back(fr->flag(FrameFlag::FF_KEYFRAME) ? 1 : 0), FI.push_back(fr->flag(FrameFlag::FF_FIX_POSE) ? 1 : 0), FI.push_back((int64_t)fr->keypoint_num());
                    FD.insert(FD.end(), {fr->pose.q.x(), fr->pose.q.y(), fr->pose.q.z(), fr->pose.q.w()});
                    for (int c = 0; c < 3; ++c) FD.push_back(fr->pose.p[c]);
                    for (int c = 0; c < 3; ++c) FD.push_back(fr->motion.v[c]);
                    for (int c = 0; c < 3; ++c) FD.push_back(fr->motion.bg[c]);
                    for (int c = 0; c < 3; ++c) FD.push_back(fr->motion.ba[c]);
                }
                FI.push_back((int64_t)wm->plane_num());
                for (size_t i = 0; i < wm->plane_num(); ++i) {
                    const Plane *pl = wm->get_plane(i);
                    FI.push_back((int64_t)pl->id()), FI.push_back((int64_t)pl->tracks.size());
                    FD.insert(FD.end(), {pl->parameter.normal[0], pl->parameter.normal[1], pl->parameter.normal[2], pl->parameter.distance});
                }
                chain_log::record(9, FI, FD);
                if (wm->last_frame()->flag(FrameFlag::FF_KEYFRAME)) ++solves_seen; // a rough count: frames after which the window ends in a keyframe
            }
            stats[0] = wm ? 1 : 0, stats[1] = wm ? (int32_t)wm->frame_num() : 0, stats[2] = (int32_t)solves_seen, stats[3] = wm ? (int32_t)wm->track_num() : 0;
        }
    } catch (const std::exception &e) {
        std::strncpy(err, e.what(), (size_t)err_len - 1);
        rc = -1;
    }
    chain_log::close();
    if (ctx) pvio_hip_destroy(ctx);
    return rc;
}
