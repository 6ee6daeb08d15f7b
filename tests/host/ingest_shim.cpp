// ingest_shim.cpp -- extern "C" handles on the dataset-ingest host code (pvio_amd/host/{undistort_maps,image_io,dataset_reader}.*)
// for the Python tests.
#include <cstdint>
#include <chrono>
#include <cstring>
#include <string>

#include "../../pvio_amd/host/dataset_reader.h"
#include "../../pvio_amd/host/image_io.h"
#include "../../pvio_amd/host/undistort_maps.h"

using namespace pvio;

namespace {
void set_err(char *err, int err_len, const char *msg) {
    if (err && err_len > 0) std::strncpy(err, msg, (size_t)err_len - 1), err[err_len - 1] = 0;
}
} // namespace

extern "C" {

int host_cv_undistort_maps(const float *K9, const float *dist, int n_dist, int w, int h, int16_t *xy, uint16_t *frac) {
    try {
        const FixedRemap m = cv_undistort_fixed_maps(K9, dist, n_dist, w, h);
        std::memcpy(xy, m.xy.data(), m.xy.size() * 2), std::memcpy(frac, m.frac.data(), m.frac.size() * 2);
        return 0;
    } catch (...) {
        return -1;
    }
}

int host_image_undistorter_maps(int w, int h, const double *K9_row_major, const double *coeffs, int n_coeffs, const char *model, int16_t *xy, uint16_t *frac,
                                int n_probe, const double *probe_uv, double *probe_out) {
    try {
        matrix<3> K;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) K(r, c) = K9_row_major[3 * r + c];
        ImageUndistorter u((size_t)w, (size_t)h, K, std::vector<double>(coeffs, coeffs + n_coeffs), model);
        std::memcpy(xy, u.maps().xy.data(), u.maps().xy.size() * 2), std::memcpy(frac, u.maps().frac.data(), u.maps().frac.size() * 2);
        for (int i = 0; i < n_probe; ++i) {
            vector<2> p;
            p[0] = probe_uv[2 * i], p[1] = probe_uv[2 * i + 1];
            const vector<2> d = u.distort_pixel(p);
            probe_out[2 * i] = d[0], probe_out[2 * i + 1] = d[1];
        }
        return 0;
    } catch (...) {
        return -1;
    }
}

// returns 0 and fills w, h, pixels (capacity cap bytes); -1 with a message otherwise
int host_read_gray_image(const char *filename, int *w, int *h, uint8_t *pixels, int cap, char *err, int err_len) {
    try {
        const GrayImage img = read_gray_image(filename);
        *w = img.width, *h = img.height;
        if ((int)img.pixels.size() > cap) {
            set_err(err, err_len, "buffer too small");
            return -1;
        }
        std::memcpy(pixels, img.pixels.data(), img.pixels.size());
        return 0;
    } catch (const std::exception &e) {
        set_err(err, err_len, e.what());
        return -1;
    }
}

// Walks a sequence the way pvio-pc's main loop does (main.cpp: next() -> read_*): records every event; every camera event is
// read, preprocessed (undistortion + CLAHE + pyramid on the device) and its level 0 copied to `images` (img_cap bytes in all).
// Returns the number of events or -1.
int host_dataset_walk(const char *uri, int max_events, int32_t *types, double *times, double *values /* [max_events][3] */, uint8_t *images, int64_t img_cap,
                      int32_t *img_wh /* [2] */, int32_t *n_images, char *err, int err_len) {
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    if (pvio_hip_create(&opts, &ctx) != 0 || !ctx) {
        set_err(err, err_len, "pvio_hip_create failed (no GPU?)");
        return -1;
    }
    int n = 0;
    *n_images = 0;
    try {
        auto reader = DatasetReader::create_reader(uri, ctx);
        if (!reader) throw std::runtime_error("unknown dataset scheme");
        int64_t used = 0;
        for (;;) {
            const DatasetReader::NextDataType type = reader->next();
            if (type == DatasetReader::END || n >= max_events) break;
            types[n] = (int32_t)type;
            values[3 * n] = values[3 * n + 1] = values[3 * n + 2] = 0;
            if (type == DatasetReader::CAMERA) {
                std::shared_ptr<Image> im = reader->read_image();
                times[n] = im->t;
                im->preprocess();
                const HipImage *hi = dynamic_cast<const HipImage *>(im.get());
                int32_t w = 0, h = 0;
                if (pvio_hip_image_download_level(ctx, hi->device_image(), 0, nullptr, nullptr, &w, &h) != 0) throw std::runtime_error("download_level");
                if ((int32_t)im->width() != w || (int32_t)im->height() != h) throw std::runtime_error("Image::width()/height() disagree with the pyramid");
                if (used + (int64_t)w * h > img_cap) throw std::runtime_error("image buffer too small");
                if (pvio_hip_image_download_level(ctx, hi->device_image(), 0, images + used, nullptr, &w, &h) != 0) throw std::runtime_error("download_level");
                used += (int64_t)w * h;
                img_wh[0] = w, img_wh[1] = h;
                ++*n_images;
            } else if (type == DatasetReader::GYROSCOPE) {
                auto [t, v] = reader->read_gyroscope();
                times[n] = t, values[3 * n] = v[0], values[3 * n + 1] = v[1], values[3 * n + 2] = v[2];
            } else if (type == DatasetReader::ACCELEROMETER) {
                auto [t, v] = reader->read_accelerometer();
                times[n] = t, values[3 * n] = v[0], values[3 * n + 1] = v[1], values[3 * n + 2] = v[2];
            }
            ++n;
        }
    } catch (const std::exception &e) {
        set_err(err, err_len, e.what());
        n = -1;
    }
    pvio_hip_destroy(ctx);
    return n;
}

// A reduced FeatureTracker::work loop (core/feature_tracker.cpp:37-130) over a sequence: every image is read, undistorted and
// preprocessed on the device, the keypoints of the previous image are tracked into it (device LK + border gate + host RANSAC),
// survivors are kept and new corners detected where there is room.  per_frame[k] = {tracked in, survived, total after
// detection, milliseconds}; mean_flow[k] = mean displacement of the survivors.  Returns the number of images or -1.
int host_replay_front_end(const char *uri, int max_frames, double keypoint_distance, int32_t *per_frame /* [max_frames][3] */, double *ms /* [max_frames] */,
                          double *mean_flow /* [max_frames][2] */, char *err, int err_len) {
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    if (pvio_hip_create(&opts, &ctx) != 0 || !ctx) {
        set_err(err, err_len, "pvio_hip_create failed (no GPU?)");
        return -1;
    }
    int n = 0;
    try {
        auto reader = DatasetReader::create_reader(uri, ctx);
        if (!reader) throw std::runtime_error("unknown dataset scheme");
        std::shared_ptr<Image> prev;
        std::vector<vector<2>> kps;
        for (;;) {
            const DatasetReader::NextDataType type = reader->next();
            if (type == DatasetReader::END || n >= max_frames) break;
            if (type == DatasetReader::GYROSCOPE) {
                reader->read_gyroscope();
                continue;
            }
            if (type == DatasetReader::ACCELEROMETER) {
                reader->read_accelerometer();
                continue;
            }
            const auto t0 = std::chrono::steady_clock::now();
            std::shared_ptr<Image> cur = reader->read_image();
            cur->preprocess();
            int tracked_in = (int)kps.size(), survived = 0;
            double fx = 0, fy = 0;
            if (prev && !kps.empty()) {
                std::vector<vector<2>> next; // no initial flow: starts from the previous positions
                std::vector<char> status;
                prev->track_keypoints(cur.get(), kps, next, status);
                std::vector<vector<2>> kept;
                for (size_t i = 0; i < kps.size(); ++i)
                    if (status[i]) {
                        fx += next[i][0] - kps[i][0], fy += next[i][1] - kps[i][1];
                        kept.push_back(next[i]);
                    }
                survived = (int)kept.size();
                kps.swap(kept);
            }
            cur->detect_keypoints(kps, 0, keypoint_distance); // appends corners that keep their distance to the existing ones
            per_frame[3 * n] = tracked_in, per_frame[3 * n + 1] = survived, per_frame[3 * n + 2] = (int32_t)kps.size();
            mean_flow[2 * n] = survived ? fx / survived : 0.0, mean_flow[2 * n + 1] = survived ? fy / survived : 0.0;
            ms[n] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            prev = cur;
            ++n;
        }
    } catch (const std::exception &e) {
        set_err(err, err_len, e.what());
        n = -1;
    }
    pvio_hip_destroy(ctx);
    return n;
}

int host_tum_write(const char *filename, int n, const double *t, const double *p /* [n][3] */, const double *q /* [n][4] x y z w */) {
    TumOutputWriter w(filename);
    if (!w.is_open()) return -1;
    for (int i = 0; i < n; ++i) {
        OutputPose pose;
        for (int k = 0; k < 3; ++k) pose.p[k] = p[3 * i + k];
        for (int k = 0; k < 4; ++k) pose.q.coeffs()[k] = q[4 * i + k];
        w.write_pose(t[i], pose);
    }
    return 0;
}

} // extern "C"
