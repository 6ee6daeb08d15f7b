// ingest_shim.cpp -- extern "C" handles on the dataset-ingest host code (pvio_amd/host/{undistort_maps,image_io,dataset_reader}.*)
// for the Python tests.
#include <cstdint>
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

int host_tum_write(const char *filename, int n, const double *t, const double *p /* [n][3] */, const double *q /* [n][4] x y z w */) {
    TumOutputWriter w(filename);
    if (!w.is_open()) return -1;
    for (int i = 0; i < n; ++i) {
        OutputPose pose;
        for (int k = 0; k < 3; ++k) pose.p[k] = p[3 * i + k];
        for (int k = 0; k < 4; ++k) pose.q.c[k] = q[4 * i + k];
        w.write_pose(t[i], pose);
    }
    return 0;
}

} // extern "C"
