// front_shim.cpp -- extern "C" handles on the host front-end classes (pvio_amd/host/feature_front.*) for the Python tests.
#include <cstdint>
#include <cstring>
#include <memory>

#include "../../pvio_amd/host/feature_front.h"
#include "../../pvio_amd/host/fundamental_ransac.h"

using namespace pvio;

extern "C" {

void host_poisson_insert(double radius, int n_preset, const double *preset_xy, int n, const double *xy, uint8_t *accepted) {
    PoissonDisk2 f(radius);
    for (int i = 0; i < n_preset; ++i) {
        vector<2> p;
        p[0] = preset_xy[2 * i], p[1] = preset_xy[2 * i + 1];
        f.preset_point(p);
    }
    for (int i = 0; i < n; ++i) {
        vector<2> p;
        p[0] = xy[2 * i], p[1] = xy[2 * i + 1];
        accepted[i] = f.insert_point(p) ? 1 : 0;
    }
}

void host_select_tracked(int n, const double *next_xy, const uint64_t *track_length, double min_distance, uint8_t *status) {
    std::vector<vector<2>> nxt((size_t)n);
    std::vector<size_t> len((size_t)n);
    std::vector<char> st((size_t)n);
    for (int i = 0; i < n; ++i) nxt[i][0] = next_xy[2 * i], nxt[i][1] = next_xy[2 * i + 1], len[i] = (size_t)track_length[i], st[i] = (char)status[i];
    select_tracked(nxt, len, min_distance, st);
    for (int i = 0; i < n; ++i) status[i] = (uint8_t)st[i];
}

void host_predict_keypoints(const double q_cam_i[4], const double q_imu_i[4], const double dq[4], const double q_imu_j[4], const double q_cam_j[4],
                            const double K_next[4], int n, const double *kp_xy, double *out_xy) {
    Frame a, b;
    for (int k = 0; k < 4; ++k) {
        a.camera.q_cs.coeffs()[k] = q_cam_i[k], a.imu.q_cs.coeffs()[k] = q_imu_i[k];
        b.camera.q_cs.coeffs()[k] = q_cam_j[k], b.imu.q_cs.coeffs()[k] = q_imu_j[k];
        b.preintegration.delta.q.coeffs()[k] = dq[k];
    }
    b.K.setZero();
    b.K(0, 0) = K_next[0], b.K(1, 1) = K_next[1], b.K(0, 2) = K_next[2], b.K(1, 2) = K_next[3], b.K(2, 2) = 1;
    for (int i = 0; i < n; ++i) a.append_keypoint(vector<2>(kp_xy[2 * i], kp_xy[2 * i + 1]));
    std::vector<vector<2>> out;
    predict_keypoints(a, b, out);
    for (int i = 0; i < n; ++i) out_xy[2 * i] = out[i][0], out_xy[2 * i + 1] = out[i][1];
}

// pvio::Image seam: two images, preprocess both, track with an initial guess; returns 0 or -1 (message in err)
int host_image_track(const uint8_t *img0, const uint8_t *img1, int w, int h, int n, const double *curr_xy, double *next_xy /* in: guess, out */,
                     int use_guess, int use_ransac, uint8_t *status, char *err, int err_len) {
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    if (pvio_hip_create(&opts, &ctx) != 0 || !ctx) {
        std::strncpy(err, "pvio_hip_create failed (no GPU?)", (size_t)err_len - 1);
        return -1;
    }
    int rc = 0;
    try {
        HipImage a(ctx, img0, w, h, w, 0.0), b(ctx, img1, w, h, w, 0.05);
        a.preprocess(), b.preprocess();
        a.enable_ransac(use_ransac != 0);
        std::vector<vector<2>> cur((size_t)n), nxt;
        for (int i = 0; i < n; ++i) cur[i][0] = curr_xy[2 * i], cur[i][1] = curr_xy[2 * i + 1];
        if (use_guess) {
            nxt.resize((size_t)n);
            for (int i = 0; i < n; ++i) nxt[i][0] = next_xy[2 * i], nxt[i][1] = next_xy[2 * i + 1];
        }
        std::vector<char> st;
        const Image *next_img = &b;
        a.track_keypoints(next_img, cur, nxt, st);
        for (int i = 0; i < n; ++i) status[i] = (uint8_t)st[i], next_xy[2 * i] = nxt[i][0], next_xy[2 * i + 1] = nxt[i][1];
    } catch (const std::exception &e) {
        std::strncpy(err, e.what(), (size_t)err_len - 1);
        rc = -1;
    }
    pvio_hip_destroy(ctx);
    return rc;
}

// pvio::Image::evaluate (both overloads): value and gradient of the bicubic sample of pyramid level `level` at n points
int host_image_evaluate(const uint8_t *img, int w, int h, int level, int n, const double *uv, double *val, double *grad, char *err, int err_len) {
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    if (pvio_hip_create(&opts, &ctx) != 0 || !ctx) {
        std::strncpy(err, "pvio_hip_create failed (no GPU?)", (size_t)err_len - 1);
        return -1;
    }
    int rc = 0;
    try {
        HipImage a(ctx, img, w, h, w, 0.0);
        a.preprocess();
        const Image *im = &a;
        for (int i = 0; i < n; ++i) {
            vector<2> u(uv[2 * i], uv[2 * i + 1]), d;
            val[i] = im->evaluate(u, d, level);
            grad[2 * i] = d[0], grad[2 * i + 1] = d[1];
            if (im->evaluate(u, level) != val[i]) throw std::runtime_error("the two evaluate() overloads disagree");
        }
    } catch (const std::exception &e) {
        std::strncpy(err, e.what(), (size_t)err_len - 1);
        rc = -1;
    }
    pvio_hip_destroy(ctx);
    return rc;
}

int host_ransac(int n, const float *p, const float *q, double threshold, double confidence, uint8_t *mask, double *F) {
    std::vector<uint8_t> m;
    const int good = find_fundamental_ransac(n, p, q, threshold, confidence, m, F);
    for (int i = 0; i < n; ++i) mask[i] = m[(size_t)i];
    return good;
}

int host_7point(const float *p, const float *q, double *F27) { return fundamental_7point(p, q, F27); }

// pvio::Image seam, detection: existing keypoints in, all keypoints out (existing first); returns the total or -1
int host_image_detect(const uint8_t *img, int w, int h, int n_existing, const double *existing_xy, double keypoint_distance, int cap, double *out_xy, char *err,
                      int err_len) {
    pvio_hip_ctx *ctx = nullptr;
    pvio_hip_opts opts;
    std::memset(&opts, 0, sizeof(opts));
    if (pvio_hip_create(&opts, &ctx) != 0 || !ctx) {
        std::strncpy(err, "pvio_hip_create failed (no GPU?)", (size_t)err_len - 1);
        return -1;
    }
    int n = -1;
    try {
        HipImage a(ctx, img, w, h, w, 0.0);
        a.preprocess();
        std::vector<vector<2>> kps((size_t)n_existing);
        for (int i = 0; i < n_existing; ++i) kps[i][0] = existing_xy[2 * i], kps[i][1] = existing_xy[2 * i + 1];
        const Image *im = &a;
        im->detect_keypoints(kps, 0, keypoint_distance);
        n = (int)kps.size() < cap ? (int)kps.size() : cap;
        for (int i = 0; i < n; ++i) out_xy[2 * i] = kps[i][0], out_xy[2 * i + 1] = kps[i][1];
    } catch (const std::exception &e) {
        std::strncpy(err, e.what(), (size_t)err_len - 1);
        n = -1;
    }
    pvio_hip_destroy(ctx);
    return n;
}

} // extern "C"
