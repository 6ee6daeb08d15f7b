// feature_front.cpp -- see feature_front.h.  Host C++ above the C ABI; no OpenCV, no Eigen.
#include "feature_front.h"

#include <cstdio>
#include <cstdlib>

#include "fundamental_ransac.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <utility>

namespace pvio {

// ------------------------------------------------------------------------------------------------------------------
// PoissonDisk2
// ------------------------------------------------------------------------------------------------------------------
PoissonDisk2::PoissonDisk2(double radius) :
    radius_(radius), radius2_(radius * radius), cell_(radius / std::sqrt(2.0)), span_((int)std::ceil(std::sqrt(2.0))) {}

void PoissonDisk2::clear() {
    points_.clear();
    grid_.clear();
}

size_t PoissonDisk2::KeyHash::operator()(const Key &k) const {
    // any hash works (the table is only probed with find()); this is the reference's combiner for reproducible bucket use
    size_t seed = 0;
    seed ^= std::hash<int>()(k.x) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    seed ^= std::hash<int>()(k.y) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
}

PoissonDisk2::Key PoissonDisk2::cell_of(const vector<2> &p) const {
    return Key{(int)std::floor(p[0] / cell_), (int)std::floor(p[1] / cell_)};
}

void PoissonDisk2::preset_point(const vector<2> &p) {
    grid_[cell_of(p)] = points_.size(); // a cell remembers only its latest point
    points_.push_back(p);
}

bool PoissonDisk2::test(const vector<2> &p, Key &cell) const {
    cell = cell_of(p);
    const int x0 = cell.x - span_, x1 = cell.x + span_, y0 = cell.y - span_, y1 = cell.y + span_;
    // The reference advances its cursor BEFORE the first probe and tests the row bound only on entry, so the visiting
    // order is (x0+1,y0) .. (x1,y0), (x0,y0+1) .. (x1,y1), and finally (x0, y1+1); (x0,y0) is never looked at.
    int cx = x0, cy = y0;
    while (cy <= y1) {
        ++cx;
        if (cx > x1) cx = x0, ++cy;
        auto it = grid_.find(Key{cx, cy});
        if (it != grid_.end()) {
            const vector<2> &q = points_[it->second];
            const double dx = p[0] - q[0], dy = p[1] - q[1];
            if (dx * dx + dy * dy < radius2_) return false;
        }
    }
    return true;
}

bool PoissonDisk2::permit_point(const vector<2> &p) const {
    Key c;
    return test(p, c);
}

bool PoissonDisk2::insert_point(const vector<2> &p) {
    Key c;
    if (!test(p, c)) return false;
    grid_[c] = points_.size();
    points_.push_back(p);
    return true;
}

void PoissonDisk2::insert_points(std::vector<vector<2>> &candidates) {
    const size_t before = points_.size();
    for (const auto &p : candidates) insert_point(p);
    candidates.assign(points_.begin() + (std::ptrdiff_t)before, points_.end());
}

// ------------------------------------------------------------------------------------------------------------------
// keypoint prediction and survivor selection
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Q { // quaternion x y z w, Hamilton product
    double x, y, z, w;
};
Q qmul(const Q &a, const Q &b) {
    return Q{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
             a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
Q qconj(const Q &a) { return Q{-a.x, -a.y, -a.z, a.w}; }
Q load(const quaternion &q) { return Q{q.x(), q.y(), q.z(), q.w()}; }
void rotate(const Q &q, const double v[3], double out[3]) { // v + 2 w (u x v) + 2 u x (u x v)
    const double ux = q.x, uy = q.y, uz = q.z;
    const double tx = 2.0 * (uy * v[2] - uz * v[1]), ty = 2.0 * (uz * v[0] - ux * v[2]), tz = 2.0 * (ux * v[1] - uy * v[0]);
    out[0] = v[0] + q.w * tx + (uy * tz - uz * ty);
    out[1] = v[1] + q.w * ty + (uz * tx - ux * tz);
    out[2] = v[2] + q.w * tz + (ux * ty - uy * tx);
}
} // namespace

void predict_keypoints(const Frame &curr, const Frame &next, std::vector<vector<2>> &next_pixels) {
    // delta = (q_cam_i^-1 * q_imu_i * dq_ij * q_imu_j^-1 * q_cam_j)^-1 : rotation taking a bearing of camera i to camera j
    const Q chain = qmul(qmul(qmul(qmul(qconj(load(curr.camera.q_cs)), load(curr.imu.q_cs)), load(next.preintegration.delta.q)), qconj(load(next.imu.q_cs))),
                         load(next.camera.q_cs));
    const Q delta = qconj(chain);
    next_pixels.resize(curr.keypoint_num());
    for (size_t i = 0; i < curr.keypoint_num(); ++i) {
        const double b[3] = {curr.get_keypoint(i)[0], curr.get_keypoint(i)[1], 1.0};
        double r[3];
        rotate(delta, b, r);
        next_pixels[i][0] = (r[0] / r[2]) * next.K(0, 0) + next.K(0, 2);
        next_pixels[i][1] = (r[1] / r[2]) * next.K(1, 1) + next.K(1, 2);
    }
}

void select_tracked(const std::vector<vector<2>> &next_pixels, const std::vector<size_t> &track_length, double min_distance, std::vector<char> &status) {
    std::vector<std::pair<size_t, size_t>> order; // (keypoint index, track length)
    order.reserve(status.size());
    for (size_t i = 0; i < status.size(); ++i)
        if (status[i] != 0 && track_length[i] != 0) order.emplace_back(i, track_length[i]);
    // same algorithm and comparator as the reference (std::sort, not stable): ties keep the library's order
    std::sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.second > b.second; });
    PoissonDisk2 filter(min_distance);
    for (const auto &kl : order) {
        const vector<2> &pt = next_pixels[kl.first];
        if (filter.permit_point(pt)) filter.preset_point(pt);
        else status[kl.first] = 0;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// HipImage
// ------------------------------------------------------------------------------------------------------------------
HipImage::HipImage(pvio_hip_ctx *ctx, const uint8_t *pixels, int width, int height, int stride, double timestamp) : ctx_(ctx), w_(width), h_(height) {
    t = timestamp;
    pixels_.resize((size_t)width * height);
    for (int y = 0; y < height; ++y) std::copy(pixels + (size_t)y * stride, pixels + (size_t)y * stride + width, pixels_.begin() + (size_t)y * width);
}

HipImage::~HipImage() {
    if (img_) pvio_hip_image_release(ctx_, img_);
}

void HipImage::preprocess() {
    if (img_) pvio_hip_image_release(ctx_, img_), img_ = nullptr;
    forget_host_levels();
    const int32_t rc = pvio_hip_image_create(ctx_, pixels_.data(), w_, h_, w_, /*apply_clahe=*/1, &img_);
    if (rc != 0) throw std::runtime_error(std::string("pvio_hip_image_create: ") + pvio_hip_last_error(ctx_)); // no CPU path
}

const HipImage::HostLevel &HipImage::host_level(int level) const {
    if (!img_) throw std::runtime_error("HipImage::evaluate: preprocess() was not called");
    if (host_levels_.empty()) host_levels_.resize(PVIO_KLT_LEVELS);
    HostLevel &L = host_levels_.at((size_t)level);
    if (L.px.empty()) {
        int32_t w = 0, h = 0;
        if (pvio_hip_image_download_level(ctx_, img_, level, nullptr, nullptr, &w, &h) != 0) throw std::runtime_error(pvio_hip_last_error(ctx_));
        L.px.resize((size_t)w * h);
        if (pvio_hip_image_download_level(ctx_, img_, level, L.px.data(), nullptr, &w, &h) != 0) throw std::runtime_error(pvio_hip_last_error(ctx_));
        L.w = w, L.h = h;
    }
    return L;
}

namespace {
// Catmull-Rom segment through p1, p2 (Ceres' CubicHermiteSpline): value and derivative at x in [0, 1]
inline void cubic(double p0, double p1, double p2, double p3, double x, double &f, double &dfdx) {
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3), b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3), c = 0.5 * (-p0 + p2);
    f = p1 + x * (c + x * (b + x * a));
    dfdx = c + x * (2.0 * b + 3.0 * a * x);
}
} // namespace

double HipImage::evaluate(const vector<2> &u, vector<2> &ddu, int level) const {
    const HostLevel &L = host_level(level);
    // opencv_image.cpp:150-154: the per-level scale is an INTEGER quotient of the extents
    const int W0 = (int)width(), H0 = (int)height(); // of level 0 (an undistorted image has the size of its maps, not of its source)
    const double sx = 1.0 / (double)((W0 - 1) / (L.w - 1)), sy = 1.0 / (double)((H0 - 1) / (L.h - 1));
    const double c = u[0] * sx, r = u[1] * sy;
    const int row = (int)std::floor(r), col = (int)std::floor(c);
    auto px = [&](int rr, int cc) { // Grid2D clamps to the image
        rr = std::min(std::max(rr, 0), L.h - 1), cc = std::min(std::max(cc, 0), L.w - 1);
        return (double)L.px[(size_t)rr * L.w + cc];
    };
    double f[4], dfdc[4];
    for (int k = 0; k < 4; ++k) cubic(px(row - 1 + k, col - 1), px(row - 1 + k, col), px(row - 1 + k, col + 1), px(row - 1 + k, col + 2), c - col, f[k], dfdc[k]);
    double val, dfdr, dc, unused;
    cubic(f[0], f[1], f[2], f[3], r - row, val, dfdr);
    cubic(dfdc[0], dfdc[1], dfdc[2], dfdc[3], r - row, dc, unused);
    ddu[0] = dc * sx, ddu[1] = dfdr * sy;
    return val;
}

double HipImage::evaluate(const vector<2> &u, int level) const {
    vector<2> ddu;
    return evaluate(u, ddu, level);
}

void HipImage::detect_keypoints(std::vector<vector<2>> &keypoints, size_t max_points, double keypoint_distance) const {
    // opencv_image.cpp:54-86.  `max_points` only gets a default there and is not passed on: the detector is the fixed
    // GFTTDetector::create(1000, 1e-3, 20, 3, useHarris = true) of :183.
    if (max_points == 0) max_points = 100;
    (void)max_points;
    if (!img_) throw std::runtime_error("HipImage::detect_keypoints: preprocess() was not called");
    constexpr int kMaxCorners = 1000;
    std::vector<float> xy(2 * kMaxCorners), resp(kMaxCorners);
    int32_t n = 0;
    const int32_t rc = pvio_hip_image_detect(ctx_, img_, kMaxCorners, 1.0e-3, 20.0, xy.data(), resp.data(), &n);
    if (rc != 0) throw std::runtime_error(std::string("pvio_hip_image_detect: ") + pvio_hip_last_error(ctx_));
    if (n == 0) return;
    // strongest first (the device call already returns them in that order; the reference re-sorts by response, :64-66)
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return resp[(size_t)a] > resp[(size_t)b]; });
    std::vector<vector<2>> fresh;
    fresh.reserve((size_t)n);
    for (int i : order) {
        vector<2> p;
        p[0] = xy[2 * (size_t)i], p[1] = xy[2 * (size_t)i + 1];
        fresh.push_back(p);
    }
    PoissonDisk2 filter(keypoint_distance);
    for (const auto &p : keypoints) filter.preset_point(p); // existing keypoints block their neighbourhood (:72-73)
    filter.insert_points(fresh);
    for (const auto &p : fresh)
        if (!(p[0] < 20 || p[1] < 20 || p[0] >= w_ - 20 || p[1] >= h_ - 20)) keypoints.push_back(p); // 20 px border (:76-82)
}

void HipImage::track_keypoints(const Image *next_image, const std::vector<vector<2>> &curr_keypoints, std::vector<vector<2>> &next_keypoints, std::vector<char> &result_status) const {
    const size_t n = curr_keypoints.size();
    std::vector<float> prev_xy(2 * n), next_xy(2 * n);
    for (size_t i = 0; i < n; ++i) prev_xy[2 * i] = (float)curr_keypoints[i][0], prev_xy[2 * i + 1] = (float)curr_keypoints[i][1];
    if (next_keypoints.size() > 0) { // initial flow given
        for (size_t i = 0; i < n; ++i) next_xy[2 * i] = (float)next_keypoints[i][0], next_xy[2 * i + 1] = (float)next_keypoints[i][1];
    } else {
        next_keypoints.resize(n);
        next_xy = prev_xy;
    }
    result_status.resize(n, 0);
    const HipImage *next = dynamic_cast<const HipImage *>(next_image);
    if (next && n > 0) {
        if (!img_ || !next->img_) throw std::runtime_error("HipImage::track_keypoints: preprocess() was not called");
        std::vector<uint8_t> st(n, 0);
        const std::vector<float> init_xy = next_xy;
        const int32_t rc = pvio_hip_klt_track(ctx_, img_, next->img_, (int32_t)n, prev_xy.data(), next_xy.data(), st.data()); // LK + 20 px border gate
        if (rc != 0) throw std::runtime_error(std::string("pvio_hip_klt_track: ") + pvio_hip_last_error(ctx_));
        if (const char *dump = std::getenv("PVIO_KLT_DUMP")) { // diagnostics: every LK call's inputs and outputs, appended to <dump>_hip.bin (tests/probe_klt_dump.py replays them)
            if (FILE *f = std::fopen((std::string(dump) + "_hip.bin").c_str(), "ab")) {
                const int32_t nn = (int32_t)n;
                std::fwrite(&nn, 4, 1, f), std::fwrite(prev_xy.data(), 4, 2 * n, f), std::fwrite(init_xy.data(), 4, 2 * n, f), std::fwrite(next_xy.data(), 4, 2 * n, f), std::fwrite(st.data(), 1, n, f);
                std::fclose(f);
            }
        }
        for (size_t i = 0; i < n; ++i) result_status[i] = (char)st[i];
    }
    if (filter_) {
        std::vector<vector<2>> nxt(n);
        for (size_t i = 0; i < n; ++i) nxt[i][0] = next_xy[2 * i], nxt[i][1] = next_xy[2 * i + 1];
        filter_(curr_keypoints, nxt, result_status);
    } else if (ransac_) {
        // opencv_image.cpp:113-129: fundamental-matrix RANSAC (threshold 1 px, confidence 0.99) over the survivors, when
        // there are at least eight of them; its outliers are dropped
        std::vector<size_t> l;
        std::vector<float> p, q;
        for (size_t i = 0; i < n; ++i)
            if (result_status[i] != 0) {
                l.push_back(i);
                p.push_back(prev_xy[2 * i]), p.push_back(prev_xy[2 * i + 1]), q.push_back(next_xy[2 * i]), q.push_back(next_xy[2 * i + 1]);
            }
        if (l.size() >= 8) {
            // hypotheses in batches on the device, the adaptive stopping rule replayed on the host (pvio_hip_fundamental_ransac; the
            // sequential host form, fundamental_ransac.h, is what the tests hold it against)
            std::vector<uint8_t> mask(l.size(), 0);
            int32_t good = 0;
            const int32_t rc = pvio_hip_fundamental_ransac(ctx_, (int32_t)l.size(), p.data(), q.data(), 1.0, 0.99, 1000, mask.data(), nullptr, &good);
            if (rc != 0) throw std::runtime_error(std::string("pvio_hip_fundamental_ransac: ") + pvio_hip_last_error(ctx_)); // no CPU path
            for (size_t i = 0; i < l.size(); ++i)
                if (mask[i] == 0) result_status[l[i]] = 0;
        }
    }
    for (size_t i = 0; i < n; ++i)
        if (result_status[i]) next_keypoints[i][0] = next_xy[2 * i], next_keypoints[i][1] = next_xy[2 * i + 1];
}

} // namespace pvio
