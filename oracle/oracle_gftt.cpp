// oracle_gftt.cpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into or called by the product).
//
// Restates what the reference gets from OpenCV in OpenCvImage::detect_keypoints
// (pvio-extra/src/pvio/extra/opencv_image.cpp:54-86, detector created at :183 as
// GFTTDetector::create(1000, 1e-3, 20, 3, /*useHarrisDetector=*/true), i.e. k = 0.04): cv::goodFeaturesToTrack with the
// Harris measure.  OpenCV is a third-party dependency (unpinned find_package, pvio-extra/depends/CMakeLists.txt) that is
// not in /root/reference and not installed here -> PARITY UNPINNED; the published algorithm is restated with one fixed,
// documented order of the float operations (the product kernels use the same order, so the two agree bit for bit):
//   1. Dx, Dy: 3x3 Sobel with BORDER_REFLECT_101; the smoothing kernel carries scale = 1 / (4 * blockSize * 255):
//        Dx = s * (d[y-1] + d[y+1]) + (2 s) * d[y],  d[y] = p[y][x+1] - p[y][x-1]   (row pass exact in integers)
//   2. cov = (Dx^2, Dx Dy, Dy^2); 3 x 3 unnormalized box sum with BORDER_REFLECT_101 on the cov images, rows top to
//      bottom, left to right inside a row
//   3. Harris response R = a c - b^2 - k (a + c)^2
//   4. threshold: R <= quality * max(R) -> 0 ; 3 x 3 dilation ; candidates = interior pixels (1-px frame excluded) whose
//      value is non-zero and equals the dilated value
//   5. candidates by response descending (ties: higher address first), greedy minimum-distance selection on a grid of
//      cell size round(minDistance), at most max_corners
// and then the reference's own post-processing (:63-84): Poisson-disk filter against the existing keypoints, 20 px border.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {
int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
} // namespace

extern "C" {

// response map (h x w floats) of steps 1-3
void oracle_harris_response(const uint8_t *img, int w, int h, float *resp) {
    const float s = (float)(1.0 / (4.0 * 3.0 * 255.0)), s2 = 2.0f * s, k = 0.04f;
    std::vector<float> cxx((size_t)w * h), cxy((size_t)w * h), cyy((size_t)w * h);
    auto px = [&](int y, int x) { return (int)img[(size_t)reflect101(y, h) * w + reflect101(x, w)]; };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int dm = px(y - 1, x + 1) - px(y - 1, x - 1), d0 = px(y, x + 1) - px(y, x - 1), dp = px(y + 1, x + 1) - px(y + 1, x - 1);
            const int em = px(y - 1, x - 1) + 2 * px(y - 1, x) + px(y - 1, x + 1), ep = px(y + 1, x - 1) + 2 * px(y + 1, x) + px(y + 1, x + 1);
            // Dx: derivative along x, smoothing [s 2s s] along y ; Dy: smoothing [1 2 1] along x (exact), derivative along y scaled by s
            float dx = s * (float)(dm + dp);
            dx = dx + s2 * (float)d0;
            const float dy = s * (float)(ep - em);
            cxx[(size_t)y * w + x] = dx * dx, cxy[(size_t)y * w + x] = dx * dy, cyy[(size_t)y * w + x] = dy * dy;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float a = 0, b = 0, c = 0;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) {
                    const size_t o = (size_t)reflect101(y + j, h) * w + reflect101(x + i, w);
                    a += cxx[o], b += cxy[o], c += cyy[o];
                }
            const float t = a + c;
            float r = a * c;
            r = r - b * b;
            r = r - (k * t) * t;
            resp[(size_t)y * w + x] = r;
        }
}

// steps 4-5; returns the number of corners, xy / response in selection order
int oracle_good_features(const float *resp, int w, int h, int max_corners, double quality, double min_distance, float *xy, float *out_resp) {
    float mx = -INFINITY;
    for (size_t i = 0; i < (size_t)w * h; ++i) mx = std::max(mx, resp[i]);
    const float thr = (float)(mx * quality);
    std::vector<float> e((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) e[i] = resp[i] > thr ? resp[i] : 0.0f;
    std::vector<size_t> cand;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const float v = e[(size_t)y * w + x];
            if (v == 0.0f) continue;
            float m = v;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) m = std::max(m, e[(size_t)(y + j) * w + x + i]);
            if (v == m) cand.push_back((size_t)y * w + x);
        }
    std::sort(cand.begin(), cand.end(), [&](size_t p, size_t q) { return e[p] > e[q] ? true : (e[p] < e[q] ? false : p > q); });
    int n = 0;
    if (min_distance >= 1) {
        const int cell = (int)std::lround(min_distance), gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<std::vector<std::pair<float, float>>> grid((size_t)gw * gh);
        const double md2 = min_distance * min_distance;
        for (size_t p : cand) {
            const int y = (int)(p / w), x = (int)(p % w), xc = x / cell, yc = y / cell;
            bool good = true;
            for (int yy = std::max(0, yc - 1); yy <= std::min(gh - 1, yc + 1) && good; ++yy)
                for (int xx = std::max(0, xc - 1); xx <= std::min(gw - 1, xc + 1) && good; ++xx)
                    for (const auto &q : grid[(size_t)yy * gw + xx]) {
                        const float dx = x - q.first, dy = y - q.second;
                        if (dx * dx + dy * dy < md2) {
                            good = false;
                            break;
                        }
                    }
            if (!good) continue;
            grid[(size_t)yc * gw + xc].push_back({(float)x, (float)y});
            xy[2 * n] = (float)x, xy[2 * n + 1] = (float)y, out_resp[n] = e[p];
            if (++n >= max_corners && max_corners > 0) break;
        }
    } else {
        for (size_t p : cand) {
            xy[2 * n] = (float)(p % w), xy[2 * n + 1] = (float)(p / w), out_resp[n] = e[p];
            if (++n >= max_corners && max_corners > 0) break;
        }
    }
    return n;
}

} // extern "C"
