// oracle_klt.cpp -- CPU oracle for the image front end.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md): checker for the HIP KLT path, never shipped or called by it.
//
// Restates what the reference obtains from OpenCV at
//   OpenCvImage::preprocess       pvio-extra/src/pvio/extra/opencv_image.cpp:138-145   CLAHE(6.0, 8x8) in place, then
//                                 buildOpticalFlowPyramid(img, pyr, Size(21,21), 3, withDerivatives=true)
//   OpenCvImage::track_keypoints  opencv_image.cpp:88-109   calcOpticalFlowPyrLK(win 21x21, maxLevel 3,
//                                 TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW) + the 20-px border kill
// OpenCV is an unpinned find_package (depends/CMakeLists.txt:9) and is NOT in /root/reference, so these are
// restatements of the published algorithms (imgproc/clahe.cpp, imgproc/pyramids.cpp, video/lkpyramid.cpp: scalar,
// non-SIMD code paths) -- SURVEY.md App. C.  PARITY UNPINNED: no OpenCV here, no reference tests; pinned only by
// invariants (tests/test_oracle_klt.py).
//
// THE ORDER OF THE FLOAT SUMS (round 4).  LK accumulates 441 products per window into float accumulators (A11, A12, A22 once per level,
// b1, b2 every iteration).  OpenCV itself has no single order: its scalar LKTrackerInvoker adds left to right, row by row; its SSE2 / NEON /
// AVX paths add four or eight lanes side by side and fold them at the end -- so even the real reference is reproducible only up to float
// rounding across builds.  This oracle therefore DEFINES the order the product implements and is held to it bit for bit
// (tests/test_gpu_klt.py, tests/test_chain_parity.py: positions identical, not "within 1e-3 px"):
//   * the window is cut into 63 runs of 7 pixels (run l = row l / 3, columns 7 (l % 3) .. 7 (l % 3) + 6); a run is summed left to right
//     from 0.0f -- a 64th, empty run holds 0.0f;
//   * the 64 run sums are folded by a fixed tree: within each group of 16 runs  s += s[i ^ 1];  s += s[i ^ 2];  s += s[mirror in 8];
//     s += s[mirror in 16]  (every run of the group then holds the group's sum), and the four group sums are added as ((g0 + g1) + g2) + g3.
// That is pvio_amd/csrc/klt.hip: lane = run, wave_sum_f = the tree (DPP quad_perm / row_half_mirror / row_mirror + four lane reads).
// oracle_klt_track_scalar_order keeps OpenCV's scalar left-to-right order; the two agree in every status byte and to <= 1e-3 px on textured
// windows (tests/test_oracle_klt.py), which is what "the same tracker" can mean across summation orders.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline int reflect101(int p, int len) { // cv::borderInterpolate(BORDER_REFLECT_101)
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}
inline int cv_round(float v) { return (int)lrintf(v); } // round half to even, like cvRound
inline int cv_floor(float v) { return (int)floorf(v); }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

constexpr int kWin = 21;        // Size(21, 21)
constexpr int kMaxLevel = 3;    // level_num()
constexpr int kMaxCount = 30;
constexpr double kEps2 = 0.01 * 0.01; // criteria.epsilon (a double, 0.01: opencv_image.cpp:103) is squared by calcOpticalFlowPyrLK: 1.0000000000000002e-4
constexpr float kMinEig = 1e-4f;

// the defined fold of 64 run sums (see the header): returns what every lane of the wave holds after klt.hip's wave_sum_f
constexpr int kRun = 7, kRuns = 64;
inline int run_of(int x, int y) { return 3 * y + x / kRun; }
inline float fold_runs(const float *run) {
    float s[kRuns], t[kRuns];
    for (int i = 0; i < kRuns; ++i) s[i] = run[i];
    for (int i = 0; i < kRuns; ++i) t[i] = s[i] + s[i ^ 1];
    for (int i = 0; i < kRuns; ++i) s[i] = t[i] + t[i ^ 2];
    for (int i = 0; i < kRuns; ++i) t[i] = s[i] + s[(i & ~7) | (7 - (i & 7))];
    for (int i = 0; i < kRuns; ++i) s[i] = t[i] + t[(i & ~15) | (15 - (i & 15))];
    return ((s[0] + s[16]) + s[32]) + s[48];
}
// one accumulator: either 64 run sums folded by the tree (the defined order) or one running sum (OpenCV's scalar order)
struct Acc {
    bool runs;
    float r[kRuns];
    float scalar;
    explicit Acc(bool by_runs) : runs(by_runs), scalar(0.f) {
        for (float &v : r) v = 0.f;
    }
    inline void add(int x, int y, float v) {
        if (runs) r[run_of(x, y)] += v;
        else scalar += v;
    }
    inline float total() const { return runs ? fold_runs(r) : scalar; }
};

} // namespace

extern "C" {

// cv::CLAHE::apply for CV_8UC1 (clipLimit, tiles x tiles), src == dst allowed
void oracle_clahe(const uint8_t *src, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y, uint8_t *dst, int dst_stride) {
    // pad to a multiple of the tile grid with BORDER_REFLECT_101 (right / bottom only)
    int ew = w, eh = h;
    if (w % tiles_x != 0 || h % tiles_y != 0) {
        ew = w + (tiles_x - (w % tiles_x));
        eh = h + (tiles_y - (h % tiles_y));
    }
    std::vector<uint8_t> ext((size_t)ew * eh);
    for (int y = 0; y < eh; ++y)
        for (int x = 0; x < ew; ++x) ext[(size_t)y * ew + x] = src[(size_t)reflect101(y, h) * stride + reflect101(x, w)];
    const int tw = ew / tiles_x, th = eh / tiles_y, tile_total = tw * th;
    const float lut_scale = 255.0f / tile_total;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_total / 256);
        if (clip < 1) clip = 1;
    }
    std::vector<uint8_t> lut((size_t)tiles_x * tiles_y * 256);
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) {
            int hist[256] = {0};
            for (int y = 0; y < th; ++y)
                for (int x = 0; x < tw; ++x) hist[ext[(size_t)(ty * th + y) * ew + tx * tw + x]]++;
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; ++i)
                    if (hist[i] > clip) clipped += hist[i] - clip, hist[i] = clip;
                int batch = clipped / 256, residual = clipped - batch * 256;
                for (int i = 0; i < 256; ++i) hist[i] += batch;
                if (residual != 0) {
                    int step = 256 / residual;
                    if (step < 1) step = 1;
                    for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            int sum = 0;
            uint8_t *tl = &lut[(size_t)(ty * tiles_x + tx) * 256];
            for (int i = 0; i < 256; ++i) {
                sum += hist[i];
                tl[i] = sat_u8(cv_round(sum * lut_scale));
            }
        }
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    std::vector<uint8_t> out((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        float tyf = y * inv_th - 0.5f;
        int ty1 = cv_floor(tyf), ty2 = ty1 + 1;
        float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        for (int x = 0; x < w; ++x) {
            float txf = x * inv_tw - 0.5f;
            int tx1 = cv_floor(txf), tx2 = tx1 + 1;
            float xa = txf - tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            const int v = src[(size_t)y * stride + x];
            const uint8_t *p1 = &lut[(size_t)(ty1 * tiles_x) * 256], *p2 = &lut[(size_t)(ty2 * tiles_x) * 256];
            const int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
            float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
            out[(size_t)y * w + x] = sat_u8(cv_round(res));
        }
    }
    for (int y = 0; y < h; ++y) std::memcpy(dst + (size_t)y * dst_stride, &out[(size_t)y * w], w);
}

// cv::pyrDown for CV_8UC1: separable [1 4 6 4 1], (sum + 128) >> 8, BORDER_REFLECT_101; dst is ((w+1)/2, (h+1)/2)
void oracle_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            int sum = 0;
            for (int j = -2; j <= 2; ++j) {
                const uint8_t *row = src + (size_t)reflect101(2 * y + j, h) * w;
                int rs = 0;
                for (int i = -2; i <= 2; ++i) rs += k[i + 2] * row[reflect101(2 * x + i, w)];
                sum += k[j + 2] * rs;
            }
            dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
        }
}

// calcSharrDeriv (lkpyramid.cpp): dst[2 * (y * w + x)] = dI/dx, +1 = dI/dy, int16, reflect-101 borders
void oracle_scharr(const uint8_t *src, int w, int h, int16_t *dst) {
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * w;
        const uint8_t *r1 = src + (size_t)y * w;
        const uint8_t *r2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * w;
        for (int x = 0; x < w; ++x) {
            auto t0 = [&](int xx) { xx = xx < 0 ? (w > 1 ? 1 : 0) : xx >= w ? (w > 1 ? w - 2 : 0) : xx; return (r0[xx] + r2[xx]) * 3 + r1[xx] * 10; };
            auto t1 = [&](int xx) { xx = xx < 0 ? (w > 1 ? 1 : 0) : xx >= w ? (w > 1 ? w - 2 : 0) : xx; return (int)r2[xx] - (int)r0[xx]; };
            dst[2 * ((size_t)y * w + x)] = (int16_t)(t0(x + 1) - t0(x - 1));
            dst[2 * ((size_t)y * w + x) + 1] = (int16_t)((t1(x + 1) + t1(x - 1)) * 3 + t1(x) * 10);
        }
    }
}

// level sizes of buildOpticalFlowPyramid(maxLevel = 3); returns the number of levels actually built
int oracle_pyramid_sizes(int w, int h, int *ws, int *hs) {
    int n = 0;
    for (int l = 0; l <= kMaxLevel; ++l) {
        if (l > 0) {
            w = (w + 1) / 2, h = (h + 1) / 2;
            if (w <= kWin || h <= kWin) break; // lkpyramid.cpp: the pyramid stops when a level is not larger than the window
        }
        ws[n] = w, hs[n] = h, ++n;
    }
    return n;
}

struct Level {
    int w, h;
    const uint8_t *img;
    const int16_t *drv;
};
// image sample with the pyramid's physical border semantics: BORDER_REFLECT_101 for pixels, zeros for derivatives
static inline int pix(const Level &L, int x, int y) { return L.img[(size_t)reflect101(y, L.h) * L.w + reflect101(x, L.w)]; }
static inline int der(const Level &L, int x, int y, int c) {
    if (x < 0 || y < 0 || x >= L.w || y >= L.h) return 0;
    return L.drv[2 * ((size_t)y * L.w + x) + c];
}

// calcOpticalFlowPyrLK (LKTrackerInvoker, scalar path) over prebuilt pyramids + the reference's 20-px border kill.
// imgs/drvs: per level pointers (level 0 first); next_xy in/out; status out.
static void klt_track_impl(bool by_runs, int n_levels, const int *ws, const int *hs, const uint8_t *const *prev_img, const int16_t *const *prev_drv,
                           const uint8_t *const *next_img, int n, const float *prev_xy, float *next_xy, uint8_t *status) {
    const float half = (kWin - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    for (int p = 0; p < n; ++p) status[p] = 1;
    std::vector<int16_t> Iwin(kWin * kWin), dIwin(kWin * kWin * 2);
    for (int level = n_levels - 1; level >= 0; --level) {
        Level I{ws[level], hs[level], prev_img[level], prev_drv[level]};
        Level J{ws[level], hs[level], next_img[level], nullptr};
        for (int p = 0; p < n; ++p) {
            float px = prev_xy[2 * p] * (float)(1. / (1 << level)), py = prev_xy[2 * p + 1] * (float)(1. / (1 << level));
            float nx, ny;
            if (level == n_levels - 1) {
                nx = next_xy[2 * p] * (float)(1. / (1 << level)), ny = next_xy[2 * p + 1] * (float)(1. / (1 << level)); // USE_INITIAL_FLOW
            } else {
                nx = next_xy[2 * p] * 2.f, ny = next_xy[2 * p + 1] * 2.f;
            }
            next_xy[2 * p] = nx, next_xy[2 * p + 1] = ny;
            px -= half, py -= half;
            int ipx = cv_floor(px), ipy = cv_floor(py);
            if (ipx < -kWin || ipx >= I.w || ipy < -kWin || ipy >= I.h) {
                if (level == 0) status[p] = 0;
                continue;
            }
            float a = px - ipx, b = py - ipy;
            const int W_BITS = 14;
            int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            Acc accA11(by_runs), accA12(by_runs), accA22(by_runs);
            for (int y = 0; y < kWin; ++y)
                for (int x = 0; x < kWin; ++x) {
                    const int X = ipx + x, Y = ipy + y;
                    int ival = (pix(I, X, Y) * iw00 + pix(I, X + 1, Y) * iw01 + pix(I, X, Y + 1) * iw10 + pix(I, X + 1, Y + 1) * iw11 + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                    int ixval = (der(I, X, Y, 0) * iw00 + der(I, X + 1, Y, 0) * iw01 + der(I, X, Y + 1, 0) * iw10 + der(I, X + 1, Y + 1, 0) * iw11 + (1 << (W_BITS - 1))) >> W_BITS;
                    int iyval = (der(I, X, Y, 1) * iw00 + der(I, X + 1, Y, 1) * iw01 + der(I, X, Y + 1, 1) * iw10 + der(I, X + 1, Y + 1, 1) * iw11 + (1 << (W_BITS - 1))) >> W_BITS;
                    Iwin[y * kWin + x] = (int16_t)ival;
                    dIwin[2 * (y * kWin + x)] = (int16_t)ixval;
                    dIwin[2 * (y * kWin + x) + 1] = (int16_t)iyval;
                    accA11.add(x, y, (float)(ixval * ixval));
                    accA12.add(x, y, (float)(ixval * iyval));
                    accA22.add(x, y, (float)(iyval * iyval));
                }
            float A11 = accA11.total() * FLT_SCALE, A12 = accA12.total() * FLT_SCALE, A22 = accA22.total() * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * kWin * kWin);
            if (minEig < kMinEig || D < 1.1920929e-07f /* FLT_EPSILON */) {
                if (level == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half, ny -= half;
            float pdx = 0, pdy = 0;
            for (int j = 0; j < kMaxCount; ++j) {
                int inx = cv_floor(nx), iny = cv_floor(ny);
                if (inx < -kWin || inx >= J.w || iny < -kWin || iny >= J.h) {
                    if (level == 0) status[p] = 0;
                    break;
                }
                a = nx - inx, b = ny - iny;
                iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                Acc accb1(by_runs), accb2(by_runs);
                for (int y = 0; y < kWin; ++y)
                    for (int x = 0; x < kWin; ++x) {
                        const int X = inx + x, Y = iny + y;
                        int diff = ((pix(J, X, Y) * iw00 + pix(J, X + 1, Y) * iw01 + pix(J, X, Y + 1) * iw10 + pix(J, X + 1, Y + 1) * iw11 + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5)) - Iwin[y * kWin + x];
                        accb1.add(x, y, (float)(diff * dIwin[2 * (y * kWin + x)]));
                        accb2.add(x, y, (float)(diff * dIwin[2 * (y * kWin + x) + 1]));
                    }
                float b1 = accb1.total() * FLT_SCALE, b2 = accb2.total() * FLT_SCALE;
                float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                nx += dx, ny += dy;
                next_xy[2 * p] = nx + half, next_xy[2 * p + 1] = ny + half;
                // both tests in double, as OpenCV makes them: delta.ddot(delta) <= criteria.epsilon; std::abs(delta.x + prevDelta.x) < 0.01
                // (the sum is a float sum, the comparison is against the double literal)
                if ((double)dx * (double)dx + (double)dy * (double)dy <= kEps2) break;
                if (j > 0 && (double)std::fabs(dx + pdx) < 0.01 && (double)std::fabs(dy + pdy) < 0.01) {
                    next_xy[2 * p] -= dx * 0.5f, next_xy[2 * p + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx, pdy = dy;
            }
            if (status[p] && level == 0) { // final-position check done together with the error measure (err is requested)
                float fx = next_xy[2 * p] - half, fy = next_xy[2 * p + 1] - half;
                int ix = cv_floor(fx), iy = cv_floor(fy);
                if (ix < -kWin || ix >= J.w || iy < -kWin || iy >= J.h) status[p] = 0;
            }
        }
    }
    // opencv_image.cpp:104-109: tracks within 20 px of the border of the full-resolution image are dropped
    for (int p = 0; p < n; ++p)
        if (next_xy[2 * p] < 20 || next_xy[2 * p] >= ws[0] - 20 || next_xy[2 * p + 1] < 20 || next_xy[2 * p + 1] >= hs[0] - 20) status[p] = 0;
}

// the defined order of the float sums (header): what the parity tests hold the HIP kernel to, bit for bit
void oracle_klt_track(int n_levels, const int *ws, const int *hs, const uint8_t *const *prev_img, const int16_t *const *prev_drv,
                      const uint8_t *const *next_img, int n, const float *prev_xy, float *next_xy, uint8_t *status) {
    klt_track_impl(true, n_levels, ws, hs, prev_img, prev_drv, next_img, n, prev_xy, next_xy, status);
}
// OpenCV's scalar LKTrackerInvoker order: one running sum, left to right, row by row
void oracle_klt_track_scalar_order(int n_levels, const int *ws, const int *hs, const uint8_t *const *prev_img, const int16_t *const *prev_drv,
                                   const uint8_t *const *next_img, int n, const float *prev_xy, float *next_xy, uint8_t *status) {
    klt_track_impl(false, n_levels, ws, hs, prev_img, prev_drv, next_img, n, prev_xy, next_xy, status);
}

} // extern "C"
