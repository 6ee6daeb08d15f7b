// undistort_maps.cpp -- see undistort_maps.h.  All arithmetic is double, in the operation order of the code restated;
// built with -ffp-contract=off.
#include "undistort_maps.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace pvio {

namespace {

constexpr int kInterBits = 5, kInterTab = 1 << kInterBits; // OpenCV INTER_BITS / INTER_TAB_SIZE

// cv::saturate_cast<int>(double) / (float): cvRound = round half to even (the default FP rounding mode)
inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_round(float v) { return (int)std::lrintf(v); }
inline int16_t saturate_short(int v) { return (int16_t)std::min(std::max(v, -32768), 32767); }

// cv::invert of a 3x3 double matrix (n <= 3 takes the explicit cofactor branch whatever the decomposition flag is)
bool cv_invert3(const double S[9], double t[9]) {
    auto s = [&](int r, int c) { return S[3 * r + c]; };
    double d = s(0, 0) * (s(1, 1) * s(2, 2) - s(1, 2) * s(2, 1)) - s(0, 1) * (s(1, 0) * s(2, 2) - s(1, 2) * s(2, 0)) +
               s(0, 2) * (s(1, 0) * s(2, 1) - s(1, 1) * s(2, 0));
    if (d == 0.) return false;
    d = 1. / d;
    t[0] = (s(1, 1) * s(2, 2) - s(1, 2) * s(2, 1)) * d;
    t[1] = (s(0, 2) * s(2, 1) - s(0, 1) * s(2, 2)) * d;
    t[2] = (s(0, 1) * s(1, 2) - s(0, 2) * s(1, 1)) * d;
    t[3] = (s(1, 2) * s(2, 0) - s(1, 0) * s(2, 2)) * d;
    t[4] = (s(0, 0) * s(2, 2) - s(0, 2) * s(2, 0)) * d;
    t[5] = (s(0, 2) * s(1, 0) - s(0, 0) * s(1, 2)) * d;
    t[6] = (s(1, 0) * s(2, 1) - s(1, 1) * s(2, 0)) * d;
    t[7] = (s(0, 1) * s(2, 0) - s(0, 0) * s(2, 1)) * d;
    t[8] = (s(0, 0) * s(1, 1) - s(0, 1) * s(1, 0)) * d;
    return true;
}

// Eigen's fixed-size 3x3 inverse (compute_inverse_size3): cofactors of the first column give the determinant, every entry
// is cofactor * (1 / det).  Sums of three terms follow Eigen's unrolled reduction a0 + (a1 + a2).
matrix<3> eigen_inverse3(const matrix<3> &m) {
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1);
    };
    const double c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    const double det = c0 * m(0, 0) + (c1 * m(1, 0) + c2 * m(2, 0));
    const double invdet = 1.0 / det;
    matrix<3> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(j, i) = cof(i, j) * invdet;
    return r;
}
inline void mat_vec3(const matrix<3> &A, const double v[3], double out[3]) {
    for (int r = 0; r < 3; ++r) out[r] = A(r, 0) * v[0] + (A(r, 1) * v[1] + A(r, 2) * v[2]);
}

} // namespace

FixedRemap cv_undistort_fixed_maps(const float Kf[9], const float *dist, int n_dist, int width, int height) {
    if (width < 1 || height < 1 || n_dist < 4) throw std::invalid_argument("cv_undistort_fixed_maps: bad size / coefficients");
    double A[9];
    for (int i = 0; i < 9; ++i) A[i] = (double)Kf[i];
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = n_dist > 4 ? (double)dist[4] : 0.0;
    const double k4 = 0, k5 = 0, k6 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    const double fx = A[0], fy = A[4], u0 = A[2], v0 = A[5];
    FixedRemap out;
    out.width = width, out.height = height;
    out.xy.resize((size_t)width * height * 2);
    out.frac.resize((size_t)width * height);
    // cv::undistort: row stripes of (1 << 12) / cols rows, the new camera matrix (= K) shifted by the stripe origin
    const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(width, 1)), height);
    for (int y0 = 0; y0 < height; y0 += stripe0) {
        const int stripe = std::min(stripe0, height - y0);
        double Ar[9];
        for (int i = 0; i < 9; ++i) Ar[i] = A[i];
        Ar[5] = v0 - y0;
        double ir[9];
        if (!cv_invert3(Ar, ir)) throw std::invalid_argument("cv_undistort_fixed_maps: singular camera matrix");
        for (int i = 0; i < stripe; ++i) {
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            int16_t *m1 = out.xy.data() + (size_t)(y0 + i) * width * 2;
            uint16_t *m2 = out.frac.data() + (size_t)(y0 + i) * width;
            for (int j = 0; j < width; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
                const double w = 1. / _w, x = _x * w, y = _y * w;
                const double x2 = x * x, y2 = y * y;
                const double r2 = x2 + y2, _2xy = 2 * x * y;
                const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
                const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
                const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
                const double invProj = 1.; // identity tilt: (xd, yd, 1)
                const double u = fx * invProj * xd + u0;
                const double v = fy * invProj * yd + v0;
                const int iu = cv_round(u * kInterTab), iv = cv_round(v * kInterTab);
                m1[j * 2] = (int16_t)(iu >> kInterBits);
                m1[j * 2 + 1] = (int16_t)(iv >> kInterBits);
                m2[j] = (uint16_t)((iv & (kInterTab - 1)) * kInterTab + (iu & (kInterTab - 1)));
            }
        }
    }
    return out;
}

FixedRemap convert_maps_fixed(const float *map_x, const float *map_y, int width, int height) {
    FixedRemap out;
    out.width = width, out.height = height;
    out.xy.resize((size_t)width * height * 2);
    out.frac.resize((size_t)width * height);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        const int ix = cv_round(map_x[i] * (float)kInterTab), iy = cv_round(map_y[i] * (float)kInterTab);
        out.xy[2 * i] = saturate_short(ix >> kInterBits);
        out.xy[2 * i + 1] = saturate_short(iy >> kInterBits);
        out.frac[i] = (uint16_t)((iy & (kInterTab - 1)) * kInterTab + (ix & (kInterTab - 1)));
    }
    return out;
}

ImageUndistorter::ImageUndistorter(size_t width, size_t height, const matrix<3> &K, const std::vector<double> &distort_coeffs, const std::string &model)
    : width_(width), height_(height), K_(K), Kinv_(eigen_inverse3(K)), coeffs_(distort_coeffs), model_(model) {
    if (model_ != "radtan" && model_ != "equidistant") throw std::runtime_error("unknown model: " + model_);
    if (coeffs_.size() < 4) throw std::runtime_error("ImageUndistorter: four distortion coefficients expected");
    std::vector<float> mx(width * height), my(width * height);
    for (size_t v = 0; v < height; ++v)
        for (size_t u = 0; u < width; ++u) {
            vector<2> p;
            p[0] = (double)u, p[1] = (double)v;
            const vector<2> d = distort_pixel(p);
            mx[v * width + u] = (float)d[0], my[v * width + u] = (float)d[1];
        }
    maps_ = convert_maps_fixed(mx.data(), my.data(), (int)width, (int)height);
}

vector<2> ImageUndistorter::distort_pixel(const vector<2> &loc) const {
    const double pix[3] = {loc[0], loc[1], 1.0};
    double n[3];
    mat_vec3(Kinv_, pix, n); // size- and focus-independent coordinates
    const double x = n[0], y = n[1];
    double d[3] = {0.0, 0.0, n[2]};
    const std::vector<double> &D = coeffs_;
    if (model_ == "radtan") {
        const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3];
        const double k3 = D.size() > 4 ? D[4] : 0.0; // the reference reads D[4] unconditionally (image_undistorter.h:68)
        const double r2 = x * x + y * y;
        const double r4 = r2 * r2;
        const double r6 = r4 * r2;
        const double kr = (1.0 + k1 * r2 + k2 * r4 + k3 * r6);
        d[0] = x * kr + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
        d[1] = y * kr + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y);
    } else {
        const double k1 = D[0], k2 = D[1], k3 = D[2], k4 = D[3];
        const double r = std::sqrt(x * x + y * y);
        if (r < 1e-10) return loc;
        const double theta = std::atan(r);
        const double theta2 = theta * theta;
        const double theta4 = theta2 * theta2;
        const double theta6 = theta2 * theta4;
        const double theta8 = theta4 * theta4;
        const double thetad = theta * (1 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
        const double scaling = (r > 1e-8) ? thetad / r : 1.0;
        d[0] = x * scaling;
        d[1] = y * scaling;
    }
    double o[3];
    mat_vec3(K_, d, o);
    vector<2> out;
    out[0] = o[0] / o[2], out[1] = o[1] / o[2]; // hnormalized
    return out;
}

} // namespace pvio
