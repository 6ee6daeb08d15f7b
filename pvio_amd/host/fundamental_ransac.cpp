// fundamental_ransac.cpp -- see fundamental_ransac.h.
#include "fundamental_ransac.h"

#include <algorithm>
#include <cfloat>
#include <cmath>


namespace pvio {
namespace {

struct Mwc { // cv::RNG
    uint64_t state;
    explicit Mwc(uint64_t s) : state(s ? s : 0xffffffffu) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

double det3(const double *m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// real roots of c[0] x^3 + c[1] x^2 + c[2] x + c[3] = 0 in the order of the closed form (three cosines, or the single real root)
int solve_cubic(const double c[4], double x[3]) {
    const double a0 = c[0], pi = 3.14159265358979323846;
    if (a0 == 0) {
        if (c[1] == 0) {
            if (c[2] == 0) return 0;
            x[0] = -c[3] / c[2];
            return 1;
        }
        double d = c[2] * c[2] - 4 * c[1] * c[3];
        if (d < 0) return 0;
        d = std::sqrt(d);
        const double q1 = (-c[2] + d) * 0.5, q2 = (c[2] + d) * -0.5;
        if (std::fabs(q1) > std::fabs(q2)) x[0] = q1 / c[1], x[1] = c[3] / q1;
        else x[0] = q2 / c[1], x[1] = c[3] / q2;
        return d > 0 ? 2 : 1;
    }
    const double a1 = c[1] / a0, a2 = c[2] / a0, a3 = c[3] / a0;
    const double Q = (a1 * a1 - 3 * a2) * (1. / 9), R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54), Qc = Q * Q * Q;
    double d = Qc - R * R;
    if (d > 0) {
        const double theta = std::acos(R / std::sqrt(Qc)), t0 = -2 * std::sqrt(Q), t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
        x[0] = t0 * std::cos(t1) - t2, x[1] = t0 * std::cos(t1 + (2. * pi / 3)) - t2, x[2] = t0 * std::cos(t1 + (4. * pi / 3)) - t2;
        return 3;
    }
    if (d == 0) {
        if (R >= 0) x[0] = -2 * std::pow(R, 1. / 3) - a1 / 3, x[1] = std::pow(R, 1. / 3) - a1 / 3;
        else x[0] = 2 * std::pow(-R, 1. / 3) - a1 / 3, x[1] = -std::pow(-R, 1. / 3) - a1 / 3;
        return 2;
    }
    d = std::sqrt(-d);
    double e = std::pow(d + std::fabs(R), 1. / 3);
    if (R > 0) e = -e;
    x[0] = (e + Q / e) - a1 * (1. / 3);
    return 1;
}

bool last_point_collinear(const float *m, int count) { // the count-th point against every pair of earlier ones
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = m[2 * j] - m[2 * i], dy1 = m[2 * j + 1] - m[2 * i + 1];
        for (int k = 0; k < j; ++k) {
            const double dx2 = m[2 * k] - m[2 * i], dy2 = m[2 * k + 1] - m[2 * i + 1];
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
        }
    }
    return false;
}

// `need`: the hypothesis only matters if it ends with MORE than `need` inliers (the best count so far); once the points that
// are left cannot lift it above that the scoring stops (the value returned is then just some count <= need, the mask is
// not used).  Most hypotheses of a run are poor: they are abandoned after n - need misses instead of after n points.
int count_inliers(int n, const float *p, const float *q, const double *F, double thr2, std::vector<uint8_t> &mask, int need) {
    int good = 0;
    for (int i = 0; i < n; ++i) {
        if (good + (n - i) <= need) return good;
        const double x1 = p[2 * i], y1 = p[2 * i + 1], x2 = q[2 * i], y2 = q[2 * i + 1];
        double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
        const double s2 = 1. / (a * a + b * b), d2 = x2 * a + y2 * b + c;
        a = F[0] * x2 + F[3] * y2 + F[6], b = F[1] * x2 + F[4] * y2 + F[7], c = F[2] * x2 + F[5] * y2 + F[8];
        const double s1 = 1. / (a * a + b * b), d1 = x1 * a + y1 * b + c;
        const float err = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
        const uint8_t in = err <= thr2 ? 1 : 0;
        mask[(size_t)i] = in, good += in;
    }
    return good;
}

int update_iterations(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.), 1.), ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN), denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num), denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

} // namespace

int fundamental_7point(const float p[14], const float q[14], double F[27]) {
    // rows (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1) . f = 0 ; f1, f2 = a basis of the null space
    // The null space comes from a Householder QR of A^T (9 x 7): the last two columns of Q are orthogonal to all seven
    // rows.  (The eigenvectors of A^T A would square the condition number -- pixel coordinates are not normalized here,
    // as in OpenCV's 7-point routine -- and lose half of the digits.)
    double M[9][7]; // A^T, overwritten by R; the reflectors are kept in v[][]
    for (int i = 0; i < 7; ++i) {
        const double x1 = p[2 * i], y1 = p[2 * i + 1], x2 = q[2 * i], y2 = q[2 * i + 1];
        const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        for (int k = 0; k < 9; ++k) M[k][i] = r[k];
    }
    double v[7][9], beta[7];
    for (int c = 0; c < 7; ++c) {
        double nrm = 0;
        for (int k = c; k < 9; ++k) nrm += M[k][c] * M[k][c];
        nrm = std::sqrt(nrm);
        for (int k = 0; k < 9; ++k) v[c][k] = 0;
        if (nrm == 0) {
            beta[c] = 0;
            continue;
        }
        const double alpha = M[c][c] > 0 ? -nrm : nrm;
        for (int k = c; k < 9; ++k) v[c][k] = M[k][c];
        v[c][c] -= alpha;
        double vv = 0;
        for (int k = c; k < 9; ++k) vv += v[c][k] * v[c][k];
        beta[c] = vv > 0 ? 2.0 / vv : 0.0;
        for (int j = c; j < 7; ++j) { // apply H_c to the remaining columns
            double d = 0;
            for (int k = c; k < 9; ++k) d += v[c][k] * M[k][j];
            d *= beta[c];
            for (int k = c; k < 9; ++k) M[k][j] -= d * v[c][k];
        }
    }
    double f1[9], f2[9];
    for (int which = 0; which < 2; ++which) { // Q e_7, Q e_8 with Q = H_0 H_1 ... H_6
        double e[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        e[7 + which] = 1.0;
        for (int c = 6; c >= 0; --c) {
            double d = 0;
            for (int k = c; k < 9; ++k) d += v[c][k] * e[k];
            d *= beta[c];
            for (int k = c; k < 9; ++k) e[k] -= d * v[c][k];
        }
        for (int k = 0; k < 9; ++k) (which == 0 ? f1 : f2)[k] = e[k];
    }
    // det(f2 + l (f1 - f2)) = c3 l^3 + c2 l^2 + c1 l + c0
    double D[9], tmp[9], c[4];
    for (int k = 0; k < 9; ++k) D[k] = f1[k] - f2[k];
    c[3] = det3(f2), c[0] = det3(D), c[2] = 0, c[1] = 0;
    for (int row = 0; row < 3; ++row) {
        for (int k = 0; k < 9; ++k) tmp[k] = f2[k];
        for (int k = 0; k < 3; ++k) tmp[3 * row + k] = D[3 * row + k];
        c[2] += det3(tmp); // linear term: one row of D
        for (int k = 0; k < 9; ++k) tmp[k] = D[k];
        for (int k = 0; k < 3; ++k) tmp[3 * row + k] = f2[3 * row + k];
        c[1] += det3(tmp); // quadratic term: one row of f2
    }
    double roots[3];
    const int n = solve_cubic(c, roots);
    int m = 0;
    for (int k = 0; k < n; ++k) {
        const double l = roots[k];
        double *Fk = F + 9 * m, nrm = 0;
        for (int e = 0; e < 9; ++e) Fk[e] = f2[e] + l * D[e], nrm += Fk[e] * Fk[e];
        if (!(nrm > 0) || !std::isfinite(nrm)) continue;
        const double s = std::fabs(Fk[8]) > DBL_EPSILON ? 1.0 / Fk[8] : 1.0 / std::sqrt(nrm); // F[2][2] = 1 where possible
        for (int e = 0; e < 9; ++e) Fk[e] *= s;
        ++m;
    }
    return m;
}

int find_fundamental_ransac(int n, const float *p, const float *q, double threshold, double confidence, std::vector<uint8_t> &mask, double F_out[9], int max_iterations) {
    constexpr int kModel = 7;
    mask.assign((size_t)std::max(n, 0), 0);
    if (n < kModel) return 0;
    Mwc rng((uint64_t)-1);
    const double thr2 = threshold * threshold;
    std::vector<uint8_t> cur((size_t)n), best((size_t)n, 0);
    int niters = max_iterations, max_good = 0;
    double bestF[9] = {0};
    float sp[14], sq[14];
    if (n == kModel) niters = 1;
    for (int iter = 0; iter < niters; ++iter) {
        if (n > kModel) { // seven distinct points whose last one is not collinear with two earlier ones (either image)
            int idx[kModel], i = 0, attempts = 0;
            const int max_attempts = 10000;
            for (; attempts < max_attempts; ++attempts) {
                for (i = 0; i < kModel && attempts < max_attempts;) {
                    const int c = idx[i] = rng.uniform(0, n);
                    int j = 0;
                    for (; j < i; ++j)
                        if (c == idx[j]) break;
                    if (j < i) continue; // drawn before: draw again
                    sp[2 * i] = p[2 * c], sp[2 * i + 1] = p[2 * c + 1], sq[2 * i] = q[2 * c], sq[2 * i + 1] = q[2 * c + 1];
                    ++i;
                }
                if (i == kModel && (last_point_collinear(sp, i) || last_point_collinear(sq, i))) continue;
                break;
            }
            if (!(i == kModel && attempts < max_attempts)) {
                if (iter == 0) return 0;
                break;
            }
        } else {
            std::copy(p, p + 14, sp), std::copy(q, q + 14, sq);
        }
        double F[27];
        const int nm = fundamental_7point(sp, sq, F);
        for (int m = 0; m < nm; ++m) {
            const int good = count_inliers(n, p, q, F + 9 * m, thr2, cur, std::max(max_good, kModel - 1));
            if (good > std::max(max_good, kModel - 1)) {
                std::swap(cur, best);
                std::copy(F + 9 * m, F + 9 * m + 9, bestF);
                max_good = good;
                niters = update_iterations(confidence, (double)(n - good) / n, kModel, niters);
            }
        }
    }
    if (max_good > 0) {
        mask = best;
        if (F_out) std::copy(bestF, bestF + 9, F_out);
    }
    return max_good;
}

} // namespace pvio
