// oracle_ransac.cpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into or called by the product).
//
// cv::findFundamentalMat(p, q, cv::FM_RANSAC, 1.0, 0.99, mask) as the reference calls it
// (pvio-extra/src/pvio/extra/opencv_image.cpp:123).  OpenCV is a third-party dependency (unpinned find_package,
// pvio-extra/depends/CMakeLists.txt) that is NOT in /root/reference: the algorithm is restated from the published one
// (calib3d: RANSACPointSetRegistrator::run / getSubset, FMEstimatorCallback::checkSubset / computeError, run7Point,
// RANSACUpdateNumIters, cv::solveCubic, cv::RNG) -- PARITY UNPINNED.  No local-optimisation / refit step: FM_RANSAC returns the
// inliers of the best minimal-sample model (the 8-point refit only changes the returned matrix, which the reference discards).
//
// Written independently of pvio_amd/host/fundamental_ransac.cpp so that the two can check each other: the null space of the
// 7 x 9 system comes from a one-sided Jacobi SVD (as in OpenCV's JacobiSVDImpl) completed to a full basis, not from a
// Householder QR; the cubic's coefficients are the spelled-out cofactor expansions; every hypothesis is scored over all points.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

struct CvRng { // cv::RNG: multiply-with-carry, operator unsigned() and uniform(int, int)
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// cv::solveCubic, coefficients highest power first; returns the number of real roots
int solve_cubic(const double c[4], double r[3]) {
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    double x0 = 0, x1 = 0, x2 = 0;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else x0 = -a3 / a2, n = 1;
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = std::sqrt(d);
                double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (std::fabs(q1) > std::fabs(q2)) x0 = q1 / a1, x1 = a3 / q1;
                else x0 = q2 / a1, x1 = a3 / q2;
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0, a1 *= a0, a2 *= a0, a3 *= a0;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9), R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54), Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            const double theta = std::acos(R / std::sqrt(Qcubed)), sqrtQ = std::sqrt(Q);
            const double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * std::cos(t1) - t2, x1 = t0 * std::cos(t1 + (2. * M_PI / 3)) - t2, x2 = t0 * std::cos(t1 + (4. * M_PI / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) x0 = -2 * std::cbrt(R) - a1 / 3, x1 = std::cbrt(R) - a1 / 3;
            else x0 = 2 * std::cbrt(-R) - a1 / 3, x1 = -std::cbrt(-R) - a1 / 3;
            n = 2;
        } else {
            d = std::sqrt(-d);
            double e = std::cbrt(d + std::fabs(R));
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    r[0] = x0, r[1] = x1, r[2] = x2;
    return n;
}

// One-sided Jacobi (Hestenes) on the columns of M (9 x 7): afterwards the columns are mutually orthogonal; normalized, they
// span the row space of the 7 x 9 system.  Two more unit vectors orthogonal to all of them complete the basis = the null space.
void null_space_7x9(const double A[7][9], double f1[9], double f2[9]) {
    double M[9][7];
    for (int i = 0; i < 7; ++i)
        for (int k = 0; k < 9; ++k) M[k][i] = A[i][k];
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool changed = false;
        for (int i = 0; i < 6; ++i)
            for (int j = i + 1; j < 7; ++j) {
                double a = 0, b = 0, p = 0;
                for (int k = 0; k < 9; ++k) a += M[k][i] * M[k][i], b += M[k][j] * M[k][j], p += M[k][i] * M[k][j];
                if (std::fabs(p) <= DBL_EPSILON * std::sqrt(a * b)) continue;
                changed = true;
                const double beta = a - b, gamma = std::hypot(2 * p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma), c = p / (gamma * s);
                } else {
                    c = std::sqrt((gamma + beta) / (2 * gamma)), s = p / (gamma * c);
                }
                for (int k = 0; k < 9; ++k) {
                    const double t0 = c * M[k][i] + s * M[k][j], t1 = -s * M[k][i] + c * M[k][j];
                    M[k][i] = t0, M[k][j] = t1;
                }
            }
        if (!changed) break;
    }
    double U[9][9]; // rows: orthonormal vectors found so far
    int nu = 0;
    double smax = 0;
    double nrm[7];
    for (int i = 0; i < 7; ++i) {
        double s = 0;
        for (int k = 0; k < 9; ++k) s += M[k][i] * M[k][i];
        nrm[i] = std::sqrt(s), smax = std::max(smax, nrm[i]);
    }
    for (int i = 0; i < 7; ++i) {
        if (!(nrm[i] > smax * 1e-14)) continue; // a rank-deficient sample: the completion below fills in
        for (int k = 0; k < 9; ++k) U[nu][k] = M[k][i] / nrm[i];
        ++nu;
    }
    // completion: the coordinate vector with the largest part outside the span, orthogonalized twice, until nine rows exist;
    // the LAST two are the basis (f1, f2) -- when the sample has rank 7 they are exactly the null space
    while (nu < 9) {
        int best = 0;
        double best_res = -1;
        for (int e = 0; e < 9; ++e) {
            double res = 1.0;
            for (int u = 0; u < nu; ++u) res -= U[u][e] * U[u][e];
            if (res > best_res) best_res = res, best = e;
        }
        double w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        w[best] = 1.0;
        for (int pass = 0; pass < 2; ++pass)
            for (int u = 0; u < nu; ++u) {
                double d = 0;
                for (int k = 0; k < 9; ++k) d += U[u][k] * w[k];
                for (int k = 0; k < 9; ++k) w[k] -= d * U[u][k];
            }
        double s = 0;
        for (int k = 0; k < 9; ++k) s += w[k] * w[k];
        s = 1.0 / std::sqrt(s);
        for (int k = 0; k < 9; ++k) U[nu][k] = w[k] * s;
        ++nu;
    }
    for (int k = 0; k < 9; ++k) f1[k] = U[7][k], f2[k] = U[8][k];
}

// run7Point: up to three matrices (row-major 3 x 3 each), F(3,3) = 1 where possible
int seven_point(const float *m1, const float *m2, double *fmatrix) {
    double A[7][9];
    for (int i = 0; i < 7; ++i) {
        const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
        A[i][0] = x1 * x0, A[i][1] = x1 * y0, A[i][2] = x1, A[i][3] = y1 * x0, A[i][4] = y1 * y0, A[i][5] = y1, A[i][6] = x0, A[i][7] = y0, A[i][8] = 1;
    }
    double f1[9], f2[9], c[4], r[3];
    null_space_7x9(A, f1, f2);
    // f ~ lambda f1 + (1 - lambda) f2 ; det(f) = 0 is a cubic in lambda
    for (int i = 0; i < 9; ++i) f1[i] -= f2[i];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
           f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) + f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7], t1 = f1[3] * f1[8] - f1[5] * f1[6], t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
           f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) + f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    const int n = solve_cubic(c, r);
    if (n < 1 || n > 3) return n < 0 ? 0 : n;
    for (int k = 0; k < n; ++k, fmatrix += 9) {
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (std::fabs(s) > DBL_EPSILON) {
            mu = 1. / s, lambda *= mu;
            fmatrix[8] = 1.;
        } else {
            fmatrix[8] = 0.;
        }
        for (int i = 0; i < 8; ++i) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

// FMEstimatorCallback::checkSubset: the last of `count` points against every pair of earlier ones
bool have_collinear_points(const float *m, int count) {
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

// FMEstimatorCallback::computeError (float) + findInliers
int find_inliers(int n, const float *m1, const float *m2, const double *F, double thresh2, uint8_t *mask) {
    int good = 0;
    for (int i = 0; i < n; ++i) {
        double a, b, c, d1, d2, s1, s2;
        a = F[0] * m1[2 * i] + F[1] * m1[2 * i + 1] + F[2];
        b = F[3] * m1[2 * i] + F[4] * m1[2 * i + 1] + F[5];
        c = F[6] * m1[2 * i] + F[7] * m1[2 * i + 1] + F[8];
        s2 = 1. / (a * a + b * b);
        d2 = m2[2 * i] * a + m2[2 * i + 1] * b + c;
        a = F[0] * m2[2 * i] + F[3] * m2[2 * i + 1] + F[6];
        b = F[1] * m2[2 * i] + F[4] * m2[2 * i + 1] + F[7];
        c = F[2] * m2[2 * i] + F[5] * m2[2 * i + 1] + F[8];
        s1 = 1. / (a * a + b * b);
        d1 = m1[2 * i] * a + m1[2 * i + 1] * b + c;
        const float err = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
        const int f = err <= thresh2;
        mask[i] = (uint8_t)f, good += f;
    }
    return good;
}

int update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = std::max(p, 0.), p = std::min(p, 1.);
    ep = std::max(ep, 0.), ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN), denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num), denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lround(num / denom);
}


// ---- the DEFINED arithmetic of the seven-point step (round 5) -------------------------------------------------------------------------------------
// RANSAC keeps the first hypothesis that reaches the best inlier count, a correspondence whose error sits on the threshold moves a count by one, and two
// hypotheses tie often enough that a long sequence meets it (tests/golden/ransac_ties.npz): the independent restatement above and the product then keep
// DIFFERENT hypotheses -- both legitimate, nothing in the reference says which one OpenCV's own rounding would keep -- and a sequence-level comparison
// ends there.  As for the LK sums (oracle_klt.cpp), the order of operations is therefore DEFINED, here, and the kernel is held to it bit for bit
// (pvio_amd/csrc/pv_fundamental.h implements the same sequence; tests/test_host_ransac.py):
//   null space   Householder QR of A^T (9 x 7), reflector c: v = column below the diagonal, v_c -= alpha with alpha = -sign(M_cc) |column|, beta = 2 / v.v,
//                applied to columns c..6; the basis is Q e_7, Q e_8 (reflectors applied in reverse order to the unit vectors)
//   cubic        det(f2 + l (f1 - f2)): c3 = det f2, c0 = det D, c2 / c1 = sums over the rows of the determinants with one row exchanged
//   roots        Q, R of the normalized cubic as in cv::solveCubic; three real roots: c = cos(theta / 3) by 64 bisection steps on 4 c^3 - 3 c = R / sqrt(Q^3) in
//                [1/2, 1], the other two from the angle-sum identities (order k = 0, 1, 2 of the closed form); one real root: cube root by 12 Newton steps from a
//                power of two; only +, -, *, /, sqrt (correctly rounded everywhere), no FMA contraction (-ffp-contract=off here, a pragma there)
//   scaling      F = f2 + l D, times 1 / F[8] when |F[8]| > DBL_EPSILON, else 1 / |F|
// The entry points above stay what the ALGORITHM is checked against (another null-space method, libm's closed form), at a tolerance.
namespace defined {

double det3(const double *m) { return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]); }

double cube_root(double x) {
    if (!(x > 0)) return 0.0;
    int e;
    (void)std::frexp(x, &e);
    const int k = e >= 0 ? e / 3 : -((-e + 2) / 3);
    double y = std::ldexp(1.0, k);
    for (int it = 0; it < 12; ++it) y = (2.0 * y + x / (y * y)) * (1.0 / 3.0);
    return y;
}

int cubic_roots(const double c[4], double x[3]) {
    const double a0 = c[0];
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
        const double r = R / std::sqrt(Qc), t0 = -2 * std::sqrt(Q), t2 = a1 * (1. / 3);
        double lo = 0.5, hi = 1.0;
        for (int it = 0; it < 64; ++it) {
            const double mid = 0.5 * (lo + hi);
            if ((4.0 * mid * mid - 3.0) * mid < r) lo = mid;
            else hi = mid;
        }
        const double cs = 0.5 * (lo + hi), sn = std::sqrt((1.0 - cs) * (1.0 + cs)), half_sqrt3 = 0.86602540378443864676;
        x[0] = t0 * cs - t2, x[1] = t0 * (-0.5 * cs - sn * half_sqrt3) - t2, x[2] = t0 * (-0.5 * cs + sn * half_sqrt3) - t2;
        return 3;
    }
    if (d == 0) {
        if (R >= 0) x[0] = -2 * cube_root(R) - a1 / 3, x[1] = cube_root(R) - a1 / 3;
        else x[0] = 2 * cube_root(-R) - a1 / 3, x[1] = -cube_root(-R) - a1 / 3;
        return 2;
    }
    d = std::sqrt(-d);
    double e = cube_root(d + std::fabs(R));
    if (R > 0) e = -e;
    x[0] = (e + Q / e) - a1 * (1. / 3);
    return 1;
}

int seven_point(const float *p, const float *q, double *F) {
    double M[9][7];
    for (int i = 0; i < 7; ++i) {
        const double x1 = p[2 * i], y1 = p[2 * i + 1], x2 = q[2 * i], y2 = q[2 * i + 1];
        const double row[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        for (int k = 0; k < 9; ++k) M[k][i] = row[k];
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
        for (int j = c; j < 7; ++j) {
            double d = 0;
            for (int k = c; k < 9; ++k) d += v[c][k] * M[k][j];
            d *= beta[c];
            for (int k = c; k < 9; ++k) M[k][j] -= d * v[c][k];
        }
    }
    double f1[9], f2[9];
    for (int which = 0; which < 2; ++which) {
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
    double D[9], tmp[9], c[4];
    for (int k = 0; k < 9; ++k) D[k] = f1[k] - f2[k];
    c[3] = det3(f2), c[0] = det3(D), c[2] = 0, c[1] = 0;
    for (int row = 0; row < 3; ++row) {
        for (int k = 0; k < 9; ++k) tmp[k] = f2[k];
        for (int k = 0; k < 3; ++k) tmp[3 * row + k] = D[3 * row + k];
        c[2] += det3(tmp);
        for (int k = 0; k < 9; ++k) tmp[k] = D[k];
        for (int k = 0; k < 3; ++k) tmp[3 * row + k] = f2[3 * row + k];
        c[1] += det3(tmp);
    }
    double roots[3];
    const int n = cubic_roots(c, roots);
    int m = 0;
    for (int k = 0; k < n; ++k) {
        const double l = roots[k];
        double *Fk = F + 9 * m, nrm = 0;
        for (int e = 0; e < 9; ++e) Fk[e] = f2[e] + l * D[e], nrm += Fk[e] * Fk[e];
        if (!(nrm > 0) || !std::isfinite(nrm)) continue;
        const double s = std::fabs(Fk[8]) > DBL_EPSILON ? 1.0 / Fk[8] : 1.0 / std::sqrt(nrm);
        for (int e = 0; e < 9; ++e) Fk[e] *= s;
        ++m;
    }
    return m;
}

} // namespace defined

} // namespace

extern "C" {

int32_t oracle_seven_point(const float *p, const float *q, double *F27) { return seven_point(p, q, F27); }
int32_t oracle_seven_point_defined(const float *p, const float *q, double *F27) { return defined::seven_point(p, q, F27); }

// -> number of inliers of the best model (0: no model); mask[n] and F[9] filled when > 0
static int32_t find_fundamental_ransac_with(int (*seven)(const float *, const float *, double *), int32_t n, const float *p, const float *q, double threshold,
                                            double confidence, int32_t max_iters, uint8_t *mask, double *F_out) {
    constexpr int kModel = 7;
    for (int i = 0; i < n; ++i) mask[i] = 0;
    if (n < kModel) return 0;
    CvRng rng((uint64_t)-1);
    const double thresh2 = threshold * threshold;
    std::vector<uint8_t> cur((size_t)n), best((size_t)n, 0);
    double best_model[9] = {0};
    int niters = max_iters, max_good = 0;
    if (n == kModel) niters = 1;
    std::vector<float> ms1(2 * kModel), ms2(2 * kModel);
    for (int iter = 0; iter < niters; ++iter) {
        if (n > kModel) {
            // getSubset: distinct indices, a point that makes the sample degenerate restarts the WHOLE sample
            std::vector<int> idx(kModel);
            int i = 0, j, iters = 0;
            const int max_attempts = 10000;
            for (; iters < max_attempts; ++iters) {
                for (i = 0; i < kModel && iters < max_attempts;) {
                    int idx_i = idx[i] = rng.uniform(0, n);
                    for (j = 0; j < i; ++j)
                        if (idx_i == idx[j]) break;
                    if (j < i) continue;
                    ms1[2 * i] = p[2 * idx_i], ms1[2 * i + 1] = p[2 * idx_i + 1], ms2[2 * i] = q[2 * idx_i], ms2[2 * i + 1] = q[2 * idx_i + 1];
                    ++i;
                }
                if (i == kModel && (have_collinear_points(ms1.data(), i) || have_collinear_points(ms2.data(), i))) continue;
                break;
            }
            const bool found = i == kModel && iters < max_attempts;
            if (!found) {
                if (iter == 0) return 0;
                break;
            }
        } else {
            std::copy(p, p + 2 * kModel, ms1.begin()), std::copy(q, q + 2 * kModel, ms2.begin());
        }
        double models[27];
        const int nmodels = seven(ms1.data(), ms2.data(), models);
        if (nmodels <= 0) continue;
        for (int m = 0; m < nmodels; ++m) {
            const int good = find_inliers(n, p, q, models + 9 * m, thresh2, cur.data());
            if (good > std::max(max_good, kModel - 1)) {
                std::swap(cur, best);
                std::copy(models + 9 * m, models + 9 * m + 9, best_model);
                max_good = good;
                niters = update_num_iters(confidence, (double)(n - good) / n, kModel, niters);
            }
        }
    }
    if (max_good > 0) {
        std::copy(best.begin(), best.end(), mask);
        if (F_out) std::copy(best_model, best_model + 9, F_out);
    }
    return max_good;
}
// the independent restatement (Jacobi null space, closed-form roots through libm): the check of the algorithm
int32_t oracle_find_fundamental_ransac(int32_t n, const float *p, const float *q, double threshold, double confidence, int32_t max_iters, uint8_t *mask, double *F_out) {
    return find_fundamental_ransac_with(seven_point, n, p, q, threshold, confidence, max_iters, mask, F_out);
}
// the same loop over the seven-point step in its DEFINED arithmetic: what the kernel is held to bit for bit, and what the sequence-level chains use
int32_t oracle_find_fundamental_ransac_defined(int32_t n, const float *p, const float *q, double threshold, double confidence, int32_t max_iters, uint8_t *mask,
                                               double *F_out) {
    return find_fundamental_ransac_with(defined::seven_point, n, p, q, threshold, confidence, max_iters, mask, F_out);
}

} // extern "C"
