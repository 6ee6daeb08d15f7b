// pv_fundamental.h -- the arithmetic of the fundamental-matrix RANSAC, shared by the device kernel (klt.hip: k_fund_hypotheses), the
// host orchestration around it (Klt::fundamental_ransac) and the host-only implementation the tests hold it against
// (pvio_amd/host/fundamental_ransac.cpp).
//
// Reference: the outlier rejection step of OpenCvImage::track_keypoints, pvio-extra/src/pvio/extra/opencv_image.cpp:113-129 --
// cv::findFundamentalMat(p, q, FM_RANSAC, 1.0, 0.99, mask).  OpenCV is a third-party dependency that is not in /root/reference: this
// restates its published algorithm (calib3d: RANSACPointSetRegistrator + FMEstimatorCallback) -- PARITY UNPINNED, see
// fundamental_ransac.h.  The pieces:
//   FmRng / fm_draw_sample   cv::RNG (multiply-with-carry, seed (uint64)-1), seven distinct indices, a sample whose last point is
//                            collinear with two earlier ones (either image) is redrawn as a whole          [host only]
//   fm_seven_point           null space of the 7 x 9 epipolar constraints (Householder QR of A^T), real roots of
//                            det(l F1 + (1 - l) F2) = 0 in the order of the closed form, F scaled to F[8] = 1   [host + device]
//   fm_error                 max of the two squared point-to-epipolar-line distances, as a float              [host + device]
//   fm_update_iterations     log(1 - confidence) / log(1 - (1 - eps)^7), capped                               [host only]
//
// DEFINED ARITHMETIC (round 5).  A hypothesis' inlier count decides which hypothesis wins, a correspondence whose error sits on the threshold moves a
// count by one, and two hypotheses tie often enough that a long sequence meets such a case (tests/golden/ransac_ties.npz: frames 34 and 63 of two rendered
// sequences).  So the matrices are computed by ONE sequence of IEEE operations wherever this header is compiled: no FMA contraction (the pragma below; hipcc
// contracts by default, g++ -ffp-contract=off does not), and no libm call whose last bit differs between glibc and the device library -- the closed
// form's acos / cos are replaced by a bisection on the trisection polynomial (4 c^3 - 3 c = cos theta, c = cos(theta / 3) in [1/2, 1]) and the angle-sum
// identities, pow(x, 1/3) by a fixed number of Newton steps; +, -, *, /, sqrt are correctly rounded on both sides.  The device form, the sequential host
// form and the oracle's `defined` entry point (oracle/oracle_ransac.cpp, restated there) are bit-identical; the oracle's independent entry point (Jacobi
// null space, libm closed form) and tests/np_ransac.py (LAPACK) stay the checks of the ALGORITHM, at a tolerance.
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "pv_math.h" // PV_HD

// (clang: the pragma is scoped to each function body below -- PV_FM_NO_CONTRACT, like pv_math.h's q_mul -- so that including this header does not change
// how the rest of a translation unit is compiled; GCC: push / pop around the header)
#if defined(__clang__)
#define PV_FM_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define PV_FM_NO_CONTRACT
#endif
#if !defined(__clang__) && defined(__GNUC__)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off") // g++ contracts by default (-ffp-contract=fast) unless the build says otherwise: the functions below must not
#endif

namespace pvfm {

// x^(1/3), x > 0: Newton on y^3 = x from a power of two within a factor of two of the root, a fixed number of steps (each: y <- (2 y + x / y^2) / 3)
PV_HD double fm_cbrt(double x) {
    PV_FM_NO_CONTRACT
    if (!(x > 0)) return 0.0;
    int e;
    (void)frexp(x, &e); // x = m 2^e, m in [1/2, 1): exact
    const int k = e >= 0 ? e / 3 : -((-e + 2) / 3);
    double y = ldexp(1.0, k);
    for (int it = 0; it < 12; ++it) y = (2.0 * y + x / (y * y)) * (1.0 / 3.0);
    return y;
}

PV_HD double fm_det3(const double *m) {
    PV_FM_NO_CONTRACT
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// real roots of c[0] x^3 + c[1] x^2 + c[2] x + c[3] = 0 in the order of the closed form (three cosines, or the single real root)
PV_HD int fm_solve_cubic(const double c[4], double x[3]) {
    PV_FM_NO_CONTRACT
    const double a0 = c[0];
    if (a0 == 0) {
        if (c[1] == 0) {
            if (c[2] == 0) return 0;
            x[0] = -c[3] / c[2];
            return 1;
        }
        double d = c[2] * c[2] - 4 * c[1] * c[3];
        if (d < 0) return 0;
        d = sqrt(d);
        const double q1 = (-c[2] + d) * 0.5, q2 = (c[2] + d) * -0.5;
        if (fabs(q1) > fabs(q2)) x[0] = q1 / c[1], x[1] = c[3] / q1;
        else x[0] = q2 / c[1], x[1] = c[3] / q2;
        return d > 0 ? 2 : 1;
    }
    const double a1 = c[1] / a0, a2 = c[2] / a0, a3 = c[3] / a0;
    const double Q = (a1 * a1 - 3 * a2) * (1. / 9), R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54), Qc = Q * Q * Q;
    double d = Qc - R * R;
    if (d > 0) {
        // three real roots t0 cos(theta / 3 + 2 pi k / 3) - t2, cos theta = R / sqrt(Q^3), in the order k = 0, 1, 2 of the closed form.
        // c = cos(theta / 3) is the root of 4 c^3 - 3 c = cos theta in [1/2, 1] (theta in [0, pi]); the polynomial is increasing there: 64 bisection steps.
        const double r = R / sqrt(Qc), t0 = -2 * sqrt(Q), t2 = a1 * (1. / 3);
        double lo = 0.5, hi = 1.0;
        for (int it = 0; it < 64; ++it) {
            const double mid = 0.5 * (lo + hi);
            if ((4.0 * mid * mid - 3.0) * mid < r) lo = mid;
            else hi = mid;
        }
        const double c3 = 0.5 * (lo + hi), s3 = sqrt((1.0 - c3) * (1.0 + c3)), h3 = 0.86602540378443864676; // sin(theta / 3) >= 0; sqrt(3) / 2
        x[0] = t0 * c3 - t2, x[1] = t0 * (-0.5 * c3 - s3 * h3) - t2, x[2] = t0 * (-0.5 * c3 + s3 * h3) - t2;
        return 3;
    }
    if (d == 0) {
        if (R >= 0) x[0] = -2 * fm_cbrt(R) - a1 / 3, x[1] = fm_cbrt(R) - a1 / 3;
        else x[0] = 2 * fm_cbrt(-R) - a1 / 3, x[1] = -fm_cbrt(-R) - a1 / 3;
        return 2;
    }
    d = sqrt(-d);
    double e = fm_cbrt(d + fabs(R));
    if (R > 0) e = -e;
    x[0] = (e + Q / e) - a1 * (1. / 3);
    return 1;
}

// up to three 3 x 3 matrices (row-major) with q^T F p = 0 for the seven correspondences
PV_HD int fm_seven_point(const float p[14], const float q[14], double F[27]) {
    PV_FM_NO_CONTRACT
    // rows (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1) . f = 0 ; f1, f2 = a basis of the null space.  The null space comes from a
    // Householder QR of A^T (9 x 7): the last two columns of Q are orthogonal to all seven rows.  (The eigenvectors of A^T A would
    // square the condition number -- pixel coordinates are not normalized here, as in OpenCV's 7-point routine.)
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
        nrm = sqrt(nrm);
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
    c[3] = fm_det3(f2), c[0] = fm_det3(D), c[2] = 0, c[1] = 0;
    for (int row = 0; row < 3; ++row) {
        for (int k = 0; k < 9; ++k) tmp[k] = f2[k];
        for (int k = 0; k < 3; ++k) tmp[3 * row + k] = D[3 * row + k];
        c[2] += fm_det3(tmp); // linear term: one row of D
        for (int k = 0; k < 9; ++k) tmp[k] = D[k];
        for (int k = 0; k < 3; ++k) tmp[3 * row + k] = f2[3 * row + k];
        c[1] += fm_det3(tmp); // quadratic term: one row of f2
    }
    double roots[3];
    const int n = fm_solve_cubic(c, roots);
    int m = 0;
    for (int k = 0; k < n; ++k) {
        const double l = roots[k];
        double *Fk = F + 9 * m, nrm = 0;
        for (int e = 0; e < 9; ++e) Fk[e] = f2[e] + l * D[e], nrm += Fk[e] * Fk[e];
        if (!(nrm > 0) || !isfinite(nrm)) continue;
        const double s = fabs(Fk[8]) > DBL_EPSILON ? 1.0 / Fk[8] : 1.0 / sqrt(nrm); // F[2][2] = 1 where possible
        for (int e = 0; e < 9; ++e) Fk[e] *= s;
        ++m;
    }
    return m;
}

// max of the two squared point-to-epipolar-line distances of (x1, y1) <-> (x2, y2) under F, rounded to float like OpenCV's error vector
PV_HD float fm_error(const double *F, double x1, double y1, double x2, double y2) {
    PV_FM_NO_CONTRACT
    double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b), d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6], b = F[1] * x2 + F[4] * y2 + F[7], c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b), d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

// ---- host only ------------------------------------------------------------------------------------------------------------------
struct FmRng { // cv::RNG
    uint64_t state;
    explicit FmRng(uint64_t s) : state(s ? s : 0xffffffffu) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

inline bool fm_last_point_collinear(const float *m, int count) { // the count-th point against every pair of earlier ones
    PV_FM_NO_CONTRACT
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = m[2 * j] - m[2 * i], dy1 = m[2 * j + 1] - m[2 * i + 1];
        for (int k = 0; k < j; ++k) {
            const double dx2 = m[2 * k] - m[2 * i], dy2 = m[2 * k + 1] - m[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}

// the next sample of the run: seven distinct points whose last one is not collinear with two earlier ones (either image).  The
// sequence depends on the points and the generator only -- not on any model -- which is what lets the hypotheses be evaluated in
// batches.  false: no admissible sample within 10 000 attempts.
inline bool fm_draw_sample(FmRng &rng, int n, const float *p, const float *q, float sp[14], float sq[14]) {
    constexpr int kModel = 7, max_attempts = 10000;
    int idx[kModel], i = 0, attempts = 0;
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
        if (i == kModel && (fm_last_point_collinear(sp, i) || fm_last_point_collinear(sq, i))) continue;
        break;
    }
    return i == kModel && attempts < max_attempts;
}

inline int fm_update_iterations(double p, double ep, int model_points, int max_iters) {
    PV_FM_NO_CONTRACT
    p = p < 0. ? 0. : (p > 1. ? 1. : p), ep = ep < 0. ? 0. : (ep > 1. ? 1. : ep);
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN, denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num), denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

} // namespace pvfm

#if !defined(__clang__) && defined(__GNUC__)
#pragma GCC pop_options
#endif
