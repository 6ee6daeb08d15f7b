// oracle_math.h -- small FP64 math for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker.
//
// Restates (does not copy) the reference's geometry helpers and the Eigen-3.3 behaviours they rely on:
//   hat/expmap/logmap/right_jacobian   -> pvio/src/pvio/geometry/lie_algebra.h:25-42, lie_algebra.cpp:22-59
//   quaternion product / rotation      -> Eigen::Quaternion semantics used throughout estimation/ceres/*.h
//   AngleAxis(q) / Quaternion(AngleAxis) -> Eigen 3.3 (angle = 2 atan2(|v|, |w|), axis flipped when w<0)
// PARITY UNPINNED: Eigen itself is not available in this environment, so these are restatements of the
// documented algorithms, validated by the invariants in tests/test_oracle_math.py.
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

struct V3 {
    double v[3];
    double &operator[](int i) { return v[i]; }
    const double &operator[](int i) const { return v[i]; }
};
struct M3 {
    double m[3][3];
};
struct Q { // Eigen coeffs() order
    double x, y, z, w;
};

inline V3 mk(double a, double b, double c) { return V3{{a, b, c}}; }
inline V3 operator+(const V3 &a, const V3 &b) { return mk(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline V3 operator-(const V3 &a, const V3 &b) { return mk(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline V3 operator-(const V3 &a) { return mk(-a[0], -a[1], -a[2]); }
inline V3 operator*(double s, const V3 &a) { return mk(s * a[0], s * a[1], s * a[2]); }
inline double dot(const V3 &a, const V3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline V3 cross(const V3 &a, const V3 &b) {
    return mk(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline double norm(const V3 &a) { return std::sqrt(dot(a, a)); }

inline M3 zero3() {
    M3 r;
    std::memset(&r, 0, sizeof r);
    return r;
}
inline M3 eye3() {
    M3 r = zero3();
    r.m[0][0] = r.m[1][1] = r.m[2][2] = 1;
    return r;
}
inline M3 operator*(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
inline V3 operator*(const M3 &a, const V3 &b) {
    return mk(a.m[0][0] * b[0] + a.m[0][1] * b[1] + a.m[0][2] * b[2], a.m[1][0] * b[0] + a.m[1][1] * b[1] + a.m[1][2] * b[2],
              a.m[2][0] * b[0] + a.m[2][1] * b[1] + a.m[2][2] * b[2]);
}
inline M3 operator*(double s, const M3 &a) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = s * a.m[i][j];
    return r;
}
inline M3 operator+(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r;
}
inline M3 operator-(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] - b.m[i][j];
    return r;
}
inline M3 operator-(const M3 &a) { return -1.0 * a; }
inline M3 transpose(const M3 &a) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
// lie_algebra.h:25-30
inline M3 hat(const V3 &w) {
    M3 r = zero3();
    r.m[0][1] = -w[2];
    r.m[0][2] = w[1];
    r.m[1][0] = w[2];
    r.m[1][2] = -w[0];
    r.m[2][0] = -w[1];
    r.m[2][1] = w[0];
    return r;
}
// Eigen's fixed-size 3x3 inverse: cofactors / determinant.
inline M3 inverse(const M3 &a) {
    const double(*m)[3] = a.m;
    double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
    double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
    double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
    double id = 1.0 / det;
    M3 r;
    r.m[0][0] = c00 * id;
    r.m[1][0] = c01 * id;
    r.m[2][0] = c02 * id;
    r.m[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
    r.m[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
    r.m[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
    r.m[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    r.m[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    r.m[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
    return r;
}

inline Q qmul(const Q &a, const Q &b) {
    Q r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
inline Q qconj(const Q &a) { return Q{-a.x, -a.y, -a.z, a.w}; }
inline Q qnormalized(const Q &a) {
    double n = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
    return Q{a.x / n, a.y / n, a.z / n, a.w / n};
}
inline M3 qmat(const Q &q) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r.m[0][0] = 1 - (tyy + tzz);
    r.m[0][1] = txy - twz;
    r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;
    r.m[1][1] = 1 - (txx + tzz);
    r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;
    r.m[2][1] = tyz + twx;
    r.m[2][2] = 1 - (txx + tyy);
    return r;
}
// Eigen: q * v  ==  v + w*t + qv x t, t = 2 qv x v
inline V3 qrot(const Q &q, const V3 &v) {
    V3 qv = mk(q.x, q.y, q.z);
    V3 uv = cross(qv, v);
    uv = uv + uv;
    return v + q.w * uv + cross(qv, uv);
}
inline Q qload(const double *p) { return Q{p[0], p[1], p[2], p[3]}; }
inline void qstore(const Q &q, double *p) {
    p[0] = q.x;
    p[1] = q.y;
    p[2] = q.z;
    p[3] = q.w;
}
inline V3 vload(const double *p) { return mk(p[0], p[1], p[2]); }
inline void vstore(const V3 &v, double *p) {
    p[0] = v[0];
    p[1] = v[1];
    p[2] = v[2];
}

// lie_algebra.h:32-37: Quaternion(AngleAxis(|w|, w.stableNormalized()))
inline Q expmap(const V3 &w) {
    double angle = norm(w);
    double mx = std::fmax(std::fabs(w[0]), std::fmax(std::fabs(w[1]), std::fabs(w[2])));
    V3 axis = w;
    if (mx > 0) {
        V3 s = mk(w[0] / mx, w[1] / mx, w[2] / mx);
        double z = dot(s, s);
        if (z > 0) axis = (1.0 / std::sqrt(z)) * s;
    }
    double ha = 0.5 * angle;
    double sh = std::sin(ha);
    return Q{sh * axis[0], sh * axis[1], sh * axis[2], std::cos(ha)};
}
// lie_algebra.h:39-42: AngleAxis(q) then angle*axis  (Eigen 3.3 semantics)
inline V3 logmap(const Q &q) {
    double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (n != 0.0) {
        double angle = 2.0 * std::atan2(n, std::fabs(q.w));
        if (q.w < 0) n = -n;
        return mk(angle * (q.x / n), angle * (q.y / n), angle * (q.z / n));
    }
    return mk(0, 0, 0);
}
// lie_algebra.cpp:22-59
inline M3 right_jacobian(const V3 &w) {
    static const double root2_eps = std::sqrt(std::numeric_limits<double>::epsilon());
    static const double root4_eps = std::sqrt(root2_eps);
    static const double qdrt720 = std::sqrt(std::sqrt(720.0));
    static const double qdrt5040 = std::sqrt(std::sqrt(5040.0));
    static const double sqrt24 = std::sqrt(24.0);
    static const double sqrt120 = std::sqrt(120.0);
    double angle = norm(w);
    double cangle = std::cos(angle), sangle = std::sin(angle);
    double angle2 = angle * angle;
    double cos_term, sin_term;
    if (angle > root4_eps * qdrt720) {
        cos_term = (1 - cangle) / angle2;
    } else {
        cos_term = 0.5;
        if (angle > root2_eps * sqrt24) cos_term -= angle2 / 24.0;
    }
    if (angle > root4_eps * qdrt5040) {
        sin_term = (angle - sangle) / (angle * angle2);
    } else {
        sin_term = 1.0 / 6.0;
        if (angle > root2_eps * sqrt120) sin_term -= angle2 / 120.0;
    }
    M3 hw = hat(w);
    return eye3() - cos_term * hw + sin_term * (hw * hw);
}

// ---- dynamic dense helpers (row-major) -------------------------------------------------------
struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};

// In-place lower Cholesky of the leading n x n of a row-major matrix with leading dimension ld.
// Returns false when a pivot is not strictly positive / not finite (LINEAR_SOLVER_FAILURE).
inline bool cholesky_lower(double *A, int n, int ld) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * ld + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * ld + k] * A[(size_t)j * ld + k];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[(size_t)j * ld + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * ld + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * ld + k] * A[(size_t)j * ld + k];
            A[(size_t)i * ld + j] = s / d;
        }
    }
    return true;
}
inline void cholesky_solve(const double *L, int n, int ld, double *b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[(size_t)i * ld + k] * b[k];
        b[i] = s / L[(size_t)i * ld + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * ld + i] * b[k];
        b[i] = s / L[(size_t)i * ld + i];
    }
}
// General inverse by LU with partial pivoting (what Eigen's .inverse() does for n > 4).
inline bool lu_inverse(const double *Ain, int n, double *inv) {
    std::vector<double> A(Ain, Ain + (size_t)n * n);
    std::vector<int> piv(n);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(A[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(A[(size_t)i * n + k]) > best) best = std::fabs(A[(size_t)i * n + k]), p = i;
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
            std::swap(piv[k], piv[p]);
        }
        for (int i = k + 1; i < n; ++i) {
            double f = A[(size_t)i * n + k] / A[(size_t)k * n + k];
            A[(size_t)i * n + k] = f;
            for (int j = k + 1; j < n; ++j) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
        }
    }
    for (int col = 0; col < n; ++col) {
        std::vector<double> x(n);
        for (int i = 0; i < n; ++i) x[i] = (piv[i] == col) ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < i; ++k) x[i] -= A[(size_t)i * n + k] * x[k];
        for (int i = n - 1; i >= 0; --i) {
            for (int k = i + 1; k < n; ++k) x[i] -= A[(size_t)i * n + k] * x[k];
            x[i] /= A[(size_t)i * n + i];
        }
        for (int i = 0; i < n; ++i) inv[(size_t)i * n + col] = x[i];
    }
    return true;
}
// Cyclic Jacobi eigen-decomposition of a symmetric matrix: A = V diag(w) V^T, eigenvalues ascending,
// V column k = eigenvector k (row-major storage V[i*n+k]).
inline void sym_eig(const double *Ain, int n, double *w, double *V) {
    std::vector<double> A(Ain, Ain + (size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) (i == j ? diag : off) += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        if (off <= 1e-60 || off <= 1e-34 * diag) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    for (int i = 0; i < n; ++i) // insertion sort ascending
        for (int j = i; j > 0 && A[(size_t)idx[j] * n + idx[j]] < A[(size_t)idx[j - 1] * n + idx[j - 1]]; --j) std::swap(idx[j], idx[j - 1]);
    std::vector<double> Vs((size_t)n * n);
    for (int k = 0; k < n; ++k) {
        w[k] = A[(size_t)idx[k] * n + idx[k]];
        for (int i = 0; i < n; ++i) Vs[(size_t)i * n + k] = V[(size_t)i * n + idx[k]];
    }
    std::memcpy(V, Vs.data(), sizeof(double) * n * n);
}

} // namespace orc
