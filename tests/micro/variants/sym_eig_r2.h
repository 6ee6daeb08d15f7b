// sym_eig.h -- host-side symmetric eigen-decomposition (Householder tridiagonalization + implicit QL), used once per
// marginalization for the 15(N-1) x 15(N-1) information matrix (replaces Eigen::SelfAdjointEigenSolver at
// pvio/src/pvio/estimation/bundle_adjustor.cpp:584).  Eigenvalues ascending; V column k = eigenvector k.
#pragma once
#include <cmath>
#include <vector>

namespace pvold {

// A: n x n row-major symmetric (lower triangle is read).  On return V (n x n row-major) holds the eigenvectors in
// its columns and w the eigenvalues in ascending order.
inline void sym_eig(const double *A, int n, double *w, double *V) {
    std::vector<double> e(n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = j <= i ? A[(size_t)i * n + j] : A[(size_t)j * n + i];
    // --- Householder reduction to tridiagonal form (classic tred2 organisation) ---
    for (int j = 0; j < n; ++j) w[j] = V[(size_t)(n - 1) * n + j];
    for (int i = n - 1; i > 0; --i) {
        double scale = 0, h = 0;
        for (int k = 0; k < i; ++k) scale += std::fabs(w[k]);
        if (scale == 0.0) {
            e[i] = w[i - 1];
            for (int j = 0; j < i; ++j) {
                w[j] = V[(size_t)(i - 1) * n + j];
                V[(size_t)i * n + j] = 0, V[(size_t)j * n + i] = 0;
            }
        } else {
            for (int k = 0; k < i; ++k) w[k] /= scale, h += w[k] * w[k];
            double f = w[i - 1], g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            w[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0;
            for (int j = 0; j < i; ++j) {
                f = w[j];
                V[(size_t)j * n + i] = f;
                g = e[j] + V[(size_t)j * n + j] * f;
                for (int k = j + 1; k <= i - 1; ++k) {
                    g += V[(size_t)k * n + j] * w[k];
                    e[k] += V[(size_t)k * n + j] * f;
                }
                e[j] = g;
            }
            f = 0;
            for (int j = 0; j < i; ++j) e[j] /= h, f += e[j] * w[j];
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * w[j];
            for (int j = 0; j < i; ++j) {
                f = w[j], g = e[j];
                for (int k = j; k <= i - 1; ++k) V[(size_t)k * n + j] -= (f * e[k] + g * w[k]);
                w[j] = V[(size_t)(i - 1) * n + j];
                V[(size_t)i * n + j] = 0;
            }
        }
        w[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) { // accumulate transformations
        V[(size_t)(n - 1) * n + i] = V[(size_t)i * n + i];
        V[(size_t)i * n + i] = 1.0;
        const double h = w[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) w[k] = V[(size_t)k * n + i + 1] / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0;
                for (int k = 0; k <= i; ++k) g += V[(size_t)k * n + i + 1] * V[(size_t)k * n + j];
                for (int k = 0; k <= i; ++k) V[(size_t)k * n + j] -= g * w[k];
            }
        }
        for (int k = 0; k <= i; ++k) V[(size_t)k * n + i + 1] = 0;
    }
    for (int j = 0; j < n; ++j) w[j] = V[(size_t)(n - 1) * n + j], V[(size_t)(n - 1) * n + j] = 0;
    V[(size_t)(n - 1) * n + n - 1] = 1.0;
    e[0] = 0;
    // --- implicit QL on the tridiagonal matrix ---
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0;
    double f = 0, tst1 = 0;
    const double eps = std::pow(2.0, -52.0);
    for (int l = 0; l < n; ++l) {
        tst1 = std::fmax(tst1, std::fabs(w[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) {
            if (std::fabs(e[m]) <= eps * tst1) break;
            ++m;
        }
        if (m > l) {
            int iter = 0;
            do {
                ++iter;
                double g = w[l], p = (w[l + 1] - g) / (2.0 * e[l]), r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                w[l] = e[l] / (p + r);
                w[l + 1] = e[l] * (p + r);
                const double dl1 = w[l + 1];
                double h = g - w[l];
                for (int i = l + 2; i < n; ++i) w[i] -= h;
                f += h;
                p = w[m];
                double c = 1, c2 = c, c3 = c, s = 0, s2 = 0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2, c2 = c, s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * w[i] - s * g;
                    w[i + 1] = h + s * (c * g + s * w[i]);
                    for (int k = 0; k < n; ++k) {
                        h = V[(size_t)k * n + i + 1];
                        V[(size_t)k * n + i + 1] = s * V[(size_t)k * n + i] + c * h;
                        V[(size_t)k * n + i] = c * V[(size_t)k * n + i] - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                w[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
        }
        w[l] += f;
        e[l] = 0;
    }
    for (int i = 0; i < n - 1; ++i) { // ascending order
        int k = i;
        double p = w[i];
        for (int j = i + 1; j < n; ++j)
            if (w[j] < p) k = j, p = w[j];
        if (k != i) {
            w[k] = w[i], w[i] = p;
            for (int j = 0; j < n; ++j) std::swap(V[(size_t)j * n + i], V[(size_t)j * n + k]);
        }
    }
}

} // namespace pvold
