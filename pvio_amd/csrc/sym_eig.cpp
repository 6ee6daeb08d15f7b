// sym_eig.cpp -- see sym_eig.h.  Compiled twice by the Makefile (baseline x86-64, -mavx2 -mfma) with
// PVBA_EIG_FN naming the build; the baseline object also carries the CPUID dispatcher.
#include "sym_eig.h"

// No templates from the standard library in here: an out-of-line instantiation emitted by the AVX builds could be the copy
// the linker keeps for everybody.
#include <cmath>
#include <cstddef>

#ifndef PVBA_EIG_FN
#define PVBA_EIG_FN sym_eig_generic
#define PVBA_EIG_DISPATCHER 1
#endif
// tests/micro/eig_bench.cpp builds extra copies with -DPVBA_EIG_TIMES=<array> (microseconds of reduction / accumulation / QL of the
// last call) and -DPVBA_EIG_HYPOT=1 (hypot() for every rotation, the form measured against); the library's builds carry neither
#ifdef PVBA_EIG_TIMES
#include <chrono>
namespace pvba {
double PVBA_EIG_TIMES[3];
}
#define PVBA_EIG_T(k)                                                                                        \
    {                                                                                                        \
        const auto now = std::chrono::steady_clock::now();                                                   \
        PVBA_EIG_TIMES[k] = std::chrono::duration<double, std::micro>(now - eig_t0).count(), eig_t0 = now;  \
    }
#else
#define PVBA_EIG_T(k)
#endif
#ifndef PVBA_EIG_HYPOT
#define PVBA_EIG_HYPOT 0
#endif

namespace pvba {
namespace eig_detail {
// sixteen partial sums = four independent vector accumulators under AVX2: one accumulator is a chain of dependent FMAs (4 cycles each),
// i.e. one double per cycle where the core can do eight
static inline double dot(const double *__restrict a, const double *__restrict b, int n) {
    double s[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t4[4] = {0, 0, 0, 0}, r = 0;
    int k = 0;
    for (; k + 16 <= n; k += 16)
        for (int t = 0; t < 16; ++t) s[t] += a[k + t] * b[k + t];
    for (; k + 4 <= n; k += 4)
        for (int t = 0; t < 4; ++t) t4[t] += a[k + t] * b[k + t];
    for (; k < n; ++k) r += a[k] * b[k];
    for (int t = 0; t < 4; ++t) t4[t] += (s[t] + s[t + 4]) + (s[t + 8] + s[t + 12]);
    return r + ((t4[0] + t4[1]) + (t4[2] + t4[3]));
}
} // namespace eig_detail

void PVBA_EIG_FN(const double *A, int n, double *__restrict w, double *__restrict Vt) {
    using eig_detail::dot;
#ifdef PVBA_EIG_TIMES
    auto eig_t0 = std::chrono::steady_clock::now();
#endif
    double *__restrict e = new double[n > 0 ? n : 1]();
    auto Z = [Vt, n](int a, int b) -> double & { return Vt[(size_t)b * n + a]; }; // Z(a, .) contiguous in a
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Z(i, j) = j <= i ? A[(size_t)i * n + j] : A[(size_t)j * n + i];
    // --- Householder reduction to tridiagonal form: step i annihilates row i left of the sub-diagonal; the reflector is
    // kept in column i of Z (contiguous), w carries the current row, e the vector p = A u / h and then q ---
    for (int j = 0; j < n; ++j) w[j] = Z(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0, h = 0;
        for (int k = 0; k < i; ++k) scale += std::fabs(w[k]);
        if (scale == 0.0) {
            e[i] = w[i - 1];
            for (int j = 0; j < i; ++j) {
                w[j] = Z(i - 1, j);
                Z(i, j) = 0, Z(j, i) = 0;
            }
        } else {
            for (int k = 0; k < i; ++k) w[k] /= scale, h += w[k] * w[k];
            double f = w[i - 1], g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            w[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0;
            for (int j = 0; j < i; ++j) { // p = A u from the stored triangle: column j below the diagonal serves row j and column j
                f = w[j];
                Z(j, i) = f;
                const double *__restrict zj = &Z(0, j);
                // two clean passes over the (L1-resident) column instead of one fused loop: the fused form did not vectorize
                e[j] = e[j] + zj[j] * f + dot(zj + j + 1, w + j + 1, i - j - 1);
                for (int k = j + 1; k < i; ++k) e[k] += zj[k] * f;
            }
            f = 0;
            for (int j = 0; j < i; ++j) e[j] /= h, f += e[j] * w[j];
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * w[j];
            for (int j = 0; j < i; ++j) { // rank-2 update of the stored triangle
                f = w[j], g = e[j];
                double *__restrict zj = &Z(0, j);
                for (int k = j; k < i; ++k) zj[k] -= (f * e[k] + g * w[k]);
                w[j] = zj[i - 1];
                zj[i] = 0;
            }
        }
        w[i] = h;
    }
    PVBA_EIG_T(0)
    for (int i = 0; i < n - 1; ++i) { // accumulate the reflectors into Z
        Z(n - 1, i) = Z(i, i);
        Z(i, i) = 1.0;
        const double h = w[i + 1];
        if (h != 0.0) {
            const double *__restrict u = &Z(0, i + 1);
            for (int k = 0; k <= i; ++k) w[k] = u[k] / h;
            for (int j = 0; j <= i; ++j) {
                double *__restrict zj = &Z(0, j);
                const double g = dot(u, zj, i + 1);
                for (int k = 0; k <= i; ++k) zj[k] -= g * w[k];
            }
        }
        for (int k = 0; k <= i; ++k) Z(k, i + 1) = 0;
    }
    for (int j = 0; j < n; ++j) w[j] = Z(n - 1, j), Z(n - 1, j) = 0;
    Z(n - 1, n - 1) = 1.0;
    // Z now holds Q with A = Q T Q^T; column j of Q = Z(., j) = row j of Vt, so the QL rotations, which mix pairs of columns
    // of the eigenvector matrix, mix pairs of contiguous rows.
    // --- implicit QL on the tridiagonal matrix ---
    PVBA_EIG_T(1)
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
#if PVBA_EIG_HYPOT
                    r = std::hypot(p, e[i]);
#else
                    // the rotation's radius sits on the serial chain of the sweep (c and p of the next rotation depend on it): a plain
                    // square root where the squares can neither overflow nor vanish, hypot() (three times the latency) otherwise
                    const double r2 = p * p + e[i] * e[i];
                    r = (r2 > 1.0e-280 && r2 < 1.0e280) ? std::sqrt(r2) : std::hypot(p, e[i]);
#endif
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * w[i] - s * g;
                    w[i + 1] = h + s * (c * g + s * w[i]);
                    double *__restrict v0 = Vt + (size_t)i * n, *__restrict v1 = Vt + (size_t)(i + 1) * n;
                    for (int k = 0; k < n; ++k) {
                        const double hk = v1[k];
                        v1[k] = s * v0[k] + c * hk;
                        v0[k] = c * v0[k] - s * hk;
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
    PVBA_EIG_T(2)
    for (int i = 0; i < n - 1; ++i) { // ascending order
        int k = i;
        double p = w[i];
        for (int j = i + 1; j < n; ++j)
            if (w[j] < p) k = j, p = w[j];
        if (k != i) {
            w[k] = w[i], w[i] = p;
            double *__restrict vi = Vt + (size_t)i * n, *__restrict vk = Vt + (size_t)k * n;
            for (int j = 0; j < n; ++j) {
                const double t = vi[j];
                vi[j] = vk[j], vk[j] = t;
            }
        }
    }
    delete[] e;
}

#ifdef PVBA_EIG_DISPATCHER
#ifdef PVBA_EIG_WIDE_BUILDS
void sym_eig_avx2(const double *A, int n, double *__restrict w, double *__restrict Vt);
#endif
namespace {
using EigFn = void (*)(const double *, int, double *, double *);
struct EigPick {
    EigFn fn = sym_eig_generic;
    const char *isa = "generic";
    EigPick() {
#ifdef PVBA_EIG_WIDE_BUILDS
        __builtin_cpu_init();
        if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) fn = sym_eig_avx2, isa = "avx2";
#endif
    }
};
const EigPick &eig_pick() {
    static const EigPick p;
    return p;
}
} // namespace
void sym_eig(const double *A, int n, double *w, double *Vt) { eig_pick().fn(A, n, w, Vt); }
const char *sym_eig_isa() { return eig_pick().isa; }
#endif

} // namespace pvba
