// eig_bench.cpp -- TEST TOOL: times the marginalization's eigen-decomposition (pvio_amd/csrc/sym_eig.h) on a matrix dumped by
// tests/prof_marg.py (binary: int32 n, then n*n doubles), next to the round-2 strided version when built with -DWITH_R2.
// Links both builds of pvio_amd/csrc/sym_eig.cpp (tests/micro/Makefile: eig_bench) and times each one the host supports.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <cmath>
#include "pvio_amd/csrc/sym_eig.h"
#include "tests/micro/variants/sym_eig_r2.h"
namespace pvba {
void sym_eig_generic(const double *, int, double *, double *);
void sym_eig_avx2(const double *, int, double *, double *);
void eig_avx2_timed(const double *, int, double *, double *);
void eig_avx2_hypot(const double *, int, double *, double *);
void eig_avx512(const double *, int, double *, double *);
extern double eig_times_a[3], eig_times_b[3], eig_times_c[3];
} // namespace pvba
int main(int argc, char **argv) {
    FILE *f = argc > 1 ? std::fopen(argv[1], "rb") : nullptr;
    int32_t n = 0;
    if (!f || std::fread(&n, 4, 1, f) != 1) return std::fprintf(stderr, "usage: eig_bench matrix.bin\n"), 2;
    std::vector<double> A((size_t)n * n), w(n), V((size_t)n * n), w2(n), V2((size_t)n * n);
    if (std::fread(A.data(), 8, A.size(), f) != A.size()) return 2;
    __builtin_cpu_init();
    const bool has2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    struct { const char *name; void (*fn)(const double *, int, double *, double *); bool ok; double best; } runs[] = {
        {"round-2 strided layout", [](const double *a, int m, double *ww, double *vv) { pvold::sym_eig(a, m, ww, vv); }, true, 1e9},
        {"generic", pvba::sym_eig_generic, true, 1e9}, {"avx2", pvba::sym_eig_avx2, has2, 1e9},
        {"dispatched", pvba::sym_eig, true, 1e9}, {"avx2 timed", pvba::eig_avx2_timed, has2, 1e9}, {"avx2 hypot()", pvba::eig_avx2_hypot, has2, 1e9},
        {"avx512", pvba::eig_avx512, has2 && __builtin_cpu_supports("avx512f"), 1e9}};
    for (int r = 0; r < 50; ++r)
        for (auto &run : runs) {
            if (!run.ok) continue;
            auto t0 = std::chrono::steady_clock::now();
            run.fn(A.data(), n, w.data(), V.data());
            run.best = std::min(run.best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    double rec = 0, amax = 0, orth = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0, o = 0;
            for (int k = 0; k < n; ++k) s += V[(size_t)k * n + i] * w[k] * V[(size_t)k * n + j], o += V[(size_t)i * n + k] * V[(size_t)j * n + k];
            rec = std::max(rec, std::fabs(s - A[(size_t)i * n + j])), amax = std::max(amax, std::fabs(A[(size_t)i * n + j])), orth = std::max(orth, std::fabs(o - (i == j)));
        }
    std::printf("n=%d (dispatcher picked %s):", n, pvba::sym_eig_isa());
    for (auto &run : runs)
        if (run.ok) std::printf("  %s %.0f us", run.name, run.best);
    std::printf("\n  phases (reduction / accumulation / QL, us): avx2 %.0f %.0f %.0f   avx2 with hypot() %.0f %.0f %.0f   avx512 %.0f %.0f %.0f\n", pvba::eig_times_a[0], pvba::eig_times_a[1],
                pvba::eig_times_a[2], pvba::eig_times_b[0], pvba::eig_times_b[1], pvba::eig_times_b[2], pvba::eig_times_c[0], pvba::eig_times_c[1], pvba::eig_times_c[2]);
    std::printf("  max|A - V L V^T| %.2e of max|A| %.2e, max|V V^T - I| %.2e, eigenvalues %.3e .. %.3e\n", rec, amax, orth, w[0], w[n - 1]);
    return 0;
}
