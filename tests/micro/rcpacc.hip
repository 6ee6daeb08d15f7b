// Accuracy of the hardware f64 reciprocal / reciprocal square root (v_rcp_f64 / v_rsq_f64) without and with Newton steps,
// against correctly rounded division / sqrt: decides how short the pivot chain of the dense kernel can be.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = r0 * fma(-v, r0, 2.0);            // one Newton step (2 dependent ops)
    double r1b = fma(fma(-v, r0, 1.0), r0, r0);   // the same step in residual form
    double q0 = __builtin_amdgcn_rsq(v);
    double q1 = q0 * fma(-0.5 * v, q0 * q0, 1.5);
    double q2 = q1 * fma(-0.5 * v, q1 * q1, 1.5);
    o[6 * i] = r0, o[6 * i + 1] = r1, o[6 * i + 2] = r1b, o[6 * i + 3] = q0, o[6 * i + 4] = q1, o[6 * i + 5] = q2;
}
static double ulps(double got, double want) {
    int64_t a, b;
    std::memcpy(&a, &got, 8), std::memcpy(&b, &want, 8);
    return std::fabs((double)(a - b));
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(6 * n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        double m = 1.0 + (double)(s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::ldexp(m, (int)(s % 80) - 40);
    }
    double *dx, *d_o;
    hipMalloc(&dx, n * 8), hipMalloc(&d_o, 6 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d_o, n);
    hipMemcpy(o.data(), d_o, 6 * n * 8, hipMemcpyDeviceToHost);
    const char *names[6] = {"rcp", "rcp + newton", "rcp + newton (residual form)", "rsq", "rsq + 1 newton", "rsq + 2 newton"};
    for (int c = 0; c < 6; ++c) {
        double mx = 0, rel = 0;
        for (int i = 0; i < n; ++i) {
            const double want = c < 3 ? 1.0 / x[i] : 1.0 / std::sqrt(x[i]);
            mx = std::fmax(mx, ulps(o[6 * i + c], want));
            rel = std::fmax(rel, std::fabs(o[6 * i + c] - want) / want);
        }
        std::printf("%-32s max %.0f ulp, max rel %.3e\n", names[c], mx, rel);
    }
    return 0;
}
