#include <hip/hip_runtime.h>
#include <cstdio>
#define PIN(x) asm volatile("" : "+v"(x))
__global__ void __launch_bounds__(256) k(long long *out, double *sink, const double *src) {
    const int tid = threadIdx.x;
    double a = src[tid], b = src[tid + 256], c = src[tid + 512];
    PIN(a); PIN(b); PIN(c);
    long long t0, t1;
    // dependent fma chain
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < 256; ++i) { a = fma(a, b, c); PIN(a); }
    t1 = clock64();
    if (tid == 0) out[0] = t1 - t0;
    // dependent mul chain
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < 256; ++i) { a = a * b; PIN(a); }
    t1 = clock64();
    if (tid == 0) out[1] = t1 - t0;
    // 4 independent fma chains
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    PIN(x0); PIN(x1); PIN(x2); PIN(x3);
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        x0 = fma(x0, b, c); x1 = fma(x1, b, c); x2 = fma(x2, b, c); x3 = fma(x3, b, c);
        PIN(x0); PIN(x1); PIN(x2); PIN(x3);
    }
    t1 = clock64();
    if (tid == 0) out[2] = t1 - t0;
    // dependent f32 fma chain
    float f = (float)a, g = (float)b, h = (float)c;
    PIN(f);
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < 256; ++i) { f = fmaf(f, g, h); PIN(f); }
    t1 = clock64();
    if (tid == 0) out[3] = t1 - t0;
    // rsq + 2 NR (fast_rsqrt) dependent
    double r = a * a + 2.0;
    PIN(r);
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        double y = __builtin_amdgcn_rsq(r);
        y = y * fma(-0.5 * r, y * y, 1.5);
        y = y * fma(-0.5 * r, y * y, 1.5);
        r = y + 2.0;
        PIN(r);
    }
    t1 = clock64();
    if (tid == 0) out[4] = t1 - t0;
    // cndmask-dependent chain: select + mul
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < 256; ++i) { a = (tid + i < 300) ? a * b : 0.0; PIN(a); }
    t1 = clock64();
    if (tid == 0) out[5] = t1 - t0;
    sink[tid] = a + x0 + x1 + x2 + x3 + f + r;
}
int main() {
    long long *out; double *sink, *src;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 256 * 8); (void)hipMalloc(&src, 768 * 8);
    double h[768]; for (int i = 0; i < 768; ++i) h[i] = 1.0 + 1e-7 * (i % 5);
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, sink, src);
    long long ho[8]; (void)hipMemcpy(ho, out, 64, hipMemcpyDeviceToHost);
    printf("dep f64 fma %.1f | dep f64 mul %.1f | 4 indep f64 fma chains: %.1f per fma | dep f32 fma %.1f | fast_rsqrt+add %.1f | select+mul %.1f\n", ho[0] / 256.0, ho[1] / 256.0, ho[2] / 256.0, ho[3] / 256.0, ho[4] / 64.0, ho[5] / 256.0);
    return 0;
}
