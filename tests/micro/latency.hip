// Latency microbenchmarks for the single-workgroup dense kernel (one wave per SIMD): prints cycles per operation.
// Build: hipcc -O3 --offload-arch=gfx950 -o latency latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N 256
#define PIN(x) asm volatile("" : "+v"(x))
__global__ void __launch_bounds__(256) k(long long *out, double *sink, const double *src) {
    __shared__ double lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    long long t0, t1;
    double acc = src[tid];
    PIN(acc);
    // 1 dependent f64 FMA chain
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) acc = fma(acc, 1.0000001, 1e-9);
    PIN(acc);
    t1 = clock64();
    if (tid == 0) out[0] = t1 - t0;
    // 2 dependent v_rsq_f64 chain
    double r = acc;
    PIN(r);
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) r = __builtin_amdgcn_rsq(r + 2.0);
    PIN(r);
    t1 = clock64();
    if (tid == 0) out[1] = t1 - t0;
    acc += r;
    // 3 dependent LDS read chain (pointer chasing through values)
    int idx = tid;
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) idx = ((int)lds[idx] + idx + 17) & 4095;
    t1 = clock64();
    if (tid == 0) out[2] = t1 - t0;
    acc += idx;
    // 4 dependent MFMA f64 16x16x4 chain
    d4 c = {acc, 0, 0, 0};
    PIN(c);
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c, 0, 0, 0);
    PIN(c);
    t1 = clock64();
    if (tid == 0) out[3] = t1 - t0;
    acc += c[0] + c[1] + c[2] + c[3];
    // 5 independent MFMA (4 accumulators)
    d4 c0 = {acc, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    PIN(c0); PIN(c1); PIN(c2); PIN(c3);
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c3, 0, 0, 0);
    }
    PIN(c0); PIN(c1); PIN(c2); PIN(c3);
    t1 = clock64();
    if (tid == 0) out[4] = t1 - t0;
    acc += c0[0] + c1[1] + c2[2] + c3[3];
    // 6 barriers
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) __syncthreads();
    t1 = clock64();
    if (tid == 0) out[5] = t1 - t0;
    // 7 LDS write -> barrier -> read round trip
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        lds[(tid + i) & 4095] = acc;
        __syncthreads();
        acc += lds[(tid + i + 1) & 4095];
    }
    t1 = clock64();
    if (tid == 0) out[6] = t1 - t0;
    // 8 dependent global load chain (L2-resident pointer chase)
    int g = tid;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) g = ((int)src[g] + g + 33) & 4095;
    t1 = clock64();
    if (tid == 0) out[7] = t1 - t0;
    acc += g;
    // 9 independent f64 FMA throughput (8 chains)
    double a[8];
    for (int j = 0; j < 8; ++j) { a[j] = acc + j; PIN(a[j]); }
    t0 = clock64();
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fma(a[j], 1.0000001, 1e-9);
    for (int j = 0; j < 8; ++j) PIN(a[j]);
    t1 = clock64();
    if (tid == 0) out[8] = t1 - t0;
    for (int j = 0; j < 8; ++j) acc += a[j];
    // 10 LDS read -> 2 MFMA -> LDS write per tile, unpipelined
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        const int base = (i * 256) & 4095;
        d4 cc;
        for (int q = 0; q < 4; ++q) cc[q] = lds[(base + tid % 64 + 64 * q) & 4095];
        const double aa = lds[(base + tid) & 4095], bb = lds[(base + 7 + tid) & 4095];
        cc = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb, cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f64_16x16x4f64(bb, aa, cc, 0, 0, 0);
        for (int q = 0; q < 4; ++q) lds[(base + tid % 64 + 64 * q) & 4095] = cc[q];
    }
    t1 = clock64();
    if (tid == 0) out[9] = t1 - t0;
    sink[tid] = acc + lds[tid];
}
int main() {
    long long *out; double *sink, *src;
    hipMalloc(&out, 16 * 8); hipMalloc(&sink, 256 * 8); hipMalloc(&src, 4096 * 8);
    double h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 1.0 + (i % 7);
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, sink, src);
    long long ho[16]; hipMemcpy(ho, out, 16 * 8, hipMemcpyDeviceToHost);
    const char *names[] = {"dep f64 fma", "dep v_rsq_f64", "dep LDS read (+cvt,add)", "dep mfma f64 16x16x4", "indep mfma f64 (x4)", "s_barrier (4 waves)",
                           "LDS write+barrier+read", "dep global load (L2)", "indep f64 fma (8 chains)", "tile: 6 lds rd + 2 mfma + 4 lds wr"};
    for (int i = 0; i < 10; ++i) printf("%-40s %8.1f cycles/op\n", names[i], (double)ho[i] / N);
    return 0;
}
