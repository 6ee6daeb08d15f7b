// Microbenchmark (gfx950): does a wave's FP64 VALU work overlap with its own v_mfma_f64_16x16x4_f64 stream?
// One workgroup of 1 or 4 waves; per wave: (a) NV dependent-free FP64 FMAs on 8 accumulator chains, (b) NM MFMAs on 4 accumulators,
// (c) both interleaved 1 MFMA : K FMAs.  Prints shader-clock cycles (s_memtime) per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int K>
__global__ void bench(double *out, long long *cyc, int iters) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001 + i;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    const double x = 1.0000001, y = 0.9999999 + threadIdx.x * 1e-12;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE & 2) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 1) {
#pragma unroll
                for (int k = 0; k < K; ++k) a[k & 7] = __builtin_fma(a[k & 7], x, y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // consume
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    asm volatile("" : "+v"(s));
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// two waves per SIMD (512 threads): waves 0-3 run the MFMA stream, waves 4-7 the FMA stream; cycles of each role
template <int K>
__global__ void bench_roles(double *out, long long *cyc, int iters, int mfma_waves, int valu_waves) {
    const int w = threadIdx.x >> 6;
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001 + i;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    const double x = 1.0000001, y = 0.9999999 + threadIdx.x * 1e-12;
    const bool do_m = w < 4 && mfma_waves, do_v = w >= 4 && valu_waves;
    __syncthreads();
    long long t0 = clock64();
    if (do_m) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[u], 0, 0, 0);
    } else if (do_v) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4 * K; ++k) a[k & 7] = __builtin_fma(a[k & 7], x, y);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    asm volatile("" : "+v"(s));
    long long t1 = clock64();
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
}

template <int K>
static void run_roles(int mfma_waves, int valu_waves, int iters) {
    double *out;
    long long *cyc, h[8];
    (void)hipMalloc(&out, 1024 * 8);
    (void)hipMalloc(&cyc, 64);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((bench_roles<K>), dim3(1), dim3(512), 0, 0, out, cyc, iters, mfma_waves, valu_waves);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("roles (2 waves / SIMD) mfma %d valu %d: wave0 (MFMA) %.1f cycles per MFMA, wave4 (%d FMA) %.1f cycles per group\n", mfma_waves, valu_waves,
           (double)h[0] / (4.0 * iters), K, (double)h[4] / (4.0 * iters));
    (void)hipFree(out), (void)hipFree(cyc);
}

template <int MODE, int K>
static void run(const char *name, int threads, int iters) {
    double *out;
    long long *cyc, h;
    (void)hipMalloc(&out, 1024 * 8);
    (void)hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((bench<MODE, K>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s threads %3d: %8lld cycles = %.1f per (MFMA%s%d FMA)\n", name, threads, h, (double)h / (4.0 * iters), (MODE & 2) ? " + " : "-less ", (MODE & 1) ? K : 0);
    (void)hipFree(out), (void)hipFree(cyc);
}

int main() {
    const int iters = 2000;
    for (int threads : {64, 256}) {
        run<1, 16>("valu only (16 FMA)", threads, iters);
        run<2, 16>("mfma only", threads, iters);
        run<3, 16>("mfma + 16 FMA interleaved", threads, iters);
        run<1, 8>("valu only (8 FMA)", threads, iters);
        run<3, 8>("mfma + 8 FMA interleaved", threads, iters);
        run<1, 32>("valu only (32 FMA)", threads, iters);
        run<3, 32>("mfma + 32 FMA interleaved", threads, iters);
    }
    run_roles<16>(1, 0, iters);
    run_roles<16>(0, 1, iters);
    run_roles<16>(1, 1, iters);
    return 0;
}
