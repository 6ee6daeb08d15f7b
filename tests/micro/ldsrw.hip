// LDS read-modify-write loop cost per iteration for 1..4 active waves (one per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ void __launch_bounds__(256) k(long long *out, double *sink, int active_waves, int iters) {
    __shared__ double lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += 256) lds[i] = 1e-3 * (i % 13);
    __syncthreads();
    long long t0 = clock64();
    double acc = 0;
    if (wv < active_waves) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            double *C = lds + ((it * 4 + wv) & 31) * 256;
            if (V == 0) { // 4 reads, fma, 4 writes
                double c[4];
                for (int r = 0; r < 4; ++r) c[r] = C[lane + 64 * r];
                for (int r = 0; r < 4; ++r) c[r] = fma(c[r], 1.0000001, 1e-9);
                for (int r = 0; r < 4; ++r) C[lane + 64 * r] = c[r];
            } else if (V == 1) { // 4 reads only
                for (int r = 0; r < 4; ++r) acc += C[lane + 64 * r];
            } else if (V == 2) { // 4 reads, 2 mfma, 4 writes
                d4 c;
                for (int r = 0; r < 4; ++r) c[r] = C[lane + 64 * r];
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c, 0, 0, 0);
                for (int r = 0; r < 4; ++r) C[lane + 64 * r] = c[r];
            } else if (V == 4) { // 16 reads only
                double c[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = C[lane + 64 * r];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc += c[r];
            } else if (V == 5) { // 16 writes only
#pragma unroll
                for (int r = 0; r < 16; ++r) C[lane + 64 * r] = acc + r;
            } else if (V == 6) { // 8 x b128 reads
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 c[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) c[r] = *reinterpret_cast<const d2 *>(C + 2 * lane + 128 * r);
#pragma unroll
                for (int r = 0; r < 8; ++r) acc += c[r][0] + c[r][1];
            } else if (V == 3) { // 4 writes only
                for (int r = 0; r < 4; ++r) C[lane + 64 * r] = acc + r;
            }
        }
    }
    asm volatile("" : "+v"(acc));
    long long t1 = clock64();
    __syncthreads();
    if (tid == 0) out[0] = t1 - t0;
    sink[tid] = acc + lds[tid];
}
int main() {
    long long *out; double *sink;
    (void)hipMalloc(&out, 16 * 8); (void)hipMalloc(&sink, 256 * 8);
    const int iters = 512;
    for (int v = 0; v < 7; ++v)
        for (int aw = 1; aw <= 4; aw *= 2) {
            long long ho;
            for (int rep = 0; rep < 2; ++rep) {
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, sink, aw, iters);
                if (v == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, out, sink, aw, iters);
                if (v == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, out, sink, aw, iters);
                if (v == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, out, sink, aw, iters);
                if (v == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 0, 0, out, sink, aw, iters);
                if (v == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(256), 0, 0, out, sink, aw, iters);
                if (v == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(256), 0, 0, out, sink, aw, iters);
            }
            (void)hipMemcpy(&ho, out, 8, hipMemcpyDeviceToHost);
            printf("variant %d active waves %d: %.1f cycles/iter\n", v, aw, (double)ho / iters);
        }
    return 0;
}
