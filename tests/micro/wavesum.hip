// wave reduction variants: __shfl_xor butterfly (ds_bpermute) vs DPP rows + lane reads.  cycles per reduction, one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double readlane_f64(double x, int src) {
    union { double d; int i[2]; } u;
    u.d = x;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src), u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
    return u.d;
}
template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
    union { double d; int i[2]; } u, t;
    u.d = x;
    t.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xF, 0xF, true), t.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xF, 0xF, true);
    return t.d;
}
__device__ __forceinline__ double sum_shfl(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double sum_dpp(double v) {
    v += dpp<0xB1>(v), v += dpp<0x4E>(v), v += dpp<0x141>(v), v += dpp<0x140>(v);
    return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}
__device__ __forceinline__ double sum_dpp_bcast(double v) { // rows via DPP, then row_bcast15 / row_bcast31 and one lane read
    v += dpp<0xB1>(v), v += dpp<0x4E>(v), v += dpp<0x141>(v), v += dpp<0x140>(v);
    double t = v + __shfl_xor(v, 16);
    return t + __shfl_xor(t, 32);
}
__global__ void __launch_bounds__(256) k(long long *out, double *sink, const double *src) {
    double a = src[threadIdx.x];
    long long t0, t1;
    asm volatile("" : "+v"(a));
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) a = sum_shfl(a) * 1e-3 + threadIdx.x;
    asm volatile("" : "+v"(a));
    t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) a = sum_dpp(a) * 1e-3 + threadIdx.x;
    asm volatile("" : "+v"(a));
    t1 = clock64();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) a = sum_dpp_bcast(a) * 1e-3 + threadIdx.x;
    asm volatile("" : "+v"(a));
    t1 = clock64();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    sink[threadIdx.x] = a;
}
int main() {
    long long *out; double *sink, *src;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 2048); (void)hipMalloc(&src, 2048);
    double h[256]; for (int i = 0; i < 256; ++i) h[i] = 1.0 + i;
    (void)hipMemcpy(src, h, 2048, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, sink, src);
    long long ho[3]; (void)hipMemcpy(ho, out, 24, hipMemcpyDeviceToHost);
    printf("wave sum of a double: shfl_xor butterfly %.0f cycles | DPP rows + 4 lane reads %.0f | DPP rows + 2 shfl %.0f\n", ho[0] / 256.0, ho[1] / 256.0, ho[2] / 256.0);
    return 0;
}
