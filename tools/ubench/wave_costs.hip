// Microbenchmark (gfx950): what the building blocks of k_dense's panel loop cost ONE wave on its own SIMD -- the numbers DESIGN.md section 5 prices
// the factor wave's chain with.  Shader-clock cycles (s_memtime), lane 0 of the measuring wave reports.
//   1. FP64 VALU: dependent chains against independent streams of v_fma_f64 / v_mul_f64, v_rsq_f64 and v_rcp_f64, the fast_rsqrt chain
//   2. LDS: dependent ds_read_b128 (latency), 12 ds_write_b128 + wait at row stride 64 B (k_dense's L rows: four lanes per bank group) and at 80 B
//   3. a hand-over between two waves on different SIMDs through a counter in LDS, as dense_signal_set / dense_wait do it: ping-pong, half a round
//      trip = one hand-over; bare, with s_sleep(1) in the poll, and with a payload (12 ds_write_b128 before the release, one dependent read behind
//      the acquire)
// build: hipcc --offload-arch=gfx950 -O2 -o wave_costs wave_costs.hip        run: ./wave_costs
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                                    \
        }                                                                \
    } while (0)

constexpr int kReps = 2000;
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ long long now() { return clock64(); }
#define PIN(x) asm volatile("" : "+v"(x))

// ---- 1. FP64 VALU ----
template <int MODE> __global__ void __launch_bounds__(64) k_valu(double *out, long long *cyc, double a, double b) {
    double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const long long t0 = now();
    for (int r = 0; r < kReps; ++r) {
        if (MODE == 0) { // 8 dependent FMAs
#pragma unroll
            for (int k = 0; k < 8; ++k) x0 = fma(x0, a, b);
        } else if (MODE == 1) { // 8 independent FMAs
            x0 = fma(x0, a, b), x1 = fma(x1, a, b), x2 = fma(x2, a, b), x3 = fma(x3, a, b), x4 = fma(x4, a, b), x5 = fma(x5, a, b), x6 = fma(x6, a, b), x7 = fma(x7, a, b);
        } else if (MODE == 2) { // 8 dependent MULs
#pragma unroll
            for (int k = 0; k < 8; ++k) x0 = x0 * a;
        } else if (MODE == 3) { // 8 dependent v_rsq_f64
#pragma unroll
            for (int k = 0; k < 8; ++k) x0 = __builtin_amdgcn_rsq(x0);
        } else if (MODE == 4) { // 8 independent v_rsq_f64
            x0 = __builtin_amdgcn_rsq(x0), x1 = __builtin_amdgcn_rsq(x1), x2 = __builtin_amdgcn_rsq(x2), x3 = __builtin_amdgcn_rsq(x3);
            x4 = __builtin_amdgcn_rsq(x4), x5 = __builtin_amdgcn_rsq(x5), x6 = __builtin_amdgcn_rsq(x6), x7 = __builtin_amdgcn_rsq(x7);
        } else if (MODE == 5) { // 8 dependent v_rcp_f64
#pragma unroll
            for (int k = 0; k < 8; ++k) x0 = __builtin_amdgcn_rcp(x0);
        } else if (MODE == 6) { // 8 x the fast_rsqrt chain (rsq + two Newton steps), dependent
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double y = __builtin_amdgcn_rsq(x0);
                y = y * fma(-0.5 * x0, y * y, 1.5);
                y = y * fma(-0.5 * x0, y * y, 1.5);
                x0 = y + b;
            }
        } else if (MODE == 7) { // 8 dependent FMAs each followed by 3 independent ones (a chain with filler, as in the pivot loop)
#pragma unroll
            for (int k = 0; k < 8; ++k) x0 = fma(x0, a, b), x1 = fma(x1, a, b), x2 = fma(x2, a, b), x3 = fma(x3, a, b);
        } else if (MODE == 8) { // 8 dependent 32-bit selects on a 64-bit value (v_cndmask pairs)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x0 = (x1 > (double)k) ? x0 : x2;
                PIN(x0);
            }
        }
        PIN(x0);
    }
    const long long t1 = now();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- 2. LDS ----
template <int MODE> __global__ void __launch_bounds__(64) k_lds(double *out, long long *cyc, int stride_doubles) {
    __shared__ __attribute__((aligned(16))) double sm[64 * 16 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 16 + 64; i += 64) sm[i] = 0.0;
    __syncthreads();
    double acc = 0;
    int idx = 0;
    const long long t0 = now();
    for (int r = 0; r < kReps; ++r) {
        if (MODE == 0) { // 4 dependent ds_read_b128 (the address of the next one comes out of the last one: the table holds zeros)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const d2 v = *reinterpret_cast<const d2 *>(sm + 2 * lane + idx);
                idx = (int)v[0];
                acc += v[1];
            }
        } else { // MODE 1: 12 ds_write_b128 of this lane's "rows" (3 rows x 4 chunks) + wait until they have landed
            d2 v;
            v[0] = acc, v[1] = (double)r;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int h = 0; h < 4; ++h) *reinterpret_cast<d2 *>(sm + (size_t)(lane % 21 + 21 * t) * stride_doubles + 2 * h) = v;
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
            asm volatile("" ::: "memory");
        }
    }
    const long long t1 = now();
    out[lane] = acc + sm[lane] + idx;
    if (lane == 0) cyc[0] = t1 - t0;
}

// ---- 3. hand-over between two waves through LDS ----
__device__ __forceinline__ int wait_for(int *flag, int target, bool sleep) {
    int val;
    while ((val = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
        if (sleep) __builtin_amdgcn_s_sleep(1);
    return val;
}
__device__ __forceinline__ void signal(int *flag, int val) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool SLEEP, bool PAYLOAD> __global__ void __launch_bounds__(256) k_hop(double *out, long long *cyc) {
    __shared__ int flags[2];
    __shared__ __attribute__((aligned(16))) double rows[2][192 * 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 2 * 192 * 8; i += 256) (&rows[0][0])[i] = 0.0;
    __syncthreads();
    double acc = 0;
    if (wv >= 2) return; // waves 0 and 1 sit on different SIMDs; 2 and 3 leave
    const long long t0 = now();
    for (int r = 1; r <= kReps; ++r) {
        const int me = wv, other = 1 - wv;
        if (wv == 1) { // wave 1 waits first
            wait_for(&flags[0], r, SLEEP);
            if (PAYLOAD) acc += rows[0][8 * lane + (r & 7)];
        }
        if (PAYLOAD) {
            d2 v;
            v[0] = acc, v[1] = (double)r;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int h = 0; h < 4; ++h) *reinterpret_cast<d2 *>(&rows[me][(size_t)(lane + 64 * t) * 8 + 2 * h]) = v;
        }
        signal(&flags[me], r);
        if (wv == 0) {
            wait_for(&flags[1], r, SLEEP);
            if (PAYLOAD) acc += rows[1][8 * lane + (r & 7)];
        }
        (void)other;
    }
    const long long t1 = now();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out;
    long long *cyc;
    CK(hipMalloc(&out, 256 * sizeof(double)));
    CK(hipMalloc(&cyc, sizeof(long long)));
    long long h = 0;
    auto get = [&]() -> double {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
        return (double)h / kReps;
    };
#define RUN_VALU(M, what, per)                                                            \
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_valu<M>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9); \
    std::printf("%-62s %7.1f cycles per instruction (%s)\n", what, get() / (per), #per " per repetition");
    RUN_VALU(0, "v_fma_f64, dependent chain", 8)
    RUN_VALU(1, "v_fma_f64, independent", 8)
    RUN_VALU(7, "v_fma_f64, one dependent + three independent per step", 32)
    RUN_VALU(2, "v_mul_f64, dependent chain", 8)
    RUN_VALU(3, "v_rsq_f64, dependent chain", 8)
    RUN_VALU(4, "v_rsq_f64, independent", 8)
    RUN_VALU(5, "v_rcp_f64, dependent chain", 8)
    RUN_VALU(6, "fast_rsqrt chain (rsq + two Newton steps) + 1 add, per chain", 8)
    RUN_VALU(8, "64-bit select (two v_cndmask_b32), dependent, per select", 8)
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_lds<0>, dim3(1), dim3(64), 0, 0, out, cyc, 8);
    std::printf("%-62s %7.1f cycles per read\n", "ds_read_b128, dependent (address from the last read)", get() / 4);
    for (int stride : {8, 10}) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_lds<1>, dim3(1), dim3(64), 0, 0, out, cyc, stride);
        std::printf("12 ds_write_b128 per lane + wait, row stride %3d B              %7.1f cycles per group of 12\n", stride * 8, get());
    }
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_hop<false, false>), dim3(1), dim3(256), 0, 0, out, cyc);
    std::printf("%-62s %7.1f cycles per hand-over\n", "LDS counter hand-over between two waves, tight poll", get() / 2);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_hop<true, false>), dim3(1), dim3(256), 0, 0, out, cyc);
    std::printf("%-62s %7.1f cycles per hand-over\n", "LDS counter hand-over, s_sleep(1) in the poll", get() / 2);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_hop<false, true>), dim3(1), dim3(256), 0, 0, out, cyc);
    std::printf("%-62s %7.1f cycles per hand-over\n", "hand-over with 12 ds_write_b128 before + 1 read behind, tight", get() / 2);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_hop<true, true>), dim3(1), dim3(256), 0, 0, out, cyc);
    std::printf("%-62s %7.1f cycles per hand-over\n", "hand-over with 12 ds_write_b128 before + 1 read behind, sleep", get() / 2);
    return 0;
}
