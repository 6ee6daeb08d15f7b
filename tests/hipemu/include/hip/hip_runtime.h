// hipemu -- a tiny single-threaded (fiber based) stand-in for <hip/hip_runtime.h>.
//
// TEST INFRASTRUCTURE ONLY.  It exists because the build container has no GPU and GPU-box minutes are
// scarce: compiling pvio_amd/csrc/*.hip against this header with g++ (-Itests/hipemu/include placed first)
// lets `pytest -m "not gpu"` exercise the kernels' indexing / reduction / control logic on the CPU and compare
// it with the oracle.  It is NOT a backend, NOT a fallback and NOT shipped: libpvio_hip.so is always built
// with hipcc for gfx950 and never sees this file; the emulated build produces a differently named test
// library (tests/hipemu/libpvio_hipemu.so) that pvio_amd/capi.py will not load by default.
//
// Model: one OS thread; every GPU thread of a block is a fiber (own stack, hand-written register switch); blocks run one after another;
// __syncthreads()/wave shuffles are cooperative yields.  Deterministic, no data races by construction
// (so it cannot find memory-model bugs -- those are left to the GPU tests).
#pragma once
#define PV_HIPEMU 1
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __constant__ static
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemu_uint3 {
    unsigned x, y, z;
};

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };

namespace hipemu {
extern const char *g_kernel_name; // name of the kernel being emulated (deadlock diagnostics)
struct Stream {
    bool capturing = false;
    std::vector<std::function<void()>> *graph = nullptr;
};
struct Graph {
    std::vector<std::function<void()>> nodes;
};
struct Event {
    double t = 0;
};
struct ThreadCtx {
    hipemu_uint3 tIdx, bIdx;
    dim3 bDim, gDim;
    int linear_tid, lane, wave;
};
extern ThreadCtx *g_cur;
extern char *g_dyn_smem;
void launch(dim3 grid, dim3 block, size_t shmem, Stream *s, std::function<void()> body);
void syncthreads();
void spin_yield(); // a polling loop hands the processor to the other fibers of the block
void wave_exchange(const void *in, void *out_all, size_t elem); // every lane contributes, receives all 64 values
double now_ms();
} // namespace hipemu

typedef hipemu::Stream *hipStream_t;
typedef hipemu::Event *hipEvent_t;
typedef hipemu::Graph *hipGraph_t;
typedef hipemu::Graph *hipGraphExec_t;

#define threadIdx (hipemu::g_cur->tIdx)
#define blockIdx (hipemu::g_cur->bIdx)
#define blockDim (hipemu::g_cur->bDim)
#define gridDim (hipemu::g_cur->gDim)
using std::isfinite;
using std::max;
using std::min;
#define warpSize 64

#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(hipemu::g_dyn_smem);
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), (shmem), (stream), [=]() { hipemu::g_kernel_name = #kernel; kernel(__VA_ARGS__); })

// ---- device intrinsics ---------------------------------------------------------------------------------
inline void __syncthreads() { hipemu::syncthreads(); }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int __mul24(int a, int b) { return a * b; } // operands are within 24 bits wherever the kernels use it
inline double __builtin_amdgcn_rsq(double x) { return (double)(float)(1.0 / std::sqrt(x)); } // deliberately low precision, like v_rsq_f64
inline long long clock64() { return (long long)(hipemu::now_ms() * 1e6); }
inline long long wall_clock64() { return (long long)(hipemu::now_ms() * 1e5); }
inline void __threadfence() {}
inline void __threadfence_block() {}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
    T all[64];
    hipemu::wave_exchange(&v, all, sizeof(T));
    int lane = hipemu::g_cur->lane;
    int base = lane & ~(width - 1);
    return all[base + (src & (width - 1))];
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    T all[64];
    hipemu::wave_exchange(&v, all, sizeof(T));
    int lane = hipemu::g_cur->lane;
    (void)width;
    return all[lane ^ mask];
}
template <typename T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
    T all[64];
    hipemu::wave_exchange(&v, all, sizeof(T));
    int lane = hipemu::g_cur->lane;
    int idx = lane + (int)delta;
    if ((idx & ~(width - 1)) != (lane & ~(width - 1)) || idx >= 64) idx = lane;
    return all[idx];
}
template <typename T>
inline T __shfl_up(T v, unsigned delta, int width = 64) {
    T all[64];
    hipemu::wave_exchange(&v, all, sizeof(T));
    int lane = hipemu::g_cur->lane;
    int idx = lane - (int)delta;
    if (idx < (lane & ~(width - 1))) idx = lane;
    return all[idx];
}
inline int __builtin_amdgcn_readlane(int v, int src) {
    int all[64];
    hipemu::wave_exchange(&v, all, sizeof(int));
    return all[src & 63];
}
inline int __builtin_amdgcn_readfirstlane(int v) {
    int all[64];
    hipemu::wave_exchange(&v, all, sizeof(int));
    return all[0];
}
// v_mov_b32 with a DPP modifier: the four lane pairings the kernels use (ctrl as in the ISA manual), all lanes enabled
inline int __builtin_amdgcn_update_dpp(int /*old*/, int src, int ctrl, int row_mask, int bank_mask, bool /*bound_ctrl*/) {
    const int lane = hipemu::g_cur->lane;
    int from;
    if (ctrl == 0xB1) from = lane ^ 1;                                // quad_perm [1,0,3,2]
    else if (ctrl == 0x4E) from = lane ^ 2;                           // quad_perm [2,3,0,1]
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));    // row_half_mirror
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15)); // row_mirror
    else std::abort();
    if (row_mask != 0xF || bank_mask != 0xF) std::abort();
    return __shfl(src, from);
}
inline float __uint_as_float(unsigned u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline unsigned __float_as_uint(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}
// v_perm_b32: byte k of the result is byte sel[k] of the 8 bytes {s1 (0-3), s0 (4-7)}; selectors >= 8 (constants / sign fills) are not used by the kernels
inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long both = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int k = 0; k < 4; ++k) {
        const unsigned b = (sel >> (8 * k)) & 255u;
        if (b > 7) std::abort();
        r |= (unsigned)((both >> (8 * b)) & 255u) << (8 * k);
    }
    return r;
}
// v_dot2_i32_i16
typedef short hipemu_short2 __attribute__((vector_size(4)));
inline int __builtin_amdgcn_sdot2(hipemu_short2 a, hipemu_short2 b, int c, bool /*clamp*/) { return (int)a[0] * (int)b[0] + (int)a[1] * (int)b[1] + c; }
// the lanes of a wave run in lockstep on the GPU; here they are fibers that run one after another: this is where they meet (the kernels
// call it in front of a release by lane 0, so that the release covers what every lane of the wave has stored)
inline void __builtin_amdgcn_wave_barrier() { (void)__shfl(0, 0); }
inline void __builtin_amdgcn_s_sleep(int) { hipemu::spin_yield(); } // a polling loop lets the other fibers run
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 4
template <typename T>
inline T __hip_atomic_load(const T *p, int, int) { return *reinterpret_cast<const volatile T *>(p); }
template <typename T>
inline void __hip_atomic_store(T *p, T v, int, int) { *reinterpret_cast<volatile T *>(p) = v; }
template <typename T>
inline T __hip_atomic_fetch_add(T *p, T v, int, int) {
    T old = *reinterpret_cast<volatile T *>(p);
    *reinterpret_cast<volatile T *>(p) = old + v;
    return old;
}
inline double unsafeAtomicAdd(double *p, double v) {
    double old = *p;
    *p = old + v;
    return old;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline unsigned long long __ballot(int pred) {
    int all[64];
    hipemu::wave_exchange(&pred, all, sizeof(int));
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i)
        if (all[i]) m |= 1ull << i;
    return m;
}
template <typename T>
inline T atomicAdd(T *p, T v) {
    T old = *p;
    *p = old + v;
    return old;
}
inline unsigned atomicInc(unsigned *p, unsigned lim) {
    unsigned old = *p;
    *p = old >= lim ? 0 : old + 1;
    return old;
}
template <typename T>
inline T atomicMax(T *p, T v) {
    T old = *p;
    if (v > old) *p = v;
    return old;
}
template <typename T>
inline T atomicExch(T *p, T v) {
    T old = *p;
    *p = v;
    return old;
}
// v_mfma_f64_16x16x4_f64: D = A(16x4) B(4x16) + C.  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
// receives D[(l>>4)+4*r][l&15] in element r (cdna_hip_programming.md section 3, f64 layout).
typedef double hipemu_double4 __attribute__((vector_size(32)));
inline hipemu_double4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hipemu_double4 c, int, int, int) {
    double A[64], B[64];
    hipemu::wave_exchange(&a, A, sizeof(double));
    hipemu::wave_exchange(&b, B, sizeof(double));
    int lane = hipemu::g_cur->lane;
    int col = lane & 15;
    hipemu_double4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fma(A[16 * k + row], B[16 * k + col], acc);
        d[r] = acc;
    }
    return d;
}

// ---- host API -------------------------------------------------------------------------------------------
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; } // (the launch checks the 160 KiB limit itself)
inline hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int) { return hipSuccess; }
struct hipDeviceProp_t {
    char name[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    char gcnArchName[256];
};
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::snprintf(p->name, sizeof p->name, "hipemu (CPU fibers)");
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "hipemu");
    p->multiProcessorCount = 8; // keep emulated grids small
    p->totalGlobalMem = 1ull << 34;
    return hipSuccess;
}
// every device allocation carries 256 guard bytes behind it; hipemu_check_guards() (after every kernel, at every free) aborts with the
// size of the allocation a kernel wrote past
namespace hipemu {
struct Guarded { char *p; size_t n; };
std::vector<Guarded> &guarded();
void check_guards(const char *where);
} // namespace hipemu
template <typename T>
inline hipError_t hipMalloc(T **p, size_t n) {
    char *q = (char *)std::malloc((n ? n : 1) + 512);
    if (q) {
        std::memset((void *)q, 0x5C, 256);
        q += 256;
        std::memset((void *)q, 0xA5, n); // poison: catches reads of never-written device memory
        std::memset((void *)(q + n), 0x5C, 256);
        hipemu::guarded().push_back({q, n});
    }
    *p = (T *)q;
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T>
inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) {
    return hipMalloc(p, n);
}
inline hipError_t hipFree(void *p) {
    hipemu::check_guards("hipFree");
    auto &g = hipemu::guarded();
    for (size_t i = 0; i < g.size(); ++i)
        if (g[i].p == p) {
            g.erase(g.begin() + (std::ptrdiff_t)i);
            std::free((char *)p - 256);
            return hipSuccess;
        }
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline void hipemu_enqueue(hipStream_t s, std::function<void()> f) {
    if (s && s->capturing)
        s->graph->push_back(std::move(f));
    else
        f();
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    std::memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st = nullptr) {
    hipemu_enqueue(st, [=]() { std::memcpy(d, s, n); });
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) {
    std::memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr) {
    hipemu_enqueue(st, [=]() { std::memset(d, v, n); });
    return hipSuccess;
}
inline hipError_t hipStreamCreate(hipStream_t *s) {
    *s = new hipemu::Stream();
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
#define hipStreamNonBlocking 1
inline hipError_t hipStreamDestroy(hipStream_t s) {
    delete s;
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new hipemu::Event();
    return hipSuccess;
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
    e->t = hipemu::now_ms();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = (float)(b->t - a->t);
    return hipSuccess;
}
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone, hipStreamCaptureStatusActive, hipStreamCaptureStatusInvalidated };
inline hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st) {
    *st = s->capturing ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
inline hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
    s->capturing = true;
    s->graph = new std::vector<std::function<void()>>();
    return hipSuccess;
}
inline hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g) {
    *g = new hipemu::Graph();
    (*g)->nodes = std::move(*s->graph);
    delete s->graph;
    s->graph = nullptr;
    s->capturing = false;
    return hipSuccess;
}
inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, void *, void *, size_t) {
    *e = new hipemu::Graph(*g);
    return hipSuccess;
}
inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
    for (auto &f : e->nodes) f();
    return hipSuccess;
}
inline hipError_t hipGraphDestroy(hipGraph_t g) {
    delete g;
    return hipSuccess;
}
inline hipError_t hipGraphExecDestroy(hipGraphExec_t g) {
    delete g;
    return hipSuccess;
}
