// Microbenchmark (gfx950): what a cross-XCD producer -> consumer hand-over costs as (a) a kernel boundary inside a hipGraph and (b) a grid barrier
// inside one persistent kernel -- the number DESIGN.md section 5 needs to say whether a persistent iteration kernel could pay.
//
// Both forms run STEPS steps on G workgroups of 256 threads (G <= the CU count: all resident).  In every step workgroup w reads a cache line that
// workgroup (w + 1) % G wrote in the previous step (the neighbour lives on another XCD: workgroups are dealt round-robin to the eight XCDs), adds one
// and writes its own line -- the shape of "every workgroup reads what the previous kernel left".
//   (a) kernels: STEPS launches captured in one graph, the boundary provides release / acquire
//   (b) persistent: one launch; after its write a workgroup does an agent-scope release fence and arrives at a counter; the last arrival
//       advances a generation word; the others poll it (agent-scope acquire), then invalidate and go on.  Counter and generation live in memory.
// Prints microseconds per step of each form and checks that both computed the same values.
// build: hipcc --offload-arch=gfx950 -O2 -o grid_sync grid_sync.hip        run: ./grid_sync [G] [STEPS]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));      \
            return 1;                                                         \
        }                                                                     \
    } while (0)

constexpr int kLine = 16; // doubles per workgroup (one 128-byte line)

__global__ void __launch_bounds__(256) k_step(double *buf, int G, int step) {
    const int w = blockIdx.x, src = (w + 1) % G;
    const double *in = buf + ((size_t)((step + 1) & 1) * G + src) * kLine;
    double *out = buf + ((size_t)(step & 1) * G + w) * kLine;
    if (threadIdx.x < kLine) out[threadIdx.x] = in[threadIdx.x] + 1.0;
}

__global__ void __launch_bounds__(256) k_persistent(double *buf, int G, int steps, int *counter, int *gen) {
    const int w = blockIdx.x, src = (w + 1) % G;
    for (int step = 0; step < steps; ++step) {
        const double *in = buf + ((size_t)((step + 1) & 1) * G + src) * kLine;
        double *out = buf + ((size_t)(step & 1) * G + w) * kLine;
        if (threadIdx.x < kLine) out[threadIdx.x] = __builtin_nontemporal_load(in + threadIdx.x) + 1.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // this workgroup's line is in memory before it arrives
            const int arrived = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == G * (step + 1) - 1) {
                __hip_atomic_store(gen, step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < step + 1) __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // nothing stale of the neighbours' lines survives in this CU's / XCD's caches
        }
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? std::atoi(argv[1]) : 240, STEPS = argc > 2 ? std::atoi(argv[2]) : 400;
    double *buf;
    int *sync;
    CK(hipMalloc(&buf, sizeof(double) * 2 * G * kLine));
    CK(hipMalloc(&sync, 2 * sizeof(int)));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // (a) graph of kernels
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < STEPS; ++s) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, st, buf, G, s);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    std::vector<double> ha((size_t)2 * G * kLine), hb(ha.size());
    double us_a = 1e30, us_b = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(buf, 0, sizeof(double) * 2 * G * kLine, st));
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        us_a = std::min(us_a, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / STEPS);
    }
    CK(hipMemcpy(ha.data(), buf, ha.size() * sizeof(double), hipMemcpyDeviceToHost));
    // (b) one persistent kernel
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(buf, 0, sizeof(double) * 2 * G * kLine, st));
        CK(hipMemsetAsync(sync, 0, 2 * sizeof(int), st));
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, st, buf, G, STEPS, sync, sync + 1);
        CK(hipStreamSynchronize(st));
        us_b = std::min(us_b, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / STEPS);
    }
    CK(hipMemcpy(hb.data(), buf, hb.size() * sizeof(double), hipMemcpyDeviceToHost));
    bool same = true;
    for (size_t i = 0; i < ha.size(); ++i) same &= ha[i] == hb[i];
    std::printf("G = %d workgroups, %d steps: kernel boundary in a graph %.2f us per step | grid barrier in a persistent kernel %.2f us per step | results %s (last value %.0f)\n", G,
                STEPS, us_a, us_b, same ? "identical" : "DIFFER", ha[((size_t)((STEPS - 1) & 1) * G) * kLine]);
    return same ? 0 : 1;
}
