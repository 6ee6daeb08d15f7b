// Test infrastructure: fills the LDS of every CU with a 64-bit pattern (LDS is not cleared between kernels: a kernel that reads LDS it
// has not written sees whatever the previous workgroup on that CU left).  Used by tests/micro/lds_poison_probe.py and the -m gpu
// test `test_gpu_solver_reads_no_stale_lds`: a solve must not change when the LDS it starts on is NaN / huge / zero.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o liblds_poison.so lds_poison.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_poison(uint64_t pattern, int n_words) {
    extern __shared__ uint64_t lds[];
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    // keep the workgroup resident for a moment so that the 4 x 256 workgroups spread over all CUs instead of queueing on a few
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
    if (lds[(threadIdx.x * 7) % n_words] != pattern) asm volatile("s_trap 2");
}

extern "C" int lds_poison(uint64_t pattern) {
    const int bytes = 160 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_poison, dim3(1024), dim3(1024), bytes, 0, pattern, bytes / 8);
    return hipDeviceSynchronize() == hipSuccess ? 0 : -2;
}
