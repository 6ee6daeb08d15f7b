// wave_placement.hip -- where do the waves of a 4-wave workgroup land?  (round 6: k_lk_track_levels puts a track's four pyramid levels on the four
// waves of a workgroup; whether those sit on four SIMDs, and which blocks share a CU, decides how the levels' searches overlap.)
// Every wave records HW_ID (wave slot, SIMD, CU, SH, SE), XCC_ID and its start / end time while spinning ~10 us, so that several workgroups are
// co-resident like in the real launch.   hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip && ./wave_placement [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) k_probe(unsigned *out, int spin) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    float x = lane;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    __syncthreads();
    const long long t1 = wall_clock64();
    if (lane == 0) {
        unsigned *o = out + 4 * (blockIdx.x * 4 + wv);
        o[0] = hw, o[1] = xcc, o[2] = (unsigned)t0, o[3] = (unsigned)(t1 - t0) + (x == 12345.f);
    }
}
int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1500;
    unsigned *d;
    hipMalloc(&d, blocks * 16 * sizeof(unsigned));
    std::vector<unsigned> h(blocks * 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, d, argc > 2 ? atoi(argv[2]) : 4000);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per block: the SIMDs of its four waves; per (xcc, se, sh, cu): which blocks
    int simd_hist[5] = {0, 0, 0, 0, 0};
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < blocks; ++b) {
        unsigned seen = 0;
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[4 * (4 * b + w)], xcc = h[4 * (4 * b + w) + 1] & 15;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            seen |= 1u << simd;
            if (w == 0) by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
            if (b < 12) printf("block %d wave %d: xcc %u se %u sh %u cu %u simd %u slot %u  t0 %u dt %u\n", b, w, xcc, se, sh, cu, simd, hw & 15, h[4 * (4 * b + w) + 2], h[4 * (4 * b + w) + 3]);
        }
        simd_hist[__builtin_popcount(seen)]++;
    }
    printf("blocks whose 4 waves sit on 1 / 2 / 3 / 4 distinct SIMDs: %d %d %d %d\n", simd_hist[1], simd_hist[2], simd_hist[3], simd_hist[4]);
    printf("distinct CUs used: %zu\n", by_cu.size());
    int shown = 0;
    for (auto &kv : by_cu) {
        if (shown++ >= 6) break;
        printf("cu key %05x:", kv.first);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    // launch ramp: when does block b start, relative to the first block?  (wall_clock64: 100 MHz, 10 ns ticks)
    {
        unsigned tmin = ~0u;
        for (int b = 0; b < blocks; ++b) tmin = h[4 * (4 * b) + 2] < tmin ? h[4 * (4 * b) + 2] : tmin;
        std::vector<unsigned> st(blocks);
        for (int b = 0; b < blocks; ++b) st[b] = h[4 * (4 * b) + 2] - tmin;
        printf("start of block b after the first block, in 10 ns ticks: ");
        for (int b = 0; b < blocks; b += blocks / 15 > 0 ? blocks / 15 : 1) printf(" b%d:%u", b, st[b]);
        unsigned mx = 0;
        for (unsigned x : st) mx = x > mx ? x : mx;
        printf("   last start %u ticks\n", mx);
    }
    // wave 0's SIMD per block: is it always the same SIMD?
    int w0[4] = {0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b) w0[(h[4 * (4 * b)] >> 4) & 3]++;
    printf("SIMD of wave 0 over all blocks: %d %d %d %d\n", w0[0], w0[1], w0[2], w0[3]);
    return 0;
}
