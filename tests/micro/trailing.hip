// Microbenchmark of the trailing-update phase of the dense kernel's panel Cholesky (tile layout + MFMA), isolated.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tile_base(int bi, int bk) { return (((bi * (bi + 1)) >> 1) + bk) << 8; }
__device__ __forceinline__ int tile_off(int r, int c) { return (r << 4) + (c ^ ((r >> 1) & 7)); }
template <int VARIANT>
__global__ void __launch_bounds__(512) k(long long *out, double *sink, int Pp) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, nbk = (Pp + 16) >> 4, LDV = nbk << 4;
    double *Lp = lds, *A = lds + 8 * LDV;
    for (int i = tid; i < 8 * LDV + nbk * (nbk + 1) / 2 * 256; i += blockDim.x) lds[i] = 1e-3 * (i % 13);
    __syncthreads();
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
    int coff[4];
    for (int r = 0; r < 4; ++r) coff[r] = tile_off(lk + 4 * r, lr);
    long long t0 = clock64();
    long long tfirst = 0;
    long long acc_t[6] = {0, 0, 0, 0, 0, 0};
    for (int j0 = 0; j0 < Pp; j0 += 8) {
        const int k0 = j0 + 8, b0 = k0 >> 4;
        const double *La = Lp + lk * LDV + lr, *Lb = Lp + (4 + lk) * LDV + lr;
        if (VARIANT == 0) { // as in the kernel: software-pipelined, round-robin
            int bi = b0, q = wv;
            while (bi < nbk && q > bi - b0) q -= bi - b0 + 1, ++bi;
            double a0 = 0, a1 = 0, p0 = 0, p1 = 0;
            d4 cin = {0, 0, 0, 0};
            double *C = A;
            bool st = false;
            if (bi < nbk) {
                C = A + tile_base(bi, b0 + q);
                a0 = La[16 * bi], a1 = Lb[16 * bi], p0 = La[16 * (b0 + q)], p1 = Lb[16 * (b0 + q)];
                for (int r = 0; r < 4; ++r) cin[r] = C[coff[r]];
                st = 16 * (b0 + q) + lr >= k0;
            }
            while (bi < nbk) {
                int bi2 = bi, q2 = q + 4;
                while (bi2 < nbk && q2 > bi2 - b0) q2 -= bi2 - b0 + 1, ++bi2;
                double na0 = 0, na1 = 0, np0 = 0, np1 = 0;
                d4 ncin = {0, 0, 0, 0};
                double *nC = A;
                bool nst = false;
                if (bi2 < nbk) {
                    nC = A + tile_base(bi2, b0 + q2);
                    na0 = La[16 * bi2], na1 = Lb[16 * bi2], np0 = La[16 * (b0 + q2)], np1 = Lb[16 * (b0 + q2)];
                    for (int r = 0; r < 4; ++r) ncin[r] = nC[coff[r]];
                    nst = 16 * (b0 + q2) + lr >= k0;
                }
                d4 cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, p0, cin, 0, 0, 0);
                cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, p1, cacc, 0, 0, 0);
                if (st)
                    for (int r = 0; r < 4; ++r) C[coff[r]] = cacc[r];
                bi = bi2, q = q2, a0 = na0, a1 = na1, p0 = np0, p1 = np1, cin = ncin, C = nC, st = nst;
            }
        } else if (VARIANT == 1) { // row-block per wave (bi = b0 + wv, + 4 ...), a operands kept, simple loops
            for (int bi = b0 + wv; bi < nbk; bi += 4) {
                const double a0 = -La[16 * bi], a1 = -Lb[16 * bi];
                for (int bk = b0; bk <= bi; ++bk) {
                    double *C = A + tile_base(bi, bk);
                    const double p0 = La[16 * bk], p1 = Lb[16 * bk];
                    d4 cacc;
                    for (int r = 0; r < 4; ++r) cacc[r] = C[coff[r]];
                    cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, p0, cacc, 0, 0, 0);
                    cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, p1, cacc, 0, 0, 0);
                    if (16 * bk + lr >= k0)
                        for (int r = 0; r < 4; ++r) C[coff[r]] = cacc[r];
                }
            }
        } else if (VARIANT >= 3 && VARIANT <= 6) {
            const int nw = blockDim.x >> 6;
            for (int bi = b0 + wv; bi < nbk; bi += nw) {
                const double a0 = -La[16 * bi], a1 = -Lb[16 * bi];
                for (int bk = b0; bk <= bi; ++bk) {
                    double *C = A + tile_base(bi, bk);
                    const double p0 = La[16 * bk], p1 = Lb[16 * bk];
                    d4 cacc;
                    for (int r = 0; r < 4; ++r) cacc[r] = C[coff[r]];
                    if (VARIANT == 3) {
                        for (int r = 0; r < 4; ++r) cacc[r] = fma(a0, p0, fma(a1, p1, cacc[r]));
                    } else if (VARIANT == 4) {
                        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0 + a1, p0 + p1, cacc, 0, 0, 0);
                    } else {
                        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, p0, cacc, 0, 0, 0);
                        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, p1, cacc, 0, 0, 0);
                    }
                    if (VARIANT == 5) {
                        if (cacc[0] == 12345.678) C[coff[0]] = cacc[1] + cacc[2] + cacc[3];
                    } else if (16 * bk + lr >= k0)
                        for (int r = 0; r < 4; ++r) C[coff[r]] = cacc[r];
                }
            }
        } else if (VARIANT >= 7 && VARIANT <= 10) { // batches of NB tiles per wave: all loads, then all MFMAs, then all stores
            constexpr int NB = (VARIANT == 7 || VARIANT == 9) ? 4 : 2;
            constexpr double SG = VARIANT >= 9 ? 1.0 : -1.0;
            int bi = b0, q = wv;
            long long tprev = clock64();
            while (bi < nbk && q > bi - b0) q -= bi - b0 + 1, ++bi;
            while (bi < nbk) {
                double *C[NB];
                double a0[NB], a1[NB], p0[NB], p1[NB];
                d4 acc[NB];
                bool st[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const bool valid = bi < nbk;
                    const int bic = valid ? bi : nbk - 1, bkc = valid ? b0 + q : nbk - 1;
                    C[u] = A + tile_base(bic, bkc);
                    a0[u] = La[16 * bic], a1[u] = Lb[16 * bic], p0[u] = La[16 * bkc], p1[u] = Lb[16 * bkc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[u][r] = C[u][coff[r]];
                    st[u] = valid && (16 * bkc + lr >= k0);
                    q += 4;
                    while (bi < nbk && q > bi - b0) q -= bi - b0 + 1, ++bi;
                }
                long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                if (VARIANT == 9) {
                    s0 = clock64();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    s1 = clock64();
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(SG * a0[u], p0[u], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NB; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(SG * a1[u], p1[u], acc[u], 0, 0, 0);
                if (VARIANT == 9) {
#pragma unroll
                    for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(acc[u]));
                    s2 = clock64();
                }
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (st[u]) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) C[u][coff[r]] = acc[u][r];
                    }
                if (VARIANT == 9) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    s3 = clock64();
                    acc_t[0] += s0 - tprev, acc_t[1] += s1 - s0, acc_t[2] += s2 - s1, acc_t[3] += s3 - s2, acc_t[4] += 1;
                    tprev = s3;
                }
            }
        } else if (VARIANT == 11 || VARIANT == 12) { // accumulator-layout tiles, b128 accesses, batches
            typedef double d2 __attribute__((ext_vector_type(2)));
            constexpr int NB = VARIANT == 11 ? 4 : 2;
            int bi = b0, q = wv;
            long long tprev = clock64();
            while (bi < nbk && q > bi - b0) q -= bi - b0 + 1, ++bi;
            while (bi < nbk) {
                double *C[NB];
                d2 av[NB], pv[NB], c01[NB], c23[NB];
                bool st[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const bool valid = bi < nbk;
                    const int bic = valid ? bi : nbk - 1, bkc = valid ? b0 + q : nbk - 1;
                    C[u] = A + tile_base(bic, bkc) + 4 * lane;
                    av[u] = *reinterpret_cast<const d2 *>(Lp + 8 * (16 * bic + lr) + 2 * lk);
                    pv[u] = *reinterpret_cast<const d2 *>(Lp + 8 * (16 * bkc + lr) + 2 * lk);
                    c01[u] = *reinterpret_cast<const d2 *>(C[u]);
                    c23[u] = *reinterpret_cast<const d2 *>(C[u] + 2);
                    st[u] = valid && (16 * bkc + lr >= k0);
                    q += 4;
                    while (bi < nbk && q > bi - b0) q -= bi - b0 + 1, ++bi;
                }
                long long s0 = clock64();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                long long s1 = clock64();
                d4 acc[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    acc[u][0] = c01[u][0], acc[u][1] = c01[u][1], acc[u][2] = c23[u][0], acc[u][3] = c23[u][1];
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][0], pv[u][0], acc[u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][1], pv[u][1], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(acc[u]));
                long long s2 = clock64();
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (st[u]) {
                        d2 w0, w1;
                        w0[0] = acc[u][0], w0[1] = acc[u][1], w1[0] = acc[u][2], w1[1] = acc[u][3];
                        *reinterpret_cast<d2 *>(C[u]) = w0;
                        *reinterpret_cast<d2 *>(C[u] + 2) = w1;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                long long s3 = clock64();
                acc_t[0] += s0 - tprev, acc_t[1] += s1 - s0, acc_t[2] += s2 - s1, acc_t[3] += s3 - s2, acc_t[4] += 1;
                tprev = s3;
            }
        } else if (VARIANT == 2) { // column-block per wave, two tiles in flight (explicit unroll by 2 over bi)
            for (int bk = b0 + wv; bk < nbk; bk += 4) {
                const double p0 = La[16 * bk], p1 = Lb[16 * bk];
                const bool st = 16 * bk + lr >= k0;
                int bi = bk;
                for (; bi + 1 < nbk; bi += 2) {
                    double *C0 = A + tile_base(bi, bk), *C1 = A + tile_base(bi + 1, bk);
                    const double a0 = -La[16 * bi], a1 = -Lb[16 * bi], e0 = -La[16 * bi + 16], e1 = -Lb[16 * bi + 16];
                    d4 x, y;
                    for (int r = 0; r < 4; ++r) x[r] = C0[coff[r]], y[r] = C1[coff[r]];
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, p0, x, 0, 0, 0);
                    y = __builtin_amdgcn_mfma_f64_16x16x4f64(e0, p0, y, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, p1, x, 0, 0, 0);
                    y = __builtin_amdgcn_mfma_f64_16x16x4f64(e1, p1, y, 0, 0, 0);
                    if (st)
                        for (int r = 0; r < 4; ++r) C0[coff[r]] = x[r], C1[coff[r]] = y[r];
                }
                if (bi < nbk) {
                    double *C0 = A + tile_base(bi, bk);
                    const double a0 = -La[16 * bi], a1 = -Lb[16 * bi];
                    d4 x;
                    for (int r = 0; r < 4; ++r) x[r] = C0[coff[r]];
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, p0, x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, p1, x, 0, 0, 0);
                    if (st)
                        for (int r = 0; r < 4; ++r) C0[coff[r]] = x[r];
                }
            }
        }
        { long long b0t = clock64(); __syncthreads(); long long b1t = clock64(); acc_t[5] += b1t - b0t; }
        if (j0 == 0) tfirst = clock64() - t0;
    }
    long long t1 = clock64();
    if (tid == 0) { out[0] = t1 - t0, out[1] = tfirst; for (int i = 0; i < 6; ++i) out[2 + i] = acc_t[i]; }
    sink[tid] = A[tid] + Lp[tid];
}
int main() {
    long long *out; double *sink;
    (void)hipMalloc(&out, 16 * 8); (void)hipMalloc(&sink, 256 * 8);
    for (int Pp : {64, 152}) {
        const int nbk = (Pp + 16) >> 4, LDV = nbk << 4;
        const size_t bytes = (8 * LDV + nbk * (nbk + 1) / 2 * 256) * 8;
        (void)hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<9>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<10>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<11>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void *)k<12>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        for (int var = 9; var < 13; ++var) {
            long long ho[2];
            for (int rep = 0; rep < 2; ++rep) {
                if (var == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(512), bytes, 0, out, sink, Pp);
                if (var == 7) hipLaunchKernelGGL(k<7>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 8) hipLaunchKernelGGL(k<8>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 9) hipLaunchKernelGGL(k<9>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 10) hipLaunchKernelGGL(k<10>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 11) hipLaunchKernelGGL(k<11>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
                if (var == 12) hipLaunchKernelGGL(k<12>, dim3(1), dim3(256), bytes, 0, out, sink, Pp);
            }
            (void)hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
            printf("Pp=%d variant %d: all panels %lld cycles, first panel %lld\n", Pp, var, ho[0], ho[1]);
            if (var == 9 || var >= 11) { long long h6[8]; (void)hipMemcpy(h6, out, 64, hipMemcpyDeviceToHost); printf("   wave 0 totals over %lld batches: decode+issue(+barrier) %lld, wait %lld, mfma %lld, store %lld, barrier %lld\n", h6[6], h6[2], h6[3], h6[4], h6[5], h6[7]); }
        }
    }
    return 0;
}
