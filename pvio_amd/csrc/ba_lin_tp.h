// ba_lin_tp.h -- the landmark role of k_linearize for LARGE windows, as a throughput kernel (round 6; included by ba_kernels.hip inside namespace pvba).
//
// What it replaces: the round 2-5 matrix-core form kept every landmark of a chunk as a record of 40 N + 46 doubles in LDS (446 at N = 10, 1246 at
// N = 30: cleared, written and re-read per chunk), walked the landmarks of a chunk one after another for the direct part of J^T J, and passed seven
// workgroup barriers per chunk: 55 k cycles per chunk of 256 factors, 162 us for the 10 x 50 000 window (1.7 % of the HBM roofline).
// Here a chunk is <= 256 factors of landmarks that share ONE anchor frame (the host cuts chunks at anchor changes), and nothing is kept per (landmark, frame):
//
//   E  evaluate     one thread per factor (reprojection_error_cost.h:40-120 through pv_factors.h): the factor's row X = [Jt | Jr | r | Jd] goes to LDS as 14
//                   (row 0, row 1) pairs, its Schur row W_t = Jd^T Jt to the landmark's dense U row (LDS) and to HBM (k_backsub reads it)
//   D  direct part  J^T J of a factor only touches the target's diagonal block, the (target, anchor) block and the target's gradient: 9 N thread tasks
//                   (target t, 3 x 3 block b), each walking the chunk's factors of ITS target in a fixed order (a per-chunk permutation by target, built by
//                   the host at upload) with the 9 sums in registers for the WHOLE chunk range of the workgroup; small N: several threads per task
//   L  per landmark H_ll, b_l, W_a, Jr^T r, Jr^T Jr are the Gram matrix of the landmark's factor rows [Jd r Jr]: v_mfma_f64_16x16x4_f64, two landmarks per
//                   instruction, two factors per K step (contiguous rows: no per-frame slots, nothing to clear)
//   P  scalars      Jacobi scale, dogleg diagonal, Schur weight; W_a completes the U row; the per-landmark outputs k_backsub needs
//   S  Schur        - sum_l w_l u_l u_l^T as a SYRK on the matrix cores, 16 x 16 tiles of the lower block triangle in accumulator registers for the whole
//                   walk (as before); the anchor's own block and the right-hand side by a few threads beside it; the NEXT chunk's landmark inputs are
//                   fetched and the other U buffer is cleared here, so a chunk passes four barriers
//   flush           once per workgroup (and at an anchor change): accumulators -> the element-major 3 x 3-task partial row the other form writes
//
// Sums are taken in fixed orders (no floating-point atomics): re-solves are bit-identical.
//
// LDS after the common part (doubles):  X [256][28] | U [2][S][US] | LMR [S][50] | DS (direct sums at a flush) | small per-chunk tables
#pragma once

constexpr int kTpXCols = 14;   // (row 0, row 1) pairs of a factor row: 0-5 Jt, 6-11 Jr, 12 r, 13 Jd
constexpr int kTpLmr = 50;     // per-landmark results: HAA[36] GA[6] | 42 w  43 bl  44..49 Wa
constexpr int kTpDirTasks = 9; // per target: TT00 TT01 TT11 | TR00 TR01 TR10 TR11 | g[0:3] g[3:6]

__host__ __device__ inline int tp_u_stride(int P6) { return ((P6 + 15) >> 4) << 4; }
// doubles of LDS the role needs behind the common part
__host__ __device__ inline size_t tp_lds_doubles(int N, int P6, int slots, int n_tasks) {
    const size_t S = (size_t)slots, US = (size_t)tp_u_stride(P6);
    size_t work = (size_t)kLinThreads * 2 * kTpXCols + 2 * S * US + S * kTpLmr;
    const size_t stage = (size_t)n_tasks * 5 + (size_t)kTpDirTasks * N * 9 + 64; // the final flush reuses the work area: stage halves + direct sums
    if (work < stage) work = stage;
    // tables: rho_eval[2][S] | accumulators vg_acc[P6] vdiag_acc[P6] | ints: active[2][S], fptr[2][S + 1], tptr[kMaxFrames + 1], perm[256]
    return work + 2 * S + 2 * (size_t)P6 + (2 * S + 2 * (S + 1) + (kMaxFrames + 1) + kLinThreads + 8) / 2 + 8;
}

template <int TW, int NDT> // TW: accumulator tiles per wave; NDT: direct tasks per thread (2 when 9 N > 256)
__device__ __forceinline__ void role_landmarks_tp(const View &v, double *lds, const Pro *pro, int wg, int n_wg) {
    const int N = v.dm.N, M = v.dm.M, P6 = v.dm.P6, tid = threadIdx.x, n_tasks = v.dm.n_tasks;
    const int lane = tid & 63, wv = tid >> 6, lk = lane >> 4, lr = lane & 15;
    const double *frec = lds + N * 16;
    double *scratch = lds + N * 16 + N * kFrameRec;
    const int S = (v.dm.lm_slots + 3) & ~3, US = tp_u_stride(P6);
    // ---- LDS carve ----
    double *work = lds + common_lds_doubles(N);
    lds_d2 *X2 = reinterpret_cast<lds_d2 *>(work);               // [256][14]
    double *Ubuf = work + (size_t)kLinThreads * 2 * kTpXCols;    // [2][S][US]
    double *LMR = Ubuf + 2 * (size_t)S * US;                     // [S][50]
    size_t work_sz = (size_t)kLinThreads * 2 * kTpXCols + 2 * (size_t)S * US + (size_t)S * kTpLmr;
    {
        const size_t stage_sz = (size_t)n_tasks * 5 + (size_t)kTpDirTasks * N * 9 + 64;
        if (work_sz < stage_sz) work_sz = stage_sz;
    }
    double *rho_eval = work + work_sz;                           // [2][S]
    double *vg_acc = rho_eval + 2 * S, *vdiag_acc = vg_acc + P6; // [P6] each: what anchor flushes have taken out of the registers
    int *active = reinterpret_cast<int *>(vdiag_acc + P6);       // [2][S]
    int *fptr = active + 2 * S;                                  // [2][S + 1] first factor (chunk-relative) of every slot
    int *tptr = fptr + 2 * (S + 1);                              // [N + 1] chunk-relative offsets of the by-target permutation
    int *perm = tptr + kMaxFrames + 1;                           // [256] factor slots sorted by target

    const int mode = pro->mode, cur = pro->cur, lin = pro->lin, oset = pro->out_set;
    const bool marg = mode == MODE_MARG;
    const int victim = v.ctrl->marg_victim;
    const double mu = pro->mu_schur, ca = pro->ca, cb = pro->cb;
    const size_t Ms = (size_t)M, Fs = (size_t)v.dm.F;
    double *o_Hll = v.Hll + oset * Ms, *o_bl = v.bl + oset * Ms, *o_Dl = v.Dl + oset * Ms, *o_ghl = v.ghl + oset * Ms;
    double *o_Wa = v.Wa + oset * Ms * 6, *o_Wt = v.Wt + oset * Fs * 6;
    const double *i_Dl = v.Dl + lin * Ms, *i_ghl = v.ghl + lin * Ms, *i_gnl = v.gnl + lin * Ms;
    const double *rho_cur = v.rho + cur * Ms;
    double *rho_cand = v.rho + (1 - cur) * Ms;
    double *pS = v.part_S + (size_t)wg * n_tasks * 9;

    // ---- Schur tiles of this wave (lower block triangle, dealt round-robin to the four waves) ----
    const int nbt = (P6 + 15) >> 4, ntile = (nbt * (nbt + 1)) >> 1;
    mfma_d4 tacc[TW];
    int tile_bb[TW]; // bi << 8 | bj, -1 past the last tile
#pragma unroll
    for (int u = 0; u < TW; ++u) {
        const int q = wv + 4 * u;
        int bi = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
        while (((bi + 1) * (bi + 2)) >> 1 <= q) ++bi;
        while (((bi * (bi + 1)) >> 1) > q) --bi;
        const int bj = q - ((bi * (bi + 1)) >> 1);
        tile_bb[u] = q < ntile ? (bi << 8 | bj) : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) tacc[u][r] = 0.0;
    }
    // ---- direct tasks of this thread: (target, block) x sub; task index = block * N + target (threads of a wave mostly share the block) ----
    const int n_dir = kTpDirTasks * N;
    const int nsub = NDT > 1 ? 1 : (kLinThreads / n_dir > 0 ? kLinThreads / n_dir : 1);
    int d_t[NDT], d_a[NDT], d_b[NDT], d_bs[NDT], d_blk[NDT]; // target, first A column, first B column, B column step, block
    const int d_sub = tid % nsub;
    double dacc[NDT][9];
#pragma unroll
    for (int q = 0; q < NDT; ++q) {
        const int task = tid / nsub + q * kLinThreads;
        const bool on = task < n_dir;
        const int b = on ? task / N : 0;
        d_t[q] = on ? task - b * N : -1, d_blk[q] = b;
        // TT00 TT01 TT11 TR00 TR01 TR10 TR11 g0 g1
        d_a[q] = (b == 0 || b == 1 || b == 3 || b == 4 || b == 7) ? 0 : 3;
        d_b[q] = b == 0 ? 0 : (b == 1 || b == 2) ? 3 : (b == 3 || b == 5) ? 6 : (b == 4 || b == 6) ? 9 : 12;
        d_bs[q] = b >= 7 ? 0 : 1;
#pragma unroll
        for (int e = 0; e < 9; ++e) dacc[q][e] = 0.0;
    }
    double aa = 0.0;                       // tid in [64, 64 + 42): the anchor's own block HAA (36) and Jr^T r (6) of the landmarks since the last flush
    double vrhs = 0.0;                     // tid < P6
    double s_cost = 0, s_g2 = 0, s_step2 = 0, s_norm2 = 0, s_bad = 0, s_bmax = 0;
    int cur_anchor = -1;
    bool row_dirty = false;
    for (int e = tid; e < 2 * P6; e += kLinThreads) vg_acc[e] = 0.0;

    const int per_wg = (v.dm.n_chunks + n_wg - 1) / n_wg;
    const int ck_begin = wg * per_wg, ck_end = ck_begin + per_wg < v.dm.n_chunks ? ck_begin + per_wg : v.dm.n_chunks;

    // landmark inputs of chunk ck -> tables of parity (ck & 1); candidate inverse depths, |step|^2, |x|^2 (threads tid < ns)
    auto prep = [&](int ck) {
        const int l0 = v.chunk_lm[ck], ns = v.chunk_lm[ck + 1] - l0, par = ck & 1;
        if (tid < ns) {
            const int l = l0 + tid;
            double r = rho_cur[l];
            const int p0 = v.lm_ptr[l], p1 = v.lm_ptr[l + 1];
            const bool used = p1 > p0;
            if (mode == MODE_CANDIDATE && used) {
                const double dl = v.cl[l] * (ca * i_ghl[l] + cb * i_gnl[l]) / i_Dl[l];
                const double rc = r + dl;
                s_step2 += (rc - r) * (rc - r);
                r = rc;
            }
            if (mode == MODE_CANDIDATE) rho_cand[l] = r;
            if (used) s_norm2 += r * r;
            rho_eval[par * S + tid] = r;
            int act = 1;
            if (marg) { // bundle_adjustor.cpp:455-461: only tracks the victim frame observes
                act = v.lm_anchor[l] == victim;
                for (int o = p0; o < p1; ++o) act |= v.obs_frame[o] == victim;
            }
            active[par * S + tid] = act;
            fptr[par * (S + 1) + tid] = p0 - v.lm_ptr[l0];
            if (tid == ns - 1) fptr[par * (S + 1) + ns] = p1 - v.lm_ptr[l0];
        }
    };
    auto clear_u = [&](int par) {
        lds_d2 z;
        z[0] = 0.0, z[1] = 0.0;
        lds_d2 *u2 = reinterpret_cast<lds_d2 *>(Ubuf + (size_t)par * S * US);
        for (int e = tid; e < ((S * US) >> 1); e += kLinThreads) u2[e] = z;
    };
    // The (target, anchor) blocks and the anchor's own block belong to ONE anchor: when it changes they leave the registers.  The workgroup's partial
    // row is zeroed the first time (row_dirty) and every entry has one writer per flush (the subs of a task are summed through LDS first): plain
    // read-modify-writes of the workgroup's own row, no atomics.  [uniform]
    auto anchor_flush = [&]() {
        if (!row_dirty) {
            for (int e = tid; e < n_tasks * 9; e += kLinThreads) pS[e] = 0.0;
            __threadfence_block();
            row_dirty = true;
        }
        double *DS = work; // [n_dir][nsub][9] -- X is free between chunks
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_t[q] >= 0 && d_blk[q] >= 3 && d_blk[q] <= 6) {
                double *dst = DS + ((size_t)(d_blk[q] * N + d_t[q]) * nsub + d_sub) * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) dst[e] = dacc[q][e], dacc[q][e] = 0.0;
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_t[q] >= 0 && d_blk[q] >= 3 && d_blk[q] <= 6 && d_sub == 0 && d_t[q] != cur_anchor) {
                const double *src = DS + (size_t)(d_blk[q] * N + d_t[q]) * nsub * 9;
                const int bi = (d_blk[q] - 3) >> 1, bj = (d_blk[q] - 3) & 1;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    double sum = src[e];
                    for (int u = 1; u < nsub; ++u) sum += src[u * 9 + e];
                    const PartialEntry pe = partial_entry(N, d_t[q], 3 * bi + e / 3, cur_anchor, 3 * bj + e % 3);
                    if (sum != 0.0) pS[pe.el * n_tasks + pe.t] += sum;
                }
            }
        if (tid >= 64 && tid < 64 + 36) {
            const int e = tid - 64;
            const PartialEntry pe = partial_entry(N, cur_anchor, e / 6, cur_anchor, e % 6);
            pS[pe.el * n_tasks + pe.t] += aa;
            if (e / 6 == e % 6) vdiag_acc[6 * cur_anchor + e / 6] += aa;
            aa = 0.0;
        } else if (tid >= 64 + 36 && tid < 64 + 42) {
            vg_acc[6 * cur_anchor + (tid - 64 - 36)] += aa;
            aa = 0.0;
        }
        __threadfence_block();
        __syncthreads();
    };

    if (ck_begin < ck_end) {
        prep(ck_begin);
        clear_u(ck_begin & 1);
    }
    __syncthreads();
    for (int ck = ck_begin; ck < ck_end; ++ck) {
        const int par = ck & 1;
        const int l0 = v.chunk_lm[ck], ns = v.chunk_lm[ck + 1] - l0;
        const int o0 = v.lm_ptr[l0], nf = v.lm_ptr[l0 + ns] - o0;
        const int a = v.lm_anchor[l0]; // the chunk's anchor (the host cuts chunks at anchor changes)
        double *U = Ubuf + (size_t)par * S * US;
        const double *rho_e = rho_eval + par * S;
        const int *act_e = active + par * S, *fp = fptr + par * (S + 1);
        if (a != cur_anchor) { // uniform
            if (cur_anchor >= 0) anchor_flush();
            cur_anchor = a;
        }
        PV_STAMP(0, 2);
        // ---- E: one thread per factor ----
        if (tid <= N) tptr[tid] = v.chunk_tptr[(size_t)ck * (N + 1) + tid];
        if (tid < nf) {
            const int o = o0 + tid, l = v.obs_lm[o], s = l - l0, t = v.obs_frame[o];
            perm[tid] = v.chunk_perm[o];
            double r[2], Jt[12], Jr[12], Jd[2];
            reproj_eval<true>(frec + t * kFrameRec, frec + a * kFrameRec, rho_e[s], v.lm_zref[2 * l], v.lm_zref[2 * l + 1],
                              v.obs_z[2 * (size_t)o], v.obs_z[2 * (size_t)o + 1], r, Jt, Jr, Jd);
            const bool act = act_e[s] != 0;
            const double sq = r[0] * r[0] + r[1] * r[1];
            // duplicate residual blocks (bundle_adjustor.cpp:165-179): m copies of the block, each robustified on its own, summed by
            // Ceres = the robustified block scaled by sqrt(m), its cost by m.  marginalize_frame lists every block once (:455-510).
            const double mult = (v.lm_mult && !marg) ? v.lm_mult[l] : 1.0;
            double bad = isfinite(sq) ? 0.0 : 1.0;
            // Corrector, rho'' < 0: sqrt(rho'); marginalization uses the un-robustified Jacobians (:487-510) of ALL blocks
            double sw = marg ? 1.0 : sqrt(fmax(DBL_MIN, 1.0 / (1.0 + sq)));
            if (mult != 1.0) sw *= sqrt(mult);
            const bool tfix = !marg && v.frame_fixed[t] != 0, afix = !marg && v.frame_fixed[a] != 0;
            if (!act) sw = 0.0; // (marginalization: a track the victim does not see contributes nothing)
            r[0] *= sw, r[1] *= sw, Jd[0] *= sw, Jd[1] *= sw;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                bad += isfinite(Jt[k]) && isfinite(Jr[k]) ? 0.0 : 1.0;
                Jt[k] = tfix ? 0.0 : Jt[k] * sw; // constant blocks have no Jacobian
                Jr[k] = afix ? 0.0 : Jr[k] * sw;
            }
            if (act) {
                s_cost += mult * (0.5 * log(1.0 + sq)); // CauchyLoss(1): rho(s) = log(1 + s)
                s_bad += bad;
            }
            lds_d2 *x = X2 + (size_t)tid * kTpXCols;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                lds_d2 p, q;
                p[0] = Jt[k], p[1] = Jt[6 + k], q[0] = Jr[k], q[1] = Jr[6 + k];
                x[k] = p, x[6 + k] = q;
            }
            {
                lds_d2 p, q;
                p[0] = r[0], p[1] = r[1], q[0] = Jd[0], q[1] = Jd[1];
                x[12] = p, x[13] = q;
            }
            double *Us = U + (size_t)s * US + 6 * t;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double wt = Jd[0] * Jt[k] + Jd[1] * Jt[6 + k];
                Us[k] = wt;
                o_Wt[(size_t)o * 6 + k] = wt;
            }
        }
        __syncthreads();
        PV_STAMP(0, 3);
        // ---- D: direct part, thread = (target, 3 x 3 block, sub) ----
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_t[q] >= 0) {
                const int e0 = tptr[d_t[q]], e1 = tptr[d_t[q] + 1];
                const int ao = d_a[q], bo = d_b[q], bs = d_bs[q];
                for (int i = e0 + d_sub; i < e1; i += nsub) {
                    const lds_d2 *x = X2 + (size_t)perm[i] * kTpXCols;
                    const lds_d2 a0 = x[ao], a1 = x[ao + 1], a2 = x[ao + 2], b0 = x[bo], b1 = x[bo + bs], b2 = x[bo + 2 * bs];
                    dacc[q][0] += a0[0] * b0[0] + a0[1] * b0[1], dacc[q][1] += a0[0] * b1[0] + a0[1] * b1[1], dacc[q][2] += a0[0] * b2[0] + a0[1] * b2[1];
                    dacc[q][3] += a1[0] * b0[0] + a1[1] * b0[1], dacc[q][4] += a1[0] * b1[0] + a1[1] * b1[1], dacc[q][5] += a1[0] * b2[0] + a1[1] * b2[1];
                    dacc[q][6] += a2[0] * b0[0] + a2[1] * b0[1], dacc[q][7] += a2[0] * b1[0] + a2[1] * b1[1], dacc[q][8] += a2[0] * b2[0] + a2[1] * b2[1];
                }
            }
        PV_STAMP(0, 4);
        // ---- L: per-landmark Gram matrix of the factor rows [Jd r Jr0..5] (two residual rows each) on the matrix cores ----
        {
            // lane (c8 = lr & 7, sub = lr >> 3, k = lk): column c8 of landmark 2 pr + sub, residual row k & 1 of the factor 2 step + (k >> 1)
            const int c8 = lr & 7, sub = lr >> 3, rho = lk & 1, fo = lk >> 1;
            const int col = c8 == 0 ? 13 : (c8 == 1 ? 12 : 4 + c8);
            for (int pr = wv; 2 * pr < ns; pr += 4) {
                const int so = 2 * pr + sub;
                const int f0 = so < ns ? fp[so] : 0, cnt = so < ns ? fp[so + 1] - f0 : 0;
                const int sA = 2 * pr, cntA = fp[sA + 1] - fp[sA], cntB = sA + 1 < ns ? fp[sA + 2] - fp[sA + 1] : 0;
                const int cmax = cntA > cntB ? cntA : cntB; // uniform
                const double *xs = reinterpret_cast<const double *>(X2 + (size_t)f0 * kTpXCols + col) + rho;
                mfma_d4 g, g2;
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = 0.0, g2[r] = 0.0;
                for (int kf = 0; kf < cmax; kf += 4) { // two independent accumulation chains
                    const double x0 = kf + fo < cnt ? xs[(size_t)(kf + fo) * 2 * kTpXCols] : 0.0;
                    const double x1 = kf + 2 + fo < cnt ? xs[(size_t)(kf + 2 + fo) * 2 * kTpXCols] : 0.0;
                    g = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, g, 0, 0, 0);
                    g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, g2, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] += g2[r];
                if (so < ns) {
                    double *W = LMR + (size_t)so * kTpLmr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = lk + 4 * r, ci = i & 7;
                        if ((i >> 3) != sub) continue;
                        if (ci == 0 && c8 == 0) W[42] = g[r];                       // Hll (the scalar phase turns it into the Schur weight)
                        else if (ci == 0 && c8 == 1) W[43] = g[r];                  // bl
                        else if (ci == 0) W[44 + (c8 - 2)] = g[r];                  // Wa
                        else if (ci == 1 && c8 >= 2) W[36 + (c8 - 2)] = g[r];       // GA = Jr^T r
                        else if (ci >= 2 && c8 >= 2) W[6 * (ci - 2) + (c8 - 2)] = g[r]; // HAA
                    }
                }
            }
        }
        __syncthreads();
        PV_STAMP(0, 5);
        // ---- P: per-landmark scalars: Jacobi scale, dogleg diagonal, Schur weight; W_a completes the U row ----
        if (tid < ns) {
            const int l = l0 + tid;
            double *W = LMR + (size_t)tid * kTpLmr;
            const double Hll = W[42], b = W[43];
            const bool used = fp[tid + 1] > fp[tid];
            double cl;
            if (marg) {
                cl = 1.0;
            } else if (mode == MODE_INIT) {
                cl = used ? 1.0 / (1.0 + sqrt(Hll)) : 1.0; // jacobi_scaling, computed once (iteration 0)
                v.cl[l] = cl;
            } else {
                cl = v.cl[l];
            }
            const double d2 = cl * cl * Hll;
            const double Dl = sqrt(fmin(fmax(d2, 1e-6), 1e32)); // DoglegStrategy diagonal (min/max_lm_diagonal)
            const double gh = cl * b / Dl;
            const double A = d2 + mu * Dl * Dl;                 // e-block: E^T E + mu D^2
            double w = used ? cl * cl / A : 0.0;                // Schur weight on the UNscaled W rows
            if (marg) { // scalar inverse of the landmark block, skipped when not finite (:537-538)
                const double inv = 1.0 / Hll;
                w = (act_e[tid] && isfinite(inv)) ? inv : 0.0;
            }
            W[42] = w;
            o_Hll[l] = Hll, o_bl[l] = b, o_Dl[l] = Dl, o_ghl[l] = used ? gh : 0.0;
            if (used) {
                s_g2 += gh * gh;
                s_bmax = fmax(s_bmax, fabs(b));
            }
            double *Us = U + (size_t)tid * US + 6 * a;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double wa = W[44 + k];
                Us[k] = wa; // the anchor is never a target of its own landmark
                o_Wa[(size_t)l * 6 + k] = wa;
            }
        }
        __syncthreads();
        PV_STAMP(0, 6);
        // ---- S: Schur complement on the matrix cores; beside it the right-hand side, the anchor's own block, the next chunk's inputs ----
        for (int s0 = 0; s0 < ns; s0 += 4) {
            const int row = s0 + lk; // rows past ns are zero (the buffer was cleared), their weight is read as 0
            const double nw = row < ns ? -LMR[(size_t)row * kTpLmr + 42] : 0.0;
            const double *Rl = U + (size_t)row * US + lr;
            constexpr int kOps = TW < 6 ? TW : 6; // operands of a batch are all requested before its first MFMA
#pragma unroll
            for (int u0 = 0; u0 < TW; u0 += kOps) {
                double a_op[kOps], b_op[kOps];
#pragma unroll
                for (int u = 0; u < kOps; ++u) {
                    const int bb = u0 + u < TW ? tile_bb[u0 + u < TW ? u0 + u : 0] : -1;
                    a_op[u] = Rl[bb >= 0 ? (bb >> 8) << 4 : 0], b_op[u] = Rl[bb >= 0 ? (bb & 255) << 4 : 0];
                }
#pragma unroll
                for (int u = 0; u < kOps; ++u)
                    if (u0 + u < TW && tile_bb[u0 + u < TW ? u0 + u : 0] >= 0) // wave-uniform
                        tacc[u0 + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(nw * a_op[u], b_op[u], tacc[u0 + u], 0, 0, 0);
            }
        }
        if (tid < P6) {
            double sum = 0.0;
            for (int s = 0; s < ns; ++s) {
                const double *W = LMR + (size_t)s * kTpLmr;
                sum += W[42] * W[43] * U[(size_t)s * US + tid];
            }
            vrhs += sum;
        }
        if (tid >= 64 && tid < 64 + 42) {
            const int e = tid - 64;
            double sum = 0.0;
            for (int s = 0; s < ns; ++s) sum += LMR[(size_t)s * kTpLmr + e];
            aa += sum;
        }
        if (ck + 1 < ck_end) prep(ck + 1);
        clear_u(1 - par);
        __syncthreads();
        PV_STAMP(0, 7);
    }

    // ---- flush: accumulators -> the workgroup's partial row (element-major 3 x 3 tasks), pose vectors, scalars ----
    // order per entry: tile entry (set), then the direct blocks (disjoint entries), then the last anchor's own block, then what earlier anchor
    // flushes left in the row
    double *stage = work;                                   // [5][n_tasks]
    double *DS = work + (size_t)n_tasks * 5;                // [n_dir][9] direct sums (subs added in order)
    {
        // subs of a task -> sub 0, through DS laid out [task][sub][9] in the stage area first (free: the chunk walk is over)
        double *tmp = work;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_t[q] >= 0) {
                double *dst = tmp + ((size_t)(d_blk[q] * N + d_t[q]) * nsub + d_sub) * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) dst[e] = dacc[q][e];
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_t[q] >= 0 && d_sub == 0) {
                const double *src = tmp + (size_t)(d_blk[q] * N + d_t[q]) * nsub * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    double sum = src[e];
                    for (int u = 1; u < nsub; ++u) sum += src[u * 9 + e];
                    dacc[q][e] = sum;
                }
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_t[q] >= 0 && d_sub == 0) {
                double *dst = DS + (size_t)(d_blk[q] * N + d_t[q]) * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) dst[e] = dacc[q][e];
            }
        __syncthreads();
    }
    // pose vectors: g_dir = Jt^T r of the targets + Jr^T r of the anchors; rhs_schur; diag of the direct H
    double aa_keep = aa; // (threads 64 .. 64 + 41)
    if (tid >= 64 + 36 && tid < 64 + 42 && cur_anchor >= 0) vg_acc[6 * cur_anchor + (tid - 64 - 36)] += aa;
    if (tid >= 64 && tid < 64 + 36 && cur_anchor >= 0 && (tid - 64) / 6 == (tid - 64) % 6) vdiag_acc[6 * cur_anchor + (tid - 64) / 6] += aa;
    __syncthreads();
    if (tid < P6) {
        const int t = tid / 6, i = tid - 6 * t;
        const double g_t = DS[(size_t)((7 + i / 3) * N + t) * 9 + 3 * (i % 3)];
        const double d_tt = DS[(size_t)((i < 3 ? 0 : 2) * N + t) * 9 + 4 * (i % 3)];
        double *pv = v.part_vec + (size_t)wg * kNumPoseVec * P6;
        pv[tid] = g_t + vg_acc[tid], pv[P6 + tid] = vrhs, pv[2 * P6 + tid] = d_tt + vdiag_acc[tid];
    }
    for (int h = 0; h < 2; ++h) {
        const int e_lo = 5 * h, e_n = h == 0 ? 5 : 4;
        __syncthreads();
        for (int e = tid; e < e_n * n_tasks; e += kLinThreads) stage[e] = 0.0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TW; ++u) {
            if (tile_bb[u] < 0) continue;
            const int I0 = ((tile_bb[u] >> 8) << 4) + lk, J = ((tile_bb[u] & 255) << 4) + lr; // this lane owns rows I0 + 4 r of column J of the tile
            const int fJ = J / 6, jJ = J - 6 * fJ;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int I = I0 + 4 * r;
                if (I >= P6 || J > I) continue;
                const int fI = I / 6, iI = I - 6 * fI;
                stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, fI, iI, fJ, jJ), tacc[u][r], false);
            }
        }
        __syncthreads();
        // direct blocks: one thread per (task, element); the entries of different tasks are disjoint
        for (int e = tid; e < n_dir * 9; e += kLinThreads) {
            const int task = e / 9, el = e - 9 * task, b = task / N, t = task - b * N, i = el / 3, j = el - 3 * i;
            const double val = DS[e];
            if (b == 0) stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, t, i, t, j), val, true);
            else if (b == 1) {
                stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, t, i, t, 3 + j), val, true);
                stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, t, 3 + j, t, i), val, true);
            } else if (b == 2) stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, t, 3 + i, t, 3 + j), val, true);
            else if (b <= 6) {
                if (cur_anchor >= 0 && t != cur_anchor) stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, t, 3 * ((b - 3) >> 1) + i, cur_anchor, 3 * ((b - 3) & 1) + j), val, true);
            }
        }
        __syncthreads();
        if (tid >= 64 && tid < 64 + 36 && cur_anchor >= 0) stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, cur_anchor, (tid - 64) / 6, cur_anchor, (tid - 64) % 6), aa_keep, true);
        if (row_dirty) __threadfence_block(); // earlier anchor flushes have landed before the row is read
        __syncthreads();
        double *dstrow = pS + (size_t)e_lo * n_tasks;
        const int n_el = e_n * n_tasks;
        if (!row_dirty) {
            for (int e = tid; e < n_el; e += kLinThreads) dstrow[e] = stage[e];
        } else {
            for (int e0 = tid; e0 < n_el; e0 += 8 * kLinThreads) { // eight row values in flight per thread
                double old[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) old[q] = e0 + q * kLinThreads < n_el ? dstrow[e0 + q * kLinThreads] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (e0 + q * kLinThreads < n_el) dstrow[e0 + q * kLinThreads] = old[q] + stage[e0 + q * kLinThreads];
            }
        }
    }
    double sc[6] = {s_cost, s_g2, s_step2, s_norm2, s_bad, s_bmax};
    block_sum<6, true>(sc, scratch);
    if (tid == 0) {
        double *ps = v.part_scal + (size_t)wg * kNumLinScal;
        ps[0] = sc[0], ps[1] = sc[1], ps[2] = sc[2], ps[3] = sc[3], ps[4] = sc[5], ps[5] = sc[4], ps[6] = 0, ps[7] = 0;
    }
    PV_STAMP(0, 8);
}
