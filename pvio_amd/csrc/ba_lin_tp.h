// ba_lin_tp.h -- the landmark role of k_linearize for LARGE windows, as a throughput kernel (round 6; included by ba_kernels.hip inside namespace pvba).
//
// What it replaces: the round 2-5 matrix-core form kept every landmark of a chunk as a record of 40 N + 46 doubles in LDS (446 at N = 10, 1246 at
// N = 30: cleared, written and re-read per chunk), walked the landmarks of a chunk one after another for the direct part of J^T J, and passed seven
// workgroup barriers per chunk: 55 k cycles per chunk of 256 factors, 162 us for the 10 x 50 000 window (1.7 % of the HBM roofline).
// Here a chunk is <= 256 factors of landmarks that share ONE anchor frame (the host cuts chunks at anchor changes), and nothing is kept per (landmark, frame):
//
//   E  evaluate     one thread per factor (reprojection_error_cost.h:40-120 through pv_factors.h): the factor's row X = [Jt | Jr | r | Jd] goes to LDS as 14
//                   (row 0, row 1) pairs, its Schur row W_t = Jd^T Jt to the landmark's dense U row (LDS) and to HBM (k_backsub reads it)
//   D  direct part  J^T J of a factor only touches the target's diagonal block, the (target, anchor) block, the anchor's own block and the two gradients.
//                   Thread tasks (target t, 3 x 3 block b) walk the chunk's factors of THEIR target in a fixed order (a per-chunk permutation by target, built
//                   by the host at upload), tasks (anchor block b) walk all of the chunk's factors; the 9 sums of a task stay in registers for the WHOLE chunk
//                   range of the workgroup; the threads left over split every task into subs
//   L  per landmark H_ll, b_l, W_a = Jd^T [Jd r Jr] over the landmark's (contiguous) factor rows: thread = (landmark, column[, part]: chunks of few landmarks deal a
//                   landmark's factors to 2 or 4 lanes, joined by DPP)
//   P  scalars      Jacobi scale, dogleg diagonal, Schur weight; W_a and b_l complete the U row; the per-landmark outputs k_backsub needs
//   S  Schur        - sum_l w_l u_l u_l^T as a SYRK on the matrix cores, 16 x 16 tiles of the lower block triangle in accumulator registers for the whole
//                   walk; b_l rides along as column 6 N of the U row, so the row 6 N of the product is the Schur right-hand side
//   A chunk passes four barriers; the inputs of chunk k + 1 (its 32-byte geometry record, then the factor and landmark arrays: two dependent trips to HBM) are
//   requested while chunk k is in E .. S.  N <= 10: two workgroups share a CU (80 KB of LDS, 256 registers each).
//   flush           once per workgroup (and at an anchor change): accumulators -> the element-major 3 x 3-task partial row the other form writes, through a
//                   scatter table of the tile entries built at upload; staged in LDS when the row fits it, straight to the row in HBM otherwise (N > ~20)
//
// Sums are taken in fixed orders (no floating-point atomics): re-solves are bit-identical.
//
// LDS after the common part (doubles):  X [256][14 pairs] (as two halves of 7, ba_types.h) | U [S][tp_u_ld] | LMR [S][kTpLmrLd] (rows padded against bank conflicts, ba_types.h) | small per-chunk tables; a flush reuses X .. LMR as its stage
// Measured per phase (shader-clock stamps, tests/prof_large_tp.py): DESIGN.md 7.4.
#pragma once

// (the LDS geometry -- kTpXCols, tp_u_stride, tp_lds_doubles ... -- is in ba_types.h: the host sizes chunks and the launch with it)

template <int TW, int NDT> // TW: accumulator tiles per wave; NDT: direct tasks per thread
__device__ __forceinline__ void role_landmarks_tp(const View &v, double *lds, const Pro *pro, int wg, int n_wg) {
    const int N = v.dm.N, M = v.dm.M, P6 = v.dm.P6, tid = threadIdx.x, n_tasks = v.dm.n_tasks;
    const int lane = tid & 63, wv = tid >> 6, lk = lane >> 4, lr = lane & 15;
    const double *frec = lds + N * 16;
    double *scratch = lds + N * 16 + N * kFrameRec;
    const int S = (v.dm.lm_slots + 3) & ~3, US = tp_u_ld(P6); // (US: leading dimension of U, tp_u_stride(P6) columns in use)
    // ---- LDS carve ----
    double *work = lds + common_lds_doubles(N);
    lds_d2 *X2 = reinterpret_cast<lds_d2 *>(work);               // [256][14]
    double *U = work + (size_t)kLinThreads * 2 * kTpXCols;       // [S][US]: every cell of a chunk's rows is written by E / P, nothing is cleared per chunk
    double *LMR = U + (size_t)S * US;                            // [S][8]
    const size_t work_sz = tp_work_doubles(N, P6, S, n_tasks);
    double *rho_e = work + work_sz;                              // [S]
    double *cl_tab = rho_e + S;                                  // [S]
    double *vg_acc = cl_tab + S, *vdiag_acc = vg_acc + P6;       // [P6] each: what anchor flushes have taken out of the registers
    int *act_e = reinterpret_cast<int *>(vdiag_acc + P6);        // [S]
    int *seen_e = act_e + S;                                     // [S] frames the landmark is observed in
    int *fp = seen_e + S;                                        // [S + 1] first factor (chunk-relative) of every slot
    int *tptr = fp + S + 1;                                      // [N + 1] chunk-relative offsets of the by-target permutation
    uint8_t *perm = reinterpret_cast<uint8_t *>(tptr + kMaxFrames + 1); // [256] factor slots sorted by target

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

    // ---- Schur tiles of this wave (lower block triangle of the (6 N + 1)-column system, dealt round-robin to the four waves) ----
    const int nbt = tp_u_stride(P6) >> 4, ntile = (nbt * (nbt + 1)) >> 1;
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
    // ---- direct tasks: unified list [target tasks x nsub | anchor tasks x nsubA], thread tid owns entries tid + 256 q ----
    const int n_dir = kTpDirTasks * N;
    const int nsub = (NDT * kLinThreads - 4 * kTpAnchTasks) / n_dir > 0 ? (NDT * kLinThreads - 4 * kTpAnchTasks) / n_dir : 1;
    const int nsubA = (NDT * kLinThreads - n_dir * nsub) / kTpAnchTasks; // >= 4 for N <= 32 (9 N + 20 <= 512)
    int d_kind[NDT], d_t[NDT], d_a[NDT], d_b[NDT], d_bs[NDT], d_blk[NDT], d_sub[NDT]; // -1 idle / 0 target / 1 anchor; target; first A column, first B column (pair offsets in an X row, tp_xcol), its step; block; sub
    double dacc[NDT][9];
#pragma unroll
    for (int q = 0; q < NDT; ++q) {
        const int u = tid + q * kLinThreads, ua = u - n_dir * nsub;
        int b;
        if (u < n_dir * nsub) {
            const int task = u / nsub;
            b = task / N;
            d_kind[q] = 0, d_t[q] = task - b * N, d_blk[q] = b, d_sub[q] = u - task * nsub;
            // TT00 TT01 TT11 TR00 TR01 TR10 TR11 g0 g1
            d_a[q] = tp_xcol((b == 0 || b == 1 || b == 3 || b == 4 || b == 7) ? 0 : 3);
            d_b[q] = tp_xcol(b == 0 ? 0 : (b == 1 || b == 2) ? 3 : (b == 3 || b == 5) ? 6 : (b == 4 || b == 6) ? 9 : 12);
            d_bs[q] = b >= 7 ? 0 : 1;
        } else if (ua < kTpAnchTasks * nsubA) {
            b = ua / nsubA;
            d_kind[q] = 1, d_t[q] = -1, d_blk[q] = b, d_sub[q] = ua - b * nsubA;
            // RR00 RR01 RR11 gR0 gR1
            d_a[q] = tp_xcol((b == 0 || b == 1 || b == 3) ? 6 : 9);
            d_b[q] = tp_xcol(b == 0 ? 6 : (b == 1 || b == 2) ? 9 : 12);
            d_bs[q] = b >= 3 ? 0 : 1;
        } else {
            d_kind[q] = -1, d_t[q] = -1, d_blk[q] = 0, d_sub[q] = 0, d_a[q] = 0, d_b[q] = 0, d_bs[q] = 0;
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) dacc[q][e] = 0.0;
    }
    double s_cost = 0, s_g2 = 0, s_step2 = 0, s_norm2 = 0, s_bad = 0, s_bmax = 0;
    int cur_anchor = -1;
    bool row_dirty = false;
    for (int e = tid; e < 2 * P6; e += kLinThreads) vg_acc[e] = 0.0;

    const int per_wg = (v.dm.n_chunks + n_wg - 1) / n_wg;
    const int ck_begin = wg * per_wg, ck_end = ck_begin + per_wg < v.dm.n_chunks ? ck_begin + per_wg : v.dm.n_chunks;

    // ---- inputs of a chunk, requested one chunk ahead: factor arrays (thread = factor) and landmark arrays (thread = landmark) ----
    int f_l = 0, f_t = 0, f_perm = 0, f_tptr = 0;
    double f_z0 = 0, f_z1 = 0, f_zr0 = 0, f_zr1 = 0;
    double p_rho = 0, p_cl = 1, p_gh = 0, p_gn = 0, p_D = 1;
    int p_p0 = 0, p_p1 = 0, p_anchor = 0;
    uint32_t p_seen = 0;
    // geometry of a chunk: first landmark, landmarks, first factor, factors, anchor frame -- one 32-byte record per chunk (built at upload): deriving it from
    // chunk_lm -> lm_ptr -> lm_anchor is a chain of dependent scalar loads, repeated by every step below it was 3-4 k cycles of a chunk's ~25 k
    struct Geo {
        int l0, ns, o0, nf, a, ck;
    };
    auto load_geo = [&](int ck) {
        const int32_t *g = v.chunk_geo + 8 * (size_t)ck;
        return Geo{g[0], g[1], g[2], g[3], g[4], ck};
    };
    auto request = [&](const Geo &G) { // independent loads
        const int l0 = G.l0, ns = G.ns, o0 = G.o0, nf = G.nf;
        if (tid <= N) f_tptr = v.chunk_tptr[(size_t)G.ck * (N + 1) + tid];
        if (tid < nf) {
            const size_t o = (size_t)o0 + tid;
            f_l = v.obs_lm[o], f_t = v.obs_frame[o], f_perm = v.chunk_perm[o], f_z0 = v.obs_z[2 * o], f_z1 = v.obs_z[2 * o + 1];
        }
        if (tid < ns) {
            const int l = l0 + tid;
            p_rho = rho_cur[l], p_p0 = v.lm_ptr[l], p_p1 = v.lm_ptr[l + 1], p_seen = v.lm_seen[l];
            if (mode != MODE_INIT && !marg) p_cl = v.cl[l];
            if (mode == MODE_CANDIDATE) p_gh = i_ghl[l], p_gn = i_gnl[l], p_D = i_Dl[l];
            if (marg) p_anchor = v.lm_anchor[l];
        }
    };
    auto request2 = [&](const Geo &G) { // loads that depend on the first ones
        if (tid < G.nf) f_zr0 = v.lm_zref[2 * (size_t)f_l], f_zr1 = v.lm_zref[2 * (size_t)f_l + 1];
    };
    // landmark inputs of chunk ck -> the tables (single-buffered: written in phase S of the chunk before, when nothing reads them any more);
    // candidate inverse depths, |step|^2, |x|^2 (threads tid < ns)
    auto prep = [&](const Geo &G) {
        const int l0 = G.l0, ns = G.ns;
        if (tid < ns) {
            const int l = l0 + tid;
            double r = p_rho;
            const bool used = p_p1 > p_p0;
            if (mode == MODE_CANDIDATE && used) {
                const double dl = p_cl * (ca * p_gh + cb * p_gn) / p_D;
                const double rc = r + dl;
                s_step2 += (rc - r) * (rc - r);
                r = rc;
            }
            if (mode == MODE_CANDIDATE) rho_cand[l] = r;
            if (used) s_norm2 += r * r;
            rho_e[tid] = r;
            cl_tab[tid] = p_cl;
            int act = 1;
            if (marg) { // bundle_adjustor.cpp:455-461: only tracks the victim frame observes
                act = p_anchor == victim;
                for (int o = p_p0; o < p_p1; ++o) act |= v.obs_frame[o] == victim;
            }
            act_e[tid] = act;
            seen_e[tid] = (int)p_seen;
            fp[tid] = p_p0 - G.o0;
            if (tid == ns - 1) fp[ns] = p_p1 - G.o0;
        }
    };
    // sums of a task's subs -> every thread of the task, through LDS (work area: X is free between chunks), in sub order
    auto sum_subs = [&](bool only_anchor_bound) {
        double *DS = work; // [NDT * 256][9]
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_kind[q] >= 0) {
                double *dst = DS + (size_t)(tid + q * kLinThreads) * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) dst[e] = dacc[q][e];
            }
        __syncthreads();
        if (!only_anchor_bound) PV_STAMP(0, 20);
        // the anchor tasks have many subs (15 at N = 10, 48 at N = 30): 45 (task, element) sums dealt to four lanes each, joined by DPP (fixed order)
        double *RES = DS + (size_t)NDT * kLinThreads * 9; // [5][9]
        {
            const int g = tid & 3, te = tid >> 2, b = te / 9, e = te - 9 * b;
            double sum = 0.0;
            if (te < 9 * kTpAnchTasks) {
                const int per = (nsubA + 3) >> 2, u0 = g * per, u1 = u0 + per < nsubA ? u0 + per : nsubA;
                const double *src = DS + (size_t)(n_dir * nsub + b * nsubA) * 9 + e;
                for (int u = u0; u < u1; ++u) sum += src[(size_t)u * 9];
            }
            sum += dpp_f64(sum, 0), sum += dpp_f64(sum, 1); // (whole waves: 180 lanes = 45 quads)
            if (te < 9 * kTpAnchTasks && g == 0) RES[te] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NDT; ++q) {
            const bool bound = d_kind[q] == 1 || (d_kind[q] == 0 && d_blk[q] >= 3 && d_blk[q] <= 6); // the sums that belong to the current anchor
            if (d_kind[q] == 1 && d_sub[q] == 0) {
#pragma unroll
                for (int e = 0; e < 9; ++e) dacc[q][e] = RES[9 * d_blk[q] + e];
            } else if (d_kind[q] == 0 && d_sub[q] == 0 && (bound || !only_anchor_bound)) {
                const double *src = DS + (size_t)(tid + q * kLinThreads) * 9;
                double sum[9]; // (sub by sub, the nine loads of a sub in flight together)
#pragma unroll
                for (int e = 0; e < 9; ++e) sum[e] = src[e];
                for (int u = 1; u < nsub; ++u) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) sum[e] += src[u * 9 + e];
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) dacc[q][e] = sum[e];
            } else if (d_kind[q] >= 0 && (bound || !only_anchor_bound)) {
#pragma unroll
                for (int e = 0; e < 9; ++e) dacc[q][e] = 0.0;
            }
        }
        __syncthreads();
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
        sum_subs(true);
        const int A = cur_anchor;
#pragma unroll
        for (int q = 0; q < NDT; ++q) {
            if (d_kind[q] == 0 && d_blk[q] >= 3 && d_blk[q] <= 6 && d_sub[q] == 0 && d_t[q] != A) {
                const int bi = (d_blk[q] - 3) >> 1, bj = (d_blk[q] - 3) & 1;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    const PartialEntry pe = partial_entry(N, d_t[q], 3 * bi + e / 3, A, 3 * bj + e % 3);
                    if (dacc[q][e] != 0.0) pS[pe.el * n_tasks + pe.t] += dacc[q][e];
                }
            }
            if (d_kind[q] == 1 && d_sub[q] == 0) {
                const int b = d_blk[q];
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    const int i = e / 3, j = e % 3;
                    const double val = dacc[q][e];
                    if (b == 0 || b == 2) {
                        const int o3 = b == 0 ? 0 : 3;
                        const PartialEntry pe = partial_entry(N, A, o3 + i, A, o3 + j);
                        pS[pe.el * n_tasks + pe.t] += val;
                        if (i == j) vdiag_acc[6 * A + o3 + i] += val;
                    } else if (b == 1) {
                        const PartialEntry pe = partial_entry(N, A, i, A, 3 + j), pt = partial_entry(N, A, 3 + j, A, i);
                        pS[pe.el * n_tasks + pe.t] += val, pS[pt.el * n_tasks + pt.t] += val;
                    } else if (j == 0) vg_acc[6 * A + 3 * (b - 3) + i] += val;
                }
            }
            if (d_kind[q] == 1 || (d_kind[q] == 0 && d_blk[q] >= 3 && d_blk[q] <= 6)) {
#pragma unroll
                for (int e = 0; e < 9; ++e) dacc[q][e] = 0.0;
            }
        }
        __threadfence_block();
        __syncthreads();
    };

    Geo cg{0, 0, 0, 0, 0, 0}, ng{0, 0, 0, 0, 0, 0}; // this chunk's / the next chunk's geometry
    if (ck_begin < ck_end) {
        cg = load_geo(ck_begin);
        request(cg);
        request2(cg);
        prep(cg);
    }
    {   // the U rows once: the columns behind 6 N + 1 stay zero for the whole walk
        lds_d2 z;
        z[0] = 0.0, z[1] = 0.0;
        lds_d2 *u2 = reinterpret_cast<lds_d2 *>(U);
        for (int e = tid; e < ((S * US) >> 1); e += kLinThreads) u2[e] = z;
    }
    __syncthreads();
    for (int ck = ck_begin; ck < ck_end; ++ck) {
        const int l0 = cg.l0, ns = cg.ns, o0 = cg.o0, nf = cg.nf;
        const int a = cg.a; // the chunk's anchor (the host cuts chunks at anchor changes)
        if (ck + 1 < ck_end) ng = load_geo(ck + 1); // (back long before phase E ends)
        if (a != cur_anchor) { // uniform
            if (cur_anchor >= 0) anchor_flush();
            cur_anchor = a;
        }
        PV_STAMP(0, 2);
        // ---- E: one thread per factor ----
        if (tid <= N) tptr[tid] = f_tptr; // (requested with the chunk's other inputs, a chunk ahead)
        if (tid < nf) {
            const int o = o0 + tid, l = f_l, s = l - l0, t = f_t;
            perm[tid] = (uint8_t)f_perm;
            double r[2], Jt[12], Jr[12], Jd[2];
            reproj_eval<true>(frec + t * kFrameRec, frec + a * kFrameRec, rho_e[s], f_zr0, f_zr1, f_z0, f_z1, r, Jt, Jr, Jd);
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
            lds_d2 *x = X2 + (size_t)tid * kTpXLd;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                lds_d2 p, q;
                p[0] = Jt[k], p[1] = Jt[6 + k], q[0] = Jr[k], q[1] = Jr[6 + k];
                x[tp_xcol(k)] = p, x[tp_xcol(6 + k)] = q;
            }
            {
                lds_d2 p, q;
                p[0] = r[0], p[1] = r[1], q[0] = Jd[0], q[1] = Jd[1];
                x[tp_xcol(12)] = p, x[tp_xcol(13)] = q;
            }
            double *Us = U + (size_t)s * US + 6 * t;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double wt = Jd[0] * Jt[k] + Jd[1] * Jt[6 + k];
                Us[k] = wt;
                o_Wt[(size_t)o * 6 + k] = wt;
            }
        }
        if (ck + 1 < ck_end) request(ng); // (in flight through D .. P)
        __syncthreads();
        PV_STAMP(0, 3);
        // ---- D: direct part, thread = (target or anchor task, sub); two factors per pass so that their loads are in flight together ----
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_kind[q] >= 0) {
                const bool tgt = d_kind[q] == 0;
                const int e0 = tgt ? tptr[d_t[q]] : 0, e1 = tgt ? tptr[d_t[q] + 1] : nf, step = tgt ? nsub : nsubA;
                const int ao = d_a[q], bo = d_b[q], bs = d_bs[q];
                int i = e0 + d_sub[q];
                for (; i + step < e1; i += 2 * step) {
                    const int fA = tgt ? perm[i] : i, fB = tgt ? perm[i + step] : i + step;
                    const lds_d2 *x = X2 + (size_t)fA * kTpXLd, *y = X2 + (size_t)fB * kTpXLd;
                    const lds_d2 a0 = x[ao], a1 = x[ao + 1], a2 = x[ao + 2], b0 = x[bo], b1 = x[bo + bs], b2 = x[bo + 2 * bs];
                    const lds_d2 c0 = y[ao], c1 = y[ao + 1], c2 = y[ao + 2], d0 = y[bo], d1 = y[bo + bs], d2 = y[bo + 2 * bs];
                    dacc[q][0] += a0[0] * b0[0] + a0[1] * b0[1], dacc[q][1] += a0[0] * b1[0] + a0[1] * b1[1], dacc[q][2] += a0[0] * b2[0] + a0[1] * b2[1];
                    dacc[q][3] += a1[0] * b0[0] + a1[1] * b0[1], dacc[q][4] += a1[0] * b1[0] + a1[1] * b1[1], dacc[q][5] += a1[0] * b2[0] + a1[1] * b2[1];
                    dacc[q][6] += a2[0] * b0[0] + a2[1] * b0[1], dacc[q][7] += a2[0] * b1[0] + a2[1] * b1[1], dacc[q][8] += a2[0] * b2[0] + a2[1] * b2[1];
                    dacc[q][0] += c0[0] * d0[0] + c0[1] * d0[1], dacc[q][1] += c0[0] * d1[0] + c0[1] * d1[1], dacc[q][2] += c0[0] * d2[0] + c0[1] * d2[1];
                    dacc[q][3] += c1[0] * d0[0] + c1[1] * d0[1], dacc[q][4] += c1[0] * d1[0] + c1[1] * d1[1], dacc[q][5] += c1[0] * d2[0] + c1[1] * d2[1];
                    dacc[q][6] += c2[0] * d0[0] + c2[1] * d0[1], dacc[q][7] += c2[0] * d1[0] + c2[1] * d1[1], dacc[q][8] += c2[0] * d2[0] + c2[1] * d2[1];
                }
                if (i < e1) {
                    const lds_d2 *x = X2 + (size_t)(tgt ? perm[i] : i) * kTpXLd;
                    const lds_d2 a0 = x[ao], a1 = x[ao + 1], a2 = x[ao + 2], b0 = x[bo], b1 = x[bo + bs], b2 = x[bo + 2 * bs];
                    dacc[q][0] += a0[0] * b0[0] + a0[1] * b0[1], dacc[q][1] += a0[0] * b1[0] + a0[1] * b1[1], dacc[q][2] += a0[0] * b2[0] + a0[1] * b2[1];
                    dacc[q][3] += a1[0] * b0[0] + a1[1] * b0[1], dacc[q][4] += a1[0] * b1[0] + a1[1] * b1[1], dacc[q][5] += a1[0] * b2[0] + a1[1] * b2[1];
                    dacc[q][6] += a2[0] * b0[0] + a2[1] * b0[1], dacc[q][7] += a2[0] * b1[0] + a2[1] * b1[1], dacc[q][8] += a2[0] * b2[0] + a2[1] * b2[1];
                }
            }
        PV_STAMP(0, 4);
        // ---- L: H_ll, b_l, W_a = Jd^T [Jd r Jr] over the landmark's factor rows; thread = (slot, column, part): a chunk of few landmarks (many frames)
        // deals a landmark's factors to 2 or 4 adjacent lanes, whose partial sums meet by DPP (fixed order) ----
        {
            const int split = ns <= 8 ? 4 : (ns <= 16 ? 2 : 1); // uniform
            const int q = tid & (split - 1), sc = tid / split, c = sc & 7, col = tp_xcol(c == 0 ? 13 : (c == 1 ? 12 : 4 + c));
            for (int sb = 0; sb < ns; sb += kLinThreads / (8 * split)) { // (uniform trip count: the DPP exchange below is executed by whole waves)
                const int s = sb + (sc >> 3);
                double s0 = 0.0, s1 = 0.0;
                if (s < ns) {
                    const int f0 = fp[s], f1 = fp[s + 1];
                    int f = f0 + q;
                    for (; f + split < f1; f += 2 * split) {
                        const lds_d2 *x = X2 + (size_t)f * kTpXLd, *y = x + (size_t)split * kTpXLd;
                        const lds_d2 jd0 = x[tp_xcol(13)], y0 = x[col], jd1 = y[tp_xcol(13)], y1 = y[col];
                        s0 += jd0[0] * y0[0] + jd0[1] * y0[1], s1 += jd1[0] * y1[0] + jd1[1] * y1[1];
                    }
                    if (f < f1) {
                        const lds_d2 *x = X2 + (size_t)f * kTpXLd;
                        const lds_d2 jd0 = x[tp_xcol(13)], y0 = x[col];
                        s0 += jd0[0] * y0[0] + jd0[1] * y0[1];
                    }
                }
                double sum = s0 + s1;
                if (split >= 2) sum += dpp_f64(sum, 0); // lane ^ 1
                if (split >= 4) sum += dpp_f64(sum, 1); // lane ^ 2
                if (s < ns && q == 0) LMR[(size_t)s * kTpLmrLd + c] = sum;
            }
        }
        __syncthreads();
        PV_STAMP(0, 5);
        // ---- P: per-landmark scalars: Jacobi scale, dogleg diagonal, Schur weight; W_a and b_l complete the U row ----
        if (tid < ns) {
            const int l = l0 + tid;
            double *W = LMR + (size_t)tid * kTpLmrLd;
            const double Hll = W[0], b = W[1];
            const bool used = fp[tid + 1] > fp[tid];
            double cl;
            if (marg) {
                cl = 1.0;
            } else if (mode == MODE_INIT) {
                cl = used ? 1.0 / (1.0 + sqrt(Hll)) : 1.0; // jacobi_scaling, computed once (iteration 0)
                v.cl[l] = cl;
            } else {
                cl = cl_tab[tid];
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
            W[0] = w;
            o_Hll[l] = Hll, o_bl[l] = b, o_Dl[l] = Dl, o_ghl[l] = used ? gh : 0.0;
            if (used) {
                s_g2 += gh * gh;
                s_bmax = fmax(s_bmax, fabs(b));
            }
            double *Us = U + (size_t)tid * US;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double wa = W[2 + k];
                Us[6 * a + k] = wa; // the anchor is never a target of its own landmark
                o_Wa[(size_t)l * 6 + k] = wa;
            }
            Us[P6] = b; // row 6 N of the SYRK below: - sum_l w_l b_l u_l = - the Schur right-hand side
            // the blocks of the frames that do not see the landmark: whatever the last chunk left there
            const unsigned unseen = ~((unsigned)seen_e[tid] | (1u << a)) & (N >= 32 ? 0xffffffffu : (1u << N) - 1u);
            for (int f = 0; f < N && unseen != 0u; ++f) // (nothing to do for a landmark every frame sees; f < N also bounds the shift: N may be 32)
                if ((unseen >> f) & 1u) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) Us[6 * f + k] = 0.0;
                }
        } else if (tid < ((ns + 3) & ~3)) { // rows that pad the chunk to the K of an MFMA: zero (their weight is read as 0, but 0 x NaN is not 0)
            double *Us = U + (size_t)tid * US;
            for (int k = 0; k <= P6; ++k) Us[k] = 0.0;
        }
        if (ck + 1 < ck_end) request2(ng);
        __syncthreads();
        PV_STAMP(0, 6);
        // ---- S: Schur complement on the matrix cores; behind it the next chunk's landmark tables (nothing reads the current ones any more) ----
        for (int s0 = 0; s0 < ns; s0 += 4) {
            const int row = s0 + lk; // rows past ns are zero (phase P), their weight is read as 0
            const double nw = row < ns ? -LMR[(size_t)row * kTpLmrLd] : 0.0;
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
        if (ck + 1 < ck_end) prep(ng);
        cg = ng;
        __syncthreads();
        PV_STAMP(0, 7);
    }

    // ---- flush: accumulators -> the workgroup's partial row (element-major 3 x 3 tasks), pose vectors, scalars ----
    // order per entry: tile entry (set), then the direct blocks (disjoint entries), then the last anchor's own block, then what earlier anchor
    // flushes left in the row
    // (the scatter table of this wave's tiles is requested first: it comes from HBM)
    int tdst[TW][4];
#pragma unroll
    for (int u = 0; u < TW; ++u) {
        const int32_t *dp = v.tp_tile_dst + ((size_t)(wv + 4 * u < ntile ? wv + 4 * u : 0) * 64 + lane) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) tdst[u][r] = tile_bb[u] >= 0 ? dp[r] : -1;
    }
    PV_STAMP(0, 18);
    sum_subs(false); // every task's sums in its sub 0
    PV_STAMP(0, 19);
    double *DS = work;                                      // [NDT * 256][9] sums of the tasks (sub 0 entries are read)
#pragma unroll
    for (int q = 0; q < NDT; ++q)
        if (d_kind[q] >= 0 && d_sub[q] == 0) {
            double *dst = DS + (size_t)(tid + q * kLinThreads) * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) dst[e] = dacc[q][e];
        }
    __syncthreads();
    PV_STAMP(0, 14);
    // where the sums of target task (t, b) / anchor task b sit in DS
    auto ds_target = [&](int t, int b) { return DS + (size_t)((b * N + t) * nsub) * 9; };
    auto ds_anchor = [&](int b) { return DS + (size_t)(n_dir * nsub + b * nsubA) * 9; };
    // the last anchor's gradient and diagonal join what the anchor flushes collected
    if (cur_anchor >= 0 && tid < 6) {
        vg_acc[6 * cur_anchor + tid] += ds_anchor(3 + tid / 3)[3 * (tid % 3)];
        vdiag_acc[6 * cur_anchor + tid] += ds_anchor(tid < 3 ? 0 : 2)[4 * (tid % 3)];
    }
    __syncthreads();
    double *pv = v.part_vec + (size_t)wg * kNumPoseVec * P6;
    if (tid < P6) {
        const int t = tid / 6, i = tid - 6 * t;
        const double g_t = ds_target(t, 7 + i / 3)[3 * (i % 3)];
        const double d_tt = ds_target(t, i < 3 ? 0 : 2)[4 * (i % 3)];
        pv[tid] = g_t + vg_acc[tid], pv[2 * P6 + tid] = d_tt + vdiag_acc[tid];
    }
    // the Schur right-hand side: row 6 N of the tiles of the last block row
#pragma unroll
    for (int u = 0; u < TW; ++u) {
        if (tile_bb[u] < 0 || (tile_bb[u] >> 8) != nbt - 1) continue;
        const int I0 = ((tile_bb[u] >> 8) << 4) + lk, J = ((tile_bb[u] & 255) << 4) + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (I0 + 4 * r == P6 && J < P6) pv[P6 + J] = -tacc[u][r];
    }
    double *stage = work + (size_t)NDT * kLinThreads * 9;   // [els][n_tasks]: as many elements of the partial row per pass as fit
    const int cap = (int)((unsigned)(work_sz - (size_t)NDT * kLinThreads * 9) / (unsigned)n_tasks);
    int els = cap >= 9 ? 9 : cap; // >= 2 (tp_work_doubles)
    if (els < 9) {
        // Many frames (the row does not fit the LDS at once: 134 KB at N = 30): straight to the workgroup's row in HBM instead of three staged passes
        // (94 k cycles at 30 x 50 000).  Every entry has ONE writer per step -- tile entries are distinct, the tasks' blocks are disjoint, the anchor's own
        // block comes last -- and the steps are separated by barriers: plain read-modify-writes of the workgroup's own row, in the same order per entry as
        // the staged form (tile, direct block, anchor block; what earlier anchor flushes left is already in the row).
        if (!row_dirty) {
            for (int e = tid; e < n_tasks * 9; e += kLinThreads) pS[e] = 0.0;
            __threadfence_block();
        }
        __syncthreads();
        PV_STAMP(0, 21);
#pragma unroll
        for (int u = 0; u < TW; ++u) {
            if (tile_bb[u] < 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (tdst[u][r] >= 0) {
                    double *dst = pS + (size_t)(tdst[u][r] >> 24) * n_tasks + (tdst[u][r] & 0xffffff);
                    *dst = row_dirty ? *dst + tacc[u][r] : tacc[u][r];
                }
        }
        __threadfence_block();
        __syncthreads();
        PV_STAMP(0, 22);
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_kind[q] == 0 && d_sub[q] == 0 && d_blk[q] < 7) {
                const int b = d_blk[q], t = d_t[q], A = cur_anchor;
                int tA, tB = -1;
                bool trA = false;
                if (b <= 2) {
                    tA = task_index(N, t, t, b == 2, b >= 1);
                    if (b == 1) tB = task_index(N, t, t, 1, 0);
                } else {
                    const int bi = (b - 3) >> 1, bj = (b - 3) & 1;
                    trA = t > A;
                    tA = (A < 0 || t == A) ? -1 : (trA ? task_index(N, A, t, bj, bi) : task_index(N, t, A, bi, bj));
                }
                // (all loads of the task first: a chain of read-modify-writes the compiler must keep in order is eighteen trips to L2)
                double oa[9], ob[9];
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    const int eT = 3 * (e % 3) + e / 3, elA = trA ? eT : e;
                    oa[e] = tA >= 0 ? pS[(size_t)elA * n_tasks + tA] : 0.0, ob[e] = tB >= 0 ? pS[(size_t)eT * n_tasks + tB] : 0.0;
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    const int eT = 3 * (e % 3) + e / 3, elA = trA ? eT : e;
                    if (tA >= 0) pS[(size_t)elA * n_tasks + tA] = oa[e] + dacc[q][e];
                    if (tB >= 0) pS[(size_t)eT * n_tasks + tB] = ob[e] + dacc[q][e];
                }
            }
        __threadfence_block();
        __syncthreads();
        PV_STAMP(0, 23);
        if (tid < 27 && cur_anchor >= 0) { // the last anchor's own block
            const int b = tid / 9, el = tid - 9 * b, i = el / 3, j = el - 3 * i, A = cur_anchor;
            const double val = ds_anchor(b)[el];
            if (b != 1) {
                const PartialEntry pe = partial_entry(N, A, (b == 0 ? 0 : 3) + i, A, (b == 0 ? 0 : 3) + j);
                pS[(size_t)pe.el * n_tasks + pe.t] += val;
            } else {
                const PartialEntry pe = partial_entry(N, A, i, A, 3 + j), pt = partial_entry(N, A, 3 + j, A, i);
                pS[(size_t)pe.el * n_tasks + pe.t] += val, pS[(size_t)pt.el * n_tasks + pt.t] += val;
            }
        }
        els = 9; // (skips the staged passes below)
    }
    for (int e_lo = els < 9 ? 0 : (cap >= 9 ? 0 : 9); e_lo < 9; e_lo += els) {
        const int e_n = e_lo + els <= 9 ? els : 9 - e_lo;
        __syncthreads();
        for (int e = tid; e < e_n * n_tasks; e += kLinThreads) stage[e] = 0.0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TW; ++u) {
            if (tile_bb[u] < 0) continue;
            // scatter table of the tile's entries (built at upload): el << 24 | task, -1 for entries nothing reads
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int el = tdst[u][r] >> 24, t = tdst[u][r] & 0xffffff;
                if (tdst[u][r] >= 0 && el >= e_lo && el < e_lo + e_n) stage[(el - e_lo) * n_tasks + t] = tacc[u][r];
            }
        }
        __syncthreads();
        if (e_lo == 0) PV_STAMP(0, 15);
        // direct blocks, from the registers of every target task's sub 0 (the entries of different tasks are disjoint): a 3 x 3 block is ONE task of the
        // partial row (its nine elements lie n_tasks apart), stored as it is or transposed
#pragma unroll
        for (int q = 0; q < NDT; ++q)
            if (d_kind[q] == 0 && d_sub[q] == 0 && d_blk[q] < 7) {
                const int b = d_blk[q], t = d_t[q], A = cur_anchor;
                int tA, tB = -1;
                bool trA = false;
                if (b <= 2) { // the target's own block: (0, 0) / (0, 1) and its mirror (1, 0) / (1, 1)
                    tA = task_index(N, t, t, b == 2, b >= 1);
                    if (b == 1) tB = task_index(N, t, t, 1, 0);
                } else { // (target, anchor): filed under the smaller frame
                    const int bi = (b - 3) >> 1, bj = (b - 3) & 1;
                    trA = t > A;
                    tA = (A < 0 || t == A) ? -1 : (trA ? task_index(N, A, t, bj, bi) : task_index(N, t, A, bi, bj));
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    const int eT = 3 * (e % 3) + e / 3, elA = trA ? eT : e;
                    if (tA >= 0 && elA >= e_lo && elA < e_lo + e_n) stage[(elA - e_lo) * n_tasks + tA] += dacc[q][e];
                    if (tB >= 0 && eT >= e_lo && eT < e_lo + e_n) stage[(eT - e_lo) * n_tasks + tB] += dacc[q][e];
                }
            }
        __syncthreads();
        if (e_lo == 0) PV_STAMP(0, 16);
        if (tid < 27 && cur_anchor >= 0) { // the last anchor's own block
            const int b = tid / 9, el = tid - 9 * b, i = el / 3, j = el - 3 * i, A = cur_anchor;
            const double val = ds_anchor(b)[el];
            if (b == 0) stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, A, i, A, j), val, true);
            else if (b == 2) stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, A, 3 + i, A, 3 + j), val, true);
            else {
                stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, A, i, A, 3 + j), val, true);
                stage_put(stage, n_tasks, e_lo, e_n, partial_entry(N, A, 3 + j, A, i), val, true);
            }
        }
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
        if (e_lo == 0) PV_STAMP(0, 17);
    }
    double sc[6] = {s_cost, s_g2, s_step2, s_norm2, s_bad, s_bmax};
    block_sum<6, true>(sc, scratch);
    if (tid == 0) {
        double *ps = v.part_scal + (size_t)wg * kNumLinScal;
        ps[0] = sc[0], ps[1] = sc[1], ps[2] = sc[2], ps[3] = sc[3], ps[4] = sc[5], ps[5] = sc[4], ps[6] = 0, ps[7] = 0;
    }
    PV_STAMP(0, 8);
}
