// ba_kernels.hip -- gfx950 kernels of the sliding-window bundle adjustment.
//
// Per trust-region iteration four launches, no host round trip (the whole state machine is in pvba::Ctrl):
//
//   k_linearize  (many WGs)  dogleg step + candidate point (every WG redundantly, tiny) ->
//                            per-factor residual/Jacobian (one thread per reprojection factor, frame records in LDS) ->
//                            per-landmark H_ll, b_l, W_l (LDS segmented sums) ->
//                            LDS-staged J^T J / Schur tile accumulation: every thread owns 3x3 tiles of the
//                            6N x 6N reduced pose system in REGISTERS across all its landmark chunks
//                            (output-stationary, no atomics, deterministic) -> one partial per WG.
//                            Extra WG roles in the same launch: plane-distance factors, IMU pre-integration
//                            factors, marginalization prior.
//   k_reduce     (many WGs)  fixed-order sum of the WG partials -> [S | vectors | scalars] (the all-reduce payload
//                            when landmark blocks are sharded across GPUs).
//   k_dense      (1 WG)      accept/reject + radius/mu update (Ceres 1.14 semantics), assemble the Jacobi-scaled
//                            reduced system, Cholesky (+ forward substitution as an augmented row), wave-level back
//                            substitution, pose parts of every dogleg scalar.
//   k_backsub    (<=64 WGs)  landmark back-substitution + landmark parts of the dogleg scalars.
//
// The linearization of the candidate is SPECULATIVE: the cost pass Ceres needs anyway and the Jacobian pass it
// would run after accepting share one evaluation; a rejected step just discards the speculative set.
//
// Reference being replaced: BundleAdjustorSolver::solve -> ceres::Solve (bundle_adjustor.cpp:63-299,
// solver_options.h:26-33); factor math in pv_factors.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>

#include <float.h>

#include "ba_kernels.h"
#include "pv_factors.h"

namespace pvba {

using namespace pv;

// phase timestamps for the profiling entry point (View::dbg != nullptr): block 0 / thread 0 only, ticks since the start of the
// SAME launch (the launch that ends a solve overwrites the first sites; absolute stamps of different launches do not subtract).
// A stamp is an s_memtime behind an s_waitcnt and a store: with all sites active the dense kernel takes 63 us instead of 50 and
// the phases do not keep their proportions -- View::dbg_sel >= 0 leaves ONE site active per run (tests/prof_phases.py sweeps it).
__device__ long long g_stamp_t0[4], g_stamp_w0[4]; // shader clock / 100 MHz wall clock at the start of the launch
#define PV_STAMP_BEGIN(kern)                                                                   \
    do {                                                                                       \
        if (v.dbg && blockIdx.x == 0 && threadIdx.x == 0) g_stamp_t0[kern] = clock64(), g_stamp_w0[kern] = wall_clock64(); \
    } while (0)
#define PV_STAMP(kern, idx)                                                                                                   \
    do {                                                                                                                      \
        if (v.dbg && (v.dbg_sel < 0 || v.dbg_sel == (idx)) && blockIdx.x == 0 && threadIdx.x == 0) v.dbg[(kern)*32 + (idx)] = clock64() - g_stamp_t0[kern]; \
    } while (0)

// k_dense: the start time is a local of the kernel (pv_t0), the stamp itself requests nothing
#define PV_STAMP2(idx)                                                                                                          \
    do {                                                                                                                        \
        if (v.dbg && (v.dbg_sel < 0 || v.dbg_sel == (idx)) && threadIdx.x == 0) v.dbg[2 * 32 + (idx)] = clock64() - pv_t0;      \
    } while (0)
#define PV_STAMP2T(idx, thread) /* the same from another thread (wave 0 owns no tiles in the look-ahead form) */             \
    do {                                                                                                                        \
        if (v.dbg && (v.dbg_sel < 0 || v.dbg_sel == (idx)) && threadIdx.x == (thread)) v.dbg[2 * 32 + (idx)] = clock64() - pv_t0; \
    } while (0)
// (an empty asm that "uses" a vector register pins the value to this point of the program; the x86 build of tests/hipemu takes the same
// line: "v" names a register class there too)
#define PV_STAMPV2(idx, var)                      \
    do {                                          \
        asm volatile("" : "+v"(var));            \
        PV_STAMP2(idx);                           \
    } while (0)
#define PV_STAMPV(kern, idx, var)                 \
    do {                                          \
        asm volatile("" : "+v"(var));            \
        PV_STAMP(kern, idx);                      \
    } while (0)

// ------------------------------------------------------------------------------------------------------
// small reductions
// ------------------------------------------------------------------------------------------------------
// Wave reductions without the LDS crossbar: `__shfl_xor` on a double is two ds_bpermute per step (~100 cycles of
// latency each, twelve per reduction); DPP moves inside the rows of 16 lanes plus four lane reads cost a tenth of that.
// dpp_f64(x, ctrl): the value of the lane the DPP pattern pairs this lane with.
__device__ __forceinline__ double dpp_f64(double x, int pattern /* 0: ^1, 1: ^2, 2: mirror in 8, 3: mirror in 16 */) {
    union { double d; int i[2]; } u, t;
    u.d = x;
    switch (pattern) {
    case 0: t.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0xB1, 0xF, 0xF, true), t.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: t.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x4E, 0xF, 0xF, true), t.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: t.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x141, 0xF, 0xF, true), t.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x141, 0xF, 0xF, true); break; // row_half_mirror
    default: t.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x140, 0xF, 0xF, true), t.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return t.d;
}
__device__ __forceinline__ double readlane_f64(double x, int src);
// sum over the wave, result in every lane; fixed pairing: inside rows of 16 (^1, ^2, the other quad pair, the other
// half row), then rows 0..3 in order
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64(v, 0);
    v += dpp_f64(v, 1);
    v += dpp_f64(v, 2);
    v += dpp_f64(v, 3);
    return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}
constexpr bool kRsqrtCubic = true;
// 1/sqrt(x) for the pivot chain of the block factorization: v_rsq_f64 (~2^-23 relative) + two Newton steps
// (-> ~1 ulp).  ocml's correctly rounded sqrt + divide is ~40 dependent FP64 instructions per pivot.
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    if (kRsqrtCubic) {
        // one third-order step instead of two Newton steps: e = 1 - x y^2 (|e| <~ 2^-22), 1/sqrt(1 - e) = 1 + e/2 + 3 e^2/8 + O(e^3): five instructions
        // instead of seven, and the pivot loop of k_dense's factor wave is bound by the instructions it issues
        const double e = fma(-x, y * y, 1.0);
        return fma(y, e * fma(e, 0.375, 0.5), y);
    }
    y = y * fma(-0.5 * x, y * y, 1.5);
    y = y * fma(-0.5 * x, y * y, 1.5);
    return y;
}
// broadcast lane `src` (wave-uniform) of a double: two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double x, int src) {
    union { double d; int i[2]; } u;
    u.d = x;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
    return u.d;
}
// sum over aligned groups of 8 lanes, result in every lane of the group: quad_perm [1,0,3,2], quad_perm [2,3,0,1], then
// row_half_mirror (lane i <-> 7 - i inside the group; the quads are uniform by then).  DPP moves, no LDS crossbar.
__device__ __forceinline__ double group8_sum(double x) {
    x += dpp_f64(x, 0);
    x += dpp_f64(x, 1);
    x += dpp_f64(x, 2);
    return x;
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64(v, 0));
    v = fmax(v, dpp_f64(v, 1));
    v = fmax(v, dpp_f64(v, 2));
    v = fmax(v, dpp_f64(v, 3));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
// sums NV values over the block; result valid in every thread.  scratch: NV * 16 doubles of LDS.
template <int NV, bool MAX_LAST = false>
__device__ __forceinline__ void block_sum(double *vals, double *scratch) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double r = (MAX_LAST && k == NV - 1) ? wave_max(vals[k]) : wave_sum(vals[k]);
        if (lane == 0) scratch[k * 16 + wv] = r;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double r = scratch[k * 16];
        for (int w = 1; w < nw; ++w) r = (MAX_LAST && k == NV - 1) ? fmax(r, scratch[k * 16 + w]) : r + scratch[k * 16 + w];
        vals[k] = r;
    }
    __syncthreads();
}

constexpr int kPanel = 8; // Cholesky panel width (= K of two f64 MFMAs)
// The factor wave of the look-ahead form is bound by the NUMBER of instructions it issues per panel (one wave on its SIMD, ~4.6 cycles per FP64
// instruction, tools/ubench/wave_costs.hip; neither the reciprocal chain nor LDS bank conflicts: profiles/r4_ab_split_rows_rcp_chain.txt,
// r4_ab_panel_loop_diet.txt), and it is the critical path of the panel loop.  Round 4 therefore takes instructions OUT of its loop:
// kDenseMaskUpper: the panel's own eight rows carry entries ABOVE the diagonal through the forward substitution (a row is eight numbers whichever row
// it is).  Rounds 1-3 stored zeros there (a compare and two selects per entry and row pass: 72 instructions per panel).  Nothing reads them: the back
// substitution and the block inverses take the strictly lower entries (lf_at(i, k), k < i), and as operands of the rank-8 update they only reach
// accumulator entries above the diagonal of a diagonal tile, which are never published for a row that is not finished.  They stay what they are.
// -DPVIO_DENSE_CONSERVATIVE: the three round-4 source forms of the look-ahead loop whose correctness on the GPU rests on the code the
// compiler happened to generate for them (uniform branches around operand loads have miscompiled before, see the note at the update waves' operand
// requests) fall back to the forms rounds 1-3 shipped.  __graft_entry__.build() passes it whenever hipcc is not the version the kernels were
// last verified with on a GPU (csrc/KNOWN_GOOD_TOOLCHAIN): ~5 % slower, nothing to trust.
#ifdef PVIO_DENSE_CONSERVATIVE
constexpr bool kDenseConservative = true;
#else
constexpr bool kDenseConservative = false;
#endif
constexpr bool kDenseMaskUpper = kDenseConservative;
// pvio_hip_opts::debug_fail_factorizations = kDbgNegativePivot + n: the next n factorizations of the look-ahead form meet a negative pivot (instead of
// having their result discarded): the detection itself is exercised (tests/test_emu_ba.py, tests/test_gpu_ba.py)
constexpr int kDbgNegativePivot = 1000;
constexpr bool kDenseFailAtEnd = !kDenseConservative; // look-ahead form: a bad pivot is detected once per panel (see the factor wave's loop)
// update waves of the look-ahead form: operands of the tile columns requested in groups of four, dead groups skipped
constexpr bool kDenseOperandGroups = !kDenseConservative;
// Round 5: the two hand-overs of a panel of the look-ahead form -- "the panel's L rows are in LDS" (A) and "the next panel's columns are published" (B) --
// through the HARDWARE barrier (two s_barrier per panel, executed by all four waves) instead of two LDS counters polled with s_sleep.  The roles and the
// overlap are the same: the factor wave passes A(p), B(p) back to back and factors panel p + 1 while the update waves, between B(p) and A(p + 1), apply panel p
// to the rest of the trailing matrix.  What goes is the poll: the timing ablations (profiles/r5_ablation_panel_loop.txt) put 1.8 k of a panel's 4.0 k cycles
// into the bare hand-shake, and a wave that sleeps in a poll sees a counter move ~190 cycles after it did (tools/ubench/wave_costs.hip); a barrier
// releases within tens.  14 127 -> 14 449 iterations/s on one box (profiles/r5_ab_la_barrier.txt).  A failed pivot no longer ends the loop early (every
// wave has to meet every barrier): the factorization finishes on NaNs and is discarded behind the loop as before.  -DPVIO_DENSE_LA_COUNTERS: the counter form.
#ifdef PVIO_DENSE_LA_COUNTERS
constexpr bool kDenseLaBarrier = false;
#else
constexpr bool kDenseLaBarrier = true;
#endif
// the per-panel profiling stamps of the look-ahead loop (sites 8-17) cost ~45 scalar instructions and ten branches per panel even when profiling is
// off: compiled in only with -DPVIO_DENSE_LOOP_STAMPS (tests/micro/build_variant.py loop_stamps)
#ifdef PVIO_DENSE_LOOP_STAMPS
#define PV_LOOP_STAMP2(idx) PV_STAMP2(idx)
#else
#define PV_LOOP_STAMP2(idx) ((void)0)
#endif
typedef double mfma_d4 __attribute__((vector_size(32))); // the accumulator of one v_mfma_f64_16x16x4_f64
typedef double lds_d2 __attribute__((vector_size(16)));  // one 16-byte LDS access

// The reduced system lives in LDS as 16 x 16 tiles of the lower block triangle (tile (bi, bk), bk <= bi, at
// (bi (bi + 1) / 2 + bk) * 256 doubles).  Inside a tile the elements are in MFMA accumulator order: lane l of a wave owns
// D[(l >> 4) + 4 r][l & 15], r = 0..3, as four consecutive doubles, so a trailing update moves a tile with two 16-byte
// loads and two 16-byte stores per lane.  Entries that are not yet factored are stored NEGATED (the update is then a plain
// multiply-accumulate, no operand negation); finished L entries are stored as they are.
__device__ __forceinline__ int tile_base(int bi, int bk) { return (((bi * (bi + 1)) >> 1) + bk) << 8; }
__device__ __forceinline__ int tile_off(int r, int c) { return ((((r & 3) << 4) + c) << 2) + (r >> 2); }
// generic (matrix in HBM) path of k_dense: columns per LDS-resident panel -- 32 when [header | 8 vectors | LDV x 34] fits the
// 160 KB of LDS, else 16
__host__ __device__ __forceinline__ int dense_panel_width(int LDV) { return (352 + 8 * LDV + LDV * 34) * 8 <= 160 * 1024 ? 32 : 16; }
__device__ __forceinline__ int mat_at(int i, int k) { return tile_base(i >> 4, k >> 4) + tile_off(i & 15, k & 15); } // k <= i

// ------------------------------------------------------------------------------------------------------
// k_linearize
// ------------------------------------------------------------------------------------------------------
struct Pro { // prologue result, one copy in LDS per WG
    int mode, cur, lin, out_set, eval_buf, valid, done, repeat;
    double mu_schur, ca, cb;
};

// LDS carve (doubles) common to every role: [est N*16][frec N*28][scratch 160][pro 16]
__device__ __forceinline__ int common_lds_doubles(int N) { return N * 16 + N * kFrameRec + 160 + 16; }

__device__ __forceinline__ void lin_prologue(const View &v, double *lds, Pro *&pro_out) {
    const int N = v.dm.N, d = v.dm.d, tid = threadIdx.x;
    double *est = lds, *frec = lds + N * 16, *scratch = frec + N * kFrameRec;
    Pro *pro = reinterpret_cast<Pro *>(scratch + 160);
    pro_out = pro;
    // Wave 0 does the whole prologue.  Everything it reads was written by the previous kernels on other XCDs, so every
    // load costs a trip through the fabric: ALL of them are requested up front, whatever the mode turns out to be (control
    // block, back-substitution partials, both state buffers of this lane's frame, both step vectors, the static frame
    // data), then used -- one trip instead of the three dependent ones of "read the mode, then the partials, then the
    // states" (10.3k -> see profiles/NOTES_r1_r3.md section 5).  The dogleg scalars stay in lane 0; ca / cb / valid reach the frame lanes by
    // lane reads, so the only barrier is the one that publishes the records to the other waves.
    if (tid < 64) {
        const Ctrl *c = v.ctrl;
        const int lane = tid, f = lane < N ? lane : N - 1;
        // (1) control block
        const int done = c->done; // (tested after everything has been requested: a test up front is a fabric trip of its own)
        const int mode = c->mode, cur = c->cur, lin = c->lin, dbg_invalid_left = c->dbg_invalid_left, iter = c->iter;
        const double c_mu = c->mu, radius = c->radius;
        // the candidate records of both iteration parities (Dims::reuse_cand), one double per lane.  Workgroup 0 of THIS launch rewrites the record of
        // this iteration's parity while other workgroups may still be loading it: only the OTHER parity's values are used below (a read of a word that
        // is being overwritten by a value nobody looks at)
        const double recv = v.cand_rec[lane & 15];
        const double c_g2 = c->pose_g2, c_lm_g2 = c->lm_g2, c_gn2 = c->pose_gn2, c_gdot = c->pose_gdot, c_qvv = c->pose_qvv, c_qvy = c->pose_qvy,
                     c_qyy = c->pose_qyy, c_gy = c->pose_gy;
        // (2) partial sums of the landmark back-substitution (<= 64 rows unless the window is large)
        double b[7] = {0, 0, 0, 0, 0, 0, 0}; // [6]: pose part of v^T H v when k_backsub forms it (Dims::qvv_back; zero otherwise)
        for (int row = lane; row < v.dm.n_back_rows; row += 64)
#pragma unroll
            for (int k = 0; k < 7; ++k) b[k] += v.back_part[row * kNumBackScal + k];
        // (3) this lane's frame: both state buffers, both step vectors, activity flags, camera / weight records
        double x0[16], x1[16], vs[15], ys[15];
#pragma unroll
        for (int k = 0; k < 16; ++k) x0[k] = v.fs[(size_t)f * 16 + k], x1[k] = v.fs[((size_t)N + f) * 16 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) vs[k] = v.vstep[f * d + k], ys[k] = v.ystep[f * d + k];
#pragma unroll
        for (int k = 6; k < 15; ++k) vs[k] = 0.0, ys[k] = 0.0;
        if (d == 15) {
#pragma unroll
            for (int k = 6; k < 15; ++k) vs[k] = v.vstep[f * d + k], ys[k] = v.ystep[f * d + k];
        }
        const bool p_act = v.pose_active[f], m_act = v.motion_active[f];
        double cam7[7], w4[4];
#pragma unroll
        for (int k = 0; k < 7; ++k) cam7[k] = v.cam_ext[7 * f + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = v.sic[4 * f + k];

        if (done) { // the solve has terminated: this launch is a no-op slot of the graph; nothing is written
            if (lane == 0) pro->done = 1, pro->valid = 0, pro->repeat = 0;
            goto prologue_out;
        }
        double ca = 0, cb = 0;
        int valid = 1;
        if (mode == MODE_CANDIDATE) {
            // ---- DoglegStrategy::ComputeTraditionalDoglegStep on the scalars of the accepted linearization ----
#pragma unroll
            for (int k = 0; k < 7; ++k) b[k] = wave_sum(b[k]);
            if (lane == 0) {
                const double g2 = c_g2 + c_lm_g2, gn2 = c_gn2 + b[0], gdot = c_gdot + b[1];
                const double qvv = (c_qvv + b[6]) + b[2], qvy = c_qvy + b[3], qyy = c_qyy + b[4], gy = c_gy + b[5];
                const double gradient_norm = sqrt(g2), gauss_newton_norm = sqrt(gn2);
                const double alpha = g2 / qvv; // |g^|^2 / |J (g^/D)|^2
                double sn;
                if (gauss_newton_norm <= radius) {
                    ca = 0.0, cb = 1.0, sn = gauss_newton_norm;
                } else if (gradient_norm * alpha >= radius) {
                    ca = -(radius / gradient_norm), cb = 0.0, sn = radius;
                } else {
                    const double b_dot_a = -alpha * gdot;
                    const double a_squared_norm = (alpha * gradient_norm) * (alpha * gradient_norm);
                    const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + gn2;
                    const double cc = b_dot_a - a_squared_norm;
                    const double dd = sqrt(cc * cc + b_minus_a_squared_norm * (radius * radius - a_squared_norm));
                    const double beta = (cc <= 0) ? (dd - cc) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (dd + cc);
                    ca = -alpha * (1.0 - beta), cb = beta;
                    sn = sqrt(ca * ca * g2 + 2 * ca * cb * gdot + cb * cb * gn2);
                }
                // model_cost_change = -(J step)^T (r + J step / 2) = -(g_s^T step + step^T H_s step / 2), step = ca v + cb y'
                const double gs = ca * g2 + cb * gy;
                const double q = ca * ca * qvv + 2 * ca * cb * qvy + cb * cb * qyy;
                const double mcc = -(gs + 0.5 * q);
                valid = (mcc > 0.0 && dbg_invalid_left <= 0) ? 1 : 0; // (fault injection: tests only)
                if (blockIdx.x == 0) {
                    Ctrl *cw = v.ctrl;
                    cw->ca = ca, cw->cb = cb, cw->dogleg_step_norm = sn, cw->model_cost_change = mcc;
                }
            }
            ca = readlane_f64(ca, 0), cb = readlane_f64(cb, 0);
            valid = __builtin_amdgcn_readlane(valid, 0);
        }
        // ---- Dims::reuse_cand: is this the candidate the last iteration evaluated and rejected? ----
        // Ceres evaluates it again: after a rejected step DoglegStrategy keeps the Gauss-Newton step and the gradient and only halves the
        // radius, so while |gn| <= radius the step -- and with it the candidate x (+) step, its cost and its Jacobian -- is bit for bit the one
        // just rejected (the benchmark window ends in four such iterations, the keyframe solves of the rendered sequence consist of them).
        // Same coefficients on the same accepted iterate, linearization set and mu = same candidate: nothing of the evaluation is redone, the
        // partial rows, the reduced system and the candidate buffers of the last launch stand, and k_dense takes the same decision again.
        int repeat = 0;
        {
            const int po = 8 * ((iter + 1) & 1); // the previous iteration's record
            const double p_ca = readlane_f64(recv, po), p_cb = readlane_f64(recv, po + 1), p_mu = readlane_f64(recv, po + 2), p_lin = readlane_f64(recv, po + 3),
                         p_cur = readlane_f64(recv, po + 4), p_it = readlane_f64(recv, po + 5), p_ev = readlane_f64(recv, po + 6);
            repeat = v.dm.reuse_cand && mode == MODE_CANDIDATE && valid && dbg_invalid_left <= 0 && p_ev == 1.0 && p_it == (double)(iter - 1) && p_lin == (double)lin &&
                     p_cur == (double)cur && p_mu == c_mu && p_ca == ca && p_cb == cb;
            if (lane == 0 && blockIdx.x == 0 && v.dm.reuse_cand) {
                double *mine = v.cand_rec + 8 * (iter & 1);
                mine[0] = ca, mine[1] = cb, mine[2] = c_mu, mine[3] = (double)lin, mine[4] = (double)cur, mine[5] = (double)iter;
                mine[6] = (mode == MODE_CANDIDATE && valid) ? 1.0 : 0.0;
                v.cand_rec[16] = repeat ? 1.0 : 0.0;
                if (repeat) v.ctrl->cand_repeats += 1;
            }
        }
        if (lane == 0) {
            pro->mode = mode, pro->cur = cur, pro->lin = lin;
            pro->valid = valid, pro->done = 0, pro->repeat = repeat;
            pro->ca = ca, pro->cb = cb;
            pro->out_set = (mode == MODE_CANDIDATE) ? 1 - lin : lin;
            pro->eval_buf = (mode == MODE_CANDIDATE) ? 1 - cur : cur;
            pro->mu_schur = (mode == MODE_CANDIDATE) ? fmax(1e-8, 2.0 * c_mu / 10.0) : c_mu; // StepAccepted's mu, assumed
        }
        // ---- evaluation-point frame states (candidate = x (+) delta) -----------------------------------
        double step2 = 0, norm2 = 0;
        if (lane < N) {
            double x[16], y[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = cur ? x1[k] : x0[k], y[k] = x[k];
            if (mode == MODE_CANDIDATE && valid) {
                if (p_act) {
                    double dl[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) dl[k] = ca * vs[k] + cb * ys[k];
                    pose_plus(y, x, dl, dl + 3);
                }
                if (m_act && d == 15) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) y[7 + k] = x[7 + k] + (ca * vs[6 + k] + cb * ys[6 + k]);
                }
                if (p_act) {
#pragma unroll
                    for (int k = 0; k < 7; ++k) step2 += (y[k] - x[k]) * (y[k] - x[k]), norm2 += y[k] * y[k];
                }
                if (m_act) {
#pragma unroll
                    for (int k = 7; k < 16; ++k) step2 += (y[k] - x[k]) * (y[k] - x[k]), norm2 += y[k] * y[k];
                }
                if (blockIdx.x == 0) {
                    double *out = v.fs + ((size_t)(1 - cur) * N + lane) * 16;
#pragma unroll
                    for (int k = 0; k < 16; ++k) out[k] = y[k];
                }
            } else {
                if (p_act) {
#pragma unroll
                    for (int k = 0; k < 7; ++k) norm2 += y[k] * y[k];
                }
                if (m_act) {
#pragma unroll
                    for (int k = 7; k < 16; ++k) norm2 += y[k] * y[k];
                }
            }
            double *ye = est + lane * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) ye[k] = y[k];
            frame_record(frec + lane * kFrameRec, y, cam7, w4);
        }
        step2 = wave_sum(step2), norm2 = wave_sum(norm2); // N <= 32 < 64: wave 0 holds all frame partials
        if (lane == 0 && blockIdx.x == 0) {
            Ctrl *cw = v.ctrl;
            cw->cand_step2_pose = step2, cw->cand_norm2_pose = norm2;
            cw->lin_result = !valid ? LIN_INVALID_STEP : (mode == MODE_INIT || mode == MODE_MARG ? LIN_INIT : (mode == MODE_CANDIDATE ? LIN_CANDIDATE : LIN_RELIN));
        }
    }
prologue_out:
    __syncthreads();
}

// One 3x3 tile task = rows 6 fi + 3 si .., cols 6 fj + 3 sj .. of the pose system (fi <= fj).
__device__ __forceinline__ void unpack_task(int t, int &fi, int &fj, int &si, int &sj) {
    fi = t & 255, fj = (t >> 8) & 255, si = (t >> 16) & 1, sj = (t >> 17) & 1;
}

// per-landmark LDS record (doubles): U[6N] JT[12N] Z[16N] GT[6N] HAA[36] GA[6] SC[4]
// Z[f] = [Jd0 Jd1 r0 r1 | J_ref row 0 (6) | J_ref row 1 (6)] of the factor targeting frame f (zero if none)
__device__ __forceinline__ int lm_rec_doubles(int N) { return 40 * N + 46; }

// index of the 3x3 tile task (fi <= fj, si, sj) in the host's enumeration (ba_solver.cpp: fi-major, fj, then 2 si + sj)
__device__ __forceinline__ int task_index(int N, int fi, int fj, int si, int sj) {
    return 4 * (fi * N - ((fi * (fi - 1)) >> 1) + (fj - fi)) + 2 * si + sj;
}
// address (element-major partial layout) of the entry that couples coordinate i of frame fa with coordinate j of frame
// fb, as the consumers of the partials read it: off-diagonal blocks once (smaller frame = row of the task), diagonal
// blocks as they are addressed
struct PartialEntry {
    int el, t; // element 0..8 of 3x3 task t: offset el * n_tasks + t
};
__device__ __forceinline__ PartialEntry partial_entry(int N, int fa, int i, int fb, int j) {
    if (fa <= fb) return PartialEntry{3 * (i % 3) + (j % 3), task_index(N, fa, fb, i / 3, j / 3)};
    return PartialEntry{3 * (j % 3) + (i % 3), task_index(N, fb, fa, j / 3, i / 3)};
}
// entry of the staged half [e_lo, e_lo + e_n) of the partial row (LDS)
__device__ __forceinline__ void stage_put(double *stage, int n_tasks, int e_lo, int e_n, PartialEntry pe, double val, bool add) {
    if (pe.el < e_lo || pe.el >= e_lo + e_n) return;
    double *dst = stage + ((pe.el - e_lo) * n_tasks + pe.t);
    *dst = add ? *dst + val : val;
}

// The landmark role of small and medium windows: every thread keeps T 3x3 tiles of the pose system in registers and adds every landmark's direct
// and Schur terms to them (right when a workgroup has a handful of landmarks: no set-up, no flush beyond one store per entry); chunks are dealt
// round-robin to the workgroups.  Large windows (Dims::lm_mm: workgroups that walk many chunks) take role_landmarks_tp (ba_lin_tp.h) instead.
template <int T>
__device__ __forceinline__ void role_landmarks(const View &v, double *lds, const Pro *pro, int wg, int n_wg) {
    const int N = v.dm.N, M = v.dm.M, P6 = v.dm.P6, tid = threadIdx.x, n_tasks = v.dm.n_tasks;
    const double *frec = lds + N * 16;
    double *scratch = lds + N * 16 + N * kFrameRec;
    double *chunk = lds + common_lds_doubles(N);
    const int rec = lm_rec_doubles(N), slots = v.dm.lm_slots;
    double *rho_eval = chunk + (size_t)slots * rec; // [slots]
    int *anch = reinterpret_cast<int *>(rho_eval + slots);
    int *active = anch + slots + (slots & 1);
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

    int tfi[T], tfj[T], tsi[T], tsj[T];
    double acc[T][9];
    {
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const int t = k * kLinThreads + tid;
            int d = (t < n_tasks) ? v.task_desc[t] : 0;
            unpack_task(d, tfi[k], tfj[k], tsi[k], tsj[k]);
#pragma unroll
            for (int e = 0; e < 9; ++e) acc[k][e] = 0.0;
        }
    }
    double *pS = v.part_S + (size_t)wg * n_tasks * 9;
    double vg = 0, vrhs = 0, vdiag = 0;                         // vector tasks (tid < P6)
    double s_cost = 0, s_g2 = 0, s_step2 = 0, s_norm2 = 0, s_bad = 0, s_bmax = 0; // scalars

    for (int ck = wg; ck < v.dm.n_chunks; ck += n_wg) { // chunks are dealt round-robin
        const int l0 = v.chunk_lm[ck], l1 = v.chunk_lm[ck + 1], ns = l1 - l0;
        const int o0 = v.lm_ptr[l0], nf = v.lm_ptr[l1] - o0;
        PV_STAMP(0, 2);
        // ---- phase 0: clear the chunk records, evaluation-point inverse depths ----
        for (int e = tid; e < ns * rec; e += kLinThreads) chunk[e] = 0.0;
        if (tid < ns) {
            const int l = l0 + tid;
            double r = rho_cur[l];
            const bool used = v.lm_ptr[l + 1] > v.lm_ptr[l];
            if (mode == MODE_CANDIDATE && used) {
                const double dl = v.cl[l] * (ca * i_ghl[l] + cb * i_gnl[l]) / i_Dl[l];
                const double rc = r + dl;
                s_step2 += (rc - r) * (rc - r);
                r = rc;
            }
            if (mode == MODE_CANDIDATE) rho_cand[l] = r;
            if (used) s_norm2 += r * r;
            rho_eval[tid] = r;
            anch[tid] = v.lm_anchor[l];
            int act = 1;
            if (marg) { // bundle_adjustor.cpp:455-461: only tracks the victim frame observes
                act = v.lm_anchor[l] == victim;
                for (int o = v.lm_ptr[l]; o < v.lm_ptr[l + 1]; ++o) act |= v.obs_frame[o] == victim;
            }
            active[tid] = act;
        }
        __syncthreads();
        PV_STAMP(0, 3);
        // ---- phase 1: one thread per reprojection factor ----
        if (tid < nf && active[v.obs_lm[o0 + tid] - l0]) {
            const int o = o0 + tid, l = v.obs_lm[o], s = l - l0, t = v.obs_frame[o], a = anch[s];
            double r[2], Jt[12], Jr[12], Jd[2];
            reproj_eval<true>(frec + t * kFrameRec, frec + a * kFrameRec, rho_eval[s], v.lm_zref[2 * l], v.lm_zref[2 * l + 1],
                              v.obs_z[2 * (size_t)o], v.obs_z[2 * (size_t)o + 1], r, Jt, Jr, Jd);
            const double sq = r[0] * r[0] + r[1] * r[1];
            // duplicate residual blocks (bundle_adjustor.cpp:165-179): m copies of the block, each robustified on its own, summed by
            // Ceres = the robustified block scaled by sqrt(m), its cost by m.  marginalize_frame lists every block once (:455-510).
            const double mult = (v.lm_mult && !marg) ? v.lm_mult[l] : 1.0;
            s_cost += mult * (0.5 * log(1.0 + sq));                         // CauchyLoss(1): rho(s) = log(1 + s)
            double bad = isfinite(sq) ? 0.0 : 1.0;
            // Corrector, rho'' < 0: sqrt(rho'); marginalization uses the un-robustified Jacobians (:487-510) of ALL blocks
            double sw = marg ? 1.0 : sqrt(fmax(DBL_MIN, 1.0 / (1.0 + sq)));
            if (mult != 1.0) sw *= sqrt(mult);
            const bool tfix = !marg && v.frame_fixed[t] != 0, afix = !marg && v.frame_fixed[a] != 0;
            r[0] *= sw, r[1] *= sw, Jd[0] *= sw, Jd[1] *= sw;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                bad += isfinite(Jt[k]) && isfinite(Jr[k]) ? 0.0 : 1.0;
                Jt[k] = tfix ? 0.0 : Jt[k] * sw;                             // constant blocks have no Jacobian
                Jr[k] = afix ? 0.0 : Jr[k] * sw;
            }
            s_bad += bad;
            double *R = chunk + (size_t)s * rec;
            double *U = R, *JT = R + 6 * N + 12 * t, *Z = R + 18 * N + 16 * t, *GT = R + 34 * N + 6 * t;
            double wt[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                wt[k] = Jd[0] * Jt[k] + Jd[1] * Jt[6 + k];
                U[6 * t + k] += wt[k]; // += : a landmark may list the same target frame twice (bundle_adjustor.cpp:165-179)
                GT[k] += Jt[k] * r[0] + Jt[6 + k] * r[1];
                o_Wt[(size_t)o * 6 + k] = wt[k];
            }
            // duplicates of one (landmark, frame) pair are rare; the records hold the LAST factor's J (used only for
            // the direct blocks), so refuse duplicates at upload time instead of silently mis-summing.
#pragma unroll
            for (int k = 0; k < 12; ++k) JT[k] = Jt[k], Z[4 + k] = Jr[k];
            Z[0] = Jd[0], Z[1] = Jd[1], Z[2] = r[0], Z[3] = r[1];
        }
        __syncthreads();
        PV_STAMP(0, 4);
        // ---- phase 1b: per-landmark sums over its factors (50 outputs per landmark) ----
        // every output is sum_f Z[f][p] Z[f][q] + Z[f][p2] Z[f][q2] for an index quadruple that depends only on k:
        // branch-free, two independent accumulators so that the LDS reads of consecutive frames overlap
        for (int e = tid; e < ns * 50; e += kLinThreads) {
            const int s = e / 50, k = e - 50 * s;
            int p, q, p2, q2;
            if (k == 0) p = 0, q = 0, p2 = 1, q2 = 1;                       // Hll  = Jd . Jd
            else if (k == 1) p = 0, q = 2, p2 = 1, q2 = 3;                  // bl   = Jd . r
            else if (k < 8) p = 0, q = 2 + k, p2 = 1, q2 = 8 + k;           // Wa_c = Jd . Jr[:, c]      (c = k - 2)
            else if (k < 14) p = k - 4, q = 2, p2 = k + 2, q2 = 3;          // GA_c = Jr[:, c] . r       (c = k - 8)
            else {                                                           // HAA_ij = Jr[:, i] . Jr[:, j]
                const int i = (k - 14) / 6, j = (k - 14) - 6 * i;
                p = 4 + i, q = 4 + j, p2 = 10 + i, q2 = 10 + j;
            }
            const double *Z = chunk + (size_t)s * rec + 18 * N;
            double s0 = 0, s1 = 0;
            int f = 0;
            for (; f + 1 < N; f += 2) {
                const double *Za = Z + 16 * f, *Zb = Za + 16;
                s0 += Za[p] * Za[q] + Za[p2] * Za[q2];
                s1 += Zb[p] * Zb[q] + Zb[p2] * Zb[q2];
            }
            if (f < N) s0 += Z[16 * f + p] * Z[16 * f + q] + Z[16 * f + p2] * Z[16 * f + q2];
            const double sum = s0 + s1;
            double *W = chunk + (size_t)s * rec + 40 * N; // HAA[36] GA[6] SC[4]
            if (k == 0) W[42 + 3] = sum;        // Hll
            else if (k == 1) W[42 + 1] = sum;   // bl
            else if (k < 8) {                   // Wa: anchor columns of the landmark's row + global copy for k_backsub
                o_Wa[(size_t)(l0 + s) * 6 + (k - 2)] = sum;
                chunk[(size_t)s * rec + 6 * anch[s] + (k - 2)] += sum; // anchor is never a target of its own landmark
            } else if (k < 14) W[36 + (k - 8)] = sum;
            else W[k - 14] = sum;
        }
        __syncthreads();
        PV_STAMP(0, 5);
        // ---- phase 1c: per-landmark scalars: Jacobi scale, dogleg diagonal, Schur weight ----
        if (tid < ns) {
            const int l = l0 + tid;
            double *SC = chunk + (size_t)tid * rec + 40 * N + 42;
            const double Hll = SC[3], b = SC[1];
            const bool used = v.lm_ptr[l + 1] > v.lm_ptr[l];
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
            const double Dl = sqrt(fmin(fmax(d2, 1e-6), 1e32));     // DoglegStrategy diagonal (min/max_lm_diagonal)
            const double gh = cl * b / Dl;
            const double A = d2 + mu * Dl * Dl;                     // e-block: E^T E + mu D^2
            SC[0] = used ? cl * cl / A : 0.0;                       // Schur weight on the UNscaled W rows
            if (marg) { // scalar inverse of the landmark block, skipped when not finite (:537-538)
                const double inv = 1.0 / Hll;
                SC[0] = (active[tid] && isfinite(inv)) ? inv : 0.0;
            }
            o_Hll[l] = Hll, o_bl[l] = b, o_Dl[l] = Dl, o_ghl[l] = used ? gh : 0.0;
            if (used) {
                s_g2 += gh * gh;
                s_bmax = fmax(s_bmax, fabs(b));
            }
        }
        __syncthreads();
        PV_STAMP(0, 6);
        // ---- phase 2: output-stationary tile accumulation over the chunk's landmarks ----
        // direct-term operand offsets per task (relative to the landmark record); which pair applies depends on the
        // landmark's anchor only:  fi == fj      : Jt(fi)^T Jt(fi)  [+ HAA when the anchor is fi]
        //                          anchor == fj  : Jt(fi)^T Jr(fi)      anchor == fi : Jr(fj)^T Jt(fj)
        for (int s = 0; s < ns; ++s) {
            const double *R = chunk + (size_t)s * rec;
            const double *U = R, *HAA = R + 40 * N, *SC = HAA + 42;
            const double w = SC[0];
            const int a = anch[s];
#pragma unroll
            for (int k = 0; k < T; ++k) {
                if (k * kLinThreads + tid >= n_tasks) continue;
                const int fi = tfi[k], fj = tfj[k], si = tsi[k], sj = tsj[k], r0 = 6 * fi + 3 * si, c0 = 6 * fj + 3 * sj;
                const bool diag = fi == fj, a_is_j = a == fj, a_is_i = a == fi;
                // JT(f) at 6N + 12 f, JR(f) at 18N + 16 f + 4 ; second row is +6 in both
                const int xo = (diag || a_is_j) ? 6 * N + 12 * fi + 3 * si : 18 * N + 16 * fj + 4 + 3 * si;
                const int yo = diag ? 6 * N + 12 * fi + 3 * sj : (a_is_j ? 18 * N + 16 * fi + 4 + 3 * sj : 6 * N + 12 * fj + 3 * sj);
                const double fl = (diag || a_is_j || a_is_i) ? 1.0 : 0.0, hf = (diag && a_is_i) ? 1.0 : 0.0;
                const double ur0 = w * U[r0], ur1 = w * U[r0 + 1], ur2 = w * U[r0 + 2];
                const double uc0 = U[c0], uc1 = U[c0 + 1], uc2 = U[c0 + 2];
                const double *X = R + xo, *Y = R + yo, *Hh = HAA + 18 * si + 3 * sj;
                const double x0 = fl * X[0], x1 = fl * X[1], x2 = fl * X[2], x3 = fl * X[6], x4 = fl * X[7], x5 = fl * X[8];
                const double y0 = Y[0], y1 = Y[1], y2 = Y[2], y3 = Y[6], y4 = Y[7], y5 = Y[8];
                acc[k][0] += x0 * y0 + x3 * y3 + hf * Hh[0] - ur0 * uc0;
                acc[k][1] += x0 * y1 + x3 * y4 + hf * Hh[1] - ur0 * uc1;
                acc[k][2] += x0 * y2 + x3 * y5 + hf * Hh[2] - ur0 * uc2;
                acc[k][3] += x1 * y0 + x4 * y3 + hf * Hh[6] - ur1 * uc0;
                acc[k][4] += x1 * y1 + x4 * y4 + hf * Hh[7] - ur1 * uc1;
                acc[k][5] += x1 * y2 + x4 * y5 + hf * Hh[8] - ur1 * uc2;
                acc[k][6] += x2 * y0 + x5 * y3 + hf * Hh[12] - ur2 * uc0;
                acc[k][7] += x2 * y1 + x5 * y4 + hf * Hh[13] - ur2 * uc1;
                acc[k][8] += x2 * y2 + x5 * y5 + hf * Hh[14] - ur2 * uc2;
            }
            if (tid < P6) {
                const int f = tid / 6, c = tid - 6 * f;
                const double *JT = R + 6 * N, *GT = R + 34 * N, *GA = HAA + 36;
                vg += GT[6 * f + c] + (a == f ? GA[c] : 0.0);
                vrhs += w * SC[1] * U[tid];
                vdiag += JT[12 * f + c] * JT[12 * f + c] + JT[12 * f + 6 + c] * JT[12 * f + 6 + c] + (a == f ? HAA[7 * c] : 0.0);
            }
        }
        __syncthreads();
        PV_STAMP(0, 7);
    }
    // ---- flush this WG's partial ----
    {
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const int t = k * kLinThreads + tid;
            if (t < n_tasks)
#pragma unroll
                for (int e = 0; e < 9; ++e) pS[(size_t)e * n_tasks + t] = acc[k][e]; // element-major: coalesced across t
        }
    }
    if (tid < P6) {
        double *pv = v.part_vec + (size_t)wg * kNumPoseVec * P6;
        pv[tid] = vg, pv[P6 + tid] = vrhs, pv[2 * P6 + tid] = vdiag;
    }
    double sc[6] = {s_cost, s_g2, s_step2, s_norm2, s_bad, s_bmax};
    block_sum<6, true>(sc, scratch);
    if (tid == 0) {
        double *ps = v.part_scal + (size_t)wg * kNumLinScal;
        ps[0] = sc[0], ps[1] = sc[1], ps[2] = sc[2], ps[3] = sc[3], ps[4] = sc[5], ps[5] = sc[4], ps[6] = 0, ps[7] = 0;
    }
    PV_STAMP(0, 8);
}

#include "ba_lin_tp.h" // role_landmarks_tp: the landmark role of large windows (Dims::lm_mm)

#if defined(__clang__)
#define PV_MIN_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, 8))) // register budget of a kernel: at most 512 / n per wave ((1, 8) is the compiler's default range)
#else
#define PV_MIN_WAVES_PER_EU(n) // (the emulator's host compiler)
#endif

// ---- plane-distance factors: one thread per factor, rows staged in LDS, same tile machinery (sign +) ----
template <int T>
__device__ __forceinline__ void role_planes(const View &v, double *lds, const Pro *pro, int wg, int n_wg, int part_row) {
    const int N = v.dm.N, P6 = v.dm.P6, tid = threadIdx.x, n_tasks = v.dm.n_tasks;
    const double *frec = lds + N * 16;
    double *scratch = lds + N * 16 + N * kFrameRec;
    double *rows = lds + common_lds_doubles(N); // [slots][P6 + 2]  (row, r, pad)
    const int slots = v.dm.plane_slots, rs = P6 + 2;
    int tfi[T], tfj[T], tsi[T], tsj[T];
    double acc[T][9];
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const int t = k * kLinThreads + tid;
        int d = (t < n_tasks) ? v.task_desc[t] : 0;
        unpack_task(d, tfi[k], tfj[k], tsi[k], tsj[k]);
#pragma unroll
        for (int e = 0; e < 9; ++e) acc[k][e] = 0.0;
    }
    double vg = 0, vdiag = 0, s_cost = 0, s_bad = 0;
    for (int ck = wg; ck < v.dm.n_plane_chunks; ck += n_wg) {
        const int f0 = v.plane_chunk[ck], ns = v.plane_chunk[ck + 1] - f0;
        for (int e = tid; e < ns * rs; e += kLinThreads) rows[e] = 0.0;
        __syncthreads();
        if (tid < ns) {
            const int f = f0 + tid, b = v.plane_ptr[f], K = v.plane_ptr[f + 1] - b;
            bool any_free = false;
            for (int k = 0; k < K; ++k) any_free |= (v.frame_fixed[v.plane_frame[b + k]] == 0);
            if (any_free) { // all-constant residual blocks are folded into Ceres' fixed_cost and dropped
                double *row = rows + (size_t)tid * rs;
                double r;
                plane_eval_row(K, v.plane_frame + b, v.plane_z + 2 * (size_t)b, frec, v.plane_normal + 3 * f, v.plane_dist[f], v.plane_sic, &r, row);
                const double sq = r * r;
                s_cost += 0.5 * log(1.0 + sq);
                s_bad += isfinite(sq) ? 0.0 : 1.0;
                const double sw = sqrt(fmax(DBL_MIN, 1.0 / (1.0 + sq)));
                for (int k = 0; k < K; ++k) {
                    const int fr = v.plane_frame[b + k];
                    const bool fx = v.frame_fixed[fr] != 0;
                    for (int c = 0; c < 6; ++c) row[6 * fr + c] = fx ? 0.0 : row[6 * fr + c];
                }
                for (int c = 0; c < P6; ++c) row[c] *= sw;
                row[P6] = r * sw;
            }
        }
        __syncthreads();
        for (int s = 0; s < ns; ++s) {
            const double *U = rows + (size_t)s * rs;
#pragma unroll
            for (int k = 0; k < T; ++k) {
                if (k * kLinThreads + tid >= n_tasks) continue;
                const int r0 = 6 * tfi[k] + 3 * tsi[k], c0 = 6 * tfj[k] + 3 * tsj[k];
                const double ur0 = U[r0], ur1 = U[r0 + 1], ur2 = U[r0 + 2];
                const double uc0 = U[c0], uc1 = U[c0 + 1], uc2 = U[c0 + 2];
                acc[k][0] += ur0 * uc0, acc[k][1] += ur0 * uc1, acc[k][2] += ur0 * uc2;
                acc[k][3] += ur1 * uc0, acc[k][4] += ur1 * uc1, acc[k][5] += ur1 * uc2;
                acc[k][6] += ur2 * uc0, acc[k][7] += ur2 * uc1, acc[k][8] += ur2 * uc2;
            }
            if (tid < P6) {
                vg += U[tid] * U[P6];
                vdiag += U[tid] * U[tid];
            }
        }
        __syncthreads();
    }
    double *pS = v.part_S + (size_t)part_row * n_tasks * 9;
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const int t = k * kLinThreads + tid;
        if (t < n_tasks)
#pragma unroll
            for (int e = 0; e < 9; ++e) pS[(size_t)e * n_tasks + t] = acc[k][e];
    }
    if (tid < P6) {
        double *pv = v.part_vec + (size_t)part_row * kNumPoseVec * P6;
        pv[tid] = vg, pv[P6 + tid] = 0.0, pv[2 * P6 + tid] = vdiag;
    }
    double sc[2] = {s_cost, s_bad};
    block_sum<2>(sc, scratch);
    if (tid == 0) {
        double *ps = v.part_scal + (size_t)part_row * kNumLinScal;
        ps[0] = sc[0], ps[1] = 0, ps[2] = 0, ps[3] = 0, ps[4] = 0, ps[5] = sc[1], ps[6] = 0, ps[7] = 0;
    }
}

// ---- IMU pre-integration factor j (frames j-1 -> j): one WG ------------------------------------------------
__device__ __forceinline__ void role_preint(const View &v, double *lds, const Pro *pro, int j) {
    const int N = v.dm.N, tid = threadIdx.x;
    const double *est = lds;
    double *work = lds + common_lds_doubles(N); // raw[16] G[450] J[450] r[16] U[225]
    double *raw = work, *G = work + 16, *J = work + 466, *r = work + 916, *Ul = work + 932;
    const int i = j - 1;
    // all threads: clear G and bring the 15 x 15 square-root information into LDS while thread 0 does the factor's
    // scalar geometry (a single thread clearing 450 LDS words alone costs ~3 us)
    for (int e = tid; e < 450; e += kLinThreads) G[e] = 0.0;
    if (tid < 225) Ul[tid] = v.pre_U[225 * (size_t)j + tid];
    __syncthreads();
    if (tid == 0) {
        // live bias read: the accepted linearization keeps the biases it was evaluated with (RELIN must reuse them)
        const double *b0 = (pro->mode == MODE_RELIN) ? v.bias0_lin + 6 * i : v.fs_user + 16 * i + 10;
        preint_raw(est + 16 * i, est + 16 * j, b0, v.pre_delta + 11 * j, v.pre_jac + 45 * j, v.imu_ext + 7 * i, v.imu_ext + 7 * j, raw, G, /*G_is_zero=*/true);
        const bool fi = pro->mode != MODE_MARG && v.frame_fixed[i] != 0, fj = pro->mode != MODE_MARG && v.frame_fixed[j] != 0;
        for (int row = 0; row < 15; ++row)
            for (int c = 0; c < 6; ++c) {
                if (fi) G[row * 30 + c] = 0.0;
                if (fj) G[row * 30 + 15 + c] = 0.0;
            }
    }
    __syncthreads();
    const double *U = Ul;
    for (int e = tid; e < 450; e += kLinThreads) {
        const int row = e / 30, c = e - 30 * row;
        double s = 0;
        for (int k = 0; k < 15; ++k) s += U[row * 15 + k] * G[k * 30 + c];
        J[e] = s;
    }
    if (tid < 15) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += U[tid * 15 + k] * raw[k];
        r[tid] = s;
    }
    __syncthreads();
    for (int e = tid; e < 900; e += kLinThreads) {
        const int a = e / 30, b = e - 30 * a;
        double s = 0;
        for (int k = 0; k < 15; ++k) s += J[k * 30 + a] * J[k * 30 + b];
        v.pre_H[(size_t)j * 900 + e] = s;
    }
    if (tid < 30) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += J[k * 30 + tid] * r[k];
        v.pre_g[(size_t)j * 30 + tid] = s;
    }
    if (tid == 0) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += r[k] * r[k];
        v.pre_cost[j] = 0.5 * s;
    }
}

// ---- marginalization prior: block b of nb -------------------------------------------------------------------
// One workgroup per prior frame b (its 15 rows of r = S e + s, t = Lambda e + eta, H = B^T Lambda B): 16 threads
// cooperate on a row, so every row product is one coalesced sweep with all loads in flight.
__device__ __forceinline__ void role_prior(const View &v, double *lds, const Pro *pro, int b, int nb) {
    const int N = v.dm.N, n = v.dm.prior_n, D = 15 * n, tid = threadIdx.x;
    const double *est = lds;
    double *work = lds + common_lds_doubles(N); // e[D] JRI[9n] rs[16] rt[16]
    double *e = work, *JRI = work + D, *rs = JRI + 9 * n, *rt = rs + 16;
    if (tid < n) prior_frame_error(est + 16 * v.prior_frames[tid], v.prior_lin + 16 * tid, e + 15 * tid, JRI + 9 * tid);
    __syncthreads();
    const bool marg = pro->mode == MODE_MARG;
    {
        const int r = tid >> 4, part = tid & 15, row = 15 * b + r; // 15 rows x 16 lanes (tid < 240)
        double ss = 0, tt = 0;
        if (r < 15) {
            const double *Sr = v.prior_S + (size_t)row * D, *Lr = v.prior_Lambda + (size_t)row * D;
            for (int k = part; k < D; k += 16) ss += Sr[k] * e[k], tt += Lr[k] * e[k];
        }
#pragma unroll
        for (int pat = 3; pat >= 0; --pat) ss += dpp_f64(ss, pat), tt += dpp_f64(tt, pat); // 16 lanes = one DPP row
        if (r < 15 && part == 0) rs[r] = ss + v.prior_s[row], rt[r] = tt + v.prior_eta[row];
    }
    __syncthreads();
    if (tid == 0) {
        double c2 = 0;
        for (int r = 0; r < 15; ++r) c2 += rs[r] * rs[r];
        v.prior_cost[b] = 0.5 * c2; // summed over the prior frames by k_dense
    }
    if (tid < 15) { // g = B^T t on this frame's block
        const double *Jm = JRI + 9 * b;
        double g = tid < 3 ? Jm[tid] * rt[0] + Jm[3 + tid] * rt[1] + Jm[6 + tid] * rt[2] : rt[tid];
        if (tid < 6 && v.frame_fixed[v.prior_frames[b]] && !marg) g = 0.0;
        v.prior_g[15 * b + tid] = g;
        // the same number where k_dense finds it without knowing the slot of a frame (prior_gd: [N][30] = gradient, diagonal of H by
        // FRAME, zero for frames the prior does not relate -- cleared at upload; written by the slot k_dense used to look up)
        const int fb = v.prior_frames[b];
        if (v.prior_slot[fb] == b) v.prior_gd[30 * fb + tid] = g;
    }
    // H rows of this frame: H = B^T Lambda B
    for (int idx = tid; idx < 15 * D; idx += kLinThreads) {
        const int ka = idx / D, c = idx - ka * D, a = 15 * b + ka;
        const int ic = c / 15, kc = c - 15 * ic;
        const double *Ja = JRI + 9 * b, *Jc = JRI + 9 * ic;
        double h = 0;
        if (ka < 3 && kc < 3) {
            for (int x = 0; x < 3; ++x)
                for (int y = 0; y < 3; ++y) h += Ja[3 * x + ka] * v.prior_Lambda[(size_t)(15 * b + x) * D + 15 * ic + y] * Jc[3 * y + kc];
        } else if (ka < 3) {
            for (int x = 0; x < 3; ++x) h += Ja[3 * x + ka] * v.prior_Lambda[(size_t)(15 * b + x) * D + c];
        } else if (kc < 3) {
            for (int y = 0; y < 3; ++y) h += v.prior_Lambda[(size_t)a * D + 15 * ic + y] * Jc[3 * y + kc];
        } else {
            h = v.prior_Lambda[(size_t)a * D + c];
        }
        if (!marg && ((ka < 6 && v.frame_fixed[v.prior_frames[b]]) || (kc < 6 && v.frame_fixed[v.prior_frames[ic]]))) h = 0.0;
        v.prior_H[(size_t)a * D + c] = h;
        if (c == a) {
            const int fb = v.prior_frames[b];
            if (v.prior_slot[fb] == b) v.prior_gd[30 * fb + 15 + ka] = h;
        }
    }
    (void)nb;
}

// a plane workgroup that has nothing to do in this mode still owns a partial row k_reduce will sum
__device__ __forceinline__ void zero_partial_row(const View &v, int row) {
    const size_t nS = (size_t)v.dm.n_tasks * 9, nV = (size_t)kNumPoseVec * v.dm.P6;
    for (size_t e = threadIdx.x; e < nS; e += kLinThreads) v.part_S[(size_t)row * nS + e] = 0.0;
    for (size_t e = threadIdx.x; e < nV; e += kLinThreads) v.part_vec[(size_t)row * nV + e] = 0.0;
    if (threadIdx.x < kNumLinScal) v.part_scal[(size_t)row * kNumLinScal + threadIdx.x] = 0.0;
}

// MFMA accumulator tiles per wave for the window sizes a tile count T stands for (T = 1: N <= 10 ... T = 9: N <= 32)
// (large windows, ba_lin_tp.h: the tiles cover 6 N + 1 columns -- b_l rides along as column 6 N -- i.e. 13 block rows = 91 tiles at N = 32)
template <int T> struct TilesPerWave { static constexpr int value = T <= 1 ? 3 : (T <= 2 ? 6 : (T <= 4 ? 12 : (T <= 6 ? 17 : 23))); };

// MM (large windows, Dims::lm_mm): the landmark workgroups run role_landmarks_tp (ba_lin_tp.h); up to 15 frames the kernel is held to 256 registers so
// that two workgroups share a CU when their LDS allows it (tp_landmark_slots: 80 KB each) -- the phases of one overlap the latencies of the other
// TW: accumulator tiles per wave of the large-window role (23 only for 32 frames: 91 tiles; 28 .. 31 frames need 66 .. 78)
template <int T, bool MM, int TW = TilesPerWave<T>::value>
__global__ void __launch_bounds__(kLinThreads) PV_MIN_WAVES_PER_EU((MM && T <= 2) ? 2 : 1) k_linearize(View v) {
    HIP_DYNAMIC_SHARED(double, lds)
    PV_STAMP_BEGIN(0);
    PV_STAMP(0, 0);
    Pro *pro;
    lin_prologue(v, lds, pro);
    PV_STAMP(0, 1);
    if (!pro->valid) return; // terminated solve (no-op slot), or invalid trust-region step: k_dense handles it (HandleInvalidStep)
    if (pro->repeat) return; // the candidate the last launch evaluated (Dims::reuse_cand): its partial rows and candidate buffers stand
    if (blockIdx.x == 0 && v.dm.n_rot > 0) {
        // rotation priors (RotationPriorFactor, no reference counterpart): one thread per frame, at the evaluation point the
        // prologue left in LDS; k_reduce / k_dense add the 3 x 3 blocks, the gradient and the cost like the IMU / prior terms
        const int N = v.dm.N, tid = threadIdx.x;
        double *rc = lds + N * 16 + N * kFrameRec; // scratch (free until the roles start)
        if (tid < N) {
            const int slot = v.rot_slot[tid];
            double r[3] = {0, 0, 0}, H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            if (slot >= 0 && (v.pose_active[tid] || pro->mode == MODE_MARG)) rot_prior_eval(lds + 16 * tid, v.rot_q0 + 4 * slot, v.rot_W + 9 * slot, r, H, g);
            for (int k = 0; k < 9; ++k) v.rot_H[9 * tid + k] = H[k];
            for (int k = 0; k < 3; ++k) v.rot_g[3 * tid + k] = g[k];
            rc[tid] = 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        }
        __syncthreads();
        if (tid == 0) {
            double c2 = 0;
            for (int f = 0; f < N; ++f) c2 += rc[f];
            v.rot_cost[0] = c2;
        }
        __syncthreads();
    }
    // launch order = [IMU | prior | planes | landmarks]: the longest workgroups are dispatched first; b keeps the role
    // numbering [landmarks | planes | IMU | prior] the partial rows use
    const int g0 = v.dm.G_lm, g1 = g0 + v.dm.G_plane, g2 = g1 + v.dm.G_pre, naux = v.dm.G_pre + v.dm.G_prior;
    const int bx = blockIdx.x;
    const int b = bx < naux ? g1 + bx : (bx < naux + v.dm.G_plane ? g0 + (bx - naux) : bx - naux - v.dm.G_plane);
    if (v.dbg && v.dbg_sel < 0 && threadIdx.x == 0 && (b == g1 || b == g2)) v.dbg[b == g1 ? 10 : 12] = clock64(); // first IMU / prior workgroup (absolute)
    if (b < g0) {
        if constexpr (MM) role_landmarks_tp<TW, (T >= 6 ? 2 : 1)>(v, lds, pro, b, g0);
        else role_landmarks<T>(v, lds, pro, b, g0);
    }
    else if (b < g1) {
        if (pro->mode != MODE_MARG) role_planes<T>(v, lds, pro, b - g0, v.dm.G_plane, b);
        else zero_partial_row(v, b);
    }
    else if (b < g2) {
        const int j = b - g1 + 1;
        const int vic = v.ctrl->marg_victim;
        if (v.pre_valid[j] && (pro->mode != MODE_MARG || j == vic || j == vic + 1)) role_preint(v, lds, pro, j);
    } else role_prior(v, lds, pro, b - g2, v.dm.G_prior);
    if (v.dbg && v.dbg_sel < 0 && threadIdx.x == 0 && (b == g1 || b == g2)) v.dbg[b == g1 ? 11 : 13] = clock64();
    PV_STAMP(0, 9);
    if (v.dbg && blockIdx.x == 0 && threadIdx.x == 0) v.dbg[30] = 0, v.dbg[31] = wall_clock64() - g_stamp_w0[0]; // 10 ns units
}

// ------------------------------------------------------------------------------------------------------
// k_reduce: fixed-order sums of the WG partials -> red = [S tiles (element-major) | 3 pose vectors | 8 scalars]
// ------------------------------------------------------------------------------------------------------
constexpr int kRedElems = 16, kRedGroups = 64; // one block = 16 output elements (one 128-byte line per partial row) x 64 partial groups:
                                                // the partials are pulled by ~8x more CUs than there are kilobytes per element
// IMU factor blocks (odd factor first) and marginalization prior added to entry ((fa, ka), (fb, kb)), fb <= fa, of the
// unscaled reduced system whose landmark / plane part is `val`.  In two steps since round 5 -- FETCH the terms (every load's address depends on the entry
// alone, so all of them can be in flight while the partial rows are summed), APPLY them to the sum in the order the one-step form used:
// k_reduce's 6 us were a chain of dependent trips (partials -> task -> factor flags -> factor blocks written by k_linearize on other XCDs).
struct EntryTerms {
    double rot, hA, hB, pri;
    int use; // 1: rotation prior, 2: IMU factor blocks, 4: odd frame (order of the two IMU terms), 8: marginalization prior
};
__device__ __forceinline__ EntryTerms fetch_entry_terms(const View &v, int fa, int ka, int fb, int kb) {
    EntryTerms t;
    t.rot = 0.0, t.hA = 0.0, t.hB = 0.0, t.pri = 0.0, t.use = 0;
    if (v.dm.n_rot > 0 && fa == fb && ka < 3 && kb < 3) t.rot = v.rot_H[9 * fa + 3 * ka + kb], t.use |= 1; // rotation prior of the frame
    if (v.dm.d != 15) return t;
    const int N = v.dm.N;
    if (v.dm.G_pre) {
        // factor j couples frames j - 1 (local 0..14) and j (local 15..29).  The blocks are requested whether or not the factor exists (an absent factor's
        // block was never written: select, below, not multiply): their addresses do not wait for the flags
        const bool same = fa == fb, adj = fa == fb + 1;
        const bool candA = fa >= 1 && (same || adj), candB = same && fa + 1 < N;
        const unsigned pvA = candA ? v.pre_valid[fa] : 0u, pvB = candB ? v.pre_valid[fa + 1] : 0u;
        const double xA = candA ? v.pre_H[(size_t)fa * 900 + (15 + ka) * 30 + (same ? 15 : 0) + kb] : 0.0;
        const double xB = candB ? v.pre_H[(size_t)(fa + 1) * 900 + ka * 30 + kb] : 0.0;
        t.hA = pvA ? xA : 0.0, t.hB = pvB ? xB : 0.0;
        t.use |= 2 | ((fa & 1) ? 4 : 0);
    }
    if (v.dm.prior_n <= 0) return t;
    const int pa = v.prior_slot[fa], pb = v.prior_slot[fb];
    if (pa >= 0 && pb >= 0) t.pri = v.prior_H[(size_t)(15 * pa + ka) * (15 * v.dm.prior_n) + 15 * pb + kb], t.use |= 8;
    return t;
}
__device__ __forceinline__ double apply_entry_terms(const EntryTerms &t, double val) {
    if (t.use & 1) val += t.rot;
    if (t.use & 2) val = (t.use & 4) ? (val + t.hA) + t.hB : (val + t.hB) + t.hA;
    if (t.use & 8) val += t.pri;
    return val;
}
__device__ __forceinline__ double reduced_entry_terms(const View &v, double val, int fa, int ka, int fb, int kb) {
    return apply_entry_terms(fetch_entry_terms(v, fa, ka, fb, kb), val);
}

// Blocks [0, nb_red) produce `red` (tiles element-major, pose vectors, scalars).  With the tile image on (single GPU,
// reduced system resident in the dense kernel's registers) the UNSCALED reduced system is also written entry by entry
// where the dense kernel's tile owners load it (lower block triangle, 16 x 16 tiles, MFMA accumulator order): the thread
// that finishes tile element red[e] adds the IMU factor blocks and the prior and stores the entry; the entries that have
// no landmark / plane part (a velocity or bias coordinate) come from the blocks behind nb_red, which read no partials.
// Entries outside the real lower triangle are never written (the image is zeroed at upload).
// phase 0: single GPU -- partials -> `red` + tile image.  Landmark-sharded runs split it around the all-reduce:
// phase 1: partials -> `red` (+ this rank's max slot), no image;  phase 2 (after the all-reduce): `red` -> tile image, so that
// the dense kernel takes the same fast path (tiles loaded straight into its accumulators) as on one GPU.
__global__ void __launch_bounds__(kRedElems *kRedGroups) k_reduce(View v, int nb_red, int phase) {
    // the control word is only needed before anything is written: its load travels together with the partials
    const int ctl_done = v.ctrl->done, ctl_result = v.ctrl->lin_result;
    if (v.dm.reuse_cand && v.cand_rec[16] != 0.0) return; // k_linearize found the candidate of the last slot again: `red` and the image stand (uniform)
    // Dims::img_scaled: once the Jacobi scaling of the solve exists the image is written as -(C S C) -- entry by entry the product
    // k_dense formed when it loaded the tile, (S_ik (c_i c_k)) negated: what its accumulators hold, so that it loads without touching
    const bool scale_img = v.dm.img_scaled && v.ctrl->scaling_ready;
    const size_t nS = (size_t)v.dm.n_tasks * 9, nV = (size_t)kNumPoseVec * v.dm.P6;
    const size_t total = nS + nV + kNumLinScal;
    if ((int)blockIdx.x >= nb_red) {
        if (ctl_done || ctl_result == LIN_INVALID_STEP) return;
        const int d = v.dm.d, P = v.dm.P;
        const size_t pos = ((size_t)(blockIdx.x - nb_red) * blockDim.x + threadIdx.x);
        if (pos >= (size_t)v.dm.img_sz) return;
        const int ti = (int)(pos >> 8), w = (int)(pos & 255), ln = w >> 2, r = w & 3;
        int bi = (int)((sqrtf(8.0f * ti + 1.0f) - 1.0f) * 0.5f);
        while (((bi + 1) * (bi + 2)) >> 1 <= ti) ++bi;
        while (((bi * (bi + 1)) >> 1) > ti) --bi;
        const int bk = ti - ((bi * (bi + 1)) >> 1);
        const int i = 16 * bi + (ln >> 4) + 4 * r, k = 16 * bk + (ln & 15);
        if (k > i || i >= P) return;
        const int fa = i / d, ka = i - d * fa, fb = k / d, kb = k - d * fb;
        if (ka < 6 && kb < 6) return; // has a landmark / plane part: written by the thread that reduces it
        double val = reduced_entry_terms(v, 0.0, fa, ka, fb, kb);
        if (scale_img) val = -(val * (v.cpl[i] * v.cpl[k]));
        v.img[pos] = val;
        return;
    }
    __shared__ double part[kRedGroups][kRedElems];
    __shared__ double quarter[4][kRedElems];
    const int G = v.dm.G_lm + v.dm.G_plane;
    const int el = threadIdx.x & (kRedElems - 1), gg = threadIdx.x / kRedElems;
    const size_t e = (size_t)blockIdx.x * kRedElems + el;
    // the image entry this thread will finish (gg == 0): its position and every other term of it are requested up front, next to the partial rows
    bool img_here = false;
    int img_row = 0, img_col = 0;
    EntryTerms img_terms;
    img_terms.rot = img_terms.hA = img_terms.hB = img_terms.pri = 0.0, img_terms.use = 0;
    double img_cr = 1.0, img_cc = 1.0;
    if (gg == 0 && v.dm.use_img && phase != 1 && e < nS) {
        const int n_tasks = v.dm.n_tasks, ee = (int)e, q = ee / n_tasks, t = ee - q * n_tasks, d = v.dm.d;
        int fi, fj, si, sj;
        unpack_task(v.task_desc[t], fi, fj, si, sj);
        const int ra = 3 * si + q / 3, ca = 3 * sj + q % 3; // coordinates inside frames fi (row of the task) and fj, fi <= fj
        if (fi != fj || ra >= ca) { // (diagonal blocks come in full: their upper half is not stored)
            img_here = true;
            img_row = fi != fj ? d * fj + ca : d * fi + ra, img_col = fi != fj ? d * fi + ra : d * fi + ca;
            img_terms = fi != fj ? fetch_entry_terms(v, fj, ca, fi, ra) : fetch_entry_terms(v, fi, ra, fi, ca);
            if (v.dm.img_scaled) img_cr = v.cpl[img_row], img_cc = v.cpl[img_col]; // (only used once the scaling exists)
        }
    }
    // stage 1: group gg sums partials g = gg, gg + 16, ... (coalesced across el), four independent accumulators
    double s = 0;
    if (phase == 2) {
        if (gg == 0 && e < total) s = v.red[e]; // the all-reduced value (everything it is combined with below is zero)
    } else if (e < nS + nV) {
        const double *src = e < nS ? v.part_S + e : v.part_vec + (e - nS);
        const size_t stride = e < nS ? nS : nV;
        // <= 4 values per thread per round (one round up to 256 partial rows), all loads issued before the first add
        for (int g0 = gg; g0 < G; g0 += 4 * kRedGroups) {
            double vals[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = g0 + q * kRedGroups;
                vals[q] = g < G ? src[(size_t)g * stride] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s += vals[q];
        }
    } else if (e < total) {
        const int q = (int)(e - nS - nV);
        for (int g = gg; g < G; g += kRedGroups) s = (q == 4) ? fmax(s, v.part_scal[(size_t)g * kNumLinScal + q]) : s + v.part_scal[(size_t)g * kNumLinScal + q];
    }
    if (ctl_done || ctl_result == LIN_INVALID_STEP) return; // uniform
    part[gg][el] = s;
    __syncthreads();
    // stage 2: fixed-order combination of the group sums, four quarters first
    const bool is_max = e >= nS + nV && e < total && (int)(e - nS - nV) == 4;
    if (gg < 4) {
        double r = part[gg * (kRedGroups / 4)][el];
        for (int q = 1; q < kRedGroups / 4; ++q) r = is_max ? fmax(r, part[gg * (kRedGroups / 4) + q][el]) : r + part[gg * (kRedGroups / 4) + q][el];
        quarter[gg][el] = r;
    }
    __syncthreads();
    if (gg == 0 && e < total) {
        double r = quarter[0][el];
        for (int q = 1; q < 4; ++q) r = is_max ? fmax(r, quarter[q][el]) : r + quarter[q][el];
        if (phase != 2) v.red[e] = r;
        if (phase != 2 && is_max && v.dm.world > 1) {
            // landmark-sharded: the maximum cannot ride in a summing all-reduce, so every rank owns one slot behind the
            // scalars (zero in the others' slots); after the sum the slots hold every rank's maximum (k_dense takes the max)
            for (int w = 0; w < v.dm.world; ++w) v.red[total + w] = (w == v.dm.rank) ? r : 0.0;
        }
        if (img_here) {
            double val = apply_entry_terms(img_terms, r);
            if (scale_img) val = -(val * (img_cr * img_cc));
            v.img[mat_at(img_row, img_col)] = val;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// k_dense: the serial part of a trust-region iteration, one workgroup.
// ------------------------------------------------------------------------------------------------------
__device__ int record_trace(const View &v, Ctrl *c, int iteration) {
    const int slot = c->trace_len;
    if (slot >= c->trace_cap) return -1;
    c->trace_len = slot + 1;
    TraceRec &r = v.trace[slot];
    r.iteration = iteration, r.step_is_valid = c->it_valid, r.step_is_successful = c->it_success, r.reserved = 0;
    r.cost = c->it_cost, r.cost_change = c->it_cost_change, r.gradient_max_norm = c->grad_max, r.step_norm = c->it_step_norm;
    r.relative_decrease = c->it_rel, r.trust_region_radius = c->radius, r.mu = c->mu;
    return slot;
}

struct DenseShared {
    int do_solve, do_trace, trace_slot, accepted, first;
    int replay_first, replay_count; // trace slots written by the linear-solver-failure replay
    double x_cost_new;
    long long stamp1, stamp2; // profiling: the two stamp sites every launch passes, published only by launches that go on to factor
};
static_assert(sizeof(DenseShared) <= 16 * sizeof(double), "LDS header layout");

// per-landmark back-substitution for landmarks l = first, first + stride, ...; vs / ys = C_p v_p, C_p y'_p (any memory)
__device__ __forceinline__ void backsub_landmarks(const View &v, int lin, double mu, const double *vs, const double *ys, int first, int stride, double *s) {
    const int M = v.dm.M, d = v.dm.d;
    const size_t Ms = (size_t)M, Fs = (size_t)v.dm.F;
    const double *Hll = v.Hll + lin * Ms, *bl = v.bl + lin * Ms, *Dl = v.Dl + lin * Ms, *ghl = v.ghl + lin * Ms;
    const double *Wa = v.Wa + lin * Ms * 6, *Wt = v.Wt + lin * Fs * 6;
    double *gnl = v.gnl + lin * Ms;
    for (int l = first; l < M; l += stride) {
        const int o0 = v.lm_ptr[l], o1 = v.lm_ptr[l + 1];
        if (o1 == o0) {
            gnl[l] = 0.0;
            continue;
        }
        const int a = v.lm_anchor[l];
        double Wv = 0, Wy = 0; // W_l . (C_p v_p), W_l . (C_p y'_p)
        for (int k = 0; k < 6; ++k) Wv += Wa[(size_t)l * 6 + k] * vs[d * a + k], Wy += Wa[(size_t)l * 6 + k] * ys[d * a + k];
        for (int o = o0; o < o1; ++o) {
            const int t = v.obs_frame[o];
            for (int k = 0; k < 6; ++k) Wv += Wt[(size_t)o * 6 + k] * vs[d * t + k], Wy += Wt[(size_t)o * 6 + k] * ys[d * t + k];
        }
        const double cl = v.cl[l], D = Dl[l], gh = ghl[l];
        const double Hs = cl * cl * Hll[l], A = Hs + mu * D * D;
        const double w = cl * cl / A;
        const double yl = -cl * (bl[l] + Wy) / A; // y'_l
        const double vl = gh / D;
        const double gn = D * yl;
        gnl[l] = gn;
        s[0] += gn * gn;
        s[1] += gh * gn;
        s[2] += w * Wv * Wv + 2 * cl * vl * Wv + Hs * vl * vl;                          // v^T H v (landmark + add-back)
        s[3] += w * Wv * Wy + cl * vl * Wy + cl * yl * Wv + Hs * vl * yl;               // v^T H y'
        s[4] += w * Wy * Wy + 2 * cl * yl * Wy + Hs * yl * yl;                          // y'^T H y'
        s[5] += cl * bl[l] * yl;                                                        // g_s^T y'
    }
}


// Builds the scaled reduced system S_s = C (H_pp - sum_l w_l W_l W_l^T) C in the tile image A (negated, see above): unit
// rows for inactive coordinates, identity in the panel padding, the scaled rhs in row Pp, zeros above it.  The sources
// (landmark / plane tiles of `red`, IMU factor blocks, marginalization prior) are swept in THEIR memory order, so every
// global load is coalesced and all of them are in flight together -- gathering per matrix entry instead costs one L2
// request per lane and load (measured 26 us for the 150 x 150 system against ~3 us for this form).  Terms are added in a
// fixed order (tiles, odd IMU factors, even IMU factors, prior); passes that touch the same entries are separated by
// barriers.  cm[a] = Jacobi scale of coordinate a, 0 if inactive.
#define PV_ORDER() asm volatile("" ::: "memory") // memory operations are not moved across this point by the compiler
#define PV_KEEP(x) asm volatile("" : "+v"(x)) // the value is needed HERE: keeps its load unconditional and where it was written
// flags[j] = IMU factor j is present, pframe[q] = frame of prior slot q (both in LDS)
template <int NT> // threads of the workgroup
__device__ __forceinline__ void dense_build(const View &v, double *A, const double *cm, const double *rhs_s, const int *flags, const int *pframe, int P, int Pp, int nbk) {
    const int tid = threadIdx.x, N = v.dm.N, d = v.dm.d, n_tasks = v.dm.n_tasks;
    constexpr int nthr = NT;
    // (one wave per SIMD: the VALU instruction count of these passes is what they cost, so index arithmetic is hoisted)
    { // pass 0: zero fill, 16 bytes per store (the tile image is contiguous)
        const int nd2 = (nbk * (nbk + 1)) << 6; // tiles * 256 / 2
        lds_d2 z;
        z[0] = 0.0, z[1] = 0.0;
        lds_d2 *A2 = reinterpret_cast<lds_d2 *>(A);
        for (int e = tid; e < nd2; e += nthr) A2[e] = z;
    }
    __syncthreads();
    PV_STAMP(2, 24);
    // pass 0b: unit diagonal of inactive / padding coordinates, scaled rhs in row Pp
    for (int a = tid; a < Pp; a += nthr) {
        if (a < P) A[mat_at(Pp, a)] = -rhs_s[a];
        if (a < P ? cm[a] == 0.0 : true) A[mat_at(a, a)] = -1.0;
    }
    PV_STAMP(2, 25);
    // pass 1: landmark + plane tiles (upper block triangle, element-major) -> lower triangle; one plain store per entry
    for (int t = tid; t < n_tasks; t += nthr) {
        int fi, fj, si, sj;
        unpack_task(v.task_desc[t], fi, fj, si, sj);
        double h[9];
#pragma unroll
        for (int el = 0; el < 9; ++el) h[el] = v.red[(size_t)el * n_tasks + t];
        const int r0 = d * fi + 3 * si, c0 = d * fj + 3 * sj; // fi <= fj => r0 <= c0
        double cr[3], cc3[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) cr[q] = cm[r0 + q], cc3[q] = cm[c0 + q];
#pragma unroll
        for (int el = 0; el < 9; ++el) {
            const int r_ = r0 + el / 3, c_ = c0 + el % 3;
            const int row = fi != fj ? c_ : r_, col = fi != fj ? r_ : c_; // diagonal blocks come in full: keep the lower half
            const double sc = cr[el / 3] * cc3[el % 3];
            if (row >= col && sc != 0.0) A[mat_at(row, col)] = -(h[el] * sc);
        }
    }
    __syncthreads();
    PV_STAMP(2, 26);
    if (d == 15) {
        if (v.dm.G_pre) {
            // IMU factor j couples frames j - 1 (local 0..14) and j (local 15..29); consecutive factors overlap on a
            // diagonal block -> odd factors, then even factors.  A thread keeps the same (a, b) entries for every factor.
            int ea[4], eb[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int el = tid + m * nthr, a = el / 30, b = el - 30 * a;
                ea[m] = (el < 900 && a >= b) ? a : -1, eb[m] = b;
            }
            // Each step is staged so that no LDS round trip waits on another: all loads (global + scales), then all
            // reads of the target entries (unconditional, in-range), then the predicated stores.
            for (int par = 0; par < 2; ++par) {
                for (int j0 = 1 + par; j0 < N; j0 += 4) { // two factors of this parity per step
                    const int j1 = j0 + 2 < N ? j0 + 2 : j0;
                    const bool on0 = flags[j0] != 0, on1 = j0 + 2 < N && flags[j1] != 0;
                    if (!on0 && !on1) continue;
                    const double *H0 = v.pre_H + (size_t)j0 * 900 + tid, *H1 = v.pre_H + (size_t)j1 * 900 + tid;
                    double h[8], sc[8], cur[8];
                    int at[8];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int o = tid + m * nthr < 900 ? m * nthr : 0;
                        h[m] = H0[o], h[4 + m] = H1[o];
                        const int am = ea[m] >= 0 ? ea[m] : 0, bm = ea[m] >= 0 ? eb[m] : 0;
                        const int r0 = 15 * (j0 - 1) + am, c0 = 15 * (j0 - 1) + bm, r1 = 15 * (j1 - 1) + am, c1 = 15 * (j1 - 1) + bm;
                        sc[m] = cm[r0] * cm[c0], sc[4 + m] = cm[r1] * cm[c1];
                        at[m] = mat_at(r0, c0), at[4 + m] = mat_at(r1, c1);
                    }
#pragma unroll
                    for (int m = 0; m < 8; ++m) cur[m] = A[at[m]];
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        PV_KEEP(h[m]);
                        PV_KEEP(cur[m]);
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (ea[m] >= 0 && on0 && sc[m] != 0.0) A[at[m]] = cur[m] - h[m] * sc[m];
                        if (ea[m] >= 0 && on1 && sc[4 + m] != 0.0) A[at[4 + m]] = cur[4 + m] - h[4 + m] * sc[4 + m];
                    }
                }
                __syncthreads();
            }
        }
        PV_STAMP(2, 27);
        if (v.dm.prior_n > 0) {
            // prior block (slot qa, slot qb): 15 x 15 entries, thread = one (a, b) of every block; only entries on or
            // below the diagonal of the reduced system contribute; same staging, kFly entries per step
            constexpr int kFly = 8;
            const int n = v.dm.prior_n, D = 15 * n;
            const int a = tid / 15, b = tid - 15 * a;
            if (tid < 225) {
                for (int qa = 0; qa < n; ++qa) {
                    const int fa = pframe[qa], ga = 15 * fa + a;
                    const double ca = cm[ga];
                    const double *Hrow = v.prior_H + (size_t)(15 * qa + a) * D + b;
                    for (int qb0 = 0; qb0 < n; qb0 += kFly) {
                        double h[kFly], sc[kFly], cur[kFly];
                        int at[kFly];
                        bool ok[kFly];
#pragma unroll
                        for (int u = 0; u < kFly; ++u) {
                            const int qb = qb0 + u < n ? qb0 + u : n - 1;
                            h[u] = Hrow[15 * qb];
                            const int gb = 15 * pframe[qb] + b;
                            ok[u] = qb0 + u < n && ga >= gb;
                            sc[u] = ca * cm[gb];
                            at[u] = ga >= gb ? mat_at(ga, gb) : mat_at(gb, ga);
                        }
#pragma unroll
                        for (int u = 0; u < kFly; ++u) cur[u] = A[at[u]];
#pragma unroll
                        for (int u = 0; u < kFly; ++u) {
                            PV_KEEP(h[u]);
                            PV_KEEP(cur[u]);
                        }
#pragma unroll
                        for (int u = 0; u < kFly; ++u)
                            if (ok[u] && sc[u] != 0.0) A[at[u]] = cur[u] - h[u] * sc[u];
                    }
                }
            }
            __syncthreads();
        }
    }
}

// The same system from the tile image k_reduce assembled (single GPU): the image already holds landmark / plane tiles +
// IMU blocks + prior, unscaled, in the layout of A, so the build is one coalesced sweep (32 bytes per lane and step)
// instead of four passes whose read-modify-writes each wait on a global round trip when A lives in HBM.
template <int NT>
__device__ __forceinline__ double dense_build_image(const View &v, double *A, const double *cm, const double *rhs_s, const double *vvec, int P, int Pp, int nbk) {
    // Four tiles of a wave are requested before the first one is used (one tile per pass was one trip to L2 / HBM per pass: 55 passes
    // per wave at P = 450), and the pose quadratic form v^T S v = sum_ik S_ik v_i v_k is taken from the values on their way out instead
    // of by a second sweep over the 1.6 MB matrix this workgroup has just written.  Returns this thread's share of it.
    const int tid = threadIdx.x;
    constexpr int nthr = NT, NW = NT / 64, kAhead = 4;
    const int ntile = (nbk * (nbk + 1)) >> 1;
    const int ln = tid & 63, lr = ln & 15, lk = ln >> 4;
    int ti = tid >> 6, bi = 0, bk = ti;
    while (bk > bi) bk -= bi + 1, ++bi;
    double q = 0.0;
    for (; ti < ntile; ti += kAhead * NW) {
        lds_d2 s01[kAhead], s23[kAhead];
        int tbi[kAhead], tbk[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int tu = ti + u * NW;
            tbi[u] = bi, tbk[u] = bk;
            const double *src = v.img + ((size_t)(tu < ntile ? tu : ti) << 8) + 4 * ln;
            s01[u] = *reinterpret_cast<const lds_d2 *>(src), s23[u] = *reinterpret_cast<const lds_d2 *>(src + 2);
            bk += NW;
            while (bk > bi) bk -= bi + 1, ++bi;
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int tu = ti + u * NW;
            if (tu >= ntile) break; // wave-uniform
            double *dst = A + ((size_t)tu << 8) + 4 * ln;
            const int k = 16 * tbk[u] + lr;
            const double ck = k < P ? cm[k] : 0.0, vk = vvec[k];
            double o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tbi[u] + lk + 4 * r;
                const double val = r < 2 ? s01[u][r & 1] : s23[u][r & 1];
                if (i == Pp) o[r] = k < P ? -rhs_s[k] : 0.0;                                   // augmented row: scaled rhs
                else if (i > Pp || k > i) o[r] = 0.0;
                else if (i == k) o[r] = (i < P && ck != 0.0) ? -(val * ck * ck) : -1.0;        // unit rows: inactive coordinates, panel padding
                else {
                    const double sc = (i < P ? cm[i] : 0.0) * ck;
                    o[r] = sc != 0.0 ? -(val * sc) : 0.0;
                }
                q -= ((i == k ? o[r] : 2.0 * o[r]) * vk) * vvec[i]; // (stored negated; v is zero past P: the augmented row and the padding add nothing)
            }
            lds_d2 w01, w23;
            w01[0] = o[0], w01[1] = o[1], w23[0] = o[2], w23[1] = o[3];
            *reinterpret_cast<lds_d2 *>(dst) = w01;
            *reinterpret_cast<lds_d2 *>(dst + 2) = w23;
        }
    }
    __syncthreads();
    return q;
}

// Trailing sweep of the generic (matrix in HBM) factorization: (-C)(16x16) += L_i (16 x W) L_k^T for every tile of the
// trailing block triangle (tile rows / columns >= b0), W / 4 v_mfma_f64_16x16x4_f64 per tile (operand layout: lane l supplies
// A[l & 15][l >> 4] and B[l >> 4][l & 15], receives D[(l >> 4) + 4 r][l & 15]).  Tiles are dealt round-robin to the four waves
// in batches of kBatch; the C tiles of the NEXT batch are requested before the current batch's operands are read from the LDS
// panel and multiplied (two batches ahead), so a batch costs max(L2 round trip, W / 4 * kBatch MFMAs), not their sum.
template <int W, int NWV>
__device__ __forceinline__ void dense_trailing_sweep(double *A, const double *Pn, int WS, int b0, int nbk, int wv, int lane) {
    constexpr int kBatch = 4, NM = W / 4;
    const int lr = lane & 15, lk = lane >> 4;
    int bi = b0, q = wv;
    while (bi < nbk && q > bi - b0) q -= bi - b0 + 1, ++bi;
    struct Batch { // one batch of C tiles on its way from L2
        double *C[kBatch];
        int bi[kBatch], bk[kBatch];
        bool valid[kBatch];
        lds_d2 c01[kBatch], c23[kBatch];
    };
    auto fetch = [&](Batch &B) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            B.valid[u] = bi < nbk;
            B.bi[u] = B.valid[u] ? bi : nbk - 1, B.bk[u] = B.valid[u] ? b0 + q : nbk - 1;
            B.C[u] = A + tile_base(B.bi[u], B.bk[u]) + 4 * lane;
            B.c01[u] = *reinterpret_cast<const lds_d2 *>(B.C[u]);
            B.c23[u] = *reinterpret_cast<const lds_d2 *>(B.C[u] + 2);
            q += NWV;
            while (bi < nbk && q > bi - b0) q -= bi - b0 + 1, ++bi;
        }
    };
    auto multiply = [&](const Batch &B) {
        mfma_d4 acc[kBatch];
        double oa[kBatch][NM], ob[kBatch][NM];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const double *pa = Pn + (16 * B.bi[u] + lr) * WS + lk, *pb = Pn + (16 * B.bk[u] + lr) * WS + lk;
#pragma unroll
            for (int m = 0; m < NM; ++m) oa[u][m] = pa[4 * m], ob[u][m] = pb[4 * m];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) acc[u][0] = B.c01[u][0], acc[u][1] = B.c01[u][1], acc[u][2] = B.c23[u][0], acc[u][3] = B.c23[u][1];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int u = 0; u < kBatch; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[u][m], ob[u][m], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
            if (B.valid[u]) {
                lds_d2 w0, w1;
                w0[0] = acc[u][0], w0[1] = acc[u][1], w1[0] = acc[u][2], w1[1] = acc[u][3];
                *reinterpret_cast<lds_d2 *>(B.C[u]) = w0;
                *reinterpret_cast<lds_d2 *>(B.C[u] + 2) = w1;
            }
    };
    // two batches ahead: a round trip to L2 is longer than one batch of MFMAs (a wave per SIMD: nobody else hides it)
    Batch B0, B1, B2;
    fetch(B0), fetch(B1);
    for (;;) {
        if (!B0.valid[0]) break;
        fetch(B2), multiply(B0);
        if (!B1.valid[0]) break;
        fetch(B0), multiply(B1);
        if (!B2.valid[0]) break;
        fetch(B1), multiply(B2);
    }
}

// slot <-> (tile column g, q) table of the register-resident factorization (see k_dense): column g has ceil((g + 1) / 4) slots
constexpr int kDenseCols = 11;
__device__ constexpr int kSlotCol[21] = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 8, 9, 9, 9, 10, 10, 10};
__device__ constexpr int kSlotQ[21] = {0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2};
__device__ constexpr int kColSlot[12] = {0, 1, 2, 3, 4, 6, 8, 10, 12, 15, 18, 21};

// q-th tile row (from the end) of wave w: 0..3 forwards, 4..7 backwards, 8..10 forwards again
__device__ __forceinline__ int dense_row_of(int w, int q) { return q == 0 ? w : (q == 1 ? 7 - w : 8 + w); }
// LOOK-AHEAD form of the register-resident factorization (template parameter LA): wave 0 owns no tiles and runs one panel AHEAD --
// it factors the 8 x 8 diagonal block of panel p + 1 and turns the rows of that panel into L while waves 1..3 apply panel p to
// the tiles; the two sides meet through two counters in LDS instead of two workgroup barriers per panel.  The tile rows
// (counted from the last one) are dealt to the three update waves 0,5,6 / 1,4,7,10 / 2,3,8,9 (19 / 18 / 18 tiles at P = 150);
// column g has max over the waves of #{rows <= g} slots: 1 1 1 2 2 2 3 3 3 4 4 = 26 slots (208 accumulator registers).
__device__ constexpr int kLaSlotCol[26] = {0, 1, 2, 3, 3, 4, 4, 5, 5, 6, 6, 6, 7, 7, 7, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10};
__device__ constexpr int kLaSlotQ[26] = {0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3};
__device__ constexpr int kLaColSlot[12] = {0, 1, 2, 3, 5, 7, 9, 12, 15, 18, 22, 26};
__device__ __forceinline__ int dense_la_row_of(int w, int q) { // w = wave index (0 = the factoring wave: no rows)
    return w == 1 ? (q == 0 ? 0 : (q == 1 ? 5 : (q == 2 ? 6 : 99)))
                  : (w == 2 ? (q == 0 ? 1 : (q == 1 ? 4 : (q == 2 ? 7 : 10))) : (w == 3 ? (q == 0 ? 2 : (q == 1 ? 3 : (q == 2 ? 8 : 9))) : 99));
}
template <bool LA> __device__ __forceinline__ constexpr int dt_slots() { return LA ? 26 : 21; }
template <bool LA> __device__ __forceinline__ constexpr int dt_nq() { return LA ? 4 : 3; }
template <bool LA> __device__ __forceinline__ constexpr int dt_slot_col(int i) { return LA ? kLaSlotCol[i] : kSlotCol[i < 21 ? i : 0]; }
template <bool LA> __device__ __forceinline__ constexpr int dt_slot_q(int i) { return LA ? kLaSlotQ[i] : kSlotQ[i < 21 ? i : 0]; }
template <bool LA> __device__ __forceinline__ constexpr int dt_col_slot(int g) { return LA ? kLaColSlot[g] : kColSlot[g]; }
template <bool LA> __device__ __forceinline__ int dt_row_of(int w, int q) { return LA ? dense_la_row_of(w, q) : (q < 3 ? dense_row_of(w, q) : 99); }
// the two LDS counters of the look-ahead form: acquire loads / release stores at workgroup scope (what has been written to LDS
// before a counter moves is visible to the wave that sees it move)
__device__ __forceinline__ int dense_wait(int *flag, int target) { // spin until the counter reaches `target` (or goes negative: failed pivot)
    int val;
    while ((val = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 0 && val < target) __builtin_amdgcn_s_sleep(1);
    return val;
}
// (called by a whole wave: lane 0 moves the counter.  The lanes of a wave run in lockstep, so everything the wave has stored before is
// covered by lane 0's release.  __builtin_amdgcn_wave_barrier() emits no instruction: it states that to the compiler -- and it is where
// the fiber emulator of tests/hipemu, which runs the lanes one after another, lets them meet.  It does change the block layout of
// k_dense<true, true> (218 lines of ISA): measured against the build without it on one box, 13 323 vs 13 275 iterations/s, same results
// (tests/micro/build_variant.py no_wave_barrier).)
__device__ __forceinline__ void dense_signal_set(int *flag, int val) {
    __builtin_amdgcn_wave_barrier();
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void dense_signal_add(int *flag) {
    __builtin_amdgcn_wave_barrier();
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// compile-time storage choice: a runtime LDS-or-global pointer select degrades every access to FLAT.  The register-resident
// form runs one wave per SIMD (its tiles live in the accumulators of exactly four waves); the HBM form runs two: its sweeps
// and passes over the matrix wait on L2 round trips that a second wave fills.
// LS: row stride, in doubles, of the two row-major LDS arrays of the register-resident factorization -- Xs (the 8 columns that are factored next) and
// Lf (the finished panels of L).  A row is 8 doubles; at LS = 8 the factor wave's lane-per-row accesses (12 ds_read_b128 + 12 ds_write_b128 per panel at
// 64-byte stride) run into 4-way bank conflicts: 75 cycles per instruction against 32 at an 80-byte stride (tools/ubench/wave_costs.hip).  LS = 10 pads
// every row to 80 bytes -- no address arithmetic, only other constants.  Round 5 built it and measured NO gain (dense_row_stride() below): LS = 8 ships,
// the padded instantiation stays reachable for the A/B.
template <bool LDSMAT, bool LA, int LS = 8>
__global__ void __launch_bounds__(LDSMAT ? kDenseThreads : 2 * kDenseThreads) k_dense(View v) {
    static_assert(LDSMAT || !LA, "the look-ahead form is a form of the register-resident factorization");
    static_assert(LS == 8 || (LDSMAT && LS == 10), "row stride of Xs / Lf: 8 doubles, or padded to 10 (80 bytes: 16-byte aligned, conflict-free)");
    // 256 threads = one wave per SIMD: the redundant 8 x 8 block factorization then costs each SIMD exactly once
    HIP_DYNAMIC_SHARED(double, lds)
    Ctrl *const cg = v.ctrl;
    const int N = v.dm.N, d = v.dm.d, P = v.dm.P, P6 = v.dm.P6, tid = threadIdx.x;
    constexpr int nthr = LDSMAT ? kDenseThreads : 2 * kDenseThreads, NW = nthr / 64;
    const size_t nS = (size_t)v.dm.n_tasks * 9;
    const double *redV = v.red + nS;
    // dynamic LDS: [header 352][8 vectors of LDV][Lp: panel, LDV x 8][A: tiles]   (no static LDS: keeps the dynamic base
    // 16-byte aligned, cdna_hip_programming.md Guideline 17).  Row Pp of A is the right-hand side, so the factorization
    // performs the forward substitution on the fly; rows above it are zero padding.
    DenseShared &sh = *reinterpret_cast<DenseShared *>(lds);
    double *red_scratch = lds + 16; // 6 * 16 doubles
    int &sh_fail = *reinterpret_cast<int *>(lds + 120);
    int *pslot = reinterpret_cast<int *>(lds + 121); // [N <= 32] frame -> prior slot (or -1), 16 doubles... see static_assert
    static_assert(kMaxFrames <= 32, "pslot overlaps the LDS header");
    const int Pp = (P + kPanel - 1) & ~(kPanel - 1); // system padded with identity rows to whole panels; the rhs is row Pp
    const int nbk = (Pp + 16) >> 4, LDV = nbk << 4;   // tile rows incl. the rhs row
    double *Dg = lds + 144; // packed lower triangle of the next 8 x 8 diagonal block (36 doubles)
    int *pvalid = reinterpret_cast<int *>(lds + 184); // [N <= 32] IMU factor j present
    int *pframe = reinterpret_cast<int *>(lds + 200); // [prior_n <= 32] frame of prior slot q
    Ctrl *const c = reinterpret_cast<Ctrl *>(lds + 224); // the control block is worked on in LDS and written back on exit
    static_assert(sizeof(Ctrl) <= 48 * sizeof(double) && sizeof(Ctrl) % sizeof(double) == 0, "LDS header layout");
    double *redS = lds + 272;     // [8] reduced scalars of the linearization
    double *aux_costs = lds + 280; // [N + prior_n <= 64] IMU factor costs (0 where absent) and prior costs
    double *vec = lds + 352;
    double *Lp = vec + 8 * (size_t)LDV;
    double *A = LDSMAT ? Lp + LS * (size_t)LDV : v.Smat;
    double *diagH = vec, *gtot = vec + LDV, *rhs = vec + 2 * LDV, *yv = vec + 3 * LDV, *vv = vec + 4 * LDV, *act = vec + 5 * LDV,
           *tmp = vec + 6 * LDV, *cpl = vec + 7 * LDV;
    double *ysol = rhs; // solution of the reduced system (rhs is dead once the augmented row has been written)
    // Tile ownership of the register-resident factorization (LDSMAT).  Tile (g, h): g = tile column, h <= g = tile row, both
    // counted from the LAST one.  Rows are dealt to the waves in a zigzag (dense_row_of: rows w, 7 - w, 8 + w -> 15 / 14 / 13 / 13
    // tiles at P = 150 instead of 21 / 18 / 15 / 12 for h & 3: the rows nearest the end are the longest); a wave keeps the
    // tile of its q-th row of column g in the
    // accumulator slot kColSlot[g] + q -- a COMPILE-TIME slot <-> (g, q) table (columns in slot order, ceil((g + 1) / 4) slots
    // per column whatever the wave: a wave with fewer rows in a column leaves a slot unused).  Only the offset nbk - 1 and
    // the wave index are runtime values, so the rank-8 update addresses accumulators and operands statically: one A operand
    // per q (3 per wave), one B operand per column, no per-slot address arithmetic, selects or slot-list branches.
    constexpr int kSlots = LDSMAT ? dt_slots<LA>() : 1; // sum over g < 11 of ceil((g + 1) / 4): LDV <= 176 (look-ahead form: 26)
    constexpr int kNQ = dt_nq<LA>();                      // tile rows a wave can own
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
    // the reduced system arrives as a tile image: loaded straight into registers (the look-ahead form is only launched on one)
    const bool from_images = LA ? true : (LDSMAT && v.dm.use_img);
    int sbi[kSlots], sbk[kSlots];
    // the accumulators of the register-resident factorization: the tile requests land in them directly (the tile image holds the
    // accumulator layout), so that a tile that needs no work -- Dims::img_scaled -- costs no instruction at all
    mfma_d4 acc[kSlots];
#define PV_LOAD_TILE(i, T)                                                       \
    do {                                                                         \
        const lds_d2 t01 = *reinterpret_cast<const lds_d2 *>(T), t23 = *reinterpret_cast<const lds_d2 *>((T) + 2); \
        acc[i][0] = t01[0], acc[i][1] = t01[1], acc[i][2] = t23[0], acc[i][3] = t23[1]; \
    } while (0)
    if constexpr (LDSMAT) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            const int g = dt_slot_col<LA>(i), h = dt_row_of<LA>(wv, dt_slot_q<LA>(i));
            const bool valid = g < nbk && h <= g;
            sbk[i] = valid ? nbk - 1 - g : -1;
            sbi[i] = valid ? nbk - 1 - h : 0;
        }
#pragma unroll
        for (int i = 0; i < kSlots; ++i) acc[i][0] = 0.0, acc[i][1] = 0.0, acc[i][2] = 0.0, acc[i][3] = 0.0;
    }
    // gradient max-norm, state / trace copies (and v^T S v: qvv_back) are finished by k_backsub (always, in the look-ahead form)
    const bool split = LA ? true : v.dm.split_fin != 0;
    // `early`: everything small this launch reads is REQUESTED in one round, every request independent of every other, and waited for
    // once; only then do the tile loads go out.  What the round-2 ISA did instead (one s_waitcnt vmcnt(0) per control input -- the
    // branches of an if-chain loading into one register; the control inputs stored to LDS, and the termination flag tested, only
    // after ALL tile loads had landed -- the per-slot branches around the tile loads leave the compiler no count of what is in
    // flight, and vmcnt retires in order) put the control section 11 600 cycles into the launch and the first panel 30 500.
    // (A property of the INSTANTIATION, not a run-time flag: a branch around the tile requests would again leave their number open.)
    constexpr bool early = LA;
    PV_STAMP_BEGIN(2);
    const long long pv_t0 = v.dbg ? clock64() : 0; // (in a register: a stamp that fetched the start time would wait for every request in flight)
    PV_STAMP2(0);
    // ---- round 1: REQUESTS ONLY.  Every load below is unconditional (clamped index, selected address: all the arrays exist for every
    // frame whatever the window holds), lands in a register of its own and is not looked at -- not even converted to a flag -- until
    // the section is over: a value used inside a branch, or an `&&` on a loaded byte, makes the compiler wait for it on the spot.
    constexpr int kCtrlWords = (int)(sizeof(Ctrl) / sizeof(double));
    const bool g_pre = v.dm.G_pre != 0;
    const size_t red_scal = nS + (size_t)kNumPoseVec * P6;
    // (a) the control inputs, one per thread at a selected address (a thread with nothing to fetch re-reads the first word)
    const double *in_ptr = reinterpret_cast<const double *>(cg);
    if (tid < kCtrlWords) in_ptr = reinterpret_cast<const double *>(cg) + tid;
    else if (tid >= 64 && tid < 64 + kNumLinScal) in_ptr = v.red + red_scal + (tid - 64);
    else if (tid >= 129 && tid < 128 + N) in_ptr = v.pre_cost + (tid - 128);
    else if (tid >= 192 && tid < 192 + v.dm.prior_n) in_ptr = v.prior_cost + (tid - 192);
    else if (tid == 255) in_ptr = v.rot_cost;
    const int jq = tid >= 129 && tid < 128 + N ? tid - 128 : 0;
    const int fq = tid < N ? tid : N - 1;
    const int a1 = tid < P ? tid : P - 1;                       // the coordinate of this thread (P <= nthr: one per thread)
    const int f1 = d == 15 ? a1 / 15 : a1 / 6, k1 = a1 - d * f1; // its frame and its index in the frame's block
    const int k6 = k1 < 6 ? k1 : 0, k3 = k1 < 3 ? k1 : 0, jB1 = f1 + 1 < N ? f1 + 1 : f1;
    static_assert(15 * kMaxFrames <= 2 * kDenseThreads, "one coordinate per thread");
    double L_in = *in_ptr;
    unsigned L_pvq = v.pre_valid[jq];
    // (b) static per-frame data of the vector assembly, the termination flag (every thread takes it from its own load: the control
    // section may set the LDS copy's `done` while slower waves are still on their way to the test)
    int L_slot = v.prior_slot[fq], L_frame = v.prior_frames[tid < v.dm.prior_n ? tid : 0]; // (both staged with at least one word)
    unsigned L_valid = v.pre_valid[fq], L_pa = v.pose_active[fq], L_ma = v.motion_active[fq];
    int was_done = cg->done;
    // (c) the unscaled vectors of coordinate a1: diag(J^T J), gradient, Schur rhs -- the landmark part from k_reduce, the IMU factors
    // on either side of the frame, the prior BY FRAME (prior_gd: zero where the prior does not reach), the rotation prior (zero
    // block when the window has none), the Jacobi scale of the coordinate (written by the solve's first factoring launch)
    double L_dg = redV[2 * P6 + 6 * f1 + k6], L_g = redV[6 * f1 + k6], L_rs = redV[P6 + 6 * f1 + k6];
    unsigned L_vA = v.pre_valid[f1], L_vB = v.pre_valid[jB1];
    double L_hA = v.pre_H[(size_t)f1 * 900 + (15 + k1) * 31], L_gA = v.pre_g[(size_t)f1 * 30 + 15 + k1];
    double L_hB = v.pre_H[(size_t)jB1 * 900 + k1 * 31], L_gB = v.pre_g[(size_t)jB1 * 30 + k1];
    double L_pg = v.prior_gd[30 * f1 + k1], L_ph = v.prior_gd[30 * f1 + 15 + k1];
    double L_rg = v.rot_g[3 * f1 + k3], L_rh = v.rot_H[9 * f1 + 4 * k3];
    unsigned L_actp = v.pose_active[f1], L_actm = v.motion_active[f1];
    double in_cp = v.cp[a1];
    if constexpr (LDSMAT) {
        // ---- round 2, requested right behind round 1: this wave's tiles of the reduced system (112 KB for the workgroup; a rejected
        // step fetches them for nothing).  One CU pulls them at ~8 bytes a cycle (14 000 cycles: bounded by the misses it can keep in
        // flight times the latency of memory the producers -- other XCDs -- wrote to), which is the longest single item in front of the
        // first panel: it has to start NOW and run under everything else.  vmcnt retires in order, so round 1 is back long before;
        // for the compiler to know that -- to wait for "all but the last 52 requests" instead of for everything -- the NUMBER of tile
        // requests must not depend on anything: a slot this wave does not own reads the all-zero tile kept behind the image.
        if constexpr (early) {
            PV_ORDER(); // round 1 is requested FIRST: requests retire in the order they were made
#pragma unroll
            for (int i = 0; i < kSlots; ++i) {
                const int ti = sbk[i] >= 0 ? ((sbi[i] * (sbi[i] + 1)) >> 1) + sbk[i] : (v.dm.img_sz >> 8); // (the tile behind the image: zeros)
                const double *T = v.img + ((size_t)ti << 8) + 4 * lane;
                PV_LOAD_TILE(i, T);
            }
        }
    }
    // every request above has been issued before any of them is looked at: the values are pinned HERE (the compiler would otherwise
    // move a load into the branch that uses it -- behind the wait for the loads that the branch condition needs: a second trip)
    PV_KEEP(L_in); PV_KEEP(L_pvq); PV_KEEP(L_slot); PV_KEEP(L_frame); PV_KEEP(L_valid); PV_KEEP(L_pa); PV_KEEP(L_ma); PV_KEEP(was_done);
    PV_KEEP(L_dg); PV_KEEP(L_g); PV_KEEP(L_rs); PV_KEEP(L_vA); PV_KEEP(L_vB); PV_KEEP(L_hA); PV_KEEP(L_gA); PV_KEEP(L_hB); PV_KEEP(L_gB);
    PV_KEEP(L_pg); PV_KEEP(L_ph); PV_KEEP(L_rg); PV_KEEP(L_rh); PV_KEEP(L_actp); PV_KEEP(L_actm); PV_KEEP(in_cp);
    PV_STAMP2(28);
    // ---- round 1: uses.  Selects, no branches on loaded values. ----
    double in_val = L_in;
    const unsigned in_pv = g_pre ? L_pvq : 0u; // IMU factor present (tid = 128 + j)
    const int in_slot = v.dm.prior_n > 0 ? L_slot : -1, in_valid = (g_pre & (L_valid != 0)) ? 1 : 0, in_frame = L_frame;
    const bool f_pose_active = (L_pa != 0) & (tid < N), f_motion_active = (L_ma != 0) & (tid < N);
    double as_dg, as_g, as_r, as_act;
    {
        double dg = k1 < 6 ? L_dg : 0.0, g = k1 < 6 ? L_g : 0.0;
        double r = g - (k1 < 6 ? L_rs : 0.0); // rhs_u = g_total - sum_l w_l W_l^T b_l
        if (d == 15) {
            // (an absent factor's block was never written: select, not multiply)
            const bool vA = g_pre & (f1 >= 1) & (L_vA != 0), vB = g_pre & (f1 + 1 < N) & (L_vB != 0);
            const double hA = vA ? L_hA : 0.0, gA = vA ? L_gA : 0.0, hB = vB ? L_hB : 0.0, gB = vB ? L_gB : 0.0;
            if (g_pre) { // (a uniform condition on a kernel argument; x + 0 is exact, the branch only keeps -0 a -0)
                if (f1 & 1) dg = (dg + hA) + hB, g = (g + gA) + gB, r = (r + gA) + gB;
                else dg = (dg + hB) + hA, g = (g + gB) + gA, r = (r + gB) + gA;
            }
            if (v.dm.prior_n > 0) dg += L_ph, g += L_pg, r += L_pg;
        }
        if (v.dm.n_rot > 0 && k1 < 3) dg += L_rh, g += L_rg, r += L_rg;
        as_act = ((k1 < 6 ? L_actp : L_actm) != 0) ? 1.0 : 0.0;
        as_dg = dg, as_g = g, as_r = r;
    }
    if (v.dm.world > 1 && tid == 64 + 4) { // sharded windows: max |b_l|, one slot per rank behind the scalars (see k_reduce); all >= 0
        in_val = 0.0;
        for (int w = 0; w < v.dm.world; ++w) in_val = fmax(in_val, v.red[red_scal + kNumLinScal + w]);
    }
    // Touch what the entry-by-entry assembly reads (not the tile-image form), one load per 128-byte line: the sources were produced on
    // other XCDs and a first touch costs a trip through the fabric -- paid once here, all lines in flight.
    double pf = 0;
    if (!from_images) {
        const size_t nR = nS + (size_t)kNumPoseVec * P6 + kNumLinScal;
#pragma unroll 4
        for (size_t e = (size_t)tid * 16; e < nR; e += (size_t)nthr * 16) pf += v.red[e];
        if (d == 15) {
            if (v.dm.G_pre) {
                const size_t nH = (size_t)N * 900, nG = (size_t)N * 30;
#pragma unroll 4
                for (size_t e = (size_t)tid * 16; e < nH; e += (size_t)nthr * 16) pf += v.pre_H[e];
                for (size_t e = (size_t)tid * 16; e < nG; e += (size_t)nthr * 16) pf += v.pre_g[e];
            }
            if (v.dm.prior_n > 0) {
                const size_t D15 = 15 * (size_t)v.dm.prior_n, nH = D15 * D15;
#pragma unroll 4
                for (size_t e = (size_t)tid * 16; e < nH; e += (size_t)nthr * 16) pf += v.prior_H[e];
                if ((size_t)tid * 16 < D15) pf += v.prior_g[(size_t)tid * 16];
            }
        }
    }
    if constexpr (LDSMAT) {
        // not `early` (split finalize off): the tile loads go out here, behind the control inputs, as in round 2
        if (split && !early) {
#pragma unroll
            for (int i = 0; i < kSlots; ++i)
                if (from_images && sbk[i] >= 0) {
                    const double *T = v.img + ((size_t)(((sbi[i] * (sbi[i] + 1)) >> 1) + sbk[i]) << 8) + 4 * lane;
                    PV_LOAD_TILE(i, T);
                }
        }
    }
    // the control inputs go to LDS (the first use of round 1: everything requested above lands within the same trip)
    {
        if (tid < kCtrlWords) reinterpret_cast<double *>(c)[tid] = in_val;
        else if (tid < 64 + kNumLinScal && tid >= 64) redS[tid - 64] = in_val;
        else if (tid >= 128) {
            // the costs of the IMU factors (wave 2) and of the prior slots + the rotation priors (wave 3), summed by their waves: the
            // control thread adds two numbers instead of walking N + prior_n + 1 LDS words one latency at a time
            double x = 0.0;
            if (tid >= 129 && tid < 128 + N) x = in_pv ? in_val : 0.0; // (select: an absent factor's cost was never written)
            else if ((tid >= 192 && tid < 192 + v.dm.prior_n) || tid == 255) x = in_val; // (rot_cost: zero block when the window has none)
            x = wave_sum(x);
            if ((tid & 63) == 0) aux_costs[tid >> 6] = x; // [2] IMU, [3] prior + rotation priors
        }
        if (tid < N) {
            pslot[tid] = in_slot, pvalid[tid] = in_valid;
            if (tid < v.dm.prior_n) pframe[tid] = in_frame;
        }
    }
    if (!split) { // (split: the passes that read these copies run in k_backsub)
        for (int e = tid; e < 32 * N; e += nthr) Lp[e] = v.fs[e];
        for (int e = tid; e < 16 * N; e += nthr) Lp[32 * N + e] = v.fs_user[e]; // the user state the finalize pass takes the old biases from
    }
    __syncthreads(); // the staged control inputs are in LDS (the tile requests stay in flight across the barrier)
    PV_STAMP2(29);
    if (was_done) return; // nothing was modified
    // ---------------- control (thread 0): Finalize the iteration in flight, decide what comes next ----------------
    if (tid == 0) {
        // on a register copy of the control block: every field read from LDS on its own is a latency of ~120 cycles in a chain of
        // dependent branches (the section took 3 100 cycles that way)
        Ctrl cc = *c;
        const int lr = cc.lin_result;
        sh.do_solve = 0, sh.do_trace = 0, sh.accepted = 0, sh.first = 0, sh.replay_first = -1, sh.replay_count = 0;
        double aux_cost = 0;
        if (lr != LIN_INVALID_STEP) {
            aux_cost = aux_costs[2] + aux_costs[3]; // IMU factors; prior slots + rotation priors (summed by waves 2 and 3)
        }
        const double lm_cost = redS[0], lm_bad = redS[5];
        double total_cost = aux_cost + lm_cost;
        const bool finite_ok = (lm_bad == 0.0) && isfinite(total_cost);
        bool finalize = false; // run FinalizeIterationAndCheckIfMinimizerCanContinue
        if (lr == LIN_INIT) {
            if (!finite_ok) {
                cc.termination = 2, cc.done = 1; // FAILURE: initial evaluation failed
            } else {
                cc.x_cost = total_cost, cc.initial_cost = total_cost;
                cc.x_norm2_pose = cc.cand_norm2_pose, cc.x_norm2_lm = redS[3];
                cc.lm_g2 = redS[1];
                cc.it_cost = total_cost, cc.it_cost_change = 0, cc.it_step_norm = 0, cc.it_rel = 0, cc.it_valid = 1, cc.it_success = 1;
                sh.first = 1, sh.accepted = 1;
                finalize = true;
            }
        } else if (lr == LIN_CANDIDATE) {
            const double cand_cost = finite_ok ? total_cost : DBL_MAX;
            const double step_norm = sqrt(cc.cand_step2_pose + redS[2]);
            const double x_norm = sqrt(cc.x_norm2_pose + cc.x_norm2_lm);
            cc.it_valid = 1, cc.it_step_norm = step_norm;
            cc.invalid_steps = 0;
            if (step_norm <= 1e-8 * (x_norm + 1e-8)) { // ParameterToleranceReached
                cc.termination = 0, cc.done = 1;
            } else {
                const double cost_change = cc.x_cost - cand_cost;
                cc.it_cost_change = cost_change;
                if (fabs(cost_change) <= 1e-6 * cc.x_cost) { // FunctionToleranceReached (candidate NOT applied)
                    cc.termination = 0, cc.done = 1;
                } else {
                    const double rel = cost_change / cc.model_cost_change;
                    cc.it_rel = rel;
                    if (rel > 1e-3) { // HandleSuccessfulStep
                        cc.cur = 1 - cc.cur, cc.lin = 1 - cc.lin;
                        cc.x_cost = cand_cost;
                        cc.x_norm2_pose = cc.cand_norm2_pose, cc.x_norm2_lm = redS[3];
                        cc.lm_g2 = redS[1];
                        if (rel < 0.25) cc.radius *= 0.5; // DoglegStrategy::StepAccepted
                        if (rel > 0.75) cc.radius = fmax(cc.radius, 3.0 * cc.dogleg_step_norm);
                        cc.mu = fmax(1e-8, 2.0 * cc.mu / 10.0);
                        cc.reuse = 0;
                        cc.it_success = 1, cc.it_cost = cand_cost;
                        sh.accepted = 1;
                    } else { // HandleUnsuccessfulStep
                        cc.radius *= 0.5; // DoglegStrategy::StepRejected
                        cc.reuse = 1;
                        cc.it_success = 0, cc.it_cost = cand_cost;
                    }
                    finalize = true;
                }
            }
        } else if (lr == LIN_INVALID_STEP) { // HandleInvalidStep
            if (cc.dbg_invalid_left > 0) cc.dbg_invalid_left--;
            cc.it_valid = 0, cc.it_success = 0, cc.it_cost = cc.x_cost, cc.it_cost_change = 0, cc.it_step_norm = 0, cc.it_rel = 0;
            if (++cc.invalid_steps >= 5) {
                cc.termination = 2, cc.done = 1;
            } else {
                cc.mu *= 10.0; // DoglegStrategy::StepIsInvalid
                cc.reuse = 0;
                finalize = true;
            }
        } else if (lr == LIN_RELIN) {
            // the accepted point re-linearized with a new mu: continue the iteration in flight
            sh.do_solve = finite_ok ? 1 : 0;
            if (!finite_ok) cc.termination = 2, cc.done = 1;
        }
        sh.x_cost_new = cc.x_cost;
        sh.do_trace = 0;
        if (finalize) sh.do_trace = 1; // the record needs grad_max of an accepted linearization -> written after the build below
        *c = cc;
    }
    __syncthreads();
    if (c->done && !sh.do_trace) {
        if (tid == 0) c->mode = MODE_DONE, *cg = *c;
        return;
    }
    if (v.dbg && tid == 0) sh.stamp1 = clock64() - pv_t0;
    if (pf == 1.2345678901234567e301) v.vstep[0] = pf; // keeps the prefetch loads alive (never true for finite data)
    const bool need_build = sh.accepted || sh.do_solve; // a new accepted linearization (or RELIN) is in `red`
    if constexpr (LDSMAT) {
        // The loads of this wave's tiles are issued here, by the launches that go on to factor (a rejected step reuses the last
        // Gauss-Newton step: three launches in ten at 10 x 1000 would pull the 112 KB image through this CU for nothing): their
        // latency -- a trip through the fabric, k_reduce ran on other XCDs -- is covered by the vector assembly, the finalize
        // pass and the scaling below.
        if (need_build && !split) {
#pragma unroll
            for (int i = 0; i < kSlots; ++i)
                if (from_images && sbk[i] >= 0) {
                    const double *T = v.img + ((size_t)(((sbi[i] * (sbi[i] + 1)) >> 1) + sbk[i]) << 8) + 4 * lane;
                    PV_LOAD_TILE(i, T);
                }
        }
    }
    // ---------------- unscaled vectors, part 2 ----------------
    if (need_build && split) {
        // every thread only ever reads back the entries it writes here (the scaling loop below walks a = tid): no barrier.  The
        // gradient max-norm of an accepted linearization is formed by k_backsub's finalize workgroup from the copy in HBM.
        if (tid < P) {
            diagH[tid] = as_dg, gtot[tid] = as_g, rhs[tid] = as_r, act[tid] = as_act;
            if (sh.accepted) v.gtot[tid] = as_g;
        }
    } else if (need_build) {
        if (tid < P) diagH[tid] = as_dg, gtot[tid] = as_g, rhs[tid] = as_r, act[tid] = as_act;
        __syncthreads();
        // gradient_max_norm = max | x - (x (+) -g) | over the free blocks (ambient coordinates)
        double gm = 0;
        if (tid < N) {
            const double *x = Lp + ((size_t)c->cur * N + tid) * 16; // staged copy of v.fs
            if (f_pose_active) {
                double ng[6], y[7];
                for (int k = 0; k < 6; ++k) ng[k] = -gtot[d * tid + k];
                pose_plus(y, x, ng, ng + 3);
                for (int k = 0; k < 7; ++k) gm = fmax(gm, fabs(x[k] - y[k]));
            }
            if (d == 15 && f_motion_active)
                for (int k = 0; k < 9; ++k) gm = fmax(gm, fabs(gtot[15 * tid + 6 + k]));
        }
        if (tid < 64) {
            gm = wave_max(gm);
            if (tid == 0) c->grad_max = fmax(gm, redS[4]);
        }
        __syncthreads();
    }
    if (v.dbg && tid == 0) sh.stamp2 = clock64() - pv_t0;
    // `early`: the scaling of the vectors (below, dense_scale_vectors) runs HERE, next to the Finalize section instead of behind it:
    // it reads the control block's mu / scaling_ready, which Finalize does not touch, and this thread's own entries of the unscaled
    // vectors; if Finalize ends the solve or finds nothing to factor its LDS results are simply not used.  Finalize runs on thread
    // 192 then -- the first lane of the one wave that has no coordinate to scale (LDV <= 176) -- so that the two overlap.
    const int fin_tid = early ? 192 : 0;
    const bool first_scaling = !c->scaling_ready;
    const double mu = c->mu;
    double keep_Da = 0, keep_gh = 0, keep_cp = 0; // dogleg diagonal, scaled gradient, Jacobi scale of coordinate a = tid (LDV <= nthr)
    auto dense_scale_vectors = [&]() {
        for (int a = tid; a < LDV; a += nthr) {
            if (a >= P) {
                vv[a] = 0.0, cpl[a] = 0.0, tmp[a] = 0.0, yv[a] = 0.0; // padding (act[] is only read below P)
                continue;
            }
            double cpa;
            if (first_scaling) cpa = act[a] != 0.0 ? 1.0 / (1.0 + sqrt(diagH[a])) : 1.0; // jacobi_scaling = 1 / (1 + sqrt(col norm^2)), once
            else cpa = early ? in_cp : v.cp[a];                                          // (early: requested in round 1; a == tid)
            keep_cp = cpa;
            cpl[a] = act[a] != 0.0 ? cpa : 0.0; // LDS copy: 0 marks an inactive coordinate
            const double d2 = cpa * cpa * diagH[a];
            const double Da = sqrt(fmin(fmax(d2, 1e-6), 1e32));
            const double gh = act[a] != 0.0 ? cpa * gtot[a] / Da : 0.0;
            keep_Da = Da, keep_gh = gh;
            vv[a] = gh / Da;                                  // v = g^ / D
            tmp[a] = Da;
            yv[a] = act[a] != 0.0 ? cpa * rhs[a] : 0.0;       // scaled reduced rhs -> augmented row P
        }
        if (from_images) {
            for (int a = tid; a < LDV; a += nthr) {
                lds_d2 pr;
                pr[0] = cpl[a], pr[1] = vv[a]; // {scale, v} pairs of the coordinates: one 16-byte read per row / column at load
                *reinterpret_cast<lds_d2 *>(A + 2 * a) = pr;
                diagH[a] = yv[a];               // keep the scaled rhs (yv is reused by the back substitution)
                // diagonal patch of the assembled system: 1 on inactive coordinates and on the panel padding, mu D^2 elsewhere
                yv[a] = a < P ? (cpl[a] == 0.0 ? 1.0 : mu * tmp[a] * tmp[a]) : (a < Pp ? 1.0 : 0.0);
                if (LA && a == 3 && c->dbg_fail_left > kDbgNegativePivot) yv[a] = -1e300; // fault injection (tests only): the fourth pivot of panel 0 turns hugely negative
            }
        }
    };
    if (early && need_build) dense_scale_vectors();
    // ---------------- Finalize: record, state-updating callback, termination tests ----------------
    if (tid == fin_tid && sh.do_trace) {
        // IN PLACE (a reference): with a register copy here AS WELL AS in the control section the first factorization of every solve
        // fails on the GPU -- the scaled system comes out wrong -- while either copy alone, and both in the emulator, are right
        // (profiles/r3_kdense_ctrl_copy.txt; reproducer: tests/micro/build_variant.py fin_copy).  Not understood; Finalize runs next
        // to the scaling in the look-ahead form, so its LDS latencies are off the critical path anyway.
        Ctrl &cc = *c;
        const int lr = cc.lin_result;
        if (cc.it_success) cc.num_success++;
        sh.trace_slot = record_trace(v, &cc, cc.iter); // (split + accepted: gradient_max_norm is patched in by k_backsub's finalize workgroup)
        if (split) {
            cc.fin_flags = kFinTrace | (sh.accepted ? kFinAccepted : 0) | (sh.first ? kFinFirst : 0);
            cc.fin_trace_slot = sh.trace_slot;
            cc.fin_lm_gmax = redS[4];
        }
        bool stop = false;
        if (cc.iter >= v.dm.max_iter) cc.termination = 1, stop = true;                           // MaxSolverIterationsReached
        // (split: the gradient of a linearization accepted in THIS launch is not known here -- the finalize workgroup applies the test
        // and takes the iteration back; a rejected step keeps the old gradient, whose test did not fire when it was accepted)
        else if (!(split && sh.accepted) && cc.it_success && cc.grad_max <= 1e-10) cc.termination = 0, stop = true; // GradientToleranceReached
        else if (cc.radius <= 1e-32) cc.termination = 0, stop = true;                            // MinTrustRegionRadiusReached
        if (stop) {
            cc.done = 1;
        } else {
            cc.iter++;
            cc.it_valid = 0, cc.it_success = 0, cc.it_cost_change = 0, cc.it_step_norm = 0, cc.it_rel = 0, cc.it_cost = cc.x_cost;
            if (lr == LIN_INIT || (lr == LIN_CANDIDATE && sh.accepted)) sh.do_solve = 1;      // new Gauss-Newton step needed
            else if (lr == LIN_CANDIDATE) cc.mode = MODE_CANDIDATE, cc.solve_ok = 0;          // rejected: reuse gn / gradient
            else if (lr == LIN_INVALID_STEP) cc.mode = MODE_RELIN, cc.solve_ok = 0, cc.retry_relin = 0;
        }
    }
    __syncthreads();
    // trace states + StateUpdatingCallback (update_state_every_iteration): user state <- accepted iterate
    if (sh.do_trace && !split) {
        const int cur = c->cur;
        if (sh.accepted && !sh.first) {
            // the accepted linearization was evaluated with the OLD user biases: keep them for a later RELIN
            for (int e = tid; e < N * 6; e += nthr) v.bias0_lin[e] = Lp[32 * N + (e / 6) * 16 + 10 + e % 6]; // staged v.fs_user
        }
        __syncthreads();
        if (sh.accepted)
            for (int e = tid; e < N * 16; e += nthr) v.fs_user[e] = Lp[(size_t)cur * N * 16 + e]; // staged v.fs
        if (sh.first)
            for (int e = tid; e < N * 6; e += nthr) v.bias0_lin[e] = Lp[(size_t)cur * N * 16 + (e / 6) * 16 + 10 + e % 6];
        if (v.trace_states && sh.trace_slot >= 0) {
            double *dst = v.trace_states + (size_t)sh.trace_slot * (N * 16 + v.dm.M);
            for (int e = tid; e < N * 16; e += nthr) dst[e] = Lp[(size_t)cur * N * 16 + e];
            for (int e = tid; e < v.dm.M; e += nthr) dst[N * 16 + e] = v.rho[(size_t)cur * v.dm.M + e];
        }
    }
    __syncthreads();
    if (c->done) {
        if (tid == 0) c->mode = MODE_DONE, *cg = *c;
        return;
    }
    if (!sh.do_solve) { // a rejected or invalid step: nothing to factor (one store per thread, see the end of the kernel)
        if (tid < (int)(sizeof(Ctrl) / sizeof(double))) reinterpret_cast<double *>(cg)[tid] = reinterpret_cast<const double *>(c)[tid];
        return;
    }

    PV_STAMP2(3);
    if (v.dbg && tid == 0 && (v.dbg_sel < 0 || v.dbg_sel == 1 || v.dbg_sel == 2)) v.dbg[2 * 32 + 1] = sh.stamp1, v.dbg[2 * 32 + 2] = sh.stamp2;
    // ---------------- Jacobi scaling (once), dogleg diagonal, scaled system ----------------
    if (!early) dense_scale_vectors();
    if (first_scaling && tid < P) v.cp[tid] = keep_cp, v.cpl[tid] = cpl[tid]; // (one coordinate per thread: P <= nthr; cpl: own entry)
    if (tid < P) v.vraw[tid] = vv[tid];                                       // v = g^ / D (k_backsub: v^T S v from the scaled image)
    const bool img_scaled = LA && v.dm.img_scaled && !first_scaling;         // what k_reduce wrote in this slot (it read the same flag)
    if (tid == 0) c->img_scaled_now = img_scaled ? 1 : 0;
    if (from_images) {
        if (!early) __syncthreads(); // (early: the barriers behind Finalize already separate the scaling from its readers)
        PV_STAMP2(21);
    } else {
        for (int e = tid; e < LS * LDV; e += nthr) Lp[e] = 0.0;
        __syncthreads();
        PV_STAMP2(21);
        const bool fused_qvv = !LDSMAT && v.dm.use_img; // (uniform) the build from the image forms v^T S v on the way
        double q_build = 0.0;
        if (fused_qvv) q_build = dense_build_image<nthr>(v, A, cpl, yv, vv, P, Pp, nbk);
        else dense_build<nthr>(v, A, cpl, yv, pvalid, pframe, P, Pp, nbk);
        PV_STAMP2(22);
        // pose quadratic form with the Schur-reduced scaled matrix (mu D^2 not yet added): v^T S v = sum S_ik v_i v_k
        // (thread = one (row, column) of every tile; the vector S v itself is not needed: v^T S y' follows from the solve)
        if (fused_qvv) {
            double s1[1] = {q_build};
            block_sum<1>(s1, red_scratch);
            if (tid == 0) c->pose_qvv = s1[0];
        } else {
            const int r = (tid & 255) >> 4, cc = tid & 15, off = tile_off(r, cc);
            double q = 0;
            for (int bi = tid >> 8; bi < nbk; bi += nthr / 256) { // 256 threads cover a tile; a second half takes every other tile row
                const int i = 16 * bi + r;
                const double vi = vv[i];
                double qr = 0;
                for (int bk = 0; bk <= bi; ++bk) {
                    const int k = 16 * bk + cc;
                    const double m = A[tile_base(bi, bk) + off]; // stored negated; strictly upper entries of diagonal tiles are 0
                    qr -= (i == k ? m : 2.0 * m) * vv[k];
                }
                q += qr * vi;
            }
            double s1[1] = {q};
            block_sum<1>(s1, red_scratch);
            if (tid == 0) c->pose_qvv = s1[0];
        }
        for (int a = tid; a < P; a += nthr) diagH[a] = yv[a]; // keep the scaled rhs (yv is reused by the back substitution)
        PV_STAMP2(23);
        for (int a = tid; a < P; a += nthr)
            if (act[a] != 0.0) A[mat_at(a, a)] -= mu * tmp[a] * tmp[a];
        __syncthreads();
        if (tid < 36) { // first diagonal block, packed
            int r = 0;
            while (((r + 1) * (r + 2)) >> 1 <= tid) ++r;
            Dg[tid] = -A[mat_at(r, tid - ((r * (r + 1)) >> 1))];
        }
        __syncthreads();
    }
    PV_STAMP2(4);
    int fail = 0;
    if constexpr (LDSMAT) {
        // ---------------- register-resident panel Cholesky (width 8), two barriers per panel ----------------
        // The trailing matrix never returns to LDS: wave w owns tiles w, w + 4, ... of the block triangle (enumerated by
        // DEscending tile column, so that the tiles still alive at any panel are a prefix of every wave's list) and keeps
        // them in accumulator registers for the whole factorization.  Per panel the LDS carries only (1) the 8 columns that
        // are factored next (Xs, published by the owners of that tile column), (2) the finished panel of L (Lf, written
        // once by the row owners as the (k, k + 4) operand pairs the MFMAs read, and kept: it is also the L the back
        // substitution uses).  Lf overlays the tile image the gather produced, which is dead once the tiles are loaded.
        // Every thread factors the 8 x 8 diagonal block redundantly in registers; the owner of row i turns its 8 panel
        // entries into L (row Pp = right-hand side -> forward substitution for free).  The system is padded with identity
        // rows to whole panels, so nothing in the loop depends on a partial panel.
        double *Xs = Lp, *Lf = A;
        if (!from_images) {
#pragma unroll
            for (int i = 0; i < kSlots; ++i) {
                acc[i][0] = 0, acc[i][1] = 0, acc[i][2] = 0, acc[i][3] = 0;
                if (sbk[i] >= 0) {
                    const double *T = A + tile_base(sbi[i], sbk[i]) + 4 * lane;
                    const lds_d2 c01 = *reinterpret_cast<const lds_d2 *>(T), c23 = *reinterpret_cast<const lds_d2 *>(T + 2);
                    acc[i][0] = c01[0], acc[i][1] = c01[1], acc[i][2] = c23[0], acc[i][3] = c23[1];
                }
            }
        } else {
            // S_s = C H C from the tile image k_reduce assembled (H = H_pp - sum_l w_l W_l W_l^T + IMU + prior), with unit
            // rows for inactive coordinates, identity in the panel padding and the scaled rhs in row Pp; v^T S v and the
            // mu D^2 diagonal on the way.  Off-diagonal tiles of real rows take the short path (C is 0 on inactive rows).
            const double *cv = A, *dgv = yv;
            const int brow = Pp >> 4; // tile row of the rhs row
            double q = 0;
            if (v.dbg) { // profiling: when the last tile request is back (requests retire in order)
                double last = acc[kSlots - 1][3];
                PV_KEEP(last);
                PV_STAMP2T(18, LA ? 64 : 0);
            }
            // the {scale, v} pairs: a tile needs the pair of its column (one per lane) and of its four rows.  With the static
            // slot table they depend on the tile column g and on the wave's row index qq only: 11 + 3 * 4 reads for all 21
            // slots instead of five per slot (rows / columns outside the system read tile row / column 0: in range, never used)
            // (the diagonal patch and the scaled rhs of a tile column are read here as well, once per column: inside the slot loop every
            // one of them was an LDS latency of its own -- 52 reads, each waited for on the spot: half of the 12 500 cycles this section took)
            lds_d2 colop[kDenseCols], rowop[kNQ][4];
            double dgcol[kDenseCols], rhcol[kDenseCols];
#pragma unroll
            for (int g = 0; g < kDenseCols; ++g) {
                const int bk = g < nbk ? nbk - 1 - g : 0;
                colop[g] = *reinterpret_cast<const lds_d2 *>(cv + 2 * (16 * bk + lr));
                dgcol[g] = dgv[16 * bk + lr];
                rhcol[g] = diagH[16 * bk + lr];
            }
#pragma unroll
            for (int qq = 0; qq < kNQ; ++qq) {
                const int h = dt_row_of<LA>(wv, qq), bi = h < nbk ? nbk - 1 - h : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) rowop[qq][r] = *reinterpret_cast<const lds_d2 *>(cv + 2 * (16 * bi + lk + 4 * r));
            }
            if (img_scaled) {
                // The image IS -(C S C) (k_reduce scaled and negated it, Dims::img_scaled) and the requests delivered it into the
                // accumulators: nothing to do here.  What the image cannot hold -- the mu D^2 / unit diagonal and the scaled rhs in row
                // Pp, both of which this launch has only just formed -- is added by the factor wave when it takes a panel's columns out
                // of Xs (`late_patch` below): 8 diagonal entries and one row per panel, from the LDS vectors, instead of a walk over all
                // 26 accumulator slots (11 500 cycles of execute-once code: instruction fetch, not arithmetic).
            } else
#pragma unroll
            for (int i = 0; i < kSlots; ++i) {
                if (sbk[i] >= 0) {
                    const int bi = sbi[i], bk = sbk[i];
                    const lds_d2 ck = colop[dt_slot_col<LA>(i)];
                    // C is 0 on inactive and padding coordinates, the image is 0 above the diagonal and outside
                    // the real rows: one formula for every entry, the structural entries are patched below
                    double val[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                    const double vk2 = 2.0 * ck[1];
                    if (v.dm.qvv_back) { // v^T S v comes from k_backsub (same image, C v from HBM): scaling only
#pragma unroll
                        for (int r = 0; r < 4; ++r) val[r] *= rowop[dt_slot_q<LA>(i)][r][0] * ck[0];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const lds_d2 ci = rowop[dt_slot_q<LA>(i)][r];
                            val[r] *= ci[0] * ck[0];
                            q += val[r] * (ci[1] * vk2);
                        }
                    }
                    { // diagonal tile: its diagonal counts once in v^T S v; unit / mu D^2 diagonal.  Selects: no branch, no read.
                        const bool isdiag = bi == bk;
                        const double dgk = dgcol[dt_slot_col<LA>(i)];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool on = isdiag & (lk + 4 * r == lr);
                            if (!v.dm.qvv_back) q -= on ? val[r] * (rowop[dt_slot_q<LA>(i)][r][1] * ck[1]) : 0.0;
                            val[r] = on ? val[r] + dgk : val[r];
                        }
                    }
                    { // the scaled rhs in row Pp
                        const bool isrhs = bi == brow;
                        const double rk = rhcol[dt_slot_col<LA>(i)];
#pragma unroll
                        for (int r = 0; r < 4; ++r) val[r] = (isrhs & (16 * bi + lk + 4 * r == Pp)) ? rk : val[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][r] = -val[r];
                }
            }
            PV_STAMPV2(22, q);
            if (LA) PV_STAMP2T(19, 64);
            if (v.dm.qvv_back) {
                if (tid == 0) c->pose_qvv = 0.0; // the whole v^T S v arrives through back_part[.][6]
            } else {
                double s1[1] = {q};
                block_sum<1>(s1, red_scratch);
                if (tid == 0) c->pose_qvv = s1[0];
            }
            PV_STAMP2(23);
        }
        // columns [o2, o2 + 8) of slot i's tile -> Xs (row-major, 8 per row; still negated)
#define PV_PUBLISH_ROW(i, brow, o2)                                                                     \
    do {                                                                                                \
        const int cX = lr - (o2);                                                                       \
        if (cX >= 0 && cX < kPanel) {                                                                   \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) Xs[(16 * (brow) + lk + 4 * r) * LS + cX] = acc[i][r]; \
        }                                                                                               \
    } while (0)
#define PV_PUBLISH(i, o2) PV_PUBLISH_ROW(i, sbi[i], o2)
#pragma unroll
        for (int i = 0; i < kSlots; ++i)
            if (sbk[i] == 0) PV_PUBLISH(i, 0);
        int *const flag_L = reinterpret_cast<int *>(lds + 216), *const flag_pub = flag_L + 1; // look-ahead form: see below
        if (LA && tid == 0) *flag_L = 0, *flag_pub = 3, sh_fail = 0; // (panel 0 has just been published by the three update waves)
        if (LA && tid == 0 && c->dbg_fail_left > kDbgNegativePivot) // fault injection (tests only): this factorization has been given a bad pivot (dense_scale_vectors)
            c->dbg_fail_left = c->dbg_fail_left - 1 == kDbgNegativePivot ? 0 : c->dbg_fail_left - 1;
        __syncthreads(); // every wave holds its tiles: the tile image may be overwritten from here on
        int lfo = 0;     // offset of the current panel in Lf (panel p keeps rows j0 .. LDV - 1, 8 doubles each)
        if constexpr (LA) {
            // ---------------- look-ahead form ----------------
            // Hand-overs: kDenseLaBarrier (round 5, default) -- two s_barrier per panel, A(p) "L rows of panel p stored" and B(p) "columns of panel p + 1
            // published"; wave 0 executes [B(p - 1)] work A(p), waves 1..3 execute A(p) next-column-update publish B(p) rest-of-the-update: the same order of
            // events as with the counters below, which remain as the other form:
            // flag_pub counts the 8-column blocks published into Xs, one count per update wave and block (3 per panel);
            // flag_L counts the panels whose L rows are complete in Lf (negative: a non-positive pivot, everybody leaves).
            //   wave 0, panel p:   wait flag_pub >= 3 (p + 1) -> diagonal block + all rows of the panel from Xs -> L rows to Lf(p)
            //                      -> flag_L = p + 1
            //   waves 1..3, panel p: wait flag_L >= p + 1 -> operands from Lf(p) -> rank-8 update of the next tile column ->
            //                      publish its 8 columns into Xs -> flag_pub += 1 -> rank-8 update of the other columns
            // Xs is single-buffered: the update waves overwrite it only after flag_L = p + 1, i.e. after wave 0 has read every
            // row of panel p; Lf(p) and Lf(p + 1) are different regions.
            const bool late_patch = img_scaled; // (uniform)
            if (wv == 0) {
                int pidx = 0;
                // this lane's row of each pass, fixed for the whole factorization (rows counted from the END: row LDV - 1 - lane - 64 t).  A row above the
                // panel (finished, or not a row at all) is read where it lies -- stale columns, valid memory -- and its result is never stored.
                constexpr int kPass = 3;
                const double *xrow[kPass];
#pragma unroll
                for (int t = 0; t < kPass; ++t) {
                    const int irow = LDV - 1 - lane - 64 * t;
                    xrow[t] = Xs + LS * (irow > 0 ? irow : 0);
                }
                for (int j0 = 0; j0 < Pp; j0 += kPanel, ++pidx) {
                    if (j0 == 0) PV_LOOP_STAMP2(8);
                    if (j0 == 80) PV_LOOP_STAMP2(13);
                    // late_patch: this panel's diagonal patch and its piece of the rhs row are requested from LDS BEFORE the wait for the
                    // update waves (they do not depend on them): the latency disappears in the wait
                    lds_d2 dg2[4], rh2[4];
                    if (late_patch) {
#pragma unroll
                        for (int h = 0; h < 4; ++h) dg2[h] = *reinterpret_cast<const lds_d2 *>(yv + j0 + 2 * h), rh2[h] = *reinterpret_cast<const lds_d2 *>(diagH + j0 + 2 * h);
                        PV_ORDER(); // (the requests stay in front of the wait)
                    }
                    if (kDenseLaBarrier) {
                        if (pidx > 0) __syncthreads(); // barrier B(p - 1): the update waves have published this panel's columns
                    } else {
                        dense_wait(flag_pub, 3 * (pidx + 1));
                    }
                    double Ld[kPanel][kPanel], inv[kPanel];
#pragma unroll
                    for (int r = 0; r < kPanel; ++r)
#pragma unroll
                        for (int h = 0; h <= (r >> 1); ++h) {
                            const lds_d2 g2 = *reinterpret_cast<const lds_d2 *>(Xs + (j0 + r) * LS + 2 * h);
                            Ld[r][2 * h] = -g2[0];
                            if (2 * h + 1 <= r) Ld[r][2 * h + 1] = -g2[1];
                        }
                    // this wave owns EVERY row of the panel: rows j0 + lane + 64 t, t < 3 (LDV <= 176), carried through the pivot loop
                    // (always three passes, rows past the end re-read the last one: skipping the empty passes was tried twice -- a uniform
                    // branch per pass inside the pivot loop, and one straight-line copy of the body per pass count -- and both were
                    // slower than carrying the dead rows: 12.0k / 12.1k against 12.6k iterations/s, profiles/r3_ab_lookahead.txt)
                    double x[kPass][kPanel];
#pragma unroll
                    for (int t = 0; t < kPass; ++t) {
                        // (rows from the end: the rhs row Pp (LDV - 16 or LDV - 8) sits in the same lane of pass 0 in EVERY panel)
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const lds_d2 g2 = *reinterpret_cast<const lds_d2 *>(xrow[t] + 2 * h);
                            x[t][2 * h] = -g2[0], x[t][2 * h + 1] = -g2[1];
                        }
                    }
                    if (late_patch) {
                        // the accumulators started from -(C S C) alone: the diagonal patch (mu D^2, or 1 on an inactive / padding
                        // coordinate) and the scaled rhs of row Pp join here, where the panel's columns leave Xs (a sum is a sum:
                        // the rank-8 updates that came first do not care)
#pragma unroll
                        for (int h = 0; h < 4; ++h) Ld[2 * h][2 * h] += dg2[h][0], Ld[2 * h + 1][2 * h + 1] += dg2[h][1];
                        const double is_rhs = lane == LDV - 1 - Pp ? 1.0 : 0.0; // row Pp: one lane of pass 0 (the rows are dealt from the end)
#pragma unroll
                        for (int h = 0; h < 4; ++h) x[0][2 * h] += is_rhs * rh2[h][0], x[0][2 * h + 1] += is_rhs * rh2[h][1];
                    }
                    double Ls[kPanel][kPanel];
#pragma unroll
                    for (int cc = 0; cc < kPanel; ++cc) {
                        const double dd = Ld[cc][cc];
                        if (!kDenseFailAtEnd) fail |= (!(dd > 0.0) || !isfinite(dd)) ? 1 : 0;
                        inv[cc] = fast_rsqrt(dd);
                        const double inv2 = inv[cc] * inv[cc];
#pragma unroll
                        for (int r = cc + 1; r < kPanel; ++r) Ls[r][cc] = Ld[r][cc] * inv2;
#pragma unroll
                        for (int r = cc + 1; r < kPanel; ++r)
#pragma unroll
                            for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ls[c2][cc];
#pragma unroll
                        for (int t = 0; t < kPass; ++t)
#pragma unroll
                            for (int c2 = cc + 1; c2 < kPanel; ++c2) x[t][c2] -= x[t][cc] * Ls[c2][cc];
                    }
                    // A pivot that is negative, zero, infinite or NaN makes fast_rsqrt return NaN (negative / NaN: v_rsq_f64 does; zero / infinite: the
                    // correction step forms 0 x inf), and a NaN multiplier reaches every entry of the block that is still to be eliminated, the last
                    // reciprocal pivot included.  A tiny positive (denormal-range) pivot does not: its reciprocal square root overflows to +inf
                    //  -- as an earlier pivot it turns the rest of the block into inf - inf = NaN, as the panel's LAST pivot it is only
                    // visible as an infinite inv[7].  One test per panel catches both: inv[7] * 0 is 0 exactly when inv[7] is finite.
                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;
                    if (j0 == 0) PV_LOOP_STAMP2(9);
                    if (j0 == 80) PV_LOOP_STAMP2(14);
                    if (fail) { // uniform
                        if (lane == 0) sh_fail = 1;
                        if (!kDenseLaBarrier) {
                            dense_signal_set(flag_L, -1);
                            break;
                        }
                        // (barrier form: every wave meets every barrier, so the loop is walked to its end on NaNs; the result is discarded behind it)
                    }
                    if (lane == 0) { // 1 / L_jj for the back substitution: every lane holds all eight, one writes them (four 16-byte writes instead of a select chain)
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            lds_d2 pr;
                            pr[0] = inv[2 * h], pr[1] = inv[2 * h + 1];
                            *reinterpret_cast<lds_d2 *>(tmp + j0 + 2 * h) = pr;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < kPass; ++t) {
                        const int irow = LDV - 1 - lane - 64 * t;
                        if (irow >= j0) {
#pragma unroll
                            for (int cc = 0; cc < kPanel; ++cc) x[t][cc] = kDenseMaskUpper ? ((j0 + cc <= irow) ? x[t][cc] * inv[cc] : 0.0) : x[t][cc] * inv[cc];
                            lds_d2 *Lrow = reinterpret_cast<lds_d2 *>(Lf + lfo + LS * (irow - j0));
#pragma unroll
                            for (int h = 0; h < 4; ++h) {
                                lds_d2 pr;
                                pr[0] = x[t][h], pr[1] = x[t][h + 4]; // operand pair (k, k + 4) of the two MFMAs
                                Lrow[h] = pr;
                            }
                        }
                    }
                    if (j0 == 0) PV_LOOP_STAMP2(10);
                    if (j0 == 80) PV_LOOP_STAMP2(15);
                    if (kDenseLaBarrier) __syncthreads(); // barrier A(p): the panel's L rows are in LDS
                    else dense_signal_set(flag_L, pidx + 1); // (release: the rows above are in LDS before the counter moves)
                    if (j0 == 0) { PV_LOOP_STAMP2(11); PV_LOOP_STAMP2(12); }
                    if (j0 == 80) { PV_LOOP_STAMP2(16); PV_LOOP_STAMP2(17); }
                    lfo += LS * (LDV - j0);
                }
                if (kDenseLaBarrier) __syncthreads(); // barrier B(last panel): the update waves' last one has a partner
            } else {
                int pidx = 0;
                for (int j0 = 0; j0 < Pp; j0 += kPanel, ++pidx) {
                    const int k0 = j0 + kPanel, b0 = k0 >> 4, o2 = k0 & 15;
                    if (kDenseLaBarrier) __syncthreads(); // barrier A(p)
                    else if (dense_wait(flag_L, pidx + 1) < 0) break;
                    const int R = nbk - b0; // live tile columns g = 0 .. R - 1 (from the end); the one that is factored next is g = R - 1
                    const double *Lpan = Lf + lfo - LS * j0 + 2 * lk + LS * lr; // + 16 LS * tile row -> this lane's operand pair
                    lds_d2 opA[kNQ], opB[kDenseCols];
                    int hq[kNQ]; // this wave's tile rows (uniform)
#pragma unroll
                    for (int q = 0; q < kNQ; ++q) {
                        hq[q] = dt_row_of<LA>(wv, q);
                        opA[q] = *reinterpret_cast<const lds_d2 *>(Lpan + 16 * LS * (hq[q] < R ? nbk - 1 - hq[q] : b0));
                    }
                    // (columns in groups of four: a group without a live column is not requested at all -- two uniform branches; inside a group the
                    // requests stay unconditional, see the note on the eleven branches of round 2 in the other form below)
#pragma unroll
                    for (int g = 0; g < 4; ++g) opB[g] = *reinterpret_cast<const lds_d2 *>(Lpan + 16 * LS * (g < R ? nbk - 1 - g : b0));
                    if (!kDenseOperandGroups || R > 4) {
#pragma unroll
                        for (int g = 4; g < 8; ++g) opB[g] = *reinterpret_cast<const lds_d2 *>(Lpan + 16 * LS * (g < R ? nbk - 1 - g : b0));
                    }
                    if (!kDenseOperandGroups || R > 8) {
#pragma unroll
                        for (int g = 8; g < kDenseCols; ++g) opB[g] = *reinterpret_cast<const lds_d2 *>(Lpan + 16 * LS * (g < R ? nbk - 1 - g : b0));
                    }
#define PV_LA_COLUMN(g)                                                                                                \
    do {                                                                                                               \
        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \
            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][0], opB[g][0], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \
        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \
            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][1], opB[g][1], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \
    } while (0)
#pragma unroll
                    for (int g = kDenseCols - 1; g >= 0; --g)
                        if (g < R) { // uniform
                            PV_LA_COLUMN(g);
                            if (g == R - 1) {
#pragma unroll
                                for (int q = 0; q < dt_col_slot<LA>(g + 1) - dt_col_slot<LA>(g); ++q)
                                    if (hq[q] <= g) {
                                        asm volatile("" ::: "memory"); // keep this a real (uniform) branch: nothing of the publish is hoisted
                                        PV_PUBLISH_ROW(dt_col_slot<LA>(g) + q, nbk - 1 - hq[q], o2);
                                    }
                                if (kDenseLaBarrier) __syncthreads(); // barrier B(p): this panel's next columns are published (R >= 1 in every panel: the rhs row's tile column)
                                else dense_signal_add(flag_pub); // (release; one count per wave and panel, whether it owns a row here or not)
                            }
                        }
#undef PV_LA_COLUMN
                    lfo += LS * (LDV - j0);
                }
            }
            __syncthreads();
            fail = sh_fail;
            {
                int lfo_end = 0; // every wave leaves with the offset behind the last panel (a failed pivot: unused)
                for (int j0 = 0; j0 < Pp; j0 += kPanel) lfo_end += LS * (LDV - j0);
                lfo = lfo_end;
            }
        } else
        for (int j0 = 0; j0 < Pp; j0 += kPanel) {
            const int k0 = j0 + kPanel, b0 = k0 >> 4, o2 = k0 & 15;
            if (j0 == 0) PV_STAMP2(8);
            if (j0 == 80) PV_STAMP2(13);
            double Ld[kPanel][kPanel], inv[kPanel];
#pragma unroll
            for (int r = 0; r < kPanel; ++r)
#pragma unroll
                for (int h = 0; h <= (r >> 1); ++h) {
                    const lds_d2 g2 = *reinterpret_cast<const lds_d2 *>(Xs + (j0 + r) * LS + 2 * h);
                    Ld[r][2 * h] = -g2[0];
                    if (2 * h + 1 <= r) Ld[r][2 * h + 1] = -g2[1];
                }
            const int irow = j0 + tid; // row owner (LDV <= 176 < 256 threads)
            double x[kPanel];
            {
                // every thread loads and carries a row (threads past the last row re-read the last one and discard the
                // result): the forward substitution below then shares a basic block with the pivot chain and fills its
                // dependency stalls instead of running after it
                const int ir = irow < LDV ? irow : LDV - 1;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const lds_d2 g2 = *reinterpret_cast<const lds_d2 *>(Xs + ir * LS + 2 * h);
                    x[2 * h] = -g2[0], x[2 * h + 1] = -g2[1];
                }
            }
            // right-looking, without ever forming the block's own L: the update of pivot cc is
            // A'[r][c2] -= A'[r][cc] * Ls[c2][cc] with Ls[c2][cc] = A'[c2][cc] / d_cc (= L[c2][cc] / L[cc][cc], which is
            // also what the forward substitution of the rows below needs); 1 / L_cc is only used for the final scaling
            double Ls[kPanel][kPanel];
#pragma unroll
            for (int cc = 0; cc < kPanel; ++cc) {
                const double dd = Ld[cc][cc];
                fail |= (!(dd > 0.0) || !isfinite(dd)) ? 1 : 0;
                inv[cc] = fast_rsqrt(dd);
                const double inv2 = inv[cc] * inv[cc];
#pragma unroll
                for (int r = cc + 1; r < kPanel; ++r) Ls[r][cc] = Ld[r][cc] * inv2;
#pragma unroll
                for (int r = cc + 1; r < kPanel; ++r)
#pragma unroll
                    for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ls[c2][cc];
                // forward substitution of this thread's row, step cc (unscaled entries: one dependent FMA per step)
#pragma unroll
                for (int c2 = cc + 1; c2 < kPanel; ++c2) x[c2] -= x[cc] * Ls[c2][cc];
            }
            if (j0 == 0) PV_STAMP2(9);
            if (j0 == 80) PV_STAMP2(14);
            if (fail) break; // uniform: every thread factored the same block
            if (j0 == 80) PV_STAMPV2(18, Ls[7][6]);
            if (tid < kPanel) {
                double iv = inv[0];
#pragma unroll
                for (int cc = 1; cc < kPanel; ++cc) iv = (tid == cc) ? inv[cc] : iv;
                tmp[j0 + tid] = iv; // 1 / L_jj for the back substitution
            }
            if (irow < LDV) {
                // the owner's row went through the forward substitution inside the pivot loop; scaling and the mask of the
                // panel's own rows (their strictly upper entries are not L) come last
                if (j0 == 80) PV_STAMPV2(19, x[7]);
#pragma unroll
                for (int cc = 0; cc < kPanel; ++cc) x[cc] = (j0 + cc <= irow) ? x[cc] * inv[cc] : 0.0;
                if (j0 == 80) PV_STAMPV2(20, x[7]);
                lds_d2 *Lrow = reinterpret_cast<lds_d2 *>(Lf + lfo + LS * tid);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    lds_d2 pr;
                    pr[0] = x[h], pr[1] = x[h + 4]; // operand pair (k, k + 4) of the two MFMAs
                    Lrow[h] = pr;
                }
            }
            __syncthreads();
            if (j0 == 0) PV_STAMP2(10);
            if (j0 == 80) PV_STAMP2(15);
            // rank-8 update of the live tiles: (-C)(16x16) += L_i (16 x 8) L_k^T, two v_mfma_f64_16x16x4_f64 per tile.
            // Operand layout (cdna_hip_programming.md section 3, f64): lane l supplies A[l & 15][l >> 4] and
            // B[l >> 4][l & 15]; it receives D[(l >> 4) + 4 r][l & 15], r = 0..3.  The live tiles are the first `na` slots;
            // they are walked from the back in groups of four (the tile column that is factored next comes first and is
            // published while the remaining groups still compute).
            {
                const int R = nbk - b0; // live tile columns g = 0 .. R - 1 (from the end); the one that is factored next is g = R - 1
                const double *Lpan = Lf + lfo - LS * j0 + 2 * lk + LS * lr; // + 16 LS * tile row -> this lane's operand pair
                // operands: A(q) = rows wv + 4 q (from the end), B(g) = column g; rows / columns outside the live range read the
                // row block b0 (a valid address; they only ever meet slots that hold no live tile)
                lds_d2 opA[3], opB[kDenseCols];
                int hq[3]; // this wave's tile rows (uniform)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    hq[q] = dense_row_of(wv, q);
                    opA[q] = *reinterpret_cast<const lds_d2 *>(Lpan + 16 * LS * (hq[q] < R ? nbk - 1 - hq[q] : b0));
                }
                // Every column's operand is requested UNCONDITIONALLY (dead columns read row block b0 like dead rows do).  Round 2 had
                // eleven uniform `if (g < R)` branches around these loads; with them, the DEscending request order compiled to a
                // k_dense whose results were wrong on the GPU (emulator fine).  Round 3 characterized it (profiles/NOTES_r1_r3.md section 4,
                // tests/micro/order_probe.py, profiles/r3_kdense_order_probe_*.txt): deterministic, independent of LDS / register
                // contents, gone when SGPRs spill to scratch instead of VGPR lanes and gone -- in BOTH orders -- without the
                // branches.  The branch-free form is what ships; it also drops eleven scalar branches per panel.
#pragma unroll
                for (int g = 0; g < kDenseCols; ++g)
                    opB[g] = *reinterpret_cast<const lds_d2 *>(Lpan + 16 * LS * (g < R ? nbk - 1 - g : b0));
                // a column's slots, unconditionally: a slot that holds no tile of this wave (its q-th row lies outside the column)
                // costs two MFMAs on registers nobody reads -- cheaper than a uniform branch per slot, which splits the MFMA
                // sequence into basic blocks (measured: 54.4 against 49.7 us)
#define PV_COLUMN(g)                                                                                                   \
    do {                                                                                                               \
        _Pragma("unroll") for (int q = 0; q < kColSlot[(g) + 1] - kColSlot[g]; ++q)                                    \
            acc[kColSlot[g] + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][0], opB[g][0], acc[kColSlot[g] + q], 0, 0, 0); \
        _Pragma("unroll") for (int q = 0; q < kColSlot[(g) + 1] - kColSlot[g]; ++q)                                    \
            acc[kColSlot[g] + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][1], opB[g][1], acc[kColSlot[g] + q], 0, 0, 0); \
    } while (0)
                // from the column that is factored next (published as soon as it is updated) down to the last one
#pragma unroll
                for (int g = kDenseCols - 1; g >= 0; --g)
                    if (g < R) { // uniform
                        PV_COLUMN(g);
                        if (g == R - 1) {
#pragma unroll
                            for (int q = 0; q < kColSlot[g + 1] - kColSlot[g]; ++q)
                                if (hq[q] <= g) {
                                    asm volatile("" ::: "memory"); // keep this a real (uniform) branch: nothing of the publish is hoisted
                                    PV_PUBLISH_ROW(kColSlot[g] + q, nbk - 1 - hq[q], o2); // (the slot tables sbi / sbk are not kept alive across the loop)
                                }
                        }
                    }
#undef PV_COLUMN
            }
            if (j0 == 0) PV_STAMP2(11);
            if (j0 == 80) PV_STAMP2(16);
            __syncthreads();
            if (j0 == 0) PV_STAMP2(12);
            if (j0 == 80) PV_STAMP2(17);
            lfo += LS * (LDV - j0);
        }
#undef PV_PUBLISH
#undef PV_PUBLISH_ROW
        PV_STAMP2(5);
        // ---------------- back substitution L^T y = z, 8 columns per step ----------------
        // L(i, k) = Lf[off(k >> 3) + (i - 8 (k >> 3)) * LS + perm(k & 7)], off(p) = LS (p LDV - 4 p (p - 1)), perm(c) = 2 (c & 3) + (c >> 2)
        if (tid == 0) sh_fail = fail;
        __syncthreads();
        if (!fail) {
            auto lf_at = [&](int i, int k) -> int {
                const int p = k >> 3, cI = k & 7;
                return LS * (p * LDV - 4 * p * (p - 1)) + (i - 8 * p) * LS + 2 * (cI & 3) + (cI >> 2);
            };
            // inverses of the 8 x 8 diagonal blocks, all at once (thread = one column of one block: L X = e_c), so that a
            // block step below is a short dot product instead of a dependent triangular solve.  Li overlays Xs.
            double *Li = Xs;
            const int npan = Pp / kPanel;
            for (int t = tid; t < npan * kPanel; t += nthr) {
                const int p = t >> 3, cI = t & 7, j0 = p * kPanel;
                double X[kPanel];
#pragma unroll
                for (int r = 0; r < kPanel; ++r) {
                    double s = (r == cI) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = 0; k < r; ++k) s -= Lf[lf_at(j0 + r, j0 + k)] * X[k];
                    X[r] = s * tmp[j0 + r];
                }
#pragma unroll
                for (int r = 0; r < kPanel; ++r) Li[p * 64 + cI * 8 + r] = X[r]; // Li[p][c][r] = (L_pp^-1)[r][c]
            }
            double zz = 0;
            for (int a = tid; a < Pp; a += nthr) {
                const double z = Lf[lf_at(Pp, a)]; // z = L^-1 rhs (row Pp went through the factorization); 0 in the padding
                yv[a] = z;
                zz += (a < P && act[a] != 0.0) ? z * z : 0.0;
            }
            double s1[1] = {zz};
            block_sum<1>(s1, red_scratch);
            if (tid == 0) c->pose_qyy = s1[0]; // y^T (S + mu D^2) y = |z|^2 ; the mu term is removed below
            __syncthreads();
            // Every wave forms y_b = L_bb^-T z_b on its own (no extra barrier): lane (c, r) multiplies (L_bb^-1)[r][c] z_r,
            // three butterfly steps sum over r, eight lane reads broadcast y_b to the wave; then the rows above the block
            // are updated in parallel.  One barrier per block.
            const int lc = lane >> 3, lrr = lane & 7;
            for (int p = npan - 1; p >= 0; --p) {
                const int jb0 = p * kPanel;
                // the two reads the dependent chain starts from go first (LDS answers in order), the L row of this thread's
                // row a -- independent of y, needed only after the lane reads -- behind them
                const double li = Li[p * 64 + lane], zr = yv[jb0 + lrr]; // Li[p][c][r] = (L_pp^-1)[r][c]
                double lrow[kPanel]; // L(jb0 + cc, a)
                const int a = tid;
                double ya = 0;
                if (a < jb0) {
                    const double *T = Lf + lf_at(jb0, a);
#pragma unroll
                    for (int cc = 0; cc < kPanel; ++cc) lrow[cc] = T[LS * cc];
                    ya = yv[a];
                }
                double part = li * zr;
                part = group8_sum(part);
                double yb[kPanel];
#pragma unroll
                for (int cc = 0; cc < kPanel; ++cc) yb[cc] = readlane_f64(part, 8 * cc);
                if (wv == 0 && lrr == 0) ysol[jb0 + lc] = part;
                if (a < jb0) {
                    double acc2 = 0, acc3 = 0; // (two chains of four)
#pragma unroll
                    for (int cc = 0; cc < kPanel / 2; ++cc) acc2 += lrow[cc] * yb[cc], acc3 += lrow[cc + 4] * yb[cc + 4];
                    yv[a] = ya - (acc2 + acc3);
                }
                __syncthreads();
            }
        }
    } else {
        // generic path (systems too large for LDS): tiles in HBM.  The factorization walks panels of W = 32 (16) columns:
        // a panel is pulled into LDS once, factored there in four (two) steps of 8 columns -- redundant 8 x 8 block
        // factorization, row owners turn their 8 entries into L, the rest of the PANEL is updated on the VALU -- written
        // back as finished L, and only then the trailing matrix is swept once with a rank-W update on the matrix cores
        // (W / 4 MFMAs per tile and read-modify-write).  One workgroup pulls ~70 GB/s from L2: with 8-column panels the
        // 57 sweeps of a 450-row system were 80 % of this kernel; a sweep per 32 columns moves a quarter of the bytes.
        const int W = dense_panel_width(LDV), WS = W + 2; // row stride W + 2 doubles: rows of a tile fall on different banks
        double *Pn = Lp;
        for (int J0 = 0; J0 < Pp; J0 += W) {
            const int Wc = Pp - J0 < W ? Pp - J0 : W, jb = J0 >> 4, ntr = nbk - jb, ntc = (Wc + 15) >> 4;
            if (J0 == 0) PV_STAMP2(8);
            // ---- panel -> LDS, un-negated: a wave takes whole tiles (32 bytes per lane = the four entries rows lk + 4 r,
            // column lr it would own in an MFMA), four tiles in flight ----
            {
                const int nt = ntr * ntc;
                for (int t0 = 4 * wv; t0 < nt; t0 += 4 * NW) {
                    lds_d2 g01[4], g23[4];
    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = t0 + u < nt ? t0 + u : nt - 1, tr = t / ntc, tc = t - tr * ntc;
                        const bool have = jb + tc <= jb + tr;
                        const double *T = A + tile_base(jb + tr, have ? jb + tc : jb) + 4 * lane;
                        g01[u] = *reinterpret_cast<const lds_d2 *>(T), g23[u] = *reinterpret_cast<const lds_d2 *>(T + 2);
                        if (!have) g01[u][0] = 0.0, g01[u][1] = 0.0, g23[u][0] = 0.0, g23[u][1] = 0.0; // tile above the diagonal
                    }
    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = t0 + u;
                        if (t < nt) {
                            const int tr = t / ntc, tc = t - tr * ntc, col = 16 * tc + lr;
                            if (col < Wc) {
                                double *dst = Pn + (16 * (jb + tr) + lk) * WS + col;
                                dst[0] = -g01[u][0], dst[4 * WS] = -g01[u][1], dst[8 * WS] = -g23[u][0], dst[12 * WS] = -g23[u][1];
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (J0 == 0) PV_STAMP2(9);
            for (int s8 = 0; s8 < Wc; s8 += kPanel) {
                const int j0 = J0 + s8;
                double Ld[kPanel][kPanel], inv[kPanel];
    #pragma unroll
                for (int r = 0; r < kPanel; ++r)
    #pragma unroll
                    for (int cc = 0; cc <= r; ++cc) Ld[r][cc] = Pn[(j0 + r) * WS + s8 + cc]; // broadcast reads
    #pragma unroll
                for (int cc = 0; cc < kPanel; ++cc) {
                    const double dd = Ld[cc][cc];
                    fail |= (!(dd > 0.0) || !isfinite(dd)) ? 1 : 0;
                    inv[cc] = fast_rsqrt(dd);
    #pragma unroll
                    for (int r = cc + 1; r < kPanel; ++r) Ld[r][cc] *= inv[cc];
    #pragma unroll
                    for (int r = cc + 1; r < kPanel; ++r)
    #pragma unroll
                        for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ld[c2][cc];
                }
                if (fail) break; // uniform: every thread factored the same block
                __syncthreads(); // everybody has read the block before its owners overwrite it
                if (tid < kPanel) {
                    double iv = inv[0];
    #pragma unroll
                    for (int cc = 1; cc < kPanel; ++cc) iv = (tid == cc) ? inv[cc] : iv;
                    tmp[j0 + tid] = iv; // 1 / L_jj for the back substitution
                }
                for (int ir = j0 + tid; ir < LDV; ir += nthr) { // row owners: 8 entries -> L (strictly upper entries of the block are not L)
                    lds_d2 *rowp = reinterpret_cast<lds_d2 *>(Pn + ir * WS + s8);
                    double x[kPanel];
    #pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const lds_d2 g = rowp[h];
                        x[2 * h] = g[0], x[2 * h + 1] = g[1];
                    }
    #pragma unroll
                    for (int cc = 0; cc < kPanel; ++cc) {
                        x[cc] = (j0 + cc <= ir) ? x[cc] * inv[cc] : 0.0;
    #pragma unroll
                        for (int c2 = cc + 1; c2 < kPanel; ++c2) x[c2] -= x[cc] * Ld[c2][cc];
                    }
    #pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        lds_d2 g;
                        g[0] = x[2 * h], g[1] = x[2 * h + 1];
                        rowp[h] = g;
                    }
                }
                __syncthreads();
                // the rest of the panel: column J0 + c of row ir loses L(ir, j0..j0+7) . L(J0 + c, j0..j0+7); the second factor is
                // a row of the panel's own square (finished entries, broadcast reads)
                const int c_lo = s8 + kPanel;
                if (c_lo < Wc) {
                    for (int ir = j0 + kPanel + tid; ir < LDV; ir += nthr) {
                        double *rowp = Pn + ir * WS;
                        double x[kPanel];
    #pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const lds_d2 g = *reinterpret_cast<const lds_d2 *>(rowp + s8 + 2 * h);
                            x[2 * h] = g[0], x[2 * h + 1] = g[1];
                        }
                        for (int c = c_lo; c < Wc; c += 2) { // two columns per step (16-byte accesses)
                            const double *l0 = Pn + (J0 + c) * WS + s8, *l1 = l0 + WS;
                            lds_d2 acc = *reinterpret_cast<const lds_d2 *>(rowp + c);
    #pragma unroll
                            for (int h = 0; h < 4; ++h) {
                                const lds_d2 a0 = *reinterpret_cast<const lds_d2 *>(l0 + 2 * h), a1 = *reinterpret_cast<const lds_d2 *>(l1 + 2 * h);
                                acc[0] -= x[2 * h] * a0[0] + x[2 * h + 1] * a0[1];
                                acc[1] -= x[2 * h] * a1[0] + x[2 * h + 1] * a1[1];
                            }
                            *reinterpret_cast<lds_d2 *>(rowp + c) = acc;
                        }
                    }
                    __syncthreads();
                }
            }
            if (fail) break;
            if (J0 == 0) PV_STAMP2(10);
            // ---- finished L of the panel -> HBM (the back substitution reads it there); same tile ownership as the load ----
            {
                const int nt = ntr * ntc;
                for (int t = wv; t < nt; t += NW) {
                    const int tr = t / ntc, tc = t - tr * ntc, col = 16 * tc + lr;
                    if (tc > tr || col >= Wc) continue; // above the diagonal / right of a narrow last panel
                    const double *src = Pn + (16 * (jb + tr) + lk) * WS + col;
                    const int row = 16 * (jb + tr) + lk, ca = J0 + col;
                    lds_d2 w0, w1;
                    w0[0] = ca <= row ? src[0] : 0.0, w0[1] = ca <= row + 4 ? src[4 * WS] : 0.0;
                    w1[0] = ca <= row + 8 ? src[8 * WS] : 0.0, w1[1] = ca <= row + 12 ? src[12 * WS] : 0.0;
                    double *T = A + tile_base(jb + tr, jb + tc) + 4 * lane;
                    *reinterpret_cast<lds_d2 *>(T) = w0, *reinterpret_cast<lds_d2 *>(T + 2) = w1;
                }
            }
            // ---- rank-W update of the trailing matrix on the matrix cores (nothing behind the last panel; every panel before
            // the last one is W wide and ends on a tile boundary) ----
            if (J0 + Wc < Pp) {
                __syncthreads(); // (the write-back above read the panel; the sweep only reads it too, but keeps the waves together)
                if (W == 32) dense_trailing_sweep<32, NW>(A, Pn, WS, (J0 + Wc) >> 4, nbk, wv, lane);
                else dense_trailing_sweep<16, NW>(A, Pn, WS, (J0 + Wc) >> 4, nbk, wv, lane);
            }
            if (J0 == 0) PV_STAMP2(11);
            __syncthreads(); // the panel buffer is free again, the trailing tiles are in place (same-workgroup visibility)
            if (J0 == 0) PV_STAMP2(12);
        }
        PV_STAMP2(5);
        // ---------------- back substitution L^T y = z, 8 columns per step ----------------
        if (tid == 0) sh_fail = fail;
        __syncthreads();
        if (!fail) {
            // inverses of the 8 x 8 diagonal blocks, all at once (thread = one column of one block: L X = e_c), so that a
            // block step below is a short dot product instead of a dependent triangular solve.  Li aliases the panel buffer.
            double *Li = Lp;
            const int npan = Pp / kPanel;
            for (int t = tid; t < npan * kPanel; t += nthr) {
                const int p = t >> 3, cI = t & 7, j0 = p * kPanel;
                const double *T = A + tile_base(j0 >> 4, j0 >> 4);
                const int o = j0 & 15;
                double X[kPanel];
    #pragma unroll
                for (int r = 0; r < kPanel; ++r) {
                    double s = (r == cI) ? 1.0 : 0.0;
    #pragma unroll
                    for (int k = 0; k < r; ++k) s -= T[tile_off(o + r, o + k)] * X[k];
                    X[r] = s * tmp[j0 + r];
                }
    #pragma unroll
                for (int r = 0; r < kPanel; ++r) Li[p * 64 + cI * 8 + r] = X[r]; // Li[p][c][r] = (L_pp^-1)[r][c]
            }
            double zz = 0;
            for (int a = tid; a < Pp; a += nthr) {
                const double z = A[mat_at(Pp, a)]; // z = L^-1 rhs (row Pp went through the factorization); 0 in the padding
                yv[a] = z;
                zz += (a < P && act[a] != 0.0) ? z * z : 0.0;
            }
            double s1[1] = {zz};
            block_sum<1>(s1, red_scratch);
            if (tid == 0) c->pose_qyy = s1[0]; // y^T (S + mu D^2) y = |z|^2 ; the mu term is removed below
            __syncthreads();
            // every thread forms y_b = L_bb^-T z_b redundantly (broadcast loads), then the rows above the block are updated
            // in parallel (one row per thread: LDV <= 496 < 512 threads).  The L entries a step needs do not depend on y:
            // those of the NEXT step are requested before this step's barrier, so a step does not wait for a round trip
            // to L2.  One barrier per block.
            static_assert(LDSMAT || (kMaxFrames * 15 + 16 <= 2 * kDenseThreads), "one row of the HBM-resident system per thread");
            auto load_lrow = [&](int jb0, double *lrow) { // L(jb0 + cc, tid), cc = 0..7
                if (tid < jb0) {
                    const double *T = A + tile_base(jb0 >> 4, tid >> 4);
                    const int o = jb0 & 15;
    #pragma unroll
                    for (int cc = 0; cc < kPanel; ++cc) lrow[cc] = T[tile_off(o + cc, tid & 15)];
                }
            };
            double lrow[kPanel], lnext[kPanel];
            load_lrow((npan - 1) * kPanel, lrow);
            for (int p = npan - 1; p >= 0; --p) {
                const int jb0 = p * kPanel;
                if (p > 0) load_lrow(jb0 - kPanel, lnext);
                double zb[kPanel], yb[kPanel];
                {
                    const lds_d2 *Z = reinterpret_cast<const lds_d2 *>(yv + jb0);
    #pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const lds_d2 z2 = Z[h];
                        zb[2 * h] = z2[0], zb[2 * h + 1] = z2[1];
                    }
                }
                const lds_d2 *L2 = reinterpret_cast<const lds_d2 *>(Li + p * 64);
    #pragma unroll
                for (int cc = 0; cc < kPanel; ++cc) {
                    double s = 0;
    #pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const lds_d2 l2 = L2[cc * 4 + h];
                        s += l2[0] * zb[2 * h] + l2[1] * zb[2 * h + 1];
                    }
                    yb[cc] = s;
                }
                if (tid < kPanel) {
                    double yo = yb[0];
    #pragma unroll
                    for (int cc = 1; cc < kPanel; ++cc) yo = (tid == cc) ? yb[cc] : yo;
                    ysol[jb0 + tid] = yo;
                }
                if (tid < jb0) {
                    double acc2 = 0;
    #pragma unroll
                    for (int cc = 0; cc < kPanel; ++cc) acc2 += lrow[cc] * yb[cc];
                    yv[tid] -= acc2;
                }
    #pragma unroll
                for (int cc = 0; cc < kPanel; ++cc) lrow[cc] = lnext[cc];
                __syncthreads();
            }
        }
    }
    PV_STAMP2(6);
    // ---------------- outputs ----------------
    // y solves (S + mu D^2) y = rhs_s ; step direction y' = -y ; gn = D y'.  The scalars of the step are summed together
    // with the count of non-finite entries (one block reduction, one barrier less than testing the solution first); a
    // failed solve leaves them unused and the step vectors unread (the next linearization is then not a candidate).
    double nbad = 0;
    {
        double s_g2 = 0, s_gn2 = 0, s_gd = 0, s_qvy = 0, s_gy = 0;
        if (!sh_fail)
            for (int a = tid; a < P; a += nthr) {
                nbad += isfinite(ysol[a]) ? 0.0 : 1.0;
                const double yp = act[a] != 0.0 ? -ysol[a] : 0.0;
                const double Da = keep_Da, gh = keep_gh, gn = Da * yp;
                v.ystep[a] = cpl[a] * yp;
                v.vstep[a] = act[a] != 0.0 ? cpl[a] * vv[a] : 0.0;
                s_g2 += gh * gh, s_gn2 += gn * gn, s_gd += gh * gn;
                s_qvy += act[a] != 0.0 ? vv[a] * (mu * Da * Da * ysol[a] - diagH[a]) : 0.0; // v^T S y' = -v^T (rhs_s - mu D^2 y), S y = rhs_s - mu D^2 y
                s_gy += act[a] != 0.0 ? cpl[a] * gtot[a] * yp : 0.0;
            }
        double sc[6] = {s_g2, s_gn2, s_gd, s_qvy, s_gy, nbad};
        block_sum<6>(sc, red_scratch);
        nbad = sc[5];
        if (tid == 0) {
            const int injected = c->dbg_fail_left > 0 && c->dbg_fail_left < kDbgNegativePivot; // fault injection (tests only)
            if (injected) c->dbg_fail_left--;
            sh.do_solve = !sh_fail && nbad == 0.0 && !injected;
            if (sh.do_solve) {
                // y'^T S y' = |z|^2 - mu sum D_a^2 y'_a^2 = |z|^2 - mu |gn_p|^2
                c->pose_g2 = sc[0], c->pose_gn2 = sc[1], c->pose_gdot = sc[2], c->pose_qvy = sc[3], c->pose_gy = sc[4];
                c->pose_qyy = c->pose_qyy - mu * sc[1];
            }
        }
    }
    if (tid == 0) {
        const int ok = sh.do_solve;
        if (ok) {
            c->solve_ok = 1, c->mode = MODE_CANDIDATE, c->scaling_ready = 1, c->retry_relin = 0;
        } else {
            // LINEAR_SOLVER_FAILURE at this mu: escalate; below max_mu re-run the Schur accumulation with the new mu
            c->solve_ok = 0;
            c->mu *= 10.0;
            c->scaling_ready = 1;
            if (c->mu < 1.0) {
                c->mode = MODE_RELIN, c->retry_relin = 1;
            } else {
                // mu >= max_mu: ComputeTrustRegionStep fails -> invalid step, and so will every following iteration
                // (x does not move, mu only grows).  Ceres walks them one by one until the fifth consecutive invalid step
                // (FAILURE) or the iteration limit; replay that bookkeeping here instead of burning launches on it.
                sh.replay_first = -1, sh.replay_count = 0;
                while (true) {
                    if (++c->invalid_steps >= 5) { // HandleInvalidStep
                        c->termination = 2;
                        break;
                    }
                    c->mu *= 10.0; // DoglegStrategy::StepIsInvalid
                    c->it_valid = 0, c->it_success = 0, c->it_cost = c->x_cost, c->it_cost_change = 0, c->it_step_norm = 0, c->it_rel = 0;
                    const int slot = record_trace(v, c, c->iter); // Finalize
                    if (slot >= 0) {
                        if (sh.replay_first < 0) sh.replay_first = slot;
                        sh.replay_count++;
                    }
                    if (c->iter >= v.dm.max_iter) {
                        c->termination = 1;
                        break;
                    }
                    if (c->radius <= 1e-32) {
                        c->termination = 0;
                        break;
                    }
                    c->iter++;
                }
                c->done = 1, c->mode = MODE_DONE;
            }
        }
        sh.do_solve = ok;
    }
    __syncthreads();
    if (!sh.do_solve) {
        if (c->done && v.trace_states && sh.replay_count > 0) { // the replayed iterations all sit at the accepted iterate
            const int cur = c->cur;
            for (int q = 0; q < sh.replay_count; ++q) {
                double *dst = v.trace_states + (size_t)(sh.replay_first + q) * (N * 16 + v.dm.M);
                for (int e = tid; e < N * 16; e += nthr) dst[e] = v.fs[(size_t)cur * N * 16 + e];
                for (int e = tid; e < v.dm.M; e += nthr) dst[N * 16 + e] = v.rho[(size_t)cur * v.dm.M + e];
            }
        }
        if (tid < (int)(sizeof(Ctrl) / sizeof(double))) reinterpret_cast<double *>(cg)[tid] = reinterpret_cast<const double *>(c)[tid];
        return;
    }
    if (v.dm.fuse_backsub) {
        // small windows: the landmark back-substitution runs right here (one launch less per iteration); the step
        // vectors are taken from LDS copies
        for (int a = tid; a < P; a += nthr) {
            const double yp = act[a] != 0.0 ? -ysol[a] : 0.0;
            tmp[a] = act[a] != 0.0 ? cpl[a] * vv[a] : 0.0; // C_p v_p
            yv[a] = cpl[a] * yp;                           // C_p y'_p
        }
        __syncthreads();
        double sb[6] = {0, 0, 0, 0, 0, 0};
        backsub_landmarks(v, c->lin, mu, tmp, yv, tid, nthr, sb);
        block_sum<6>(sb, red_scratch);
        if (tid == 0) {
            for (int k = 0; k < 6; ++k) v.back_part[k] = sb[k];
            v.back_part[6] = v.back_part[7] = 0;
        }
    }
    __syncthreads();
    // the control block goes back as one store per thread (a struct copy by thread 0 is 24 dependent LDS reads and stores)
    if (tid < (int)(sizeof(Ctrl) / sizeof(double))) reinterpret_cast<double *>(cg)[tid] = reinterpret_cast<const double *>(c)[tid];
    PV_STAMP2(7);
    if (v.dbg && threadIdx.x == 0) v.dbg[2 * 32 + 30] = 0, v.dbg[2 * 32 + 31] = wall_clock64() - g_stamp_w0[2]; // 10 ns units
}

// ------------------------------------------------------------------------------------------------------
// split finalize (Dims::split_fin): the part of Ceres' FinalizeIterationAndCheckIfMinimizerCanContinue that does not decide what
// k_dense factors -- run by one extra workgroup of k_backsub, beside the landmark back-substitution instead of in front of the
// factorization: the gradient max-norm of the linearization just accepted (patched into the iteration's trace record; the
// gradient-tolerance exit takes the iteration k_dense has started back), the state-updating callback's copies
// (update_state_every_iteration: user state <- accepted iterate, the biases the linearization was evaluated with) and the trace
// states.  Runs whenever k_dense left fin_flags set, also on the last launch of a solve.
// ------------------------------------------------------------------------------------------------------
__device__ void backsub_finalize(const View &v) {
    Ctrl *c = v.ctrl;
    const int N = v.dm.N, d = v.dm.d, tid = threadIdx.x, nthr = blockDim.x;
    // one round of loads, whatever the flags turn out to be (every dependent round is a trip through the fabric, and this workgroup
    // must not outlast the landmark workgroups beside it): control fields, and per frame thread the gradient, both state buffers
    // and the user state's biases
    const int flags = c->fin_flags, cur = c->cur, slot = c->fin_trace_slot, was_done = c->done;
    const double lm_gmax = c->fin_lm_gmax;
    const int f = tid < N ? tid : 0;
    double g[15], x0[16], x1[16], ub[6];
#pragma unroll
    for (int k = 0; k < 15; ++k) g[k] = k < d ? v.gtot[d * f + k] : 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) x0[k] = v.fs[(size_t)f * 16 + k], x1[k] = v.fs[((size_t)N + f) * 16 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) ub[k] = v.fs_user[(size_t)f * 16 + 10 + k];
    const bool p_act = v.pose_active[f] != 0, m_act = v.motion_active[f] != 0;
#ifdef PV_DEBUG_FIN
    if (threadIdx.x == 0) printf("fin: flags %d slot %d cur %d done %d ts %p\n", flags, slot, cur, was_done, (void *)v.trace_states);
#endif
    if (!flags) return; // uniform
    const bool accepted = (flags & kFinAccepted) != 0, first = (flags & kFinFirst) != 0;
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = cur ? x1[k] : x0[k];
    if (accepted) {
        // gradient_max_norm = max | x - (x (+) -g) | over the free blocks (ambient coordinates)
        double gm = 0;
        if (tid < N) {
            if (p_act) {
                double ng[6], y[7];
                for (int k = 0; k < 6; ++k) ng[k] = -g[k];
                pose_plus(y, x, ng, ng + 3);
                for (int k = 0; k < 7; ++k) gm = fmax(gm, fabs(x[k] - y[k]));
            }
            if (d == 15 && m_act)
                for (int k = 0; k < 9; ++k) gm = fmax(gm, fabs(g[6 + k]));
        }
        if (tid < 64) {
            gm = wave_max(gm); // N <= 32 < 64: wave 0 holds every frame
            if (tid == 0) {
                const double gmax = fmax(gm, lm_gmax);
                c->grad_max = gmax;
                if (slot >= 0) v.trace[slot].gradient_max_norm = gmax;
                if (!was_done && gmax <= 1e-10) { // GradientToleranceReached: the iteration k_dense has started does not happen
                    // INVARIANT: this write races with the landmark workgroups of the same launch, which read c->done at their
                    // start -- one that starts late may see it and return without its back_part row while others have written theirs.  That is
                    // allowed because NOTHING reads back_part, gnl or the k_backsub scalars once done is set: k_linearize's prologue returns on
                    // `done` before it touches them, k_dense's control section returns on it, and the host reads states only.  A consumer that
                    // is added later must test `done` first, like those do.
                    c->termination = 0, c->done = 1, c->mode = MODE_DONE, c->iter = c->iter - 1;
                }
            }
        }
        // StateUpdatingCallback (update_state_every_iteration), one frame per thread: user state <- accepted iterate.  The accepted
        // linearization was evaluated with the OLD user biases: kept for a later RELIN (the first one with the initial state's own)
        if (tid < N) {
#pragma unroll
            for (int k = 0; k < 6; ++k) v.bias0_lin[(size_t)tid * 6 + k] = first ? x[10 + k] : ub[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) v.fs_user[(size_t)tid * 16 + k] = x[k];
        }
    }
    if (v.trace_states && slot >= 0) {
        double *dst = v.trace_states + (size_t)slot * (N * 16 + v.dm.M);
        if (tid < N)
            for (int k = 0; k < 16; ++k) dst[(size_t)tid * 16 + k] = x[k];
        for (int e = tid; e < v.dm.M; e += nthr) dst[N * 16 + e] = v.rho[(size_t)cur * v.dm.M + e];
    }
    __syncthreads(); // every thread has read the flags (and the control fields above) before they are cleared
    if (tid == 0) c->fin_flags = 0;
}

// ------------------------------------------------------------------------------------------------------
// k_backsub: landmark back-substitution + landmark parts of the dogleg scalars; <= 64 WGs, one row each
// ------------------------------------------------------------------------------------------------------
// Sixteen lanes (one DPP row) per landmark: lane `sub` takes observations sub, sub + 16, ... so that the dependent
// obs_frame -> step loads of all observations travel together (a thread walking its landmark's ~10 observations in turn
// pays ten L2 round trips back to back); the two dot products are summed over the row and lane 0 finishes the landmark.
// The control word and the first operands are requested together: nothing is written before the word has arrived.
__global__ void __launch_bounds__(256) k_backsub(View v) {
    // Two dependent rounds of loads instead of four (each is a trip through the fabric, ~2000 cycles: the producers ran on
    // other XCDs): round 1 = control word, the landmarks' CSR entries and anchors, and the two step vectors, which go to LDS
    // (the per-observation lookups vs[t] / ys[t] then cost an LDS read instead of a third trip behind obs_frame[o]);
    // round 2 = everything addressed by `lin` / the CSR offsets, including the per-landmark scalars of the lanes that
    // finish a landmark (requested before the sums they are combined with, not after).
    const Ctrl *c = v.ctrl;
    if (v.dm.split_fin && (int)blockIdx.x == v.dm.G_back) { // the extra workgroup: what k_dense left to be finished (uniform)
        backsub_finalize(v);
        return;
    }
    const int done = c->done, solve_ok = c->solve_ok, lin = c->lin;
    const double mu = c->mu;
    __shared__ double scratch[7 * 16];
    __shared__ double stepv[kMaxFrames * 15], stepy[kMaxFrames * 15];
    const int M = v.dm.M, d = v.dm.d, P = v.dm.P;
    const size_t Ms = (size_t)M, Fs = (size_t)v.dm.F;
    const int sub = threadIdx.x & 15, per_block = blockDim.x >> 4;
    const int l_first = blockIdx.x * per_block + (threadIdx.x >> 4);
    const int lc_first = l_first < M ? l_first : M - 1;
    int o0n = v.lm_ptr[lc_first], o1n = v.lm_ptr[lc_first + 1], an = v.lm_anchor[lc_first];
    for (int e = threadIdx.x; e < P; e += blockDim.x) stepv[e] = v.vstep[e], stepy[e] = v.ystep[e];
    __shared__ double rawv[kMaxFrames * 15];
    const bool scaled_img = v.dm.img_scaled && c->img_scaled_now;
    if (v.dm.img_scaled)
        for (int e = threadIdx.x; e < P; e += blockDim.x) rawv[e] = v.vraw[e];
    // qvv_back: this workgroup's tiles of the reduced system's image (written by k_reduce, loaded by k_dense) for the pose part of
    // v^T H v = (C v)^T H (C v): thread = one entry of a tile (MFMA accumulator order: entry e of lane e >> 2, r = e & 3)
    const int n_lm_wg = v.dm.split_fin ? v.dm.G_back : (int)gridDim.x; // landmark workgroups (the finalize workgroup is not one)
    double himg = 0;
    int hi = 0, hk = 0;
    const int n_img_tiles = (v.dm.qvv_back && (v.dm.world <= 1 || v.dm.rank == 0)) ? v.dm.img_sz >> 8 : 0; // (shards: rank 0's rows carry it)
    const int my_tile = (int)blockIdx.x;
    if (my_tile < n_img_tiles) himg = v.img[((size_t)my_tile << 8) + threadIdx.x];
    if (done || !solve_ok) return; // uniform (the loads above were only issued)
    __syncthreads();
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int t = my_tile; t < n_img_tiles; t += n_lm_wg) {
        if (t != my_tile) himg = v.img[((size_t)t << 8) + threadIdx.x];
        int bi = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
        while (((bi + 1) * (bi + 2)) >> 1 <= t) ++bi;
        while (((bi * (bi + 1)) >> 1) > t) --bi;
        const int bk = t - ((bi * (bi + 1)) >> 1), ln = threadIdx.x >> 2, r = threadIdx.x & 3;
        hi = 16 * bi + (ln >> 4) + 4 * r, hk = 16 * bk + (ln & 15);
        // (the image is the lower triangle; scaled_img: it holds -(C S C), and v^T (C S C) v takes v = g^ / D itself)
        if (hi < P && hk <= hi) s[6] += scaled_img ? (hi == hk ? -himg : -2.0 * himg) * rawv[hi] * rawv[hk] : (hi == hk ? himg : 2.0 * himg) * stepv[hi] * stepv[hk];
    }
    for (int l0 = blockIdx.x * per_block; l0 < M; l0 += n_lm_wg * per_block) { // uniform trip count per block
        const int l = l0 + (threadIdx.x >> 4);
        const bool in = l < M;
        const int lc = in ? l : M - 1;
        const int o0 = o0n, o1 = o1n, a = an;
        {   // CSR entries of this thread's next landmark (large windows: several per thread)
            const int ln = l + n_lm_wg * per_block, lcn = ln < M ? ln : M - 1;
            if (l0 + n_lm_wg * per_block < M) o0n = v.lm_ptr[lcn], o1n = v.lm_ptr[lcn + 1], an = v.lm_anchor[lcn];
        }
        const double *Wa = v.Wa + lin * Ms * 6, *Wt = v.Wt + lin * Fs * 6;
        const bool fin = sub == 0 && in && o1 != o0; // this lane finishes the landmark
        double Hll = 0, bl = 0, D = 1, gh = 0, cl = 0;
        if (fin) Hll = v.Hll[lin * Ms + l], bl = v.bl[lin * Ms + l], D = v.Dl[lin * Ms + l], gh = v.ghl[lin * Ms + l], cl = v.cl[l];
        double Wv = 0, Wy = 0; // W_l . (C_p v_p), W_l . (C_p y'_p)
        if (sub < 6) {
            const double wa = Wa[(size_t)lc * 6 + sub];
            Wv = wa * stepv[d * a + sub], Wy = wa * stepy[d * a + sub];
        }
        for (int o = o0 + sub; o < o1; o += 16) {
            const int t = v.obs_frame[o];
            const double *w = Wt + (size_t)o * 6;
            double wk[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) wk[k] = w[k];
            const double *vs = stepv + d * t, *ys = stepy + d * t;
            double av = 0, ay = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) av += wk[k] * vs[k], ay += wk[k] * ys[k];
            Wv += av, Wy += ay;
        }
#pragma unroll
        for (int pat = 0; pat < 4; ++pat) Wv += dpp_f64(Wv, pat), Wy += dpp_f64(Wy, pat);
        if (sub == 0 && in) {
            double *gnl = v.gnl + lin * Ms;
            if (o1 == o0) {
                gnl[l] = 0.0;
            } else {
                const double Hs = cl * cl * Hll, A = Hs + mu * D * D;
                const double w = cl * cl / A;
                const double yl = -cl * (bl + Wy) / A; // y'_l
                const double vl = gh / D;
                const double gn = D * yl;
                gnl[l] = gn;
                s[0] += gn * gn;
                s[1] += gh * gn;
                s[2] += w * Wv * Wv + 2 * cl * vl * Wv + Hs * vl * vl;            // v^T H v (landmark + add-back)
                s[3] += w * Wv * Wy + cl * vl * Wy + cl * yl * Wv + Hs * vl * yl; // v^T H y'
                s[4] += w * Wy * Wy + 2 * cl * yl * Wy + Hs * yl * yl;            // y'^T H y'
                s[5] += cl * bl * yl;                                             // g_s^T y'
            }
        }
    }
    if (done || !solve_ok) return;
    block_sum<7>(s, scratch);
    if (threadIdx.x == 0) {
        double *row = v.back_part + (size_t)blockIdx.x * kNumBackScal;
        for (int k = 0; k < 7; ++k) row[k] = s[k];
        row[7] = 0;
    }
}

// Landmark-sharded solves only: back_local keeps this rank's latest k_backsub sums (they stay valid across rejected
// steps, when k_backsub does not run); every slot copies them into back_red, which the host then all-reduces IN PLACE --
// so the collective can be issued unconditionally without accumulating stale values.
// (one wave.  The rows are summed the way the prologue of k_linearize sums them on one GPU -- lane-strided partial sums, then the wave
// reduction -- with every load of a lane independent of the others: seven threads walking the rows one dependent load at a time took 21 us
// for the 256 rows of a 10 x 50 000 window, inside the serial part of every sharded iteration)
__global__ void __launch_bounds__(64) k_back_reduce(View v, int rows, double *back_local) {
    const Ctrl *c = v.ctrl;
    const int lane = threadIdx.x;
    const bool live = !c->done && c->solve_ok; // uniform
    double b[kNumBackScal];
#pragma unroll
    for (int k = 0; k < kNumBackScal; ++k) b[k] = 0.0;
    if (live) {
        for (int row = lane; row < rows; row += 64)
#pragma unroll
            for (int k = 0; k < kNumBackScal; ++k) b[k] += v.back_part[(size_t)row * kNumBackScal + k];
#pragma unroll
        for (int k = 0; k < kNumBackScal; ++k) b[k] = wave_sum(b[k]);
    }
    if (lane < kNumBackScal) {
        if (live) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < kNumBackScal; ++k) s = lane == k ? b[k] : s;
            back_local[lane] = s;
        }
        v.back_red[lane] = back_local[lane];
    }
}

// ------------------------------------------------------------------------------------------------------
// post-solve quality pass (bundle_adjustor.cpp:277-296) and compute_reprojection_error (:321-336)
// ------------------------------------------------------------------------------------------------------
// `pack` (optional, solve path): the accepted iterate and the pass's outputs in ONE contiguous buffer -- frame states [16N], inverse depths [M],
// quality [M] (0 where the landmark was invalidated), valid bytes [M] -- so that a solve through the API reads its results back with one copy
// queued behind the iterations instead of a second round of launches, pageable copies and synchronizations (BASolver::solve, fused read-back)
__global__ void k_quality(View v, int buf_from_ctrl, double *err_sum /* [2] optional: sum, count */, double *pack) {
    __shared__ double scratch[2 * 16];
    const int N = v.dm.N, M = v.dm.M;
    const int cur = buf_from_ctrl ? v.ctrl->cur : 0;
    const double *fs = v.fs + (size_t)cur * N * 16, *rho = v.rho + (size_t)cur * M;
    double es = 0, en = 0;
    if (pack && blockIdx.x == 0)
        for (int e = threadIdx.x; e < 16 * N; e += blockDim.x) pack[e] = fs[e];
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < M; l += gridDim.x * blockDim.x) {
        const int a = v.lm_anchor[l];
        double rec_a[kFrameRec], Rwc[9], pwc[3], y0[3], x[3], t[3];
        frame_record(rec_a, fs + 16 * a, v.cam_ext + 7 * a, v.sic + 4 * a);
        m3_mul(Rwc, rec_a, rec_a + 12);
        m3_vec(t, rec_a, rec_a + 21);
        v3_add(pwc, rec_a + 9, t);
        const double inv = 1.0 / rho[l];
        v3_set(y0, v.lm_zref[2 * l] * inv, v.lm_zref[2 * l + 1] * inv, inv);
        m3_vec(x, Rwc, y0);
        v3_add(x, x, pwc); // Track::get_landmark_point (track.cpp:137-141)
        double q = 0, qn = 0;
        bool ok = true;
        const int o0 = v.lm_ptr[l], nobs = v.lm_ptr[l + 1] - o0;
        for (int k = -1; k < nobs; ++k) {
            const int f = k < 0 ? a : v.obs_frame[o0 + k];
            const double zu = k < 0 ? v.lm_zref[2 * l] : v.obs_z[2 * (size_t)(o0 + k)], zv = k < 0 ? v.lm_zref[2 * l + 1] : v.obs_z[2 * (size_t)(o0 + k) + 1];
            double rec[kFrameRec], Rf[9], pf[3], dd[3], y[3];
            frame_record(rec, fs + 16 * f, v.cam_ext + 7 * f, v.sic + 4 * f);
            m3_mul(Rf, rec, rec + 12);
            m3_vec(t, rec, rec + 21);
            v3_add(pf, rec + 9, t);
            v3_sub(dd, x, pf);
            m3_tvec(y, Rf, dd);
            if (!err_sum && (y[2] <= 1.0e-3 || y[2] > 50)) {
                ok = false;
                break;
            }
            const double *K = v.intr + 4 * f;
            const double du = (y[0] / y[2]) * K[0] + K[2] - (zu * K[0] + K[2]), dv = (y[1] / y[2]) * K[1] + K[3] - (zv * K[1] + K[3]);
            q += sqrt(du * du + dv * dv);
            qn += 1.0;
        }
        if (err_sum) {
            es += q, en += qn;
        } else {
            if (v.lm_valid) v.lm_valid[l] = ok ? 1 : 0;
            if (ok && v.lm_quality) v.lm_quality[l] = q / fmax(qn, 1.0);
            if (pack) {
                pack[16 * (size_t)N + l] = rho[l];
                pack[16 * (size_t)N + M + l] = ok ? q / fmax(qn, 1.0) : 0.0;
                reinterpret_cast<unsigned char *>(pack + 16 * (size_t)N + 2 * (size_t)M)[l] = ok ? 1 : 0;
            }
        }
    }
    if (err_sum) {
        double s[2] = {es, en};
        block_sum<2>(s, scratch);
        if (threadIdx.x == 0) {
            atomicAdd(&err_sum[0], s[0]);
            atomicAdd(&err_sum[1], s[1]);
        }
    }
}

// Lambda = S^T S, eta = S^T s of the marginalization prior (once per upload)
// One workgroup = one 16 x 16 tile of Lambda (or, in the last tile column, sixteen entries of eta): the two column blocks of S it needs
// go through LDS 160 rows at a time, coalesced, so that a thread's D-long sum reads LDS instead of making D / 15 dependent trips to L2
// (15 us per upload at D = 150, 94 us at D = 435 that way).  Every entry is still summed over the rows in ascending order.
__global__ void __launch_bounds__(256) k_prior_prep(const double *S, const double *s, int D, double *Lambda, double *eta, double *ST) {
    constexpr int kRows = 160; // (one pass for the 150 x 150 prior of a ten-frame window; 40 KB of LDS)
    __shared__ double Sa[kRows][16], Sb[kRows][16];
    const int tid = threadIdx.x, i = tid >> 4, j = tid & 15;
    const int nt = (D + 15) >> 4, ta = blockIdx.y, tb = blockIdx.x; // tb == nt: the eta column
    const int a = 16 * ta + i, b = 16 * tb + j;
    const bool is_eta = tb == nt;
    double acc = 0;
    for (int r0 = 0; r0 < D; r0 += kRows) {
        const int nr = D - r0 < kRows ? D - r0 : kRows;
        for (int rr = i; rr < nr; rr += 16) { // 16 rows per pass, 16 consecutive doubles (128 bytes) per row and block
            const size_t row = (size_t)(r0 + rr) * D;
            Sa[rr][j] = 16 * ta + j < D ? S[row + 16 * ta + j] : 0.0;
            Sb[rr][j] = is_eta ? (j == 0 ? s[r0 + rr] : 0.0) : (b < D ? S[row + b] : 0.0);
        }
        __syncthreads();
        for (int rr = 0; rr < nr; ++rr) acc += Sa[rr][i] * Sb[rr][j];
        __syncthreads();
    }
    if (a >= D) return;
    if (is_eta) {
        if (j == 0) eta[a] = acc;
    } else if (b < D) {
        Lambda[(size_t)a * D + b] = acc;
        ST[(size_t)a * D + b] = S[(size_t)b * D + a];
    }
}

// ------------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------------
// landmarks per chunk of the large-window role (ba_lin_tp.h): a multiple of four (the K of one MFMA), at most 64.  Two workgroups per CU (80 KB of LDS
// each: the phases of one overlap the latencies of the other) when that leaves room for the landmarks 256 factors belong to (`want`); else what fits in
// 150 KB
int tp_landmark_slots(const Dims &dm, int want) {
    const size_t common = (size_t)(dm.N * 16 + dm.N * kFrameRec + 160 + 16);
    auto fits = [&](int s, size_t bytes) { return (common + tp_lds_doubles(dm.N, dm.P6, s, dm.n_tasks)) * sizeof(double) <= bytes; };
    const int up = std::min(64, std::max(4, (want + 3) & ~3)), down = std::min(64, std::max(4, want & ~3));
    if (fits(up, 80 * 1024)) return up;
    if (fits(down, 80 * 1024)) return down; // (10 frames seen by all: 28 landmarks = 252 factors per chunk)
    int s = 64;
    while (s > 4 && !fits(s, 150 * 1024)) s -= 4;
    return s;
}

size_t linearize_lds_bytes(const Dims &dm) {
    const int N = dm.N;
    size_t common = (size_t)(N * 16 + N * kFrameRec + 160 + 16);
    const size_t slots = (size_t)dm.lm_slots;
    size_t lm = dm.lm_mm ? tp_lds_doubles(N, dm.P6, (dm.lm_slots + 3) & ~3, dm.n_tasks) : slots * (40 * N + 46) + slots + 2 * ((slots + 1) / 2) + 4;
    size_t pl = dm.n_plane > 0 ? (size_t)dm.plane_slots * (dm.P6 + 2) : 0; // (a role without workgroups needs no room)
    size_t pre = dm.use_inertial ? 16 + 450 + 450 + 16 + 225 : 0;
    size_t pri = dm.prior_n > 0 ? (size_t)dm.prior_n * (15 + 9) + 40 : 0;
    size_t role = lm;
    if (pl > role) role = pl;
    if (pre > role) role = pre;
    if (pri > role) role = pri;
    return (common + role) * sizeof(double);
}

int tiles_per_thread(const Dims &dm) { return (dm.n_tasks + kLinThreads - 1) / kLinThreads; }

template <int T, bool MM, int TW = TilesPerWave<T>::value>
static hipError_t launch_lin_TM(const View &v, hipStream_t st) {
    const int grid = v.dm.G_lm + v.dm.G_plane + v.dm.G_pre + v.dm.G_prior;
    const size_t lds = linearize_lds_bytes(v.dm);
    static size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linearize<T, MM, TW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linearize<T, MM, TW>), dim3(grid), dim3(kLinThreads), lds, st, v);
    return hipGetLastError();
}
template <int T>
static hipError_t launch_lin_T(const View &v, hipStream_t st) {
    if (!v.dm.lm_mm) return launch_lin_TM<T, false>(v, st);
    if (tp_tiles(v.dm.P6) > 4 * TilesPerWave<T>::value) return hipErrorInvalidValue; // (cannot happen: see TilesPerWave)
    if constexpr (T == 9) {
        if (tp_tiles(v.dm.P6) <= 80) return launch_lin_TM<9, true, 20>(v, st); // 28 .. 31 frames
    }
    return launch_lin_TM<T, true>(v, st);
}

hipError_t launch_linearize(const View &v, hipStream_t st) {
    const int T = tiles_per_thread(v.dm);
    if (T <= 1) return launch_lin_T<1>(v, st);
    if (T <= 2) return launch_lin_T<2>(v, st);
    if (T <= 4) return launch_lin_T<4>(v, st);
    if (T <= 6) return launch_lin_T<6>(v, st);
    return launch_lin_T<9>(v, st);
}

hipError_t launch_reduce(const View &v, hipStream_t st, int phase) {
    const size_t total = (size_t)v.dm.n_tasks * 9 + (size_t)kNumPoseVec * v.dm.P6 + kNumLinScal;
    const int nb_red = (int)((total + kRedElems - 1) / kRedElems);
    const int nb_img = (v.dm.use_img && phase != 1) ? (v.dm.img_sz + kRedElems * kRedGroups - 1) / (kRedElems * kRedGroups) : 0;
    hipLaunchKernelGGL(k_reduce, dim3(nb_red + nb_img), dim3(kRedElems * kRedGroups), 0, st, v, nb_red, phase);
    return hipGetLastError();
}

size_t dense_tile_doubles(const Dims &dm) {
    const size_t nbk = ((((size_t)dm.P + 7) & ~(size_t)7) + 16) >> 4;
    return nbk * (nbk + 1) / 2 * 256; // 16 x 16 tiles of the lower block triangle incl. the rhs row
}
// bytes of dynamic LDS of the register-resident form at row stride `ls` of Xs / Lf (k_dense's template parameter LS)
static size_t dense_lds_bytes_at(const Dims &dm, int ls) {
    const size_t nbk = ((((size_t)dm.P + 7) & ~(size_t)7) + 16) >> 4, LDV = nbk << 4;
    const size_t vec = (352 + 8 * LDV + (size_t)ls * LDV) * sizeof(double); // header, 8 vectors, Xs
    const size_t npan = (((size_t)dm.P + 7) & ~(size_t)7) / 8;
    const size_t lfull = (size_t)ls * (npan * LDV - 4 * npan * (npan - 1)); // finished panels of L (overlays the tile image)
    return std::max(dense_tile_doubles(dm), lfull) * sizeof(double) + vec;
}
// 8, or -- PVIO_HIP_DENSE_ROW_STRIDE=10, experiments only -- rows padded to 80 bytes where the LDS holds them.  Measured on the metric window
// (profiles/r5_ab_row_stride.txt): 13 964 iterations/s padded against 14 056 unpadded, i.e. the bank conflicts of the factor wave's row accesses
// that tools/ubench/wave_costs.hip shows in isolation (899 against 391 cycles per 12 ds_write_b128) are NOT on the panel loop's critical path.
int dense_row_stride(const Dims &dm) {
    static const int forced = std::getenv("PVIO_HIP_DENSE_ROW_STRIDE") ? std::atoi(std::getenv("PVIO_HIP_DENSE_ROW_STRIDE")) : 0;
    return (forced == 10 && dm.dense_la && dense_lds_bytes_at(dm, 10) <= 160 * 1024) ? 10 : 8;
}
size_t dense_lds_bytes(const Dims &dm, int *lds_matrix) {
    const size_t nbk = ((((size_t)dm.P + 7) & ~(size_t)7) + 16) >> 4, LDV = nbk << 4;
    *lds_matrix = (dense_lds_bytes_at(dm, 8) <= 160 * 1024 && LDV <= 176) ? 1 : 0;
    if (*lds_matrix) return dense_lds_bytes_at(dm, dense_row_stride(dm));
    return (352 + 8 * LDV + LDV * (size_t)(dense_panel_width((int)LDV) + 2)) * sizeof(double); // header, 8 vectors, LDS panel
}

hipError_t launch_dense(const View &v, hipStream_t st) {
    int lm;
    const size_t lds = dense_lds_bytes(v.dm, &lm);
    static size_t configured = 0, configured_g = 0;
    if (lm && lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dense<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dense<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dense<true, true, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    if (!lm && lds > configured_g) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dense<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured_g = lds;
    }
    const bool la = lm && v.dm.dense_la && v.dm.split_fin && v.dm.use_img;
    const int ls = la ? dense_row_stride(v.dm) : 8;
    static const bool say = std::getenv("PVIO_HIP_DEBUG_LAUNCH") != nullptr;
    static bool said = false;
    if (say && !said) said = true, std::fprintf(stderr, "launch_dense: lds matrix %d, look-ahead %d, split finalize %d, qvv in backsub %d, row stride %d, lds %zu\n", lm, v.dm.dense_la, v.dm.split_fin, v.dm.qvv_back, ls, lds);
    if (la && ls == 10) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense<true, true, 10>), dim3(1), dim3(kDenseThreads), lds, st, v);
    else if (la) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense<true, true>), dim3(1), dim3(kDenseThreads), lds, st, v);
    else if (lm) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense<true, false>), dim3(1), dim3(kDenseThreads), lds, st, v);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dense<false, false>), dim3(1), dim3(2 * kDenseThreads), lds, st, v);
    return hipGetLastError();
}

hipError_t launch_backsub(const View &v, hipStream_t st) {
    hipLaunchKernelGGL(k_backsub, dim3(v.dm.G_back + (v.dm.split_fin ? 1 : 0)), dim3(256), 0, st, v);
    return hipGetLastError();
}
hipError_t launch_back_reduce(const View &v, double *back_local, hipStream_t st) {
    hipLaunchKernelGGL(k_back_reduce, dim3(1), dim3(64), 0, st, v, v.dm.G_back, back_local);
    return hipGetLastError();
}
hipError_t launch_quality(const View &v, hipStream_t st, int buf_from_ctrl, double *err_sum, double *pack) {
    int grid = (v.dm.M + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_quality, dim3(grid), dim3(256), 0, st, v, buf_from_ctrl, err_sum, pack);
    return hipGetLastError();
}
// The reset in front of a solve -- accepted iterate <- initial state, user state <- initial state, control block <- its template -- as ONE
// launch (three device-to-device copies and a host-to-device copy of the control block cost four stream operations per solve before)
__global__ void __launch_bounds__(256) k_reset(View v, const double *fs_init, const double *rho_init, const Ctrl *tmpl) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const size_t nf = (size_t)v.dm.N * 16, M = (size_t)v.dm.M;
    for (size_t e = i; e < nf; e += stride) {
        const double x = fs_init[e];
        v.fs[e] = x, v.fs_user[e] = x;
    }
    for (size_t e = i; e < M; e += stride) v.rho[e] = rho_init[e];
    if (blockIdx.x == 0 && threadIdx.x < sizeof(Ctrl) / sizeof(double))
        reinterpret_cast<double *>(v.ctrl)[threadIdx.x] = reinterpret_cast<const double *>(tmpl)[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x < 24) v.cand_rec[threadIdx.x] = (threadIdx.x & 7) == 5 && threadIdx.x < 16 ? -2.0 : 0.0; // no candidate on record
}
hipError_t launch_reset(const View &v, const double *fs_init, const double *rho_init, const Ctrl *tmpl, hipStream_t st) {
    int grid = (int)(((size_t)v.dm.M + 255) / 256);
    grid = grid < 1 ? 1 : (grid > 512 ? 512 : grid);
    hipLaunchKernelGGL(k_reset, dim3(grid), dim3(256), 0, st, v, fs_init, rho_init, tmpl);
    return hipGetLastError();
}
// Eight result arrays of a marginalization pass -> one contiguous buffer (one D2H copy instead of eight; 32-bit words, so that the int32
// task table travels the same way as the FP64 arrays)
__global__ void __launch_bounds__(256) k_gather(GatherArgs a, uint32_t *dst) {
    const int seg = blockIdx.y;
    const uint32_t *src = static_cast<const uint32_t *>(a.src[seg]);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.words[seg]; i += gridDim.x * blockDim.x) dst[a.off[seg] + i] = src[i];
}
hipError_t launch_gather(const GatherArgs &a, void *dst, hipStream_t st) {
    hipLaunchKernelGGL(k_gather, dim3(32, 8), dim3(256), 0, st, a, static_cast<uint32_t *>(dst));
    return hipGetLastError();
}
hipError_t launch_prior_prep(const double *S, const double *s, int D, double *Lambda, double *eta, double *ST, hipStream_t st) {
    const int nt = (D + 15) / 16;
    hipLaunchKernelGGL(k_prior_prep, dim3(nt + 1, nt), dim3(256), 0, st, S, s, D, Lambda, eta, ST);
    return hipGetLastError();
}

} // namespace pvba
