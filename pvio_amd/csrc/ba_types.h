// ba_types.h -- device-side views of one bundle-adjustment window and the on-device solver state.
//
// HBM layout (all FP64 unless noted; everything is SoA and stays resident for the whole solve):
//   frames     fs[2][N][16]           two state buffers (accepted iterate / candidate), index ctrl->cur
//              fs_user[N][16]         the reference's "user state" (live bias read, preintegration_error_cost.h:57-58)
//   landmarks  rho[2][M]              inverse depths, accepted / candidate
//              lin[2] x {Hll,bl,Dl,ghl,gnl}[M], Wa[M][6], Wt[F][6]   two linearization sets (accepted / speculative)
//              cl[M]                  Jacobi column scale of the inverse-depth columns (computed once)
//   factors    obs_frame[F] i32, obs_z[F][2], obs_lm[F] i32  -- landmark-major CSR (lm_ptr[M+1])
//   partials   one row per workgroup of k_linearize: S tiles, 3 pose vectors, 8 scalars; summed by k_reduce
//   dense      Smat[(dN)^2], vectors of length dN  (d = 6 vision-only, 15 with IMU / prior)
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace pvba {

constexpr int kMaxFrames = 32;
constexpr int kLinThreads = 256;   // workgroup size of k_linearize
constexpr int kDenseThreads = 256;
constexpr int kNumLinScal = 8;     // cost, sum g^_l^2, |step_l|^2, |x_l|^2, max|b_l|, nonfinite, spare, spare
constexpr int kNumBackScal = 8;    // sum gn_l^2, sum g^_l gn_l, Qvv, Qvy, Qyy, Gy, spare, spare
constexpr int kNumPoseVec = 3;     // g_dir, rhs_schur, diagH_dir

enum Mode : int32_t {
    MODE_INIT = 0,       // evaluate + linearize the initial point (iteration 0)
    MODE_CANDIDATE = 1,  // form the dogleg step, evaluate + speculatively linearize the candidate
    MODE_RELIN = 2,      // re-linearize the accepted point (mu changed after a failed factorization / invalid step)
    MODE_DONE = 3,
    MODE_MARG = 4        // marginalize_frame: un-robustified J^T J of the victim's landmarks, Schur weight 1 / H_ll
};
enum LinResult : int32_t { LIN_NONE = 0, LIN_INIT = 1, LIN_CANDIDATE = 2, LIN_RELIN = 3, LIN_INVALID_STEP = 4 };

struct TraceRec { // mirrors pvio_ba_iteration
    int32_t iteration, step_is_valid, step_is_successful, reserved;
    double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius, mu;
};

// Control block: the whole trust-region state machine lives on the device so that a solve needs no host
// round trip per iteration (Ceres-1.14 TrustRegionMinimizer + DoglegStrategy semantics).
struct Ctrl {
    int32_t mode;            // what the next k_linearize does
    int32_t lin_result;      // what the last k_linearize did
    int32_t cur;             // state buffer of the accepted iterate
    int32_t lin;             // linearization set of the accepted iterate
    int32_t iter;            // trust-region iterations started so far
    int32_t num_success;
    int32_t invalid_steps;
    int32_t termination;
    int32_t done;
    int32_t reuse;           // DoglegStrategy::reuse_
    int32_t solve_ok;        // k_dense produced a Gauss-Newton step -> k_backsub must run
    int32_t scaling_ready;   // Jacobi scaling has been computed (iteration 0)
    int32_t trace_len, trace_cap;
    int32_t retry_relin;     // RELIN triggered by a mu escalation inside one iteration (not a new iteration)
    int32_t marg_victim;     // MODE_MARG: frame being marginalized
    double radius, mu;
    double x_cost, x_norm2_pose, x_norm2_lm, grad_max;
    double initial_cost;
    // scalars of the current linearization (pose parts written by k_dense, landmark parts summed from k_backsub)
    double pose_g2, pose_gn2, pose_gdot, pose_qvv, pose_qvy, pose_qyy, pose_gy;
    double lm_g2;            // sum over landmarks of g^_l^2 of the accepted linearization
    // current trust-region step (written by block 0 of k_linearize in MODE_CANDIDATE)
    double ca, cb, dogleg_step_norm, model_cost_change;
    double cand_step2_pose, cand_norm2_pose;
    // bookkeeping for the record of the iteration in flight
    double it_cost, it_cost_change, it_step_norm, it_rel;
    int32_t it_valid, it_success;
    int32_t dbg_fail_left, dbg_invalid_left; // fault injection (pvio_hip_opts::debug_*), counted down by k_dense
    // split finalize (Dims::split_fin): what k_dense leaves for the extra workgroup of k_backsub to finish, off the critical path
    // -- the gradient max-norm of the linearization just accepted (trace record, gradient-tolerance exit), the state-updating
    // callback's copies and the trace states
    int32_t fin_flags;       // kFinTrace | kFinAccepted | kFinFirst; 0 = nothing pending
    int32_t fin_trace_slot;  // trace record of the iteration (or -1)
    double fin_lm_gmax;      // max |b_l| of the accepted linearization (landmark part of the gradient max-norm)
    // Dims::img_scaled: the tile image k_dense factored in this slot was the SCALED, negated system -(C S C) (k_reduce applies the
    // Jacobi scaling once it exists); k_backsub forms v^T S v from the same image and has to know which one it is
    int32_t img_scaled_now;
    int32_t cand_repeats;    // Dims::reuse_cand: candidate evaluations short-circuited in this solve (written by workgroup 0 of k_linearize)
};
enum : int32_t { kFinTrace = 1, kFinAccepted = 2, kFinFirst = 4 };

struct Dims {
    int32_t N, M, F;
    int32_t d;          // tangent dims per frame in the dense system: 6 or 15
    int32_t P;          // d * N
    int32_t P6;         // 6 * N (pose part the landmark tiles live on)
    int32_t n_tasks;    // 3x3 tile tasks: 4 * N (N + 1) / 2
    int32_t n_chunks;   // landmark chunks
    int32_t lm_slots;   // max landmarks per chunk (LDS budget)
    int32_t n_plane, n_plane_chunks, plane_slots;
    int32_t prior_n;
    int32_t use_inertial;
    int32_t max_iter;
    int32_t G_lm, G_plane, G_pre, G_prior; // workgroups per role in k_linearize
    int32_t G_back;
    int32_t n_back_rows;  // rows of back partials k_linearize must sum (G_back, or 1 after an all-reduce)
    int32_t world, rank;
    int32_t fuse_backsub; // landmark back-substitution inside k_dense (small windows, single GPU)
    int32_t use_img;      // k_reduce also assembles the reduced system as a tile image the dense kernel loads straight into registers
    int32_t img_sz;       // doubles in the image (tiles * 256)
    int32_t n_rot;        // rotation priors (RotationPriorFactor, no reference counterpart): evaluated by workgroup 0 of k_linearize
    int32_t split_fin;    // single GPU, separate k_backsub launch: k_dense defers the gradient max-norm, the state / trace copies and
                          // (qvv_back) the pose part of v^T H v to k_backsub, where they run beside the landmark back-substitution
    int32_t qvv_back;     // v^T S v from the tile image in k_backsub (partials in back_part[.][6]) instead of k_dense
    int32_t dense_la;     // register-resident factorization in its look-ahead form (wave 0 factors panel p + 1 while waves 1..3 apply panel p)
    int32_t img_scaled;   // look-ahead form: once the Jacobi scaling exists (after the first factoring launch of a solve) k_reduce writes the
                          // image as -(C S C), the form the accumulators hold, and k_dense loads it without touching it (Ctrl::img_scaled_now)
    int32_t reuse_cand;   // a candidate that is bit-identical to the one just rejected is not evaluated again (View::cand_rec)
    int32_t lm_mm;        // landmark workgroups accumulate the Schur complement as 16x16 f64 MFMA tiles and walk a contiguous chunk range
};

struct View { // passed by value to every kernel
    Dims dm;
    Ctrl *ctrl;
    // static problem
    const uint8_t *frame_fixed;   // [N]
    const uint8_t *pose_active;   // [N] pose block is free and referenced
    const uint8_t *motion_active; // [N]
    const double *cam_ext, *imu_ext, *sic, *intr;
    const int32_t *lm_anchor, *lm_ptr, *obs_frame, *obs_lm;
    const double *lm_zref, *obs_z;
    const double *lm_mult; // [M] multiplicity of the landmark's residual blocks as a double, or null (every block once)
    const int32_t *chunk_lm;      // [n_chunks+1] landmark ranges
    // Dims::lm_mm (large windows, ba_lin_tp.h): every chunk's factors sorted by target frame -- chunk_perm[o] = chunk-relative factor index of the
    // o-th entry of the chunk's sorted list, chunk_tptr[chunk][t] = where target t's entries start in it
    const int32_t *chunk_geo;     // [n_chunks][8]: first landmark, landmarks, first factor, factors, anchor frame, 0, 0, 0
    const int32_t *chunk_tptr;    // [n_chunks][N + 1]
    const uint8_t *chunk_perm;    // [F]
    const uint32_t *lm_seen;      // [M] bit f: the landmark is observed in frame f (large windows: the U row's zero blocks)
    const int32_t *tp_tile_dst;   // [tiles][64 lanes][4]: where entry r of lane's accumulator of a Schur tile goes in the partial row: el << 24 | task, -1: nowhere
    const int32_t *task_desc;     // [n_tasks] packed fi | fj<<8 | si<<16 | sj<<17
    const uint8_t *pre_valid;     // [N]
    const double *pre_delta, *pre_U, *pre_jac;
    const int32_t *prior_frames;
    const int32_t *prior_slot;    // [N] slot of the frame in the prior or -1
    const double *prior_S, *prior_s, *prior_lin, *prior_Lambda, *prior_eta, *prior_ST; // Lambda = S^T S, eta = S^T s, ST = S^T
    const int32_t *plane_ptr, *plane_frame, *plane_chunk; // CSR + chunk ranges
    const double *plane_z, *plane_normal, *plane_dist;
    double plane_sic;
    const int32_t *rot_slot;      // [N] rotation prior of the frame or -1
    const double *rot_q0, *rot_W; // [n_rot][4], [n_rot][9]
    double *rot_H, *rot_g, *rot_cost; // [N][9], [N][3], [1]: J^T J, J^T r per frame; 1/2 sum |r|^2 (zero without a prior)
    // state
    double *fs;        // [2][N][16]
    double *fs_user;   // [N][16]
    double *bias0_lin; // [N][6] live biases the accepted linearization was evaluated with
    double *rho;       // [2][M]
    double *cl;        // [M]
    double *Hll, *bl, *Dl, *ghl, *gnl, *Wa, *Wt; // [2] sets each
    // partials and reduced quantities
    double *part_S;    // [G][n_tasks*9]
    double *part_vec;  // [G][3][P6]
    double *part_scal; // [G][8]
    double *red;       // [n_tasks*9 + 3*P6 + 8]  (the all-reduce payload)
    double *back_part; // [G_back][8]
    double *back_red;  // [8]
    double *pre_H, *pre_g, *pre_cost;       // [N][900], [N][30], [N]
    double *prior_H, *prior_g, *prior_cost; // [(15n)^2], [15n], [1]
    double *prior_gd;                       // [N][30] by FRAME: prior gradient [15], diagonal of the prior's H [15]; 0 where no prior
    // dense system
    double *Smat;      // tile image of the reduced system (systems too large for LDS)
    double *img;       // the unscaled reduced system as a tile image (lower block triangle, MFMA accumulator order), written by
                       // k_reduce; zero wherever nothing is ever written
    double *cp, *Dp, *gtot, *ghp, *vstep, *ystep; // [P] each
    // Dims::reuse_cand: [2][8] the candidate of the last two trust-region iterations by iteration parity -- ca, cb, mu, lin, cur, iteration, evaluated,
    // unused -- written by workgroup 0 of k_linearize, read by every workgroup of the NEXT launch; [16] = this slot repeats the last candidate
    // INVARIANT the short-circuit rests on: between the k_linearize launch that evaluated a candidate and the launch that finds it again,
    // nothing writes what the skipped evaluation would have produced and the rest of the slot still reads -- the partial rows part_*, `red`, the tile
    // image `img`, the candidate buffers fs[1 - cur] / rho[1 - cur] and the cost buffers pre_cost / prior_cost / rot_cost of the speculative
    // linearization set.  Holds because a rejected step makes k_dense return before it builds anything (need_build false: it only READS red / img) and
    // k_backsub only writes step vectors and back_part.  Does NOT hold on landmark shards, whose all-reduce sums `red` in place -- upload() turns the
    // mode off there.  Held by tests: the whole case matrix runs with the mode on against the oracle per iteration (test_emu_ba.py, test_gpu_ba.py).
    double *cand_rec;
    double *cpl, *vraw; // [P] each: Jacobi scale with 0 on inactive coordinates (img_scaled: read by k_reduce); v = g^ / D before C is applied
    // trace
    TraceRec *trace;
    double *trace_states;
    double *lm_quality;
    uint8_t *lm_valid;
    long long *dbg; // optional phase timestamps (profiling entry point): [kernel 0..3][32] shader-clock ticks since the launch's start, block 0 / thread 0
    int dbg_sel;    // -1: every stamp site stores; k >= 0: only site k does (one store per launch: the stamps' own s_waitcnt / store do not add up)
};

inline int lin_set_stride_M() { return 1; }

// ---- LDS geometry of the large-window landmark role (ba_lin_tp.h), shared by the kernel, the chunker of upload() and the launch ----
#if defined(__HIPCC__)
#define PVBA_HD __host__ __device__
#else
#define PVBA_HD
#endif
constexpr int kTpXCols = 14;   // (row 0, row 1) pairs of a factor row: 0-5 Jt, 6-11 Jr, 12 r, 13 Jd
constexpr int kTpLmr = 8;      // per-landmark results: 0 H_ll (then the Schur weight w)  1 b_l  2..7 W_a
// LDS has 32 four-byte banks.  (PVBA_TP_PAD=0 builds the round's first layout for the A/B: X rows of 14 pairs, U rows of 16 k doubles, LMR rows of 8.)
//  * X rows of 14 pairs = 56 dwords put a given column of ANY row on one of four bank offsets: half the banks are never used by a column, every access where
//    lanes read one column of different rows takes twice the cycles.  X is therefore kept as TWO arrays of 7 pairs per row (28 dwords: eight offsets, all
//    banks) -- the same bytes: [Jt 0-5, r] and [Jr 0-5, Jd]; a direct task's column triplets never straddle the two.  tp_xcol(c) = pair offset of column c
//    from the row's base X2 + row * kTpXLd.
//  * a U row of 16 k doubles and an LMR row of 8 put every row on the SAME banks (phase P's lane-per-row stores serialize completely): one double of padding.
// The arithmetic and its order are untouched.
#ifndef PVBA_TP_PAD
#define PVBA_TP_PAD 1
#endif
constexpr int kTpXLd = PVBA_TP_PAD ? kTpXCols / 2 : kTpXCols; // pairs per X row (of one half)
constexpr int kTpXHalf = kLinThreads * (kTpXCols / 2);        // pair offset of the second half
PVBA_HD constexpr int tp_xcol(int c) { return PVBA_TP_PAD ? (c < 6 ? c : (c < 12 ? kTpXHalf + c - 6 : (c == 12 ? 6 : kTpXHalf + 6))) : c; }
constexpr int kTpLmrLd = kTpLmr + PVBA_TP_PAD;   // doubles per LMR row
constexpr int kTpDirTasks = 9; // per target: TT00 TT01 TT11 | TR00 TR01 TR10 TR11 | g[0:3] g[3:6]
constexpr int kTpAnchTasks = 5; // per anchor: RR00 RR01 RR11 | gR[0:3] gR[3:6]

// U rows: 6 N pose columns, then b_l in column 6 N (the row 6 N of the SYRK is then the Schur right-hand side), padded to whole 16 x 16 tiles
PVBA_HD inline int tp_u_stride(int P6) { return ((P6 + 1 + 15) >> 4) << 4; }
PVBA_HD inline int tp_u_ld(int P6) { return tp_u_stride(P6) + PVBA_TP_PAD; } // doubles per U row
PVBA_HD inline int tp_tiles(int P6) { const int nbt = tp_u_stride(P6) >> 4; return (nbt * (nbt + 1)) >> 1; }
// doubles of LDS the role needs behind the common part: X [256][14 pairs] | U [S][US] | LMR [S][8]; a flush reuses them as its stage
PVBA_HD inline size_t tp_work_doubles(int N, int P6, int slots, int n_tasks) {
    const size_t S = (size_t)slots, US = (size_t)tp_u_ld(P6);
    size_t work = (size_t)kLinThreads * 2 * kTpXCols + S * US + S * kTpLmrLd;
    const size_t flush = (size_t)2 * n_tasks + 2 * (size_t)kLinThreads * 9 + 64; // a flush: >= two elements of the partial row per pass + the tasks' sums
    return work < flush ? flush : work;
}
PVBA_HD inline size_t tp_lds_doubles(int N, int P6, int slots, int n_tasks) {
    const size_t S = (size_t)slots;
    // tables: rho_eval[S] cl_tab[S] | vg_acc[P6] vdiag_acc[P6] | ints: active[S], seen[S], fptr[S + 1], tptr[kMaxFrames + 1] | bytes: perm[256]
    return tp_work_doubles(N, P6, slots, n_tasks) + 2 * S + 2 * (size_t)P6 + (3 * S + 1 + (kMaxFrames + 1) + 1) / 2 + kLinThreads / 8 + 8;
}

} // namespace pvba
