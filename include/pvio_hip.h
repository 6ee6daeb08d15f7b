/* pvio_hip.h -- C ABI of the MI355X-native PVIO back-end (bundle adjustment + KLT).
 *
 * This is the drop-in boundary (seam B3).  Everything above it is host
 * C++ that mirrors the reference's own seams:
 *   - pvio::BundleAdjustor::{solve, marginalize_frame, compute_reprojection_error}
 *       reference: pvio/src/pvio/estimation/bundle_adjustor.h:29-42 (impl bundle_adjustor.cpp:63-599)
 *   - pvio::Image::{preprocess, track_keypoints}
 *       reference: pvio/include/pvio/pvio.h:114-133 (impl pvio-extra/src/pvio/extra/opencv_image.cpp:88-160)
 *   - pvio::PreIntegrator::integrate
 *       reference: pvio/src/pvio/estimation/preintegrator.cpp:84-100
 * Everything below it is hand-written HIP for gfx950.
 *
 * Conventions
 *   - plain C, POD structs, fixed-width ints, no size_t, no ownership transfer: every host buffer is
 *     owned by the caller and only read/written during the call; device memory is owned by the ctx.
 *   - all BA data is FP64.  Dense matrices are ROW-major (the adapter transposes Eigen's col-major).
 *   - quaternions are stored x,y,z,w (Eigen coeffs() order, bundle_adjustor.cpp:77).
 *   - a frame state is 16 doubles: q[4] p[3] v[3] bg[3] ba[3]; the tangent ("error state") order is
 *     ES_Q=0, ES_P=3, ES_V=6, ES_BG=9, ES_BA=12 (estimation/state.h:29-36).
 *   - every function returns 0 on success or a negative pvio_status; nothing throws or aborts.
 *   - one ctx per process and per GPU; calls on one ctx are serialized by the caller (this is the
 *     reference's threading contract: BA is never called concurrently with itself).
 */
#ifndef PVIO_HIP_H
#define PVIO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVIO_ES_SIZE 15
#define PVIO_FRAME_STATE_DIM 16
#define PVIO_MAX_FRAMES 32
#define PVIO_KLT_LEVELS 4 /* maxLevel 3 -> 4 levels, opencv_image.cpp:103 with level_num()==3 */

typedef enum pvio_status {
    PVIO_OK = 0,
    PVIO_ERR_INVALID_ARGUMENT = -1,
    PVIO_ERR_NO_DEVICE = -2,
    PVIO_ERR_HIP = -3,
    PVIO_ERR_OUT_OF_MEMORY = -4,
    PVIO_ERR_UNSUPPORTED = -5,
    PVIO_ERR_COMM = -6
} pvio_status;

/* Ceres TerminationType values the reference's `IsSolutionUsable()` looks at (bundle_adjustor.cpp:298). */
typedef enum pvio_termination {
    PVIO_TERM_CONVERGENCE = 0,
    PVIO_TERM_NO_CONVERGENCE = 1,
    PVIO_TERM_FAILURE = 2
} pvio_termination;

typedef struct pvio_hip_ctx pvio_hip_ctx;
typedef struct pvio_hip_image pvio_hip_image;
typedef struct pvio_hip_undistort pvio_hip_undistort;

/* Layout version of the structs below.  pvio_hip_opts is copied BY VALUE by pvio_hip_create and has grown a trailing field since 0.1
 * (reuse_identical_candidates): a caller built against an older header would have 4 bytes past its struct read as that field.  A caller
 * therefore checks pvio_hip_abi_version() == PVIO_HIP_ABI_VERSION once before the first pvio_hip_create (pvio_amd/host/bundle_adjustor.cpp
 * does).  Bumped whenever a struct of this header changes size or field order. */
#define PVIO_HIP_ABI_VERSION 2

typedef struct pvio_hip_opts {
    int32_t device;          /* HIP device ordinal */
    int32_t rank;            /* landmark-shard index of this process (0 when single GPU) */
    int32_t world_size;      /* number of landmark shards / processes (1 when single GPU) */
    int32_t use_graph;       /* 1: replay the per-iteration kernel sequence from a hipGraph */
    /* fault injection for tests of the rarely taken solver paths (0 in production): the first N Cholesky factorizations
     * of a solve are treated as failed (-> mu escalation, re-linearization), the first N trust-region steps as invalid
     * (-> HandleInvalidStep; five in a row = FAILURE) */
    int32_t debug_fail_factorizations;
    int32_t debug_invalid_steps;
    /* the landmark role of k_linearize: 0 = choose by window size (register 3x3 tiles on the FP64 VALU, a landmark chunk per
     * workgroup, below 40 000 reprojection factors; from there on the large-window role -- chunks of 256 factors walked by
     * workgroups that keep their accumulators, Schur complement on 16x16 f64 MFMA tiles), 1 = always the register tiles,
     * 2 = always the large-window role (tests; a window without landmarks has nothing for it to walk and takes the register
     * tiles' empty walk).  Results agree within rounding (another summation order). */
    int32_t linearize_mode;
    /* tests only: run the landmark-sharded code path (eager launches, the all-reduces through the communicator given to
     * pvio_hip_comm_init, the reduced system assembled from the all-reduced buffer) even with world_size == 1 -- a one-rank
     * RCCL communicator exercises the collectives on a single-GPU box.  0 in production. */
    int32_t debug_force_sharded;
    /* 1: a trust-region candidate that is bit-identical to the one the previous iteration evaluated and rejected is not evaluated again.
     * Ceres does evaluate it again -- after a rejected step DoglegStrategy keeps its Gauss-Newton step and only halves the radius, so while
     * |step| <= radius the next candidate IS the rejected one -- and with the reference's live-bias read (preintegration_error_cost.h:57-58)
     * such runs are the rule: the keyframe solves of a VIO sequence spend most of their ten iterations re-evaluating one rejected point.
     * The iterations, their records and every result are unchanged (same cost, same decision, same radius update); only the repeated
     * evaluation is skipped.  0 (default): every candidate is evaluated, like the reference -- what bench.py's headline measures.
     * Ignored on landmark shards. */
    int32_t reuse_identical_candidates;
} pvio_hip_opts;

/* ------------------------------------------------------------------------------------------------
 * Flat bundle-adjustment problem.  Replaces the ceres::Problem assembled at bundle_adjustor.cpp:75-242.
 * The adapter (pvio_amd/host/bundle_adjustor.cpp) fills it from the Map in the reference's block order.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pvio_ba_problem {
    int32_t n_frames;              /* N: frames in the window, Map index order                         */
    int32_t n_landmarks;           /* M: inverse-depth parameter blocks (VALID non-PLANE tracks, :91-103)*/
    int32_t n_obs;                 /* F: reprojection residual blocks = non-anchor observations (:142-161)*/
    int32_t use_inertial;          /* solve(..., use_inertial) (:63)                                     */

    /* per frame */
    const uint8_t *frame_fixed;    /* [N]   FF_FIX_POSE: q,p blocks constant (:79-82)                   */
    const double *cam_extrinsic;   /* [N][7] frame->camera: q_cs(xyzw) p_cs                              */
    const double *imu_extrinsic;   /* [N][7] frame->imu                                                  */
    const double *sqrt_inv_cov;    /* [N][4] frame->sqrt_inv_cov, 2x2 row-major (core.cpp:114-116)       */
    const double *intrinsics;      /* [N][4] fx fy cx cy of frame->K (quality pass, :277-296)            */

    /* landmarks, CSR over their non-anchor observations; anchor = track->first_keypoint()              */
    const int32_t *lm_anchor_frame;/* [M]                                                                */
    const double *lm_anchor_z;     /* [M][2] normalized keypoint in the anchor frame                     */
    const int32_t *lm_obs_ptr;     /* [M+1]                                                              */
    const int32_t *obs_frame;      /* [F]   target frame index                                           */
    const double *obs_z;           /* [F][2] normalized keypoint in the target frame                     */

    /* IMU pre-integration factors; entry j couples frame j-1 -> j; entry 0 unused (:220-242)            */
    const uint8_t *preint_valid;   /* [N]                                                                */
    const double *preint_delta;    /* [N][11] dt, dq(xyzw), dp, dv (PreIntegrator::Delta)                */
    const double *preint_sqrt_inv_cov; /* [N][225] 15x15 row-major                                       */
    const double *preint_jacobian; /* [N][45] dq_dbg dp_dbg dp_dba dv_dbg dv_dba, 3x3 row-major each     */

    /* marginalization prior (:126-139); prior_n == 0 when Map has no marginalization factor             */
    int32_t prior_n;               /* n: frames the prior relates                                        */
    int32_t n_plane_factors;       /* P: AugmentedPlaneDistanceErrorCost blocks (:180-195)               */
    const int32_t *prior_frames;   /* [n] window index of each related frame                             */
    const double *prior_S;         /* [15n][15n] row-major sqrt-information matrix                       */
    const double *prior_s;         /* [15n]     sqrt-information vector                                   */
    const double *prior_lin_state; /* [n][16]   pose_0/motion_0 captured at construction                 */

    /* plane-distance factors, CSR over ALL observations of the track (anchor included)                  */
    const int32_t *plane_obs_ptr;  /* [P+1]                                                              */
    const int32_t *plane_obs_frame;/* [..]                                                               */
    const double *plane_obs_z;     /* [..][2]                                                            */
    const double *plane_normal;    /* [P][3] constant block                                              */
    const double *plane_distance;  /* [P]    constant block                                              */
    double plane_sqrt_inv_cov;     /* sqrt(1/config->plane_distance_cov()) (:181)                        */

    /* solver options (solver_options.h:26-33); everything else is a Ceres 1.14 default                   */
    int32_t max_iterations;        /* config->solver_iteration_limit()                                   */
    int32_t reserved0;
    double max_solver_time;        /* config->solver_time_limit() [s]; checked between replays of the iteration
                                      graph (one replay = every iteration of an ordinary solve), <= 0: no limit     */

    /* Rotation priors -- the `RotationPriorFactor` BASELINE.json's north_star names.  NO REFERENCE COUNTERPART: the class
     * does not exist in the reference @ v0 (name-mapping note).  Defined as that note prescribes, as
     * the rotation rows of the marginalization factor (marginalization_error_cost.h:65,77) with a 3x3 sqrt-information:
     *   r = W log(q0^-1 (x) q_body),   dr/dtheta = W Jr^-1(log(q0^-1 (x) q_body)),   no loss function.
     * At most one per frame; a prior on an FF_FIX_POSE frame is a constant and is left out (Ceres folds such blocks into
     * fixed_cost).  marginalize_frame folds the victim's rotation prior into the new marginalization prior. */
    int32_t n_rot_priors;          /* R                                                                  */
    int32_t reserved1;
    const int32_t *rot_prior_frame;    /* [R]    window index                                            */
    const double *rot_prior_q0;        /* [R][4] x y z w                                                 */
    const double *rot_prior_sqrt_info; /* [R][9] W, row-major                                            */

    /* Duplicate residual blocks (bundle_adjustor.cpp:165-179).  The reference adds the reprojection blocks of every track of a
     * plane with fewer than 20 tracks a second time -- once more per such plane the track sits in -- on top of the blocks the track
     * got at :142-161 as a VALID non-PLANE track.  Ceres sums duplicate blocks: a track listed m times weighs m times in the cost,
     * the gradient and the normal equations (each copy robustified on its own, so the factor applies AFTER the loss function).
     * lm_multiplicity[l] = m >= 1 for landmark l; NULL = every landmark once.  Not used by pvio_hip_ba_marginalize
     * (marginalize_frame adds each block once, :455-510) nor by the quality pass. */
    const int32_t *lm_multiplicity;    /* [M] or NULL                                                    */
} pvio_ba_problem;

/* In/out states, updated in place exactly like Frame::pose/motion and Track::landmark.inv_depth. */
typedef struct pvio_ba_state {
    double *frame_state;           /* [N][16] */
    double *lm_inv_depth;          /* [M]     */
    /* post-solve pass (:277-296); either may be NULL */
    double *lm_quality;            /* [M] mean pixel reprojection error; untouched when the landmark is invalidated */
    uint8_t *lm_valid;             /* [M] 0 when the depth gate (z<=1e-3 or z>50) cleared TF_VALID */
} pvio_ba_state;

typedef struct pvio_ba_iteration {
    int32_t iteration;
    int32_t step_is_valid;
    int32_t step_is_successful;
    int32_t reserved;
    double cost;                   /* cost of the accepted point after this iteration (candidate cost if rejected) */
    double cost_change;
    double gradient_max_norm;
    double step_norm;
    double relative_decrease;
    double trust_region_radius;    /* radius after the iteration */
    double mu;                     /* Dogleg regularization multiplier after the iteration */
} pvio_ba_iteration;

typedef struct pvio_ba_summary {
    int32_t termination;           /* pvio_termination */
    int32_t is_usable;             /* == ceres::Solver::Summary::IsSolutionUsable() */
    int32_t num_iterations;        /* trust-region iterations executed (iteration 0 excluded) */
    int32_t num_successful_steps;
    double initial_cost;
    double final_cost;
    double solve_seconds;          /* wall time inside the call */
    double device_seconds;         /* hipEvent time of the kernel sequence only */
    int32_t trace_capacity;        /* in: entries available in `trace` (0 allowed) */
    int32_t trace_len;             /* out */
    pvio_ba_iteration *trace;      /* optional per-iteration records (iteration 0 first) */
    double *trace_states;          /* optional [trace_capacity][N*16+M] states after each iteration */
} pvio_ba_summary;

/* New marginalization prior produced by marginalize_frame (:583-598). */
typedef struct pvio_ba_prior {
    int32_t n;                     /* out: number of remaining frames (N-1) */
    int32_t reserved;
    double *S;                     /* [15n][15n] row-major, caller-allocated for n = N-1 */
    double *s;                     /* [15n] */
    double *info_matrix;           /* optional [15n][15n]: the Schur complement before the eigendecomposition */
    double *info_vector;           /* optional [15n] */
} pvio_ba_prior;

int32_t pvio_hip_create(const pvio_hip_opts *opts, pvio_hip_ctx **out);
void pvio_hip_destroy(pvio_hip_ctx *ctx);
const char *pvio_hip_last_error(const pvio_hip_ctx *ctx);
const char *pvio_hip_version(void);
/* PVIO_HIP_ABI_VERSION of the header the LIBRARY was built against */
int32_t pvio_hip_abi_version(void);

/* replaces BundleAdjustor::solve (bundle_adjustor.cpp:63-299) */
int32_t pvio_hip_ba_solve(pvio_hip_ctx *ctx, const pvio_ba_problem *problem, pvio_ba_state *state,
                          pvio_ba_summary *summary);

/* replaces BundleAdjustor::marginalize_frame (bundle_adjustor.cpp:348-599) */
int32_t pvio_hip_ba_marginalize(pvio_hip_ctx *ctx, const pvio_ba_problem *problem,
                                const pvio_ba_state *state, int32_t victim, pvio_ba_prior *out);

/* replaces BundleAdjustor::compute_reprojection_error (bundle_adjustor.cpp:321-336) */
int32_t pvio_hip_ba_reprojection_error(pvio_hip_ctx *ctx, const pvio_ba_problem *problem,
                                       const pvio_ba_state *state, double *mean_pixel_error);

/* Device-resident variant used by bench.py / multi-GPU: upload once, solve repeatedly from the same
 * initial state without re-uploading (inputs resident in HBM when the timed region starts). */
int32_t pvio_hip_ba_upload(pvio_hip_ctx *ctx, const pvio_ba_problem *problem, const pvio_ba_state *state);
int32_t pvio_hip_ba_solve_resident(pvio_hip_ctx *ctx, pvio_ba_summary *summary);
int32_t pvio_hip_ba_download(pvio_hip_ctx *ctx, pvio_ba_state *state);

/* Per-kernel timing of one resident solve (bench.py roofline leg): the same kernels, launched eagerly one slot at a
 * time with hipEvents on the solver's own stream around every launch.  Index: 0 linearize, 1 reduce, 2 dense, 3 backsub. */
typedef struct pvio_ba_kernel_times {
    double total_ms[4];
    int32_t launches[4];
    int64_t phase_ticks[4][32]; /* shader-clock timestamps of the LAST working slot, block 0 (kernel phase breakdown) */
    /* landmark-sharded solves: the two exchange steps of an iteration, same events.  [0] all-reduce of the reduced system (+ the
     * k_reduce pass that turns it into the tile image), [1] the 8-double all-reduce behind the back-substitution (+ k_back_reduce).
     * Zero on a single GPU. */
    double comm_ms[2];
    int32_t comm_launches[2];
} pvio_ba_kernel_times;
int32_t pvio_hip_ba_profile_resident(pvio_hip_ctx *ctx, pvio_ba_summary *summary, pvio_ba_kernel_times *times);
/* candidate evaluations the last solve short-circuited (pvio_hip_opts::reuse_identical_candidates; 0 when the option is off) */
int32_t pvio_hip_ba_last_candidate_repeats(const pvio_hip_ctx *ctx);
/* diagnostics: how many solves of this context ran as a replay of the captured slot graph (the others launched eagerly: the first solve of
 * an upload, contexts created with use_graph = 0, a landmark shard whose capture failed) */
int32_t pvio_hip_ba_graph_replays(const pvio_hip_ctx *ctx);

/* multi-GPU (landmark shards, one process per GPU): RCCL communicator bootstrap.
 * rank 0 creates the 128-byte unique id, the launcher broadcasts it (torch.distributed), every rank inits. */
int32_t pvio_hip_comm_unique_id(uint8_t id[128]);
int32_t pvio_hip_comm_init(pvio_hip_ctx *ctx, const uint8_t id[128], int32_t rank, int32_t world_size);

/* replaces PreIntegrator::integrate(t, bg, ba, true, true) (preintegrator.cpp:84-100); host-side, serial. */
typedef struct pvio_imu_noise {
    double cov_w[9], cov_a[9], cov_bg[9], cov_ba[9]; /* 3x3 row-major, continuous-time */
} pvio_imu_noise;
int32_t pvio_preintegrate(int32_t n_samples, const double *imu_t, const double *imu_w, const double *imu_a,
                          double t_end, const double bg[3], const double ba[3], const pvio_imu_noise *noise,
                          double delta[11], double cov[225], double sqrt_inv_cov[225], double jacobian[45]);

/* ------------------------------------------------------------------------------------------------
 * KLT front end.  Replaces OpenCvImage::preprocess / track_keypoints.
 * ---------------------------------------------------------------------------------------------- */
/* upload + CLAHE(6.0, 8x8) + 4-level pyramid + Scharr derivatives, all on device (opencv_image.cpp:138-145) */
int32_t pvio_hip_image_create(pvio_hip_ctx *ctx, const uint8_t *pixels, int32_t width, int32_t height,
                              int32_t stride, int32_t apply_clahe, pvio_hip_image **out);
void pvio_hip_image_release(pvio_hip_ctx *ctx, pvio_hip_image *img);
/* Image undistortion in front of the pyramid.  Replaces cv::undistort(img, K, dist) of the EuRoC reader
 * (pvio-pc/src/euroc_dataset_reader.cpp:72-75) and ImageUndistorter::undistort_image = cv::remap(map_x, map_y, INTER_LINEAR,
 * BORDER_CONSTANT) of the TUM-VI reader (pvio-extra/include/pvio/extra/image_undistorter.h:44-46, tum_dataset_reader.cpp:81).
 * The maps are what those calls hold internally (OpenCV's fixed-point pair): map_xy [h][w][2] int16 = integer source
 * position (x, y), map_frac [h][w] uint16 = (fy << 5) | fx with 5-bit fractions.  The host side builds them once per camera
 * (pvio_amd/host/undistort_maps.h); they stay resident, every frame is uploaded distorted and remapped on the device. */
int32_t pvio_hip_undistort_create(pvio_hip_ctx *ctx, const int16_t *map_xy, const uint16_t *map_frac, int32_t width,
                                  int32_t height, pvio_hip_undistort **out);
void pvio_hip_undistort_release(pvio_hip_ctx *ctx, pvio_hip_undistort *ud);
/* pvio_hip_image_create on the undistorted pixels: (width, height, stride) describe the DISTORTED source, the pyramid
 * has the map's size.  With apply_clahe = 0 level 0 is the remap output itself (pvio_hip_image_download_level). */
int32_t pvio_hip_image_create_undistorted(pvio_hip_ctx *ctx, const pvio_hip_undistort *ud, const uint8_t *pixels,
                                          int32_t width, int32_t height, int32_t stride, int32_t apply_clahe,
                                          pvio_hip_image **out);
/* copy a pyramid level back (tests): level l image u8 [h_l][w_l] and/or derivatives int16 [h_l][w_l][2] */
int32_t pvio_hip_image_download_level(pvio_hip_ctx *ctx, const pvio_hip_image *img, int32_t level,
                                      uint8_t *pixels, int16_t *deriv, int32_t *w, int32_t *h);

/* pyramidal LK, win 21x21, levels 3..0, <=30 iterations, eps 0.01, minEigThreshold 1e-4,
 * OPTFLOW_USE_INITIAL_FLOW, followed by the 20-px border kill (opencv_image.cpp:103-109).
 * next_xy is in/out (initial guess in, tracked position out; failed tracks keep their last estimate).
 * The F-matrix RANSAC (:121-129) stays on the host adapter. */
int32_t pvio_hip_klt_track(pvio_hip_ctx *ctx, const pvio_hip_image *prev, const pvio_hip_image *next,
                           int32_t n, const float *prev_xy, float *next_xy, uint8_t *status);

/* Harris corners of the preprocessed image, replaces `gftt()->detect(image, keypoints)` (opencv_image.cpp:61, detector
 * created at :183 as GFTTDetector::create(1000, 1e-3, 20, 3, useHarrisDetector = true) -> k = 0.04, blockSize 3,
 * Sobel aperture 3): response map, 3x3 non-maximum suppression and compaction on the device, ordering by response and
 * the minimum-distance grid selection of cv::goodFeaturesToTrack on the host side of this call.  xy / response take up
 * to max_corners entries, in selection order (strongest first); *n receives the count. */
int32_t pvio_hip_image_detect(pvio_hip_ctx *ctx, const pvio_hip_image *img, int32_t max_corners, double quality_level,
                              double min_distance, float *xy, float *response, int32_t *n);
/* tests: response map (h x w floats) of the last pvio_hip_image_detect call on an image of this size */
int32_t pvio_hip_image_download_response(pvio_hip_ctx *ctx, const pvio_hip_image *img, float *response);

/* hipEvent duration [ms] of the LK kernel of the last pvio_hip_klt_track call (bench.py: tracks/ms, pyramids resident) */
double pvio_hip_klt_last_device_ms(const pvio_hip_ctx *ctx);

/* Fundamental-matrix RANSAC of OpenCvImage::track_keypoints (opencv_image.cpp:113-129: cv::findFundamentalMat(p, q, FM_RANSAC, 1.0, 0.99,
 * mask), of which the reference uses the inlier mask only).  The samples are drawn on the host in cv::RNG's order, the hypotheses of a
 * batch are solved (7-point) and scored against all n matches on the device, the host replays the adaptive iteration count: the result is
 * the sequential algorithm's.  p_xy / q_xy: n points (x, y) each, single precision like cv::Point2f; mask[n] = 1 for the inliers of the best
 * model; F row-major (may be NULL); *n_inliers = 0 and an all-zero mask when no model was found or n < 7. */
int32_t pvio_hip_fundamental_ransac(pvio_hip_ctx *ctx, int32_t n, const float *p_xy, const float *q_xy, double threshold, double confidence,
                                    int32_t max_iterations, uint8_t *mask, double F[9], int32_t *n_inliers);
int32_t pvio_hip_ransac_last_hypotheses(const pvio_hip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PVIO_HIP_H */
