// capi.cpp -- extern "C" entry points declared in include/pvio_hip.h.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/pvio_hip.h"
#include "ba_solver.h"
#include "klt.h"

struct pvio_hip_ctx {
    pvio_hip_opts opts;
    pvba::BASolver *ba = nullptr;
    pvba::Comm *comm = nullptr;
    pvklt::Klt *klt = nullptr;
    std::string err;
};

extern "C" {

const char *pvio_hip_version(void) { return "pvio-mi355x 0.2 (gfx950; ABI 2)"; }
int32_t pvio_hip_abi_version(void) { return PVIO_HIP_ABI_VERSION; }

int32_t pvio_hip_create(const pvio_hip_opts *opts, pvio_hip_ctx **out) {
    if (!out) return PVIO_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    pvio_hip_opts o{};
    o.world_size = 1;
    o.use_graph = 1;
    if (opts) o = *opts;
    if (o.world_size < 1) o.world_size = 1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PVIO_ERR_NO_DEVICE; // no GPU, no fallback
    if (o.device < 0 || o.device >= n) return PVIO_ERR_INVALID_ARGUMENT;
    pvio_hip_ctx *c = new (std::nothrow) pvio_hip_ctx();
    if (!c) return PVIO_ERR_OUT_OF_MEMORY;
    c->opts = o;
    c->ba = new (std::nothrow) pvba::BASolver(o.device, o.rank, o.world_size, o.use_graph != 0);
    if (c->ba) c->ba->set_fault_injection(o.debug_fail_factorizations, o.debug_invalid_steps), c->ba->set_linearize_mode(o.linearize_mode), c->ba->set_force_sharded(o.debug_force_sharded != 0), c->ba->set_reuse_candidates(o.reuse_identical_candidates != 0);
    c->klt = new (std::nothrow) pvklt::Klt(o.device);
    if (!c->ba || !c->klt) {
        pvio_hip_destroy(c);
        return PVIO_ERR_OUT_OF_MEMORY;
    }
    *out = c;
    return PVIO_OK;
}

void pvio_hip_destroy(pvio_hip_ctx *ctx) {
    if (!ctx) return;
    delete ctx->ba;
    delete ctx->klt;
    if (ctx->comm) pvba::comm_destroy(ctx->comm);
    delete ctx;
}

const char *pvio_hip_last_error(const pvio_hip_ctx *ctx) {
    if (!ctx) return "null context";
    if (!ctx->err.empty()) return ctx->err.c_str();
    if (ctx->ba && !ctx->ba->error().empty()) return ctx->ba->error().c_str();
    if (ctx->klt && !ctx->klt->error().empty()) return ctx->klt->error().c_str();
    return "";
}

int32_t pvio_hip_ba_upload(pvio_hip_ctx *ctx, const pvio_ba_problem *problem, const pvio_ba_state *state) {
    if (!ctx) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->ba->upload(problem, state);
}
int32_t pvio_hip_ba_solve_resident(pvio_hip_ctx *ctx, pvio_ba_summary *summary) {
    if (!ctx) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->ba->solve(summary);
}
int32_t pvio_hip_ba_profile_resident(pvio_hip_ctx *ctx, pvio_ba_summary *summary, pvio_ba_kernel_times *times) {
    if (!ctx || !times) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->ba->solve(summary, times);
}
int32_t pvio_hip_ba_graph_replays(const pvio_hip_ctx *ctx) { return ctx && ctx->ba ? ctx->ba->graph_replays() : 0; }
int32_t pvio_hip_ba_last_candidate_repeats(const pvio_hip_ctx *ctx) { return ctx && ctx->ba ? ctx->ba->last_candidate_repeats() : 0; }
int32_t pvio_hip_ba_download(pvio_hip_ctx *ctx, pvio_ba_state *state) {
    if (!ctx) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->ba->download(state);
}
int32_t pvio_hip_ba_solve(pvio_hip_ctx *ctx, const pvio_ba_problem *problem, pvio_ba_state *state, pvio_ba_summary *summary) {
    if (!ctx || !problem || !state) return PVIO_ERR_INVALID_ARGUMENT;
    static const bool timing = std::getenv("PVIO_HIP_TIMING") != nullptr; // diagnostics: host time of the two halves of a call
    const auto t0 = std::chrono::steady_clock::now();
    int rc = ctx->ba->upload(problem, state, /*may_return_early=*/true); // (the solve below synchronizes before the arrays go back to the caller)
    if (rc != PVIO_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    rc = ctx->ba->solve(summary, nullptr, state); // the accepted iterate and the quality pass come back in the solve's own stream round
    if (timing)
        std::fprintf(stderr, "[pvio-hip] ba_solve: staging + upload enqueue %.0f us, iterations + read-back %.0f us (device: %.0f us)\n",
                     std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count(),
                     summary ? 1e6 * summary->device_seconds : 0.0);
    return rc;
}
int32_t pvio_hip_ba_marginalize(pvio_hip_ctx *ctx, const pvio_ba_problem *problem, const pvio_ba_state *state, int32_t victim, pvio_ba_prior *out) {
    if (!ctx || !problem || !state || !out) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->ba->marginalize(problem, state, victim, out);
}
int32_t pvio_hip_ba_reprojection_error(pvio_hip_ctx *ctx, const pvio_ba_problem *problem, const pvio_ba_state *state, double *mean_pixel_error) {
    if (!ctx || !problem || !state || !mean_pixel_error) return PVIO_ERR_INVALID_ARGUMENT;
    int rc = ctx->ba->upload(problem, state);
    if (rc != PVIO_OK) return rc;
    return ctx->ba->reprojection_error(mean_pixel_error);
}

int32_t pvio_hip_comm_unique_id(uint8_t id[128]) { return pvba::comm_unique_id(id) ? PVIO_ERR_COMM : PVIO_OK; }
int32_t pvio_hip_comm_init(pvio_hip_ctx *ctx, const uint8_t id[128], int32_t rank, int32_t world_size) {
    if (!ctx || !id) return PVIO_ERR_INVALID_ARGUMENT;
    if (ctx->comm) pvba::comm_destroy(ctx->comm), ctx->comm = nullptr;
    if (pvba::comm_init(&ctx->comm, id, rank, world_size, ctx->opts.device)) {
        ctx->err = "RCCL communicator init failed";
        return PVIO_ERR_COMM;
    }
    ctx->ba->set_comm(ctx->comm);
    return PVIO_OK;
}

int32_t pvio_hip_image_create(pvio_hip_ctx *ctx, const uint8_t *pixels, int32_t width, int32_t height, int32_t stride, int32_t apply_clahe,
                              pvio_hip_image **out) {
    if (!ctx || !pixels || !out) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->klt->create_image(pixels, width, height, stride, apply_clahe != 0, reinterpret_cast<pvklt::Image **>(out));
}
int32_t pvio_hip_undistort_create(pvio_hip_ctx *ctx, const int16_t *map_xy, const uint16_t *map_frac, int32_t width, int32_t height,
                                  pvio_hip_undistort **out) {
    if (!ctx || !out) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->klt->create_undistort(map_xy, map_frac, width, height, reinterpret_cast<pvklt::Undistort **>(out));
}
void pvio_hip_undistort_release(pvio_hip_ctx *ctx, pvio_hip_undistort *ud) {
    if (ctx && ud) ctx->klt->release_undistort(reinterpret_cast<pvklt::Undistort *>(ud));
}
int32_t pvio_hip_image_create_undistorted(pvio_hip_ctx *ctx, const pvio_hip_undistort *ud, const uint8_t *pixels, int32_t width, int32_t height,
                                          int32_t stride, int32_t apply_clahe, pvio_hip_image **out) {
    if (!ctx || !ud || !pixels || !out) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->klt->create_image(pixels, width, height, stride, apply_clahe != 0, reinterpret_cast<pvklt::Image **>(out),
                                  reinterpret_cast<const pvklt::Undistort *>(ud));
}
void pvio_hip_image_release(pvio_hip_ctx *ctx, pvio_hip_image *img) {
    if (ctx && img) ctx->klt->release_image(reinterpret_cast<pvklt::Image *>(img));
}
int32_t pvio_hip_image_download_level(pvio_hip_ctx *ctx, const pvio_hip_image *img, int32_t level, uint8_t *pixels, int16_t *deriv, int32_t *w,
                                      int32_t *h) {
    if (!ctx || !img) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->klt->download_level(reinterpret_cast<const pvklt::Image *>(img), level, pixels, deriv, w, h);
}
int32_t pvio_hip_image_detect(pvio_hip_ctx *ctx, const pvio_hip_image *img, int32_t max_corners, double quality_level, double min_distance, float *xy,
                              float *response, int32_t *n) {
    if (!ctx || !img || !xy || !response || !n || max_corners <= 0 || !(quality_level > 0)) return PVIO_ERR_INVALID_ARGUMENT;
    int cnt = 0;
    const int rc = ctx->klt->detect(reinterpret_cast<const pvklt::Image *>(img), max_corners, quality_level, min_distance, xy, response, &cnt);
    *n = cnt;
    return rc;
}

int32_t pvio_hip_image_download_response(pvio_hip_ctx *ctx, const pvio_hip_image *img, float *response) {
    if (!ctx || !img || !response) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->klt->download_response(reinterpret_cast<const pvklt::Image *>(img), response);
}

int32_t pvio_hip_klt_track(pvio_hip_ctx *ctx, const pvio_hip_image *prev, const pvio_hip_image *next, int32_t n, const float *prev_xy,
                           float *next_xy, uint8_t *status) {
    if (!ctx || !prev || !next || n < 0 || (n > 0 && (!prev_xy || !next_xy || !status))) return PVIO_ERR_INVALID_ARGUMENT;
    return ctx->klt->track(reinterpret_cast<const pvklt::Image *>(prev), reinterpret_cast<const pvklt::Image *>(next), n, prev_xy, next_xy, status);
}

double pvio_hip_klt_last_device_ms(const pvio_hip_ctx *ctx) { return ctx ? ctx->klt->last_track_ms() : 0.0; }

int32_t pvio_hip_fundamental_ransac(pvio_hip_ctx *ctx, int32_t n, const float *p_xy, const float *q_xy, double threshold, double confidence,
                                    int32_t max_iterations, uint8_t *mask, double F[9], int32_t *n_inliers) {
    if (!ctx || n < 0 || (n > 0 && (!p_xy || !q_xy || !mask)) || !n_inliers || max_iterations < 1) return PVIO_ERR_INVALID_ARGUMENT;
    int good = 0;
    const int rc = ctx->klt->fundamental_ransac(n, p_xy, q_xy, threshold, confidence, max_iterations, mask, F, &good);
    *n_inliers = good;
    return rc;
}
int32_t pvio_hip_ransac_last_hypotheses(const pvio_hip_ctx *ctx) { return ctx ? ctx->klt->last_ransac_hypotheses() : 0; }

} // extern "C"
