// ba_solver.cpp -- host side of the device-resident bundle adjustment.
//
// Replaces the ceres::Problem assembly + ceres::Solve call of BundleAdjustorSolver::solve
// (pvio/src/pvio/estimation/bundle_adjustor.cpp:63-299): the flat problem is uploaded once, the whole
// trust-region loop runs on the device (kernels in ba_kernels.hip), the host only replays a graph of
// "slots" (linearize -> reduce -> dense -> backsub) until the device-side state machine reports done.
#include "ba_solver.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "ba_kernels.h"
#include "sym_eig.h"
#include <cmath>

namespace pvba {
constexpr bool kDenseLookAheadDefault = true; // same-box A/B (profiles/r3_ab_lookahead.txt): 12 076 -> 12 678 iterations/s


DevicePool::~DevicePool() { release(); }
void DevicePool::release() {
    for (auto &s : slots_)
        if (s.ptr) (void)hipFree(s.ptr);
    slots_.clear();
    total_ = 0;
}
void *DevicePool::get(const char *name, size_t bytes, bool *grew) {
    if (bytes == 0) bytes = 8;
    for (auto &s : slots_)
        if (s.name == name) {
            if (s.bytes >= bytes) return s.ptr;
            (void)hipFree(s.ptr);
            total_ -= s.bytes;
            s.ptr = nullptr;
            const size_t nb = bytes + bytes / 2;
            if (hipMalloc(&s.ptr, nb) != hipSuccess) return nullptr;
            s.bytes = nb;
            total_ += nb;
            if (grew) *grew = true;
            return s.ptr;
        }
    // Headroom from the first allocation on: the windows of a sequence change shape with every keyframe (landmarks +-30 %), and a slot that is
    // sized exactly is freed and allocated again -- ~40 slots x (hipFree + hipMalloc) showed up as 0.3-0.4 ms on individual keyframe solves of the
    // rendered sequence (profiles/r4_prof_sequence_solves.txt).  288 GB of HBM: 50 % slack on the largest window (30 x 50 000: 0.4 GB) is nothing.
    Slot s;
    s.name = name;
    bytes = bytes + bytes / 2 < 4096 ? 4096 : bytes + bytes / 2;
    if (hipMalloc(&s.ptr, bytes) != hipSuccess) return nullptr;
    s.bytes = bytes;
    total_ += bytes;
    slots_.push_back(s);
    if (grew) *grew = true;
    return s.ptr;
}

BASolver::BASolver(int device, int rank, int world, bool use_graph) : device_(device), rank_(rank), world_(world), sharded_(world > 1), use_graph_(use_graph) {
    (void)hipSetDevice(device_);
    (void)hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking);
    (void)hipEventCreate(&ev0_);
    (void)hipEventCreate(&ev1_);
    (void)hipHostMalloc(&h_ctrl_, sizeof(Ctrl));
}
BASolver::~BASolver() {
    invalidate_graph();
    if (h_ctrl_) (void)hipHostFree(h_ctrl_);
    if (h_stage_) (void)hipHostFree(h_stage_);
    if (h_back_) (void)hipHostFree(h_back_);
    if (h_pack_) (void)hipHostFree(h_pack_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (stream_) (void)hipStreamDestroy(stream_);
}
int BASolver::fail(int code, const std::string &msg) {
    err_ = msg;
    return code;
}
int BASolver::check(hipError_t e, const char *what) {
    if (e == hipSuccess) return PVIO_OK;
    return fail(PVIO_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
void BASolver::invalidate_graph() {
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    graph_exec_ = nullptr;
    graph_ = nullptr;
    graph_slots_ = 0;
}

// Host inputs of one upload: gathered in ONE pinned staging buffer and sent with ONE DMA into one device slab (a window is
// ~35 small arrays; a pageable hipMemcpyAsync each cost more than the solve's first iterations).  Offsets are 256-byte
// aligned; an array that is not given (null) keeps its place uninitialized, like the separate allocations it replaces.
struct StagedUpload {
    struct Item {
        const void *src;
        size_t bytes, off;
        const void **dst;
    };
    std::vector<Item> items; // offsets are assigned when the upload is flushed
    template <typename T>
    void add(const T *src, size_t n, const T **dst) {
        const size_t bytes = n * sizeof(T);
        items.push_back({(n && src) ? static_cast<const void *>(src) : nullptr, bytes, 0, reinterpret_cast<const void **>(dst)});
    }
};

template <typename T>
static bool dev(DevicePool &pool, const char *name, size_t n, T **dst, bool *grew) {
    T *p = static_cast<T *>(pool.get(name, n * sizeof(T), grew));
    *dst = p;
    return p != nullptr;
}

int BASolver::upload(const pvio_ba_problem *pb, const pvio_ba_state *st, bool may_return_early) {
    if (!pb || !st || !st->frame_state) return fail(PVIO_ERR_INVALID_ARGUMENT, "null problem/state");
    const int N = pb->n_frames, M = pb->n_landmarks, F = pb->n_obs;
    if (N < 1 || N > kMaxFrames) return fail(PVIO_ERR_UNSUPPORTED, "n_frames must be in [1, 32]");
    if (M < 0 || F < 0 || (M > 0 && !st->lm_inv_depth)) return fail(PVIO_ERR_INVALID_ARGUMENT, "bad landmark arrays");
    if (!pb->frame_fixed || !pb->cam_extrinsic || !pb->imu_extrinsic || !pb->sqrt_inv_cov || !pb->intrinsics) return fail(PVIO_ERR_INVALID_ARGUMENT, "null per-frame array");
    if (M > 0 && (!pb->lm_anchor_frame || !pb->lm_anchor_z || !pb->lm_obs_ptr)) return fail(PVIO_ERR_INVALID_ARGUMENT, "null landmark array");
    if (F > 0 && (!pb->obs_frame || !pb->obs_z)) return fail(PVIO_ERR_INVALID_ARGUMENT, "null observation array");
    if (M > 0 && (pb->lm_obs_ptr[0] != 0 || pb->lm_obs_ptr[M] != F)) return fail(PVIO_ERR_INVALID_ARGUMENT, "lm_obs_ptr must start at 0 and end at n_obs");
    if (pb->prior_n < 0 || pb->prior_n > N || pb->max_iterations < 0 || pb->n_plane_factors < 0) return fail(PVIO_ERR_INVALID_ARGUMENT, "negative count / prior_n > n_frames");
    if (pb->prior_n > 0 && (!pb->prior_frames || !pb->prior_S || !pb->prior_s || !pb->prior_lin_state)) return fail(PVIO_ERR_INVALID_ARGUMENT, "null prior array");
    if (pb->use_inertial && (!pb->preint_valid || !pb->preint_delta || !pb->preint_sqrt_inv_cov || !pb->preint_jacobian)) return fail(PVIO_ERR_INVALID_ARGUMENT, "null pre-integration array");
    if (pb->n_plane_factors > 0) {
        if (!pb->plane_obs_ptr || !pb->plane_obs_frame || !pb->plane_obs_z || !pb->plane_normal || !pb->plane_distance) return fail(PVIO_ERR_INVALID_ARGUMENT, "null plane array");
        if (pb->plane_obs_ptr[0] != 0) return fail(PVIO_ERR_INVALID_ARGUMENT, "plane_obs_ptr must start at 0");
        for (int f = 0; f < pb->n_plane_factors; ++f) {
            if (pb->plane_obs_ptr[f + 1] < pb->plane_obs_ptr[f]) return fail(PVIO_ERR_INVALID_ARGUMENT, "plane_obs_ptr is not a valid CSR");
            for (int o = pb->plane_obs_ptr[f]; o < pb->plane_obs_ptr[f + 1]; ++o)
                if (pb->plane_obs_frame[o] < 0 || pb->plane_obs_frame[o] >= N) return fail(PVIO_ERR_INVALID_ARGUMENT, "plane observation frame out of range");
        }
    }
    if (check(hipSetDevice(device_), "hipSetDevice")) return PVIO_ERR_HIP;
    Dims dm{};
    dm.N = N, dm.M = M, dm.F = F;
    dm.use_inertial = pb->use_inertial ? 1 : 0;
    dm.prior_n = pb->prior_n;
    dm.d = (dm.use_inertial || dm.prior_n > 0) ? 15 : 6;
    dm.P = dm.d * N, dm.P6 = 6 * N;
    dm.n_tasks = 4 * N * (N + 1) / 2;
    dm.max_iter = pb->max_iterations;
    max_solver_time_ = pb->max_solver_time > 0 ? pb->max_solver_time : 0.0;
    dm.world = world_, dm.rank = rank_;
    dm.n_plane = pb->n_plane_factors;

    // ---- block usage (Ceres removes constant and unreferenced parameter blocks from the reduced program) ----
    std::vector<uint8_t> pose_used(N, 0), motion_used(N, 0), pose_active(N, 0), motion_active(N, 0), pre_valid(N, 0);
    std::vector<int32_t> obs_lm(F, 0);
    std::vector<uint32_t> lm_seen((size_t)std::max(M, 1), 0u); // bit f: the landmark is observed in frame f
    int maxK = 1;
    for (int l = 0; l < M; ++l) {
        const int b = pb->lm_obs_ptr[l], e = pb->lm_obs_ptr[l + 1];
        if (b < 0 || e < b || e > F) return fail(PVIO_ERR_INVALID_ARGUMENT, "lm_obs_ptr is not a valid CSR");
        const int a = pb->lm_anchor_frame[l];
        if (a < 0 || a >= N) return fail(PVIO_ERR_INVALID_ARGUMENT, "anchor frame out of range");
        if (e > b) pose_used[a] = 1;
        maxK = std::max(maxK, e - b);
        uint32_t seen = 0;
        for (int o = b; o < e; ++o) {
            const int t = pb->obs_frame[o];
            if (t < 0 || t >= N || t == a) return fail(PVIO_ERR_INVALID_ARGUMENT, "observation frame out of range or equal to the anchor");
            if (seen & (1u << t)) return fail(PVIO_ERR_UNSUPPORTED, "a landmark lists the same target frame twice");
            seen |= 1u << t;
            obs_lm[o] = l;
            pose_used[t] = 1;
        }
        lm_seen[(size_t)l] = seen;
    }
    if (maxK > kLinThreads) return fail(PVIO_ERR_UNSUPPORTED, "too many observations per landmark");
    if (dm.use_inertial && pb->preint_valid)
        for (int j = 1; j < N; ++j)
            if (pb->preint_valid[j]) {
                pre_valid[j] = 1;
                pose_used[j - 1] = pose_used[j] = motion_used[j - 1] = motion_used[j] = 1;
            }
    std::vector<int32_t> prior_slot(N, -1); // frame -> slot of the prior (the last one that names it, like the kernels' old search)
    for (int i = 0; i < dm.prior_n; ++i) {
        const int f = pb->prior_frames[i];
        if (f < 0 || f >= N) return fail(PVIO_ERR_INVALID_ARGUMENT, "prior frame out of range");
        pose_used[f] = motion_used[f] = 1;
        prior_slot[f] = i;
    }
    // rotation priors: at most one per frame; one on a fixed frame is a constant block and is dropped
    std::vector<int32_t> rot_slot(N, -1);
    if (pb->n_rot_priors < 0 || (pb->n_rot_priors > 0 && (!pb->rot_prior_frame || !pb->rot_prior_q0 || !pb->rot_prior_sqrt_info)))
        return fail(PVIO_ERR_INVALID_ARGUMENT, "bad rotation prior arrays");
    for (int i = 0; i < pb->n_rot_priors; ++i) {
        const int f = pb->rot_prior_frame[i];
        if (f < 0 || f >= N) return fail(PVIO_ERR_INVALID_ARGUMENT, "rotation prior frame out of range");
        if (rot_slot[f] >= 0) return fail(PVIO_ERR_UNSUPPORTED, "more than one rotation prior on a frame");
        rot_slot[f] = i;
        pose_used[f] = 1;
    }
    dm.n_rot = pb->n_rot_priors;
    for (int f = 0; f < dm.n_plane; ++f) {
        bool any_free = false;
        for (int o = pb->plane_obs_ptr[f]; o < pb->plane_obs_ptr[f + 1]; ++o) any_free |= !pb->frame_fixed[pb->plane_obs_frame[o]];
        if (any_free)
            for (int o = pb->plane_obs_ptr[f]; o < pb->plane_obs_ptr[f + 1]; ++o) pose_used[pb->plane_obs_frame[o]] = 1;
    }
    for (int i = 0; i < N; ++i) {
        pose_active[i] = pose_used[i] && !pb->frame_fixed[i];
        motion_active[i] = (dm.d == 15) && motion_used[i];
    }

    if (cus_ <= 0) { // asked once: the query is slow
        hipDeviceProp_t prop;
        if (check(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties")) return PVIO_ERR_HIP;
        cus_ = std::max(1, prop.multiProcessorCount);
    }
    const int cus = cus_;
    // ---- landmark chunks: <= lm_slots landmarks and <= 256 factors each (one thread per factor).  A small window is
    // latency-bound per workgroup, so chunks are sized to put one chunk on every CU (256 on MI355X) rather than to fill
    // the LDS; a large window falls back to LDS-filling chunks that each workgroup walks in a grid-stride loop. ----
    const size_t lds_budget = 112 * 1024;
    dm.plane_slots = (int)std::max<size_t>(1, std::min<size_t>(lds_budget / 8 / (dm.P6 + 2), (size_t)kLinThreads));
    std::vector<int32_t> plane_chunk;
    for (int f = 0; f <= dm.n_plane; f += dm.plane_slots) plane_chunk.push_back(f);
    if (plane_chunk.empty() || plane_chunk.back() != dm.n_plane) plane_chunk.push_back(dm.n_plane);
    dm.n_plane_chunks = (int)plane_chunk.size() - 1;
    dm.G_plane = dm.n_plane > 0 ? std::max(1, std::min(dm.n_plane_chunks, std::max(1, cus / 4))) : 0;
    dm.G_pre = dm.use_inertial ? N - 1 : 0;
    dm.G_prior = dm.prior_n; // one workgroup per prior frame
    // One workgroup fits on a CU (registers, LDS) and the IMU / prior workgroups are the longest: the landmark chunks get
    // the CUs that are left, so that the whole grid is resident at once instead of queueing a second round behind it.
    // One landmark chunk per free CU.  Round 5 measured the alternative the 22x traffic of the partial rows suggests -- fewer, fatter workgroups, i.e.
    // fewer rows for k_reduce to sum (PVIO_HIP_LM_WGS=n caps the count; profiles/r5_ab_partials.txt): the traffic falls with the count and the
    // iteration rate with it, because a workgroup's chain grows with the landmarks it holds (the per-landmark sums and the tile accumulation walk
    // them one after another): 48 rows = 12 161 against 14 112 iterations/s, k_linearize 18.4 -> 29.2 us for 0.4 us less in k_reduce.
    // (an experiment switch, unsupported: read once per process, clamped to [1, 4 x CUs] -- the partial-row buffers are sized from the resulting grid below,
    // so any value in the range is safe; anything else is ignored)
    static const int lm_wgs_env = std::getenv("PVIO_HIP_LM_WGS") ? std::atoi(std::getenv("PVIO_HIP_LM_WGS")) : 0;
    const int lm_wgs_cap = lm_wgs_env >= 1 ? std::min(lm_wgs_env, 4 * cus) : 0;
    int lm_cus = std::max(1, cus - dm.G_plane - dm.G_pre - dm.G_prior);
    if (lm_wgs_cap > 0) lm_cus = lm_wgs_cap; // (may exceed the free CUs: more than one workgroup per CU where registers and LDS allow)
    const int slots_lds = (int)std::max<size_t>(1, std::min<size_t>(lds_budget / 8 / (40 * N + 46), (size_t)kLinThreads));
    const int slots_spread = std::max(1, (M + lm_cus - 1) / lm_cus);
    dm.lm_slots = std::min(slots_lds, slots_spread);
    std::vector<int32_t> chunk_lm;
    chunk_lm.push_back(0);
    {
        int cnt = 0, fac = 0;
        for (int l = 0; l < M; ++l) {
            const int k = pb->lm_obs_ptr[l + 1] - pb->lm_obs_ptr[l];
            if (cnt > 0 && (cnt + 1 > dm.lm_slots || fac + k > kLinThreads)) {
                chunk_lm.push_back(l);
                cnt = 0, fac = 0;
            }
            ++cnt, fac += k;
        }
        if (M > 0) chunk_lm.push_back(M);
    }
    dm.n_chunks = (int)chunk_lm.size() - 1;
    dm.G_lm = std::max(1, std::min(dm.n_chunks, lm_cus));
    // Large windows take the throughput role (ba_lin_tp.h: chunks of 256 factors walked by workgroups that keep their accumulators, Schur complement on the
    // matrix cores); its prologue + flush + first chunk cost ~30 us whatever the window, the register-tile role's time grows with the landmarks per workgroup:
    // same-box A/B over 6 .. 24 frames (profiles/r6_lin_mode_ab.txt): they cross at ~40 000 factors (10 x 5000: 35.7 -> 32.4 us, 10 x 10 000: 66.7 -> 40.2 us,
    // 16 x 3000: 48.4 -> 41.0 us; 10 x 3000: 28.1 against 31.3 us, 6 x 3000: 24.5 against 34.5 us)
    dm.lm_mm = (lin_mode_ == 2 && M > 0) || (lin_mode_ == 0 && F >= 40000); // (a window without landmarks has no chunks: the register-tile role's empty walk handles it)
    // Large windows (ba_lin_tp.h): chunks of <= 256 factors whose landmarks share ONE anchor frame (a chunk is cut where the anchor changes: the
    // reference's block order is anchor-sorted, any other order only makes more chunks), as many landmarks as the LDS holds U rows for; every
    // chunk's factors sorted by target frame (the direct part of J^T J is accumulated per target)
    std::vector<int32_t> chunk_tptr, chunk_geo, tp_tile_dst;
    std::vector<uint8_t> chunk_perm;
    if (dm.lm_mm) {
        // scatter table of the Schur tiles' accumulator entries into the element-major 3 x 3-task partial row (the flush of ba_lin_tp.h): entry r of
        // lane l of tile (bi, bj) is element (16 bi + (l >> 4) + 4 r, 16 bj + (l & 15)) of the lower triangle
        const int nbt = tp_u_stride(dm.P6) >> 4;
        auto task_index = [&](int fi, int fj, int si, int sj) { return 4 * (fi * N - ((fi * (fi - 1)) >> 1) + (fj - fi)) + 2 * si + sj; };
        for (int bi = 0; bi < nbt; ++bi)
            for (int bj = 0; bj <= bi; ++bj)
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 4; ++r) {
                        const int I = 16 * bi + (l >> 4) + 4 * r, J = 16 * bj + (l & 15);
                        int32_t dst = -1;
                        if (I < dm.P6 && J <= I) {
                            const int fI = I / 6, iI = I % 6, fJ = J / 6, jJ = J % 6; // fJ <= fI
                            // partial_entry(N, fI, iI, fJ, jJ) of ba_kernels.hip: off-diagonal blocks once (smaller frame = row of the task)
                            const int el = fI == fJ ? 3 * (iI % 3) + (jJ % 3) : 3 * (jJ % 3) + (iI % 3);
                            const int t = fI == fJ ? task_index(fI, fI, iI / 3, jJ / 3) : task_index(fJ, fI, jJ / 3, iI / 3);
                            dst = (el << 24) | t;
                        }
                        tp_tile_dst.push_back(dst);
                    }
    }
    if (dm.lm_mm) {
        dm.lm_slots = tp_landmark_slots(dm, F > 0 ? (int)(((long long)kLinThreads * M + F - 1) / F) : 64);
        chunk_lm.assign(1, 0);
        int cnt = 0, fac = 0, anchor = -1;
        for (int l = 0; l < M; ++l) {
            const int k = pb->lm_obs_ptr[l + 1] - pb->lm_obs_ptr[l];
            if (cnt > 0 && (cnt + 1 > dm.lm_slots || fac + k > kLinThreads || pb->lm_anchor_frame[l] != anchor)) {
                chunk_lm.push_back(l);
                cnt = 0, fac = 0;
            }
            anchor = pb->lm_anchor_frame[l];
            ++cnt, fac += k;
        }
        if (M > 0) chunk_lm.push_back(M);
        dm.n_chunks = (int)chunk_lm.size() - 1;
        // two workgroups per CU where their LDS allows (tp_landmark_slots) and the kernel is held to 256 registers (k_linearize<T <= 2, true>)
        dm.lm_mm = 1;
        const int per_cu = (linearize_lds_bytes(dm) <= 80 * 1024 && tiles_per_thread(dm) <= 2) ? 2 : 1;
        dm.G_lm = std::max(1, std::min(dm.n_chunks, lm_wgs_cap > 0 ? lm_wgs_cap : per_cu * cus - (dm.G_plane + dm.G_pre + dm.G_prior)));
        chunk_geo.assign((size_t)dm.n_chunks * 8, 0);
        for (int c = 0; c < dm.n_chunks; ++c) {
            int32_t *g = chunk_geo.data() + (size_t)c * 8;
            g[0] = chunk_lm[c], g[1] = chunk_lm[c + 1] - chunk_lm[c], g[2] = pb->lm_obs_ptr[chunk_lm[c]], g[3] = pb->lm_obs_ptr[chunk_lm[c + 1]] - g[2];
            g[4] = pb->lm_anchor_frame[chunk_lm[c]];
        }
        chunk_tptr.assign((size_t)dm.n_chunks * (N + 1), 0);
        chunk_perm.assign((size_t)std::max(F, 1), 0);
        for (int c = 0; c < dm.n_chunks; ++c) {
            const int o0 = pb->lm_obs_ptr[chunk_lm[c]], o1 = pb->lm_obs_ptr[chunk_lm[c + 1]];
            int32_t *tp = chunk_tptr.data() + (size_t)c * (N + 1);
            for (int o = o0; o < o1; ++o) ++tp[pb->obs_frame[o] + 1];
            for (int t = 0; t < N; ++t) tp[t + 1] += tp[t];
            int fill[kMaxFrames] = {0};
            for (int o = o0; o < o1; ++o) { // counting sort, stable: a target's factors in landmark order
                const int t = pb->obs_frame[o];
                chunk_perm[(size_t)o0 + tp[t] + fill[t]++] = (uint8_t)(o - o0);
            }
        }
    }
    // pvio_hip_opts::reuse_identical_candidates.  Not on landmark shards: their k_reduce feeds an all-reduce that sums IN PLACE, a skipped slot
    // would add the last slot's sums to themselves.
    dm.reuse_cand = (reuse_cand_ && !sharded_) ? 1 : 0;
    if (dm.lm_mm && dm.n_chunks > 0) { // contiguous chunk ranges (a range mostly shares one anchor frame): no more workgroups than ranges
        const int per_wg = (dm.n_chunks + dm.G_lm - 1) / dm.G_lm;
        dm.G_lm = std::max(1, (dm.n_chunks + per_wg - 1) / per_wg);
    }
    // 16 landmarks per workgroup pass (16 lanes each); <= 64 partial rows (one per lane of the wave that sums them) until a
    // large window needs the whole chip
    dm.G_back = M <= 4096 ? std::max(1, std::min(64, (M + 15) / 16)) : std::min(cus, (M + 63) / 64);
    // (PVIO_HIP_FUSE_BACKSUB=0: tests -- small windows through the separate k_backsub launch, i.e. the split-finalize form)
    static const bool fuse_off = std::getenv("PVIO_HIP_FUSE_BACKSUB") != nullptr && std::atoi(std::getenv("PVIO_HIP_FUSE_BACKSUB")) == 0;
    // (PVIO_HIP_FUSE_BACKSUB_MAX: experiments -- the landmark count up to which the back-substitution stays inside k_dense; clamped to 16 .. 4096)
    static const int fuse_max = std::getenv("PVIO_HIP_FUSE_BACKSUB_MAX") ? std::min(4096, std::max(16, std::atoi(std::getenv("PVIO_HIP_FUSE_BACKSUB_MAX")))) : 256;
    dm.fuse_backsub = (!sharded_ && M <= fuse_max && !fuse_off) ? 1 : 0; // beyond one landmark per thread the separate launch is faster
    dm.n_back_rows = (sharded_ || dm.fuse_backsub) ? 1 : dm.G_back;

    // 3x3 tile tasks over the upper block triangle
    std::vector<int32_t> task_desc;
    for (int fi = 0; fi < N; ++fi)
        for (int fj = fi; fj < N; ++fj)
            for (int s = 0; s < 4; ++s) task_desc.push_back(fi | (fj << 8) | ((s >> 1) << 16) | ((s & 1) << 17));

    // ---- device buffers ----
    bool grew = false, sent_directly = false;
    View v{};
    v.dm = dm;
    const size_t Ns = N, Ms = std::max(M, 1), Fs = std::max(F, 1);
    bool ok = true;
    StagedUpload stage;
    stage.add(pb->frame_fixed, Ns, &v.frame_fixed);
    stage.add(pose_active.data(), Ns, &v.pose_active);
    stage.add(motion_active.data(), Ns, &v.motion_active);
    stage.add(pb->cam_extrinsic, Ns * 7, &v.cam_ext);
    stage.add(pb->imu_extrinsic, Ns * 7, &v.imu_ext);
    stage.add(pb->sqrt_inv_cov, Ns * 4, &v.sic);
    stage.add(pb->intrinsics, Ns * 4, &v.intr);
    stage.add(pb->lm_anchor_frame, (size_t)M, &v.lm_anchor);
    stage.add(pb->lm_obs_ptr, (size_t)M + 1, &v.lm_ptr);
    stage.add(pb->lm_anchor_z, (size_t)M * 2, &v.lm_zref);
    std::vector<double> lm_mult; // duplicate residual blocks (bundle_adjustor.cpp:165-179); staged only when some landmark counts twice
    if (pb->lm_multiplicity) {
        bool any = false;
        for (int l = 0; l < M; ++l) {
            if (pb->lm_multiplicity[l] < 1) return fail(PVIO_ERR_INVALID_ARGUMENT, "lm_multiplicity must be >= 1");
            any |= pb->lm_multiplicity[l] != 1;
        }
        if (any) {
            lm_mult.resize((size_t)M);
            for (int l = 0; l < M; ++l) lm_mult[(size_t)l] = (double)pb->lm_multiplicity[l];
        }
    }
    stage.add(lm_mult.empty() ? (const double *)nullptr : lm_mult.data(), lm_mult.size(), &v.lm_mult);
    stage.add(pb->obs_frame, (size_t)F, &v.obs_frame);
    stage.add(pb->obs_z, (size_t)F * 2, &v.obs_z);
    stage.add(obs_lm.data(), (size_t)F, &v.obs_lm);
    stage.add(lm_seen.data(), (size_t)M, &v.lm_seen);
    stage.add(chunk_lm.data(), chunk_lm.size(), &v.chunk_lm);
    stage.add(chunk_tptr.empty() ? (const int32_t *)nullptr : chunk_tptr.data(), chunk_tptr.size(), &v.chunk_tptr);
    stage.add(chunk_geo.empty() ? (const int32_t *)nullptr : chunk_geo.data(), chunk_geo.size(), &v.chunk_geo);
    stage.add(chunk_perm.empty() ? (const uint8_t *)nullptr : chunk_perm.data(), chunk_perm.size(), &v.chunk_perm);
    stage.add(tp_tile_dst.empty() ? (const int32_t *)nullptr : tp_tile_dst.data(), tp_tile_dst.size(), &v.tp_tile_dst);
    stage.add(task_desc.data(), task_desc.size(), &v.task_desc);
    stage.add(pre_valid.data(), Ns, &v.pre_valid);
    stage.add(dm.use_inertial ? pb->preint_delta : nullptr, Ns * 11, &v.pre_delta);
    stage.add(dm.use_inertial ? pb->preint_sqrt_inv_cov : nullptr, Ns * 225, &v.pre_U);
    stage.add(dm.use_inertial ? pb->preint_jacobian : nullptr, Ns * 45, &v.pre_jac);
    const size_t Dp = 15 * (size_t)dm.prior_n;
    stage.add(pb->prior_frames, (size_t)dm.prior_n, &v.prior_frames);
    stage.add(pb->prior_S, Dp * Dp, &v.prior_S);
    stage.add(pb->prior_s, Dp, &v.prior_s);
    stage.add(pb->prior_lin_state, (size_t)dm.prior_n * 16, &v.prior_lin);
    double *Lambda = nullptr, *eta = nullptr, *ST = nullptr;
    ok &= dev(pool_, "prior_Lambda", Dp * Dp, &Lambda, &grew);
    ok &= dev(pool_, "prior_eta", Dp, &eta, &grew);
    ok &= dev(pool_, "prior_ST", Dp * Dp, &ST, &grew);
    v.prior_Lambda = Lambda, v.prior_eta = eta, v.prior_ST = ST;
    const size_t npo = dm.n_plane ? (size_t)pb->plane_obs_ptr[dm.n_plane] : 0;
    stage.add(pb->plane_obs_ptr, (size_t)dm.n_plane + 1, &v.plane_ptr);
    stage.add(pb->plane_obs_frame, npo, &v.plane_frame);
    stage.add(plane_chunk.data(), plane_chunk.size(), &v.plane_chunk);
    stage.add(pb->plane_obs_z, npo * 2, &v.plane_z);
    stage.add(pb->plane_normal, (size_t)dm.n_plane * 3, &v.plane_normal);
    stage.add(pb->plane_distance, (size_t)dm.n_plane, &v.plane_dist);
    v.plane_sic = pb->plane_sqrt_inv_cov;
    stage.add(rot_slot.data(), Ns, &v.rot_slot);
    stage.add(prior_slot.data(), Ns, &v.prior_slot);
    stage.add(pb->rot_prior_q0, (size_t)dm.n_rot * 4, &v.rot_q0);
    stage.add(pb->rot_prior_sqrt_info, (size_t)dm.n_rot * 9, &v.rot_W);
    // state + work
    ok &= dev(pool_, "ctrl", 1, &v.ctrl, &grew);
    ok &= dev(pool_, "cand_rec", 24, &v.cand_rec, &grew);
    ok &= dev(pool_, "fs", 2 * Ns * 16, &v.fs, &grew);
    ok &= dev(pool_, "fs_user", Ns * 16, &v.fs_user, &grew);
    ok &= dev(pool_, "bias0_lin", Ns * 6, &v.bias0_lin, &grew);
    ok &= dev(pool_, "rho", 2 * Ms, &v.rho, &grew);
    ok &= dev(pool_, "cl", Ms, &v.cl, &grew);
    ok &= dev(pool_, "Hll", 2 * Ms, &v.Hll, &grew);
    ok &= dev(pool_, "bl", 2 * Ms, &v.bl, &grew);
    ok &= dev(pool_, "Dl", 2 * Ms, &v.Dl, &grew);
    ok &= dev(pool_, "ghl", 2 * Ms, &v.ghl, &grew);
    ok &= dev(pool_, "gnl", 2 * Ms, &v.gnl, &grew);
    ok &= dev(pool_, "Wa", 2 * Ms * 6, &v.Wa, &grew);
    ok &= dev(pool_, "Wt", 2 * Fs * 6, &v.Wt, &grew);
    const size_t G = (size_t)dm.G_lm + dm.G_plane, nS = (size_t)dm.n_tasks * 9, nV = (size_t)kNumPoseVec * dm.P6;
    ok &= dev(pool_, "part_S", G * nS, &v.part_S, &grew);
    ok &= dev(pool_, "part_vec", G * nV, &v.part_vec, &grew);
    ok &= dev(pool_, "part_scal", G * kNumLinScal, &v.part_scal, &grew);
    ok &= dev(pool_, "red", nS + nV + kNumLinScal + (size_t)world_, &v.red, &grew); // + one max slot per rank
    ok &= dev(pool_, "back_part", (size_t)std::max(dm.G_back, 1) * kNumBackScal, &v.back_part, &grew);
    ok &= dev(pool_, "back_red", kNumBackScal, &v.back_red, &grew);
    ok &= dev(pool_, "pre_H", Ns * 900, &v.pre_H, &grew);
    ok &= dev(pool_, "pre_g", Ns * 30, &v.pre_g, &grew);
    ok &= dev(pool_, "pre_cost", Ns, &v.pre_cost, &grew);
    ok &= dev(pool_, "prior_H", Dp * Dp, &v.prior_H, &grew);
    ok &= dev(pool_, "prior_g", Dp, &v.prior_g, &grew);
    ok &= dev(pool_, "prior_cost", (size_t)std::max(dm.prior_n, 1), &v.prior_cost, &grew);
    {
        // arrays the kernels read for every frame whether the window has the factor or not: the prior's gradient / diagonal by frame
        // (k_dense's vector assembly; frames without a prior slot stay zero) and the rotation-prior blocks (k_reduce / k_dense; zero
        // when the window has none).  One block, one memset.
        double *zb = nullptr;
        ok &= dev(pool_, "zero_block", Ns * 42 + 1, &zb, &grew);
        v.prior_gd = zb, v.rot_H = zb + Ns * 30, v.rot_g = v.rot_H + Ns * 9, v.rot_cost = v.rot_g + Ns * 3;
        if (ok && check(hipMemsetAsync(zb, 0, (dm.n_rot == 0 ? Ns * 42 + 1 : Ns * 30) * sizeof(double), stream_), "memset zero block")) return PVIO_ERR_HIP;
    }
    const size_t P = dm.P;
    ok &= dev(pool_, "Smat", dense_tile_doubles(dm), &v.Smat, &grew);
    // tile image: k_reduce also assembles the reduced system entry by entry where the dense kernel's tile owners load it
    // from (single GPU, system resident in LDS/registers); zero wherever nothing is ever written
    {
        int lds_matrix = 0;
        dense_lds_bytes(dm, &lds_matrix);
        const size_t img_sz = dense_tile_doubles(dm);
        v.dm.use_img = 1; // (landmark-sharded runs build it from the all-reduced `red`: k_reduce phase 2)
        v.dm.img_sz = (int)img_sz;
        dm.use_img = v.dm.use_img, dm.img_sz = v.dm.img_sz;
        // split finalize: with a separate k_backsub launch on one GPU, k_dense leaves the gradient max-norm, the state / trace copies
        // and (register-resident factorization: the image is what it factors) the pose part of v^T H v to that launch.
        // PVIO_HIP_SPLIT_FIN=0 keeps everything in k_dense (A/B timing, tests of both forms).
        static const bool split_off = std::getenv("PVIO_HIP_SPLIT_FIN") != nullptr && std::atoi(std::getenv("PVIO_HIP_SPLIT_FIN")) == 0;
        // (landmark shards: every rank runs the same finalize workgroup on the same all-reduced data; the pose part of v^T H v is formed by
        // rank 0's k_backsub alone -- the exchange step behind it sums the ranks' rows)
        dm.split_fin = (!dm.fuse_backsub && !split_off) ? 1 : 0;
        dm.qvv_back = (dm.split_fin && dm.use_img && lds_matrix) ? 1 : 0;
        v.dm.split_fin = dm.split_fin, v.dm.qvv_back = dm.qvv_back;
        // look-ahead form of the register-resident factorization (PVIO_HIP_DENSE_LA=0 / 1 overrides the default)
        static const int la_env = std::getenv("PVIO_HIP_DENSE_LA") ? std::atoi(std::getenv("PVIO_HIP_DENSE_LA")) : -1;
        dm.dense_la = (lds_matrix && dm.use_img && (la_env < 0 ? kDenseLookAheadDefault : la_env != 0)) ? 1 : 0;
        v.dm.dense_la = dm.dense_la;
        // the image in the form the accumulators hold once the scaling exists (PVIO_HIP_IMG_SCALED=0: k_dense scales every time)
        static const bool img_scaled_off = std::getenv("PVIO_HIP_IMG_SCALED") != nullptr && std::atoi(std::getenv("PVIO_HIP_IMG_SCALED")) == 0;
        dm.img_scaled = (dm.dense_la && dm.split_fin && dm.qvv_back && !img_scaled_off) ? 1 : 0;
        v.dm.img_scaled = dm.img_scaled;
        // (+ one tile of zeros behind the image: what k_dense's waves load into the accumulator slots they do not own)
        ok &= dev(pool_, "img", img_sz + 256, &v.img, &grew);
        if (ok && v.dm.use_img && check(hipMemsetAsync(v.img, 0, (img_sz + 256) * sizeof(double), stream_), "memset img")) return PVIO_ERR_HIP;
    }
    ok &= dev(pool_, "cp", P, &v.cp, &grew);
    ok &= dev(pool_, "cpl", P, &v.cpl, &grew);
    ok &= dev(pool_, "vraw", P, &v.vraw, &grew);
    ok &= dev(pool_, "Dp", P, &v.Dp, &grew);
    ok &= dev(pool_, "gtot", P, &v.gtot, &grew);
    ok &= dev(pool_, "ghp", P, &v.ghp, &grew);
    ok &= dev(pool_, "vstep", P, &v.vstep, &grew);
    ok &= dev(pool_, "ystep", P, &v.ystep, &grew);
    ok &= dev(pool_, "lm_quality", Ms, &v.lm_quality, &grew);
    ok &= dev(pool_, "lm_valid", Ms, &v.lm_valid, &grew);
    trace_cap_ = dm.max_iter + 2;
    ok &= dev(pool_, "trace", (size_t)trace_cap_, &v.trace, &grew);
    v.trace_states = nullptr;
    if (!ok) return fail(PVIO_ERR_OUT_OF_MEMORY, "device allocation / upload failed");

    // initial state: kept on the host (tiny for frames) and on the device in buffer 0
    h_init_fs_.assign(st->frame_state, st->frame_state + Ns * 16);
    h_init_rho_.assign(st->lm_inv_depth, st->lm_inv_depth + M);
    stage.add(h_init_fs_.data(), Ns * 16, &fs_init_);
    stage.add(h_init_rho_.data(), (size_t)M, &rho_init_);
    // ---- one staging pass, one DMA ----
    {
        // small arrays ride in the staging buffer (front of the slab); a big one (the observation arrays of a 50 000-landmark
        // window) is cheaper sent from where it lies than copied twice, and lives behind them
        constexpr size_t kDirect = 1 << 20;
        size_t staged = 0, total = 0;
        for (auto &it : stage.items)
            if (it.bytes < kDirect) it.off = staged, staged += (std::max<size_t>(it.bytes, 8) + 255) & ~(size_t)255;
        total = staged;
        for (auto &it : stage.items)
            if (it.bytes >= kDirect) it.off = total, total += (it.bytes + 255) & ~(size_t)255;
        char *slab = static_cast<char *>(pool_.get("inputs", total, &grew));
        if (!slab) return fail(PVIO_ERR_OUT_OF_MEMORY, "device allocation failed");
        if (stage_in_flight_) {
            // the last upload returned with its staged copy queued and the call that was to synchronize behind it ended early (an error in
            // solve() / marginalize() before their own synchronization): drain the stream before the staging buffer is written again
            if (check(hipStreamSynchronize(stream_), "staging buffer still in flight")) return PVIO_ERR_HIP;
            stage_in_flight_ = false;
        }
        if (staged > h_stage_cap_) {
            if (h_stage_) (void)hipHostFree(h_stage_);
            h_stage_ = nullptr, h_stage_cap_ = 0;
            const size_t cap = staged + staged / 2; // (pinned host memory is the expensive allocation: headroom for the next windows of the sequence)
            if (hipHostMalloc(&h_stage_, cap) != hipSuccess) return fail(PVIO_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
            h_stage_cap_ = cap;
        }
        for (const auto &it : stage.items) {
            if (it.src && it.bytes < kDirect) std::memcpy(static_cast<char *>(h_stage_) + it.off, it.src, it.bytes);
            *it.dst = slab + it.off;
        }
        if (check(hipMemcpyAsync(slab, h_stage_, staged, hipMemcpyHostToDevice, stream_), "H2D inputs")) return PVIO_ERR_HIP;
        for (const auto &it : stage.items)
            if (it.src && it.bytes >= kDirect) {
                if (check(hipMemcpyAsync(slab + it.off, it.src, it.bytes, hipMemcpyHostToDevice, stream_), "H2D inputs")) return PVIO_ERR_HIP;
                sent_directly = true;
            }
    }
    if (lm_mult.empty()) v.lm_mult = nullptr; // no duplicates: the kernels take the unscaled path
    if (dm.prior_n > 0 && check(launch_prior_prep(v.prior_S, v.prior_s, (int)Dp, Lambda, eta, ST, stream_), "k_prior_prep")) return PVIO_ERR_HIP;

    const bool dims_changed = std::memcmp(&v.dm, &v_.dm, sizeof(Dims)) != 0;
    if (grew || dims_changed || std::memcmp(&v, &v_, sizeof(View)) != 0) invalidate_graph();
    v_ = v;
    uploaded_ = true;
    solves_since_upload_ = 0;
    // The caller of the public entry point may release its arrays as soon as this returns, and arrays that went straight from where they
    // lay (>= 1 MB; some of them temporaries of this function) must have left: wait.  A solve that follows immediately on the same stream
    // (pvio_hip_ba_solve: may_return_early) synchronizes anyway before ITS caller gets the arrays back; the staged part was copied into the
    // pinned buffer above, which is not written again before the next upload.
    if (may_return_early && !sent_directly) {
        stage_in_flight_ = true; // the DMA may still be reading h_stage_: the next upload waits for it before it writes there
        return PVIO_OK;
    }
    stage_in_flight_ = false;
    return check(hipStreamSynchronize(stream_), "upload sync");
}

int BASolver::enqueue_slot(hipEvent_t *ev) {
    hipError_t e;
    View vl = v_;
    if (sharded_) vl.back_part = v_.back_red; // k_linearize reads the all-reduced row
    if (ev) (void)hipEventRecord(ev[0], stream_);
    if ((e = launch_linearize(vl, stream_)) != hipSuccess) return check(e, "k_linearize");
    if (ev) (void)hipEventRecord(ev[1], stream_);
    if ((e = launch_reduce(v_, stream_, sharded_ ? 1 : 0)) != hipSuccess) return check(e, "k_reduce");
    if (ev) (void)hipEventRecord(ev[2], stream_);
    if (sharded_) {
        const size_t n = (size_t)v_.dm.n_tasks * 9 + (size_t)kNumPoseVec * v_.dm.P6 + kNumLinScal;
        // ONE summing all-reduce per linearization: scalar 4 (max |b_l|) travels as one slot per rank behind the scalars
        // (k_reduce fills this rank's slot, zeros the others; k_dense takes the maximum), its summed copy is not used
        if (comm_allreduce(comm_, v_.red, n + (size_t)world_, 0, stream_)) return fail(PVIO_ERR_COMM, "all-reduce failed");
        // the all-reduced system -> tile image (every rank the same): the dense kernel then runs as on one GPU
        if ((e = launch_reduce(v_, stream_, 2)) != hipSuccess) return check(e, "k_reduce (image)");
    }
    if (ev) (void)hipEventRecord(ev[3], stream_);
    // diagnostics (PVIO_HIP_DEBUG_CTRL=1, plain launches only): the control block as each kernel of the slot leaves it
    static const bool dump_ctrl = std::getenv("PVIO_HIP_DEBUG_CTRL") != nullptr;
    auto dump = [&](const char *after) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream_, &cs);
        if (!dump_ctrl || cs != hipStreamCaptureStatusNone) return;
        Ctrl h;
        (void)hipStreamSynchronize(stream_);
        (void)hipMemcpy(&h, v_.ctrl, sizeof h, hipMemcpyDeviceToHost);
        std::fprintf(stderr, "[ctrl after %-11s] mode %d lin_result %d cur %d lin %d iter %d succ %d invalid %d term %d done %d reuse %d solve_ok %d scaling %d trace %d retry %d | radius %.6e mu %.3e x_cost %.9e "
                             "xn2 %.6e %.6e gmax %.3e | g2 %.6e gn2 %.6e gdot %.6e qvv %.6e qvy %.6e qyy %.6e gy %.6e lm_g2 %.6e | ca %.6e cb %.6e sn %.6e mcc %.6e | it %.6e %.3e %.3e %.3e v%d s%d | fin %d %d %.3e\n",
                     after, h.mode, h.lin_result, h.cur, h.lin, h.iter, h.num_success, h.invalid_steps, h.termination, h.done, h.reuse, h.solve_ok, h.scaling_ready, h.trace_len, h.retry_relin, h.radius, h.mu, h.x_cost,
                     h.x_norm2_pose, h.x_norm2_lm, h.grad_max, h.pose_g2, h.pose_gn2, h.pose_gdot, h.pose_qvv, h.pose_qvy, h.pose_qyy, h.pose_gy, h.lm_g2, h.ca, h.cb, h.dogleg_step_norm, h.model_cost_change,
                     h.it_cost, h.it_cost_change, h.it_step_norm, h.it_rel, h.it_valid, h.it_success, h.fin_flags, h.fin_trace_slot, h.fin_lm_gmax);
    };
    dump("k_reduce");
    if ((e = launch_dense(v_, stream_)) != hipSuccess) return check(e, "k_dense");
    dump("k_dense");
    if (ev) (void)hipEventRecord(ev[4], stream_);
    if (!v_.dm.fuse_backsub && (e = launch_backsub(v_, stream_)) != hipSuccess) return check(e, "k_backsub");
    dump("k_backsub");
    if (ev) (void)hipEventRecord(ev[5], stream_);
    if (sharded_) {
        double *back_local = static_cast<double *>(pool_.get("back_local", kNumBackScal * sizeof(double)));
        if ((e = launch_back_reduce(v_, back_local, stream_)) != hipSuccess) return check(e, "k_back_reduce");
        if (comm_allreduce(comm_, v_.back_red, kNumBackScal, 0, stream_)) return fail(PVIO_ERR_COMM, "all-reduce failed");
    }
    if (ev) (void)hipEventRecord(ev[6], stream_);
    return PVIO_OK;
}

int BASolver::run_slots(int n_slots) {
    // A window that has just been uploaded with a new shape is solved with plain launches (they pipeline on the stream: the
    // state machine is on the device); capturing + instantiating a graph costs more than it saves on a single solve and
    // is left to the second solve of the same resident window.
    const bool have_graph = graph_exec_ && graph_slots_ == n_slots;
    // Landmark-sharded runs are captured too (RCCL supports stream capture; eager launches + two collectives per iteration cost
    // 9 257 against 9 779 iterations/s with one rank).  Only ever exercised with a one-rank communicator on this pool, hence the
    // safety net: if capturing or instantiating the sharded graph fails, this context falls back to eager launches for good.
    // PVIO_HIP_SHARDED_GRAPH=0 turns the capture off.
    static const bool sharded_graph = !(std::getenv("PVIO_HIP_SHARDED_GRAPH") != nullptr && std::atoi(std::getenv("PVIO_HIP_SHARDED_GRAPH")) == 0);
    if (use_graph_ && (!sharded_ || (sharded_graph && !sharded_graph_failed_)) && (have_graph || solves_since_upload_ > 0)) {
        bool ok = true;
        if (!have_graph) {
            invalidate_graph();
            if (check(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal), "begin capture")) {
                if (!sharded_) return PVIO_ERR_HIP;
                ok = false;
            }
            if (ok) {
                int rc = PVIO_OK;
                for (int s = 0; s < n_slots && rc == PVIO_OK; ++s) rc = enqueue_slot();
                hipError_t e = hipStreamEndCapture(stream_, &graph_);
                if (rc != PVIO_OK || e != hipSuccess || hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0) != hipSuccess) {
                    if (!sharded_) return rc != PVIO_OK ? rc : fail(PVIO_ERR_HIP, "graph capture / instantiate failed");
                    ok = false;
                } else {
                    graph_slots_ = n_slots;
                }
            }
            if (!ok) { // sharded only: remember, clean up, run this and every later solve eagerly
                (void)hipGetLastError();
                invalidate_graph();
                sharded_graph_failed_ = true;
            }
        }
        if (ok) {
            ++graph_replays_;
            return check(hipGraphLaunch(graph_exec_, stream_), "graph launch");
        }
    }
    for (int s = 0; s < n_slots; ++s) {
        int rc = enqueue_slot();
        if (rc != PVIO_OK) return rc;
    }
    return PVIO_OK;
}

int BASolver::solve(pvio_ba_summary *sum, pvio_ba_kernel_times *prof, pvio_ba_state *read_back) {
    if (!uploaded_) return fail(PVIO_ERR_INVALID_ARGUMENT, "no problem uploaded");
    auto t0 = std::chrono::steady_clock::now();
    if (check(hipSetDevice(device_), "hipSetDevice")) return PVIO_ERR_HIP;
    const Dims &dm = v_.dm;
    const size_t Ns = dm.N, Ms = std::max(dm.M, 1);
    // fused read-back: frame states, inverse depths, quality, valid bytes in one pinned buffer
    const size_t n_pack = Ns * 16 + 2 * (size_t)dm.M + ((size_t)dm.M + 7) / 8;
    double *d_pack = nullptr;
    if (read_back) {
        bool grew = false;
        if (!dev(pool_, "result_pack", n_pack, &d_pack, &grew)) return fail(PVIO_ERR_OUT_OF_MEMORY, "result_pack");
        if (n_pack > h_pack_cap_) {
            if (h_pack_) (void)hipHostFree(h_pack_);
            h_pack_ = nullptr, h_pack_cap_ = 0;
            const size_t cap = n_pack + n_pack / 2;
            if (hipHostMalloc(reinterpret_cast<void **>(&h_pack_), cap * sizeof(double)) != hipSuccess) return fail(PVIO_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
            h_pack_cap_ = cap;
        }
    }
    // trace states are optional and live in a separate buffer so that bench runs do not pay for them
    const bool want_states = sum && sum->trace_states && sum->trace_capacity > 0;
    {
        double *ts = nullptr;
        if (want_states) {
            bool grew = false;
            if (!dev(pool_, "trace_states", (size_t)trace_cap_ * (Ns * 16 + dm.M), &ts, &grew)) return fail(PVIO_ERR_OUT_OF_MEMORY, "trace_states");
        }
        if (ts != v_.trace_states) {
            v_.trace_states = ts;
            invalidate_graph();
        }
    }
    // reset: state buffer 0 <- initial state, user state, control block
    // (one launch: k_reset; the control block's template lives on the device and is sent again only when it changes)
    {
        Ctrl t;
        std::memset(&t, 0, sizeof t);
        t.mode = MODE_INIT;
        t.radius = 1e4;        // initial_trust_region_radius
        t.mu = 1e-8;           // DoglegStrategy kMinMu
        t.termination = PVIO_TERM_NO_CONVERGENCE;
        t.trace_cap = trace_cap_;
        t.dbg_fail_left = dbg_fail_, t.dbg_invalid_left = dbg_invalid_;
        bool grew = false;
        Ctrl *d_tmpl = nullptr;
        if (!dev(pool_, "ctrl_template", 1, &d_tmpl, &grew)) return fail(PVIO_ERR_OUT_OF_MEMORY, "ctrl template");
        if (grew || d_tmpl != d_ctrl_tmpl_ || std::memcmp(&t, &h_ctrl_tmpl_, sizeof t) != 0) {
            h_ctrl_tmpl_ = t, d_ctrl_tmpl_ = d_tmpl;
            // (pageable source: the copy is staged by the runtime before the call returns)
            if (check(hipMemcpyAsync(d_tmpl, &h_ctrl_tmpl_, sizeof t, hipMemcpyHostToDevice, stream_), "ctrl template")) return PVIO_ERR_HIP;
        }
        if (check(launch_reset(v_, fs_init_, rho_init_, d_tmpl, stream_), "k_reset")) return PVIO_ERR_HIP;
    }
    if (check(hipEventRecord(ev0_, stream_), "event")) return PVIO_ERR_HIP;
    // every slot = one pass of [linearize, reduce, dense, backsub]; iteration 0 + max_iter iterations; relaunch while the device
    // has not reported done
    // (no slack slot: a solve whose mu escalates needs more slots than iterations and simply gets a second replay from the loop below;
    // every other solve saved four no-op launches per replay)
    // A real-time configuration (a limit far below the reference's default of 1e6 s, config.cpp:86-88) looks at the clock every two
    // slots instead of once per solve: Ceres tests its limit every iteration
    const bool time_limited = max_solver_time_ > 0 && max_solver_time_ < 1.0e5;
    const int n_slots = time_limited ? std::min(dm.max_iter + 1, 2) : dm.max_iter + 1;
    int rounds = 0;
    hipEvent_t pev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    long long *dbg_saved = v_.dbg;
    bool prof_graph = false;
    if (prof) {
        std::memset(prof, 0, sizeof *prof);
        for (auto &e : pev) (void)hipEventCreate(&e);
        bool grew = false;
        long long *dbg = nullptr;
        if (!dev(pool_, "dbg", 4 * 32, &dbg, &grew)) return fail(PVIO_ERR_OUT_OF_MEMORY, "dbg");
        (void)hipMemsetAsync(dbg, 0, 4 * 32 * sizeof(long long), stream_);
        // PVIO_HIP_STAMP_SEL: -1 (default) every stamp site stores, k >= 0 only site k (one store per launch), -2 none (events only)
        const char *sel = std::getenv("PVIO_HIP_STAMP_SEL");
        v_.dbg_sel = sel ? std::atoi(sel) : -1;
        v_.dbg = v_.dbg_sel == -2 ? nullptr : dbg;
        // PVIO_HIP_PROFILE_GRAPH=1: the stamps are taken inside a REPLAY of the slot graph (kernels back to back, as in a normal
        // solve) instead of eager launches with a host synchronization per slot; no per-launch events then
        prof_graph = std::getenv("PVIO_HIP_PROFILE_GRAPH") != nullptr && std::atoi(std::getenv("PVIO_HIP_PROFILE_GRAPH")) != 0;
        if (prof_graph) invalidate_graph(); // the graph bakes the View in: capture one that carries the stamp buffer
    }
    for (;;) {
        int rc;
        if (prof && !prof_graph) { // one slot at a time, events around every launch; counts only slots that did work
            rc = enqueue_slot(pev);
            if (rc == PVIO_OK && check(hipStreamSynchronize(stream_), "profile sync")) rc = PVIO_ERR_HIP;
            if (rc == PVIO_OK) {
                const int ia[4] = {0, 1, 3, 4}, ib[4] = {1, 2, 4, 5};
                for (int k = 0; k < 4; ++k) {
                    float ms1 = 0;
                    (void)hipEventElapsedTime(&ms1, pev[ia[k]], pev[ib[k]]);
                    prof->total_ms[k] += ms1;
                    prof->launches[k] += 1;
                }
                if (sharded_) { // the two exchange steps: between k_reduce and k_dense, behind k_backsub
                    const int ca[2] = {2, 5}, cb[2] = {3, 6};
                    for (int k = 0; k < 2; ++k) {
                        float ms1 = 0;
                        (void)hipEventElapsedTime(&ms1, pev[ca[k]], pev[cb[k]]);
                        prof->comm_ms[k] += ms1;
                        prof->comm_launches[k] += 1;
                    }
                }
            }
            --rounds; // the round cap below is for graph replays
        } else {
            rc = run_slots(n_slots);
        }
        if (rc != PVIO_OK) return rc;
        const auto t_enq = std::chrono::steady_clock::now();
        if (check(hipMemcpyAsync(h_ctrl_, v_.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, stream_), "ctrl D2H")) return PVIO_ERR_HIP;
        if (check(hipEventRecord(ev1_, stream_), "event")) return PVIO_ERR_HIP;
        if (read_back) { // speculative: all but never is the state machine still running after one replay (then this is repeated)
            if (check(launch_quality(v_, stream_, 1, nullptr, d_pack), "k_quality")) return PVIO_ERR_HIP;
            if (check(hipMemcpyAsync(h_pack_, d_pack, n_pack * sizeof(double), hipMemcpyDeviceToHost, stream_), "results D2H")) return PVIO_ERR_HIP;
        }
        if (check(hipStreamSynchronize(stream_), "solve sync")) return PVIO_ERR_HIP;
        stage_in_flight_ = false;
        {   // PVIO_HIP_TIMING=1: where the host's time of a solve goes (enqueueing the slots against waiting for them)
            static const bool timing = std::getenv("PVIO_HIP_TIMING") != nullptr;
            if (timing)
                std::fprintf(stderr, "[pvio-hip] solve host: reset + %d slots enqueued after %.0f us (%s), stream drained after %.0f us\n", n_slots,
                             std::chrono::duration<double, std::micro>(t_enq - t0).count(), graph_exec_ && graph_slots_ == n_slots ? "graph" : "plain launches",
                             std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        if (h_ctrl_->done || ++rounds > (time_limited ? 16 * (dm.max_iter + 1) : 16)) break;
        // max_solver_time_in_seconds (solver_options.h:30): the state machine runs on the device, so the wall clock is looked at
        // between replays of the slot graph only (one replay covers every iteration of an ordinary solve): NO_CONVERGENCE at
        // the iterate reached, like Ceres' time-limit exit
        if (max_solver_time_ > 0) {
            int timed_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > max_solver_time_ ? 1 : 0;
            // tests: only rank PVIO_HIP_DEBUG_TIMEOUT_RANK sees its clock run out (the ranks must still leave in the same round)
            static const int dbg_timeout_rank = std::getenv("PVIO_HIP_DEBUG_TIMEOUT_RANK") ? std::atoi(std::getenv("PVIO_HIP_DEBUG_TIMEOUT_RANK")) : -1;
            if (dbg_timeout_rank >= 0) timed_out = rank_ == dbg_timeout_rank ? 1 : 0;
            if (sharded_) {
                // every rank looks at its OWN clock, but the next replay contains collectives: a rank that left alone would leave the
                // others waiting in an all-reduce for good.  `done` is identical on all ranks (identical all-reduced inputs),
                // so they all arrive here in the same round; the decision is all-reduced (max) before anybody acts on it.
                double *flag = static_cast<double *>(pool_.get("timeout_flag", sizeof(double)));
                double hflag = timed_out;
                if (!flag || check(hipMemcpyAsync(flag, &hflag, sizeof(double), hipMemcpyHostToDevice, stream_), "timeout flag H2D")) return PVIO_ERR_HIP;
                if (comm_allreduce(comm_, flag, 1, 1, stream_)) return fail(PVIO_ERR_COMM, "all-reduce failed");
                if (check(hipMemcpyAsync(&hflag, flag, sizeof(double), hipMemcpyDeviceToHost, stream_), "timeout flag D2H")) return PVIO_ERR_HIP;
                if (check(hipStreamSynchronize(stream_), "timeout flag sync")) return PVIO_ERR_HIP;
                timed_out = hflag > 0.0;
            }
            if (timed_out) {
                h_ctrl_->done = 1, h_ctrl_->termination = PVIO_TERM_NO_CONVERGENCE;
                break;
            }
        }
    }
    if (prof) {
        for (auto &e : pev) (void)hipEventDestroy(e);
        if (v_.dbg) (void)hipMemcpy(prof->phase_ticks, v_.dbg, 4 * 32 * sizeof(long long), hipMemcpyDeviceToHost);
        v_.dbg = dbg_saved;
        if (prof_graph) invalidate_graph(); // the next solve captures a graph without the stamp buffer again
    }
    ++solves_since_upload_;
    if (!h_ctrl_->done) return fail(PVIO_ERR_HIP, "device state machine did not terminate");
    last_repeats_ = h_ctrl_->cand_repeats;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0_, ev1_);
    if (sum) {
        sum->termination = h_ctrl_->termination;
        sum->is_usable = h_ctrl_->termination != PVIO_TERM_FAILURE;
        sum->num_iterations = h_ctrl_->iter;
        sum->num_successful_steps = h_ctrl_->num_success;
        sum->initial_cost = h_ctrl_->initial_cost;
        sum->final_cost = h_ctrl_->x_cost;
        sum->device_seconds = ms * 1e-3;
        sum->trace_len = 0;
        if (sum->trace && sum->trace_capacity > 0) {
            const int n = std::min(std::min(h_ctrl_->trace_len, trace_cap_), sum->trace_capacity);
            std::vector<TraceRec> tr(n);
            if (n && check(hipMemcpy(tr.data(), v_.trace, n * sizeof(TraceRec), hipMemcpyDeviceToHost), "trace D2H")) return PVIO_ERR_HIP;
            static_assert(sizeof(TraceRec) == sizeof(pvio_ba_iteration), "trace layout");
            std::memcpy(sum->trace, tr.data(), n * sizeof(TraceRec));
            sum->trace_len = n;
            if (want_states && n)
                if (check(hipMemcpy(sum->trace_states, v_.trace_states, (size_t)n * (Ns * 16 + dm.M) * sizeof(double), hipMemcpyDeviceToHost), "trace states D2H")) return PVIO_ERR_HIP;
        }
        sum->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (read_back) { // same semantics as download(): invalidated landmarks keep their old quality (bundle_adjustor.cpp:294)
        const size_t M = dm.M;
        if (read_back->frame_state) std::memcpy(read_back->frame_state, h_pack_, Ns * 16 * sizeof(double));
        if (M && read_back->lm_inv_depth) std::memcpy(read_back->lm_inv_depth, h_pack_ + Ns * 16, M * sizeof(double));
        const double *q = h_pack_ + Ns * 16 + M;
        const unsigned char *val = reinterpret_cast<const unsigned char *>(h_pack_ + Ns * 16 + 2 * M);
        for (size_t l = 0; l < M; ++l) {
            if (read_back->lm_valid) read_back->lm_valid[l] = val[l];
            if (read_back->lm_quality && val[l]) read_back->lm_quality[l] = q[l];
        }
    }
    return PVIO_OK;
}

int BASolver::download(pvio_ba_state *st) {
    if (!uploaded_ || !st) return fail(PVIO_ERR_INVALID_ARGUMENT, "nothing to download");
    const Dims &dm = v_.dm;
    const int cur = h_ctrl_->cur;
    if (check(launch_quality(v_, stream_, 1, nullptr), "k_quality")) return PVIO_ERR_HIP;
    if (check(hipMemcpyAsync(st->frame_state, v_.fs + (size_t)cur * dm.N * 16, (size_t)dm.N * 16 * sizeof(double), hipMemcpyDeviceToHost, stream_), "D2H")) return PVIO_ERR_HIP;
    if (dm.M) {
        if (check(hipMemcpyAsync(st->lm_inv_depth, v_.rho + (size_t)cur * dm.M, (size_t)dm.M * sizeof(double), hipMemcpyDeviceToHost, stream_), "D2H")) return PVIO_ERR_HIP;
        std::vector<double> q;
        std::vector<uint8_t> val;
        if (st->lm_quality || st->lm_valid) {
            q.resize(dm.M), val.resize(dm.M);
            if (check(hipMemcpyAsync(q.data(), v_.lm_quality, (size_t)dm.M * sizeof(double), hipMemcpyDeviceToHost, stream_), "D2H")) return PVIO_ERR_HIP;
            if (check(hipMemcpyAsync(val.data(), v_.lm_valid, (size_t)dm.M, hipMemcpyDeviceToHost, stream_), "D2H")) return PVIO_ERR_HIP;
            if (check(hipStreamSynchronize(stream_), "sync")) return PVIO_ERR_HIP;
            for (int l = 0; l < dm.M; ++l) {
                if (st->lm_valid) st->lm_valid[l] = val[l];
                if (st->lm_quality && val[l]) st->lm_quality[l] = q[l]; // invalidated landmarks keep their old quality (:294)
            }
        }
    }
    return check(hipStreamSynchronize(stream_), "download sync");
}

int BASolver::reprojection_error(double *out) {
    if (!uploaded_ || !out) return fail(PVIO_ERR_INVALID_ARGUMENT, "nothing uploaded");
    double *acc = static_cast<double *>(pool_.get("err_acc", 2 * sizeof(double)));
    if (!acc) return fail(PVIO_ERR_OUT_OF_MEMORY, "err_acc");
    if (check(hipMemsetAsync(acc, 0, 2 * sizeof(double), stream_), "memset")) return PVIO_ERR_HIP;
    // evaluates the uploaded initial state (buffer 0 after a reset)
    const size_t Ns = v_.dm.N, Ms = std::max(v_.dm.M, 1);
    const double *fs_init = fs_init_, *rho_init = rho_init_;
    if (check(hipMemcpyAsync(v_.fs, fs_init, Ns * 16 * sizeof(double), hipMemcpyDeviceToDevice, stream_), "reset fs")) return PVIO_ERR_HIP;
    if (v_.dm.M && check(hipMemcpyAsync(v_.rho, rho_init, (size_t)v_.dm.M * sizeof(double), hipMemcpyDeviceToDevice, stream_), "reset rho")) return PVIO_ERR_HIP;
    if (check(launch_quality(v_, stream_, 0, acc), "k_quality")) return PVIO_ERR_HIP;
    double h[2];
    if (check(hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, stream_), "D2H")) return PVIO_ERR_HIP;
    if (check(hipStreamSynchronize(stream_), "sync")) return PVIO_ERR_HIP;
    *out = h[0] / std::max(h[1], 1.0);
    return PVIO_OK;
}

} // namespace pvba

namespace pvba {

static bool lu_inverse15(const double *Ain, double *inv) { // Eigen's .inverse() for a 15 x 15 block is PartialPivLU based
    const int n = 15;
    double A[225];
    int piv[15];
    std::memcpy(A, Ain, sizeof A);
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(A[i * n + k]) > best) best = std::fabs(A[i * n + k]), p = i;
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[p * n + j]);
            std::swap(piv[k], piv[p]);
        }
        for (int i = k + 1; i < n; ++i) {
            const double f = A[i * n + k] / A[k * n + k];
            A[i * n + k] = f;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
        }
    }
    for (int col = 0; col < n; ++col) {
        double x[15];
        for (int i = 0; i < n; ++i) x[i] = piv[i] == col ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < i; ++k) x[i] -= A[i * n + k] * x[k];
        for (int i = n - 1; i >= 0; --i) {
            for (int k = i + 1; k < n; ++k) x[i] -= A[i * n + k] * x[k];
            x[i] /= A[i * n + i];
        }
        for (int i = 0; i < n; ++i) inv[i * n + col] = x[i];
    }
    return true;
}

// BundleAdjustor::marginalize_frame (bundle_adjustor.cpp:348-599).  The O(F) part -- un-robustified J^T J / J^T r of
// every reprojection factor of the victim's landmarks, the scalar landmark elimination, the prior and the two IMU
// factors -- runs on the GPU through k_linearize (MODE_MARG) + k_reduce; the 15N-dimensional dense tail (victim block
// elimination, eigen-decomposition) is a few hundred kflop and stays on the host.
int BASolver::marginalize(const pvio_ba_problem *pb, const pvio_ba_state *st, int victim, pvio_ba_prior *out) {
    if (!pb || !st || !out || !out->S || !out->s) return fail(PVIO_ERR_INVALID_ARGUMENT, "null argument");
    const int N = pb->n_frames;
    if (victim < 0 || victim >= N || N < 2) return fail(PVIO_ERR_INVALID_ARGUMENT, "victim frame out of range");
    // marginalization always works on the full 15-dim error state and ignores FF_FIX_POSE; IMU factors are used
    // whenever they exist (:416-450), regardless of how the last solve was configured
    static const bool timing = std::getenv("PVIO_HIP_TIMING") != nullptr; // diagnostics: where a marginalization's time goes
    const auto tm0 = std::chrono::steady_clock::now();
    pvio_ba_problem p2 = *pb;
    p2.use_inertial = 1;
    int rc = upload(&p2, st, /*may_return_early=*/true); // (this call synchronizes below, before the caller's arrays can change)
    if (rc != PVIO_OK) return rc;
    const auto tm1 = std::chrono::steady_clock::now();
    const Dims &dm = v_.dm;
    const size_t Ns = N;
    const double *fs_init = fs_init_, *rho_init = rho_init_;
    // working states <- the uploaded ones, in one launch (the control block it also copies is its own content: overwritten right below)
    if (check(launch_reset(v_, fs_init, rho_init, v_.ctrl, stream_), "k_reset")) return PVIO_ERR_HIP;
    std::memset(h_ctrl_, 0, sizeof(Ctrl));
    h_ctrl_->mode = MODE_MARG;
    h_ctrl_->marg_victim = victim;
    h_ctrl_->mu = 0.0;
    if (check(hipMemcpyAsync(v_.ctrl, h_ctrl_, sizeof(Ctrl), hipMemcpyHostToDevice, stream_), "ctrl")) return PVIO_ERR_HIP;
    hipError_t e;
    if ((e = launch_linearize(v_, stream_)) != hipSuccess) return check(e, "k_linearize");
    if ((e = launch_reduce(v_, stream_, sharded_ ? 1 : 0)) != hipSuccess) return check(e, "k_reduce");
    const size_t nS = (size_t)dm.n_tasks * 9, nV = (size_t)kNumPoseVec * dm.P6;
    // landmark shards: every rank holds the victim's landmarks of its own range only -> sum the reduced buffers (the IMU
    // factors, the old prior and the rotation prior are replicated and are added once, below, on every rank alike)
    if (sharded_ && comm_allreduce(comm_, v_.red, nS + nV + kNumLinScal + (size_t)world_, 0, stream_)) return fail(PVIO_ERR_COMM, "all-reduce failed");
    // read-back as ONE copy into a pinned buffer: a D2H copy into pageable memory is staged synchronously by the runtime (about 20 us
    // each, eight of them); the eight arrays are gathered on the device first (k_gather)
    const size_t Dp = 15 * (size_t)dm.prior_n;
    const size_t n_back[8] = {nS + nV + kNumLinScal, Ns * 900, Ns * 30, Dp * Dp, Dp, Ns * 9, Ns * 3, ((size_t)dm.n_tasks + 1) / 2};
    size_t off_back[9] = {0};
    for (int k = 0; k < 8; ++k) off_back[k + 1] = off_back[k] + ((n_back[k] + 7) & ~(size_t)7);
    if (off_back[8] > h_back_cap_) {
        if (h_back_) (void)hipHostFree(h_back_);
        h_back_ = nullptr, h_back_cap_ = 0;
        const size_t cap = off_back[8] + off_back[8] / 2;
        if (hipHostMalloc(reinterpret_cast<void **>(&h_back_), cap * sizeof(double)) != hipSuccess) return fail(PVIO_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
        h_back_cap_ = cap;
    }
    const double *red = h_back_ + off_back[0], *preH = h_back_ + off_back[1], *preg = h_back_ + off_back[2], *priH = h_back_ + off_back[3];
    const double *prig = h_back_ + off_back[4], *rotH = h_back_ + off_back[5], *rotg = h_back_ + off_back[6];
    const int32_t *tasks = reinterpret_cast<const int32_t *>(h_back_ + off_back[7]);
    const void *src_back[8] = {v_.red, v_.pre_H, v_.pre_g, v_.prior_H, v_.prior_g, v_.rot_H, v_.rot_g, v_.task_desc};
    GatherArgs ga;
    for (int k = 0; k < 8; ++k) {
        ga.src[k] = src_back[k];
        ga.words[k] = (uint32_t)(k == 7 ? (size_t)dm.n_tasks : 2 * n_back[k]);
        ga.off[k] = (uint32_t)(2 * off_back[k]);
    }
    double *d_back = nullptr;
    {
        bool grew = false;
        if (!dev(pool_, "marg_back", off_back[8], &d_back, &grew)) return fail(PVIO_ERR_OUT_OF_MEMORY, "marg_back");
    }
    if (check(launch_gather(ga, d_back, stream_), "k_gather")) return PVIO_ERR_HIP;
    if (check(hipMemcpyAsync(h_back_, d_back, off_back[8] * sizeof(double), hipMemcpyDeviceToHost, stream_), "D2H")) return PVIO_ERR_HIP;
    if (check(hipStreamSynchronize(stream_), "marginalize sync")) return PVIO_ERR_HIP;
    stage_in_flight_ = false;
    const auto tm2 = std::chrono::steady_clock::now();

    // ---- assemble the 15N information matrix / vector ----
    const int D = 15 * N;
    // (the large work arrays are members: a fresh 180 KB vector per call is an mmap, its page faults and an munmap)
    std::vector<double> &H = marg_H_, &C = marg_C_, &V = marg_V_;
    std::vector<double> b(D, 0.0);
    H.assign((size_t)D * D, 0.0);
    for (int t = 0; t < dm.n_tasks; ++t) {
        const int fi = tasks[t] & 255, fj = (tasks[t] >> 8) & 255, si = (tasks[t] >> 16) & 1, sj = (tasks[t] >> 17) & 1;
        for (int el = 0; el < 9; ++el) {
            const int r = 15 * fi + 3 * si + el / 3, c = 15 * fj + 3 * sj + el % 3;
            const double val = red[(size_t)el * dm.n_tasks + t];
            if (fi != fj) H[(size_t)r * D + c] = H[(size_t)c * D + r] = val;
            else if (r >= c) H[(size_t)r * D + c] = H[(size_t)c * D + r] = val; // only the lower half of a diagonal block is complete
        }
    }
    for (int f = 0; f < N; ++f)
        for (int k = 0; k < 6; ++k) b[15 * f + k] = red[nS + 6 * f + k] - red[nS + dm.P6 + 6 * f + k]; // J^T r - sum_l W_l^T b_l / H_ll
    for (int j = victim; j <= victim + 1; ++j) {
        if (j == 0 || j >= N || !(pb->preint_valid && pb->preint_valid[j])) continue;
        for (int a = 0; a < 30; ++a) {
            b[15 * (j - 1) + a] += preg[(size_t)j * 30 + a];
            for (int c = 0; c < 30; ++c) H[(size_t)(15 * (j - 1) + a) * D + 15 * (j - 1) + c] += preH[(size_t)j * 900 + a * 30 + c];
        }
    }
    for (int a = 0; a < 3; ++a) { // the victim's rotation prior (evaluated un-gated in MODE_MARG; zero when it has none)
        b[15 * victim + a] += rotg[(size_t)victim * 3 + a];
        for (int c = 0; c < 3; ++c) H[(size_t)(15 * victim + a) * D + 15 * victim + c] += rotH[(size_t)victim * 9 + 3 * a + c];
    }
    std::vector<int> gprior(Dp); // prior coordinate -> window coordinate (a division per ELEMENT of the 135 x 135 block cost 40 us here)
    for (size_t a = 0; a < Dp; ++a) gprior[a] = 15 * pb->prior_frames[a / 15] + (int)(a % 15);
    for (size_t a = 0; a < Dp; ++a) {
        const int ga = gprior[a];
        b[ga] += prig[a];
        double *hrow = &H[(size_t)ga * D];
        const double *prow = &priH[a * Dp];
        for (size_t c = 0; c < Dp; ++c) hrow[gprior[c]] += prow[c];
    }
    // ---- eliminate the victim's 15 x 15 block (:547-581) ----
    const int R = D - 15;
    double Hvv[225], Hinv[225];
    for (int x = 0; x < 15; ++x)
        for (int y = 0; y < 15; ++y) Hvv[x * 15 + y] = H[(size_t)(15 * victim + x) * D + 15 * victim + y];
    if (!lu_inverse15(Hvv, Hinv)) return fail(PVIO_ERR_INVALID_ARGUMENT, "singular victim block");
    auto gidx = [&](int k) { return k < 15 * victim ? k : k + 15; };
    std::vector<double> cv(R, 0.0), T((size_t)R * 15);
    C.assign((size_t)R * R, 0.0);
    for (int i = 0; i < R; ++i) { // T = H_rv Hvv^-1: fifteen independent sums per row, each still taken over x in ascending order
        const double *hiv = &H[(size_t)gidx(i) * D + 15 * victim];
        double trow[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int x = 0; x < 15; ++x) {
            const double hx = hiv[x];
            for (int y = 0; y < 15; ++y) trow[y] += hx * Hinv[x * 15 + y];
        }
        for (int y = 0; y < 15; ++y) T[(size_t)i * 15 + y] = trow[y];
    }
    const int split = 15 * victim;
    // the victim's rows with the victim's own columns removed, so that the inner loops below run over contiguous memory (the sums are
    // taken in the same order as a loop over y per element would take them; that form cost 100 us per marginalization)
    std::vector<double> Hv((size_t)15 * R), acc(R);
    for (int y = 0; y < 15; ++y)
        for (int j = 0; j < R; ++j) Hv[(size_t)y * R + j] = H[(size_t)(15 * victim + y) * D + gidx(j)];
    for (int i = 0; i < R; ++i) {
        double s = 0;
        for (int y = 0; y < 15; ++y) s += T[(size_t)i * 15 + y] * b[15 * victim + y];
        cv[i] = b[gidx(i)] - s;
        const int j0 = i >= split ? split : 0; // lower-left = transpose of upper-right (:572-576), filled in below
        for (int j = j0; j < R; ++j) acc[j] = 0;
        for (int y = 0; y < 15; ++y) {
            const double ty = T[(size_t)i * 15 + y];
            const double *hv = &Hv[(size_t)y * R];
            for (int j = j0; j < R; ++j) acc[j] += ty * hv[j];
        }
        const double *hi = &H[(size_t)gidx(i) * D];
        double *ci = &C[(size_t)i * R];
        for (int j = j0; j < split; ++j) ci[j] = hi[j] - acc[j];
        for (int j = j0 > split ? j0 : split; j < R; ++j) ci[j] = hi[j + 15] - acc[j];
    }
    for (int i = split; i < R; ++i)
        for (int j = 0; j < split; ++j) C[(size_t)i * R + j] = C[(size_t)j * R + i];
    if (out->info_matrix) std::memcpy(out->info_matrix, C.data(), sizeof(double) * R * R);
    if (out->info_vector) std::memcpy(out->info_vector, cv.data(), sizeof(double) * R);
    // ---- sqrt information: sqrt(L) V^T and L^-1/2 V^T b, eigenvalues <= 1e-8 zeroed (:583-590) ----
    // Coordinates without any information -- an exactly zero row and column: velocity / biases of a frame no IMU factor or prior reaches --
    // are eigenvectors of eigenvalue 0 by themselves and stay out of the eigen-problem: a backward-stable solver would hand them back with
    // an eigenvalue of a few eps |C| and a few eps of every other coordinate mixed in, which the 1e-8 cut then keeps or drops as the rounding
    // falls (profiles/NOTES_r1_r3.md section 2a).  Their rows of S are zero (they come first, like the zero eigenvalues of the full problem).
    std::vector<int> keep;
    keep.reserve(R);
    for (int i = 0; i < R; ++i) {
        bool zero = true;
        for (int j = 0; j < R && zero; ++j) zero = C[(size_t)i * R + j] == 0.0 && C[(size_t)j * R + i] == 0.0;
        if (!zero) keep.push_back(i);
    }
    const int Rk = (int)keep.size(), nz = R - Rk;
    std::vector<double> w(std::max(Rk, 1));
    V.resize((size_t)std::max(Rk, 1) * std::max(Rk, 1));
    const auto tm3 = std::chrono::steady_clock::now();
    if (nz > 0) { // compress the kept coordinates to the front of C's storage (the copies above have been made)
        for (int a = 0; a < Rk; ++a)
            for (int b = 0; b < Rk; ++b) C[(size_t)a * Rk + b] = C[(size_t)keep[a] * R + keep[b]];
    }
    if (Rk > 0) sym_eig(C.data(), Rk, w.data(), V.data());
    const auto tm4 = std::chrono::steady_clock::now();
    out->n = N - 1;
    std::fill(out->S, out->S + (size_t)R * R, 0.0);
    std::fill(out->s, out->s + R, 0.0);
    for (int k = 0; k < Rk; ++k) {
        const double lam = w[k] > 1.0e-8 ? w[k] : 0.0, lam_inv = w[k] > 1.0e-8 ? 1.0 / w[k] : 0.0;
        const double sl = std::sqrt(lam), sli = std::sqrt(lam_inv);
        double acc = 0;
        double *Srow = out->S + (size_t)(nz + k) * R;
        for (int i = 0; i < Rk; ++i) {
            Srow[keep[i]] = sl * V[(size_t)k * Rk + i];
            acc += V[(size_t)k * Rk + i] * cv[keep[i]];
        }
        out->s[nz + k] = sli * acc;
    }
    uploaded_ = false; // the device buffers now hold a marginalization pass, not a solvable window
    if (timing) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "[pvio-hip] marginalize: upload %.0f us, kernels + read-back %.0f us, assembly + elimination %.0f us, eigen-decomposition (%d x %d) %.0f us, sqrt-information %.0f us\n",
                     us(tm0, tm1), us(tm1, tm2), us(tm2, tm3), Rk, Rk, us(tm3, tm4), us(tm4, std::chrono::steady_clock::now()));
    }
    return PVIO_OK;
}

} // namespace pvba
