// bundle_adjustor.cpp -- pvio::BundleAdjustor on top of the HIP C-ABI (include/pvio_hip.h).
//
// Drop-in for pvio/src/pvio/estimation/bundle_adjustor.cpp: same class (pimpl included, bundle_adjustor.h:29-42), same
// three methods, same in-place semantics (states are read from and written back into Frame::pose/motion and
// Track::landmark, flags and Plane::tracks are updated like :251-296), same return value (IsSolutionUsable(), :298), no
// exceptions.  What it does NOT contain any more: ceres::Problem assembly and ceres::Solve -- the Map is flattened, in the
// reference's residual-block order, into a pvio_ba_problem and handed to the GPU.
//
// ONE source, two builds (host_seam.h): against the reference's real headers inside the PVIO tree
// (-DPVIO_HOST_USE_REFERENCE_TYPES; `make -C tests/host refcheck` proves that it compiles there), and against the
// look-alike declarations of pvio_min.h for the standalone tests.  Only members that exist in the reference are used:
// frame->image->t, get_preintegration_factor(), Track::flag(f) = v, Factor::create_marginalization_error,
// Map::set/get_marginalization_factor, MarginalizationErrorCost::related_frames() (+ the four accessors of the Ceres-free
// holder, dropin/.../marginalization_error_cost.h), Track::try_triangulate / set_landmark_point / get_landmark_point,
// PlaneExtractor::enough_baseline, Frame::get_pose.  Eigen objects are touched through (i, j), [i], .data() of vectors,
// .coeffs(), .dot() only; every matrix crosses the C ABI element by element (the ABI is row-major, Eigen column-major).
#include "host_seam.h"

#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <algorithm>
#include <cstdint>
#include <unordered_map>
#include <unordered_set>

#include "../../include/pvio_hip.h"

namespace pvio {

namespace {

// One GPU context per process: the reference constructs a BundleAdjustor temporary per call
// (sliding_window_tracker.cpp:113), so the context must outlive the object.
pvio_hip_ctx *process_ctx() {
    static pvio_hip_ctx *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        pvio_hip_opts o;
        std::memset(&o, 0, sizeof o);
        o.world_size = 1, o.use_graph = 1;
        // A candidate that is bit-identical to the one just rejected is not evaluated again (include/pvio_hip.h): same iterations, records and
        // results as the reference, which does re-evaluate it; PVIO_HIP_REUSE_CANDIDATES=0 turns it off.
        const char *reuse = std::getenv("PVIO_HIP_REUSE_CANDIDATES");
        o.reuse_identical_candidates = (reuse && std::atoi(reuse) == 0) ? 0 : 1;
        if (const char *dev = std::getenv("PVIO_HIP_DEVICE")) o.device = std::atoi(dev); // one ctx per process and GPU
        if (pvio_hip_abi_version() != PVIO_HIP_ABI_VERSION) { // pvio_hip_opts is copied by value: a library built against another header reads another layout
            std::fprintf(stderr, "[pvio-hip] libpvio_hip.so has ABI version %d, this adapter was built against %d\n", (int)pvio_hip_abi_version(), PVIO_HIP_ABI_VERSION);
            ctx = nullptr;
            return;
        }
        if (pvio_hip_create(&o, &ctx) != PVIO_OK) {
            std::fprintf(stderr, "[pvio-hip] no usable GPU context: BundleAdjustor::solve will report failure\n");
            ctx = nullptr;
        }
    });
    return ctx;
}

struct Flat { // owns the arrays a pvio_ba_problem points into
    std::vector<uint8_t> frame_fixed, pre_valid;
    std::vector<double> cam, imu, sic, intr, fstate, lm_z, obs_z, rho, pre_delta, pre_U, pre_jac, prior_S, prior_s, prior_lin, plane_z, plane_n, plane_d;
    std::vector<int32_t> lm_anchor, lm_ptr, obs_frame, prior_frames, plane_ptr, plane_frame, lm_mult;
    std::vector<Track *> lm_track; // landmark index -> track (write-back)
    std::vector<double> quality;
    std::vector<uint8_t> valid;
    pvio_ba_problem pb;
};

// Pointer-keyed open-addressing tables for the two lookups the flattening does per observation (frame -> window index) and per
// keypoint (track already listed?): std::unordered_map / unordered_set cost a node allocation per insert and a pointer
// chase per probe, which made the flattening as expensive as the upload it feeds (measured 225 us for 10 x 300).
inline size_t ptr_hash(const void *p) {
    uint64_t x = (uint64_t)(uintptr_t)p;
    x ^= x >> 17, x *= 0x9E3779B97F4A7C15ull, x ^= x >> 29;
    return (size_t)x;
}
constexpr int kMaxWindow = 64;
class FrameIndex { // <= kMaxFrames entries
  public:
    void clear() {
        for (auto &k : key_) k = nullptr;
    }
    void put(const Frame *f, int idx) {
        size_t h = ptr_hash(f) & (kSlots - 1);
        while (key_[h] && key_[h] != f) h = (h + 1) & (kSlots - 1);
        key_[h] = f, val_[h] = idx;
    }
    int find(const Frame *f) const { // -1 when the frame is not in the window
        size_t h = ptr_hash(f) & (kSlots - 1);
        while (key_[h]) {
            if (key_[h] == f) return val_[h];
            h = (h + 1) & (kSlots - 1);
        }
        return -1;
    }

  private:
    static constexpr size_t kSlots = 256;
    const Frame *key_[kSlots] = {};
    int val_[kSlots] = {};
};
class PtrSet {
  public:
    void reset(size_t expected) {
        size_t n = 64;
        while (n < 2 * expected + 16) n <<= 1;
        if (slots_.size() != n) slots_.assign(n, nullptr);
        else std::fill(slots_.begin(), slots_.end(), nullptr);
        used_ = 0;
    }
    bool insert(const void *p) { // true when p was not in the set
        if (2 * (used_ + 1) > slots_.size()) grow();
        const size_t mask = slots_.size() - 1;
        size_t h = ptr_hash(p) & mask;
        while (slots_[h]) {
            if (slots_[h] == p) return false;
            h = (h + 1) & mask;
        }
        slots_[h] = p, ++used_;
        return true;
    }
    bool contains(const void *p) const {
        const size_t mask = slots_.size() - 1;
        size_t h = ptr_hash(p) & mask;
        while (slots_[h]) {
            if (slots_[h] == p) return true;
            h = (h + 1) & mask;
        }
        return false;
    }

  private:
    void grow() {
        std::vector<const void *> old;
        old.swap(slots_);
        slots_.assign(old.size() * 2, nullptr);
        used_ = 0;
        for (const void *p : old)
            if (p) insert(p);
    }
    std::vector<const void *> slots_ = std::vector<const void *>(64, nullptr);
    size_t used_ = 0;
};

// ---- incremental flattening -------------------------------------------------------------
// From one keyframe solve to the next almost every track is the track it was, plus one observation.  Walking its
// std::map of observations again (pointer chasing through the reference's object graph: 215 us of a 1.1 ms keyframe solve
// at 10 x 1000) is replaced by a per-track cache of the flattened list -- (frame id, keypoint) of every non-anchor
// observation, in keypoint_map() order -- validated by a signature that every way the reference changes a track alters:
// (track id, number of keypoints, first frame id, last frame id).  Hit: the list is replayed (frame ids -> window indices by
// a merge, both ascending).  Grown by exactly the newest observation: that one is appended.  Anything else: rebuilt.
// No hooks in the reference's map layer are needed; the output is identical to a walk from scratch
// (PVIO_HIP_FLATTEN_VERIFY=1 re-walks every window and compares; the headless tests run with it).
class ObsCache {
  public:
    struct Entry {
        const Track *key = nullptr;
        size_t id = 0, nkp = 0, first_id = 0, last_id = 0;
        uint32_t off = 0, cnt = 0;
    };
    Entry *find(const Track *t) {
        if (slots_.empty()) return nullptr;
        const size_t mask = slots_.size() - 1;
        for (size_t h = ptr_hash(t) & mask;; h = (h + 1) & mask) {
            if (!slots_[h].key) return nullptr;
            if (slots_[h].key == t) return &slots_[h];
        }
    }
    Entry *insert(const Track *t) {
        if (2 * (used_ + 1) > slots_.size()) rehash(std::max<size_t>(256, 4 * (used_ + 1)));
        const size_t mask = slots_.size() - 1;
        size_t h = ptr_hash(t) & mask;
        while (slots_[h].key && slots_[h].key != t) h = (h + 1) & mask;
        if (!slots_[h].key) ++used_;
        slots_[h].key = t;
        return &slots_[h];
    }
    // lists are append-only between resets; dead lists are dropped wholesale once they outweigh the live ones
    void begin_window(size_t live_obs_estimate) {
        if (fid_.size() > 8 * live_obs_estimate + 65536) clear();
    }
    void clear() {
        slots_.clear(), fid_.clear(), z_.clear(), used_ = 0;
    }
    std::vector<size_t> fid_; // frame id per cached observation
    std::vector<double> z_;   // 2 per cached observation

  private:
    void rehash(size_t n) {
        size_t cap = 256;
        while (cap < n) cap <<= 1;
        std::vector<Entry> old;
        old.swap(slots_);
        slots_.assign(cap, Entry());
        used_ = 0;
        for (const Entry &e : old)
            if (e.key) *insert(e.key) = e;
    }
    std::vector<Entry> slots_;
    size_t used_ = 0;
};

void put_q(std::vector<double> &v, size_t off, const quaternion &q) {
    for (int k = 0; k < 4; ++k) v[off + k] = q.coeffs()[k]; // x y z w
}
void put_m3(std::vector<double> &v, size_t off, const matrix<3> &m) { // -> row-major
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) v[off + 3 * r + c] = m(r, c);
}

// The reference's block order: frames in map index order (:75-88); inverse depths by first encounter scanning frames then
// keypoint index (:91-103); reprojection blocks per track in ascending frame id with the anchor first (track.h:69).
// (Ceres sums residual blocks frame-major; the flat CSR is landmark-major -- a different summation order, i.e. a
// difference at rounding level only.)
// `victim` (marginalization only): the landmarks are the tracks the victim frame observes (:453-457), in its keypoint order -- the other
// frames' keypoints are not walked at all.
void flatten(Map *map, Config *config, bool use_inertial, bool for_marginalization, Flat &F, ObsCache *cache = nullptr, int victim = -1) {
    const int N = (int)map->frame_num();
    // the device path takes windows of at most PVIO_MAX_FRAMES frames (the reference's sliding window holds 10): say so before
    // walking a larger map -- and before the pointer table below, which has no room for 256 frames
    if (N > PVIO_MAX_FRAMES) throw std::out_of_range("window of " + std::to_string(N) + " frames: the HIP back-end takes at most " + std::to_string(PVIO_MAX_FRAMES));
    FrameIndex fidx;
    for (int i = 0; i < N; ++i) fidx.put(map->get_frame(i), i);
    F.frame_fixed.assign(N, 0), F.pre_valid.assign(N, 0);
    F.cam.assign(7 * N, 0), F.imu.assign(7 * N, 0), F.sic.assign(4 * N, 0), F.intr.assign(4 * N, 0), F.fstate.assign(16 * N, 0);
    F.pre_delta.assign(11 * N, 0), F.pre_U.assign(225 * N, 0), F.pre_jac.assign(45 * N, 0);
    for (int i = 0; i < N; ++i) {
        Frame *f = map->get_frame(i);
        F.frame_fixed[i] = f->flag(FrameFlag::FF_FIX_POSE) ? 1 : 0;
        put_q(F.cam, 7 * i, f->camera.q_cs), put_q(F.imu, 7 * i, f->imu.q_cs);
        for (int k = 0; k < 3; ++k) F.cam[7 * i + 4 + k] = f->camera.p_cs[k], F.imu[7 * i + 4 + k] = f->imu.p_cs[k];
        F.sic[4 * i] = f->sqrt_inv_cov(0, 0), F.sic[4 * i + 1] = f->sqrt_inv_cov(0, 1), F.sic[4 * i + 2] = f->sqrt_inv_cov(1, 0), F.sic[4 * i + 3] = f->sqrt_inv_cov(1, 1);
        F.intr[4 * i] = f->K(0, 0), F.intr[4 * i + 1] = f->K(1, 1), F.intr[4 * i + 2] = f->K(0, 2), F.intr[4 * i + 3] = f->K(1, 2);
        put_q(F.fstate, 16 * i, f->pose.q);
        for (int k = 0; k < 3; ++k) {
            F.fstate[16 * i + 4 + k] = f->pose.p[k], F.fstate[16 * i + 7 + k] = f->motion.v[k];
            F.fstate[16 * i + 10 + k] = f->motion.bg[k], F.fstate[16 * i + 13 + k] = f->motion.ba[k];
        }
    }
    // planes with >= 20 tracks constrain their best-plane tracks through the plane-distance factor; tracks of smaller
    // planes fall back to plain reprojection blocks (:165-195)
    std::vector<std::pair<Track *, Plane *>> plane_factors;
    if (!for_marginalization)
        for (size_t i = 0; i < map->plane_num(); ++i) {
            Plane *pl = map->get_plane(i);
            if (pl->tracks.size() < 20) continue;
            for (Track *t : pl->tracks)
                if (t->landmark.plane_id == pl->id()) plane_factors.emplace_back(t, pl); // only under its best plane (:183)
        }
    static thread_local PtrSet visited; // keeps its table between solves
    visited.reset(F.lm_track.size());
    F.lm_track.clear(), F.lm_anchor.clear(), F.lm_ptr.assign(1, 0), F.obs_frame.clear(), F.lm_z.clear(), F.obs_z.clear(), F.rho.clear();
    size_t win_id[kMaxWindow];
    for (int i = 0; i < N && i < kMaxWindow; ++i) win_id[i] = map->get_frame(i)->id();
    if (N > kMaxWindow) cache = nullptr;
    if (cache) cache->begin_window(F.obs_frame.capacity());
    auto walk = [&](Track *track, Frame *anchor_frame, ObsCache::Entry *e) { // from scratch; fills the cache entry as well
        if (e) e->off = (uint32_t)cache->fid_.size(), e->cnt = 0;
        for (const auto &kv : track->keypoint_map()) {
            if (kv.first == anchor_frame) continue;              // the anchor observation has no factor (:149)
            const auto &z = kv.first->get_keypoint(kv.second);
            if (e) cache->fid_.push_back(kv.first->id()), cache->z_.push_back(z(0)), cache->z_.push_back(z(1)), ++e->cnt;
            const int t = fidx.find(kv.first);
            if (t < 0) continue;
            F.obs_frame.push_back(t);
            F.obs_z.push_back(z(0)), F.obs_z.push_back(z(1));
        }
    };
    auto add_landmark = [&](Track *track) {
        if (!visited.insert(track)) return;
        auto first = track->first_keypoint();
        const int anchor = fidx.find(first.first);
        if (anchor < 0) return;
        F.lm_track.push_back(track);
        F.lm_anchor.push_back(anchor);
        const auto &za = first.first->get_keypoint(first.second);
        F.lm_z.push_back(za(0)), F.lm_z.push_back(za(1));
        F.rho.push_back(track->landmark.inv_depth);
        if (!cache) {
            walk(track, first.first, nullptr);
        } else {
            const size_t nkp = track->keypoint_num(), first_id = first.first->id();
            const auto last = track->last_keypoint();
            const size_t last_id = last.first->id();
            ObsCache::Entry *e = cache->find(track);
            bool hit = e && e->id == track->id() && e->first_id == first_id;
            if (hit && e->nkp == nkp && e->last_id == last_id) {
                // unchanged
            } else if (hit && e->nkp + 1 == nkp && nkp >= 2 && std::prev(track->keypoint_map().end(), 2)->first->id() == e->last_id) {
                // grown by its newest observation: the list moves to the end of the store with the new entry behind it
                const uint32_t off = (uint32_t)cache->fid_.size();
                for (uint32_t k = 0; k < e->cnt; ++k) {
                    cache->fid_.push_back(cache->fid_[e->off + k]);
                    cache->z_.push_back(cache->z_[2 * (size_t)(e->off + k)]), cache->z_.push_back(cache->z_[2 * (size_t)(e->off + k) + 1]);
                }
                const auto &z = last.first->get_keypoint(last.second);
                cache->fid_.push_back(last_id), cache->z_.push_back(z(0)), cache->z_.push_back(z(1));
                e->off = off, e->cnt += 1, e->nkp = nkp, e->last_id = last_id;
            } else {
                hit = false;
            }
            if (hit) { // replay: cached frame ids and the window's frame ids are both ascending
                int w = 0;
                for (uint32_t k = 0; k < e->cnt; ++k) {
                    const size_t fid = cache->fid_[e->off + k];
                    while (w < N && win_id[w] < fid) ++w;
                    if (w < N && win_id[w] == fid) {
                        F.obs_frame.push_back(w);
                        F.obs_z.push_back(cache->z_[2 * (size_t)(e->off + k)]), F.obs_z.push_back(cache->z_[2 * (size_t)(e->off + k) + 1]);
                    }
                }
            } else {
                e = cache->insert(track);
                e->id = track->id(), e->nkp = nkp, e->first_id = first_id, e->last_id = last_id;
                walk(track, first.first, e);
            }
        }
        F.lm_ptr.push_back((int32_t)F.obs_frame.size());
    };
    for (int i = 0; i < N; ++i) {
        if (victim >= 0 && i != victim) continue;
        Frame *f = map->get_frame(i);
        for (size_t j = 0; j < f->keypoint_num(); ++j) {
            Track *track = f->get_track(j);
            if (!track || !track->flag(TrackFlag::TF_VALID) || track->flag(TrackFlag::TF_PLANE)) continue;
            add_landmark(track);
        }
    }
    // Tracks of planes with fewer than 20 members get their reprojection blocks (again) at :165-179 -- whatever their flags, once per
    // such plane.  A track that already has its blocks from the loop above (VALID, not PLANE), or from another small plane, is listed
    // twice in the ceres::Problem and weighs twice: that is the landmark's multiplicity (include/pvio_hip.h).
    F.lm_mult.clear();
    if (!for_marginalization) {
        bool any_small = false;
        for (size_t i = 0; i < map->plane_num(); ++i) any_small |= map->get_plane(i)->tracks.size() < 20 && !map->get_plane(i)->tracks.empty();
        if (any_small) {
            std::unordered_map<const Track *, size_t> index;
            for (size_t l = 0; l < F.lm_track.size(); ++l) index.emplace(F.lm_track[l], l);
            F.lm_mult.assign(F.lm_track.size(), 1);
            for (size_t i = 0; i < map->plane_num(); ++i) {
                if (map->get_plane(i)->tracks.size() >= 20) continue;
                for (Track *t : map->get_plane(i)->tracks) {
                    auto it = index.find(t);
                    if (it != index.end()) {
                        F.lm_mult[it->second] += 1;
                        continue;
                    }
                    const size_t before = F.lm_track.size();
                    add_landmark(t);
                    if (F.lm_track.size() > before) index.emplace(t, before), F.lm_mult.push_back(1);
                }
            }
            if (std::all_of(F.lm_mult.begin(), F.lm_mult.end(), [](int32_t m) { return m == 1; })) F.lm_mult.clear();
        }
    }
    // plane-distance factors (:180-195)
    F.plane_ptr.assign(1, 0), F.plane_frame.clear(), F.plane_z.clear(), F.plane_n.clear(), F.plane_d.clear();
    for (auto &tp : plane_factors) {
        for (const auto &kv : tp.first->keypoint_map()) {
            const int t = fidx.find(kv.first);
            if (t < 0) continue;
            const auto &z = kv.first->get_keypoint(kv.second);
            F.plane_frame.push_back(t);
            F.plane_z.push_back(z(0)), F.plane_z.push_back(z(1));
        }
        F.plane_ptr.push_back((int32_t)F.plane_frame.size());
        for (int k = 0; k < 3; ++k) F.plane_n.push_back(tp.second->parameter.normal[k]);
        F.plane_d.push_back(tp.second->parameter.distance);
    }
    // IMU factors: re-integrate at frame_i's current biases first (:224); marginalization reuses the stored delta (:416-450)
    if (use_inertial || for_marginalization)
        for (int j = 1; j < N; ++j) {
            Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
            const bool ok = for_marginalization ? fj->get_preintegration_factor() != nullptr
                                                : fj->preintegration.integrate(fj->image->t, fi->motion.bg, fi->motion.ba, true, true);
            if (!ok) continue;
            F.pre_valid[j] = 1;
            const PreIntegrator::Delta &d = fj->preintegration.delta;
            F.pre_delta[11 * j] = d.t;
            put_q(F.pre_delta, 11 * j + 1, d.q);
            for (int k = 0; k < 3; ++k) F.pre_delta[11 * j + 5 + k] = d.p[k], F.pre_delta[11 * j + 8 + k] = d.v[k];
            for (int r = 0; r < 15; ++r)
                for (int c = 0; c < 15; ++c) F.pre_U[225 * j + 15 * r + c] = d.sqrt_inv_cov(r, c);
            const PreIntegrator::Jacobian &jc = fj->preintegration.jacobian;
            put_m3(F.pre_jac, 45 * j, jc.dq_dbg), put_m3(F.pre_jac, 45 * j + 9, jc.dp_dbg), put_m3(F.pre_jac, 45 * j + 18, jc.dp_dba);
            put_m3(F.pre_jac, 45 * j + 27, jc.dv_dbg), put_m3(F.pre_jac, 45 * j + 36, jc.dv_dba);
        }
    // marginalization prior (:126-139): S, s, the related frames and the linearization states the holder captured
    F.prior_frames.clear(), F.prior_S.clear(), F.prior_s.clear(), F.prior_lin.clear();
    if (Factor *mf = map->get_marginalization_factor()) {
        const MarginalizationErrorCost *mc = mf->get_cost_function<MarginalizationErrorCost>();
        const std::vector<Frame *> &rel = mc->related_frames();
        for (size_t i = 0; i < rel.size(); ++i) {
            const int pf = fidx.find(rel[i]);
            if (pf < 0) throw std::out_of_range("marginalization prior refers to a frame outside the window"); // unordered_map::at threw here too; caught by the callers
            F.prior_frames.push_back(pf);
            const size_t o = F.prior_lin.size();
            F.prior_lin.resize(o + 16);
            const PoseState &p0 = mc->linearization_pose(i);
            const MotionState &m0 = mc->linearization_motion(i);
            put_q(F.prior_lin, o, p0.q);
            for (int k = 0; k < 3; ++k) F.prior_lin[o + 4 + k] = p0.p[k], F.prior_lin[o + 7 + k] = m0.v[k], F.prior_lin[o + 10 + k] = m0.bg[k], F.prior_lin[o + 13 + k] = m0.ba[k];
        }
        const matrix<> &S = mc->sqrt_information();
        const vector<> &sv = mc->information_vector();
        const size_t D = 15 * rel.size();
        F.prior_S.resize(D * D), F.prior_s.resize(D);
        for (size_t r = 0; r < D; ++r) {
            F.prior_s[r] = sv[r];
            for (size_t c = 0; c < D; ++c) F.prior_S[r * D + c] = S(r, c);
        }
    }
    pvio_ba_problem &pb = F.pb;
    std::memset(&pb, 0, sizeof pb);
    pb.n_frames = N, pb.n_landmarks = (int32_t)F.lm_anchor.size(), pb.n_obs = (int32_t)F.obs_frame.size();
    pb.use_inertial = use_inertial ? 1 : 0;
    pb.frame_fixed = F.frame_fixed.data(), pb.cam_extrinsic = F.cam.data(), pb.imu_extrinsic = F.imu.data();
    pb.sqrt_inv_cov = F.sic.data(), pb.intrinsics = F.intr.data();
    pb.lm_anchor_frame = F.lm_anchor.data(), pb.lm_anchor_z = F.lm_z.data(), pb.lm_obs_ptr = F.lm_ptr.data();
    pb.obs_frame = F.obs_frame.data(), pb.obs_z = F.obs_z.data();
    pb.preint_valid = F.pre_valid.data(), pb.preint_delta = F.pre_delta.data(), pb.preint_sqrt_inv_cov = F.pre_U.data(), pb.preint_jacobian = F.pre_jac.data();
    pb.prior_n = (int32_t)F.prior_frames.size(), pb.prior_frames = F.prior_frames.data();
    pb.prior_S = F.prior_S.data(), pb.prior_s = F.prior_s.data(), pb.prior_lin_state = F.prior_lin.data();
    pb.lm_multiplicity = F.lm_mult.empty() ? nullptr : F.lm_mult.data();
    pb.n_plane_factors = (int32_t)F.plane_d.size();
    pb.plane_obs_ptr = F.plane_ptr.data(), pb.plane_obs_frame = F.plane_frame.data(), pb.plane_obs_z = F.plane_z.data();
    pb.plane_normal = F.plane_n.data(), pb.plane_distance = F.plane_d.data();
    pb.plane_sqrt_inv_cov = std::sqrt(1.0 / config->plane_distance_cov()); // :181
    pb.max_iterations = (int32_t)config->solver_iteration_limit();
    pb.max_solver_time = config->solver_time_limit();
}

// ---- post-solve passes (bundle_adjustor.cpp:251-296), host side ---------------------------------------------------
// depth gate + mean pixel error of one track from the map's current states (:279-295); false when the gate fails
bool track_quality(const Track *track, double &quality) {
    const vector<3> x = track->get_landmark_point();
    double sum = 0, num = 0;
    for (const auto &fk : track->keypoint_map()) {
        const Frame *frame = fk.first;
        const PoseState cam = frame->get_pose(frame->camera);
        const vector<3> y = cam.q.conjugate() * (x - cam.p);
        if (y.z() <= 1.0e-3 || y.z() > 50) return false;
        const vector<2> &z = frame->get_keypoint(fk.second);
        const double du = (y.x() / y.z() * frame->K(0, 0) + frame->K(0, 2)) - (z[0] * frame->K(0, 0) + frame->K(0, 2)); // apply_k, stereo.h:25-27
        const double dv = (y.y() / y.z() * frame->K(1, 1) + frame->K(1, 2)) - (z[1] * frame->K(1, 1) + frame->K(1, 2));
        sum += std::sqrt(du * du + dv * dv), num += 1.0;
    }
    quality = sum / std::max(num, 1.0);
    return true;
}

// :251-275 -- PLANE tracks that can be triangulated on their own are checked against the planes they sit in (0.1 m): a
// plane that does not hold the point loses the track, a track that no plane holds goes back to being an ordinary VALID
// landmark at the triangulated point.  Returns the tracks whose inverse depth it replaced.
std::vector<Track *> revalidate_plane_tracks(Map *map) {
    std::vector<Track *> moved;
    for (size_t i = 0; i < map->track_num(); ++i) {
        Track *track = map->get_track(i);
        if (!track->flag(TrackFlag::TF_PLANE)) continue;
        if (track->keypoint_num() < 2) continue;
        vector<3> p;
        if (!(track->life > 10 && PlaneExtractor::enough_baseline(track) && track->try_triangulate(p))) continue;
        bool held = false;
        for (size_t j = 0; j < map->plane_num(); ++j) {
            Plane *plane = map->get_plane(j);
            if (plane->tracks.count(track) == 0) continue;
            if (std::abs(plane->parameter.normal.dot(p) - plane->parameter.distance) > 0.1) plane->tracks.erase(track);
            else held = true;
        }
        if (!held) {
            track->flag(TrackFlag::TF_PLANE) = false;
            track->flag(TrackFlag::TF_VALID) = true;
            track->set_landmark_point(p);
            moved.push_back(track);
        }
    }
    return moved;
}

struct DefaultConfig : Config { // solver_iteration_limit / solver_time_limit / plane_distance_cov defaults (config.cpp:24-26,82-88)
    matrix<3> camera_intrinsic() const override { return matrix<3>(); }
    quaternion camera_to_body_rotation() const override { return quaternion(); }
    vector<3> camera_to_body_translation() const override { return vector<3>(); }
    quaternion imu_to_body_rotation() const override { return quaternion(); }
    vector<3> imu_to_body_translation() const override { return vector<3>(); }
    matrix<2> keypoint_noise_cov() const override { return matrix<2>(); }
    matrix<3> gyroscope_noise_cov() const override { return matrix<3>(); }
    matrix<3> accelerometer_noise_cov() const override { return matrix<3>(); }
    matrix<3> gyroscope_bias_noise_cov() const override { return matrix<3>(); }
    matrix<3> accelerometer_bias_noise_cov() const override { return matrix<3>(); }
};

} // namespace

// diagnostics (tests/host/roundtrip.cpp): seconds per flattening of `map`, steady state (the arrays keep their capacity)
// reps > 0: walk from scratch every time; reps < 0: |reps| repetitions with the observation cache (steady state of an unchanged window)
double flatten_seconds(Map *map, bool use_inertial, int reps) {
    DefaultConfig dc;
    Flat F;
    ObsCache cache;
    ObsCache *c = reps < 0 ? &cache : nullptr;
    if (reps < 0) reps = -reps;
    flatten(map, &dc, use_inertial, false, F, c);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) flatten(map, &dc, use_inertial, false, F, c);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (reps > 0 ? reps : 1);
}

struct BundleAdjustor::BundleAdjustorSolver { // the pimpl of bundle_adjustor.h:30,41; the GPU context is process-wide, see process_ctx()
    Flat F; // a BundleAdjustor temporary lives for one call: the arrays that should keep their capacity are thread_local below

    static Flat &window() {
        static thread_local Flat F; // keeps the capacity of its arrays from one keyframe to the next
        return F;
    }
    static ObsCache &obs_cache() {
        static thread_local ObsCache c; // the flattened observation lists of the tracks, kept from one keyframe to the next
        return c;
    }

    bool solve(Map *map, Config *config, bool use_inertial) {
        pvio_hip_ctx *ctx = process_ctx();
        if (!ctx) return false;
        DefaultConfig dc;
        Flat &F = window();
        static const bool timing = std::getenv("PVIO_HIP_TIMING") != nullptr; // diagnostics: host share of a keyframe solve
        const bool verify = std::getenv("PVIO_HIP_FLATTEN_VERIFY") != nullptr; // tests: the cached flattening against a walk from scratch
        static const bool no_cache = std::getenv("PVIO_HIP_FLATTEN_NOCACHE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        flatten(map, config ? config : &dc, use_inertial, false, F, no_cache ? nullptr : &obs_cache());
        const auto t1 = std::chrono::steady_clock::now();
        if (verify) {
            Flat G;
            flatten(map, config ? config : &dc, use_inertial, false, G, nullptr);
            if (G.lm_track != F.lm_track || G.lm_anchor != F.lm_anchor || G.lm_ptr != F.lm_ptr || G.obs_frame != F.obs_frame || G.obs_z != F.obs_z || G.lm_z != F.lm_z || G.rho != F.rho)
                throw std::logic_error("incremental flattening differs from the walk from scratch");
        }
        F.quality.assign(F.lm_track.size(), 0.0), F.valid.assign(F.lm_track.size(), 1);
        pvio_ba_state st{F.fstate.data(), F.rho.data(), F.quality.data(), F.valid.data()};
        pvio_ba_summary sum;
        std::memset(&sum, 0, sizeof sum);
        if (pvio_hip_ba_solve(ctx, &F.pb, &st, &sum) != PVIO_OK) {
            std::fprintf(stderr, "[pvio-hip] ba_solve failed: %s\n", pvio_hip_last_error(ctx));
            return false;
        }
        const auto t2 = std::chrono::steady_clock::now();
        // write the states back in place, like Ceres does through the parameter-block pointers
        const bool motion_blocks = use_inertial || map->get_marginalization_factor() != nullptr;
        for (size_t i = 0; i < map->frame_num(); ++i) {
            Frame *f = map->get_frame(i);
            for (int k = 0; k < 4; ++k) f->pose.q.coeffs()[k] = F.fstate[16 * i + k];
            for (int k = 0; k < 3; ++k) {
                f->pose.p[k] = F.fstate[16 * i + 4 + k];
                if (motion_blocks) f->motion.v[k] = F.fstate[16 * i + 7 + k], f->motion.bg[k] = F.fstate[16 * i + 10 + k], f->motion.ba[k] = F.fstate[16 * i + 13 + k];
            }
        }
        for (size_t l = 0; l < F.lm_track.size(); ++l) F.lm_track[l]->landmark.inv_depth = F.rho[l];
        // ---- post-solve passes (:251-296).  The depth gate and the mean pixel error of the flattened landmarks come from the
        // device (k_quality, evaluated on the final states); the plane-track re-validation and the tracks the device never
        // saw (PLANE tracks of planes with >= 20 tracks, tracks the re-validation has just moved) are done here. ----
        const std::vector<Track *> moved = revalidate_plane_tracks(map);
        static thread_local PtrSet on_device;
        on_device.reset(F.lm_track.size());
        for (size_t l = 0; l < F.lm_track.size(); ++l) {
            Track *t = F.lm_track[l];
            if (std::find(moved.begin(), moved.end(), t) != moved.end()) continue; // its point changed after the solve
            if (!t->flag(TrackFlag::TF_VALID) && !t->flag(TrackFlag::TF_PLANE)) continue; // (:279)
            on_device.insert(t);
            if (!F.valid[l]) t->flag(TrackFlag::TF_VALID) = false, t->flag(TrackFlag::TF_PLANE) = false; // depth gate (:286-290)
            else if (t->flag(TrackFlag::TF_VALID)) t->landmark.quality = F.quality[l];                  // (:294-295)
        }
        for (size_t i = 0; i < map->track_num(); ++i) {
            Track *t = map->get_track(i);
            if (!t->flag(TrackFlag::TF_VALID) && !t->flag(TrackFlag::TF_PLANE)) continue;
            if (on_device.contains(t)) continue;
            double q = 0;
            if (!track_quality(t, q)) t->flag(TrackFlag::TF_VALID) = false, t->flag(TrackFlag::TF_PLANE) = false;
            else if (t->flag(TrackFlag::TF_VALID)) t->landmark.quality = q;
        }
        if (timing) {
            const auto t3 = std::chrono::steady_clock::now();
            auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            std::fprintf(stderr, "[pvio-hip] solve: %d frames, %d landmarks, %d factors: flatten %.1f us, upload+solve+download %.1f us (device %.1f us, %d iterations), write-back + post passes %.1f us\n",
                         F.pb.n_frames, F.pb.n_landmarks, F.pb.n_obs, us(t0, t1), us(t1, t2), 1e6 * sum.device_seconds, sum.num_iterations, us(t2, t3));
        }
        return sum.is_usable != 0;
    }

    void marginalize_frame(Map *map, size_t index) {
        // Whatever goes wrong below, Map::marginalize_frame erases the victim right after this call (map.cpp:78-87): a prior that
        // still names it would hold a dangling Frame* and make every later flatten fail or match a recycled address.
        // So the map never keeps the old prior past a failed marginalization: it is dropped (the window loses its gauge prior
        // until the next successful one -- the solve stays well-posed through the IMU factors, and says so on stderr).
        struct DropStalePrior {
            Map *map;
            bool armed = true;
            ~DropStalePrior() {
                if (armed && map->get_marginalization_factor()) {
                    std::fprintf(stderr, "[pvio-hip] marginalize_frame failed: the old prior is dropped (it names the frame that is being erased)\n");
                    map->set_marginalization_factor(nullptr);
                }
            }
        } guard{map};
        pvio_hip_ctx *ctx = process_ctx();
        if (!ctx || index >= map->frame_num() || map->frame_num() < 2) return;
        DefaultConfig dc;
        static const bool timing = std::getenv("PVIO_HIP_TIMING") != nullptr; // diagnostics: host share of a marginalization
        const auto t0 = std::chrono::steady_clock::now();
        flatten(map, &dc, true, true, F, &obs_cache(), (int)index); // the tracks were flattened by the last solve: their lists are replayed
        const auto t1 = std::chrono::steady_clock::now();
        const size_t n = map->frame_num() - 1, D = 15 * n;
        std::vector<double> S(D * D, 0.0), s(D, 0.0); // row-major, like the C ABI
        pvio_ba_state st{F.fstate.data(), F.rho.data(), nullptr, nullptr};
        pvio_ba_prior out;
        std::memset(&out, 0, sizeof out);
        out.S = S.data(), out.s = s.data();
        if (pvio_hip_ba_marginalize(ctx, &F.pb, &st, (int32_t)index, &out) != PVIO_OK) {
            std::fprintf(stderr, "[pvio-hip] ba_marginalize failed: %s\n", pvio_hip_last_error(ctx));
            return;
        }
        const auto t2 = std::chrono::steady_clock::now();
        matrix<> sqrt_infomat;
        vector<> sqrt_infovec;
        sqrt_infomat.resize((int)D, (int)D), sqrt_infovec.resize((int)D);
        for (size_t r = 0; r < D; ++r) {
            sqrt_infovec[(int)r] = s[r];
            for (size_t c = 0; c < D; ++c) sqrt_infomat((int)r, (int)c) = S[r * D + c];
        }
        std::vector<Frame *> remaining; // linearized at their current states by the holder's constructor (:592-598)
        for (size_t i = 0; i < map->frame_num(); ++i)
            if (i != index) remaining.emplace_back(map->get_frame(i));
        map->set_marginalization_factor(Factor::create_marginalization_error(sqrt_infomat, sqrt_infovec, std::move(remaining)));
        guard.armed = false;
        if (timing) {
            auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            std::fprintf(stderr, "[pvio-hip] marginalize_frame: %d frames, %d landmarks: flatten %.1f us, pvio_hip_ba_marginalize %.1f us, new prior factor %.1f us\n",
                         F.pb.n_frames, F.pb.n_landmarks, us(t0, t1), us(t1, t2), us(t2, std::chrono::steady_clock::now()));
        }
    }

    double compute_reprojection_error(Map *map) {
        pvio_hip_ctx *ctx = process_ctx();
        if (!ctx) return 0.0;
        DefaultConfig dc;
        flatten(map, &dc, false, true, F);
        pvio_ba_state st{F.fstate.data(), F.rho.data(), nullptr, nullptr};
        double err = 0;
        if (pvio_hip_ba_reprojection_error(ctx, &F.pb, &st, &err) != PVIO_OK) return 0.0;
        return err;
    }
};

BundleAdjustor::BundleAdjustor() { solver = std::make_unique<BundleAdjustorSolver>(); }
BundleAdjustor::~BundleAdjustor() = default;

// The reference's methods do not throw (Ceres reports failure through the summary); neither do these: anything the
// flattening or the standard library raises (bad_alloc, a prior that names a frame outside the window) ends the call as
// "not usable" / "no new prior" with one line on stderr.
// The reference's forensics timers at the two seams (bundle_adjustor.cpp:308-319 running average of the solve time, :349-353
// marginalization time: pvio-pc's "BA Time" graphs, main.cpp:165-167).  Inside the PVIO tree they are the reference's own
// `forensics` / `make_timer` (forensics.h:96-101, utility/unique_timer.h) and obey PVIO_ENABLE_FORENSICS like there; the standalone
// test build has neither header and no GUI to feed.
#ifdef PVIO_HOST_USE_REFERENCE_TYPES
#define PVIO_HOST_SOLVE_TIMER()                                              \
    auto ba_timer = make_timer([](double t) {                                \
        forensics(bundle_adjustor_solve_time, time) {                        \
            static double avg_time = 0;                                      \
            static double avg_count = 0;                                     \
            avg_time = (avg_time * avg_count + t) / (avg_count + 1);         \
            avg_count += 1.0;                                                \
            time = avg_time;                                                 \
        }                                                                    \
    })
#define PVIO_HOST_MARG_TIMER()                                               \
    auto ba_timer = make_timer([](double t) {                                \
        forensics(bundle_adjustor_marginalization_time, time) { time = t; }  \
    })
#else
#define PVIO_HOST_SOLVE_TIMER() ((void)0)
#define PVIO_HOST_MARG_TIMER() ((void)0)
#endif

bool BundleAdjustor::solve(Map *map, Config *config, bool use_inertial) {
    PVIO_HOST_SOLVE_TIMER();
    try {
        return solver->solve(map, config, use_inertial);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[pvio-hip] BundleAdjustor::solve: %s\n", e.what());
        return false;
    }
}

void BundleAdjustor::marginalize_frame(Map *map, size_t index) {
    PVIO_HOST_MARG_TIMER();
    try {
        solver->marginalize_frame(map, index);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[pvio-hip] BundleAdjustor::marginalize_frame: %s\n", e.what());
    }
}

double BundleAdjustor::compute_reprojection_error(Map *map) {
    try {
        return solver->compute_reprojection_error(map);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[pvio-hip] BundleAdjustor::compute_reprojection_error: %s\n", e.what());
        return 0.0;
    }
}

} // namespace pvio
