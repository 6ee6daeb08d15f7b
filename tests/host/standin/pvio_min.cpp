// pvio_min.cpp -- standalone stand-in for the part of the reference's map layer that pvio_min.h declares.
//
// NOT part of the drop-in: inside the PVIO tree these functions are the reference's own (map/{map,frame,track,plane}.cpp,
// estimation/{factor,preintegrator}.cpp, core/plane_extractor.cpp:184-203, geometry/stereo.h:67-128) and stay untouched.
// Outside the tree (tests, bench, tools/pvio_headless) the adapter still needs a Map to work on, so the few operations
// it calls are restated here, each citing what it follows.  Scalar bookkeeping, no hot path.
#include <cmath>
#include <cstring>
#include <iterator>
#include <limits>

#include "../../../include/pvio_hip.h"
#include "dropin/pvio/estimation/ceres/marginalization_error_cost.h"
#include "dropin/pvio/estimation/ceres/preintegration_error_cost.h"
#include "dropin/pvio/estimation/ceres/reprojection_error_cost.h"
#include "feature_front.h"
#include "pvio_min.h"

namespace pvio {

// ---- estimation/preintegrator.cpp:84-108 -------------------------------------------------------------------------
bool PreIntegrator::integrate(double t, const vector<3> &bg, const vector<3> &ba, bool, bool) {
    if (data.empty()) return false;
    std::vector<double> ts(data.size()), w(3 * data.size()), a(3 * data.size());
    for (size_t i = 0; i < data.size(); ++i) {
        ts[i] = data[i].t;
        for (int k = 0; k < 3; ++k) w[3 * i + k] = data[i].w[k], a[3 * i + k] = data[i].a[k];
    }
    pvio_imu_noise nz;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) nz.cov_w[3 * r + c] = cov_w(r, c), nz.cov_a[3 * r + c] = cov_a(r, c), nz.cov_bg[3 * r + c] = cov_bg(r, c), nz.cov_ba[3 * r + c] = cov_ba(r, c);
    double d[11], cov[225], U[225], jac[45]; // the C ABI is row-major
    if (pvio_preintegrate((int32_t)data.size(), ts.data(), w.data(), a.data(), t, bg.data(), ba.data(), &nz, d, cov, U, jac) != PVIO_OK) return false;
    delta.t = d[0];
    for (int k = 0; k < 4; ++k) delta.q.coeffs()[k] = d[1 + k];
    for (int k = 0; k < 3; ++k) delta.p[k] = d[5 + k], delta.v[k] = d[8 + k];
    for (int r = 0; r < 15; ++r)
        for (int c = 0; c < 15; ++c) delta.cov(r, c) = cov[15 * r + c], delta.sqrt_inv_cov(r, c) = U[15 * r + c];
    matrix<3> *J[5] = {&jacobian.dq_dbg, &jacobian.dp_dbg, &jacobian.dp_dba, &jacobian.dv_dbg, &jacobian.dv_dba};
    for (int b = 0; b < 5; ++b)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) (*J[b])(r, c) = jac[9 * b + 3 * r + c];
    return true;
}

void PreIntegrator::predict(const Frame *old_frame, Frame *new_frame) {
    const vector<3> gravity(0, 0, -9.80665);
    new_frame->motion.bg = old_frame->motion.bg;
    new_frame->motion.ba = old_frame->motion.ba;
    new_frame->motion.v = old_frame->motion.v + gravity * delta.t + old_frame->pose.q * delta.v;
    new_frame->pose.p = old_frame->pose.p + 0.5 * gravity * delta.t * delta.t + old_frame->motion.v * delta.t + old_frame->pose.q * delta.p;
    new_frame->pose.q = old_frame->pose.q * delta.q;
}

// ---- estimation/factor.cpp:27-43 ---------------------------------------------------------------------------------
std::unique_ptr<Factor> Factor::create_marginalization_error(const matrix<> &sqrt_inv_cov, const vector<> &infovec, std::vector<Frame *> &&frames) {
    return std::make_unique<Factor>(std::make_unique<MarginalizationErrorCost>(sqrt_inv_cov, infovec, std::move(frames)), factor_construct_t());
}
std::unique_ptr<Factor> Factor::create_reprojection_error(Track *track, Frame *frame, size_t keypoint_index) {
    return std::make_unique<Factor>(std::make_unique<ReprojectionErrorCost>(track, frame, keypoint_index), factor_construct_t());
}
std::unique_ptr<Factor> Factor::create_preintegration_error(Frame *frame_i, Frame *frame_j) {
    return std::make_unique<Factor>(std::make_unique<PreIntegrationErrorCost>(frame_i, frame_j), factor_construct_t());
}

// ---- map/frame.cpp:27-139,187-200 --------------------------------------------------------------------------------
create_if_empty_t create_if_empty{};

std::unique_ptr<Frame> Frame::clone() const {
    std::unique_ptr<Frame> f = std::make_unique<Frame>(id(), construct_by_frame_t());
    f->K = K, f->sqrt_inv_cov = sqrt_inv_cov, f->image = image, f->pose = pose, f->motion = motion, f->camera = camera, f->imu = imu;
    f->preintegration = preintegration;
    f->keypoints = keypoints;
    f->tracks.assign(keypoints.size(), nullptr);
    f->reprojection_factors.resize(keypoints.size());
    return f;
}

Track *Frame::get_track(size_t keypoint_index, const create_if_empty_t &) {
    if (tracks[keypoint_index] == nullptr) map->create_track()->add_keypoint(this, keypoint_index);
    return tracks[keypoint_index];
}

// pixels in, pixels out through the Image seam; new corners appended behind the existing keypoints (frame.cpp:72-87)
void Frame::detect_keypoints(Config *config) {
    std::vector<vector<2>> px(keypoints.size());
    for (size_t i = 0; i < keypoints.size(); ++i) px[i] = vector<2>(keypoints[i][0] * K(0, 0) + K(0, 2), keypoints[i][1] * K(1, 1) + K(1, 2));
    image->detect_keypoints(px, config->feature_tracker_max_keypoint_detection(), config->feature_tracker_min_keypoint_distance());
    const size_t old = keypoints.size();
    keypoints.resize(px.size()), tracks.resize(px.size(), nullptr), reprojection_factors.resize(px.size());
    for (size_t i = old; i < px.size(); ++i) keypoints[i] = vector<2>((px[i][0] - K(0, 2)) / K(0, 0), (px[i][1] - K(1, 2)) / K(1, 1));
}

// gyro-only prediction -> Image::track_keypoints -> longest-track-first Poisson-disk acceptance -> survivors appended to the
// next frame and to their tracks (frame.cpp:89-139); the first three steps are feature_front.h
void Frame::track_keypoints(Frame *next_frame, Config *config) {
    std::vector<vector<2>> curr(keypoints.size()), next;
    for (size_t i = 0; i < keypoints.size(); ++i) curr[i] = vector<2>(keypoints[i][0] * K(0, 0) + K(0, 2), keypoints[i][1] * K(1, 1) + K(1, 2));
    if (config->feature_tracker_predict_keypoints()) predict_keypoints(*this, *next_frame, next);
    std::vector<char> status;
    image->track_keypoints(next_frame->image.get(), curr, next, status);
    std::vector<size_t> length(curr.size(), 0);
    for (size_t i = 0; i < curr.size(); ++i)
        if (status[i] && tracks[i]) length[i] = tracks[i]->keypoint_num();
    select_tracked(next, length, config->feature_tracker_min_keypoint_distance(), status);
    for (size_t i = 0; i < curr.size(); ++i)
        if (status[i]) {
            const size_t k = next_frame->keypoint_num();
            next_frame->append_keypoint(vector<2>((next[i][0] - next_frame->K(0, 2)) / next_frame->K(0, 0), (next[i][1] - next_frame->K(1, 2)) / next_frame->K(1, 1)));
            get_track(i, create_if_empty)->add_keypoint(next_frame, k);
        }
}

void Frame::append_keypoint(const vector<2> &keypoint) {
    keypoints.emplace_back(keypoint), tracks.emplace_back(nullptr), reprojection_factors.emplace_back(nullptr);
}
PoseState Frame::get_pose(const ExtrinsicParams &sensor) const {
    PoseState r;
    r.q = pose.q * sensor.q_cs;
    r.p = pose.p + pose.q * sensor.p_cs;
    return r;
}
void Frame::set_pose(const ExtrinsicParams &sensor, const PoseState &sp) {
    pose.q = sp.q * sensor.q_cs.conjugate();
    pose.p = sp.p - pose.q * sensor.p_cs;
}

// ---- map/map.cpp:28-186 ------------------------------------------------------------------------------------------
Map::Map() = default;
Map::~Map() = default;

void Map::put_frame(std::unique_ptr<Frame> frame, size_t position) {
    frame->map = this;
    if (position == nil()) frames.emplace_back(std::move(frame)), position = frames.size() - 1;
    else frames.emplace(frames.begin() + (std::ptrdiff_t)position, std::move(frame));
    if (position > 0) frames[position]->preintegration_factor = Factor::create_preintegration_error(frames[position - 1].get(), frames[position].get());
    if (position + 1 < frames.size()) frames[position + 1]->preintegration_factor = Factor::create_preintegration_error(frames[position].get(), frames[position + 1].get());
}

void Map::erase_frame(size_t index) {
    Frame *frame = frames[index].get();
    for (size_t i = 0; i < frame->keypoint_num(); ++i)
        if (Track *track = frame->get_track(i)) track->remove_keypoint(frame);
    frames.erase(frames.begin() + (std::ptrdiff_t)index);
    if (index > 0 && index < frames.size()) frames[index]->preintegration_factor = Factor::create_preintegration_error(frames[index - 1].get(), frames[index].get());
}

void Map::marginalize_frame(size_t index) {
    BundleAdjustor().marginalize_frame(this, index);
    Frame *frame = frames[index].get();
    for (size_t i = 0; i < frame->keypoint_num(); ++i)
        if (Track *track = frame->get_track(i)) track->remove_keypoint(frame);
    frames.erase(frames.begin() + (std::ptrdiff_t)index);
    if (index > 0 && index < frames.size()) frames[index]->preintegration_factor.reset();
}

size_t Map::frame_index_by_id(size_t id) const { // frames are kept in ascending id order (map.cpp:88-106)
    size_t lo = 0, hi = frames.size();
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (frames[mid]->id() < id) lo = mid + 1;
        else hi = mid;
    }
    return (lo < frames.size() && frames[lo]->id() == id) ? lo : nil();
}

Track *Map::create_track() {
    std::unique_ptr<Track> track = std::make_unique<Track>(construct_by_map_t());
    track->map_index = tracks.size(), track->map = this;
    tracks.emplace_back(std::move(track));
    return tracks.back().get();
}

void Map::erase_track(Track *track) {
    while (track->keypoint_num() > 0) track->remove_keypoint(track->keypoint_map().begin()->first, false);
    recycle_track(track);
}

void Map::prune_tracks(const std::function<bool(const Track *)> &condition) {
    std::vector<Track *> doomed;
    for (size_t i = 0; i < track_num(); ++i)
        if (condition(get_track(i))) doomed.push_back(get_track(i));
    for (Track *t : doomed) erase_track(t);
}

void Map::put_plane(std::unique_ptr<Plane> plane) { planes.emplace_back(std::move(plane)); }
void Map::set_marginalization_factor(std::unique_ptr<Factor> factor) { marginalization_factor = std::move(factor); }

void Map::recycle_track(Track *track) { // swap with the last slot, drop from every plane
    if (track->map_index != tracks.back()->map_index) {
        tracks[track->map_index].swap(tracks.back());
        tracks[track->map_index]->map_index = track->map_index;
    }
    for (auto &pl : planes) pl->tracks.erase(track);
    tracks.pop_back();
}

// ---- geometry/stereo.h:76-128: DLT triangulation, null vector of the 2K x 4 system ---------------------------------
// (the reference takes V.col(3) of Eigen's JacobiSVD; here a one-sided Jacobi SVD: rotate column pairs of A until they
// are orthogonal, the column with the smallest norm pairs with the sought right singular vector.  Both are backward
// stable, so the direction agrees to a few ulp times the conditioning; its sign is irrelevant to every use below.)
namespace {
void null_vector4(std::vector<double> &A /* rows x 4, row-major, destroyed */, size_t rows, double v[4]) {
    double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double a = 0, b = 0, c = 0;
                for (size_t r = 0; r < rows; ++r) a += A[4 * r + p] * A[4 * r + p], b += A[4 * r + q] * A[4 * r + q], c += A[4 * r + p] * A[4 * r + q];
                if (c == 0.0) continue;
                off = std::max(off, std::fabs(c) / std::sqrt(a * b + std::numeric_limits<double>::min()));
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (size_t r = 0; r < rows; ++r) {
                    const double x = A[4 * r + p], y = A[4 * r + q];
                    A[4 * r + p] = cs * x - sn * y, A[4 * r + q] = sn * x + cs * y;
                }
                for (int r = 0; r < 4; ++r) {
                    const double x = V[4 * r + p], y = V[4 * r + q];
                    V[4 * r + p] = cs * x - sn * y, V[4 * r + q] = sn * x + cs * y;
                }
            }
        if (off < 1e-15) break;
    }
    int best = 0;
    double bn = std::numeric_limits<double>::max();
    for (int c = 0; c < 4; ++c) {
        double n = 0;
        for (size_t r = 0; r < rows; ++r) n += A[4 * r + c] * A[4 * r + c];
        if (n < bn) bn = n, best = c;
    }
    for (int r = 0; r < 4; ++r) v[r] = V[4 * r + best];
}

// triangulate_point_scored over all observations (stereo.h:104-128): true when the point lies in front of every camera
// and closer than 100 (in units of the homogeneous scale)
bool triangulate_track(const Track *track, vector<3> &p) {
    const auto &obs = track->keypoint_map();
    std::vector<double> A(8 * obs.size()), P(12 * obs.size());
    size_t k = 0;
    for (const auto &fk : obs) {
        const PoseState cam = fk.first->get_pose(fk.first->camera);
        const matrix<3> R = cam.q.conjugate().toRotationMatrix(); // world -> camera
        const vector<3> T = -(R * cam.p);
        const vector<2> &z = fk.first->get_keypoint(fk.second);
        double *Pk = &P[12 * k];
        for (int r = 0; r < 3; ++r) Pk[4 * r] = R(r, 0), Pk[4 * r + 1] = R(r, 1), Pk[4 * r + 2] = R(r, 2), Pk[4 * r + 3] = T[r];
        for (int c = 0; c < 4; ++c) A[8 * k + c] = z[0] * Pk[8 + c] - Pk[c], A[8 * k + 4 + c] = z[1] * Pk[8 + c] - Pk[4 + c];
        ++k;
    }
    double q[4];
    null_vector4(A, 2 * obs.size(), q);
    bool in_front = true;
    for (size_t i = 0; i < obs.size(); ++i) {
        const double *Pk = &P[12 * i];
        const double z = Pk[8] * q[0] + Pk[9] * q[1] + Pk[10] * q[2] + Pk[11] * q[3];
        if (!(z * q[3] > 0)) in_front = false;
        if (!(z / q[3] < 100)) in_front = false;
    }
    if (in_front) p = vector<3>(q[0] / q[3], q[1] / q[3], q[2] / q[3]);
    else p = vector<3>(q[0], q[1], q[2]).normalized();
    return in_front;
}
} // namespace

// ---- map/track.cpp:28-147 ----------------------------------------------------------------------------------------
const vector<2> &Track::get_keypoint(Frame *frame) const { return frame->get_keypoint(keypoint_refs.at(frame)); }

void Track::add_keypoint(Frame *frame, size_t keypoint_index) {
    frame->tracks[keypoint_index] = this;
    frame->reprojection_factors[keypoint_index] = Factor::create_reprojection_error(this, frame, keypoint_index);
    keypoint_refs[frame] = keypoint_index;
    life++;
}

void Track::remove_keypoint(Frame *frame, bool suicide_if_empty) {
    const size_t keypoint_index = keypoint_refs.at(frame);
    if (keypoint_refs.size() > 1) {
        if (frame == first_frame()) { // the anchor leaves: re-express the inverse depth in the next observing frame (:42-49)
            const auto next = *std::next(keypoint_refs.begin());
            const PoseState cam = frame->get_pose(frame->camera), ncam = next.first->get_pose(next.first->camera);
            const vector<3> point = (cam.q * frame->get_keypoint(keypoint_index).homogeneous()) / landmark.inv_depth + cam.p;
            landmark.inv_depth = 1.0 / (ncam.q.conjugate() * (point - ncam.p)).z();
        }
    } else {
        flag(TrackFlag::TF_VALID) = false;
    }
    frame->tracks[keypoint_index] = nullptr;
    frame->reprojection_factors[keypoint_index].reset();
    keypoint_refs.erase(frame);
    if (suicide_if_empty && keypoint_refs.empty()) map->recycle_track(this);
}

bool Track::try_triangulate(vector<3> &p) { return triangulate_track(this, p); }

bool Track::triangulate() {
    vector<3> p;
    if (triangulate_track(this, p)) set_landmark_point(p), flag(TrackFlag::TF_VALID) = true;
    else flag(TrackFlag::TF_VALID) = false;
    flag(TrackFlag::TF_TRIANGULATED) = true;
    return flag(TrackFlag::TF_VALID);
}

double Track::compute_baseline() const {
    double total = 0;
    for (auto i = keypoint_refs.begin(), j = std::next(i); j != keypoint_refs.end(); ++i, ++j) total += (i->first->pose.p - j->first->pose.p).norm();
    return total;
}

vector<3> Track::get_landmark_point() const {
    const auto fk = first_keypoint();
    const PoseState cam = fk.first->get_pose(fk.first->camera);
    return cam.q * fk.first->get_keypoint(fk.second).homogeneous() / landmark.inv_depth + cam.p;
}

void Track::set_landmark_point(const vector<3> &p) {
    const auto fk = first_keypoint();
    const PoseState cam = fk.first->get_pose(fk.first->camera);
    landmark.inv_depth = 1.0 / (cam.q.conjugate() * (p - cam.p)).z();
}

// ---- map/plane.cpp:116-134 ---------------------------------------------------------------------------------------
double Plane::point_to_plane_abs_distance(const vector<3> &point) const { return std::abs(parameter.normal.dot(point) - parameter.distance); }
double Plane::cast_to_depth(const vector<3> &origin, const vector<3> &direction) const {
    return (parameter.distance - parameter.normal.dot(origin)) / parameter.normal.dot(direction);
}
vector<3> Plane::cast_to_point(const vector<3> &origin, const vector<3> &direction) const { return origin + (direction * cast_to_depth(origin, direction)); }
bool Plane::is_parallel(const vector<3> &direction, double angle) const {
    return std::abs(direction.normalized().dot(parameter.normal)) < std::sin(angle * (M_PI / 180));
}

// ---- core/plane_extractor.cpp:184-203 ----------------------------------------------------------------------------
double PlaneExtractor::compute_reprojection_error(const Map *, const Track *track, const vector<3> &point) {
    double rpe = 0;
    size_t n = 0;
    for (const auto &fk : track->keypoint_map()) {
        const Frame *f = fk.first;
        const PoseState cam = f->get_pose(f->camera);
        const vector<3> y = cam.q.conjugate() * (point - cam.p);
        const vector<2> &z = f->get_keypoint(fk.second);
        const double du = (y.x() / y.z()) * f->K(0, 0) + f->K(0, 2) - (z[0] * f->K(0, 0) + f->K(0, 2));
        const double dv = (y.y() / y.z()) * f->K(1, 1) + f->K(1, 2) - (z[1] * f->K(1, 1) + f->K(1, 2));
        rpe += std::sqrt(du * du + dv * dv), ++n;
    }
    return n == 0 ? std::numeric_limits<double>::max() : rpe / (double)n;
}

double PlaneExtractor::enough_baseline(const Track *track) {
    const double baseline = track->compute_baseline();
    return (baseline > 0.5) || (track->landmark.inv_depth < (1 / 0.2) && baseline * track->landmark.inv_depth > 0.5);
}

} // namespace pvio
