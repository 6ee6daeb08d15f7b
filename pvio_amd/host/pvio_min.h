// pvio_min.h -- minimal, Eigen-free stand-ins for the reference types the BundleAdjustor seam touches.
//
// The reference's public value types are Eigen typedefs (pvio/include/pvio/pvio.h:28-40) and Eigen is not installed in
// this environment, so the adapter (bundle_adjustor.cpp) is compiled here against these look-alikes: same names, same
// members, same memory layout (column vectors contiguous, quaternion coefficients x,y,z,w) and -- crucially -- the same
// ORDERING guarantees the reference's flattening order depends on (Track::keypoint_map() is a std::map ordered by
// frame id: map/track.h:69, common.h:81-86, utility/identifiable.h:32-34).  Inside the PVIO tree the adapter is built
// with -DPVIO_HOST_USE_REFERENCE_TYPES against the real headers instead.  This is NOT a re-implementation of the map
// layer (out of scope): no triangulation, no pruning, no plane extraction.
#pragma once
#include <bitset>
#include <cstddef>
#include <map>
#include <memory>
#include <set>
#include <vector>

namespace pvio {

template <int N>
struct vector {
    double v[N] = {};
    double *data() { return v; }
    const double *data() const { return v; }
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double &x() { return v[0]; }
    double &y() { return v[1]; }
    double &z() { return v[2]; }
};
template <int R, int C = R>
struct matrix { // column-major like Eigen's default
    double m[R * C] = {};
    double &operator()(int r, int c) { return m[c * R + r]; }
    double operator()(int r, int c) const { return m[c * R + r]; }
    double *data() { return m; }
    const double *data() const { return m; }
};
struct quaternion {
    double c[4] = {0, 0, 0, 1}; // x y z w
    struct Coeffs {
        double *p;
        double *data() { return p; }
    };
    Coeffs coeffs() { return Coeffs{c}; }
    const double *coeffs_data() const { return c; }
};

enum ErrorStateLocation { ES_Q = 0, ES_P = 3, ES_V = 6, ES_BG = 9, ES_BA = 12, ES_SIZE = 15 }; // estimation/state.h:29-36
struct ExtrinsicParams {
    quaternion q_cs;
    vector<3> p_cs;
};
struct PoseState {
    quaternion q;
    vector<3> p;
};
struct MotionState {
    vector<3> v, bg, ba;
};
struct LandmarkState {
    double inv_depth = 0, quality = 0;
    size_t plane_id = size_t(-1);
};
struct ImuData {
    double t;
    vector<3> w, a;
};

struct PreIntegrator { // estimation/preintegrator.h:27-62
    struct Delta {
        double t = 0;
        quaternion q;
        vector<3> p, v;
        double cov[225] = {}, sqrt_inv_cov[225] = {}; // row-major here
    } delta;
    struct Jacobian {
        double dq_dbg[9] = {}, dp_dbg[9] = {}, dp_dba[9] = {}, dv_dbg[9] = {}, dv_dba[9] = {};
    } jacobian;
    double cov_w[9] = {}, cov_a[9] = {}, cov_bg[9] = {}, cov_ba[9] = {};
    std::vector<ImuData> data;
    bool integrate(double t, const vector<3> &bg, const vector<3> &ba, bool compute_jacobian, bool compute_covariance); // -> pvio_preintegrate
};

enum class FrameFlag { FF_KEYFRAME = 0, FF_FIX_POSE, FLAG_NUM };
enum class TrackFlag { TF_VALID = 0, TF_TRIANGULATED, TF_PLANE, FLAG_NUM };

class Track;
class Frame {
  public:
    size_t id_ = 0;
    size_t id() const { return id_; }
    std::bitset<2> flags;
    bool flag(FrameFlag f) const { return flags[(size_t)f]; }
    matrix<3> K;
    matrix<2> sqrt_inv_cov;
    double image_t = 0; // frame->image->t
    PoseState pose;
    MotionState motion;
    ExtrinsicParams camera, imu;
    PreIntegrator preintegration;
    bool has_preintegration_factor = false; // frame->get_preintegration_factor() != nullptr
    std::vector<vector<2>> keypoints;
    std::vector<Track *> tracks;
    size_t keypoint_num() const { return keypoints.size(); }
    const vector<2> &get_keypoint(size_t i) const { return keypoints[i]; }
    Track *get_track(size_t i) const { return tracks[i]; }
};
struct FrameIdLess {
    bool operator()(const Frame *a, const Frame *b) const { return a->id() < b->id(); }
};
class Track {
  public:
    size_t id_ = 0, life = 0;
    std::bitset<3> flags;
    bool flag(TrackFlag f) const { return flags[(size_t)f]; }
    void set_flag(TrackFlag f, bool v) { flags[(size_t)f] = v; }
    LandmarkState landmark;
    std::map<Frame *, size_t, FrameIdLess> keypoint_refs;
    const std::map<Frame *, size_t, FrameIdLess> &keypoint_map() const { return keypoint_refs; }
    Frame *first_frame() const { return keypoint_refs.begin()->first; }
    std::pair<Frame *, size_t> first_keypoint() const { return *keypoint_refs.begin(); }
    size_t keypoint_num() const { return keypoint_refs.size(); }
};
struct Plane {
    size_t id_ = 0;
    size_t id() const { return id_; }
    struct {
        vector<3> normal;
        double distance = 0;
    } parameter;
    std::set<Track *> tracks;
};
struct MarginalizationPrior { // MarginalizationErrorCost's state (marginalization_error_cost.h:96-105)
    std::vector<double> sqrt_infomat, sqrt_infovec; // row-major 15n x 15n, 15n
    std::vector<Frame *> frames;
    std::vector<PoseState> pose_0;
    std::vector<MotionState> motion_0;
};
class Map {
  public:
    std::vector<std::unique_ptr<Frame>> frames;
    std::vector<std::unique_ptr<Track>> tracks;
    std::vector<std::unique_ptr<Plane>> planes;
    std::unique_ptr<MarginalizationPrior> prior;
    size_t frame_num() const { return frames.size(); }
    Frame *get_frame(size_t i) const { return frames[i].get(); }
    size_t track_num() const { return tracks.size(); }
    Track *get_track(size_t i) const { return tracks[i].get(); }
    size_t plane_num() const { return planes.size(); }
    Plane *get_plane(size_t i) const { return planes[i].get(); }
    MarginalizationPrior *get_marginalization_factor() const { return prior.get(); }
    void set_marginalization_factor(std::unique_ptr<MarginalizationPrior> p) { prior = std::move(p); }
};
class Config { // the three values the seam reads (pvio.h:70-112)
  public:
    virtual ~Config() = default;
    virtual size_t solver_iteration_limit() const { return 10; }
    virtual double solver_time_limit() const { return 1.0e6; }
    virtual double plane_distance_cov() const { return 1.0e-4; }
};

class Image { // pvio/include/pvio/pvio.h:114-133 (evaluate() has no caller in the library and is left out)
  public:
    double t = 0;
    virtual size_t width() const = 0;
    virtual size_t height() const = 0;
    virtual size_t level_num() const { return 0; }
    virtual ~Image() = default;
    virtual void preprocess() {}
    virtual void detect_keypoints(std::vector<vector<2>> &keypoints, size_t max_points = 0, double keypoint_distance = 0.5) const = 0;
    virtual void track_keypoints(const Image *next_image, const std::vector<vector<2>> &curr_keypoints, std::vector<vector<2>> &next_keypoints, std::vector<char> &result_status) const = 0;
};

class BundleAdjustor { // estimation/bundle_adjustor.h:29-42
  public:
    BundleAdjustor();
    virtual ~BundleAdjustor();
    bool solve(Map *map, Config *config, bool use_inertial = true);
    void marginalize_frame(Map *map, size_t index);
    double compute_reprojection_error(Map *map);
};

} // namespace pvio
