// TEST SCAFFOLDING (tests/host/standin/, moved out of the product directory in round 3): look-alike declarations so that the adapter
// sources can RUN in the tests; inside the PVIO tree the reference's own headers are used and nothing here is compiled.
//
// pvio_min.h -- the part of the reference's class surface the host seam touches, for builds OUTSIDE the PVIO tree.
//
// The adapter sources (bundle_adjustor.cpp, pnp.cpp, feature_front.cpp, feature_tracker.cpp) are written against the
// reference's own headers and are compiled two ways from the same text:
//   * inside the PVIO tree (or `make -C tests/host refcheck`): -DPVIO_HOST_USE_REFERENCE_TYPES, includes
//       <pvio/pvio.h>, <pvio/common.h>, <pvio/estimation/{state,preintegrator,factor,bundle_adjustor}.h>,
//       <pvio/map/{map,frame,track,plane}.h>, <pvio/core/plane_extractor.h>
//     straight from pvio/include and pvio/src -- nothing in this file is used;
//   * standalone (tests, bench, the headless driver): this file, which declares the SAME public names with the same
//     signatures, value types (Eigen typedefs, pvio.h:28-40), containers and ORDERINGS the flattening order depends on
//     (Track::keypoint_map() ordered by frame id: map/track.h:69,108 + common.h:81-86 + utility/identifiable.h:32-34;
//     Plane::tracks ordered by track id: map/plane.h:45).  The member functions of the map layer that the reference
//     implements in map/*.cpp, core/plane_extractor.cpp and geometry/stereo.h (outside the hot path)
//     have a minimal stand-in in pvio_min.cpp so that the adapter can run without the PVIO tree.
// Anything the adapter does with these types must therefore compile against both; `refcheck` is part of build().
#pragma once
#include <Eigen/Eigen> // the real Eigen when installed; tests/host/eigen_stub otherwise

#include <algorithm>
#include <bitset>
#include <cstddef>
#include <deque>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>
#include <vector>

namespace pvio {

// ---- pvio/include/pvio/pvio.h:28-40 ------------------------------------------------------------------------------
template <int Rows = Eigen::Dynamic, int Cols = Rows, bool UseRowMajor = false, typename T = double>
using matrix = typename std::conditional<Rows != 1 && Cols != 1, Eigen::Matrix<T, Rows, Cols, UseRowMajor ? Eigen::RowMajor : Eigen::ColMajor>,
                                         Eigen::Matrix<T, Rows, Cols>>::type;
template <int Dimension = Eigen::Dynamic, bool RowVector = false, typename T = double>
using vector = typename std::conditional<RowVector, matrix<1, Dimension, false, T>, matrix<Dimension, 1, false, T>>::type;
using quaternion = Eigen::Quaternion<double>;

struct OutputPose { // pvio.h:42-45
    quaternion q;
    vector<3> p;
};

class Config { // pvio.h:70-112, defaults of config.cpp:24-93
  public:
    virtual ~Config() = default;
    virtual matrix<3> camera_intrinsic() const = 0;
    virtual quaternion camera_to_body_rotation() const = 0;
    virtual vector<3> camera_to_body_translation() const = 0;
    virtual quaternion imu_to_body_rotation() const = 0;
    virtual vector<3> imu_to_body_translation() const = 0;
    virtual matrix<2> keypoint_noise_cov() const = 0;
    virtual matrix<3> gyroscope_noise_cov() const = 0;
    virtual matrix<3> accelerometer_noise_cov() const = 0;
    virtual matrix<3> gyroscope_bias_noise_cov() const = 0;
    virtual matrix<3> accelerometer_bias_noise_cov() const = 0;
    virtual double plane_distance_cov() const { return 0.01 * 0.01; }
    virtual quaternion output_to_body_rotation() const { return quaternion::Identity(); }
    virtual vector<3> output_to_body_translation() const { return vector<3>::Zero(); }
    virtual size_t sliding_window_size() const { return 10; }
    virtual double feature_tracker_min_keypoint_distance() const { return 20.0; }
    virtual size_t feature_tracker_max_keypoint_detection() const { return 150; }
    virtual size_t feature_tracker_max_init_frames() const { return 60; }
    virtual size_t feature_tracker_max_frames() const { return 20; }
    virtual bool feature_tracker_predict_keypoints() const { return true; }
    virtual size_t solver_iteration_limit() const { return 10; }
    virtual double solver_time_limit() const { return 1.0e6; }
    virtual int random() const { return 648; }
};

class Image { // pvio.h:114-133
  public:
    double t;
    virtual size_t width() const = 0;
    virtual size_t height() const = 0;
    virtual size_t level_num() const { return 0; }
    virtual double evaluate(const vector<2> &u, int level = 0) const = 0;
    virtual double evaluate(const vector<2> &u, vector<2> &ddu, int level = 0) const = 0;
    virtual ~Image() = default;
    virtual void preprocess() {}
    virtual void detect_keypoints(std::vector<vector<2>> &keypoints, size_t max_points = 0, double keypoint_distance = 0.5) const = 0;
    virtual void track_keypoints(const Image *next_image, const std::vector<vector<2>> &curr_keypoints, std::vector<vector<2>> &next_keypoints, std::vector<char> &result_status) const = 0;
};

// ---- pvio/src/pvio/common.h:69-129 -------------------------------------------------------------------------------
inline constexpr size_t nil() { return size_t(-1); }

template <typename T>
struct compare;
template <typename T>
struct compare<T *> {
    constexpr bool operator()(const T *a, const T *b) const { return std::less<T>()(*a, *b); }
};

template <class FlagEnum>
struct Flagged {
    static const size_t flag_num = static_cast<size_t>(FlagEnum::FLAG_NUM);
    bool flag(FlagEnum f) const { return flags[static_cast<size_t>(f)]; }
    typename std::bitset<flag_num>::reference flag(FlagEnum f) { return flags[static_cast<size_t>(f)]; }
    bool any_of(std::initializer_list<FlagEnum> fs) const {
        return std::any_of(fs.begin(), fs.end(), [this](FlagEnum f) { return flag(f); });
    }

  private:
    std::bitset<flag_num> flags;
};

struct ImuData {
    double t;
    vector<3> w;
    vector<3> a;
};

// ---- utility/identifiable.h:24-56 --------------------------------------------------------------------------------
template <typename T>
class Identifiable {
  public:
    size_t id() const { return id_value; }
    bool operator<(const T &other) const { return id_value < other.id_value; }

  protected:
    Identifiable() : Identifiable(generate_id()) {}
    Identifiable(size_t id_value) : id_value(id_value) {}

  private:
    static size_t generate_id() {
        static size_t s_id = 0;
        return ++s_id;
    }
    const size_t id_value;
};

// ---- estimation/state.h:29-88 ------------------------------------------------------------------------------------
enum ErrorStateLocation { ES_Q = 0, ES_P = 3, ES_V = 6, ES_BG = 9, ES_BA = 12, ES_SIZE = 15 };
struct ExtrinsicParams {
    quaternion q_cs;
    vector<3> p_cs;
};
struct PoseState {
    PoseState() { q.setIdentity(), p.setZero(); }
    quaternion q;
    vector<3> p;
};
struct MotionState {
    MotionState() { v.setZero(), bg.setZero(), ba.setZero(); }
    vector<3> v, bg, ba;
};
struct LandmarkState {
    double inv_depth = 0, quality = 0;
    size_t plane_id = nil();
};
struct PlaneState {
    vector<3> normal;
    double distance;
    vector<3> reference_point;
};

class Frame;
class Track;
class Map;
class Plane;

// ---- estimation/preintegrator.h:27-62 (integrate() -> pvio_preintegrate of the C ABI) ----------------------------
struct PreIntegrator {
    struct Delta {
        double t;
        quaternion q;
        vector<3> p, v;
        matrix<15> cov, sqrt_inv_cov;
    };
    struct Jacobian {
        matrix<3> dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba;
    };
    bool integrate(double t, const vector<3> &bg, const vector<3> &ba, bool compute_jacobian, bool compute_covariance);
    void predict(const Frame *old_frame, Frame *new_frame); // preintegrator.cpp:102-108
    matrix<3> cov_w, cov_a, cov_bg, cov_ba;
    Delta delta;
    Jacobian jacobian;
    std::vector<ImuData> data;
};

// ---- estimation/factor.h:28-52 -----------------------------------------------------------------------------------
class Factor {
    struct factor_construct_t {};

  public:
    struct FactorCostFunction {
        virtual ~FactorCostFunction() = default;
        virtual void update() = 0;
    };
    static std::unique_ptr<Factor> create_marginalization_error(const matrix<> &sqrt_inv_cov, const vector<> &infovec, std::vector<Frame *> &&frames);
    static std::unique_ptr<Factor> create_reprojection_error(Track *track, Frame *frame, size_t keypoint_index);
    static std::unique_ptr<Factor> create_preintegration_error(Frame *frame_i, Frame *frame_j);
    template <typename T>
    T *get_cost_function() { return static_cast<T *>(cost_function.get()); }
    Factor(std::unique_ptr<FactorCostFunction> cost_function, const factor_construct_t &) : cost_function(std::move(cost_function)) {}
    virtual ~Factor() = default;

  private:
    std::unique_ptr<FactorCostFunction> cost_function;
};

// ---- map/frame.h:37-111 ------------------------------------------------------------------------------------------
struct create_if_empty_t {};
extern create_if_empty_t create_if_empty;

enum class FrameFlag { FF_KEYFRAME = 0, FF_FIX_POSE, FLAG_NUM };

class Frame : public Flagged<FrameFlag>, public Identifiable<Frame> {
    friend class Track;
    friend class Map;
    Map *map = nullptr;

  public:
    struct construct_by_frame_t {};
    Frame() = default;
    Frame(size_t id, const construct_by_frame_t &) : Identifiable(id) {}
    virtual ~Frame() = default;
    std::unique_ptr<Frame> clone() const; // same id, keypoints kept, tracks / factors / map dropped (frame.cpp:43-58)
    size_t keypoint_num() const { return keypoints.size(); }
    void append_keypoint(const vector<2> &keypoint);
    const vector<2> &get_keypoint(size_t keypoint_index) const { return keypoints[keypoint_index]; }
    Track *get_track(size_t keypoint_index) const { return tracks[keypoint_index]; }
    Track *get_track(size_t keypoint_index, const create_if_empty_t &);
    void detect_keypoints(Config *config);                   // frame.cpp:72-87
    void track_keypoints(Frame *next_frame, Config *config); // frame.cpp:89-139
    Factor *get_reprojection_factor(size_t keypoint_index) { return reprojection_factors[keypoint_index].get(); }
    Factor *get_preintegration_factor() { return preintegration_factor.get(); }
    PoseState get_pose(const ExtrinsicParams &sensor) const;
    void set_pose(const ExtrinsicParams &sensor, const PoseState &pose);
    bool has_map() const { return map != nullptr; }

    matrix<3> K;
    matrix<2> sqrt_inv_cov;
    std::shared_ptr<Image> image;
    PoseState pose;
    MotionState motion;
    ExtrinsicParams camera, imu;
    PreIntegrator preintegration;

  private:
    std::vector<vector<2>> keypoints;
    std::vector<Track *> tracks;
    std::vector<std::unique_ptr<Factor>> reprojection_factors;
    std::unique_ptr<Factor> preintegration_factor;
};

// ---- map/map.h:29-112 --------------------------------------------------------------------------------------------
class Map {
    friend class Track;
    struct construct_by_map_t {};

  public:
    Map();
    virtual ~Map();
    size_t frame_num() const { return frames.size(); }
    Frame *get_frame(size_t index) const { return frames[index].get(); }
    Frame *first_frame() const { return frames[0].get(); }
    Frame *last_frame() const { return frames[frames.size() - 1].get(); }
    void put_frame(std::unique_ptr<Frame> frame, size_t position = nil());
    void erase_frame(size_t index);
    void marginalize_frame(size_t index);
    size_t frame_index_by_id(size_t id) const;
    size_t track_num() const { return tracks.size(); }
    Track *get_track(size_t index) const { return tracks[index].get(); }
    Track *create_track();
    void erase_track(Track *track);
    void prune_tracks(const std::function<bool(const Track *)> &condition);
    size_t plane_num() const { return planes.size(); }
    Plane *get_plane(size_t index) const { return planes[index].get(); }
    void put_plane(std::unique_ptr<Plane> plane); // stand-in: appends (the reference merges overlapping planes, map.cpp:143-166)
    void set_marginalization_factor(std::unique_ptr<Factor> factor);
    Factor *get_marginalization_factor() { return marginalization_factor.get(); }

  private:
    void recycle_track(Track *track);
    std::deque<std::unique_ptr<Frame>> frames;
    std::vector<std::unique_ptr<Plane>> planes;
    std::vector<std::unique_ptr<Track>> tracks;
    std::unique_ptr<Factor> marginalization_factor;
};

// ---- map/track.h:29-108 ------------------------------------------------------------------------------------------
enum class TrackFlag { TF_VALID = 0, TF_TRIANGULATED, TF_PLANE, FLAG_NUM };

class Track : public Flagged<TrackFlag>, public Identifiable<Track> {
    friend class Map;
    size_t map_index = 0;
    Map *map = nullptr;
    Track() : life(0) {}

  public:
    Track(const Map::construct_by_map_t &) : Track() {}
    virtual ~Track() = default;
    size_t keypoint_num() const { return keypoint_refs.size(); }
    std::pair<Frame *, size_t> first_keypoint() const { return *keypoint_refs.begin(); }
    std::pair<Frame *, size_t> last_keypoint() const { return *keypoint_refs.rbegin(); }
    Frame *first_frame() const { return keypoint_refs.begin()->first; }
    Frame *last_frame() const { return keypoint_refs.rbegin()->first; }
    const std::map<Frame *, size_t, compare<Frame *>> &keypoint_map() const { return keypoint_refs; }
    bool has_keypoint(Frame *frame) const { return keypoint_refs.count(frame) > 0; }
    size_t get_keypoint_index(Frame *frame) const { return has_keypoint(frame) ? keypoint_refs.at(frame) : nil(); }
    const vector<2> &get_keypoint(Frame *frame) const;
    void add_keypoint(Frame *frame, size_t keypoint_index);
    void remove_keypoint(Frame *frame, bool suicide_if_empty = true);
    bool triangulate();
    bool try_triangulate(vector<3> &p);
    double compute_baseline() const;
    vector<3> get_landmark_point() const;
    void set_landmark_point(const vector<3> &p);

    LandmarkState landmark;
    size_t life;

  private:
    std::map<Frame *, size_t, compare<Frame *>> keypoint_refs;
};

// ---- map/plane.h:31-47 (sector area left out) --------------------------------------------------------------------
class Plane : public Identifiable<Plane> {
  public:
    double point_to_plane_abs_distance(const vector<3> &point) const;
    double cast_to_depth(const vector<3> &origin, const vector<3> &direction) const;
    vector<3> cast_to_point(const vector<3> &origin, const vector<3> &direction) const;
    bool is_parallel(const vector<3> &direction, double angle = 10) const;
    PlaneState parameter;
    std::set<Track *, compare<Track *>> tracks;
};

// ---- core/plane_extractor.h:43-45 (the two static helpers the estimation seam calls) -----------------------------
class PlaneExtractor {
  public:
    static double compute_reprojection_error(const Map *map, const Track *track, const vector<3> &point);
    static double enough_baseline(const Track *track);
};

// ---- estimation/bundle_adjustor.h:29-42, estimation/pnp.h:26 -----------------------------------------------------
class BundleAdjustor {
    struct BundleAdjustorSolver; // pimpl

  public:
    BundleAdjustor();
    virtual ~BundleAdjustor();
    bool solve(Map *map, Config *config, bool use_inertial = true);
    double compute_reprojection_error(Map *map);
    void marginalize_frame(Map *map, size_t index);

  private:
    std::unique_ptr<BundleAdjustorSolver> solver;
};

void visual_inertial_pnp(Map *map, Frame *frame, Config *config, bool use_inertial = true);

} // namespace pvio
