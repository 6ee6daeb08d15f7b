// feature_front.h -- host side of the FeatureTracker seam above the C ABI (SURVEY.md section 8a rows K3, K5, section 8b
// row B2): the keypoint bookkeeping around the device LK tracker.  Everything here is scalar, order-dependent logic that
// decides WHICH tracks survive, so it stays on the host and follows the reference's behaviour decision by decision:
//
//   PoissonDisk2            pvio/src/pvio/utility/poisson_disk_filter.h:25-130 (2-D instance)
//   predict_keypoints       pvio/src/pvio/map/frame.cpp:97-103   (gyro-only rotation prediction)
//   select_tracked          pvio/src/pvio/map/frame.cpp:108-130  (track-length order + Poisson-disk acceptance)
//   HipImage                pvio-extra/src/pvio/extra/opencv_image.cpp:88-160 behind pvio::Image (pvio.h:114-133):
//                           preprocess() = CLAHE + pyramid + Scharr on the GPU, track_keypoints() = device LK + the 20 px
//                           border gate, detect_keypoints() = Harris corners on the GPU (goodFeaturesToTrack semantics) +
//                           Poisson-disk filter against the existing points + 20 px border; the fundamental-matrix RANSAC
//                           of :113-129 runs on the host (fundamental_ransac.h, a restatement of OpenCV's published
//                           algorithm); set_outlier_filter() replaces it with a caller-supplied rejection step.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <unordered_map>
#include <vector>

#include "../../include/pvio_hip.h"
#include "host_seam.h"

namespace pvio {

// Sparse-grid Poisson-disk acceptance test in the plane.  Cell size r / sqrt(2), neighbourhood +-2 cells, one point slot
// per cell (a later point overwrites the slot).  The neighbourhood walk reproduces the reference's probe sequence
// (poisson_disk_filter.h:92-110): the first cell of the block is skipped and one cell past the last row is probed.
class PoissonDisk2 {
  public:
    explicit PoissonDisk2(double radius);
    void clear();
    void preset_point(const vector<2> &p);            // unconditional insert
    bool permit_point(const vector<2> &p) const;      // no point closer than radius among the probed cells
    bool insert_point(const vector<2> &p);            // permit + insert
    void insert_points(std::vector<vector<2>> &candidates); // keeps the accepted ones, in order
    const std::vector<vector<2>> &points() const { return points_; }

  private:
    struct Key {
        int x, y;
        bool operator==(const Key &o) const { return x == o.x && y == o.y; }
    };
    struct KeyHash {
        size_t operator()(const Key &k) const;
    };
    Key cell_of(const vector<2> &p) const;
    bool test(const vector<2> &p, Key &cell) const;
    double radius_, radius2_, cell_;
    int span_;
    std::vector<vector<2>> points_;
    std::unordered_map<Key, size_t, KeyHash> grid_;
};

// next = K_next * hnormalized( (q_cam_i^-1 q_imu_i dq q_imu_j^-1 q_cam_j)^-1 * [kp; 1] ) for every normalized keypoint
void predict_keypoints(const Frame &curr, const Frame &next, std::vector<vector<2>> &next_pixels);

// Survivors of the tracker (status != 0), longest track first, pass a Poisson-disk filter of radius min_distance on
// their NEXT-image pixel positions; the ones that do not are cleared in `status`.  track_length[i] == 0 means "no track"
// (dropped from the candidates but its status byte is left alone, exactly like the reference's `continue`).
void select_tracked(const std::vector<vector<2>> &next_pixels, const std::vector<size_t> &track_length, double min_distance, std::vector<char> &status);

// pvio::Image on the GPU.  The pixels are uploaded once at construction; preprocess() builds the resident pyramid.
class HipImage : public Image {
  public:
    HipImage(pvio_hip_ctx *ctx, const uint8_t *pixels, int width, int height, int stride, double timestamp);
    ~HipImage() override;
    size_t width() const override { return (size_t)w_; }
    size_t height() const override { return (size_t)h_; }
    size_t level_num() const override { return 3; } // opencv_image.h: level_num() = 3 -> maxLevel 3, four levels
    // bicubic sample of the preprocessed pyramid level (and its gradient), opencv_image.cpp:36-52 (ceres::BiCubicInterpolator
    // over the level's pixels).  No caller inside the library; the level is copied back from the device on first use.
    double evaluate(const vector<2> &u, int level = 0) const override;
    double evaluate(const vector<2> &u, vector<2> &ddu, int level = 0) const override;
    void preprocess() override;
    void detect_keypoints(std::vector<vector<2>> &keypoints, size_t max_points, double keypoint_distance) const override;
    void track_keypoints(const Image *next_image, const std::vector<vector<2>> &curr_keypoints, std::vector<vector<2>> &next_keypoints, std::vector<char> &result_status) const override;
    // optional rejection step after the border gate (the place of the reference's fundamental-matrix RANSAC)
    using OutlierFilter = std::function<void(const std::vector<vector<2>> &curr, const std::vector<vector<2>> &next, std::vector<char> &status)>;
    void set_outlier_filter(OutlierFilter f) { filter_ = std::move(f); }
    void enable_ransac(bool on) { ransac_ = on; } // default on, like the reference; a custom filter takes precedence
    const pvio_hip_image *device_image() const { return img_; }

  protected: // UndistortedHipImage (dataset_reader.h) builds the pyramid through another C-ABI entry point
    void forget_host_levels() { host_levels_.clear(); } // evaluate()'s copies belong to the previous pyramid
    pvio_hip_ctx *ctx_;
    std::vector<uint8_t> pixels_;
    int w_, h_;
    pvio_hip_image *img_ = nullptr;

  private:
    struct HostLevel {
        int w = 0, h = 0;
        std::vector<uint8_t> px;
    };
    const HostLevel &host_level(int level) const;
    mutable std::vector<HostLevel> host_levels_;
    OutlierFilter filter_;
    bool ransac_ = true;
};

} // namespace pvio
