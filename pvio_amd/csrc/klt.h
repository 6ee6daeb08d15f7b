// klt.h -- device-resident image pyramids + pyramidal Lucas-Kanade tracker (host interface).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace pvklt {

struct Image;     // device pyramid (klt.hip)
struct Undistort; // device-resident fixed-point remap tables (klt.hip)

class Klt {
  public:
    explicit Klt(int device);
    ~Klt();
    // `ud` != nullptr: the pixels are first undistorted on the device (cv::remap semantics), the pyramid has the map's size
    int create_image(const uint8_t *pixels, int w, int h, int stride, bool clahe, Image **out, const Undistort *ud = nullptr);
    int create_undistort(const int16_t *map_xy, const uint16_t *map_frac, int w, int h, Undistort **out);
    void release_undistort(Undistort *u);
    void release_image(Image *img);
    int download_level(const Image *img, int level, uint8_t *pixels, int16_t *deriv, int32_t *w, int32_t *h);
    int track(const Image *prev, const Image *next, int n, const float *prev_xy, float *next_xy, uint8_t *status);
    // Harris corners of level 0 (cv::goodFeaturesToTrack semantics); xy / response need room for max_corners entries
    int detect(const Image *img, int max_corners, double quality, double min_distance, float *xy, float *response, int *n_out);
    int download_response(const Image *img, float *resp);
    // cv::findFundamentalMat(p, q, FM_RANSAC, threshold, confidence) with the hypotheses evaluated in batches on the device; mask[n]
    int fundamental_ransac(int n, const float *p, const float *q, double threshold, double confidence, int max_iterations, uint8_t *mask, double *F_out, int *n_inliers);
    int last_ransac_hypotheses() const { return last_fm_hypotheses_; } // hypotheses the last run evaluated (diagnostics)
    const std::string &error() const { return err_; }
    double last_track_ms() const { return last_ms_; } // hipEvent time of the last k_lk_track launch

  private:
    int device_;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    double last_ms_ = 0;
    std::string err_;
    void *d_pts_ = nullptr;
    size_t pts_cap_ = 0;
    void *h_pts_ = nullptr; // pinned mirror of d_pts_
    int n_simds_ = 0;    // 4 x CUs
    int lk_form_ = 0;    // PVIO_HIP_LK_FORM (experiments and tests): 0 (default) k_lk_track_levels (a workgroup per track, a wave per level) up to three tracks per CU, k_lk_track (a wave per track) beyond; 1 always k_lk_track; 2 the unit queue (k_lk_track_units); 3 always k_lk_track_levels
    int lk_blocks_ = 0;  // PVIO_HIP_LK_BLOCKS: blocks of the unit queue (default one per CU)
    void *d_fm_ = nullptr, *h_fm_ = nullptr; // fundamental_ransac: points, samples, counts, models, mask words (+ pinned mirror)
    size_t fm_cap_ = 0;
    int last_fm_hypotheses_ = 0;
    void *d_src_ = nullptr; // distorted source pixels of the image being built
    size_t src_cap_ = 0;
    void *d_det_ = nullptr; // detection scratch: cov planes, response map, candidates
    size_t det_cap_ = 0;
    void *det_host_ = nullptr;         // pinned: counts + the first slice of the candidate list
    std::vector<uint64_t> det_keys_;   // (response bits, address) keys of the candidates, reused
    std::vector<int> det_grid_cnt_;    // minimum-distance grid of the selection, reused
    std::vector<float> det_grid_xy_;
    // released pyramid slabs, reused by the next image of the same size: a camera stream allocates once (hipMalloc +
    // hipFree per frame cost more than the whole pyramid build)
    std::vector<std::pair<size_t, void *>> slab_pool_;
    void *staging_ = nullptr; // pinned host buffer for the pixel upload
    size_t staging_cap_ = 0;
};

} // namespace pvklt
