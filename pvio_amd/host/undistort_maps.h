// undistort_maps.h -- host side of the image-undistortion step in front of the feature tracker:
// builds, once per camera, the fixed-point remap tables the device kernel k_remap consumes (pvio_hip_undistort_create).
//
//   cv_undistort_fixed_maps   the map cv::undistort(img, K, dist) applies in the EuRoC reader
//                             (pvio-pc/src/euroc_dataset_reader.cpp:72-75; K and dist are float32 matrices there):
//                             OpenCV's initUndistortRectifyMap(K, dist, I, K, size, CV_16SC2) evaluated in row stripes,
//                             restated from the published algorithm (OpenCV itself is not in /root/reference: unpinned)
//   convert_maps_fixed        cv::convertMaps(map_x, map_y, CV_16SC2) (image_undistorter.h:41)
//   ImageUndistorter          pvio-extra/include/pvio/extra/image_undistorter.h:27-110: radtan / equidistant
//                             distort_pixel() per destination pixel in double, float32 maps, convertMaps; the remap itself
//                             (undistort_image, :44-46) runs on the device
//
// The remap is NOT done here: there is no host pixel path (the product has no CPU fallback).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "host_seam.h" // pvio::matrix / vector (the reference's value types, or the test stand-ins)

namespace pvio {

struct FixedRemap {
    int width = 0, height = 0;
    std::vector<int16_t> xy;    // [height][width][2]: integer source position (x, y)
    std::vector<uint16_t> frac; // [height][width]: (fy << 5) | fx, 5-bit fractions (OpenCV INTER_BITS = 5)
};

// K row-major 3x3, dist = (k1, k2, p1, p2[, k3]) as float32 (converted to double like cv::undistort does)
FixedRemap cv_undistort_fixed_maps(const float K[9], const float *dist, int n_dist, int width, int height);

FixedRemap convert_maps_fixed(const float *map_x, const float *map_y, int width, int height);

class ImageUndistorter {
  public:
    ImageUndistorter(size_t width, size_t height, const matrix<3> &K, const std::vector<double> &distort_coeffs, const std::string &model);
    vector<2> distort_pixel(const vector<2> &undistort_location) const;
    const FixedRemap &maps() const { return maps_; }
    size_t width() const { return width_; }
    size_t height() const { return height_; }

  private:
    size_t width_, height_;
    matrix<3> K_, Kinv_;
    std::vector<double> coeffs_;
    std::string model_;
    FixedRemap maps_;
};

} // namespace pvio
