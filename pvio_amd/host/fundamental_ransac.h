// fundamental_ransac.h -- the outlier rejection step of OpenCvImage::track_keypoints
// (pvio-extra/src/pvio/extra/opencv_image.cpp:113-129): cv::findFundamentalMat(p, q, FM_RANSAC, 1.0, 0.99, mask), of which the
// reference only uses the inlier mask.  OpenCV is a third-party dependency that is neither in /root/reference nor
// installed here, so this is a restatement of its published RANSAC (calib3d: RANSACPointSetRegistrator +
// FMEstimatorCallback) from the algorithm's description -- PARITY UNPINNED:
//   * sampler: the multiply-with-carry generator of cv::RNG seeded with (uint64)-1, `uniform(0, n)` = next() % n, seven
//     distinct indices per sample, a sample is redrawn when its last point is collinear with two earlier ones;
//   * model: 7-point algorithm (null space of the 7 x 9 epipolar constraints, real roots of det(l F1 + (1 - l) F2) = 0);
//   * score: max of the two squared point-to-epipolar-line distances, inlier when <= threshold^2;
//   * the best model needs strictly more inliers than the previous best (and at least 7); the iteration count adapts as
//     log(1 - confidence) / log(1 - (1 - eps)^7), capped at 1000.
// Host code (a few hundred points, a few hundred samples): SURVEY.md section 8f row 2.
#pragma once
#include "host_namespace.h"
#include <cstdint>
#include <vector>

namespace pvio {

// p, q: n points (x, y) each, single precision like the reference's cv::Point2f.  mask[i] = 1 for inliers of the best model.
// Returns the number of inliers (0: no model found, mask all zero).  n < 7 returns 0.
int find_fundamental_ransac(int n, const float *p_xy, const float *q_xy, double threshold, double confidence, std::vector<uint8_t> &mask, double F_out[9] = nullptr,
                            int max_iterations = 1000);

// the 7-point solver alone: up to three 3 x 3 matrices (row-major) with q^T F p = 0 for the seven correspondences
int fundamental_7point(const float p_xy[14], const float q_xy[14], double F[27]);

} // namespace pvio
