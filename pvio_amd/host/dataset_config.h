// dataset_config.h -- the camera / IMU constants of the two public datasets pvio-pc ships configurations for, as a pvio::Config
// (pvio/include/pvio/pvio.h:70-112).  The reference parses them from config/euroc.yaml:11-45 / config/tum-vi.yaml:13-43 with yaml-cpp
// (pvio-extra/.../yaml_config.cpp); the headless driver (tools/pvio_headless.cpp) has no yaml parser and takes
// them from here.  Everything the yaml files leave at the defaults of pvio/src/pvio/config.cpp stays at those defaults.
#pragma once
#include <memory>

#include "host_seam.h"

namespace pvio {

class DatasetConfig : public Config {
  public:
    static std::shared_ptr<DatasetConfig> euroc();   // config/euroc.yaml
    static std::shared_ptr<DatasetConfig> tum_vi();  // config/tum-vi.yaml
    // constants of a rig given by the caller (tests: rendered sequences), noise parameters as in euroc.yaml
    static std::shared_ptr<DatasetConfig> make(const double K4[4], const double q_bc_xyzw[4], const double p_bc[3], double cov_gyr, double cov_acc,
                                               double cov_bias_gyr, double cov_bias_acc, bool normalize_q = true);
    matrix<3> camera_intrinsic() const override { return K; }
    quaternion camera_to_body_rotation() const override { return q_bc; }
    vector<3> camera_to_body_translation() const override { return p_bc; }
    quaternion imu_to_body_rotation() const override { return q_bi; }
    vector<3> imu_to_body_translation() const override { return p_bi; }
    matrix<2> keypoint_noise_cov() const override { return cov_kp; }
    matrix<3> gyroscope_noise_cov() const override { return cov_g; }
    matrix<3> accelerometer_noise_cov() const override { return cov_a; }
    matrix<3> gyroscope_bias_noise_cov() const override { return cov_bg; }
    matrix<3> accelerometer_bias_noise_cov() const override { return cov_ba; }
    size_t sliding_window_size() const override { return window; }
    double feature_tracker_min_keypoint_distance() const override { return min_keypoint_distance; }
    size_t solver_iteration_limit() const override { return iteration_limit; }
#ifdef PVIO_HOST_USE_REFERENCE_TYPES
    size_t initializer_keyframe_gap() const override { return keyframe_gap; } // (the stand-in Config of the test build has no initializer)
#endif
    size_t initializer_keyframe_gap_() const { return keyframe_gap; }

    matrix<3> K;
    quaternion q_bc, q_bi;
    vector<3> p_bc, p_bi;
    matrix<2> cov_kp;
    matrix<3> cov_g, cov_a, cov_bg, cov_ba;
    size_t window = 8, keyframe_gap = 5, iteration_limit = 10; // sliding_window_size, initializer_keyframe_gap, solver_iteration_limit of the yaml files
    double min_keypoint_distance = 25.0;
};

} // namespace pvio
