// dataset_config.cpp -- see dataset_config.h.
#include "dataset_config.h"

namespace pvio {

namespace {
matrix<3> diag3(double v) {
    matrix<3> m;
    m.setZero();
    m(0, 0) = m(1, 1) = m(2, 2) = v;
    return m;
}
} // namespace

std::shared_ptr<DatasetConfig> DatasetConfig::make(const double K4[4], const double q_bc[4], const double p_bc[3], double g, double a, double bg, double ba, bool normalize_q) {
    auto c = std::make_shared<DatasetConfig>();
    c->K.setZero();
    c->K(0, 0) = K4[0], c->K(1, 1) = K4[1], c->K(0, 2) = K4[2], c->K(1, 2) = K4[3], c->K(2, 2) = 1.0;
    c->q_bc = quaternion(q_bc[3], q_bc[0], q_bc[1], q_bc[2]);
    // the dataset presets keep the yaml's coefficients as they are, like the reference (pvio-extra/src/pvio/extra/yaml_config.cpp:128-129 assigns them
    // raw: tum-vi.yaml's q_bc has |q|^2 = 1.0000009, and a normalized copy differs from pvio-pc's rig at 5e-7 -- ADVICE r5); a caller's rig is normalized
    if (normalize_q) c->q_bc.normalize();
    c->p_bc = vector<3>(p_bc[0], p_bc[1], p_bc[2]);
    c->q_bi = quaternion::Identity(), c->p_bi = vector<3>::Zero(); // imu: extrinsic identity in both files (euroc.yaml:44-45, tum-vi.yaml:42-43)
    c->cov_kp.setZero();
    c->cov_kp(0, 0) = c->cov_kp(1, 1) = 0.5;                      // camera: noise [0.5, 0, 0, 0.5] pixel^2
    c->cov_g = diag3(g), c->cov_a = diag3(a), c->cov_bg = diag3(bg), c->cov_ba = diag3(ba);
    return c;
}

std::shared_ptr<DatasetConfig> DatasetConfig::euroc() { // config/euroc.yaml:15-45
    const double K4[4] = {458.654, 457.296, 367.215, 248.375};
    const double q[4] = {-7.7071797555374275e-03, 1.0499323370587278e-02, 7.0175280029197162e-01, 7.1230146066895372e-01};
    const double p[3] = {-0.0216401454975, -0.064676986768, 0.00981073058949};
    return make(K4, q, p, 2.8791302399999997e-08, 4.0e-6, 3.7608844899999997e-10, 9.0e-6, false);
}

std::shared_ptr<DatasetConfig> DatasetConfig::tum_vi() { // config/tum-vi.yaml:13-43
    const double K4[4] = {190.97847715128717, 190.9733070521226, 254.93170605935475, 256.8974428996504};
    const double q[4] = {-0.013272, -0.694726, 0.719112, 0.007648};
    const double p[3] = {0.04536566, -0.071996, -0.04478181};
    return make(K4, q, p, 2.56e-08, 7.84e-6, 4.84e-10, 7.396e-07, false);
}

} // namespace pvio
