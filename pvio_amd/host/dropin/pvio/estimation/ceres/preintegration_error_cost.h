// Drop-in replacement for pvio/src/pvio/estimation/ceres/preintegration_error_cost.h (same path and class name; the
// constructor estimation/factor.cpp:39-43 calls).  The reference object is a ceres::SizedCostFunction<15, 4,3,3,3,3,
// 4,3,3,3,3> (:30-160); here it records the frame pair -- the residual / Jacobians of :40-160 are evaluated by the IMU
// role of k_linearize (pvio_amd/csrc/pv_factors.h preint_raw), the prior form of :167-206 by host/pnp.cpp.
#ifndef PVIO_PREINTEGRATION_ERROR_COST_H
#define PVIO_PREINTEGRATION_ERROR_COST_H

#include "host_types.h"

namespace pvio {

class PreIntegrationErrorCost : public Factor::FactorCostFunction {
  public:
    PreIntegrationErrorCost(Frame *frame_i, Frame *frame_j) : frame_i(frame_i), frame_j(frame_j) {}
    void update() override {}
    Frame *const frame_i;
    Frame *const frame_j;
};

} // namespace pvio

#endif // PVIO_PREINTEGRATION_ERROR_COST_H
