// Drop-in replacement for pvio/src/pvio/estimation/ceres/marginalization_error_cost.h (same path, same class name, same
// constructor: the reference's estimation/factor.cpp:27-31 compiles against it unedited).
//
// The reference class is a ceres::CostFunction whose Evaluate() the solver calls (:53-94).  With the solve on the GPU the
// object is only the HOLDER of the prior between marginalize_frame (which creates it, bundle_adjustor.cpp:583-598) and
// the next solve / marginalize_frame (which read it, :126-139, :369-413): sqrt-information matrix and vector, the related
// frames, and the linearization states captured at construction (:36-47).  The residual / Jacobian arithmetic of :53-94
// lives in the HIP kernels (pvio_amd/csrc/ba_kernels.hip, prior role) -- nothing here evaluates anything.
//
// Additions over the reference's public surface (its members are private and have no accessors besides
// related_frames()): sqrt_information(), information_vector(), linearization_pose(i), linearization_motion(i).
#ifndef PVIO_MARGINALIZATION_ERROR_COST_H
#define PVIO_MARGINALIZATION_ERROR_COST_H

#include "host_types.h"

namespace pvio {

class MarginalizationErrorCost : public Factor::FactorCostFunction {
  public:
    MarginalizationErrorCost(const matrix<> &sqrt_inv_cov, const vector<> &infovec, std::vector<Frame *> &&frames) :
        sqrt_inv_cov(sqrt_inv_cov), infovec(infovec), frames(std::move(frames)) {
        pose_0.reserve(this->frames.size()), motion_0.reserve(this->frames.size());
        for (Frame *f : this->frames) pose_0.push_back(f->pose), motion_0.push_back(f->motion); // the linearization point
    }

    void update() override {}

    const std::vector<Frame *> &related_frames() const { return frames; }
    const matrix<> &sqrt_information() const { return sqrt_inv_cov; }
    const vector<> &information_vector() const { return infovec; }
    const PoseState &linearization_pose(size_t i) const { return pose_0[i]; }
    const MotionState &linearization_motion(size_t i) const { return motion_0[i]; }

  private:
    std::vector<PoseState> pose_0;
    std::vector<MotionState> motion_0;
    matrix<> sqrt_inv_cov;
    vector<> infovec;
    std::vector<Frame *> frames;
};

} // namespace pvio

#endif // PVIO_MARGINALIZATION_ERROR_COST_H
