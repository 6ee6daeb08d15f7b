// pnp_problem.h -- host `visual_inertial_pnp` without Ceres.
//
// Reference: pvio/src/pvio/estimation/pnp.cpp:32-100 -- the per-frame pose refinement that runs right before the sliding
// window bundle adjustment (sliding_window_tracker.cpp:79, initializer.cpp:186) and produces its initial guess: one free
// frame (pose, and velocity / biases when inertial), every VALID track of the frame that is also seen by the map's last
// frame as a pose-only reprojection factor with CauchyLoss(1) (reprojection_error_cost.h:128-157), plus the IMU
// pre-integration prior against the last frame (preintegration_error_cost.h:167-206).  One pose and a few hundred residual
// rows: not worth a kernel; it runs on the host with the same factor code the kernels use
// (pvio_amd/csrc/pv_factors.h compiles for the host) and the dense trust-region loop of dense_minimizer.h.
#pragma once
#include "host_namespace.h"
#include <vector>

#include "dense_minimizer.h"

namespace pvio {

struct PnpFactor { // PoseOnlyReprojectionErrorCost: landmark = anchor keypoint at inverse depth rho in the (fixed) anchor frame
    double anchor_state[16]; // q(xyzw) p v bg ba of the anchor frame's body
    double anchor_cam[7];    // its camera extrinsics q_cs, p_cs
    double z_ref[2], z_tgt[2], inv_depth;
};
struct PnpPointFactor { // PoseOnlyReprojectionXYZErrorCost (reprojection_error_cost.h:159-203): a fixed world point
    double point[3], z_tgt[2];
};
struct PnpProblem {
    double cam[7], imu[7], sqrt_inv_cov[4]; // of the frame being solved (camera / IMU extrinsics, 2x2 row-major)
    std::vector<PnpFactor> factors;
    std::vector<PnpPointFactor> point_factors;
    bool use_inertial = false;
    double last_state[16], last_imu[7];                // the map's last frame (fixed) ...
    double delta[11], sqrt_inv_cov_imu[225], jac[45];  // ... and the pre-integration between it and this frame
};

// minimizes over state16 in place (q, p and -- when inertial -- v, bg, ba)
dense::Summary solve_pnp(const PnpProblem &pb, double state16[16], int max_iterations);

// The reference's entry point `void visual_inertial_pnp(Map *, Frame *, Config *, bool use_inertial = true)`
// (estimation/pnp.h:26) is declared by host_seam.h (the reference's own header inside the PVIO tree) and defined in
// pnp.cpp: it flattens `frame` against `map` into a PnpProblem -- best-plane search of pnp.cpp:61-88 included -- and writes
// the result back into frame->pose / motion.

} // namespace pvio
