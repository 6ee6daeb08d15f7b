// pnp_solve.cpp -- solve_pnp of pnp_problem.h: the flat pose-only problem (PnpProblem) evaluated with the factor code the kernels
// use (pvio_amd/csrc/pv_factors.h compiles for the host) and minimized by the dense trust-region loop of dense_minimizer.h.
// Residual / Jacobian conventions: pvio/src/pvio/estimation/ceres/reprojection_error_cost.h:128-203 (pose-only factors, CauchyLoss(1)),
// preintegration_error_cost.h:167-206 (the pre-integration prior against the last frame).  No reference types in this file: the
// reference's entry point visual_inertial_pnp (pnp.cpp) flattens Map / Frame into a PnpProblem and calls solve_pnp.
#include "pnp_problem.h"

#include <cmath>
#include <cstring>
#include <limits>

#include "../csrc/pv_factors.h"

namespace pvio {
namespace {

using namespace pv;

struct Problem {
    const PnpProblem &pb;
    std::vector<double> anchor_rec; // 28 doubles per factor (the anchors do not move)
    explicit Problem(const PnpProblem &p) : pb(p), anchor_rec(p.factors.size() * (size_t)kFrameRec) {
        for (size_t k = 0; k < p.factors.size(); ++k)
            frame_record(&anchor_rec[k * kFrameRec], p.factors[k].anchor_state, p.factors[k].anchor_cam, p.sqrt_inv_cov); // W of the anchor is not used
    }
    int ambient() const { return pb.use_inertial ? 16 : 7; }
    int tangent() const { return pb.use_inertial ? 15 : 6; }

    void plus(const double *x, const double *d, double *out) const {
        double full[16];
        std::memcpy(full, x, sizeof(double) * (size_t)ambient());
        double y[7];
        pose_plus(y, full, d, d + 3);
        std::memcpy(out, y, 7 * sizeof(double));
        if (pb.use_inertial)
            for (int k = 0; k < 9; ++k) out[7 + k] = x[7 + k] + d[6 + k];
    }

    bool evaluate(const double *x, double &cost, std::vector<double> *r, std::vector<double> *J) const {
        const int n = tangent();
        const size_t rows = 2 * (pb.factors.size() + pb.point_factors.size()) + (pb.use_inertial ? 15 : 0);
        if (r) r->assign(rows, 0.0);
        if (J) J->assign(rows * (size_t)n, 0.0);
        double state[16] = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::memcpy(state, x, sizeof(double) * (size_t)ambient());
        double Ft[kFrameRec];
        frame_record(Ft, state, pb.cam, pb.sqrt_inv_cov);
        cost = 0;
        size_t row = 0;
        if (pb.use_inertial) { // residual block order of pnp.cpp: the prior first, then the keypoints in index order
            double raw[15], G[450];
            preint_raw(pb.last_state, state, pb.last_state + 10, pb.delta, pb.jac, pb.last_imu, pb.imu, raw, G);
            for (int a = 0; a < 15; ++a) {
                double s = 0;
                for (int k = 0; k < 15; ++k) s += pb.sqrt_inv_cov_imu[a * 15 + k] * raw[k];
                if (!std::isfinite(s)) return false;
                cost += 0.5 * s * s;
                if (r) (*r)[row + a] = s;
                if (J)
                    for (int c = 0; c < 15; ++c) {
                        double t = 0;
                        for (int k = 0; k < 15; ++k) t += pb.sqrt_inv_cov_imu[a * 15 + k] * G[k * 30 + 15 + c];
                        (*J)[(row + a) * n + c] = t;
                    }
            }
            row += 15;
        }
        auto robust = [&](const double res[2], const double Jt[12]) { // CauchyLoss(1): rho = log(1 + s), r, J *= sqrt(rho')
            const double s = res[0] * res[0] + res[1] * res[1];
            if (!std::isfinite(s)) return false;
            const double w = std::sqrt(1.0 / (1.0 + s));
            cost += 0.5 * std::log(1.0 + s);
            if (r) (*r)[row] = w * res[0], (*r)[row + 1] = w * res[1];
            if (J)
                for (int c = 0; c < 6; ++c) (*J)[row * n + c] = w * Jt[c], (*J)[(row + 1) * n + c] = w * Jt[6 + c];
            row += 2;
            return true;
        };
        for (size_t k = 0; k < pb.factors.size(); ++k) {
            const PnpFactor &f = pb.factors[k];
            double res[2], Jt[12], Jr[12], Jd[2];
            reproj_eval<true>(Ft, &anchor_rec[k * kFrameRec], f.inv_depth, f.z_ref[0], f.z_ref[1], f.z_tgt[0], f.z_tgt[1], res, Jt, Jr, Jd);
            if (!robust(res, Jt)) return false;
        }
        for (const PnpPointFactor &f : pb.point_factors) {
            // y_c = Rb^T (x - p_b) ; y = Rc^T (y_c - p_c) ; r = W (pi(y) - z)          reprojection_error_cost.h:170-180
            double d[3], yc[3], e[3], y[3];
            v3_sub(d, f.point, Ft + 9);
            m3_tvec(yc, Ft, d);
            v3_sub(e, yc, Ft + 21);
            m3_tvec(y, Ft + 12, e);
            const double iz = 1.0 / y[2], u = y[0] * iz - f.z_tgt[0], v = y[1] * iz - f.z_tgt[1];
            const double *W = Ft + 24;
            const double res[2] = {W[0] * u + W[1] * v, W[2] * u + W[3] * v};
            const double xz = -y[0] * iz * iz, yz = -y[1] * iz * iz;
            const double P[6] = {W[0] * iz, W[1] * iz, W[0] * xz + W[1] * yz, W[2] * iz, W[3] * iz, W[2] * xz + W[3] * yz};
            double Jt[12];
            for (int i = 0; i < 2; ++i) {
                double A[3], B[3];
                for (int j = 0; j < 3; ++j) A[j] = P[3 * i] * Ft[12 + 3 * j] + P[3 * i + 1] * Ft[12 + 3 * j + 1] + P[3 * i + 2] * Ft[12 + 3 * j + 2]; // Jpi Rc^T
                for (int j = 0; j < 3; ++j) B[j] = A[0] * Ft[3 * j] + A[1] * Ft[3 * j + 1] + A[2] * Ft[3 * j + 2];                                  // .. Rb^T
                Jt[6 * i + 0] = A[1] * yc[2] - A[2] * yc[1], Jt[6 * i + 1] = A[2] * yc[0] - A[0] * yc[2], Jt[6 * i + 2] = A[0] * yc[1] - A[1] * yc[0]; // A hat(y_c)  :186
                Jt[6 * i + 3] = -B[0], Jt[6 * i + 4] = -B[1], Jt[6 * i + 5] = -B[2];                                                                  // :191
            }
            if (!robust(res, Jt)) return false;
        }
        return std::isfinite(cost);
    }
};

} // namespace

dense::Summary solve_pnp(const PnpProblem &pb, double state16[16], int max_iterations) {
    Problem P(pb);
    double x[16];
    std::memcpy(x, state16, sizeof x);
    const dense::Summary s = dense::minimize(P, x, max_iterations);
    std::memcpy(state16, x, sizeof(double) * (size_t)P.ambient());
    return s;
}

} // namespace pvio
