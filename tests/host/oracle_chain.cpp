// oracle_chain.cpp -- TEST INFRASTRUCTURE: the ORACLE CHAIN.
//
// libpvio_chain_oracle.so is the headless driver (standin/headless.*, standin/pvio_min.*) and the data-structure glue of the adapter
// (bundle_adjustor.cpp: Map -> flat problem; pnp.cpp: Map / Frame -> PnpProblem; feature_tracker.cpp: the per-frame order of
// operations) linked against THIS file instead of libpvio_hip.so and the arithmetic host sources.  Every piece that computes is the
// CPU oracle's (liboracle.so), none of it is the product's:
//
//   product (libpvio_chain_hip.so)                         oracle chain (this file)
//   pvio_hip_image_create   k_clahe, k_pyr, k_scharr       oracle_clahe, oracle_pyr_down, oracle_scharr          (oracle_klt.cpp)
//   pvio_hip_klt_track      k_klt                          oracle_klt_track                                       (oracle_klt.cpp)
//   pvio_hip_image_detect   k_harris, k_select ...         oracle_harris_response, oracle_good_features           (oracle_gftt.cpp)
//   find_fundamental_ransac host/fundamental_ransac.cpp    oracle_find_fundamental_ransac                         (oracle_ransac.cpp)
//   PoissonDisk2, select_tracked, predict_keypoints        oracle_poisson_insert, oracle_select_tracked,
//                           host/feature_front.cpp         oracle_predict_keypoints                               (oracle_front.cpp)
//   solve_pnp               host/pnp_solve.cpp             oracle_pnp_flat                                        (oracle_pnp.cpp)
//   pvio_preintegrate       csrc (host FP64)               oracle_preintegrate                                    (oracle_ba.cpp)
//   pvio_hip_ba_solve       k_linearize .. k_backsub       oracle_ba_solve (incl. the depth gate / quality pass)  (oracle_ba.cpp, oracle_post.cpp)
//   pvio_hip_ba_marginalize k_marg_*                       oracle_ba_marginalize                                  (oracle_ba.cpp)
//   pvio_hip_ba_reprojection_error                         oracle_ba_reprojection_error
//
// Not part of any product path: nothing under pvio_amd/ links or loads this.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pvio_hip.h"
#include "../../pvio_amd/host/feature_front.h" // declarations of predict_keypoints / select_tracked (the definitions are below)
#include "../../pvio_amd/host/fundamental_ransac.h"
#include "../../pvio_amd/host/pnp_problem.h"
#include "host_seam.h"

extern "C" {
// liboracle.so
int32_t oracle_ba_solve(const pvio_ba_problem *, pvio_ba_state *, pvio_ba_summary *);
int32_t oracle_ba_marginalize(const pvio_ba_problem *, const pvio_ba_state *, int32_t, pvio_ba_prior *);
int32_t oracle_ba_reprojection_error(const pvio_ba_problem *, const pvio_ba_state *, double *);
int32_t oracle_preintegrate(int32_t, const double *, const double *, const double *, double, const double *, const double *, const pvio_imu_noise *, double *,
                            double *, double *, double *);
void oracle_clahe(const uint8_t *, int, int, int, double, int, int, uint8_t *, int);
void oracle_pyr_down(const uint8_t *, int, int, uint8_t *);
void oracle_scharr(const uint8_t *, int, int, int16_t *);
int oracle_pyramid_sizes(int, int, int *, int *);
void oracle_klt_track(int, const int *, const int *, const uint8_t *const *, const int16_t *const *, const uint8_t *const *, int, const float *, float *, uint8_t *);
void oracle_harris_response(const uint8_t *, int, int, float *);
int oracle_good_features(const float *, int, int, int, double, double, float *, float *);
int32_t oracle_find_fundamental_ransac_defined(int32_t, const float *, const float *, double, double, int32_t, uint8_t *, double *);
void oracle_poisson_insert(double, int, const double *, int, const double *, uint8_t *);
void oracle_select_tracked(int, const double *, const uint64_t *, double, uint8_t *);
void oracle_predict_keypoints(const double *, const double *, const double *, const double *, const double *, const double *, int, const double *, double *);
int32_t oracle_pnp_flat(const double *, const double *, const double *, int32_t, const double *, const double *, const double *, const double *, const double *, int32_t,
                        const double *, const double *, int32_t, const double *, const double *, const double *, const double *, const double *, int32_t, double *,
                        int32_t *, int32_t *, double *);
}

// ---- the C ABI, oracle behind it ------------------------------------------------------------------------------------------------
struct pvio_hip_ctx {
    std::string err;
};
extern "C" {
int32_t pvio_hip_abi_version(void) { return PVIO_HIP_ABI_VERSION; }
int32_t pvio_hip_create(const pvio_hip_opts *, pvio_hip_ctx **out) {
    *out = new pvio_hip_ctx();
    return PVIO_OK;
}
void pvio_hip_destroy(pvio_hip_ctx *ctx) { delete ctx; }
const char *pvio_hip_last_error(const pvio_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }
int32_t pvio_hip_ba_solve(pvio_hip_ctx *, const pvio_ba_problem *pb, pvio_ba_state *st, pvio_ba_summary *sum) { return oracle_ba_solve(pb, st, sum); }
int32_t pvio_hip_ba_marginalize(pvio_hip_ctx *, const pvio_ba_problem *pb, const pvio_ba_state *st, int32_t victim, pvio_ba_prior *out) {
    return oracle_ba_marginalize(pb, st, victim, out);
}
int32_t pvio_hip_ba_reprojection_error(pvio_hip_ctx *, const pvio_ba_problem *pb, const pvio_ba_state *st, double *out) { return oracle_ba_reprojection_error(pb, st, out); }
int32_t pvio_preintegrate(int32_t n, const double *t, const double *w, const double *a, double t_end, const double bg[3], const double ba[3],
                          const pvio_imu_noise *noise, double delta[11], double cov[225], double U[225], double jac[45]) {
    return oracle_preintegrate(n, t, w, a, t_end, bg, ba, noise, delta, cov, U, jac);
}
}

namespace pvio {

// ---- host arithmetic, oracle behind the product's declarations ------------------------------------------------------------------
int find_fundamental_ransac(int n, const float *p, const float *q, double threshold, double confidence, std::vector<uint8_t> &mask, double F_out[9], int max_iterations) {
    mask.assign((size_t)std::max(n, 0), 0);
    double F[9];
    return oracle_find_fundamental_ransac_defined(n, p, q, threshold, confidence, max_iterations, mask.data(), F_out ? F_out : F);
}

void predict_keypoints(const Frame &curr, const Frame &next, std::vector<vector<2>> &next_pixels) {
    auto q4 = [](const quaternion &q, double o[4]) { o[0] = q.x(), o[1] = q.y(), o[2] = q.z(), o[3] = q.w(); };
    double qci[4], qii[4], dq[4], qij[4], qcj[4];
    q4(curr.camera.q_cs, qci), q4(curr.imu.q_cs, qii), q4(next.preintegration.delta.q, dq), q4(next.imu.q_cs, qij), q4(next.camera.q_cs, qcj);
    const double K4[4] = {next.K(0, 0), next.K(1, 1), next.K(0, 2), next.K(1, 2)};
    const size_t n = curr.keypoint_num();
    std::vector<double> kp(2 * n), out(2 * n);
    for (size_t i = 0; i < n; ++i) kp[2 * i] = curr.get_keypoint(i)[0], kp[2 * i + 1] = curr.get_keypoint(i)[1];
    oracle_predict_keypoints(qci, qii, dq, qij, qcj, K4, (int)n, kp.data(), out.data());
    next_pixels.resize(n);
    for (size_t i = 0; i < n; ++i) next_pixels[i][0] = out[2 * i], next_pixels[i][1] = out[2 * i + 1];
}

void select_tracked(const std::vector<vector<2>> &next_pixels, const std::vector<size_t> &track_length, double min_distance, std::vector<char> &status) {
    const size_t n = status.size();
    std::vector<double> xy(2 * n);
    std::vector<uint64_t> len(n);
    std::vector<uint8_t> st(n);
    for (size_t i = 0; i < n; ++i) xy[2 * i] = next_pixels[i][0], xy[2 * i + 1] = next_pixels[i][1], len[i] = track_length[i], st[i] = (uint8_t)status[i];
    oracle_select_tracked((int)n, xy.data(), len.data(), min_distance, st.data());
    for (size_t i = 0; i < n; ++i) status[i] = (char)st[i];
}

dense::Summary solve_pnp(const PnpProblem &pb, double state16[16], int max_iterations) {
    const size_t n = pb.factors.size(), m = pb.point_factors.size();
    std::vector<double> A(16 * n), Cm(7 * n), zr(2 * n), zt(2 * n), rho(n), pts(3 * m), zp(2 * m);
    for (size_t k = 0; k < n; ++k) {
        const PnpFactor &f = pb.factors[k];
        std::memcpy(&A[16 * k], f.anchor_state, 128), std::memcpy(&Cm[7 * k], f.anchor_cam, 56);
        zr[2 * k] = f.z_ref[0], zr[2 * k + 1] = f.z_ref[1], zt[2 * k] = f.z_tgt[0], zt[2 * k + 1] = f.z_tgt[1], rho[k] = f.inv_depth;
    }
    for (size_t k = 0; k < m; ++k) {
        std::memcpy(&pts[3 * k], pb.point_factors[k].point, 24);
        zp[2 * k] = pb.point_factors[k].z_tgt[0], zp[2 * k + 1] = pb.point_factors[k].z_tgt[1];
    }
    int32_t it = 0, term = 0;
    double costs[2] = {0, 0};
    oracle_pnp_flat(pb.cam, pb.imu, pb.sqrt_inv_cov, (int32_t)n, A.data(), Cm.data(), zr.data(), zt.data(), rho.data(), (int32_t)m, pts.data(), zp.data(),
                    pb.use_inertial ? 1 : 0, pb.last_state, pb.last_imu, pb.delta, pb.sqrt_inv_cov_imu, pb.jac, max_iterations, state16, &it, &term, costs);
    dense::Summary s;
    s.iterations = it, s.termination = term, s.initial_cost = costs[0], s.final_cost = costs[1];
    return s;
}

} // namespace pvio

#include "oracle_image.h" // class OracleImage : pvio::Image (shared with oracle/ref/seq_capi.cpp)

std::shared_ptr<pvio::Image> oracle_chain_make_image(const uint8_t *pixels, int w, int h, double t) { return std::make_shared<pvio::OracleImage>(pixels, w, h, t); }
