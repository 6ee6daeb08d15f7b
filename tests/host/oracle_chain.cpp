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
int32_t oracle_find_fundamental_ransac(int32_t, const float *, const float *, double, double, int32_t, uint8_t *, double *);
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
    return oracle_find_fundamental_ransac(n, p, q, threshold, confidence, max_iterations, mask.data(), F_out ? F_out : F);
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

// ---- pvio::Image, oracle behind it (opencv_image.cpp:54-160) -------------------------------------------------------------------
class OracleImage : public Image {
  public:
    OracleImage(const uint8_t *pixels, int width, int height, double timestamp) : w_(width), h_(height), raw_(pixels, pixels + (size_t)width * height) { t = timestamp; }
    size_t width() const override { return (size_t)w_; }
    size_t height() const override { return (size_t)h_; }
    size_t level_num() const override { return 3; }
    double evaluate(const vector<2> &, int = 0) const override { throw std::logic_error("OracleImage::evaluate: not used by the chain"); }
    double evaluate(const vector<2> &, vector<2> &, int = 0) const override { throw std::logic_error("OracleImage::evaluate: not used by the chain"); }

    void preprocess() override { // :138-145: CLAHE(6.0, 8 x 8) in place, then buildOpticalFlowPyramid(.., maxLevel 3, withDerivatives)
        std::vector<uint8_t> eq((size_t)w_ * h_);
        oracle_clahe(raw_.data(), w_, h_, w_, 6.0, 8, 8, eq.data(), w_);
        int ws[4], hs[4];
        const int n = oracle_pyramid_sizes(w_, h_, ws, hs);
        ws_.assign(ws, ws + n), hs_.assign(hs, hs + n);
        img_.assign((size_t)n, {}), drv_.assign((size_t)n, {});
        img_[0] = std::move(eq);
        for (int l = 0; l < n; ++l) {
            if (l > 0) {
                img_[(size_t)l].resize((size_t)ws[l] * hs[l]);
                oracle_pyr_down(img_[(size_t)l - 1].data(), ws[l - 1], hs[l - 1], img_[(size_t)l].data());
            }
            drv_[(size_t)l].resize((size_t)2 * ws[l] * hs[l]);
            oracle_scharr(img_[(size_t)l].data(), ws[l], hs[l], drv_[(size_t)l].data());
        }
    }

    void detect_keypoints(std::vector<vector<2>> &keypoints, size_t, double keypoint_distance) const override { // :54-86
        if (img_.empty()) throw std::runtime_error("OracleImage::detect_keypoints: preprocess() was not called");
        std::vector<float> resp((size_t)w_ * h_), xy(2000), r(1000);
        oracle_harris_response(img_[0].data(), w_, h_, resp.data());
        const int n = oracle_good_features(resp.data(), w_, h_, 1000, 1.0e-3, 20.0, xy.data(), r.data());
        if (n == 0) return;
        std::vector<int> order((size_t)n);
        for (int i = 0; i < n; ++i) order[(size_t)i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return r[(size_t)a] > r[(size_t)b]; });
        std::vector<double> preset(2 * keypoints.size()), cand(2 * (size_t)n);
        for (size_t i = 0; i < keypoints.size(); ++i) preset[2 * i] = keypoints[i][0], preset[2 * i + 1] = keypoints[i][1];
        for (int i = 0; i < n; ++i) cand[2 * (size_t)i] = xy[2 * (size_t)order[(size_t)i]], cand[2 * (size_t)i + 1] = xy[2 * (size_t)order[(size_t)i] + 1];
        std::vector<uint8_t> acc((size_t)n);
        oracle_poisson_insert(keypoint_distance, (int)keypoints.size(), preset.data(), n, cand.data(), acc.data());
        for (int i = 0; i < n; ++i) {
            if (!acc[(size_t)i]) continue;
            const double x = cand[2 * (size_t)i], y = cand[2 * (size_t)i + 1];
            if (x < 20 || y < 20 || x >= w_ - 20 || y >= h_ - 20) continue;
            vector<2> p;
            p[0] = x, p[1] = y;
            keypoints.push_back(p);
        }
    }

    void track_keypoints(const Image *next_image, const std::vector<vector<2>> &curr, std::vector<vector<2>> &next, std::vector<char> &status) const override { // :88-135
        const size_t n = curr.size();
        std::vector<float> p(2 * n), q(2 * n);
        for (size_t i = 0; i < n; ++i) p[2 * i] = (float)curr[i][0], p[2 * i + 1] = (float)curr[i][1];
        if (next.size() > 0) {
            for (size_t i = 0; i < n; ++i) q[2 * i] = (float)next[i][0], q[2 * i + 1] = (float)next[i][1];
        } else {
            next.resize(n);
            q = p;
        }
        status.resize(n, 0);
        const OracleImage *nx = dynamic_cast<const OracleImage *>(next_image);
        if (nx && n > 0) {
            if (img_.empty() || nx->img_.empty()) throw std::runtime_error("OracleImage::track_keypoints: preprocess() was not called");
            const size_t L = img_.size();
            std::vector<const uint8_t *> pi(L), ni(L);
            std::vector<const int16_t *> pd(L);
            for (size_t l = 0; l < L; ++l) pi[l] = img_[l].data(), pd[l] = drv_[l].data(), ni[l] = nx->img_[l].data();
            std::vector<uint8_t> st(n, 0);
            oracle_klt_track((int)L, ws_.data(), hs_.data(), pi.data(), pd.data(), ni.data(), (int)n, p.data(), q.data(), st.data());
            for (size_t i = 0; i < n; ++i) status[i] = (char)st[i];
        }
        std::vector<size_t> l;
        std::vector<float> pp, qq;
        for (size_t i = 0; i < n; ++i)
            if (status[i] != 0) l.push_back(i), pp.push_back(p[2 * i]), pp.push_back(p[2 * i + 1]), qq.push_back(q[2 * i]), qq.push_back(q[2 * i + 1]);
        if (l.size() >= 8) { // :113-129
            std::vector<uint8_t> mask(l.size(), 0);
            double F[9];
            oracle_find_fundamental_ransac((int32_t)l.size(), pp.data(), qq.data(), 1.0, 0.99, 1000, mask.data(), F);
            for (size_t i = 0; i < l.size(); ++i)
                if (mask[i] == 0) status[l[i]] = 0;
        }
        for (size_t i = 0; i < n; ++i)
            if (status[i]) next[i][0] = q[2 * i], next[i][1] = q[2 * i + 1];
    }

  private:
    int w_, h_;
    std::vector<uint8_t> raw_;
    std::vector<int> ws_, hs_;
    std::vector<std::vector<uint8_t>> img_;
    std::vector<std::vector<int16_t>> drv_;
};

} // namespace pvio

std::shared_ptr<pvio::Image> oracle_chain_make_image(const uint8_t *pixels, int w, int h, double t) { return std::make_shared<pvio::OracleImage>(pixels, w, h, t); }
