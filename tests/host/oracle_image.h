// oracle_image.h -- TEST INFRASTRUCTURE: pvio::Image with the CPU oracle (liboracle.so) behind every method -- CLAHE, pyramid, Scharr,
// LK, Harris / goodFeaturesToTrack, F-RANSAC (oracle/oracle_klt.cpp, oracle_gftt.cpp, oracle_ransac.cpp, oracle_front.cpp) -- following
// pvio-extra/src/pvio/extra/opencv_image.cpp:54-160 call by call.  Written against pvio::Image / vector<2> only, so it compiles against the
// look-alike declarations (tests/host/standin/pvio_min.h: the oracle chain of tests/host/oracle_chain.cpp) and against the reference's real
// pvio.h (oracle/ref/seq_capi.cpp: the reference's own pvio::PVIO driven over a sequence).  Include after the header that declares pvio::Image.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <stdexcept>
#include <vector>

extern "C" {
void oracle_clahe(const uint8_t *, int, int, int, double, int, int, uint8_t *, int);
void oracle_pyr_down(const uint8_t *, int, int, uint8_t *);
void oracle_scharr(const uint8_t *, int, int, int16_t *);
int oracle_pyramid_sizes(int, int, int *, int *);
void oracle_klt_track(int, const int *, const int *, const uint8_t *const *, const int16_t *const *, const uint8_t *const *, int, const float *, float *, uint8_t *);
void oracle_harris_response(const uint8_t *, int, int, float *);
int oracle_good_features(const float *, int, int, int, double, double, float *, float *);
int32_t oracle_find_fundamental_ransac_defined(int32_t, const float *, const float *, double, double, int32_t, uint8_t *, double *);
void oracle_poisson_insert(double, int, const double *, int, const double *, uint8_t *);
}

namespace pvio {

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
            const std::vector<float> q_init = q;
            oracle_klt_track((int)L, ws_.data(), hs_.data(), pi.data(), pd.data(), ni.data(), (int)n, p.data(), q.data(), st.data());
            if (const char *dump = std::getenv("PVIO_KLT_DUMP")) { // diagnostics (see HipImage::track_keypoints): <dump>_oracle.bin
                if (FILE *f = std::fopen((std::string(dump) + "_oracle.bin").c_str(), "ab")) {
                    const int32_t nn = (int32_t)n;
                    std::fwrite(&nn, 4, 1, f), std::fwrite(p.data(), 4, 2 * n, f), std::fwrite(q_init.data(), 4, 2 * n, f), std::fwrite(q.data(), 4, 2 * n, f), std::fwrite(st.data(), 1, n, f);
                    std::fclose(f);
                }
            }
            for (size_t i = 0; i < n; ++i) status[i] = (char)st[i];
        }
        std::vector<size_t> l;
        std::vector<float> pp, qq;
        for (size_t i = 0; i < n; ++i)
            if (status[i] != 0) l.push_back(i), pp.push_back(p[2 * i]), pp.push_back(p[2 * i + 1]), qq.push_back(q[2 * i]), qq.push_back(q[2 * i + 1]);
        if (l.size() >= 8) { // :113-129
            std::vector<uint8_t> mask(l.size(), 0);
            double F[9];
            oracle_find_fundamental_ransac_defined((int32_t)l.size(), pp.data(), qq.data(), 1.0, 0.99, 1000, mask.data(), F);
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
