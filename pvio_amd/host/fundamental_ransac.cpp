// fundamental_ransac.cpp -- see fundamental_ransac.h.  The host-only form of the RANSAC: sequential hypothesise-and-verify, each
// hypothesis scored until it can no longer beat the best.  The arithmetic (7-point solver, error measure, sampler, stopping rule) is
// pvio_amd/csrc/pv_fundamental.h, shared with the batched device form (pvio_hip_fundamental_ransac), which HipImage uses; this one is
// what the tests hold it -- and the CPU oracle -- against.
#include "fundamental_ransac.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

#include "../csrc/pv_fundamental.h"

namespace pvio {
namespace {

// `need`: the hypothesis only matters if it ends with MORE than `need` inliers (the best count so far); once the points that
// are left cannot lift it above that the scoring stops (the value returned is then just some count <= need, the mask is
// not used).  Most hypotheses of a run are poor: they are abandoned after n - need misses instead of after n points.
int count_inliers(int n, const float *p, const float *q, const double *F, double thr2, std::vector<uint8_t> &mask, int need) {
    int good = 0;
    for (int i = 0; i < n; ++i) {
        if (good + (n - i) <= need) return good;
        const float err = pvfm::fm_error(F, p[2 * i], p[2 * i + 1], q[2 * i], q[2 * i + 1]);
        const uint8_t in = err <= thr2 ? 1 : 0;
        mask[(size_t)i] = in, good += in;
    }
    return good;
}

} // namespace

int fundamental_7point(const float p[14], const float q[14], double F[27]) { return pvfm::fm_seven_point(p, q, F); }

int find_fundamental_ransac(int n, const float *p, const float *q, double threshold, double confidence, std::vector<uint8_t> &mask, double F_out[9], int max_iterations) {
    constexpr int kModel = 7;
    mask.assign((size_t)std::max(n, 0), 0);
    if (n < kModel) return 0;
    pvfm::FmRng rng((uint64_t)-1);
    const double thr2 = threshold * threshold;
    std::vector<uint8_t> cur((size_t)n), best((size_t)n, 0);
    int niters = max_iterations, max_good = 0;
    double bestF[9] = {0};
    float sp[14], sq[14];
    if (n == kModel) niters = 1;
    for (int iter = 0; iter < niters; ++iter) {
        if (n > kModel) {
            if (!pvfm::fm_draw_sample(rng, n, p, q, sp, sq)) {
                if (iter == 0) return 0;
                break;
            }
        } else {
            std::copy(p, p + 14, sp), std::copy(q, q + 14, sq);
        }
        double F[27];
        const int nm = pvfm::fm_seven_point(sp, sq, F);
        for (int m = 0; m < nm; ++m) {
            const int good = count_inliers(n, p, q, F + 9 * m, thr2, cur, std::max(max_good, kModel - 1));
            if (good > std::max(max_good, kModel - 1)) {
                std::swap(cur, best);
                std::copy(F + 9 * m, F + 9 * m + 9, bestF);
                max_good = good;
                niters = pvfm::fm_update_iterations(confidence, (double)(n - good) / n, kModel, niters);
            }
        }
    }
    if (max_good > 0) {
        mask = best;
        if (F_out) std::copy(bestF, bestF + 9, F_out);
    }
    return max_good;
}

} // namespace pvio
