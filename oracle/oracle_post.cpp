// oracle_post.cpp -- CPU restatement of BundleAdjustorSolver::solve's post-solve passes (TEST INFRASTRUCTURE: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/).
//
// Follows, on flat arrays, pvio/src/pvio/estimation/bundle_adjustor.cpp:251-296 (the PVIO_ENABLE_PLANE_CONSTRAINT build,
// ON by default) with the helpers it calls:
//   Track::try_triangulate            map/track.cpp:61-81          camera matrices [R | -R p] of every observation
//   triangulate_point(_scored)        geometry/stereo.h:76-83,104-128   DLT rows u P2 - P0, v P2 - P1; V.col(3) of the SVD;
//                                                                   in front of every camera and depth / w < 100
//   PlaneExtractor::enough_baseline   core/plane_extractor.cpp:200-203 + Track::compute_baseline map/track.cpp:125-135
//   Track::get/set_landmark_point     map/track.cpp:137-147
//   Frame::get_pose                   map/frame.cpp:187-192
// Eigen's JacobiSVD is third-party and absent; the right singular vector of the smallest singular value is computed
// here as the eigenvector of A^T A (cyclic Jacobi, oracle_math.h sym_eig) -- a DIFFERENT algorithm from the one-sided
// Jacobi of the host stand-in (pvio_amd/host/pvio_min.cpp), so the comparison is not a self-comparison.  Parity unpinned
// (no reference fixtures exist for this path).
#include <cmath>
#include <cstdint>
#include <vector>

#include "../include/pvio_hip.h"
#include "oracle_math.h"

using namespace orc;

namespace {
struct Cam {
    Q q;
    V3 p;
};
Cam camera_pose(const pvio_ba_problem *pb, const double *fs, int f) { // Frame::get_pose(frame->camera)
    const Q qb = qload(fs + 16 * f), qe = qload(pb->cam_extrinsic + 7 * f);
    return Cam{qmul(qb, qe), vload(fs + 16 * f + 4) + qrot(qb, vload(pb->cam_extrinsic + 7 * f + 4))};
}
} // namespace

extern "C" int oracle_post_passes(const pvio_ba_problem *pb, const double *frame_state, int32_t n_tracks, const int32_t *trk_obs_ptr,
                                  const int32_t *trk_obs_frame, const double *trk_obs_z, const int64_t *trk_life, uint8_t *trk_valid,
                                  uint8_t *trk_plane, double *trk_inv_depth, double *trk_quality, int32_t n_planes,
                                  const double *plane_normal, const double *plane_distance, uint8_t *membership /* [n_planes][n_tracks] */) {
    const double *fs = frame_state;
    // ---- :251-275 plane-track re-validation -------------------------------------------------------------------------
    for (int t = 0; t < n_tracks; ++t) {
        if (!trk_plane[t]) continue;
        const int b = trk_obs_ptr[t], e = trk_obs_ptr[t + 1], K = e - b;
        if (K < 2) continue;
        if (!(trk_life[t] > 10)) continue;
        double baseline = 0; // body positions of consecutive observing frames
        for (int o = b; o + 1 < e; ++o) baseline += norm(vload(fs + 16 * trk_obs_frame[o] + 4) - vload(fs + 16 * trk_obs_frame[o + 1] + 4));
        if (!((baseline > 0.5) || (trk_inv_depth[t] < (1 / 0.2) && baseline * trk_inv_depth[t] > 0.5))) continue;
        // DLT
        std::vector<double> A((size_t)2 * K * 4), P((size_t)K * 12);
        for (int k = 0; k < K; ++k) {
            const Cam c = camera_pose(pb, fs, trk_obs_frame[b + k]);
            const M3 R = qmat(qconj(c.q));
            const V3 T = -(R * c.p);
            double *Pk = &P[(size_t)12 * k];
            for (int r = 0; r < 3; ++r) Pk[4 * r] = R.m[r][0], Pk[4 * r + 1] = R.m[r][1], Pk[4 * r + 2] = R.m[r][2], Pk[4 * r + 3] = T[r];
            const double *z = trk_obs_z + 2 * (size_t)(b + k);
            for (int c4 = 0; c4 < 4; ++c4) A[(size_t)(2 * k) * 4 + c4] = z[0] * Pk[8 + c4] - Pk[c4], A[(size_t)(2 * k + 1) * 4 + c4] = z[1] * Pk[8 + c4] - Pk[4 + c4];
        }
        double G[16] = {0}, w[4], V[16];
        for (int r = 0; r < 2 * K; ++r)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) G[4 * i + j] += A[(size_t)r * 4 + i] * A[(size_t)r * 4 + j];
        sym_eig(G, 4, w, V); // ascending eigenvalues, eigenvectors in columns
        int kmin = 0;
        for (int k = 1; k < 4; ++k)
            if (w[k] < w[kmin]) kmin = k;
        const double q[4] = {V[0 * 4 + kmin], V[1 * 4 + kmin], V[2 * 4 + kmin], V[3 * 4 + kmin]};
        bool ok = true;
        for (int k = 0; k < K; ++k) {
            const double *Pk = &P[(size_t)12 * k];
            const double z = Pk[8] * q[0] + Pk[9] * q[1] + Pk[10] * q[2] + Pk[11] * q[3];
            if (!(z * q[3] > 0)) ok = false;
            if (!(z / q[3] < 100)) ok = false;
        }
        if (!ok) continue;
        const V3 p = mk(q[0] / q[3], q[1] / q[3], q[2] / q[3]);
        bool held = false;
        for (int j = 0; j < n_planes; ++j) {
            uint8_t &in = membership[(size_t)j * n_tracks + t];
            if (!in) continue;
            if (std::abs(dot(vload(plane_normal + 3 * j), p) - plane_distance[j]) > 0.1) in = 0;
            else held = true;
        }
        if (!held) {
            trk_plane[t] = 0, trk_valid[t] = 1;
            const Cam c = camera_pose(pb, fs, trk_obs_frame[b]); // set_landmark_point: depth in the anchor camera
            trk_inv_depth[t] = 1.0 / qrot(qconj(c.q), p - c.p)[2];
        }
    }
    // ---- :277-296 depth gate + mean pixel reprojection error --------------------------------------------------------
    for (int t = 0; t < n_tracks; ++t) {
        if (!trk_valid[t] && !trk_plane[t]) continue;
        const int b = trk_obs_ptr[t], e = trk_obs_ptr[t + 1];
        const Cam ca = camera_pose(pb, fs, trk_obs_frame[b]);
        const double *za = trk_obs_z + 2 * (size_t)b;
        const V3 x = (1.0 / trk_inv_depth[t]) * qrot(ca.q, mk(za[0], za[1], 1.0)) + ca.p;
        double quality = 0, num = 0;
        for (int o = b; o < e; ++o) {
            const int f = trk_obs_frame[o];
            const Cam c = camera_pose(pb, fs, f);
            const V3 y = qrot(qconj(c.q), x - c.p);
            if (y[2] <= 1.0e-3 || y[2] > 50) {
                trk_valid[t] = 0, trk_plane[t] = 0;
                break;
            }
            const double *Kf = pb->intrinsics + 4 * f, *z = trk_obs_z + 2 * (size_t)o;
            const double du = (y[0] / y[2]) * Kf[0] + Kf[2] - (z[0] * Kf[0] + Kf[2]), dv = (y[1] / y[2]) * Kf[1] + Kf[3] - (z[1] * Kf[1] + Kf[3]);
            quality += std::sqrt(du * du + dv * dv), num += 1.0;
        }
        if (!trk_valid[t]) continue;
        trk_quality[t] = quality / std::max(num, 1.0);
    }
    return 0;
}
