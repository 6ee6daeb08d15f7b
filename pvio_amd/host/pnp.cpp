// pnp.cpp -- see pnp_problem.h.  Drop-in for pvio/src/pvio/estimation/pnp.cpp (compiled against the reference's headers
// inside the PVIO tree, host_seam.h): flattens `frame` against `map` into a PnpProblem and hands it to solve_pnp (pnp_solve.cpp).
#include "host_seam.h"
#include "pnp_problem.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace pvio {
namespace {

void put_state(double s[16], const Frame *f) {
    for (int k = 0; k < 4; ++k) s[k] = f->pose.q.coeffs()[k];
    for (int k = 0; k < 3; ++k) s[4 + k] = f->pose.p[k], s[7 + k] = f->motion.v[k], s[10 + k] = f->motion.bg[k], s[13 + k] = f->motion.ba[k];
}
void put_ext(double e[7], const ExtrinsicParams &x) {
    for (int k = 0; k < 4; ++k) e[k] = x.q_cs.coeffs()[k];
    for (int k = 0; k < 3; ++k) e[4 + k] = x.p_cs[k];
}
void put_m3(double *dst, const matrix<3> &m) { // -> row-major
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) dst[3 * r + c] = m(r, c);
}

} // namespace

void visual_inertial_pnp(Map *map, Frame *frame, Config *config, bool use_inertial) {
    Frame *last = map->last_frame();
    PnpProblem pb;
    put_ext(pb.cam, frame->camera), put_ext(pb.imu, frame->imu);
    pb.sqrt_inv_cov[0] = frame->sqrt_inv_cov(0, 0), pb.sqrt_inv_cov[1] = frame->sqrt_inv_cov(0, 1);
    pb.sqrt_inv_cov[2] = frame->sqrt_inv_cov(1, 0), pb.sqrt_inv_cov[3] = frame->sqrt_inv_cov(1, 1);
    pb.use_inertial = use_inertial;
    put_state(pb.last_state, last), put_ext(pb.last_imu, last->imu);
    if (use_inertial) {
        const PreIntegrator::Delta &d = frame->preintegration.delta;
        pb.delta[0] = d.t;
        for (int k = 0; k < 4; ++k) pb.delta[1 + k] = d.q.coeffs()[k];
        for (int k = 0; k < 3; ++k) pb.delta[5 + k] = d.p[k], pb.delta[8 + k] = d.v[k];
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) pb.sqrt_inv_cov_imu[15 * r + c] = d.sqrt_inv_cov(r, c);
        const PreIntegrator::Jacobian &jc = frame->preintegration.jacobian;
        put_m3(pb.jac, jc.dq_dbg), put_m3(pb.jac + 9, jc.dp_dbg), put_m3(pb.jac + 18, jc.dp_dba), put_m3(pb.jac + 27, jc.dv_dbg), put_m3(pb.jac + 36, jc.dv_dba);
    }
    // Residual blocks in the reference's order (pnp.cpp:55-93): keypoint index order, plane-point and anchored factors
    // interleaved.  PnpProblem keeps the two kinds in separate lists, i.e. the cost sums them in another order -- a
    // rounding-level difference, like the landmark-major order of the window solve.
    for (size_t i = 0; i < frame->keypoint_num(); ++i) {
        Track *track = frame->get_track(i);
        if (!track) continue;
        if (!track->has_keypoint(last)) continue;
        if (!track->flag(TrackFlag::TF_VALID)) continue;
        const vector<2> &zt = frame->get_keypoint(i);
        if (track->flag(TrackFlag::TF_PLANE)) { // PVIO_ENABLE_PLANE_CONSTRAINT (ON by default, CMakeLists.txt:10): pnp.cpp:61-88
            // "find best plane via reprojection error": the reference never updates max_rpe, so every plane that is not
            // parallel to the anchor's viewing ray and whose reprojection error is below DBL_MAX replaces the previous
            // choice -- the LAST such plane of the map wins.  Reproduced as is.
            bool have = false;
            vector<3> best_plane_point;
            const double max_rpe = std::numeric_limits<double>::max();
            const auto ref = track->first_keypoint();
            const PoseState cam = ref.first->get_pose(ref.first->camera);
            const vector<3> direction = cam.q * ref.first->get_keypoint(ref.second).homogeneous();
            for (size_t j = 0; j < map->plane_num(); ++j) {
                Plane *plane = map->get_plane(j);
                if (plane->is_parallel(direction)) continue;
                const vector<3> plane_point = plane->cast_to_point(cam.p, direction);
                const double rpe = PlaneExtractor::compute_reprojection_error(map, track, plane_point);
                if (rpe < max_rpe) best_plane_point = plane_point, have = true;
            }
            // (with no admissible plane the reference hands an uninitialized point to the factor; here the keypoint is left out)
            if (have) {
                PnpPointFactor f;
                for (int k = 0; k < 3; ++k) f.point[k] = best_plane_point[k];
                f.z_tgt[0] = zt[0], f.z_tgt[1] = zt[1];
                pb.point_factors.push_back(f);
            }
            continue;
        }
        const auto anchor = track->first_keypoint();
        PnpFactor f;
        put_state(f.anchor_state, anchor.first), put_ext(f.anchor_cam, anchor.first->camera);
        const vector<2> &zr = anchor.first->get_keypoint(anchor.second);
        f.z_ref[0] = zr[0], f.z_ref[1] = zr[1], f.z_tgt[0] = zt[0], f.z_tgt[1] = zt[1];
        f.inv_depth = track->landmark.inv_depth;
        pb.factors.push_back(f);
    }
    double x[16];
    put_state(x, frame);
    solve_pnp(pb, x, (int)config->solver_iteration_limit());
    for (int k = 0; k < 4; ++k) frame->pose.q.coeffs()[k] = x[k];
    for (int k = 0; k < 3; ++k) frame->pose.p[k] = x[4 + k];
    if (use_inertial)
        for (int k = 0; k < 3; ++k) frame->motion.v[k] = x[7 + k], frame->motion.bg[k] = x[10 + k], frame->motion.ba[k] = x[13 + k];
}

} // namespace pvio
