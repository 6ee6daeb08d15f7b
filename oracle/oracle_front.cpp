// oracle_front.cpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into or called by the product).
//
// Restates the keypoint bookkeeping around the LK tracker, decision by decision, from the reference:
//   PoissonDiskFilter<2>        pvio/src/pvio/utility/poisson_disk_filter.h:25-130
//   Frame::track_keypoints      pvio/src/pvio/map/frame.cpp:89-139  (prediction :97-103, selection :108-130)
// Written independently of pvio_amd/host/feature_front.cpp (ordered std::map grid, the probe list spelled out cell by
// cell) so that the two can check each other.  PARITY UNPINNED: the reference has no tests or vectors for this code
// and cannot be built here (Eigen absent); what pins it is tests/test_host_frontend.py (brute-force properties).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <utility>
#include <vector>

namespace {

struct Filter {
    double r2, cell;
    std::map<std::pair<int, int>, size_t> grid; // cell -> index of the LAST point stored there (:40-44)
    std::vector<std::pair<double, double>> pts;
    explicit Filter(double radius) : r2(radius * radius), cell(radius / std::sqrt(2.0)) {}
    std::pair<int, int> cell_of(double x, double y) const { return {(int)std::floor(x / cell), (int)std::floor(y / cell)}; }
    // poisson_disk_filter.h:92-110 with grid_span = ceil(sqrt(2)) = 2: the cursor is advanced before the first probe
    // and the row bound is only checked on entry -> the 24 cells of the 5 x 5 block except its first one, plus the cell
    // below the block's first column.
    bool permit(double x, double y) const {
        const auto c = cell_of(x, y);
        std::vector<std::pair<int, int>> probes;
        for (int dy = -2; dy <= 2; ++dy)
            for (int dx = -2; dx <= 2; ++dx)
                if (!(dx == -2 && dy == -2)) probes.push_back({c.first + dx, c.second + dy});
        probes.push_back({c.first - 2, c.second + 3});
        for (const auto &pc : probes) {
            auto it = grid.find(pc);
            if (it == grid.end()) continue;
            const double ddx = x - pts[it->second].first, ddy = y - pts[it->second].second;
            if (ddx * ddx + ddy * ddy < r2) return false;
        }
        return true;
    }
    void preset(double x, double y) {
        grid[cell_of(x, y)] = pts.size();
        pts.push_back({x, y});
    }
};

} // namespace

extern "C" {

// insert_points (:65-74): candidates in order; accepted[i] = 1 if candidate i was inserted. presets are inserted first.
void oracle_poisson_insert(double radius, int n_preset, const double *preset_xy, int n, const double *xy, uint8_t *accepted) {
    Filter f(radius);
    for (int i = 0; i < n_preset; ++i) f.preset(preset_xy[2 * i], preset_xy[2 * i + 1]);
    for (int i = 0; i < n; ++i) {
        accepted[i] = f.permit(xy[2 * i], xy[2 * i + 1]) ? 1 : 0;
        if (accepted[i]) f.preset(xy[2 * i], xy[2 * i + 1]);
    }
}

// frame.cpp:108-130: survivors (status != 0, has a track) sorted by track length, longest first (std::sort with the
// same comparator: the order of equal lengths is whatever the library produces), Poisson-disk acceptance on the
// next-image pixels; rejected ones get status 0.
void oracle_select_tracked(int n, const double *next_xy, const uint64_t *track_length, double min_distance, uint8_t *status) {
    std::vector<std::pair<size_t, size_t>> order;
    order.reserve((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (status[i] == 0) continue;
        if (track_length[i] == 0) continue; // get_track(i) == nullptr
        order.emplace_back((size_t)i, (size_t)track_length[i]);
    }
    std::sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.second > b.second; });
    Filter f(min_distance);
    for (auto &[idx, len] : order) {
        (void)len;
        if (f.permit(next_xy[2 * idx], next_xy[2 * idx + 1])) f.preset(next_xy[2 * idx], next_xy[2 * idx + 1]);
        else status[idx] = 0;
    }
}

// frame.cpp:97-103: delta = (q_ci^-1 q_ii dq q_ij^-1 q_cj)^-1 applied to the homogeneous normalized keypoint, then
// hnormalized and mapped through the next frame's K.  Quaternions x,y,z,w; the rotation is done with the rotation matrix
// (Eigen's operator* uses v + w t + u x t, t = 2 u x v: the same numbers up to rounding).
void oracle_predict_keypoints(const double q_cam_i[4], const double q_imu_i[4], const double dq[4], const double q_imu_j[4], const double q_cam_j[4],
                              const double K_next[4] /* fx fy cx cy */, int n, const double *kp_xy, double *out_xy) {
    auto mul = [](const double *a, const double *b, double *o) {
        o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
        o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
        o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
        o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    };
    auto conj = [](const double *a, double *o) { o[0] = -a[0], o[1] = -a[1], o[2] = -a[2], o[3] = a[3]; };
    double t0[4], t1[4], t2[4], t3[4], c[4];
    conj(q_cam_i, c);
    mul(c, q_imu_i, t0);
    mul(t0, dq, t1);
    conj(q_imu_j, c);
    mul(t1, c, t2);
    mul(t2, q_cam_j, t3);
    double d[4];
    conj(t3, d);
    const double x = d[0], y = d[1], z = d[2], w = d[3];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                         2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    for (int i = 0; i < n; ++i) {
        const double b[3] = {kp_xy[2 * i], kp_xy[2 * i + 1], 1.0};
        const double r0 = R[0] * b[0] + R[1] * b[1] + R[2] * b[2], r1 = R[3] * b[0] + R[4] * b[1] + R[5] * b[2], r2 = R[6] * b[0] + R[7] * b[1] + R[8] * b[2];
        out_xy[2 * i] = (r0 / r2) * K_next[0] + K_next[2];
        out_xy[2 * i + 1] = (r1 / r2) * K_next[1] + K_next[3];
    }
}

} // extern "C"
