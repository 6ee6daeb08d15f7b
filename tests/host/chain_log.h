// chain_log.h -- TEST INFRASTRUCTURE (tests/host): the record stream of a sequence run (tests/test_chain_parity.py).
//
// Both chains -- the product (HIP kernels behind the C ABI + the host code of pvio_amd/host) and the oracle chain
// (oracle_chain.cpp: every arithmetic piece replaced by the CPU oracle's) -- write the same records while they run the same
// rendered sequence through the same driver (standin/headless.*); the Python test compares the two files.
//
// Binary stream of records: int32 tag, int32 n_ints, int32 n_doubles, int64 ints[n_ints], double doubles[n_doubles].
//   tag 1  camera frame    ints: frame index, frame id, initialized, window frames, n keypoints, then per keypoint (track id or 0, track length)
//                          doubles: per keypoint x y (normalized), then the reported pose t p(3) q(xyzw)
//   tag 2  window solve    ints: N, M, n_obs, use_inertial, prior_n, termination, is_usable, num_iterations, num_successful_steps, trace_len,
//                                then per trace entry (iteration, step_is_valid, step_is_successful), then lm_valid[M]
//                          doubles: initial_cost, final_cost, per trace entry (cost, cost_change, gradient_max_norm, step_norm,
//                                relative_decrease, trust_region_radius, mu), trace states [trace_len][16N+M], final frame states [16N],
//                                inverse depths [M], quality [M]
//   tag 3  marginalization ints: N, victim, n (remaining), rc     doubles: S [15n][15n], s [15n]
//   tag 6  PnP solve       ints: anchored factors, point factors, inertial, iterations, termination   doubles: state in [16], state out [16], costs [2]
//   tag 8  window tracks   ints: frame index, window frames, n tracks, then per track (id, TF_VALID, observations)   doubles: per track inv_depth, quality
//   tags 4, 5, 7 (product chain only): the oracle run on the inputs of the preceding record 2, 3, 6 -- same layout
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>

namespace chain_log {

inline FILE *&file() {
    static FILE *f = nullptr;
    return f;
}
inline void open(const char *path) {
    if (file()) std::fclose(file());
    file() = path && *path ? std::fopen(path, "wb") : nullptr;
}
inline void close() {
    if (file()) std::fclose(file());
    file() = nullptr;
}
inline void record(int32_t tag, const std::vector<int64_t> &ints, const std::vector<double> &doubles) {
    FILE *f = file();
    if (!f) return;
    const int32_t head[3] = {tag, (int32_t)ints.size(), (int32_t)doubles.size()};
    std::fwrite(head, sizeof head, 1, f);
    if (!ints.empty()) std::fwrite(ints.data(), sizeof(int64_t), ints.size(), f);
    if (!doubles.empty()) std::fwrite(doubles.data(), sizeof(double), doubles.size(), f);
}

} // namespace chain_log
