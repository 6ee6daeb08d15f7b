// ba_solver.h -- host orchestration of the on-device bundle adjustment (one instance per pvio_hip_ctx).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/pvio_hip.h"
#include "ba_types.h"

namespace pvba {

struct Comm; // RCCL communicator wrapper (ba_comm.cpp)
int comm_unique_id(uint8_t id[128]);
int comm_init(Comm **out, const uint8_t id[128], int rank, int world, int device);
int comm_allreduce(Comm *c, double *buf, size_t n, int op_max, hipStream_t st);
void comm_destroy(Comm *c);

class DevicePool { // grow-only device allocations keyed by name
  public:
    ~DevicePool();
    void *get(const char *name, size_t bytes, bool *grew = nullptr);
    void release();
    size_t total_bytes() const { return total_; }

  private:
    struct Slot {
        std::string name;
        void *ptr = nullptr;
        size_t bytes = 0;
    };
    std::vector<Slot> slots_;
    size_t total_ = 0;
};

class BASolver {
  public:
    BASolver(int device, int rank, int world, bool use_graph);
    void set_force_sharded(bool on) { sharded_ = world_ > 1 || on; }
    void set_fault_injection(int fail_factorizations, int invalid_steps) { dbg_fail_ = fail_factorizations, dbg_invalid_ = invalid_steps; }
    void set_linearize_mode(int m) { lin_mode_ = m; }
    void set_reuse_candidates(bool on) { reuse_cand_ = on; }
    int last_candidate_repeats() const { return last_repeats_; }
    int graph_replays() const { return graph_replays_; }
    ~BASolver();
    int upload(const pvio_ba_problem *pb, const pvio_ba_state *st, bool may_return_early = false);   // H2D of the flat problem + initial state
    // runs from the uploaded initial state; `read_back`: the accepted iterate + quality pass land in the caller's arrays as part of the same
    // stream round (what upload + solve + download cost as three round trips before)
    int solve(pvio_ba_summary *sum, pvio_ba_kernel_times *prof = nullptr, pvio_ba_state *read_back = nullptr);
    int download(pvio_ba_state *st);                                   // D2H of the accepted iterate (+ quality pass)
    int reprojection_error(double *out);
    int marginalize(const pvio_ba_problem *pb, const pvio_ba_state *st, int victim, pvio_ba_prior *out);
    void set_comm(Comm *c) { comm_ = c; }
    const std::string &error() const { return err_; }
    hipStream_t stream() const { return stream_; }

  private:
    int fail(int code, const std::string &msg);
    int check(hipError_t e, const char *what);
    int run_slots(int n_slots);
    int enqueue_slot(hipEvent_t *ev = nullptr);
    void invalidate_graph();

    int device_, rank_, world_;
    bool sharded_; // world_ > 1, or forced for tests: eager launches + all-reduces + assembly from the reduced buffer
    bool use_graph_;
    int lin_mode_ = 0; // pvio_hip_opts::linearize_mode
    bool reuse_cand_ = false; // pvio_hip_opts::reuse_identical_candidates
    int last_repeats_ = 0;    // Ctrl::cand_repeats of the last solve
    int graph_replays_ = 0;   // hipGraphLaunch calls so far (diagnostics: pvio_hip_ba_graph_replays)
    int dbg_fail_ = 0, dbg_invalid_ = 0; // tests only: forced factorization failures / invalid steps per solve
    hipStream_t stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    DevicePool pool_;
    View v_{};
    Ctrl *h_ctrl_ = nullptr; // pinned
    bool uploaded_ = false;
    int solves_since_upload_ = 0;
    int cus_ = 0;
    void *h_stage_ = nullptr; // pinned staging of one upload's inputs
    size_t h_stage_cap_ = 0;
    bool stage_in_flight_ = false; // upload(may_return_early) left its staged copy queued; cleared by the next stream synchronization
    double *h_back_ = nullptr; // pinned read-back of one marginalization pass
    size_t h_back_cap_ = 0;    // in doubles
    std::vector<double> marg_H_, marg_C_, marg_V_; // host work arrays of a marginalization (15N x 15N, twice 15(N-1) x 15(N-1))
    double *h_pack_ = nullptr; // pinned: a solve's packed results (k_quality `pack`)
    size_t h_pack_cap_ = 0;    // in doubles
    const double *fs_init_ = nullptr, *rho_init_ = nullptr; // initial state inside the inputs slab
    Ctrl h_ctrl_tmpl_{};                                    // what a solve's control block starts from (k_reset copies the device copy)
    Ctrl *d_ctrl_tmpl_ = nullptr;
    int trace_cap_ = 0;
    hipGraph_t graph_ = nullptr;
    hipGraphExec_t graph_exec_ = nullptr;
    int graph_slots_ = 0;
    double max_solver_time_ = 0.0; // seconds; checked between graph replays
    bool sharded_graph_failed_ = false; // capturing the collectives failed once: eager launches from then on
    Comm *comm_ = nullptr;
    std::string err_;
    std::vector<double> h_init_fs_, h_init_rho_;
};

} // namespace pvba
