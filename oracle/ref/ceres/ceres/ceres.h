// <ceres/ceres.h> -- a miniature of the part of ceres-solver 1.14's public interface the reference's hot path uses.
//
// TEST INFRASTRUCTURE (oracle/ref): ceres-solver (pinned 1.14.0 at pvio/depends/CMakeLists.txt:31-35) is a third-party
// dependency whose source is NOT in /root/reference and cannot be fetched (no network).  This header lets the
// reference's own estimation/bundle_adjustor.cpp, estimation/pnp.cpp and estimation/ceres/*_error_cost.h compile
// UNEDITED (oracle/ref/Makefile -> oracle/_ref/libpvio_ref.so).  Two kinds of content:
//   * interface classes (CostFunction, SizedCostFunction, LocalParameterization, LossFunction/CauchyLoss, Problem,
//     Solver::Options/Summary, IterationCallback): same names and semantics as Ceres 1.14;
//   * ceres::Solve (mini_ceres.cpp): a RESTATEMENT of Ceres 1.14's TrustRegionMinimizer + DoglegStrategy
//     (TRADITIONAL_DOGLEG) + Corrector + Jacobi scaling with a dense Cholesky of the full normal equations in place of
//     SPARSE_SCHUR -- a generic loop over whatever residual blocks the reference's code adds.  It is NOT Ceres: the
//     trust-region loop stays "parity unpinned".  What _ref pins is everything the
//     reference itself contributes: which blocks are added, constant, in which parameterization, every residual and
//     Jacobian, the state-updating callback reading live biases, the post-solve passes and marginalize_frame.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace ceres {

enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum DoglegType { TRADITIONAL_DOGLEG, SUBSPACE_DOGLEG };
enum MinimizerType { LINE_SEARCH, TRUST_REGION };
enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

typedef int int32;

class CostFunction {
  public:
    CostFunction() : num_residuals_(0) {}
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    const std::vector<int32> &parameter_block_sizes() const { return parameter_block_sizes_; }
    int num_residuals() const { return num_residuals_; }

  protected:
    std::vector<int32> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
    void set_num_residuals(int n) { num_residuals_ = n; }

  private:
    std::vector<int32> parameter_block_sizes_;
    int num_residuals_;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
  public:
    SizedCostFunction() {
        set_num_residuals(kNumResiduals);
        *mutable_parameter_block_sizes() = std::vector<int32>{Ns...};
    }
    virtual ~SizedCostFunction() {}
};

class LocalParameterization {
  public:
    virtual ~LocalParameterization() {}
    virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
    virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0; // GlobalSize x LocalSize, row-major
    virtual int GlobalSize() const = 0;
    virtual int LocalSize() const = 0;
};

// Only ever attached to CONSTANT blocks by the reference (plane normals, bundle_adjustor.cpp:106-109): the reduced program
// never calls it.  Plus / ComputeJacobian abort so that a use that would need Ceres' real implementation cannot go unnoticed.
class HomogeneousVectorParameterization : public LocalParameterization {
  public:
    explicit HomogeneousVectorParameterization(int size) : size_(size) {}
    bool Plus(const double *, const double *, double *) const override;
    bool ComputeJacobian(const double *, double *) const override;
    int GlobalSize() const override { return size_; }
    int LocalSize() const override { return size_ - 1; }

  private:
    int size_;
};

class LossFunction {
  public:
    virtual ~LossFunction() {}
    virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class TrivialLoss : public LossFunction {
  public:
    void Evaluate(double s, double rho[3]) const override { rho[0] = s, rho[1] = 1.0, rho[2] = 0.0; }
};
class CauchyLoss : public LossFunction { // loss_function.cc: rho(s) = b log(1 + s / b), b = a^2
  public:
    explicit CauchyLoss(double a) : b_(a * a), c_(1 / b_) {}
    void Evaluate(double s, double rho[3]) const override {
        const double sum = 1.0 + s * c_;
        const double inv = 1.0 / sum;
        rho[0] = b_ * std::log(sum);
        rho[1] = std::max(std::numeric_limits<double>::min(), inv);
        rho[2] = -c_ * (inv * inv);
    }

  private:
    const double b_, c_;
};
class HuberLoss : public LossFunction {
  public:
    explicit HuberLoss(double a) : a_(a), b_(a * a) {}
    void Evaluate(double s, double rho[3]) const override {
        if (s > b_) {
            const double r = std::sqrt(s);
            rho[0] = 2.0 * a_ * r - b_, rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r), rho[2] = -rho[1] / (2.0 * s);
        } else {
            rho[0] = s, rho[1] = 1.0, rho[2] = 0.0;
        }
    }

  private:
    const double a_, b_;
};

struct IterationSummary {
    int iteration = 0;
    bool step_is_valid = false;
    bool step_is_nonmonotonic = false;
    bool step_is_successful = false;
    double cost = 0, cost_change = 0, gradient_max_norm = 0, gradient_norm = 0, step_norm = 0, relative_decrease = 0;
    double trust_region_radius = 0, eta = 0, step_size = 0;
    int line_search_function_evaluations = 0, linear_solver_iterations = 0;
    double iteration_time_in_seconds = 0, step_solver_time_in_seconds = 0, cumulative_time_in_seconds = 0;
    double mu = 0; // (not in Ceres) DoglegStrategy's regularization multiplier after the iteration
};

class IterationCallback {
  public:
    virtual ~IterationCallback() {}
    virtual CallbackReturnType operator()(const IterationSummary &summary) = 0;
};

class Problem {
  public:
    struct Options {
        Ownership cost_function_ownership = TAKE_OWNERSHIP;
        Ownership loss_function_ownership = TAKE_OWNERSHIP;
        Ownership local_parameterization_ownership = TAKE_OWNERSHIP;
        bool enable_fast_removal = false;
        bool disable_all_safety_checks = false;
    };
    struct ParameterBlock {
        double *user = nullptr;
        int size = 0;
        LocalParameterization *local = nullptr;
        bool constant = false;
        int order = 0; // insertion order
    };
    struct ResidualBlock {
        CostFunction *cost = nullptr;
        LossFunction *loss = nullptr;
        std::vector<double *> params;
    };
    typedef const ResidualBlock *ResidualBlockId;

    Problem() {}
    explicit Problem(const Options &o) : options_(o) {}
    ~Problem();
    Problem(const Problem &) = delete;
    Problem &operator=(const Problem &) = delete;

    void AddParameterBlock(double *values, int size);
    void AddParameterBlock(double *values, int size, LocalParameterization *local);
    void SetParameterBlockConstant(double *values);
    void SetParameterBlockVariable(double *values);
    void SetParameterization(double *values, LocalParameterization *local);
    ResidualBlockId AddResidualBlock(CostFunction *cost, LossFunction *loss, const std::vector<double *> &params);
    template <typename... Ps>
    ResidualBlockId AddResidualBlock(CostFunction *cost, LossFunction *loss, double *x0, Ps *... xs) {
        return AddResidualBlock(cost, loss, std::vector<double *>{x0, xs...});
    }
    int NumParameterBlocks() const { return (int)blocks_.size(); }
    int NumResidualBlocks() const { return (int)residuals_.size(); }

    // implementation access (mini_ceres.cpp)
    const std::vector<std::unique_ptr<ResidualBlock>> &residual_blocks() const { return residuals_; }
    const std::map<double *, ParameterBlock> &parameter_blocks() const { return blocks_; }

  private:
    Options options_;
    std::map<double *, ParameterBlock> blocks_;
    std::vector<std::unique_ptr<ResidualBlock>> residuals_;
};

class Solver {
  public:
    struct Options {
        MinimizerType minimizer_type = TRUST_REGION;
        LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
        TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
        DoglegType dogleg_type = TRADITIONAL_DOGLEG;
        bool use_nonmonotonic_steps = false;
        int max_num_iterations = 50;
        double max_solver_time_in_seconds = 1e9;
        int num_threads = 1;
        double initial_trust_region_radius = 1e4;
        double max_trust_region_radius = 1e16;
        double min_trust_region_radius = 1e-32;
        double min_relative_decrease = 1e-3;
        double min_lm_diagonal = 1e-6;
        double max_lm_diagonal = 1e32;
        int max_num_consecutive_invalid_steps = 5;
        double function_tolerance = 1e-6;
        double gradient_tolerance = 1e-10;
        double parameter_tolerance = 1e-8;
        bool jacobi_scaling = true;
        bool minimizer_progress_to_stdout = false;
        bool update_state_every_iteration = false;
        std::vector<IterationCallback *> callbacks;
    };
    struct Summary {
        TerminationType termination_type = FAILURE;
        std::string message;
        double initial_cost = 0, final_cost = 0, fixed_cost = 0;
        std::vector<IterationSummary> iterations;
        int num_successful_steps = 0, num_unsuccessful_steps = 0;
        int num_parameter_blocks_reduced = 0, num_parameters_reduced = 0, num_effective_parameters_reduced = 0;
        int num_residual_blocks_reduced = 0, num_residuals_reduced = 0;
        double total_time_in_seconds = 0;
        int iterations_started = 0; // (not in Ceres) trust-region iterations begun, incl. one ended by a tolerance test before it was recorded
        bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE || termination_type == USER_SUCCESS; }
        std::string BriefReport() const;
        std::string FullReport() const { return BriefReport(); }
    };
};

void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary);

// ---- not in Ceres: hooks for the test harness (oracle/ref/ref_capi.cpp) -------------------------------------------
namespace mini {
// called once per iteration AFTER the state-updating callback and the user's callbacks (iteration 0 included)
void set_observer(std::function<void(const IterationSummary &)> fn);
// the summary of the last ceres::Solve on this thread (the reference keeps its Summary in a local variable)
const Solver::Summary &last_summary();
// fault injection mirroring pvio_hip_opts::debug_*: the first N factorizations of a Solve fail, the first N steps are invalid
void set_fault_injection(int fail_factorizations, int invalid_steps);
} // namespace mini

} // namespace ceres
