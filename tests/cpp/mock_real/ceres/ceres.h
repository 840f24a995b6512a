// API-subset mock (see ../README.md): declarations only, real Ceres names only.
#pragma once
#include <string>
namespace ceres {
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
class LossFunction {
public:
    virtual ~LossFunction();
    virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class ScaledLoss : public LossFunction {
public:
    ScaledLoss(const LossFunction *rho, double a, Ownership ownership);
    void Evaluate(double sq_norm, double out[3]) const override;
};
class LossFunctionWrapper : public LossFunction {
public:
    LossFunctionWrapper(LossFunction *rho, Ownership ownership);
    void Evaluate(double sq_norm, double out[3]) const override;
    void Reset(LossFunction *rho, Ownership ownership);
};
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY };
enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
struct IterationSummary {
    int iteration;
    bool step_is_successful;
    double cost, cost_change;
};
class IterationCallback {
public:
    virtual ~IterationCallback();
    virtual CallbackReturnType operator()(const IterationSummary &summary) = 0;
};
class Solver {
public:
    struct Options {
        LinearSolverType linear_solver_type;
        bool use_nonmonotonic_steps;
        bool minimizer_progress_to_stdout;
        int max_num_iterations;
        double function_tolerance;
        int num_threads;
    };
    struct Summary {
        double initial_cost, final_cost;
        int num_successful_steps;
        std::string FullReport() const;
    };
};
}  // namespace ceres
