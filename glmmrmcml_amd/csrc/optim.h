// optim.h -- host-side derivative-free optimiser and finite differences.
//
// The reference minimises its objective functors with rminqa's Rbobyqa (Powell's
// BOBYQA; mcmloptim.h:56-113) and takes standard errors from rminqa's
// Functor::Hessian (R's optimhess central differences; mcmloptim.h:296-355).
// rminqa is not in the image, so both are restated from their published
// algorithms: PARITY UNPINNED against the real rminqa trajectory; what is tested
// is convergence to the same optimum (tests/test_optim_cpu.py).
//
// bobyqa(): bound-constrained trust-region DFO in the BOBYQA family: npt
// interpolation points, minimum-Frobenius-norm quadratic model (built from the
// explicit KKT system: the problems here have <= ~20 parameters), trust-region
// step by projected truncated conjugate gradients (TRSBOX's job), geometry
// steps that maximise the outgoing point's Lagrange polynomial (ALTMOV's job),
// Powell's rho / delta schedule.  Defaults follow minqa/rminqa: npt = min(n+2,
// 2n), rhobeg = min(0.95, 0.2 max|x0|), rhoend = 1e-6 rhobeg, maxfun = 10000.
#pragma once
#include <functional>
#include <vector>

namespace mcml {

typedef std::function<int(const std::vector<double>& x, double* f)> objective_fn;
// evaluates every point of X (in whatever order, on whatever devices) and returns all the values: one exchange
typedef std::function<int(const std::vector<std::vector<double>>& X, std::vector<double>* F)> batch_objective_fn;

struct BobyqaOpts {
    int    npt = 0;          // 0 = min(n + 2, 2 n)
    double rhobeg = 0.0;     // 0 = min(0.95, 0.2 max |x0|)
    double rhoend = 0.0;     // 0 = 1e-6 rhobeg
    int    maxfun = 10000;
    int    iprint = 0;
};

struct BobyqaResult {
    std::vector<double> x;
    double fval = 0.0;
    int    nfev = 0;
    int    status = 0;       // 0 converged (rho reached rhoend), 1 maxfun reached
    int    rounds = 0;       // bobyqa_batch: batches evaluated (the sequential depth)
};

// lower/upper may hold -HUGE_VAL / +HUGE_VAL.  Returns 0 or the first nonzero
// code the objective returned.
int bobyqa(const objective_fn& f, const std::vector<double>& x0, const std::vector<double>& lower,
           const std::vector<double>& upper, const BobyqaOpts& opts, BobyqaResult* res);

// bobyqa_batch(): the same model-based trust-region method when `width` objective evaluations cost the time of one
// -- the theta-step of a chain-sharded job, where every rank holds all the samples and evaluates ONE candidate
// theta per round (drivers.hip::d_optim).  Same objective, same interpolation models (minimum Frobenius norm; the
// full quadratic when (n+1)(n+2)/2 <= width), same trust-region / geometry sub-problems; what changes is the
// schedule: a round evaluates the trust-region step at several radii at once (delta, 2 delta, delta/2, 4 delta: the
// radius is chosen by the values, not by a ratio test over successive rounds), a replacement for EVERY badly placed
// interpolation point (chosen one after the other against the set as it will be), and, when the model's step is
// shorter than rho/2, the points the next, smaller rho will want around the model's minimiser -- so every value of
// rho costs one round instead of 3-6 sequential evaluations.  Deterministic given (f, x0, width): every rank runs it
// on the same values and proposes the same points.  Sequential depth: res->rounds.
int bobyqa_batch(const batch_objective_fn& f, const std::vector<double>& x0, const std::vector<double>& lower,
                 const std::vector<double>& upper, const BobyqaOpts& opts, int width, BobyqaResult* res);

// R's optimhess / rminqa Functor::Hessian: central differences of the central-
// difference gradient, steps ndeps, optional bounds (one-sided at a bound),
// symmetrised.  H is n x n column-major.
int fd_gradient(const objective_fn& f, const std::vector<double>& x, const std::vector<double>& ndeps,
                bool usebounds, const std::vector<double>& lower, const std::vector<double>& upper,
                std::vector<double>* grad);
int fd_hessian(const objective_fn& f, const std::vector<double>& x, const std::vector<double>& ndeps,
               bool usebounds, const std::vector<double>& lower, const std::vector<double>& upper,
               std::vector<double>* H);

}  // namespace mcml
