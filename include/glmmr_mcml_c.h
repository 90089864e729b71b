/*
 * glmmr_mcml_c.h -- C ABI of libglmmr_mcml_hip.so, the MI355X (gfx950) build of
 * glmmrMCML's MCML inner loop.  Plain pointers and sizes only; every matrix is
 * column-major float64, every integer array int32, exactly as R hands them to
 * the reference's Rcpp exports (src/RcppExports.cpp:15-310).
 *
 * Conventions: every function returns 0 or a negative error code and never
 * throws or calls the R API; glmmr_mcml_last_error() gives the text.  The
 * caller owns every buffer it passes; a context owns its device memory and is
 * used from one host thread at a time.  There is no CPU fallback: without a
 * visible gfx950 device compute entry points return GLMMR_MCML_ENODEVICE.
 */
#ifndef GLMMR_MCML_C_H
#define GLMMR_MCML_C_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLMMR_MCML_OK            0
#define GLMMR_MCML_EINVAL       -1
#define GLMMR_MCML_EUNSUPPORTED -2   /* reference: unordered_map::at throws (mcmlmodel.h:89) */
#define GLMMR_MCML_ENOTPD       -3   /* reference does not check (SURVEY 8b "Errors") */
#define GLMMR_MCML_ESINGULAR    -4
#define GLMMR_MCML_EHIP         -5
#define GLMMR_MCML_ENODEVICE    -6
#define GLMMR_MCML_ENOMEM       -7

/* text of the calling thread's last error; replaces Rcpp's BEGIN_RCPP/END_RCPP
 * exception -> R condition translation (src/RcppExports.cpp:18,43) */
const char* glmmr_mcml_last_error(void);

/* ------------------------------------------------------------------------- */
/* Device-resident problem                                                    */
/* ------------------------------------------------------------------------- */

/* The arguments every Rcpp export receives (R6ModelExtMCML.R:399-419):
 * covariance$get_D_data() = (cov, data), eff_range, Z, X, y, family, link.
 * Z, X, y may be NULL (n = 0) for a context that only evaluates mvn_ll. */
typedef struct glmmr_mcml_problem {
    const int32_t* cov;       /* cov_rows x 5 column-major (mcml_optim.cpp:20-21) */
    int            cov_rows;
    const double*  data;      /* flattened block data (mcml_optim.cpp:22) */
    int            data_len;
    const double*  eff_range; /* per-function effective range (mcml_optim.cpp:23) */
    int            eff_len;
    const double*  Z;         /* n x Q */
    const double*  X;         /* n x P */
    const double*  y;         /* n */
    int            n, Q, P;
    const char*    family;    /* "poisson" | "binomial" | "gaussian" | "gamma" | "beta" */
    const char*    link;
} glmmr_mcml_problem;

/* reduction hook: sum `n` doubles at `dev_buf` (device memory) over all ranks,
 * in place.  NULL = single process. */
typedef int (*glmmr_mcml_reduce_fn)(void* user, double* dev_buf, int n);

typedef struct glmmr_mcml_dev_opts {
    int   device;          /* HIP device ordinal */
    void* stream;          /* hipStream_t to run on, or NULL for a private stream */
    int   rank, world;     /* this process's shard of the chains / sample columns */
    glmmr_mcml_reduce_fn reduce;
    void* reduce_user;
} glmmr_mcml_dev_opts;

typedef struct glmmr_mcml_ctx glmmr_mcml_ctx;

int glmmr_mcml_ctx_create(const glmmr_mcml_problem* prob, const glmmr_mcml_dev_opts* opts,
                          glmmr_mcml_ctx** out);
int glmmr_mcml_ctx_destroy(glmmr_mcml_ctx* ctx);

/* Native multi-GPU exchange (no reference counterpart: the reference is one process, src/Makevars:14-15;
 * north-star row 8(e)).  One process per GPU; rank 0 makes an id, the host program hands the 128 bytes to the
 * other ranks by whatever transport it has (MPI, a file, R's parallel sockets, torch.distributed), every rank
 * calls comm_init_rccl on its context.  From then on every statistic the path exchanges -- P*P+P+2 doubles per
 * MCNR step, (sum, count) per objective evaluation -- is ONE ncclAllReduce(sum, f64) on the context's stream
 * (RCCL over xGMI), with no host synchronisation in front of it; `reduce` above is ignored.  librccl.so is
 * loaded on first use. */
#define GLMMR_MCML_RCCL_ID_BYTES 128
int glmmr_mcml_rccl_unique_id(unsigned char* id128);
int glmmr_mcml_ctx_comm_init_rccl(glmmr_mcml_ctx* ctx, const unsigned char* id128, int rank, int world);
/* Sums vals[0..n) over the ranks of the context's group through the very path the statistics take (native RCCL
 * communicator or reduce hook; identity for a single process) and returns the sums in place: a caller's self-test
 * of the exchange before it commits to it (glmmrmcml_amd/dist.py::init_native_rccl). */
int glmmr_mcml_ctx_comm_allreduce(glmmr_mcml_ctx* ctx, double* vals, int n);

/* collectives issued so far, doubles summed, 1 if the native communicator is in use (all nullable) */
int glmmr_mcml_ctx_comm_stats(glmmr_mcml_ctx* ctx, long long* calls, long long* doubles, int* native);
/* out6 = [all-gathers issued, doubles received by them, theta-step rounds (one all-reduce each), candidate thetas
 * evaluated on this rank, candidate thetas evaluated over all ranks, 0] since the context was made */
int glmmr_mcml_ctx_shard_stats(glmmr_mcml_ctx* ctx, long long* out6);

/* samples u (Q x ncols, this rank's columns).  niter = columns the beta-step
 * reads (mcmlmodel.h:73,296); the theta-step reads all ncols (mcmldmatrix.h:24). */
int glmmr_mcml_set_u(glmmr_mcml_ctx* ctx, const double* u, int Q, int ncols, int niter);
int glmmr_mcml_get_u(glmmr_mcml_ctx* ctx, double* u, int ldu);
/* All ranks' samples (Q x world * ncols, rank r's columns at column r * ncols): what the reference's mcml_full
 * returns as u (mcml_full.cpp:144-145) when the chains are sharded.  COLLECTIVE unless the samples have not changed
 * since the last theta-step (which gathers them): after glmmr_mcml_ctx_full it moves no data between ranks.
 * ncols_out (nullable) = columns written.  Single process: the same as glmmr_mcml_get_u. */
int glmmr_mcml_get_u_all(glmmr_mcml_ctx* ctx, double* u, int ldu, int* ncols_out);

/* MCMLDmatrix::loglik(u) at theta (mcmldmatrix.h:23-41) */
int glmmr_mcml_ctx_mvn_ll(glmmr_mcml_ctx* ctx, const double* theta, double* out);
/* the same at k candidate thetas (npar x k, column-major) FACTORISED SIDE BY SIDE: when D consists of large dense blocks
 * only, the k matrices sit in one workspace and every launch of the factorisation's schedule covers all of them -- one
 * evaluation of such a block is a latency chain that leaves most of the chip idle.  Other models: one
 * after the other.  out[j] = log-likelihood at candidate j (equal to the single evaluation to rounding), NaN where
 * D(theta_j) is not positive definite.  k <= 64. */
int glmmr_mcml_ctx_mvn_ll_batch(glmmr_mcml_ctx* ctx, const double* thetas, int k, double* out);
/* DMatrix::genD(0, chol, false) (mcml_full.cpp:68) -> out (Q x Q) */
int glmmr_mcml_ctx_gen_D(glmmr_mcml_ctx* ctx, const double* theta, int chol, double* out, int ldo);

/* mcmlModel::log_likelihood() at beta / var_par (mcmlmodel.h:284-304; L_likelihood,
 * likelihood.h:57-64 returns its negative) */
int glmmr_mcml_ctx_loglik(glmmr_mcml_ctx* ctx, const double* beta, double var_par, double* out);
/* one mcmloptim::mcnr() step (mcmloptim.h:198-236).  stats_out (nullable):
 * P*P + P + 1 doubles = sum_i X'W_iX | sum_i X'(W_i detadmu resid_i) | sum_i sigma_i */
int glmmr_mcml_ctx_mcnr(glmmr_mcml_ctx* ctx, const double* beta, double var_par, double* beta_out,
                        double* sigma_out, double* stats_out);
/* dmat.update_parameters(theta); L = genD(0,true,false); model.update_L()
 * (mcml_full.cpp:120-125) */
int glmmr_mcml_ctx_update_L(glmmr_mcml_ctx* ctx, const double* theta);
/* take L as given (export mcmc_sample receives it, mcml_full.cpp:315) */
int glmmr_mcml_ctx_set_L(glmmr_mcml_ctx* ctx, const double* L, int ldl);

/* mcmcRunHMC options (mhmcmc.h:37-43; R field mcmc_options, R6ModelExtMCML.R:867-872) */
typedef struct glmmr_mcml_hmc_opts {
    int    warmup;
    int    nsamp;          /* samples wanted in total (m) */
    int    adapt;          /* proposals that adapt the step size; reference: 100 (mhmcmc.h:123) */
    double lambda;         /* trajectory length */
    int    max_steps;
    double target_accept;
    int    chains;         /* 1: the reference's one sequential chain, u is Q x (nsamp+1);
                              C > 1: C chains x ceil(nsamp/C) post-warmup draws, u is Q x C*ceil(nsamp/C) */
    int    chain_offset;   /* global id of this rank's first chain (keys the RNG streams) */
} glmmr_mcml_hmc_opts;

typedef struct glmmr_mcml_hmc_diag {
    double    accept_rate, mean_e, min_e, max_e;
    int       max_steps_used;
    long long leapfrog_total;
} glmmr_mcml_hmc_diag;

/* mcmcRunHMC::sample (mhmcmc.h:121-157): fills the context's samples with L * v.
 * inj_init (Q x chains) / inj_mom (Q x chains x proposals) replace the generated
 * initial state / momenta (parity tests); flags_out / probs_out (chains x proposals)
 * receive every accept decision and acceptance probability.  All four nullable. */
int glmmr_mcml_ctx_hmc_sample(glmmr_mcml_ctx* ctx, const double* beta, double var_par,
                              const glmmr_mcml_hmc_opts* opts, uint64_t seed, uint32_t iter_idx,
                              const double* inj_init, const double* inj_mom, uint8_t* flags_out,
                              double* probs_out, glmmr_mcml_hmc_diag* diag, int* ncols_out);

/* No-U-Turn sampler standing where the reference calls Stan through cmdstanr (R/gen_u_samples.R:38-69,
 * R6ModelExtMCML.R:234-257 with inst/stan/mcml_*.stan: gamma ~ std_normal(), y ~ family(Xb + Z L gamma)).
 * Stan's multinomial NUTS with the generalised U-turn criterion, dual-averaging step size and diag_e metric
 * adaptation (csrc/nuts.h lists what is and is not reproduced).  Stan's defaults apply where a field is 0. */
typedef struct glmmr_mcml_nuts_opts {
    int    warmup;          /* iter_warmup */
    int    nsamp;           /* iter_sampling: draws wanted in total */
    int    max_treedepth;   /* 0 -> 10 */
    double adapt_delta;     /* 0 -> 0.8 */
    double stepsize;        /* initial step size before the init_stepsize search; 0 -> 1 */
    int    chains;          /* C chains x ceil(nsamp/C) draws each; u is Q x C*ceil(nsamp/C) (no column 0) */
    int    chain_offset;    /* global id of this rank's first chain */
    int    metric;          /* 0: diag_e with Stan's windowed adaptation (its default); 1: unit_e */
} glmmr_mcml_nuts_opts;

typedef struct glmmr_mcml_nuts_diag {
    double    mean_e, min_e, max_e;          /* step sizes after adaptation */
    long long divergent;                     /* transitions that ended in a divergence (all chains, warm-up included) */
    long long treedepth_hits;                /* transitions stopped by max_treedepth */
    long long batched_leapfrogs;             /* leapfrog steps launched (each advances every growing chain) */
    long long stepsize_search_leapfrogs;
} glmmr_mcml_nuts_diag;

/* Fills the context's samples with L * gamma like glmmr_mcml_ctx_hmc_sample.  depth_out / nleap_out (int) and
 * eps_out / accept_out (double), each chains x (warmup + ceil(nsamp/chains)), nullable: tree depth, leapfrog count,
 * step size used and mean acceptance statistic of every transition. */
int glmmr_mcml_ctx_nuts_sample(glmmr_mcml_ctx* ctx, const double* beta, double var_par,
                               const glmmr_mcml_nuts_opts* opts, uint64_t seed, uint32_t iter_idx,
                               int* depth_out, int* nleap_out, double* eps_out, double* accept_out,
                               glmmr_mcml_nuts_diag* diag, int* ncols_out);

/* ------------------------------------------------------------------------- */
/* Mirrors of the Rcpp exports (host buffers in, host buffers out)            */
/* ------------------------------------------------------------------------- */

/* What the reference leaves to chance or to its dependencies' defaults. */
typedef struct glmmr_mcml_ext {
    uint64_t seed;      /* 0: draw one from std::random_device as the reference does (mhmcmc.h:55) */
    int      chains;    /* <= 1: the reference's single sequential chain; C: C concurrent chains */
    int      maxfun;    /* objective evaluations per optimiser call; 0 = 10000 (minqa default) */
    int      device;    /* HIP device ordinal */
    int      theta_batch; /* candidate thetas per rank and round of the batch schedule (csrc/optim.h bobyqa_batch) of the
                             theta-step and of mcml_simlik.  0 = default: when D consists of large dense blocks only -- a
                             round's candidates are then factorised side by side in one pass -- rounds of 8 in all (8 for a
                             single process, 8 / world per rank of a sharded job), else 1 per rank; 1 = a single process
                             runs the reference's sequential BOBYQA, a sharded job one candidate per rank and round */
} glmmr_mcml_ext;

/* gen_u_samples(y, X, Z, L, beta, family, sigma, warmup_iter, m) -> Q x m      -- R/gen_u_samples.R:38-69
 * (the R function takes a family object; here family / link strings as in the other exports) */
int glmmr_mcml_gen_u_samples(const double* Z, const double* L, const double* X, const double* y, int n, int Q, int P,
                             const double* beta, const char* family, const char* link, double sigma, int warmup_iter,
                             int m, const glmmr_mcml_nuts_opts* opts, const glmmr_mcml_ext* ext, double* samples,
                             int lds, int* ncols);

/* columns of u a sampler call returns for m samples: m + 1 for one chain
 * (mhmcmc.h:126), chains * ceil(m / chains) otherwise */
int glmmr_mcml_sample_cols(int m, int chains);
int glmmr_mcml_ctx_ncols(glmmr_mcml_ctx* ctx);
/* HIP-event timing of the sampler's two GEMMs on the context's stream.
 * out8 (nullable) = [forward ms, forward launches, backward ms, backward launches so far,
 * executed flops per forward / per backward launch, dense flops per launch (2 n Q C),
 * operator kind: 0 dense GEMM, 1 banded GEMM (structural zeros of ZL skipped), 2 sparse] */
int glmmr_mcml_ctx_profile(glmmr_mcml_ctx* ctx, int enable, int reset, double* out8);
/* The sampler TIMES one proposal in four (a marker between two dependent launches costs ~2.5 us of idle GPU): out8's
 * launch counts are the timed launches; these are all forward / backward launches since the last reset. */
int glmmr_mcml_ctx_profile_launches(glmmr_mcml_ctx* ctx, long long* fwd, long long* bwd);
/* kernel family that served the sampler's last forward / backward product: 0 streamed few-column kernel
 * (dgemm_skinny.h), 1 banded FP64 MFMA kernel (dgemm_band.h), 2 dense direct-to-LDS MFMA kernel (dgemm_dlds.h),
 * 3 register-staged MFMA kernel (dgemm_mfma.h), 4 sparse chain-major operator (hmc_cm.h); -1 none yet */
int glmmr_mcml_ctx_last_kernels(glmmr_mcml_ctx* ctx, int* fwd, int* bwd);
/* Host wall-clock time per phase of the MCML iterations run by this process since the last reset (csrc/trace.h):
 * out8 (nullable) = [sample, beta-step, theta-step, refresh] ms, then the four phase counts.  enable / reset as above. */
int glmmr_mcml_dbg_phase_ms(int enable, int reset, double* out8);
int glmmr_mcml_ctx_npar(glmmr_mcml_ctx* ctx);

/* The same drivers on a resident context (what bench.py times). */
int glmmr_mcml_ctx_optim(glmmr_mcml_ctx* ctx, const double* start, int nstart, int trace, int mcnr,
                         const glmmr_mcml_ext* ext, double* beta, double* theta, double* sigma);
int glmmr_mcml_ctx_simlik(glmmr_mcml_ctx* ctx, const double* start, int nstart, int trace,
                          const glmmr_mcml_ext* ext, double* beta, double* theta, double* sigma);
int glmmr_mcml_ctx_hess(glmmr_mcml_ctx* ctx, const double* start, int nstart, double tol, int trace,
                        double* H /* (P+R) x (P+R) */);
int glmmr_mcml_ctx_aic(glmmr_mcml_ctx* ctx, const double* beta_par, int nbeta, const double* cov_par,
                       int ncov, double* out);
int glmmr_mcml_ctx_full(glmmr_mcml_ctx* ctx, const double* start, int nstart, int mcnr, int m, int maxiter,
                        int warmup, double tol, int verbose, double lambda, int trace, int refresh,
                        int maxsteps, double target_accept, const glmmr_mcml_ext* ext, double* beta,
                        double* theta, double* sigma, int* converged, int* iters,
                        glmmr_mcml_hmc_diag* last_diag);

/* mcml_full(cov, data, eff_range, Z, X, y, family, link, start, mcnr, m, maxiter, warmup, tol,
 *           verbose, lambda, trace, refresh, maxsteps, target_accept)
 *   -> list(beta, theta, sigma, converged, u)               -- src/mcml_full.cpp:41-148
 * u receives Q x *ucols columns (glmmr_mcml_sample_cols(m, chains)). */
int glmmr_mcml_full(const glmmr_mcml_problem* prob, const double* start, int nstart, int mcnr, int m,
                    int maxiter, int warmup, double tol, int verbose, double lambda, int trace,
                    int refresh, int maxsteps, double target_accept, const glmmr_mcml_ext* ext,
                    double* beta, double* theta, double* sigma, int* converged, double* u, int ldu,
                    int* ucols);

/* mcmc_sample(Z, L, X, y, beta, family, link, warmup, nsamp, lambda, var_par, trace, refresh,
 *             maxsteps, target_accept) -> Q x (nsamp+1)      -- src/mcml_full.cpp:314-338 */
int glmmr_mcml_mcmc_sample(const double* Z, const double* L, const double* X, const double* y, int n,
                           int Q, int P, const double* beta, const char* family, const char* link,
                           int warmup, int nsamp, double lambda, double var_par, int trace, int refresh,
                           int maxsteps, double target_accept, const glmmr_mcml_ext* ext,
                           double* samples, int lds, int* ncols);

/* mcml_optim(cov, data, eff_range, Z, X, y, u, family, link, start, trace, mcnr)
 *   -> list(beta, theta, sigma)                              -- src/mcml_optim.cpp:35-68 */
int glmmr_mcml_optim(const glmmr_mcml_problem* prob, const double* u, int ucols, const double* start,
                     int nstart, int trace, int mcnr, const glmmr_mcml_ext* ext, double* beta,
                     double* theta, double* sigma);
/* mcml_simlik(..., u, family, link, start, trace)            -- src/mcml_optim.cpp:90-117 */
int glmmr_mcml_simlik(const glmmr_mcml_problem* prob, const double* u, int ucols, const double* start,
                      int nstart, int trace, const glmmr_mcml_ext* ext, double* beta, double* theta,
                      double* sigma);
/* mcml_optim_sparse / mcml_simlik_sparse (..., Ap, Ai, ...)  -- src/mcml_optim.cpp:147-184,210-239.
 * D is block diagonal, so the (Ap, Ai) pattern is checked against the blocks and the same
 * block-factorised path is used (the reference's sparse loglik reads column 0 of u for every
 * sample, defect D1: not reproduced). */
int glmmr_mcml_optim_sparse(const glmmr_mcml_problem* prob, const int32_t* Ap, const int32_t* Ai, int nnz,
                            const double* u, int ucols, const double* start, int nstart, int trace,
                            int mcnr, const glmmr_mcml_ext* ext, double* beta, double* theta,
                            double* sigma,
                            /* LDL' factor of D(theta), as the reference returns it (mcml_optim.cpp:180-182):
                             * unit lower L column-compressed without its diagonal (Lp: Q+1, Li/Lx: lcap
                             * entries, size from glmmr_mcml_sparse_factor_nnz) and the pivots D (Q).
                             * All four nullable. */
                            int32_t* Lp, int32_t* Li, double* Lx, double* D, int lcap);
int glmmr_mcml_sparse_factor_nnz(const int32_t* cov, int cov_rows, const double* data, int data_len);
int glmmr_mcml_simlik_sparse(const glmmr_mcml_problem* prob, const int32_t* Ap, const int32_t* Ai, int nnz,
                             const double* u, int ucols, const double* start, int nstart, int trace,
                             const glmmr_mcml_ext* ext, double* beta, double* theta, double* sigma);
/* mcml_hess(..., u, family, link, start, tol = 1e-5, trace)  -- src/mcml_optim.cpp:263-285 */
int glmmr_mcml_hess(const glmmr_mcml_problem* prob, const double* u, int ucols, const double* start,
                    int nstart, double tol, int trace, const glmmr_mcml_ext* ext, double* H);
int glmmr_mcml_hess_sparse(const glmmr_mcml_problem* prob, const int32_t* Ap, const int32_t* Ai, int nnz,
                           const double* u, int ucols, const double* start, int nstart, double tol,
                           int trace, const glmmr_mcml_ext* ext, double* H);
/* aic_mcml(..., u, family, link, beta_par, cov_par)          -- src/mcml_optim.cpp:356-392 */
int glmmr_mcml_aic(const glmmr_mcml_problem* prob, const double* u, int ucols, const double* beta_par,
                   int nbeta, const double* cov_par, int ncov, const glmmr_mcml_ext* ext, double* out);

/* mcml_la / mcml_la_nr(cov, data, eff_range, Z, X, y, family, link, start, usehess, tol, verbose, trace,
 * maxiter) -> List(beta, theta, sigma, se, u)   -- src/mcml_la.cpp:28-155 / 174-290
 * Laplace-approximation fits: (beta, v) by BOBYQA (mcml_la) or one Newton-Raphson step per iteration
 * (mcml_la_nr), theta (+ sigma) by BOBYQA on ll - v'v/2 - logdet(ZL' W ZL + I)/2, then a joint polish.
 * se: nstart entries (zeros unless usehess); u: Q entries (= L v).  ext->maxfun bounds each BOBYQA run. */
int glmmr_mcml_la(const glmmr_mcml_problem* prob, const double* start, int nstart, int usehess, double tol,
                  int verbose, int trace, int maxiter, const glmmr_mcml_ext* ext, double* beta, double* theta,
                  double* sigma, double* se, double* u);
int glmmr_mcml_la_nr(const glmmr_mcml_problem* prob, const double* start, int nstart, int usehess, double tol,
                     int verbose, int trace, int maxiter, const glmmr_mcml_ext* ext, double* beta, double* theta,
                     double* sigma, double* se, double* u);
/* the same on a resident context (nr = 0 / 1); converged / iters nullable */
int glmmr_mcml_ctx_la(glmmr_mcml_ctx* ctx, const double* start, int nstart, int nr, int usehess, double tol,
                      int verbose, int trace, int maxiter, const glmmr_mcml_ext* ext, double* beta, double* theta,
                      double* sigma, double* se, double* u, int* converged, int* iters);

/* mvn_ll(cov, data, eff_range, gamma, u)  -- src/mcml_optim.cpp:406-414 */
int glmmr_mcml_mvn_ll(const int32_t* cov, int cov_rows, const double* data, int data_len,
                      const double* eff_range, int eff_len, const double* gamma, int ngamma,
                      const double* u, int Q, int m, double* out);

/* ---- test hooks (building blocks exposed for tests/ and bench.py only) ---- */
/* every evaluation of the theta-step's objective on this context as a row (theta_1 .. theta_R, MVN log-likelihood):
 * enable 1 clears the log and starts it, 0 stops it, -1 leaves it as it is; out (nullable) receives the LAST cap_rows rows
 * (row-major, R + 1 doubles each), *nrows the number written (or held, when out is null) */
int glmmr_mcml_dbg_theta_log(glmmr_mcml_ctx* ctx, int enable, double* out, int cap_rows, int* nrows);
/* One rank of an N-rank job on ONE GPU, for timing what a rank executes when no N-GPU node is at hand (bench.py
 * --as-rank-of N).  The context stays a single process; its peers are emulated as copies of itself: a sum is N times
 * the local value, the sample all-gather N copies of the local block.  mode 1: the candidate thetas of ALL ranks are
 * evaluated here and their values recorded; mode 2: only rank 0's share is evaluated, the other values come from the
 * record (the same fit must be re-run from the same start: every own value is checked against the record) -- the
 * launches of mode 2 are exactly those of rank 0 of the real job; what it leaves out is the time of the collectives
 * themselves.  world <= 1 switches the emulation off. */
int glmmr_mcml_dbg_emulate_world(glmmr_mcml_ctx* ctx, int world, int mode);
/* shader-clock timestamps of the phases of one 128 x 128 Cholesky leaf: [start, loaded, sum (a) diagonal tiles,
 * sum (b) panel solves, sum (c) trailing updates, factor done, L written, inverse diag done, inverse done, end] */
int glmmr_mcml_dbg_leaf_profile(glmmr_mcml_ctx* ctx, unsigned long long* out10);
/* Laplace path pieces from the state the drivers start in (model built from `start`, v given, W as
 * update_W leaves it): kind 0 LA_likelihood(par = beta, v), 1 LA_likelihood_cov(par = theta[, var_par]),
 * 2 LA_likelihood_btheta(par = beta, theta[, var_par]) -> *out (likelihood.h:112-230);
 * kind 3 one mcnr_b step (mcmloptim.h:238-293) -> v_out (Q), beta_out (P), sigma_out */
int glmmr_mcml_dbg_la_probe(glmmr_mcml_ctx* ctx, const double* start, int nstart, int kind, const double* v,
                            double var_par, const double* par, int npar, double* out, double* v_out,
                            double* beta_out, double* sigma_out);
int glmmr_mcml_dbg_dgemm(int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                         int b_nmajor, double alpha, double beta, double* C, int ldc,
                         int lower_only, int force_tile /* -1 = auto; 0.. tiles of the register-staged kernel; 20..25 the
                                                           Cholesky's deep-ring kernel; 40 the sampler's dense direct-to-LDS
                                                           kernel (dgemm_dlds_asm_kernel) */);
/* mcmlModel::log_prob / log_grad (mcmlmodel.h:138-279) of every column of V (Q x ncols) */
int glmmr_mcml_dbg_log_prob_grad(glmmr_mcml_ctx* ctx, const double* beta, double var_par, const double* V,
                                 int ncols, double* lp, double* G);
/* RNG contract on the device: Philox/AS241 normals and the minstd canonical stream */
int glmmr_mcml_dbg_normals(uint64_t seed, uint32_t chain, uint32_t prop, uint32_t tag, int n, double* out);
int glmmr_mcml_dbg_minstd(uint32_t seed, int n, double* out);
/* host optimiser / finite differences on a caller-supplied objective (CPU only) */
typedef double (*glmmr_mcml_objective)(const double* x, int n, void* user);
int glmmr_mcml_dbg_bobyqa(glmmr_mcml_objective f, void* user, int n, const double* x0, const double* lower,
                          const double* upper, double rhobeg, double rhoend, int maxfun, double* x_out,
                          double* f_out, int* nfev_out);
/* the batch schedule of the same optimiser (csrc/optim.h bobyqa_batch): `width` points per round; rounds_out = the
 * sequential depth.  The callback is still called once per point. */
int glmmr_mcml_dbg_bobyqa_batch(glmmr_mcml_objective f, void* user, int n, const double* x0, const double* lower,
                                const double* upper, double rhobeg, double rhoend, int maxfun, int width,
                                double* x_out, double* f_out, int* nfev_out, int* rounds_out);
/* the same driven by a batch callback: one call per round with all its points (X is n x k column-major, F receives k values;
 * returns 0 or an error code that aborts the run) -- the shape of a rank's work in a sharded theta-step */
typedef int (*glmmr_mcml_batch_objective)(const double* X, int n, int k, double* F, void* user);
int glmmr_mcml_dbg_bobyqa_rounds(glmmr_mcml_batch_objective fb, void* user, int n, const double* x0, const double* lower,
                                 const double* upper, double rhobeg, double rhoend, int maxfun, int width,
                                 double* x_out, double* f_out, int* nfev_out, int* rounds_out);
/* host -> device -> host through the library's staged copies (csrc/common.hip copy_h2d_2d / copy_d2h_2d): `cols` columns of
 * `rows` doubles, host pitches ld_in / ld_out (in doubles), a padded pitch on the device */
int glmmr_mcml_dbg_copy_roundtrip(const double* in, long long ld_in, double* out, long long ld_out, long long rows, long long cols);
int glmmr_mcml_dbg_fd_hessian(glmmr_mcml_objective f, void* user, int n, const double* x, double ndeps,
                              int usebounds, const double* lower, const double* upper, double* H);
int glmmr_mcml_dbg_dgemm_bench(int M, int N, int K, int b_nmajor, int iters, int force_tile,
                               double* ms_per_launch);
/* the same with lower tiles only and C read-modify-written (beta != 0), as the Cholesky updates run */
int glmmr_mcml_dbg_dgemm_bench2(int M, int N, int K, int b_nmajor, int iters, int force_tile, int lower_only,
                                double beta, double* ms_per_launch);
/* C = A B through the banded zero-skipping kernel of the sampler (dgemm_band.h); tiles_executed (nullable) =
 * K tiles of 32 actually multiplied, summed over the 80-row bands */
int glmmr_mcml_dbg_dgemm_band(int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C,
                              int ldc, int* tiles_executed);
/* sustained shader clock under the banded FP64 MFMA kernel: out3 = [ms per launch, shader MHz, executed TFLOP/s] */
int glmmr_mcml_dbg_band_clocks(int M, int N, int iters, int mode, double* out3);
/* mode: 0 the real kernel; timing experiments (results meaningless): 1 no LDS-DMA, 2 no barriers, 4 no ds_reads */

#ifdef __cplusplus
}
#endif
#endif
