/*
 * mcml_oracle.h -- CPU ORACLE for the glmmrMCML hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under glmmrmcml_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline.
 *
 * It is a plain-C restatement of the reference's Eigen/C++ algorithm
 * (samuel-watson/glmmrMCML v0.2.2); every function cites the reference
 * file:line it follows.  The reference cannot be compiled here (no R, Rcpp,
 * Eigen, glmmrBase, SparseChol, rminqa in the image), so:
 *
 *   PARITY UNPINNED at the glmmrBase / SparseChol / rminqa boundary: the
 *   covariance-function table, dhdmu/mod_inv_func and the BOBYQA trajectory are
 *   restated from the interface facts visible in the reference sources and from
 *   their published definitions; the reference holds no tests or golden vectors.
 *   What IS pinned: closed-form known answers (scipy.stats), libstdc++'s
 *   minstd_rand/uniform_real_distribution stream (tests compile a probe with
 *   g++), the Random123 Philox4x32-10 known-answer vectors and AS241 vs
 *   scipy.stats.norm.ppf.
 */
#ifndef MCML_ORACLE_H
#define MCML_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG contract ---- */
void     orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double   orc_u52(uint32_t w0, uint32_t w1);
double   orc_dlog(double x);
double   orc_ppnd16(double p);
double   orc_normal(uint64_t seed, uint32_t elem, uint32_t chain, uint32_t prop, uint32_t tag);
uint32_t orc_minstd_next(uint32_t *x);
double   orc_minstd_canonical(uint32_t *x);
uint32_t orc_chain_minstd_seed(uint64_t seed, uint32_t chain, uint32_t iter);

/* ---- GLM scalar maths (moremaths.h) ---- */
int    orc_flink(const char *family, const char *link);
int    orc_link_code(const char *link);
double orc_log_factorial_approx(double n);
double orc_logpdf(double y, double mu, double var_par, int flink);
double orc_mod_inv(double eta, int link_code);
double orc_dhdmu(double eta, int flink);
double orc_detadmu(double eta, int link_code);
double orc_digamma(double x);

/* ---- covariance / MVN log-likelihood (mcmldmatrix.h + glmmrBase restated) ---- */
int orc_cov_npar(const int32_t *cov, int rows);
int orc_cov_nblocks(const int32_t *cov, int rows);
int orc_cov_N(const int32_t *cov, int rows);
int orc_gen_D(const int32_t *cov, int rows, const double *data, const double *eff,
              const double *gamma, int chol, double *D /* N x N col-major */);
int orc_mvn_ll(const int32_t *cov, int rows, const double *data, const double *eff,
               const double *gamma, const double *u, int Q, int m,
               int per_column_refactor, double *out);
int orc_chol_lower(double *A, int n, int lda);

/* ---- model kernels (mcmlmodel.h) ---- */
void   orc_gemv_n(int n, int Q, const double *A, const double *v, double *out);     /* out += A v   */
void   orc_gemv_t(int n, int Q, const double *A, const double *s, double *out);     /* out  = A' s  */
void   orc_gemm_nn(int M, int N, int K, const double *A, int lda, const double *B, int ldb,
                   double *C, int ldc);                                             /* C = A B      */
double orc_log_prob(int n, int Q, const double *xb, const double *ZL, const double *y,
                    double var_par, int flink, const double *v);
void   orc_log_grad(int n, int Q, const double *xb, const double *ZL, const double *y,
                    double var_par, int flink, const double *v, double *grad);
double orc_model_loglik(int n, int Q, int m, const double *Z, const double *xb,
                        const double *y, const double *u, int ldu, double var_par,
                        int flink, int recompute_zu, const double *zu_cached);

/* ---- HMC (mhmcmc.h) ---- */
typedef struct {
    int    warmup;
    int    nsamp;
    int    adapt;          /* 100 in the reference (mhmcmc.h:123) */
    double lambda;
    int    max_steps;
    double target_accept;
} orc_hmc_opts;

typedef struct {
    int    accept;
    double e;
    double ebar;
    int    steps;
} orc_hmc_diag;

int orc_hmc_chain(int n, int Q, const double *xb, const double *ZL, const double *y,
                  double var_par, int flink, const orc_hmc_opts *o,
                  uint64_t seed, uint32_t chain_id, uint32_t iter_idx,
                  const double *inj_init, const double *inj_mom,
                  double *samples /* Q x (nsamp+1), whitened v */,
                  uint8_t *accept_flags /* warmup+nsamp or NULL */,
                  double *probs /* warmup+nsamp or NULL */,
                  orc_hmc_diag *diag);

/* ---- MCNR step (mcmloptim.h:198-236) ---- */
int orc_mcnr(int n, int Q, int P, int m, const double *X, const double *Z, const double *y,
             const double *u, int ldu, const double *beta, double var_par,
             int flink, int link_code, int family_code,
             double *XtWX_sum /* P*P */, double *XtWr_sum /* P */, double *sigma_sum,
             double *beta_out, double *sigma_out);

int orc_num_threads(void);
void orc_set_num_threads(int t);

#ifdef __cplusplus
}
#endif
#endif
