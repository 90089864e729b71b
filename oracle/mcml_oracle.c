/*
 * mcml_oracle.c -- CPU ORACLE (test infrastructure only; see mcml_oracle.h).
 *
 * Plain-C restatement of glmmrMCML v0.2.2's MCML hot path.  Citations are
 * file:line relative to the reference tree.  PARITY UNPINNED at the
 * glmmrBase/SparseChol/rminqa boundary (see header).
 *
 * Everything is column-major f64 / int32, as R hands it to the Rcpp exports.
 * Build: see oracle/Makefile (-ffp-contract=off so that the arithmetic is the
 * plain IEEE sequence the RNG contract relies on).
 */
#include "mcml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void orc_set_num_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================== */
/* RNG contract                                                              */
/* ======================================================================== */

/* Philox4x32-10 (Salmon et al., SC'11; Random123).  Counter-based: the build's
 * replacement for R's rnorm stream, which the reference uses for momenta
 * (mhmcmc.h:48-51,62) and which cannot be reproduced without R. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 52-bit uniform strictly inside (0,1); every operation is exact. */
double orc_u52(uint32_t w0, uint32_t w1)
{
    double a = (double)(w0 >> 6);
    double b = (double)(w1 >> 6);
    return (a * 67108864.0 + b + 0.5) * (1.0 / 4503599627370496.0);
}

/* Deterministic natural log: +,*,/ and bit manipulation only, so that the CPU
 * oracle and the HIP device code return the same bits (libm and ocml log are
 * each faithful but not identical).  |rel err| ~ 2e-16 for normal x > 0. */
double orc_dlog(double x)
{
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 27.0;
    p = p * s2 + 1.0 / 25.0;
    p = p * s2 + 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2;
    double ed = (double)e;
    double hi = ed * 0.693147180369123816490;      /* ln2 high part */
    double lo = ed * 1.90821492927058770002e-10;   /* ln2 low part  */
    double t = 2.0 * s;
    return hi + (t + (t * p + lo));
}

/* Wichura's AS241 PPND16 (the algorithm behind R's qnorm), with orc_dlog in
 * the tails so the result is bit-reproducible on the GPU. */
double orc_ppnd16(double p)
{
    double q = p - 0.5, r, val;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        double num = 2.5090809287301226727e+3;
        num = num * r + 3.3430575583588128105e+4;
        num = num * r + 6.7265770927008700853e+4;
        num = num * r + 4.5921953931549871457e+4;
        num = num * r + 1.3731693765509461125e+4;
        num = num * r + 1.9715909503065514427e+3;
        num = num * r + 1.3314166789178437745e+2;
        num = num * r + 3.3871328727963666080e0;
        double den = 5.2264952788528545610e+3;
        den = den * r + 2.8729085735721942674e+4;
        den = den * r + 3.9307895800092710610e+4;
        den = den * r + 2.1213794301586595867e+4;
        den = den * r + 5.3941960214247511077e+3;
        den = den * r + 6.8718700749205790830e+2;
        den = den * r + 4.2313330701600911252e+1;
        den = den * r + 1.0;
        return q * num / den;
    }
    r = (q < 0.0) ? p : 1.0 - p;
    r = sqrt(-orc_dlog(r));
    if (r <= 5.0) {
        r = r - 1.6;
        double num = 7.74545014278341407640e-4;
        num = num * r + 2.27238449892691845833e-2;
        num = num * r + 2.41780725177450611770e-1;
        num = num * r + 1.27045825245236838258e0;
        num = num * r + 3.64784832476320460504e0;
        num = num * r + 5.76949722146069140550e0;
        num = num * r + 4.63033784615654529590e0;
        num = num * r + 1.42343711074968357734e0;
        double den = 1.05075007164441684324e-9;
        den = den * r + 5.47593808499534494600e-4;
        den = den * r + 1.51986665636164571966e-2;
        den = den * r + 1.48103976427480074590e-1;
        den = den * r + 6.89767334985100004550e-1;
        den = den * r + 1.67638483018380384940e0;
        den = den * r + 2.05319162663775882187e0;
        den = den * r + 1.0;
        val = num / den;
    } else {
        r = r - 5.0;
        double num = 2.01033439929228813265e-7;
        num = num * r + 2.71155556874348757815e-5;
        num = num * r + 1.24266094738807843860e-3;
        num = num * r + 2.65321895265761230930e-2;
        num = num * r + 2.96560571828504891230e-1;
        num = num * r + 1.78482653991729133580e0;
        num = num * r + 5.46378491116411436990e0;
        num = num * r + 6.65790464350110377720e0;
        double den = 2.04426310338993978564e-15;
        den = den * r + 1.42151175831644588870e-7;
        den = den * r + 1.84631831751005468180e-5;
        den = den * r + 7.86869131145613259100e-4;
        den = den * r + 1.48753612908506148525e-2;
        den = den * r + 1.36929880922735805310e-1;
        den = den * r + 5.99832206555887937690e-1;
        den = den * r + 1.0;
        val = num / den;
    }
    return (q < 0.0) ? -val : val;
}

/* One standard normal, addressed by (seed; element, chain, proposal, tag).
 * tag: 0 = initial state u (mhmcmc.h:48-49), 2 = momentum of proposal `prop`
 * (mhmcmc.h:62); tag + 16*iter separates MCML iterations. */
double orc_normal(uint64_t seed, uint32_t elem, uint32_t chain, uint32_t prop, uint32_t tag)
{
    uint32_t ctr[4] = { elem, chain, prop, tag };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    return orc_ppnd16(orc_u52(o[0], o[1]));
}

/* std::minstd_rand: x <- 48271 x mod (2^31 - 1)   (mhmcmc.h:27,55) */
uint32_t orc_minstd_next(uint32_t *x)
{
    *x = (uint32_t)(((uint64_t)(*x) * 48271u) % 2147483647u);
    return *x;
}

/* libstdc++ std::uniform_real_distribution<double>(0,1)(minstd_rand)
 * = generate_canonical<double,53>: two engine draws, R = max-min+1 = 2^31-2.
 * (mhmcmc.h:28,56,85) */
double orc_minstd_canonical(uint32_t *x)
{
    const double R = 2147483646.0;
    double sum = 0.0, tmp = 1.0;
    sum += (double)(orc_minstd_next(x) - 1u) * tmp;
    tmp *= R;
    sum += (double)(orc_minstd_next(x) - 1u) * tmp;
    tmp *= R;
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

/* Reference seeds gen_ from std::random_device (mhmcmc.h:55): not
 * reproducible.  The build derives the per-chain seed from Philox. */
uint32_t orc_chain_minstd_seed(uint64_t seed, uint32_t chain, uint32_t iter)
{
    uint32_t ctr[4] = { 0u, chain, iter, 3u };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    uint32_t s = o[0] % 2147483647u;
    return s == 0u ? 1u : s;
}

/* ======================================================================== */
/* GLM scalar maths                                                          */
/* ======================================================================== */

/* mcmlmodel.h:74-89 string_to_case; returns 0 where the reference's
 * unordered_map::at would throw. */
int orc_flink(const char *family, const char *link)
{
    static const char *tab[12][2] = {
        {"poisson", "log"}, {"poisson", "identity"}, {"binomial", "logit"},
        {"binomial", "log"}, {"binomial", "identity"}, {"binomial", "probit"},
        {"gaussian", "identity"}, {"gaussian", "log"}, {"gamma", "log"},
        {"gamma", "inverse"}, {"gamma", "identity"}, {"beta", "logit"}};
    for (int i = 0; i < 12; i++)
        if (!strcmp(family, tab[i][0]) && !strcmp(link, tab[i][1])) return i + 1;
    /* R passes "Gamma"; the map key is "gamma..." so Gamma throws in the
     * reference too (mcmlmodel.h:83-85 vs R6ModelExtMCML.R:148). */
    return 0;
}

/* moremaths.h:123-129 */
int orc_link_code(const char *link)
{
    if (!strcmp(link, "log")) return 1;
    if (!strcmp(link, "identity")) return 2;
    if (!strcmp(link, "logit")) return 3;
    if (!strcmp(link, "probit")) return 4;
    if (!strcmp(link, "inverse")) return 5;
    return 0;
}

/* moremaths.h:16-24 (Ramanujan, with the literal 3.141593) */
double orc_log_factorial_approx(double n)
{
    if (n == 0) return 0;
    return n * log(n) - n + log(n * (1 + 4 * n * (1 + 2 * n))) / 6 + log(3.141593) / 2;
}

static double pnorm_std(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }
static double dnorm_std(double x) { return exp(-0.5 * x * x) * 0.39894228040143267794; }

/* moremaths.h:26-102 */
double orc_logpdf(double y, double mu, double var_par, int flink)
{
    double logl = 0.0;
    switch (flink) {
    case 1: {
        double lf1 = orc_log_factorial_approx(y);
        logl = y * mu - exp(mu) - lf1;
        break;
    }
    case 2: {
        double lf1 = orc_log_factorial_approx(y);
        logl = y * log(mu) - mu - lf1;
        break;
    }
    case 3:
        if (y == 1) logl = log(1 / (1 + exp(-1.0 * mu)));
        else if (y == 0) logl = log(1 - 1 / (1 + exp(-1.0 * mu)));
        break;
    case 4:
        if (y == 1) logl = mu;
        else if (y == 0) logl = log(1 - exp(mu));
        break;
    case 5:
        if (y == 1) logl = log(mu);
        else if (y == 0) logl = log(1 - mu);
        break;
    case 6:
        if (y == 1) logl = log(pnorm_std(mu));
        else if (y == 0) logl = log(1 - pnorm_std(mu));
        break;
    case 7:
        logl = -1 * log(var_par) - 0.5 * log(2 * 3.141593) -
               0.5 * ((y - mu) / var_par) * ((y - mu) / var_par);
        break;
    case 8:
        logl = -1 * log(var_par) - 0.5 * log(2 * 3.141593) -
               0.5 * ((log(y) - mu) / var_par) * ((log(y) - mu) / var_par);
        break;
    case 9: {
        double ymu = var_par * y / exp(mu);
        logl = log(1 / (tgamma(var_par) * y)) + var_par * log(ymu) - ymu;
        break;
    }
    case 10: {
        double ymu = var_par * y * mu;
        logl = log(1 / (tgamma(var_par) * y)) + var_par * log(ymu) - ymu;
        break;
    }
    case 11:
        logl = log(1 / (tgamma(var_par) * y)) + var_par * log(var_par * y / mu) - var_par * y / mu;
        break;
    case 12:
        logl = (mu * var_par - 1) * log(y) + ((1 - mu) * var_par - 1) * log(1 - y) -
               lgamma(mu * var_par) - lgamma((1 - mu) * var_par) + lgamma(var_par);
        break;
    }
    return logl;
}

/* glmmrBase maths::mod_inv_func (mcmloptim.h:214 call site) -- RESTATED:
 * inverse link. */
double orc_mod_inv(double eta, int link_code)
{
    switch (link_code) {
    case 1: return exp(eta);
    case 2: return eta;
    case 3: return exp(eta) / (1 + exp(eta));
    case 4: return pnorm_std(eta);
    case 5: return 1 / eta;
    }
    return eta;
}

/* glmmrBase maths::dhdmu (mcmlmodel.h:122 call site) -- RESTATED so that
 * W = 1/(dhdmu*phi) is the GLM working weight (poisson-log: mu; binomial-logit:
 * p(1-p); gaussian-identity: 1/sigma^2).  Cases 1,3,7 are the in-scope ones. */
double orc_dhdmu(double eta, int flink)
{
    double p;
    switch (flink) {
    case 1: return exp(-1.0 * eta);
    case 2: return exp(eta);
    case 3: p = orc_mod_inv(eta, 3); return 1 / (p * (1.0 - p));
    case 4: p = orc_mod_inv(eta, 3); return (1.0 - p) / p;
    case 5: p = orc_mod_inv(eta, 3); return p * (1.0 - p);
    case 6: p = pnorm_std(eta); return (p * (1 - p)) / dnorm_std(eta);
    case 7: return 1.0;
    case 8: return 1 / exp(eta);
    case 9: return 1.0;
    case 10: return 1 / (eta * eta);
    case 11: return eta * eta;
    case 12: p = orc_mod_inv(eta, 3); return 1 / (p * (1.0 - p));
    }
    return 1.0;
}

/* moremaths.h:118-161 */
double orc_detadmu(double eta, int link_code)
{
    double p;
    switch (link_code) {
    case 1: return exp(-1.0 * eta);
    case 2: return 1.0;
    case 3: p = orc_mod_inv(eta, 3); return 1 / (p * (1.0 - p));
    case 4: return 1 / dnorm_std(eta);
    case 5: return -1.0 * eta * eta;
    }
    return 1.0;
}

/* ======================================================================== */
/* Covariance data (glmmrBase DData / DSubMatrix, RESTATED from the interface  */
/* facts in mcml_optim.cpp:20-23, mcmldmatrix.h:26-30,59-65,                   */
/* R6ModelExtMCML.R:430).                                                      */
/* ======================================================================== */

#define COV(r, c) cov[(r) + (c) * rows]

/* parameters per function id 1..14: R6ModelExtMCML.R:430 */
static const int fn_npar[15] = {0, 1, 1, 1, 2, 2, 1, 2, 2, 2, 2, 2, 2, 2, 1};

int orc_cov_npar(const int32_t *cov, int rows)
{
    int np = 0;
    for (int r = 0; r < rows; r++) {
        int fn = COV(r, 2);
        int e = COV(r, 4) + ((fn >= 1 && fn <= 14) ? fn_npar[fn] : 0);
        if (e > np) np = e;
    }
    return np;
}

int orc_cov_nblocks(const int32_t *cov, int rows)
{
    int b = 0;
    for (int r = 0; r < rows; r++)
        if (COV(r, 0) + 1 > b) b = COV(r, 0) + 1;
    return b;
}

int orc_cov_N(const int32_t *cov, int rows)
{
    int N = 0, last = -1;
    for (int r = 0; r < rows; r++)
        if (COV(r, 0) != last) { N += COV(r, 1); last = COV(r, 0); }
    return N;
}

/* The build's covariance-function table.  Ids follow the parameter-count
 * vector c(1,1,1,2,2,1,2,2,2,2,2,2,2,1) (R6ModelExtMCML.R:430); id 1 = gr is
 * certain (mcmldmatrix.h:61-65).  The others are INFERRED, compatibility with
 * glmmrBase unverified:
 *   1 gr      : d==0 ? v*theta^2 : 0      2 fexp0 : v*exp(-d/theta)
 *   3 ar1     : v*theta^d                 4 sqexp : v*t0*exp(-d^2/t1^2)
 *   7 fexp    : v*t0*exp(-d/t1)          14 sqexp0: v*exp(-d^2/theta^2)
 * 5,6,8-13 (matern, bessel, wendland, prod*) are not built. */
static int cov_apply(int fn, double dist, const double *g, double *val)
{
    switch (fn) {
    case 1: *val = (dist == 0) ? (*val) * g[0] * g[0] : 0.0; return 0;
    case 2: *val = (*val) * exp(-1.0 * dist / g[0]); return 0;
    case 3: *val = (*val) * pow(g[0], dist); return 0;
    case 4: *val = (*val) * g[0] * exp(-1.0 * dist * dist / (g[1] * g[1])); return 0;
    case 7: *val = (*val) * g[0] * exp(-1.0 * dist / g[1]); return 0;
    case 14: *val = (*val) * exp(-1.0 * dist * dist / (g[0] * g[0])); return 0;
    }
    return -1;
}

/* one block: rows [r0,r1) of cov, data block at bd (dim x ncol col-major) */
static int gen_block(const int32_t *cov, int rows, int r0, int r1, const double *bd,
                     const double *gamma, double *A, int lda)
{
    int dim = COV(r0, 1);
    for (int j = 0; j < dim; j++) {
        for (int i = j; i < dim; i++) {
            double val = 1.0;
            int coff = 0;
            for (int k = r0; k < r1; k++) {
                int nv = COV(k, 3);
                double d2 = 0.0;
                for (int p = 0; p < nv; p++) {
                    double df = bd[i + (coff + p) * dim] - bd[j + (coff + p) * dim];
                    d2 += df * df;
                }
                double dist = sqrt(d2);
                if (cov_apply(COV(k, 2), dist, gamma + COV(k, 4), &val)) return -2;
                coff += nv;
            }
            A[i + j * lda] = val;
            A[j + i * lda] = val;
        }
    }
    return 0;
}

/* plain lower Cholesky, in place; upper triangle zeroed.  Stands in for
 * glmmrBase gen_block_mat(b,true,false) (mcmldmatrix.h:59). */
int orc_chol_lower(double *A, int n, int lda)
{
    for (int j = 0; j < n; j++) {
        double d = A[j + j * lda];
        for (int k = 0; k < j; k++) d -= A[j + k * lda] * A[j + k * lda];
        if (!(d > 0.0)) return -3;
        d = sqrt(d);
        A[j + j * lda] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i + j * lda];
            for (int k = 0; k < j; k++) s -= A[i + k * lda] * A[j + k * lda];
            A[i + j * lda] = s / d;
        }
        for (int i = 0; i < j; i++) A[i + j * lda] = 0.0;
    }
    return 0;
}

static void block_extent(const int32_t *cov, int rows, int r0, int *r1, int *ncol)
{
    int r = r0, nc = 0;
    while (r < rows && COV(r, 0) == COV(r0, 0)) { nc += COV(r, 3); r++; }
    *r1 = r; *ncol = nc;
}

/* DMatrix::genD(0,chol,false): the block-diagonal D (or its lower Cholesky
 * factor), N x N (mcml_full.cpp:68). */
int orc_gen_D(const int32_t *cov, int rows, const double *data, const double *eff,
              const double *gamma, int chol, double *D)
{
    (void)eff;
    int N = orc_cov_N(cov, rows);
    memset(D, 0, sizeof(double) * (size_t)N * N);
    int r0 = 0, mstart = 0;
    size_t doff = 0;
    while (r0 < rows) {
        int r1, ncol, dim = COV(r0, 1);
        block_extent(cov, rows, r0, &r1, &ncol);
        double *A = D + mstart + (size_t)mstart * N;
        int rc = gen_block(cov, rows, r0, r1, data + doff, gamma, A, N);
        if (rc) return rc;
        if (chol) { rc = orc_chol_lower(A, dim, N); if (rc) return rc; }
        doff += (size_t)dim * ncol; mstart += dim; r0 = r1;
    }
    return 0;
}

/* moremaths.h:166-179 */
static void forward_sub(const double *U, int ldu, const double *u, int n, double *y)
{
    for (int i = 0; i < n; i++) {
        double lsum = 0;
        for (int j = 0; j < i; j++) lsum += U[i + j * ldu] * y[j];
        y[i] = (u[i] - lsum) / U[i + i * ldu];
    }
}

/* mcmldmatrix.h:57-78 given the block's Cholesky factor */
static double loglik_block(const double *L, int n, int all_gr, const double *u, double *work)
{
    double logl = 0;
    if (all_gr) {
        for (int k = 0; k < n; k++) {
            double d = L[k + k * n];
            logl += -0.5 * log(d * d) - 0.5 * log(2 * M_PI) - 0.5 * u[k] * u[k] / (d * d);
        }
    } else {
        double logdetD = 0;
        for (int i = 0; i < n; i++) logdetD += 2 * log(L[i + i * n]);
        forward_sub(L, n, u, n, work);
        double quadform = 0;
        for (int i = 0; i < n; i++) quadform += work[i] * work[i];
        logl = (-0.5 * n * log(2 * M_PI) - 0.5 * logdetD - 0.5 * quadform);
    }
    return logl;
}

/* MCMLDmatrix::loglik (mcmldmatrix.h:23-41) == export mvn_ll
 * (mcml_optim.cpp:406-414).  per_column_refactor=1 rebuilds and refactors the
 * block for every column exactly as the reference does (defect D2); 0 factors
 * once per block.  Both give identical numbers. */
int orc_mvn_ll(const int32_t *cov, int rows, const double *data, const double *eff,
               const double *gamma, const double *u, int Q, int m,
               int per_column_refactor, double *out)
{
    (void)eff;
    double loglV = 0;
    int r0 = 0, mstart = 0, rc_all = 0;
    size_t doff = 0;
    while (r0 < rows) {
        int r1, ncol, dim = COV(r0, 1);
        block_extent(cov, rows, r0, &r1, &ncol);
        int all_gr = 1;
        for (int k = r0; k < r1; k++) if (COV(k, 2) != 1) all_gr = 0;
        double *loglB = (double *)calloc((size_t)m, sizeof(double));
        double *Lshared = NULL;
        if (!per_column_refactor) {
            Lshared = (double *)malloc(sizeof(double) * (size_t)dim * dim);
            int rc = gen_block(cov, rows, r0, r1, data + doff, gamma, Lshared, dim);
            if (!rc) rc = orc_chol_lower(Lshared, dim, dim);
            if (rc) rc_all = rc;
        }
        if (!rc_all) {
#pragma omp parallel if ((long long)dim * dim * (per_column_refactor ? dim : 1) * m > 2000000)
            {
                double *work = (double *)malloc(sizeof(double) * (size_t)dim);
                double *Lp = per_column_refactor
                                 ? (double *)malloc(sizeof(double) * (size_t)dim * dim) : NULL;
#pragma omp for schedule(dynamic)
                for (int i = 0; i < m; i++) {
                    const double *L = Lshared;
                    if (per_column_refactor) {
                        int rc = gen_block(cov, rows, r0, r1, data + doff, gamma, Lp, dim);
                        if (!rc) rc = orc_chol_lower(Lp, dim, dim);
                        if (rc) {
#pragma omp critical
                            rc_all = rc;
                            continue;
                        }
                        L = Lp;
                    }
                    loglB[i] = loglik_block(L, dim, all_gr, u + mstart + (size_t)i * Q, work);
                }
                free(work);
                free(Lp);
            }
        }
        for (int i = 0; i < m; i++) loglV += loglB[i];
        free(loglB);
        free(Lshared);
        if (rc_all) return rc_all;
        doff += (size_t)dim * ncol; mstart += dim; r0 = r1;
    }
    *out = loglV / m;
    return 0;
}

/* ======================================================================== */
/* Dense helpers                                                             */
/* ======================================================================== */

/* out += A v ; A is n x Q col-major.  Each out[i] is summed in column order,
 * whatever the thread count. */
void orc_gemv_n(int n, int Q, const double *A, const double *v, double *out)
{
#pragma omp parallel if ((long long)n * Q > 200000)
    {
        int nt = 1, t = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads(); t = omp_get_thread_num();
#endif
        int lo = (int)((long long)n * t / nt), hi = (int)((long long)n * (t + 1) / nt);
        for (int j = 0; j < Q; j++) {
            double vj = v[j];
            const double *a = A + (size_t)j * n;
            for (int i = lo; i < hi; i++) out[i] += a[i] * vj;
        }
    }
}

/* out = A' s */
void orc_gemv_t(int n, int Q, const double *A, const double *s, double *out)
{
#pragma omp parallel for if ((long long)n * Q > 200000)
    for (int j = 0; j < Q; j++) {
        const double *a = A + (size_t)j * n;
        double acc = 0;
        for (int i = 0; i < n; i++) acc += a[i] * s[i];
        out[j] = acc;
    }
}

void orc_gemm_nn(int M, int N, int K, const double *A, int lda, const double *B, int ldb,
                 double *C, int ldc)
{
#pragma omp parallel for if ((long long)M * N * K > 2000000)
    for (int j = 0; j < N; j++) {
        double *c = C + (size_t)j * ldc;
        for (int i = 0; i < M; i++) c[i] = 0;
        for (int k = 0; k < K; k++) {
            double b = B[k + (size_t)j * ldb];
            if (b == 0.0) continue;
            const double *a = A + (size_t)k * lda;
            for (int i = 0; i < M; i++) c[i] += a[i] * b;
        }
    }
}

/* ======================================================================== */
/* Model kernels                                                             */
/* ======================================================================== */

/* mcmlModel::log_prob (mcmlmodel.h:138-153) */
double orc_log_prob(int n, int Q, const double *xb, const double *ZL, const double *y,
                    double var_par, int flink, const double *v)
{
    double *mu = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(mu, xb, sizeof(double) * (size_t)n);
    orc_gemv_n(n, Q, ZL, v, mu);
    double ll = 0, lp = 0;
    for (int i = 0; i < n; i++) ll += orc_logpdf(y[i], mu[i], var_par, flink);
    for (int i = 0; i < Q; i++) lp += orc_logpdf(v[i], 0, 1, 7);
    free(mu);
    return ll + lp;
}

/* digamma(x), x > 0 -- stands where boost::math::digamma is called
 * (mcmlmodel.h:271; boost is not in the image).  Published algorithm, the build's own
 * statement of it: psi(x) = psi(x+1) - 1/x up to x >= 10, then the asymptotic series
 * ln x - 1/(2x) - sum_k B_2k / (2k x^2k), k = 1..7 (next term < 5e-17 at x = 10).
 * Same expression order as glm_digamma in glmmrmcml_amd/csrc/glm.h.  KAT: scipy.special.digamma
 * (tests/test_oracle_kat.py). */
double orc_digamma(double x)
{
    if (!(x > 0)) return 0.0 / 0.0;
    double r = 0.0;
    while (x < 10.0) { r = r - 1 / x; x = x + 1; }
    const double i2 = 1 / (x * x);
    double t = 1.0 / 12;                       /* B14/14 */
    t = 691.0 / 32760 - t * i2;                /* B12/12 */
    t = 1.0 / 132 - t * i2;                    /* B10/10 */
    t = 1.0 / 240 - t * i2;                    /* B8/8   */
    t = 1.0 / 252 - t * i2;                    /* B6/6   */
    t = 1.0 / 120 - t * i2;                    /* B4/4   */
    t = 1.0 / 12 - t * i2;                     /* B2/2   */
    return r + (log(x) - 0.5 / x - t * i2);
}

/* the score s(y,mu) applied before ZL' (mcmlmodel.h:169-276) and the scalar
 * applied after it */
static double score(double y, double mu, double var_par, int flink, double *post)
{
    *post = 1.0;
    switch (flink) {
    case 1: return y - exp(mu);
    case 2: return y * (1 / mu) - 1;
    case 3: { double t = exp(mu); t = t + 1; t = 1 / t; t = t + y; return t - 1; }
    case 4: if (y == 1) return 1; else if (y == 0) return exp(mu) / (1 - exp(mu)); return mu;
    case 5: if (y == 1) return 1 / mu; else if (y == 0) return -1 / (1 - mu); return mu;
    case 6:
        if (y == 1) return dnorm_std(mu) / pnorm_std(mu);
        else if (y == 0) return -1.0 * dnorm_std(mu) / (1 - pnorm_std(mu));
        return mu;
    case 7: case 8: *post = 1.0 / (var_par * var_par); return y - mu;
    case 9: *post = var_par; return y * exp(-1.0 * mu) - 1;
    case 10: *post = var_par; return (1 / mu) - y;
    case 11: *post = var_par; { double im = 1 / mu; return y * im * im - im; }
    case 12: {   /* mcmlmodel.h:266-275, literally: the second line reads the UPDATED mu(i) (= p), so the
                  * factor is p/(1+exp(p)), not p(1-p) */
        double p = exp(mu) / (exp(mu) + 1);
        return (p / (1 + exp(p))) * var_par *
               (log(y) - log(1 - y) - orc_digamma(p * var_par) + orc_digamma((1 - p) * var_par));
    }
    }
    return 0;
}

/* mcmlModel::log_grad, usezl=true (mcmlmodel.h:156-279) */
void orc_log_grad(int n, int Q, const double *xb, const double *ZL, const double *y,
                  double var_par, int flink, const double *v, double *grad)
{
    double *mu = (double *)malloc(sizeof(double) * (size_t)n);
    double *g = (double *)malloc(sizeof(double) * (size_t)Q);
    memcpy(mu, xb, sizeof(double) * (size_t)n);
    orc_gemv_n(n, Q, ZL, v, mu);
    double post = 1.0;
    for (int i = 0; i < n; i++) mu[i] = score(y[i], mu[i], var_par, flink, &post);
    orc_gemv_t(n, Q, ZL, mu, g);
    for (int k = 0; k < Q; k++) grad[k] = -1.0 * v[k] + post * g[k];
    free(mu); free(g);
}

/* mcmlModel::log_likelihood (mcmlmodel.h:284-304).  recompute_zu=1 redoes
 * Z*u as the reference does on every call (:286); otherwise zu_cached (n x m)
 * is used.  Same value either way. */
double orc_model_loglik(int n, int Q, int m, const double *Z, const double *xb,
                        const double *y, const double *u, int ldu, double var_par,
                        int flink, int recompute_zu, const double *zu_cached)
{
    double *zd = NULL;
    const double *zu = zu_cached;
    if (recompute_zu || !zu_cached) {
        zd = (double *)malloc(sizeof(double) * (size_t)n * m);
        orc_gemm_nn(n, m, Q, Z, n, u, ldu, zd, n);
        zu = zd;
    }
    double *ll = (double *)calloc((size_t)m, sizeof(double));
#pragma omp parallel for if ((long long)n * m > 200000)
    for (int j = 0; j < m; j++) {
        double acc = 0;
        for (int i = 0; i < n; i++)
            acc += orc_logpdf(y[i], xb[i] + zu[i + (size_t)j * n], var_par, flink);
        ll[j] = acc;
    }
    double s = 0;
    for (int j = 0; j < m; j++) s += ll[j];
    free(ll); free(zd);
    return s / m;
}

/* ======================================================================== */
/* HMC: one chain, mcmcRunHMC::{initialise_u,new_proposal,sample}            */
/* (mhmcmc.h:47-157)                                                         */
/* ======================================================================== */
int orc_hmc_chain(int n, int Q, const double *xb, const double *ZL, const double *y,
                  double var_par, int flink, const orc_hmc_opts *o,
                  uint64_t seed, uint32_t chain_id, uint32_t iter_idx,
                  const double *inj_init, const double *inj_mom,
                  double *samples, uint8_t *accept_flags, double *probs,
                  orc_hmc_diag *diag)
{
    double *u = (double *)malloc(sizeof(double) * (size_t)Q);
    double *up = (double *)malloc(sizeof(double) * (size_t)Q);
    double *r = (double *)malloc(sizeof(double) * (size_t)Q);
    double *grad = (double *)malloc(sizeof(double) * (size_t)Q);
    uint32_t tagbase = 16u * iter_idx;

    /* initialise_u (mhmcmc.h:47-59) */
    for (int k = 0; k < Q; k++)
        u[k] = inj_init ? inj_init[k] : orc_normal(seed, (uint32_t)k, chain_id, 0u, tagbase + 0u);
    int accept_count = 0;
    double H = 0, e = 0.001, ebar = 1.0;
    uint32_t gen = orc_chain_minstd_seed(seed, chain_id, iter_idx);
    int steps = 1;

    int total = o->warmup + o->nsamp;
    for (int it = 0; it < total; it++) {
        int adapt = (it < o->warmup) && (it < o->adapt);   /* mhmcmc.h:131-136 */
        int iter = it + 1;
        /* new_proposal (mhmcmc.h:61-119) */
        for (int k = 0; k < Q; k++)
            r[k] = inj_mom ? inj_mom[k + (size_t)it * Q]
                           : orc_normal(seed, (uint32_t)k, chain_id, (uint32_t)it, tagbase + 2u);
        orc_log_grad(n, Q, xb, ZL, y, var_par, flink, u, grad);
        double lpr = 0;
        for (int k = 0; k < Q; k++) lpr += r[k] * r[k];
        lpr *= 0.5;
        memcpy(up, u, sizeof(double) * (size_t)Q);
        steps = (int)round(o->lambda / e);
        if (steps < 1) steps = 1;
        if (steps > o->max_steps) steps = o->max_steps;
        for (int s = 0; s < steps; s++) {
            for (int k = 0; k < Q; k++) r[k] += (e / 2) * grad[k];
            for (int k = 0; k < Q; k++) up[k] += e * r[k];
            orc_log_grad(n, Q, xb, ZL, y, var_par, flink, up, grad);
            for (int k = 0; k < Q; k++) r[k] += (e / 2) * grad[k];
        }
        double lprt = 0;
        for (int k = 0; k < Q; k++) lprt += r[k] * r[k];
        lprt *= 0.5;
        double l1 = orc_log_prob(n, Q, xb, ZL, y, var_par, flink, u);
        double l2 = orc_log_prob(n, Q, xb, ZL, y, var_par, flink, up);
        double prob = fmin(1.0, exp(-l1 + lpr + l2 - lprt));
        double runif = orc_minstd_canonical(&gen);
        int acc = runif < prob;
        if (acc) { memcpy(u, up, sizeof(double) * (size_t)Q); accept_count++; }
        if (accept_flags) accept_flags[it] = (uint8_t)acc;
        if (probs) probs[it] = prob;
        if (adapt) {
            double f1 = 1.0 / (iter + 10);
            H = (1 - f1) * H + f1 * (o->target_accept - prob);
            double loge = -4.60517 - (sqrt((double)iter / 0.05)) * H;
            double powm = pow((double)iter, -0.75);
            double logbare = powm * loge + (1 - powm) * log(ebar);
            e = exp(loge);
            ebar = exp(logbare);
        } else {
            e = ebar;
        }
        if (it == o->warmup - 1 || (o->warmup == 0 && it == 0)) { /* handled below */ }
        if (it >= o->warmup)
            memcpy(samples + (size_t)(it - o->warmup + 1) * Q, u, sizeof(double) * (size_t)Q);
        else if (it == o->warmup - 1)
            memcpy(samples, u, sizeof(double) * (size_t)Q);   /* samples.col(0)=u, :142 */
    }
    if (o->warmup == 0) {
        /* col 0 is the initial state when there is no warmup (mhmcmc.h:142) */
        for (int k = 0; k < Q; k++)
            samples[k] = inj_init ? inj_init[k]
                                  : orc_normal(seed, (uint32_t)k, chain_id, 0u, tagbase + 0u);
    }
    if (diag) { diag->accept = accept_count; diag->e = e; diag->ebar = ebar; diag->steps = steps; }
    free(u); free(up); free(r); free(grad);
    return 0;
}

/* ======================================================================== */
/* MCNR step (mcmloptim.h:198-236), serial semantics (defect D3 not kept)    */
/* ======================================================================== */
static int inv_spd_small(double *A, int P)
{
    /* Gauss-Jordan with partial pivoting, in place (Eigen .inverse() stand-in) */
    int *piv = (int *)malloc(sizeof(int) * (size_t)P);
    double *B = (double *)calloc((size_t)P * P, sizeof(double));
    for (int i = 0; i < P; i++) B[i + i * P] = 1.0;
    for (int c = 0; c < P; c++) {
        int p = c; double best = fabs(A[c + c * P]);
        for (int i = c + 1; i < P; i++) if (fabs(A[i + c * P]) > best) { best = fabs(A[i + c * P]); p = i; }
        if (best == 0.0) { free(piv); free(B); return -4; }
        if (p != c) for (int j = 0; j < P; j++) {
            double t = A[c + j * P]; A[c + j * P] = A[p + j * P]; A[p + j * P] = t;
            t = B[c + j * P]; B[c + j * P] = B[p + j * P]; B[p + j * P] = t;
        }
        double d = A[c + c * P];
        for (int j = 0; j < P; j++) { A[c + j * P] /= d; B[c + j * P] /= d; }
        for (int i = 0; i < P; i++) if (i != c) {
            double f = A[i + c * P];
            if (f != 0.0) for (int j = 0; j < P; j++) { A[i + j * P] -= f * A[c + j * P]; B[i + j * P] -= f * B[c + j * P]; }
        }
    }
    memcpy(A, B, sizeof(double) * (size_t)P * P);
    free(piv); free(B);
    return 0;
}

int orc_mcnr(int n, int Q, int P, int m, const double *X, const double *Z, const double *y,
             const double *u, int ldu, const double *beta, double var_par,
             int flink, int link_code, int unused,
             double *XtWX_sum, double *XtWr_sum, double *sigma_sum,
             double *beta_out, double *sigma_out)
{
    (void)unused;
    double *xb = (double *)calloc((size_t)n, sizeof(double));
    for (int p = 0; p < P; p++) for (int i = 0; i < n; i++) xb[i] += X[i + (size_t)p * n] * beta[p];
    double *zd = (double *)malloc(sizeof(double) * (size_t)n * m);
    orc_gemm_nn(n, m, Q, Z, n, u, ldu, zd, n);               /* :207 */
    /* mcmlmodel.h:123-130 */
    double nvar_par = 1.0;
    if (flink == 7 || flink == 8) nvar_par *= var_par * var_par;
    else if (flink >= 9 && flink <= 11) nvar_par *= var_par;
    else if (flink == 12) nvar_par *= (1 + var_par);
    double *S1 = (double *)calloc((size_t)P * P, sizeof(double));
    double *S2 = (double *)calloc((size_t)P, sizeof(double));
    double S3 = 0;
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *wu = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < m; i++) {
        double mean = 0;
        for (int j = 0; j < n; j++) {
            double eta = xb[j] + zd[j + (size_t)i * n];
            w[j] = 1 / (orc_dhdmu(eta, flink) * nvar_par);          /* :213 */
            double resid = y[j] - orc_mod_inv(eta, link_code);       /* :214-215 */
            wu[j] = w[j] * orc_detadmu(eta, link_code) * resid;      /* :218-223 */
            zd[j + (size_t)i * n] = resid;
            mean += resid;
        }
        mean /= n;
        double ss = 0;
        for (int j = 0; j < n; j++) { double d = zd[j + (size_t)i * n] - mean; ss += d * d; }
        S3 += sqrt(ss / (n - 1));                                     /* :216 */
        for (int a = 0; a < P; a++) {
            double t = 0;
            for (int j = 0; j < n; j++) t += X[j + (size_t)a * n] * wu[j];
            S2[a] += t;
            for (int b = 0; b < P; b++) {
                double s = 0;
                for (int j = 0; j < n; j++) s += X[j + (size_t)a * n] * w[j] * X[j + (size_t)b * n];
                S1[a + b * P] += s;                                  /* :217 */
            }
        }
    }
    if (XtWX_sum) memcpy(XtWX_sum, S1, sizeof(double) * (size_t)P * P);
    if (XtWr_sum) memcpy(XtWr_sum, S2, sizeof(double) * (size_t)P);
    if (sigma_sum) *sigma_sum = S3;
    int rc = 0;
    if (beta_out) {
        for (int k = 0; k < P * P; k++) S1[k] *= (double)1 / m;      /* :227 */
        rc = inv_spd_small(S1, P);                                   /* :230 */
        for (int a = 0; a < P; a++) {
            double inc = 0;
            for (int b = 0; b < P; b++) inc += S1[a + b * P] * (S2[b] / m);   /* :231-232 */
            beta_out[a] = beta[a] + inc;                             /* :234 */
        }
    }
    if (sigma_out) *sigma_out = S3 / m;                              /* :235 */
    free(xb); free(zd); free(S1); free(S2); free(w); free(wu);
    return rc;
}
