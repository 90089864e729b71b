// glm.h -- per-observation GLM scalar maths on the device.
//   glm_logpdf : maths::log_likelihood (moremaths.h:26-102), log_factorial_approx (:16-24)
//   glm_score  : the vector log_grad hands to ZL' and the scalar applied after it
//                (mcmlmodel.h:169-276)
//   glm_mod_inv / glm_dhdmu / glm_detadmu : glmmrBase maths::mod_inv_func, maths::dhdmu
//                (restated from their call sites mcmloptim.h:214, mcmlmodel.h:122)
//                and maths::detadmu (moremaths.h:118-161)
// Expression order follows the reference (and so the CPU oracle) literally,
// including its constants (3.141593) and its unstable logistic form.
#pragma once
#include <hip/hip_runtime.h>

namespace mcml {

__device__ __forceinline__ double glm_pnorm(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }
__device__ __forceinline__ double glm_dnorm(double x) { return exp(-0.5 * x * x) * 0.39894228040143267794; }

__device__ __forceinline__ double glm_log_factorial_approx(double n)
{
    if (n == 0) return 0;
    return n * log(n) - n + log(n * (1 + 4 * n * (1 + 2 * n))) / 6 + log(3.141593) / 2;
}

__device__ __forceinline__ double glm_logpdf(double y, double mu, double var_par, int flink)
{
    double logl = 0.0;
    switch (flink) {
    case 1: logl = y * mu - exp(mu) - glm_log_factorial_approx(y); break;
    case 2: logl = y * log(mu) - mu - glm_log_factorial_approx(y); break;
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
        if (y == 1) logl = log(glm_pnorm(mu));
        else if (y == 0) logl = log(1 - glm_pnorm(mu));
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

// scalar applied after ZL' s (mcmlmodel.h:235,241,251,257,263)
__host__ __device__ __forceinline__ double glm_score_post(double var_par, int flink)
{
    if (flink == 7 || flink == 8) return 1.0 / (var_par * var_par);
    if (flink >= 9 && flink <= 11) return var_par;
    return 1.0;
}

// digamma(x), x > 0: stands where boost::math::digamma is called (mcmlmodel.h:271).  Recurrence
// psi(x) = psi(x+1) - 1/x up to x >= 10, then ln x - 1/(2x) - sum_k B_2k/(2k x^2k), k = 1..7;
// the same expression order as orc_digamma in oracle/mcml_oracle.c.
__device__ __forceinline__ double glm_digamma(double x)
{
    if (!(x > 0)) return __builtin_nan("");
    double r = 0.0;
    while (x < 10.0) { r = r - 1 / x; x = x + 1; }
    const double i2 = 1 / (x * x);
    double t = 1.0 / 12;
    t = 691.0 / 32760 - t * i2;
    t = 1.0 / 132 - t * i2;
    t = 1.0 / 240 - t * i2;
    t = 1.0 / 252 - t * i2;
    t = 1.0 / 120 - t * i2;
    t = 1.0 / 12 - t * i2;
    return r + (log(x) - 0.5 / x - t * i2);
}

// beta/logit (mcmlmodel.h:266-275): the only case where var_par enters the vector itself (the others apply it
// after the product, glm_score_post).  Literally: the reference's second line reads the UPDATED mu(i) = p, so
// the leading factor is p/(1+exp(p)).  Kept out of glm_score's switch so that the hot GEMM epilogues of the
// other eleven cases do not carry the digamma loops (they cost the forward band kernel 13 VGPRs and a scratch
// frame); callers select it at compile time (EpiForwardT<12>).
__device__ __forceinline__ double glm_score_beta(double y, double mu, double var_par)
{
    const double p = exp(mu) / (exp(mu) + 1);
    return (p / (1 + exp(p))) * var_par *
           (log(y) - log(1 - y) - glm_digamma(p * var_par) + glm_digamma((1 - p) * var_par));
}

__device__ __forceinline__ double glm_score(double y, double mu, int flink)
{
    switch (flink) {
    case 1: return y - exp(mu);
    case 2: return y * (1 / mu) - 1;
    case 3: { double t = exp(mu); t = t + 1; t = 1 / t; t = t + y; return t - 1; }
    case 4: if (y == 1) return 1; else if (y == 0) return exp(mu) / (1 - exp(mu)); return mu;
    case 5: if (y == 1) return 1 / mu; else if (y == 0) return -1 / (1 - mu); return mu;
    case 6:
        if (y == 1) return glm_dnorm(mu) / glm_pnorm(mu);
        else if (y == 0) return -1.0 * glm_dnorm(mu) / (1 - glm_pnorm(mu));
        return mu;
    case 7: case 8: return y - mu;
    case 9: return y * exp(-1.0 * mu) - 1;
    case 10: return (1 / mu) - y;
    case 11: { double im = 1 / mu; return y * im * im - im; }
    }
    return 0.0;
}

__device__ __forceinline__ double glm_mod_inv(double eta, int link_code)
{
    switch (link_code) {
    case 1: return exp(eta);
    case 2: return eta;
    case 3: return exp(eta) / (1 + exp(eta));
    case 4: return glm_pnorm(eta);
    case 5: return 1 / eta;
    }
    return eta;
}

__device__ __forceinline__ double glm_dhdmu(double eta, int flink)
{
    double p;
    switch (flink) {
    case 1: return exp(-1.0 * eta);
    case 2: return exp(eta);
    case 3: p = glm_mod_inv(eta, 3); return 1 / (p * (1.0 - p));
    case 4: p = glm_mod_inv(eta, 3); return (1.0 - p) / p;
    case 5: p = glm_mod_inv(eta, 3); return p * (1.0 - p);
    case 6: p = glm_pnorm(eta); return (p * (1 - p)) / glm_dnorm(eta);
    case 7: return 1.0;
    case 8: return 1 / exp(eta);
    case 9: return 1.0;
    case 10: return 1 / (eta * eta);
    case 11: return eta * eta;
    case 12: p = glm_mod_inv(eta, 3); return 1 / (p * (1.0 - p));
    }
    return 1.0;
}

__device__ __forceinline__ double glm_detadmu(double eta, int link_code)
{
    double p;
    switch (link_code) {
    case 1: return exp(-1.0 * eta);
    case 2: return 1.0;
    case 3: p = glm_mod_inv(eta, 3); return 1 / (p * (1.0 - p));
    case 4: return 1 / glm_dnorm(eta);
    case 5: return -1.0 * eta * eta;
    }
    return 1.0;
}

}  // namespace mcml
