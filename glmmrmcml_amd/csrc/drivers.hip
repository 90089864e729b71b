// drivers.hip -- the algorithm objects and drivers above the kernels:
//   objective functors D_/L_/F_likelihood        likelihood.h:31-110
//   mcmloptim::{d_optim,l_optim,f_optim,mcnr,f_hess}   mcmloptim.h:56-113,198-236,333-355
//   drivers mcml_full, mcmc_sample                src/mcml_full.cpp:41-148,314-338
//   mcml_optim, mcml_simlik, mcml_hess, aic_mcml  src/mcml_optim.cpp:35-392
// Host code; every objective evaluation runs on the device with u, Z, X, y
// resident and returns one scalar.
//
// Deliberate departures from the reference:
//   D4  the importance ratio exp(ll+logl)/exp(denom) underflows to NaN for any
//       realistic n (likelihood.h:101-105): evaluated as -(ll + logl - denom).
//   D3  mcnr's OpenMP race on W_/zu_: the statistics here are those of the
//       serial loop.
//   D11 the sampler is re-initialised every MCML iteration exactly as the
//       reference does (mhmcmc.h:127), with seeds derived from (seed, iteration).
#include "../../include/glmmr_mcml_c.h"
#include "ctx.h"
#include "optim.h"
#include "trace.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <random>

namespace mcml {

int hmc_sample(Ctx& c, const double* beta, double var_par, const glmmr_mcml_hmc_opts* o, uint64_t seed,
               uint32_t iter_idx, const double* inj_init, const double* inj_mom, uint8_t* flags_out,
               double* probs_out, glmmr_mcml_hmc_diag* diag, int* ncols_out);

static bool is_gaussian(int flink) { return flink == 7 || flink == 8; }
// family_=="gaussian"||"Gamma"||"beta" (likelihood.h:61,95): the map keys are
// lower-case "gamma" (mcmlmodel.h:83-85), so "Gamma" never reaches this point
static bool has_var_par(int flink) { return flink == 7 || flink == 8 || flink == 12; }

// ---- device-backed scalar evaluations (all-reduced over ranks) ----
static int eval_loglik(Ctx& c, const double* beta, double var_par, double* ll)
{
    MCML_TRY(model_update_beta(c, beta));
    double s = 0;
    MCML_TRY(model_loglik_sum(c, var_par, &s));
    double tot[2] = {s, (double)c.niter};
    MCML_TRY(allreduce_host(c, tot, 2));
    *ll = tot[0] / tot[1];
    return MCML_OK;
}

// A theta at which some block is not positive definite has no likelihood: the
// reference would return NaN from its unchecked Cholesky; the optimiser is told
// "infinitely bad" so that it backs away instead of aborting the fit.
static int eval_mvn(Ctx& c, const double* theta, double* logl)
{
    double s = 0;
    int rc = mvn_loglik_sum(c, theta, &s);
    if (rc == MCML_ENOTPD) { *logl = -HUGE_VAL; return MCML_OK; }
    MCML_TRY(rc);
    double tot[2] = {s, (double)c.mcols};
    MCML_TRY(allreduce_host(c, tot, 2));
    *logl = tot[0] / tot[1];
    if (c.theta_log_on) { c.theta_log.insert(c.theta_log.end(), theta, theta + c.cov.npar); c.theta_log.push_back(*logl); }
    return MCML_OK;
}

// mcmloptim<T> (mcmloptim.h:16-369)
struct McmlOptim {
    Ctx& c;
    int P, R;
    std::vector<double> beta, theta, cov_par_fix;
    double sigma;
    int trace, maxfun;
    int theta_batch = 0;      // candidates per rank and round of the sharded theta-step (glmmr_mcml_ext.theta_batch)
    double model_var_par;     // M_->var_par_

    McmlOptim(Ctx& ctx, const double* start, int trace_, int maxfun_, double var_par0)
        : c(ctx), P(ctx.P), R(ctx.cov.npar), trace(trace_), maxfun(maxfun_), model_var_par(var_par0)
    {
        beta.assign(start, start + P);
        theta.assign(start + P, start + P + R);
        cov_par_fix = theta;
        sigma = is_gaussian(c.flink) ? start[P + R] : 0;         // mcmloptim.h:30
    }

    BobyqaOpts bopts() const { BobyqaOpts o; o.iprint = trace; if (maxfun > 0) o.maxfun = maxfun; return o; }

    // d_optim of a chain-sharded job (no reference counterpart: the reference is one process).  The
    // factorisation of D(theta) does not shard -- replicated, it was 43 % of a rank's step at 8 GPUs -- but the
    // optimiser's EVALUATIONS do: the sample columns are all-gathered once per iteration (every rank then holds all of
    // u), the optimiser proposes `width` candidate thetas per round (optim.h bobyqa_batch), candidate j is evaluated on
    // ALL columns by rank j mod world, and one all-reduce per round carries the values (each slot is nonzero on one
    // rank).  Same objective (mcmloptim.h:56-68, likelihood.h:31-46, mcmldmatrix.h:23-41), sequential depth = rounds.
    // The search runs over log(theta): the parameters are positive scales (lower bound 1e-6, mcmloptim.h:35-38) and
    // the MVN log-likelihood is far closer to a quadratic there, which is what a round of simultaneous points needs.
    // Every rank runs the same deterministic optimiser on the same values: no further agreement is needed.
    int d_optim_sharded(int width)
    {
        MCML_TRY(gather_samples(c));
        const int wr = comm_world(c), mall = c.mcols * wr;
        batch_objective_fn fb = [&](const std::vector<std::vector<double>>& Zs, std::vector<double>* F) -> int {
            const int nc = (int)Zs.size();
            std::vector<double> vals(nc, 0.0);
            const bool emu = c.world <= 1 && !c.comm && c.emu_world > 1;
            int first_rc = MCML_OK;
            // this rank's candidates of the round, factorised side by side (mvn.hip mvn_loglik_batch).  Rank emulation in
            // recording mode evaluates every rank's share here, one rank's share at a time -- exactly the batches the
            // ranks of the real job would run, so that the replay reproduces the recorded values to the bit
            const int r_lo = (emu && c.emu_mode == 1) ? 0 : c.rank, r_hi = (emu && c.emu_mode == 1) ? wr : c.rank + 1;
            for (int rr = r_lo; rr < r_hi; ++rr) {
                std::vector<int> own;
                for (int j = 0; j < nc; ++j) if ((j % wr) == rr) own.push_back(j);
                if (own.empty()) continue;
                std::vector<double> ths((size_t)R * own.size()), sums(own.size(), 0.0);
                std::vector<int> rcs(own.size(), 0);
                for (size_t q = 0; q < own.size(); ++q)
                    for (int i = 0; i < R; ++i) ths[q * R + i] = std::exp(Zs[own[q]][i]);
                const int brc = mvn_loglik_batch(c, ths.data(), (int)own.size(), c.Uall.d(), c.Uall.ld, mall, sums.data(), rcs.data());
                for (size_t q = 0; q < own.size(); ++q) {
                    const int j = own[q];
                    const int rc = (rcs[q] == MCML_OK && brc != MCML_OK) ? brc : rcs[q];
                    if (rc != MCML_OK && getenv("GLMMR_MCML_BOBYQA_TRACE")) fprintf(stderr, "theta-step: candidate %d (theta %.6g %.6g) rc %d brc %d: %s\n", j, ths[q * R], R > 1 ? ths[q * R + 1] : 0.0, rcs[q], brc, last_error());
                    if (rc == MCML_ENOTPD) vals[j] = HUGE_VAL;               // as eval_mvn: infinitely bad, not an error
                    else if (rc != MCML_OK) { vals[j] = NAN; if (first_rc == MCML_OK) first_rc = rc; }
                    else vals[j] = -1 * (sums[q] / mall);
                    if (rr == c.rank) ++c.theta_evals_own;
                    if (c.theta_log_on && rc == MCML_OK) { c.theta_log.insert(c.theta_log.end(), ths.begin() + q * R, ths.begin() + (q + 1) * R); c.theta_log.push_back(sums[q] / mall); }
                }
            }
            c.theta_rounds += 1; c.theta_evals_all += nc;
            if (!emu) MCML_TRY(allreduce_host(c, vals.data(), nc));      // every slot is zero on all ranks but its owner
            else if (c.emu_mode == 1) c.emu_trace.push_back(vals);
            else {
                MCML_REQUIRE(c.emu_pos < c.emu_trace.size() && (int)c.emu_trace[c.emu_pos].size() == nc,
                             "rank emulation: replay ran past its record (round %zu)", c.emu_pos);
                const std::vector<double>& rec = c.emu_trace[c.emu_pos++];
                for (int j = 0; j < nc; ++j) {
                    if ((j % wr) == c.rank) MCML_REQUIRE(vals[j] == rec[j] || (vals[j] != vals[j]), "rank emulation: replayed value differs from the recorded one (%.17g vs %.17g)", vals[j], rec[j]);
                    else vals[j] = rec[j];
                }
            }
            if (first_rc != MCML_OK) return first_rc;
            for (double v : vals)         // a rank that failed put NaN in its slot: every rank stops here, together
                if (v != v) { set_error("theta-step: a rank failed to evaluate its candidate"); return MCML_EHIP; }
            *F = vals;
            return MCML_OK; };
        std::vector<double> z(R), lo(R, std::log(1e-6)), up(R, HUGE_VAL);
        for (int i = 0; i < R; ++i) z[i] = std::log(std::max(theta[i], 1e-6));
        BobyqaOpts o = bopts();
        o.rhobeg = 0.25; o.rhoend = 1e-7;
        BobyqaResult r;
        MCML_TRY(bobyqa_batch(fb, z, lo, up, o, width, &r));
        for (int i = 0; i < R; ++i) theta[i] = std::exp(r.x[i]);
        return MCML_OK;
    }

    // d_optim (mcmloptim.h:56-68), D_likelihood (likelihood.h:40-45)
    int d_optim()
    {
        static const bool shard = !(getenv("GLMMR_MCML_THETA_SHARD") && !strcmp(getenv("GLMMR_MCML_THETA_SHARD"), "0"));
        const int wr = comm_world(c);
        // candidates per rank and round.  A model whose D is large dense blocks only (the geospatial configs) evaluates
        // a round's candidates in ONE pass of the factorisation's schedule (mvn.hip mvn_loglik_batch): there the batch
        // schedule is the default even for a single process, 8 candidates per round (batch_width());
        // glmmr_mcml_ext.theta_batch wins, 1 = the reference's sequential BOBYQA.
        // a sharded job: rounds of about eight candidates in all -- 8 / world per rank for a dense-block model (at 2 ranks
        // four candidates per rank factorised side by side: 6 rounds of 8.7 ms instead of 20 of 3.6 ms), one otherwise
        const bool dense_only = c.maxdim_large > 0 && c.n_small == 0 && c.n_diag_rows == 0;
        const int k = wr > 1 ? (theta_batch > 0 ? theta_batch : dense_only ? std::max(1, 8 / wr) : 1) : batch_width();
        if ((wr > 1 && shard) || k > 1) return d_optim_sharded(wr * std::max(1, k));
        objective_fn f = [&](const std::vector<double>& par, double* v) {
            double logl; MCML_TRY(eval_mvn(c, par.data(), &logl)); *v = -1 * logl; return (int)MCML_OK; };
        std::vector<double> lo(R, 1e-6), up(R, HUGE_VAL);
        BobyqaResult r;
        MCML_TRY(bobyqa(f, theta, lo, up, bopts(), &r));
        theta = r.x;
        return MCML_OK;
    }

    // l_optim (mcmloptim.h:71-88), L_likelihood (likelihood.h:57-64)
    int l_optim()
    {
        if (c.flink == 12) { set_error("l_optim: beta family is not built (reference reads par[P] out of range, defect D8)"); return MCML_EUNSUPPORTED; }
        const bool g = is_gaussian(c.flink);
        objective_fn f = [&](const std::vector<double>& par, double* v) {
            model_var_par = has_var_par(c.flink) ? par[P] : 0.0;   // fix_var_ = false, fix_var_par_ = 0
            double ll; MCML_TRY(eval_loglik(c, par.data(), model_var_par, &ll)); *v = -1 * ll; return (int)MCML_OK; };
        std::vector<double> x = beta, lo(P, -HUGE_VAL), up(P, HUGE_VAL);
        if (g) { x.push_back(sigma); lo.push_back(0.0); up.push_back(HUGE_VAL); }
        BobyqaResult r;
        MCML_TRY(bobyqa(f, x, lo, up, bopts(), &r));
        beta.assign(r.x.begin(), r.x.begin() + P);
        if (g) sigma = r.x[P];
        return MCML_OK;
    }

    // mcnr (mcmloptim.h:198-236)
    int mcnr()
    {
        MCML_TRY(model_update_beta(c, beta.data()));
        std::vector<double> st((size_t)P * P + P + 2), nb(P);
        MCML_TRY(model_mcnr_stats(c, model_var_par, st.data()));
        double s = 0;
        MCML_TRY(mcnr_finish(P, st.data(), beta.data(), nb.data(), &s));
        beta = nb; sigma = s;
        return MCML_OK;
    }

    // F_likelihood::operator() (likelihood.h:88-109) with fix_var = true
    // memo: the MVN term depends on theta alone.  The central differences of f_hess move one or two coordinates at a time,
    // so about half of its 4 (P + R)^2 points repeat a theta already factorised (19 distinct of 36 at P = 1, R = 2): those
    // reuse the value (same bits) instead of another build + Cholesky + solve.
    typedef std::map<std::vector<double>, double> ThetaMemo;
    objective_fn make_F(bool importance, double fix_var_par, double denomD, bool memoise = false,
                        std::shared_ptr<ThetaMemo> memo = std::make_shared<ThetaMemo>())
    {
        return [this, importance, fix_var_par, denomD, memoise, memo](const std::vector<double>& par, double* v) {
            model_var_par = fix_var_par;
            double ll, logl;
            MCML_TRY(eval_loglik(c, par.data(), model_var_par, &ll));
            if (memoise) {
                const std::vector<double> th(par.begin() + P, par.begin() + P + R);
                auto it = memo->find(th);
                if (it != memo->end()) logl = it->second;
                else { MCML_TRY(eval_mvn(c, th.data(), &logl)); (*memo)[th] = logl; }
            } else
            MCML_TRY(eval_mvn(c, par.data() + P, &logl));
            *v = importance ? -1.0 * (ll + logl - denomD) : -1.0 * (ll + logl);
            return (int)MCML_OK; };
    }

    // candidates per round of the batch schedule for this model in a single process (see d_optim): 8 when D consists of
    // large dense blocks only, else 1 = the sequential optimiser; glmmr_mcml_ext.theta_batch / GLMMR_MCML_THETA_BATCH override
    int batch_width() const
    {
        if (theta_batch > 0) return theta_batch;
        static const int envk = getenv("GLMMR_MCML_THETA_BATCH") ? atoi(getenv("GLMMR_MCML_THETA_BATCH")) : 0;
        if (envk > 0) return envk;
        return (c.maxdim_large > 0 && c.n_small == 0 && c.n_diag_rows == 0) ? 8 : 1;
    }

    // f_optim on the batch schedule (single process, dense-block models): a round's candidates (beta, log theta[, sigma])
    // get their MVN terms from ONE pass of the factorisation (mvn_loglik_batch) and their log-likelihood terms one after
    // the other (cheap: n m log-pdf evaluations on the cached Z u).  Same objective (likelihood.h:88-109), same optimum.
    // (For the gaussian family the reference lets BOBYQA carry sigma as a variable the objective never reads,
    // mcmloptim.h:102-105 -- a flat direction whose final value is whatever the trust region left it at; here it simply stays
    // at its start value.)
    int f_optim_batch(int width, double denomD)
    {
        const double fix_var_par = sigma;
        batch_objective_fn fb = [&](const std::vector<std::vector<double>>& Zs, std::vector<double>* F) -> int {
            const int nc = (int)Zs.size();
            std::vector<double> ths((size_t)R * nc), sums(nc, 0.0);
            std::vector<int> rcs(nc, 0);
            for (int j = 0; j < nc; ++j) for (int i = 0; i < R; ++i) ths[(size_t)j * R + i] = std::exp(Zs[j][P + i]);
            MCML_TRY(mvn_loglik_batch(c, ths.data(), nc, c.U.d(), c.U.ld, c.mcols, sums.data(), rcs.data()));
            F->assign(nc, 0.0);
            for (int j = 0; j < nc; ++j) {
                model_var_par = fix_var_par;
                double ll = 0;
                MCML_TRY(eval_loglik(c, Zs[j].data(), model_var_par, &ll));
                const double logl = rcs[j] == MCML_OK ? sums[j] / c.mcols : -HUGE_VAL;
                (*F)[j] = -1.0 * (ll + logl - denomD);
            }
            return MCML_OK; };
        std::vector<double> x = beta, lo(P, -HUGE_VAL), up;
        for (int i = 0; i < R; ++i) { x.push_back(std::log(std::max(theta[i], 1e-6))); lo.push_back(std::log(1e-6)); }
        up.assign(x.size(), HUGE_VAL);
        BobyqaOpts o = bopts();
        o.rhobeg = 0.25; o.rhoend = 1e-7;
        BobyqaResult r;
        MCML_TRY(bobyqa_batch(fb, x, lo, up, o, width, &r));
        beta.assign(r.x.begin(), r.x.begin() + P);
        for (int i = 0; i < R; ++i) theta[i] = std::exp(r.x[P + i]);
        return MCML_OK;
    }

    // f_optim (mcmloptim.h:91-113)
    int f_optim()
    {
        const bool g = is_gaussian(c.flink);
        double denomD = 0;
        MCML_TRY(eval_mvn(c, cov_par_fix.data(), &denomD));      // constant in the parameters
        if (comm_world(c) == 1 && batch_width() > 1) return f_optim_batch(batch_width(), denomD);
        objective_fn f = make_F(true, sigma, denomD);
        std::vector<double> x = beta, lo(P, -HUGE_VAL), up;
        for (int i = 0; i < R; ++i) { x.push_back(theta[i]); lo.push_back(1e-6); }
        if (g) { x.push_back(sigma); lo.push_back(0.0); }
        up.assign(x.size(), HUGE_VAL);
        BobyqaResult r;
        MCML_TRY(bobyqa(f, x, lo, up, bopts(), &r));
        beta.assign(r.x.begin(), r.x.begin() + P);
        theta.assign(r.x.begin() + P, r.x.begin() + P + R);
        if (g) sigma = r.x[P + R];
        return MCML_OK;
    }

    // f_hess (mcmloptim.h:333-355)
    int f_hess(double tol, double* H)
    {
        auto memo = std::make_shared<ThetaMemo>();
        objective_fn f = make_F(false, sigma, 0.0, true, memo);
        const int nv = P + R;
        std::vector<double> x = beta, lo(P, -HUGE_VAL), up(nv, HUGE_VAL), nd(nv, tol), h;
        for (int i = 0; i < R; ++i) { x.push_back(theta[i]); lo.push_back(1e-6); }
        // The points of the finite differences do not depend on the values: a dry pass lists the thetas they will ask
        // for, and (single process) those are factorised side by side, a round at a time, before the real pass runs
        if (comm_world(c) == 1 && c.maxdim_large > 0) {
            std::vector<std::vector<double>> want;
            objective_fn dry = [&](const std::vector<double>& par, double* v) {
                std::vector<double> th(par.begin() + P, par.begin() + P + R);
                if (std::find(want.begin(), want.end(), th) == want.end()) want.push_back(th);
                *v = 0.0; return (int)MCML_OK; };
            std::vector<double> hdry;
            MCML_TRY(fd_hessian(dry, x, nd, true, lo, up, &hdry));
            std::vector<double> ths((size_t)R * want.size()), sums(want.size(), 0.0);
            std::vector<int> rcs(want.size(), 0);
            for (size_t q = 0; q < want.size(); ++q) for (int i = 0; i < R; ++i) ths[q * R + i] = want[q][i];
            MCML_TRY(mvn_loglik_batch(c, ths.data(), (int)want.size(), c.U.d(), c.U.ld, c.mcols, sums.data(), rcs.data()));
            for (size_t q = 0; q < want.size(); ++q)
                (*memo)[want[q]] = rcs[q] == MCML_OK ? sums[q] / c.mcols : -HUGE_VAL;       // as eval_mvn
        }
        MCML_TRY(fd_hessian(f, x, nd, true, lo, up, &h));
        memcpy(H, h.data(), sizeof(double) * (size_t)nv * nv);
        return MCML_OK;
    }
};

static uint64_t default_seed(const glmmr_mcml_ext* e)
{
    if (e && e->seed) return e->seed;
    std::random_device rd;            // what the reference does (mhmcmc.h:55)
    return ((uint64_t)rd() << 32) | rd();
}

// ---------------------------------------------------------------- ctx-level drivers
int drv_optim(Ctx& c, const double* start, int nstart, int trace, int mcnr, const glmmr_mcml_ext* e,
              double* beta, double* theta, double* sigma)
{
    const int P = c.P, R = c.cov.npar;
    MCML_REQUIRE(start && nstart >= P + R + (is_gaussian(c.flink) ? 1 : 0), "start has %d values, need %d", nstart, P + R + (is_gaussian(c.flink) ? 1 : 0));
    MCML_REQUIRE(c.mcols > 0, "no samples u set");
    McmlOptim mc(c, start, trace, e ? e->maxfun : 0, 1.0);      // model built with var_par = 1 (mcml_optim.cpp:51)
    mc.theta_batch = e ? e->theta_batch : 0;
    {
        PhaseRange r("mcml:beta-step");
        if (!mcnr) MCML_TRY(mc.l_optim()); else MCML_TRY(mc.mcnr());
    }
    PhaseRange r("mcml:theta-step");
    MCML_TRY(mc.d_optim());
    memcpy(beta, mc.beta.data(), sizeof(double) * P);
    memcpy(theta, mc.theta.data(), sizeof(double) * R);
    *sigma = mc.sigma;
    return MCML_OK;
}

int drv_simlik(Ctx& c, const double* start, int nstart, int trace, const glmmr_mcml_ext* e, double* beta,
               double* theta, double* sigma)
{
    const int P = c.P, R = c.cov.npar;
    MCML_REQUIRE(start && nstart >= P + R + (is_gaussian(c.flink) ? 1 : 0), "start too short");
    MCML_REQUIRE(c.mcols > 0, "no samples u set");
    McmlOptim mc(c, start, trace, e ? e->maxfun : 0, 1.0);
    mc.theta_batch = e ? e->theta_batch : 0;
    MCML_TRY(mc.f_optim());
    memcpy(beta, mc.beta.data(), sizeof(double) * P);
    memcpy(theta, mc.theta.data(), sizeof(double) * R);
    *sigma = mc.sigma;
    return MCML_OK;
}

int drv_hess(Ctx& c, const double* start, int nstart, double tol, int trace, double* H)
{
    const int P = c.P, R = c.cov.npar;
    MCML_REQUIRE(start && nstart >= P + R + (is_gaussian(c.flink) ? 1 : 0), "start too short");
    MCML_REQUIRE(c.mcols > 0, "no samples u set");
    McmlOptim mc(c, start, trace, 0, 1.0);
    return mc.f_hess(tol, H);
}

// aic_mcml (mcml_optim.cpp:356-392)
int drv_aic(Ctx& c, const double* beta_par, int nbeta, const double* cov_par, int ncov, double* out)
{
    const int P = c.P;
    const bool var = (c.flink == 7 || c.flink == 8 || c.flink == 12);
    MCML_REQUIRE(nbeta >= P + (var ? 1 : 0) && ncov >= c.cov.npar, "aic_mcml: parameter vectors too short");
    MCML_REQUIRE(c.mcols > 0, "no samples u set");
    const double var_par = var ? beta_par[P] : 0;
    const int dof = nbeta + ncov;
    double dmv, ll;
    MCML_TRY(eval_mvn(c, cov_par, &dmv));
    MCML_TRY(eval_loglik(c, beta_par, var_par, &ll));
    *out = (-2 * (ll + dmv) + 2 * dof);
    return MCML_OK;
}

// mcml_full (mcml_full.cpp:41-148)
int drv_full(Ctx& c, const double* start, int nstart, int mcnr, int m, int maxiter, int warmup, double tol,
             int verbose, double lambda, int trace, int refresh, int maxsteps, double target_accept,
             const glmmr_mcml_ext* e, double* beta_out, double* theta_out, double* sigma_out,
             int* converged_out, int* iters_out, glmmr_mcml_hmc_diag* last_diag)
{
    (void)refresh;
    const int P = c.P, R = c.cov.npar;
    MCML_REQUIRE(start && nstart >= P + R + 1, "mcml_full: start needs c(beta, theta, sigma|1) = %d values", P + R + 1);
    MCML_REQUIRE(m > 0 && maxiter >= 1, "mcml_full: bad m / maxiter");
    std::vector<double> theta(start + P, start + P + R), beta(start, start + P);
    double var_par = is_gaussian(c.flink) ? start[nstart - 1] : 1;          // :65
    const uint64_t seed = default_seed(e);
    const int chains = (e && e->chains > 0) ? e->chains : 1;
    MCML_TRY(mvn_gen_L(c, theta.data(), true));                               // :66-68
    MCML_TRY(model_update_beta(c, beta.data()));
    MCML_TRY(model_update_L(c));
    McmlOptim mc(c, start, trace, e ? e->maxfun : 0, var_par);                // :71
    mc.theta_batch = e ? e->theta_batch : 0;
    double maxdiff = 1;
    int iter = 1;
    bool converged = false;
    glmmr_mcml_hmc_opts ho{};
    ho.warmup = warmup; ho.nsamp = m; ho.adapt = 100; ho.lambda = lambda; ho.max_steps = maxsteps;
    ho.target_accept = target_accept; ho.chains = chains; ho.chain_offset = c.rank * chains;
    while (maxdiff > tol && iter <= maxiter) {                                 // :83
        glmmr_mcml_hmc_diag dg{};
        {
            PhaseRange r("mcml:sample");
            MCML_TRY(hmc_sample(c, beta.data(), var_par, &ho, seed, (uint32_t)iter, nullptr, nullptr, nullptr,
                                nullptr, &dg, nullptr));                       // :92
        }
        if (last_diag) *last_diag = dg;
        mc.model_var_par = var_par;
        {
            PhaseRange r("mcml:beta-step");
            if (!mcnr) MCML_TRY(mc.l_optim()); else MCML_TRY(mc.mcnr());       // :95-99
        }
        {
            PhaseRange r("mcml:theta-step");
            MCML_TRY(mc.d_optim());                                            // :101
        }
        const std::vector<double>& nb = mc.beta; const std::vector<double>& nt = mc.theta;
        double new_var_par = 1;
        if (is_gaussian(c.flink)) new_var_par = mc.sigma;                      // :105
        else new_var_par = var_par;
        maxdiff = 0;
        for (int i = 0; i < P; ++i) maxdiff = std::max(maxdiff, std::fabs(beta[i] - nb[i]));
        for (int i = 0; i < R; ++i) maxdiff = std::max(maxdiff, std::fabs(theta[i] - nt[i]));
        maxdiff = std::max(maxdiff, std::fabs(var_par - new_var_par));
        if (maxdiff < tol) converged = true;                                   // :113
        beta = nb; theta = nt; var_par = new_var_par;
        if (!converged) {                                                      // :119-126
            PhaseRange r("mcml:refresh");
            MCML_TRY(mvn_gen_L(c, theta.data(), true));
            MCML_TRY(model_update_beta(c, beta.data()));
            MCML_TRY(model_update_L(c));
        }
        if (verbose && c.rank == 0) {
            printf("Iter %d  beta:", iter);
            for (double b : beta) printf(" %.6g", b);
            printf("  theta:");
            for (double t : theta) printf(" %.6g", t);
            if (is_gaussian(c.flink)) printf("  sigma: %.6g", var_par);
            printf("  max diff %.4g  accept %.3f%s\n", maxdiff, dg.accept_rate, converged ? "  CONVERGED" : "");
            fflush(stdout);
        }
        ++iter;
    }
    memcpy(beta_out, beta.data(), sizeof(double) * P);
    memcpy(theta_out, theta.data(), sizeof(double) * R);
    *sigma_out = var_par;
    if (converged_out) *converged_out = converged ? 1 : 0;
    if (iters_out) *iters_out = iter - 1;
    return MCML_OK;
}

}  // namespace mcml

extern "C" int glmmr_mcml_dbg_phase_ms(int enable, int reset, double* out8)
{
    mcml::PhaseClock& pc = mcml::phase_clock();
    std::lock_guard<std::mutex> g(pc.mu);
    if (out8) for (int i = 0; i < 4; ++i) { out8[i] = pc.ms[i]; out8[4 + i] = (double)pc.n[i]; }
    if (reset) for (int i = 0; i < 4; ++i) { pc.ms[i] = 0; pc.n[i] = 0; }
    pc.on.store(enable != 0, std::memory_order_relaxed);
    return 0;
}
