// cabi.hip -- the extern "C" boundary (include/glmmr_mcml_c.h).
#include "../../include/glmmr_mcml_c.h"
#include "ctx.h"
#include <random>

using namespace mcml;


namespace mcml {
int hmc_sample(Ctx& c, const double* beta, double var_par, const glmmr_mcml_hmc_opts* o, uint64_t seed,
               uint32_t iter_idx, const double* inj_init, const double* inj_mom, uint8_t* flags_out,
               double* probs_out, glmmr_mcml_hmc_diag* diag, int* ncols_out);
int nuts_sample(Ctx& c, const double* beta, double var_par, const glmmr_mcml_nuts_opts* o, uint64_t seed, uint32_t iter_idx,
                int* depth_out, int* nleap_out, double* eps_out, double* accept_out, glmmr_mcml_nuts_diag* diag,
                int* ncols_out);
}

static int flink_of(const char* family, const char* link)
{
    // mcmlmodel.h:74-87 string_to_case
    static const char* tab[12][2] = {
        {"poisson", "log"}, {"poisson", "identity"}, {"binomial", "logit"},
        {"binomial", "log"}, {"binomial", "identity"}, {"binomial", "probit"},
        {"gaussian", "identity"}, {"gaussian", "log"}, {"gamma", "log"},
        {"gamma", "inverse"}, {"gamma", "identity"}, {"beta", "logit"}};
    for (int i = 0; i < 12; i++)
        if (!strcmp(family, tab[i][0]) && !strcmp(link, tab[i][1])) return i + 1;
    return 0;
}

static int link_code_of(const char* link)
{
    // moremaths.h:123-129
    if (!strcmp(link, "log")) return 1;
    if (!strcmp(link, "identity")) return 2;
    if (!strcmp(link, "logit")) return 3;
    if (!strcmp(link, "probit")) return 4;
    if (!strcmp(link, "inverse")) return 5;
    return 0;
}

static int require_device(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: libglmmr_mcml_hip has no CPU fallback");
        return MCML_ENODEVICE;
    }
    MCML_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    MCML_HIP(hipSetDevice(device));
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_create(const glmmr_mcml_problem* p, const glmmr_mcml_dev_opts* o,
                                     glmmr_mcml_ctx** out)
{
    MCML_REQUIRE(p && out, "ctx_create: null argument");
    *out = nullptr;
    MCML_TRY(require_device(o ? o->device : 0));
    auto* h = new (std::nothrow) glmmr_mcml_ctx();
    MCML_REQUIRE(h, "out of host memory");
    Ctx& c = h->c;
    auto fail = [&](int rc) { delete h; return rc; };
    c.device = o ? o->device : 0;
    if (o && o->stream) { c.stream = (hipStream_t)o->stream; c.own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); return fail(MCML_EHIP); }
        c.own_stream = true;
    }
    c.rank = o ? o->rank : 0;
    c.world = (o && o->world > 0) ? o->world : 1;
    c.reduce = o ? (reduce_fn)o->reduce : nullptr;
    c.reduce_user = o ? o->reduce_user : nullptr;
    int rc = MCML_OK;
    if (p->cov) {
        rc = c.cov.parse(p->cov, p->cov_rows, p->data, p->data_len, p->eff_range, p->eff_len);
        if (rc) return fail(rc);
    } else {
        // no covariance description (export mcmc_sample receives L itself, mcml_full.cpp:315)
        if (p->Q <= 0) { set_error("ctx_create: neither cov nor Q given"); return fail(MCML_EINVAL); }
        c.cov.N = p->Q;
    }
    c.Q = c.cov.N;
    if (p->Q > 0 && p->Q != c.cov.N) {
        set_error("Z has %d columns but the covariance blocks sum to %d", p->Q, c.cov.N);
        return fail(MCML_EINVAL);
    }
    c.n = p->n; c.P = p->P;
    if (p->n > 0) {
        if (!(p->Z && p->X && p->y && p->family && p->link && p->P > 0)) {
            set_error("ctx_create: n > 0 needs Z, X, y, family, link");
            return fail(MCML_EINVAL);
        }
        c.flink = flink_of(p->family, p->link);
        c.link_code = link_code_of(p->link);
        if (!c.flink) {
            set_error("family '%s' with link '%s' is not a known combination", p->family, p->link);
            return fail(MCML_EUNSUPPORTED);
        }
        if ((rc = model_setup(c, p->Z, p->X, p->y))) return fail(rc);
    }
    if ((rc = mvn_setup(c))) return fail(rc);
    *out = h;
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_destroy(glmmr_mcml_ctx* h)
{
    if (!h) return MCML_OK;
    (void)hipSetDevice(h->c.device);
    if (h->c.stream) (void)hipStreamSynchronize(h->c.stream);
    hipStream_t s = h->c.own_stream ? h->c.stream : nullptr;
    delete h;
    if (s) (void)hipStreamDestroy(s);
    return MCML_OK;
}

extern "C" int glmmr_mcml_set_u(glmmr_mcml_ctx* h, const double* u, int Q, int ncols, int niter)
{
    MCML_REQUIRE(h && u, "set_u: null argument");
    Ctx& c = h->c;
    MCML_REQUIRE(Q == c.Q && ncols > 0 && niter > 0 && niter <= ncols, "set_u: bad shape %d x %d (niter %d)", Q, ncols, niter);
    MCML_HIP(hipSetDevice(c.device));
    MCML_TRY(upload_matrix(c.U, u, Q, ncols, Q, c.stream));
    c.mcols = ncols; c.niter = niter;
    c.m_global = ncols; c.niter_global = niter;
    c.zu_valid = false; c.uall_valid = false;
    return c.sync();
}

extern "C" int glmmr_mcml_get_u(glmmr_mcml_ctx* h, double* u, int ldu)
{
    MCML_REQUIRE(h && u, "get_u: null argument");
    Ctx& c = h->c;
    MCML_REQUIRE(c.mcols > 0 && ldu >= c.Q, "get_u: no samples / bad ldu");
    MCML_HIP(hipSetDevice(c.device));
    return download_matrix(u, ldu, c.U.d(), c.U.ld, c.Q, c.mcols, c.stream);
}

extern "C" int glmmr_mcml_get_u_all(glmmr_mcml_ctx* h, double* u, int ldu, int* ncols_out)
{
    MCML_REQUIRE(h && u, "get_u_all: null argument");
    Ctx& c = h->c;
    MCML_REQUIRE(c.mcols > 0 && ldu >= c.Q, "get_u_all: no samples / bad ldu");
    MCML_HIP(hipSetDevice(c.device));
    if (comm_world(c) <= 1) {
        if (ncols_out) *ncols_out = c.mcols;
        return download_matrix(u, ldu, c.U.d(), c.U.ld, c.Q, c.mcols, c.stream);
    }
    MCML_TRY(gather_samples(c));
    if (ncols_out) *ncols_out = c.Uall.cols;
    return download_matrix(u, ldu, c.Uall.d(), c.Uall.ld, c.Q, c.Uall.cols, c.stream);
}

extern "C" int glmmr_mcml_dbg_theta_log(glmmr_mcml_ctx* h, int enable, double* out, int cap_rows, int* nrows)
{
    MCML_REQUIRE(h, "theta_log: null context");
    Ctx& c = h->c;
    const int w = c.cov.npar + 1;
    const int have = (int)(c.theta_log.size() / (size_t)w);
    if (nrows) *nrows = have;
    if (out) {
        const int take = have < cap_rows ? have : cap_rows;       // the LAST `take` rows
        memcpy(out, c.theta_log.data() + (size_t)(have - take) * w, sizeof(double) * (size_t)take * w);
        if (nrows) *nrows = take;
    }
    if (enable >= 0) { c.theta_log_on = enable != 0; if (enable) c.theta_log.clear(); }
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_last_kernels(glmmr_mcml_ctx* h, int* fwd, int* bwd)
{
    MCML_REQUIRE(h, "last_kernels: null context");
    if (fwd) *fwd = h->c.last_kernel[0];
    if (bwd) *bwd = h->c.last_kernel[1];
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_shard_stats(glmmr_mcml_ctx* h, long long* out6)
{
    MCML_REQUIRE(h && out6, "shard_stats: null argument");
    const Ctx& c = h->c;
    out6[0] = c.gather_calls; out6[1] = c.gather_doubles; out6[2] = c.theta_rounds; out6[3] = c.theta_evals_own;
    out6[4] = c.theta_evals_all; out6[5] = 0;
    return MCML_OK;
}

extern "C" int glmmr_mcml_dbg_emulate_world(glmmr_mcml_ctx* h, int world, int mode)
{
    MCML_REQUIRE(h, "emulate_world: null context");
    Ctx& c = h->c;
    MCML_REQUIRE(c.world <= 1 && !c.comm, "emulate_world: the context is rank %d of a real group of %d", c.rank, c.world);
    MCML_REQUIRE(world <= 1 || mode == 1 || mode == 2, "emulate_world: mode must be 1 (record) or 2 (replay)");
    if (world <= 1) { c.emu_world = 0; c.emu_mode = 0; c.emu_trace.clear(); c.emu_pos = 0; c.uall_valid = false; return MCML_OK; }
    if (c.emu_world != world) { c.emu_trace.clear(); c.uall_valid = false; }
    c.emu_world = world; c.emu_mode = mode; c.emu_pos = 0;
    if (mode == 1) c.emu_trace.clear();
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_mvn_ll(glmmr_mcml_ctx* h, const double* theta, double* out)
{
    MCML_REQUIRE(h && theta && out, "mvn_ll: null argument");
    Ctx& c = h->c;
    MCML_HIP(hipSetDevice(c.device));
    double s = 0;
    MCML_TRY(mvn_loglik_sum(c, theta, &s));
    double tot[2] = {s, (double)c.mcols};
    MCML_TRY(allreduce_host(c, tot, 2));
    *out = tot[0] / tot[1];        // loglV / ncols, mcmldmatrix.h:40
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_mvn_ll_batch(glmmr_mcml_ctx* h, const double* thetas, int k, double* out)
{
    MCML_REQUIRE(h && thetas && out && k >= 1 && k <= 64, "mvn_ll_batch: bad argument");
    Ctx& c = h->c;
    MCML_HIP(hipSetDevice(c.device));
    MCML_REQUIRE(c.mcols > 0 && c.U.d(), "mvn_ll: no samples set");
    std::vector<double> tot(k + 1, 0.0);
    std::vector<int> rcs(k, 0);
    MCML_TRY(mvn_loglik_batch(c, thetas, k, c.U.d(), c.U.ld, c.mcols, tot.data(), rcs.data()));
    for (int j = 0; j < k; ++j) if (rcs[j] == MCML_ENOTPD) tot[j] = NAN;          // a sum with NaN stays NaN on every rank
    tot[k] = (double)c.mcols;
    MCML_TRY(allreduce_host(c, tot.data(), k + 1));
    for (int j = 0; j < k; ++j) out[j] = tot[j] / tot[k];
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_gen_D(glmmr_mcml_ctx* h, const double* theta, int chol, double* out, int ldo)
{
    MCML_REQUIRE(h && theta && out, "gen_D: null argument");
    Ctx& c = h->c;
    MCML_REQUIRE(ldo >= c.Q, "gen_D: ldo too small");
    MCML_HIP(hipSetDevice(c.device));
    MCML_TRY(mvn_gen_L(c, theta, chol != 0));
    return download_matrix(out, ldo, c.L.d(), c.L.ld, c.Q, c.Q, c.stream);
}

// beta += (1/m sum X'W_iX)^-1 X' (1/m sum W_i detadmu resid_i); sigma = mean sigma_i
// (mcmloptim.h:227-235); Gauss-Jordan with partial pivoting stands in for Eigen's .inverse()
int mcml::mcnr_finish(int P, const double* stats, const double* beta, double* beta_out, double* sigma_out)
{
    const double m = stats[P * P + P + 1];
    MCML_REQUIRE(m > 0, "mcnr: no samples");
    std::vector<double> A(stats, stats + P * P), B((size_t)P * P, 0.0);
    for (auto& v : A) v *= (double)1 / m;
    for (int i = 0; i < P; i++) B[i + (size_t)i * P] = 1.0;
    for (int c = 0; c < P; c++) {
        int p = c; double best = fabs(A[c + (size_t)c * P]);
        for (int i = c + 1; i < P; i++) if (fabs(A[i + (size_t)c * P]) > best) { best = fabs(A[i + (size_t)c * P]); p = i; }
        if (best == 0.0 || best != best) { set_error("mcnr: X'WX is singular"); return MCML_ESINGULAR; }
        if (p != c) for (int j = 0; j < P; j++) { std::swap(A[c + (size_t)j * P], A[p + (size_t)j * P]); std::swap(B[c + (size_t)j * P], B[p + (size_t)j * P]); }
        double d = A[c + (size_t)c * P];
        for (int j = 0; j < P; j++) { A[c + (size_t)j * P] /= d; B[c + (size_t)j * P] /= d; }
        for (int i = 0; i < P; i++) if (i != c) {
            double f = A[i + (size_t)c * P];
            if (f != 0.0) for (int j = 0; j < P; j++) { A[i + (size_t)j * P] -= f * A[c + (size_t)j * P]; B[i + (size_t)j * P] -= f * B[c + (size_t)j * P]; }
        }
    }
    for (int a = 0; a < P; a++) {
        double inc = 0;
        for (int b = 0; b < P; b++) inc += B[a + (size_t)b * P] * (stats[P * P + b] / m);
        beta_out[a] = beta[a] + inc;
    }
    *sigma_out = stats[P * P + P] / m;
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_loglik(glmmr_mcml_ctx* h, const double* beta, double var_par, double* out)
{
    MCML_REQUIRE(h && beta && out, "loglik: null argument");
    Ctx& c = h->c;
    MCML_HIP(hipSetDevice(c.device));
    MCML_TRY(model_update_beta(c, beta));
    double s = 0;
    MCML_TRY(model_loglik_sum(c, var_par, &s));
    double tot[2] = {s, (double)c.niter};
    MCML_TRY(allreduce_host(c, tot, 2));
    *out = tot[0] / tot[1];        // ll.mean(), mcmlmodel.h:303
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_mcnr(glmmr_mcml_ctx* h, const double* beta, double var_par,
                                   double* beta_out, double* sigma_out, double* stats_out)
{
    MCML_REQUIRE(h && beta && beta_out && sigma_out, "mcnr: null argument");
    Ctx& c = h->c;
    MCML_HIP(hipSetDevice(c.device));
    MCML_TRY(model_update_beta(c, beta));
    std::vector<double> st((size_t)c.P * c.P + c.P + 2);
    MCML_TRY(model_mcnr_stats(c, var_par, st.data()));
    if (stats_out) memcpy(stats_out, st.data(), sizeof(double) * ((size_t)c.P * c.P + c.P + 1));
    return mcnr_finish(c.P, st.data(), beta, beta_out, sigma_out);
}

extern "C" int glmmr_mcml_ctx_update_L(glmmr_mcml_ctx* h, const double* theta)
{
    MCML_REQUIRE(h && theta, "update_L: null argument");
    Ctx& c = h->c;
    MCML_HIP(hipSetDevice(c.device));
    MCML_TRY(mvn_gen_L(c, theta, true));
    MCML_TRY(model_update_L(c));
    return c.sync();
}

extern "C" int glmmr_mcml_ctx_set_L(glmmr_mcml_ctx* h, const double* L, int ldl)
{
    MCML_REQUIRE(h && L, "set_L: null argument");
    Ctx& c = h->c;
    MCML_REQUIRE(ldl >= c.Q, "set_L: ldl too small");
    MCML_HIP(hipSetDevice(c.device));
    MCML_TRY(upload_matrix(c.L, L, c.Q, c.Q, ldl, c.stream));
    c.have_L = true;
    // a caller-supplied L need not have the block pattern the sparse ZL operator assumes (entries outside the
    // covariance blocks would be dropped): the dense products take whatever L holds
    c.l_foreign = true;
    MCML_TRY(model_update_L(c));
    return c.sync();
}

extern "C" int glmmr_mcml_ctx_hmc_sample(glmmr_mcml_ctx* h, const double* beta, double var_par,
                                         const glmmr_mcml_hmc_opts* opts, uint64_t seed, uint32_t iter_idx,
                                         const double* inj_init, const double* inj_mom, uint8_t* flags_out,
                                         double* probs_out, glmmr_mcml_hmc_diag* diag, int* ncols_out)
{
    MCML_REQUIRE(h && beta && opts, "hmc_sample: null argument");
    MCML_HIP(hipSetDevice(h->c.device));
    return hmc_sample(h->c, beta, var_par, opts, seed, iter_idx, inj_init, inj_mom, flags_out, probs_out,
                      diag, ncols_out);
}

extern "C" int glmmr_mcml_ctx_nuts_sample(glmmr_mcml_ctx* h, const double* beta, double var_par,
                                          const glmmr_mcml_nuts_opts* opts, uint64_t seed, uint32_t iter_idx,
                                          int* depth_out, int* nleap_out, double* eps_out, double* accept_out,
                                          glmmr_mcml_nuts_diag* diag, int* ncols_out)
{
    MCML_REQUIRE(h && beta && opts, "nuts_sample: null argument");
    MCML_HIP(hipSetDevice(h->c.device));
    return nuts_sample(h->c, beta, var_par, opts, seed, iter_idx, depth_out, nleap_out, eps_out, accept_out, diag,
                       ncols_out);
}

extern "C" int glmmr_mcml_dbg_log_prob_grad(glmmr_mcml_ctx* h, const double* beta, double var_par,
                                            const double* V, int ncols, double* lp, double* G)
{
    MCML_REQUIRE(h && beta && V && lp && G && ncols > 0, "log_prob_grad: bad argument");
    MCML_HIP(hipSetDevice(h->c.device));
    return hmc_dbg_log_prob_grad(h->c, beta, var_par, V, ncols, lp, G);
}

extern "C" int glmmr_mcml_mvn_ll(const int32_t* cov, int cov_rows, const double* data, int data_len,
                                 const double* eff_range, int eff_len, const double* gamma, int ngamma,
                                 const double* u, int Q, int m, double* out)
{
    glmmr_mcml_problem p{};
    p.cov = cov; p.cov_rows = cov_rows; p.data = data; p.data_len = data_len;
    p.eff_range = eff_range; p.eff_len = eff_len; p.Q = Q;
    glmmr_mcml_ctx* h = nullptr;
    MCML_TRY(glmmr_mcml_ctx_create(&p, nullptr, &h));
    int rc = MCML_OK;
    if (ngamma < h->c.cov.npar) {
        set_error("mvn_ll: %d covariance parameters given, the functions need %d", ngamma, h->c.cov.npar);
        rc = MCML_EINVAL;
    }
    if (!rc) rc = glmmr_mcml_set_u(h, u, Q, m, m);
    if (!rc) rc = glmmr_mcml_ctx_mvn_ll(h, gamma, out);
    glmmr_mcml_ctx_destroy(h);
    return rc;
}

// ------------------------------------------------------------------------- drivers
namespace mcml {
int drv_optim(Ctx& c, const double* start, int nstart, int trace, int mcnr, const glmmr_mcml_ext* e,
              double* beta, double* theta, double* sigma);
int drv_simlik(Ctx& c, const double* start, int nstart, int trace, const glmmr_mcml_ext* e, double* beta,
               double* theta, double* sigma);
int drv_hess(Ctx& c, const double* start, int nstart, double tol, int trace, double* H);
int drv_aic(Ctx& c, const double* beta_par, int nbeta, const double* cov_par, int ncov, double* out);
int drv_full(Ctx& c, const double* start, int nstart, int mcnr, int m, int maxiter, int warmup, double tol,
             int verbose, double lambda, int trace, int refresh, int maxsteps, double target_accept,
             const glmmr_mcml_ext* e, double* beta_out, double* theta_out, double* sigma_out,
             int* converged_out, int* iters_out, glmmr_mcml_hmc_diag* last_diag);
int drv_la(Ctx& c, const double* start, int nstart, int nr, int usehess, double tol, int verbose, int trace,
           int maxiter, const glmmr_mcml_ext* e, double* beta, double* theta, double* sigma, double* se, double* u,
           int* converged, int* iters);
int drv_la_probe(Ctx& c, const double* start, int nstart, int kind, const double* v, double var_par,
                 const double* par, int npar, double* out, double* v_out, double* beta_out, double* sigma_out);
}

extern "C" int glmmr_mcml_sample_cols(int m, int chains)
{
    if (m <= 0) return 0;
    if (chains <= 1) return m + 1;                 // mhmcmc.h:126
    return chains * ((m + chains - 1) / chains);
}

// kernel timing (HIP events on the context's stream).
// out8: [fwd_ms, fwd_count, bwd_ms, bwd_count, executed flops per forward launch, per backward
// launch, dense flops per launch (2 n Q C), operator kind (0 dense GEMM, 1 banded GEMM, 2 sparse)]
extern "C" int glmmr_mcml_ctx_profile(glmmr_mcml_ctx* h, int enable, int reset, double* out8)
{
    MCML_REQUIRE(h, "profile: null context");
    Ctx& c = h->c;
    KernelProf& p = c.prof;
    if (out8) {
        out8[0] = p.ms[0]; out8[1] = (double)p.cnt[0]; out8[2] = p.ms[1]; out8[3] = (double)p.cnt[1];
        const double C = (double)c.hmc.C, dense = 2.0 * c.n * (double)c.Q * C;
        double ff = dense, fb = dense, kind = 0;
        if (c.sp.active) { ff = fb = 2.0 * (double)c.sp.nnz * C; kind = 2; }
        else {
            if (c.band_fwd) { ff = 2.0 * 80.0 * 32.0 * (double)c.band_fwd_tiles * C; kind = 1; }
            if (c.band_bwd) { fb = 2.0 * 80.0 * 32.0 * (double)c.band_bwd_tiles * C; kind = 1; }
        }
        out8[4] = ff; out8[5] = fb; out8[6] = dense; out8[7] = kind;
    }
    if (reset) for (int i = 0; i < 4; ++i) { p.ms[i] = 0; p.cnt[i] = 0; p.seen[i] = 0; }
    if (enable && !p.on) {
        // the markers' events are made HERE, not at their first use inside whatever is being timed: a sampler call of
        // 100 proposals records ~1000 of them, and creating those on the fly cost the first timed MCML iteration of a
        // small configuration 20-75 ms (config 4 in bench.py: 45 or 70 ms per iteration from one process to the next)
        MCML_HIP(hipSetDevice(c.device));
        while (p.ev.size() < 4096) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); break; } p.ev.push_back(e); }
    }
    p.on = enable != 0;
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_profile_launches(glmmr_mcml_ctx* h, long long* fwd, long long* bwd)
{
    MCML_REQUIRE(h, "profile_launches: null context");
    if (fwd) *fwd = h->c.prof.seen[0];
    if (bwd) *bwd = h->c.prof.seen[1];
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_ncols(glmmr_mcml_ctx* h) { return h ? h->c.mcols : 0; }
extern "C" int glmmr_mcml_ctx_npar(glmmr_mcml_ctx* h) { return h ? h->c.cov.npar : 0; }

extern "C" int glmmr_mcml_ctx_optim(glmmr_mcml_ctx* h, const double* start, int nstart, int trace, int mcnr,
                                    const glmmr_mcml_ext* ext, double* beta, double* theta, double* sigma)
{
    MCML_REQUIRE(h && start && beta && theta && sigma, "optim: null argument");
    MCML_REQUIRE(h->c.n > 0, "optim: context has no model");
    MCML_HIP(hipSetDevice(h->c.device));
    return drv_optim(h->c, start, nstart, trace, mcnr, ext, beta, theta, sigma);
}

extern "C" int glmmr_mcml_ctx_simlik(glmmr_mcml_ctx* h, const double* start, int nstart, int trace,
                                     const glmmr_mcml_ext* ext, double* beta, double* theta, double* sigma)
{
    MCML_REQUIRE(h && start && beta && theta && sigma, "simlik: null argument");
    MCML_REQUIRE(h->c.n > 0, "simlik: context has no model");
    MCML_HIP(hipSetDevice(h->c.device));
    return drv_simlik(h->c, start, nstart, trace, ext, beta, theta, sigma);
}

extern "C" int glmmr_mcml_ctx_hess(glmmr_mcml_ctx* h, const double* start, int nstart, double tol, int trace,
                                   double* H)
{
    MCML_REQUIRE(h && start && H, "hess: null argument");
    MCML_REQUIRE(h->c.n > 0 && tol > 0, "hess: context has no model / bad tol");
    MCML_HIP(hipSetDevice(h->c.device));
    return drv_hess(h->c, start, nstart, tol, trace, H);
}

extern "C" int glmmr_mcml_ctx_aic(glmmr_mcml_ctx* h, const double* beta_par, int nbeta, const double* cov_par,
                                  int ncov, double* out)
{
    MCML_REQUIRE(h && beta_par && cov_par && out, "aic: null argument");
    MCML_REQUIRE(h->c.n > 0, "aic: context has no model");
    MCML_HIP(hipSetDevice(h->c.device));
    return drv_aic(h->c, beta_par, nbeta, cov_par, ncov, out);
}

extern "C" int glmmr_mcml_ctx_full(glmmr_mcml_ctx* h, const double* start, int nstart, int mcnr, int m,
                                   int maxiter, int warmup, double tol, int verbose, double lambda, int trace,
                                   int refresh, int maxsteps, double target_accept, const glmmr_mcml_ext* ext,
                                   double* beta, double* theta, double* sigma, int* converged, int* iters,
                                   glmmr_mcml_hmc_diag* diag)
{
    MCML_REQUIRE(h && start && beta && theta && sigma, "mcml_full: null argument");
    MCML_REQUIRE(h->c.n > 0, "mcml_full: context has no model");
    MCML_HIP(hipSetDevice(h->c.device));
    return drv_full(h->c, start, nstart, mcnr, m, maxiter, warmup, tol, verbose, lambda, trace, refresh, maxsteps,
                    target_accept, ext, beta, theta, sigma, converged, iters, diag);
}

// Laplace-approximation fits (src/mcml_la.cpp): nr = 0 mcml_la, 1 mcml_la_nr
extern "C" int glmmr_mcml_ctx_la(glmmr_mcml_ctx* h, const double* start, int nstart, int nr, int usehess, double tol,
                                 int verbose, int trace, int maxiter, const glmmr_mcml_ext* ext, double* beta,
                                 double* theta, double* sigma, double* se, double* u, int* converged, int* iters)
{
    MCML_REQUIRE(h && start && beta && theta && sigma, "mcml_la: null argument");
    MCML_HIP(hipSetDevice(h->c.device));
    return drv_la(h->c, start, nstart, nr, usehess, tol, verbose, trace, maxiter, ext, beta, theta, sigma, se, u,
                  converged, iters);
}

extern "C" int glmmr_mcml_dbg_leaf_profile(glmmr_mcml_ctx* h, unsigned long long* out10)
{
    MCML_REQUIRE(h && out10, "leaf_profile: null argument");
    return potrf_leaf_profile(h->c, out10);
}

extern "C" int glmmr_mcml_dbg_la_probe(glmmr_mcml_ctx* h, const double* start, int nstart, int kind, const double* v,
                                       double var_par, const double* par, int npar, double* out, double* v_out,
                                       double* beta_out, double* sigma_out)
{
    MCML_REQUIRE(h && start, "la_probe: null argument");
    MCML_REQUIRE(kind >= 0 && kind <= 3, "la_probe: kind %d", kind);
    MCML_REQUIRE(kind == 3 ? (v_out && beta_out && sigma_out) : (par && out && npar > 0), "la_probe: null output");
    MCML_HIP(hipSetDevice(h->c.device));
    return drv_la_probe(h->c, start, nstart, kind, v, var_par, par, npar, out, v_out, beta_out, sigma_out);
}

// ---- host-buffer mirrors of the Rcpp exports ----
namespace {
struct CtxGuard {
    glmmr_mcml_ctx* h = nullptr;
    ~CtxGuard() { if (h) glmmr_mcml_ctx_destroy(h); }
};
int open_ctx(const glmmr_mcml_problem* prob, const glmmr_mcml_ext* ext, CtxGuard& g)
{
    glmmr_mcml_dev_opts o{};
    o.device = ext ? ext->device : 0; o.world = 1;
    return glmmr_mcml_ctx_create(prob, &o, &g.h);
}
// the sparse exports pass the CSC pattern (Ap, Ai) of D (R6ModelExtMCML.R:193-195); D is
// block diagonal, so the pattern must fit inside the blocks the cov matrix describes
int check_pattern(const Ctx& c, const int32_t* Ap, const int32_t* Ai, int nnz)
{
    MCML_REQUIRE(Ap && Ai && nnz >= 0, "sparse: null pattern");
    std::vector<int> blk(c.Q);
    for (int b = 0; b < c.cov.B; ++b) for (int k = 0; k < c.cov.blocks[b].dim; ++k) blk[c.cov.blocks[b].matstart + k] = b;
    MCML_REQUIRE(Ap[0] == 0 && Ap[c.Q] == nnz, "sparse: Ap does not describe %d columns / %d entries", c.Q, nnz);
    for (int j = 0; j < c.Q; ++j)
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
            MCML_REQUIRE(Ai[p] >= 0 && Ai[p] < c.Q, "sparse: row index out of range");
            MCML_REQUIRE(blk[Ai[p]] == blk[j], "sparse: entry (%d,%d) lies outside the covariance blocks", Ai[p], j);
        }
    return MCML_OK;
}
}  // namespace

extern "C" int glmmr_mcml_optim(const glmmr_mcml_problem* prob, const double* u, int ucols, const double* start,
                                int nstart, int trace, int mcnr, const glmmr_mcml_ext* ext, double* beta,
                                double* theta, double* sigma)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    MCML_TRY(glmmr_mcml_set_u(g.h, u, prob->Q, ucols, ucols));
    return glmmr_mcml_ctx_optim(g.h, start, nstart, trace, mcnr, ext, beta, theta, sigma);
}

extern "C" int glmmr_mcml_simlik(const glmmr_mcml_problem* prob, const double* u, int ucols, const double* start,
                                 int nstart, int trace, const glmmr_mcml_ext* ext, double* beta, double* theta,
                                 double* sigma)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    MCML_TRY(glmmr_mcml_set_u(g.h, u, prob->Q, ucols, ucols));
    return glmmr_mcml_ctx_simlik(g.h, start, nstart, trace, ext, beta, theta, sigma);
}

// LDL' factor of the block-diagonal D(theta) in the layout SparseChol hands back
// (mcml_optim.cpp:180-182: chol_->L->{Ap,Ai,Ax}, chol_->D): L unit lower triangular stored
// column-compressed WITHOUT its diagonal, D the pivots; chol(D) = L * diag(sqrt(D)).
// From the dense block factors: L_ldl = L_chol * diag(1 / diag(L_chol)), D = diag(L_chol)^2.
static int sparse_factor_nnz(const Ctx& c)
{
    long nnz = 0;
    for (const auto& b : c.cov.blocks) if (!b.all_gr) nnz += (long)b.dim * (b.dim - 1) / 2;
    return (int)nnz;
}

extern "C" int glmmr_mcml_sparse_factor_nnz(const int32_t* cov, int cov_rows, const double* data, int data_len)
{
    CovSpec cs;
    if (cs.parse(cov, cov_rows, data, data_len, nullptr, 0)) return -1;
    long nnz = 0;
    for (const auto& b : cs.blocks) if (!b.all_gr) nnz += (long)b.dim * (b.dim - 1) / 2;
    return (int)nnz;
}

static int sparse_factor(Ctx& c, const double* theta, int32_t* Lp, int32_t* Li, double* Lx, double* D)
{
    MCML_TRY(mvn_gen_L(c, theta, true));
    std::vector<double> blk;
    int nz = 0;
    for (const auto& b : c.cov.blocks) {
        blk.assign((size_t)b.dim * b.dim, 0.0);
        MCML_TRY(download_matrix(blk.data(), b.dim, c.L.at(b.matstart, b.matstart), c.L.ld, b.dim, b.dim, c.stream));
        for (int j = 0; j < b.dim; ++j) {
            const double d = blk[j + (size_t)j * b.dim];
            Lp[b.matstart + j] = nz;
            D[b.matstart + j] = d * d;
            if (!b.all_gr)
                for (int i = j + 1; i < b.dim; ++i) { Li[nz] = b.matstart + i; Lx[nz] = blk[i + (size_t)j * b.dim] / d; ++nz; }
        }
    }
    Lp[c.Q] = nz;
    return MCML_OK;
}

extern "C" int glmmr_mcml_optim_sparse(const glmmr_mcml_problem* prob, const int32_t* Ap, const int32_t* Ai,
                                       int nnz, const double* u, int ucols, const double* start, int nstart,
                                       int trace, int mcnr, const glmmr_mcml_ext* ext, double* beta, double* theta,
                                       double* sigma, int32_t* Lp, int32_t* Li, double* Lx, double* D, int lcap)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    MCML_TRY(check_pattern(g.h->c, Ap, Ai, nnz));
    MCML_TRY(glmmr_mcml_set_u(g.h, u, prob->Q, ucols, ucols));
    MCML_TRY(glmmr_mcml_ctx_optim(g.h, start, nstart, trace, mcnr, ext, beta, theta, sigma));
    if (Lp && Li && Lx && D) {
        MCML_REQUIRE(lcap >= sparse_factor_nnz(g.h->c), "optim_sparse: factor needs %d entries, %d given",
                     sparse_factor_nnz(g.h->c), lcap);
        MCML_TRY(sparse_factor(g.h->c, theta, Lp, Li, Lx, D));
    }
    return MCML_OK;
}

extern "C" int glmmr_mcml_simlik_sparse(const glmmr_mcml_problem* prob, const int32_t* Ap, const int32_t* Ai,
                                        int nnz, const double* u, int ucols, const double* start, int nstart,
                                        int trace, const glmmr_mcml_ext* ext, double* beta, double* theta,
                                        double* sigma)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    MCML_TRY(check_pattern(g.h->c, Ap, Ai, nnz));
    MCML_TRY(glmmr_mcml_set_u(g.h, u, prob->Q, ucols, ucols));
    return glmmr_mcml_ctx_simlik(g.h, start, nstart, trace, ext, beta, theta, sigma);
}

extern "C" int glmmr_mcml_hess(const glmmr_mcml_problem* prob, const double* u, int ucols, const double* start,
                               int nstart, double tol, int trace, const glmmr_mcml_ext* ext, double* H)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    MCML_TRY(glmmr_mcml_set_u(g.h, u, prob->Q, ucols, ucols));
    return glmmr_mcml_ctx_hess(g.h, start, nstart, tol, trace, H);
}

extern "C" int glmmr_mcml_hess_sparse(const glmmr_mcml_problem* prob, const int32_t* Ap, const int32_t* Ai,
                                      int nnz, const double* u, int ucols, const double* start, int nstart,
                                      double tol, int trace, const glmmr_mcml_ext* ext, double* H)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    MCML_TRY(check_pattern(g.h->c, Ap, Ai, nnz));
    MCML_TRY(glmmr_mcml_set_u(g.h, u, prob->Q, ucols, ucols));
    return glmmr_mcml_ctx_hess(g.h, start, nstart, tol, trace, H);
}

extern "C" int glmmr_mcml_aic(const glmmr_mcml_problem* prob, const double* u, int ucols, const double* beta_par,
                              int nbeta, const double* cov_par, int ncov, const glmmr_mcml_ext* ext, double* out)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    MCML_TRY(glmmr_mcml_set_u(g.h, u, prob->Q, ucols, ucols));
    return glmmr_mcml_ctx_aic(g.h, beta_par, nbeta, cov_par, ncov, out);
}

extern "C" int glmmr_mcml_full(const glmmr_mcml_problem* prob, const double* start, int nstart, int mcnr, int m,
                               int maxiter, int warmup, double tol, int verbose, double lambda, int trace,
                               int refresh, int maxsteps, double target_accept, const glmmr_mcml_ext* ext,
                               double* beta, double* theta, double* sigma, int* converged, double* u, int ldu,
                               int* ucols)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    int iters = 0;
    MCML_TRY(glmmr_mcml_ctx_full(g.h, start, nstart, mcnr, m, maxiter, warmup, tol, verbose, lambda, trace, refresh,
                                 maxsteps, target_accept, ext, beta, theta, sigma, converged, &iters, nullptr));
    if (ucols) *ucols = g.h->c.mcols;
    if (u) MCML_TRY(glmmr_mcml_get_u(g.h, u, ldu));
    return MCML_OK;
}

extern "C" int glmmr_mcml_la(const glmmr_mcml_problem* prob, const double* start, int nstart, int usehess, double tol,
                             int verbose, int trace, int maxiter, const glmmr_mcml_ext* ext, double* beta,
                             double* theta, double* sigma, double* se, double* u)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    return glmmr_mcml_ctx_la(g.h, start, nstart, 0, usehess, tol, verbose, trace, maxiter, ext, beta, theta, sigma, se,
                             u, nullptr, nullptr);
}

extern "C" int glmmr_mcml_la_nr(const glmmr_mcml_problem* prob, const double* start, int nstart, int usehess,
                                double tol, int verbose, int trace, int maxiter, const glmmr_mcml_ext* ext,
                                double* beta, double* theta, double* sigma, double* se, double* u)
{
    CtxGuard g;
    MCML_TRY(open_ctx(prob, ext, g));
    return glmmr_mcml_ctx_la(g.h, start, nstart, 1, usehess, tol, verbose, trace, maxiter, ext, beta, theta, sigma, se,
                             u, nullptr, nullptr);
}

extern "C" int glmmr_mcml_mcmc_sample(const double* Z, const double* L, const double* X, const double* y, int n,
                                      int Q, int P, const double* beta, const char* family, const char* link,
                                      int warmup, int nsamp, double lambda, double var_par, int trace, int refresh,
                                      int maxsteps, double target_accept, const glmmr_mcml_ext* ext,
                                      double* samples, int lds, int* ncols)
{
    (void)trace; (void)refresh;
    MCML_REQUIRE(Z && L && X && y && beta && samples, "mcmc_sample: null argument");
    glmmr_mcml_problem p{};
    p.Z = Z; p.X = X; p.y = y; p.n = n; p.Q = Q; p.P = P; p.family = family; p.link = link;
    CtxGuard g;
    MCML_TRY(open_ctx(&p, ext, g));
    MCML_TRY(glmmr_mcml_ctx_set_L(g.h, L, Q));
    glmmr_mcml_hmc_opts ho{};
    ho.warmup = warmup; ho.nsamp = nsamp; ho.adapt = 100; ho.lambda = lambda; ho.max_steps = maxsteps;
    ho.target_accept = target_accept; ho.chains = (ext && ext->chains > 0) ? ext->chains : 1; ho.chain_offset = 0;
    uint64_t seed = (ext && ext->seed) ? ext->seed : 0;
    if (!seed) { std::random_device rd; seed = ((uint64_t)rd() << 32) | rd(); }
    MCML_TRY(glmmr_mcml_ctx_hmc_sample(g.h, beta, var_par, &ho, seed, 0, nullptr, nullptr, nullptr, nullptr, nullptr, ncols));
    return glmmr_mcml_get_u(g.h, samples, lds);
}

extern "C" int glmmr_mcml_gen_u_samples(const double* Z, const double* L, const double* X, const double* y, int n, int Q,
                                        int P, const double* beta, const char* family, const char* link, double sigma,
                                        int warmup_iter, int m, const glmmr_mcml_nuts_opts* opts, const glmmr_mcml_ext* ext,
                                        double* samples, int lds, int* ncols)
{
    MCML_REQUIRE(Z && L && X && y && beta && samples, "gen_u_samples: null argument");
    glmmr_mcml_problem p{};
    p.Z = Z; p.X = X; p.y = y; p.n = n; p.Q = Q; p.P = P; p.family = family; p.link = link;
    CtxGuard g;
    MCML_TRY(open_ctx(&p, ext, g));
    MCML_TRY(glmmr_mcml_ctx_set_L(g.h, L, Q));
    glmmr_mcml_nuts_opts no{};
    if (opts) no = *opts;
    no.warmup = warmup_iter; no.nsamp = m;
    if (no.chains <= 0) no.chains = (ext && ext->chains > 0) ? ext->chains : 1;      // gen_u_samples.R:59: chains = 1
    uint64_t seed = (ext && ext->seed) ? ext->seed : 0;
    if (!seed) { std::random_device rd; seed = ((uint64_t)rd() << 32) | rd(); }
    MCML_TRY(glmmr_mcml_ctx_nuts_sample(g.h, beta, sigma, &no, seed, 0, nullptr, nullptr, nullptr, nullptr, nullptr, ncols));
    return glmmr_mcml_get_u(g.h, samples, lds);
}
