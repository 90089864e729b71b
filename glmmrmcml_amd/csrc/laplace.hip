// laplace.hip -- the Laplace-approximation fits mcml_la / mcml_la_nr on the device.
//   functors  LA_likelihood / LA_likelihood_cov / LA_likelihood_btheta   likelihood.h:112-230
//   steps     la_optim / la_optim_cov / la_optim_bcov / hess_la / mcnr_b mcmloptim.h:116-195,238-293
//   drivers   mcml_la / mcml_la_nr                                       src/mcml_la.cpp:28-290
//   state     mcmlModel ctor, update_W(i, useL), log_grad(v, usezl=false) mcmlmodel.h:51-134,156-168
// Every objective evaluation runs on the device (Z, X, y, L, ZL resident) and returns one
// scalar; the Q x Q x n product ZL' W ZL and its Cholesky use the same MFMA GEMM / potrf as
// the MCML path, the vector-sized pieces are plain streaming kernels.
//
// Reference behaviour reproduced on purpose (also restated in oracle/la.py):
//   * v = u column 0 is the whitened effect, yet update_W(useL = false) forms Z v and
//     log_grad(v, false) forms xb + Z v and -D v (mcmlmodel.h:121,165-166);
//   * D_ is built once from the starting theta and never refreshed (src/mcml_la.cpp:247);
//   * var_par starts at 1 whatever `start` holds (src/mcml_la.cpp:45,191);
//   * at convergence L keeps the previous iteration's theta, and u = L v uses it (:76-97,:147).
// Departures: a theta with a non-positive-definite block is an infinitely bad objective
// instead of NaN; after an optimisation the model is left at the optimum (the reference
// leaves it at rminqa's last evaluated point, which is optimiser-specific); hess_la for the
// beta family is refused (its functor mis-sizes theta, likelihood.h:196-201).
#include "../../include/glmmr_mcml_c.h"
#include "ctx.h"
#include "dgemm_mfma.h"
#include "glm.h"
#include "optim.h"
#include "reduce.h"
#include <cmath>

namespace mcml {

static bool la_is_gaussian(int flink) { return flink == 7 || flink == 8; }
static bool la_has_var_par(int flink) { return flink == 7 || flink == 8 || flink == 12; }

// ------------------------------------------------------------------ kernels
// y = A x  (A rows x cols, column-major): one thread per row
__global__ __launch_bounds__(256) void k_la_gemv_n(const double* A, int lda, int rows, int cols, const double* x,
                                                   double* y)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    double s = 0;
    for (int j = 0; j < cols; ++j) s += A[i + (size_t)j * lda] * x[j];
    y[i] = s;
}

// y[j] = beta y[j] + alpha sum_i A[i + j lda] x[i]  (A' x): one wave per column
__global__ __launch_bounds__(256) void k_la_gemv_t(const double* A, int lda, int rows, int cols, const double* x,
                                                   double* y, double alpha, double beta)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= cols) return;
    double s = 0;
    for (int i = lane; i < rows; i += 64) s += A[i + (size_t)j * lda] * x[i];
    s = wave_sum(s);
    if (lane == 0) y[j] = (beta != 0.0 ? beta * y[j] : 0.0) + alpha * s;
}

// partial sums of log f(y_i | xb_i + zv_i)
__global__ __launch_bounds__(256) void k_la_ll(const double* y, const double* xb, const double* zv, int n,
                                               double var_par, int flink, double* partials)
{
    __shared__ double sh[4];
    double acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        acc += glm_logpdf(y[i], xb[i] + zv[i], var_par, flink);
    const double r = block_sum(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_la_sumsq(const double* v, int n, double* partials)
{
    __shared__ double sh[4];
    double acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += v[i] * v[i];
    const double r = block_sum(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_la_sum(const double* partials, int n, double* out)
{
    __shared__ double sh[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
    const double r = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = r;
}

// W_i = 1 / (dhdmu(xb_i + zv_i) nvar)   (mcmlmodel.h:122-133)
__global__ void k_la_W(const double* xb, const double* zv, int n, int flink, double nvar, double* W)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) W[i] = 1 / (glm_dhdmu(xb[i] + zv[i], flink) * nvar);
}

// out[q + i ld] = A[q + i ld] W[i]   (columns = observations)
__global__ void k_la_scale_cols(const double* A, int lda, int rows, int cols, const double* W, double* out, int ldo)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows) return;
    for (int i = blockIdx.y; i < cols; i += gridDim.y) out[q + (size_t)i * ldo] = A[q + (size_t)i * lda] * W[i];
}

__global__ void k_la_add_identity(double* M, int ld, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) M[i + (size_t)i * ld] += 1.0;
}

__global__ __launch_bounds__(256) void k_la_logdet(const double* A, int lda, int n, double* out)
{
    __shared__ double sh[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += log(A[i + (size_t)i * lda]);
    const double r = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = 2 * r;                     // moremaths.h:105-116
}

// mcnr_b's per-observation pieces (mcmloptim.h:248-266): resid, Wu = W detadmu resid, score(xb + Zv)
__global__ void k_la_nr_obs(const double* y, const double* xb, const double* zlv, const double* zv, const double* W,
                            int n, int flink, int link_code, double var_par, double* resid, double* Wu, double* score)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double eta = xb[i] + zlv[i];
    const double r = y[i] - glm_mod_inv(eta, link_code);
    resid[i] = r;
    Wu[i] = W[i] * glm_detadmu(eta, link_code) * r;
    score[i] = flink == 12 ? glm_score_beta(y[i], xb[i] + zv[i], var_par) : glm_score(y[i], xb[i] + zv[i], flink);
}

__global__ void k_la_axpy(double* y, const double* x, double a, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}

// X' diag(W) X (P x P) and X' Wu (P): tiny P, one block per (a, b) pair / per a
__global__ __launch_bounds__(256) void k_la_xtwx(const double* X, int ldx, int n, int P, const double* W,
                                                 const double* Wu, double* out /* P*P + P */)
{
    __shared__ double sh[4];
    const int idx = blockIdx.x;
    double acc = 0;
    if (idx < P * P) {
        const int a = idx % P, b = idx / P;
        for (int i = threadIdx.x; i < n; i += 256) acc += X[i + (size_t)a * ldx] * W[i] * X[i + (size_t)b * ldx];
    } else {
        const int a = idx - P * P;
        for (int i = threadIdx.x; i < n; i += 256) acc += X[i + (size_t)a * ldx] * Wu[i];
    }
    const double r = block_sum(acc, sh);
    if (threadIdx.x == 0) out[idx] = r;
}

__global__ void k_la_copy(double* dst, const double* src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// x <- L'^-1 (L^-1 x): forward solve through the blocked TRSM (one right-hand side), backward
// solve block by block with the inverted diagonal blocks potrf_lower left in c.linv
int potrs_lower_vec(Ctx& c, const double* L, int ldl, int n, double* x, double* tmp)
{
    MCML_TRY(trsm_left_lower(c, L, ldl, n, x, pad_ld(n), 1));
    const int nblk = (n + CHOL_NB - 1) / CHOL_NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * CHOL_NB, nb = (n - k0 < CHOL_NB) ? n - k0 : CHOL_NB;
        const double* Linv = c.linv.d() + (size_t)kb * CHOL_NB * CHOL_NB;
        hipLaunchKernelGGL(k_la_gemv_t, dim3((nb + 3) / 4), dim3(256), 0, c.stream, Linv, CHOL_NB, nb, nb, x + k0, tmp, 1.0, 0.0);
        hipLaunchKernelGGL(k_la_copy, dim3(1), dim3(128), 0, c.stream, x + k0, tmp, nb);
        if (k0 > 0)
            hipLaunchKernelGGL(k_la_gemv_t, dim3((k0 + 3) / 4), dim3(256), 0, c.stream, L + k0, ldl, nb, k0, x + k0, x, -1.0, 1.0);
        MCML_HIP(hipGetLastError());
    }
    return MCML_OK;
}

// ------------------------------------------------------------------ state
struct LaFit {
    Ctx& c;
    int n, Q, P, R, flink, link_code;
    std::vector<double> beta, theta, theta_model;   // theta_model: the theta the model's L / ZL belong to
    double sigma = 0, var_par = 1.0;
    bool model_L_valid = false;                     // c.L / c.ZL / c.ZLT currently hold L(theta_model)
    int trace = 0, maxfun = 0;
    DevBuf v, W, zv, tmpn, tmpn2, tmpn3, tmpq, tmpq2, part, small;
    DevMat M, ZLTW, D0;
    std::vector<double> hv;                         // host copy of v

    LaFit(Ctx& ctx) : c(ctx), n(ctx.n), Q(ctx.Q), P(ctx.P), R(ctx.cov.npar), flink(ctx.flink), link_code(ctx.link_code) {}

    int nblk() const { int b = (n + 255) / 256; return b > 1024 ? 1024 : (b < 1 ? 1 : b); }

    int init(const double* start)
    {
        beta.assign(start, start + P);
        theta.assign(start + P, start + P + R);
        theta_model = theta;
        sigma = la_is_gaussian(flink) ? start[P + R] : 0;         // mcmloptim.h:30
        var_par = 1.0;
        hv.assign(Q, 0.0);
        MCML_TRY(v.ensure(sizeof(double) * (size_t)(Q + 64)));
        MCML_TRY(W.ensure(sizeof(double) * (size_t)n));
        MCML_TRY(zv.ensure(sizeof(double) * (size_t)n));
        MCML_TRY(tmpn.ensure(sizeof(double) * (size_t)n));
        MCML_TRY(tmpn2.ensure(sizeof(double) * (size_t)n));
        MCML_TRY(tmpn3.ensure(sizeof(double) * (size_t)n));
        MCML_TRY(tmpq.ensure(sizeof(double) * (size_t)(Q + 64)));
        MCML_TRY(tmpq2.ensure(sizeof(double) * (size_t)(Q + 64)));
        MCML_TRY(part.ensure(sizeof(double) * 1100));
        MCML_TRY(small.ensure(sizeof(double) * (size_t)(P * P + P + 16)));
        MCML_HIP(hipMemsetAsync(v.p, 0, sizeof(double) * (size_t)(Q + 64), c.stream));
        c.no_sparse_zl = true;                                    // this path works on the dense ZL
        // D_ = L L' at the starting theta (mcmlmodel.h:71); genD(chol = false) gives it directly
        MCML_TRY(mvn_gen_L(c, theta.data(), false));
        MCML_TRY(D0.alloc(Q, Q));
        MCML_HIP(hipMemcpyAsync(D0.d(), c.L.d(), sizeof(double) * (size_t)c.L.ld * Q, hipMemcpyDeviceToDevice, c.stream));
        model_L_valid = false;
        MCML_TRY(ensure_model_L());
        MCML_TRY(model_update_beta(c, beta.data()));
        MCML_TRY(update_W(false));                                // mcmlmodel.h:94
        return MCML_OK;
    }

    int gen_L(const double* th)                                   // c.L, c.ZL, c.ZLT <- theta
    {
        MCML_TRY(mvn_gen_L(c, th, true));
        MCML_TRY(model_update_L(c));
        model_L_valid = false;
        return MCML_OK;
    }
    int ensure_model_L()
    {
        if (model_L_valid) return MCML_OK;
        MCML_TRY(gen_L(theta_model.data()));
        model_L_valid = true;
        return MCML_OK;
    }
    int set_v(const double* host_v)
    {
        hv.assign(host_v, host_v + Q);
        MCML_TRY(copy_h2d(v.p, hv.data(), sizeof(double) * (size_t)Q, c.stream));
        return MCML_OK;
    }
    int get_v()
    {
        MCML_TRY(copy_d2h(hv.data(), v.p, sizeof(double) * (size_t)Q, c.stream));
        MCML_HIP(hipStreamSynchronize(c.stream));
        return MCML_OK;
    }
    // out = ZL v from the current c.ZLT (Q x n): column i of ZLT is row i of ZL
    int zl_times_v(double* out)
    {
        hipLaunchKernelGGL(k_la_gemv_t, dim3((n + 3) / 4), dim3(256), 0, c.stream, c.ZLT.d(), c.ZLT.ld, Q, n, v.d(), out,
                           1.0, 0.0);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
    int z_times_v(double* out)
    {
        hipLaunchKernelGGL(k_la_gemv_n, dim3((n + 255) / 256), dim3(256), 0, c.stream, c.Z.d(), c.Z.ld, n, Q, v.d(), out);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
    int scalar_from(const double* dev, double* host)
    {
        MCML_TRY(copy_d2h(host, dev, sizeof(double), c.stream));
        MCML_HIP(hipStreamSynchronize(c.stream));
        return MCML_OK;
    }

    // mcmlmodel.h:120-134; useL needs the MODEL's ZL
    int update_W(bool useL)
    {
        if (useL) { MCML_TRY(ensure_model_L()); MCML_TRY(zl_times_v(zv.d())); }
        else MCML_TRY(z_times_v(zv.d()));
        double nvar = 1.0;
        if (la_is_gaussian(flink)) nvar = var_par * var_par;
        else if (flink == 12) nvar = 1 + var_par;
        hipLaunchKernelGGL(k_la_W, dim3((n + 255) / 256), dim3(256), 0, c.stream, c.xb.d(), zv.d(), n, flink, nvar, W.d());
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }

    // sum_i log f(y_i | xb_i + (ZL v)_i) with the CURRENT c.ZLT, and v'v
    int ll_and_vv(double* ll, double* vv)
    {
        MCML_TRY(zl_times_v(tmpn.d()));
        const int nb = nblk();
        hipLaunchKernelGGL(k_la_ll, dim3(nb), dim3(256), 0, c.stream, c.y.d(), c.xb.d(), tmpn.d(), n, var_par, flink, part.d());
        hipLaunchKernelGGL(k_la_sum, dim3(1), dim3(256), 0, c.stream, part.d(), nb, small.d());
        int qb = (Q + 255) / 256; if (qb > 1024) qb = 1024;
        hipLaunchKernelGGL(k_la_sumsq, dim3(qb), dim3(256), 0, c.stream, v.d(), Q, part.d());
        hipLaunchKernelGGL(k_la_sum, dim3(1), dim3(256), 0, c.stream, part.d(), qb, small.d() + 1);
        MCML_HIP(hipGetLastError());
        double h[2];
        MCML_TRY(copy_d2h(h, small.p, sizeof(double) * 2, c.stream));
        MCML_HIP(hipStreamSynchronize(c.stream));
        *ll = h[0]; *vv = h[1];
        return MCML_OK;
    }

    // M = ZL' W ZL + I from the current c.ZL / c.ZLT
    int build_M()
    {
        MCML_TRY(M.alloc(Q, Q));
        MCML_TRY(ZLTW.alloc(Q, n, 32));
        if ((size_t)ZLTW.cols_alloc > (size_t)n)
            MCML_HIP(hipMemsetAsync(ZLTW.at(0, n), 0, sizeof(double) * (size_t)ZLTW.ld * (ZLTW.cols_alloc - n), c.stream));
        int gy = n < 1024 ? n : 1024;
        hipLaunchKernelGGL(k_la_scale_cols, dim3((Q + 255) / 256, gy), dim3(256), 0, c.stream, c.ZLT.d(), c.ZLT.ld, Q, n,
                           W.d(), ZLTW.d(), ZLTW.ld);
        MCML_HIP(hipGetLastError());
        EpiAxpby epi{M.d(), M.ld, 1.0, 0.0};
        MCML_TRY(launch_gemm<false>(c.stream, Q, Q, n, ZLTW.d(), ZLTW.ld, c.ZL.d(), c.ZL.ld, epi));
        hipLaunchKernelGGL(k_la_add_identity, dim3((Q + 255) / 256), dim3(256), 0, c.stream, M.d(), M.ld, Q);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
    int logdet_M(double* out)
    {
        MCML_TRY(build_M());
        int rc = potrf_lower_checked(c, M.d(), Q, M.ld);
        if (rc) return rc;
        hipLaunchKernelGGL(k_la_logdet, dim3(1), dim3(256), 0, c.stream, M.d(), M.ld, Q, small.d() + 2);
        MCML_HIP(hipGetLastError());
        return scalar_from(small.d() + 2, out);
    }

    // LA_likelihood (likelihood.h:112-140): par = (beta, v); uses the model's ZL
    int obj_bv(const std::vector<double>& par, double* val)
    {
        MCML_TRY(ensure_model_L());
        MCML_TRY(model_update_beta(c, par.data()));
        MCML_TRY(set_v(par.data() + P));
        double ll, vv;
        MCML_TRY(ll_and_vv(&ll, &vv));
        *val = -1.0 * (ll - 0.5 * vv);
        return MCML_OK;
    }
    // shared tail of LA_likelihood_cov / _btheta: theta -> L, ZL; ll - v'v/2 - logdet/2
    int obj_theta_tail(const double* th, double* val)
    {
        int rc = gen_L(th);
        if (rc == MCML_ENOTPD) { *val = HUGE_VAL; return MCML_OK; }
        MCML_TRY(rc);
        double ll, vv, ld;
        MCML_TRY(ll_and_vv(&ll, &vv));
        rc = logdet_M(&ld);
        if (rc == MCML_ENOTPD) { *val = HUGE_VAL; return MCML_OK; }
        MCML_TRY(rc);
        *val = -1 * (ll - 0.5 * vv - 0.5 * ld);
        return MCML_OK;
    }
    // LA_likelihood_cov (likelihood.h:142-183): par = (theta[, var_par])
    int obj_cov(const std::vector<double>& par, double* val)
    {
        const int Rp = la_has_var_par(flink) ? (int)par.size() - 1 : (int)par.size();
        if (la_has_var_par(flink)) var_par = par[Rp];
        return obj_theta_tail(par.data(), val);
    }
    // LA_likelihood_btheta (likelihood.h:185-230): par = (beta, theta[, var_par if gaussian])
    int obj_btheta(const std::vector<double>& par, double* val)
    {
        if (la_is_gaussian(flink)) var_par = par.back();
        MCML_TRY(model_update_beta(c, par.data()));
        MCML_TRY(update_W(false));
        return obj_theta_tail(par.data() + P, val);
    }

    BobyqaOpts bopts() const { BobyqaOpts o; o.iprint = trace; if (maxfun > 0) o.maxfun = maxfun; return o; }

    // mcmloptim.h:116-129
    int la_optim()
    {
        std::vector<double> x(beta);
        x.insert(x.end(), hv.begin(), hv.end());
        std::vector<double> lo(x.size(), -HUGE_VAL), up(x.size(), HUGE_VAL);
        objective_fn f = [&](const std::vector<double>& par, double* val) { return obj_bv(par, val); };
        BobyqaResult r;
        MCML_TRY(bobyqa(f, x, lo, up, bopts(), &r));
        beta.assign(r.x.begin(), r.x.begin() + P);
        MCML_TRY(set_v(r.x.data() + P));
        return MCML_OK;
    }
    // mcmloptim.h:131-151
    int la_optim_cov()
    {
        std::vector<double> x(theta), lo(R, 1e-6), up;
        if (la_has_var_par(flink)) { x.push_back(sigma); lo.push_back(0.0); }
        up.assign(x.size(), HUGE_VAL);
        objective_fn f = [&](const std::vector<double>& par, double* val) { return obj_cov(par, val); };
        BobyqaResult r;
        MCML_TRY(bobyqa(f, x, lo, up, bopts(), &r));
        theta.assign(r.x.begin(), r.x.begin() + R);
        if (la_has_var_par(flink)) { sigma = r.x[R]; var_par = sigma; }   // model left at the optimum
        return MCML_OK;
    }
    // mcmloptim.h:153-178
    int la_optim_bcov()
    {
        std::vector<double> x(beta), lo(P, -HUGE_VAL), up;
        for (int i = 0; i < R; ++i) { x.push_back(theta[i]); lo.push_back(1e-6); }
        if (la_is_gaussian(flink)) { x.push_back(sigma); lo.push_back(0.0); }
        up.assign(x.size(), HUGE_VAL);
        objective_fn f = [&](const std::vector<double>& par, double* val) { return obj_btheta(par, val); };
        BobyqaResult r;
        MCML_TRY(bobyqa(f, x, lo, up, bopts(), &r));
        beta.assign(r.x.begin(), r.x.begin() + P);
        theta.assign(r.x.begin() + P, r.x.begin() + P + R);
        if (la_is_gaussian(flink)) { sigma = r.x[P + R]; var_par = sigma; }
        return MCML_OK;
    }
    // mcmloptim.h:180-195
    int hess_la(double tol, std::vector<double>* H)
    {
        if (flink == 12) { set_error("hess_la: the beta family is not built (the reference functor mis-sizes theta)"); return MCML_EUNSUPPORTED; }
        std::vector<double> x(beta);
        x.insert(x.end(), theta.begin(), theta.end());
        if (la_has_var_par(flink)) x.push_back(sigma);
        std::vector<double> nd(x.size(), tol), none;
        objective_fn f = [&](const std::vector<double>& par, double* val) { return obj_btheta(par, val); };
        return fd_hessian(f, x, nd, false, none, none, H);
    }

    // mcmloptim.h:238-293
    int mcnr_b()
    {
        MCML_TRY(ensure_model_L());
        MCML_TRY(zl_times_v(tmpn.d()));                           // zd = ZL v
        MCML_TRY(z_times_v(zv.d()));                              // Z v (log_grad with usezl = false)
        hipLaunchKernelGGL(k_la_nr_obs, dim3((n + 255) / 256), dim3(256), 0, c.stream, c.y.d(), c.xb.d(), tmpn.d(), zv.d(),
                           W.d(), n, flink, link_code, var_par, tmpn2.d(), tmpn3.d(), tmpn.d());
        MCML_HIP(hipGetLastError());
        // tmpn2 = resid, tmpn3 = Wu, tmpn = score(xb + Z v)
        hipLaunchKernelGGL(k_la_xtwx, dim3(P * P + P), dim3(256), 0, c.stream, c.X.d(), c.X.ld, n, P, W.d(), tmpn3.d(), small.d());
        MCML_HIP(hipGetLastError());
        std::vector<double> st((size_t)P * P + P), resid(n);
        MCML_TRY(copy_d2h(st.data(), small.p, sizeof(double) * st.size(), c.stream));
        MCML_TRY(copy_d2h(resid.data(), tmpn2.p, sizeof(double) * (size_t)n, c.stream));
        // vgrad = -D0 v + post * ZL' score
        hipLaunchKernelGGL(k_la_gemv_t, dim3((Q + 3) / 4), dim3(256), 0, c.stream, c.ZL.d(), c.ZL.ld, n, Q, tmpn.d(), tmpq.d(),
                           glm_score_post(var_par, flink), 0.0);
        hipLaunchKernelGGL(k_la_gemv_t, dim3((Q + 3) / 4), dim3(256), 0, c.stream, D0.d(), D0.ld, Q, Q, v.d(), tmpq.d(), -1.0, 1.0);
        MCML_HIP(hipGetLastError());
        // vincr = (ZL' W ZL + I)^-1 vgrad
        MCML_TRY(build_M());
        MCML_TRY(potrf_lower_checked(c, M.d(), Q, M.ld));
        MCML_TRY(potrs_lower_vec(c, M.d(), M.ld, Q, tmpq.d(), tmpq2.d()));
        hipLaunchKernelGGL(k_la_axpy, dim3((Q + 255) / 256), dim3(256), 0, c.stream, v.d(), tmpq.d(), 1.0, Q);
        MCML_HIP(hipGetLastError());
        MCML_TRY(get_v());                                        // also synchronises st / resid
        double mean = 0;
        for (int i = 0; i < n; ++i) mean += resid[i];
        mean /= n;
        double ss = 0;
        for (int i = 0; i < n; ++i) ss += (resid[i] - mean) * (resid[i] - mean);
        // beta += (X'WX)^-1 X' Wu: Gaussian elimination with partial pivoting on the P x P system
        std::vector<double> A(st.begin(), st.begin() + (size_t)P * P), b(st.begin() + (size_t)P * P, st.end());
        for (int k = 0; k < P; ++k) {
            int piv = k;
            for (int i = k + 1; i < P; ++i) if (fabs(A[i + (size_t)k * P]) > fabs(A[piv + (size_t)k * P])) piv = i;
            if (A[piv + (size_t)k * P] == 0.0) { set_error("mcnr_b: X'WX is singular"); return MCML_ESINGULAR; }
            if (piv != k) { for (int j = 0; j < P; ++j) std::swap(A[k + (size_t)j * P], A[piv + (size_t)j * P]); std::swap(b[k], b[piv]); }
            for (int i = k + 1; i < P; ++i) {
                const double f = A[i + (size_t)k * P] / A[k + (size_t)k * P];
                for (int j = k; j < P; ++j) A[i + (size_t)j * P] -= f * A[k + (size_t)j * P];
                b[i] -= f * b[k];
            }
        }
        for (int k = P - 1; k >= 0; --k) {
            double s = b[k];
            for (int j = k + 1; j < P; ++j) s -= A[k + (size_t)j * P] * b[j];
            b[k] = s / A[k + (size_t)k * P];
        }
        for (int k = 0; k < P; ++k) beta[k] += b[k];
        sigma = sqrt(ss / (n - 1));
        return MCML_OK;
    }

    // src/mcml_la.cpp:62-103 / 204-243 and the tail :105-155 / 245-289
    int run(bool nr, bool usehess, double tol, int maxiter, int verbose, double* beta_out, double* theta_out,
            double* sigma_out, double* se_out, int nstart, double* u_out, int* converged_out, int* iters_out)
    {
        std::vector<double> b0 = beta, t0 = theta;
        double vp = 1.0, new_vp = 1.0;
        if (nr) MCML_TRY(update_W(true));                         // :195
        int it = 1; double maxdiff = 1; bool converged = false;
        while (maxdiff > tol && it <= maxiter) {
            if (nr) MCML_TRY(mcnr_b()); else MCML_TRY(la_optim());
            std::vector<double> nb = beta;
            MCML_TRY(model_update_beta(c, nb.data()));
            MCML_TRY(update_W(nr));
            MCML_TRY(la_optim_cov());
            std::vector<double> nt = theta;
            if (la_is_gaussian(flink) || (nr && flink == 12)) new_vp = sigma;     // :84 vs :222
            maxdiff = fabs(vp - new_vp);
            for (int i = 0; i < P; ++i) maxdiff = fmax(maxdiff, fabs(b0[i] - nb[i]));
            for (int i = 0; i < R; ++i) maxdiff = fmax(maxdiff, fabs(t0[i] - nt[i]));
            converged = maxdiff < tol;
            b0 = nb; t0 = nt; vp = new_vp;
            if (!converged) {
                MCML_TRY(model_update_beta(c, b0.data()));
                if (nr) {
                    // update_W(0, true) runs BEFORE update_L: it still sees the previous ZL (:245-247)
                    var_par = new_vp;
                    MCML_TRY(update_W(true));
                } else {
                    MCML_TRY(update_W(false));
                    var_par = new_vp;
                }
                theta_model = t0;
                model_L_valid = false;
            } else {
                // the functor evaluations left other thetas in c.L: the model keeps theta_model
                model_L_valid = false;
            }
            if (verbose) {
                printf("\nIter %d  beta:", it);
                for (double x : b0) printf(" %g", x);
                printf("  theta:");
                for (double x : t0) printf(" %g", x);
                printf("  max diff %g%s\n", maxdiff, converged ? " CONVERGED" : "");
            }
            ++it;
        }
        MCML_TRY(la_optim_bcov());
        if (la_is_gaussian(flink)) vp = sigma;
        for (int i = 0; i < P; ++i) beta_out[i] = beta[i];
        for (int i = 0; i < R; ++i) theta_out[i] = theta[i];
        *sigma_out = vp;
        if (se_out) {
            for (int i = 0; i < nstart; ++i) se_out[i] = 0.0;
            if (usehess) {
                std::vector<double> H;
                MCML_TRY(hess_la(1e-4, &H));
                const int nv = P + R + (la_has_var_par(flink) ? 1 : 0);
                std::vector<double> Hi;
                MCML_TRY(spd_inverse_host(H, nv, &Hi));
                for (int i = 0; i < nv && i < nstart; ++i) se_out[i] = sqrt(Hi[i + (size_t)i * nv]);
            }
        }
        if (u_out) {                                              // u = L v with the driver's L
            model_L_valid = false;
            MCML_TRY(ensure_model_L());
            hipLaunchKernelGGL(k_la_gemv_n, dim3((Q + 255) / 256), dim3(256), 0, c.stream, c.L.d(), c.L.ld, Q, Q, v.d(), tmpq.d());
            MCML_HIP(hipGetLastError());
            MCML_TRY(copy_d2h(u_out, tmpq.p, sizeof(double) * (size_t)Q, c.stream));
            MCML_HIP(hipStreamSynchronize(c.stream));
        }
        if (converged_out) *converged_out = converged ? 1 : 0;
        if (iters_out) *iters_out = it - 1;
        return MCML_OK;
    }

    // hess.llt().solve(I) (src/mcml_la.cpp:120-124): Cholesky inverse of the small Hessian on the host
    static int spd_inverse_host(const std::vector<double>& H, int nv, std::vector<double>* out)
    {
        std::vector<double> L(H);
        for (int j = 0; j < nv; ++j) {
            double d = L[j + (size_t)j * nv];
            for (int k = 0; k < j; ++k) d -= L[j + (size_t)k * nv] * L[j + (size_t)k * nv];
            if (!(d > 0.0)) { set_error("hess_la: Hessian is not positive definite"); return MCML_ENOTPD; }
            d = sqrt(d);
            L[j + (size_t)j * nv] = d;
            for (int i = j + 1; i < nv; ++i) {
                double s = L[i + (size_t)j * nv];
                for (int k = 0; k < j; ++k) s -= L[i + (size_t)k * nv] * L[j + (size_t)k * nv];
                L[i + (size_t)j * nv] = s / d;
            }
        }
        out->assign((size_t)nv * nv, 0.0);
        for (int col = 0; col < nv; ++col) {
            std::vector<double> x(nv, 0.0);
            x[col] = 1.0;
            for (int i = 0; i < nv; ++i) {
                double s = x[i];
                for (int k = 0; k < i; ++k) s -= L[i + (size_t)k * nv] * x[k];
                x[i] = s / L[i + (size_t)i * nv];
            }
            for (int i = nv - 1; i >= 0; --i) {
                double s = x[i];
                for (int k = i + 1; k < nv; ++k) s -= L[k + (size_t)i * nv] * x[k];
                x[i] = s / L[i + (size_t)i * nv];
            }
            for (int i = 0; i < nv; ++i) (*out)[i + (size_t)col * nv] = x[i];
        }
        return MCML_OK;
    }
};

// the Laplace path forces the dense ZL and leaves c.L at an arbitrary theta: restore / invalidate on every exit
struct LaScope {
    Ctx& c;
    explicit LaScope(Ctx& ctx) : c(ctx) {}
    ~LaScope() { c.no_sparse_zl = false; c.have_L = false; }
};

// ------------------------------------------------------------------ entry points (cabi.hip)
int drv_la(Ctx& c, const double* start, int nstart, int nr, int usehess, double tol, int verbose, int trace,
           int maxiter, const glmmr_mcml_ext* e, double* beta, double* theta, double* sigma, double* se, double* u,
           int* converged, int* iters)
{
    MCML_REQUIRE(c.n > 0 && c.cov.npar > 0, "mcml_la: the context has no model / covariance");
    MCML_REQUIRE(start && nstart >= c.P + c.cov.npar + (la_is_gaussian(c.flink) ? 1 : 0), "mcml_la: start too short");
    MCML_REQUIRE(beta && theta && sigma, "mcml_la: null output");
    LaScope scope(c);
    LaFit f(c);
    f.trace = trace;
    f.maxfun = (e && e->maxfun > 0) ? e->maxfun : 0;
    MCML_TRY(f.init(start));
    return f.run(nr != 0, usehess != 0, tol, maxiter, verbose, beta, theta, sigma, se, nstart, u, converged, iters);
}

// test hook: one functor value / one mcnr_b step from a given state (see include/glmmr_mcml_c.h)
int drv_la_probe(Ctx& c, const double* start, int nstart, int kind, const double* v, double var_par,
                 const double* par, int npar, double* out, double* v_out, double* beta_out, double* sigma_out)
{
    MCML_REQUIRE(c.n > 0 && c.cov.npar > 0, "la_probe: the context has no model / covariance");
    MCML_REQUIRE(start && nstart >= c.P + c.cov.npar, "la_probe: start too short");
    LaScope scope(c);
    LaFit f(c);
    MCML_TRY(f.init(start));
    f.var_par = var_par;
    if (v) MCML_TRY(f.set_v(v));
    int rc = MCML_OK;
    if (kind <= 2) {
        std::vector<double> p(par, par + npar);
        if (kind == 1 || kind == 2) MCML_TRY(f.update_W(false));       // W as the constructor / driver leaves it
        rc = kind == 0 ? f.obj_bv(p, out) : kind == 1 ? f.obj_cov(p, out) : f.obj_btheta(p, out);
    } else {
        MCML_TRY(f.update_W(true));
        rc = f.mcnr_b();
        if (!rc) {
            for (int i = 0; i < c.Q; ++i) v_out[i] = f.hv[i];
            for (int i = 0; i < c.P; ++i) beta_out[i] = f.beta[i];
            *sigma_out = f.sigma;
        }
    }
    return rc;
}

}  // namespace mcml
