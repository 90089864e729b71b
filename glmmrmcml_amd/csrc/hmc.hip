// hmc.hip -- the random-effects HMC sampler, mcmcRunHMC (mhmcmc.h:16-160), run as
// C independent chains at once.
//
// The reference runs ONE chain whose every leapfrog step is two n x Q GEMVs that
// stream ZL (200 MB at n = Q = 5000) and are strictly sequential.  Here the
// chains are the columns of Q x C state matrices, so one leapfrog step of all
// chains is two FP64-MFMA GEMMs that read ZL once:
//     forward   MU = xb + ZL * UP ,  S = s(y, MU)          (n x Q x C)
//     backward  g  = -UP + post * ZL' * S                   (Q x n x C)
// with the score and the leapfrog update fused into the GEMM epilogues.  Each
// chain keeps its own step size / dual-averaging state (mhmcmc.h:107-116), its
// own minstd accept stream (:27,55,85) and its own number of steps; chains that
// have finished their trajectory are masked in the backward epilogue.
// log_prob(u) and log_grad(u) of the current state are cached from the step that
// produced it instead of being recomputed (the reference recomputes them,
// :64,82): same numbers, 2*steps GEMMs per proposal instead of 2*steps + 4.
#include <chrono>
#include "../../include/glmmr_mcml_c.h"
#include "ctx.h"
#include "dgemm_mfma.h"
#include "dgemm_dlds.h"
#include "dgemm_band.h"
#include "dgemm_skinny.h"
#include "glm.h"
#include "reduce.h"
#include "rng.h"
#include "hmc_cm.h"

namespace mcml {

struct ChainArrays {
    double *e, *ebar, *H, *lpcur, *K0;
    int *steps, *acc;
    uint32_t* gen;
    long long* leap;
};

static ChainArrays chain_arrays(const HmcState& h)
{
    ChainArrays a;
    const size_t C = (size_t)round_up(h.C, 16);
    double* d = h.chain.d();
    a.e = d; a.ebar = d + C; a.H = d + 2 * C; a.lpcur = d + 3 * C; a.K0 = d + 4 * C;
    a.leap = reinterpret_cast<long long*>(d + 5 * C);
    a.steps = reinterpret_cast<int*>(d + 6 * C);
    a.acc = a.steps + C;
    a.gen = reinterpret_cast<uint32_t*>(a.acc + C);
    return a;
}

// ------------------------------------------------------------------ GEMM epilogues
// forward: MU = xb + acc ; S = score(y, MU)          (mcmlmodel.h:160-162,169-276)
// store_mu = 0: inside a leapfrog trajectory only the score feeds the next product; the linear
// predictor itself is read (by k_hmc_accept) after the LAST step only, so its 8 n C bytes per
// launch are not written
// FL: the family / link code as a compile-time constant (12 = beta/logit, whose digamma score is its own function,
// glm.h; 1, 3, 7 = poisson/log, binomial/logit, gaussian/identity), 0 = run-time code: the epilogue evaluates the score
// 20 times per lane, unrolled -- with a run-time code that is twenty copies of glm_score's switch
template <int FL>
struct EpiForwardT {
    double* MU; double* S; int ld; const double* xb; const double* y; int flink; int store_mu; double var_par;
    __device__ __forceinline__ double score(double yv, double mu) const {
        if constexpr (FL == 12) return glm_score_beta(yv, mu, var_par);
        else return glm_score(yv, mu, FL ? FL : flink);
    }
    __device__ __forceinline__ void elem(int m, int n, double accv) const {
        const double mu = xb[m] + accv;
        if (store_mu) MU[m + (size_t)n * ld] = mu;
        S[m + (size_t)n * ld] = score(y[m], mu);
    }
    // Three phases -- all loads, all arithmetic, all stores -- with no use of a loaded register after the first
    // store.  The compiler's waitcnt insertion cannot count stores issued under divergent control flow, so any
    // such use gets `s_waitcnt vmcnt(0)`, and vector memory completes in order: that wait also drains every store
    // issued so far (one memory round trip per row group, ~1-2 us each with every CU storing at once).
    template <int TM, int TN>
    __device__ __forceinline__ void operator()(d4 (&acc)[TM][TN], int mB, int nB, int lane, int M, int N,
                                               int) const {
        double xbv[TM], yv[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mB + 16 * i + (lane & 15);
            const int mm = m < M ? m : 0;
            xbv[i] = xb[mm]; yv[i] = y[mm];
        }
        double sc[TM][TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double mu = xbv[i] + acc[i][j][r];
                    acc[i][j][r] = mu;
                    sc[i][j][r] = score(yv[i], mu);
                }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nB + 16 * j + (lane >> 4) + 4 * r;
                if (n >= N) continue;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = mB + 16 * i + (lane & 15);
                    if (m < M) {
                        if (store_mu) MU[m + (size_t)n * ld] = acc[i][j][r];
                        S[m + (size_t)n * ld] = sc[i][j][r];
                    }
                }
            }
    }
};

// backward: g = -x + post*acc.  mode 0: GRAD = g (initial state).
// mode 1 (leapfrog step s): for chains with s < steps: GRADP = g; R += e/2 g;
// and, unless it was the chain's last step, R += e/2 g; UP += e R   (mhmcmc.h:73-78)
struct EpiBackward {
    const double* Xs; double* G; double* R; double* UP; int ld;
    const double* e; const int* steps; int s; double post; int mode;
    __device__ __forceinline__ void elem(int m, int n, double accv) const {
        int st = 0; double en = 0.0;
        if (mode == 1) { st = steps[n]; en = e[n]; if (s >= st) return; }
        const size_t off = m + (size_t)n * ld;
        const double x = Xs[off];
        double g = -1.0 * x;
        g = g + post * accv;
        if (mode != 1 || s + 1 >= st) G[off] = g;     // mid-trajectory gradients are never read
        if (mode == 1) {
            double rr = R[off];
            rr = rr + (en / 2) * g;
            if (s + 1 < st) {
                rr = rr + (en / 2) * g;
                UP[off] = x + en * rr;
            }
            R[off] = rr;
        }
    }
    // Per 16-column group: all loads (from clamped, always valid addresses), then all arithmetic into
    // registers, then all stores, and no use of a loaded register after the first store (see EpiForwardT: such a
    // use costs `s_waitcnt vmcnt(0)`, which drains the stores issued so far -- one memory round trip per
    // element group).  Arithmetic order per element is elem()'s; the choices it makes with branches are selects.
    template <int TM, int TN>
    __device__ __forceinline__ void operator()(d4 (&acc)[TM][TN], int mB, int nB, int lane, int M, int N,
                                               int) const {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int stv[4]; double env[4]; bool inN[4]; size_t cb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nB + 16 * j + (lane >> 4) + 4 * r;
                inN[r] = n < N;
                const int nn = inN[r] ? n : 0;
                stv[r] = mode == 1 ? steps[nn] : 0;
                env[r] = mode == 1 ? e[nn] : 0.0;
                cb[r] = (size_t)nn * ld;
            }
            double xv[TM][4], rv[TM][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mB + 16 * i + (lane & 15);
                const int mm = m < M ? m : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xv[i][r] = Xs[mm + cb[r]];
                    rv[i][r] = mode == 1 ? R[mm + cb[r]] : 0.0;
                }
            }
            // g -> acc, new R -> rv, new UP -> xv
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double en = env[r];
                const bool more = s + 1 < stv[r];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const double x = xv[i][r];
                    double g = -1.0 * x;
                    g = g + post * acc[i][j][r];
                    acc[i][j][r] = g;
                    double rr = rv[i][r];
                    rr = rr + (en / 2) * g;
                    const double rr2 = rr + (en / 2) * g;
                    rr = more ? rr2 : rr;
                    rv[i][r] = rr;
                    xv[i][r] = x + en * rr;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool act = inN[r] && (mode != 1 || s < stv[r]);
                if (!act) continue;
                const bool more = s + 1 < stv[r];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = mB + 16 * i + (lane & 15);
                    if (m >= M) continue;
                    const size_t off = m + cb[r];
                    if (mode != 1 || !more) G[off] = acc[i][j][r];       // mid-trajectory gradients are never read
                    if (mode == 1) {
                        if (more) UP[off] = xv[i][r];
                        R[off] = rv[i][r];
                    }
                }
            }
        }
    }
};

// ------------------------------------------------------------------ sparse ZL products
// The forward / backward products of the sparse operator and the per-chain kernels that go with them live in
// hmc_cm.h: with a sparse ZL the whole sampler state is chain-major.

// U = L V for a block-diagonal L with small blocks (the sparse-ZL configurations): row q of L has entries in
// columns start(q) .. q only.  Dense, this product is Q x Q x m (22 ms at config 5 for what is a diagonal scaling).
__global__ __launch_bounds__(256) void k_blockdiag_LV(int Q, int ncols, const int* start, const double* L, int ldl,
                                                      const double* V, int ldv, double* U, int ldu)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    const int s0 = start[q];
    for (int c = blockIdx.y; c < ncols; c += gridDim.y) {
        const double* v = V + (size_t)c * ldv;
        double acc = 0.0;
        for (int t = s0; t <= q; ++t) acc += L[q + (size_t)t * ldl] * v[t];
        U[q + (size_t)c * ldu] = acc;
    }
}

// ------------------------------------------------------------------ per-chain kernels
__global__ __launch_bounds__(256) void k_hmc_init(double* V, int ld, int Q, ChainArrays ca, uint64_t seed,
                                                  uint32_t chain_offset, uint32_t iter_idx, const double* inj_init)
{
    const int c = blockIdx.x;
    const uint32_t gid = chain_offset + (uint32_t)c;
    for (int q = threadIdx.x; q < Q; q += 256)
        V[q + (size_t)c * ld] = inj_init ? inj_init[q + (size_t)c * Q]
                                         : rng_normal(seed, (uint32_t)q, gid, 0u, 16u * iter_idx + 0u);
    if (threadIdx.x == 0) {                                   // initialise_u, mhmcmc.h:47-59
        ca.e[c] = 0.001; ca.ebar[c] = 1.0; ca.H[c] = 0.0; ca.acc[c] = 0; ca.leap[c] = 0;
        ca.gen[c] = chain_minstd_seed(seed, gid, iter_idx);
        ca.steps[c] = 1;
    }
}

// log_prob of column c: sum_i logf(y_i | MU_ic) + sum_k logN(x_k; 0, 1)   (mcmlmodel.h:138-153)
// FL != 0: the family / link is a compile-time constant -- the 12-way switch of glm_logpdf with its lgamma / tgamma / erfc
// bodies inlined costs 302 VGPRs (one wave per SIMD) in every kernel that calls it with a run-time code
template <int FL>
__device__ __forceinline__ double chain_log_prob(const double* MU, int ldm, int n, const double* X, int ldx,
                                                 int Q, const double* y, double var_par, int flink_rt, int c,
                                                 double* sh)
{
    const int flink = FL ? FL : flink_rt;
    // loads batched four deep (the kernel is latency-bound: 4 workgroups per CU); each thread still adds its
    // own elements in index order, so the sums are bit-identical to the plain loop
    double ll = 0, lp = 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 1024) {
        double m4[4], y4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            m4[u] = i < n ? MU[i + (size_t)c * ldm] : 0.0;
            y4[u] = i < n ? y[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < n) ll += glm_logpdf(y4[u], m4[u], var_par, flink);
    }
    for (int k0 = threadIdx.x; k0 < Q; k0 += 1024) {
        double x4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = k0 + 256 * u; x4[u] = k < Q ? X[k + (size_t)c * ldx] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + 256 * u < Q) lp += glm_logpdf(x4[u], 0, 1, 7);
    }
    double a = block_sum(ll, sh);
    double b = block_sum(lp, sh);
    return a + b;    // valid in thread 0
}

template <int FL>
__global__ __launch_bounds__(256) void k_hmc_lp0(const double* MU, int ldm, int n, const double* V, int ld, int Q,
                                                 const double* y, double var_par, int flink, double* lpcur)
{
    __shared__ double sh[4];
    double v = chain_log_prob<FL>(MU, ldm, n, V, ld, Q, y, var_par, flink, blockIdx.x, sh);
    if (threadIdx.x == 0) lpcur[blockIdx.x] = v;
}

// new_proposal, first part (mhmcmc.h:62-75): momentum, K0, steps, first half step + position
__global__ __launch_bounds__(256) void k_hmc_propose(const double* V, const double* GRAD, double* R, double* UP,
                                                     int ld, int Q, ChainArrays ca, double lambda, int max_steps,
                                                     uint64_t seed, uint32_t chain_offset, uint32_t iter_idx,
                                                     int it, const double* inj_mom, int C)
{
    __shared__ double sh[4];
    const int c = blockIdx.x;
    const uint32_t gid = chain_offset + (uint32_t)c;
    const double e = ca.e[c];
    double ss = 0;
    for (int q = threadIdx.x; q < Q; q += 256) {
        const size_t off = q + (size_t)c * ld;
        double r = inj_mom ? inj_mom[q + ((size_t)it * C + c) * Q]
                           : rng_normal(seed, (uint32_t)q, gid, (uint32_t)it, 16u * iter_idx + 2u);
        ss += r * r;
        const double g = GRAD[off], v = V[off];
        r = r + (e / 2) * g;
        R[off] = r;
        UP[off] = v + e * r;
    }
    double tot = block_sum(ss, sh);
    if (threadIdx.x == 0) {
        ca.K0[c] = 0.5 * tot;
        double st = round(lambda / e);                       // mhmcmc.h:69-70
        if (!(st >= 1.0)) st = 1.0;
        if (st > (double)max_steps) st = (double)max_steps;
        ca.steps[c] = (int)st;
        ca.leap[c] += (long long)st;
    }
}

// slot (nullable): host memory mapped into the device -- the count goes there too, tagged with the proposal's sequence number,
// as one 64-bit system-scope store (hmc_sample reads it back with plain loads)
__global__ void k_max_steps(const int* steps, int C, int* out, unsigned long long* slot = nullptr, unsigned seq = 0)
{
    __shared__ int sh[256];
    int v = 0;
    for (int i = threadIdx.x; i < C; i += 256) v = max(v, steps[i]);
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] = max(sh[threadIdx.x], sh[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) {
        out[0] = sh[0];
        if (slot) __hip_atomic_store(slot, ((unsigned long long)seq << 32) | (unsigned)sh[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// new_proposal, second part (mhmcmc.h:80-117)
template <int FL>
__global__ __launch_bounds__(256) void k_hmc_accept(double* V, double* GRAD, const double* R, const double* UP,
                                                    const double* GRADP, int ld, int Q, const double* MU, int ldm,
                                                    int n, const double* y, double var_par, int flink,
                                                    ChainArrays ca, double target_accept, int adapt, int it,
                                                    int C, uint8_t* flags, double* probs)
{
    __shared__ double sh[4];
    __shared__ int acc_s;
    const int c = blockIdx.x;
    double l2 = chain_log_prob<FL>(MU, ldm, n, UP, ld, Q, y, var_par, flink, c, sh);
    double kin = 0;
    for (int k0 = threadIdx.x; k0 < Q; k0 += 1024) {
        double r4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = k0 + 256 * u; r4[u] = k < Q ? R[k + (size_t)c * ld] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + 256 * u < Q) kin += r4[u] * r4[u];
    }
    kin = block_sum(kin, sh);
    if (threadIdx.x == 0) {
        const double lprt = 0.5 * kin, lpr = ca.K0[c], l1 = ca.lpcur[c];
        const double prob = fmin(1.0, exp(-l1 + lpr + l2 - lprt));
        uint32_t g = ca.gen[c];
        const double runif = minstd_canonical(g);
        ca.gen[c] = g;
        const int acc = runif < prob;
        acc_s = acc;
        if (acc) { ca.lpcur[c] = l2; ca.acc[c] += 1; }
        if (flags) flags[c + (size_t)it * C] = (uint8_t)acc;
        if (probs) probs[c + (size_t)it * C] = prob;
        if (adapt) {                                         // mhmcmc.h:107-114
            const int iter = it + 1;
            const double f1 = 1.0 / (iter + 10);
            const double H = (1 - f1) * ca.H[c] + f1 * (target_accept - prob);
            ca.H[c] = H;
            const double loge = -4.60517 - (sqrt((double)iter / 0.05)) * H;
            const double powm = pow((double)iter, -0.75);
            const double logbare = powm * loge + (1 - powm) * log(ca.ebar[c]);
            ca.e[c] = exp(loge);
            ca.ebar[c] = exp(logbare);
        } else {
            ca.e[c] = ca.ebar[c];                            // :116
        }
    }
    __syncthreads();
    const int acc = acc_s;
    if (acc)
        for (int k0 = threadIdx.x; k0 < Q; k0 += 1024) {
            double u4[4], g4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 256 * u;
                const size_t off = (k < Q ? k : 0) + (size_t)c * ld;
                u4[u] = UP[off]; g4[u] = GRADP[off];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 256 * u;
                if (k < Q) { const size_t off = k + (size_t)c * ld; V[off] = u4[u]; GRAD[off] = g4[u]; }
            }
        }
}

// store the current state of every chain as sample columns c*stride + col
__global__ __launch_bounds__(256) void k_hmc_store(const double* V, int ld, int Q, double* SAMP, int lds,
                                                   int stride, int col)
{
    const int c = blockIdx.x;
    for (int k = threadIdx.x; k < Q; k += 256) SAMP[k + (size_t)(c * stride + col) * lds] = V[k + (size_t)c * ld];
}

__global__ void k_hmc_diag(ChainArrays ca, int C, double* out)
{
    // out: [0] sum accept, [1] sum e, [2] min e, [3] max e, [4] max steps, [5] sum leapfrog
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sa = 0, se = 0, mn = 1e300, mx = 0, ms = 0, sl = 0;
    for (int c = 0; c < C; ++c) {
        sa += ca.acc[c]; se += ca.e[c];
        mn = fmin(mn, ca.e[c]); mx = fmax(mx, ca.e[c]);
        ms = fmax(ms, (double)ca.steps[c]); sl += (double)ca.leap[c];
    }
    out[0] = sa; out[1] = se; out[2] = mn; out[3] = mx; out[4] = ms; out[5] = sl;
}

// ------------------------------------------------------------------ host
static int hmc_alloc(Ctx& c, int C)
{
    HmcState& h = c.hmc;
    h.C = C; h.Cw = C;
    h.cm = c.sp.active;
    if (h.cm) {
        // chain-major (hmc_cm.h): "rows" of the DevMat are the chains
        MCML_TRY(h.V.alloc(C, c.Q)); MCML_TRY(h.R.alloc(C, c.Q)); MCML_TRY(h.UP.alloc(C, c.Q));
        MCML_TRY(h.GRAD.alloc(C, c.Q)); MCML_TRY(h.GRADP.alloc(C, c.Q));
        MCML_TRY(h.MU.alloc(C, c.n)); MCML_TRY(h.S.alloc(C, c.n));
        MCML_TRY(h.chain.ensure(sizeof(double) * (size_t)round_up(C, 16) * 8));
        const size_t nchn = (size_t)(c.n + CM_ROWS - 1) / CM_ROWS, nchq = (size_t)(c.Q + cm_qrows(c.Q) - 1) / cm_qrows(c.Q);
        MCML_TRY(h.cm_part.ensure(sizeof(double) * (nchn + 3 * nchq + 4) * (size_t)h.V.ld));
        MCML_TRY(h.cm_acc.ensure(sizeof(int) * (size_t)round_up(C, 64)));
        if (c.sp.factored) { MCML_TRY(h.LX.alloc(C, c.Q)); MCML_TRY(h.ZS.alloc(C, c.Q)); }
        return MCML_OK;
    }
    MCML_TRY(h.V.alloc(c.Q, C)); MCML_TRY(h.R.alloc(c.Q, C)); MCML_TRY(h.UP.alloc(c.Q, C));
    MCML_TRY(h.GRAD.alloc(c.Q, C)); MCML_TRY(h.GRADP.alloc(c.Q, C));
    MCML_TRY(h.MU.alloc(c.n, C)); MCML_TRY(h.S.alloc(c.n, C));
    MCML_TRY(h.chain.ensure(sizeof(double) * (size_t)round_up(C, 16) * 8));
    // padding rows of the operands the GEMMs read must hold finite values
    MCML_HIP(hipMemsetAsync(h.V.d(), 0, sizeof(double) * (size_t)h.V.ld * C, c.stream));
    MCML_HIP(hipMemsetAsync(h.UP.d(), 0, sizeof(double) * (size_t)h.UP.ld * C, c.stream));
    MCML_HIP(hipMemsetAsync(h.S.d(), 0, sizeof(double) * (size_t)h.S.ld * C, c.stream));
    return MCML_OK;
}

// direct-to-LDS GEMM (dgemm_dlds.h) unless GLMMR_MCML_GEMM=reg asks for the register-staged one
static bool use_dlds()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("GLMMR_MCML_GEMM"); v = (e && !strcmp(e, "reg")) ? 0 : 1; }
    return v == 1;
}

static bool use_skinny()                       // GLMMR_MCML_SKINNY=0: the MFMA kernels whatever the column count
{                                              // (read per call: the parity tests run both paths in one process)
    const char* e = getenv("GLMMR_MCML_SKINNY");
    return !(e && !strcmp(e, "0"));
}

// MU = xb + ZL * X ; S = score
template <class Epi>
static int hmc_forward_launch(Ctx& c, const double* X, int ldx, const Epi& epi)
{
    HmcState& h = c.hmc;
    // at most 16 chains (chains = 1: the reference's layout; the tail of a NUTS doubling): an HBM-bound stream, not an MFMA tile
    // c.last_kernel[]: which kernel family served the product (tests assert the path they mean to compare)
    if (use_skinny() && skinny_applicable(c.plan_fwd, c.n, c.Q, h.Cw, c.ZL.ld)) {
        c.last_kernel[0] = KERNEL_SKINNY;
        return launch_skinny(c.stream, c.plan_fwd, h.Cw, c.ZL.d(), c.ZL.ld, X, ldx, epi);
    }
    if (c.band_fwd && dlds_applicable(c.n, h.Cw, c.Q, c.ZL.d(), c.ZL.ld, c.ZL.cols_alloc, X, ldx)) {
        c.last_kernel[0] = KERNEL_BAND;
        return launch_gemm_band(c.stream, c.plan_fwd, h.Cw, c.ZL.d(), c.ZL.ld, X, ldx, epi);
    }
    if (use_dlds() && dlds_applicable(c.n, h.Cw, c.Q, c.ZL.d(), c.ZL.ld, c.ZL.cols_alloc, X, ldx)) {
        c.last_kernel[0] = KERNEL_DLDS;
        return launch_gemm_dlds(c.stream, c.n, h.Cw, c.Q, c.ZL.d(), c.ZL.ld, X, ldx, epi);
    }
    c.last_kernel[0] = KERNEL_REG;
    return launch_gemm<false>(c.stream, c.n, h.Cw, c.Q, c.ZL.d(), c.ZL.ld, X, ldx, epi);
}

// want_ll (sparse operator only): leave the per-chain partial sums of log f(y | MU) in h.cm_part_fwd instead of MU
// lx_ready (factored sparse operator only): LX = L X is already there, left by k_cm_Lcol_Lrow of the previous leapfrog step
static int hmc_forward(Ctx& c, const double* X, int ldx, double var_par, bool store_mu = true, bool chain = false,
                       bool want_ll = false, bool lx_ready = false)
{
    HmcState& h = c.hmc;
    const int slot = c.prof.begin(c.stream, 0, chain);
    int rc;
    if (h.cm) {
        c.last_kernel[0] = KERNEL_SPARSE;
        const int rpw = CM_FR;
        dim3 grid((c.n + 4 * rpw - 1) / (4 * rpw), (h.Cw + 63) / 64);
        // factored operator: LX = L X first, then the rows of Z gather from LX
        int W = c.sp.W; const int* col = c.sp.ell_col.as<int>(); const double* val = c.sp.ell_val.d(); const double* Xin = X;
        if (c.sp.factored) {
            if (!lx_ready)
            hipLaunchKernelGGL(k_cm_Lrow, dim3((c.Q + 3) / 4, (h.Cw + 63) / 64), dim3(256), 0, c.stream, c.Q, h.Cw, h.V.ld,
                               c.sp.row_start.as<int>(), c.L.d(), c.L.ld, X, h.LX.d());
            W = c.z_width; col = c.z_idx.as<int>(); val = c.z_val.d(); Xin = h.LX.d();
        }
        double* pll = nullptr;
        if (want_ll) {
            MCML_TRY(h.cm_part_fwd.ensure(sizeof(double) * (size_t)grid.x * h.V.ld));
            pll = h.cm_part_fwd.d();
        }
#define MCML_CMF(FL) hipLaunchKernelGGL((k_cm_forward<FL>), grid, dim3(256), 0, c.stream, c.n, h.Cw, h.V.ld, W, col, val, Xin, \
                                        c.xb.d(), c.y.d(), c.flink, var_par, store_mu ? 1 : 0, h.MU.d(), h.S.d(), rpw, pll, h.V.ld)
        switch (c.flink) {
        case 1: MCML_CMF(1); break;
        case 3: MCML_CMF(3); break;
        case 7: MCML_CMF(7); break;
        case 12: MCML_CMF(12); break;
        default: MCML_CMF(0); break;
        }
#undef MCML_CMF
        rc = (hipGetLastError() == hipSuccess) ? MCML_OK : MCML_EHIP;
    } else {
#define MCML_FWD(FL) rc = hmc_forward_launch(c, X, ldx, EpiForwardT<FL>{h.MU.d(), h.S.d(), h.MU.ld, c.xb.d(), c.y.d(), c.flink, \
                                                                        store_mu ? 1 : 0, var_par})
        switch (c.flink) {
        case 1: MCML_FWD(1); break;
        case 3: MCML_FWD(3); break;
        case 7: MCML_FWD(7); break;
        case 12: MCML_FWD(12); break;
        default: MCML_FWD(0); break;
        }
#undef MCML_FWD
    }
    c.prof.end(c.stream, slot);
    return rc;
}

// GLMMR_MCML_CM_LFUSE=0: the factored operator's k_cm_Lcol and the next step's k_cm_Lrow as separate launches (the A/B switch)
static bool cm_lfuse(const Ctx& c)
{
    const char* e = getenv("GLMMR_MCML_CM_LFUSE");          // read per hmc_sample call: a test compares the two in one process
    const bool v = !(e && !strcmp(e, "0"));
    return v && c.sp.factored && c.sp.nblk > 0 && c.sp.max_blk <= 16;
}

// next_lx (factored sparse operator, inside a trajectory): also leave LX = L * UP for the next step's forward product
static int hmc_backward(Ctx& c, const double* Xs, double* G, int s, double var_par, int mode, bool chain = false,
                        bool next_lx = false)
{
    HmcState& h = c.hmc;
    ChainArrays ca = chain_arrays(h);
    EpiBackward epi{Xs, G, h.R.d(), h.UP.d(), h.V.ld, ca.e, ca.steps, s, glm_score_post(var_par, c.flink), mode};
    const int slot = c.prof.begin(c.stream, 1, chain);
    int rc;
    if (h.cm) {
        // factored operator: T = Z' S (mode 2: the raw sums), then g = -x + post * L' T with the leapfrog update
        const bool f = c.sp.factored;
        const int* ptr = f ? c.sp.zcsr_ptr.as<int>() : c.sp.csr_ptr.as<int>();
        const int* ci = f ? c.sp.zcsr_i.as<int>() : c.sp.csr_i.as<int>();
        const double* cv = f ? c.sp.zcsr_val.d() : c.sp.csr_val.d();
        double* out = f ? h.ZS.d() : G;
        const int m1 = f ? 2 : mode;
        const double post = glm_score_post(var_par, c.flink);
        if ((f ? c.sp.nnz_z : c.sp.nnz) >= 24L * c.Q) {          // long rows: a workgroup per (random effect, 64 chains)
            const int ncb = (h.Cw + 63) / 64;
            hipLaunchKernelGGL(k_cm_backward_long, dim3((c.Q * ncb + 7) / 8 * 8), dim3(256), 0, c.stream, c.Q, h.Cw, h.V.ld,
                               ptr, ci, cv, h.S.d(), Xs, out, h.R.d(), h.UP.d(), ca.e, ca.steps, s, post, m1, ncb);
        } else {
            const int rpw = 2;
            dim3 grid((c.Q + 4 * rpw - 1) / (4 * rpw), (h.Cw + 63) / 64);
            hipLaunchKernelGGL(k_cm_backward, grid, dim3(256), 0, c.stream, c.Q, h.Cw, h.V.ld, ptr, ci, cv, h.S.d(), Xs, out,
                               h.R.d(), h.UP.d(), ca.e, ca.steps, s, post, m1, rpw);
        }
        if (f && next_lx && mode == 1) {
            const dim3 grid((c.sp.nblk + 3) / 4, (h.Cw + 63) / 64);
            if (c.sp.max_blk <= 8)
                hipLaunchKernelGGL((k_cm_Lcol_Lrow<8>), grid, dim3(256), 0, c.stream, c.sp.nblk, h.Cw, h.V.ld, c.sp.blk_ptr.as<int>(),
                                   c.L.d(), c.L.ld, h.ZS.d(), Xs, G, h.R.d(), h.UP.d(), ca.e, ca.steps, s, post, h.LX.d());
            else
                hipLaunchKernelGGL((k_cm_Lcol_Lrow<16>), grid, dim3(256), 0, c.stream, c.sp.nblk, h.Cw, h.V.ld, c.sp.blk_ptr.as<int>(),
                                   c.L.d(), c.L.ld, h.ZS.d(), Xs, G, h.R.d(), h.UP.d(), ca.e, ca.steps, s, post, h.LX.d());
        } else if (f)
            hipLaunchKernelGGL(k_cm_Lcol, dim3((c.Q + 3) / 4, (h.Cw + 63) / 64), dim3(256), 0, c.stream, c.Q, h.Cw, h.V.ld,
                               c.sp.row_end.as<int>(), c.L.d(), c.L.ld, h.ZS.d(), Xs, G, h.R.d(), h.UP.d(), ca.e, ca.steps, s,
                               post, mode);
        rc = (hipGetLastError() == hipSuccess) ? MCML_OK : MCML_EHIP;
        c.last_kernel[1] = KERNEL_SPARSE;
        c.prof.end(c.stream, slot);
        return rc;
    }
    if (use_skinny() && skinny_applicable(c.plan_bwd, c.Q, c.n, h.Cw, c.ZLT.ld)) {
        c.last_kernel[1] = KERNEL_SKINNY;
        rc = launch_skinny(c.stream, c.plan_bwd, h.Cw, c.ZLT.d(), c.ZLT.ld, h.S.d(), h.S.ld, epi);
    } else if (c.band_bwd && dlds_applicable(c.Q, h.Cw, c.n, c.ZLT.d(), c.ZLT.ld, c.ZLT.cols_alloc, h.S.d(), h.S.ld)) {
        c.last_kernel[1] = KERNEL_BAND;
        rc = launch_gemm_band(c.stream, c.plan_bwd, h.Cw, c.ZLT.d(), c.ZLT.ld, h.S.d(), h.S.ld, epi);
    } else if (use_dlds() && dlds_applicable(c.Q, h.Cw, c.n, c.ZLT.d(), c.ZLT.ld, c.ZLT.cols_alloc, h.S.d(), h.S.ld)) {
        c.last_kernel[1] = KERNEL_DLDS;
        rc = launch_gemm_dlds(c.stream, c.Q, h.Cw, c.n, c.ZLT.d(), c.ZLT.ld, h.S.d(), h.S.ld, epi);
    } else {
        c.last_kernel[1] = KERNEL_REG;
        rc = launch_gemm<false>(c.stream, c.Q, h.Cw, c.n, c.ZLT.d(), c.ZLT.ld, h.S.d(), h.S.ld, epi);
    }
    c.prof.end(c.stream, slot);
    return rc;
}

// ---- chain-major helpers (sparse ZL operator, hmc_cm.h) ----
static CmChain cm_chain(const ChainArrays& a)
{
    return CmChain{a.e, a.ebar, a.H, a.lpcur, a.K0, a.steps, a.acc, a.gen, a.leap};
}
static int cm_fwd_chunks(const Ctx& c) { return (c.n + 4 * CM_FR - 1) / (4 * CM_FR); }     // workgroups of k_cm_forward = its ll partials
// GLMMR_MCML_CM_FUSE=0: the round-2 sequence (MU stored and read back by k_cm_logprob_partials, separate k_cm_commit): the A/B switch
static bool cm_fuse()
{
    static const bool v = !(getenv("GLMMR_MCML_CM_FUSE") && !strcmp(getenv("GLMMR_MCML_CM_FUSE"), "0"));
    return v;
}
struct CmParts { double *ll, *lp, *kin, *ss; int nchn, nchq, ldp; };
static CmParts cm_parts(const Ctx& c)
{
    const HmcState& h = c.hmc;
    CmParts p;
    p.nchn = (c.n + CM_ROWS - 1) / CM_ROWS; p.nchq = (c.Q + cm_qrows(c.Q) - 1) / cm_qrows(c.Q); p.ldp = h.V.ld;
    p.ll = h.cm_part.d(); p.lp = p.ll + (size_t)p.nchn * p.ldp; p.kin = p.lp + (size_t)p.nchq * p.ldp;
    p.ss = p.kin + (size_t)p.nchq * p.ldp;
    return p;
}
// partial sums of log f(y | MU) + log N(X; 0, 1) (+ R^2) of every chain
// skip_ll: the observation part came out of the last forward product (hmc_forward want_ll); only the prior / kinetic part runs
static int cm_logprob_partials(Ctx& c, const double* X, const double* R, double var_par, bool skip_ll = false)
{
    HmcState& h = c.hmc;
    const CmParts p = cm_parts(c);
    const int nchn = skip_ll ? 0 : p.nchn;
    const dim3 grid((h.Cw + 63) / 64, nchn + p.nchq);
#define MCML_LP_LAUNCH(FL) hipLaunchKernelGGL((k_cm_logprob_partials<FL>), grid, dim3(256), 0, c.stream, h.MU.d(), X, R, h.V.ld, \
                       c.n, c.Q, h.Cw, c.y.d(), var_par, c.flink, nchn, p.ll, p.lp, p.kin, p.ldp)
    switch (c.flink) {                       // the common families get their own instantiation
    case 1: MCML_LP_LAUNCH(1); break;        // poisson / log
    case 3: MCML_LP_LAUNCH(3); break;        // binomial / logit
    case 7: MCML_LP_LAUNCH(7); break;        // gaussian / identity
    default: MCML_LP_LAUNCH(0); break;
    }
#undef MCML_LP_LAUNCH
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

// log_prob and log_grad of every column of the current V
static int hmc_eval_state(Ctx& c, double var_par)
{
    HmcState& h = c.hmc;
    ChainArrays ca = chain_arrays(h);
    const bool fuse = h.cm && cm_fuse();
    MCML_TRY(hmc_forward(c, h.V.d(), h.V.ld, var_par, !fuse, false, fuse));
    if (h.cm) {
        const CmParts p = cm_parts(c);
        MCML_TRY(cm_logprob_partials(c, h.V.d(), nullptr, var_par, fuse));
        hipLaunchKernelGGL(k_cm_lp0_fin, dim3((h.C + 63) / 64), dim3(256), 0, c.stream, fuse ? h.cm_part_fwd.d() : p.ll, p.lp,
                           fuse ? cm_fwd_chunks(c) : p.nchn, p.nchq, p.ldp, h.C, ca.lpcur);
    } else
    MCML_FL_DISPATCH(c.flink, k_hmc_lp0, dim3(h.C), dim3(256), 0, c.stream, h.MU.d(), h.MU.ld, c.n, h.V.d(), h.V.ld, c.Q,
                       c.y.d(), var_par, c.flink, ca.lpcur);
    MCML_HIP(hipGetLastError());
    return hmc_backward(c, h.V.d(), h.GRAD.d(), 0, var_par, 0);
}

int hmc_sample(Ctx& c, const double* beta, double var_par, const glmmr_mcml_hmc_opts* o, uint64_t seed,
               uint32_t iter_idx, const double* inj_init, const double* inj_mom, uint8_t* flags_out,
               double* probs_out, glmmr_mcml_hmc_diag* diag, int* ncols_out)
{
    MCML_REQUIRE(c.n > 0 && c.have_L && (c.ZL.d() || c.sp.active), "hmc: model / L not set (call update_L or set_L first)");
    MCML_REQUIRE(o && o->warmup >= 0 && o->nsamp > 0 && o->max_steps >= 1 && o->lambda > 0,
                 "hmc: bad options");
    MCML_REQUIRE(beta, "hmc: beta is null");
    const int C = o->chains > 0 ? o->chains : 1;
    const int d = (C == 1) ? o->nsamp : (o->nsamp + C - 1) / C;   // draws per chain
    const int total = o->warmup + d;
    const int ncols = (C == 1) ? d + 1 : C * d;                   // mhmcmc.h:126: Q x (nsamp+1)
    const int Q = c.Q, n = c.n;
    HmcState& h = c.hmc;
    // GLMMR_MCML_HMC_TIMING=1: host wall-clock of this call's segments on stderr (set-up | proposals | tail), and of the slowest
    // proposal's enqueue -- to tell a slow call's cause from outside (run-to-run jitter)
    static const bool timing = getenv("GLMMR_MCML_HMC_TIMING") != nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    MCML_TRY(model_update_beta(c, beta));
    MCML_TRY(hmc_alloc(c, C));
    ChainArrays ca = chain_arrays(h);
    DevMat samp;
    MCML_TRY(samp.alloc(Q, ncols));
    MCML_HIP(hipMemsetAsync(samp.d(), 0, sizeof(double) * (size_t)samp.ld * ncols, c.stream));

    DevBuf d_init, d_mom, d_flags, d_probs;
    const double* p_init = nullptr; const double* p_mom = nullptr;
    if (inj_init) {
        MCML_TRY(d_init.ensure(sizeof(double) * (size_t)Q * C));
        MCML_TRY(copy_h2d(d_init.p, inj_init, sizeof(double) * (size_t)Q * C, c.stream));
        p_init = d_init.d();
    }
    if (inj_mom) {
        MCML_TRY(d_mom.ensure(sizeof(double) * (size_t)Q * C * total));
        MCML_TRY(copy_h2d(d_mom.p, inj_mom, sizeof(double) * (size_t)Q * C * total, c.stream));
        p_mom = d_mom.d();
    }
    if (flags_out) MCML_TRY(d_flags.ensure((size_t)C * total));
    if (probs_out) MCML_TRY(d_probs.ensure(sizeof(double) * (size_t)C * total));

    const int nchq = (Q + cm_qrows(Q) - 1) / cm_qrows(Q);
    auto store = [&](int stride, int col) {
        if (h.cm)      // SAMP[k + (c * stride + col) * lds] = V[c + k * ldc]
            hipLaunchKernelGGL(k_cm_transpose, dim3((Q + 31) / 32, (C + 31) / 32), dim3(256), 0, c.stream, h.V.d(), h.V.ld, Q, C,
                               samp.d(), (size_t)samp.ld, (size_t)(stride > 0 ? stride : 1), col);
        else
            hipLaunchKernelGGL(k_hmc_store, dim3(C), dim3(256), 0, c.stream, h.V.d(), h.V.ld, Q, samp.d(), samp.ld, stride, col);
    };
    if (h.cm)
        hipLaunchKernelGGL(k_cm_init, dim3((C + 63) / 64, nchq), dim3(256), 0, c.stream, h.V.d(), h.V.ld, Q, C, cm_chain(ca), seed,
                           (uint32_t)o->chain_offset, iter_idx, p_init);
    else
    hipLaunchKernelGGL(k_hmc_init, dim3(C), dim3(256), 0, c.stream, h.V.d(), h.V.ld, Q, ca, seed,
                       (uint32_t)o->chain_offset, iter_idx, p_init);
    MCML_HIP(hipGetLastError());
    MCML_TRY(hmc_eval_state(c, var_par));
    if (C == 1 && o->warmup == 0) store(0, 0);

    int* d_maxs = c.scalars.as<int>() + 34;
    // The number of leapfrog iterations to launch is the largest step count over the chains, a device
    // value.  Reading it back costs a host synchronisation per proposal (~50 us of idle GPU).  While
    // the step counts observed so far sit at the cap (lambda / e >= max_steps, the usual regime), the
    // cap itself is launched without waiting -- iterations beyond a chain's own count are masked
    // no-ops, so results are identical -- and the true value comes back on its own: k_max_steps stores
    // (proposal sequence number, count) into a ring of host memory mapped into the device (StepRing,
    // ctx.h), which the host reads with plain loads; an observation below the cap switches back to the
    // exact, synchronous path.  GLMMR_MCML_HMC_SPEC=0 disables the speculation.
    // (Until round 3 the read-back was a hipMemcpyAsync into a pinned ring allocated per call plus an
    // event per slot: in about one process in four ONE such enqueue stalled for 65-70 ms inside the
    // runtime -- config 4's "slow first repetition", DESIGN.md 6 -- and every call paid a hipHostMalloc,
    // four event creations and their release.  The loop now makes no HIP call besides kernel launches
    // and, on the synchronous path, the stream synchronisation.)
    static const bool spec_allowed = !(getenv("GLMMR_MCML_HMC_SPEC") && atoi(getenv("GLMMR_MCML_HMC_SPEC")) == 0);
    constexpr int RING = StepRing::SLOTS, AHEAD = 4;      // the host runs at most AHEAD proposals ahead of the last count it has seen
    StepRing& ring = h.ring;
    if (!ring.h) {
        MCML_HIP(hipHostMalloc((void**)&ring.h, sizeof(unsigned long long) * RING, hipHostMallocMapped | hipHostMallocCoherent));
        memset(ring.h, 0, sizeof(unsigned long long) * RING);
        MCML_HIP(hipHostGetDevicePointer((void**)&ring.d, ring.h, 0));
    }
    int seen_maxs = -1;                         // latest step count actually observed
    const unsigned seq0 = ring.seq + 1;          // sequence number of this call's first proposal
    unsigned seen_seq = seq0 - 1;                // newest proposal of this call whose count has arrived
    // Speculating costs a whole masked leapfrog step whenever the true count is below the cap (config 5: 290 us against ~30 us
    // for the wait it saves), so it needs evidence: SPEC_STREAK consecutive proposals at the cap, and the first count below it
    // ends it.  (With "the last count seen was at the cap" as the only condition a model whose longest chain hovers round the
    // cap flipped between the two paths by the timing of the read-back: config 5 measured 337-386 ms per iteration from one
    // process to the next on one box.)
    constexpr int SPEC_STREAK = 8;
    int streak = 0;                              // consecutive proposals observed at the cap
    auto observe = [&](unsigned sq, int v) { seen_seq = sq; seen_maxs = v; streak = (v == o->max_steps) ? streak + 1 : 0; };
    auto harvest = [&]() {                       // the counts that have arrived, in proposal order (at most AHEAD + 1 are outstanding)
        for (;;) {
            const unsigned want = seen_seq + 1;
            if ((int)(ring.seq - want) < 0) return;                  // nothing launched beyond what has been seen
            const unsigned long long tok = __atomic_load_n(ring.h + (want % RING), __ATOMIC_ACQUIRE);
            if ((unsigned)(tok >> 32) != want) return;               // not there yet
            observe(want, (int)(unsigned)tok);
        }
    };
    bool pending_commit = false;                // sparse operator: the last decisions are applied by the next k_cm_propose
    const bool lf = h.cm && cm_lfuse(c);        // factored operator: the backward pass of step s leaves LX for step s + 1
    const double t_setup = since(tc0);
    const auto tc1 = std::chrono::steady_clock::now();
    auto tp_prev = tc1;
    double t_sync = 0, t_slowest = 0; int n_sync = 0;
    double seg[5] = {0, 0, 0, 0, 0};   // slowest enqueue of: [0] propose launches, [1] count look-ahead (speculative path), [3] trajectory launches, [4] accept / commit / store
    auto mark = [&](int k, std::chrono::steady_clock::time_point t) { if (timing) { const double d = since(t); if (d > seg[k]) seg[k] = d; } };
    for (int it = 0; it < total; ++it) {
        const auto tp0 = std::chrono::steady_clock::now();
        if (it > 0 && timing) { const double d = since(tp_prev); if (d > t_slowest) t_slowest = d; }
        tp_prev = tp0;
        if (h.cm) {
            const CmParts p = cm_parts(c);
            hipLaunchKernelGGL(k_cm_propose, dim3((C + 63) / 64, p.nchq), dim3(256), 0, c.stream, h.V.d(), h.GRAD.d(), h.R.d(),
                               h.UP.d(), h.V.ld, Q, C, cm_chain(ca), seed, (uint32_t)o->chain_offset, iter_idx, it, p_mom,
                               p.ss, p.ldp, pending_commit ? h.cm_acc.as<int>() : nullptr, h.GRADP.d());
            pending_commit = false;
            hipLaunchKernelGGL(k_cm_propose_fin, dim3((C + 63) / 64), dim3(256), 0, c.stream, p.ss, p.nchq, p.ldp, C,
                               cm_chain(ca), o->lambda, o->max_steps);
        } else
        hipLaunchKernelGGL(k_hmc_propose, dim3(C), dim3(256), 0, c.stream, h.V.d(), h.GRAD.d(), h.R.d(), h.UP.d(),
                           h.V.ld, Q, ca, o->lambda, o->max_steps, seed, (uint32_t)o->chain_offset, iter_idx, it,
                           p_mom, C);
        const unsigned seq = ++ring.seq;
        const int slot = (int)(seq % RING);
        hipLaunchKernelGGL(k_max_steps, dim3(1), dim3(256), 0, c.stream, ca.steps, C, d_maxs, ring.d + slot, seq);
        MCML_HIP(hipGetLastError());
        mark(0, tp0);
        const auto tq0 = std::chrono::steady_clock::now();
        harvest();
        int maxs = 0;
        bool spec = spec_allowed && streak >= SPEC_STREAK;
        if (spec) {
            // bounded look-ahead: a count below the cap must be noticed within AHEAD proposals (in the dense path a masked
            // step is a full product).  Plain loads of host memory; a count that does not arrive falls back to the wait below
            const auto tw0 = std::chrono::steady_clock::now();
            while ((int)(seq - seen_seq) > AHEAD) {
                harvest();
                if ((int)(seq - seen_seq) > AHEAD && since(tw0) > 2000.0) { spec = false; break; }
            }
            if (spec && streak < SPEC_STREAK) spec = false;
        }
        if (spec) {
            maxs = o->max_steps;
            mark(1, tq0);
        } else {
            const auto ts0 = std::chrono::steady_clock::now();
            MCML_HIP(hipStreamSynchronize(c.stream));
            if (timing) { t_sync += since(ts0); ++n_sync; }
            const unsigned long long tok = __atomic_load_n(ring.h + slot, __ATOMIC_ACQUIRE);
            MCML_REQUIRE((unsigned)(tok >> 32) == seq, "hmc: the step count of proposal %d did not arrive (token %llx, expected sequence %u)", it, tok, seq);
            maxs = (int)(unsigned)tok;
            harvest();                                               // everything up to and including this proposal, in order
            MCML_REQUIRE(seen_seq == seq && seen_maxs == maxs, "hmc: step-count ring out of order");
        }
        MCML_REQUIRE(maxs >= 1 && maxs <= o->max_steps, "hmc: step count %d out of range", maxs);
        // kernel timing (bench.py's roofline): every marker between two dependent launches costs ~2.5 us of idle GPU,
        // so one proposal in four is timed -- still hundreds of launches per MCML iteration behind the average
        c.prof.skip = (it & 3) != 0;
        const auto tt0 = std::chrono::steady_clock::now();
        int rc_traj = MCML_OK;
        for (int s = 0; s < maxs && rc_traj == MCML_OK; ++s) {
            const bool fuse = h.cm && cm_fuse();
            rc_traj = hmc_forward(c, h.UP.d(), h.UP.ld, var_par, !fuse && s == maxs - 1, s > 0, fuse && s == maxs - 1, lf && s > 0);
            if (rc_traj == MCML_OK) rc_traj = hmc_backward(c, h.UP.d(), h.GRADP.d(), s, var_par, 1, true, lf && s + 1 < maxs);
        }
        c.prof.skip = false;
        mark(3, tt0);
        const auto ta0 = std::chrono::steady_clock::now();
        MCML_TRY(rc_traj);
        c.prof.unchain();
        const int adapt = (it < o->warmup) && (it < o->adapt);     // mhmcmc.h:131-136
        if (h.cm) {
            const CmParts p = cm_parts(c);
            const bool fuse = cm_fuse();
            MCML_TRY(cm_logprob_partials(c, h.UP.d(), h.R.d(), var_par, fuse));
            hipLaunchKernelGGL(k_cm_accept_fin, dim3((C + 63) / 64), dim3(256), 0, c.stream, fuse ? h.cm_part_fwd.d() : p.ll, p.lp, p.kin,
                               fuse ? cm_fwd_chunks(c) : p.nchn, p.nchq, p.ldp, C, cm_chain(ca), o->target_accept, adapt, it,
                               flags_out ? d_flags.as<uint8_t>() : nullptr, probs_out ? d_probs.d() : nullptr,
                               h.cm_acc.as<int>());
            // the accepted chains' V <- UP, GRAD <- GRADP: folded into the next proposal's first pass unless V is read
            // before that (a draw is stored after this proposal, or it is the last one)
            const bool stores_now = (C == 1) ? (it >= o->warmup - 1) : (it >= o->warmup);
            if (fuse && it + 1 < total && !stores_now) pending_commit = true;
            else hipLaunchKernelGGL(k_cm_commit, dim3((C + 63) / 64, p.nchq), dim3(256), 0, c.stream, h.V.d(), h.GRAD.d(), h.UP.d(),
                                    h.GRADP.d(), h.V.ld, Q, C, h.cm_acc.as<int>());
        } else
        MCML_FL_DISPATCH(c.flink, k_hmc_accept, dim3(C), dim3(256), 0, c.stream, h.V.d(), h.GRAD.d(), h.R.d(), h.UP.d(),
                           h.GRADP.d(), h.V.ld, Q, h.MU.d(), h.MU.ld, n, c.y.d(), var_par, c.flink, ca,
                           o->target_accept, adapt, it, C, flags_out ? d_flags.as<uint8_t>() : nullptr,
                           probs_out ? d_probs.d() : nullptr);
        int col = -1, stride = 0;
        if (C == 1) {
            if (it == o->warmup - 1) col = 0;                      // samples.col(0) = u_, :142
            else if (it >= o->warmup) col = it - o->warmup + 1;    // samples.col(i+1) = u_, :147
        } else if (it >= o->warmup) { col = it - o->warmup; stride = d; }
        if (col >= 0) store(stride, col);
        MCML_HIP(hipGetLastError());
        mark(4, ta0);
    }
    const double t_loop = since(tc1);
    const auto tc2 = std::chrono::steady_clock::now();
    // return (L * samples)  (mhmcmc.h:155)
    MCML_TRY(c.U.alloc(Q, ncols));
    MCML_HIP(hipMemsetAsync(c.U.d(), 0, sizeof(double) * (size_t)c.U.ld * ncols, c.stream));
    if (c.sp.active && c.sp.row_start.p) {
        int gy = ncols < 1024 ? ncols : 1024;
        hipLaunchKernelGGL(k_blockdiag_LV, dim3((Q + 255) / 256, gy), dim3(256), 0, c.stream, Q, ncols,
                           c.sp.row_start.as<int>(), c.L.d(), c.L.ld, samp.d(), samp.ld, c.U.d(), c.U.ld);
        MCML_HIP(hipGetLastError());
    } else {
        EpiAxpby epi{c.U.d(), c.U.ld, 1.0, 0.0};
        MCML_TRY(launch_gemm<false>(c.stream, Q, ncols, Q, c.L.d(), c.L.ld, samp.d(), samp.ld, epi));
    }
    c.mcols = ncols;
    c.niter = (C == 1) ? d : ncols;                                // mcmlmodel.h:73 vs mhmcmc.h:126 (D5)
    c.zu_valid = false; c.uall_valid = false;
    if (flags_out) MCML_TRY(copy_d2h(flags_out, d_flags.p, (size_t)C * total, c.stream));
    if (probs_out) MCML_TRY(copy_d2h(probs_out, d_probs.p, sizeof(double) * (size_t)C * total, c.stream));
    double dg[6] = {0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_hmc_diag, dim3(1), dim3(64), 0, c.stream, ca, C, c.scalars.d() + 8);
    MCML_TRY(copy_d2h(dg, c.scalars.d() + 8, sizeof dg, c.stream));
    MCML_HIP(hipStreamSynchronize(c.stream));
    if (timing)
        fprintf(stderr, "hmc_sample: set-up %.2f ms | %d proposals %.2f ms enqueue (%d synchronous, %.2f ms waiting; slowest proposal %.2f ms: propose %.2f, count look-ahead %.2f, trajectory %.2f, accept %.2f) | tail %.2f ms\n",
                t_setup, total, t_loop, n_sync, t_sync, t_slowest, seg[0], seg[1], seg[3], seg[4], since(tc2));
    c.prof.collect();
    if (diag) {
        diag->accept_rate = dg[0] / ((double)C * total);
        diag->mean_e = dg[1] / C; diag->min_e = dg[2]; diag->max_e = dg[3];
        diag->max_steps_used = (int)dg[4]; diag->leapfrog_total = (long long)dg[5];
    }
    if (ncols_out) *ncols_out = ncols;
    return MCML_OK;
}

// test hook: log_prob (A4) and log_grad (A5) of every column of V
int hmc_dbg_log_prob_grad(Ctx& c, const double* beta, double var_par, const double* V, int ncols, double* lp,
                          double* G)
{
    MCML_REQUIRE(c.n > 0 && c.have_L && (c.ZL.d() || c.sp.active), "log_prob_grad: model / L not set");
    MCML_TRY(model_update_beta(c, beta));
    MCML_TRY(hmc_alloc(c, ncols));
    HmcState& h = c.hmc;
    ChainArrays ca = chain_arrays(h);
    if (h.cm) {
        DevMat tmp;                                   // Q x ncols column-major staging
        MCML_TRY(tmp.alloc(c.Q, ncols));
        MCML_TRY(copy_h2d_2d(tmp.d(), sizeof(double) * tmp.ld, V, sizeof(double) * c.Q, sizeof(double) * c.Q, ncols, c.stream));
        // V[c + q * ldc] = tmp[q + c * ld]
        hipLaunchKernelGGL(k_cm_transpose, dim3((ncols + 31) / 32, (c.Q + 31) / 32), dim3(256), 0, c.stream, tmp.d(), tmp.ld,
                           ncols, c.Q, h.V.d(), (size_t)h.V.ld, (size_t)1, 0);
        MCML_HIP(hipGetLastError());
        MCML_TRY(hmc_eval_state(c, var_par));
        hipLaunchKernelGGL(k_cm_transpose, dim3((c.Q + 31) / 32, (ncols + 31) / 32), dim3(256), 0, c.stream, h.GRAD.d(), h.GRAD.ld,
                           c.Q, ncols, tmp.d(), (size_t)tmp.ld, (size_t)1, 0);
        MCML_HIP(hipGetLastError());
        MCML_TRY(copy_d2h(lp, ca.lpcur, sizeof(double) * ncols, c.stream));
        MCML_TRY(download_matrix(G, c.Q, tmp.d(), tmp.ld, c.Q, ncols, c.stream));
        return c.sync();                              // tmp goes out of scope
    }
    MCML_TRY(copy_h2d_2d(h.V.d(), sizeof(double) * h.V.ld, V, sizeof(double) * c.Q, sizeof(double) * c.Q, ncols, c.stream));
    MCML_TRY(hmc_eval_state(c, var_par));
    MCML_TRY(copy_d2h(lp, ca.lpcur, sizeof(double) * ncols, c.stream));
    return download_matrix(G, c.Q, h.GRAD.d(), h.GRAD.ld, c.Q, ncols, c.stream);
}

}  // namespace mcml

#include "nuts.h"
