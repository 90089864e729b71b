// hmc_cm.h -- the sampler's kernels for the SPARSE ZL operator (configs 1, 4, 5) with the HMC state stored
// CHAIN-MAJOR: element (chain c, row r) of V, R, UP, GRAD, GRADP (rows = random effects) and of MU, S (rows =
// observations) lives at  c + r * ldc,  64 consecutive chains = one wave's 512 contiguous bytes.
//
// Why: the sparse products are HBM-bound gathers.  With the dense layout (chain = column) a thread owns a row and
// every lane carries its own row's pointers, indices and values: 11 load instructions per element of which 3 are HBM
// streams (profiles/r02_cfg5_*: 2.1-3.0 TB/s).  Chain-major, a wave is 64 chains of ONE row: the row's metadata is
// wave-uniform (scalar loads), every state access is one coalesced 512-byte line, and the sums over a row's entries
// need no cross-lane reduction.  Measured on the config-5 backward shape (scripts/sp_proto.hip): 184 us = 4.8 TB/s
// algorithmic against 401 us.
//
// Per-chain sums (log-density, kinetic energy) are two-stage with a fixed order: a workgroup = 64 chains x 4 waves
// owns SUM_ROWS consecutive rows and leaves one partial per (row chunk, chain); the finishing kernels add the
// chunks in order.  The RNG is keyed by (element, GLOBAL chain, proposal, tag) exactly as in the dense path, so
// the momenta and accept streams are bit-identical to it and to the oracle.
#pragma once
#include "ctx.h"
#include "glm.h"
#include "rng.h"

namespace mcml {

constexpr int CM_ROWS = 64;       // rows per workgroup in the elementwise / partial-sum kernels (16 per wave)

struct CmChain {                  // the per-chain scalars of hmc.hip::ChainArrays
    double *e, *ebar, *H, *lpcur, *K0;
    int *steps, *acc;
    uint32_t* gen;
    long long* leap;
};

// ---- scalar loads of wave-uniform, read-only metadata --------------------------------------------------------------
// hipcc keeps such loads on the vector pipe (64 lanes fetching one address) because it cannot prove that the kernel's
// stores do not alias them; these issue them on the scalar pipe, a batch at a time with one wait.  The pointers must be
// wave-uniform (SGPR operands); dword / dwordx2 loads need 4-byte alignment only.
__device__ __forceinline__ void sload8(const int* ip, const double* dp, int (&iv)[8], double (&dv)[8])
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dword %0, %16, 0x0\n\ts_load_dword %1, %16, 0x4\n\ts_load_dword %2, %16, 0x8\n\ts_load_dword %3, %16, 0xc\n\t"
                 "s_load_dword %4, %16, 0x10\n\ts_load_dword %5, %16, 0x14\n\ts_load_dword %6, %16, 0x18\n\ts_load_dword %7, %16, 0x1c\n\t"
                 "s_load_dwordx2 %8, %17, 0x0\n\ts_load_dwordx2 %9, %17, 0x8\n\ts_load_dwordx2 %10, %17, 0x10\n\ts_load_dwordx2 %11, %17, 0x18\n\t"
                 "s_load_dwordx2 %12, %17, 0x20\n\ts_load_dwordx2 %13, %17, 0x28\n\ts_load_dwordx2 %14, %17, 0x30\n\ts_load_dwordx2 %15, %17, 0x38\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(iv[0]), "=&s"(iv[1]), "=&s"(iv[2]), "=&s"(iv[3]), "=&s"(iv[4]), "=&s"(iv[5]), "=&s"(iv[6]), "=&s"(iv[7]),
                   "=&s"(dv[0]), "=&s"(dv[1]), "=&s"(dv[2]), "=&s"(dv[3]), "=&s"(dv[4]), "=&s"(dv[5]), "=&s"(dv[6]), "=&s"(dv[7])
                 : "s"(ip), "s"(dp) : "memory");
#else
    for (int u = 0; u < 8; ++u) { iv[u] = ip[u]; dv[u] = dp[u]; }
#endif
}
__device__ __forceinline__ void sload_f64x2(const double* a, const double* b, double& x, double& y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(x), "=&s"(y) : "s"(a), "s"(b) : "memory");      // early clobber: an output must not reuse a pointer's registers
#else
    x = *a; y = *b;
#endif
}

// ---- products -------------------------------------------------------------------------------------------------
// forward: MU[c, i] = xb_i + sum_k val_k X[c, col_k] ; S = score(y_i, MU)      (ELL row of observation i)
// a wave owns CM_FR consecutive observations: the k loop is outermost so that CM_FR independent gathers are in
// flight per iteration, then CM_FR independent scores and stores
constexpr int CM_FR = 4;       // 8 measured slower (152 vs 121 us at config 5)
// part_ll (nullable): the workgroup's 16 observations' log f(y_i | MU[c, i]) summed per chain -- rows in order inside a wave,
// the four waves in wave order through LDS -- at part_ll[blockIdx.x * ldp + c].  On the last step of a trajectory the
// accept kernel needs exactly that sum: taking it here saves the store of MU (n x C) and the pass that read it back
// (k_cm_logprob_partials: 173 us per proposal at config 5).
template <int FL>      // family / link code at compile time (12 = beta/logit), 0 = run-time code (EpiForwardT, hmc.hip)
__global__ __launch_bounds__(256) void k_cm_forward(int n, int C, int ldc, int W, const int* col, const double* val,
                                                    const double* X, const double* xb, const double* y, int flink,
                                                    double var_par, int store_mu, double* MU, double* S, int rpw,
                                                    double* part_ll, int ldp)
{
    (void)rpw;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.y * 64 + lane;
    const int i0 = (blockIdx.x * 4 + w) * CM_FR;
    double llsum = 0.0;
    if (c < C && i0 < n) {
    double acc[CM_FR];
#pragma unroll
    for (int r = 0; r < CM_FR; ++r) acc[r] = 0.0;
    // the row metadata is wave-uniform and read-only: hand-written SCALAR loads (hipcc keeps them on the vector
    // pipe because it cannot prove that the stores to MU / S do not alias them): CM_FR columns and values per
    // batch, one wait
    const bool full = i0 + CM_FR <= n;
    for (int k = 0; k < W; ++k) {
        int qv[CM_FR]; double vv[CM_FR];
        if (full) {
            const int* cp_ = col + i0 + (size_t)k * n;
            const double* vp_ = val + i0 + (size_t)k * n;
#if defined(__HIP_DEVICE_COMPILE__)
            static_assert(CM_FR == 4, "scalar batch below");
            typedef int i4s_ __attribute__((ext_vector_type(4)));
            typedef double d2s_ __attribute__((ext_vector_type(2)));
            if ((((size_t)cp_) & 15) == 0 && (((size_t)vp_) & 15) == 0) {
                i4s_ cq; d2s_ v01, v23;
                asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %4, 0x0\n\ts_load_dwordx4 %2, %4, 0x10\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(cq), "=&s"(v01), "=&s"(v23) : "s"(cp_), "s"(vp_) : "memory");
                qv[0] = cq.x; qv[1] = cq.y; qv[2] = cq.z; qv[3] = cq.w;
                vv[0] = v01.x; vv[1] = v01.y; vv[2] = v23.x; vv[3] = v23.y;
            } else
#endif
            {
#pragma unroll
                for (int r = 0; r < CM_FR; ++r) { qv[r] = __builtin_amdgcn_readfirstlane(cp_[r]); vv[r] = vp_[r]; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < CM_FR; ++r) {
                const int i = (i0 + r < n) ? i0 + r : n - 1;
                qv[r] = __builtin_amdgcn_readfirstlane(col[i + (size_t)k * n]);
                vv[r] = val[i + (size_t)k * n];
            }
        }
        double xv[CM_FR];
#pragma unroll
        for (int r = 0; r < CM_FR; ++r) xv[r] = X[c + (size_t)qv[r] * ldc];
#pragma unroll
        for (int r = 0; r < CM_FR; ++r) acc[r] += vv[r] * xv[r];
    }
#pragma unroll
    for (int r = 0; r < CM_FR; ++r) {
        const int i = i0 + r;
        if (i < n) {
            double xbi, yi;
            sload_f64x2(xb + i, y + i, xbi, yi);
            const double mu = xbi + acc[r];
            const size_t off = c + (size_t)i * ldc;
            if (store_mu) MU[off] = mu;
            if constexpr (FL == 12) S[off] = glm_score_beta(yi, mu, var_par);
            else S[off] = glm_score(yi, mu, FL ? FL : flink);
            if (part_ll) llsum += glm_logpdf(yi, mu, var_par, FL ? FL : flink);
        }
    }
    }
    if (part_ll) {                                    // uniform: every wave of the workgroup gets here
        __shared__ double sh[4][64];
        sh[w][lane] = llsum;
        __syncthreads();
        if (w == 0 && c < C) part_ll[(size_t)blockIdx.x * ldp + c] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
    }
}

// the update that follows the backward product (hmc.hip::EpiBackward): acc = (ZL' S)[c, q]
//   mode 0: G = -x + post * acc ; mode 1: the same inside a leapfrog trajectory of st steps (step s) with the momentum and
//   position updates ; mode 2: G = acc (the first factor of the factored operator, below)
__device__ __forceinline__ void cm_leapfrog(double acc, size_t off, const double* Xs, double* G, double* R, double* UP,
                                            double en, int st, int s, double post, int mode)
{
    if (mode == 2) { G[off] = acc; return; }
    if (mode == 1 && s >= st) return;
    const double x = Xs[off];
    double g = -1.0 * x;
    g = g + post * acc;
    if (mode != 1 || s + 1 >= st) G[off] = g;                          // mid-trajectory gradients are never read
    if (mode == 1) {
        double rr = R[off];
        rr = rr + (en / 2) * g;
        if (s + 1 < st) { rr = rr + (en / 2) * g; UP[off] = x + en * rr; }
        R[off] = rr;
    }
}

// backward: g = -x + post * sum_t val_t S[c, i_t]  (CSR row q of ZL'), then the leapfrog update of hmc.hip::EpiBackward
__global__ __launch_bounds__(256) void k_cm_backward(int Q, int C, int ldc, const int* ptr, const int* ci,
                                                     const double* val, const double* S, const double* Xs, double* G,
                                                     double* R, double* UP, const double* e, const int* steps, int s,
                                                     double post, int mode, int rpw)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const int q0 = (blockIdx.x * 4 + w) * rpw;
    if (c >= C) return;
    int st = 0; double en = 0.0;
    if (mode == 1) { st = steps[c]; en = e[c]; }
    for (int q = q0; q < q0 + rpw && q < Q; ++q) {
        const int t0 = __builtin_amdgcn_readfirstlane(ptr[q]), t1 = __builtin_amdgcn_readfirstlane(ptr[q + 1]);
        double acc = 0.0;
        int t = t0;
        for (; t + 4 <= t1; t += 4) {                                  // four gathers in flight
            const int i0 = __builtin_amdgcn_readfirstlane(ci[t]), i1 = __builtin_amdgcn_readfirstlane(ci[t + 1]);
            const int i2 = __builtin_amdgcn_readfirstlane(ci[t + 2]), i3 = __builtin_amdgcn_readfirstlane(ci[t + 3]);
            const double s0 = S[c + (size_t)i0 * ldc], s1 = S[c + (size_t)i1 * ldc];
            const double s2 = S[c + (size_t)i2 * ldc], s3 = S[c + (size_t)i3 * ldc];
            acc += val[t] * s0; acc += val[t + 1] * s1; acc += val[t + 2] * s2; acc += val[t + 3] * s3;
        }
        for (; t < t1; ++t) acc += val[t] * S[c + (size_t)__builtin_amdgcn_readfirstlane(ci[t]) * ldc];
        cm_leapfrog(acc, c + (size_t)q * ldc, Xs, G, R, UP, en, st, s, post, mode);
    }
}

// backward for long rows (tens to hundreds of observations per random effect, config 4): a workgroup = ONE random
// effect x 64 chains; its four waves split the row's entries (contiguous quarters), eight gathers in flight each; the
// four partial sums are added in wave order through LDS (fixed order) and wave 0 applies the leapfrog update
__global__ __launch_bounds__(256) void k_cm_backward_long(int Q, int C, int ldc, const int* ptr, const int* ci,
                                                          const double* val, const double* S, const double* Xs, double* G,
                                                          double* R, double* UP, const double* e, const int* steps, int s,
                                                          double post, int mode, int ncb)
{
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup id -> (effect, chain block): ids go round-robin over the 8 XCDs, so XCD x takes the CONTIGUOUS range
    // [x, x + 1) * per8 of the (chain block, effect) order -- the effects of one covariance block gather the same rows
    // of S and now do so through one L2
    const int T = Q * ncb, per8 = (T + 7) >> 3;
    const int vid = (int)(blockIdx.x & 7) * per8 + (int)(blockIdx.x >> 3);
    if (vid >= T) return;
    const int q = vid % Q;
    const int c = (vid / Q) * 64 + lane;
    const bool cin = c < C;
    const int cc = cin ? c : 0;
    const int t0 = __builtin_amdgcn_readfirstlane(ptr[q]), t1 = __builtin_amdgcn_readfirstlane(ptr[q + 1]);
    const int len = t1 - t0, per = (len + 3) >> 2;
    const int a = t0 + w * per, b = (a + per < t1) ? a + per : t1;
    double acc = 0.0;
    int t = a;
    for (; t + 8 <= b; t += 8) {
        int iv[8]; double vv[8], sv[8];
        sload8(ci + t, val + t, iv, vv);                               // indices and values on the scalar pipe
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] = S[cc + (size_t)iv[u] * ldc];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += vv[u] * sv[u];
    }
    for (; t < b; ++t) acc += val[t] * S[cc + (size_t)__builtin_amdgcn_readfirstlane(ci[t]) * ldc];
    sh[w][lane] = acc;
    __syncthreads();
    if (w != 0 || !cin) return;
    acc = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
    int st = 0; double en = 0.0;
    if (mode == 1) { st = steps[c]; en = e[c]; }
    cm_leapfrog(acc, c + (size_t)q * ldc, Xs, G, R, UP, en, st, s, post, mode);
}

// (A block-at-a-time variant -- one workgroup per covariance block, S read once, the block's effects in DMAX
// accumulators with wave-uniform values -- measured 121 us against 72 us for the kernel above at config 4: the
// scalar index loads per observation cost more than the re-reads of S from L2 save.)

// ---- factored operator ZL = Z * L ------------------------------------------------------------------------------------
// When the covariance blocks are not tiny a row of ZL repeats a row of L for every observation of that random effect
// (config 4: 16000 x 8.5 entries for 320 x 8.5 distinct values).  The sampler then applies the two factors in turn:
//   forward   LX = L X        (k_cm_Lrow, Q x C, tiny)      MU = xb + Z LX    (k_cm_forward on the rows of Z)
//   backward  T  = Z' S       (k_cm_backward*, mode 2)      g  = -x + post * L' T, leapfrog update (k_cm_Lcol)
// so that X / S are gathered nnz(Z) times instead of nnz(ZL) times.
__device__ __forceinline__ double sload_f64(const double* a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double x;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(x) : "s"(a) : "memory");
    return x;
#else
    return *a;
#endif
}

// LX[c, q] = sum_{j = start(q)}^{q} L[q, j] X[c, j]  (row q of the block-diagonal lower-triangular L, ascending j)
__global__ __launch_bounds__(256) void k_cm_Lrow(int Q, int C, int ldc, const int* start, const double* L, int ldl,
                                                 const double* X, double* LX)
{
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 4 + w;
    const int c = blockIdx.y * 64 + lane;
    if (q >= Q || c >= C) return;
    const int s0 = __builtin_amdgcn_readfirstlane(start[q]);
    double acc = 0.0;
    int j = s0;
    for (; j + 4 <= q + 1; j += 4) {
        const double l0 = sload_f64(L + q + (size_t)j * ldl), l1 = sload_f64(L + q + (size_t)(j + 1) * ldl);
        const double l2 = sload_f64(L + q + (size_t)(j + 2) * ldl), l3 = sload_f64(L + q + (size_t)(j + 3) * ldl);
        const double x0 = X[c + (size_t)j * ldc], x1 = X[c + (size_t)(j + 1) * ldc];
        const double x2 = X[c + (size_t)(j + 2) * ldc], x3 = X[c + (size_t)(j + 3) * ldc];
        acc += l0 * x0; acc += l1 * x1; acc += l2 * x2; acc += l3 * x3;
    }
    for (; j <= q; ++j) acc += sload_f64(L + q + (size_t)j * ldl) * X[c + (size_t)j * ldc];
    LX[c + (size_t)q * ldc] = acc;
}

// acc = sum_{j = q}^{end(q) - 1} L[j, q] T[c, j]  (column q of L, ascending j), then the update of cm_leapfrog
__global__ __launch_bounds__(256) void k_cm_Lcol(int Q, int C, int ldc, const int* end, const double* L, int ldl,
                                                 const double* T, const double* Xs, double* G, double* R, double* UP,
                                                 const double* e, const int* steps, int s, double post, int mode)
{
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 4 + w;
    const int c = blockIdx.y * 64 + lane;
    if (q >= Q || c >= C) return;
    const int j1 = __builtin_amdgcn_readfirstlane(end[q]);
    const double* Lq = L + (size_t)q * ldl;
    double acc = 0.0;
    int j = q;
    for (; j + 4 <= j1; j += 4) {
        const double l0 = sload_f64(Lq + j), l1 = sload_f64(Lq + j + 1), l2 = sload_f64(Lq + j + 2), l3 = sload_f64(Lq + j + 3);
        const double t0 = T[c + (size_t)j * ldc], t1 = T[c + (size_t)(j + 1) * ldc];
        const double t2 = T[c + (size_t)(j + 2) * ldc], t3 = T[c + (size_t)(j + 3) * ldc];
        acc += l0 * t0; acc += l1 * t1; acc += l2 * t2; acc += l3 * t3;
    }
    for (; j < j1; ++j) acc += sload_f64(Lq + j) * T[c + (size_t)j * ldc];
    int st = 0; double en = 0.0;
    if (mode == 1) { st = steps[c]; en = e[c]; }
    cm_leapfrog(acc, c + (size_t)q * ldc, Xs, G, R, UP, en, st, s, post, mode);
}

// k_cm_Lcol of leapfrog step s and k_cm_Lrow of step s + 1 in one launch (a trajectory of the factored operator): a WAVE owns
// one covariance block x 64 chains.  The block of L goes through LDS once (read back as broadcasts), the block's rows of T and the
// updated positions stay in registers, so that LX = L * UP of the next step comes out of the same kernel that produced UP -- three
// launches per leapfrog step instead of four (config 4: k_cm_Lcol 4.9 us + k_cm_Lrow 4.6 us + a kernel boundary).  The sums run in
// the order of the two kernels above (ascending j, one fma per term), so the results are theirs to the bit.  DMAX >= the largest
// block; models with larger blocks keep the separate kernels (hmc.hip).
template <int DMAX>
__global__ __launch_bounds__(256) void k_cm_Lcol_Lrow(int B, int C, int ldc, const int* blk_ptr, const double* L, int ldl,
                                                      const double* T, const double* Xs, double* G, double* R, double* UP,
                                                      const double* e, const int* steps, int s, double post, double* LX)
{
    __shared__ double Ls[4][DMAX * DMAX];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * 4 + w;
    const bool bin = b < B;
    const int r0 = bin ? __builtin_amdgcn_readfirstlane(blk_ptr[b]) : 0;
    const int d = bin ? __builtin_amdgcn_readfirstlane(blk_ptr[b + 1]) - r0 : 0;
    const int c = blockIdx.y * 64 + lane;
    const bool valid = bin && c < C;
    // every load first (the block of L for LDS and 3 d independent loads of the chain's state, all in flight together), then the
    // arithmetic, then the stores: the stores may alias the loads as far as the compiler knows (Xs IS UP), and a load behind
    // each store made this a chain of d memory round trips
    const int st = valid ? steps[c] : 0;
    const double en = valid ? e[c] : 0.0;
    const bool live = s < st, last = s + 1 >= st;
    double t[DMAX], x[DMAX], rr[DMAX], g[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; ++i)
        if (i < d && valid) {
            const size_t off = c + (size_t)(r0 + i) * ldc;
            t[i] = T[off]; x[i] = Xs[off]; rr[i] = live ? R[off] : 0.0;
        }
    for (int idx = lane; idx < d * d; idx += 64) {
        const int i = idx % d, j = idx / d;
        Ls[w][i + j * DMAX] = (i >= j) ? L[(size_t)(r0 + i) + (size_t)(r0 + j) * ldl] : 0.0;
    }
    __syncthreads();                                   // every wave gets here (no early exit above)
    if (!valid) return;
#pragma unroll
    for (int q = 0; q < DMAX; ++q) {
        if (q < d) {
            double acc = 0.0;
#pragma unroll
            for (int j = q; j < DMAX; ++j) if (j < d) acc += Ls[w][j + q * DMAX] * t[j];       // column q of L, ascending j
            if (live) {                                // cm_leapfrog, mode 1
                double gq = -1.0 * x[q];
                gq = gq + post * acc;
                double r = rr[q];
                r = r + (en / 2) * gq;
                if (!last) { r = r + (en / 2) * gq; x[q] = x[q] + en * r; }
                rr[q] = r; g[q] = gq;
            }
        }
    }
    if (live) {
#pragma unroll
        for (int q = 0; q < DMAX; ++q)
            if (q < d) {
                const size_t off = c + (size_t)(r0 + q) * ldc;
                if (last) G[off] = g[q]; else UP[off] = x[q];
                R[off] = rr[q];
            }
    }
#pragma unroll
    for (int q = 0; q < DMAX; ++q) {
        if (q < d) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j <= q; ++j) acc += Ls[w][q + j * DMAX] * x[j];                    // row q of L, ascending j
            LX[c + (size_t)(r0 + q) * ldc] = acc;
        }
    }
}

// ---- per-chain kernels -------------------------------------------------------------------------------------------
// grid (chains / 64, row chunks of CM_ROWS); wave w of a workgroup takes rows w, w + 4, ... of the chunk and the four
// waves' sums are added in wave order through LDS: one partial per (chunk, chain)
__device__ __forceinline__ void cm_block_partial(double v, double* part, int chunk, int ldp, int c, bool cin)
{
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    sh[w][lane] = v;
    __syncthreads();
    if (w == 0 && cin) part[(size_t)chunk * ldp + c] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
    __syncthreads();
}

// sum over the chunks of one partial array, for the 64 chains of a 256-thread workgroup: wave w adds its contiguous
// quarter of the chunks in order (eight loads in flight), the quarters are added in wave order through LDS.  One
// thread per chain walking all chunks was a chain of nchunk dependent loads (63 us at config 4).
__device__ __forceinline__ double cm_sum_chunks(const double* part, int nchunk, int ldp, int cc)
{
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per = (nchunk + 3) >> 2, k0 = w * per, k1 = (k0 + per < nchunk) ? k0 + per : nchunk;
    double a = 0.0;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * ldp + cc];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; k < k1; ++k) a += part[(size_t)k * ldp + cc];
    __syncthreads();
    sh[w][lane] = a;
    __syncthreads();
    return ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
}

// rows of the random-effect-sized arrays per workgroup of the per-chain kernels: small chunks when Q is small, so
// that config 4's 320 effects still spread over the chip
__host__ __device__ inline int cm_qrows(int Q) { return Q <= 4096 ? 16 : CM_ROWS; }

__global__ __launch_bounds__(256) void k_cm_init(double* V, int ldc, int Q, int C, CmChain ca, uint64_t seed,
                                                 uint32_t chain_offset, uint32_t iter_idx, const double* inj_init)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    if (c >= C) return;
    const uint32_t gid = chain_offset + (uint32_t)c;
    const int QR = cm_qrows(Q), q0 = blockIdx.y * QR;
    for (int q = q0 + w; q < q0 + QR && q < Q; q += 4)
        V[c + (size_t)q * ldc] = inj_init ? inj_init[q + (size_t)c * Q]
                                          : rng_normal(seed, (uint32_t)q, gid, 0u, 16u * iter_idx + 0u);
    if (blockIdx.y == 0 && w == 0) {                              // initialise_u, mhmcmc.h:47-59
        ca.e[c] = 0.001; ca.ebar[c] = 1.0; ca.H[c] = 0.0; ca.acc[c] = 0; ca.leap[c] = 0;
        ca.gen[c] = chain_minstd_seed(seed, gid, iter_idx);
        ca.steps[c] = 1;
    }
}

// partial sums of log f(y_i | MU[c,i]) over observation chunks [0, nchunk_n) and of log N(X[c,q]; 0, 1) (+ R[c,q]^2
// when R is given) over random-effect chunks: part_ll / part_lp / part_kin, ld = ldp
// FL != 0: the family / link is a compile-time constant (the 12-way switch of glm_logpdf with its lgamma / tgamma / erfc
// bodies inlined cost 290 VGPRs, one wave per SIMD: 109 us at config 4)
template <int FL>
__global__ __launch_bounds__(256) void k_cm_logprob_partials(const double* MU, const double* X, const double* R, int ldc,
                                                             int n, int Q, int C, const double* y, double var_par,
                                                             int flink, int nchunk_n, double* part_ll, double* part_lp,
                                                             double* part_kin, int ldp)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool cin = c < C;
    const int cc = cin ? c : 0;
    if ((int)blockIdx.y < nchunk_n) {
        const int fl = FL ? FL : flink;
        const int i0 = blockIdx.y * CM_ROWS, i1 = (i0 + CM_ROWS < n) ? i0 + CM_ROWS : n;
        double ll = 0.0;
        for (int i = i0 + w; i < i1; i += 16) {                       // rows i, i + 4, i + 8, i + 12: four loads in flight
            double m[4], yy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = (i + 4 * u < i1) ? i + 4 * u : i1 - 1;
                m[u] = MU[cc + (size_t)r * ldc]; yy[u] = y[r];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + 4 * u < i1) ll += glm_logpdf(yy[u], m[u], var_par, fl);
        }
        cm_block_partial(ll, part_ll, blockIdx.y, ldp, c, cin);
    } else {
        const int QR = cm_qrows(Q);
        const int ch = blockIdx.y - nchunk_n, q0 = ch * QR;
        double lp = 0.0, kin = 0.0;
        for (int q = q0 + w; q < q0 + QR && q < Q; q += 4) {
            lp += glm_logpdf(X[cc + (size_t)q * ldc], 0, 1, 7);
            if (R) { const double r = R[cc + (size_t)q * ldc]; kin += r * r; }
        }
        cm_block_partial(lp, part_lp, ch, ldp, c, cin);
        if (R) cm_block_partial(kin, part_kin, ch, ldp, c, cin);
    }
}

// the three kernels below: one 256-thread workgroup per 64 chains (cm_sum_chunks), wave 0 finishes
__global__ __launch_bounds__(256) void k_cm_lp0_fin(const double* part_ll, const double* part_lp, int nchunk_n,
                                                    int nchunk_q, int ldp, int C, double* lpcur)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cc = c < C ? c : 0;
    const double a = cm_sum_chunks(part_ll, nchunk_n, ldp, cc);
    const double b = cm_sum_chunks(part_lp, nchunk_q, ldp, cc);
    if (threadIdx.x >= 64 || c >= C) return;
    lpcur[c] = a + b;                                             // ll.sum() + lp.sum(), mcmlmodel.h:151
}

// new_proposal, first part (mhmcmc.h:62-75): momentum, first half step + position; partial sums of r^2.
// accflag (nullable): the PREVIOUS proposal's decisions, not yet applied to V / GRAD -- an accepted chain's state is read
// from UP / GRADP and written through to V / GRAD here, so the separate commit pass (k_cm_commit: one more read and write of
// the Q x C state, 131 us per proposal at config 5) is folded into this one.
__global__ __launch_bounds__(256) void k_cm_propose(double* V, double* GRAD, double* R, double* UP, int ldc,
                                                    int Q, int C, CmChain ca, uint64_t seed, uint32_t chain_offset,
                                                    uint32_t iter_idx, int it, const double* inj_mom, double* part_ss,
                                                    int ldp, const int* accflag, const double* GRADP)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool cin = c < C;
    const int cc = cin ? c : 0;
    const uint32_t gid = chain_offset + (uint32_t)cc;
    const double e = ca.e[cc];
    const int QR = cm_qrows(Q), q0 = blockIdx.y * QR;
    double ss = 0.0;
    if (cin)
        for (int q = q0 + w; q < q0 + QR && q < Q; q += 4) {
            const size_t off = c + (size_t)q * ldc;
            double r = inj_mom ? inj_mom[q + ((size_t)it * C + c) * Q]
                               : rng_normal(seed, (uint32_t)q, gid, (uint32_t)it, 16u * iter_idx + 2u);
            ss += r * r;
            double g, v;
            if (accflag && accflag[c]) { g = GRADP[off]; v = UP[off]; GRAD[off] = g; V[off] = v; }
            else { g = GRAD[off]; v = V[off]; }
            r = r + (e / 2) * g;
            R[off] = r;
            UP[off] = v + e * r;
        }
    cm_block_partial(ss, part_ss, blockIdx.y, ldp, c, cin);
}

__global__ __launch_bounds__(256) void k_cm_propose_fin(const double* part_ss, int nchunk_q, int ldp, int C, CmChain ca,
                                                        double lambda, int max_steps)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const double tot = cm_sum_chunks(part_ss, nchunk_q, ldp, c < C ? c : 0);
    if (threadIdx.x >= 64 || c >= C) return;
    ca.K0[c] = 0.5 * tot;
    double st = round(lambda / ca.e[c]);                          // mhmcmc.h:69-70
    if (!(st >= 1.0)) st = 1.0;
    if (st > (double)max_steps) st = (double)max_steps;
    ca.steps[c] = (int)st;
    ca.leap[c] += (long long)st;
}

// new_proposal, second part (mhmcmc.h:80-117): the decision of every chain
__global__ __launch_bounds__(256) void k_cm_accept_fin(const double* part_ll, const double* part_lp, const double* part_kin,
                                                       int nchunk_n, int nchunk_q, int ldp, int C, CmChain ca,
                                                       double target_accept, int adapt, int it, uint8_t* flags,
                                                       double* probs, int* accflag)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cc = c < C ? c : 0;
    const double a = cm_sum_chunks(part_ll, nchunk_n, ldp, cc);
    const double b = cm_sum_chunks(part_lp, nchunk_q, ldp, cc);
    const double kin = cm_sum_chunks(part_kin, nchunk_q, ldp, cc);
    if (threadIdx.x >= 64 || c >= C) return;
    const double l2 = a + b;
    const double lprt = 0.5 * kin, lpr = ca.K0[c], l1 = ca.lpcur[c];
    const double prob = fmin(1.0, exp(-l1 + lpr + l2 - lprt));
    uint32_t g = ca.gen[c];
    const double runif = minstd_canonical(g);
    ca.gen[c] = g;
    const int acc = runif < prob;
    accflag[c] = acc;
    if (acc) { ca.lpcur[c] = l2; ca.acc[c] += 1; }
    if (flags) flags[c + (size_t)it * C] = (uint8_t)acc;
    if (probs) probs[c + (size_t)it * C] = prob;
    if (adapt) {                                                  // mhmcmc.h:107-114
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
        ca.e[c] = ca.ebar[c];                                     // :116
    }
}

// accepted chains: V <- UP, GRAD <- GRADP
__global__ __launch_bounds__(256) void k_cm_commit(double* V, double* GRAD, const double* UP, const double* GRADP,
                                                   int ldc, int Q, int C, const int* accflag)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    if (c >= C || !accflag[c]) return;
    const int QR = cm_qrows(Q), q0 = blockIdx.y * QR;
    for (int q = q0 + w; q < q0 + QR && q < Q; q += 4) {
        const size_t off = c + (size_t)q * ldc;
        V[off] = UP[off]; GRAD[off] = GRADP[off];
    }
}

// B (rows x cols, column-major, ldb) <- A' where A is chain-major (A[c + r * ldc], c < cols_src = rows of B ...):
// generic 32 x 32 LDS transpose: OUT[i + j * ldo] = IN[j + i * ldi], i < ni, j < nj
__global__ __launch_bounds__(256) void k_cm_transpose(const double* IN, int ldi, int ni, int nj, double* OUT, size_t ldo,
                                                      size_t out_col_stride_mult, int out_col_offset)
{
    // out column index of source index j: j * out_col_stride_mult + out_col_offset (used to interleave the draws of
    // several chains in the sample matrix)
    __shared__ double tile[32][33];
    const int bi = blockIdx.x * 32, bj = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int j = bj + tx, i = bi + r;
        tile[r][tx] = (i < ni && j < nj) ? IN[j + (size_t)i * ldi] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = bi + tx, j = bj + r;
        if (i < ni && j < nj) OUT[i + ((size_t)j * out_col_stride_mult + out_col_offset) * ldo] = tile[tx][r];
    }
}

}  // namespace mcml
