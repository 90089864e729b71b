// mvn.hip -- multivariate-normal log-likelihood of the random effects,
//   (1/m) sum_b sum_j log N(u_{b,j}; 0, D_b(theta)),
// i.e. MCMLDmatrix::loglik / logdet / loglik_block (mcmldmatrix.h:23-78), the
// D_likelihood functor (likelihood.h:31-46) and export mvn_ll
// (mcml_optim.cpp:406-414), plus DMatrix::genD(0,chol,false) (mcml_full.cpp:68).
//
// The reference rebuilds and refactorises each block once per sample column
// (defect D2); here every block is built and factorised once per theta:
//   * all-gr blocks (diagonal, mcmldmatrix.h:61-65): closed form, one streaming pass
//     over u (HBM-bound: 8 B per element);
//   * blocks of dim <= 32: one wave builds + factorises the block in LDS and
//     forward-substitutes 64 sample columns at a time;
//   * larger blocks: fused covariance build -> recursive blocked Cholesky whose
//     trailing updates are FP64-MFMA GEMMs (dgemm_mfma.h), 128-wide leaves
//     factorised and inverted in LDS -> recursive TRSM of all m sample columns
//     (MFMA GEMMs against the inverted leaves) -> Frobenius norm + log-det.
// Reductions are two-stage with a fixed order, so the result is run-to-run
// bit-reproducible.
#include "ctx.h"
#include "dgemm_mfma.h"
#include "dgemm_dl.h"
#include "dgemm_dlds.h"
#include "reduce.h"

namespace mcml {

#define LOG_2PI 1.8378770664093454835606594728112   /* log(2*M_PI), mcmldmatrix.h:63,75 */

// ------------------------------------------------------------------ reductions
// out[0] = (accumulate ? out[0] : 0) + scale * sum(partials[0..n))
__global__ __launch_bounds__(256) void k_sum_partials(const double* partials, int n, double scale,
                                                      double* out, int accumulate)
{
    __shared__ double sh[4];
    double v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    double r = block_sum(v, sh);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0) + scale * r;
}

int device_sum(Ctx& c, const double* partials, int n, double* dev_out)
{
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, partials, n, 1.0, dev_out, 0);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

// ------------------------------------------------------------------ diagonal blocks
// dd[k] = prod_k theta^2 ("dmat(k,k)*dmat(k,k)"), dc[k] = -0.5 log(dd) - 0.5 log(2 pi)
__global__ void k_diag_prep(int Q, const int* rowblock, const CovBlock* blocks, const int32_t* cov,
                            int rows, ThetaArg th, double* dd, double* dc)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Q) return;
    int b = rowblock[k];
    if (b < 0) { dd[k] = 1.0; dc[k] = 0.0; return; }
    CovBlock blk = blocks[b];
    double val = 1.0;
    for (int r = blk.r0; r < blk.r1; ++r) val = cov_term(1, 0.0, &th.v[cov[r + 4 * rows]], val);
    double d = sqrt(val);           // the Cholesky factor of a diagonal block
    dd[k] = d * d;
    dc[k] = -0.5 * log(d * d) - 0.5 * LOG_2PI;
}

__global__ __launch_bounds__(256) void k_diag_ll(const double* U, int ldu, int Q, int mcols,
                                                 const int* rowblock, const double* dd,
                                                 const double* dc, double* partials)
{
    __shared__ double sh[4];
    int k = blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    if (k < Q && rowblock[k] >= 0) {
        const double idd = dd[k], c0 = dc[k];
        for (int col = blockIdx.y; col < mcols; col += gridDim.y) {
            double u = U[k + (size_t)col * ldu];
            acc += c0 - 0.5 * u * u / idd;
        }
    }
    double r = block_sum(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// ------------------------------------------------------------------ covariance entry
__device__ __forceinline__ double cov_entry(const CovBlock& blk, const int32_t* cov, int rows,
                                            const double* bd, const ThetaArg& th, int i, int j)
{
    double val = 1.0;
    int coff = 0;
    for (int r = blk.r0; r < blk.r1; ++r) {
        const int fn = cov[r + 2 * rows], nv = cov[r + 3 * rows], pi = cov[r + 4 * rows];
        double d2 = 0.0;
        for (int p = 0; p < nv; ++p) {
            double df = bd[i + (size_t)(coff + p) * blk.dim] - bd[j + (size_t)(coff + p) * blk.dim];
            d2 += df * df;
        }
        val = cov_term(fn, sqrt(d2), &th.v[pi], val);
        coff += nv;
    }
    return val;
}

// ------------------------------------------------------------------ small blocks
__global__ __launch_bounds__(64) void k_small_ll(const double* U, int ldu, int mcols,
                                                 const int* small_ids, const CovBlock* blocks,
                                                 const int32_t* cov, int rows, const double* data,
                                                 ThetaArg th, double* partials, int* errflag)
{
    __shared__ double S[SMALL_BLOCK * (SMALL_BLOCK + 1)];
    __shared__ double Zs[SMALL_BLOCK * 64];
    const CovBlock blk = blocks[small_ids[blockIdx.x]];
    const int dim = blk.dim, lane = threadIdx.x;
    constexpr int LS = SMALL_BLOCK + 1;
    const double* bd = data + blk.doff;
    for (int e = lane; e < dim * dim; e += 64) {
        int i = e % dim, j = e / dim;
        if (i >= j) S[i * LS + j] = cov_entry(blk, cov, rows, bd, th, i, j);
    }
    __syncthreads();
    // left-looking column Cholesky, same operation order as the CPU oracle
    for (int j = 0; j < dim; ++j) {
        if (lane == j) {
            double d = S[j * LS + j];
            for (int k = 0; k < j; ++k) d -= S[j * LS + k] * S[j * LS + k];
            if (!(d > 0.0)) { atomicExch(errflag, 1); d = 1.0; }
            S[j * LS + j] = sqrt(d);
        }
        __syncthreads();
        if (lane > j && lane < dim) {
            double s = S[lane * LS + j];
            for (int k = 0; k < j; ++k) s -= S[lane * LS + k] * S[j * LS + k];
            S[lane * LS + j] = s / S[j * LS + j];
        }
        __syncthreads();
    }
    double logdet = 0;
    for (int i = 0; i < dim; ++i) logdet += 2 * log(S[i * LS + i]);
    double acc = 0;
    for (int col = blockIdx.y * 64 + lane; col < mcols; col += gridDim.y * 64) {
        const double* u = U + blk.matstart + (size_t)col * ldu;
        double quad = 0;
        for (int i = 0; i < dim; ++i) {                      // algo::forward_sub, moremaths.h:166-179
            double lsum = 0;
            for (int j = 0; j < i; ++j) lsum += S[i * LS + j] * Zs[j * 64 + lane];
            double z = (u[i] - lsum) / S[i * LS + i];
            Zs[i * 64 + lane] = z;
            quad += z * z;
        }
        acc += (-0.5 * dim * LOG_2PI - 0.5 * logdet - 0.5 * quad);
    }
    acc = wave_sum(acc);
    if (lane == 0) partials[blockIdx.x * gridDim.y + blockIdx.y] = acc;
}

// ------------------------------------------------------------------ large blocks
// lower triangle (mirrored) of one block's covariance matrix
// (dpad > dim: rows / columns dim .. dpad are an identity border, so that an odd block can be factorised as an
// even one -- every panel pointer stays 16-byte aligned; the border does not change L, the log-determinant or
// the solves)
// the candidate thetas of one round (mvn_loglik_batch): blockIdx.z picks the candidate and its matrix
constexpr int MVN_MAXBATCH = 8;
struct ThetaBatch { ThetaArg t[MVN_MAXBATCH]; };

template <class TH>
__device__ __forceinline__ void build_dense_body(double* A, int lda, int bidx, const CovBlock* blocks, const int32_t* cov,
                                                 int rows, const double* data, const TH& th, int mirror, int dpad);

__global__ __launch_bounds__(256) void k_build_dense_batch(double* A, int lda, size_t bsA, int bidx, const CovBlock* blocks,
                                                           const int32_t* cov, int rows, const double* data,
                                                           ThetaBatch tb, int dpad)
{
    build_dense_body(A + (size_t)blockIdx.z * bsA, lda, bidx, blocks, cov, rows, data, tb.t[blockIdx.z], 0, dpad);
}

__global__ __launch_bounds__(256) void k_build_dense(double* A, int lda, int bidx, const CovBlock* blocks,
                                                     const int32_t* cov, int rows, const double* data,
                                                     ThetaArg th, int mirror, int dpad)
{
    build_dense_body(A, lda, bidx, blocks, cov, rows, data, th, mirror, dpad);
}

template <class TH>
__device__ __forceinline__ void build_dense_body(double* A, int lda, int bidx, const CovBlock* blocks, const int32_t* cov,
                                                 int rows, const double* data, const TH& th, int mirror, int dpad)
{
    // a workgroup = 64 rows x 16 columns, a wave = the 64 rows of four columns: every store is 512 contiguous bytes
    // (16 x 16 tiles with 128-byte row groups ran 71 us for the 5000 x 5000 lower triangle)
    if ((int)blockIdx.x * 64 + 63 < (int)blockIdx.y * 16) return;     // strictly above the diagonal
    const CovBlock blk = blocks[bidx];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = blockIdx.y * 16 + (threadIdx.x >> 6) * 4 + u;
        if (i >= blk.dim && i < dpad && j <= i) { A[i + (size_t)j * lda] = (i == j) ? 1.0 : 0.0; continue; }
        if (i >= blk.dim || j >= blk.dim || i < j) continue;
        const double v = cov_entry(blk, cov, rows, data + blk.doff, th, i, j);
        A[i + (size_t)j * lda] = v;
        if (mirror && i != j) A[j + (size_t)i * lda] = v;
    }
}

// broadcast lane `src` (compile-time constant) of a double: two v_readlane_b32, no LDS round trip
template <int SRC>
__device__ __forceinline__ double bcast_lane(double v)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], SRC);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], SRC);
    return u.d;
}

template <int C, int J>
__device__ __forceinline__ void potrf16_upd(double (&a)[16], double l, int r)
{
    // rows r < J hold (never read) upper-triangle entries: updating them too saves the select
    const double lj = bcast_lane<J>(l);
    a[J] = __builtin_fma(-l, lj, a[J]);
    if constexpr (J < 15) potrf16_upd<C, J + 1>(a, l, r);
}

// 1 / sqrt(x) to double precision without the division: v_rsq_f64 seed (~2^-26) + two Newton
// steps, all fused multiply-adds (about 8 dependent instructions instead of ~35 for sqrt + div:
// the 16 x 16 diagonal tile is a serial chain, one of these per column)
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double e = __builtin_fma(-(h * y), y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-(h * y), y, 0.5);
    y = __builtin_fma(y, e, y);
    return y;
}

// One column of the 16 x 16 diagonal tile, branch-free (the 16 columns are one serial dependency
// chain on a single wave: every exec-mask branch in it costs a pipeline bubble).  Lane `ln` holds one
// ROW of the 16 columns being eliminated: lanes 0-15 are the tile's own rows (lane C is the pivot row of
// column C), lanes >= 16 are rows that ride along (x L11' = a for any row a costs no extra instruction on
// a 64-lane wave).  `bad` accumulates failed pivots, `yv` collects lane C's reciprocal pivot 1 / L_CC.
template <int C>
__device__ __forceinline__ void potrf16_col(double (&a)[16], int ln, int& bad, double& yv)
{
    double diag = bcast_lane<C>(a[C]);
    const bool ok = diag > 0.0;
    bad |= ok ? 0 : 1;
    diag = ok ? diag : 1.0;
    const double y = rsqrt_nr(diag);
    // every row, the pivot row included, is scaled by the reciprocal pivot: L_CC = diag * rsqrt(diag) is sqrt(diag) to
    // ~1.5 ulp (no separate square root, no lane-dependent branch on the serial chain)
    const double l = a[C] * y;
    yv = (ln == C) ? y : yv;
    a[C] = l;
    if constexpr (C < 15) {
        // a[j] -= l * l_j for j > C, l_j = lane j's l
        potrf16_upd<C, C + 1>(a, l, ln);
    }
}

// Leaf of the blocked factorisation: n <= 128.  One workgroup keeps the block in LDS (row stride 129:
// conflict-free column walks) and factorises it in 16-wide steps; the unused upper triangle receives the
// inverse X = inv(L) (X[i][j], i > j, at S[j][i]; its diagonal = the reciprocal pivots, in xd).  Writes L over
// the lower triangle of A and inv(L) to Linv (128 x 128, ld 128; the caller has zeroed it: entries above the
// diagonal and beyond n are never written).
//
// Step kt (columns kb = 16 kt ...) is software-pipelined so that the serial part is as short as possible:
//   P1  wave 0: the 16 x 16 diagonal tile in registers, one row per lane, readlane broadcasts.  Lanes 16-31
//       carry the 16 rows of the identity (their solution is X_II, the diagonal tile of the inverse) and lanes
//       32-63 the first 32 rows of the panel below -- on a 64-lane wave those rows cost nothing.
//       waves 1-7 meanwhile: the trailing-update tiles of step kt-1 that wave 0 did not need, then block row
//       kt-1 of the inverse (MFMA), stored straight to Linv.
//   P2  all waves: the rest of the panel, rows x X_II' on the matrix cores (4 MFMAs per 16 rows).
//   P3  wave 0: the three trailing tiles the next step's P1 reads (next diagonal tile + its 32 ride-along
//       rows), then straight into P1 of step kt+1 with no workgroup barrier in between.
constexpr int LEAF_NT = 512, LEAF_NW = LEAF_NT / 64;
// workgroup barrier for LDS traffic only.  __syncthreads() also waits for vmcnt(0): with Linv stored to global
// memory as the leaf goes, every barrier would wait for those stores to be acknowledged (~1-2 us each).
__device__ __forceinline__ void leaf_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__global__ __launch_bounds__(LEAF_NT) void k_potrf_leaf(double* A, int lda, int n, double* Linv, int* errflag,
                                                    unsigned long long* prof, size_t bsA = 0, size_t bsL = 0)
{
#pragma clang fp contract(fast)      // the factorisation is not part of the bit-exact RNG / leapfrog contract
    // blockIdx.x: one of several matrices factorised side by side (mvn_loglik_batch), each with its own flag
    A += (size_t)blockIdx.x * bsA; Linv += (size_t)blockIdx.x * bsL; errflag += blockIdx.x;
#define LEAF_T(i) do { if (prof && threadIdx.x == 0) prof[i] = __builtin_amdgcn_s_memtime(); } while (0)
    LEAF_T(0);
    extern __shared__ __attribute__((aligned(16))) double S[];
    constexpr int LS = 129;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    {   // load the lower triangle: thread = (row, column parity); 16 loads in flight per thread
        constexpr int JG = LEAF_NT / 128;
        const int i = tid & 127, j0 = tid >> 7;
        for (int jb = 0; jb < 128; jb += 16 * JG) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = jb + j0 + JG * q;
                v[q] = (i < n && j < n && i >= j) ? A[i + (size_t)j * lda] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) S[i * LS + jb + j0 + JG * q] = v[q];
        }
    }
    __syncthreads();
    LEAF_T(1);
    unsigned long long ta = 0, tb = 0, tc = 0, t_, q0 = 0, q1 = 0, q2 = 0, q3 = 0, u_;
    const int ntile = (n + 15) >> 4;
    double* xd = S + 128 * LS;            // 128 doubles: reciprocal pivots = diagonal of the inverse
    double* Tt = xd + 128;                // 7 scratch tiles of 16 x 16 (one per wave 1..7)
    double* dummy = Tt + 7 * 256;         // 64 doubles nobody reads (address-select stores)

    // block row I of the inverse (waves 1..7: wave w owns column tile J = w - 1)
    auto inverse_row = [&](int I) {
        const int J = wave - 1;
        if (J >= I) return;
        // T_J = sum_{K=J}^{I-1} L_IK X_KJ on the matrix cores
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int K = J; K < I; ++K) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const int t = kc * 4 + lk;
                // A operand: L_IK[row l15][t];  B operand: X_KJ[t][col l15]
                const double pa = S[(I * 16 + l15) * LS + K * 16 + t];
                double pb;
                if (K == J) pb = (t == l15) ? xd[J * 16 + l15] : (t > l15 ? S[(J * 16 + l15) * LS + J * 16 + t] : 0.0);
                else pb = S[(J * 16 + l15) * LS + K * 16 + t];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, acc, 0, 0, 0);
            }
        }
        // T_J[row][col]: row = lk + 4 r, col = l15 -> this wave's scratch tile
#pragma unroll
        for (int r = 0; r < 4; ++r) Tt[J * 256 + (lk + 4 * r) * 16 + l15] = acc[r];
        // X_IJ = -X_II T_J
        d4 acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int t = kc * 4 + lk;
            // A operand: X_II[row l15][t] (lower triangular);  B operand: T_J[t][col l15]
            const double pa = (t == l15) ? xd[I * 16 + l15]
                                         : (t < l15 ? S[(I * 16 + t) * LS + I * 16 + l15] : 0.0);
            const double pb = Tt[J * 256 + t * 16 + l15];
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = I * 16 + lk + 4 * r, gj = J * 16 + l15;
            if (gi < n) {
                S[gj * LS + gi] = -acc2[r];
                Linv[gi + gj * 128] = -acc2[r];
            }
        }
    };
    // trailing tile (ti, tj) -= P_ti P_tj' with the panel of step kb (K = 16), on the matrix cores
    auto upd_tile = [&](int ti, int tj, int kb) {
        const int ri = ti * 16, rj = tj * 16;
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = S[(ri + lk + 4 * r) * LS + rj + l15];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const double pa = -S[(ri + l15) * LS + kb + kc * 4 + lk];
            const double pb = S[(rj + l15) * LS + kb + kc * 4 + lk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(ri + lk + 4 * r) * LS + rj + l15] = acc[r];
    };
    // the tiles of step kt's trailing update that wave 0 does NOT do itself, shared out over waves 1..7
    auto rest_update = [&](int kt) {
        int cnt = 0;
        for (int tj = kt + 1; tj < ntile; ++tj)
            for (int ti = tj; ti < ntile; ++ti) {
                if (tj == kt + 1 && ti <= kt + 3) continue;            // wave 0's (P3)
                if (cnt % (LEAF_NW - 1) == wave - 1) upd_tile(ti, tj, kt * 16);
                ++cnt;
            }
    };

    for (int kt = 0; kt < ntile; ++kt) {
        const int kb = kt * 16;
        t_ = __builtin_amdgcn_s_memtime();
        if (wave == 0) {
            // P1: lane < 16 tile row kb + lane; 16..31 identity row lane - 16; 32..63 panel row kb + 16 + (lane - 32)
            const int prow = kb + 16 + (lane - 32);
            const bool pvalid = lane >= 32 && prow < ntile * 16;
            int bad = 0; double yv = 1.0;
            double a[16];
            u_ = __builtin_amdgcn_s_memtime();
            // branch-free: one load per column from a clamped, always valid row; selects supply the identity
            // (padding rows of the tile, lanes 16-31) and zeros (padding rows of the panel)
            const int lrow = lane < 32 ? kb + l15 : (pvalid ? prow : 0);
            const bool ld_row = lane < 16 ? (kb + lane < n) : pvalid;
            const double* srow = S + lrow * LS + kb;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double v = srow[c];
                const bool use = ld_row && (lane >= 32 || kb + c < n);
                a[c] = use ? v : ((lane < 32 && l15 == c) ? 1.0 : 0.0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            q0 += __builtin_amdgcn_s_memtime() - u_; u_ = __builtin_amdgcn_s_memtime();
            potrf16_col<0>(a, lane, bad, yv);  potrf16_col<1>(a, lane, bad, yv);
            potrf16_col<2>(a, lane, bad, yv);  potrf16_col<3>(a, lane, bad, yv);
            potrf16_col<4>(a, lane, bad, yv);  potrf16_col<5>(a, lane, bad, yv);
            potrf16_col<6>(a, lane, bad, yv);  potrf16_col<7>(a, lane, bad, yv);
            potrf16_col<8>(a, lane, bad, yv);  potrf16_col<9>(a, lane, bad, yv);
            potrf16_col<10>(a, lane, bad, yv); potrf16_col<11>(a, lane, bad, yv);
            potrf16_col<12>(a, lane, bad, yv); potrf16_col<13>(a, lane, bad, yv);
            potrf16_col<14>(a, lane, bad, yv); potrf16_col<15>(a, lane, bad, yv);
            q1 += __builtin_amdgcn_s_memtime() - u_; u_ = __builtin_amdgcn_s_memtime();
            // lanes 0-15 write row kb + l15 up to the diagonal (L), lanes 16-31 the SAME row above it (x[c] =
            // X_II[c][jj], c > jj: kept in the tile's upper part for the MFMA phases), lanes 32-63 their panel row.
            // Full leaves (n = 128) store through ADDRESS selects (a lane with nothing to store hits a dummy slot):
            // per-column exec masks cost this wave ~100 SGPRs, which spill, and the serial chain pays the reloads.
            double* drow = S + lrow * LS + kb;
            if (n == 128) {
                double* dmy = dummy + lane;
                double* w1 = (lane < 16 || pvalid) ? drow : nullptr;
#pragma unroll
                for (int c = 0; c < 16; ++c) *(w1 ? w1 + c : dmy) = a[c];          // rows, all 16 columns
                // then (LDS keeps a wave's accesses in order) the identity lanes overwrite the part above the diagonal
#pragma unroll
                for (int c = 0; c < 16; ++c) *((lk == 1 && c > l15) ? drow + c : dmy) = a[c];
                if (lane < 16) xd[kb + lane] = yv;         // reciprocal diagonal = the inverse's diagonal
                if (lk == 1) {
                    // the diagonal tile of the inverse is final: x[c] = X_II[c][jj] for c >= jj, exact zeros above
                    double* g = Linv + (kb + l15) * 128 + kb;
#pragma unroll
                    for (int c = 0; c < 16; ++c) g[c] = a[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const bool st = lane < 16 ? (l15 >= c && kb + l15 < n && kb + c < n) : (lane < 32 ? c > l15 : pvalid);
                    if (st) drow[c] = a[c];
                }
                if (lane < 16) xd[kb + lane] = yv;
                if (lk == 1) {
                    double* g = Linv + (kb + l15) * 128 + kb;
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        if (c >= l15 && kb + c < n && kb + l15 < n) g[c] = a[c];
                }
            }
            if (bad && lane == 0) atomicExch(errflag, 1);
            q2 += __builtin_amdgcn_s_memtime() - u_; u_ = __builtin_amdgcn_s_memtime();
        } else if (kt >= 1) {
            rest_update(kt - 1);
            inverse_row(kt - 1);
        }
        leaf_sync();
        q3 += __builtin_amdgcn_s_memtime() - u_;
        ta += __builtin_amdgcn_s_memtime() - t_; t_ = __builtin_amdgcn_s_memtime();
        // P2: panel rows beyond the 32 that rode along: x = a inv(L11)' = a X_II' on the matrix cores
        for (int t = kt + 3 + wave; t < ntile; t += LEAF_NW) {
            const int r0 = t * 16;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            double pa[4], pb[4];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const int k = kc * 4 + lk;
                pa[kc] = S[(r0 + l15) * LS + kb + k];                      // a[i = l15][k]
                // B[k][j = l15] = X_II[j][k]
                pb[kc] = (l15 == k) ? xd[kb + k] : (l15 > k ? S[(kb + k) * LS + kb + l15] : 0.0);
            }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[kc], pb[kc], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(r0 + lk + 4 * r) * LS + kb + l15] = acc[r];
        }
        leaf_sync();
        tb += __builtin_amdgcn_s_memtime() - t_; t_ = __builtin_amdgcn_s_memtime();
        // P3: wave 0 updates what its next P1 reads and goes on without a barrier; the other waves pick up the
        // rest of this step's trailing update at the top of the next iteration
        if (wave == 0) {
            for (int ti = kt + 1; ti < ntile && ti <= kt + 3; ++ti) upd_tile(ti, kt + 1, kb);
            // the next P1 reads, one row per lane, what other lanes of this wave have just written: LDS executes a
            // wave's accesses in order; this keeps the compiler from moving them
            asm volatile("" ::: "memory");
        }
        tc += __builtin_amdgcn_s_memtime() - t_;
    }
    leaf_sync();
    if (prof && threadIdx.x == 0) { prof[2] = ta; prof[3] = tb; prof[4] = tc; }
    LEAF_T(5);
    {   // write L
        const int i = tid & 127, j0 = tid >> 7;
        if (i < n)
            for (int j = j0; j <= i && j < n; j += LEAF_NT / 128) A[i + (size_t)j * lda] = S[i * LS + j];
    }
    LEAF_T(6);
    // the last block row of the inverse (the earlier ones were done under the later steps' P1)
    if (wave >= 1) inverse_row(ntile - 1);
    LEAF_T(9);
    if (prof && threadIdx.x == 0) { prof[6] = q0; prof[7] = q1; prof[8] = q2; prof[1] = q3; }
#undef LEAF_T
}

// debug: phase timestamps (shader clock) of one 128 x 128 leaf on a random SPD block
int potrf_leaf_profile(Ctx& c, unsigned long long* host10)
{
    DevMat A; DevBuf prof;
    MCML_TRY(A.alloc(128, 128));
    MCML_TRY(prof.ensure(80));
    MCML_TRY(c.linv.ensure(sizeof(double) * 2 * CHOL_NB * CHOL_NB));
    std::vector<double> h((size_t)A.ld * 128, 0.0);
    for (int j = 0; j < 128; ++j) for (int i = 0; i < 128; ++i) h[i + (size_t)j * A.ld] = (i == j ? 130.0 : 1.0 / (1 + abs(i - j)));
    MCML_TRY(copy_h2d(A.d(), h.data(), sizeof(double) * h.size(), 0)); MCML_HIP(hipDeviceSynchronize());
    MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_potrf_leaf), (int)(sizeof(double) * (128 * 129 + 128 + 7 * 256 + 64))));
    MCML_HIP(hipMemset(c.linv.p, 0, sizeof(double) * 2 * CHOL_NB * CHOL_NB));
    for (int rep = 0; rep < 3; ++rep) {
        MCML_TRY(copy_h2d(A.d(), h.data(), sizeof(double) * h.size(), 0)); MCML_HIP(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_potrf_leaf, dim3(1), dim3(LEAF_NT), sizeof(double) * (128 * 129 + 128 + 7 * 256 + 64), c.stream, A.d(), A.ld,
                           128, c.linv.d(), c.scalars.as<int>() + 32, prof.as<unsigned long long>());
        MCML_HIP(hipStreamSynchronize(c.stream));
    }
    MCML_HIP(hipMemcpy(host10, prof.p, 80, hipMemcpyDeviceToHost));
    return MCML_OK;
}

// AT (cols x rows) = A' through a 32 x 33 LDS tile
__global__ __launch_bounds__(256) void k_transpose_in(const double* A, int lda, int rows, int cols, double* AT, int ldt,
                                                      size_t bsT = 0)
{
    AT += (size_t)blockIdx.z * bsT;            // the same samples under every candidate's matrix
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = bx + tx, j = by + r;
        tile[r][tx] = (i < rows && j < cols) ? A[i + (size_t)j * lda] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = by + tx, i = bx + r;
        if (i < rows && j < cols) AT[j + (size_t)i * ldt] = tile[tx][r];
    }
}

__global__ void k_copy_block(double* dst, int ldd, const double* src, int lds, int rows, int cols)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    for (int j = blockIdx.y; j < cols; j += gridDim.y) dst[i + (size_t)j * ldd] = src[i + (size_t)j * lds];
}

__global__ __launch_bounds__(256) void k_sumsq(const double* U, int ldu, int rows, int cols, double* partials)
{
    __shared__ double sh[4];
    int i = blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    if (i < rows)
        for (int j = blockIdx.y; j < cols; j += gridDim.y) {
            double u = U[i + (size_t)j * ldu];
            acc += u * u;
        }
    double r = block_sum(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// scal[1] = logdet = sum 2 log L_ii (mcmldmatrix.h:67-70)
__global__ __launch_bounds__(256) void k_logdet(const double* A, int lda, int n, double* out)
{
    __shared__ double sh[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += 2 * log(A[i + (size_t)i * lda]);
    double r = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = r;
}

// scal[0] += mcols*(-0.5 dim log 2pi - 0.5 logdet) - 0.5 sumsq   (mcmldmatrix.h:75)
__global__ void k_finish_large(double* scal, int dim, int mcols)
{
    scal[0] += (double)mcols * (-0.5 * dim * LOG_2PI - 0.5 * scal[1]) - 0.5 * scal[2];
}

__global__ void k_zero_upper(double* A, int lda, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y;
    if (i < n && j < n && i < j) A[i + (size_t)j * lda] = 0.0;
}

// LDS of k_potrf_leaf: the 128 x 129 block, the inverse's diagonal, 7 scratch tiles
static constexpr size_t POTRF_LDS = sizeof(double) * (128 * 129 + 128 + 7 * 256 + 64);

// ------------------------------------------------------------------ recursion (host)
static inline int split128(int n)
{
    int h = (n / 2) / CHOL_NB * CHOL_NB;
    return h < CHOL_NB ? CHOL_NB : h;
}

static int potrf_rec(Ctx& c, double* A0, int lda, int off, int n);
static int trsm_right_rec(Ctx& c, const double* A0, int lda, int off, int n, double* X, int ldx, int M);

// X (M x n) <- X * inv(L)^T, L = A0[off:off+n, off:off+n] lower
static int trsm_right_rec(Ctx& c, const double* A0, int lda, int off, int n, double* X, int ldx, int M)
{
    if (n <= CHOL_NB) {
        const double* Linv = c.linv.d() + (size_t)(off / CHOL_NB) * CHOL_NB * CHOL_NB;
        EpiAxpby epi{X, ldx, 1.0, 0.0};
        // in place: a workgroup reads its whole row band (K = n <= 128) before it writes
        return launch_gemm<true>(c.stream, M, n, n, X, ldx, Linv, CHOL_NB, epi, false, 1);
    }
    const int n1 = split128(n), n2 = n - n1;
    MCML_TRY(trsm_right_rec(c, A0, lda, off, n1, X, ldx, M));
    const double* L21 = A0 + (off + n1) + (size_t)off * lda;
    double* X2 = X + (size_t)n1 * ldx;
    EpiAxpby epi{X2, ldx, -1.0, 1.0};
    MCML_TRY(launch_gemm<true>(c.stream, M, n2, n1, X, ldx, L21, lda, epi));
    return trsm_right_rec(c, A0, lda, off + n1, n2, X2, ldx, M);
}

static int potrf_rec(Ctx& c, double* A0, int lda, int off, int n)
{
    double* A = A0 + off + (size_t)off * lda;
    if (n <= CHOL_NB) {
        double* Linv = c.linv.d() + (size_t)(off / CHOL_NB) * CHOL_NB * CHOL_NB;
        hipLaunchKernelGGL(k_potrf_leaf, dim3(1), dim3(LEAF_NT), POTRF_LDS, c.stream,
                           A, lda, n, Linv, c.scalars.as<int>() + 32, nullptr);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
    const int n1 = split128(n), n2 = n - n1;
    MCML_TRY(potrf_rec(c, A0, lda, off, n1));
    double* A21 = A + n1;
    MCML_TRY(trsm_right_rec(c, A0, lda, off, n1, A21, lda, n2));
    double* A22 = A + n1 + (size_t)n1 * lda;
    EpiAxpby epi{A22, lda, -1.0, 1.0};
    MCML_TRY(launch_gemm<true>(c.stream, n2, n2, n1, A21, lda, A21, lda, epi, true));
    return potrf_rec(c, A0, lda, off + n1, n2);
}

// The K = 128 panel GEMMs of the blocked factorisation / solve: deep-ring direct-to-LDS kernel
// (dgemm_dl.h) when its contract holds, the register-staged kernel otherwise.
//   inplace: 0 none, 1 = C aliases A (N <= 128), 2 = C aliases B (M <= 128)
//   GLMMR_MCML_CHOL_GEMM=reg : always the register-staged kernel
template <bool BNMAJOR>
static int chol_gemm(hipStream_t s, int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                     const EpiAxpby& epi, bool lower_only, int inplace)
{
    static const bool use_dl = !(getenv("GLMMR_MCML_CHOL_GEMM") && !strcmp(getenv("GLMMR_MCML_CHOL_GEMM"), "reg"));
    if (use_dl && dl_applicable(M, N, K, A, lda, B, ldb, BNMAJOR))
        return launch_gemm_dl<BNMAJOR>(s, M, N, K, A, lda, B, ldb, epi, lower_only, 0, inplace);
    return launch_gemm<BNMAJOR>(s, M, N, K, A, lda, B, ldb, epi, lower_only, inplace ? inplace : -1);
}

// Right-looking variant: one 128-wide panel at a time -- leaf (factor + invert the diagonal
// block in LDS), panel TRSM as a GEMM against the inverted block, SYRK of the trailing matrix.
//
// Look-ahead: the leaf is a single-workgroup, latency-bound kernel (~90 us) and the late
// panels' GEMMs are small, so running them back to back leaves the chip idle most of the
// time.  The trailing update is therefore split: the next panel's 128 columns are updated
// first, the next leaf then runs on a high-priority side stream while the main stream updates
// the rest of the trailing matrix (columns the leaf never touches).
//   GLMMR_MCML_CHOL=rec    : the recursive variant;  =nola : this one without look-ahead
static int chol_mode()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GLMMR_MCML_CHOL");
        v = (e && !strcmp(e, "rec")) ? 0 : (e && !strcmp(e, "nola")) ? 2 : 1;
    }
    return v;
}
static bool chol_blocked() { return chol_mode() != 0; }

static int lookahead_setup(Ctx& c)
{
    if (c.aux) return MCML_OK;
    int lo = 0, hi = 0;
    MCML_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    MCML_HIP(hipStreamCreateWithPriority(&c.aux, hipStreamNonBlocking, hi));
    MCML_HIP(hipStreamCreateWithPriority(&c.aux_lo, hipStreamNonBlocking, lo));
    MCML_HIP(hipEventCreateWithFlags(&c.ev_col, hipEventDisableTiming));
    MCML_HIP(hipEventCreateWithFlags(&c.ev_leaf, hipEventDisableTiming));
    MCML_HIP(hipEventCreateWithFlags(&c.ev_ps, hipEventDisableTiming));
    MCML_HIP(hipEventCreateWithFlags(&c.ev_b, hipEventDisableTiming));
    return MCML_OK;
}

// Right-looking blocked Cholesky of the n x n lower triangle of A, 128-wide panels, with `extra` more ROWS
// carried below the matrix (rows n .. n+extra of the same column-major array): they receive every panel
// operation the rows of the matrix receive, i.e. X <- X inv(L)'.  With X = U' (the sample columns, one per
// extra row) that is the forward substitution inv(L) U of mvn_ll, done inside the factorisation's own GEMMs
// -- no separate TRSM pass, and the late panels, whose trailing matrices are tiny, still fill the chip.
//
// Step t, on the main stream: the ONE 128 x 128 panel block that becomes the next diagonal block's multiplier
// and that block's own update (two single-block products spread over 8 / 16 CUs); then, forked to a
// high-priority side stream, leaf(t+1) -- while the main stream runs the bulk: the rest of panel t and the
// whole trailing update except that block, one launch, lower tiles only, balanced over the XCDs.  The leaf
// (ONE workgroup that needs a whole CU's LDS) becomes ready together with the bulk and is dispatched first, so
// it never waits for a CU to drain.  The main stream joins the leaf before step t+1.  Period: two small GEMMs
// + max(bulk, leaf).  (A freer two-queue schedule measured ~10 % faster live but
// hung intermittently; it exists as a captured graph only -- see "the hang" above potrf_la2_capture.)
//   GLMMR_MCML_CHOL=rec : the recursive variant;  =nola : everything on one stream
// Several matrices of the same shape factorised side by side (the candidate thetas of one theta-step round): every
// launch of the schedule covers all of them (leaf: one workgroup per matrix; products: blockIdx.y), so the latency
// chain of the late steps -- leaf, two single-block products, their gaps -- is paid once per round, not once per
// candidate.  sA / sL: element strides from one matrix / one set of inverted diagonal blocks to the next.
struct Bat { int n = 1; size_t sA = 0, sL = 0; int* err = nullptr; };   // err: one flag per matrix (null: the context's own)

static int potrf_blocked(Ctx& c, double* A, int lda, int n, int extra, Bat bt = Bat(), bool one_stream = false)
{
    int* errflag = bt.err ? bt.err : c.scalars.as<int>() + 32;
    const bool two = chol_mode() == 1 && n > 2 * CHOL_NB && !one_stream;
    hipStream_t sM = c.stream, sL = c.stream;
    if (two) { MCML_TRY(lookahead_setup(c)); sL = c.aux; }
    auto leaf = [&](hipStream_t s, int k, int nb) -> int {
        hipLaunchKernelGGL(k_potrf_leaf, dim3(bt.n), dim3(LEAF_NT), POTRF_LDS, s, A + k + (size_t)k * lda, lda, nb,
                           c.linv.d() + (size_t)(k / CHOL_NB) * CHOL_NB * CHOL_NB, errflag, nullptr, bt.sA, bt.sL);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    };
    // C (M x N) = alpha A B' + beta C with B N-major, preferring the LDS-DMA kernel with a given tile
    auto gemm_nt = [&](int M, int N, int K, const double* Ap, const double* Bp, int ldb, double* Cp,
                       double alpha, double beta, bool lower, int tile, int inplace, int shift) -> int {
        EpiAxpby epi{Cp, lda, alpha, beta, bt.sA};
        const size_t sB = (ldb == CHOL_NB) ? bt.sL : bt.sA;            // B is an inverted diagonal block, or part of the matrix
        if (dl_applicable(M, N, K, Ap, lda, Bp, ldb, true))
            return launch_gemm_dl<true>(sM, M, N, K, Ap, lda, Bp, ldb, epi, lower, tile, inplace, shift, bt.n, bt.sA, sB);
        MCML_REQUIRE(bt.n == 1, "potrf: a batch of factorisations needs the LDS-DMA kernel for every panel product");
        MCML_REQUIRE(shift == 0 || !lower, "potrf: shifted lower-only update needs the LDS-DMA kernel");
        return launch_gemm<true>(sM, M, N, K, Ap, lda, Bp, ldb, epi, lower, inplace ? inplace : -1);
    };
    const int nsteps = (n + CHOL_NB - 1) / CHOL_NB;
    static const int SPW = CHOL_NB * (getenv("GLMMR_MCML_CHOL_SPW") ? atoi(getenv("GLMMR_MCML_CHOL_SPW")) : 8);   // super-panel width (panels)
    static const bool tl_off = getenv("GLMMR_MCML_CHOL_TWOLEVEL") && !strcmp(getenv("GLMMR_MCML_CHOL_TWOLEVEL"), "0");
    const bool twolevel = bt.n > 1 && n > SPW + CHOL_NB && !tl_off;
    MCML_TRY(leaf(sM, 0, n < CHOL_NB ? n : CHOL_NB));
    bool forked = false;
    for (int t = 0; t < nsteps; ++t) {
        const int k = t * CHOL_NB;
        const int nb = (n - k < CHOL_NB) ? n - k : CHOL_NB;
        double* A11 = A + k + (size_t)k * lda;
        const double* Linv = c.linv.d() + (size_t)t * CHOL_NB * CHOL_NB;
        const int rem = n - k - nb, R = rem + extra;
        if (R <= 0) break;
        double* A21 = A11 + nb;                                   // R x nb: the panel below the diagonal block
        const int nb2 = rem < CHOL_NB ? rem : CHOL_NB;            // rows of the next diagonal block (0 at the end)
        if (forked) { MCML_HIP(hipStreamWaitEvent(sM, c.ev_leaf, 0)); forked = false; }   // leaf(t) done
        // Two levels of blocking for a BATCH (bt.n > 1), whose rounds are bound by the trailing updates, not by the chain:
        // inside a super-panel of 8 panels the K = 128 updates touch the super-panel's own columns only; the columns to
        // its right receive all eight panels in ONE pass with K = 1024 when the super-panel is done (the same LDS-DMA
        // kernel: 48 against 35 TF on the trailing update of a 4000 x 4000 block, scripts/dl_k_sweep.py) -- five
        // read-modify-write passes over the trailing matrix at Q = 5000 instead of forty.  The sums are regrouped, so a
        // batched value equals the single evaluation to rounding (1e-13), not to the bit.
        const int sp0 = twolevel ? (k / SPW) * SPW : 0, sp1 = twolevel ? (sp0 + SPW < n ? sp0 + SPW : n) : n;
        const int inner = sp1 - (k + nb);                         // trailing columns this step's K = 128 update covers
        if (twolevel && inner == 0 && rem > 0) {
            // last panel of a super-panel: the whole panel, then everything to the right in one pass, then the next leaf
            MCML_TRY(gemm_nt(R, nb, nb, A21, Linv, CHOL_NB, A21, 1.0, 0.0, false, 0, 1, 0));
            const double* Asp = A + sp1 + (size_t)sp0 * lda;      // rows below the super-panel, its columns
            double* Csp = A + sp1 + (size_t)sp1 * lda;
            MCML_TRY(gemm_nt(R, rem, sp1 - sp0, Asp, Asp, lda, Csp, -1.0, 1.0, true, 0, 0, 0));
            MCML_TRY(leaf(sM, k + nb, nb2));
            continue;
        }
        if (nb2 > 0) {
            // the next diagonal block: its multiplier, its update, its factorisation
            MCML_TRY(gemm_nt(nb2, nb, nb, A21, Linv, CHOL_NB, A21, 1.0, 0.0, false, 7, 1, 0));
            double* T = A11 + nb + (size_t)nb * lda;
            MCML_TRY(gemm_nt(nb2, nb2, nb, A21, A21, lda, T, -1.0, 1.0, false, 8, 0, 0));
            if (two) {
                MCML_HIP(hipEventRecord(c.ev_col, sM));
                MCML_HIP(hipStreamWaitEvent(sL, c.ev_col, 0));
            }
            MCML_TRY(leaf(sL, k + nb, nb2));
            if (two) { MCML_HIP(hipEventRecord(c.ev_leaf, sL)); forked = true; }
        }
        // the bulk: the rest of the panel, then the whole trailing update except the block above
        if (R - nb2 > 0)
            MCML_TRY(gemm_nt(R - nb2, nb, nb, A21 + nb2, Linv, CHOL_NB, A21 + nb2, 1.0, 0.0, false, 0, 1, 0));
        if (rem > 0 && R - nb2 > 0) {
            double* C2 = A11 + nb + nb2 + (size_t)nb * lda;       // rows nb2.. of the trailing matrix, all its columns
            MCML_TRY(gemm_nt(R - nb2, twolevel ? inner : rem, nb, A21 + nb2, A21, lda, C2, -1.0, 1.0, true, 0, 0, nb2));
        }
    }
    if (forked) MCML_HIP(hipStreamWaitEvent(sM, c.ev_leaf, 0));
    return MCML_OK;
}

// The dependency structure the GRAPH is captured with (never launched eagerly, see "the hang" below).
// Two chains that only meet through events:
//   critical (capture stream):  P_small(t)  the 128 x 128 panel block under the diagonal block, X <- X inv(L_t)'
//                               U_small(t)  D_{t+1} -= X X'
//                               leaf(t+1)
//   bulk (side stream):         a(t)  the rest of panel t                       (needs leaf(t))
//                               b(t)  column blocks t+1 AND t+2, rows below block t+1, -= panel t   (needs P_small(t))
//                               c(t)  column blocks t+3.., lower tiles only, -= panel t
// P_small(t+1) / U_small(t+1) touch block (t+2, t+1) and D_{t+2}: both were brought up to date by b(t), so the
// critical chain waits for b(t) and never for the big update c(t) OF THE SAME STEP.  It is not independent of the big
// updates: b(t) sits behind c(t-1) on the bulk queue (and must: column block t+2 was last written by c(t-1)), so the
// critical chain has one step of slack against the bulk chain, which is what the early, throughput-bound steps use up;
// in the late steps the bulk chain runs ahead and the three critical kernels sit back to back on one queue.
// Every tile still receives its updates in panel order from kernels that accumulate k = 0..127 in order: the factor
// is bit-identical to potrf_blocked's.
//
// The hang (round 2: this structure as two LIVE streams, cross waits both ways, hung the process -- not the GPU -- on
// three of ~20 boxes; no log of it was kept).  What the code says: the dependency structure is acyclic if every
// hipStreamWaitEvent binds to the event's record AT THE TIME OF THE CALL, which is what HIP documents (CUDA's
// semantics) and what a capture does by construction -- the same structure has replayed > 27 000 times as a graph.
// The live version re-recorded the SAME four events every step while the other stream could still hold a pending
// wait on the previous record.  If the runtime ever resolves such a wait against the event's LATEST record instead,
// the bulk stream's wait for leaf(t+1) can land on leaf(t+2), which waits (through P_small(t+2)) for b(t+1) on the
// bulk stream behind that very wait: a cycle, and a host thread blocked in the next enqueue on a full queue -- the
// observed symptom.  That reading cannot be proved without the runtime's source; what is done instead is to remove
// the question: every dependency edge below has ITS OWN event (a ring, never re-recorded within a capture), so no wait
// can be bound to anything but the one record it was written for, and live launches keep the fork-join of
// potrf_blocked, whose waits always follow the record they mean on the same host thread with no later record pending.
// If the two-chain schedule is ever wanted live again, it must use the same ring.
static hipEvent_t ring_event(Ctx& c, size_t idx)
{
    while (c.ev_ring.size() <= idx) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        c.ev_ring.push_back(e);
    }
    return c.ev_ring[idx];
}
static int potrf_la2_capture(Ctx& c, double* A, int lda, int n, int extra, Bat bt = Bat())
{
    int* errflag = bt.err ? bt.err : c.scalars.as<int>() + 32;
    hipStream_t sC = c.stream, sB = c.aux_lo;
    auto leaf = [&](int k, int nb) -> int {
        hipLaunchKernelGGL(k_potrf_leaf, dim3(bt.n), dim3(LEAF_NT), POTRF_LDS, sC, A + k + (size_t)k * lda, lda, nb,
                           c.linv.d() + (size_t)(k / CHOL_NB) * CHOL_NB * CHOL_NB, errflag, nullptr, bt.sA, bt.sL);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    };
    auto gemm_nt = [&](hipStream_t st, int M, int N, int K, const double* Ap, const double* Bp, int ldb, double* Cp,
                       double alpha, double beta, bool lower, int tile, int inplace) -> int {
        EpiAxpby epi{Cp, lda, alpha, beta, bt.sA};
        const size_t sBb = (ldb == CHOL_NB) ? bt.sL : bt.sA;
        if (dl_applicable(M, N, K, Ap, lda, Bp, ldb, true))
            return launch_gemm_dl<true>(st, M, N, K, Ap, lda, Bp, ldb, epi, lower, tile, inplace, 0, bt.n, bt.sA, sBb);
        MCML_REQUIRE(bt.n == 1, "potrf: a batch of factorisations needs the LDS-DMA kernel for every panel product");
        return launch_gemm<true>(st, M, N, K, Ap, lda, Bp, ldb, epi, lower, inplace ? inplace : -1);
    };
    const int nsteps = (n + CHOL_NB - 1) / CHOL_NB;
    // one event per dependency edge: 3 per step (leaf, P_small, b) + the first leaf + the join
    auto EV = [&](int step, int kind) -> hipEvent_t { return ring_event(c, (size_t)3 * (step + 1) + kind); };
    MCML_REQUIRE(ring_event(c, (size_t)3 * (nsteps + 2)) != nullptr, "potrf: could not create the capture's events");
    MCML_TRY(leaf(0, n < CHOL_NB ? n : CHOL_NB));
    MCML_HIP(hipEventRecord(EV(-1, 0), sC));
    MCML_HIP(hipStreamWaitEvent(sB, EV(-1, 0), 0));                  // the side stream joins the capture; a(0) needs leaf(0)
    hipEvent_t last_b = nullptr;
    for (int t = 0; t < nsteps; ++t) {
        const int k = t * CHOL_NB;
        const int nb = (n - k < CHOL_NB) ? n - k : CHOL_NB;
        double* A11 = A + k + (size_t)k * lda;
        const double* Linv = c.linv.d() + (size_t)t * CHOL_NB * CHOL_NB;
        const int rem = n - k - nb, R = rem + extra;
        if (R <= 0) break;
        double* A21 = A11 + nb;
        const int nb2 = rem < CHOL_NB ? rem : CHOL_NB;
        const int nb3 = (rem - nb2 < CHOL_NB) ? rem - nb2 : CHOL_NB;
        const int w = nb2 + nb3;                                      // columns b(t) covers
        // ---- critical chain
        if (nb2 > 0) {
            if (last_b) MCML_HIP(hipStreamWaitEvent(sC, last_b, 0));  // b(t-1)
            MCML_TRY(gemm_nt(sC, nb2, nb, nb, A21, Linv, CHOL_NB, A21, 1.0, 0.0, false, 7, 1));
            double* T = A11 + nb + (size_t)nb * lda;
            MCML_TRY(gemm_nt(sC, nb2, nb2, nb, A21, A21, lda, T, -1.0, 1.0, false, 8, 0));
            MCML_HIP(hipEventRecord(EV(t, 1), sC));
            MCML_TRY(leaf(k + nb, nb2));
            MCML_HIP(hipEventRecord(EV(t, 0), sC));
        }
        // ---- bulk chain
        const int Rb = R - nb2;
        if (Rb > 0) {
            MCML_TRY(gemm_nt(sB, Rb, nb, nb, A21 + nb2, Linv, CHOL_NB, A21 + nb2, 1.0, 0.0, false, 0, 1));
            if (nb2 > 0) {
                MCML_HIP(hipStreamWaitEvent(sB, EV(t, 1), 0));
                double* Cb = A11 + nb + nb2 + (size_t)nb * lda;       // rows below block t+1, columns of blocks t+1, t+2
                MCML_TRY(gemm_nt(sB, Rb, w, nb, A21 + nb2, A21, lda, Cb, -1.0, 1.0, false, 0, 0));
                MCML_HIP(hipEventRecord(EV(t, 2), sB));
                last_b = EV(t, 2);
                if (rem - w > 0) {
                    double* Cc = A11 + nb + w + (size_t)(nb + w) * lda;
                    MCML_TRY(gemm_nt(sB, R - w, rem - w, nb, A21 + w, A21 + w, lda, Cc, -1.0, 1.0, true, 0, 0));
                }
            }
        } else if (nb2 > 0) {
            MCML_HIP(hipStreamWaitEvent(sB, EV(t, 1), 0));            // keep the side stream a descendant of every node
            MCML_HIP(hipEventRecord(EV(t, 2), sB));
            last_b = EV(t, 2);
        }
        if (nb2 > 0) MCML_HIP(hipStreamWaitEvent(sB, EV(t, 0), 0));   // a(t+1) needs leaf(t+1)
    }
    MCML_HIP(hipEventRecord(EV(nsteps, 0), sB));                      // join
    MCML_HIP(hipStreamWaitEvent(sC, EV(nsteps, 0), 0));
    return MCML_OK;
}

// The same factorisation replayed as a hipGraph.  A theta-step evaluates one (matrix, shape) 40 times per MCML
// iteration and every evaluation is ~200 launches, ~80 event operations and the host calls behind them: the first
// call with a given key runs eagerly (function attributes, allocations), the second is captured (fork / join of the
// look-ahead included: the side stream joins the capture through its event waits), later ones are one hipGraphLaunch.
// GLMMR_MCML_CHOL_GRAPH=0 keeps the eager launches.
// GLMMR_MCML_CHOL_GRAPH: 0 eager launches; old = the graph of potrf_blocked's own fork-join; default = potrf_la2_capture
static int chol_graph_kind()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GLMMR_MCML_CHOL_GRAPH");
        v = (e && !strcmp(e, "0")) ? 0 : (e && !strcmp(e, "old")) ? 1 : 2;
    }
    return v;
}
static bool chol_graph_on() { return chol_graph_kind() != 0; }
static int potrf_graphed(Ctx& c, double* A, int lda, int n, int extra, Bat bt = Bat())
{
    if (!chol_graph_on() || !(chol_mode() == 1 && n > 2 * CHOL_NB)) return potrf_blocked(c, A, lda, n, extra, bt);
    // A batch is launched EAGERLY with the two-stream fork-join of potrf_blocked (the leaf of step t+1 beside the bulk of
    // step t): 13.4 ms per round of 8 at Q = 5000 and 1.95 ms at Q = 2000, against 14.4 / 2.21 ms for the one-chain graph
    // (GLMMR_MCML_BATCH_GRAPH=1) -- a graph without a parallel branch cannot overlap the two, and one with a branch is
    // subject to the executable lottery described at CholGraph.  The host keeps ahead easily (~250 launches per round).
    static const bool batch_graph = getenv("GLMMR_MCML_BATCH_GRAPH") && !strcmp(getenv("GLMMR_MCML_BATCH_GRAPH"), "1");
    if (bt.n > 1 && !batch_graph) return potrf_blocked(c, A, lda, n, extra, bt);
    CholGraph& g = c.chol_graphs.find(A, c.linv.d(), lda, n, extra + (bt.n << 24), bt.n == 1);
    auto timed_begin = [&]() -> bool {
        if (!g.t0 && (hipEventCreate(&g.t0) != hipSuccess || hipEventCreate(&g.t1) != hipSuccess)) { (void)hipGetLastError(); return false; }
        return hipEventRecord(g.t0, c.stream) == hipSuccess;
    };
    if (g.exec) {
        // calibration of a fresh executable (see CholGraph): the previous replay's time is known by now -- every caller
        // synchronises for its result -- and an executable slower than the eager launches is instantiated again
        if (!g.settled && g.timed) {
            float ms = 0.f;
            g.timed = false;
            bool done = true;
            const hipError_t ee = hipEventElapsedTime(&ms, g.t0, g.t1);
            static const bool cal_trace = getenv("GLMMR_MCML_GRAPH_TRACE") != nullptr;
            if (cal_trace) fprintf(stderr, "graph calibration: n=%d extra=%d trial %d: %.3f ms (eager %.3f, best %.3f) rc=%d\n", g.n, g.extra & 0xffffff, g.tries, ms, g.eager_ms, g.best_ms, (int)ee);
            if (ee == hipSuccess && g.eager_ms > 0.f && g.tmpl) {
                // keep the fastest executable seen; try another instantiation unless this one clearly beats the eager
                // launches or four have been tried
                if (!g.best || ms < g.best_ms) {
                    if (g.best && g.best != g.exec) (void)hipGraphExecDestroy(g.best);
                    g.best = g.exec; g.best_ms = ms;
                } else if (g.exec != g.best) (void)hipGraphExecDestroy(g.exec);
                g.exec = g.best;
                if (!(g.best_ms <= 0.93f * g.eager_ms) && g.tries < 3) {
                    hipGraphExec_t ex2 = nullptr;
                    if (hipGraphInstantiate(&ex2, g.tmpl, nullptr, nullptr, 0) == hipSuccess) { g.exec = ex2; ++g.tries; g.trial_launches = 0; done = false; }
                    else (void)hipGetLastError();
                }
            } else (void)hipGetLastError();
            if (done) {
                g.settled = true; g.best = nullptr;
                if (g.tmpl) { (void)hipGraphDestroy(g.tmpl); g.tmpl = nullptr; }
            }
        }
        const bool tm = !g.settled && g.trial_launches >= 1 && timed_begin();
        MCML_HIP(hipGraphLaunch(g.exec, c.stream));
        if (tm) g.timed = hipEventRecord(g.t1, c.stream) == hipSuccess;
        ++g.trial_launches;
        return MCML_OK;
    }
    if (g.seen++ <= (bt.n == 1 ? 1 : 0)) {   // eager first (attributes, allocations); a single evaluation twice: the second run's time is the calibration's yardstick
        const bool tm = timed_begin();
        MCML_TRY(potrf_blocked(c, A, lda, n, extra, bt));
        if (tm && hipEventRecord(g.t1, c.stream) == hipSuccess && hipEventSynchronize(g.t1) == hipSuccess)
            (void)hipEventElapsedTime(&g.eager_ms, g.t0, g.t1);
        (void)hipGetLastError();
        return MCML_OK;
    }
    MCML_TRY(lookahead_setup(c));
    if (hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();                                           // e.g. the legacy default stream: eager for good
        g.seen = -(1 << 30);
        return potrf_blocked(c, A, lda, n, extra, bt);
    }
    // A batch is captured as ONE chain (potrf_blocked on a single stream): with k matrices per launch the late steps'
    // chain is amortised anyway (measured eagerly: 17.1 against 16.0 ms per round of 8 at Q = 5000), and a graph without
    // a parallel branch does not depend on which hardware queue the runtime gives that branch -- see CholGraph: of
    // several two-chain executables alive in a process only the first two reliably overlap their chains, and a theta-step
    // uses a different batch width for its last, partial rounds.  The single evaluation keeps the two-chain graph.
    const bool linear = bt.n > 1;
    const int rc = linear ? potrf_blocked(c, A, lda, n, extra, bt, true)
                          : chol_graph_kind() == 2 ? potrf_la2_capture(c, A, lda, n, extra, bt) : potrf_blocked(c, A, lda, n, extra, bt);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c.stream, &graph);
    if (rc != MCML_OK || e != hipSuccess || !graph) {
        // whatever went wrong while recording (nothing has run): eager launches from now on, the error if that fails too
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        g.seen = -(1 << 30);
        return potrf_blocked(c, A, lda, n, extra, bt);
    }
    const hipError_t ei = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        g.exec = nullptr; g.seen = -(1 << 30);
        return potrf_blocked(c, A, lda, n, extra, bt);
    }
    g.tmpl = graph; g.tries = 0; g.settled = linear || !(g.eager_ms > 0.f); g.timed = false; g.best = nullptr; g.best_ms = 0.f;
    if (g.settled) { (void)hipGraphDestroy(g.tmpl); g.tmpl = nullptr; }
    g.trial_launches = 1;                     // this first launch is not timed (it carries the executable's upload)
    MCML_HIP(hipGraphLaunch(g.exec, c.stream));
    return MCML_OK;
}

static int trsm_left_blocked(Ctx& c, const double* L, int ldl, int n, double* U, int ldu, int m)
{
    for (int k = 0; k < n; k += CHOL_NB) {
        const int nb = (n - k < CHOL_NB) ? n - k : CHOL_NB;
        const double* Linv = c.linv.d() + (size_t)(k / CHOL_NB) * CHOL_NB * CHOL_NB;
        double* Uk = U + k;
        {
            EpiAxpby epi{Uk, ldu, 1.0, 0.0};
            MCML_TRY(chol_gemm<false>(c.stream, nb, m, nb, Linv, CHOL_NB, Uk, ldu, epi, false, 2));
        }
        const int rem = n - k - nb;
        if (rem <= 0) break;
        const double* L21 = L + (k + nb) + (size_t)k * ldl;
        EpiAxpby epi{Uk + nb, ldu, -1.0, 1.0};
        MCML_TRY(chol_gemm<false>(c.stream, rem, m, nb, L21, ldl, Uk, ldu, epi, false, 0));
    }
    return MCML_OK;
}

int potrf_lower(Ctx& c, double* A, int n, int lda)
{
    MCML_REQUIRE(n > 0 && lda >= n && (lda & 1) == 0, "potrf: bad shape n=%d lda=%d", n, lda);
    MCML_TRY(c.linv.ensure(sizeof(double) * (size_t)(n / CHOL_NB + 1) * CHOL_NB * CHOL_NB));
    // the leaves write the lower triangles of their inverses only (k_potrf_leaf)
    MCML_HIP(hipMemsetAsync(c.linv.p, 0, sizeof(double) * (size_t)((n + CHOL_NB - 1) / CHOL_NB) * CHOL_NB * CHOL_NB, c.stream));
    MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_potrf_leaf), (int)POTRF_LDS));
    if (chol_blocked()) return potrf_blocked(c, A, lda, n, 0);
    return potrf_rec(c, A, lda, 0, n);
}

// U (n x m) <- inv(L) U, L = A0[off.., off..]; needs the leaf inverses potrf_lower left in c.linv.
// Recursive splitting: the update U2 -= L21 U1 is ONE product with K = n/2, n/4, ... (long-K GEMMs
// at full MFMA efficiency) instead of n/128 products with K = 128.
static int trsm_left_rec(Ctx& c, const double* A0, int lda, int off, int n, double* U, int ldu, int m)
{
    if (n <= CHOL_NB) {
        const double* Linv = c.linv.d() + (size_t)(off / CHOL_NB) * CHOL_NB * CHOL_NB;
        EpiAxpby epi{U, ldu, 1.0, 0.0};
        // in place: a workgroup reads its whole column band (K = n <= 128) before it writes
        return chol_gemm<false>(c.stream, n, m, n, Linv, CHOL_NB, U, ldu, epi, false, 2);
    }
    const int n1 = split128(n), n2 = n - n1;
    MCML_TRY(trsm_left_rec(c, A0, lda, off, n1, U, ldu, m));
    const double* L21 = A0 + (off + n1) + (size_t)off * lda;
    double* U2 = U + n1;
    EpiAxpby epi{U2, ldu, -1.0, 1.0};
    static const bool big_dlds = getenv("GLMMR_MCML_TRSM_BIG") && !strcmp(getenv("GLMMR_MCML_TRSM_BIG"), "dlds");
    if (big_dlds && n1 > CHOL_NB && dlds_applicable(n2, m, n1, L21, lda, n1 + 32, U, ldu))
        MCML_TRY(launch_gemm_dlds(c.stream, n2, m, n1, L21, lda, U, ldu, epi));
    else
        MCML_TRY(chol_gemm<false>(c.stream, n2, m, n1, L21, lda, U, ldu, epi, false, 0));
    return trsm_left_rec(c, A0, lda, off + n1, n2, U2, ldu, m);
}

int trsm_left_lower(Ctx& c, const double* L, int ldl, int n, double* U, int ldu, int m)
{
    // GLMMR_MCML_TRSM=rec: the recursive variant (long-K updates; measured 3-15 % slower at Q = 5000, m = 1024
    // than the panel-by-panel one with the two-per-CU K = 128 tiles)
    static const bool rec = getenv("GLMMR_MCML_TRSM") && !strcmp(getenv("GLMMR_MCML_TRSM"), "rec");
    if (chol_blocked() && !(rec && m >= 64)) return trsm_left_blocked(c, L, ldl, n, U, ldu, m);
    return trsm_left_rec(c, L, ldl, 0, n, U, ldu, m);
}

// ------------------------------------------------------------------ setup
int mvn_setup(Ctx& c)
{
    const CovSpec& cs = c.cov;
    MCML_TRY(c.d_cov.ensure(sizeof(int32_t) * cs.cov.size()));
    MCML_TRY(copy_h2d(c.d_cov.p, cs.cov.data(), sizeof(int32_t) * cs.cov.size(), c.stream));
    MCML_TRY(c.d_data.ensure(sizeof(double) * (cs.data.size() + 1)));
    if (!cs.data.empty())
        MCML_TRY(copy_h2d(c.d_data.p, cs.data.data(), sizeof(double) * cs.data.size(), c.stream));
    MCML_TRY(c.d_blocks.ensure(sizeof(CovBlock) * cs.blocks.size()));
    MCML_TRY(copy_h2d(c.d_blocks.p, cs.blocks.data(), sizeof(CovBlock) * cs.blocks.size(), c.stream));
    std::vector<int> rowblock(cs.N, -1), small;
    c.maxdim_large = 0; c.n_small = 0; c.n_diag_rows = 0;
    for (int b = 0; b < cs.B; ++b) {
        const CovBlock& blk = cs.blocks[b];
        if (blk.all_gr) {
            for (int k = 0; k < blk.dim; ++k) rowblock[blk.matstart + k] = b;
            c.n_diag_rows += blk.dim;
        } else if (blk.dim <= SMALL_BLOCK) {
            ++c.n_small;
            small.push_back(b);
        } else if (blk.dim > c.maxdim_large) {
            c.maxdim_large = blk.dim;
        }
    }
    MCML_TRY(c.d_rowblock.ensure(sizeof(int) * (size_t)(cs.N + 1)));
    MCML_TRY(copy_h2d(c.d_rowblock.p, rowblock.data(), sizeof(int) * cs.N, c.stream));
    MCML_TRY(c.scalars.ensure(sizeof(double) * 64));
    MCML_HIP(hipMemsetAsync(c.scalars.p, 0, sizeof(double) * 64, c.stream));
    if (c.n_small) {
        MCML_TRY(c.small_ids.ensure(sizeof(int) * small.size()));
        MCML_TRY(copy_h2d(c.small_ids.p, small.data(), sizeof(int) * small.size(), c.stream));
    }
    if (c.maxdim_large) {
        MCML_TRY(c.Dwork.alloc(c.maxdim_large, c.maxdim_large));
        MCML_HIP(hipMemsetAsync(c.Dwork.d(), 0, sizeof(double) * (size_t)c.Dwork.ld * c.maxdim_large, c.stream));
        // room for a whole round of candidates (mvn_loglik_batch) from the start: growing the buffer later would change
        // the pointer the single evaluation's captured graph is keyed on
        MCML_TRY(c.linv.ensure(sizeof(double) * (size_t)(round_up(c.maxdim_large, 16) / CHOL_NB + 1) * CHOL_NB * CHOL_NB * MVN_MAXBATCH));
    }
    MCML_HIP(hipStreamSynchronize(c.stream));
    return MCML_OK;
}

static int theta_arg(const Ctx& c, const double* theta, ThetaArg& th)
{
    MCML_REQUIRE(theta, "theta is null");
    memset(&th, 0, sizeof th);
    for (int i = 0; i < c.cov.npar; ++i) th.v[i] = theta[i];
    return MCML_OK;
}

static int check_errflag(Ctx& c, const char* what)
{
    int flag = 0;
    MCML_TRY(copy_d2h(&flag, c.scalars.as<int>() + 32, sizeof(int), c.stream));
    MCML_HIP(hipStreamSynchronize(c.stream));
    if (flag) {
        MCML_HIP(hipMemsetAsync(c.scalars.as<int>() + 32, 0, sizeof(int), c.stream));
        set_error("%s: covariance block is not positive definite", what);
        return MCML_ENOTPD;
    }
    return MCML_OK;
}

int potrf_lower_checked(Ctx& c, double* A, int n, int lda)
{
    MCML_TRY(potrf_lower(c, A, n, lda));
    return check_errflag(c, "potrf");
}

// ------------------------------------------------------------------ loglik
int mvn_loglik_sum(Ctx& c, const double* theta, double* sum_out)
{
    MCML_REQUIRE(c.mcols > 0 && c.U.d(), "mvn_ll: no samples set");
    return mvn_loglik_sum_on(c, theta, c.U.d(), c.U.ld, c.mcols, sum_out);
}

static int mvn_loglik_enqueue(Ctx& c, const double* theta, const double* Us, int ldu, int m);

int mvn_loglik_sum_on(Ctx& c, const double* theta, const double* Us, int ldu, int m, double* sum_out)
{
    MCML_TRY(mvn_loglik_enqueue(c, theta, Us, ldu, m));
    MCML_TRY(copy_d2h(sum_out, c.scalars.d(), sizeof(double), c.stream));
    MCML_TRY(check_errflag(c, "mvn_ll"));
    return MCML_OK;
}

// k candidate thetas in ONE pass of the factorisation's schedule.  A single evaluation of a large dense block is a
// latency chain (40 steps of leaf + two single-block products, ~70 us each, on a handful of CUs; 0.24 of the FP64 MFMA
// peak at Q = 5000); with the k matrices side by side in one workspace every launch covers all of them, so the chain
// is paid once per round.  (Measured dead end: the k evaluations as k independent streams / graphs -- "lanes" -- do
// not overlap on this stack: 3.6 ms per evaluation alone, 4.1 / 4.8 / 5.0 ms each with 2 / 4 / 8 lanes, worse with
// more hardware queues.)  Models whose D has diagonal or small blocks besides, and k = 1, take the single path.
int mvn_loglik_batch(Ctx& c, const double* thetas, int k, const double* Us, int ldu, int m, double* sums, int* rcs)
{
    MCML_REQUIRE(k >= 1 && thetas && sums && rcs && m > 0 && Us, "mvn_ll batch: bad arguments");
    const CovSpec& cs = c.cov;
    const int R = cs.npar;
    static const bool off = getenv("GLMMR_MCML_MVN_BATCH") && !strcmp(getenv("GLMMR_MCML_MVN_BATCH"), "0");
    const bool batchable = !off && k > 1 && c.maxdim_large > 0 && c.n_small == 0 && c.n_diag_rows == 0 && chol_blocked();
    if (!batchable) {
        int first_rc = MCML_OK;
        for (int j = 0; j < k; ++j) {
            rcs[j] = mvn_loglik_sum_on(c, thetas + (size_t)j * R, Us, ldu, m, sums + j);
            if (rcs[j] != MCML_OK && rcs[j] != MCML_ENOTPD && first_rc == MCML_OK) first_rc = rcs[j];
        }
        return first_rc;
    }
    for (int j0 = 0; j0 < k; j0 += MVN_MAXBATCH) {
        const int kb = (k - j0 < MVN_MAXBATCH) ? k - j0 : MVN_MAXBATCH;
        ThetaBatch tb;
        memset(&tb, 0, sizeof tb);
        for (int j = 0; j < kb; ++j)
            for (int i = 0; i < R; ++i) tb.t[j].v[i] = thetas[(size_t)(j0 + j) * R + i];
        const int dmax = round_up(c.maxdim_large, 16);
        // kb matrices side by side: matrix j = columns [j * dmax, (j + 1) * dmax), the m sample rows below each
        if (c.Dbatch.rows < dmax + m || c.Dbatch.cols < kb * dmax) {
            MCML_TRY(c.Dbatch.alloc(dmax + m, MVN_MAXBATCH * dmax));
            MCML_HIP(hipMemsetAsync(c.Dbatch.d(), 0, sizeof(double) * (size_t)c.Dbatch.ld * MVN_MAXBATCH * dmax, c.stream));
        }
        const int ld = c.Dbatch.ld;
        const size_t sA = (size_t)ld * dmax;
        const int nst = dmax / CHOL_NB + 1;
        const size_t sL = (size_t)nst * CHOL_NB * CHOL_NB;
        MCML_TRY(c.linv.ensure(sizeof(double) * sL * MVN_MAXBATCH));
        // results: 4 doubles per candidate, then one "not positive definite" flag (int) per candidate -- a buffer of
        // their own (c.scalars is shared with the sampler's diagnostics)
        MCML_TRY(c.bscal.ensure(sizeof(double) * 5 * MVN_MAXBATCH));
        MCML_HIP(hipMemsetAsync(c.bscal.p, 0, sizeof(double) * 5 * MVN_MAXBATCH, c.stream));
        int* bflags = reinterpret_cast<int*>(c.bscal.d() + 4 * MVN_MAXBATCH);
        const int32_t* dcov = c.d_cov.as<int32_t>();
        const CovBlock* dblk = c.d_blocks.as<CovBlock>();
        for (int b = 0; b < cs.B; ++b) {
            const CovBlock& blk = cs.blocks[b];
            const int d = blk.dim, dp = round_up(d, 16);
            hipLaunchKernelGGL(k_build_dense_batch, dim3((dp + 63) / 64, (dp + 15) / 16, kb), dim3(256), 0, c.stream,
                               c.Dbatch.d(), ld, sA, b, dblk, dcov, cs.rows, c.d_data.d(), tb, dp);
            hipLaunchKernelGGL(k_transpose_in, dim3((d + 31) / 32, (m + 31) / 32, kb), dim3(256), 0, c.stream,
                               Us + blk.matstart, ldu, d, m, c.Dbatch.d() + dp, ld, sA);
            MCML_HIP(hipGetLastError());
            if (dp > d)          // the border columns of the sample rows must be finite: they meet zeros only
                for (int j = 0; j < kb; ++j)
                    MCML_HIP(hipMemset2DAsync(c.Dbatch.d() + j * sA + dp + (size_t)d * ld, sizeof(double) * ld, 0, sizeof(double) * m, dp - d, c.stream));
            MCML_HIP(hipMemsetAsync(c.linv.p, 0, sizeof(double) * sL * kb, c.stream));
            MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_potrf_leaf), (int)POTRF_LDS));
            Bat bt; bt.n = kb; bt.sA = sA; bt.sL = sL; bt.err = bflags;
            MCML_TRY(potrf_graphed(c, c.Dbatch.d(), ld, dp, m, bt));
            const int gx = (m + 255) / 256, gy = d < 64 ? d : 64;
            MCML_TRY(c.partials.ensure(sizeof(double) * (size_t)(gx * gy + 16)));
            for (int j = 0; j < kb; ++j) {
                double* scal = c.bscal.d() + 4 * j;
                const double* Aj = c.Dbatch.d() + j * sA;
                hipLaunchKernelGGL(k_sumsq, dim3(gx, gy), dim3(256), 0, c.stream, Aj + dp, ld, m, d, c.partials.d());
                hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, c.partials.d(), gx * gy, 1.0, scal + 2, 0);
                hipLaunchKernelGGL(k_logdet, dim3(1), dim3(256), 0, c.stream, Aj, ld, d, scal + 1);
                hipLaunchKernelGGL(k_finish_large, dim3(1), dim3(1), 0, c.stream, scal, d, m);
            }
            MCML_HIP(hipGetLastError());
        }
        double hs[4 * MVN_MAXBATCH]; int hf[MVN_MAXBATCH];
        MCML_TRY(copy_d2h(hs, c.bscal.p, sizeof(double) * 4 * kb, c.stream));
        MCML_TRY(copy_d2h(hf, bflags, sizeof(int) * kb, c.stream));
        MCML_HIP(hipStreamSynchronize(c.stream));
        for (int j = 0; j < kb; ++j) { sums[j0 + j] = hs[4 * j]; rcs[j0 + j] = hf[j] ? MCML_ENOTPD : MCML_OK; }
    }
    return MCML_OK;
}

static int mvn_loglik_enqueue(Ctx& c, const double* theta, const double* Us, int ldu, int m)
{
    MCML_REQUIRE(m > 0 && Us, "mvn_ll: no samples set");
    ThetaArg th;
    MCML_TRY(theta_arg(c, theta, th));
    const CovSpec& cs = c.cov;
    const int Q = cs.N;
    double* scal = c.scalars.d();
    const int32_t* dcov = c.d_cov.as<int32_t>();
    const CovBlock* dblk = c.d_blocks.as<CovBlock>();
    const int* drb = c.d_rowblock.as<int>();
    MCML_HIP(hipMemsetAsync(scal, 0, sizeof(double) * 4, c.stream));

    if (c.n_diag_rows > 0) {
        MCML_TRY(c.partials.ensure(sizeof(double) * (size_t)(2 * Q + 65536)));
        double* dd = c.partials.d();
        double* dc = dd + Q;
        double* part = dc + Q;
        hipLaunchKernelGGL(k_diag_prep, dim3((Q + 255) / 256), dim3(256), 0, c.stream, Q, drb, dblk, dcov,
                           cs.rows, th, dd, dc);
        dim3 grid((Q + 255) / 256, 1);
        int gy = 65536 / (int)grid.x; if (gy > m) gy = m; if (gy > 64) gy = 64; if (gy < 1) gy = 1;
        grid.y = gy;
        hipLaunchKernelGGL(k_diag_ll, grid, dim3(256), 0, c.stream, Us, ldu, Q, m, drb, dd, dc, part);
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, part, (int)(grid.x * grid.y), 1.0, scal, 1);
        MCML_HIP(hipGetLastError());
    }
    if (c.n_small > 0) {
        // the block ids were uploaded once by mvn_setup: nothing here needs the host to wait (this was a host
        // synchronisation per evaluation, 40 of them in a theta-step of config 4)
        int ny = (m + 63) / 64; if (ny > 16) ny = 16;
        DevBuf& wb = c.scratch;      // scratch that outlives the launch
        MCML_TRY(wb.ensure(sizeof(double) * (size_t)c.n_small * ny + 1024));
        double* part = wb.d();
        hipLaunchKernelGGL(k_small_ll, dim3((unsigned)c.n_small, ny), dim3(64), 0, c.stream, Us, ldu, m,
                           c.small_ids.as<int>(), dblk, dcov, cs.rows, c.d_data.d(), th, part, c.scalars.as<int>() + 32);
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, part, c.n_small * ny, 1.0, scal, 1);
        MCML_HIP(hipGetLastError());
    }
    if (c.maxdim_large > 0) {
        // blocked path: the m sample columns ride below the matrix as m extra rows (U', one sample per row), so
        // the forward substitution happens inside the factorisation (potrf_blocked); recursive path: separate TRSM
        const bool aug = chol_blocked();
        if (aug) {
            if (c.Dwork.rows < c.maxdim_large + 16 + m || c.Dwork.cols < c.maxdim_large + 16) {
                MCML_TRY(c.Dwork.alloc(c.maxdim_large + 16 + m, c.maxdim_large + 16));
                MCML_HIP(hipMemsetAsync(c.Dwork.d(), 0, sizeof(double) * (size_t)c.Dwork.ld * (c.maxdim_large + 16), c.stream));
            }
        } else {
            MCML_TRY(c.Uwork.alloc(c.maxdim_large, m));
        }
        for (int b = 0; b < cs.B; ++b) {
            const CovBlock& blk = cs.blocks[b];
            if (blk.all_gr || blk.dim <= SMALL_BLOCK) continue;
            const int d = blk.dim;
            dim3 g((d + 15) / 16, (d + 15) / 16);
            // a multiple of 16 (identity border): the extra rows and every panel stay 16-byte aligned and the last, ragged
            // panel still has a K the LDS-DMA kernel takes (5000 = 39 x 128 + 8 would fall back to the register-staged one)
            const int dp = aug ? round_up(d, 16) : d;
            g = dim3((dp + 63) / 64, (dp + 15) / 16);
            hipLaunchKernelGGL(k_build_dense, g, dim3(256), 0, c.stream, c.Dwork.d(), c.Dwork.ld, b, dblk, dcov,
                               cs.rows, c.d_data.d(), th, 0, dp);
            MCML_HIP(hipGetLastError());
            const double* Z; int ldz, zr, zc;        // the solved samples: zr x zc, sum of squares wanted
            if (aug) {
                hipLaunchKernelGGL(k_transpose_in, dim3((d + 31) / 32, (m + 31) / 32), dim3(256), 0, c.stream,
                                   Us + blk.matstart, ldu, d, m, c.Dwork.d() + dp, c.Dwork.ld);
                if (dp > d)        // the border columns of the sample rows must be finite: they meet zeros only
                    MCML_HIP(hipMemset2DAsync(c.Dwork.d() + dp + (size_t)d * c.Dwork.ld, sizeof(double) * c.Dwork.ld, 0, sizeof(double) * m, dp - d, c.stream));
                MCML_HIP(hipGetLastError());
                MCML_TRY(c.linv.ensure(sizeof(double) * (size_t)(dp / CHOL_NB + 1) * CHOL_NB * CHOL_NB));
                MCML_HIP(hipMemsetAsync(c.linv.p, 0, sizeof(double) * (size_t)((dp + CHOL_NB - 1) / CHOL_NB) * CHOL_NB * CHOL_NB, c.stream));
                MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_potrf_leaf), (int)POTRF_LDS));
                MCML_TRY(potrf_graphed(c, c.Dwork.d(), c.Dwork.ld, dp, m));
                Z = c.Dwork.d() + dp; ldz = c.Dwork.ld; zr = m; zc = d;
            } else {
                MCML_TRY(potrf_lower(c, c.Dwork.d(), d, c.Dwork.ld));
                int gy = m < 256 ? m : 256;
                hipLaunchKernelGGL(k_copy_block, dim3((d + 255) / 256, gy), dim3(256), 0, c.stream, c.Uwork.d(),
                                   c.Uwork.ld, Us + blk.matstart, ldu, d, m);
                MCML_HIP(hipGetLastError());
                MCML_TRY(trsm_left_lower(c, c.Dwork.d(), c.Dwork.ld, d, c.Uwork.d(), c.Uwork.ld, m));
                Z = c.Uwork.d(); ldz = c.Uwork.ld; zr = d; zc = m;
            }
            int gx = (zr + 255) / 256, gy = zc < 64 ? zc : 64;
            MCML_TRY(c.partials.ensure(sizeof(double) * (size_t)(gx * gy + 16)));
            hipLaunchKernelGGL(k_sumsq, dim3(gx, gy), dim3(256), 0, c.stream, Z, ldz, zr, zc, c.partials.d());
            hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, c.partials.d(), gx * gy, 1.0, scal + 2, 0);
            hipLaunchKernelGGL(k_logdet, dim3(1), dim3(256), 0, c.stream, c.Dwork.d(), c.Dwork.ld, d, scal + 1);
            hipLaunchKernelGGL(k_finish_large, dim3(1), dim3(1), 0, c.stream, scal, d, m);
            MCML_HIP(hipGetLastError());
        }
    }
    return MCML_OK;
}

// ------------------------------------------------------------------ genD
__global__ void k_diag_fill(double* L, int ldl, int Q, const int* rowblock, const CovBlock* blocks,
                            const int32_t* cov, int rows, ThetaArg th, int chol)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Q) return;
    int b = rowblock[k];
    if (b < 0) return;
    CovBlock blk = blocks[b];
    double val = 1.0;
    for (int r = blk.r0; r < blk.r1; ++r) val = cov_term(1, 0.0, &th.v[cov[r + 4 * rows]], val);
    L[k + (size_t)k * ldl] = chol ? sqrt(val) : val;
}

int mvn_gen_L(Ctx& c, const double* theta, bool chol)
{
    ThetaArg th;
    MCML_TRY(theta_arg(c, theta, th));
    const CovSpec& cs = c.cov;
    const int Q = cs.N;
    MCML_TRY(c.L.alloc(Q, Q));
    MCML_HIP(hipMemsetAsync(c.L.d(), 0, sizeof(double) * (size_t)c.L.ld * Q, c.stream));
    const int32_t* dcov = c.d_cov.as<int32_t>();
    const CovBlock* dblk = c.d_blocks.as<CovBlock>();
    if (c.n_diag_rows > 0)
        hipLaunchKernelGGL(k_diag_fill, dim3((Q + 255) / 256), dim3(256), 0, c.stream, c.L.d(), c.L.ld, Q,
                           c.d_rowblock.as<int>(), dblk, dcov, cs.rows, th, chol ? 1 : 0);
    for (int b = 0; b < cs.B; ++b) {
        const CovBlock& blk = cs.blocks[b];
        if (blk.all_gr) continue;
        const int d = blk.dim;
        double* A = c.L.at(blk.matstart, blk.matstart);
        dim3 g((d + 63) / 64, (d + 15) / 16);
        hipLaunchKernelGGL(k_build_dense, g, dim3(256), 0, c.stream, A, c.L.ld, b, dblk, dcov, cs.rows,
                           c.d_data.d(), th, chol ? 0 : 1, d);
        MCML_HIP(hipGetLastError());
        if (chol) {
            // blocks start at arbitrary (possibly odd) offsets: factorise in the
            // aligned workspace when the view is not 16-byte aligned
            if (d <= CHOL_NB || (blk.matstart & 1) == 0) {
                MCML_TRY(potrf_lower(c, A, d, c.L.ld));
            } else {
                DevMat tmp;
                MCML_TRY(tmp.alloc(d, d));
                hipLaunchKernelGGL(k_copy_block, dim3((d + 255) / 256, d < 256 ? d : 256), dim3(256), 0, c.stream,
                                   tmp.d(), tmp.ld, A, c.L.ld, d, d);
                MCML_TRY(potrf_lower(c, tmp.d(), d, tmp.ld));
                hipLaunchKernelGGL(k_copy_block, dim3((d + 255) / 256, d < 256 ? d : 256), dim3(256), 0, c.stream,
                                   A, c.L.ld, tmp.d(), tmp.ld, d, d);
                MCML_HIP(hipStreamSynchronize(c.stream));
            }
            // the SYRK updates of diagonal tiles also touch their upper halves
            if (d > CHOL_NB)
                hipLaunchKernelGGL(k_zero_upper, dim3((d + 255) / 256, d), dim3(256), 0, c.stream, A, c.L.ld, d);
        }
    }
    MCML_HIP(hipGetLastError());
    MCML_TRY(check_errflag(c, "genD"));
    if (chol) c.l_foreign = false;
    c.have_L = chol;
    return MCML_OK;
}

}  // namespace mcml
