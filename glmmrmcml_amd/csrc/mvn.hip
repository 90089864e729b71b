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
__global__ __launch_bounds__(256) void k_build_dense(double* A, int lda, int bidx, const CovBlock* blocks,
                                                     const int32_t* cov, int rows, const double* data,
                                                     ThetaArg th, int mirror)
{
    if (blockIdx.x < blockIdx.y) return;
    const CovBlock blk = blocks[bidx];
    const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= blk.dim || j >= blk.dim || i < j) return;
    double v = cov_entry(blk, cov, rows, data + blk.doff, th, i, j);
    A[i + (size_t)j * lda] = v;
    if (mirror && i != j) A[j + (size_t)i * lda] = v;
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
// chain on a single wave: every exec-mask branch in it costs a pipeline bubble).  `bad`
// accumulates failed pivots, `yv` collects lane r's reciprocal pivot 1 / L_rr.
template <int C>
__device__ __forceinline__ void potrf16_col(double (&a)[16], int r, int& bad, double& yv)
{
    double diag = bcast_lane<C>(a[C]);
    const bool ok = diag > 0.0;
    bad |= ok ? 0 : 1;
    diag = ok ? diag : 1.0;
    const double y = rsqrt_nr(diag);
    double d = diag * y;                                   // sqrt(diag), one correction step
    d = __builtin_fma(__builtin_fma(-d, d, diag), 0.5 * y, d);
    const double l = (r == C) ? d : a[C] * y;
    yv = (r == C) ? y : yv;
    a[C] = l;
    if constexpr (C < 15) {
        // a[j] -= l * l_j for j > C, l_j = lane j's l
        potrf16_upd<C, C + 1>(a, l, r);
    }
}

// Leaf of the recursive factorisation: n <= 128.  One workgroup keeps the
// block in LDS (row stride 129: conflict-free column walks), factorises it in
// 16-wide panels (the 16x16 diagonal tile in registers via wave shuffles), then
// inverts the factor in place into the unused upper triangle.  Writes L over
// the lower triangle of A and inv(L) to Linv (128 x 128, ld 128, upper zeroed).
constexpr int LEAF_NT = 512, LEAF_NW = LEAF_NT / 64;      // 8 waves: the panel / trailing / inverse phases are wave-parallel
__global__ __launch_bounds__(LEAF_NT) void k_potrf_leaf(double* A, int lda, int n, double* Linv, int* errflag,
                                                    unsigned long long* prof)
{
#pragma clang fp contract(fast)      // the factorisation is not part of the bit-exact RNG / leapfrog contract
#define LEAF_T(i) do { if (prof && threadIdx.x == 0) prof[i] = __builtin_amdgcn_s_memtime(); } while (0)
    LEAF_T(0);
    extern __shared__ __attribute__((aligned(16))) double S[];
    constexpr int LS = 129;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    unsigned long long ta = 0, tb = 0, tc = 0, t_;
    const int ntile = (n + 15) >> 4;
    // ---- inverse X = inv(L), 16 x 16 tiles; X[i][j] (i > j) is kept at S[j][i] (the unused upper
    // triangle), its diagonal (= the reciprocal pivots phase (a) stores) in xd.  Block row I is
    // complete once tile (I, I) is factorised, so waves 1..7 compute row I of the inverse while
    // wave 0 runs the serial phase (a) of step I + 1: off the critical path.  Wave w owns column
    // tile J = w - 1 and keeps its T_J in a private scratch tile -- no barrier is needed inside
    // the row.  The diagonal tiles X_II cost nothing: phase (b) solves x L11' = e_j for the 16
    // rows of the identity along with the panel rows, which is X_II' written straight into the
    // upper part of the diagonal tile.
    double* xd = S + 128 * LS;            // 128 doubles
    double* Tt = xd + 128;                // 7 tiles of 16 x 16
    auto inverse_row = [&](int I) {
        const int J = wave - 1;
        const int l15 = lane & 15, lk = lane >> 4;
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
            if (gi < n) S[gj * LS + gi] = -acc2[r];
        }
    };
    for (int kt = 0; kt < ntile; ++kt) {
        const int kb = kt * 16;
        t_ = __builtin_amdgcn_s_memtime();
        if (wave == 0) {
            // (a) 16x16 diagonal tile, one row per lane (lanes >= 16 mirror lane&15)
            const int r = lane & 15;
            int bad = 0; double yv = 1.0;                  // failed pivots; this lane's reciprocal pivot
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c)
                a[c] = (kb + r < n && kb + c < n) ? S[(kb + r) * LS + kb + c] : (r == c ? 1.0 : 0.0);
            potrf16_col<0>(a, r, bad, yv);  potrf16_col<1>(a, r, bad, yv);
            potrf16_col<2>(a, r, bad, yv);  potrf16_col<3>(a, r, bad, yv);
            potrf16_col<4>(a, r, bad, yv);  potrf16_col<5>(a, r, bad, yv);
            potrf16_col<6>(a, r, bad, yv);  potrf16_col<7>(a, r, bad, yv);
            potrf16_col<8>(a, r, bad, yv);  potrf16_col<9>(a, r, bad, yv);
            potrf16_col<10>(a, r, bad, yv); potrf16_col<11>(a, r, bad, yv);
            potrf16_col<12>(a, r, bad, yv); potrf16_col<13>(a, r, bad, yv);
            potrf16_col<14>(a, r, bad, yv); potrf16_col<15>(a, r, bad, yv);
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (r >= c && kb + r < n && kb + c < n) S[(kb + r) * LS + kb + c] = a[c];
                S[128 * LS + kb + r] = yv;                 // reciprocal diagonal = the inverse's diagonal (xd)
                if (bad && lane == 0) atomicExch(errflag, 1);
            }
        } else if (kt >= 1) {
            inverse_row(kt - 1);
        }
        __syncthreads();
        ta += __builtin_amdgcn_s_memtime() - t_; t_ = __builtin_amdgcn_s_memtime();
        // (b) panel below the tile: x L11^T = a, one row per thread (rows padded to the tile grid
        // are zero and stay zero).  The tile's own 16 rows ride along with right-hand side e_j:
        // their solution is row j of inv(L11)', i.e. X_II, stored in the tile's upper part.
        // Right-looking substitution: once x[k] is known every later column's partial sum is updated
        // at once, so the 120 multiply-adds of a row pipeline instead of forming 16 long chains.
        {
            const bool ident = wave == LEAF_NW - 1;        // the last wave: lanes 0-15 = the identity rows
            const int jj = lane;
            const int i = ident ? kb + jj : kb + 16 + tid;
            if (ident ? (lane < 16) : (i < ntile * 16)) {
                double sv[16], x[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) sv[c] = ident ? (c == jj ? 1.0 : 0.0) : S[i * LS + kb + c];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    x[k] = (kb + k < n) ? sv[k] * S[128 * LS + kb + k] : 0.0;
#pragma unroll
                    for (int c = k + 1; c < 16; ++c) sv[c] -= x[k] * S[(kb + c) * LS + kb + k];
                }
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (!ident || c > jj) S[i * LS + kb + c] = x[c];
            }
        }
        __syncthreads();
        tb += __builtin_amdgcn_s_memtime() - t_; t_ = __builtin_amdgcn_s_memtime();
        // (c) trailing update of the lower tiles on the matrix cores: C -= P P^T, K = 16
        const int t0 = kt + 1, nrem = ntile - t0;
        const int ntri = nrem * (nrem + 1) / 2;
        for (int t = wave; t < ntri; t += LEAF_NW) {
            int ti = 0, rem = t;                       // t -> (ti >= tj) in row-major triangle order
            while (rem > ti) { rem -= ti + 1; ++ti; }
            const int tj = rem;
            const int ri = (t0 + ti) * 16, rj = (t0 + tj) * 16;
            d4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = S[(ri + (lane >> 4) + 4 * r) * LS + rj + (lane & 15)];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const double pa = -S[(ri + (lane & 15)) * LS + kb + kc * 4 + (lane >> 4)];
                const double pb = S[(rj + (lane & 15)) * LS + kb + kc * 4 + (lane >> 4)];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(ri + (lane >> 4) + 4 * r) * LS + rj + (lane & 15)] = acc[r];
        }
        __syncthreads();
        tc += __builtin_amdgcn_s_memtime() - t_;
    }
    if (prof && threadIdx.x == 0) { prof[2] = ta; prof[3] = tb; prof[4] = tc; }
    LEAF_T(5);
    {   // write L
        const int i = tid & 127, j0 = tid >> 7;
        if (i < n)
            for (int j = j0; j <= i && j < n; j += LEAF_NT / 128) A[i + (size_t)j * lda] = S[i * LS + j];
    }
    // the last block row of the inverse (the earlier ones were done under the later steps' phase (a))
    LEAF_T(6);
    if (wave >= 1) inverse_row(ntile - 1);
    __syncthreads();
    LEAF_T(7);
    LEAF_T(8);
    for (int e = tid; e < 128 * 128; e += LEAF_NT) {
        int i = e & 127, j = e >> 7;
        double v = 0.0;
        if (i < n && j < n) {
            if (i == j) v = xd[j];
            else if (i > j) v = S[j * LS + i];
        }
        Linv[i + j * 128] = v;
    }
    LEAF_T(9);
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
    MCML_HIP(hipMemcpy(A.d(), h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_potrf_leaf), (int)(sizeof(double) * (128 * 129 + 128 + 7 * 256))));
    for (int rep = 0; rep < 3; ++rep) {
        MCML_HIP(hipMemcpy(A.d(), h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_potrf_leaf, dim3(1), dim3(LEAF_NT), sizeof(double) * (128 * 129 + 128 + 7 * 256), c.stream, A.d(), A.ld,
                           128, c.linv.d(), c.scalars.as<int>() + 32, prof.as<unsigned long long>());
        MCML_HIP(hipStreamSynchronize(c.stream));
    }
    MCML_HIP(hipMemcpy(host10, prof.p, 80, hipMemcpyDeviceToHost));
    return MCML_OK;
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
static constexpr size_t POTRF_LDS = sizeof(double) * (128 * 129 + 128 + 7 * 256);

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
    MCML_HIP(hipEventCreateWithFlags(&c.ev_col, hipEventDisableTiming));
    MCML_HIP(hipEventCreateWithFlags(&c.ev_leaf, hipEventDisableTiming));
    return MCML_OK;
}

static int potrf_blocked(Ctx& c, double* A, int lda, int n)
{
    // look-ahead pays while the rest of the trailing update is at least as long as a leaf
    static const int la_min = getenv("GLMMR_MCML_LA_MIN") ? atoi(getenv("GLMMR_MCML_LA_MIN")) : 1536;
    const bool la_any = chol_mode() == 1 && n - 2 * CHOL_NB >= la_min;
    if (la_any) MCML_TRY(lookahead_setup(c));
    int* errflag = c.scalars.as<int>() + 32;
    auto leaf = [&](hipStream_t s, int k, int nb) -> int {
        hipLaunchKernelGGL(k_potrf_leaf, dim3(1), dim3(LEAF_NT), POTRF_LDS, s, A + k + (size_t)k * lda, lda, nb,
                           c.linv.d() + (size_t)(k / CHOL_NB) * CHOL_NB * CHOL_NB, errflag, nullptr);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    };
    MCML_TRY(leaf(c.stream, 0, n < CHOL_NB ? n : CHOL_NB));
    for (int k = 0; k < n; k += CHOL_NB) {
        const int nb = (n - k < CHOL_NB) ? n - k : CHOL_NB;
        double* A11 = A + k + (size_t)k * lda;
        const double* Linv = c.linv.d() + (size_t)(k / CHOL_NB) * CHOL_NB * CHOL_NB;
        const int rem = n - k - nb;
        if (rem <= 0) break;
        double* A21 = A11 + nb;
        {
            EpiAxpby epi{A21, lda, 1.0, 0.0};
            MCML_TRY(chol_gemm<true>(c.stream, rem, nb, nb, A21, lda, Linv, CHOL_NB, epi, false, 1));
        }
        double* A22 = A11 + nb + (size_t)nb * lda;
        const int nb2 = rem < CHOL_NB ? rem : CHOL_NB;
        const bool la = la_any && rem - nb2 >= la_min;
        if (!la) {
            EpiAxpby epi{A22, lda, -1.0, 1.0};
            MCML_TRY(chol_gemm<true>(c.stream, rem, rem, nb, A21, lda, A21, lda, epi, true, 0));
            MCML_TRY(leaf(c.stream, k + nb, nb2));
            continue;
        }
        {   // the next panel's columns (its upper triangle inside the diagonal block is scratch)
            EpiAxpby epi{A22, lda, -1.0, 1.0};
            MCML_TRY(chol_gemm<true>(c.stream, rem, nb2, nb, A21, lda, A21, lda, epi, false, 0));
        }
        MCML_HIP(hipEventRecord(c.ev_col, c.stream));
        MCML_HIP(hipStreamWaitEvent(c.aux, c.ev_col, 0));
        MCML_TRY(leaf(c.aux, k + nb, nb2));
        MCML_HIP(hipEventRecord(c.ev_leaf, c.aux));
        const int rem2 = rem - nb2;
        if (rem2 > 0) {
            EpiAxpby epi{A22 + nb2 + (size_t)nb2 * lda, lda, -1.0, 1.0};
            MCML_TRY(chol_gemm<true>(c.stream, rem2, rem2, nb, A21 + nb2, lda, A21 + nb2, lda, epi, true, 0));
        }
        MCML_HIP(hipStreamWaitEvent(c.stream, c.ev_leaf, 0));
    }
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
    MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_potrf_leaf), (int)POTRF_LDS));
    if (chol_blocked()) return potrf_blocked(c, A, lda, n);
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
    MCML_HIP(hipMemcpyAsync(c.d_cov.p, cs.cov.data(), sizeof(int32_t) * cs.cov.size(), hipMemcpyHostToDevice, c.stream));
    MCML_TRY(c.d_data.ensure(sizeof(double) * (cs.data.size() + 1)));
    if (!cs.data.empty())
        MCML_HIP(hipMemcpyAsync(c.d_data.p, cs.data.data(), sizeof(double) * cs.data.size(), hipMemcpyHostToDevice, c.stream));
    MCML_TRY(c.d_blocks.ensure(sizeof(CovBlock) * cs.blocks.size()));
    MCML_HIP(hipMemcpyAsync(c.d_blocks.p, cs.blocks.data(), sizeof(CovBlock) * cs.blocks.size(), hipMemcpyHostToDevice, c.stream));
    std::vector<int> rowblock(cs.N, -1);
    c.maxdim_large = 0; c.n_small = 0; c.n_diag_rows = 0;
    for (int b = 0; b < cs.B; ++b) {
        const CovBlock& blk = cs.blocks[b];
        if (blk.all_gr) {
            for (int k = 0; k < blk.dim; ++k) rowblock[blk.matstart + k] = b;
            c.n_diag_rows += blk.dim;
        } else if (blk.dim <= SMALL_BLOCK) {
            ++c.n_small;
        } else if (blk.dim > c.maxdim_large) {
            c.maxdim_large = blk.dim;
        }
    }
    MCML_TRY(c.d_rowblock.ensure(sizeof(int) * (size_t)(cs.N + 1)));
    MCML_HIP(hipMemcpyAsync(c.d_rowblock.p, rowblock.data(), sizeof(int) * cs.N, hipMemcpyHostToDevice, c.stream));
    MCML_TRY(c.scalars.ensure(sizeof(double) * 64));
    MCML_HIP(hipMemsetAsync(c.scalars.p, 0, sizeof(double) * 64, c.stream));
    if (c.maxdim_large) {
        MCML_TRY(c.Dwork.alloc(c.maxdim_large, c.maxdim_large));
        MCML_HIP(hipMemsetAsync(c.Dwork.d(), 0, sizeof(double) * (size_t)c.Dwork.ld * c.maxdim_large, c.stream));
        MCML_TRY(c.linv.ensure(sizeof(double) * (size_t)(c.maxdim_large / CHOL_NB + 1) * CHOL_NB * CHOL_NB));
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
    MCML_HIP(hipMemcpyAsync(&flag, c.scalars.as<int>() + 32, sizeof(int), hipMemcpyDeviceToHost, c.stream));
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
    ThetaArg th;
    MCML_TRY(theta_arg(c, theta, th));
    const CovSpec& cs = c.cov;
    const int Q = cs.N, m = c.mcols;
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
        hipLaunchKernelGGL(k_diag_ll, grid, dim3(256), 0, c.stream, c.U.d(), c.U.ld, Q, m, drb, dd, dc, part);
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, part, (int)(grid.x * grid.y), 1.0, scal, 1);
        MCML_HIP(hipGetLastError());
    }
    if (c.n_small > 0) {
        std::vector<int> ids;
        for (int b = 0; b < cs.B; ++b)
            if (!cs.blocks[b].all_gr && cs.blocks[b].dim <= SMALL_BLOCK) ids.push_back(b);
        int ny = (m + 63) / 64; if (ny > 16) ny = 16;
        // ids + partials live in one buffer: [ids | partials]
        size_t idbytes = round_up_sz(sizeof(int) * ids.size(), 16);
        DevBuf& wb = c.scratch;      // scratch that outlives the launch
        MCML_TRY(wb.ensure(idbytes + sizeof(double) * ids.size() * ny + 1024));
        MCML_HIP(hipMemcpyAsync(wb.p, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice, c.stream));
        double* part = reinterpret_cast<double*>(static_cast<char*>(wb.p) + idbytes);
        hipLaunchKernelGGL(k_small_ll, dim3((unsigned)ids.size(), ny), dim3(64), 0, c.stream, c.U.d(), c.U.ld, m,
                           wb.as<int>(), dblk, dcov, cs.rows, c.d_data.d(), th, part, c.scalars.as<int>() + 32);
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, part, (int)ids.size() * ny, 1.0, scal, 1);
        MCML_HIP(hipGetLastError());
        MCML_HIP(hipStreamSynchronize(c.stream));   // ids is a host temporary
    }
    if (c.maxdim_large > 0) {
        MCML_TRY(c.Uwork.alloc(c.maxdim_large, m));
        for (int b = 0; b < cs.B; ++b) {
            const CovBlock& blk = cs.blocks[b];
            if (blk.all_gr || blk.dim <= SMALL_BLOCK) continue;
            const int d = blk.dim;
            dim3 g((d + 15) / 16, (d + 15) / 16);
            hipLaunchKernelGGL(k_build_dense, g, dim3(256), 0, c.stream, c.Dwork.d(), c.Dwork.ld, b, dblk, dcov,
                               cs.rows, c.d_data.d(), th, 0);
            MCML_HIP(hipGetLastError());
            MCML_TRY(potrf_lower(c, c.Dwork.d(), d, c.Dwork.ld));
            int gy = m < 256 ? m : 256;
            hipLaunchKernelGGL(k_copy_block, dim3((d + 255) / 256, gy), dim3(256), 0, c.stream, c.Uwork.d(),
                               c.Uwork.ld, c.U.d() + blk.matstart, c.U.ld, d, m);
            MCML_HIP(hipGetLastError());
            MCML_TRY(trsm_left_lower(c, c.Dwork.d(), c.Dwork.ld, d, c.Uwork.d(), c.Uwork.ld, m));
            int gx = (d + 255) / 256; gy = m < 64 ? m : 64;
            MCML_TRY(c.partials.ensure(sizeof(double) * (size_t)(gx * gy + 16)));
            hipLaunchKernelGGL(k_sumsq, dim3(gx, gy), dim3(256), 0, c.stream, c.Uwork.d(), c.Uwork.ld, d, m, c.partials.d());
            hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, c.partials.d(), gx * gy, 1.0, scal + 2, 0);
            hipLaunchKernelGGL(k_logdet, dim3(1), dim3(256), 0, c.stream, c.Dwork.d(), c.Dwork.ld, d, scal + 1);
            hipLaunchKernelGGL(k_finish_large, dim3(1), dim3(1), 0, c.stream, scal, d, m);
            MCML_HIP(hipGetLastError());
        }
    }
    MCML_HIP(hipMemcpyAsync(sum_out, scal, sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MCML_TRY(check_errflag(c, "mvn_ll"));
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
        dim3 g((d + 15) / 16, (d + 15) / 16);
        hipLaunchKernelGGL(k_build_dense, g, dim3(256), 0, c.stream, A, c.L.ld, b, dblk, dcov, cs.rows,
                           c.d_data.d(), th, chol ? 0 : 1);
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
    c.have_L = chol;
    return MCML_OK;
}

}  // namespace mcml
