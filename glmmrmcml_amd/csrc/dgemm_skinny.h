// dgemm_skinny.h -- the sampler's two products when at most 4 chains take part: chains = 1 (the reference's one
// sequential chain, mhmcmc.h:121-157) and the last chains of a deep No-U-Turn doubling (nuts.h).  The MFMA kernels
// work on 128-column tiles: one column costs what 128 do (66-76 us per product at n = Q = 5000).  With so few columns
// the product is a matrix-vector stream bound by HBM -- every element of A is used at most 4 times -- so:
//   stage 1  k_skinny_partial: a workgroup = 256 consecutive rows (two per thread: 16-byte loads of A, M-contiguous)
//            x one chunk of 128 K columns, the chunk of X staged in LDS ([k][n], broadcast reads); the structurally
//            zero K range of the row block (BandPlan::kr, the same ranges the banded GEMM skips) is never launched
//            into; partial sums part[chunk][n][row];
//   stage 2  k_skinny_finish<Epi>: one thread per row adds the chunks in K order (fixed order: bit-reproducible) and
//            applies the element form of the same epilogue functor the GEMMs use (EpiForwardT / EpiBackward::elem).
// Measured per product at n = Q = 5000 (scripts/time_fewchains.py): 1 column 28 us, 4 columns 43-46 us (MFMA path:
// 66-76 us); 16 columns in this form run 135 us -- LDS-read bound, 16 broadcast reads per 16-byte load of A -- which
// is why the path stops at 4 (SK_NUSE) and the MFMA tiles take over.
#pragma once
#include "band_plan.h"

namespace mcml {

constexpr int SK_ROWS = 256, SK_KC = 128, SK_NMAX = 16;       // SkinnyPlan: band_plan.h; SK_NMAX: columns the partial layout holds
constexpr int SK_NUSE = 4;                                    // columns up to which this path is used

template <int N>
__global__ __launch_bounds__(128) void k_skinny_partial(const double* A, int lda, int M, int K, const double* X, int ldx,
                                                        int ncols, const int* range, double* part, int ldp)
{
    __shared__ double xs[SK_KC][N];
    const int rb = blockIdx.x, kc = blockIdx.y;
    if (kc < range[2 * rb] || kc >= range[2 * rb + 1]) return;
    const int k0 = kc * SK_KC;
    for (int idx = threadIdx.x; idx < SK_KC * N; idx += 128) {
        const int k = idx / N, n = idx - k * N;
        xs[k][n] = (k0 + k < K && n < ncols) ? X[(k0 + k) + (size_t)n * ldx] : 0.0;
    }
    __syncthreads();
    const int r = rb * SK_ROWS + 2 * (int)threadIdx.x;
    if (r >= M) return;
    const bool two = r + 1 < M;
    double a0[N], a1[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { a0[n] = 0.0; a1[n] = 0.0; }
    const int kn = (K - k0 < SK_KC) ? K - k0 : SK_KC;
    const double* Ap = A + r + (size_t)k0 * lda;
    typedef double d2_ __attribute__((ext_vector_type(2)));
    int k = 0;
    for (; k + 8 <= kn; k += 8) {                                       // eight 16-byte loads in flight
        d2_ v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const d2_*>(Ap + (size_t)(k + u) * lda);   // row r is even, lda even
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < N; ++n) { const double x = xs[k + u][n]; a0[n] += v[u].x * x; a1[n] += v[u].y * x; }
    }
    for (; k < kn; ++k) {
        const d2_ v = *reinterpret_cast<const d2_*>(Ap + (size_t)k * lda);
#pragma unroll
        for (int n = 0; n < N; ++n) { const double x = xs[k][n]; a0[n] += v.x * x; a1[n] += v.y * x; }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double* P = part + ((size_t)kc * SK_NMAX + n) * ldp + r;
        P[0] = a0[n];
        if (two) P[1] = a1[n];
    }
}

template <class Epi>
__global__ __launch_bounds__(256) void k_skinny_finish(const double* part, int ldp, int M, int ncols, const int* range, Epi epi)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int rb = m / SK_ROWS;
    const int c0 = range[2 * rb], c1 = range[2 * rb + 1];
    for (int n = 0; n < ncols; ++n) {
        double acc = 0.0;
        int kc = c0;
        for (; kc + 4 <= c1; kc += 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = part[((size_t)(kc + u) * SK_NMAX + n) * ldp + m];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
        for (; kc < c1; ++kc) acc += part[((size_t)kc * SK_NMAX + n) * ldp + m];
        epi.elem(m, n, acc);
    }
}

// chunk range of every row block from the K-tile ranges of the 80-row bands
static int skinny_plan(const BandPlan& bp, hipStream_t s, SkinnyPlan& sp)
{
    const int M = bp.M, K = bp.K;
    sp.nrb = (M + SK_ROWS - 1) / SK_ROWS; sp.nkc = (K + SK_KC - 1) / SK_KC; sp.ldp = round_up(M, SK_ROWS);
    std::vector<int> rg(2 * (size_t)sp.nrb);
    for (int rb = 0; rb < sp.nrb; ++rb) {
        const int r0 = rb * SK_ROWS, r1 = std::min(M, r0 + SK_ROWS) - 1;
        int kmin = 1 << 30, kmax = 0;
        for (int b = r0 / BD_BM; b <= r1 / BD_BM && b < bp.nbands; ++b) {
            if (bp.kr[2 * b + 1] <= bp.kr[2 * b]) continue;            // an all-zero band
            kmin = std::min(kmin, bp.kr[2 * b] * BD_BK); kmax = std::max(kmax, bp.kr[2 * b + 1] * BD_BK);
        }
        if (kmax > K) kmax = K;
        if (kmax <= kmin) { rg[2 * rb] = 0; rg[2 * rb + 1] = 0; continue; }
        rg[2 * rb] = kmin / SK_KC; rg[2 * rb + 1] = (kmax + SK_KC - 1) / SK_KC;
    }
    MCML_TRY(sp.range.ensure(sizeof(int) * rg.size()));
    MCML_HIP(hipMemcpyAsync(sp.range.p, rg.data(), sizeof(int) * rg.size(), hipMemcpyHostToDevice, s));
    MCML_HIP(hipStreamSynchronize(s));                                 // rg is a local
    MCML_TRY(sp.part.ensure(sizeof(double) * (size_t)sp.nkc * SK_NMAX * sp.ldp));
    return MCML_OK;
}

// Y = A X + epilogue for ncols <= SK_NMAX columns; A is M x K column-major with an even leading dimension
template <class Epi>
static int launch_skinny(hipStream_t s, BandPlan& bp, int ncols, const double* A, int lda, const double* X, int ldx,
                         const Epi& epi)
{
    if (!bp.skinny) {
        bp.skinny.reset(new SkinnyPlan());
        int rc = skinny_plan(bp, s, *bp.skinny);
        if (rc != MCML_OK) { bp.skinny.reset(); return rc; }
    }
    SkinnyPlan& sp = *bp.skinny;
    const dim3 grid(sp.nrb, sp.nkc);
    if (ncols <= 1)
        hipLaunchKernelGGL((k_skinny_partial<1>), grid, dim3(128), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    else if (ncols <= 2)
        hipLaunchKernelGGL((k_skinny_partial<2>), grid, dim3(128), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    else if (ncols <= 4)
        hipLaunchKernelGGL((k_skinny_partial<4>), grid, dim3(128), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    else
        hipLaunchKernelGGL((k_skinny_partial<16>), grid, dim3(128), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    hipLaunchKernelGGL((k_skinny_finish<Epi>), dim3((bp.M + 255) / 256), dim3(256), 0, s, sp.part.d(), sp.ldp, bp.M, ncols,
                       sp.range.as<int>(), epi);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

inline bool skinny_applicable(const BandPlan& bp, int M, int K, int ncols, int lda)
{
    return ncols >= 1 && ncols <= SK_NUSE && bp.M == M && bp.K == K && !bp.kr.empty() && (lda & 1) == 0;
}

}  // namespace mcml
