// dgemm_skinny.h -- the sampler's two products when at most 16 chains take part: chains = 1 (the reference's one
// sequential chain, mhmcmc.h:121-157), small chain counts, and the last chains of a deep No-U-Turn doubling (nuts.h).
// The MFMA kernels work on 128-column tiles: one column costs what 128 do (66-69 us per product at n = Q = 5000).
// With so few columns the product is a matrix-vector stream bound by HBM -- every element of A is used at most 16
// times -- so:
//   stage 1  k_skinny_partial: a workgroup = 128 consecutive rows (two per lane: 16-byte loads of A, M-contiguous)
//            x one chunk of 128 K columns split over its four waves (fixed-order LDS combine), the chunk of X staged
//            in LDS ([k][n], broadcast reads); the structurally zero K range of the row block (BandPlan::kr, the same
//            ranges the banded GEMM skips) is never launched into; partial sums part[chunk][n][row];
//   stage 2  k_skinny_finish<Epi>: one thread per (row, column) adds the chunks in K order (fixed order:
//            bit-reproducible) and applies the element form of the same epilogue functor the GEMMs use
//            (EpiForwardT / EpiBackward::elem).
// Measured per product at n = Q = 5000 (scripts/time_fewchains.py, both stages and the timing marker): 1 column 27 us
// (stage 1 alone 18.5 us = 100 MB of triangular ZL at 5.4 TB/s), 4 columns 29 us, 8 columns 32-33 us, 9-16 columns
// 49-62 us (the 16-wide instantiation is LDS-read bound), MFMA path 66-69 us.
#pragma once
#include "band_plan.h"

namespace mcml {

constexpr int SK_ROWS = 256, SK_KC = 128, SK_NMAX = 16;       // SkinnyPlan: band_plan.h; SK_NMAX: columns the partial layout holds
constexpr int SK_NUSE = 16;                                   // columns up to which this path is used (= SK_NMAX)
constexpr int SK_WROWS = 128;                                 // rows of a workgroup of k_skinny_partial

// workgroup = 4 waves on the SAME 128 rows (two per lane): wave w takes K columns [32 w, 32 w + 32) of the chunk, the
// four sums are added in wave order through LDS (fixed order).  Four times the waves in flight of a 2-wave workgroup
// with all 128 K columns each: the kernel is a pure HBM stream and lives on memory-level parallelism.
template <int N>
__global__ __launch_bounds__(256) void k_skinny_partial(const double* A, int lda, int M, int K, const double* X, int ldx,
                                                        int ncols, const int* range, double* part, int ldp)
{
    __shared__ double xs[SK_KC][N];
    __shared__ double red[3][2 * N][64];
    const int rb = blockIdx.x, kc = blockIdx.y;
    // the chunk range table is per 256-row block (two workgroups share an entry)
    const int rblk = (rb * SK_WROWS) / SK_ROWS;
    if (kc < range[2 * rblk] || kc >= range[2 * rblk + 1]) return;
    const int k0 = kc * SK_KC;
    for (int idx = threadIdx.x; idx < SK_KC * N; idx += 256) {
        const int k = idx / N, n = idx - k * N;
        xs[k][n] = (k0 + k < K && n < ncols) ? X[(k0 + k) + (size_t)n * ldx] : 0.0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = rb * SK_WROWS + 2 * lane;
    const bool rin = r < M;
    const bool two = r + 1 < M;
    double a0[N], a1[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { a0[n] = 0.0; a1[n] = 0.0; }
    const int kq0 = w * (SK_KC / 4);
    int kn = K - k0 - kq0;                                            // K columns this wave still has
    if (kn > SK_KC / 4) kn = SK_KC / 4;
    typedef double d2_ __attribute__((ext_vector_type(2)));
    if (rin && kn > 0) {
        const double* Ap = A + r + (size_t)(k0 + kq0) * lda;
        int k = 0;
        for (; k + 8 <= kn; k += 8) {                                   // eight 16-byte loads in flight
            d2_ v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const d2_*>(Ap + (size_t)(k + u) * lda);   // row r is even, lda even
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int n = 0; n < N; ++n) { const double x = xs[kq0 + k + u][n]; a0[n] += v[u].x * x; a1[n] += v[u].y * x; }
        }
        for (; k < kn; ++k) {
            const d2_ v = *reinterpret_cast<const d2_*>(Ap + (size_t)k * lda);
#pragma unroll
            for (int n = 0; n < N; ++n) { const double x = xs[kq0 + k][n]; a0[n] += v.x * x; a1[n] += v.y * x; }
        }
    }
    if (w > 0) {
#pragma unroll
        for (int n = 0; n < N; ++n) { red[w - 1][2 * n][lane] = a0[n]; red[w - 1][2 * n + 1][lane] = a1[n]; }
    }
    __syncthreads();
    if (w != 0 || !rin) return;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const double s0 = ((a0[n] + red[0][2 * n][lane]) + red[1][2 * n][lane]) + red[2][2 * n][lane];
        const double s1 = ((a1[n] + red[0][2 * n + 1][lane]) + red[1][2 * n + 1][lane]) + red[2][2 * n + 1][lane];
        double* P = part + ((size_t)kc * SK_NMAX + n) * ldp + r;
        P[0] = s0;
        if (two) P[1] = s1;
    }
}

// one wave per (64 rows, column): the kernel is a handful of dependent-latency steps, so it wants many small workgroups
template <class Epi>
__global__ __launch_bounds__(64) void k_skinny_finish(const double* part, int ldp, int M, int ncols, const int* range, Epi epi)
{
    const int m = blockIdx.x * 64 + threadIdx.x, n = blockIdx.y;
    if (m >= M) return;
    const int rb = m / SK_ROWS;
    const int c0 = range[2 * rb], c1 = range[2 * rb + 1];
    double acc = 0.0;
    int kc = c0;
    for (; kc + 8 <= c1; kc += 8) {                                     // eight loads in flight, adds in K order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(kc + u) * SK_NMAX + n) * ldp + m];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; kc < c1; ++kc) acc += part[((size_t)kc * SK_NMAX + n) * ldp + m];
    epi.elem(m, n, acc);
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
    const dim3 grid((bp.M + SK_WROWS - 1) / SK_WROWS, sp.nkc);
    if (ncols <= 1)
        hipLaunchKernelGGL((k_skinny_partial<1>), grid, dim3(256), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    else if (ncols <= 2)
        hipLaunchKernelGGL((k_skinny_partial<2>), grid, dim3(256), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    else if (ncols <= 4)
        hipLaunchKernelGGL((k_skinny_partial<4>), grid, dim3(256), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    else if (ncols <= 8)
        hipLaunchKernelGGL((k_skinny_partial<8>), grid, dim3(256), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    else
        hipLaunchKernelGGL((k_skinny_partial<16>), grid, dim3(256), 0, s, A, lda, bp.M, bp.K, X, ldx, ncols, sp.range.as<int>(), sp.part.d(), sp.ldp);
    hipLaunchKernelGGL((k_skinny_finish<Epi>), dim3((bp.M + 63) / 64, ncols), dim3(64), 0, s, sp.part.d(), sp.ldp, bp.M, ncols,
                       sp.range.as<int>(), epi);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

inline bool skinny_applicable(const BandPlan& bp, int M, int K, int ncols, int lda)
{
    return ncols >= 1 && ncols <= SK_NUSE && bp.M == M && bp.K == K && !bp.kr.empty() && (lda & 1) == 0;
}

}  // namespace mcml
