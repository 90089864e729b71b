// dgemm_mfma.h -- the FP64 matrix-core GEMM every dense contraction of the MCML
// hot path runs on (HMC forward/backward products, Z*L, Z*u, L*V, the trailing
// updates of the Cholesky factorisation and of the triangular solves).
//
//   C(M x N) = epilogue( A(M x K) * B(K x N) )
//
// A is column-major (M contiguous).  B is either K-major (column-major K x N,
// as the Q x m sample matrices are) or N-major (B^T stored column-major, as the
// L21 panels of the Cholesky update are).  All operands f64.
//
// CDNA4 mapping (one 256-thread workgroup = 4 waves = one wave per SIMD):
//   * v_mfma_f64_16x16x4_f64, 64-lane waves, operands swapped so that the MFMA's
//     D tile is C^T: lane l, reg r holds C[m = l&15][n = (l>>4)+4r], i.e. each
//     16-lane group stores 128 contiguous bytes of a column of C.
//   * block tile (32*TM) x (32*TN), K step 16, wave tile (16*TM) x (16*TN);
//     TM=5,TN=4 gives 160 x 128 tiles = exactly 256 workgroups for the
//     5000 x 1024 products of the n=5000, m=1024 configuration.
//   * operands staged global -> VGPR (16-B loads) -> LDS, double buffered, one
//     barrier per K step; the next tile's global loads are issued before the
//     MFMAs of the current one.
//   * LDS rows padded so that the two 16-lane halves of a ds_read_b64 land in
//     disjoint bank halves (A rows: stride = 16 mod 32 doubles; K-major B rows:
//     18 doubles).
//   * blockIdx -> tile map is XCD-aware: each XCD (blockIdx % 8) owns a
//     contiguous band of row tiles, so the A band and the B panels it shares
//     stay in that XCD's L2.
#pragma once
#include "common.h"

namespace mcml {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int GEMM_BK = 16;

struct GemmP {
    int M, N, K;
    const double* A; int lda;   // A[i + k*lda]
    const double* B; int ldb;   // K-major: B[k + j*ldb];  N-major: B[j + k*ldb]
    int gm, gn;                 // tile grid
    int lower_only;             // skip tiles that lie strictly above the diagonal
    int dbg_nostep;             // timing experiment only: do not advance the operand pointers
    int diag_shift = 0;         // dgemm_dl lower_only: row m of C is row m + diag_shift of the square matrix
    size_t bsA = 0, bsB = 0;    // dgemm_dl batches (blockIdx.y): element strides of A and B from one problem to the next
};

// WTM x WTN: 16x16 MFMA tiles per wave; WM x WN: waves per workgroup.
template <int WTM, int WTN, int WM, int WN, bool BNMAJOR, int BK = 16>
struct GemmCfg {
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int BM = 16 * WTM * WM, BN = 16 * WTN * WN;
    static constexpr int SA = BM + 16;
    static constexpr int SB = BNMAJOR ? (BN + 16) : (BK + 2);
    static constexpr int A_TILE = BK * SA;
    static constexpr int B_TILE = BNMAJOR ? BK * SB : BN * SB;
    static constexpr size_t LDS_BYTES = sizeof(double) * 2 * (A_TILE + B_TILE);
    // 16-byte staging loads per thread (ceil)
    static constexpr int A_VEC = BK * BM / 2, B_VEC = BK * BN / 2;
    static constexpr int A_LD = (A_VEC + THREADS - 1) / THREADS;
    static constexpr int B_LD = (B_VEC + THREADS - 1) / THREADS;
};

// Epilogues receive the wave's accumulator fragment:
//   acc[i][j][r] = C[mB + 16 i + (lane & 15)][nB + 16 j + (lane >> 4) + 4 r]
#define MCML_EPI_FOREACH(TM_, TN_)                                   \
    _Pragma("unroll") for (int i = 0; i < TM_; ++i)                  \
    _Pragma("unroll") for (int j = 0; j < TN_; ++j)                  \
    _Pragma("unroll") for (int r = 0; r < 4; ++r)

struct EpiAxpby {   // C = alpha*A*B + beta*C
    double* C; int ldc; double alpha, beta;
    size_t bsC = 0;                                              // dgemm_dl batches: element stride of C
    __device__ __forceinline__ void shift(int b) { C += (size_t)b * bsC; }
    // With beta != 0 all reads of C are issued (from clamped, always valid addresses) before the first
    // store: a store to C followed by a load from C cannot be reordered by the compiler (may alias), so the
    // straightforward per-element read-modify-write is a chain of dependent memory round trips.
    template <int TM, int TN>
    __device__ __forceinline__ void operator()(d4 (&acc)[TM][TN], int mB, int nB, int lane,
                                               int M, int N, int /*rowslot*/) const {
        if (beta != 0.0) {
            double old[TM][TN][4];
            MCML_EPI_FOREACH(TM, TN) {
                int m = mB + 16 * i + (lane & 15), n = nB + 16 * j + (lane >> 4) + 4 * r;
                const bool ok = m < M && n < N;
                old[i][j][r] = C[(ok ? m : 0) + (size_t)(ok ? n : 0) * ldc];
            }
            MCML_EPI_FOREACH(TM, TN) {
                int m = mB + 16 * i + (lane & 15), n = nB + 16 * j + (lane >> 4) + 4 * r;
                if (m < M && n < N) {
                    double v = alpha * acc[i][j][r];
                    v += beta * old[i][j][r];
                    C[m + (size_t)n * ldc] = v;
                }
            }
        } else {
            MCML_EPI_FOREACH(TM, TN) {
                int m = mB + 16 * i + (lane & 15), n = nB + 16 * j + (lane >> 4) + 4 * r;
                if (m < M && n < N) C[m + (size_t)n * ldc] = alpha * acc[i][j][r];
            }
        }
    }
};

template <int WTM, int WTN, int WM, int WN, bool BNMAJOR, int BK, bool STAGGER, int INNER, class Epi>
__global__ __launch_bounds__(64 * WM * WN) void dgemm_mfma_kernel(GemmP p, Epi epi)
{
    using Cfg = GemmCfg<WTM, WTN, WM, WN, BNMAJOR, BK>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, SA = Cfg::SA, SB = Cfg::SB, NT = Cfg::THREADS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;
    double* Bs = smem + 2 * Cfg::A_TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave - wr * WN;

    // XCD-aware, bijective blockIdx -> tile map
    const int nblk = p.gm * p.gn;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int bi = nid / p.gn, bj = nid - bi * p.gn;
    if (p.lower_only && (bi + 1) * BM <= bj * BN) return;
    const int m0 = bi * BM, n0 = bj * BN;

    d2 ra[Cfg::A_LD], rb[Cfg::B_LD];
    // Loop-invariant staging state: one pointer per 16-byte load (advanced by a constant per K
    // step), the row/column validity of that load, and its k offset inside the tile.  Invalid
    // rows/columns point at row/column 0 (always readable) and are zeroed when written to LDS,
    // so the K loop carries no address arithmetic beyond pointer increments and no branches.
    const double* pa[Cfg::A_LD]; const double* pb[Cfg::B_LD];
    unsigned oka = 0, okb = 0;          // bit i: row / column of load i is inside the operand
    int kra[Cfg::A_LD], krb[Cfg::B_LD];
#pragma unroll
    for (int i = 0; i < Cfg::A_LD; ++i) {
        int idx = tid + i * NT;
        if (Cfg::A_VEC % NT != 0 && idx >= Cfg::A_VEC) idx = Cfg::A_VEC - 1;
        const int kr = idx / (BM / 2), mp = idx - kr * (BM / 2);
        const int m = m0 + 2 * mp;
        const bool okm = m < p.M;
        oka |= (okm ? 1u : 0u) << i; kra[i] = kr;
        pa[i] = p.A + (okm ? m : 0) + (size_t)kr * p.lda;
    }
#pragma unroll
    for (int j = 0; j < Cfg::B_LD; ++j) {
        int idx = tid + j * NT;
        if (Cfg::B_VEC % NT != 0 && idx >= Cfg::B_VEC) idx = Cfg::B_VEC - 1;
        if (BNMAJOR) {
            const int kr = idx / (BN / 2), np = idx - kr * (BN / 2);
            const int n = n0 + 2 * np;
            const bool okn = n < p.N;
            okb |= (okn ? 1u : 0u) << j; krb[j] = kr;
            pb[j] = p.B + (okn ? n : 0) + (size_t)kr * p.ldb;
        } else {
            const int ncol = idx / (BK / 2), kp = idx - ncol * (BK / 2);
            const int n = n0 + ncol;
            const bool okn = n < p.N;
            okb |= (okn ? 1u : 0u) << j; krb[j] = 2 * kp;
            pb[j] = p.B + 2 * kp + (size_t)(okn ? n : 0) * p.ldb;
        }
    }
    const size_t stepA = (p.dbg_nostep & 1) ? 0 : (size_t)BK * p.lda;
    const size_t stepB = (p.dbg_nostep & 1) ? 0 : (BNMAJOR ? (size_t)BK * p.ldb : (size_t)BK);
    unsigned kva = 0, kvb0 = 0, kvb1 = 0;   // bit i: load i of the tile in flight lies inside K

    // full = the whole tile lies inside K (every K step but possibly the last)
    auto load_tiles = [&](int k0, bool full) {
        if (full) {
#pragma unroll
            for (int i = 0; i < Cfg::A_LD; ++i) ra[i] = *reinterpret_cast<const d2*>(pa[i]);
#pragma unroll
            for (int j = 0; j < Cfg::B_LD; ++j) rb[j] = *reinterpret_cast<const d2*>(pb[j]);
            kva = kvb0 = kvb1 = ~0u;
        } else {
            kva = kvb0 = kvb1 = 0;
#pragma unroll
            for (int i = 0; i < Cfg::A_LD; ++i) {
                const bool v = (k0 + kra[i] < p.K);
                kva |= (v ? 1u : 0u) << i;
                ra[i] = v ? *reinterpret_cast<const d2*>(pa[i]) : d2{0.0, 0.0};
            }
#pragma unroll
            for (int j = 0; j < Cfg::B_LD; ++j) {
                const bool v0 = (k0 + krb[j] < p.K);
                const bool v1 = BNMAJOR ? v0 : (k0 + krb[j] + 1 < p.K);
                kvb0 |= (v0 ? 1u : 0u) << j; kvb1 |= (v1 ? 1u : 0u) << j;
                rb[j] = v0 ? *reinterpret_cast<const d2*>(pb[j]) : d2{0.0, 0.0};
            }
        }
#pragma unroll
        for (int i = 0; i < Cfg::A_LD; ++i) pa[i] += stepA;
#pragma unroll
        for (int j = 0; j < Cfg::B_LD; ++j) pb[j] += stepB;
    };
    auto store_tiles = [&](int buf) {
        double* as = As + buf * Cfg::A_TILE;
        double* bs = Bs + buf * Cfg::B_TILE;
#pragma unroll
        for (int i = 0; i < Cfg::A_LD; ++i) {
            int idx = tid + i * NT;
            if (Cfg::A_VEC % NT == 0 || idx < Cfg::A_VEC) {
                int kr = idx / (BM / 2), mp = idx - kr * (BM / 2);
                d2 v = ((oka & kva) >> i & 1u) ? ra[i] : d2{0.0, 0.0};
                *reinterpret_cast<d2*>(as + kr * SA + 2 * mp) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < Cfg::B_LD; ++j) {
            int idx = tid + j * NT;
            if (Cfg::B_VEC % NT == 0 || idx < Cfg::B_VEC) {
                d2 v;
                v[0] = ((okb & kvb0) >> j & 1u) ? rb[j][0] : 0.0;
                v[1] = ((okb & kvb1) >> j & 1u) ? rb[j][1] : 0.0;
                if (BNMAJOR) {
                    int kr = idx / (BN / 2), np = idx - kr * (BN / 2);
                    *reinterpret_cast<d2*>(bs + kr * SB + 2 * np) = v;
                } else {
                    int ncol = idx / (BK / 2), kp = idx - ncol * (BK / 2);
                    *reinterpret_cast<d2*>(bs + ncol * SB + 2 * kp) = v;
                }
            }
        }
    };

    d4 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = (p.K + BK - 1) / BK;
    const int l15 = lane & 15, lk = lane >> 4;
    auto compute = [&](int cur) {
        const double* as = As + cur * Cfg::A_TILE + wr * 16 * WTM + l15;
        const double* bs = BNMAJOR ? (Bs + cur * Cfg::B_TILE + wc * 16 * WTN + l15)
                                   : (Bs + cur * Cfg::B_TILE + (wc * 16 * WTN + l15) * SB);
        if constexpr (INNER == 2) {
            // all fragments of the K step first (BK/4 * (WTM + WTN) doubles), then the MFMAs
            double a[BK / 4][WTM], b[BK / 4][WTN];
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                const int kk = 4 * ks + lk;
#pragma unroll
                for (int i = 0; i < WTM; ++i) a[ks][i] = as[kk * SA + 16 * i];
#pragma unroll
                for (int j = 0; j < WTN; ++j) b[ks][j] = BNMAJOR ? bs[kk * SB + 16 * j] : bs[16 * j * SB + kk];
            }
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks)
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[ks][j], a[ks][i], acc[i][j], 0, 0, 0);
        } else {
            if constexpr (INNER == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                const int kk = 4 * ks + lk;
                double a[WTM], b[WTN];
#pragma unroll
                for (int i = 0; i < WTM; ++i) a[i] = as[kk * SA + 16 * i];
#pragma unroll
                for (int j = 0; j < WTN; ++j) b[j] = BNMAJOR ? bs[kk * SB + 16 * j] : bs[16 * j * SB + kk];
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[i][j], 0, 0, 0);
            }
            if constexpr (INNER == 1) __builtin_amdgcn_s_setprio(0);
        }
    };

    load_tiles(0, BK <= p.K);
    store_tiles(0);
    int cur = 0;
    if constexpr (!STAGGER) {
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1 < nk) && !(p.dbg_nostep & 2);   // bit 1: timing experiment, no staging
            if (more) load_tiles((kt + 1) * BK, (kt + 2) * BK <= p.K);
            __builtin_amdgcn_sched_barrier(0);   // loads are issued, not consumed, above here
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);   // the loads' first use stays below the MFMAs
            if (more) store_tiles(cur ^ 1);
            if (!(p.dbg_nostep & 4)) __syncthreads();               // bit 2: timing experiment, no barrier
            if (!(p.dbg_nostep & 2)) cur ^= 1;
        }
    } else {
        // The two waves of a SIMD (w and w + 4: a workgroup's waves are dealt to SIMDs cyclically)
        // run half a K step apart: the "early" wave writes the next tile to LDS and re-issues its
        // loads BEFORE its MFMAs, the "late" wave after them.  One wave alone can keep the FP64
        // matrix pipe busy (64 cycles per MFMA), so while one partner executes its staging
        // instructions the other's MFMAs fill the pipe instead of both idling at once.
        const bool early = __builtin_amdgcn_readfirstlane(wave) < (WM * WN) / 2;
        if (early && nk > 1) load_tiles(BK, 2 * BK <= p.K);          // tile 1, consumed in step 0
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1 < nk);
            if (early) {
                if (more) store_tiles(cur ^ 1);                        // tile kt + 1 (in registers)
                if (kt + 2 < nk) load_tiles((kt + 2) * BK, (kt + 3) * BK <= p.K);
            } else {
                if (more) load_tiles((kt + 1) * BK, (kt + 2) * BK <= p.K);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            if (!early && more) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    epi(acc, m0 + wr * 16 * WTM, n0 + wc * 16 * WTN, lane, p.M, p.N, bi * WM + wr);
}

// Tile variants: id -> (block tile, waves)
//   0: 160 x 128, 8 waves (5x2 MFMA tiles per wave)   1: 128 x 128, 8 waves (4x2)
//   2: 128 x  64, 4 waves (4x2)                        3:  64 x  64, 4 waves (2x2)
//   4: 160 x 128, 16 waves (5x1)                       5: 128 x 128, 16 waves (4x1)
//   6/7: 160 x 128 with K step 32 (8 / 16 waves)      8-10: staggered SIMD partners
//   11/12: 160 x 128, 4 waves (5x4 per wave)
// v_mfma_f64_16x16x4_f64 issues every 64 cycles from one wave (scripts/mfma_peak.hip: 77.4
// TFLOP/s = 98.5 % of the 78.6 vendor peak with one wave per SIMD).
struct TileChoice { int id; };
static const struct { int bm, bn; double eff; } kTileTab[16] = {
    {160, 128, 1.00}, {128, 128, 0.97}, {128, 64, 0.90}, {64, 64, 0.80},
    {160, 128, 0.50}, {128, 128, 0.50},    // 4, 5: the same tiles with 16 waves (not picked by default)
    {160, 128, 0.50}, {160, 128, 0.50},    // 6, 7: K step 32, 8 / 16 waves
    {160, 128, 0.50}, {160, 128, 0.50}, {128, 128, 0.50},    // 8, 9, 10: staggered SIMD partners
    {160, 128, 0.50}, {160, 128, 0.50},    // 11, 12: 4 waves (one per SIMD), 5x4 tiles per wave
    {160, 128, 0.50}, {160, 128, 0.50}, {160, 128, 0.50}};   // 13-15: inner-loop variants (setprio, preload)

// Smallest estimated time: workgroups are dealt 256 at a time (one per CU);
// smaller tiles pay more staging per flop.  Only ids 0-3 and 6 are candidates; the others
// are kept for A/B measurements (scripts/one_gemm.py): 16 waves, staggered SIMD partners
// and one-wave-per-SIMD variants all measured slower on MI355X ("GEMM log").
static inline TileChoice pick_tile(int M, int N, int K = 1 << 30)
{
    TileChoice best{3};
    double bestc = 1e300;
    for (int id = 0; id < 4; ++id) {
        long gm = (M + kTileTab[id].bm - 1) / kTileTab[id].bm, gn = (N + kTileTab[id].bn - 1) / kTileTab[id].bn;
        long rounds = (gm * gn + 255) / 256;
        double cost = (double)rounds * kTileTab[id].bm * kTileTab[id].bn / kTileTab[id].eff;
        if (cost < bestc) { bestc = cost; best = TileChoice{id}; }
    }
    // the 160 x 128 tile with K step 32 halves the per-step staging bubbles (+5 % measured) once
    // K is long enough to amortise its larger prologue
    if (best.id == 0 && K >= 1024) best.id = 6;
    return best;
}

template <int WTM, int WTN, int WM, int WN, bool BNMAJOR, int BK, bool STAGGER, int INNER, class Epi>
static inline int launch_gemm_tile(hipStream_t s, GemmP p, const Epi& epi)
{
    using Cfg = GemmCfg<WTM, WTN, WM, WN, BNMAJOR, BK>;
    p.gm = (p.M + Cfg::BM - 1) / Cfg::BM;
    p.gn = (p.N + Cfg::BN - 1) / Cfg::BN;
    MCML_TRY(ensure_dynamic_lds(
        reinterpret_cast<const void*>(&dgemm_mfma_kernel<WTM, WTN, WM, WN, BNMAJOR, BK, STAGGER, INNER, Epi>),
        (int)Cfg::LDS_BYTES));
    hipLaunchKernelGGL((dgemm_mfma_kernel<WTM, WTN, WM, WN, BNMAJOR, BK, STAGGER, INNER, Epi>), dim3(p.gm * p.gn),
                       dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, p, epi);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

// number of row slots an epilogue that keeps per-(row slot, column) partials needs
static inline int gemm_row_slots(int M, int tile_id)
{
    int bm = kTileTab[tile_id].bm;
    return ((M + bm - 1) / bm) * 2;   // every variant has WM = 2
}

// Host-side shape contract of the kernel (checked before every launch: a
// faulting kernel can take the whole node down).
static inline int check_gemm_args(const GemmP& p)
{
    MCML_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "dgemm: empty shape %dx%dx%d", p.M, p.N, p.K);
    MCML_REQUIRE(p.A && p.B, "dgemm: null operand");
    MCML_REQUIRE((p.lda & 1) == 0 && (p.ldb & 1) == 0, "dgemm: odd leading dimension");
    MCML_REQUIRE(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0,
                 "dgemm: operand not 16-byte aligned");
    MCML_REQUIRE(p.lda >= p.M, "dgemm: lda %d < M %d", p.lda, p.M);
    return MCML_OK;
}

template <bool BNMAJOR, class Epi>
static inline int launch_gemm(hipStream_t s, int M, int N, int K, const double* A, int lda,
                              const double* B, int ldb, const Epi& epi, bool lower_only = false,
                              int force_tile = -1)
{
    static const int dbg_nostep = getenv("GLMMR_MCML_DBG_NOSTEP") ? atoi(getenv("GLMMR_MCML_DBG_NOSTEP")) : 0;
    GemmP p{M, N, K, A, lda, B, ldb, 0, 0, lower_only ? 1 : 0, dbg_nostep};
    MCML_TRY(check_gemm_args(p));
    MCML_REQUIRE(BNMAJOR ? ldb >= N : ldb >= K, "dgemm: ldb %d too small", ldb);
    int id = force_tile >= 0 ? force_tile : pick_tile(M, N, K).id;
    switch (id) {
    case 0: return launch_gemm_tile<5, 2, 2, 4, BNMAJOR, 16, false, 0, Epi>(s, p, epi);
    case 1: return launch_gemm_tile<4, 2, 2, 4, BNMAJOR, 16, false, 0, Epi>(s, p, epi);
    case 2: return launch_gemm_tile<4, 2, 2, 2, BNMAJOR, 16, false, 0, Epi>(s, p, epi);
    case 4: return launch_gemm_tile<5, 1, 2, 8, BNMAJOR, 16, false, 0, Epi>(s, p, epi);
    case 5: return launch_gemm_tile<4, 1, 2, 8, BNMAJOR, 16, false, 0, Epi>(s, p, epi);
    case 6: return launch_gemm_tile<5, 2, 2, 4, BNMAJOR, 32, false, 0, Epi>(s, p, epi);
    case 7: return launch_gemm_tile<5, 1, 2, 8, BNMAJOR, 32, false, 0, Epi>(s, p, epi);
    case 8: return launch_gemm_tile<5, 2, 2, 4, BNMAJOR, 16, true, 0, Epi>(s, p, epi);
    case 9: return launch_gemm_tile<5, 2, 2, 4, BNMAJOR, 32, true, 0, Epi>(s, p, epi);
    case 10: return launch_gemm_tile<4, 2, 2, 4, BNMAJOR, 16, true, 0, Epi>(s, p, epi);
    case 11: return launch_gemm_tile<5, 4, 2, 2, BNMAJOR, 16, false, 0, Epi>(s, p, epi);
    case 12: return launch_gemm_tile<5, 4, 2, 2, BNMAJOR, 32, false, 0, Epi>(s, p, epi);
    case 13: return launch_gemm_tile<5, 2, 2, 4, BNMAJOR, 32, false, 1, Epi>(s, p, epi);
    case 14: return launch_gemm_tile<5, 2, 2, 4, BNMAJOR, 16, false, 2, Epi>(s, p, epi);
    case 15: return launch_gemm_tile<5, 2, 2, 4, BNMAJOR, 16, false, 1, Epi>(s, p, epi);
    default: return launch_gemm_tile<2, 2, 2, 2, BNMAJOR, 16, false, 0, Epi>(s, p, epi);
    }
}

}  // namespace mcml
