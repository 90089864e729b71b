// dgemm_dl.h -- short-K GEMM for the Cholesky / triangular-solve updates.
//
// The panel updates of potrf_blocked / trsm_left_blocked (mvn.hip) are GEMMs with K = 128:
// eight K steps of 16.  With register staging (dgemm_mfma.h) every step waits for a global
// load issued only one step earlier, so a launch costs ~8 memory latencies (20-40 us) for a
// few microseconds of MFMA work.  This kernel stages global -> LDS with LDS-DMA
// (`global_load_lds_dwordx4`) into a deep ring (4-5 stages: three or four K steps in flight
// behind counted `s_waitcnt vmcnt` + raw `s_barrier`), so the whole K = 128 panel streams in
// about two latencies.
//
// Operands as in dgemm_mfma.h: A column-major (M contiguous); B K-major (B[k + n*ldb]) or
// N-major (B[n + k*ldb]).  LDS images (linear in the order LDS-DMA fills them; swizzles live
// in the per-lane SOURCE address):
//   M-major operand  [k][BM doubles]; when BM % 32 == 0 the 16-double blocks of odd rows k
//                    are swapped pairwise so rows k, k+1 (the two 16-lane halves of a
//                    ds_read_b64) hit different bank halves;  BM % 32 == 16 needs no swizzle.
//   K-major B        [k/2][n][2 doubles]
// Contract (checked on the host, dl_applicable): K % 16 == 0, operands 16-byte aligned with
// even leading dimensions.  Rows of A beyond M / columns of B beyond N are read from a clamped
// valid address and only feed outputs that are never stored.
// `lower_only`: workgroups whose tile lies strictly above the diagonal exit at once.
// In-place use (C aliases A with N <= BN, or C aliases B with M <= BM) is safe: a workgroup
// has consumed its whole K range before the epilogue stores, and no other workgroup reads
// the rows (columns) it overwrites.
#pragma once
#include "dgemm_dlds.h"

namespace mcml {

template <int WTM, int WTN, int WM, int WN, bool BNMAJOR, int STAGES>
struct DlgCfg {
    static constexpr int BK = 16;
    static constexpr int NW = WM * WN, THREADS = 64 * NW;
    static constexpr int BM = 16 * WTM * WM, BN = 16 * WTN * WN;
    static constexpr int A_BYTES = BK * BM * 8, B_BYTES = BK * BN * 8;
    static constexpr int A_CHUNKS = A_BYTES / 1024, B_CHUNKS = B_BYTES / 1024;   // 1 KiB per wave LDS-DMA
    static constexpr int CHUNKS = A_CHUNKS + B_CHUNKS;
    static constexpr int PER = (CHUNKS + NW - 1) / NW;        // pieces per wave per K step
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr size_t LDS_BYTES = (size_t)STAGES * STAGE_BYTES;
    static constexpr bool SWZ_A = (BM % 32) == 0;
    static constexpr bool SWZ_B = BNMAJOR && (BN % 32) == 0;
    static_assert(BM % 8 == 0 && BN % 8 == 0, "tile sides are multiples of 8 doubles");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS ring does not fit");
    static_assert(PER * (STAGES - 1) <= 24, "vmcnt immediates in dl_wait_vm");
};

// s_waitcnt vmcnt(N) with a literal immediate (N <= 24)
template <int N>
__device__ __forceinline__ void dl_wait_vm()
{
#define MCML_VM(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    if constexpr (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MCML_VM(1) MCML_VM(2) MCML_VM(3) MCML_VM(4) MCML_VM(5) MCML_VM(6) MCML_VM(7) MCML_VM(8)
    MCML_VM(9) MCML_VM(10) MCML_VM(11) MCML_VM(12) MCML_VM(13) MCML_VM(14) MCML_VM(15) MCML_VM(16)
    MCML_VM(17) MCML_VM(18) MCML_VM(19) MCML_VM(20) MCML_VM(21) MCML_VM(22) MCML_VM(23) MCML_VM(24)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef MCML_VM
}

// leave at most `tiles` K steps (PER pieces each) in flight
template <int PER, int MAXT>
__device__ __forceinline__ void dl_wait_leave(int tiles)
{
    if constexpr (MAXT <= 0) dl_wait_vm<0>();
    else {
        if (tiles >= MAXT) dl_wait_vm<PER * MAXT>();
        else dl_wait_leave<PER, MAXT - 1>(tiles);
    }
}

template <int WTM, int WTN, int WM, int WN, bool BNMAJOR, int STAGES, class Epi>
__global__ __launch_bounds__(64 * WM * WN) void dgemm_dl_kernel(GemmP p, Epi epi)
{
    using Cfg = DlgCfg<WTM, WTN, WM, WN, BNMAJOR, STAGES>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NW = Cfg::NW, PER = Cfg::PER;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave - wr * WN;

    const int nblk = gridDim.x;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int bi, bj;
    if (p.lower_only) {
        // only the tiles that touch the lower triangle were launched (dl_lower_tiles): row band bi holds
        // its first min(gn, last column tile under the diagonal + 1) column tiles; a short scalar walk finds
        // (bi, bj).  Every XCD gets the same number of working tiles (the plain map gave the XCD of the
        // bottom rows twice the average).
        int rem = nid;
        for (bi = 0; bi < p.gm; ++bi) {
            int nc = (bi * BM + BM - 1 + p.diag_shift) / BN + 1;
            if (nc > p.gn) nc = p.gn;
            if (rem < nc) break;
            rem -= nc;
        }
        bj = rem;
    } else {
        bi = nid / p.gn; bj = nid - bi * p.gn;
    }
    const int m0 = bi * BM, n0 = bj * BN;

    // ---- this wave's LDS-DMA pieces: a uniform base per operand (advanced per K step by scalar adds) plus a
    // per-lane 32-bit byte offset that never changes (lds_dma16, dgemm_dlds.h); piece s of a wave belongs to A
    // or to B depending on the wave only, so the choice of base is scalar
    unsigned voff[PER]; int loff[PER]; bool isA[PER];
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        int c = wave + NW * s;
        if (c >= Cfg::CHUNKS) c = wave;                    // repeat a piece: uniform vmcnt bookkeeping
        isA[s] = c < Cfg::A_CHUNKS;
        if (c < Cfg::A_CHUNKS) {
            const int o = c * 1024 + lane * 16;
            const int k = o / (BM * 8), pos = (o - k * BM * 8) >> 3;
            const int m = Cfg::SWZ_A ? ((((pos >> 4) ^ (k & 1)) << 4) | (pos & 15)) : pos;
            int gm = m0 + m;
            if (gm >= p.M) gm = 0;
            voff[s] = (unsigned)((gm + (size_t)k * p.lda) * 8);
            loff[s] = c * 1024;
        } else {
            const int cb = c - Cfg::A_CHUNKS;
            const int o = cb * 1024 + lane * 16;
            if (BNMAJOR) {
                const int k = o / (BN * 8), pos = (o - k * BN * 8) >> 3;
                const int n = Cfg::SWZ_B ? ((((pos >> 4) ^ (k & 1)) << 4) | (pos & 15)) : pos;
                int gn = n0 + n;
                if (gn >= p.N) gn = 0;
                voff[s] = (unsigned)((gn + (size_t)k * p.ldb) * 8);
            } else {
                const int q = o >> 4;                       // 16-byte slot: (kp, n)
                const int kp = q / BN, n = q - kp * BN;
                int gn = n0 + n;
                if (gn >= p.N) gn = 0;
                voff[s] = (unsigned)((2 * kp + (size_t)gn * p.ldb) * 8);
            }
            loff[s] = Cfg::A_BYTES + cb * 1024;
        }
    }
    const size_t stepA = (size_t)BK * p.lda * 8, stepB = BNMAJOR ? (size_t)BK * p.ldb * 8 : (size_t)BK * 8;
    // blockIdx.y: one of several independent problems of the same shape (the candidate thetas of a theta-step round
    // factorised side by side, mvn.hip)
    const int by = blockIdx.y;
    const char* sA = reinterpret_cast<const char*>(p.A + (size_t)by * p.bsA);
    const char* sB = reinterpret_cast<const char*>(p.B + (size_t)by * p.bsB);

    auto issue = [&](int stage) {
        const unsigned lbase = (unsigned)(size_t)(lds_ptr_t)lds + stage * Cfg::STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < PER; ++s) lds_dma16(lbase + loff[s], voff[s], isA[s] ? sA : sB);
        sA += stepA;
        sB += stepB;
    };

    d4 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int l15 = lane & 15, lk = lane >> 4;
    auto compute = [&](int stage) {
        const double* as = reinterpret_cast<const double*>(lds + stage * Cfg::STAGE_BYTES);
        const double* bs = reinterpret_cast<const double*>(lds + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int kk = 4 * ks + lk;
            double a[WTM], b[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                const int blk = wr * WTM + i;
                a[i] = as[kk * BM + ((Cfg::SWZ_A ? (blk ^ (kk & 1)) : blk) << 4) + l15];
            }
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int blk = wc * WTN + j;
                if (BNMAJOR) b[j] = bs[kk * BN + ((Cfg::SWZ_B ? (blk ^ (kk & 1)) : blk) << 4) + l15];
                else b[j] = bs[((kk >> 1) * BN + (blk << 4) + l15) * 2 + (kk & 1)];
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = p.K / BK;
    int issued = 0;
    for (; issued < STAGES - 1 && issued < nk; ++issued) issue(issued);
    dl_wait_leave<PER, STAGES - 2>(issued - 1);            // K step 0 landed
    __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (issued < nk) {                                 // its stage was last read in step kt-1
            int sn = st + STAGES - 1; if (sn >= STAGES) sn -= STAGES;
            issue(sn);
            ++issued;
        }
        compute(st);
        dl_wait_leave<PER, STAGES - 2>(issued - kt - 2);   // step kt+1 landed, later ones stay in flight
        __builtin_amdgcn_s_barrier();
        st = st + 1; if (st >= STAGES) st = 0;
    }

    Epi e = epi;
    e.shift(by);
    e(acc, m0 + wr * 16 * WTM, n0 + wc * 16 * WTN, lane, p.M, p.N, bi * WM + wr);
}

static inline bool dl_applicable(int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                                 bool bnmajor)
{
    return M >= 1 && N >= 1 && K >= 16 && (K & 15) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
           (lda & 1) == 0 && (ldb & 1) == 0 && lda >= M && (bnmajor ? ldb >= N : ldb >= K) &&
           dma_offsets_fit((size_t)lda * 16 + M, bnmajor ? (size_t)ldb * 16 + N : (size_t)ldb * N + 16);
}

template <int WTM, int WTN, int WM, int WN, bool BNMAJOR, int STAGES, class Epi>
static inline int launch_gemm_dl_cfg(hipStream_t s, GemmP p, const Epi& epi, int nbatch = 1)
{
    using Cfg = DlgCfg<WTM, WTN, WM, WN, BNMAJOR, STAGES>;
    p.gm = (p.M + Cfg::BM - 1) / Cfg::BM;
    p.gn = (p.N + Cfg::BN - 1) / Cfg::BN;
    long ntiles = (long)p.gm * p.gn;
    if (p.lower_only) {
        ntiles = 0;
        for (int bi = 0; bi < p.gm; ++bi) {
            int nc = (bi * Cfg::BM + Cfg::BM - 1 + p.diag_shift) / Cfg::BN + 1;
            ntiles += nc > p.gn ? p.gn : nc;
        }
    }
    MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&dgemm_dl_kernel<WTM, WTN, WM, WN, BNMAJOR, STAGES, Epi>), (int)Cfg::LDS_BYTES));
    hipLaunchKernelGGL((dgemm_dl_kernel<WTM, WTN, WM, WN, BNMAJOR, STAGES, Epi>), dim3((unsigned)ntiles, (unsigned)nbatch),
                       dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, p, epi);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

// tile: 0 = pick;  1 = 128 x 128, 4-stage ring (one workgroup per CU);  2 = 64 x 128, 5 stages;
// 3 = 128 x 32 (the 128 x m diagonal-block products of the blocked TRSM: m / 32 workgroups);
// 4 = 128 x 128 with a 2-stage ring (64 KB) and 5 = 64 x 128 with 3 stages (72 KB): TWO workgroups per
// CU, so one's prologue / epilogue overlaps the other's MFMA work and the tile count quantises over 512
// slots -- measured best for every K = 128 update of the Q = 5000 factorisation;  6 = 64 x 64, 4 waves,
// 48 KB (three per CU);  7 = 16 x 128 and 8 = 32 x 32, 4 waves: the single-block products on the
// factorisation's critical path (one 128 x 128 x 128 product spread over 8 / 16 CUs).
// inplace: 0 none; 1 = C aliases A (needs BN >= N); 2 = C aliases B (needs BM >= M).
template <bool BNMAJOR, class Epi>
static inline int launch_gemm_dl(hipStream_t s, int M, int N, int K, const double* A, int lda,
                                 const double* B, int ldb, const Epi& epi, bool lower_only = false, int tile = 0,
                                 int inplace = 0, int diag_shift = 0, int nbatch = 1, size_t bsA = 0, size_t bsB = 0)
{
    MCML_REQUIRE(dl_applicable(M, N, K, A, lda, B, ldb, BNMAJOR), "dgemm_dl: shape/alignment contract violated "
                 "(M %d N %d K %d lda %d ldb %d)", M, N, K, lda, ldb);
    GemmP p{M, N, K, A, lda, B, ldb, 0, 0, lower_only ? 1 : 0, 0, diag_shift, bsA, bsB};
    if (tile == 0) {
        long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        if (lower_only) t128 = t128 / 2 + (M + 127) / 128;
        static const int big = getenv("GLMMR_MCML_DL_BIG") ? atoi(getenv("GLMMR_MCML_DL_BIG")) : 4;
        static const int small = getenv("GLMMR_MCML_DL_SMALL") ? atoi(getenv("GLMMR_MCML_DL_SMALL")) : 5;
        static const int thr = getenv("GLMMR_MCML_DL_THR") ? atoi(getenv("GLMMR_MCML_DL_THR")) : 768;
        static const int thr64 = getenv("GLMMR_MCML_DL_THR64") ? atoi(getenv("GLMMR_MCML_DL_THR64")) : 128;
        tile = t128 >= thr ? big : small;
        if (inplace == 0 && t128 < thr64) tile = 6;
        if (inplace == 2) tile = N >= 256 ? 3 : 1;
    }
    MCML_REQUIRE(!(inplace == 1 && (tile == 6 || tile == 8) && N > (tile == 6 ? 64 : 32)) &&
                 !(inplace == 2 && (tile == 2 || tile == 5 || tile == 6 || tile == 7 || tile == 8) && M > (tile == 7 ? 16 : tile == 8 ? 32 : 64)),
                 "dgemm_dl: tile %d cannot run this product in place", tile);
    if (tile == 1) return launch_gemm_dl_cfg<4, 2, 2, 4, BNMAJOR, 4, Epi>(s, p, epi, nbatch);
    if (tile == 3) return launch_gemm_dl_cfg<1, 2, 8, 1, BNMAJOR, 6, Epi>(s, p, epi, nbatch);
    if (tile == 4) return launch_gemm_dl_cfg<4, 2, 2, 4, BNMAJOR, 2, Epi>(s, p, epi, nbatch);
    if (tile == 5) return launch_gemm_dl_cfg<2, 2, 2, 4, BNMAJOR, 3, Epi>(s, p, epi, nbatch);
    if (tile == 6) return launch_gemm_dl_cfg<2, 2, 2, 2, BNMAJOR, 3, Epi>(s, p, epi, nbatch);
    if (tile == 7) return launch_gemm_dl_cfg<1, 2, 1, 4, BNMAJOR, 4, Epi>(s, p, epi, nbatch);    // 16 x 128, 4 waves
    if (tile == 8) return launch_gemm_dl_cfg<1, 1, 2, 2, BNMAJOR, 4, Epi>(s, p, epi, nbatch);    // 32 x 32, 4 waves
    return launch_gemm_dl_cfg<2, 2, 2, 4, BNMAJOR, 5, Epi>(s, p, epi, nbatch);
}

}  // namespace mcml
