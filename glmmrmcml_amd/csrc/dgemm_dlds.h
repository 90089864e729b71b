// dgemm_dlds.h -- the HMC products' GEMM with direct-to-LDS staging.
//
// Same contraction, tile and epilogue interface as dgemm_mfma_kernel (160 x 128 tile, 8
// waves = 2 per SIMD, 5 x 2 v_mfma_f64_16x16x4 tiles per wave, K step 16), but the operands
// go global -> LDS with `global_load_lds_dwordx4` (LDS-DMA): no VGPR staging, no ds_write
// pass, no zero-selects, and a 3-stage LDS ring keeps TWO tiles in flight behind a counted
// `s_waitcnt vmcnt` and a raw `s_barrier` (a __syncthreads() would drain them).
//
// Measured motivation: with register staging the kernel runs at ~50 TF while
// the same loop without the staging instructions runs at 61.5 TF.
//
// LDS-DMA writes wave-uniform base + lane*16 contiguously (1 KiB per wave instruction), so
// the LDS images are linear in the order the lanes fill them and any swizzle lives in the
// per-lane SOURCE address:
//   A tile  [k][160 doubles], 16-double blocks of odd rows k swapped pairwise (XOR 1 on the
//           block index): the two 16-lane halves of a ds_read_b64 (rows k, k+1) then hit
//           different 128-B halves of the bank row -> conflict-free.
//   B tile  [k/2][n][2 doubles] (B is K-major: a 16-byte piece = rows k,k+1 of one column):
//           lanes n..n+15 read consecutive 16-byte slots -> conflict-free.
// Contract (checked on the host): K is a multiple of 16 AFTER padding, i.e. A holds zero
// columns and B zero rows up to round_up(K,16); rows of A beyond M and columns of B beyond N
// are read from a clamped (valid) address and only feed outputs that are never stored.
#pragma once
#include "dgemm_mfma.h"

namespace mcml {

constexpr int DL_BM = 160, DL_BN = 128;
// (K step, ring stages): (16, 3) keeps two tiles in flight in 108 KB; (32, 2) halves the
// barriers with one tile in flight in 144 KB
template <int BK, int STAGES>
struct DlCfg {
    static constexpr int A_BYTES = BK * DL_BM * 8;         // BK=16: 20 chunks of 1 KiB; 32: 40
    static constexpr int B_BYTES = BK * DL_BN * 8;         // BK=16: 16 chunks; 32: 32
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr size_t LDS_BYTES = (size_t)STAGES * STAGE_BYTES;
    static constexpr int A_CHUNKS = A_BYTES / 1024, B_CHUNKS = B_BYTES / 1024;
    static constexpr int NA = (A_CHUNKS + 7) / 8, NB = B_CHUNKS / 8;   // LDS-DMA pieces per wave per tile
    static constexpr int PER_TILE = NA + NB;
};
constexpr int DL_BK = 16;     // K padding granule the callers provide (a multiple of it serves both)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One LDS-DMA piece (1 KiB per wave: lane l's 16 bytes land at lds_addr + 16 l) in the SCALAR-BASE form:
// address = sbase (uniform, advanced per K step by scalar adds) + voff (per-lane 32-bit byte offset that
// never changes inside a K loop).  Hand-written: hipcc materialises base + offset into a 64-bit VGPR
// temporary per load, and with per-lane 64-bit pointers advanced by VALU adds it may place an add right
// behind the load that reads the same register -- that write-after-read stalls the wave until the load
// has left the queue (measured on the banded kernel: 458 vs 419 us per launch).  The s_nop is the wait
// state an LDS-DMA needs behind a SALU write of M0 (hipcc pads nothing inside asm).
__device__ __forceinline__ void lds_dma16(unsigned lds_addr, unsigned voff, const void* sbase)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
#else
    (void)lds_addr; (void)voff; (void)sbase;
#endif
}
// per-lane byte offsets are 32-bit: the operands of one launch must span less than 4 GiB
static inline bool dma_offsets_fit(size_t elems_a, size_t elems_b)
{
    return elems_a * 8 < 0xffffffffull && elems_b * 8 < 0xffffffffull;
}

template <int BK, int STAGES, class Epi>
__global__ __launch_bounds__(512) void dgemm_dlds_kernel(GemmP p, Epi epi)
{
    using Cfg = DlCfg<BK, STAGES>;
    constexpr int DL_A_BYTES = Cfg::A_BYTES, DL_STAGE_BYTES = Cfg::STAGE_BYTES, DL_STAGES = STAGES;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;               // 2 x 4 waves, wave tile 80 x 32

    const int nblk = p.gm * p.gn;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int bi = nid / p.gn, bj = nid - bi * p.gn;
    const int m0 = bi * DL_BM, n0 = bj * DL_BN;

    // ---- per-lane source pointers of this wave's LDS-DMA pieces (advance by a constant per K step)
    // A: chunks c = wave, wave + 8, and wave + 16 for waves 0-3 (waves 4-7 repeat their second
    // chunk: identical bytes to the same place, keeps the vmcnt bookkeeping uniform)
    const double* pa[Cfg::NA]; int la[Cfg::NA];
#pragma unroll
    for (int s = 0; s < Cfg::NA; ++s) {
        int c = wave + 8 * s;
        if (c >= Cfg::A_CHUNKS) c = wave + 8;
        const int o = c * 1024 + lane * 16;               // byte offset inside the A image
        const int k = o / (DL_BM * 8), pos = (o - k * DL_BM * 8) >> 3;      // position (doubles) in row k
        const int blk = pos >> 4, within = pos & 15;
        const int m = (((blk ^ (k & 1)) << 4) | within);   // logical row of C this piece holds
        int gm = m0 + m;
        if (gm >= p.M) gm = 0;                             // clamped: feeds rows that are never stored
        pa[s] = p.A + gm + (size_t)k * p.lda;
        la[s] = c * 1024;
    }
    const double* pb[Cfg::NB]; int lb[Cfg::NB];
#pragma unroll
    for (int s = 0; s < Cfg::NB; ++s) {
        const int c = wave + 8 * s;                        // 16 chunks: kp = c >> 1, half = c & 1
        const int kp = c >> 1, n = ((c & 1) << 6) + lane;
        int gn = n0 + n;
        if (gn >= p.N) gn = 0;
        pb[s] = p.B + 2 * kp + (size_t)gn * p.ldb;
        lb[s] = DL_A_BYTES + c * 1024;
    }
    const size_t stepA = (size_t)BK * p.lda;

    auto issue = [&](int stage) {
#if defined(__HIP_DEVICE_COMPILE__)
        // (the host pass must not see the amdgcn builtin in this template-dependent context: clang
        // would silently drop the whole kernel stub)
        char* base = lds + stage * DL_STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < Cfg::NA; ++s) {
            __builtin_amdgcn_global_load_lds(pa[s], (lds_ptr_t)(base + la[s]), 16, 0, 0);
            pa[s] += stepA;
        }
#pragma unroll
        for (int s = 0; s < Cfg::NB; ++s) {
            __builtin_amdgcn_global_load_lds(pb[s], (lds_ptr_t)(base + lb[s]), 16, 0, 0);
            pb[s] += BK;
        }
#else
        (void)stage; (void)stepA;
#endif
    };

    d4 acc[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int l15 = lane & 15, lk = lane >> 4;
    auto compute = [&](int stage) {
        const double* as = reinterpret_cast<const double*>(lds + stage * DL_STAGE_BYTES);
        const double* bs = reinterpret_cast<const double*>(lds + stage * DL_STAGE_BYTES + DL_A_BYTES);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int kk = 4 * ks + lk;
            double a[5], b[2];
#pragma unroll
            for (int i = 0; i < 5; ++i) a[i] = as[kk * DL_BM + (((wr * 5 + i) ^ (kk & 1)) << 4) + l15];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = bs[((kk >> 1) * DL_BN + wc * 32 + 16 * j + l15) * 2 + (kk & 1)];
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = (p.K + BK - 1) / BK;                    // operands are zero-padded to nk * BK
    // each wave issues PER_TILE LDS-DMA pieces per tile; "leave t tiles in flight" = vmcnt(PER_TILE t)
    auto wait_leave = [&](int tiles) {
        static_assert(Cfg::PER_TILE == 5 || Cfg::PER_TILE == 9, "vmcnt immediates below");
        if constexpr (Cfg::PER_TILE == 5) {
            if (tiles >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (tiles == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (tiles >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else if (tiles == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    int issued = 0;
    for (; issued < DL_STAGES - 1 && issued < nk; ++issued) issue(issued);
    wait_leave(issued - 1);                                // tile 0 landed
    __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (issued < nk) {                                 // its stage was last read in step kt-1
            int sn = st + DL_STAGES - 1; if (sn >= DL_STAGES) sn -= DL_STAGES;
            issue(sn);
            ++issued;
        }
        compute(st);
        wait_leave(issued - kt - 2);                       // tile kt+1 landed, later ones stay in flight
        __builtin_amdgcn_s_barrier();
        st = st + 1; if (st >= DL_STAGES) st = 0;
    }

    epi(acc, m0 + wr * 80, n0 + wc * 32, lane, p.M, p.N, bi * 2 + wr);
}

// ---- the same kernel (K step 16, 3 stages) with hand-scheduled operand reads ------------------
// As in dgemm_band.h: the compiler's `s_waitcnt lgkmcnt(0)` before every MFMA group also waits for
// the reads just issued for the next group.  Here the ds_reads go through inline asm, double
// buffered in registers with counted lgkmcnt(7) waits (5 A + 2 B reads per K substep), and the
// end-of-tile vmcnt + barrier sits before the last MFMA group of a tile.
// The XOR-1 block swizzle of the A image depends on the parity of the k row a lane reads (= lk & 1)
// and on the parity of the block (wr * 5 + i): two per-lane base addresses, compile-time offsets.
#if defined(__HIP_DEVICE_COMPILE__)
#define DL_RD(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#else
#define DL_RD(dst, addr, off) (dst = 0.0)
#endif
template <int KS>
__device__ __forceinline__ void dl_read(double (&a)[5], double (&b)[2], unsigned a_i_even, unsigned a_i_odd, unsigned baddr)
{
    DL_RD(a[0], a_i_even, KS * 4 * DL_BM * 8);
    DL_RD(a[1], a_i_odd, KS * 4 * DL_BM * 8 + 128);
    DL_RD(a[2], a_i_even, KS * 4 * DL_BM * 8 + 256);
    DL_RD(a[3], a_i_odd, KS * 4 * DL_BM * 8 + 384);
    DL_RD(a[4], a_i_even, KS * 4 * DL_BM * 8 + 512);
    DL_RD(b[0], baddr, KS * 2 * DL_BN * 16);
    DL_RD(b[1], baddr, KS * 2 * DL_BN * 16 + 256);
}
template <int LEAVE>
__device__ __forceinline__ void dl_wait(double (&a)[5], double (&b)[2])
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LEAVE == 7)
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]), "+v"(b[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]), "+v"(b[1]));
#endif
}
__device__ __forceinline__ void dl_mfma(d4 (&acc)[5][2], const double (&a)[5], const double (&b)[2])
{
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[i][j], 0, 0, 0);
}

template <class Epi>
__global__ __launch_bounds__(512) void dgemm_dlds_asm_kernel(GemmP p, Epi epi)
{
    using Cfg = DlCfg<16, 3>;
    constexpr int BK = 16, STAGES = 3;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nblk = p.gm * p.gn;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int bi = nid / p.gn, bj = nid - bi * p.gn;
    const int m0 = bi * DL_BM, n0 = bj * DL_BN;

    unsigned voa[Cfg::NA]; int la[Cfg::NA];
#pragma unroll
    for (int s = 0; s < Cfg::NA; ++s) {
        int c = wave + 8 * s;
        if (c >= Cfg::A_CHUNKS) c = wave + 8;
        const int o = c * 1024 + lane * 16;
        const int k = o / (DL_BM * 8), pos = (o - k * DL_BM * 8) >> 3;
        const int blk = pos >> 4, within = pos & 15;
        const int m = (((blk ^ (k & 1)) << 4) | within);
        int gm = m0 + m;
        if (gm >= p.M) gm = 0;
        voa[s] = (unsigned)((gm + (size_t)k * p.lda) * 8);
        la[s] = c * 1024;
    }
    unsigned vob[Cfg::NB]; int lb[Cfg::NB];
#pragma unroll
    for (int s = 0; s < Cfg::NB; ++s) {
        const int c = wave + 8 * s;
        const int kp = c >> 1, n = ((c & 1) << 6) + lane;
        int gn = n0 + n;
        if (gn >= p.N) gn = 0;
        vob[s] = (unsigned)((2 * kp + (size_t)gn * p.ldb) * 8);
        lb[s] = Cfg::A_BYTES + c * 1024;
    }
    const size_t stepA = (size_t)BK * p.lda * 8;           // bytes per K step
    const char* sA = reinterpret_cast<const char*>(p.A);   // uniform bases, advanced by scalar adds
    const char* sB = reinterpret_cast<const char*>(p.B);
    auto issue = [&](int stage) {
        const unsigned lbase = (unsigned)(size_t)(lds_ptr_t)lds + stage * Cfg::STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < Cfg::NA; ++s) lds_dma16(lbase + la[s], voa[s], sA);
#pragma unroll
        for (int s = 0; s < Cfg::NB; ++s) lds_dma16(lbase + lb[s], vob[s], sB);
        sA += stepA;
        sB += BK * 8;
    };
    auto wait_leave = [&](int tiles) {
        static_assert(Cfg::PER_TILE == 5, "vmcnt immediates below");
        if (tiles >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (tiles == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    d4 acc[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    const int l15 = lane & 15, lk = lane >> 4, par = lk & 1;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds;
    // block wr*5 + i read by a lane of k-parity `par` sits at block (wr*5 + i) ^ par
    const unsigned abase = lds0 + lk * (DL_BM * 8) + l15 * 8 + wr * 5 * 128;
    const unsigned a_blk_even = abase + par * 128, a_blk_odd = abase - par * 128;       // for even / odd block index
    const unsigned a_i_even = (wr & 1) ? a_blk_odd : a_blk_even;                         // parity of wr*5 + i, i even
    const unsigned a_i_odd = (wr & 1) ? a_blk_even : a_blk_odd;
    const unsigned boff = lds0 + Cfg::A_BYTES + (lk >> 1) * (DL_BN * 16) + wc * 512 + l15 * 16 + par * 8;

    const int nk = (p.K + BK - 1) / BK;
    int issued = 0;
    for (; issued < STAGES - 1 && issued < nk; ++issued) issue(issued);
    wait_leave(issued - 1);
    __builtin_amdgcn_s_barrier();
    int st = 0;
    double ra[2][5], rb[2][2];
    dl_read<0>(ra[0], rb[0], a_i_even, a_i_odd, boff);
    for (int kt = 0; kt < nk; ++kt) {
        if (issued < nk) {
            int sn = st + STAGES - 1; if (sn >= STAGES) sn -= STAGES;
            issue(sn);
            ++issued;
        }
        const unsigned so = st * Cfg::STAGE_BYTES;
        int stn = st + 1; if (stn >= STAGES) stn = 0;
#define DL_STEP(KS, CUR, NXT)                                                            \
        dl_read<KS + 1>(ra[NXT], rb[NXT], a_i_even + so, a_i_odd + so, boff + so);         \
        dl_wait<7>(ra[CUR], rb[CUR]);                                                      \
        dl_mfma(acc, ra[CUR], rb[CUR]);                                                    \
        __builtin_amdgcn_sched_barrier(0);
        DL_STEP(0, 0, 1) DL_STEP(1, 1, 0) DL_STEP(2, 0, 1)
#undef DL_STEP
        wait_leave(issued - kt - 2);
        dl_wait<0>(ra[1], rb[1]);
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) {
            const unsigned sn2 = stn * Cfg::STAGE_BYTES;
            dl_read<0>(ra[0], rb[0], a_i_even + sn2, a_i_odd + sn2, boff + sn2);
        }
        dl_mfma(acc, ra[1], rb[1]);
        __builtin_amdgcn_sched_barrier(0);
        st = stn;
    }
    epi(acc, m0 + wr * 80, n0 + wc * 32, lane, p.M, p.N, bi * 2 + wr);
}

// Host contract of the direct-to-LDS kernel; returns false when the generic kernel must be used.
static inline bool dlds_applicable(int M, int N, int K, const double* A, int lda, int a_cols_alloc,
                                   const double* B, int ldb)
{
    const int kpad = round_up(K, 32);      // serves both K steps
    return M >= 1 && N >= 1 && K >= 1 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
           (lda & 1) == 0 && (ldb & 1) == 0 && lda >= M && ldb >= kpad && a_cols_alloc >= kpad &&
           dma_offsets_fit((size_t)lda * 32 + M, (size_t)ldb * N + 32);
}

template <int BK, int STAGES, class Epi>
static inline int launch_gemm_dlds_cfg(hipStream_t s, const GemmP& p, const Epi& epi)
{
    using Cfg = DlCfg<BK, STAGES>;
    MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&dgemm_dlds_kernel<BK, STAGES, Epi>), (int)Cfg::LDS_BYTES));
    hipLaunchKernelGGL((dgemm_dlds_kernel<BK, STAGES, Epi>), dim3(p.gm * p.gn), dim3(512), Cfg::LDS_BYTES, s, p, epi);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

template <class Epi>
static inline int launch_gemm_dlds(hipStream_t s, int M, int N, int K, const double* A, int lda,
                                   const double* B, int ldb, const Epi& epi)
{
    GemmP p{M, N, K, A, lda, B, ldb, (M + DL_BM - 1) / DL_BM, (N + DL_BN - 1) / DL_BN, 0, 0};
    // GLMMR_MCML_DLDS: 0 = hand-scheduled reads (default), 1 = (32, 2) compiler-scheduled, 2 = (16, 3) compiler-scheduled
    static const int variant = getenv("GLMMR_MCML_DLDS") ? atoi(getenv("GLMMR_MCML_DLDS")) : 0;
    if (variant == 1) return launch_gemm_dlds_cfg<32, 2, Epi>(s, p, epi);
    if (variant == 2) return launch_gemm_dlds_cfg<16, 3, Epi>(s, p, epi);
    {
        using Cfg = DlCfg<16, 3>;
        MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&dgemm_dlds_asm_kernel<Epi>), (int)Cfg::LDS_BYTES));
        hipLaunchKernelGGL((dgemm_dlds_asm_kernel<Epi>), dim3(p.gm * p.gn), dim3(512), Cfg::LDS_BYTES, s, p, epi);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
}

}  // namespace mcml
