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

constexpr int DL_BM = 160, DL_BN = 128, DL_BK = 16;
#ifndef DL_STAGES_N
#define DL_STAGES_N 3
#endif
constexpr int DL_STAGES = DL_STAGES_N;      // LDS ring: DL_STAGES - 1 tiles in flight
constexpr int DL_A_BYTES = DL_BK * DL_BM * 8;          // 20480 = 20 chunks of 1 KiB
constexpr int DL_B_BYTES = DL_BK * DL_BN * 8;          // 16384 = 16 chunks
constexpr int DL_STAGE_BYTES = DL_A_BYTES + DL_B_BYTES;
constexpr size_t DL_LDS_BYTES = (size_t)DL_STAGES * DL_STAGE_BYTES;   // 110592

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <class Epi>
__global__ __launch_bounds__(512) void dgemm_dlds_kernel(GemmP p, Epi epi)
{
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
    const double* pa[3]; int la[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        int c = wave + 8 * s;
        if (c >= 20) c = wave + 8;
        const int o = c * 1024 + lane * 16;               // byte offset inside the A image
        const int k = o / (DL_BM * 8), pos = (o - k * DL_BM * 8) >> 3;      // position (doubles) in row k
        const int blk = pos >> 4, within = pos & 15;
        const int m = (((blk ^ (k & 1)) << 4) | within);   // logical row of C this piece holds
        int gm = m0 + m;
        if (gm >= p.M) gm = 0;                             // clamped: feeds rows that are never stored
        pa[s] = p.A + gm + (size_t)k * p.lda;
        la[s] = c * 1024;
    }
    const double* pb[2]; int lb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int c = wave + 8 * s;                        // 16 chunks: kp = c >> 1, half = c & 1
        const int kp = c >> 1, n = ((c & 1) << 6) + lane;
        int gn = n0 + n;
        if (gn >= p.N) gn = 0;
        pb[s] = p.B + 2 * kp + (size_t)gn * p.ldb;
        lb[s] = DL_A_BYTES + c * 1024;
    }
    const size_t stepA = (size_t)DL_BK * p.lda;

    auto issue = [&](int stage) {
        char* base = lds + stage * DL_STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            __builtin_amdgcn_global_load_lds(pa[s], (lds_ptr_t)(base + la[s]), 16, 0, 0);
            pa[s] += stepA;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            __builtin_amdgcn_global_load_lds(pb[s], (lds_ptr_t)(base + lb[s]), 16, 0, 0);
            pb[s] += DL_BK;
        }
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
        for (int ks = 0; ks < DL_BK / 4; ++ks) {
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

    const int nk = (p.K + DL_BK - 1) / DL_BK;              // operands are zero-padded to nk * 16
    // each wave issues 5 LDS-DMA pieces per tile; "leave t tiles in flight" = vmcnt(5 t)
    auto wait_leave = [&](int tiles) {
        if (tiles >= 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (tiles == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (tiles == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// Host contract of the direct-to-LDS kernel; returns false when the generic kernel must be used.
static inline bool dlds_applicable(int M, int N, int K, const double* A, int lda, int a_cols_alloc,
                                   const double* B, int ldb)
{
    const int kpad = round_up(K, DL_BK);
    return M >= 1 && N >= 1 && K >= 1 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
           (lda & 1) == 0 && (ldb & 1) == 0 && lda >= M && ldb >= kpad && a_cols_alloc >= kpad;
}

template <class Epi>
static inline int launch_gemm_dlds(hipStream_t s, int M, int N, int K, const double* A, int lda,
                                   const double* B, int ldb, const Epi& epi)
{
    GemmP p{M, N, K, A, lda, B, ldb, (M + DL_BM - 1) / DL_BM, (N + DL_BN - 1) / DL_BN, 0, 0};
    static bool attr_set = false;
    if (!attr_set) {
        MCML_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dgemm_dlds_kernel<Epi>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)DL_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((dgemm_dlds_kernel<Epi>), dim3(p.gm * p.gn), dim3(512), DL_LDS_BYTES, s, p, epi);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

}  // namespace mcml
