// dgemm_band2.h -- the banded zero-skipping GEMM (dgemm_band.h) re-cut so that TWO workgroups
// share a CU: 80 x 64 tile, 4 waves (each a 80 x 16 strip), K step 32, 2-stage LDS ring = 72 KB.
//
// Same operands, LDS images, hand-scheduled reads and band pairing as dgemm_band_kernel.  What
// changes is who fills the bubbles: with one 156 KB workgroup per CU the matrix pipe idles during
// that workgroup's barriers, pipeline fills and epilogues; with two independent 4-wave workgroups
// (8 waves per CU, as before) one's MFMA work covers the other's.  Co-resident workgroups would
// run in lock step (equal work), so odd-numbered pairs process their bands in the opposite order
// (long band first): the mid-kernel epilogues then fall at different times.
// 63 bands x 16 column tiles -> 32 x 16 = 512 workgroups for the 5000 x 1024 products.
#pragma once
#include "dgemm_band.h"

namespace mcml {

constexpr int B2_BN = 64, B2_NW = 4, B2_STAGES = 2;
constexpr int B2_A_BYTES = BD_BK * BD_BM * 8;      // 20480: 20 chunks of 1 KiB
constexpr int B2_B_BYTES = BD_BK * B2_BN * 8;      // 16384: 16 chunks (chunk = one k-pair row of 64 columns)
constexpr int B2_STAGE_BYTES = B2_A_BYTES + B2_B_BYTES;
constexpr size_t B2_LDS_BYTES = (size_t)B2_STAGES * B2_STAGE_BYTES;   // 73728
constexpr int B2_NA = 5, B2_NB = 4;                // LDS-DMA pieces per wave per tile

template <int KS>
__device__ __forceinline__ void b2_read(double (&a)[5], double& b, unsigned aaddr, unsigned baddr)
{
    BD_RD(a[0], aaddr, KS * 4 * BD_BM * 8);
    BD_RD(a[1], aaddr, KS * 4 * BD_BM * 8 + 128);
    BD_RD(a[2], aaddr, KS * 4 * BD_BM * 8 + 256);
    BD_RD(a[3], aaddr, KS * 4 * BD_BM * 8 + 384);
    BD_RD(a[4], aaddr, KS * 4 * BD_BM * 8 + 512);
    BD_RD(b, baddr, KS * 2 * B2_BN * 16);
}

template <class Epi>
__global__ __launch_bounds__(256) void dgemm_band2_kernel(BandP bp, Epi epi)
{
    const GemmP& p = bp.g;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lk = lane >> 4;

    const int npairs = (bp.nbands + 1) >> 1;
    const int nblk = npairs * p.gn;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int pb = nid / p.gn, bj = nid - pb * p.gn;
    const int n0 = bj * B2_BN;

    // B pieces: chunk c = k-pair row c of the tile, lane = column
    const double* pb0[B2_NB]; int lb[B2_NB];
#pragma unroll
    for (int s = 0; s < B2_NB; ++s) {
        const int c = wave + B2_NW * s;
        int gn = n0 + lane;
        if (gn >= p.N) gn = 0;
        pb0[s] = p.B + 2 * c + (size_t)gn * p.ldb;
        lb[s] = B2_A_BYTES + c * 1024;
    }
    const size_t stepA = (size_t)BD_BK * p.lda;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds;
    const unsigned aoff = lds0 + lk * (BD_BM * 8) + l15 * 8;
    const unsigned boff = lds0 + B2_A_BYTES + (((lk >> 1) * B2_BN + wave * 16 + l15) << 4) + ((lk & 1) << 3);

    int bandA = pb, bandB = bp.nbands - 1 - pb;            // bandB == bandA: odd count, the middle band runs once
    if ((nid & 1) && bandB > bandA) { const int t = bandA; bandA = bandB; bandB = t; }

    for (int pass = 0; pass < 2; ++pass) {
        const int band = pass == 0 ? bandA : bandB;
        if (pass == 1 && bandB == bandA) break;
        const int m0 = band * BD_BM;
        const int kt0 = bp.krange[2 * band], kt1 = bp.krange[2 * band + 1];

        const double* pa[B2_NA]; int la[B2_NA];
#pragma unroll
        for (int s = 0; s < B2_NA; ++s) {
            const int c = wave + B2_NW * s;                // 0..19
            const int o = c * 1024 + lane * 16;
            const int k = o / (BD_BM * 8), m = (o - k * BD_BM * 8) >> 3;
            int gm = m0 + m;
            if (gm >= p.M) gm = 0;
            pa[s] = p.A + gm + (size_t)(kt0 * BD_BK + k) * p.lda;
            la[s] = c * 1024;
        }
        const double* pbb[B2_NB];
#pragma unroll
        for (int s = 0; s < B2_NB; ++s) pbb[s] = pb0[s] + (size_t)kt0 * BD_BK;

        auto issue = [&](int stage) {
#if defined(__HIP_DEVICE_COMPILE__)
            char* base = lds + stage * B2_STAGE_BYTES;
#pragma unroll
            for (int s = 0; s < B2_NA; ++s) {
                __builtin_amdgcn_global_load_lds(pa[s], (lds_ptr_t)(base + la[s]), 16, 0, 0);
                pa[s] += stepA;
            }
#pragma unroll
            for (int s = 0; s < B2_NB; ++s) {
                __builtin_amdgcn_global_load_lds(pbb[s], (lds_ptr_t)(base + lb[s]), 16, 0, 0);
                pbb[s] += BD_BK;
            }
#else
            (void)stage; (void)stepA;
#endif
        };

        d4 acc[5][1];
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i][0] = d4{0.0, 0.0, 0.0, 0.0};

        const int nk = kt1 - kt0;
        if (nk > 0) {
            // (the previous band's epilogue stores share vmcnt: vmcnt(0) drains those too)
            issue(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int st = 0;
            double ra[2][5], rb[2];
            b2_read<0>(ra[0], rb[0], aoff, boff);
            for (int kt = 0; kt < nk; ++kt) {
                const int stn = st ^ 1;
                if (kt + 1 < nk) issue(stn);               // its stage was last read in step kt-1 (barrier passed)
                const unsigned aaddr = aoff + st * B2_STAGE_BYTES, baddr = boff + st * B2_STAGE_BYTES;
#define B2_STEP(KS, CUR, NXT)                                    \
                b2_read<KS + 1>(ra[NXT], rb[NXT], aaddr, baddr);  \
                bd_wait<6>(ra[CUR], rb[CUR]);                     \
                bd_mfma(acc, ra[CUR], rb[CUR]);                   \
                __builtin_amdgcn_sched_barrier(0);
                B2_STEP(0, 0, 1) B2_STEP(1, 1, 0) B2_STEP(2, 0, 1) B2_STEP(3, 1, 0)
                B2_STEP(4, 0, 1) B2_STEP(5, 1, 0) B2_STEP(6, 0, 1)
#undef B2_STEP
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+1 landed
                bd_wait<0>(ra[1], rb[1]);
                __builtin_amdgcn_s_barrier();
                if (kt + 1 < nk) b2_read<0>(ra[0], rb[0], aoff + stn * B2_STAGE_BYTES, boff + stn * B2_STAGE_BYTES);
                bd_mfma(acc, ra[1], rb[1]);
                __builtin_amdgcn_sched_barrier(0);
                st = stn;
            }
        }
        epi(acc, m0, n0 + wave * 16, lane, p.M, p.N, band);
        // the next band's first LDS-DMA may overwrite a stage another wave is still reading
        __builtin_amdgcn_s_barrier();
    }
}

template <class Epi>
static inline int launch_gemm_band2(hipStream_t s, int M, int N, int K, const double* A, int lda,
                                    const double* B, int ldb, const int* krange, const Epi& epi)
{
    BandP bp;
    bp.clocks = nullptr;
    bp.mode = 0;
    bp.g = GemmP{M, N, K, A, lda, B, ldb, 0, (N + B2_BN - 1) / B2_BN, 0, 0};
    bp.krange = krange;
    bp.nbands = (M + BD_BM - 1) / BD_BM;
    const int npairs = (bp.nbands + 1) / 2;
    MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&dgemm_band2_kernel<Epi>), (int)B2_LDS_BYTES));
    hipLaunchKernelGGL((dgemm_band2_kernel<Epi>), dim3(npairs * bp.g.gn), dim3(256), B2_LDS_BYTES, s, bp, epi);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}

}  // namespace mcml
