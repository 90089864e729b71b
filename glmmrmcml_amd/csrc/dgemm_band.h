// dgemm_band.h -- the HMC products with structural-zero skipping, any number of chains.
//
// For the geospatial designs Z = I, so ZL = L is lower triangular (and ZL' upper): half of
// the K tiles a dense GEMM multiplies are exactly zero.  More generally, whenever the rows of
// ZL (or ZL') touch only a range of columns, the products outside that range contribute
// nothing.  When L is refreshed, k_band_ranges records for every 80-row band of the A operand
// the first and last K tile that holds a nonzero; the kernel runs each band's K loop over that
// range only.  Skipping exact zeros does not change any sum (0*x adds nothing for finite x).
//
// Work decomposition (BandPlan, built on the host per number of column tiles):
//   * paired: a workgroup owns band p and its mirror nbands-1-p of one column tile, back to
//     back, so for a triangular operand every workgroup executes the same number of K steps:
//     63 bands x 8 column tiles -> 32 x 8 = 256 workgroups (one per CU) for the 5000 x 1024
//     products.  No partial sums.
//   * streamed (few column tiles: a rank of the chain-sharded job holds 1024/8 = 128 chains = ONE
//     column tile, which paired would run on 32 of the 256 CUs): the (band, K tile) iteration
//     space of a column tile is flattened and cut into equal runs of K tiles, one run per
//     workgroup, ~256 workgroups in all.  A run may end one band and start the next.  A band
//     covered by one run is finished by that workgroup; a band cut into several pieces has each
//     piece stored as a raw accumulator tile (in register order: 512-byte coalesced stores) and
//     k_band_reduce sums the pieces IN K ORDER and applies the same epilogue functor -- a
//     fixed-order two-stage reduction, bit-reproducible for a given (operand, chain count).
//     (Measured dead end, round 3: folding the second stage into the piece that ARRIVES LAST -- write-through
//     (sc1) tile stores, one relaxed agent-scope counter per band, sc1 loads by the last arriver -- is correct
//     (parity suite green) but no faster: 80.4 / 82.7 us against 76.7 / 82.4 us per forward / backward product at
//     n = 5000 with 128 chains, 44.5 / 47.9 against 39.2 / 41.7 us at n = 2000 with 256 -- one workgroup per band
//     reads up to eight 80 KB pieces back from memory in dependent batches, where this kernel spreads a band
//     over four workgroups and finds the pieces in L2.  With agent-scope release / acquire fences instead of
//     write-through stores every wave issues a buffer_wbl2, a walk over the XCD's whole L2: 160 us per product.)
//
// Same direct-to-LDS machinery as dgemm_dlds.h: 80 x 128 tile, 8 waves (each a 80 x 16 strip =
// 5 x 1 v_mfma_f64_16x16x4 tiles), K step 32, 3-stage LDS ring (156 KB), counted vmcnt + raw
// barriers.  The A image [k][80 doubles] has 640-byte rows = 32 banks mod 64, so the two
// 16-lane halves of a ds_read_b64 (rows k, k+1) are conflict-free without a swizzle.
#pragma once
#include "dgemm_dlds.h"
#include "band_plan.h"

namespace mcml {

struct BandP {
    GemmP g;
    const BandItem* items;
    const int* wg_ptr;     // items of workgroup w (of any column tile): [wg_ptr[w], wg_ptr[w+1])
    int nwg;               // workgroups per column tile
    double* part;          // partial tiles: [(slot * gn + column tile) * BD_TILE_ELEMS]
    int mode;              // DBG kernels only -- timing experiments: 1 no LDS-DMA, 2 no barriers, 4 no ds_reads, 8 compiler-scheduled reads
    unsigned long long* clocks;   // DBG kernels only (nullable): workgroup 0 adds its shader-clock and 100 MHz real-time ticks
};

// first / last nonzero K tile of every 80-row band of A (M x K, column-major)
static __global__ __launch_bounds__(256) void k_band_ranges(const double* A, int lda, int M, int K, int* krange)
{
    __shared__ int smin[256], smax[256];
    const int band = blockIdx.x, r0 = band * BD_BM;
    const int rows = (M - r0 < BD_BM) ? (M - r0) : BD_BM;
    int kmin = 1 << 30, kmax = -1;
    for (int k = threadIdx.x; k < K; k += 256) {
        const double* col = A + r0 + (size_t)k * lda;
        bool nz = false;
        for (int r = 0; r < rows; ++r) nz |= (col[r] != 0.0);
        if (nz) { kmin = min(kmin, k); kmax = max(kmax, k); }
    }
    smin[threadIdx.x] = kmin; smax[threadIdx.x] = kmax;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smin[threadIdx.x] = min(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (smax[0] < 0) { krange[2 * band] = 0; krange[2 * band + 1] = 0; }
        else { krange[2 * band] = smin[0] / BD_BK; krange[2 * band + 1] = smax[0] / BD_BK + 1; }
    }
}

// ---- hand-scheduled operand reads -------------------------------------------------------
// The compiler's waitcnt insertion puts `s_waitcnt lgkmcnt(0)` in front of every MFMA group,
// which also waits for the reads just issued for the NEXT group: measured (debug modes of
// glmmr_mcml_dbg_band_clocks) the matrix pipe then idles ~14% of the K loop on LDS latency.
// The reads are therefore issued through inline asm the compiler does not track, and the
// counted waits are placed by hand (LDS returns in order: lgkmcnt(6) = "everything but the six
// reads just issued").  The waits carry the operand registers as in/out operands so the MFMAs
// that consume them cannot be scheduled above the wait.
#if defined(__HIP_DEVICE_COMPILE__)
#define BD_RD(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#else
#define BD_RD(dst, addr, off) (dst = 0.0)
#endif
// operands of K substep KS (4 k's) of the stage whose per-lane byte addresses are aaddr / baddr
template <int KS>
__device__ __forceinline__ void bd_read(double (&a)[5], double& b, unsigned aaddr, unsigned baddr)
{
    BD_RD(a[0], aaddr, KS * 4 * BD_BM * 8);
    BD_RD(a[1], aaddr, KS * 4 * BD_BM * 8 + 128);
    BD_RD(a[2], aaddr, KS * 4 * BD_BM * 8 + 256);
    BD_RD(a[3], aaddr, KS * 4 * BD_BM * 8 + 384);
    BD_RD(a[4], aaddr, KS * 4 * BD_BM * 8 + 512);
    BD_RD(b, baddr, KS * 2 * BD_BN * 16);
}
template <int LEAVE>
__device__ __forceinline__ void bd_wait(double (&a)[5], double& b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LEAVE == 6)
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b));
#endif
}
__device__ __forceinline__ void bd_mfma(d4 (&acc)[5][1], const double (&a)[5], double b)
{
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a[i], acc[i][0], 0, 0, 0);
}

// DBG = true only for the timing hooks of debug_hooks.hip (bp.mode / bp.clocks); the sampler's instantiation
// carries neither the experiment branches nor the clock reads
template <class Epi, bool DBG>
__global__ __launch_bounds__(512) void dgemm_band_kernel(BandP bp, Epi epi)
{
    const GemmP& p = bp.g;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lk = lane >> 4;
    unsigned long long t0c = 0, t0r = 0;
    if constexpr (DBG)
        if (bp.clocks && blockIdx.x == 0 && tid == 0) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }

    // XCD-aware map over (workgroup of the plan, column tile): the column tiles of one run of K tiles
    // share an XCD (they read the same pieces of A)
    const int nblk = bp.nwg * p.gn;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int pw = nid / p.gn, bj = nid - pw * p.gn;
    const int n0 = bj * BD_BN;

    // Operand addressing: a UNIFORM base (SGPR pair, advanced per K tile by scalar adds) plus a per-lane 32-bit
    // byte offset that never changes inside a K loop (`global_load_lds_dwordx4 voff, s[base]`).  With per-lane
    // 64-bit pointers advanced by VALU adds the compiler may place an add right behind the LDS-DMA that reads
    // the same address register; that write-after-read stalls the wave until the load has left the queue
    // (measured: 458 vs 419 us per launch, the difference in SQ_WAIT_INST_ANY).
    // B pieces do not depend on the band: 32 chunks, chunk c -> (kp = c >> 1, half = c & 1)
    unsigned vob[BD_NB]; int lb[BD_NB];
#pragma unroll
    for (int s = 0; s < BD_NB; ++s) {
        const int c = wave + 8 * s;
        const int kp = c >> 1, n = ((c & 1) << 6) + lane;
        int gn = n0 + n;
        if (gn >= p.N) gn = 0;
        vob[s] = (unsigned)((2 * kp + (size_t)gn * p.ldb) * 8);
        lb[s] = BD_A_BYTES + c * 1024;
    }
    const size_t stepA = (size_t)BD_BK * p.lda * 8;      // bytes per K tile

    // The item list is read with a SCALAR load (hand-written: the compiler will not scalarise a load it
    // cannot prove unclobbered by the epilogue's stores).  A vector load here sits behind the previous item's
    // epilogue stores -- vector memory completes in order, its `s_waitcnt vmcnt(0)` drains every store before
    // the next item's first LDS-DMA can even be issued: measured 471 vs 428 us per launch at 1024 chains.
    const int it0 = bp.wg_ptr[pw], it1 = bp.wg_ptr[pw + 1];
    // the item being worked on and its operand addressing; load_item() replaces them (after a K loop they are dead)
    BandItem item{0, 0, 0, -1};
    int m0 = 0;
    unsigned voa[BD_NA]; int la[BD_NA];
#pragma unroll
    for (int s = 0; s < BD_NA; ++s) {
        int c = wave + 8 * s;
        if (c >= 20) c = wave + 8;                          // waves 4-7 repeat a chunk: uniform vmcnt
        la[s] = c * 1024;                                   // LDS offset of the piece: the same for every item
    }
    const char* sA = nullptr; const char* sB = nullptr;                 // uniform
    auto load_item = [&](int it) {
#if defined(__HIP_DEVICE_COMPILE__)
        {
            typedef int i4s __attribute__((ext_vector_type(4)));
            i4s raw;
            const BandItem* ip = bp.items + it;
            asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(raw) : "s"(ip) : "memory");
            item.band = raw.x; item.kt0 = raw.y; item.kt1 = raw.z; item.slot = raw.w;
        }
#else
        item = bp.items[it];
#endif
        m0 = item.band * BD_BM;
#pragma unroll
        for (int s = 0; s < BD_NA; ++s) {
            int c = wave + 8 * s;
            if (c >= 20) c = wave + 8;                      // waves 4-7 repeat a chunk: uniform vmcnt
            const int o = c * 1024 + lane * 16;
            const int k = o / (BD_BM * 8), m = (o - k * BD_BM * 8) >> 3;
            int gm = m0 + m;
            if (gm >= p.M) gm = 0;
            voa[s] = (unsigned)((gm + (size_t)k * p.lda) * 8);
        }
        sA = reinterpret_cast<const char*>(p.A) + (size_t)item.kt0 * stepA;
        sB = reinterpret_cast<const char*>(p.B) + (size_t)item.kt0 * (BD_BK * 8);
    };
    auto issue = [&](int stage) {
#if defined(__HIP_DEVICE_COMPILE__)
        // hand-written so that the scalar-base form is what runs (hipcc materialises base + offset into one
        // 64-bit temporary per load instead); M0 = LDS byte address of the piece, lane l lands at M0 + 16 l;
        // the s_nop is the wait state an LDS-DMA needs behind a SALU write of M0 (hipcc pads nothing inside asm)
        const unsigned lbase = (unsigned)(size_t)(lds_ptr_t)lds + stage * BD_STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < BD_NA; ++s)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(lbase + la[s]), "v"(voa[s]), "s"(sA) : "memory", "m0");
#pragma unroll
        for (int s = 0; s < BD_NB; ++s)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(lbase + lb[s]), "v"(vob[s]), "s"(sB) : "memory", "m0");
        sA += stepA;
        sB += BD_BK * 8;
#else
        (void)stage; (void)stepA; (void)sA; (void)sB;
#endif
    };
    auto wait_leave = [&](int tiles) {
        static_assert(BD_PER_TILE == 7, "vmcnt immediates below");
        if (tiles >= 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else if (tiles == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int mode = DBG ? (bp.mode & 15) : 0;
    // DBG, mode bit 16: every workgroup leaves 100 MHz real-time stamps of its phases at clocks[8 + 8 * bid ..]:
    // [start, first ring stages landed, K loop of the first item done, its epilogue / partial store issued, end]
    const bool stamps = DBG && (bp.mode & 16) && bp.clocks;
    auto stamp = [&](int i) {
        if constexpr (DBG)
            if (stamps && tid == 0) bp.clocks[8 + 8 * (size_t)blockIdx.x + i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    bool first_item = true;
    // `pre`: the first ring stages of the item about to start are already in flight -- they were issued BEFORE the
    // previous item's epilogue (the ring is free once every wave has passed the last barrier of a K loop), so the
    // fill's memory latency runs under the epilogue's loads and stores instead of after them
    bool pre = false;
    int issued = 0;
    if (it0 < it1) load_item(it0);
    for (int it = it0; it < it1;) {
        const int band = item.band, e_m0 = m0, slot = item.slot;
        const int nk = item.kt1 - item.kt0;

        d4 acc[5][1];
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i][0] = d4{0.0, 0.0, 0.0, 0.0};

        if (nk > 0 && mode == 0) {
            // per-lane LDS byte addresses of the operand reads inside stage 0
            const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds;
            const unsigned aoff = lds0 + lk * (BD_BM * 8) + l15 * 8;
            const unsigned boff = lds0 + BD_A_BYTES + (((lk >> 1) * BD_BN + wave * 16 + l15) << 4) + ((lk & 1) << 3);
            if (!pre) { issued = 0; for (; issued < BD_STAGES - 1 && issued < nk; ++issued) issue(issued); }
            wait_leave(issued - 1);
            __builtin_amdgcn_s_barrier();
            if (first_item) stamp(1);
            int st = 0;
            double ra[2][5], rb[2];
            bd_read<0>(ra[0], rb[0], aoff, boff);
            for (int kt = 0; kt < nk; ++kt) {
                if (issued < nk) {
                    int sn = st + BD_STAGES - 1; if (sn >= BD_STAGES) sn -= BD_STAGES;
                    issue(sn);
                    ++issued;
                }
                const unsigned aaddr = aoff + st * BD_STAGE_BYTES, baddr = boff + st * BD_STAGE_BYTES;
                int stn = st + 1; if (stn >= BD_STAGES) stn = 0;
#define BD_STEP(KS, CUR, NXT)                                   \
                bd_read<KS + 1>(ra[NXT], rb[NXT], aaddr, baddr); \
                bd_wait<6>(ra[CUR], rb[CUR]);                    \
                bd_mfma(acc, ra[CUR], rb[CUR]);                  \
                __builtin_amdgcn_sched_barrier(0);
                BD_STEP(0, 0, 1) BD_STEP(1, 1, 0) BD_STEP(2, 0, 1) BD_STEP(3, 1, 0)
                BD_STEP(4, 0, 1) BD_STEP(5, 1, 0) BD_STEP(6, 0, 1)
#undef BD_STEP
                // last substep: synchronise first (this wave's pieces of tile kt+1 landed, its own
                // reads of tile kt complete), start the next tile's first reads, then the MFMAs
                wait_leave(issued - kt - 2);
                bd_wait<0>(ra[1], rb[1]);
                __builtin_amdgcn_s_barrier();
                if (kt + 1 < nk) bd_read<0>(ra[0], rb[0], aoff + stn * BD_STAGE_BYTES, boff + stn * BD_STAGE_BYTES);
                bd_mfma(acc, ra[1], rb[1]);
                __builtin_amdgcn_sched_barrier(0);
                st = stn;
            }
        } else if (DBG && nk > 0 && mode == 8) {
            // compiler-scheduled operand reads (the loop the hand-scheduled one replaced)
            int issued = 0;
            for (; issued < BD_STAGES - 1 && issued < nk; ++issued) issue(issued);
            wait_leave(issued - 1);
            __builtin_amdgcn_s_barrier();
            int st = 0;
            for (int kt = 0; kt < nk; ++kt) {
                if (issued < nk) {
                    int sn = st + BD_STAGES - 1; if (sn >= BD_STAGES) sn -= BD_STAGES;
                    issue(sn);
                    ++issued;
                }
                const double* as = reinterpret_cast<const double*>(lds + st * BD_STAGE_BYTES);
                const double* bs = reinterpret_cast<const double*>(lds + st * BD_STAGE_BYTES + BD_A_BYTES);
#pragma unroll
                for (int ks = 0; ks < BD_BK / 4; ++ks) {
                    const int kk = 4 * ks + lk;
                    double a[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) a[i] = as[kk * BD_BM + 16 * i + l15];
                    const double b = bs[((kk >> 1) * BD_BN + wave * 16 + l15) * 2 + (kk & 1)];
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a[i], acc[i][0], 0, 0, 0);
                }
                wait_leave(issued - kt - 2);
                __builtin_amdgcn_s_barrier();
                st = st + 1; if (st >= BD_STAGES) st = 0;
            }
        } else if (DBG && nk > 0) {
            // timing experiments: the same loop with parts switched off (results meaningless)
            int issued = 0;
            if (!(mode & 1)) {
                for (; issued < BD_STAGES - 1 && issued < nk; ++issued) issue(issued);
                wait_leave(issued - 1);
            }
            __builtin_amdgcn_s_barrier();
            int st = 0;
            double a[5] = {1.0 + lane, 2.0, 3.0, 4.0, 5.0}, b = 0.5 + lane;
            for (int kt = 0; kt < nk; ++kt) {
                if (!(mode & 1) && issued < nk) {
                    int sn = st + BD_STAGES - 1; if (sn >= BD_STAGES) sn -= BD_STAGES;
                    issue(sn);
                    ++issued;
                }
                const double* as = reinterpret_cast<const double*>(lds + st * BD_STAGE_BYTES);
                const double* bs = reinterpret_cast<const double*>(lds + st * BD_STAGE_BYTES + BD_A_BYTES);
#pragma unroll
                for (int ks = 0; ks < BD_BK / 4; ++ks) {
                    const int kk = 4 * ks + lk;
                    if (!(mode & 4)) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) a[i] = as[kk * BD_BM + 16 * i + l15];
                        b = bs[((kk >> 1) * BD_BN + wave * 16 + l15) * 2 + (kk & 1)];
                    }
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a[i], acc[i][0], 0, 0, 0);
                }
                if (!(mode & 1)) wait_leave(issued - kt - 2);
                if (!(mode & 2)) __builtin_amdgcn_s_barrier();
                st = st + 1; if (st >= BD_STAGES) st = 0;
            }
        }
        if (first_item) stamp(2);
        ++it;
        pre = false;
        if (it < it1) {
            load_item(it);
            const int nkn = item.kt1 - item.kt0;
            if (mode == 0 && nk > 0 && nkn > 0) {
                issued = 0;
                for (; issued < BD_STAGES - 1 && issued < nkn; ++issued) issue(issued);
                pre = true;
            }
        }
        if (slot < 0) {
            epi(acc, e_m0, n0 + wave * 16, lane, p.M, p.N, band);
        } else {
            // raw accumulator tile in register order: element (wave, i, r, lane) -- each store instruction
            // writes 512 contiguous bytes; k_band_reduce reads it back with the same thread mapping
            double* P = bp.part + ((size_t)slot * p.gn + bj) * BD_TILE_ELEMS + (size_t)wave * (20 * 64) + lane;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[(i * 4 + r) * 64] = acc[i][0][r];
        }
        if (first_item) { stamp(3); first_item = false; }
        // without a prefetch the next item's first LDS-DMA may overwrite a stage another wave is still reading
        if (!pre) __builtin_amdgcn_s_barrier();
    }
    if constexpr (DBG)
        if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(4); }
    if constexpr (DBG)
        if (bp.clocks && blockIdx.x == 0 && tid == 0) {
            atomicAdd(bp.clocks, __builtin_amdgcn_s_memtime() - t0c);
            atomicAdd(bp.clocks + 1, __builtin_amdgcn_s_memrealtime() - t0r);
        }
}

// second stage of the streamed decomposition: one workgroup of 2 waves per (split band, column tile,
// wave pair) sums the band's pieces in K order and applies the epilogue with the GEMM kernel's own
// register mapping
template <class Epi>
__global__ __launch_bounds__(128) void k_band_reduce(const BandRed* red, const double* part, int gn, int M, int N, Epi epi)
{
    const BandRed rd = red[blockIdx.x];
    const int bj = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = blockIdx.z * 2 + (threadIdx.x >> 6);
    d4 acc[5][1];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i][0] = d4{0.0, 0.0, 0.0, 0.0};
    // four pieces' loads in flight at a time (the kernel is latency-bound: a handful of pieces per band); the adds
    // stay in K order
    const size_t pstride = (size_t)gn * BD_TILE_ELEMS;
    const double* P0 = part + ((size_t)rd.s0 * gn + bj) * BD_TILE_ELEMS + (size_t)wave * (20 * 64) + lane;
    for (int s = rd.s0; s < rd.s1; s += 4) {
        double v[4][20];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double* P = P0 + (size_t)(s - rd.s0 + (s + u < rd.s1 ? u : 0)) * pstride;
#pragma unroll
            for (int t = 0; t < 20; ++t) v[u][t] = P[t * 64];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s + u < rd.s1) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][0][r] += v[u][i * 4 + r];
            }
    }
    epi(acc, rd.band * BD_BM, bj * BD_BN + wave * 16, lane, M, N, rd.band);
}

template <class Epi, bool DBG = false>
static inline int launch_gemm_band(hipStream_t s, BandPlan& plan, int N, const double* A, int lda,
                                   const double* B, int ldb, const Epi& epi,
                                   unsigned long long* clocks = nullptr, int mode = 0)
{
    const int gn = (N + BD_BN - 1) / BD_BN;
    BandPlanDev* d = nullptr;
    MCML_TRY(plan.device_plan(gn, s, &d));
    BandP bp;
    bp.clocks = clocks;
    bp.mode = mode;
    bp.g = GemmP{plan.M, N, plan.K, A, lda, B, ldb, 0, gn, 0, 0};
    bp.items = d->items.as<BandItem>();
    bp.wg_ptr = d->wg_ptr.as<int>();
    bp.nwg = d->nwg;
    bp.part = d->part.d();
    MCML_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&dgemm_band_kernel<Epi, DBG>), (int)BD_LDS_BYTES));
    hipLaunchKernelGGL((dgemm_band_kernel<Epi, DBG>), dim3(d->nwg * gn), dim3(512), BD_LDS_BYTES, s, bp, epi);
    if (d->nred > 0)
        hipLaunchKernelGGL((k_band_reduce<Epi>), dim3(d->nred, gn, 4), dim3(128), 0, s, d->red.as<BandRed>(),
                           d->part.d(), gn, plan.M, N, epi);
    MCML_HIP(hipGetLastError());
    return MCML_OK;
}


}  // namespace mcml
