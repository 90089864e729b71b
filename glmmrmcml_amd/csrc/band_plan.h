// band_plan.h -- host side of dgemm_band.h: tile constants and the work decomposition of the banded
// HMC products (paired bands, or the flattened (band, K tile) space cut into equal runs with a
// fixed-order second stage).  Kept apart from the kernels so that ctx.h stays light.
#pragma once
#include "common.h"
#include <algorithm>
#include <map>
#include <memory>

namespace mcml {

constexpr int BD_BM = 80, BD_BN = 128, BD_BK = 32, BD_STAGES = 3;
constexpr int BD_A_BYTES = BD_BK * BD_BM * 8;      // 20480: 20 chunks of 1 KiB
constexpr int BD_B_BYTES = BD_BK * BD_BN * 8;      // 32768: 32 chunks
constexpr int BD_STAGE_BYTES = BD_A_BYTES + BD_B_BYTES;
constexpr size_t BD_LDS_BYTES = (size_t)BD_STAGES * BD_STAGE_BYTES;   // 159744
constexpr int BD_NA = 3, BD_NB = 4, BD_PER_TILE = BD_NA + BD_NB;      // LDS-DMA pieces per wave per tile
constexpr int BD_TILE_ELEMS = BD_BM * BD_BN;                          // one raw accumulator tile (doubles)

// one run of K tiles of one band.  slot < 0: the band's only run, the epilogue is applied here;
// slot >= 0: the raw accumulator tile goes to partial slot `slot` of this column tile
struct BandItem { int band, kt0, kt1, slot; };
// a band whose pieces k_band_reduce sums: slots [s0, s1) in K order
struct BandRed { int band, s0, s1, pad; };

// ---- host: the decomposition ------------------------------------------------------------
struct BandPlanDev {
    int gn = 0, nwg = 0, nred = 0, nslots = 0;
    DevBuf items, wg_ptr, red, part;
};

// the few-column products (dgemm_skinny.h): chunk range of every 256-row block, partial sums
struct SkinnyPlan {
    int nrb = 0, nkc = 0, ldp = 0;
    DevBuf range;                 // int2 per row block: first chunk of 128 K columns, one past the last
    DevBuf part;                  // nkc x 16 x ldp doubles
};

struct BandPlan {
    int M = 0, K = 0, nbands = 0;
    std::vector<int> kr;                  // host copy of the K-tile ranges, 2 per band
    long tiles = 0;                       // sum over bands of K tiles multiplied
    std::map<int, std::unique_ptr<BandPlanDev>> by_gn;
    std::unique_ptr<SkinnyPlan> skinny;

    void reset(int M_, int K_, const std::vector<int>& kr_)
    {
        M = M_; K = K_; kr = kr_; nbands = (M + BD_BM - 1) / BD_BM;
        tiles = 0;
        for (int b = 0; b < nbands; ++b) tiles += kr[2 * b + 1] - kr[2 * b];
        by_gn.clear();
        skinny.reset();
    }

    // target_wg: workgroups the launch should have in all (one per CU)
    static void decompose(const std::vector<int>& kr, int nbands, int gn, int target_wg, std::vector<BandItem>& items,
                          std::vector<int>& wg_ptr, std::vector<BandRed>& red, int& nslots)
    {
        items.clear(); wg_ptr.clear(); red.clear(); nslots = 0;
        const int npairs = (nbands + 1) / 2;
        if ((long)npairs * gn * 5 >= (long)target_wg * 4) {          // paired fills >= 80% of the CUs: no partial sums
            for (int p = 0; p < npairs; ++p) {
                wg_ptr.push_back((int)items.size());
                items.push_back({p, kr[2 * p], kr[2 * p + 1], -1});
                const int q = nbands - 1 - p;
                if (q > p) items.push_back({q, kr[2 * q], kr[2 * q + 1], -1});
            }
            wg_ptr.push_back((int)items.size());
            return;
        }
        // Streamed: walk the bands in order and fill one workgroup after another up to a cost cap.  Cost in
        // half K tiles: 2 per K tile + OVH per item (ring fill, barrier, epilogue or partial store).  Per-workgroup
        // phase stamps (scripts/band_clocks.py 5000 128 1 48) put a second item at about one K tile: with OVH = 5
        // the two-item workgroups ended 8 us before the single-item ones and only 249 of 256 CUs were used at
        // n = 5000, 128 chains; OVH = 3 fills all 256 and the longest run drops from 21 to 20 K tiles.  The cap is
        // the smallest one whose greedy fill needs at most nwg workgroups (bisection).
        int nwg = target_wg / gn; if (nwg < 1) nwg = 1;
        constexpr long OVH = 3;
        auto fill = [&](long cap, bool emit) -> int {
            int used = 1; long room = cap;
            if (emit) wg_ptr.push_back(0);
            for (int b = 0; b < nbands; ++b) {
                int k0 = kr[2 * b]; const int k1 = kr[2 * b + 1];
                const int first = (int)items.size();
                int pieces = 0;
                do {
                    if (room < OVH + 2) { ++used; room = cap; if (emit) wg_ptr.push_back((int)items.size()); }
                    long take = (room - OVH) / 2; if (take > k1 - k0) take = k1 - k0;
                    if (emit) items.push_back({b, k0, k0 + (int)take, -1});
                    k0 += (int)take; room -= OVH + 2 * take; ++pieces;
                } while (k0 < k1);
                if (emit && pieces > 1) {
                    for (int t = 0; t < pieces; ++t) items[first + t].slot = nslots + t;
                    red.push_back({b, nslots, nslots + pieces, 0});
                    nslots += pieces;
                }
            }
            if (emit) wg_ptr.push_back((int)items.size());
            return used;
        };
        long T = 0;
        for (int b = 0; b < nbands; ++b) T += kr[2 * b + 1] - kr[2 * b];
        long lo = OVH + 2, hi = 2 * T + OVH * nbands + OVH + 2;      // hi: everything in one workgroup
        while (lo < hi) {
            const long mid = (lo + hi) / 2;
            if (fill(mid, false) <= nwg) hi = mid; else lo = mid + 1;
        }
        fill(lo, true);
        return;
    }

    int device_plan(int gn, hipStream_t s, BandPlanDev** out)
    {
        auto it = by_gn.find(gn);
        if (it != by_gn.end()) { *out = it->second.get(); return MCML_OK; }
        std::unique_ptr<BandPlanDev> d(new BandPlanDev());
        std::vector<BandItem> items; std::vector<int> wg_ptr; std::vector<BandRed> red; int nslots = 0;
        decompose(kr, nbands, gn, 256, items, wg_ptr, red, nslots);
        d->gn = gn; d->nwg = (int)wg_ptr.size() - 1; d->nred = (int)red.size(); d->nslots = nslots;
        MCML_TRY(d->items.ensure(sizeof(BandItem) * (items.size() + 1)));
        MCML_TRY(d->wg_ptr.ensure(sizeof(int) * wg_ptr.size()));
        MCML_HIP(hipMemcpyAsync(d->items.p, items.data(), sizeof(BandItem) * items.size(), hipMemcpyHostToDevice, s));
        MCML_HIP(hipMemcpyAsync(d->wg_ptr.p, wg_ptr.data(), sizeof(int) * wg_ptr.size(), hipMemcpyHostToDevice, s));
        if (nslots > 0) {
            MCML_TRY(d->red.ensure(sizeof(BandRed) * red.size()));
            MCML_HIP(hipMemcpyAsync(d->red.p, red.data(), sizeof(BandRed) * red.size(), hipMemcpyHostToDevice, s));
            MCML_TRY(d->part.ensure(sizeof(double) * (size_t)nslots * gn * BD_TILE_ELEMS));
        }
        MCML_HIP(hipStreamSynchronize(s));       // the host vectors go out of scope
        *out = d.get();
        by_gn[gn] = std::move(d);
        return MCML_OK;
    }
};

}  // namespace mcml
