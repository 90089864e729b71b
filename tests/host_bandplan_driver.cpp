// Host-side work decomposition of the banded HMC products (glmmrmcml_amd/csrc/band_plan.h) under
// AddressSanitizer + UBSan: for triangular, banded, dense and partly empty operands and 1..9 column tiles, every
// K tile of every band is covered exactly once and in order, slots are contiguous per band, single-piece bands
// carry no slot, the workgroup count stays within the target, and the costliest workgroup of the streamed cut is
// within a small factor of the mean.  Built and run by tests/test_host_sanitizers.py.
#include "band_plan.h"
#include <cstdio>
using namespace mcml;

static int check(const std::vector<int>& kr, int nbands, int gn, const char* what)
{
    std::vector<BandItem> items; std::vector<int> wg; std::vector<BandRed> red; int nslots = 0;
    BandPlan::decompose(kr, nbands, gn, 256, items, wg, red, nslots);
    int fails = 0;
    const int nwg = (int)wg.size() - 1;
    if (nwg < 1 || wg[0] != 0 || wg.back() != (int)items.size()) { printf("%s: bad wg_ptr\n", what); return 1; }
    if ((long)nwg * gn > 256 + gn && nwg > (nbands + 1) / 2) { printf("%s: %d workgroups x %d tiles\n", what, nwg, gn); ++fails; }
    std::vector<int> next(nbands), pieces(nbands, 0), lastslot(nbands, -2);
    for (int b = 0; b < nbands; ++b) next[b] = kr[2 * b];
    long maxcost = 0, totcost = 0;
    for (int w = 0; w < nwg; ++w) {
        if (wg[w + 1] < wg[w]) { printf("%s: wg_ptr not monotone\n", what); return 1; }
        long cost = 0;
        for (int it = wg[w]; it < wg[w + 1]; ++it) {
            const BandItem& x = items[it];
            if (x.band < 0 || x.band >= nbands || x.kt0 > x.kt1) { ++fails; continue; }
            cost += 5 + 2 * (x.kt1 - x.kt0);
            ++pieces[x.band];
        }
        maxcost = std::max(maxcost, cost); totcost += cost;
    }
    // coverage in order: the items of a band appear in increasing K order across the list (paired mode lists band p
    // then its mirror; streamed mode walks the bands in order)
    std::vector<std::vector<BandItem>> per(nbands);
    for (const BandItem& x : items) per[x.band].push_back(x);
    for (int b = 0; b < nbands; ++b) {
        int k = kr[2 * b];
        if (per[b].empty()) { printf("%s: band %d has no item (its epilogue would not run)\n", what, b); ++fails; continue; }
        for (size_t t = 0; t < per[b].size(); ++t) {
            if (per[b][t].kt0 != k) { printf("%s: band %d gap at %d\n", what, b, k); ++fails; }
            k = per[b][t].kt1;
            const int want = per[b].size() == 1 ? -1 : (t == 0 ? per[b][0].slot : per[b][t - 1].slot + 1);
            if (per[b][t].slot != want || (per[b].size() > 1 && per[b][t].slot < 0)) { printf("%s: band %d slot\n", what, b); ++fails; }
        }
        if (k != kr[2 * b + 1]) { printf("%s: band %d ends at %d, not %d\n", what, b, k, kr[2 * b + 1]); ++fails; }
    }
    int nred = 0, slots = 0;
    for (int b = 0; b < nbands; ++b) if (per[b].size() > 1) { ++nred; slots += (int)per[b].size(); }
    if (nred != (int)red.size() || slots != nslots) { printf("%s: red list %d/%d slots %d/%d\n", what, (int)red.size(), nred, nslots, slots); ++fails; }
    for (const BandRed& r : red)
        if (r.s1 - r.s0 != (int)per[r.band].size() || per[r.band][0].slot != r.s0) { printf("%s: red entry of band %d\n", what, r.band); ++fails; }
    if (!red.empty() && nwg > 8 && maxcost * nwg > 3 * totcost) { printf("%s: imbalance max %ld mean %.1f\n", what, maxcost, (double)totcost / nwg); ++fails; }
    return fails;
}

int main()
{
    int fails = 0;
    for (int M : {80, 333, 1000, 2000, 5000, 20000}) {
        const int nbands = (M + BD_BM - 1) / BD_BM, ktiles = (M + BD_BK - 1) / BD_BK;
        std::vector<int> lower(2 * nbands), upper(2 * nbands), dense(2 * nbands), holes(2 * nbands);
        for (int b = 0; b < nbands; ++b) {
            const int last = std::min(M, (b + 1) * BD_BM) - 1;
            lower[2 * b] = 0; lower[2 * b + 1] = last / BD_BK + 1;
            upper[2 * b] = (b * BD_BM) / BD_BK; upper[2 * b + 1] = ktiles;
            dense[2 * b] = 0; dense[2 * b + 1] = ktiles;
            holes[2 * b] = (b % 3 == 1) ? 0 : upper[2 * b]; holes[2 * b + 1] = (b % 3 == 1) ? 0 : std::min(ktiles, upper[2 * b] + 7);
        }
        for (int gn = 1; gn <= 9; ++gn) {
            char tag[64];
            snprintf(tag, sizeof tag, "lower M=%d gn=%d", M, gn); fails += check(lower, nbands, gn, tag);
            snprintf(tag, sizeof tag, "upper M=%d gn=%d", M, gn); fails += check(upper, nbands, gn, tag);
            snprintf(tag, sizeof tag, "dense M=%d gn=%d", M, gn); fails += check(dense, nbands, gn, tag);
            snprintf(tag, sizeof tag, "holes M=%d gn=%d", M, gn); fails += check(holes, nbands, gn, tag);
        }
    }
    printf("fails=%d\n", fails);
    return fails ? 1 : 0;
}
