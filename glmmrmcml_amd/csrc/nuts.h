// nuts.h -- a No-U-Turn sampler for the random effects, standing where the reference calls Stan
// (R/gen_u_samples.R:38-69, R6ModelExtMCML.R:234-257: cmdstanr, inst/stan/mcml_*.stan).  Included at the end of
// hmc.hip: it drives the same two products (hmc_forward / hmc_backward) and the same log-density kernels.
//
// The target is the one the Stan programs declare: gamma ~ std_normal(), y ~ family(Xb + (Z L) gamma) -- log_prob of
// mcmlmodel.h:138-153.  The transition is Stan's base_nuts (multinomial NUTS, Betancourt 2017) restated for C chains in
// lock step:
//   * one tree per chain and iteration, grown by doubling in a random direction; every chain that is still growing is
//     at the same (doubling j, leaf n), so ONE batched leapfrog (two products over all chains) extends every tree;
//   * a leaf has weight exp(H0 - H); a chain whose leaf has H - H0 > 1000 is divergent and stops;
//   * balanced subtrees are merged like the carries of a binary counter: after leaf n, one merge per trailing one bit
//     of n.  A merge adds the momentum sums (rho), keeps the earlier-built begin and the later-built end, picks the later subtree's
//     proposal with probability w_right / (w_left + w_right), and applies the generalised no-U-turn criterion
//     p_begin . rho > 0 && p_end . rho > 0 ; a subtree that turns invalidates the doubling and stops the chain;
//   * a completed doubling replaces the tree's proposal with probability min(1, w_subtree / w_tree) (biased
//     progressive sampling), then the criterion is applied to the whole tree;
//   * step size: Stan's dual averaging on the mean of min(1, exp(H0 - H)) over the leaves (delta, gamma = 0.05,
//     t0 = 10, kappa = 0.75, mu = log(10 eps0)) during warm-up, eps = exp(xbar) after; eps0 from Stan's doubling /
//     halving heuristic (init_stepsize).
//   * metric: Stan's default `diag_e` with its windowed adaptation (windowed_adaptation / var_adaptation: Welford
//     variance of every coordinate over doubling windows between init_buffer and term_buffer, regularised
//     (n / (n + 5)) var + 1e-3 * 5 / (n + 5); at the end of a window the step size is searched again and the dual
//     averaging restarts with mu = log(10 eps)); every chain adapts its own diagonal.  `metric = 1` keeps unit_e.
//   * every merge (and the tree itself after every doubling) also applies Stan's checks BETWEEN the two subtrees
//     (base_nuts.hpp since 2.23): rho_left + p_first(right) against the sharp momenta of left's begin and right's begin,
//     and rho_right + p_last(left) against left's end and right's end.
//   * initial values: uniform(-2, 2) on every coordinate, Stan's default (`init = 2`).
// Not reproduced: Stan's RNG (the Philox / minstd streams of the HMC sampler stand in), its retries when the
// initial log density is not finite.  cmdstan
// does not exist in this image: parity of this sampler is UNPINNED; it is checked against oracle/nuts.py (same
// algorithm, same streams: every tree depth, leapfrog count and divergence identical) and against the exact
// posterior of the gaussian model.
//
// State: every vector is a Q x C (or chain-major C x Q) matrix like the HMC state.  Edges (theta, r, grad) x 2, the
// tree's rho and proposal, a node under construction (rho, p_begin, proposal) and one stored node per level.  All
// growing chains share the merge schedule, so "push the node under construction on level l" is a host-side swap of
// array pointers.
#pragma once

namespace mcml {

struct NutsChain {                                  // per-chain scalars (arrays of Cp)
    double *eps, *H0, *lw_tree, *lw_stack, *lp_new, *sum_acc, *xbar, *sbar, *mu, *dH;
    int *active, *valid, *dir, *depth, *nleap, *counter, *ndiv, *accsub, *was, *hdir, *hdone, *nhit, *slot, *ndsave;
    uint8_t* choose;                                // [NUTS_MAXD][Cp]
    uint32_t* gen;
    int Cp;
};

static size_t nuts_chain_bytes(int C)
{
    const size_t Cp = (size_t)round_up(C, 64);
    return sizeof(double) * Cp * (10 + NUTS_MAXD + 1) + sizeof(int) * Cp * 15 + (size_t)NUTS_MAXD * Cp + 256;
}
static NutsChain nuts_chain(void* base, int C)
{
    NutsChain a;
    const size_t Cp = (size_t)round_up(C, 64);
    double* d = static_cast<double*>(base);
    a.eps = d; a.H0 = d + Cp; a.lw_tree = d + 2 * Cp; a.lp_new = d + 3 * Cp; a.sum_acc = d + 4 * Cp;
    a.xbar = d + 5 * Cp; a.sbar = d + 6 * Cp; a.mu = d + 7 * Cp; a.dH = d + 8 * Cp;
    a.lw_stack = d + 10 * Cp;                                       // (NUTS_MAXD + 1) x Cp
    int* i = reinterpret_cast<int*>(d + (10 + NUTS_MAXD + 1) * Cp);
    a.active = i; a.valid = i + Cp; a.dir = i + 2 * Cp; a.depth = i + 3 * Cp; a.nleap = i + 4 * Cp;
    a.counter = i + 5 * Cp; a.ndiv = i + 6 * Cp; a.accsub = i + 7 * Cp; a.was = i + 8 * Cp; a.hdir = i + 9 * Cp;
    a.hdone = i + 10 * Cp; a.nhit = i + 11 * Cp; a.slot = i + 12 * Cp; a.ndsave = i + 13 * Cp;
    a.gen = reinterpret_cast<uint32_t*>(i + 14 * Cp);
    a.choose = reinterpret_cast<uint8_t*>(i + 15 * Cp);
    a.Cp = (int)Cp;
    return a;
}

// log(exp(a) + exp(b)), Stan's log_sum_exp(a, b)
__host__ __device__ inline double nuts_logaddexp(double a, double b)
{
    if (a == -INFINITY) return b;
    if (a == INFINITY && b == INFINITY) return INFINITY;
    if (a > b) return a + log1p(exp(b - a));
    return b + log1p(exp(a - b));
}

// ---- vector kernels: a workgroup = 256 x SB elements of the (fast, slow) index space; fast = random effect for the
// column-major state, chain for the chain-major state
template <bool CM> struct NutsTile { static constexpr int SB = CM ? 16 : 4; };

#define NUTS_IDX()                                                           \
    constexpr int SB = NutsTile<CM>::SB;                                     \
    const int f = blockIdx.x * 256 + threadIdx.x, s0 = blockIdx.y * SB;      \
    const int F = CM ? C : Q, S = CM ? Q : C;                                \
    const bool fin = f < F;                                                  \
    (void)S

// per-chain partial sums: part[k][chunk][chain]; chunk = slow tile (chain-major) or fast block (column-major)
template <bool CM, int NV>
__device__ __forceinline__ void nuts_store_partials(double (&acc)[NV][NutsTile<CM>::SB], double* part, size_t pstride,
                                                    int ldp, int C)
{
    constexpr int SB = NutsTile<CM>::SB;
    if (CM) {
        const int f = blockIdx.x * 256 + threadIdx.x;
        if (f < C)
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                double t = 0.0;
#pragma unroll
                for (int u = 0; u < SB; ++u) t += acc[k][u];
                part[k * pstride + (size_t)blockIdx.y * ldp + f] = t;
            }
    } else {
        __shared__ double sh[4];
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const double r = block_sum(acc[k][u], sh);
                const int s = blockIdx.y * SB + u;
                if (threadIdx.x == 0 && s < C) part[k * pstride + (size_t)blockIdx.x * ldp + s] = r;
            }
    }
}

struct NutsVecs {                                   // device pointers, all with the leading dimension of the state
    double *TM, *RM, *GM, *TP, *RP, *GP;            // backward / forward edge: position, momentum, gradient
    double *Trho, *Tth;                             // the tree: sum of momenta, proposal
    double *Crho, *Cpb, *Cpe, *Cth;                 // node under construction: rho, momentum of its first- / last-built leaf, proposal
    double *Mi, *Wm, *Ws;                           // inverse metric (diagonal, per chain); Welford mean and sum of squares
    double *Tadj;                                   // momentum of the tree's edge the current doubling grows from (before it)
};

// iteration start: fresh momentum, one-node tree
template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_begin(const double* V, const double* GRAD, NutsVecs nv, int ld, int Q, int C,
                                                    uint64_t seed, uint32_t chain_offset, uint32_t it, uint32_t stream,
                                                    double* part, size_t pstride, int ldp)
{
    NUTS_IDX();
    double acc[1][SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        acc[0][u] = 0.0;
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const int ch = CM ? f : s, q = CM ? s : f;
        const size_t off = f + (size_t)s * ld;
        const double mi = nv.Mi[off];
        const double r = rng_normal(seed, (uint32_t)q, chain_offset + (uint32_t)ch, it, stream) / sqrt(mi);   // diag_e sample_p
        const double v = V[off], g = GRAD[off];
        nv.TM[off] = v; nv.TP[off] = v; nv.Tth[off] = v;
        nv.RM[off] = r; nv.RP[off] = r; nv.Trho[off] = r;
        nv.GM[off] = g; nv.GP[off] = g;
        acc[0][u] = r * (mi * r);
    }
    nuts_store_partials<CM, 1>(acc, part, pstride, ldp, C);
}

// first half of a leapfrog step from the edge the chain grows: WR = r + (es / 2) g ; WX = theta + es WR, es = dir * eps.
// WX (and with it MU, S, the new gradient and the new log density) lives in the chain's PACKED column nc.slot: only
// the chains whose trees still grow take part in the two products
template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_leap_pre(NutsVecs nv, double* WX, double* WR, int ld, int Q, int C, NutsChain nc)
{
    NUTS_IDX();
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const int ch = CM ? f : s;
        if (!nc.active[ch]) continue;
        const size_t off = f + (size_t)s * ld;
        const int d = nc.dir[ch];
        const double es = d * nc.eps[ch];
        const double th = d > 0 ? nv.TP[off] : nv.TM[off], r = d > 0 ? nv.RP[off] : nv.RM[off];
        const double g = d > 0 ? nv.GP[off] : nv.GM[off];
        const double rh = r + (0.5 * es) * g;
        WR[off] = rh;
        const int sl = nc.slot[ch];                                // the products run on the packed columns
        WX[CM ? sl + (size_t)s * ld : f + (size_t)sl * ld] = th + es * (nv.Mi[off] * rh);
    }
}

// second half: r' = WR + (es / 2) grad(WX); the new state becomes the edge and a one-leaf node under construction
template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_leap_post(const double* WX, const double* WR, const double* GN, NutsVecs nv,
                                                        int ld, int Q, int C, NutsChain nc, double* part, size_t pstride,
                                                        int ldp)
{
    NUTS_IDX();
    double acc[1][SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        acc[0][u] = 0.0;
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const int ch = CM ? f : s;
        if (!nc.active[ch]) continue;
        const size_t off = f + (size_t)s * ld;
        const int d = nc.dir[ch];
        const double es = d * nc.eps[ch];
        const int sl = nc.slot[ch];
        const size_t offc = CM ? sl + (size_t)s * ld : f + (size_t)sl * ld;
        const double g = GN[offc], x = WX[offc];
        const double rn = WR[off] + (0.5 * es) * g;
        if (d > 0) { nv.TP[off] = x; nv.RP[off] = rn; nv.GP[off] = g; }
        else { nv.TM[off] = x; nv.RM[off] = rn; nv.GM[off] = g; }
        nv.Crho[off] = rn; nv.Cpb[off] = rn; nv.Cpe[off] = rn; nv.Cth[off] = x;
        acc[0][u] = rn * (nv.Mi[off] * rn);
    }
    nuts_store_partials<CM, 1>(acc, part, pstride, ldp, C);
}

// merge the stored node of a level (built earlier: "left") with the node under construction (built later: "right").
// Six dot products: around the merged subtree, and the two checks between the subtrees.
template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_merge(const double* Srho, const double* Spb, const double* Spe, const double* Sth,
                                                    NutsVecs nv, const uint8_t* choose, int ld, int Q, int C, NutsChain nc,
                                                    double* part, size_t pstride, int ldp)
{
    NUTS_IDX();
    double acc[6][SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k][u] = 0.0;
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const int ch = CM ? f : s;
        if (!nc.active[ch]) continue;
        const size_t off = f + (size_t)s * ld;
        const double mi = nv.Mi[off];
        const double lrho = Srho[off], rrho = nv.Crho[off];
        const double lpb = Spb[off], lpe = Spe[off], rpb = nv.Cpb[off], rpe = nv.Cpe[off];
        const double rho = lrho + rrho;
        nv.Crho[off] = rho; nv.Cpb[off] = lpb;                      // the end stays the right subtree's
        if (!choose[ch]) nv.Cth[off] = Sth[off];
        acc[0][u] = (mi * lpb) * rho; acc[1][u] = (mi * rpe) * rho;
        const double e1 = lrho + rpb;                              // rho_left + first momentum of the right subtree
        acc[2][u] = (mi * lpb) * e1; acc[3][u] = (mi * rpb) * e1;
        const double e2 = rrho + lpe;                              // rho_right + last momentum of the left subtree
        acc[4][u] = (mi * lpe) * e2; acc[5][u] = (mi * rpe) * e2;
    }
    nuts_store_partials<CM, 6>(acc, part, pstride, ldp, C);
}

// the edge a doubling grows from, before it grows
template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_save_adj(NutsVecs nv, int ld, int Q, int C, NutsChain nc)
{
    NUTS_IDX();
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const int ch = CM ? f : s;
        if (!nc.active[ch]) continue;
        const size_t off = f + (size_t)s * ld;
        nv.Tadj[off] = nc.dir[ch] > 0 ? nv.RP[off] : nv.RM[off];
    }
}

// a completed doubling (stored node of level j: rho, first-built momentum) joins the tree; the same three checks
template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_tree_update(const double* Srho, const double* Spb, const double* Sth, NutsVecs nv,
                                                          int ld, int Q, int C, NutsChain nc, double* part, size_t pstride,
                                                          int ldp)
{
    NUTS_IDX();
    double acc[6][SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k][u] = 0.0;
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const int ch = CM ? f : s;
        if (!(nc.was[ch] && nc.valid[ch])) continue;
        const size_t off = f + (size_t)s * ld;
        const double mi = nv.Mi[off];
        const double rho_old = nv.Trho[off], rho_sub = Srho[off];
        const double rho = rho_old + rho_sub;
        nv.Trho[off] = rho;
        if (nc.accsub[ch]) nv.Tth[off] = Sth[off];
        const double rm = nv.RM[off], rp = nv.RP[off];
        acc[0][u] = (mi * rm) * rho; acc[1][u] = (mi * rp) * rho;
        const bool fwd = nc.dir[ch] > 0;
        const double far_old = fwd ? rm : rp, new_edge = fwd ? rp : rm, pb = Spb[off], padj = nv.Tadj[off];
        const double e1 = rho_old + pb;                            // the old tree + the subtree's first momentum
        acc[2][u] = (mi * far_old) * e1; acc[3][u] = (mi * pb) * e1;
        const double e2 = rho_sub + padj;                          // the subtree + the old tree's adjacent edge
        acc[4][u] = (mi * padj) * e2; acc[5][u] = (mi * new_edge) * e2;
    }
    nuts_store_partials<CM, 6>(acc, part, pstride, ldp, C);
}

template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_commit(const double* Tth, double* V, int ld, int Q, int C)
{
    NUTS_IDX();
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const size_t off = f + (size_t)s * ld;
        V[off] = Tth[off];
    }
}

// Stan's default initial values: uniform(-2, 2)
template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_init_uniform(double* V, int ld, int Q, int C, uint64_t seed, uint32_t chain_offset,
                                                           uint32_t stream)
{
    NUTS_IDX();
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const int ch = CM ? f : s, q = CM ? s : f;
        V[f + (size_t)s * ld] = 4.0 * rng_uniform(seed, (uint32_t)q, chain_offset + (uint32_t)ch, 0u, stream) - 2.0;
    }
}

// metric adaptation (var_adaptation): Welford update with the state after a transition; end of a window
template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_welford(const double* V, NutsVecs nv, int ld, int Q, int C, double nsamp)
{
    NUTS_IDX();
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const size_t off = f + (size_t)s * ld;
        const double q = V[off];
        double m = nv.Wm[off];
        const double delta = q - m;
        m = m + delta / nsamp;
        nv.Wm[off] = m;
        nv.Ws[off] = nv.Ws[off] + (q - m) * delta;
    }
}

template <bool CM>
__global__ __launch_bounds__(256) void k_nuts_metric(NutsVecs nv, int ld, int Q, int C, double nsamp, int reset_only)
{
    NUTS_IDX();
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int s = s0 + u;
        if (!fin || s >= S) continue;
        const size_t off = f + (size_t)s * ld;
        if (reset_only) nv.Mi[off] = 1.0;
        else {
            const double var = nv.Ws[off] / (nsamp - 1.0);
            nv.Mi[off] = (nsamp / (nsamp + 5.0)) * var + 1e-3 * (5.0 / (nsamp + 5.0));
        }
        nv.Wm[off] = 0.0; nv.Ws[off] = 0.0;
    }
}

// ---- per-chain kernels: a 256-thread workgroup = 64 chains (cm_sum_chunks adds the partial sums), wave 0 decides
__global__ __launch_bounds__(256) void k_nuts_chain_init(int C, NutsChain nc, double eps0, uint64_t seed, uint32_t chain_offset,
                                                         uint32_t iter_idx)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    nc.eps[c] = eps0; nc.counter[c] = 0; nc.sbar[c] = 0.0; nc.xbar[c] = 0.0; nc.mu[c] = log(10 * eps0);
    nc.ndiv[c] = 0; nc.nhit[c] = 0; nc.hdone[c] = 0; nc.hdir[c] = 0; nc.ndsave[c] = 0;
    nc.gen[c] = chain_minstd_seed(seed, chain_offset + (uint32_t)c, iter_idx);
}

__global__ __launch_bounds__(256) void k_nuts_begin_fin(const double* part, int nchunk, int ldp, int C, NutsChain nc,
                                                        const double* lpcur)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const double kin = cm_sum_chunks(part, nchunk, ldp, c < C ? c : 0);
    if (threadIdx.x >= 64 || c >= C) return;
    nc.H0[c] = -1 * lpcur[c] + 0.5 * kin;
    nc.lw_tree[c] = 0.0; nc.active[c] = 1; nc.valid[c] = 1; nc.depth[c] = 0; nc.sum_acc[c] = 0.0; nc.nleap[c] = 0;
    nc.was[c] = 0; nc.accsub[c] = 0;
}

__global__ __launch_bounds__(256) void k_nuts_begin_doubling(int C, NutsChain nc, int force_forward)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    nc.was[c] = nc.active[c];
    nc.accsub[c] = 0;
    if (!nc.active[c]) return;
    if (force_forward) { nc.dir[c] = 1; nc.valid[c] = 1; return; }
    uint32_t g = nc.gen[c];
    const double u = minstd_canonical(g);
    nc.gen[c] = g;
    nc.dir[c] = u > 0.5 ? 1 : -1;                                  // base_nuts: go_forward = rand_uniform() > 0.5
    nc.valid[c] = 1;
}

// a leaf: its energy, divergence, weight; the draws of the merges this leaf triggers (levels 0 .. tz-1)
__global__ __launch_bounds__(256) void k_nuts_leaf_fin(const double* part, int nchunk, int ldp, int C, NutsChain nc, int tz)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const double kin = cm_sum_chunks(part, nchunk, ldp, c < C ? c : 0);
    if (threadIdx.x >= 64 || c >= C || !nc.active[c]) return;
    double h = -1 * nc.lp_new[nc.slot[c]] + 0.5 * kin;
    if (isnan(h)) h = INFINITY;
    const double H0 = nc.H0[c];
    nc.nleap[c] += 1;
    nc.dH[c] = H0 - h;
    if ((h - H0) > 1000.0) {                                       // max_deltaH: divergent, the tree stops here
        nc.ndiv[c] += 1; nc.valid[c] = 0; nc.active[c] = 0;
        return;
    }
    nc.sum_acc[c] += (H0 - h > 0) ? 1.0 : exp(H0 - h);
    double lw = H0 - h;
    uint32_t g = nc.gen[c];
    for (int l = 0; l < tz; ++l) {
        const double lwm = nuts_logaddexp(nc.lw_stack[(size_t)l * nc.Cp + c], lw);
        const double u = minstd_canonical(g);
        nc.choose[(size_t)l * nc.Cp + c] = (uint8_t)(u < exp(lw - lwm));
        lw = lwm;
    }
    nc.gen[c] = g;
    nc.lw_stack[(size_t)tz * nc.Cp + c] = lw;
}

__global__ __launch_bounds__(256) void k_nuts_merge_fin(const double* part, size_t pstride, int nchunk, int ldp, int C,
                                                        NutsChain nc)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cc = c < C ? c : 0;
    bool ok = true;
    for (int k = 0; k < 6; ++k) ok = (cm_sum_chunks(part + k * pstride, nchunk, ldp, cc) > 0) && ok;
    if (threadIdx.x >= 64 || c >= C || !nc.active[c]) return;
    if (!ok) { nc.valid[c] = 0; nc.active[c] = 0; }
}

__global__ __launch_bounds__(256) void k_nuts_end_doubling(int C, NutsChain nc, int j)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (!(nc.was[c] && nc.valid[c])) return;
    nc.depth[c] = j + 1;
    const double lws = nc.lw_stack[(size_t)j * nc.Cp + c], lwt = nc.lw_tree[c];
    int acc;
    if (lws > lwt) acc = 1;
    else {
        uint32_t g = nc.gen[c];
        const double u = minstd_canonical(g);
        nc.gen[c] = g;
        acc = u < exp(lws - lwt);
    }
    nc.accsub[c] = acc;
    nc.lw_tree[c] = nuts_logaddexp(lwt, lws);
}

__global__ __launch_bounds__(256) void k_nuts_tree_fin(const double* part, size_t pstride, int nchunk, int ldp, int C,
                                                       NutsChain nc, int max_depth)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cc = c < C ? c : 0;
    bool ok = true;
    for (int k = 0; k < 6; ++k) ok = (cm_sum_chunks(part + k * pstride, nchunk, ldp, cc) > 0) && ok;
    if (threadIdx.x >= 64 || c >= C) return;
    if (!(nc.was[c] && nc.valid[c])) return;
    if (!ok) nc.active[c] = 0;
    if (nc.depth[c] >= max_depth) { if (nc.active[c]) nc.nhit[c] += 1; nc.active[c] = 0; }
}

__global__ void k_nuts_count(const int* flag, int C, int* out)
{
    __shared__ int sh[256];
    int v = 0;
    for (int i = threadIdx.x; i < C; i += 256) v += flag[i] ? 1 : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// packed column of every growing chain (ascending chain order), -1 for the others; one 256-thread workgroup
__global__ __launch_bounds__(256) void k_nuts_slots(const int* active, int C, int* slot, int identity)
{
    __shared__ int cnt[256];
    const int per = (C + 255) / 256, c0 = threadIdx.x * per, c1 = (c0 + per < C) ? c0 + per : C;
    int n = 0;
    for (int c = c0; c < c1; ++c) n += (identity || active[c]) ? 1 : 0;
    cnt[threadIdx.x] = n;
    __syncthreads();
    int base = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) base += cnt[t];
    for (int c = c0; c < c1; ++c) slot[c] = (identity || active[c]) ? base++ : -1;
}

// end of a transition: stepsize_adaptation::learn_stepsize / complete_adaptation
__global__ __launch_bounds__(256) void k_nuts_end_iter(int C, NutsChain nc, int adapt, int last_warm, double delta, int it,
                                                       int* depth_out, int* nleap_out, double* eps_out, double* acc_out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double eps_used = nc.eps[c];
    double stat = nc.nleap[c] > 0 ? nc.sum_acc[c] / nc.nleap[c] : 0.0;
    if (adapt) {
        const int cnt = nc.counter[c] + 1;
        nc.counter[c] = cnt;
        if (stat > 1) stat = 1;
        const double eta = 1.0 / (cnt + 10.0);                     // t0 = 10
        const double sbar = (1.0 - eta) * nc.sbar[c] + eta * (delta - stat);
        nc.sbar[c] = sbar;
        const double x = nc.mu[c] - sbar * sqrt((double)cnt) / 0.05;   // gamma = 0.05
        const double xeta = pow((double)cnt, -0.75);               // kappa = 0.75
        const double xbar = (1.0 - xeta) * nc.xbar[c] + xeta * x;
        nc.xbar[c] = xbar;
        nc.eps[c] = last_warm ? exp(xbar) : exp(x);
    }
    if (depth_out) depth_out[c + (size_t)it * C] = nc.depth[c];
    if (nleap_out) nleap_out[c + (size_t)it * C] = nc.nleap[c];
    if (eps_out) eps_out[c + (size_t)it * C] = eps_used;
    if (acc_out) acc_out[c + (size_t)it * C] = stat;
}

// init_stepsize (base_hmc.hpp): after a one-step trajectory with fresh momentum, delta_H = H0 - h against log(0.8)
// gives the direction (first round) or ends the search; otherwise eps doubles / halves
__global__ __launch_bounds__(256) void k_nuts_heur_fin(int C, NutsChain nc, int round)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (nc.hdone[c]) { nc.active[c] = 0; return; }
    // dH holds H0 - h of the leaf just taken (a divergent leaf left it too: -inf / very negative)
    const double dH = nc.dH[c];
    const double thr = log(0.8);
    if (round == 0) {
        nc.hdir[c] = dH > thr ? 1 : -1;
    } else {
        const int d = nc.hdir[c];
        if ((d == 1 && !(dH > thr)) || (d == -1 && !(dH < thr))) { nc.hdone[c] = 1; return; }
        double e = nc.eps[c];
        e = d == 1 ? 2 * e : 0.5 * e;
        if (e > 1e7 || e < 1e-300) { nc.hdone[c] = 1; return; }  // Stan throws here; keep the last usable value
        nc.eps[c] = e;
    }
}

__global__ __launch_bounds__(256) void k_nuts_heur_reset(int C, NutsChain nc)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    nc.hdone[c] = 0; nc.hdir[c] = 0;
    nc.ndsave[c] = nc.ndiv[c];                                     // the search's own divergent leaves do not count
}
__global__ __launch_bounds__(256) void k_nuts_heur_done(int C, NutsChain nc)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    nc.mu[c] = log(10 * nc.eps[c]);                                // stepsize_adaptation::set_mu
    nc.counter[c] = 0; nc.sbar[c] = 0.0; nc.xbar[c] = 0.0;         // stepsize_adaptation::restart
    nc.ndiv[c] = nc.ndsave[c];
}

__global__ void k_nuts_diag(NutsChain nc, int C, double* out)
{
    // out: [0] sum eps, [1] min eps, [2] max eps, [3] divergent transitions, [4] transitions that hit max depth
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double se = 0, mn = 1e300, mx = 0, nd = 0, nh = 0;
    for (int c = 0; c < C; ++c) {
        se += nc.eps[c]; mn = fmin(mn, nc.eps[c]); mx = fmax(mx, nc.eps[c]); nd += nc.ndiv[c]; nh += nc.nhit[c];
    }
    out[0] = se; out[1] = mn; out[2] = mx; out[3] = nd; out[4] = nh;
}

// ---- host -------------------------------------------------------------------------------------------------------
struct NutsRun {
    Ctx& c; HmcState& h; NutsState& ns; NutsChain nc; NutsVecs nv;
    double* Srho[NUTS_MAXD + 1]; double* Spb[NUTS_MAXD + 1]; double* Spe[NUTS_MAXD + 1]; double* Sth[NUTS_MAXD + 1];
    int C, Q, ld, nchunk, ldp; size_t pstride; double* part; dim3 vgrid; double var_par;
    uint64_t seed; uint32_t chain_offset, iter_idx;
    long long leapfrogs = 0;

    template <class K, class... A> void vec(K k, A... a) { hipLaunchKernelGGL(k, vgrid, dim3(256), 0, c.stream, a...); }

    // gradient and log density of WX = h.UP for every chain: GRADP, nc.lp_new
    int eval_new()
    {
        MCML_TRY(hmc_forward(c, h.UP.d(), h.UP.ld, var_par, true));
        if (h.cm) {
            const CmParts p = cm_parts(c);
            MCML_TRY(cm_logprob_partials(c, h.UP.d(), nullptr, var_par));
            hipLaunchKernelGGL(k_cm_lp0_fin, dim3((h.Cw + 63) / 64), dim3(256), 0, c.stream, p.ll, p.lp, p.nchn, p.nchq, p.ldp,
                               h.Cw, nc.lp_new);
        } else
            MCML_FL_DISPATCH(c.flink, k_hmc_lp0, dim3(h.Cw), dim3(256), 0, c.stream, h.MU.d(), h.MU.ld, c.n, h.UP.d(), h.UP.ld, Q,
                               c.y.d(), var_par, c.flink, nc.lp_new);
        MCML_HIP(hipGetLastError());
        return hmc_backward(c, h.UP.d(), h.GRADP.d(), 0, var_par, 0);
    }
    // one leapfrog step of every growing chain, leaf bookkeeping with tz merges to follow
    int leaf(int tz)
    {
        if (h.cm) vec(k_nuts_leap_pre<true>, nv, h.UP.d(), h.R.d(), ld, Q, C, nc);
        else vec(k_nuts_leap_pre<false>, nv, h.UP.d(), h.R.d(), ld, Q, C, nc);
        MCML_TRY(eval_new());
        if (h.cm) vec(k_nuts_leap_post<true>, h.UP.d(), h.R.d(), h.GRADP.d(), nv, ld, Q, C, nc, part, pstride, ldp);
        else vec(k_nuts_leap_post<false>, h.UP.d(), h.R.d(), h.GRADP.d(), nv, ld, Q, C, nc, part, pstride, ldp);
        hipLaunchKernelGGL(k_nuts_leaf_fin, dim3((C + 63) / 64), dim3(256), 0, c.stream, part, nchunk, ldp, C, nc, tz);
        MCML_HIP(hipGetLastError());
        ++leapfrogs;
        return MCML_OK;
    }
    // pack the growing chains (nact of them) into the first columns the products process
    int pack(int nact)
    {
        hipLaunchKernelGGL(k_nuts_slots, dim3(1), dim3(256), 0, c.stream, nc.active, C, nc.slot, nact >= C ? 1 : 0);
        MCML_HIP(hipGetLastError());
        const int g = h.cm ? 64 : 128;                                 // a wave of chains / a column tile of the GEMMs
        int cw = round_up(nact, g);
        if (!h.cm && nact <= SK_NUSE) cw = nact;                       // few enough for the streamed products (dgemm_skinny.h)
        h.Cw = cw < C ? cw : C;
        return MCML_OK;
    }
    int begin(uint32_t it, uint32_t stream)
    {
        h.Cw = C;
        MCML_TRY(hmc_eval_state(c, var_par));                          // lpcur, GRAD at V
        ChainArrays ca = chain_arrays(h);
        if (h.cm) vec(k_nuts_begin<true>, h.V.d(), h.GRAD.d(), nv, ld, Q, C, seed, chain_offset, it, stream, part, pstride, ldp);
        else vec(k_nuts_begin<false>, h.V.d(), h.GRAD.d(), nv, ld, Q, C, seed, chain_offset, it, stream, part, pstride, ldp);
        hipLaunchKernelGGL(k_nuts_begin_fin, dim3((C + 63) / 64), dim3(256), 0, c.stream, part, nchunk, ldp, C, nc, ca.lpcur);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
    int count_active(int* out)
    {
        int* d = c.scalars.as<int>() + 40;
        hipLaunchKernelGGL(k_nuts_count, dim3(1), dim3(256), 0, c.stream, nc.active, C, d);
        MCML_TRY(copy_d2h(out, d, sizeof(int), c.stream));
        MCML_HIP(hipStreamSynchronize(c.stream));
        return MCML_OK;
    }
    // one transition of every chain
    int transition(uint32_t it, int max_depth)
    {
        MCML_TRY(begin(it, 16u * iter_idx + 4u));
        int nact = C;
        for (int j = 0; j < max_depth; ++j) {
            MCML_TRY(pack(nact));
            hipLaunchKernelGGL(k_nuts_begin_doubling, dim3((C + 255) / 256), dim3(256), 0, c.stream, C, nc, 0);
            if (h.cm) vec(k_nuts_save_adj<true>, nv, ld, Q, C, nc);
            else vec(k_nuts_save_adj<false>, nv, ld, Q, C, nc);
            const int nleaf = 1 << j;
            for (int n = 0; n < nleaf; ++n) {
                int tz = 0;
                while ((n >> tz) & 1) ++tz;                            // merges this leaf completes
                MCML_TRY(leaf(tz));
                for (int l = 0; l < tz; ++l) {
                    if (h.cm) vec(k_nuts_merge<true>, Srho[l], Spb[l], Spe[l], Sth[l], nv, nc.choose + (size_t)l * nc.Cp, ld, Q, C, nc, part, pstride, ldp);
                    else vec(k_nuts_merge<false>, Srho[l], Spb[l], Spe[l], Sth[l], nv, nc.choose + (size_t)l * nc.Cp, ld, Q, C, nc, part, pstride, ldp);
                    hipLaunchKernelGGL(k_nuts_merge_fin, dim3((C + 63) / 64), dim3(256), 0, c.stream, part, pstride, nchunk, ldp, C, nc);
                }
                // push: the node under construction becomes the stored node of level tz (all growing chains agree)
                std::swap(Srho[tz], nv.Crho); std::swap(Spb[tz], nv.Cpb); std::swap(Spe[tz], nv.Cpe); std::swap(Sth[tz], nv.Cth);
                MCML_HIP(hipGetLastError());
                if ((n & 15) == 15 && n + 1 < nleaf) {                 // deep doublings: stop once every chain has, and
                    int na = 0;                                        // re-pack when a quarter of the packed chains
                    MCML_TRY(count_active(&na));                       // have stopped (nothing packed outlives a leaf)
                    if (na == 0) break;
                    if (4 * na <= 3 * nact) { MCML_TRY(pack(na)); nact = na; }
                }
            }
            hipLaunchKernelGGL(k_nuts_end_doubling, dim3((C + 255) / 256), dim3(256), 0, c.stream, C, nc, j);
            if (h.cm) vec(k_nuts_tree_update<true>, Srho[j], Spb[j], Sth[j], nv, ld, Q, C, nc, part, pstride, ldp);
            else vec(k_nuts_tree_update<false>, Srho[j], Spb[j], Sth[j], nv, ld, Q, C, nc, part, pstride, ldp);
            hipLaunchKernelGGL(k_nuts_tree_fin, dim3((C + 63) / 64), dim3(256), 0, c.stream, part, pstride, nchunk, ldp, C, nc, max_depth);
            MCML_HIP(hipGetLastError());
            int na = 0;
            MCML_TRY(count_active(&na));
            if (na == 0) break;
            nact = na;
        }
        h.Cw = C;
        if (h.cm) vec(k_nuts_commit<true>, nv.Tth, h.V.d(), ld, Q, C);
        else vec(k_nuts_commit<false>, nv.Tth, h.V.d(), ld, Q, C);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
    // Stan's init_stepsize
    int nsearch = 0;
    int find_stepsize()
    {
        hipLaunchKernelGGL(k_nuts_heur_reset, dim3((C + 255) / 256), dim3(256), 0, c.stream, C, nc);
        const uint32_t base = 1000u * (uint32_t)nsearch++;             // every search draws its own momenta
        for (int round = 0; round < 80; ++round) {
            MCML_TRY(begin(base + (uint32_t)round, 16u * iter_idx + 5u));
            MCML_TRY(pack(C));
            hipLaunchKernelGGL(k_nuts_begin_doubling, dim3((C + 255) / 256), dim3(256), 0, c.stream, C, nc, 1);
            MCML_TRY(leaf(0));
            hipLaunchKernelGGL(k_nuts_heur_fin, dim3((C + 255) / 256), dim3(256), 0, c.stream, C, nc, round);
            MCML_HIP(hipGetLastError());
            if (round > 0) {
                int* d = c.scalars.as<int>() + 40;
                int nd = 0;
                hipLaunchKernelGGL(k_nuts_count, dim3(1), dim3(256), 0, c.stream, nc.hdone, C, d);
                MCML_TRY(copy_d2h(&nd, d, sizeof(int), c.stream));
                MCML_HIP(hipStreamSynchronize(c.stream));
                if (nd == C) break;
            }
        }
        hipLaunchKernelGGL(k_nuts_heur_done, dim3((C + 255) / 256), dim3(256), 0, c.stream, C, nc);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
};

int nuts_sample(Ctx& c, const double* beta, double var_par, const glmmr_mcml_nuts_opts* o, uint64_t seed, uint32_t iter_idx,
                int* depth_out, int* nleap_out, double* eps_out, double* accept_out, glmmr_mcml_nuts_diag* diag,
                int* ncols_out)
{
    MCML_REQUIRE(c.n > 0 && c.have_L && (c.ZL.d() || c.sp.active), "nuts: model / L not set (call update_L or set_L first)");
    MCML_REQUIRE(o->warmup >= 0 && o->nsamp > 0 && o->chains >= 1, "nuts: bad options");
    const int max_depth = o->max_treedepth > 0 ? o->max_treedepth : 10;
    MCML_REQUIRE(max_depth <= NUTS_MAXD, "nuts: max_treedepth %d > %d", max_depth, NUTS_MAXD);
    const double delta = o->adapt_delta > 0 ? o->adapt_delta : 0.8;
    MCML_REQUIRE(delta < 1, "nuts: adapt_delta must be in (0, 1)");
    const double eps0 = o->stepsize > 0 ? o->stepsize : 1.0;
    const int C = o->chains, Q = c.Q;
    const int d = (o->nsamp + C - 1) / C, total = o->warmup + d, ncols = C * d;
    HmcState& h = c.hmc;
    MCML_TRY(model_update_beta(c, beta));
    MCML_TRY(hmc_alloc(c, C));
    ChainArrays ca = chain_arrays(h);
    NutsState& ns = c.nuts;
    const int nvec = 16 + 4 * (max_depth + 1);
    const bool diag_metric = o->metric == 0;
    for (int i = 0; i < nvec; ++i) MCML_TRY(h.cm ? ns.vecs[i].alloc(C, Q) : ns.vecs[i].alloc(Q, C));
    MCML_TRY(ns.chain.ensure(nuts_chain_bytes(C)));
    NutsRun r{c, h, ns, nuts_chain(ns.chain.p, C), NutsVecs{}};
    double* v[16];
    for (int i = 0; i < 16; ++i) v[i] = ns.vecs[i].d();
    r.nv = NutsVecs{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
    for (int l = 0; l <= max_depth; ++l) {
        r.Srho[l] = ns.vecs[16 + 4 * l].d(); r.Spb[l] = ns.vecs[17 + 4 * l].d(); r.Spe[l] = ns.vecs[18 + 4 * l].d();
        r.Sth[l] = ns.vecs[19 + 4 * l].d();
    }
    r.C = C; r.Q = Q; r.ld = h.V.ld; r.var_par = var_par; r.seed = seed; r.chain_offset = (uint32_t)o->chain_offset;
    r.iter_idx = iter_idx;
    const int F = h.cm ? C : Q, S = h.cm ? Q : C;
    const int SB = h.cm ? NutsTile<true>::SB : NutsTile<false>::SB;
    r.vgrid = dim3((F + 255) / 256, (S + SB - 1) / SB);
    r.nchunk = h.cm ? (int)r.vgrid.y : (int)r.vgrid.x;
    r.ldp = round_up(C, 64);
    r.pstride = (size_t)r.nchunk * r.ldp;
    MCML_TRY(ns.part.ensure(sizeof(double) * 6 * r.pstride));
    r.part = ns.part.d();
    MCML_REQUIRE(h.V.ld == ns.vecs[0].ld, "nuts: state leading dimensions differ");

    DevMat samp;
    MCML_TRY(samp.alloc(Q, ncols));
    MCML_HIP(hipMemsetAsync(samp.d(), 0, sizeof(double) * (size_t)samp.ld * ncols, c.stream));
    DevBuf d_depth, d_nleap, d_eps, d_acc;
    if (depth_out) MCML_TRY(d_depth.ensure(sizeof(int) * (size_t)C * total));
    if (nleap_out) MCML_TRY(d_nleap.ensure(sizeof(int) * (size_t)C * total));
    if (eps_out) MCML_TRY(d_eps.ensure(sizeof(double) * (size_t)C * total));
    if (accept_out) MCML_TRY(d_acc.ensure(sizeof(double) * (size_t)C * total));

    // the HMC sampler's initialisation sets up the per-chain arrays; the state itself is then Stan's uniform(-2, 2)
    const int nchq = (Q + cm_qrows(Q) - 1) / cm_qrows(Q);
    if (h.cm)
        hipLaunchKernelGGL(k_cm_init, dim3((C + 63) / 64, nchq), dim3(256), 0, c.stream, h.V.d(), h.V.ld, Q, C, cm_chain(ca), seed,
                           (uint32_t)o->chain_offset, iter_idx, (const double*)nullptr);
    else
        hipLaunchKernelGGL(k_hmc_init, dim3(C), dim3(256), 0, c.stream, h.V.d(), h.V.ld, Q, ca, seed, (uint32_t)o->chain_offset,
                           iter_idx, (const double*)nullptr);
    if (h.cm) r.vec(k_nuts_init_uniform<true>, h.V.d(), r.ld, Q, C, seed, (uint32_t)o->chain_offset, 16u * iter_idx + 6u);
    else r.vec(k_nuts_init_uniform<false>, h.V.d(), r.ld, Q, C, seed, (uint32_t)o->chain_offset, 16u * iter_idx + 6u);
    hipLaunchKernelGGL(k_nuts_chain_init, dim3((C + 255) / 256), dim3(256), 0, c.stream, C, r.nc, eps0, seed,
                       (uint32_t)o->chain_offset, iter_idx);
    MCML_HIP(hipGetLastError());
    // unit metric to start with (Stan's diag_e starts from ones too), Welford state cleared
    if (h.cm) r.vec(k_nuts_metric<true>, r.nv, r.ld, Q, C, 0.0, 1);
    else r.vec(k_nuts_metric<false>, r.nv, r.ld, Q, C, 0.0, 1);
    MCML_TRY(r.find_stepsize());
    long long heur_leaps = r.leapfrogs;
    // windowed_adaptation (stan/mcmc/windowed_adaptation.hpp): all integer bookkeeping, the same for every chain
    struct Windows {
        int W, init = 75, term = 50, base = 25, counter = 0, next = 0, size = 0;
        explicit Windows(int W_) : W(W_) {
            if (W >= 20 && init + base + term > W) { init = (int)(0.15 * W); term = (int)(0.1 * W); base = W - (init + term); }
            next = init + base - 1; size = base;
        }
        bool in_window() const { return counter >= init && counter < W - term && counter != W; }
        bool at_end() const { return counter == next && counter != W; }
        void compute_next() {
            if (next == W - term - 1) return;
            size *= 2;
            next = counter + size;
            if (next == W - term - 1) return;
            const int boundary = next + 2 * size;
            if (boundary >= W - term) next = W - term - 1;
        }
    } win(o->warmup);
    int wsamp = 0;

    long long sum_leap_chain = 0;                                     // filled from the device at the end
    for (int it = 0; it < total; ++it) {
        MCML_TRY(r.transition((uint32_t)it, max_depth));
        const int adapt = it < o->warmup, last = it == o->warmup - 1;
        hipLaunchKernelGGL(k_nuts_end_iter, dim3((C + 255) / 256), dim3(256), 0, c.stream, C, r.nc, adapt, last, delta, it,
                           depth_out ? d_depth.as<int>() : nullptr, nleap_out ? d_nleap.as<int>() : nullptr,
                           eps_out ? d_eps.d() : nullptr, accept_out ? d_acc.d() : nullptr);
        if (adapt && diag_metric) {                                      // var_adaptation::learn_variance
            if (win.in_window()) {
                ++wsamp;
                if (h.cm) r.vec(k_nuts_welford<true>, h.V.d(), r.nv, r.ld, Q, C, (double)wsamp);
                else r.vec(k_nuts_welford<false>, h.V.d(), r.nv, r.ld, Q, C, (double)wsamp);
            }
            const bool update = win.at_end();
            if (update) {
                win.compute_next();
                if (h.cm) r.vec(k_nuts_metric<true>, r.nv, r.ld, Q, C, (double)wsamp, 0);
                else r.vec(k_nuts_metric<false>, r.nv, r.ld, Q, C, (double)wsamp, 0);
                wsamp = 0;
            }
            ++win.counter;
            MCML_HIP(hipGetLastError());
            if (update) {                                                // adapt_diag_e_nuts::transition
                const long long before = r.leapfrogs;
                MCML_TRY(r.find_stepsize());
                heur_leaps += r.leapfrogs - before;
            }
        }
        if (it >= o->warmup) {
            const int col = it - o->warmup;
            if (h.cm)
                hipLaunchKernelGGL(k_cm_transpose, dim3((Q + 31) / 32, (C + 31) / 32), dim3(256), 0, c.stream, h.V.d(), h.V.ld, Q, C,
                                   samp.d(), (size_t)samp.ld, (size_t)d, col);
            else
                hipLaunchKernelGGL(k_hmc_store, dim3(C), dim3(256), 0, c.stream, h.V.d(), h.V.ld, Q, samp.d(), samp.ld, d, col);
        }
        MCML_HIP(hipGetLastError());
    }
    (void)sum_leap_chain;
    // u = L * gamma (gen_u_samples.R:66)
    MCML_TRY(c.U.alloc(Q, ncols));
    MCML_HIP(hipMemsetAsync(c.U.d(), 0, sizeof(double) * (size_t)c.U.ld * ncols, c.stream));
    if (c.sp.active && c.sp.row_start.p) {
        int gy = ncols < 1024 ? ncols : 1024;
        hipLaunchKernelGGL(k_blockdiag_LV, dim3((Q + 255) / 256, gy), dim3(256), 0, c.stream, Q, ncols, c.sp.row_start.as<int>(),
                           c.L.d(), c.L.ld, samp.d(), samp.ld, c.U.d(), c.U.ld);
        MCML_HIP(hipGetLastError());
    } else {
        EpiAxpby epi{c.U.d(), c.U.ld, 1.0, 0.0};
        MCML_TRY(launch_gemm<false>(c.stream, Q, ncols, Q, c.L.d(), c.L.ld, samp.d(), samp.ld, epi));
    }
    c.mcols = ncols; c.niter = ncols; c.zu_valid = false; c.uall_valid = false;
    if (depth_out) MCML_TRY(copy_d2h(depth_out, d_depth.p, sizeof(int) * (size_t)C * total, c.stream));
    if (nleap_out) MCML_TRY(copy_d2h(nleap_out, d_nleap.p, sizeof(int) * (size_t)C * total, c.stream));
    if (eps_out) MCML_TRY(copy_d2h(eps_out, d_eps.p, sizeof(double) * (size_t)C * total, c.stream));
    if (accept_out) MCML_TRY(copy_d2h(accept_out, d_acc.p, sizeof(double) * (size_t)C * total, c.stream));
    double dg[5] = {0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_nuts_diag, dim3(1), dim3(64), 0, c.stream, r.nc, C, c.scalars.d() + 8);
    MCML_TRY(copy_d2h(dg, c.scalars.d() + 8, sizeof dg, c.stream));
    MCML_HIP(hipStreamSynchronize(c.stream));
    c.prof.collect();
    if (diag) {
        diag->mean_e = dg[0] / C; diag->min_e = dg[1]; diag->max_e = dg[2];
        diag->divergent = (long long)dg[3]; diag->treedepth_hits = (long long)dg[4];
        diag->batched_leapfrogs = r.leapfrogs - heur_leaps;
        diag->stepsize_search_leapfrogs = heur_leaps;
    }
    if (ncols_out) *ncols_out = ncols;
    return MCML_OK;
}

}  // namespace mcml
