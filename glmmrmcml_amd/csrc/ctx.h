// ctx.h -- the device-resident MCML problem: everything the hot path needs
// stays in HBM between calls (Z, X, y, the samples u, L, ZL, workspaces); only
// parameter vectors go down and scalars / small statistics come back.
#pragma once
#include "common.h"
#include "covspec.h"
#include "band_plan.h"

namespace mcml {

constexpr int CHOL_NB = 128;     // leaf size of the recursive Cholesky / TRSM
constexpr int SMALL_BLOCK = 32;  // blocks up to this size are factorised one wave each

// cross-rank reduction hook (sum, f64, in place on a device buffer).  Single
// process: null.  bench.py / the distributed front end install a callback that
// all-reduces the buffer with torch.distributed (backend "nccl" = RCCL).
typedef int (*reduce_fn)(void* user, double* dev_buf, int n);

// kernel family that served an HMC product (glmmr_mcml_ctx_last_kernels)
enum { KERNEL_SKINNY = 0, KERNEL_BAND = 1, KERNEL_DLDS = 2, KERNEL_REG = 3, KERNEL_SPARSE = 4 };

// the sampler's step-count read-back (hmc.hip hmc_sample): a ring of host memory mapped into the device, written by
// k_max_steps with (proposal sequence number << 32 | count), read by the host with plain loads; lives as long as the context
struct StepRing {
    static constexpr int SLOTS = 8;
    unsigned long long* h = nullptr;    // host address
    unsigned long long* d = nullptr;    // the same memory as the device sees it
    unsigned seq = 0;                   // sequence number of the last proposal launched
    StepRing() = default;
    StepRing(const StepRing&) = delete;
    StepRing& operator=(const StepRing&) = delete;
    ~StepRing() { if (h) (void)hipHostFree(h); }
};

struct HmcState {
    StepRing ring;
    int C = 0;                  // chains resident
    int Cw = 0;                 // columns the two products and the log-density kernels process (= C; the No-U-Turn
                                // sampler packs the chains whose trees still grow into the first Cw columns)
    DevMat V, R, UP, GRAD, GRADP;   // Q x C
    DevMat MU, S;               // n x C
    DevBuf chain;               // per-chain scalars, see hmc.hip
    DevBuf partial;             // per-(row slot, chain) partial sums
    int partial_slots = 0;
    bool cm = false;            // chain-major state (sparse ZL operator, hmc_cm.h): element (c, r) at c + r * ld
    DevBuf cm_part, cm_acc;     // chain-major path: partial sums (ll | lp | kin | ss), accept flags
    DevBuf cm_part_fwd;         // ... and the log-density partials the last forward product of a trajectory leaves (one per workgroup)
    DevMat LX, ZS;              // factored operator (SparseZL::factored): L X and Z' S, C x Q
};

// No-U-Turn sampler (nuts.h): edges, tree, node under construction and one stored node per tree level (Q x C each),
// per-chain scalars, partial sums
constexpr int NUTS_MAXD = 12;                       // deepest tree supported (Stan's default max_treedepth: 10)
struct NutsState {
    DevMat vecs[16 + 4 * (NUTS_MAXD + 1)];
    DevBuf chain, part;
};

// ZL = Z L held as padded-CSR (ELL) rows plus its transpose in CSR: used by the sampler (chain-major state, hmc_cm.h)
// instead of the dense n x Q products when Z is indicator-like and every covariance block is
// diagonal or small, so that a row of ZL has only a few nonzeros (configs 1, 4, 5): the two
// products are then gathers bound by HBM bandwidth, not n x Q x C GEMMs.
struct SparseZL {
    bool possible = false, active = false, built = false;
    int W = 0;                       // ELL width
    long nnz = 0;
    DevBuf ell_col, ell_src, ell_z, ell_val;   // n x W (column-major): column of ZL, flat index into L, Z value, value
    DevBuf csr_ptr, csr_i, csr_pos, csr_val;   // rows of ZL' : q -> (observation, position in ell_val, value)
    DevBuf row_start;                          // first column of row q of the block-diagonal L (U = L V, hmc.hip)
    // factored form ZL = Z * L (hmc_cm.h): chosen when a row of ZL mostly repeats a row of L, i.e. when gathering
    // through Z and applying the blocks of L separately touches far fewer entries than nnz(ZL)
    bool factored = false;
    long nnz_z = 0, nnz_l = 0;
    DevBuf zcsr_ptr, zcsr_i, zcsr_val;         // rows of Z' : q -> (observation, value)
    DevBuf row_end;                            // one past the last row of column q of L (= end of q's block)
    DevBuf blk_ptr;                            // first row of every covariance block, B + 1 entries (k_cm_Lcol_Lrow)
    int nblk = 0, max_blk = 0;
};

// HIP-event timing of the dominant kernels, on the stream they are launched on
// (bench.py's roofline line).  kind 0 = HMC forward GEMM, 1 = HMC backward GEMM.
struct KernelProf {
    bool on = false;
    bool skip = false;                          // on, but this launch is only counted (the sampler times one proposal in four)
    long long seen[4] = {0, 0, 0, 0};           // launches since the last reset, timed or not
    std::vector<hipEvent_t> ev;                 // pool
    std::vector<int> kind, e0, e1;              // per timed launch: kind, start / stop event index
    size_t used = 0, nev = 0;
    int last_stop = -1;                         // event recorded right after the previous timed launch
    double ms[4] = {0, 0, 0, 0};
    long long cnt[4] = {0, 0, 0, 0};
    int fresh() {
        if (nev >= ev.size())
            for (int i = 0; i < 512; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return -1; ev.push_back(e); }
        return (int)nev++;
    }
    // chain = true: this launch directly follows the previous timed launch on the same stream, so that
    // launch's stop marker doubles as this one's start (one marker per kernel instead of two: every
    // marker costs a few microseconds of idle GPU between dependent kernels)
    int begin(hipStream_t s, int k, bool chain = false) {
        if (!on) return -1;
        ++seen[k];
        if (skip) return -1;
        int st = (chain && last_stop >= 0) ? last_stop : fresh();
        if (st < 0) return -1;
        if (!(chain && last_stop >= 0)) (void)hipEventRecord(ev[st], s);
        if (used >= kind.size()) { kind.resize(used + 512); e0.resize(used + 512); e1.resize(used + 512); }
        kind[used] = k; e0[used] = st; e1[used] = -1;
        return (int)used++;
    }
    void end(hipStream_t s, int slot) {
        if (slot < 0) return;
        const int sp = fresh();
        if (sp < 0) return;
        (void)hipEventRecord(ev[sp], s);
        e1[slot] = sp; last_stop = sp;
    }
    void unchain() { last_stop = -1; }          // something else was launched in between
    void collect() {       // call after the stream has been synchronised
        for (size_t i = 0; i < used; ++i) {
            float t = 0;
            if (e1[i] >= 0 && hipEventElapsedTime(&t, ev[e0[i]], ev[e1[i]]) == hipSuccess) { ms[kind[i]] += t; cnt[kind[i]] += 1; }
        }
        used = 0; nev = 0; last_stop = -1;
    }
    ~KernelProf() { for (auto e : ev) (void)hipEventDestroy(e); }
};

// the factorisation's launch sequence as a hipGraph (mvn.hip potrf_graphed): the theta-step evaluates the same
// (matrix, shape) dozens of times per MCML iteration; key = what the captured kernels' arguments depend on
struct CholGraph {
    hipGraphExec_t exec = nullptr;
    const double* A = nullptr; const double* linv = nullptr; int lda = 0, n = 0, extra = 0, seen = 0;
    long long used = 0;                 // launch counter value at the last use (the least recently used entry is replaced)
    // Calibration (mvn.hip potrf_graphed).  On this stack every other hipGraphInstantiate in a process yields an
    // executable whose two chains run 1.5-2x slower than the same graph instantiated before or after it (measured: the
    // 3rd and 5th instantiation; the parallel branch lands on a hardware queue that does not overlap with the launch
    // stream's).  The first replays are therefore timed against the eager launch sequence and a slow executable is
    // instantiated again from the kept template.
    hipGraph_t tmpl = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    float eager_ms = 0.f, best_ms = 0.f;
    hipGraphExec_t best = nullptr;      // fastest executable seen so far while `exec` is the one on trial
    int tries = 0, trial_launches = 0;  // an executable's first launch carries its upload: the second one is timed
    bool timed = false, settled = false, branchy = false;
    void release() {
        if (best && best != exec) (void)hipGraphExecDestroy(best);
        best = nullptr;
        if (exec) { (void)hipGraphExecDestroy(exec); exec = nullptr; }
        if (tmpl) { (void)hipGraphDestroy(tmpl); tmpl = nullptr; }
        if (t0) { (void)hipEventDestroy(t0); t0 = nullptr; }
        if (t1) { (void)hipEventDestroy(t1); t1 = nullptr; }
    }
};
// a few graphs side by side: a model with two or more large covariance blocks of different size evaluates them in turn,
// and a one-entry cache would re-capture (i.e. run eagerly) every time
struct CholGraphCache {
    static constexpr int CAP = 4;
    std::vector<CholGraph> g;
    long long tick = 0;
    void clear() { for (CholGraph& e : g) e.release(); g.clear(); }
    ~CholGraphCache() { clear(); }
    CholGraphCache() = default;
    CholGraphCache(const CholGraphCache&) = delete;
    CholGraphCache& operator=(const CholGraphCache&) = delete;
    CholGraphCache(CholGraphCache&& o) noexcept : g(std::move(o.g)), tick(o.tick) { o.g.clear(); }
    CholGraphCache& operator=(CholGraphCache&& o) noexcept { if (this != &o) { clear(); g = std::move(o.g); tick = o.tick; o.g.clear(); } return *this; }
    // branchy: a two-chain graph (a single evaluation) -- at most two of those are kept alive (of three or more alive in
    // a process every new instantiation measured slow, whatever the number of tries); one-chain graphs (batches): four
    CholGraph& find(const double* A, const double* linv, int lda, int n, int extra, bool branchy) {
        ++tick;
        for (CholGraph& e : g)
            if (e.A == A && e.linv == linv && e.lda == lda && e.n == n && e.extra == extra) { e.used = tick; return e; }
        const int cap = branchy ? 2 : CAP;
        int have = 0; size_t old = g.size();
        for (size_t i = 0; i < g.size(); ++i)
            if (g[i].branchy == branchy) { ++have; if (old == g.size() || g[i].used < g[old].used) old = i; }
        if (have >= cap) {
            g[old].release();
            g[old] = CholGraph();
            std::swap(g[old], g.back());
        } else g.push_back(CholGraph());
        CholGraph& e = g.back();
        e.A = A; e.linv = linv; e.lda = lda; e.n = n; e.extra = extra; e.seen = 0; e.used = tick; e.branchy = branchy;
        return e;
    }
};

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // problem
    int n = 0, Q = 0, P = 0, flink = 0, link_code = 0;
    int z_width = 0;            // > 0: Z also held as padded-CSR rows of this width
    DevBuf z_idx, z_val;        // n x z_width (column-major)
    std::vector<int> h_zidx; std::vector<double> h_zval;   // host copy of the same
    SparseZL sp;
    bool no_sparse_zl = false;  // the Laplace path works on the dense ZL / ZLT
    bool l_foreign = false;     // L came from the caller (set_L), not from theta: it need not have the block pattern the
                                // sparse ZL operator assumes; cleared as soon as L is regenerated from theta (mvn_gen_L)
    CovSpec cov;
    DevMat Z, X;                // n x Q, n x P
    DevBuf y;                   // n
    DevBuf d_cov, d_data, d_blocks, d_rowblock;   // covariance spec on the device

    // samples
    DevMat U;                   // Q x mcols (u = L v)
    int mcols = 0;              // columns held (theta-step uses all: mcmldmatrix.h:24)
    int niter = 0;              // columns the beta-step uses (mcmlmodel.h:73,296)
    int m_global = 0, niter_global = 0;   // across ranks (== local when single rank)
    DevMat ZU;                  // n x mcols, cached Z u
    bool zu_valid = false;

    // model state
    DevMat L, ZL, ZLT;          // Q x Q lower factor of D; n x Q; Q x n
    BandPlan plan_fwd, plan_bwd;   // nonzero K-tile range of every 80-row band of ZL / ZLT + the work decomposition (dgemm_band.h)
    DevBuf kr_scratch;
    bool band_fwd = false, band_bwd = false;   // the structural zeros are worth skipping
    long band_fwd_tiles = 0, band_bwd_tiles = 0;   // sum over bands of the K tiles (of 32) actually multiplied
    DevBuf xb;                  // n
    bool have_L = false;

    // MVN workspaces
    DevMat Dwork;               // maxdim x maxdim
    DevMat Uwork;               // maxdim x mcols
    DevBuf linv;                // (maxdim/128 + 1) x 128 x 128
    DevBuf partials;            // reduction partials
    DevBuf scalars;             // small device scalars: [0..15] results, [16] error flag (int)
    int maxdim_large = 0;       // largest block that takes the recursive path
    int n_small = 0, n_diag_rows = 0;
    DevBuf small_ids;                           // ids of the small dense blocks (mvn_setup), for k_small_ll

    // sampler
    int last_kernel[2] = {-1, -1};   // kernel family of the last forward / backward product (KERNEL_*)
    HmcState hmc;
    NutsState nuts;
    KernelProf prof;

    // distribution
    int rank = 0, world = 1;
    reduce_fn reduce = nullptr;
    void* reduce_user = nullptr;
    void* comm = nullptr;       // ncclComm_t of the native RCCL path (comm.hip); null = hook or single process
    long long coll_calls = 0, coll_doubles = 0;   // collectives issued / doubles summed (tests, bench)
    long long gather_calls = 0, gather_doubles = 0;   // all-gathers issued / doubles received
    // every rank's sample columns (comm.hip allgather_dev): rank r's mcols columns at column r * mcols.  Valid until the
    // samples change.  The theta-step of a sharded job reads these (drivers.hip::d_optim).
    DevMat Uall;
    bool uall_valid = false;
    bool theta_log_on = false;          // tests: every MVN objective evaluation as (theta..., log-likelihood)
    std::vector<double> theta_log;
    long long theta_rounds = 0, theta_evals_own = 0, theta_evals_all = 0;   // sharded theta-step: exchanges, evaluations here / everywhere
    // rank emulation on one GPU (bench.py --as-rank-of N, include/glmmr_mcml_c.h glmmr_mcml_dbg_emulate_world): the
    // other ranks are copies of this one -- a sum is `emu_world` times the local value, a gather `emu_world` copies of
    // the local block; the candidate thetas of the other ranks are evaluated here too (mode 1, values recorded) or taken
    // from that record (mode 2: only this rank's share runs, which is what a rank of the real job executes)
    int emu_world = 0, emu_mode = 0;
    std::vector<std::vector<double>> emu_trace; size_t emu_pos = 0;
    DevBuf reduce_buf;          // doubles handed to the collective
    DevBuf scratch;             // short-lived per-call scratch

    // side stream + events for the Cholesky look-ahead (mvn.hip potrf_blocked)
    hipStream_t aux = nullptr;      // high priority: the leaf of the eager fork-join
    hipStream_t aux_lo = nullptr;   // lowest priority: the bulk chain of the captured schedule
    hipEvent_t ev_col = nullptr, ev_leaf = nullptr, ev_ps = nullptr, ev_b = nullptr;
    CholGraphCache chol_graphs;
    std::vector<hipEvent_t> ev_ring;                 // one event per dependency edge of the captured schedule (mvn.hip)
    DevMat Dbatch;                                   // mvn_loglik_batch: the candidates' matrices side by side
    DevBuf bscal;                                    // ... and their result scalars (4 per candidate)
    ~Ctx() {
        for (hipEvent_t e : ev_ring) if (e) (void)hipEventDestroy(e);
        comm_release_hook();
        if (ev_col) (void)hipEventDestroy(ev_col);
        if (ev_leaf) (void)hipEventDestroy(ev_leaf);
        if (ev_ps) (void)hipEventDestroy(ev_ps);
        if (ev_b) (void)hipEventDestroy(ev_b);
        if (aux) (void)hipStreamDestroy(aux);
        if (aux_lo) (void)hipStreamDestroy(aux_lo);
    }

    int sync() { MCML_HIP(hipStreamSynchronize(stream)); return MCML_OK; }
    void comm_release_hook();
};

// ---- comm.hip ----
void comm_release(Ctx& c);
int allreduce_dev(Ctx& c, double* dev, int n);              // in place on device memory, on c.stream
int allgather_dev(Ctx& c, const double* send, double* recv, size_t count);
int gather_samples(Ctx& c);                                  // c.Uall <- all ranks' sample columns
inline int comm_world(const Ctx& c) { return c.emu_world > 1 ? c.emu_world : c.world; }
inline void Ctx::comm_release_hook() { comm_release(*this); }

// ---- mvn.hip ----
int mvn_setup(Ctx& c);
// sum over the locally held columns of sum_b log N(u_b; 0, D_b(theta))
int mvn_loglik_sum(Ctx& c, const double* theta, double* sum_out);
// the same over the m columns of any resident sample matrix (Q x m, leading dimension ldu)
int mvn_loglik_sum_on(Ctx& c, const double* theta, const double* U, int ldu, int m, double* sum_out);
// k candidate thetas (npar x k, column-major) factorised side by side: sums[j], rcs[j] = MCML_OK | MCML_ENOTPD
int mvn_loglik_batch(Ctx& c, const double* thetas, int k, const double* U, int ldu, int m, double* sums, int* rcs);
// L = genD(0, chol=true, upper=false) (mcml_full.cpp:68): block-diagonal lower factor
int mvn_gen_L(Ctx& c, const double* theta, bool chol);
int potrf_lower(Ctx& c, double* A, int n, int lda);                          // in place
int potrf_leaf_profile(Ctx& c, unsigned long long* host10);                 // debug: phase clocks of one leaf
int potrf_lower_checked(Ctx& c, double* A, int n, int lda);                  // + MCML_ENOTPD if a pivot failed
// x <- (L L')^-1 x for one vector, L from the LAST potrf_lower (its diagonal-block inverses are in c.linv)
int potrs_lower_vec(Ctx& c, const double* L, int ldl, int n, double* x, double* tmp);
int trsm_left_lower(Ctx& c, const double* L, int ldl, int n, double* U, int ldu, int m);

// ---- model.hip ----
int model_setup(Ctx& c, const double* Z, const double* X, const double* y);
int allreduce_host(Ctx& c, double* vals, int n);             // comm.hip
int model_update_beta(Ctx& c, const double* beta);          // xb = X beta
int model_update_zu(Ctx& c);                                // ZU = Z U (cached)
int model_update_L(Ctx& c);                                 // ZL = Z L, ZLT (dense) or the ELL/CSR pair (sparse)
int model_loglik_sum(Ctx& c, double var_par, double* sum_out);
int model_mcnr_stats(Ctx& c, double var_par, double* stats /* P*P + P + 2 */);
int mcnr_finish(int P, const double* stats, const double* beta, double* beta_out, double* sigma_out);

// ---- hmc.hip ----
struct glmmr_mcml_hmc_opts_fwd;
int hmc_dbg_log_prob_grad(Ctx& c, const double* beta, double var_par, const double* V, int ncols, double* lp,
                          double* G);

// ---- reductions shared by several modules ----
int device_sum(Ctx& c, const double* partials, int n, double* dev_out);

}  // namespace mcml

// the opaque handle of include/glmmr_mcml_c.h
struct glmmr_mcml_ctx { mcml::Ctx c; };
