// debug_hooks.hip -- C-ABI test/bench hooks for the GEMM building block
// (declared in include/glmmr_mcml_c.h under "test hooks").
#include "dgemm_mfma.h"
#include "dgemm_dl.h"
#include "dgemm_band.h"

using namespace mcml;

extern "C" int glmmr_mcml_dbg_dgemm(int M, int N, int K, const double* A, int lda,
                                    const double* B, int ldb, int b_nmajor,
                                    double alpha, double beta, double* C, int ldc,
                                    int lower_only, int force_tile)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device: libglmmr_mcml_hip has no CPU fallback");
        return MCML_ENODEVICE;
    }
    hipStream_t s = nullptr;
    DevMat dA, dB, dC;
    if (force_tile == 40) {
        // the sampler's dense direct-to-LDS kernel (dgemm_dlds.h: dgemm_dlds_asm_kernel unless GLMMR_MCML_DLDS picks a
        // compiler-scheduled variant), with the operand contract the sampler meets: zero columns / rows up to the
        // next multiple of 32 in K
        MCML_REQUIRE(!b_nmajor && !lower_only, "dbg_dgemm: the direct-to-LDS kernel takes a K-major B and all tiles");
        const int kpad = round_up(K, 32);
        MCML_TRY(dA.alloc(M, K, 32));
        MCML_HIP(hipMemset(dA.d(), 0, sizeof(double) * (size_t)dA.ld * dA.cols_alloc));
        MCML_HIP(hipMemcpy2D(dA.d(), sizeof(double) * dA.ld, A, sizeof(double) * lda, sizeof(double) * M, K, hipMemcpyHostToDevice));
        MCML_TRY(dB.alloc(kpad, N));
        MCML_HIP(hipMemset(dB.d(), 0, sizeof(double) * (size_t)dB.ld * N));
        MCML_HIP(hipMemcpy2D(dB.d(), sizeof(double) * dB.ld, B, sizeof(double) * ldb, sizeof(double) * K, N, hipMemcpyHostToDevice));
        MCML_TRY(upload_matrix(dC, C, M, N, ldc, s));
        MCML_REQUIRE(dlds_applicable(M, N, K, dA.d(), dA.ld, dA.cols_alloc, dB.d(), dB.ld), "dbg_dgemm: operands do not meet the direct-to-LDS contract");
        EpiAxpby epi2{dC.d(), dC.ld, alpha, beta};
        MCML_TRY(launch_gemm_dlds(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi2));
        MCML_TRY(download_matrix(C, ldc, dC.d(), dC.ld, M, N, s));
        return MCML_OK;
    }
    MCML_TRY(upload_matrix(dA, A, M, K, lda, s));
    if (b_nmajor) MCML_TRY(upload_matrix(dB, B, N, K, ldb, s));
    else MCML_TRY(upload_matrix(dB, B, K, N, ldb, s));
    MCML_TRY(upload_matrix(dC, C, M, N, ldc, s));
    EpiAxpby epi{dC.d(), dC.ld, alpha, beta};
    int rc;
    if (force_tile >= 20)      // 20..25: the deep-ring direct-to-LDS kernel (dgemm_dl.h), tiles 1..6
        rc = b_nmajor ? launch_gemm_dl<true>(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi, lower_only != 0, force_tile - 19)
                      : launch_gemm_dl<false>(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi, lower_only != 0, force_tile - 19);
    else
        rc = b_nmajor ? launch_gemm<true>(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi, lower_only != 0, force_tile)
                      : launch_gemm<false>(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi, lower_only != 0, force_tile);
    MCML_TRY(rc);
    MCML_TRY(download_matrix(C, ldc, dC.d(), dC.ld, M, N, s));
    return MCML_OK;
}

// times `iters` launches of the M x N x K product on resident random operands
extern "C" int glmmr_mcml_dbg_dgemm_bench2(int M, int N, int K, int b_nmajor, int iters, int force_tile,
                                           int lower_only, double beta, double* ms_per_launch);
extern "C" int glmmr_mcml_dbg_dgemm_bench(int M, int N, int K, int b_nmajor, int iters,
                                          int force_tile, double* ms_per_launch)
{
    return glmmr_mcml_dbg_dgemm_bench2(M, N, K, b_nmajor, iters, force_tile, 0, 0.0, ms_per_launch);
}

// the same with the SYRK-like options of the Cholesky updates: lower tiles only, C read-modify-written
extern "C" int glmmr_mcml_dbg_dgemm_bench2(int M, int N, int K, int b_nmajor, int iters, int force_tile,
                                           int lower_only, double beta, double* ms_per_launch)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device: libglmmr_mcml_hip has no CPU fallback");
        return MCML_ENODEVICE;
    }
    hipStream_t s = nullptr;
    std::vector<double> hA((size_t)M * K), hB((size_t)K * N), hC((size_t)M * N, 0.0);
    uint64_t x = 88172645463325252ULL;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 * 2 - 1; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    DevMat dA, dB, dC;
    MCML_TRY(upload_matrix(dA, hA.data(), M, K, M, s));
    if (b_nmajor) MCML_TRY(upload_matrix(dB, hB.data(), N, K, N, s));
    else MCML_TRY(upload_matrix(dB, hB.data(), K, N, K, s));
    MCML_TRY(upload_matrix(dC, hC.data(), M, N, M, s));
    EpiAxpby epi{dC.d(), dC.ld, beta != 0.0 ? 1e-3 : 1.0, beta};
    const bool lo = lower_only != 0;
    auto go = [&]() {
        if (force_tile >= 20)
            return b_nmajor ? launch_gemm_dl<true>(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi, lo, force_tile - 19)
                            : launch_gemm_dl<false>(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi, lo, force_tile - 19);
        return b_nmajor ? launch_gemm<true>(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi, lo, force_tile)
                        : launch_gemm<false>(s, M, N, K, dA.d(), dA.ld, dB.d(), dB.ld, epi, lo, force_tile);
    };
    for (int i = 0; i < 3; i++) MCML_TRY(go());
    hipEvent_t e0, e1;
    MCML_HIP(hipEventCreate(&e0));
    MCML_HIP(hipEventCreate(&e1));
    MCML_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++) MCML_TRY(go());
    MCML_HIP(hipEventRecord(e1, s));
    MCML_HIP(hipEventSynchronize(e1));
    float ms = 0;
    MCML_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MCML_OK;
}

// C = A * B through the banded kernel (dgemm_band.h) exactly as the sampler uses it: A is copied into a
// buffer whose columns are zero-padded to a multiple of 32, B's rows likewise, the K-tile ranges come from
// k_band_ranges.  For tests/ only.
extern "C" int glmmr_mcml_dbg_dgemm_band(int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                                         double* C, int ldc, int* tiles_executed)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device: libglmmr_mcml_hip has no CPU fallback");
        return MCML_ENODEVICE;
    }
    MCML_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "dbg_dgemm_band: bad argument");
    hipStream_t s = nullptr;
    DevMat dA, dB, dC;
    MCML_TRY(dA.alloc(M, K, 32));
    MCML_HIP(hipMemset(dA.d(), 0, sizeof(double) * (size_t)dA.ld * dA.cols_alloc));
    MCML_HIP(hipMemcpy2D(dA.d(), sizeof(double) * dA.ld, A, sizeof(double) * lda, sizeof(double) * M, K, hipMemcpyHostToDevice));
    MCML_TRY(dB.alloc(round_up(K, 32), N));
    MCML_HIP(hipMemset(dB.d(), 0, sizeof(double) * (size_t)dB.ld * N));
    MCML_HIP(hipMemcpy2D(dB.d(), sizeof(double) * dB.ld, B, sizeof(double) * ldb, sizeof(double) * K, N, hipMemcpyHostToDevice));
    MCML_TRY(dC.alloc(M, N));
    MCML_HIP(hipMemset(dC.d(), 0, sizeof(double) * (size_t)dC.ld * N));
    MCML_REQUIRE(dlds_applicable(M, N, K, dA.d(), dA.ld, dA.cols_alloc, dB.d(), dB.ld), "dbg_dgemm_band: contract");
    const int nbands = (M + BD_BM - 1) / BD_BM;
    DevBuf kr;
    MCML_TRY(kr.ensure(sizeof(int) * 2 * nbands));
    hipLaunchKernelGGL(k_band_ranges, dim3(nbands), dim3(256), 0, s, dA.d(), dA.ld, M, K, kr.as<int>());
    MCML_HIP(hipGetLastError());
    std::vector<int> hk(2 * nbands);
    MCML_HIP(hipMemcpy(hk.data(), kr.p, sizeof(int) * 2 * nbands, hipMemcpyDeviceToHost));
    BandPlan plan;
    plan.reset(M, K, hk);
    if (tiles_executed) *tiles_executed = (int)plan.tiles;
    EpiAxpby epi{dC.d(), dC.ld, 1.0, 0.0};
    MCML_TRY(launch_gemm_band(s, plan, N, dA.d(), dA.ld, dB.d(), dB.ld, epi));
    MCML_TRY(download_matrix(C, ldc, dC.d(), dC.ld, M, N, s));
    return MCML_OK;
}

// Sustained shader clock under the banded FP64 MFMA kernel: a lower-triangular M x M operand
// times an M x N matrix, `iters` launches; workgroup 0 accumulates s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz) over its lifetime.  out3 = [ms per launch, shader MHz, executed TFLOP/s]
extern "C" int glmmr_mcml_dbg_band_clocks(int M, int N, int iters, int mode, double* out3)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device: libglmmr_mcml_hip has no CPU fallback");
        return MCML_ENODEVICE;
    }
    hipStream_t s = nullptr;
    if (getenv("GLMMR_MCML_DBG_STREAM")) MCML_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<double> hA((size_t)M * M, 0.0), hB((size_t)M * N);
    uint64_t x = 88172645463325252ULL;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 * 2 - 1; };
    // GLMMR_MCML_DBG_DATA=const: every entry 1.0 instead of random mantissas (MFMA power draw depends on the data)
    const bool cst = getenv("GLMMR_MCML_DBG_DATA") && !strcmp(getenv("GLMMR_MCML_DBG_DATA"), "const");
    // =decay: entries fall off away from the diagonal like a Cholesky factor of an exponential covariance
    const bool dec = getenv("GLMMR_MCML_DBG_DATA") && !strcmp(getenv("GLMMR_MCML_DBG_DATA"), "decay");
    for (int j = 0; j < M; ++j) for (int i = j; i < M; ++i) hA[i + (size_t)j * M] = cst ? 1.0 : dec ? rnd() * std::exp(-(double)(i - j) / 40.0) : rnd();
    for (auto& v : hB) v = cst ? 1.0 : rnd();
    DevMat dA, dB, dC;
    MCML_TRY(dA.alloc(M, M, 32));
    MCML_HIP(hipMemset(dA.d(), 0, sizeof(double) * (size_t)dA.ld * dA.cols_alloc));
    MCML_HIP(hipMemcpy2D(dA.d(), sizeof(double) * dA.ld, hA.data(), sizeof(double) * M, sizeof(double) * M, M, hipMemcpyHostToDevice));
    MCML_TRY(dB.alloc(round_up(M, 32), N));
    MCML_HIP(hipMemset(dB.d(), 0, sizeof(double) * (size_t)dB.ld * N));
    MCML_HIP(hipMemcpy2D(dB.d(), sizeof(double) * dB.ld, hB.data(), sizeof(double) * M, sizeof(double) * M, N, hipMemcpyHostToDevice));
    MCML_TRY(dC.alloc(M, N));
    const int nbands = (M + BD_BM - 1) / BD_BM;
    DevBuf kr, clk;
    MCML_TRY(kr.ensure(sizeof(int) * 2 * nbands));
    MCML_TRY(clk.ensure(16));
    MCML_HIP(hipMemset(clk.p, 0, 16));
    hipLaunchKernelGGL(k_band_ranges, dim3(nbands), dim3(256), 0, s, dA.d(), dA.ld, M, M, kr.as<int>());
    std::vector<int> hk(2 * nbands);
    MCML_HIP(hipMemcpy(hk.data(), kr.p, sizeof(int) * 2 * nbands, hipMemcpyDeviceToHost));
    BandPlan plan;
    plan.reset(M, M, hk);
    const double tiles = (double)plan.tiles;
    EpiAxpby epi{dC.d(), dC.ld, 1.0, 0.0};
    if (mode < 0) {
        // the sampler's own instantiation (no experiment branches, no clock reads): what rocprofv3 should look at
        for (int i = 0; i < 3; i++)
            MCML_TRY((launch_gemm_band<EpiAxpby, false>(s, plan, N, dA.d(), dA.ld, dB.d(), dB.ld, epi)));
        hipEvent_t f0, f1;
        MCML_HIP(hipEventCreate(&f0));
        MCML_HIP(hipEventCreate(&f1));
        MCML_HIP(hipEventRecord(f0, s));
        for (int i = 0; i < iters; i++)
            MCML_TRY((launch_gemm_band<EpiAxpby, false>(s, plan, N, dA.d(), dA.ld, dB.d(), dB.ld, epi)));
        MCML_HIP(hipEventRecord(f1, s));
        MCML_HIP(hipEventSynchronize(f1));
        float fms = 0;
        MCML_HIP(hipEventElapsedTime(&fms, f0, f1));
        out3[0] = fms / iters; out3[1] = 0;
        out3[2] = 2.0 * 80.0 * 32.0 * tiles * N / (fms / iters * 1e-3) / 1e12;
        (void)hipEventDestroy(f0);
        (void)hipEventDestroy(f1);
        return MCML_OK;
    }
    if (mode & 16) {
        // phase stamps of every workgroup of ONE launch (dgemm_band.h), printed as offsets from the earliest start
        BandPlanDev* d = nullptr;
        MCML_TRY(plan.device_plan((N + BD_BN - 1) / BD_BN, s, &d));
        const int nb = d->nwg * d->gn;
        DevBuf st;
        MCML_TRY(st.ensure(sizeof(unsigned long long) * (8 + 8 * (size_t)nb)));
        for (int rep = 0; rep < 3; ++rep) {
            MCML_HIP(hipMemset(st.p, 0, sizeof(unsigned long long) * (8 + 8 * (size_t)nb)));
            MCML_TRY((launch_gemm_band<EpiAxpby, true>(s, plan, N, dA.d(), dA.ld, dB.d(), dB.ld, epi, st.as<unsigned long long>(), mode)));
            MCML_HIP(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> h(8 + 8 * (size_t)nb);
        MCML_HIP(hipMemcpy(h.data(), st.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ULL;
        for (int w = 0; w < nb; ++w) if (h[8 + 8 * w] && h[8 + 8 * w] < t0) t0 = h[8 + 8 * w];
        double mx[5] = {0, 0, 0, 0, 0}, sm[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < nb; ++w)
            for (int i = 0; i < 5; ++i) { double v = (double)(h[8 + 8 * w + i] - t0) * 0.01; sm[i] += v; if (v > mx[i]) mx[i] = v; }
        printf("band stamps (us from the first workgroup's start; %d workgroups, mean / max):\n", nb);
        const char* nm[5] = {"start", "ring filled", "K loop done", "epilogue issued", "end"};
        for (int i = 0; i < 5; ++i) printf("  %-16s %7.2f / %7.2f\n", nm[i], sm[i] / nb, mx[i]);
        if (mode & 32) for (int w = 0; w < nb; ++w) {
            printf("  wg %3d:", w);
            for (int i = 0; i < 5; ++i) printf(" %7.2f", (double)(h[8 + 8 * w + i] - t0) * 0.01);
            printf("\n");
        }
        fflush(stdout);
        out3[0] = out3[1] = out3[2] = 0;
        return MCML_OK;
    }
    for (int i = 0; i < 3; i++)
        MCML_TRY((launch_gemm_band<EpiAxpby, true>(s, plan, N, dA.d(), dA.ld, dB.d(), dB.ld, epi)));
    hipEvent_t e0, e1;
    MCML_HIP(hipEventCreate(&e0));
    MCML_HIP(hipEventCreate(&e1));
    MCML_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++)
        MCML_TRY((launch_gemm_band<EpiAxpby, true>(s, plan, N, dA.d(), dA.ld, dB.d(), dB.ld, epi, clk.as<unsigned long long>(), mode)));
    MCML_HIP(hipEventRecord(e1, s));
    MCML_HIP(hipEventSynchronize(e1));
    float ms = 0;
    MCML_HIP(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc[2] = {0, 0};
    MCML_HIP(hipMemcpy(hc, clk.p, 16, hipMemcpyDeviceToHost));
    out3[0] = ms / iters;
    out3[1] = hc[1] ? (double)hc[0] / (double)hc[1] * 100.0 : 0.0;
    out3[2] = 2.0 * 80.0 * 32.0 * tiles * N / (ms / iters * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MCML_OK;
}

extern "C" const char* glmmr_mcml_last_error(void) { return mcml::last_error(); }

// ---- RNG contract on the device (bitwise twins of oracle/mcml_oracle.c) ----
#include "rng.h"
__global__ void k_dbg_normals(uint64_t seed, uint32_t chain, uint32_t prop, uint32_t tag, int n, double* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rng_normal(seed, (uint32_t)i, chain, prop, tag);
}
__global__ void k_dbg_minstd(uint32_t seed, int n, double* out)
{
    if (threadIdx.x || blockIdx.x) return;
    uint32_t x = seed % 2147483647u;
    if (x == 0) x = 1;
    for (int i = 0; i < n; ++i) out[i] = minstd_canonical(x);
}
extern "C" int glmmr_mcml_dbg_normals(uint64_t seed, uint32_t chain, uint32_t prop, uint32_t tag, int n,
                                      double* out)
{
    DevBuf b;
    MCML_TRY(b.ensure(sizeof(double) * (size_t)n));
    hipLaunchKernelGGL(k_dbg_normals, dim3((n + 255) / 256), dim3(256), 0, nullptr, seed, chain, prop, tag, n, b.d());
    MCML_HIP(hipGetLastError());
    MCML_HIP(hipMemcpy(out, b.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return MCML_OK;
}
extern "C" int glmmr_mcml_dbg_minstd(uint32_t seed, int n, double* out)
{
    DevBuf b;
    MCML_TRY(b.ensure(sizeof(double) * (size_t)n));
    hipLaunchKernelGGL(k_dbg_minstd, dim3(1), dim3(64), 0, nullptr, seed, n, b.d());
    MCML_HIP(hipGetLastError());
    MCML_HIP(hipMemcpy(out, b.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return MCML_OK;
}

// ---- host optimiser hooks (no GPU needed) ----
#include "../../include/glmmr_mcml_c.h"
#include "optim.h"
#include <cmath>
extern "C" int glmmr_mcml_dbg_bobyqa(glmmr_mcml_objective f, void* user, int n, const double* x0,
                                     const double* lower, const double* upper, double rhobeg, double rhoend,
                                     int maxfun, double* x_out, double* f_out, int* nfev_out)
{
    MCML_REQUIRE(f && n > 0 && x0 && x_out, "dbg_bobyqa: bad argument");
    objective_fn obj = [&](const std::vector<double>& x, double* v) { *v = f(x.data(), n, user); return 0; };
    std::vector<double> x(x0, x0 + n), lo(n, -HUGE_VAL), up(n, HUGE_VAL);
    if (lower) lo.assign(lower, lower + n);
    if (upper) up.assign(upper, upper + n);
    BobyqaOpts o; o.rhobeg = rhobeg; o.rhoend = rhoend; if (maxfun > 0) o.maxfun = maxfun;
    BobyqaResult r;
    MCML_TRY(bobyqa(obj, x, lo, up, o, &r));
    for (int i = 0; i < n; ++i) x_out[i] = r.x[i];
    if (f_out) *f_out = r.fval;
    if (nfev_out) *nfev_out = r.nfev;
    return MCML_OK;
}
extern "C" int glmmr_mcml_dbg_bobyqa_batch(glmmr_mcml_objective f, void* user, int n, const double* x0,
                                           const double* lower, const double* upper, double rhobeg, double rhoend,
                                           int maxfun, int width, double* x_out, double* f_out, int* nfev_out,
                                           int* rounds_out)
{
    MCML_REQUIRE(f && n > 0 && x0 && x_out && width >= 1, "dbg_bobyqa_batch: bad argument");
    batch_objective_fn obj = [&](const std::vector<std::vector<double>>& X, std::vector<double>* F) {
        F->resize(X.size());
        for (size_t i = 0; i < X.size(); ++i) (*F)[i] = f(X[i].data(), n, user);
        return 0; };
    std::vector<double> x(x0, x0 + n), lo(n, -HUGE_VAL), up(n, HUGE_VAL);
    if (lower) lo.assign(lower, lower + n);
    if (upper) up.assign(upper, upper + n);
    BobyqaOpts o; o.rhobeg = rhobeg; o.rhoend = rhoend; if (maxfun > 0) o.maxfun = maxfun;
    BobyqaResult r;
    MCML_TRY(bobyqa_batch(obj, x, lo, up, o, width, &r));
    for (int i = 0; i < n; ++i) x_out[i] = r.x[i];
    if (f_out) *f_out = r.fval;
    if (nfev_out) *nfev_out = r.nfev;
    if (rounds_out) *rounds_out = r.rounds;
    return MCML_OK;
}
// the same with a BATCH callback: fb(X /* n x k, column-major */, n, k, F /* k values out */, user) evaluates a whole round --
// what a rank of a sharded job does (its own candidates, then the exchange); tests/test_dist_gloo.py runs it over gloo
extern "C" int glmmr_mcml_dbg_bobyqa_rounds(glmmr_mcml_batch_objective fb, void* user, int n, const double* x0,
                                            const double* lower, const double* upper, double rhobeg, double rhoend,
                                            int maxfun, int width, double* x_out, double* f_out, int* nfev_out,
                                            int* rounds_out)
{
    MCML_REQUIRE(fb && n > 0 && x0 && x_out && width >= 1, "dbg_bobyqa_rounds: bad argument");
    batch_objective_fn obj = [&](const std::vector<std::vector<double>>& X, std::vector<double>* F) {
        const int k = (int)X.size();
        std::vector<double> flat((size_t)n * k);
        for (int j = 0; j < k; ++j) for (int i = 0; i < n; ++i) flat[(size_t)j * n + i] = X[j][i];
        F->assign(k, 0.0);
        return fb(flat.data(), n, k, F->data(), user); };
    std::vector<double> x(x0, x0 + n), lo(n, -HUGE_VAL), up(n, HUGE_VAL);
    if (lower) lo.assign(lower, lower + n);
    if (upper) up.assign(upper, upper + n);
    BobyqaOpts o; o.rhobeg = rhobeg; o.rhoend = rhoend; if (maxfun > 0) o.maxfun = maxfun;
    BobyqaResult r;
    MCML_TRY(bobyqa_batch(obj, x, lo, up, o, width, &r));
    for (int i = 0; i < n; ++i) x_out[i] = r.x[i];
    if (f_out) *f_out = r.fval;
    if (nfev_out) *nfev_out = r.nfev;
    if (rounds_out) *rounds_out = r.rounds;
    return MCML_OK;
}
extern "C" int glmmr_mcml_dbg_fd_hessian(glmmr_mcml_objective f, void* user, int n, const double* x, double ndeps,
                                         int usebounds, const double* lower, const double* upper, double* H)
{
    MCML_REQUIRE(f && n > 0 && x && H, "dbg_fd_hessian: bad argument");
    objective_fn obj = [&](const std::vector<double>& xx, double* v) { *v = f(xx.data(), n, user); return 0; };
    std::vector<double> xv(x, x + n), nd(n, ndeps), lo(n, -HUGE_VAL), up(n, HUGE_VAL), h;
    if (lower) lo.assign(lower, lower + n);
    if (upper) up.assign(upper, upper + n);
    MCML_TRY(fd_hessian(obj, xv, nd, usebounds != 0, lo, up, &h));
    for (int i = 0; i < n * n; ++i) H[i] = h[i];
    return MCML_OK;
}

extern "C" int glmmr_mcml_dbg_copy_roundtrip(const double* in, long long ld_in, double* out, long long ld_out, long long rows, long long cols)
{
    MCML_REQUIRE(in && out && rows > 0 && cols > 0 && ld_in >= rows && ld_out >= rows, "copy_roundtrip: bad arguments");
    const size_t ldd = (size_t)rows + 3;                 // an odd device pitch
    DevBuf buf;
    MCML_TRY(buf.ensure(sizeof(double) * ldd * (size_t)cols));
    MCML_HIP(hipMemset(buf.p, 0xff, sizeof(double) * ldd * (size_t)cols));
    MCML_TRY(copy_h2d_2d(buf.p, sizeof(double) * ldd, in, sizeof(double) * (size_t)ld_in, sizeof(double) * (size_t)rows, (size_t)cols, 0));
    MCML_TRY(copy_d2h_2d(out, sizeof(double) * (size_t)ld_out, buf.p, sizeof(double) * ldd, sizeof(double) * (size_t)rows, (size_t)cols, 0));
    MCML_HIP(hipDeviceSynchronize());
    return MCML_OK;
}
