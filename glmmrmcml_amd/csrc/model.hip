// model.hip -- mcmlModel on the device (mcmlmodel.h:28-307): xb = X beta, the
// cached Z u, ZL = Z L, the Monte-Carlo log-likelihood (log_likelihood,
// :284-304; functor L_likelihood, likelihood.h:48-65) and the sufficient
// statistics of the MCNR step (mcmloptim.h:198-236).
//
// What the reference recomputes, this keeps: Z u is formed once per set of
// samples (the reference redoes the n x Q x m GEMM on every objective
// evaluation, mcmlmodel.h:286, and once per sample inside mcnr, :121); W stays
// a diagonal (the reference materialises a dense n x n matrix, :62).
// Z is held dense for the MFMA products and, when it is an indicator-like
// matrix, also as padded-CSR rows so that Z u and Z L are row gathers.
#include "ctx.h"
#include "dgemm_mfma.h"
#include "dgemm_band.h"
#include "glm.h"
#include "reduce.h"

namespace mcml {

// ------------------------------------------------------------------ setup
int model_setup(Ctx& c, const double* Z, const double* X, const double* y)
{
    const int n = c.n, Q = c.Q, P = c.P;
    MCML_TRY(upload_matrix(c.Z, Z, n, Q, n, c.stream));
    MCML_TRY(upload_matrix(c.X, X, n, P, n, c.stream));
    MCML_TRY(c.y.ensure(sizeof(double) * (size_t)pad_ld(n)));
    MCML_TRY(c.xb.ensure(sizeof(double) * (size_t)pad_ld(n)));
    MCML_HIP(hipMemsetAsync(c.y.p, 0, sizeof(double) * (size_t)pad_ld(n), c.stream));
    MCML_HIP(hipMemsetAsync(c.xb.p, 0, sizeof(double) * (size_t)pad_ld(n), c.stream));
    std::vector<double> yy(y, y + n);
    if (c.flink == 8)                      // mcmlmodel.h:90-92: y_ = y_.log()
        for (auto& v : yy) v = log(v);
    MCML_TRY(copy_h2d(c.y.p, yy.data(), sizeof(double) * n, c.stream));
    // sparse rows of Z (indicator designs): padded CSR, width = max nnz per row
    size_t nnz = 0; int maxrow = 0;
    std::vector<int> cnt(n, 0);
    for (int j = 0; j < Q; ++j)
        for (int i = 0; i < n; ++i)
            if (Z[i + (size_t)j * n] != 0.0) { ++cnt[i]; ++nnz; }
    for (int i = 0; i < n; ++i) if (cnt[i] > maxrow) maxrow = cnt[i];
    c.z_width = 0;
    if (maxrow > 0 && maxrow <= 8 && (double)nnz <= 0.02 * (double)n * Q + 8.0 * n) {
        c.z_width = maxrow;
        std::vector<int> zi((size_t)n * maxrow, 0);
        std::vector<double> zv((size_t)n * maxrow, 0.0);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int j = 0; j < Q; ++j)
            for (int i = 0; i < n; ++i) {
                double v = Z[i + (size_t)j * n];
                if (v != 0.0) { int k = cnt[i]++; zi[i + (size_t)k * n] = j; zv[i + (size_t)k * n] = v; }
            }
        c.h_zidx = zi; c.h_zval = zv;
        MCML_TRY(c.z_idx.ensure(sizeof(int) * zi.size()));
        MCML_TRY(c.z_val.ensure(sizeof(double) * zv.size()));
        MCML_TRY(copy_h2d(c.z_idx.p, zi.data(), sizeof(int) * zi.size(), c.stream));
        MCML_TRY(copy_h2d(c.z_val.p, zv.data(), sizeof(double) * zv.size(), c.stream));
        MCML_HIP(hipStreamSynchronize(c.stream));
    }
    MCML_HIP(hipStreamSynchronize(c.stream));
    return MCML_OK;
}

// ------------------------------------------------------------------ xb = X beta
__global__ void k_xb(const double* X, int ldx, int n, int P, const double* beta, double* xb)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0;
    for (int p = 0; p < P; ++p) s += X[i + (size_t)p * ldx] * beta[p];   // Eigen: X * beta
    xb[i] = s;
}

int model_update_beta(Ctx& c, const double* beta)
{
    MCML_REQUIRE(c.n > 0, "no model in this context");
    MCML_TRY(c.scratch.ensure(sizeof(double) * 64));
    MCML_REQUIRE(c.P <= 64, "P > 64 fixed effects");
    MCML_TRY(copy_h2d(c.scratch.p, beta, sizeof(double) * c.P, c.stream));
    hipLaunchKernelGGL(k_xb, dim3((c.n + 255) / 256), dim3(256), 0, c.stream, c.X.d(), c.X.ld, c.n, c.P,
                       c.scratch.d(), c.xb.d());
    MCML_HIP(hipGetLastError());
    MCML_HIP(hipStreamSynchronize(c.stream));    // beta is a caller buffer
    return MCML_OK;
}

// ------------------------------------------------------------------ Z * B (row gather or MFMA)
// out[i, j] = sum_k zval[i,k] * B[zidx[i,k], j]
__global__ __launch_bounds__(256) void k_zgather(const int* zidx, const double* zval, int width, int n,
                                                 const double* B, int ldb, int ncols, double* out, int ldo)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
        double s = 0;
        for (int k = 0; k < width; ++k) s += zval[i + (size_t)k * n] * B[zidx[i + (size_t)k * n] + (size_t)j * ldb];
        out[i + (size_t)j * ldo] = s;
    }
}

static int z_times(Ctx& c, const double* B, int ldb, int ncols, double* out, int ldo)
{
    if (c.z_width > 0) {
        int gy = ncols < 1024 ? ncols : 1024;
        hipLaunchKernelGGL(k_zgather, dim3((c.n + 255) / 256, gy), dim3(256), 0, c.stream, c.z_idx.as<int>(),
                           c.z_val.d(), c.z_width, c.n, B, ldb, ncols, out, ldo);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
    EpiAxpby epi{out, ldo, 1.0, 0.0};
    return launch_gemm<false>(c.stream, c.n, ncols, c.Q, c.Z.d(), c.Z.ld, B, ldb, epi);
}

// zu_ = Z * u (mcmlmodel.h:116-118), cached until the samples change
int model_update_zu(Ctx& c)
{
    MCML_REQUIRE(c.n > 0 && c.mcols > 0, "update_zu: no model / samples");
    if (c.zu_valid) return MCML_OK;
    MCML_TRY(c.ZU.alloc(c.n, c.mcols));
    MCML_TRY(z_times(c, c.U.d(), c.U.ld, c.mcols, c.ZU.d(), c.ZU.ld));
    c.zu_valid = true;
    return MCML_OK;
}

__global__ void k_transpose(const double* A, int lda, int rows, int cols, double* AT, int ldt)
{
    __shared__ double tile[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int i = bx + tx, j = by + r;
        tile[r][tx] = (i < rows && j < cols) ? A[i + (size_t)j * lda] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int j = by + tx, i = bx + r;
        if (i < rows && j < cols) AT[j + (size_t)i * ldt] = tile[tx][r];
    }
}

// ---- sparse ZL (configs whose D has only diagonal / small blocks and whose Z is indicator-like)
__global__ void k_ell_fill(const int* src, const double* z, const double* L, long total, double* val)
{
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) val[e] = z[e] * L[src[e]];
}

// the same values in the order of the rows of ZL' (CSR): the backward product reads them without the
// position indirection
__global__ void k_csr_fill(const int* pos, const double* ell_val, long nnz, double* csr_val)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nnz) csr_val[t] = ell_val[pos[t]];
}

static int sparse_zl_setup(Ctx& c)
{
    SparseZL& sp = c.sp;
    sp.built = true; sp.possible = false;
    if (const char* e = getenv("GLMMR_MCML_ZL")) if (!strcmp(e, "dense")) return MCML_OK;
    if (c.z_width <= 0 || c.cov.B <= 0 || c.maxdim_large > 0) return MCML_OK;
    if ((double)c.Q * pad_ld(c.Q) >= 2.0e9) return MCML_OK;           // flat index into L must fit an int
    const int n = c.n, Q = c.Q, zw = c.z_width, ldL = pad_ld(Q);
    std::vector<int> blk_of(Q), start_of(Q);
    for (int b = 0; b < c.cov.B; ++b)
        for (int k = 0; k < c.cov.blocks[b].dim; ++k) { blk_of[c.cov.blocks[b].matstart + k] = b; start_of[c.cov.blocks[b].matstart + k] = c.cov.blocks[b].matstart; }
    // row i of ZL = sum over the nonzeros (j, z) of row i of Z of z * L[j, start(j) .. j]
    int W = 0; long nnz = 0;
    std::vector<int> width(n, 0);
    for (int i = 0; i < n; ++i) {
        int w = 0;
        for (int k = 0; k < zw; ++k) { if (c.h_zval[i + (size_t)k * n] == 0.0) continue; int j = c.h_zidx[i + (size_t)k * n]; w += j - start_of[j] + 1; }
        width[i] = w; nnz += w; if (w > W) W = w;
    }
    if (W <= 0 || W > 64) return MCML_OK;
    std::vector<int> col((size_t)n * W, 0), src((size_t)n * W, 0);
    std::vector<double> zz((size_t)n * W, 0.0);
    std::vector<int> cnt(Q, 0);
    for (int i = 0; i < n; ++i) {
        int w = 0;
        for (int k = 0; k < zw; ++k) {
            double z = c.h_zval[i + (size_t)k * n];
            if (z == 0.0) continue;
            int j = c.h_zidx[i + (size_t)k * n];
            for (int t = start_of[j]; t <= j; ++t) { col[i + (size_t)w * n] = t; src[i + (size_t)w * n] = j + t * ldL; zz[i + (size_t)w * n] = z; ++cnt[t]; ++w; }
        }
    }
    std::vector<int> ptr(Q + 1, 0);
    for (int q = 0; q < Q; ++q) ptr[q + 1] = ptr[q] + cnt[q];
    std::vector<int> ci((size_t)nnz), cp((size_t)nnz), fill(ptr.begin(), ptr.end() - 1);
    for (int i = 0; i < n; ++i)
        for (int w = 0; w < width[i]; ++w) { int q = col[i + (size_t)w * n]; int t = fill[q]++; ci[t] = i; cp[t] = i + w * n; }
    const size_t tot = (size_t)n * W;
    MCML_TRY(sp.ell_col.ensure(sizeof(int) * tot)); MCML_TRY(sp.ell_src.ensure(sizeof(int) * tot));
    MCML_TRY(sp.ell_z.ensure(sizeof(double) * tot)); MCML_TRY(sp.ell_val.ensure(sizeof(double) * tot));
    MCML_TRY(sp.csr_ptr.ensure(sizeof(int) * (size_t)(Q + 1))); MCML_TRY(sp.csr_i.ensure(sizeof(int) * (size_t)(nnz + 1)));
    MCML_TRY(sp.csr_pos.ensure(sizeof(int) * (size_t)(nnz + 1)));
    MCML_TRY(copy_h2d(sp.ell_col.p, col.data(), sizeof(int) * tot, c.stream));
    MCML_TRY(copy_h2d(sp.ell_src.p, src.data(), sizeof(int) * tot, c.stream));
    MCML_TRY(copy_h2d(sp.ell_z.p, zz.data(), sizeof(double) * tot, c.stream));
    MCML_TRY(copy_h2d(sp.csr_ptr.p, ptr.data(), sizeof(int) * (size_t)(Q + 1), c.stream));
    MCML_TRY(copy_h2d(sp.csr_i.p, ci.data(), sizeof(int) * (size_t)nnz, c.stream));
    MCML_TRY(copy_h2d(sp.csr_pos.p, cp.data(), sizeof(int) * (size_t)nnz, c.stream));
    MCML_TRY(sp.row_start.ensure(sizeof(int) * (size_t)Q));
    MCML_TRY(copy_h2d(sp.row_start.p, start_of.data(), sizeof(int) * (size_t)Q, c.stream));
    // the factored form: rows of Z' (CSR) and the end of each column of L
    {
        std::vector<int> zcnt(Q, 0), end_of(Q);
        long nz = 0, nl = 0;
        for (int b = 0; b < c.cov.B; ++b) {
            const int d = c.cov.blocks[b].dim, m0 = c.cov.blocks[b].matstart;
            for (int k = 0; k < d; ++k) end_of[m0 + k] = m0 + d;
            nl += (long)d * (d + 1) / 2;
        }
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < zw; ++k) if (c.h_zval[i + (size_t)k * n] != 0.0) { ++zcnt[c.h_zidx[i + (size_t)k * n]]; ++nz; }
        std::vector<int> zptr(Q + 1, 0);
        for (int q = 0; q < Q; ++q) zptr[q + 1] = zptr[q] + zcnt[q];
        std::vector<int> zi((size_t)nz + 1), zfill(zptr.begin(), zptr.end() - 1);
        std::vector<double> zv((size_t)nz + 1);
        for (int i = 0; i < n; ++i)                                    // ascending observation within a row
            for (int k = 0; k < zw; ++k) {
                const double z = c.h_zval[i + (size_t)k * n];
                if (z == 0.0) continue;
                const int t = zfill[c.h_zidx[i + (size_t)k * n]]++;
                zi[t] = i; zv[t] = z;
            }
        MCML_TRY(sp.zcsr_ptr.ensure(sizeof(int) * (size_t)(Q + 1))); MCML_TRY(sp.zcsr_i.ensure(sizeof(int) * (size_t)(nz + 1)));
        MCML_TRY(sp.zcsr_val.ensure(sizeof(double) * (size_t)(nz + 1))); MCML_TRY(sp.row_end.ensure(sizeof(int) * (size_t)Q));
        MCML_TRY(copy_h2d(sp.zcsr_ptr.p, zptr.data(), sizeof(int) * (size_t)(Q + 1), c.stream));
        MCML_TRY(copy_h2d(sp.zcsr_i.p, zi.data(), sizeof(int) * (size_t)nz, c.stream));
        MCML_TRY(copy_h2d(sp.zcsr_val.p, zv.data(), sizeof(double) * (size_t)nz, c.stream));
        MCML_TRY(copy_h2d(sp.row_end.p, end_of.data(), sizeof(int) * (size_t)Q, c.stream));
        sp.nnz_z = nz; sp.nnz_l = nl;
        // the blocks in row order, for the kernel that takes a whole block per wave (contiguous and covering 0 .. Q, or not used)
        {
            std::vector<int> bp;
            int at = 0; bool ok = true;
            sp.max_blk = 0;
            for (int b = 0; b < c.cov.B && ok; ++b) {
                if (c.cov.blocks[b].matstart != at) ok = false;
                bp.push_back(at); at += c.cov.blocks[b].dim;
                if (c.cov.blocks[b].dim > sp.max_blk) sp.max_blk = c.cov.blocks[b].dim;
            }
            bp.push_back(at);
            sp.nblk = (ok && at == Q) ? c.cov.B : 0;
            if (sp.nblk) {
                MCML_TRY(sp.blk_ptr.ensure(sizeof(int) * bp.size()));
                MCML_HIP(hipMemcpy(sp.blk_ptr.p, bp.data(), sizeof(int) * bp.size(), hipMemcpyHostToDevice));
            }
        }
        // entries gathered per chain and leapfrog step: nnz(ZL) against nnz(Z) + nnz(L) + the extra pass over Q
        sp.factored = 4 * (nz + nl + 2L * Q) < 3 * nnz;
        if (const char* e = getenv("GLMMR_MCML_ZL")) {
            if (!strcmp(e, "factored")) sp.factored = true;
            if (!strcmp(e, "product")) sp.factored = false;
        }
    }
    MCML_HIP(hipStreamSynchronize(c.stream));
    sp.W = W; sp.nnz = nnz; sp.possible = true;
    return MCML_OK;
}

// ZL_ = Z * L (mcmlmodel.h:104-106): dense with its transpose (so that both HMC products read
// their A operand M-contiguous), or the ELL/CSR pair when ZL is sparse
int model_update_L(Ctx& c)
{
    MCML_REQUIRE(c.n > 0 && c.have_L, "update_L: no model / L");
    if (!c.sp.built) MCML_TRY(sparse_zl_setup(c));
    if (c.sp.possible && !c.no_sparse_zl && !c.l_foreign && c.L.ld == pad_ld(c.Q)) {
        const long tot = (long)c.n * c.sp.W;
        hipLaunchKernelGGL(k_ell_fill, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c.stream,
                           c.sp.ell_src.as<int>(), c.sp.ell_z.d(), c.L.d(), tot, c.sp.ell_val.d());
        MCML_TRY(c.sp.csr_val.ensure(sizeof(double) * (size_t)(c.sp.nnz + 1)));
        hipLaunchKernelGGL(k_csr_fill, dim3((unsigned)((c.sp.nnz + 255) / 256)), dim3(256), 0, c.stream,
                           c.sp.csr_pos.as<int>(), c.sp.ell_val.d(), c.sp.nnz, c.sp.csr_val.d());
        MCML_HIP(hipGetLastError());
        c.sp.active = true;
        return MCML_OK;
    }
    c.sp.active = false;
    // K-padding columns (zero) so that the direct-to-LDS GEMM needs no K guards
    if (c.ZL.rows != c.n || c.ZL.cols != c.Q || !c.ZL.d()) {
        MCML_TRY(c.ZL.alloc(c.n, c.Q, 32));
        MCML_TRY(c.ZLT.alloc(c.Q, c.n, 32));
        MCML_HIP(hipMemsetAsync(c.ZL.d(), 0, sizeof(double) * (size_t)c.ZL.ld * c.ZL.cols_alloc, c.stream));
        MCML_HIP(hipMemsetAsync(c.ZLT.d(), 0, sizeof(double) * (size_t)c.ZLT.ld * c.ZLT.cols_alloc, c.stream));
    }
    if (c.z_width > 0) {
        MCML_TRY(z_times(c, c.L.d(), c.L.ld, c.Q, c.ZL.d(), c.ZL.ld));
    } else {
        EpiAxpby epi{c.ZL.d(), c.ZL.ld, 1.0, 0.0};
        MCML_TRY(launch_gemm<false>(c.stream, c.n, c.Q, c.Q, c.Z.d(), c.Z.ld, c.L.d(), c.L.ld, epi));
    }
    hipLaunchKernelGGL(k_transpose, dim3((c.n + 31) / 32, (c.Q + 31) / 32), dim3(256), 0, c.stream, c.ZL.d(),
                       c.ZL.ld, c.n, c.Q, c.ZLT.d(), c.ZLT.ld);
    MCML_HIP(hipGetLastError());
    // structural zeros of ZL / ZL' (triangular when Z = I): record each 80-row band's range of
    // nonzero K tiles; the banded kernel is used when it skips at least a fifth of the tiles
    c.band_fwd = c.band_bwd = false;
    const char* env = getenv("GLMMR_MCML_GEMM");
    if (!(env && (!strcmp(env, "reg") || !strcmp(env, "dlds")))) {
        auto ranges = [&](const DevMat& A, int M, int K, BandPlan& plan, bool& use, long& ntiles) -> int {
            const int nb = (M + BD_BM - 1) / BD_BM;
            MCML_TRY(c.kr_scratch.ensure(sizeof(int) * 2 * (size_t)nb));
            hipLaunchKernelGGL(k_band_ranges, dim3(nb), dim3(256), 0, c.stream, A.d(), A.ld, M, K, c.kr_scratch.as<int>());
            MCML_HIP(hipGetLastError());
            std::vector<int> h(2 * (size_t)nb);
            MCML_TRY(copy_d2h(h.data(), c.kr_scratch.p, sizeof(int) * h.size(), c.stream));
            MCML_HIP(hipStreamSynchronize(c.stream));
            // the K-tile ranges of a triangular ZL do not change from one MCML iteration to the next: keep the device
            // plans (work lists, partial-tile buffers) unless they did
            if (!(plan.M == M && plan.K == K && plan.kr == h))
                plan.reset(M, K, h);           // the per-chain-count decompositions are rebuilt on first use
            const long dense = (long)nb * ((K + BD_BK - 1) / BD_BK);
            use = (env && !strcmp(env, "band")) || plan.tiles * 5 <= dense * 4;
            ntiles = plan.tiles;
            return MCML_OK;
        };
        MCML_TRY(ranges(c.ZL, c.n, c.Q, c.plan_fwd, c.band_fwd, c.band_fwd_tiles));
        MCML_TRY(ranges(c.ZLT, c.Q, c.n, c.plan_bwd, c.band_bwd, c.band_bwd_tiles));
    }
    return MCML_OK;
}

// ------------------------------------------------------------------ log_likelihood
// FL != 0: the family / link as a compile-time constant (glm_logpdf's 12-way switch inlined costs 288 VGPRs)
template <int FL>
__global__ __launch_bounds__(256) void k_loglik(const double* ZU, int ldz, int n, int ncols, const double* xb,
                                                const double* y, double var_par, int flink_rt, double* partials)
{
    const int flink = FL ? FL : flink_rt;
    __shared__ double sh[4];
    int i = blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    if (i < n) {
        const double yi = y[i], xbi = xb[i];
        for (int j = blockIdx.y; j < ncols; j += gridDim.y)
            acc += glm_logpdf(yi, xbi + ZU[i + (size_t)j * ldz], var_par, flink);
    }
    double r = block_sum(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// sum_{j < niter} sum_i logf(y_i | xb_i + (Z u)_ij) over the local columns
int model_loglik_sum(Ctx& c, double var_par, double* sum_out)
{
    MCML_TRY(model_update_zu(c));
    int gx = (c.n + 255) / 256, gy = c.niter < 64 ? c.niter : 64;
    MCML_TRY(c.partials.ensure(sizeof(double) * (size_t)(gx * gy + 16)));
    MCML_FL_DISPATCH(c.flink, k_loglik, dim3(gx, gy), dim3(256), 0, c.stream, c.ZU.d(), c.ZU.ld, c.n, c.niter,
                       c.xb.d(), c.y.d(), var_par, c.flink, c.partials.d());
    MCML_HIP(hipGetLastError());
    MCML_TRY(device_sum(c, c.partials.d(), gx * gy, c.scalars.d() + 4));
    MCML_TRY(copy_d2h(sum_out, c.scalars.d() + 4, sizeof(double), c.stream));
    MCML_HIP(hipStreamSynchronize(c.stream));
    return MCML_OK;
}

// ------------------------------------------------------------------ MCNR statistics
// one workgroup per sample column: sigma_i = sd(y - h^-1(xb + zd_i)) (mcmloptim.h:214-216)
// FL != 0 (k_mcnr_col / k_mcnr_row): family / link as compile-time constants -- flink 1, 3, 7 are poisson/log,
// binomial/logit, gaussian/identity, whose link codes are 1, 3, 2.  With run-time codes the 16-fold unrolled body of
// k_mcnr_row is sixteen copies of three switch statements (jump tables, ~1100 basic blocks): 37 us at config 3
template <int FL>
__device__ __forceinline__ int mcnr_link(int link_code) { return FL == 1 ? 1 : FL == 3 ? 3 : FL == 7 ? 2 : link_code; }

template <int FL>
__global__ __launch_bounds__(256) void k_mcnr_col(const double* ZU, int ldz, int n, const double* xb,
                                                  const double* y, int link_code_rt, double* sig)
{
    const int link_code = mcnr_link<FL>(link_code_rt);
    __shared__ double sh[4];
    __shared__ double mean_s;
    const double* z = ZU + (size_t)blockIdx.x * ldz;
    double acc = 0;
    for (int j = threadIdx.x; j < n; j += 256) acc += y[j] - glm_mod_inv(xb[j] + z[j], link_code);
    double r = block_sum(acc, sh);
    if (threadIdx.x == 0) mean_s = r / n;
    __syncthreads();
    const double mean = mean_s;
    acc = 0;
    for (int j = threadIdx.x; j < n; j += 256) {
        double d = (y[j] - glm_mod_inv(xb[j] + z[j], link_code)) - mean;
        acc += d * d;
    }
    r = block_sum(acc, sh);
    if (threadIdx.x == 0) sig[blockIdx.x] = sqrt(r / (n - 1));
}

// wsum_j = sum_i W_i,jj ; wusum_j = sum_i W_i,jj * detadmu * resid   (mcmlmodel.h:120-134, mcmloptim.h:213-225)
// 2-D grid: (256 observations) x (a chunk of sample columns); each workgroup leaves its chunk's partial sums,
// k_mcnr_rowsum adds the chunks in order (fixed-order two-stage reduction: n/256 workgroups with a serial loop
// over all m columns ran at 1 % of the HBM rate)
constexpr int MCNR_CHUNK = 16;
template <int FL>
__global__ __launch_bounds__(256) void k_mcnr_row(const double* ZU, int ldz, int n, int ncols, const double* xb,
                                                  const double* y, int flink_rt, int link_code_rt, double nvar_par,
                                                  double* pw, double* pwu, int ldp)
{
    const int flink = FL ? FL : flink_rt;
    const int link_code = mcnr_link<FL>(link_code_rt);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const double yj = y[j], xbj = xb[j];
    const int i0 = blockIdx.y * MCNR_CHUNK, i1 = (i0 + MCNR_CHUNK < ncols) ? i0 + MCNR_CHUNK : ncols;
    double zu[MCNR_CHUNK];
#pragma unroll
    for (int u = 0; u < MCNR_CHUNK; ++u) zu[u] = ZU[j + (size_t)((i0 + u < i1) ? i0 + u : i1 - 1) * ldz];
    double a = 0, b = 0;
#pragma unroll
    for (int u = 0; u < MCNR_CHUNK; ++u)
        if (i0 + u < i1) {
            const double eta = xbj + zu[u];
            const double w = 1 / (glm_dhdmu(eta, flink) * nvar_par);
            const double resid = yj - glm_mod_inv(eta, link_code);
            a += w;
            b += w * glm_detadmu(eta, link_code) * resid;
        }
    pw[j + (size_t)blockIdx.y * ldp] = a; pwu[j + (size_t)blockIdx.y * ldp] = b;
}

__global__ __launch_bounds__(256) void k_mcnr_rowsum(const double* pw, const double* pwu, int ldp, int n, int nchunks,
                                                     double* wsum, double* wusum)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;           // one wave per workgroup: n / 64 workgroups
    if (j >= n) return;
    double a = 0, b = 0;
    int k = 0;
    for (; k + 8 <= nchunks; k += 8) {                              // eight loads of each array in flight, adds in order
        double va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { va[u] = pw[j + (size_t)(k + u) * ldp]; vb[u] = pwu[j + (size_t)(k + u) * ldp]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; }
    }
    for (; k < nchunks; ++k) { a += pw[j + (size_t)k * ldp]; b += pwu[j + (size_t)k * ldp]; }
    wsum[j] = a; wusum[j] = b;
}

// out[0 .. P*P) = X' diag(wsum) X ; out[P*P .. P*P+P) = X' wusum ; out[P*P+P] = sum sig ;
// out[P*P+P+1] = number of sample columns summed.  One workgroup per output.
__global__ __launch_bounds__(256) void k_mcnr_fin(const double* X, int ldx, int n, int P, const double* wsum,
                                                  const double* wusum, const double* sig, int ncols, double* out)
{
    __shared__ double sh[4];
    const int o = blockIdx.x;
    double acc = 0;
    if (o < P * P) {
        const int a = o % P, b = o / P;
        for (int j = threadIdx.x; j < n; j += 256) acc += X[j + (size_t)a * ldx] * wsum[j] * X[j + (size_t)b * ldx];
    } else if (o < P * P + P) {
        const int a = o - P * P;
        for (int j = threadIdx.x; j < n; j += 256) acc += X[j + (size_t)a * ldx] * wusum[j];
    } else if (o == P * P + P) {
        for (int i = threadIdx.x; i < ncols; i += 256) acc += sig[i];
    }
    const double r = block_sum(acc, sh);
    if (threadIdx.x == 0) out[o] = (o == P * P + P + 1) ? (double)ncols : r;
}

// stats (host, P*P + P + 2 doubles): sum_i X'W_iX, sum_i X'(W_i detadmu resid_i), sum_i sigma_i
// and the column count, summed over this rank's first `niter` columns and then over ranks:
// the per-chain sufficient statistics of the MCNR step, one all-reduce.
int model_mcnr_stats(Ctx& c, double var_par, double* stats)
{
    MCML_TRY(model_update_zu(c));
    const int n = c.n, P = c.P, m = c.niter;
    double nvar_par = 1.0;                       // mcmlmodel.h:123-130
    if (c.flink == 7 || c.flink == 8) nvar_par *= var_par * var_par;
    else if (c.flink >= 9 && c.flink <= 11) nvar_par *= var_par;
    else if (c.flink == 12) nvar_par *= (1 + var_par);
    const int ns = P * P + P + 2;
    const int nchunks = (m + MCNR_CHUNK - 1) / MCNR_CHUNK, ldp = pad_ld(n);
    MCML_TRY(c.partials.ensure(sizeof(double) * ((size_t)(2 + 2 * (size_t)nchunks) * ldp + m + ns + 64)));
    double* wsum = c.partials.d();
    double* wusum = wsum + ldp;
    double* sig = wusum + ldp;
    double* pw = sig + round_up(m + 16, 32);
    double* pwu = pw + (size_t)nchunks * ldp;
    MCML_TRY(c.reduce_buf.ensure(sizeof(double) * (size_t)ns));
    MCML_FL_DISPATCH(c.flink, k_mcnr_col, dim3(m), dim3(256), 0, c.stream, c.ZU.d(), c.ZU.ld, n, c.xb.d(), c.y.d(),
                     c.link_code, sig);
    MCML_FL_DISPATCH(c.flink, k_mcnr_row, dim3((n + 255) / 256, nchunks), dim3(256), 0, c.stream, c.ZU.d(), c.ZU.ld, n, m,
                     c.xb.d(), c.y.d(), c.flink, c.link_code, nvar_par, pw, pwu, ldp);
    hipLaunchKernelGGL(k_mcnr_rowsum, dim3((n + 63) / 64), dim3(64), 0, c.stream, pw, pwu, ldp, n, nchunks, wsum, wusum);
    hipLaunchKernelGGL(k_mcnr_fin, dim3(ns), dim3(256), 0, c.stream, c.X.d(), c.X.ld, n, P, wsum, wusum, sig, m,
                       c.reduce_buf.d());
    MCML_HIP(hipGetLastError());
    MCML_TRY(allreduce_dev(c, c.reduce_buf.d(), ns));              // RCCL all-reduce of the statistics (comm.hip)
    MCML_TRY(copy_d2h(stats, c.reduce_buf.p, sizeof(double) * ns, c.stream));
    MCML_HIP(hipStreamSynchronize(c.stream));
    return MCML_OK;
}

}  // namespace mcml
