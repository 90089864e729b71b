// covspec.h -- the (cov, data, eff_range) triple that glmmrBase's
// Covariance$get_D_data() hands to every Rcpp export (src/mcml_optim.cpp:20-23),
// parsed once on the host and mirrored on the device.
//
// cov: int32 rows x 5 column-major = (block id, block dim, function id,
// n variables, parameter index); data: every block's (dim x nvar_total) data
// matrix flattened column-major and concatenated.
//
// Covariance-function table (ids follow the parameter-count vector
// c(1,1,1,2,2,1,2,2,2,2,2,2,2,1), R6ModelExtMCML.R:430; id 1 = gr is certain from
// mcmldmatrix.h:61-65, the others are INFERRED -- glmmrBase is not in the image):
//   1 gr     d==0 ? v*t^2 : 0        2 fexp0  v*exp(-d/t)
//   3 ar1    v*t^d                   4 sqexp  v*t0*exp(-d^2/t1^2)
//   7 fexp   v*t0*exp(-d/t1)        14 sqexp0 v*exp(-d^2/t^2)
// 5,6,8..13 (matern, bessel, wendland, prod*) -> MCML_EUNSUPPORTED.
#pragma once
#include "common.h"

namespace mcml {

constexpr int MAX_COV_PAR = 32;
struct ThetaArg { double v[MAX_COV_PAR]; };

struct CovBlock {
    int dim;        // block dimension
    int r0, r1;     // rows [r0, r1) of cov
    int ncol;       // total variables (columns of the block's data matrix)
    int doff;       // offset of the block's data in `data`
    int matstart;   // first random-effect index of the block
    int all_gr;     // every function id == 1 -> diagonal fast path (mcmldmatrix.h:61)
};

struct CovSpec {
    int rows = 0, B = 0, N = 0, npar = 0;
    std::vector<int32_t> cov;   // rows x 5 column-major
    std::vector<double> data;
    std::vector<double> eff;
    std::vector<CovBlock> blocks;

    static int fn_npar(int fn) {
        static const int t[15] = {0, 1, 1, 1, 2, 2, 1, 2, 2, 2, 2, 2, 2, 2, 1};
        return (fn >= 1 && fn <= 14) ? t[fn] : -1;
    }
    static bool fn_built(int fn) { return fn == 1 || fn == 2 || fn == 3 || fn == 4 || fn == 7 || fn == 14; }
    int c(int r, int col) const { return cov[r + (size_t)col * rows]; }

    int parse(const int32_t* cov_, int rows_, const double* data_, int data_len,
              const double* eff_, int eff_len)
    {
        MCML_REQUIRE(cov_ && rows_ > 0, "cov: empty");
        rows = rows_;
        cov.assign(cov_, cov_ + (size_t)rows * 5);
        eff.assign(eff_ ? eff_ : nullptr, eff_ ? eff_ + (eff_len > 0 ? eff_len : 0) : nullptr);
        blocks.clear();
        N = 0; npar = 0;
        int r = 0;
        size_t doff = 0;
        while (r < rows) {
            CovBlock b{};
            b.dim = c(r, 1); b.r0 = r; b.ncol = 0; b.all_gr = 1; b.matstart = N; b.doff = (int)doff;
            MCML_REQUIRE(b.dim > 0, "cov row %d: block dimension %d", r, b.dim);
            int id = c(r, 0);
            while (r < rows && c(r, 0) == id) {
                int fn = c(r, 2), nv = c(r, 3), pi = c(r, 4);
                MCML_REQUIRE(c(r, 1) == b.dim, "cov row %d: dimension differs within block", r);
                if (fn_npar(fn) < 0 || !fn_built(fn)) {
                    set_error("cov row %d: covariance function id %d is not built (gr, fexp0, ar1, "
                              "sqexp, fexp, sqexp0 only)", r, fn);
                    return MCML_EUNSUPPORTED;
                }
                MCML_REQUIRE(nv >= 0 && pi >= 0, "cov row %d: negative field", r);
                MCML_REQUIRE(pi + fn_npar(fn) <= MAX_COV_PAR, "more than %d covariance parameters", MAX_COV_PAR);
                if (pi + fn_npar(fn) > npar) npar = pi + fn_npar(fn);
                if (fn != 1) b.all_gr = 0;
                b.ncol += nv;
                ++r;
            }
            b.r1 = r;
            doff += (size_t)b.dim * b.ncol;
            N += b.dim;
            blocks.push_back(b);
        }
        B = (int)blocks.size();
        MCML_REQUIRE((size_t)data_len >= doff, "data has %d values, the blocks need %zu", data_len, doff);
        data.assign(data_, data_ + doff);
        return MCML_OK;
    }
};

// one term of the product kernel; identical expression order to the oracle
__host__ __device__ static inline double cov_term(int fn, double dist, const double* g, double val)
{
    switch (fn) {
    case 1: return (dist == 0) ? val * g[0] * g[0] : 0.0;
    case 2: return val * exp(-1.0 * dist / g[0]);
    case 3: return val * pow(g[0], dist);
    case 4: return val * g[0] * exp(-1.0 * dist * dist / (g[1] * g[1]));
    case 7: return val * g[0] * exp(-1.0 * dist / g[1]);
    case 14: return val * exp(-1.0 * dist * dist / (g[0] * g[0]));
    }
    return val;
}

}  // namespace mcml
