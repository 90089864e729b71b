/*
 * glmmr_mcml_c.h -- C ABI of libglmmr_mcml_hip.so, the MI355X (gfx950) build of
 * glmmrMCML's MCML inner loop.  Plain pointers and sizes only; every matrix is
 * column-major float64, every integer array int32, exactly as R hands them to
 * the reference's Rcpp exports (src/RcppExports.cpp:15-310).
 *
 * Conventions: every function returns 0 or a negative error code and never
 * throws or calls the R API; glmmr_mcml_last_error() gives the text.  The
 * caller owns every buffer it passes.  There is no CPU fallback: without a
 * visible gfx950 device compute entry points return GLMMR_MCML_ENODEVICE.
 */
#ifndef GLMMR_MCML_C_H
#define GLMMR_MCML_C_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLMMR_MCML_OK            0
#define GLMMR_MCML_EINVAL       -1
#define GLMMR_MCML_EUNSUPPORTED -2   /* reference: unordered_map::at throws (mcmlmodel.h:89) */
#define GLMMR_MCML_ENOTPD       -3   /* reference does not check (SURVEY 8b "Errors") */
#define GLMMR_MCML_ESINGULAR    -4
#define GLMMR_MCML_EHIP         -5
#define GLMMR_MCML_ENODEVICE    -6
#define GLMMR_MCML_ENOMEM       -7

/* text of the calling thread's last error; replaces Rcpp's BEGIN_RCPP/END_RCPP
 * exception -> R condition translation (src/RcppExports.cpp:18,43) */
const char* glmmr_mcml_last_error(void);

/* ---- test hooks (building blocks exposed for tests/ and bench.py only) ---- */
int glmmr_mcml_dbg_dgemm(int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                         int b_nmajor, double alpha, double beta, double* C, int ldc,
                         int lower_only, int force_tile /* -1 = auto */);
int glmmr_mcml_dbg_dgemm_bench(int M, int N, int K, int b_nmajor, int iters, int force_tile,
                               double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif
