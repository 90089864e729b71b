// common.h -- error plumbing and device-buffer helpers for libglmmr_mcml_hip.
// gfx950 (MI355X) only; no CUDA shims, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mcml {

// error codes returned through the C ABI (never throw across it)
enum : int {
    MCML_OK = 0,
    MCML_EINVAL = -1,       // bad argument / shape
    MCML_EUNSUPPORTED = -2, // family/link or covariance function not built
    MCML_ENOTPD = -3,       // covariance block not positive definite
    MCML_ESINGULAR = -4,    // singular X'WX in MCNR
    MCML_EHIP = -5,         // HIP runtime error
    MCML_ENODEVICE = -6,    // no MI355X visible: the product has no CPU fallback
    MCML_ENOMEM = -7,
};

void set_error(const char* fmt, ...);
const char* last_error();

#define MCML_HIP(expr)                                                              \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            ::mcml::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),\
                              __FILE__, __LINE__);                                  \
            return ::mcml::MCML_EHIP;                                               \
        }                                                                           \
    } while (0)

#define MCML_TRY(expr)                       \
    do {                                     \
        int _rc = (expr);                    \
        if (_rc != ::mcml::MCML_OK) return _rc; \
    } while (0)

#define MCML_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            ::mcml::set_error(__VA_ARGS__);  \
            return ::mcml::MCML_EINVAL;      \
        }                                    \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that needs more than 64 KB of LDS: set once per
// (kernel, device) -- a process may hold contexts on several GPUs
int ensure_dynamic_lds(const void* kernel, int bytes);

// launch KERNEL<FL>: the common families get their own instantiation (1 poisson/log, 3 binomial/logit, 7 gaussian/identity)
#define MCML_FL_DISPATCH(FLINK, KERNEL, ...)                                          \
    do {                                                                              \
        switch (FLINK) {                                                              \
        case 1: hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__); break;                  \
        case 3: hipLaunchKernelGGL((KERNEL<3>), __VA_ARGS__); break;                  \
        case 7: hipLaunchKernelGGL((KERNEL<7>), __VA_ARGS__); break;                  \
        default: hipLaunchKernelGGL((KERNEL<0>), __VA_ARGS__); break;                 \
        }                                                                             \
    } while (0)


static inline int round_up(int x, int a) { return (x + a - 1) / a * a; }
static inline size_t round_up_sz(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Leading dimension used for every device matrix: a multiple of 32 doubles
// (256 B) so that any column starts on a full cache line and 16-byte vector
// loads of row pairs are always aligned.
static inline int pad_ld(int rows) { return round_up(rows < 1 ? 1 : rows, 32); }

// Owning device allocation.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
    ~DevBuf() { release(); }
    void release() {
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    }
    int ensure(size_t nbytes) {
        if (nbytes <= bytes) return MCML_OK;
        release();
        // +256 B slack: vector loads at a ragged edge may touch the pad
        hipError_t e = hipMalloc(&p, nbytes + 256);
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(e));
            return MCML_ENOMEM;
        }
        bytes = nbytes;
        return MCML_OK;
    }
    double* d() const { return static_cast<double*>(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// Column-major device matrix with padded leading dimension.
struct DevMat {
    DevBuf buf;
    int rows = 0, cols = 0, ld = 0;
    int cols_alloc = 0;     // columns actually allocated (>= cols; round_up(cols, col_align))
    int alloc(int r, int c, int col_align = 1) {
        int nld = pad_ld(r);
        int ca = round_up(c < 1 ? 1 : c, col_align);
        MCML_TRY(buf.ensure(sizeof(double) * (size_t)nld * (size_t)ca));
        rows = r; cols = c; ld = nld; cols_alloc = ca;
        return MCML_OK;
    }
    double* d() const { return buf.d(); }
    double* at(int i, int j) const { return buf.d() + i + (size_t)j * ld; }
};

// Host <-> device copies of CALLER (or temporary) host memory.  Everything above a few KB goes through a pinned staging
// buffer the library owns, in chunks, and is complete when the call returns -- so that the runtime never pins pageable
// memory it does not own: a pinned registration of memory the caller (or a destroyed std::vector) later unmaps makes the
// kernel driver evict and restore the process's queues, a 60-70 ms stall of whatever HIP call comes next (found as config
// 4's "slow first repetition").  Below the threshold: hipMemcpyAsync, which the runtime stages itself;
// for device -> host the caller synchronises the stream as before.
int copy_h2d(void* dev, const void* host, size_t bytes, hipStream_t s);
int copy_d2h(void* host, const void* dev, size_t bytes, hipStream_t s);
// the same for `cols` columns of `rowbytes` bytes each with a pitch on either side
int copy_h2d_2d(void* dev, size_t dpitch, const void* host, size_t hpitch, size_t rowbytes, size_t cols, hipStream_t s);
int copy_d2h_2d(void* host, size_t hpitch, const void* dev, size_t dpitch, size_t rowbytes, size_t cols, hipStream_t s);

// host (ldh) <-> device (padded ld) copies of a column-major matrix
int upload_matrix(DevMat& dst, const double* host, int rows, int cols, int ldh, hipStream_t s);
int download_matrix(double* host, int ldh, const double* dev, int ldd, int rows, int cols,
                    hipStream_t s);

}  // namespace mcml
