// common.hip -- error string, matrix upload/download.
#include "common.h"

#include <mutex>
#include <set>
#include <utility>

namespace mcml {

int ensure_dynamic_lds(const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    MCML_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({kernel, dev})) return MCML_OK;
    MCML_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({kernel, dev});
    return MCML_OK;
}


static thread_local std::string g_err;

void set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

const char* last_error() { return g_err.c_str(); }

int upload_matrix(DevMat& dst, const double* host, int rows, int cols, int ldh, hipStream_t s)
{
    MCML_REQUIRE(host && rows >= 0 && cols >= 0 && ldh >= rows, "upload_matrix: bad shape");
    MCML_TRY(dst.alloc(rows, cols));
    if (rows == 0 || cols == 0) return MCML_OK;
    // zero the padding rows too: kernels may read (never use) them
    MCML_HIP(hipMemsetAsync(dst.d(), 0, sizeof(double) * (size_t)dst.ld * cols, s));
    MCML_HIP(hipMemcpy2DAsync(dst.d(), sizeof(double) * dst.ld, host, sizeof(double) * ldh,
                              sizeof(double) * rows, cols, hipMemcpyHostToDevice, s));
    return MCML_OK;
}

int download_matrix(double* host, int ldh, const double* dev, int ldd, int rows, int cols,
                    hipStream_t s)
{
    if (rows == 0 || cols == 0) return MCML_OK;
    MCML_HIP(hipMemcpy2DAsync(host, sizeof(double) * ldh, dev, sizeof(double) * ldd,
                              sizeof(double) * rows, cols, hipMemcpyDeviceToHost, s));
    MCML_HIP(hipStreamSynchronize(s));
    return MCML_OK;
}

}  // namespace mcml
