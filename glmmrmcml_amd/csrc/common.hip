// common.hip -- error string, matrix upload/download.
#include "common.h"

#include <cstring>
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

// ---- staged copies ----------------------------------------------------------------------------------------------
namespace {
constexpr size_t STAGE_MIN = 16 << 10;        // smaller copies: the runtime's own staging
constexpr size_t STAGE_BYTES = 16 << 20;
struct HostStage {
    std::mutex mu;
    void* p = nullptr;
    int get(void** out) {
        if (!p) {
            hipError_t e = hipHostMalloc(&p, STAGE_BYTES, hipHostMallocPortable);
            if (e != hipSuccess) { p = nullptr; set_error("hipHostMalloc(%zu) failed: %s", STAGE_BYTES, hipGetErrorString(e)); return MCML_ENOMEM; }
        }
        *out = p;
        return MCML_OK;
    }
    // never freed: the buffer lives as long as the process (a static destructor would run after the runtime's own)
};
HostStage& host_stage() { static HostStage* hs = new HostStage; return *hs; }
}  // namespace

int copy_h2d_2d(void* dev, size_t dpitch, const void* host, size_t hpitch, size_t rowbytes, size_t cols, hipStream_t s)
{
    if (rowbytes == 0 || cols == 0) return MCML_OK;
    MCML_REQUIRE(dev && host && dpitch >= rowbytes && hpitch >= rowbytes, "copy_h2d: bad arguments");
    if (rowbytes * cols < STAGE_MIN) {
        MCML_HIP(hipMemcpy2DAsync(dev, dpitch, host, hpitch, rowbytes, cols, hipMemcpyHostToDevice, s));
        return MCML_OK;
    }
    HostStage& hs = host_stage();
    std::lock_guard<std::mutex> lk(hs.mu);
    char* st; MCML_TRY(hs.get((void**)&st));
    if (rowbytes > STAGE_BYTES) {                  // a column longer than the buffer: piece by piece
        for (size_t j = 0; j < cols; ++j)
            for (size_t o = 0; o < rowbytes; o += STAGE_BYTES) {
                const size_t nb = rowbytes - o < STAGE_BYTES ? rowbytes - o : STAGE_BYTES;
                memcpy(st, (const char*)host + j * hpitch + o, nb);
                MCML_HIP(hipMemcpyAsync((char*)dev + j * dpitch + o, st, nb, hipMemcpyHostToDevice, s));
                MCML_HIP(hipStreamSynchronize(s));
            }
        return MCML_OK;
    }
    const size_t per = STAGE_BYTES / rowbytes;     // columns per chunk, packed
    for (size_t j0 = 0; j0 < cols; j0 += per) {
        const size_t nc = cols - j0 < per ? cols - j0 : per;
        if (hpitch == rowbytes) memcpy(st, (const char*)host + j0 * hpitch, nc * rowbytes);
        else for (size_t j = 0; j < nc; ++j) memcpy(st + j * rowbytes, (const char*)host + (j0 + j) * hpitch, rowbytes);
        if (nc == 1 || dpitch == rowbytes) MCML_HIP(hipMemcpyAsync((char*)dev + j0 * dpitch, st, nc * rowbytes, hipMemcpyHostToDevice, s));
        else MCML_HIP(hipMemcpy2DAsync((char*)dev + j0 * dpitch, dpitch, st, rowbytes, rowbytes, nc, hipMemcpyHostToDevice, s));
        MCML_HIP(hipStreamSynchronize(s));         // the buffer is reused by the next chunk / the next caller
    }
    return MCML_OK;
}

int copy_d2h_2d(void* host, size_t hpitch, const void* dev, size_t dpitch, size_t rowbytes, size_t cols, hipStream_t s)
{
    if (rowbytes == 0 || cols == 0) return MCML_OK;
    MCML_REQUIRE(dev && host && dpitch >= rowbytes && hpitch >= rowbytes, "copy_d2h: bad arguments");
    if (rowbytes * cols < STAGE_MIN) {
        MCML_HIP(hipMemcpy2DAsync(host, hpitch, dev, dpitch, rowbytes, cols, hipMemcpyDeviceToHost, s));
        return MCML_OK;
    }
    HostStage& hs = host_stage();
    std::lock_guard<std::mutex> lk(hs.mu);
    char* st; MCML_TRY(hs.get((void**)&st));
    if (rowbytes > STAGE_BYTES) {
        for (size_t j = 0; j < cols; ++j)
            for (size_t o = 0; o < rowbytes; o += STAGE_BYTES) {
                const size_t nb = rowbytes - o < STAGE_BYTES ? rowbytes - o : STAGE_BYTES;
                MCML_HIP(hipMemcpyAsync(st, (const char*)dev + j * dpitch + o, nb, hipMemcpyDeviceToHost, s));
                MCML_HIP(hipStreamSynchronize(s));
                memcpy((char*)host + j * hpitch + o, st, nb);
            }
        return MCML_OK;
    }
    const size_t per = STAGE_BYTES / rowbytes;
    for (size_t j0 = 0; j0 < cols; j0 += per) {
        const size_t nc = cols - j0 < per ? cols - j0 : per;
        if (nc == 1 || dpitch == rowbytes) MCML_HIP(hipMemcpyAsync(st, (const char*)dev + j0 * dpitch, nc * rowbytes, hipMemcpyDeviceToHost, s));
        else MCML_HIP(hipMemcpy2DAsync(st, rowbytes, (const char*)dev + j0 * dpitch, dpitch, rowbytes, nc, hipMemcpyDeviceToHost, s));
        MCML_HIP(hipStreamSynchronize(s));
        if (hpitch == rowbytes) memcpy((char*)host + j0 * hpitch, st, nc * rowbytes);
        else for (size_t j = 0; j < nc; ++j) memcpy((char*)host + (j0 + j) * hpitch, st + j * rowbytes, rowbytes);
    }
    return MCML_OK;
}

int copy_h2d(void* dev, const void* host, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return MCML_OK;
    if (bytes < STAGE_MIN) { MCML_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s)); return MCML_OK; }   // 1-D: the plain call
    return copy_h2d_2d(dev, bytes, host, bytes, bytes, 1, s);
}
int copy_d2h(void* host, const void* dev, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return MCML_OK;
    if (bytes < STAGE_MIN) { MCML_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s)); return MCML_OK; }
    return copy_d2h_2d(host, bytes, dev, bytes, bytes, 1, s);
}

int upload_matrix(DevMat& dst, const double* host, int rows, int cols, int ldh, hipStream_t s)
{
    MCML_REQUIRE(host && rows >= 0 && cols >= 0 && ldh >= rows, "upload_matrix: bad shape");
    MCML_TRY(dst.alloc(rows, cols));
    if (rows == 0 || cols == 0) return MCML_OK;
    // zero the padding rows too: kernels may read (never use) them
    MCML_HIP(hipMemsetAsync(dst.d(), 0, sizeof(double) * (size_t)dst.ld * cols, s));
    return copy_h2d_2d(dst.d(), sizeof(double) * dst.ld, host, sizeof(double) * ldh, sizeof(double) * rows, cols, s);
}

int download_matrix(double* host, int ldh, const double* dev, int ldd, int rows, int cols,
                    hipStream_t s)
{
    if (rows == 0 || cols == 0) return MCML_OK;
    MCML_TRY(copy_d2h_2d(host, sizeof(double) * ldh, dev, sizeof(double) * ldd, sizeof(double) * rows, cols, s));
    MCML_HIP(hipStreamSynchronize(s));
    return MCML_OK;
}

}  // namespace mcml
