// comm.hip -- the cross-rank exchange of the MCML path: ONE collective, an in-place sum of a few doubles
// (per-chain sufficient statistics of the MCNR step, (sum, count) of an objective evaluation), SURVEY 8(e).
//
// Native path: an RCCL communicator owned by the context; ncclAllReduce(sum, f64) is enqueued on the
// context's stream behind the kernels that produced the statistics -- no host synchronisation, no
// callback.  This is what a host without torch (R, the reference's host language) uses.  librccl is
// resolved with dlopen when a communicator is first asked for, so the library loads on machines
// without RCCL and single-process use never touches it.
// Hook path: glmmr_mcml_dev_opts.reduce (bench.py / glmmrmcml_amd.dist install torch.distributed's
// all_reduce): needs a stream synchronisation either side of the callback.
#include "../../include/glmmr_mcml_c.h"
#include "ctx.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace mcml {

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int rccl_api(RcclApi** out)
{
    static RcclApi api;
    static int state = 0;     // 0 untried, 1 ready, -1 failed
    if (state == 0) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) { api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (api.handle) break; }
        if (api.handle) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
            api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
        }
        state = (api.handle && api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce &&
                 api.GetErrorString) ? 1 : -1;
    }
    if (state != 1) { set_error("librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing"); return MCML_EUNSUPPORTED; }
    *out = &api;
    return MCML_OK;
}
}  // namespace

void comm_release(Ctx& c)
{
    if (c.comm) {
        RcclApi* a = nullptr;
        if (rccl_api(&a) == MCML_OK) (void)a->CommDestroy((ncclComm_t)c.comm);
        c.comm = nullptr;
    }
}

// sum n doubles at dev (device memory) over the ranks, in place, ordered behind everything already on c.stream
int allreduce_dev(Ctx& c, double* dev, int n)
{
    if (!c.comm && c.world <= 1) return MCML_OK;
    c.coll_calls += 1; c.coll_doubles += n;
    if (c.comm) {
        RcclApi* a = nullptr;
        MCML_TRY(rccl_api(&a));
        ncclResult_t r = a->AllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)c.comm, c.stream);
        if (r != ncclSuccess) { set_error("ncclAllReduce failed: %s", a->GetErrorString(r)); return MCML_EHIP; }
        return MCML_OK;
    }
    if (c.reduce) {
        MCML_HIP(hipStreamSynchronize(c.stream));
        int rc = c.reduce(c.reduce_user, dev, n);
        if (rc) { set_error("reduce hook failed (%d)", rc); return MCML_EINVAL; }
        return MCML_OK;
    }
    set_error("context is rank %d of %d but has neither an RCCL communicator (glmmr_mcml_ctx_comm_init_rccl) nor a "
              "reduce hook", c.rank, c.world);
    return MCML_EINVAL;
}

// sums `n` host doubles over all ranks (identity when single-process)
int allreduce_host(Ctx& c, double* vals, int n)
{
    if (!c.comm && c.world <= 1) return MCML_OK;
    MCML_TRY(c.reduce_buf.ensure(sizeof(double) * (size_t)(n < 64 ? 64 : n)));
    MCML_HIP(hipMemcpyAsync(c.reduce_buf.p, vals, sizeof(double) * n, hipMemcpyHostToDevice, c.stream));
    MCML_TRY(allreduce_dev(c, c.reduce_buf.d(), n));
    MCML_HIP(hipMemcpyAsync(vals, c.reduce_buf.p, sizeof(double) * n, hipMemcpyDeviceToHost, c.stream));
    MCML_HIP(hipStreamSynchronize(c.stream));
    return MCML_OK;
}

}  // namespace mcml

using namespace mcml;

extern "C" int glmmr_mcml_rccl_unique_id(unsigned char* id128)
{
    MCML_REQUIRE(id128, "rccl_unique_id: null argument");
    RcclApi* a = nullptr;
    MCML_TRY(rccl_api(&a));
    ncclUniqueId id;
    ncclResult_t r = a->GetUniqueId(&id);
    if (r != ncclSuccess) { set_error("ncclGetUniqueId failed: %s", a->GetErrorString(r)); return MCML_EHIP; }
    static_assert(sizeof(id.internal) == GLMMR_MCML_RCCL_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, id.internal, sizeof(id.internal));
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_comm_init_rccl(glmmr_mcml_ctx* h, const unsigned char* id128, int rank, int world)
{
    MCML_REQUIRE(h && id128, "comm_init_rccl: null argument");
    MCML_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init_rccl: rank %d of %d", rank, world);
    Ctx& c = h->c;
    MCML_HIP(hipSetDevice(c.device));
    RcclApi* a = nullptr;
    MCML_TRY(rccl_api(&a));
    comm_release(c);
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    ncclComm_t comm = nullptr;
    ncclResult_t r = a->CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, a->GetErrorString(r)); return MCML_EHIP; }
    c.comm = comm; c.rank = rank; c.world = world;
    return MCML_OK;
}

extern "C" int glmmr_mcml_ctx_comm_allreduce(glmmr_mcml_ctx* h, double* vals, int n)
{
    MCML_REQUIRE(h && vals && n > 0 && n <= 4096, "comm_allreduce: bad argument");
    MCML_HIP(hipSetDevice(h->c.device));
    return allreduce_host(h->c, vals, n);
}

extern "C" int glmmr_mcml_ctx_comm_stats(glmmr_mcml_ctx* h, long long* calls, long long* doubles, int* native)
{
    MCML_REQUIRE(h, "comm_stats: null context");
    if (calls) *calls = h->c.coll_calls;
    if (doubles) *doubles = h->c.coll_doubles;
    if (native) *native = h->c.comm ? 1 : 0;
    return MCML_OK;
}
