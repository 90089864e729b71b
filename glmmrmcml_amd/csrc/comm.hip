// comm.hip -- the cross-rank exchanges of the MCML path:
//   * an in-place sum of a few doubles (per-chain sufficient statistics of the MCNR step, (sum, count) of an objective
//     evaluation, the values of one round of candidate thetas);
//   * ONE all-gather of the sample columns per MCML iteration (Q x m doubles, 41 MB at config 3): every rank then holds
//     all of u, so the theta-step shards over candidate thetas instead of replicating its factorisation on every rank
//     (drivers.hip::d_optim), and mcml_full can hand back all of u as the reference does (mcml_full.cpp:144-145).
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
#include <string>

namespace mcml {

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

// resolved once per process, whichever host thread asks first (two contexts may be driven from two threads)
int rccl_api(RcclApi** out)
{
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            a.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
            const char* e = dlerror();          // read once: the call clears it
            a.why = e ? e : "dlopen failed";
        }
        if (!a.handle) return a;
        auto sym = [&](const char* nm) -> void* {
            void* p = dlsym(a.handle, nm);
            if (!p) { const char* e = dlerror(); a.why = std::string("symbol ") + nm + " missing" + (e ? std::string(": ") + e : std::string()); }
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.GetErrorString;
        return a;
    }();
    if (!api.ok) { set_error("librccl.so could not be loaded (%s)", api.why.c_str()); return MCML_EUNSUPPORTED; }
    *out = &api;
    return MCML_OK;
}
}  // namespace

// rank emulation (ctx.h emu_world): the sum over emu_world identical ranks
__global__ void k_emu_scale(double* v, int n, double f)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= f;
}

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
    if (!c.comm && c.world <= 1 && c.emu_world > 1) {
        c.coll_calls += 1; c.coll_doubles += n;
        hipLaunchKernelGGL(k_emu_scale, dim3((n + 255) / 256), dim3(256), 0, c.stream, dev, n, (double)c.emu_world);
        MCML_HIP(hipGetLastError());
        return MCML_OK;
    }
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

// every rank contributes `count` doubles at `send`; afterwards recv[r * count .. (r+1) * count) holds rank r's, on every
// rank.  send may alias recv + rank * count.  Ordered behind everything already on c.stream.
int allgather_dev(Ctx& c, const double* send, double* recv, size_t count)
{
    double* mine = recv + (size_t)c.rank * count;
    if (!c.comm && c.world <= 1 && c.emu_world > 1) {
        c.gather_calls += 1; c.gather_doubles += (long long)count * c.emu_world;
        for (int r = 0; r < c.emu_world; ++r)
            if (recv + (size_t)r * count != send)
                MCML_HIP(hipMemcpyAsync(recv + (size_t)r * count, send, sizeof(double) * count, hipMemcpyDeviceToDevice, c.stream));
        return MCML_OK;
    }
    if (!c.comm && c.world <= 1) {
        if (send != mine) MCML_HIP(hipMemcpyAsync(mine, send, sizeof(double) * count, hipMemcpyDeviceToDevice, c.stream));
        return MCML_OK;
    }
    c.gather_calls += 1; c.gather_doubles += (long long)count * c.world;
    if (c.comm) {
        RcclApi* a = nullptr;
        MCML_TRY(rccl_api(&a));
        ncclResult_t r = a->AllGather(send, recv, count, ncclDouble, (ncclComm_t)c.comm, c.stream);
        if (r != ncclSuccess) { set_error("ncclAllGather failed: %s", a->GetErrorString(r)); return MCML_EHIP; }
        return MCML_OK;
    }
    if (c.reduce) {
        // the hook only sums: every other rank's slot is zero on this rank, and x + 0 + ... + 0 is x exactly
        MCML_REQUIRE(count * (size_t)c.world < ((size_t)1 << 31), "all-gather through the reduce hook: %zu doubles do not fit its int count", count * (size_t)c.world);
        if (send != mine) MCML_HIP(hipMemcpyAsync(mine, send, sizeof(double) * count, hipMemcpyDeviceToDevice, c.stream));
        if (c.rank > 0) MCML_HIP(hipMemsetAsync(recv, 0, sizeof(double) * count * (size_t)c.rank, c.stream));
        if (c.rank + 1 < c.world)
            MCML_HIP(hipMemsetAsync(mine + count, 0, sizeof(double) * count * (size_t)(c.world - 1 - c.rank), c.stream));
        MCML_HIP(hipStreamSynchronize(c.stream));
        int rc = c.reduce(c.reduce_user, recv, (int)(count * (size_t)c.world));
        if (rc) { set_error("reduce hook failed (%d)", rc); return MCML_EINVAL; }
        return MCML_OK;
    }
    set_error("context is rank %d of %d but has neither an RCCL communicator (glmmr_mcml_ctx_comm_init_rccl) nor a "
              "reduce hook", c.rank, c.world);
    return MCML_EINVAL;
}

// c.Uall <- every rank's sample columns (rank r's block at column r * mcols); a no-op while the samples are unchanged
int gather_samples(Ctx& c)
{
    MCML_REQUIRE(c.mcols > 0 && c.U.d(), "gather_samples: no samples set");
    if (c.uall_valid) return MCML_OK;
    const int wr = comm_world(c);
    // the blocks must be equally wide (one count per rank in the collective): agree on it first
    double chk[2] = {(double)c.mcols, (double)c.mcols * c.mcols};
    MCML_TRY(allreduce_host(c, chk, 2));
    MCML_REQUIRE(chk[0] * chk[0] == wr * chk[1], "gather_samples: the ranks hold different numbers of sample columns (this rank: %d)", c.mcols);
    if (c.Uall.rows != c.Q || c.Uall.cols != c.mcols * wr) MCML_TRY(c.Uall.alloc(c.Q, c.mcols * wr));
    MCML_REQUIRE(c.Uall.ld == c.U.ld, "gather_samples: leading dimensions differ (%d vs %d)", c.Uall.ld, c.U.ld);
    MCML_TRY(allgather_dev(c, c.U.d(), c.Uall.d(), (size_t)c.U.ld * c.mcols));
    c.uall_valid = true;
    return MCML_OK;
}

// sums `n` host doubles over all ranks (identity when single-process)
int allreduce_host(Ctx& c, double* vals, int n)
{
    if (!c.comm && c.world <= 1 && c.emu_world <= 1) return MCML_OK;
    MCML_TRY(c.reduce_buf.ensure(sizeof(double) * (size_t)(n < 64 ? 64 : n)));
    MCML_TRY(copy_h2d(c.reduce_buf.p, vals, sizeof(double) * n, c.stream));
    MCML_TRY(allreduce_dev(c, c.reduce_buf.d(), n));
    MCML_TRY(copy_d2h(vals, c.reduce_buf.p, sizeof(double) * n, c.stream));
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
