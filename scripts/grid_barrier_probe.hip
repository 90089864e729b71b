// Probe: cost of a device-side grid barrier across the 8 XCDs of an MI355X with write-through (sc1) stores and
// a per-wave buffer_inv afterwards -- what a persistent per-trajectory kernel would pay between the phases of a
// product.  256 workgroups x 512 threads, 156 KB of LDS each (one per CU, as dgemm_band_kernel).  Every spin is bounded.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/grid_barrier_probe scripts/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err)
{
    __builtin_amdgcn_s_waitcnt(0);            // this thread's stores have left (vmcnt(0)); write-through stores are then visible
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000) { atomicExch(err, 1); ok = false; break; }     // bounded: never hangs the GPU
        }
    }
    __builtin_amdgcn_s_barrier();
    return ok;
}

// mode 0: barrier only; 1: + buffer_inv sc1 per wave; 2: + every thread writes (sc1) and reads another workgroup's values
__global__ __launch_bounds__(512) void k_probe(unsigned* counter, int* err, double* buf, int iters, int mode, double* out)
{
    extern __shared__ double lds[];
    const int nwg = gridDim.x;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        if (mode >= 2) __hip_atomic_store(&buf[(size_t)blockIdx.x * 512 + threadIdx.x], (double)(it + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(counter, (unsigned)(it + 1) * nwg, err)) break;
        if (*(volatile int*)err) break;
        if (mode >= 1) asm volatile("buffer_inv sc1" ::: "memory");
        if (mode >= 2) {
            const int other = (blockIdx.x + 37) % nwg;
            const double v = buf[(size_t)other * 512 + threadIdx.x];
            if (v != (double)(it + other)) atomicExch(err, 2);          // stale data: the protocol is wrong
            acc += v;
        }
        if (mode >= 2) {       // second barrier: nobody overwrites buf before everybody has read it
            if (!grid_barrier(counter + 32, (unsigned)(it + 1) * nwg, err)) break;
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc + lds[0] * 0.0;
}

int main()
{
    unsigned* counter; int* err; double *buf, *out;
    CHECK(hipMalloc(&counter, 256)); CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&buf, sizeof(double) * 256 * 512)); CHECK(hipMalloc(&out, sizeof(double) * 256));
    CHECK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    for (int nwg : {256, 64}) for (int mode : {0, 1, 2}) {
        const int iters = 2000;
        CHECK(hipMemset(counter, 0, 256)); CHECK(hipMemset(err, 0, 4));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(512), 156 * 1024, 0, counter, err, buf, iters, mode, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int herr = 0; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("workgroups %3d mode %d: %.2f us per iteration (%s)%s\n", nwg, mode, ms * 1e3 / iters,
               mode == 0 ? "one barrier" : mode == 1 ? "one barrier + buffer_inv sc1" : "write-through store, barrier, inv, read, barrier",
               herr == 0 ? "" : herr == 1 ? "  [SPIN LIMIT HIT]" : "  [STALE DATA SEEN]");
        fflush(stdout);
    }
    return 0;
}
