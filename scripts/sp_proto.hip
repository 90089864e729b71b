// Prototype: config-5-shaped backward sparse product + leapfrog epilogue with the state stored CHAIN-MAJOR
// (chain contiguous): a wave = 64 chains of one random effect, row metadata wave-uniform (scalar loads).
// Measures the achievable rate before committing to a refactor of the sparse sampler state.
// build: hipcc --offload-arch=gfx950 -O3 scripts/sp_proto.hip -o scripts/sp_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// chain-major: S[c + i*C], X/R/UP[c + q*C]
__global__ __launch_bounds__(256) void k_bwd_cm(int Q, int C, const int* ptr, const int* ci, const double* val,
                                                const double* S, const double* Xs, double* R, double* UP,
                                                const double* e, const int* steps, int s, double post, int rpw)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const int q0 = (blockIdx.x * 4 + w) * rpw;
    if (c >= C) return;
    const int st = steps[c]; const double en = e[c];
    for (int q = q0; q < q0 + rpw && q < Q; ++q) {
        const int t0 = __builtin_amdgcn_readfirstlane(ptr[q]), t1 = __builtin_amdgcn_readfirstlane(ptr[q + 1]);
        double acc = 0.0;
        for (int t = t0; t < t1; ++t) {
            const int i = __builtin_amdgcn_readfirstlane(ci[t]);
            const double v = val[t];
            acc += v * S[c + (size_t)i * C];
        }
        if (s >= st) continue;
        const size_t off = c + (size_t)q * C;
        const double x = Xs[off];
        double g = -1.0 * x; g = g + post * acc;
        double rr = R[off]; rr = rr + (en / 2) * g;
        if (s + 1 < st) { rr = rr + (en / 2) * g; UP[off] = x + en * rr; }
        R[off] = rr;
    }
}

int main()
{
    const int nsubj = 2000, nvis = 10, n = nsubj * nvis, Q = nsubj + n, C = 1024;
    std::vector<int> ptr(Q + 1), ci; std::vector<double> val;
    ptr[0] = 0;
    for (int q = 0; q < Q; ++q) {
        if (q < nsubj) for (int v = 0; v < nvis; ++v) { ci.push_back(q * nvis + v); val.push_back(0.5); }
        else { ci.push_back(q - nsubj); val.push_back(0.2); }
        ptr[q + 1] = (int)ci.size();
    }
    int *dptr, *dci, *dsteps; double *dval, *dS, *dX, *dR, *dUP, *de;
    CK(hipMalloc(&dptr, 4 * (Q + 1))); CK(hipMalloc(&dci, 4 * ci.size())); CK(hipMalloc(&dval, 8 * val.size()));
    CK(hipMalloc(&dS, 8ull * n * C)); CK(hipMalloc(&dX, 8ull * Q * C)); CK(hipMalloc(&dR, 8ull * Q * C)); CK(hipMalloc(&dUP, 8ull * Q * C));
    CK(hipMalloc(&de, 8 * C)); CK(hipMalloc(&dsteps, 4 * C));
    CK(hipMemcpy(dptr, ptr.data(), 4 * (Q + 1), hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), 4 * ci.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dval, val.data(), 8 * val.size(), hipMemcpyHostToDevice));
    CK(hipMemset(dS, 0, 8ull * n * C)); CK(hipMemset(dX, 0, 8ull * Q * C)); CK(hipMemset(dR, 0, 8ull * Q * C)); CK(hipMemset(dUP, 0, 8ull * Q * C));
    std::vector<double> he(C, 0.05); std::vector<int> hs(C, 10);
    CK(hipMemcpy(de, he.data(), 8 * C, hipMemcpyHostToDevice)); CK(hipMemcpy(dsteps, hs.data(), 4 * C, hipMemcpyHostToDevice));
    for (int rpw : {1, 2, 4, 8, 16}) {
        dim3 grid((Q + 4 * rpw - 1) / (4 * rpw), (C + 63) / 64);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_bwd_cm, grid, dim3(256), 0, 0, Q, C, dptr, dci, dval, dS, dX, dR, dUP, de, dsteps, 3, 1.0, rpw);
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k_bwd_cm, grid, dim3(256), 0, 0, Q, C, dptr, dci, dval, dS, dX, dR, dUP, de, dsteps, 3, 1.0, rpw);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms / reps * 1e3, bytes = 8.0 * ((double)n * C + 4.0 * Q * C);
        printf("chain-major backward, rows per wave %2d: %.1f us per launch, %.2f TB/s algorithmic (8(nC+4QC) = %.0f MB)\n", rpw, us, bytes / us / 1e6, bytes / 1e6);
    }
    return 0;
}
