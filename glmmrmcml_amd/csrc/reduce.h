// reduce.h -- fixed-order wave / workgroup sums (deterministic: no atomics).
#pragma once
#include <hip/hip_runtime.h>
namespace mcml {
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;   // lane 0
}
// result valid in thread 0; sh holds blockDim.x/64 doubles
__device__ __forceinline__ double block_sum(double v, double* sh)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}
}  // namespace mcml
