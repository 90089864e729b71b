// Microbenchmark: FP64 MFMA issue ceiling + in-kernel clock under sustained load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0; }
}
template <int NACC> void run(int blocks, int threads, int iters, int reps) {
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  int nw = blocks * threads / 64;
  unsigned long long* st; hipMalloc(&st, 16 * nw);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, threads>>>(out, st, iters, 0.37, 0.73);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; r++) k<NACC><<<blocks, threads>>>(out, st, iters, 0.37, 0.73);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h(2 * nw); hipMemcpy(h.data(), st, 16 * nw, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int w = 0; w < nw; w++) { clk.push_back((double)h[2*w] / (double)h[2*w+1] * 100.0); cyc.push_back((double)h[2*w] / ((double)iters * NACC)); }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  double flops = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
  printf("nacc=%d blocks=%d threads=%d: %.3f ms %.2f TFLOP/s | median clock %.0f MHz, cycles/MFMA/wave median %.1f\n", NACC, blocks, threads, ms, flops / ms / 1e9, clk[nw/2], cyc[nw/2]);
  hipFree(out); hipFree(st);
}
int main() {
  run<8>(256, 256, 20000, 20);   // 1 wave / SIMD
  run<8>(512, 256, 20000, 20);   // 2 waves / SIMD
  run<8>(1024, 256, 10000, 20);  // 4 waves / SIMD
  run<8>(64, 256, 20000, 20);    // a quarter of the CUs only
  run<8>(256, 64, 20000, 20);    // one wave per CU
  run<8>(256, 1024, 5000, 20);   // 4 waves / SIMD, co-resident by construction
  run<8>(256, 768, 5000, 20);    // 3 waves / SIMD
  run<4>(256, 1024, 10000, 20);
  return 0;
}
