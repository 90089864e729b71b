// Microbenchmark: FP64 MFMA issue ceiling on this chip under sustained load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, int threads, int iters) {
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, threads>>>(out, iters, 0.37, 0.73);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) k<NACC><<<blocks, threads>>>(out, iters, 0.37, 0.73);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
  printf("nacc=%d blocks=%d threads=%d: %.3f ms %.2f TFLOP/s\n", NACC, blocks, threads, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(256, 256, 20000); run<8>(256, 256, 10000); run<16>(256, 256, 5000);
  run<8>(512, 256, 10000); run<8>(256, 512, 10000); run<8>(1024, 256, 4000);
  run<8>(256, 256, 200000);
  return 0;
}
