// Measured ceiling of v_mfma_f64_16x16x4_f64 on this part: register-only MFMA chains, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(512) void k(double* out, int iters) {
  d4 c[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) c[j] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < CH; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < CH; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
void run(int blocks, int threads, int iters) {
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * iters * CH * 2048.0;
  printf("chains %d  blocks %d x %d threads  iters %d: %.3f ms  %.1f TFLOP/s\n", CH, blocks, threads, iters, ms, flop / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(256, 512, 20000);     // one dependent chain per wave, 2 waves / SIMD (the solver kernels' shape)
  run<2>(256, 512, 10000);
  run<4>(256, 512, 5000);
  run<1>(512, 512, 20000);     // 4 waves / SIMD
  run<4>(256, 256, 10000);     // 1 wave / SIMD, 4 chains
  run<4>(256, 512, 50000);     // long run (clock settles)
  return 0;
}
