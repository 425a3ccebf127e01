// Where does the linear-RHS tile kernel lose MFMA throughput?  Same per-stage instruction mix as LinCtx::rhs_eval
// (D = 128, fp64, 16-row tile, 8 waves), switched on piece by piece.
// hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
struct alignas(16) CH { double v[2]; };
constexpr int D = 128, LD = D + 2, KS = D / 4;

// MODE 4: as 3, plus the tile traffic of the attempt kernel every 6 evaluations: prefetch y0/f0 of the next tile
//         (accumulator layout, 8-byte lanes, 128-byte row segments), store y1/f1 of this one
template <int MODE>   // 0: MFMA only (operands in registers)  1: + A from LDS (ds_read_b128)  2: + LDS write + 2 barriers (= rhs_eval)
                      // 3: as 2 plus a stage-6-sized combine (12 fp64 mul/add per element) between the evaluations
__global__ __launch_bounds__(512) void k(double* out, const double* W, int iters, const double* pin = nullptr, double* pout = nullptr, long long nrows = 0) {
  __shared__ __attribute__((aligned(16))) double s_ys[16 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4, col = 16 * wave + li;
  double bf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) bf[s] = W[(lg * KS + s) * D + col];
  double ys[4] = {tid * 1e-3, 1.0, 2.0, 3.0};
  for (int i = tid; i < 16 * LD; i += 512) s_ys[i] = i * 1e-4;
  __syncthreads();
  double acc_out = 0;
  double pre[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tile = blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    if ((MODE == 4 || MODE == 5) && it % 6 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ys[i] += pre[i] + pre[4 + i]; }
      const long long nt = nrows / 16;
      const long long tn = (tile + gridDim.x) % nt;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long idx = (tn * 16 + lg + 4 * i) * D + col;
        pre[i] = pin[idx]; pre[4 + i] = pin[nrows * D + idx];
      }
    }
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s_ys[(lg + 4 * i) * LD + col] = ys[i];
      __syncthreads();
    }
    d4 c0 = {0, 0, 0, 0};
    const double* ap = s_ys + li * LD + lg * KS;
#pragma unroll
    for (int m = 0; m < KS / 2; ++m) {
      CH a0;
      if (MODE >= 1) a0 = *(const CH*)(ap + m * 2);
      else { a0.v[0] = ys[0] + m; a0.v[1] = ys[1] + m; }
#pragma unroll
      for (int v = 0; v < 2; ++v) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.v[v], bf[m * 2 + v], c0, 0, 0, 0);
    }
    if (MODE >= 2) __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double y = c0[i];
      if (MODE >= 3) {
#pragma unroll
        for (int j = 0; j < 6; ++j) y = y + (1e-3 * (j + 1)) * ys[(i + j) & 3];
      }
      ys[i] = y * 1e-3;
    }
    acc_out += ys[0];
    if ((MODE == 4 || MODE == 6) && it % 6 == 5) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long idx = (tile * 16 + lg + 4 * i) * D + col;
        pout[idx] = ys[i]; pout[nrows * D + idx] = ys[i] + 1.0;
      }
      tile = (tile + gridDim.x) % (nrows / 16);
    }
  }
  out[blockIdx.x * 512 + tid] = acc_out + ys[1] + ys[2] + ys[3];
}

template <int MODE>
void run(int blocks, int iters, const double* W, const char* what) {
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, W, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, W, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 8 * iters * KS * 2048.0;
  printf("mode %d (%s), %d workgroups x 512: %.3f ms  %.1f TFLOP/s\n", MODE, what, blocks, ms, flop / ms / 1e9);
  hipFree(out);
}
int main() {
  double* W;
  hipMalloc(&W, sizeof(double) * D * D);
  hipMemset(W, 0, sizeof(double) * D * D);
  const int it = 600;     // ~ 100 tiles x 6 stages: the duration of one attempt kernel
  for (int blocks : {256, 512}) {
    run<0>(blocks, it, W, "MFMA only");
    run<1>(blocks, it, W, "+ A operand from LDS");
    run<2>(blocks, it, W, "+ LDS write, 2 barriers");
    run<3>(blocks, it, W, "+ combine VALU");
  }
  {
    const long long nrows = 65536;
    double *pin, *pout, *out;
    hipMalloc(&pin, sizeof(double) * 2 * nrows * D); hipMalloc(&pout, sizeof(double) * 2 * nrows * D);
    hipMemset(pin, 0, sizeof(double) * 2 * nrows * D);
    hipMalloc(&out, sizeof(double) * 256 * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, W, 96, pin, pout, nrows);   // 16 tiles x 6 evaluations = one attempt
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode 4 (+ tile loads/stores, one attempt = 96 evaluations): %.3f ms  %.1f TFLOP/s\n", ms, 256.0 * 8 * 96 * KS * 2048.0 / ms / 1e9);
    }
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, out, W, 96, pin, pout, nrows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    { float ms5; hipEventElapsedTime(&ms5, e0, e1); printf("mode 5 (tile loads only): %.3f ms  %.1f TFLOP/s\n", ms5, 256.0 * 8 * 96 * KS * 2048.0 / ms5 / 1e9); }
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, out, W, 96, pin, pout, nrows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    { float ms6; hipEventElapsedTime(&ms6, e0, e1); printf("mode 6 (tile stores only): %.3f ms  %.1f TFLOP/s\n", ms6, 256.0 * 8 * 96 * KS * 2048.0 / ms6 / 1e9); }
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, W, 96);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode 3, 96 evaluations (same length, no tile traffic): %.3f ms  %.1f TFLOP/s\n", ms, 256.0 * 8 * 96 * KS * 2048.0 / ms / 1e9);
  }
  run<2>(256, 6000, W, "+ LDS write, 2 barriers, long");
  return 0;
}
