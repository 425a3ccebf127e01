// Feasibility probe for a barrier-free layout of the ODEFunc MLP (config 5: 64 -> 128 -> 128 -> 64, fp32):
// every WAVE owns a 16-row tile and runs all three layers on it; the (transposed) weights are the MFMA A operand and come
// from LDS (136 KB, one copy per CU), the activations are the B operand and never leave registers: the accumulator layout
// of v_mfma_f32_16x16x4_f32 (lane (n, g), register r  <->  H[n][16 mb + 4 g + r]) IS a legal B layout of the next layer
// (k-slot g of step (blk, r) = column 16 blk + 4 g + r), so there is no LDS round trip and no s_barrier per evaluation.
// The product kernel (csrc/mi_ode_mlp.h) keeps the weights in registers and shares the activations through LDS with five
// barriers per evaluation; it sits at 0.55 of the fp32 matrix peak.  This probe measures the ceiling of the other layout.
// hipcc --offload-arch=gfx950 -O3 -o mlp_wavetile mlp_wavetile.hip && ./mlp_wavetile
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int D = 64, H = 128;
constexpr int LW1 = D + 4, LW2 = H + 4, LW3 = H + 4;                   // row strides = 4 mod 64 banks: ds_read_b128 conflict-free
constexpr int OFF_W1 = 0, OFF_W2 = OFF_W1 + H * LW1, OFF_W3 = OFF_W2 + H * LW2, OFF_B = OFF_W3 + D * LW3;
constexpr int LDS_FLOATS = OFF_B + H + H + D;

__device__ __forceinline__ float act_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float x2 = x * x;
  const float small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
  return fabsf(x) < 0.25f ? small : big;
}
template <int ACT> __device__ __forceinline__ float act(float x) {
  if constexpr (ACT == 0) return act_tanh(x);
  else if constexpr (ACT == 1) return x > 0.f ? x : 0.f;
  else return x;
}

// one layer: out[mb] (MB blocks of 16 outputs) = bias + sum over KB blocks of 16 inputs; weights transposed [out][in] in LDS
template <int MB, int KB, int LW, int ACT>
__device__ __forceinline__ void layer(const float* wt, const float* bias, const f4* in, f4* out, int m, int g) {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    f4 c = *(const f4*)(bias + 16 * mb + 4 * g);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f4 a = *(const f4*)(wt + (16 * mb + m) * LW + 16 * kb + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], in[kb][r], c, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = act<ACT>(c[r]);
    out[mb] = c;
    __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from hoisting every block's LDS reads to the top (442 spills without)
  }
}

template <int ACT, int NW>
__global__ __launch_bounds__(64 * NW) void k(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                              const float* b3, const float* X, float* Y, int ntiles, int iters, float fb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < H * D; i += 64 * NW) { const int o = i / D, kk = i % D; lds[OFF_W1 + o * LW1 + kk] = W1[kk * H + o]; }
  for (int i = threadIdx.x; i < H * H; i += 64 * NW) { const int o = i / H, kk = i % H; lds[OFF_W2 + o * LW2 + kk] = W2[kk * H + o]; }
  for (int i = threadIdx.x; i < D * H; i += 64 * NW) { const int o = i / H, kk = i % H; lds[OFF_W3 + o * LW3 + kk] = W3[kk * D + o]; }
  for (int i = threadIdx.x; i < H; i += 64 * NW) { lds[OFF_B + i] = b1[i]; lds[OFF_B + H + i] = b2[i]; }
  for (int i = threadIdx.x; i < D; i += 64 * NW) lds[OFF_B + 2 * H + i] = b3[i];
  __syncthreads();                                                       // the only barrier of the kernel
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  for (int t = blockIdx.x * NW + wave; t < ntiles; t += gridDim.x * NW) {
    f4 x[4], h1[8], h2[8], o[4];
    const float* xr = X + (size_t)(16 * t + m) * D + 4 * g;              // lane (n = m, g): row n, columns 16 kb + 4 g .. + 3
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) x[kb] = *(const f4*)(xr + 16 * kb);
    for (int it = 0; it < iters; ++it) {
      layer<8, 4, LW1, ACT>(lds + OFF_W1, lds + OFF_B, x, h1, m, g);
      layer<8, 8, LW2, ACT>(lds + OFF_W2, lds + OFF_B + H, h1, h2, m, g);
      layer<4, 8, LW3, 2>(lds + OFF_W3, lds + OFF_B + 2 * H, h2, o, m, g);
      if (it + 1 < iters) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) x[kb] = x[kb] + fb * o[kb];       // an Euler-like update keeps the evaluations dependent
      }
    }
    float* yr = Y + (size_t)(16 * t + m) * D + 4 * g;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) *(f4*)(yr + 16 * kb) = o[kb];
  }
}

static void cpu_eval(const std::vector<float>& W1, const std::vector<float>& b1, const std::vector<float>& W2, const std::vector<float>& b2,
                     const std::vector<float>& W3, const std::vector<float>& b3, const float* x, float* out, int actk) {
  auto a = [&](double v) { return actk == 0 ? std::tanh(v) : (v > 0 ? v : 0.0); };
  double h1[H], h2[H];
  for (int o = 0; o < H; ++o) { double s = b1[o]; for (int kk = 0; kk < D; ++kk) s += (double)x[kk] * W1[kk * H + o]; h1[o] = a(s); }
  for (int o = 0; o < H; ++o) { double s = b2[o]; for (int kk = 0; kk < H; ++kk) s += h1[kk] * W2[kk * H + o]; h2[o] = a(s); }
  for (int o = 0; o < D; ++o) { double s = b3[o]; for (int kk = 0; kk < H; ++kk) s += h2[kk] * W3[kk * D + o]; out[o] = (float)s; }
}

template <int ACT, int NW>
static void run(const char* what, int rows, int iters, const float* dW1, const float* db1, const float* dW2, const float* db2,
                const float* dW3, const float* db3, const float* dX, float* dY) {
  const int ntiles = rows / 16;
  const size_t lds_bytes = LDS_FLOATS * sizeof(float);
  hipFuncSetAttribute((const void*)k<ACT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  const int grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ACT, NW>), dim3(grid), dim3(64 * NW), lds_bytes, 0, dW1, db1, dW2, db2, dW3, db3, dX, dY, ntiles, iters, 1e-3f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ACT, NW>), dim3(grid), dim3(64 * NW), lds_bytes, 0, dW1, db1, dW2, db2, dW3, db3, dX, dY, ntiles, iters, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double flop = (double)rows * iters * 2.0 * (D * H + H * H + H * D);
  printf("%-34s rows %6d  evaluations %3d  %d waves/CU: %.3f ms  %.1f TFLOP/s  (%.2f of 157.3)   %s\n", what, rows, iters, NW, best,
         flop / best / 1e9, flop / best / 1e9 / 157.3, hipGetErrorString(hipGetLastError()));
}

int main() {
  const int rows = 32768;
  std::vector<float> W1(D * H), b1(H), W2(H * H), b2(H), W3(H * D), b3(D), X((size_t)2 * rows * D), Y((size_t)2 * rows * D);
  srand(1);
  auto rnd = [](float s) { return s * (2.f * rand() / RAND_MAX - 1.f); };
  for (auto& v : W1) v = rnd(0.18f);
  for (auto& v : W2) v = rnd(0.15f);
  for (auto& v : W3) v = rnd(0.18f);
  for (auto& v : b1) v = rnd(0.1f);
  for (auto& v : b2) v = rnd(0.1f);
  for (auto& v : b3) v = rnd(0.1f);
  for (auto& v : X) v = rnd(1.5f);
  float *dW1, *db1, *dW2, *db2, *dW3, *db3, *dX, *dY;
  auto up = [](float** d, const std::vector<float>& h) { hipMalloc(d, h.size() * 4); hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice); };
  up(&dW1, W1); up(&db1, b1); up(&dW2, W2); up(&db2, b2); up(&dW3, W3); up(&db3, b3); up(&dX, X);
  hipMalloc(&dY, Y.size() * 4);
  // correctness of the layout algebra: one evaluation against a double-precision CPU evaluation of a few rows
  for (int actk = 0; actk < 2; ++actk) {
    const size_t lds_bytes = LDS_FLOATS * sizeof(float);
    if (actk == 0) {
      hipFuncSetAttribute((const void*)k<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      hipLaunchKernelGGL((k<0, 8>), dim3(256), dim3(512), lds_bytes, 0, dW1, db1, dW2, db2, dW3, db3, dX, dY, rows / 16, 1, 0.f);
    } else {
      hipFuncSetAttribute((const void*)k<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      hipLaunchKernelGGL((k<1, 8>), dim3(256), dim3(512), lds_bytes, 0, dW1, db1, dW2, db2, dW3, db3, dX, dY, rows / 16, 1, 0.f);
    }
    hipDeviceSynchronize();
    hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int row : {0, 1, 15, 16, 17, 255, 4097, rows - 1}) {
      float ref[D];
      cpu_eval(W1, b1, W2, b2, W3, b3, &X[(size_t)row * D], ref, actk);
      for (int c = 0; c < D; ++c) worst = fmax(worst, fabs((double)Y[(size_t)row * D + c] - ref[c]));
    }
    printf("%s: max |gpu - cpu(double)| over 8 rows = %.2e   %s\n", actk == 0 ? "tanh" : "relu", worst, hipGetErrorString(hipGetLastError()));
  }
  run<0, 8>("tanh", rows, 14, dW1, db1, dW2, db2, dW3, db3, dX, dY);
  run<0, 8>("tanh", rows, 140, dW1, db1, dW2, db2, dW3, db3, dX, dY);
  run<1, 8>("relu", rows, 140, dW1, db1, dW2, db2, dW3, db3, dX, dY);
  run<2, 8>("no activation (MFMA + LDS only)", rows, 140, dW1, db1, dW2, db2, dW3, db3, dX, dY);
  run<0, 16>("tanh", 2 * rows, 140, dW1, db1, dW2, db2, dW3, db3, dX, dY);   // 4 waves per SIMD (128 VGPRs each)
  return 0;
}
