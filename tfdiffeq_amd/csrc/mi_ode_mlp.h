// Fused kernels for f(t, y) = W3^T tanh(W2^T tanh(W1^T y + b1) + b2) + b3   (ODEFunc of
// /root/reference/tfdiffeq/models/dense_odenet.py:41-92, time independent; BASELINE config 5: 64-128-128-64, fp32).
//
// The whole RK attempt runs per 32-row tile: the three layers go through v_mfma_f32_16x16x4_f32 (exact fp32), every
// wave keeps its 16-column slices of W1/W2/W3 in registers, activations travel through LDS, the stage derivatives
// k_1..k_{S+1} of the tile stay in registers (accumulator layout of layer 3).  HBM traffic per attempt: y0, f0 in;
// y1, f1 out = 4 planes (+ speculative dense output, see mi_ode_step_fused.h).  Bound: fp32 matrix pipe (1024 flop per state element per stage vs 20 B).
// DP / HP are the padded widths (multiples of 16) the kernel is instantiated for; the real dim / hidden may be smaller
// (weights are zero-padded in registers, state columns are masked).
#pragma once
#include <type_traits>
#include "mi_ode_step_fused.h"

namespace mi {

enum MlpMode { MLP_F0 = 0, MLP_INITB = 1, MLP_STEP = 2 };

template <int DP, int HP>
struct MlpGeom {
  static constexpr int CB3 = DP / 16;                       // column blocks of the output layer
  static constexpr int NW12 = HP / 16;                      // waves busy in layers 1, 2 (16 hidden columns each)
  static constexpr int NW3 = 2 * CB3;                       // waves that own state elements (2 row blocks x CB3)
  static constexpr int NW = NW12 > NW3 ? NW12 : NW3;        // waves per workgroup
  static constexpr int R = 32;                              // rows per tile
  static constexpr int LDX = DP + 4, LDH = HP + 4;          // LDS row strides (16-byte pad)
  static constexpr size_t lds_bytes() { return (size_t)R * (LDX + 2 * LDH) * sizeof(float) + 80 * sizeof(double); }
};

// One evaluation of the MLP for the tile whose input rows sit in s_x.  Every thread of the workgroup must call it.
// Owner threads (wave < NW3) receive their 4 output elements (rows rb*16 + 4*(lane>>4) + i, column 16*cb + (lane&15)).
template <int DP, int HP>
__device__ __forceinline__ void mlp_eval(float* s_x, float* s_h1, float* s_h2, const float* w1f, const float* w2f,
                                         const float* w3f, float b1v, float b2v, float b3v, float* out4) {
  using G = MlpGeom<DP, HP>;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  constexpr int KS1 = DP / 4, KS2 = HP / 4;
  __syncthreads();                                          // s_x is complete
  if (wave < G::NW12) {                                     // layer 1: [32 x DP] @ [DP x 16]
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const float* a0p = s_x + li * G::LDX + lg * KS1;
    const float* a1p = s_x + (16 + li) * G::LDX + lg * KS1;
#pragma unroll
    for (int m = 0; m < KS1 / 4; ++m) {
      const f4 a0 = *(const f4*)(a0p + 4 * m), a1 = *(const f4*)(a1p + 4 * m);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[v], w1f[4 * m + v], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[v], w1f[4 * m + v], c1, 0, 0, 0);
      }
    }
    const int col = 16 * wave + li;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s_h1[(4 * lg + i) * G::LDH + col] = tanhf(c0[i] + b1v);
      s_h1[(16 + 4 * lg + i) * G::LDH + col] = tanhf(c1[i] + b1v);
    }
  }
  __syncthreads();
  if (wave < G::NW12) {                                     // layer 2: [32 x HP] @ [HP x 16]
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const float* a0p = s_h1 + li * G::LDH + lg * KS2;
    const float* a1p = s_h1 + (16 + li) * G::LDH + lg * KS2;
#pragma unroll
    for (int m = 0; m < KS2 / 4; ++m) {
      const f4 a0 = *(const f4*)(a0p + 4 * m), a1 = *(const f4*)(a1p + 4 * m);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[v], w2f[4 * m + v], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[v], w2f[4 * m + v], c1, 0, 0, 0);
      }
    }
    const int col = 16 * wave + li;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s_h2[(4 * lg + i) * G::LDH + col] = tanhf(c0[i] + b2v);
      s_h2[(16 + 4 * lg + i) * G::LDH + col] = tanhf(c1[i] + b2v);
    }
  }
  __syncthreads();
  if (wave < G::NW3) {                                      // layer 3: one 16-row block x 16 output columns per wave
    const int rb = wave / G::CB3;
    f4 c = {0, 0, 0, 0};
    const float* ap = s_h2 + (16 * rb + li) * G::LDH + lg * KS2;
#pragma unroll
    for (int m = 0; m < KS2 / 4; ++m) {
      const f4 a = *(const f4*)(ap + 4 * m);
#pragma unroll
      for (int v = 0; v < 4; ++v) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v], w3f[4 * m + v], c, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out4[i] = c[i] + b3v;
  }
}

struct MlpArgs {
  // STEP: the adaptive attempt (controller mode).  F0 / INITB: explicit y0 etc. as in StageArgs.
  StepArgs step;
  const void* x_y0;          // F0: caller's y0
  void* copy_a;              // F0: state plane
  void* copy_b;              // F0: solution[0] (nullable)
  double rtol, atol;
  int hidden;                // real hidden width
};

template <int DP, int HP, int MODE, int S, bool TS>
__global__ __launch_bounds__((64 * MlpGeom<DP, HP>::NW)) void k_mlp(MlpArgs M) {
  using G = MlpGeom<DP, HP>;
  const StepArgs& A = M.step;
  StepPlanes<float, S> P;
  if (MODE == MLP_F0) {
    P.y0 = (const float*)M.x_y0;
    P.f0 = nullptr; P.y1 = nullptr; P.hs = 0.f; P.t0 = 0.f; P.j_lo = P.j_hi = 0;
    P.f1 = (float*)(A.planes + 2 * A.stride);               // F0 writes f0 into idx_k[0] of a fresh handle
  } else {
    if (!resolve_step<float, S>(A, P)) return;
    if (MODE == MLP_INITB) P.hs = (float)A.ctl->h0;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_x = (float*)smem_raw;
  float* s_h1 = s_x + G::R * G::LDX;
  float* s_h2 = s_h1 + G::R * G::LDH;
  double* red = (double*)(s_h2 + G::R * G::LDH);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int d = A.dim, hd = M.hidden;
  const float* W1 = (const float*)A.rhs.w[0];
  const float* W2 = (const float*)A.rhs.w[1];
  const float* W3 = (const float*)A.rhs.w[2];
  const float* B1 = (const float*)A.rhs.b[0];
  const float* B2 = (const float*)A.rhs.b[1];
  const float* B3 = (const float*)A.rhs.b[2];
  const float sign = (float)A.rhs.sign;
  constexpr int KS1 = DP / 4, KS2 = HP / 4;

  // resident weight slices, zero padded: lane (col = li, group lg) holds W[k = lg*KS + s][16*block + li]
  float w1f[KS1], w2f[KS2], w3f[KS2];
  {
    const int c12 = 16 * wave + li;
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      const int k = lg * KS1 + s;
      w1f[s] = (wave < G::NW12 && k < d && c12 < hd) ? W1[(long long)k * hd + c12] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
      const int k = lg * KS2 + s;
      w2f[s] = (wave < G::NW12 && k < hd && c12 < hd) ? W2[(long long)k * hd + c12] : 0.f;
    }
    const int c3 = 16 * (wave % G::CB3) + li;
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
      const int k = lg * KS2 + s;
      w3f[s] = (wave < G::NW3 && k < hd && c3 < d) ? W3[(long long)k * d + c3] : 0.f;
    }
  }
  const int c12 = 16 * wave + li;
  const float b1v = (B1 != nullptr && wave < G::NW12 && c12 < hd) ? B1[c12] : 0.f;
  const float b2v = (B2 != nullptr && wave < G::NW12 && c12 < hd) ? B2[c12] : 0.f;
  const int col = 16 * (wave % G::CB3) + li;                // this thread's state column (owner waves)
  const float b3v = (B3 != nullptr && wave < G::NW3 && col < d) ? B3[col] : 0.f;
  const bool owner = wave < G::NW3 && col < d;
  const int rbase = 16 * (wave / G::CB3) + 4 * lg;          // + i : row inside the tile

  Acc acc;
  const long long ntiles = (A.batch + G::R - 1) / G::R;
  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    const long long row0 = tile_i * G::R;
    float hs = P.hs;
    asm volatile("" : "+v"(hs));                            // keep dt*coefficient products out of long-lived registers
    float y0e[4], k[S + 1][4], ys[4], kn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long row = row0 + rbase + i;
      const bool ok = owner && row < A.batch;
      y0e[i] = ok ? P.y0[row * d + col] : 0.f;
      k[0][i] = (ok && MODE != MLP_F0) ? P.f0[row * d + col] : 0.f;
      if (MODE == MLP_F0 && ok) {
        if (M.copy_a != nullptr) ((float*)M.copy_a)[row * d + col] = y0e[i];
        if (M.copy_b != nullptr) ((float*)M.copy_b)[row * d + col] = y0e[i];
      }
    }
    // input tile -> LDS (owner threads write their elements; padded columns of s_x must read as zero)
    auto put_x = [&](const float* v4) {
      if (wave < G::NW3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_x[(rbase + i) * G::LDX + col] = (col < d) ? v4[i] : 0.f;
      }
    };
    if (MODE == MLP_F0) {
      put_x(y0e);
      mlp_eval<DP, HP>(s_x, s_h1, s_h2, w1f, w2f, w3f, b1v, b2v, b3v, kn);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long row = row0 + rbase + i;
        if (owner && row < A.batch) {
          const float f0 = sign * kn[i];
          P.f1[row * d + col] = f0;
          const float sc = (float)M.atol + fabsf(y0e[i]) * (float)M.rtol;      // misc.py:225
          const double q0 = (double)(y0e[i] / sc), q1 = (double)(f0 / sc);
          acc.suma += q0 * q0; acc.sumb += q1 * q1;
          if (!finite_(y0e[i])) acc.flag = 1;
        }
      }
      __syncthreads();
      continue;
    }
    if (MODE == MLP_INITB) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ys[i] = y0e[i] + hs * k[0][i];               // misc.py:235
      put_x(ys);
      mlp_eval<DP, HP>(s_x, s_h1, s_h2, w1f, w2f, w3f, b1v, b2v, b3v, kn);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long row = row0 + rbase + i;
        if (owner && row < A.batch) {
          const float sc = (float)M.atol + fabsf(y0e[i]) * (float)M.rtol;
          const double q = (double)((sign * kn[i] - k[0][i]) / sc);            // misc.py:237
          acc.suma += q * q;
        }
      }
      __syncthreads();
      continue;
    }
    // ---- MLP_STEP: all S stages --------------------------------------------------------------------------
    auto stage = [&](auto sg_c) {
      constexpr int SG = decltype(sg_c)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float kk[SG];
#pragma unroll
        for (int j = 0; j < SG; ++j) kk[j] = k[j][i];
        ys[i] = step_combine<float, SG>(y0e[i], kk, hs, A);
      }
      put_x(ys);
      mlp_eval<DP, HP>(s_x, s_h1, s_h2, w1f, w2f, w3f, b1v, b2v, b3v, kn);
#pragma unroll
      for (int i = 0; i < 4; ++i) k[SG][i] = sign * kn[i];
    };
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    if constexpr (S == 6) {
      stage(std::integral_constant<int, 4>{});
      stage(std::integral_constant<int, 5>{});
      stage(std::integral_constant<int, 6>{});
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long row = row0 + rbase + i;
      if (owner && row < A.batch) {
        float kk[S + 1];
#pragma unroll
        for (int j = 0; j <= S; ++j) kk[j] = k[j][i];
        float err, ymid;
        step_finish<float, S>(y0e[i], kk, hs, A, err, ymid);
        const long long idx = row * d + col;
        P.y1[idx] = ys[i];
        P.f1[idx] = k[S][i];
        step_emit<float, S, TS>(A, P, y0e[i], ys[i], kk, ymid, idx, A.t_out);
        acc.maxa = fmax(acc.maxa, (double)fabsf(y0e[i]));
        acc.maxb = fmax(acc.maxb, (double)fabsf(ys[i]));
        acc.suma += (double)err * (double)err;
      }
    }
    __syncthreads();
  }
  if constexpr (MODE == MLP_STEP) finish_attempt(A, acc, red);
  else block_reduce_store(acc, red, A.partials + (long long)blockIdx.x * kRec);
}

}  // namespace mi
