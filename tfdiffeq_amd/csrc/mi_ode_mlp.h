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
#include "mi_ode_persist.h"
#include "mi_ode_step_fused.h"

namespace mi {

enum MlpMode { MLP_F0 = 0, MLP_INITB = 1, MLP_STEP = 2 };

// As in the MFMA-linear family (MI_LIN_FMA, mi_ode_dev.h): the stage / error combinations of the MLP tile kernels are chains of
// fused multiply-adds - one rounding per term where the reference's add_n((scale * c_j) * k_j) has two.  The fp32 vector ALU shares
// its issue slots with the fp32 matrix pipe (scripts/micro/mfma_fill.hip), and this family never was bit-comparable with the
// reference (tanh / softplus from the transcendental unit, the matrix products' summation order): its parity bar are the float32
// bands of tests/bands.py.  0 restores the two roundings.
#ifndef MI_MLP_FMA
#define MI_MLP_FMA 1
#endif
template <int SG>
__device__ __forceinline__ float mlp_combine(float y0, const float* k, float hs, const StepArgs& A) {
  float acc = (hs * (float)A.beta[SG - 1][0]) * k[0];
#pragma unroll
  for (int j = 1; j < SG; ++j) acc = madd<(MI_MLP_FMA != 0), float>(hs * (float)A.beta[SG - 1][j], k[j], acc);
  return y0 + acc;
}
template <int S>
__device__ __forceinline__ float mlp_error(const float* k, float hs, const StepArgs& A) {
  float er = (hs * (float)A.e[0]) * k[0];
#pragma unroll
  for (int j = 1; j <= S; ++j) er = madd<(MI_MLP_FMA != 0), float>(hs * (float)A.e[j], k[j], er);
  return er;
}

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

// tanh for the hidden layers: 2^(x * 2/ln 2) on the transcendental unit (v_exp_f32), one reciprocal, and the odd Taylor
// polynomial below |x| = 1/4 where 1 - 2/(e+1) would cancel: 15 VALU instructions instead of the 27 of ocml's tanhf (the
// fp32 VALU shares its issue slots with the fp32 matrix pipe, scripts/micro/mfma_fill.hip: the activation was ~1/5 of the
// kernel).  Max relative error 5e-7 (absolute 1.3e-7) over the whole range (the reference's tf.tanh is a rational fp32 approximation of the
// same class, not a correctly rounded one either).
// (Round 5 A/B, -DMI_MLP_TANH_RATIONAL: x P(x^2) / Q(x^2) on [-7.9, 7.9], deg P = 5, deg Q = 3 - a least-squares / Lawson fit made for this
// file, 3.0e-7 relative in fp32 with fused multiply-adds - ONE transcendental instruction and 12 others per value instead of two and
// 15, all packed (the measured build also started the layers' MFMA chains from the bias instead of adding it behind them).  Measured
// (profiles/r05_tanh_ab.txt): config 5 0.379 -> 0.374 ms per call, kernel 0.339 -> 0.330 ms; against a float64
// solve the fused adjoint's gradients are closer with the form below in two cases of three (theta 2.1e-7 vs 1.1e-6, 2.6e-5 vs 3.2e-5,
// 5.0e-6 vs 4.1e-6).  A third fewer issue slots for 1.3 - 2.6 %: v_exp_f32 / v_rcp_f32 are not quarter rate on this part, and the
// activation is less of the critical path than its instruction count says.  Not adopted: the bits of every MLP path would change for that.)
#ifdef MI_MLP_TANH_RATIONAL
#define MI_TANH_P1 1.298491806e-01f
#define MI_TANH_P2 2.992105903e-03f
#define MI_TANH_P3 9.939260963e-06f
#define MI_TANH_P4 -1.490644053e-08f
#define MI_TANH_P5 2.410562548e-11f
#define MI_TANH_Q1 4.631823599e-01f
#define MI_TANH_Q2 2.405313030e-02f
#define MI_TANH_Q3 2.380880178e-04f
#define MI_TANH_XM 7.9f
__device__ __forceinline__ float mlp_tanh(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -MI_TANH_XM, MI_TANH_XM);
  const float u = xc * xc;
  float p = __builtin_fmaf(MI_TANH_P5, u, MI_TANH_P4);
  p = __builtin_fmaf(p, u, MI_TANH_P3);
  p = __builtin_fmaf(p, u, MI_TANH_P2);
  p = __builtin_fmaf(p, u, MI_TANH_P1);
  p = __builtin_fmaf(p, u, 1.0f);
  float q = __builtin_fmaf(MI_TANH_Q3, u, MI_TANH_Q2);
  q = __builtin_fmaf(q, u, MI_TANH_Q1);
  q = __builtin_fmaf(q, u, 1.0f);
  const float t = (xc * p) * __builtin_amdgcn_rcpf(q);
  return __builtin_fmaf(0.0f, x, t);
}
#else
__device__ __forceinline__ float mlp_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);      // exp(2x); +inf / 0 at the ends give exactly +-1
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float x2 = x * x;
  const float small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
  return fabsf(x) < 0.25f ? small : big;
}
#endif

// The hidden activation: mi_ode_rhs.scalars[0], a TEMPLATE parameter of the kernels (a run-time switch made the compiler evaluate
// every branch and select: config 5 went from 0.42 to 0.54 ms).  0 = tanh (above); 1 = relu - the DEFAULT of the reference's
// ODEFunc (dense_odenet.py:14, 46-47); 2 = softplus (dense_odenet.py:48-49).  mlp_act_deriv: d act / d z from the activation's
// OUTPUT h (what the adjoint kernel keeps): tanh 1 - h^2, relu [h > 0], softplus sigmoid(z) = 1 - exp(-h).
enum { MLP_ACT_TANH = 0, MLP_ACT_RELU = 1, MLP_ACT_SOFTPLUS = 2 };
// (the softplus arithmetic is built from the transcendental unit's exp2 / log2 like mlp_tanh - ocml's expf / log1pf inlined at
// sixteen call sites cost the tanh path registers.  log1p(u) = log(w) u / (w - 1) with w = fl(1 + u): the classic correction
// for the rounding of 1 + u.)
__device__ __forceinline__ float mlp_softplus(float x) {
  const float u = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);              // e^x
  const float w = 1.0f + u, dw = w - 1.0f;
  const float l1p = dw == 0.f ? u : (__builtin_amdgcn_logf(w) * 0.6931471805599453f) * (u * __builtin_amdgcn_rcpf(dw));   // v_log_f32 is log2
  return x > 20.f ? x : l1p;                                                    // x > 20: log(1 + e^x) = x in fp32 (every framework's guard)
}
template <int ACT>
__device__ __forceinline__ float mlp_act(float x) {
  if constexpr (ACT == MLP_ACT_TANH) return mlp_tanh(x);
  else if constexpr (ACT == MLP_ACT_RELU) return x > 0.f ? x : (x != x ? x : 0.f);            // NaN stays NaN
  else return mlp_softplus(x);
}
// Two activations at once (plus the bias add in front): for tanh the affine / polynomial parts run as PACKED fp32 instructions
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two IEEE operations per lane and issue slot, same roundings - the results are
// the bits mlp_tanh gives, element for element); v_exp_f32 / v_rcp_f32 and the final select stay scalar.  Vector instructions do
// not overlap with the matrix pipe of their SIMD on this part (profiles/r03_mfma_pair.txt) and tanh is ~3/4 of the vector
// work of an evaluation, so halving the issue slots of its arithmetic is worth more than anything a schedule can hide.
typedef float mlp_f2 __attribute__((ext_vector_type(2)));
#ifdef MI_MLP_TANH_RATIONAL
__device__ __forceinline__ mlp_f2 mlp_f2_splat(float v) { mlp_f2 r = {v, v}; return r; }
__device__ __forceinline__ mlp_f2 mlp_tanh2(mlp_f2 x) {                        // mlp_tanh on two values: the same operations, the same bits
  mlp_f2 xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -MI_TANH_XM, MI_TANH_XM); xc.y = __builtin_amdgcn_fmed3f(x.y, -MI_TANH_XM, MI_TANH_XM);
  const mlp_f2 u = xc * xc;
  mlp_f2 p = __builtin_elementwise_fma(mlp_f2_splat(MI_TANH_P5), u, mlp_f2_splat(MI_TANH_P4));
  p = __builtin_elementwise_fma(p, u, mlp_f2_splat(MI_TANH_P3));
  p = __builtin_elementwise_fma(p, u, mlp_f2_splat(MI_TANH_P2));
  p = __builtin_elementwise_fma(p, u, mlp_f2_splat(MI_TANH_P1));
  p = __builtin_elementwise_fma(p, u, mlp_f2_splat(1.0f));
  mlp_f2 q = __builtin_elementwise_fma(mlp_f2_splat(MI_TANH_Q3), u, mlp_f2_splat(MI_TANH_Q2));
  q = __builtin_elementwise_fma(q, u, mlp_f2_splat(MI_TANH_Q1));
  q = __builtin_elementwise_fma(q, u, mlp_f2_splat(1.0f));
  mlp_f2 r;
  r.x = __builtin_amdgcn_rcpf(q.x); r.y = __builtin_amdgcn_rcpf(q.y);
  const mlp_f2 t = (xc * p) * r;
  return __builtin_elementwise_fma(mlp_f2_splat(0.0f), x, t);
}
#else
__device__ __forceinline__ mlp_f2 mlp_tanh2(mlp_f2 x) {
  const mlp_f2 t = x * 2.8853900817779268f;
  mlp_f2 e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const mlp_f2 d = e + 1.0f;
  mlp_f2 r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  const mlp_f2 big = 1.0f - 2.0f * r;                                          // (2 r is exact: the same value fused or not)
  const mlp_f2 x2 = x * x;
  const mlp_f2 small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
  mlp_f2 out;
  out.x = fabsf(x.x) < 0.25f ? small.x : big.x;
  out.y = fabsf(x.y) < 0.25f ? small.y : big.y;
  return out;
}
#endif
template <int ACT>
__device__ __forceinline__ void mlp_act_pair(float a, float b, float bias, float& oa, float& ob) {
  if constexpr (ACT == MLP_ACT_TANH) {
    mlp_f2 x = {a, b};
    x = x + bias;
    const mlp_f2 y = mlp_tanh2(x);
    oa = y.x; ob = y.y;
  } else {
    oa = mlp_act<ACT>(a + bias); ob = mlp_act<ACT>(b + bias);
  }
}
template <int ACT>
__device__ __forceinline__ float mlp_act_deriv(float h) {
  if constexpr (ACT == MLP_ACT_TANH) return 1.0f - h * h;
  else if constexpr (ACT == MLP_ACT_RELU) return h > 0.f ? 1.0f : 0.f;
  else {
    const float small = h * (1.0f - h * (0.5f - h * 0.16666667f));                // 1 - e^-h without the cancellation, h < 2^-6
    return h < 0.015625f ? small : 1.0f - __builtin_amdgcn_exp2f(h * -1.4426950408889634f);
  }
}

// One evaluation of the MLP for the tile whose input rows sit in s_x.  Every thread of the workgroup must call it.
// Owner threads (wave < NW3) receive their 4 output elements (rows rb*16 + 4*(lane>>4) + i, column 16*cb + (lane&15)).
#ifndef MI_MLP_ABL
#define MI_MLP_ABL 0
#endif
#ifdef MI_TRACE                                              // tuning aid: cycle stamps of the layer phases, workgroup 0, wavefronts 0 and 4
__device__ long long mi_mlp_tr[2][1024];
__device__ int mi_mlp_tn[2];
__device__ __forceinline__ void mlp_stamp() {
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) {
    const int w = threadIdx.x ? 1 : 0;
    const int n = mi_mlp_tn[w];
    if (n < 1024) { mi_mlp_tr[w][n] = (long long)__builtin_readcyclecounter(); mi_mlp_tn[w] = n + 1; }
  }
}
#else
__device__ __forceinline__ void mlp_stamp() {}
#endif
// (Round 4: a 16-row tile on FOUR wavefronts - each owning 32 hidden columns, so that an activation element is read from LDS by 4
// wavefronts instead of 8 - with two such workgroups per CU to fill each other's barrier bubbles was built for the 64 x 128 geometry:
// it needs 354 registers per wavefront (128 of them weight slices); at two wavefronts per SIMD it spills 87 and takes 0.455 ms per
// config-5 call against 0.333, with the whole register file (one wavefront per SIMD, nothing spilled) 0.370 ms.  Removed.)
// (Round 3: issuing the LDS reads of the NEXT group of MFMAs before the current group - explicit operand pipelining with
// sched_group_barrier - changed nothing, 0.351 vs 0.343 ms at config 5: the two wavefronts of a SIMD interleave their fp32 chains and
// cover each other's LDS round trips.  What the reads cost is the start of every chain after its barrier (profiles/r03_mlp_timeline.txt).
// Two accumulators in the dependent chain of layer 3: no change either - 2816 vs 2812 cycles.)
template <int DP, int HP, int ACT>
__device__ __forceinline__ void mlp_eval(float* s_x, float* s_h1, float* s_h2, const float* w1f, const float* w2f,
                                         const float* w3f, float b1v, float b2v, float b3v, float* out4) {
  using G = MlpGeom<DP, HP>;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  constexpr int KS1 = DP / 4, KS2 = HP / 4;
#if (MI_MLP_ABL & 1)
  f4 fake = {1.0f, 0.5f, 0.25f, 0.125f};
  asm volatile("" : "+v"(fake));
#define MI_MLP_LD(p) fake
#elif (MI_MLP_ABL & 2)                                      // the reads are issued (and their results kept alive) but nothing waits for them
  f4 fake = {1.0f, 0.5f, 0.25f, 0.125f};
  asm volatile("" : "+v"(fake));
  auto mi_mlp_ld2 = [&](const float* p_) {
    f4 t_;
    asm volatile("ds_read_b128 %0, %1" : "=v"(t_) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const float*)p_));
    return fake;
  };
#define MI_MLP_LD(p) mi_mlp_ld2(p)
#else
#define MI_MLP_LD(p) (*(const f4*)(p))
#endif
  mlp_stamp();
  __syncthreads();                                          // s_x is complete
  mlp_stamp();
  if (wave < G::NW12) {                                     // layer 1: [32 x DP] @ [DP x 16]
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const float* a0p = s_x + li * G::LDX + lg * KS1;
    const float* a1p = s_x + (16 + li) * G::LDX + lg * KS1;
#pragma unroll
    for (int m = 0; m < KS1 / 4; ++m) {
      const f4 a0 = MI_MLP_LD(a0p + 4 * m), a1 = MI_MLP_LD(a1p + 4 * m);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[v], w1f[4 * m + v], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[v], w1f[4 * m + v], c1, 0, 0, 0);
      }
    }
    mlp_stamp();
    const int col = 16 * wave + li;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float h0, h1;
      mlp_act_pair<ACT>(c0[i], c1[i], b1v, h0, h1);
      s_h1[(4 * lg + i) * G::LDH + col] = h0;
      s_h1[(16 + 4 * lg + i) * G::LDH + col] = h1;
    }
  }
  mlp_stamp();
  __syncthreads();
  mlp_stamp();
  if (wave < G::NW12) {                                     // layer 2: [32 x HP] @ [HP x 16]
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const float* a0p = s_h1 + li * G::LDH + lg * KS2;
    const float* a1p = s_h1 + (16 + li) * G::LDH + lg * KS2;
#pragma unroll
    for (int m = 0; m < KS2 / 4; ++m) {
      const f4 a0 = MI_MLP_LD(a0p + 4 * m), a1 = MI_MLP_LD(a1p + 4 * m);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[v], w2f[4 * m + v], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[v], w2f[4 * m + v], c1, 0, 0, 0);
      }
    }
    mlp_stamp();
    const int col = 16 * wave + li;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float h0, h1;
      mlp_act_pair<ACT>(c0[i], c1[i], b2v, h0, h1);
      s_h2[(4 * lg + i) * G::LDH + col] = h0;
      s_h2[(16 + 4 * lg + i) * G::LDH + col] = h1;
    }
  }
  mlp_stamp();
  __syncthreads();
  mlp_stamp();
  if (wave < G::NW3) {                                      // layer 3: one 16-row block x 16 output columns per wave
    const int rb = wave / G::CB3;
    f4 c = {0, 0, 0, 0};
    const float* ap = s_h2 + (16 * rb + li) * G::LDH + lg * KS2;
#pragma unroll
    for (int m = 0; m < KS2 / 4; ++m) {
      const f4 a = MI_MLP_LD(ap + 4 * m);
#pragma unroll
      for (int v = 0; v < 4; ++v) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v], w3f[4 * m + v], c, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out4[i] = c[i] + b3v;
  }
  mlp_stamp();
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

// Per-thread context of the MLP tile kernels: resident weight slices (zero padded), biases, LDS tiles, element map.
template <int DP, int HP, int ACT>
struct MlpCtx {
  using G = MlpGeom<DP, HP>;
  static constexpr int KS1 = DP / 4, KS2 = HP / 4;
  float *s_x, *s_h1, *s_h2;
  float w1f[KS1], w2f[KS2], w3f[KS2];
  float b1v, b2v, b3v, sign;
  float wtv;                  // time-dependent first layer (rhs.s[1] != 0): W1 is [d + 1, hd], row 0 multiplies t; else 0
  int lane, wave, li, lg, d, hd, col, rbase;
  bool owner;

  __device__ __forceinline__ void init(const RhsParams& rhs, int dim, char* smem) {
    s_x = (float*)smem;
    s_h1 = s_x + G::R * G::LDX;
    s_h2 = s_h1 + G::R * G::LDH;
    lane = threadIdx.x & 63; wave = threadIdx.x >> 6; li = lane & 15; lg = lane >> 4;
    d = dim; hd = rhs.hidden;
    const float* W1 = (const float*)rhs.w[0];
    const float* W2 = (const float*)rhs.w[1];
    const float* W3 = (const float*)rhs.w[2];
    const float* B1 = (const float*)rhs.b[0];
    const float* B2 = (const float*)rhs.b[1];
    const float* B3 = (const float*)rhs.b[2];
    sign = (float)rhs.sign;
    // resident weight slices, zero padded: lane (col = li, group lg) holds W[k = lg*KS + s][16*block + li]
    const int c12 = 16 * wave + li;
    const int td = rhs.s[1] != 0.0 ? 1 : 0;                   // dense_odenet.py:79-84: fc1 sees concat([t, x])
    wtv = (td && wave < G::NW12 && c12 < hd) ? W1[c12] : 0.f;
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      const int k = lg * KS1 + s;
      w1f[s] = (wave < G::NW12 && k < d && c12 < hd) ? W1[(long long)(k + td) * hd + c12] : 0.f;
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
    b1v = (B1 != nullptr && wave < G::NW12 && c12 < hd) ? B1[c12] : 0.f;
    b2v = (B2 != nullptr && wave < G::NW12 && c12 < hd) ? B2[c12] : 0.f;
    col = 16 * (wave % G::CB3) + li;                          // this thread's state column (owner waves)
    b3v = (B3 != nullptr && wave < G::NW3 && col < d) ? B3[col] : 0.f;
    owner = wave < G::NW3 && col < d;
    rbase = 16 * (wave / G::CB3) + 4 * lg;                    // + i : row inside the tile
  }
  // input tile -> LDS (owner threads write their elements; padded columns of s_x must read as zero)
  __device__ __forceinline__ void put_x(const float* v4) {
    if (wave < G::NW3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s_x[(rbase + i) * G::LDX + col] = (col < d) ? v4[i] : 0.f;
    }
  }
  // ts: the time the network sees (already multiplied by the direction sign); it only shifts the first layer's bias
  __device__ __forceinline__ void eval(float* out4, float ts) {
    mlp_eval<DP, HP, ACT>(s_x, s_h1, s_h2, w1f, w2f, w3f, b1v + ts * wtv, b2v, b3v, out4);
  }
};

// One pass over this workgroup's tiles: MODE F0 (f0 + the norms of misc._select_initial_step, seeds copy_a / copy_b),
// INITB (second half of _select_initial_step), STEP (one adaptive attempt).  SC0 as in the linear kernels.
template <int DP, int HP, int ACT, int MODE, int S, bool TS, bool SC0>
__device__ __forceinline__ void mlp_pass(const StepArgs& A, const StepPlanes<float, S>& P, void* copy_a, void* copy_b,
                                         MlpCtx<DP, HP, ACT>& cx, Acc& acc, const double* t_out) {
  using G = MlpGeom<DP, HP>;
  const int d = cx.d, col = cx.col, rbase = cx.rbase;
  const bool owner = cx.owner;
  const float sign = cx.sign;
  const long long ntiles = (A.batch + G::R - 1) / G::R;
  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    const long long row0 = tile_i * G::R;
    float hs = P.hs;
    asm volatile("" : "+v"(hs));                            // keep dt*coefficient products out of long-lived registers
    float y0e[4], k[S + 1][4], ys[4], kn[4];
    // (measured in round 3: prefetching the next tile's y0 / f0 into registers, as the linear kernels do, changes nothing here -
    // 0.3385 vs 0.3386 ms per config-5 call - and costs 12 more spilled registers: the loads stay where they are)
    // (scalar tile base + 32-bit element offsets: no 64-bit vector index arithmetic per access, as in the linear tile kernels)
    const long long left = A.batch - row0;
    const int nr = left < G::R ? (int)left : G::R;           // rows of this tile that exist
    const long long tb = row0 * d;
    const unsigned e0 = (unsigned)(rbase * d + col);         // element i of this thread: tb + e0 + i * d
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = owner && rbase + i < nr;
      const unsigned eo = e0 + (unsigned)(i * d);
      y0e[i] = ok ? stream_load<SC0>(P.y0 + tb + eo) : 0.f;
      k[0][i] = (ok && MODE != MLP_F0) ? stream_load<SC0>(P.f0 + tb + eo) : 0.f;
      if (MODE == MLP_F0 && ok) {
        if (copy_a != nullptr) ((float*)copy_a + tb)[eo] = y0e[i];
        if (copy_b != nullptr) ((float*)copy_b + tb)[eo] = y0e[i];
      }
    }
    if (MODE == MLP_F0) {
      cx.put_x(y0e);
      cx.eval(kn, sign * P.t0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (owner && rbase + i < nr) {
          const float f0 = sign * kn[i];
          (P.f1 + tb)[e0 + (unsigned)(i * d)] = f0;
          const float sc = (float)A.cp.atol + fabsf(y0e[i]) * (float)A.cp.rtol;      // misc.py:225
          const double q0 = (double)(y0e[i] / sc), q1 = (double)(f0 / sc);
          acc.suma += q0 * q0; acc.sumb += q1 * q1;
          if (!finite_(y0e[i])) acc.flag = 1;
        }
      }
      
      continue;
    }
    if (MODE == MLP_INITB) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ys[i] = y0e[i] + hs * k[0][i];               // misc.py:235
      cx.put_x(ys);
      cx.eval(kn, sign * (P.t0 + hs));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (owner && rbase + i < nr) {
          const float sc = (float)A.cp.atol + fabsf(y0e[i]) * (float)A.cp.rtol;
          const double q = (double)((sign * kn[i] - k[0][i]) / sc);            // misc.py:237
          acc.suma += q * q;
        }
      }
      
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
        ys[i] = mlp_combine<SG>(y0e[i], kk, hs, A);
      }
      cx.put_x(ys);
      cx.eval(kn, sign * (P.t0 + (float)A.alpha[SG - 1] * hs));                 // rk_common.py:50, in the state dtype
#pragma unroll
      for (int i = 0; i < 4; ++i) k[SG][i] = sign * kn[i];
    };
    for_stages<1, S>(stage);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (owner && rbase + i < nr) {
        float kk[S + 1];
#pragma unroll
        for (int j = 0; j <= S; ++j) kk[j] = k[j][i];
        float err, ymid;
        if (!TS && P.j_hi > P.j_lo) step_finish<float, S>(y0e[i], kk, hs, A, err, ymid, true);     // (y_mid: the plain form, rare)
        else ymid = y0e[i];
        err = mlp_error<S>(kk, hs, A);
        const unsigned eo = e0 + (unsigned)(i * d);
        const long long idx = tb + eo;
        (P.y1 + tb)[eo] = ys[i];
        (P.f1 + tb)[eo] = k[S][i];
        step_emit<float, S, TS>(A, P, y0e[i], ys[i], kk, ymid, idx, t_out);
        acc.maxa = fmax(acc.maxa, (double)fabsf(y0e[i]));
        acc.maxb = fmax(acc.maxb, (double)fabsf(ys[i]));
        acc.suma += (double)err * (double)err;
      }
    }
    // (no barrier at the end of a tile: the next tile's first LDS write is to s_x, which every wavefront finished reading before it
    // passed the second barrier of this tile's last evaluation; s_h1 / s_h2 are only written behind the next evaluation's barriers)
  }
}

// Fixed grid (solvers.py:82-104) for the ODEFunc MLP: trajectories never interact on a fixed grid, so a 32-row tile runs through
// EVERY grid interval with y and k_1..k_4 in registers (Euler: fixed_grid.py:6-7; RK4 3/8 rule: rk_common.py:73-81 - the
// arithmetic of k_fixed_rowlocal / k_fixed_linear_mfma, true divisions included), the weight slices stay resident, solution[i+1]
// is streamed; one launch per call.  Traffic = y0 in + T solution rows out; bound: fp32 matrix pipe.  The time the network
// sees (time-dependent first layer, dense_odenet.py:79-84) is the step function's t + eps and the 3/8 rule's stage times.
template <int DP, int HP, int ACT>
__global__ __launch_bounds__((64 * MlpGeom<DP, HP>::NW)) void k_fixed_mlp(FixedArgs A) {
  using G = MlpGeom<DP, HP>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  MlpCtx<DP, HP, ACT> cx;
  cx.init(A.rhs, A.dim, smem_raw);
  const int d = cx.d;
  const float sign = cx.sign;
  const long long ntiles = (A.batch + G::R - 1) / G::R;
  const long long n = A.batch * d;
  const float* y0p = (const float*)A.y0;
  float* out = (float*)A.out;
  const float eps = (float)A.eps;
  FixedClk clk;
  clk.begin();
  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    float y[4];
    long long idx[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long row = tile_i * G::R + cx.rbase + i;
      ok[i] = cx.owner && row < A.batch;
      idx[i] = row * d + cx.col;
      y[i] = ok[i] ? y0p[idx[i]] : 0.f;
      if (ok[i]) out[idx[i]] = y[i];                          // solution = [y0]
    }
    int j = 1;
    for (int s = 0; s < A.M; ++s) {
      const float t0 = (float)A.grid[s];                      // solvers.py:84: the grid is cast to the STATE dtype
      const float t1 = (float)A.grid[s + 1];
      const float dt = t1 - t0;
      const float te = t0 + eps;                              // fixed_grid.py:7 / :42
      float k1[4], k2[4], k3[4], k4[4], ys[4], yn[4], kn[4];
      cx.put_x(y);
      cx.eval(kn, sign * te);
#pragma unroll
      for (int i = 0; i < 4; ++i) k1[i] = sign * kn[i];
      if (!A.rk4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) yn[i] = y[i] + dt * k1[i];                          // fixed_grid.py:7
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ys[i] = y[i] + dt * k1[i] / 3.0f;                   // rk_common.py:77
        cx.put_x(ys);
        cx.eval(kn, sign * (te + dt / 3.0f));
#pragma unroll
        for (int i = 0; i < 4; ++i) { k2[i] = sign * kn[i]; ys[i] = y[i] + dt * (k1[i] / -3.0f + k2[i]); }    // :78
        cx.put_x(ys);
        cx.eval(kn, sign * (te + dt * 2.0f / 3.0f));
#pragma unroll
        for (int i = 0; i < 4; ++i) { k3[i] = sign * kn[i]; ys[i] = y[i] + dt * (k1[i] - k2[i] + k3[i]); }    // :79
        cx.put_x(ys);
        cx.eval(kn, sign * (te + dt));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          k4[i] = sign * kn[i];
          yn[i] = y[i] + (k1[i] + 3.0f * k2[i] + 3.0f * k3[i] + k4[i]) * (dt / 8.0f);                       // :81
        }
      }
      while (j < A.T && t1 >= (float)A.t[j]) {               // solvers.py:97-100, _linear_interp :106-115
        const float tj = (float)A.t[j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (ok[i]) out[(long long)j * n + idx[i]] = (tj == t0) ? y[i] : ((tj == t1) ? yn[i] : y[i] + ((yn[i] - y[i]) / (t1 - t0)) * (tj - t0));
        ++j;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = yn[i];
    }
  }
  clk.end(A.clk);
}

template <int DP, int HP, int ACT, int MODE, int S, bool TS>
__global__ __launch_bounds__((64 * MlpGeom<DP, HP>::NW)) void k_mlp(MlpArgs M) {
  using G = MlpGeom<DP, HP>;
  const StepArgs& A = M.step;
  StepPlanes<float, S> P;
  if (MODE == MLP_F0) {
    P.y0 = (const float*)M.x_y0;
    P.f0 = nullptr; P.y1 = nullptr; P.hs = 0.f; P.t0 = (float)A.ctl->t1; P.j_lo = P.j_hi = 0;
    P.f1 = (float*)(A.planes + 2 * A.stride);               // F0 writes f0 into idx_k[0] of a fresh handle
  } else {
    if (!resolve_step<float, S>(A, P)) return;
    if (MODE == MLP_INITB) P.hs = (float)A.ctl->h0;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  MlpCtx<DP, HP, ACT> cx;
  cx.init(A.rhs, A.dim, smem_raw);
  double* red = (double*)(cx.s_h2 + G::R * G::LDH);
  Acc acc;
  mlp_pass<DP, HP, ACT, MODE, S, TS, false>(A, P, M.copy_a, M.copy_b, cx, acc, A.t_out);
  if constexpr (MODE == MLP_STEP) finish_attempt(A, acc, red);
  else block_reduce_store(acc, red, A.partials + (long long)blockIdx.x * kRec);
}

// The whole call in one launch (see mi_ode_persist.h): before_integrate, every attempt, controller and dense output on
// the persistent tile grid; weights are loaded once per call.  Same planes / hand-off / redundant controller as
// k_persist_linear_mfma.
template <int DP, int HP, int ACT, int S, bool TS>
__global__ __launch_bounds__((64 * MlpGeom<DP, HP>::NW)) void k_persist_mlp(PersistArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ PersistShared sh;
  Ctl& s_c = sh.c;
  MlpCtx<DP, HP, ACT> cx;
  cx.init(A.s.rhs, A.s.dim, smem_raw);
  CtrlParams cp = A.s.cp;
  cp.t_out = persist_stage_tout(A, sh.tout);
  const double* t_out = cp.t_out;
  unsigned gen = 0;
  double r[5], rec[kRec], n_tot = 0.0;
  if (threadIdx.x == 0) { persist_init_ctl(s_c, A); sh.ok = 1; }
  __syncthreads();

  float* const ya = (float*)(A.s.planes);
  float* const yb = (float*)(A.s.planes + A.s.stride);
  float* const fa = (float*)(A.s.planes + 2 * A.s.stride);
  float* const fb = (float*)(A.s.planes + (long long)(2 + S) * A.s.stride);
  const float* const y_user = (const float*)A.y0;

  bool ok;
  {
    StepPlanes<float, S> P;
    P.y0 = y_user; P.f0 = nullptr; P.y1 = nullptr; P.f1 = fa; P.hs = 0.f; P.t0 = (float)A.t0; P.j_lo = P.j_hi = 0;
    Acc acc;
    mlp_pass<DP, HP, ACT, MLP_F0, S, TS, true>(A.s, P, nullptr, A.out0, cx, acc, t_out);
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_F0, cp); }
    __syncthreads();
  }
  if (cp.auto_first_step && ok) {
    StepPlanes<float, S> P;
    P.y0 = y_user; P.f0 = fa; P.y1 = nullptr; P.f1 = nullptr; P.hs = (float)uniform_d(s_c.h0); P.t0 = (float)A.t0; P.j_lo = P.j_hi = 0;
    Acc acc;
    mlp_pass<DP, HP, ACT, MLP_INITB, S, TS, true>(A.s, P, nullptr, nullptr, cx, acc, t_out);
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_INITB, cp); }
  }
  auto publish = [&](const AttemptState& st) {                // thread 0: what the next attempt needs
    sh.pub.dt = st.dt; sh.pub.t1 = st.t1; sh.pub.accepted = st.accepted; sh.pub.done = st.done;
    int j = st.next_out;                                      // speculative output range of the NEXT attempt (resolve_step)
    const double t_new = st.t1 + st.dt;
    while (j < st.n_out && !(t_out[j] > t_new)) ++j;
    sh.pub.emit_lo = st.next_out; sh.pub.emit_hi = j;
  };
  if (threadIdx.x == 0) {
    if (!ok) { s_c.status |= MI_ODE_ST_SYNC_TIMEOUT; s_c.done = 1; }
    else set_outputs_apply(&s_c, A.n_out);
    AttemptState st;
    st.load(s_c);
    st.accepted = 0;
    publish(st);
    sh.st = st;
  }
  __syncthreads();

  const float* cur_y = y_user;
  float* cur_f = fa;
  while (!uniform_i(sh.pub.done)) {
    StepPlanes<float, S> P;
    const double dt_u = uniform_d(sh.pub.dt), t1_u = uniform_d(sh.pub.t1);
    P.y0 = cur_y; P.f0 = cur_f;
    P.y1 = (cur_y == ya) ? yb : ya;
    P.f1 = (cur_f == fa) ? fb : fa;
    P.hs = (float)dt_u; P.t0 = (float)t1_u;
    P.t_start = t1_u; P.dt64 = dt_u; P.t_new = t1_u + dt_u;
    P.j_lo = uniform_i(sh.pub.emit_lo); P.j_hi = uniform_i(sh.pub.emit_hi);
    Acc acc;
    mlp_pass<DP, HP, ACT, MLP_STEP, S, TS, true>(A.s, P, nullptr, nullptr, cx, acc, t_out);
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0) {
      AttemptState st = sh.st;
      if (!ok) { st.status |= MI_ODE_ST_SYNC_TIMEOUT; st.done = 1; st.accepted = 0; }
      else { fill_record(rec, r, n_tot); attempt_core(st, rec, cp); }
      publish(st);
      sh.st = st;
    }
    __syncthreads();
    if (uniform_i(sh.pub.accepted)) { cur_y = P.y1; cur_f = P.f1; }
  }

  if (cur_y == y_user) {                                      // no accepted step (error exit): seed plane 0 with y0
    const long long n = A.s.batch * (long long)A.s.dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) ya[i] = y_user[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sh.st.store(s_c);
    s_c.idx_y0 = (cur_y == yb) ? 1 : 0; s_c.idx_y1 = (cur_y == yb) ? 0 : 1;
    s_c.idx_k[0] = (cur_f == fa) ? 2 : 2 + S; s_c.idx_k[S] = (cur_f == fa) ? 2 + S : 2;
    persist_write_back(A, s_c);
#ifdef MI_TRACE
    for (int w = 0; w < 2; ++w) {                             // 9 stamps per evaluation: [0] entry [1] barrier1 [2] L1 chain [3] act1+write [4] barrier2
      const long long* q = mi_mlp_tr[w];                      // [5] L2 chain [6] act2+write [7] barrier3 [8] L3 chain... [9] end
      for (int e = 20; e < 34 && 9 * e + 9 < mi_mlp_tn[w]; ++e) {
        const long long* p = q + 9 * e;
        printf("[mlp trace] wave %d eval %2d: bar1 %5lld L1 %5lld act1 %5lld bar2 %5lld L2 %5lld act2 %5lld bar3 %5lld L3 %5lld | to next %5lld\n", 4 * w, e,
               p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5] - p[4], p[6] - p[5], p[7] - p[6], p[8] - p[7], p[9] - p[8]);
      }
      mi_mlp_tn[w] = 0;
    }
#endif
  }
}

}  // namespace mi
