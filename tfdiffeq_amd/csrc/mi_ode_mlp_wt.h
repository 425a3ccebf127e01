// EXPERIMENTAL (opt-in: mi_ode_rhs.scalars[2] != 0): barrier-free "wave tile" layout of the whole-call MLP kernel.
//
// k_persist_mlp (mi_ode_mlp.h) keeps the weights in registers and shares the activations of a 32-row tile through LDS: five
// s_barriers per evaluation phase-lock the eight waves of a workgroup, and the kernel sits at 0.55 of the fp32 matrix peak.
// Here every WAVE owns a 16-row tile and runs all three layers on it.  The TRANSPOSED weights are the MFMA A operand and
// live in LDS (136 KB for 64-128-128-64, staged once per launch); the activations are the B operand and never leave
// registers: the accumulator layout of v_mfma_f32_16x16x4_f32 - lane (n, g), register r of block mb holds H[n][16 mb + 4 g + r] -
// IS a legal B layout for the next layer (k-slot g of step (kb, r) = input 16 kb + 4 g + r; any k permutation is legal when A
// and B agree), so there is no LDS round trip and no barrier after the weights are staged.  scripts/micro/mlp_wavetile.hip
// measured the evaluation loop alone at 0.73 (tanh) / 0.88 (relu) of the peak (profiles/r02_mlp_wavetile_probe.txt).
// A lane owns row n = lane & 15 of its wave's tile and the DP/4 columns 16 kb + 4 g + r: 16 elements of y, f and of every
// stage derivative at DP = 64.  Same passes, planes, hand-off and redundant controller as k_persist_mlp; the accumulation
// order of the contractions differs (k runs 4 g + r inside 16-blocks), so results agree to rounding, not bit for bit.
#pragma once
#include "mi_ode_mlp.h"

namespace mi {

typedef float wt_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) float* wt_lds;          // LDS pointers keep their address space (ds_read, 32-bit address)
typedef const __attribute__((address_space(3))) wt_f4* wt_lds4;

template <int DP, int HP>
struct WtGeom {
  static constexpr int NW = 4;                               // waves per workgroup = 1 per SIMD: 374 registers per lane (7 x 16 stage derivatives + activations).
                                                             // NW = 8 (2 per SIMD, 128 + 128 registers, ~40 scratch accesses per evaluation) measured the
                                                             // same 0.474 ms at config 5 and is slower on small batches (fewer, longer workgroups)
  static constexpr int R = 16 * NW;                          // rows per workgroup tile
  static constexpr int KB1 = DP / 16, MB1 = HP / 16, MB3 = DP / 16;
  static constexpr int LW1 = DP + 4, LW2 = HP + 4, LW3 = HP + 4;   // row strides = 4 (mod 64 banks): ds_read_b128 conflict-free
  static constexpr int OFF_W1 = 0, OFF_W2 = OFF_W1 + HP * LW1, OFF_W3 = OFF_W2 + HP * LW2, OFF_B1 = OFF_W3 + DP * LW3;
  static constexpr int OFF_B2 = OFF_B1 + HP, OFF_B3 = OFF_B2 + HP, OFF_WT = OFF_B3 + DP, OFF_TAB = OFF_WT + HP, FLOATS = OFF_TAB + 64;
  static constexpr int TAB_E = 36, TAB_MID = 43;            // float tableau in LDS: beta[6][6], c_error[7], c_mid[7]
  static constexpr size_t lds_bytes() { return (size_t)FLOATS * sizeof(float) + 80 * sizeof(double); }
};


// ---- packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per lane and instruction, same roundings as the
// scalar forms; contraction is off for the whole library) for the VALU share of an evaluation: the hidden activations and the
// stage combinations.  The transcendental v_exp_f32 / v_rcp_f32 have no packed form.
typedef float wt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wt_f2 wt_tanh2(wt_f2 x) {                  // mlp_tanh on two values, operation for operation
  const wt_f2 a = x * 2.8853900817779268f;
  wt_f2 e;
  e[0] = __builtin_amdgcn_exp2f(a[0]); e[1] = __builtin_amdgcn_exp2f(a[1]);
  const wt_f2 ep = e + 1.0f;
  wt_f2 rc;
  rc[0] = __builtin_amdgcn_rcpf(ep[0]); rc[1] = __builtin_amdgcn_rcpf(ep[1]);
  const wt_f2 big = 1.0f - 2.0f * rc;
  const wt_f2 x2 = x * x;
  const wt_f2 small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
  wt_f2 out;
  out[0] = fabsf(x[0]) < 0.25f ? small[0] : big[0];
  out[1] = fabsf(x[1]) < 0.25f ? small[1] : big[1];
  return out;
}
template <int ACT>
__device__ __forceinline__ void wt_act4(wt_f4& c) {
  if constexpr (ACT == MLP_ACT_TANH) {
    wt_f2 lo = {c[0], c[1]}, hi = {c[2], c[3]};
    lo = wt_tanh2(lo); hi = wt_tanh2(hi);
    c[0] = lo[0]; c[1] = lo[1]; c[2] = hi[0]; c[3] = hi[1];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = mlp_act<ACT>(c[r]);
  }
}

// one layer on a 16-row tile: out[mb] = act(bias + ts * tw + sum_k Wt[16 mb + .][k] * in[.][k]), weights transposed [out][in] in
// LDS; two output blocks at a time = two independent accumulator chains (a dependent MFMA issues every 64 cycles, its latency is 40)
template <int MB, int KB, int LW, int ACT, bool TW>
__device__ __forceinline__ void wt_layer(wt_lds wt, wt_lds bias, wt_lds tw, float ts, const wt_f4* in, wt_f4* out, int m, int g) {
  static_assert(MB % 2 == 0, "blocks are processed in pairs");
#pragma unroll
  for (int mb = 0; mb < MB; mb += 2) {
    wt_f4 c0 = *(wt_lds4)(bias + 16 * mb + 4 * g);
    wt_f4 c1 = *(wt_lds4)(bias + 16 * (mb + 1) + 4 * g);
    if constexpr (TW) {
      const wt_f4 t0 = *(wt_lds4)(tw + 16 * mb + 4 * g);
      const wt_f4 t1 = *(wt_lds4)(tw + 16 * (mb + 1) + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) { c0[r] = c0[r] + ts * t0[r]; c1[r] = c1[r] + ts * t1[r]; }
    }
    // software pipeline over the 16-column blocks of the input: the A operands of block kb + 1 are read from LDS while the eight
    // MFMAs of block kb run (one wave per SIMD: nobody else hides the LDS latency); only two blocks of operands are ever live
    wt_lds w0 = wt + (16 * mb + m) * LW + 4 * g;
    wt_lds w1 = wt + (16 * (mb + 1) + m) * LW + 4 * g;
    wt_f4 a0n = *(wt_lds4)w0, a1n = *(wt_lds4)w1;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const wt_f4 a0 = a0n, a1 = a1n;
      if (kb + 1 < KB) { a0n = *(wt_lds4)(w0 + 16 * (kb + 1)); a1n = *(wt_lds4)(w1 + 16 * (kb + 1)); }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], in[kb][r], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], in[kb][r], c1, 0, 0, 0);
      }
      // The accumulators pass through an empty asm that also clobbers memory: instruction selection orders the (pure) MFMAs only
      // by their operands, so without this tie it emitted the LDS reads of a whole layer (64 x 16 bytes per lane) in one batch
      // and sank the MFMAs below them - the stage derivatives then lived in scratch.  sched_barrier keeps the machine scheduler
      // from undoing it.
      asm volatile("" : "+a"(c0), "+a"(c1) : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ACT >= 0) {
      wt_act4<ACT>(c0); wt_act4<ACT>(c1);
    }
    out[mb] = c0; out[mb + 1] = c1;
    __builtin_amdgcn_sched_barrier(0);     // keeps the scheduler from hoisting every block's LDS reads to the top (hundreds of spills)
  }
}

// An LDS pointer the optimiser cannot see through: without it the weight reads of the six stage evaluations of an attempt (same
// addresses, no LDS store in between) are merged and ALL weight values of a lane - 2048 of them - are kept live (1900 spills).
__device__ __forceinline__ wt_lds wt_opaque(wt_lds p) {
  asm volatile("" : "+v"(p));
  return p;
}
template <int DP, int HP, int ACT>
struct WtCtx {
  using G = WtGeom<DP, HP>;
  static constexpr int E = DP / 4;                           // state elements per lane
  wt_lds l;                                                  // LDS: transposed weights, biases, time weights
  float sign;
  int lane, wave, m, g, d, hd;

  __device__ __forceinline__ void init(const StepArgs& A, char* smem) {
    const RhsParams& rhs = A.rhs;
    const int dim = A.dim;
    float* w = (float*)smem;
    l = (wt_lds)w;
    lane = threadIdx.x & 63; wave = threadIdx.x >> 6; m = lane & 15; g = lane >> 4;
    d = dim; hd = rhs.hidden; sign = (float)rhs.sign;
    const float* W1 = (const float*)rhs.w[0];
    const float* W2 = (const float*)rhs.w[1];
    const float* W3 = (const float*)rhs.w[2];
    const float* B1 = (const float*)rhs.b[0];
    const float* B2 = (const float*)rhs.b[1];
    const float* B3 = (const float*)rhs.b[2];
    const int td = rhs.s[1] != 0.0 ? 1 : 0;                   // time-dependent first layer: W1 is [d + 1, hd], row 0 for t
    const int nt = blockDim.x, tid = threadIdx.x;
    // [in][out] in memory -> zero-padded [out][in] in LDS.  Branch-free with eight loads in flight per thread: written as a
    // guarded element loop the compiler waited for every single load (128 round trips to L2 = ~0.1 ms per launch).
    auto stage = [&](float* dst, const float* W, int K, int O, int LW, int kreal, int oreal, int ld, int koff) {
      const int N = K * O;
      for (int base = 0; base < N; base += 8 * nt) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * nt + tid, k = i / O, o = i - k * O;
          const bool in = i < N && k < kreal && o < oreal;
          const float x = W[in ? (long long)(k + koff) * ld + o : 0];
          v[u] = in ? x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * nt + tid, k = i / O, o = i - k * O;
          if (i < N) dst[o * LW + k] = v[u];
        }
      }
    };
    stage(w + G::OFF_W1, W1, DP, HP, G::LW1, d, hd, hd, td);
    stage(w + G::OFF_W2, W2, HP, HP, G::LW2, hd, hd, hd, 0);
    stage(w + G::OFF_W3, W3, HP, DP, G::LW3, hd, d, d, 0);
    for (int i = tid; i < HP; i += nt) {
      w[G::OFF_B1 + i] = (B1 != nullptr && i < hd) ? B1[i] : 0.f;
      w[G::OFF_B2 + i] = (B2 != nullptr && i < hd) ? B2[i] : 0.f;
      w[G::OFF_WT + i] = (td && i < hd) ? W1[i] : 0.f;
    }
    for (int i = tid; i < DP; i += nt) w[G::OFF_B3 + i] = (B3 != nullptr && i < d) ? B3[i] : 0.f;
    // the tableau in the state dtype (rk_common.py:49-53 multiplies python floats into float32 tensors): read from LDS stage by
    // stage - converted in registers, the ~50 coefficients were hoisted out of every loop and spilled to scratch
    if (tid < 36) w[G::OFF_TAB + tid] = (float)A.beta[tid / 6][tid % 6];
    if (tid < 7) { w[G::OFF_TAB + G::TAB_E + tid] = (float)A.e[tid]; w[G::OFF_TAB + G::TAB_MID + tid] = (float)A.cmid[tid]; }
    __syncthreads();
  }
  __device__ __forceinline__ int col(int i) const { return 16 * (i >> 2) + 4 * g + (i & 3); }

  // ys[E] -> out[E] (same element order); ts: the time the network sees (already multiplied by the direction sign)
  __device__ __forceinline__ void eval(const float* ys, float* out, float ts) const {
    wt_f4 x[G::KB1], h1[G::MB1], h2[G::MB1], o[G::MB3];
    wt_lds l = wt_opaque(this->l);
#pragma unroll
    for (int kb = 0; kb < G::KB1; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[kb][r] = ys[4 * kb + r];
    wt_layer<G::MB1, G::KB1, G::LW1, ACT, true>(l + G::OFF_W1, l + G::OFF_B1, l + G::OFF_WT, ts, x, h1, m, g);
    wt_layer<G::MB1, G::MB1, G::LW2, ACT, false>(l + G::OFF_W2, l + G::OFF_B2, l, 0.f, h1, h2, m, g);
    wt_layer<G::MB3, G::MB1, G::LW3, -1, false>(l + G::OFF_W3, l + G::OFF_B3, l, 0.f, h2, o, m, g);
#pragma unroll
    for (int mb = 0; mb < G::MB3; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[4 * mb + r] = o[mb][r];
  }
};


// the stage combination / error / mid-point sums of mi_ode_step_fused.h with the float tableau read from LDS; `tab` is made
// opaque per use so that the loads stay inside the stage (same operation order: (hs * c) * k, summed left to right)
template <int SG>
__device__ __forceinline__ float wt_combine(float y0, const float* k, float hs, const float* row) {
  float acc = (hs * row[0]) * k[0];
#pragma unroll
  for (int j = 1; j < SG; ++j) acc = acc + (hs * row[j]) * k[j];
  return y0 + acc;
}
template <int S>
__device__ __forceinline__ void wt_finish(float y0, const float* k, float hs, const float* e, const float* cmid, float& err, float& ymid, bool need_mid) {
  float er = (hs * e[0]) * k[0];
#pragma unroll
  for (int j = 1; j <= S; ++j) er = er + (hs * e[j]) * k[j];
  err = er;
  ymid = y0;
  if (need_mid) {
    float ym = (hs * cmid[0]) * k[0];
#pragma unroll
    for (int j = 1; j <= S; ++j) ym = ym + (hs * cmid[j]) * k[j];
    ymid = y0 + ym;
  }
}

// One pass over this workgroup's tiles (cf. mlp_pass): every wave walks its own 16 rows, no workgroup synchronisation.
template <int DP, int HP, int ACT, int MODE, int S, bool TS>
__device__ __forceinline__ void wt_pass(const StepArgs& A, const StepPlanes<float, S>& P, void* copy_b, const WtCtx<DP, HP, ACT>& cx,
                                        Acc& acc, const double* t_out) {
  using G = WtGeom<DP, HP>;
  constexpr int E = DP / 4;
  const int d = cx.d;
  const float sign = cx.sign;
  const long long ntiles = (A.batch + G::R - 1) / G::R;
  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    const long long row = tile_i * G::R + 16 * cx.wave + cx.m;
    const bool rowok = row < A.batch;
    float hs = P.hs;
    asm volatile("" : "+v"(hs));
    float y0e[E], k[S + 1][E], ys[E], kn[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      // branch-free (clamped address + select): as guarded loads every element got its own branch and its own s_waitcnt - 32
      // serialized round trips per tile and pass
      const int col = cx.col(i);
      const bool oki = rowok && col < d;
      const long long off = oki ? row * d + col : 0;
      const float yv = stream_load<true>(P.y0 + off);
      y0e[i] = oki ? yv : 0.f;
      if constexpr (MODE != MLP_F0) {
        const float fv = stream_load<true>(P.f0 + off);
        k[0][i] = oki ? fv : 0.f;
      } else {
        k[0][i] = 0.f;
      }
    }
    if (MODE == MLP_F0) {
      if (copy_b != nullptr) {                                                       // solution[0] (after all loads were issued)
#pragma unroll
        for (int i = 0; i < E; ++i)
          if (rowok && cx.col(i) < d) ((float*)copy_b)[row * d + cx.col(i)] = y0e[i];
      }
      cx.eval(y0e, kn, sign * P.t0);
#pragma unroll
      for (int i = 0; i < E; ++i) {
        if (rowok && cx.col(i) < d) {
          const float f0 = sign * kn[i];
          P.f1[row * d + cx.col(i)] = f0;
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
      for (int i = 0; i < E; ++i) ys[i] = y0e[i] + hs * k[0][i];                     // misc.py:235
      cx.eval(ys, kn, sign * (P.t0 + hs));
#pragma unroll
      for (int i = 0; i < E; ++i) {
        if (rowok && cx.col(i) < d) {
          const float sc = (float)A.cp.atol + fabsf(y0e[i]) * (float)A.cp.rtol;
          const double q = (double)((sign * kn[i] - k[0][i]) / sc);                  // misc.py:237
          acc.suma += q * q;
        }
      }
      continue;
    }
    auto stage = [&](auto sg_c) {
      constexpr int SG = decltype(sg_c)::value;
      wt_lds brow = wt_opaque(cx.l + G::OFF_TAB + 6 * (SG - 1));
      float bc[SG];
#pragma unroll
      for (int j = 0; j < SG; ++j) bc[j] = brow[j];
#pragma unroll
      for (int i = 0; i < E; i += 2) {                                               // two elements per packed operation
        wt_f2 acc = {k[0][i], k[0][i + 1]};
        acc = (hs * bc[0]) * acc;
#pragma unroll
        for (int j = 1; j < SG; ++j) {
          const wt_f2 kj = {k[j][i], k[j][i + 1]};
          acc = acc + (hs * bc[j]) * kj;
        }
        ys[i] = y0e[i] + acc[0]; ys[i + 1] = y0e[i + 1] + acc[1];
      }
      cx.eval(ys, kn, sign * (P.t0 + (float)A.alpha[SG - 1] * hs));                  // rk_common.py:50, in the state dtype
#pragma unroll
      for (int i = 0; i < E; ++i) k[SG][i] = sign * kn[i];
    };
    for_stages<1, S>(stage);
    wt_lds tab = wt_opaque(cx.l + G::OFF_TAB);
    float ec[S + 1], mc[S + 1];
#pragma unroll
    for (int j = 0; j <= S; ++j) { ec[j] = tab[G::TAB_E + j]; mc[j] = tab[G::TAB_MID + j]; }
#pragma unroll
    for (int i = 0; i < E; ++i) {
      if (rowok && cx.col(i) < d) {
        float kk[S + 1];
#pragma unroll
        for (int j = 0; j <= S; ++j) kk[j] = k[j][i];
        float err, ymid;
        wt_finish<S>(y0e[i], kk, hs, ec, mc, err, ymid, !TS && P.j_hi > P.j_lo);
        const long long idx = row * d + cx.col(i);
        P.y1[idx] = ys[i];
        P.f1[idx] = k[S][i];
        step_emit<float, S, TS>(A, P, y0e[i], ys[i], kk, ymid, idx, t_out);
        acc.maxa = fmax(acc.maxa, (double)fabsf(y0e[i]));
        acc.maxb = fmax(acc.maxb, (double)fabsf(ys[i]));
        acc.suma += (double)err * (double)err;
      }
    }
  }
}

// The whole call in one launch on the wave-tile layout; control flow identical to k_persist_mlp.
template <int DP, int HP, int ACT, int S, bool TS>
__global__ __launch_bounds__((64 * WtGeom<DP, HP>::NW)) void k_persist_mlp_wt(PersistArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ PersistSharedT<256, 64> sh;                       // grid <= one workgroup per CU; more than 64 output times are read from memory
  Ctl& s_c = sh.c;
  WtCtx<DP, HP, ACT> cx;
  cx.init(A.s, smem_raw);
  CtrlParams cp = A.s.cp;
  cp.t_out = persist_stage_tout(A, sh.tout, 64);
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
    wt_pass<DP, HP, ACT, MLP_F0, S, TS>(A.s, P, A.out0, cx, acc, t_out);
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_F0, cp); }
    __syncthreads();
  }
  if (cp.auto_first_step && ok) {
    StepPlanes<float, S> P;
    P.y0 = y_user; P.f0 = fa; P.y1 = nullptr; P.f1 = nullptr; P.hs = (float)uniform_d(s_c.h0); P.t0 = (float)A.t0; P.j_lo = P.j_hi = 0;
    Acc acc;
    wt_pass<DP, HP, ACT, MLP_INITB, S, TS>(A.s, P, nullptr, cx, acc, t_out);
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_INITB, cp); }
  }
  auto publish = [&](const AttemptState& st) {
    sh.pub.dt = st.dt; sh.pub.t1 = st.t1; sh.pub.accepted = st.accepted; sh.pub.done = st.done;
    int j = st.next_out;
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
    wt_pass<DP, HP, ACT, MLP_STEP, S, TS>(A.s, P, nullptr, cx, acc, t_out);
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

  if (cur_y == y_user) {
    const long long n = A.s.batch * (long long)A.s.dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) ya[i] = y_user[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sh.st.store(s_c);
    s_c.idx_y0 = (cur_y == yb) ? 1 : 0; s_c.idx_y1 = (cur_y == yb) ? 0 : 1;
    s_c.idx_k[0] = (cur_f == fa) ? 2 : 2 + S; s_c.idx_k[S] = (cur_f == fa) ? 2 + S : 2;
    persist_write_back(A, s_c);
  }
}

}  // namespace mi
