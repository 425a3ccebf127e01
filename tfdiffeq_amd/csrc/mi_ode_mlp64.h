// The ODEFunc network dim -> hidden -> hidden -> dim (models/dense_odenet.py:41-92) in FLOAT64 on the matrix cores (round 6; the review
// of round 5, item 3: "fp64 ODEFunc at training-size batches has no kernel" - the reference's test problems are float64,
// tests/problems.py:10, and its ODEFunc takes any width).
//
// Same decomposition as the float32 tile kernels of mi_ode_mlp.h - a 32-row tile per workgroup, the three layers as chains of
// v_mfma_f64_16x16x4_f64, activations through LDS, the stage derivatives k_1 .. k_{S+1} of the tile in registers (accumulator layout
// of layer 3), the persistent grid / stamped-record hand-off / redundant controller of mi_ode_persist.h - with ONE structural
// difference: the weight slices are NOT resident in registers (64-128-128-64 in float64 is 264 KB: neither the register file at two
// wavefronts per SIMD nor LDS next to the activations holds it).  They are read from a PACKED copy in global memory at every use -
// [wave][k-pair][lane] double2, one coalesced 1-KB load per wavefront per two MFMA steps, hits in L1 / L2 (every tile of every
// workgroup reads the same 264 KB) - which the fp64 matrix pipe can afford: an fp64 16x16x4 MFMA occupies its SIMD for 64 cycles (the
// fp32 one 32), so a wavefront consumes 16 B of weights per 128 cycles per lane; the loads of the next steps are in flight under the
// current ones.  The pack (zero padded to DP x HP, the time row and the biases behind it) is refreshed by k_mlp64_pack before every
// launch (a few microseconds; weights change between calls of a training loop).
// Bound: fp64 matrix pipe (1024 flop per state element per evaluation at 64-128-128-64) + the fp64 activations on the vector ALU
// (which shares the SIMD's issue time with the matrix pipe on this part).
#pragma once
#include "mi_ode_mlp.h"

namespace mi {

typedef double mlp_d4 __attribute__((ext_vector_type(4)));
typedef double mlp_d2 __attribute__((ext_vector_type(2)));

template <int DP, int HP>
struct MlpGeom64 {
  static constexpr int CB3 = DP / 16;
  static constexpr int NW12 = HP / 16;                      // VIRTUAL wavefronts busy in layers 1, 2 (16 hidden columns each)
  static constexpr int NW3 = 2 * CB3;                       // virtual wavefronts that own state elements (2 row blocks x CB3)
  static constexpr int NV = NW12 > NW3 ? NW12 : NW3;        // virtual wavefronts of the decomposition
  // One wavefront per SIMD (V virtual wavefronts each): float64 needs twice the registers of the float32 kernel for the same tile - at two
  // wavefronts per SIMD (256 registers) the 64 x 128 geometry spilled 190 - 490 of them whatever was kept or re-read (measured); a
  // wavefront alone on its SIMD has all 512.  What the second wavefront bought the float32 kernel - the other's barrier bubbles and LDS
  // round trips filled - weighs half as much here: an fp64 MFMA occupies the pipe twice as long.
  static constexpr int V = NV > 4 ? NV / 4 : 1;             // virtual wavefronts per physical wavefront
  static constexpr int NW = NV / V;                         // physical wavefronts per workgroup
  static constexpr int R = 32;
  static constexpr int KS1 = DP / 4, KS2 = HP / 4;          // k steps per lane group in layers 1 / 2, 3
  static constexpr int LDX = DP + 2, LDH = HP + 2;          // LDS row strides in doubles (16-byte pad)
  // the pack: W1 [NW12][KS1][64] | W2 [NW12][KS2][64] | W3 [CB3][KS2][64] | wt [HP] | b1 [HP] | b2 [HP] | b3 [DP]
  static constexpr int OFF_W1 = 0;
  static constexpr int OFF_W2 = OFF_W1 + NW12 * KS1 * 64;
  static constexpr int OFF_W3 = OFF_W2 + NW12 * KS2 * 64;
  static constexpr int OFF_WT = OFF_W3 + CB3 * KS2 * 64;
  static constexpr int OFF_B1 = OFF_WT + HP, OFF_B2 = OFF_B1 + HP, OFF_B3 = OFF_B2 + HP;
  static constexpr int PACK = OFF_B3 + DP;                  // doubles
  static constexpr size_t lds_bytes() { return (size_t)R * (LDX + 2 * LDH) * sizeof(double) + 80 * sizeof(double); }
};

// weights of this call -> the pack (launched on the stream in front of every MLP kernel of the float64 family).
// Element (wave, s, lane) of a layer: k = lg * KS + s, column = 16 * wave + li, stored at [(wave * KS/2 + s/2) * 64 + lane] * 2 + (s & 1).
template <int DP, int HP>
__global__ __launch_bounds__(256) void k_mlp64_pack(RhsParams rhs, int d, double* pack) {
  using G = MlpGeom64<DP, HP>;
  const int hd = rhs.hidden;
  const int tid = (int)(blockIdx.x * blockDim.x + threadIdx.x), nt = (int)(gridDim.x * blockDim.x);   // (a grid of workgroups: one alone took ~0.2 ms)
  const double* W1 = (const double*)rhs.w[0];
  const double* W2 = (const double*)rhs.w[1];
  const double* W3 = (const double*)rhs.w[2];
  const double* B1 = (const double*)rhs.b[0];
  const double* B2 = (const double*)rhs.b[1];
  const double* B3 = (const double*)rhs.b[2];
  const int td = rhs.s[1] != 0.0 ? 1 : 0;                    // dense_odenet.py:79-84: fc1 sees concat([t, x]); row 0 multiplies t
  auto slot = [](int blk, int KS, int s, int lane) { return ((blk * (KS / 2) + (s >> 1)) * 64 + lane) * 2 + (s & 1); };
  for (int e = tid; e < G::NW12 * G::KS1 * 64; e += nt) {
    const int lane = e & 63, s = (e >> 6) % G::KS1, w = (e >> 6) / G::KS1;
    const int k = (lane >> 4) * G::KS1 + s, c = 16 * w + (lane & 15);
    pack[G::OFF_W1 + slot(w, G::KS1, s, lane)] = (k < d && c < hd) ? W1[(long long)(k + td) * hd + c] : 0.0;
  }
  for (int e = tid; e < G::NW12 * G::KS2 * 64; e += nt) {
    const int lane = e & 63, s = (e >> 6) % G::KS2, w = (e >> 6) / G::KS2;
    const int k = (lane >> 4) * G::KS2 + s, c = 16 * w + (lane & 15);
    pack[G::OFF_W2 + slot(w, G::KS2, s, lane)] = (k < hd && c < hd) ? W2[(long long)k * hd + c] : 0.0;
  }
  for (int e = tid; e < G::CB3 * G::KS2 * 64; e += nt) {
    const int lane = e & 63, s = (e >> 6) % G::KS2, cb = (e >> 6) / G::KS2;
    const int k = (lane >> 4) * G::KS2 + s, c = 16 * cb + (lane & 15);
    pack[G::OFF_W3 + slot(cb, G::KS2, s, lane)] = (k < hd && c < d) ? W3[(long long)k * d + c] : 0.0;
  }
  for (int c = tid; c < HP; c += nt) {
    pack[G::OFF_WT + c] = (td && c < hd) ? W1[c] : 0.0;
    pack[G::OFF_B1 + c] = (B1 != nullptr && c < hd) ? B1[c] : 0.0;
    pack[G::OFF_B2 + c] = (B2 != nullptr && c < hd) ? B2[c] : 0.0;
  }
  for (int c = tid; c < DP; c += nt) pack[G::OFF_B3 + c] = (B3 != nullptr && c < d) ? B3[c] : 0.0;
}

// float64 activations.  tanh = 1 - 2 / (e^{2x} + 1) with the odd Taylor polynomial below |x| = 2^-4 (where the difference cancels);
// relu keeps NaN; softplus = log1p(e^x) with the usual guard.  (Round 6, measured statically on the 64 x 128 kernel: a hand-rolled
// e^{2x} (rint / two-part ln 2 / degree-12 polynomial / ldexp) with v_rcp_f64 + two Newton steps instead of ocml's exp() and the
// correctly rounded division takes the kernel from 9.8 to 9.4 vector instructions per MFMA at 3e-15 instead of 1e-15 relative error
// - the activations are NOT most of the vector work here (accumulator <-> vector register moves of a kernel that uses all 512
// registers, the stage combinations and the address arithmetic are): not adopted.  Host replica: scripts/micro/tanh64_check.cpp.)
__device__ __forceinline__ double mlp64_tanh(double x) {
  const double e = exp(2.0 * x);                             // +inf / 0 at the ends give exactly +-1
  const double big = 1.0 - 2.0 / (e + 1.0);
  const double x2 = x * x;
  const double small = x * (1.0 + x2 * (-1.0 / 3.0 + x2 * (2.0 / 15.0 + x2 * (-17.0 / 315.0 + x2 * (62.0 / 2835.0 + x2 * (-1382.0 / 155925.0))))));
  return fabs(x) < 0.0625 ? small : big;
}
template <int ACT>
__device__ __forceinline__ double mlp64_act(double x) {
  if constexpr (ACT == MLP_ACT_TANH) return mlp64_tanh(x);
  else if constexpr (ACT == MLP_ACT_RELU) return x > 0.0 ? x : (x != x ? x : 0.0);
  else return x > 30.0 ? x : log1p(exp(x));
}

// One evaluation of the network for the tile whose input rows sit in s_x.  Every thread of the workgroup calls it.  A wavefront plays V
// virtual wavefronts (vw = wave + v * NW); as owner of state elements, virtual wavefront vw < NW3 holds rows 16 * rb + lg + 4 * i
// (the C / D layout of v_mfma_f64_16x16x4_f64), column 16 * cb + li, rb = vw / CB3, cb = vw % CB3.
// Weight slices travel in CHUNKS of CH k-pairs (32 registers), double buffered: while the chain of one chunk runs, the next chunk's loads
// are in flight - across column blocks, layers, barriers and evaluations (the first chunk of the next evaluation is requested under layer
// 3 and carried in `wn`).  (Measured with the loads at their uses: 84 us per evaluation of a tile at 128 workgroups - every workgroup
// waiting on the same L2 lines at every step.)
template <int N>
__device__ __forceinline__ void mlp64_load(const mlp_d2* wp, mlp_d2* w) {     // N coalesced 1-KB loads of this wavefront, issued together
#pragma unroll
  for (int s = 0; s < N; ++s) w[s] = wp[s * 64];
}
// N k-pairs of a chain.  The activation operands of the WHOLE chunk are read from LDS first (one wavefront per SIMD: nobody else covers
// an LDS round trip in front of an MFMA), then the MFMAs run back to back on two independent accumulators.
template <int N, bool TWO>
__device__ __forceinline__ void mlp64_chain(const double* a0p, const double* a1p, const mlp_d2* w, mlp_d4& c0, mlp_d4& c1) {
  mlp_d2 a0[N], a1[TWO ? N : 1];
#pragma unroll
  for (int m = 0; m < N; ++m) {
    a0[m] = *(const mlp_d2*)(a0p + 2 * m);
    if constexpr (TWO) a1[m] = *(const mlp_d2*)(a1p + 2 * m);
  }
  __builtin_amdgcn_sched_barrier(0);                         // (the scheduler otherwise sinks every read back in front of its MFMA)
#pragma unroll
  for (int m = 0; m < N; ++m) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[m][0], w[m][0], c0, 0, 0, 0);
    if constexpr (TWO) {
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[m][0], w[m][0], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[m][1], w[m][1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[m][1], w[m][1], c1, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[m][1], w[m][1], c0, 0, 0, 0);
    }
  }
}

template <int DP, int HP, int ACT>
struct MlpCtx64 {
  using G = MlpGeom64<DP, HP>;
  static constexpr int V = G::V, NW = G::NW;
  static constexpr int P1 = G::KS1 / 2, P2 = G::KS2 / 2;     // k-pairs of a layer-1 / layer-2, 3 slice
  static constexpr int CH = P2 >= 16 ? 8 : P2;               // chunk: k-pairs per buffer (layer 1's slice is one chunk: P1 <= CH)
  static constexpr int NC2 = P2 / CH;                        // chunks per slice in layers 2, 3
  static_assert(P1 <= CH && P2 % CH == 0, "chunking");
  // The weights of one evaluation as a STREAM of chunks: V of layer 1 (one per virtual wavefront), V * NC2 of layer 2, V * NC2 of
  // layer 3; chunk c + PD is requested while chunk c is consumed (PD = 2: two 8-KB requests per wavefront in flight - with one, the
  // kernel ran at 6 TB/s of L2 traffic over 256 workgroups, every chain waiting out most of a round trip), across column blocks,
  // layers, barriers and evaluations (the stream wraps: the first PD chunks of the next evaluation are carried in `wn`).
  static constexpr int NCH = V + 2 * V * NC2;
  static constexpr int PD = 2, NB = PD + 1;
  double *s_x, *s_h1, *s_h2;
  const double* pack;
  double sign;
  mlp_d2 wn[PD][CH];
  int lane, wave, li, lg, d;

  __device__ __forceinline__ int vw(int v) const { return wave + v * NW; }
  __device__ __forceinline__ int col(int v) const { return 16 * (vw(v) % G::CB3) + li; }
  __device__ __forceinline__ int row_of(int v, int i) const { return 16 * (vw(v) / G::CB3) + lg + 4 * i; }
  __device__ __forceinline__ bool owner(int v) const { return vw(v) < G::NW3 && col(v) < d; }
  // chunk c of the stream (c taken modulo NCH): layer, virtual wavefront, part of the slice
  static __device__ __forceinline__ constexpr int c_layer(int c) { return c < V ? 1 : (c < V + V * NC2 ? 2 : 3); }
  static __device__ __forceinline__ constexpr int c_v(int c) { return c < V ? c : ((c - V) % (V * NC2)) / NC2; }
  static __device__ __forceinline__ constexpr int c_h(int c) { return c < V ? 0 : ((c - V) % (V * NC2)) % NC2; }
  __device__ __forceinline__ void request(int c, mlp_d2* w) const {       // (guarded by the chunk's consumer: a wavefront idles where it owns nothing)
    c = c % NCH;
    const int L = c_layer(c), v = c_v(c), h = c_h(c);
    if (L == 1) { if (vw(v) < G::NW12) mlp64_load<P1>((const mlp_d2*)(pack + G::OFF_W1) + (vw(v) * P1) * 64 + lane, w); }
    else if (L == 2) { if (vw(v) < G::NW12) mlp64_load<CH>((const mlp_d2*)(pack + G::OFF_W2) + (vw(v) * P2 + h * CH) * 64 + lane, w); }
    else { if (vw(v) < G::NW3) mlp64_load<CH>((const mlp_d2*)(pack + G::OFF_W3) + ((vw(v) % G::CB3) * P2 + h * CH) * 64 + lane, w); }
  }

  __device__ __forceinline__ void init(const RhsParams& rhs, int dim, char* smem) {
    s_x = (double*)smem;
    s_h1 = s_x + G::R * G::LDX;
    s_h2 = s_h1 + G::R * G::LDH;
    lane = threadIdx.x & 63; wave = threadIdx.x >> 6; li = lane & 15; lg = lane >> 4;
    d = dim;
    pack = (const double*)rhs.w[0];                          // (the launcher put the pack here)
    sign = rhs.sign;
#pragma unroll
    for (int c = 0; c < PD; ++c) request(c, wn[c]);
  }
  __device__ __forceinline__ void put_x(const double (*v4)[4]) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (vw(v) < G::NW3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_x[row_of(v, i) * G::LDX + col(v)] = (col(v) < d) ? v4[v][i] : 0.0;
      }
    }
  }
  // ts: the time the network sees (already multiplied by the direction sign); it only shifts the first layer's bias
  __device__ __forceinline__ void eval(double (*out)[4], double ts) {
    mlp_d2 wb[NB][CH];
#pragma unroll
    for (int c = 0; c < PD; ++c) {
#pragma unroll
      for (int s = 0; s < CH; ++s) wb[c][s] = wn[c][s];
    }
    mlp_d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    __syncthreads();                                         // s_x is complete
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int L = c_layer(c), v = c_v(c), h = c_h(c);
      if (c == V || c == V + V * NC2) __syncthreads();       // layer 2 reads s_h1, layer 3 reads s_h2
      request(c + PD, wb[(c + PD) % NB]);
      const mlp_d2* w = wb[c % NB];
      if (L == 1) {                                          // [32 x DP] @ [DP x 16] per virtual wavefront
        if (vw(v) < G::NW12) {
          c0 = mlp_d4{0, 0, 0, 0}; c1 = mlp_d4{0, 0, 0, 0};
          mlp64_chain<P1, true>(s_x + li * G::LDX + lg * G::KS1, s_x + (16 + li) * G::LDX + lg * G::KS1, w, c0, c1);
          const int cc = 16 * vw(v) + li;
          const double b = pack[G::OFF_B1 + cc] + ts * pack[G::OFF_WT + cc];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s_h1[(lg + 4 * i) * G::LDH + cc] = mlp64_act<ACT>(c0[i] + b);
            s_h1[(16 + lg + 4 * i) * G::LDH + cc] = mlp64_act<ACT>(c1[i] + b);
          }
        }
      } else if (L == 2) {                                   // [32 x HP] @ [HP x 16]
        if (vw(v) < G::NW12) {
          if (h == 0) { c0 = mlp_d4{0, 0, 0, 0}; c1 = mlp_d4{0, 0, 0, 0}; }
          mlp64_chain<CH, true>(s_h1 + li * G::LDH + lg * G::KS2 + 2 * CH * h, s_h1 + (16 + li) * G::LDH + lg * G::KS2 + 2 * CH * h, w, c0, c1);
          if (h == NC2 - 1) {
            const int cc = 16 * vw(v) + li;
            const double b = pack[G::OFF_B2 + cc];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              s_h2[(lg + 4 * i) * G::LDH + cc] = mlp64_act<ACT>(c0[i] + b);
              s_h2[(16 + lg + 4 * i) * G::LDH + cc] = mlp64_act<ACT>(c1[i] + b);
            }
          }
        }
      } else {                                               // one 16-row block x 16 output columns per virtual wavefront
        if (vw(v) < G::NW3) {
          if (h == 0) c0 = mlp_d4{0, 0, 0, 0};
          const double* ap = s_h2 + (16 * (vw(v) / G::CB3) + li) * G::LDH + lg * G::KS2 + 2 * CH * h;
          mlp64_chain<CH, false>(ap, ap, w, c0, c1);
          if (h == NC2 - 1) {
            const double b = pack[G::OFF_B3 + col(v)];
#pragma unroll
            for (int i = 0; i < 4; ++i) out[v][i] = c0[i] + b;
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < PD; ++c) {                           // the next evaluation's first chunks, requested under layer 3
#pragma unroll
      for (int s = 0; s < CH; ++s) wn[c][s] = wb[(NCH + c) % NB][s];
    }
  }
};

template <int SG>
__device__ __forceinline__ double mlp64_combine(double y0, const double* k, double hs, const StepArgs& A) {
  double acc = (hs * A.beta[SG - 1][0]) * k[0];
#pragma unroll
  for (int j = 1; j < SG; ++j) acc = madd<(MI_MLP_FMA != 0), double>(hs * A.beta[SG - 1][j], k[j], acc);
  return y0 + acc;
}
template <int S>
__device__ __forceinline__ double mlp64_error(const double* k, double hs, const StepArgs& A) {
  double er = (hs * A.e[0]) * k[0];
#pragma unroll
  for (int j = 1; j <= S; ++j) er = madd<(MI_MLP_FMA != 0), double>(hs * A.e[j], k[j], er);
  return er;
}

template <int DP, int HP, int ACT, int MODE, int S, bool TS, bool SC0>
__device__ __forceinline__ void mlp_pass64(const StepArgs& A, const StepPlanes<double, S>& P, void* copy_a, void* copy_b,
                                           MlpCtx64<DP, HP, ACT>& cx, Acc& acc, const double* t_out) {
  using G = MlpGeom64<DP, HP>;
  constexpr int V = G::V;
  const int d = cx.d;
  const double sign = cx.sign;
  const long long ntiles = (A.batch + G::R - 1) / G::R;
  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    const long long row0 = tile_i * G::R;
    double hs = P.hs;
    asm volatile("" : "+v"(hs));                            // keep dt * coefficient products out of long-lived registers
    double y0e[V][4], k[S + 1][V][4], ys[V][4], kn[V][4];
    const long long left = A.batch - row0;
    const int nr = left < G::R ? (int)left : G::R;
    const long long tb = row0 * d;
    unsigned eo[V][4];
    bool ok[V][4];
#pragma unroll
    for (int v = 0; v < V; ++v) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ok[v][i] = cx.owner(v) && cx.row_of(v, i) < nr;
        eo[v][i] = (unsigned)(cx.row_of(v, i) * d + cx.col(v));
        y0e[v][i] = ok[v][i] ? stream_load<SC0>(P.y0 + tb + eo[v][i]) : 0.0;
        k[0][v][i] = (ok[v][i] && MODE != MLP_F0) ? stream_load<SC0>(P.f0 + tb + eo[v][i]) : 0.0;
        if (MODE == MLP_F0 && ok[v][i]) {
          if (copy_a != nullptr) ((double*)copy_a + tb)[eo[v][i]] = y0e[v][i];
          if (copy_b != nullptr) ((double*)copy_b + tb)[eo[v][i]] = y0e[v][i];
        }
      }
    }
    if (MODE == MLP_F0) {
      cx.put_x(y0e);
      cx.eval(kn, sign * P.t0);
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (ok[v][i]) {
            const double f0 = sign * kn[v][i];
            (P.f1 + tb)[eo[v][i]] = f0;
            const double sc = A.cp.atol + fabs(y0e[v][i]) * A.cp.rtol;               // misc.py:225
            const double q0 = y0e[v][i] / sc, q1 = f0 / sc;
            acc.suma += q0 * q0; acc.sumb += q1 * q1;
            if (!finite_(y0e[v][i])) acc.flag = 1;
          }
        }
      }
      continue;
    }
    if (MODE == MLP_INITB) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ys[v][i] = y0e[v][i] + hs * k[0][v][i];          // misc.py:235
      }
      cx.put_x(ys);
      cx.eval(kn, sign * (P.t0 + hs));
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (ok[v][i]) {
            const double sc = A.cp.atol + fabs(y0e[v][i]) * A.cp.rtol;
            const double q = (sign * kn[v][i] - k[0][v][i]) / sc;                    // misc.py:237
            acc.suma += q * q;
          }
        }
      }
      continue;
    }
    auto stage = [&](auto sg_c) {
      constexpr int SG = decltype(sg_c)::value;
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          double kk[SG];
#pragma unroll
          for (int j = 0; j < SG; ++j) kk[j] = k[j][v][i];
          ys[v][i] = mlp64_combine<SG>(y0e[v][i], kk, hs, A);
        }
      }
      cx.put_x(ys);
      cx.eval(kn, sign * (P.t0 + A.alpha[SG - 1] * hs));                             // rk_common.py:50
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) k[SG][v][i] = sign * kn[v][i];
      }
    };
    for_stages<1, S>(stage);
#pragma unroll
    for (int v = 0; v < V; ++v) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ok[v][i]) {
          double kk[S + 1];
#pragma unroll
          for (int j = 0; j <= S; ++j) kk[j] = k[j][v][i];
          double err, ymid;
          if (!TS && P.j_hi > P.j_lo) step_finish<double, S>(y0e[v][i], kk, hs, A, err, ymid, true);
          else ymid = y0e[v][i];
          err = mlp64_error<S>(kk, hs, A);
          const long long idx = tb + eo[v][i];
          (P.y1 + tb)[eo[v][i]] = ys[v][i];
          (P.f1 + tb)[eo[v][i]] = k[S][v][i];
          step_emit<double, S, TS>(A, P, y0e[v][i], ys[v][i], kk, ymid, idx, t_out);
          acc.maxa = fmax(acc.maxa, fabs(y0e[v][i]));
          acc.maxb = fmax(acc.maxb, fabs(ys[v][i]));
          acc.suma += err * err;
        }
      }
    }
  }
}

// Euler / RK4 (3/8 rule) on a fixed grid, one launch per call: the arithmetic of k_fixed_mlp, in float64.
template <int DP, int HP, int ACT>
__global__ __launch_bounds__((64 * MlpGeom64<DP, HP>::NW)) void k_fixed_mlp64(FixedArgs A) {
  using G = MlpGeom64<DP, HP>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  MlpCtx64<DP, HP, ACT> cx;
  cx.init(A.rhs, A.dim, smem_raw);
  const int d = cx.d;
  const double sign = cx.sign;
  const long long ntiles = (A.batch + G::R - 1) / G::R;
  const long long n = A.batch * d;
  const double* y0p = (const double*)A.y0;
  double* out = (double*)A.out;
  const double eps = A.eps;
  FixedClk clk;
  clk.begin();
  constexpr int V = G::V;
  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    double y[V][4];
    long long idx[V][4];
    bool ok[V][4];
#pragma unroll
    for (int v = 0; v < V; ++v) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long row = tile_i * G::R + cx.row_of(v, i);
        ok[v][i] = cx.owner(v) && row < A.batch;
        idx[v][i] = row * d + cx.col(v);
        y[v][i] = ok[v][i] ? y0p[idx[v][i]] : 0.0;
        if (ok[v][i]) out[idx[v][i]] = y[v][i];
      }
    }
    int j = 1;
    for (int s = 0; s < A.M; ++s) {
      const double t0 = A.grid[s], t1 = A.grid[s + 1];
      const double dt = t1 - t0;
      const double te = t0 + eps;                             // fixed_grid.py:7 / :42
      double k1[V][4], k2[V][4], k3[V][4], ys[V][4], yn[V][4], kn[V][4];
      cx.put_x(y);
      cx.eval(kn, sign * te);
#define MI64_ALL(stmt) _Pragma("unroll") for (int v = 0; v < V; ++v) { _Pragma("unroll") for (int i = 0; i < 4; ++i) { stmt; } }
      MI64_ALL(k1[v][i] = sign * kn[v][i])
      if (!A.rk4) {
        MI64_ALL(yn[v][i] = y[v][i] + dt * k1[v][i])
      } else {
        MI64_ALL(ys[v][i] = y[v][i] + dt * k1[v][i] / 3.0)                             // rk_common.py:77
        cx.put_x(ys);
        cx.eval(kn, sign * (te + dt / 3.0));
        MI64_ALL(k2[v][i] = sign * kn[v][i]; ys[v][i] = y[v][i] + dt * (k1[v][i] / -3.0 + k2[v][i]))
        cx.put_x(ys);
        cx.eval(kn, sign * (te + dt * 2.0 / 3.0));
        MI64_ALL(k3[v][i] = sign * kn[v][i]; ys[v][i] = y[v][i] + dt * (k1[v][i] - k2[v][i] + k3[v][i]))
        cx.put_x(ys);
        cx.eval(kn, sign * (te + dt));
        MI64_ALL(const double k4 = sign * kn[v][i]; yn[v][i] = y[v][i] + (k1[v][i] + 3.0 * k2[v][i] + 3.0 * k3[v][i] + k4) * (dt / 8.0))   // :81
      }
      while (j < A.T && t1 >= A.t[j]) {                      // solvers.py:97-100, _linear_interp :106-115
        const double tj = A.t[j];
        MI64_ALL(if (ok[v][i]) out[(long long)j * n + idx[v][i]] = (tj == t0) ? y[v][i] : ((tj == t1) ? yn[v][i] : y[v][i] + ((yn[v][i] - y[v][i]) / (t1 - t0)) * (tj - t0)))
        ++j;
      }
      MI64_ALL(y[v][i] = yn[v][i])
#undef MI64_ALL
    }
  }
  clk.end(A.clk);
}

template <int DP, int HP, int ACT, int MODE, int S, bool TS>
__global__ __launch_bounds__((64 * MlpGeom64<DP, HP>::NW)) void k_mlp64(MlpArgs M) {
  using G = MlpGeom64<DP, HP>;
  const StepArgs& A = M.step;
  StepPlanes<double, S> P;
  if (MODE == MLP_F0) {
    P.y0 = (const double*)M.x_y0;
    P.f0 = nullptr; P.y1 = nullptr; P.hs = 0.0; P.t0 = A.ctl->t1; P.j_lo = P.j_hi = 0;
    P.f1 = (double*)(A.planes + 2 * A.stride);
  } else {
    if (!resolve_step<double, S>(A, P)) return;
    if (MODE == MLP_INITB) P.hs = A.ctl->h0;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  MlpCtx64<DP, HP, ACT> cx;
  cx.init(A.rhs, A.dim, smem_raw);
  double* red = cx.s_h2 + G::R * G::LDH;
  Acc acc;
  mlp_pass64<DP, HP, ACT, MODE, S, TS, false>(A, P, M.copy_a, M.copy_b, cx, acc, A.t_out);
  if constexpr (MODE == MLP_STEP) finish_attempt(A, acc, red);
  else block_reduce_store(acc, red, A.partials + (long long)blockIdx.x * kRec);
}

// The whole call in one launch: k_persist_mlp in float64.
template <int DP, int HP, int ACT, int S, bool TS>
__global__ __launch_bounds__((64 * MlpGeom64<DP, HP>::NW)) void k_persist_mlp64(PersistArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ PersistShared sh;
  Ctl& s_c = sh.c;
  MlpCtx64<DP, HP, ACT> cx;
  cx.init(A.s.rhs, A.s.dim, smem_raw);
  CtrlParams cp = A.s.cp;
  cp.t_out = persist_stage_tout(A, sh.tout);
  const double* t_out = cp.t_out;
  unsigned gen = 0;
  double r[5], rec[kRec], n_tot = 0.0;
  if (threadIdx.x == 0) { persist_init_ctl(s_c, A); sh.ok = 1; }
  __syncthreads();

  double* const ya = (double*)(A.s.planes);
  double* const yb = (double*)(A.s.planes + A.s.stride);
  double* const fa = (double*)(A.s.planes + 2 * A.s.stride);
  double* const fb = (double*)(A.s.planes + (long long)(2 + S) * A.s.stride);
  const double* const y_user = (const double*)A.y0;

  bool ok;
  {
    StepPlanes<double, S> P;
    P.y0 = y_user; P.f0 = nullptr; P.y1 = nullptr; P.f1 = fa; P.hs = 0.0; P.t0 = A.t0; P.j_lo = P.j_hi = 0;
    Acc acc;
    mlp_pass64<DP, HP, ACT, MLP_F0, S, TS, true>(A.s, P, nullptr, A.out0, cx, acc, t_out);
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_F0, cp); }
    __syncthreads();
  }
  if (cp.auto_first_step && ok) {
    StepPlanes<double, S> P;
    P.y0 = y_user; P.f0 = fa; P.y1 = nullptr; P.f1 = nullptr; P.hs = uniform_d(s_c.h0); P.t0 = A.t0; P.j_lo = P.j_hi = 0;
    Acc acc;
    mlp_pass64<DP, HP, ACT, MLP_INITB, S, TS, true>(A.s, P, nullptr, nullptr, cx, acc, t_out);
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

  const double* cur_y = y_user;
  double* cur_f = fa;
  while (!uniform_i(sh.pub.done)) {
    StepPlanes<double, S> P;
    const double dt_u = uniform_d(sh.pub.dt), t1_u = uniform_d(sh.pub.t1);
    P.y0 = cur_y; P.f0 = cur_f;
    P.y1 = (cur_y == ya) ? yb : ya;
    P.f1 = (cur_f == fa) ? fb : fa;
    P.hs = dt_u; P.t0 = t1_u;
    P.t_start = t1_u; P.dt64 = dt_u; P.t_new = t1_u + dt_u;
    P.j_lo = uniform_i(sh.pub.emit_lo); P.j_hi = uniform_i(sh.pub.emit_hi);
    Acc acc;
    mlp_pass64<DP, HP, ACT, MLP_STEP, S, TS, true>(A.s, P, nullptr, nullptr, cx, acc, t_out);
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
