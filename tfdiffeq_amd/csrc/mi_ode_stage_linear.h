// Fused RK stage kernels for f(t, y) = g(y) @ W (+ b), g = identity (MI_ODE_RHS_LINEAR, config 4)
// or g = cube (MI_ODE_RHS_CUBIC_LINEAR) on a [batch, dim] row-major state.
//
// The stage is HBM-bound by construction (reads y0 and NK stage planes, writes one plane; the
// contraction re-uses W, which is dim*dim elements and lives in registers / L2):
//   algorithmic traffic = (NK + 2) planes (+1 when y1 is written)        SURVEY.md 8(d)
//   arithmetic          = 2*dim flop per element                         (256 flop/elt at dim 128)
// so the contraction has to run at >= ~45 % of the fp64 peak UNDER the memory stream; that is
// what the MFMA tile kernel below is for.  The VALU kernel is the any-dim fallback.
#pragma once
#include <type_traits>
#include "mi_ode_dev.h"

#ifndef MI_PF_MAX
#define MI_PF_MAX 6      // planes prefetched during the MFMA phase (stages with more planes: see MI_PF_LAST)
#endif
#ifndef MI_PF_LAST
#define MI_PF_LAST 4     // ... for the last stage (7 planes would spill)
#endif

namespace mi {

// ------------------------------------------------------------------------------------------------
// (1) VALU fallback: any dim <= 256.  Thread (r, c) owns element (row r of the tile, column c).
// ------------------------------------------------------------------------------------------------
template <typename T, int NK, int MODE>
__global__ __launch_bounds__(256) void k_stage_linear_valu(StageArgs A, int dim_p2) {
  Resolved<T> R;
  if (!resolve<T, NK, MODE>(A, R)) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* tile = (T*)smem_raw;                                  // [rows_per_block][D]
  const int D = A.dim;
  const int rpb = 256 / dim_p2;
  const int r = threadIdx.x / dim_p2, c = threadIdx.x % dim_p2;
  const T* W = (const T*)A.rhs.w[0];
  const T* bias = (const T*)A.rhs.b[0];
  const T sign = (T)A.rhs.sign;
  const bool cube = A.rhs.cube != 0;
  Acc acc;
  const long long ntiles = (A.batch + rpb - 1) / rpb;
  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    const long long row = tile_i * rpb + r;
    const bool active = (c < D) && (row < A.batch);
    const long long idx = row * D + c;
    T y0 = (T)0, ys = (T)0, aux = (T)0, kk[NK > 0 ? NK : 1];
    kk[0] = (T)0;
    if (active) {
      y0 = R.y0[idx];
#pragma unroll
      for (int j = 0; j < NK; ++j) kk[j] = R.k[j][idx];
      ys = combine_elem<T, NK, MODE>(y0, kk, R.hs, A, aux);
      reduce_flat<T, MODE>(y0, ys, A, acc);
      if constexpr (MODE == M_F0) {
        if (A.copy_a != nullptr) ((T*)A.copy_a)[idx] = y0;
        if (A.copy_b != nullptr) ((T*)A.copy_b)[idx] = y0;
      }
      tile[r * D + c] = cube ? ys * ys * ys : ys;
    }
    __syncthreads();
    if (active) {
      T kn = (T)0;
      const T* trow = tile + r * D;
      for (int k = 0; k < D; ++k) kn = fma(trow[k], W[(long long)k * D + c], kn);
      if (bias != nullptr) kn = kn + bias[c];
      kn = sign * kn;
      if (R.k_out != nullptr) R.k_out[idx] = kn;
      const T v = epilogue_elem<T, NK, MODE>(y0, kk[0], kn, aux, R.hs, A, acc);
      if constexpr (mode_writes_y1(MODE)) R.y1[idx] = (MODE == M_LAST_FSAL) ? ys : v;
    }
    __syncthreads();
  }
  if constexpr (mode_has_reduction(MODE)) {
    __shared__ double red[80];
    block_reduce_store(acc, red, A.partials + (long long)blockIdx.x * kRec);
  }
}

// ------------------------------------------------------------------------------------------------
// (2) MFMA tile kernel: dim D in {16, 32, 64, 128}, fp64 (v_mfma_f64_16x16x4_f64) or fp32
//     (v_mfma_f32_16x16x4_f32).  Workgroup = D/16 wavefronts; wave w owns output columns
//     [16w, 16w+16) and keeps its 16-column slice of W in registers for the whole kernel
//     (D/4 values per lane).  Per tile of R = 32 rows:
//       flat phase : every thread streams 16-byte chunks of y0 / k_j (fully coalesced), combines
//                    them into ys (registers) and parks ys (and the partial error sum) in LDS;
//       MFMA phase : each wave multiplies the [32 x D] LDS tile by its W slice: lane (i, g) feeds
//                    A-operand ys[i][g*D/4 + s] and B-operand W[g*D/4 + s][16w + i] at step s
//                    (any k-permutation is legal as long as A and B agree - this one makes every
//                    lane read a CONTIGUOUS run of ys, i.e. ds_read_b128, conflict-free with the
//                    16-byte row pad);
//       epilogue   : k_new leaves from the accumulator layout in 128-byte row segments.
// ------------------------------------------------------------------------------------------------
template <typename T> struct MfmaTraits;
template <> struct MfmaTraits<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static constexpr int VEC = 2;   // elements per 16-byte chunk
  __device__ static __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  __device__ static __forceinline__ int acc_row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct MfmaTraits<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static constexpr int VEC = 4;
  __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  __device__ static __forceinline__ int acc_row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

template <typename T, int V> struct alignas(16) Chunk { T v[V]; };

template <typename T, int D, int NK, int MODE, bool CUBE>
__global__ __launch_bounds__(D * 4) void k_stage_linear_mfma(StageArgs A) {
  using TR = MfmaTraits<T>;
  using acc_t = typename TR::acc_t;
  constexpr int VEC = TR::VEC;
  constexpr int NT = D * 4;                 // threads = 64 * (D / 16)
  constexpr int R_ = 32;                    // rows per tile
  constexpr int CPT = R_ * D / VEC / NT;    // 16-byte chunks per thread per plane (4 fp64, 2 fp32)
  constexpr int LD = D + VEC;               // LDS row stride (16-byte pad)
  constexpr int KS = D / 4;                 // MFMA steps; lane group g covers k in [g*KS, (g+1)*KS)
  constexpr bool NEED_AUX = (MODE == M_LAST_FSAL || MODE == M_FX_RK4_4);
  using CH = Chunk<T, VEC>;
  static_assert(CPT >= 1, "tile too small");

  Resolved<T> R;
  if (!resolve<T, NK, MODE>(A, R)) return;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* s_ys = (T*)smem_raw;                                   // [R_][LD]
  T* s_aux = s_ys + R_ * LD;                                // [R_][LD] (only when NEED_AUX)
  double* red = (double*)(s_ys + (NEED_AUX ? 2 : 1) * R_ * LD);   // 80 doubles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const T* W = (const T*)A.rhs.w[0];
  const T* bias = (const T*)A.rhs.b[0];
  const T sign = (T)A.rhs.sign;

  // this wave's slice of W, resident for the whole kernel
  T bf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) bf[s] = W[(long long)(lg * KS + s) * D + 16 * wave + li];
  const T bias_v = bias != nullptr ? bias[16 * wave + li] : (T)0;

  Acc acc;
  const long long ntiles = (A.batch + R_ - 1) / R_;

  // Registers of the tile in flight.  Plane p = 0 is y0, plane p = j + 1 is k_j.  The loads of the first PF
  // planes of tile i+1 are issued right after tile i has been handed to LDS, so they travel while the matrix
  // pipe works on tile i: with one 8-wave workgroup per CU this is what overlaps the HBM stream with the MFMA
  // phase (without it a stage costs T_mem + T_mfma).  The MFMA phase lasts about as long as ~3 planes take to
  // stream, so PF is capped where the register file would spill (last stage: 7 planes x 16 VGPRs).
  constexpr int NP = NK + 1;
  constexpr int PF = (NP <= MI_PF_MAX) ? NP : ((MODE == M_LAST_FSAL) ? MI_PF_LAST : MI_PF_MAX);
  CH pl[NP][CPT];
  auto load_planes = [&](long long t_i, auto p_begin, auto p_end) {
    constexpr int P0 = decltype(p_begin)::value, P1 = decltype(p_end)::value;
    const long long base = t_i * R_ * D;
    const long long left = (A.batch - t_i * R_) * D;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int e0 = (c * NT + tid) * VEC;
      const bool ok = e0 < left;                             // D % VEC == 0: chunks never straddle the end
#pragma unroll
      for (int p = P0; p < P1; ++p) {
        if (ok) {
          pl[p][c] = *(const CH*)((p == 0 ? R.y0 : R.k[p > 0 ? p - 1 : 0]) + base + e0);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) pl[p][c].v[v] = (T)0;
        }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using IPF = std::integral_constant<int, PF>;
  using INP = std::integral_constant<int, NP>;
  if ((long long)blockIdx.x < ntiles) load_planes(blockIdx.x, I0{}, IPF{});

  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    const long long row0 = tile_i * R_;
    const long long tile_base = row0 * D;                    // tile rows are contiguous in memory
    const long long elems_left = (A.batch - row0) * D;       // valid elements from tile_base on
    if constexpr (PF < NP) load_planes(tile_i, IPF{}, INP{});   // the planes that were not prefetched
    const T hs = R.hs;
    if (tile_i != (long long)blockIdx.x) __syncthreads();    // previous tile's LDS reads are done

    // ---- flat phase: combine the landed planes, park ys (and the partial error sum) in LDS ------
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int e0 = (c * NT + tid) * VEC;                   // element offset inside the tile
      const bool ok = e0 < elems_left;
      CH ysc, auxc;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        T kk[NK > 0 ? NK : 1];
#pragma unroll
        for (int j = 0; j < NK; ++j) kk[j] = pl[j + 1][c].v[v];
        T aux;
        const T ys = combine_elem<T, NK, MODE, (MI_LIN_FMA != 0)>(pl[0][c].v[v], kk, hs, A, aux);
        if (ok) reduce_flat<T, MODE>(pl[0][c].v[v], ys, A, acc);
        ysc.v[v] = ys;
        auxc.v[v] = aux;
      }
      if constexpr (MODE == M_LAST_FSAL) {
        if (ok) *(CH*)(R.y1 + tile_base + e0) = ysc;         // y1 = y_last (rk_common.py:58)
      }
      if constexpr (MODE == M_F0) {                          // ys == y0: seed the state plane and solution[0]
        if (ok && A.copy_a != nullptr) *(CH*)((T*)A.copy_a + tile_base + e0) = ysc;
        if (ok && A.copy_b != nullptr) *(CH*)((T*)A.copy_b + tile_base + e0) = ysc;
      }
      const int rr = e0 / D, cc = e0 % D;
      if constexpr (CUBE) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) ysc.v[v] = ysc.v[v] * ysc.v[v] * ysc.v[v];
      }
      *(CH*)(s_ys + rr * LD + cc) = ysc;
      if constexpr (NEED_AUX) *(CH*)(s_aux + rr * LD + cc) = auxc;
    }
    __syncthreads();
    if (tile_i + gridDim.x < ntiles) load_planes(tile_i + gridDim.x, I0{}, IPF{});   // prefetch: in flight during the MFMA phase

    // ---- MFMA phase: two 16-row blocks per wave, interleaved for issue-level parallelism ------
    acc_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const T* a0p = s_ys + li * LD + lg * KS;
    const T* a1p = s_ys + (16 + li) * LD + lg * KS;
#pragma unroll
    for (int m = 0; m < KS / VEC; ++m) {
      const CH a0 = *(const CH*)(a0p + m * VEC);
      const CH a1 = *(const CH*)(a1p + m * VEC);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        c0 = TR::mfma(a0.v[v], bf[m * VEC + v], c0);
        c1 = TR::mfma(a1.v[v], bf[m * VEC + v], c1);
      }
    }

    // ---- epilogue in the accumulator layout ------------------------------------------------
    const int col = 16 * wave + li;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = rb * 16 + TR::acc_row(lane, i);
        const long long row = row0 + rr;
        if (row < A.batch) {
          T kn = (rb == 0 ? c0[i] : c1[i]);
          if (bias != nullptr) kn = kn + bias_v;
          kn = sign * kn;
          const long long idx = row * D + col;
          if (R.k_out != nullptr) R.k_out[idx] = kn;
          T y0v = (T)0, k0v = (T)0, auxv = (T)0;
          if constexpr (mode_needs_y0_epi(MODE)) y0v = R.y0[idx];
          if constexpr (MODE == M_INITB) k0v = R.k[0][idx];
          if constexpr (NEED_AUX) auxv = s_aux[rr * LD + col];
          const T v = epilogue_elem<T, NK, MODE, (MI_LIN_FMA != 0)>(y0v, k0v, kn, auxv, hs, A, acc);
          if constexpr (MODE == M_FX_EULER || MODE == M_FX_RK4_4) R.y1[idx] = v;
        }
      }
    }
  }
  if constexpr (mode_has_reduction(MODE)) {
    block_reduce_store(acc, red, A.partials + (long long)blockIdx.x * kRec);
  }
}

template <typename T, int D, int MODE>
constexpr size_t linear_mfma_lds_bytes() {
  constexpr int VEC = MfmaTraits<T>::VEC;
  constexpr bool NEED_AUX = (MODE == M_LAST_FSAL || MODE == M_FX_RK4_4);
  return (size_t)(NEED_AUX ? 2 : 1) * 32 * (D + VEC) * sizeof(T) + 80 * sizeof(double);
}

}  // namespace mi
