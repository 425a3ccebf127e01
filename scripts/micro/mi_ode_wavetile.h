// "Wave-tile" kernels for the linear RHS f = y @ W (+ b): ONE WAVEFRONT OWNS ONE 16-ROW TILE FOR THE WHOLE ATTEMPT.
//
// The older tile kernels (mi_ode_step_fused.h, LinCtx) spread W over the 8 waves of a workgroup (16 output columns
// each, resident in VGPRs); every RHS evaluation then has to publish the stage state y_sigma to the other waves
// through LDS behind two workgroup barriers, which phase-locks the two waves of every SIMD (both combine, both
// multiply) and leaves the matrix pipe idle a third of the time (profiles/r01_whole_mfma_pmc.jsonl).
//
// Here the product is computed TRANSPOSED, k^T = W^T y^T, so that the state is the B operand:
//   v_mfma_f64_16x16x4_f64  D[m][n] += sum_kk A[m][kk] B[kk][n],  lane l = (r = l & 15, g = l >> 4) supplies A[m = r][kk = g]
//   and B[kk = g][n = r] and receives D[m = g + 4 reg][n = r] (f32: m = 4 g + reg).
// With n = trajectory (row r of the tile) and the D columns of the state dealt to the four lane groups in contiguous
// runs of NE = D/4 (lane (r, g) owns y[r][g NE .. g NE + NE)), step s of an evaluation takes B = the lane's OWN element
// s, and the 16 rows of output block b can be NAMED so that lane (r, g) receives exactly the columns it owns
// (block-row m <-> column (m & 3) NE + 4 b + (m >> 2)).  Consequences:
//   * the state never changes hands: no LDS exchange, no barrier, no transposition between stages - a wave runs all
//     S evaluations of an attempt (and the before_integrate passes) on registers it alone owns;
//   * W is the A operand, shared by every wave: it is staged ONCE per launch into LDS in operand order (D*D elements
//     = 128 KiB at D = 128 fp64; gfx950 has 160 KiB) and streamed with ds_read_b128 (two or four operands per read,
//     lane-contiguous, conflict-free), 32 B/clk/CU at the full fp64 MFMA rate (LDS gives 256); the reversed-time sign
//     (misc.py:318-321) is folded into the staged copy (negation is exact, so sign*(y@W + b) keeps its bits);
//   * an evaluation is NB = D/16 INDEPENDENT accumulator chains (one per output block), so MFMAs issue back to back
//     without the dependent-accumulator bubble of a single chain;
//   * k-grouping and accumulation order are those of the older kernels (step s multiplies columns {kk NE + s}), and
//     a*b = b*a: results are bit-identical to them.
//
// THE STAGE ARITHMETIC STREAMS THROUGH THE MFMA LOOP.  k-step s of evaluation J needs exactly ONE element of y_J as
// its B operand - element s.  It is formed just in time, inside the k-loop, from element s of the previous
// evaluation's accumulators (k_J), and every other use of k_J[s] (the running sums of the later stages, the error
// estimate) happens in the same step, in the shadow of the step's NB MFMAs (512 matrix-pipe cycles at D = 128 fp64
// against <= ~45 VALU operations).  There is no separate combine phase, no y_sigma plane, and k_J[s] is dead after
// step s.  Only the very first element of an evaluation waits for the previous evaluation to drain.
//
// Register budget (D = 128 fp64: one "plane" = 32 elements = 64 registers per lane; a wave owns the whole 512-entry
// file: one wave per SIMD, 256-thread workgroups).  Keeping k_1..k_7 would need 9 planes; instead k_1..k_{PS-1} are
// kept and from k_PS on the stage sums are carried as RUNNING ACCUMULATORS (evaluation PS converts element by element:
// 4 values die, 3 are born).  misc._scaled_dot_product adds its terms left to right (misc.py:118-121), so a running
// sum performs the same operations in the same order - same bits.  Peak: 6 planes (y0, three k / sums, the previous
// and the current accumulators).  The dt*coefficient products are formed once per attempt by the first lanes of the
// workgroup and read back from LDS (broadcast reads) - the same IEEE product each lane would form itself.
#pragma once
#include "mi_ode_step_fused.h"

namespace mi {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <typename T, int D>
struct Wt {
  using TR = MfmaTraits<T>;
  using acc_t = typename TR::acc_t;
  using CH = Chunk<T, TR::VEC>;
  static constexpr int VEC = TR::VEC;            // elements per 16 bytes
  static constexpr int NE = D / 4;               // elements of a tile row owned by one lane
  static constexpr int NB = D / 16;              // 16-column output blocks = independent accumulator chains
  static constexpr int GB = NB < VEC ? NB : VEC; // blocks served by one 16-byte LDS read ...
  static constexpr int GS = VEC / GB;            // ... times k-steps served by it
  static constexpr int NBG = NB / GB, NSG = NE / GS;
  static constexpr int kWaves = 4;               // wavefronts per workgroup (one per SIMD)
  static constexpr int kThreads = 64 * kWaves;
  static constexpr size_t kWBytes = (size_t)D * D * sizeof(T);
  static constexpr size_t kLdsBytes = kWBytes + (size_t)D * sizeof(T);   // W, bias
  static constexpr bool kF64 = std::is_same<T, double>::value;

  // state column received in row m of output block b (see the header comment)
  __host__ __device__ static constexpr int outcol(int b, int m) {
    return kF64 ? (m & 3) * NE + 4 * b + (m >> 2) : (m >> 2) * NE + 4 * b + (m & 3);
  }

  // sign * W ([d, d] row-major, f = y @ W) -> LDS in operand order, zero padded to D: element (sg, bg, lane, vs, vb) is
  // the A operand of lane `lane` for k-step s = sg GS + vs and output block b = bg GB + vb.
  __device__ static __forceinline__ void stage_w(T* Wl, const T* W, int d, T sign) {
    for (int idx = threadIdx.x; idx < D * D; idx += blockDim.x) {
      const int v = idx % VEC, lane = (idx / VEC) % 64, grp = idx / (VEC * 64);
      const int bg = grp % NBG, sg = grp / NBG;
      const int s = sg * GS + v / GB, b = bg * GB + v % GB;
      const int in = (lane >> 4) * NE + s, out = outcol(b, lane & 15);
      Wl[idx] = (in < d && out < d) ? sign * W[(long long)in * d + out] : (T)0;
    }
  }
};

// ---- tile I/O: 16 bytes per lane per instruction, NE*sizeof(T) contiguous bytes per lane per plane.  A wave-uniform base
// (SGPR pair: plane + first row of the tile) plus one 32-bit per-lane byte offset that is the same for every plane of the
// tile (global_load_dwordx4 v, v_off, s[base] offset:imm): no address registers per plane, no buffer descriptors (four
// SGPRs each - the stage products already fill the scalar file).  Rows past the end of the batch: the lane's row is
// clamped to the last valid one for loads (finite duplicates that nobody accumulates or stores).  Planes of a launch
// that are rewritten and re-read are always re-read by the wave that wrote them (fixed tile -> wave map): program order
// through the CU's own L1 is enough, no cache-policy bits needed.
template <typename T>
struct TileIo {
  const char* base;
  __device__ __forceinline__ void open(const T* plane, long long row0, long long /*batch*/, int d) { base = (const char*)(plane + row0 * d); }
  template <bool SC0>
  __device__ __forceinline__ Chunk<T, 16 / sizeof(T)> load(int voff) const { return *(const Chunk<T, 16 / sizeof(T)>*)(base + (unsigned)voff); }
};

// Stores go through a buffer descriptor instead (hardware bounds check: the rows past the end of the batch are dropped
// without a branch - a branch would split the straight-line k-step stream); a descriptor lives only while a tile's
// outputs leave (last evaluation + tile epilogue).
template <typename T>
struct TileOut {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ void open(T* plane, long long row0, long long batch, int d) {
    long long rows = batch - row0;
    rows = rows < 0 ? 0 : (rows > 16 ? 16 : rows);
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(plane + row0 * d), (short)0, (int)(rows * d * (long long)sizeof(T)), 0x00020000);
  }
  // voff: the UNCLAMPED byte offset of the lane inside the tile
  __device__ __forceinline__ void store(int voff, const Chunk<T, 16 / sizeof(T)>& c) const {
    u32x4_t raw;
    __builtin_memcpy(&raw, &c, 16);
    __builtin_amdgcn_raw_buffer_store_b128(raw, rsrc, voff, 0, 0);
  }
};

template <typename T, int NE, bool SC0>
__device__ __forceinline__ void tile_load(const TileIo<T>& io, int voff, T (&v)[NE]) {
  constexpr int VEC = 16 / sizeof(T);
#pragma unroll
  for (int q = 0; q < NE / VEC; ++q) {
    const Chunk<T, VEC> c = io.template load<SC0>(voff + q * 16);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[q * VEC + e] = c.v[e];
  }
}

template <typename T, int NE>
__device__ __forceinline__ void tile_store(const TileOut<T>& io, int voff, const T (&v)[NE]) {
  constexpr int VEC = 16 / sizeof(T);
#pragma unroll
  for (int q = 0; q < NE / VEC; ++q) {
    Chunk<T, VEC> c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) c.v[e] = v[q * VEC + e];
    io.store(voff + q * 16, c);
  }
}

// per-wave context
template <typename T, int D, bool BIAS>
struct WtCtx {
  using G = Wt<T, D>;
  using acc_t = typename G::acc_t;
  using CH = typename G::CH;
  const T* Wl;                 // LDS: sign * W in operand order
  const T* bias_lds;           // LDS: sign * bias (padded to D)
  int lane, r, g;
  int wave;                    // wave-uniform (SGPR)
  int lane_off;                // element offset of this lane's first A operand
  int dd;                      // row length (elements)
#ifndef MI_WT_FILL
#define MI_WT_FILL 10
#endif
#ifndef MI_WT_LOOK
#define MI_WT_LOOK 8
#endif
  static constexpr int kFill = MI_WT_FILL;   // VALU instructions dealt out behind each MFMA of a step (as many as there are)

  // `lds` = G::kLdsBytes of dynamic LDS.  Ends with a workgroup barrier.
  __device__ __forceinline__ void init(const RhsParams& rhs, char* lds, int d) {
    lane = threadIdx.x & 63; r = lane & 15; g = lane >> 4;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    T* W_ = (T*)lds;
    const T sign = (T)rhs.sign;
    G::stage_w(W_, (const T*)rhs.w[0], d, sign);
    Wl = W_;
    T* bl = (T*)(lds + G::kWBytes);
    if constexpr (BIAS)
      for (int c = threadIdx.x; c < D; c += blockDim.x) bl[c] = c < d ? sign * ((const T*)rhs.b[0])[c] : (T)0;
    bias_lds = bl;
    lane_off = lane * G::VEC;
    dd = d;
    __syncthreads();
  }

  // One evaluation k = sign * (y @ W (+ b)) for the wave's tile, streamed and software-pipelined: f(s) returns element s
  // of y (the B operand of k-step s) and may do whatever else belongs to step s.  While the NB MFMAs of step s occupy the
  // matrix pipe (NB x 64 cycles at fp64), the wave issues f(s + 1) and the A-operand reads of step s + 1 BETWEEN them:
  // an in-order wave can only hide work behind an MFMA that has already been issued, so the fillers are dealt out
  // one group per MFMA (sched_group_barrier), not clumped in front of the step.
  // C receives the NB accumulators (element e of k = C[e / 4][e % 4], bias NOT yet added: use k_of()).
  // byte offset of the lane's first element inside the tile that starts at row0 (rows past the batch: clamped)
  __device__ __forceinline__ int voff_of(long long row0, long long batch) const {
    const long long left = batch - row0;
    const int rr = (left >= 16 || r < (int)left) ? r : (int)left - 1;
    return (rr * dd + g * G::NE) * (int)sizeof(T);
  }
  __device__ __forceinline__ int voff_raw() const { return (r * dd + g * G::NE) * (int)sizeof(T); }
  template <class F>
  __device__ __forceinline__ void eval(acc_t (&C)[G::NB], F&& f) const {
    constexpr int NBG = G::NBG, GS = G::GS, GB = G::GB, NSG = G::NSG, VEC = G::VEC, NB = G::NB;
    // W never changes, so the compiler would merge the operand reads of successive evaluations (and then spill a whole
    // W worth of registers to "save" the LDS reads): make the lane's offset opaque once per evaluation (the offset,
    // not the pointer: the pointer keeps its LDS address space)
    int off = lane_off;
    asm volatile("" : "+v"(off));
    const T* base = Wl + off;
    CH a[NSG + 1][NBG];
#pragma unroll
    for (int bg = 0; bg < NBG; ++bg) a[0][bg] = *(const CH*)(base + (size_t)bg * 64 * VEC);
#pragma unroll
    for (int b = 0; b < NB; ++b) C[b] = acc_t{0, 0, 0, 0};
    T bop[G::NE + 1];
    bop[0] = f(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, G::NE>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int sg = s / GS, vs = s % GS;
      if constexpr (vs == GS - 1 && sg + 1 < NSG) {
#pragma unroll
        for (int bg = 0; bg < NBG; ++bg) a[sg + 1][bg] = *(const CH*)(base + (size_t)((sg + 1) * NBG + bg) * 64 * VEC);
      }
      if constexpr (s + 1 < G::NE) bop[s + 1] = f(std::integral_constant<int, s + 1>{});
#pragma unroll
      for (int bg = 0; bg < NBG; ++bg)
#pragma unroll
        for (int vb = 0; vb < GB; ++vb)
          C[bg * GB + vb] = G::TR::mfma(a[sg][bg].v[vs * GB + vb], bop[s], C[bg * GB + vb]);
      // pipeline of the step: one MFMA, then a share of the fillers (masks: 0x8 MFMA, 0x2 VALU, 0x100 DS read,
      // 0x20 VMEM read, 0x40 VMEM write, 0x4 SALU)
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, kFill, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x40, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x4, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);         // keep the steps apart: the scheduler must not clump the stage arithmetic
    });
  }
  // element e of k from the accumulators of the evaluation that produced it
  template <int e>
  __device__ __forceinline__ T k_of(const acc_t (&C)[G::NB]) const {
    T k_ = C[e / 4][e % 4];
    if constexpr (BIAS) k_ = k_ + bias_lds[g * G::NE + e];
    return k_;
  }
};

// Make a value opaque at this point of the program: the optimiser may neither sink the computation that produced it
// out of the k-step it belongs to nor fold it into a later one (the stage arithmetic must stay in the shadow of ITS step's
// MFMAs; sunk to the end of the tile it would also keep the previous evaluation's accumulators alive).
template <typename V>
__device__ __forceinline__ void wt_pin(V& x) { asm volatile("" : "+v"(x)); }

// wave-uniform double -> SGPR pair (the dt * coefficient products: one VALU multiply, then scalar for the whole evaluation)
__device__ __forceinline__ double wt_uniform(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float wt_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// ------------------------------------------------------------------------------------------------
// One adaptive attempt for the wave's tiles (tile = first, first + stride, ...).  FSAL-shaped tableau with S rows.
// Arithmetic = step_combine / step_finish of mi_ode_step_fused.h, term for term.
// y0 is NOT held in registers: every evaluation needs element s of it once, at step s, so it is streamed from L2 again
// (16-byte chunks, kLook steps ahead) - one plane less in the register file, five L2 plane reads more per attempt.
// NEED_MID: an output time falls into this attempt (wave-uniform): carry the y_mid sum (dopri5.py:42) as one more
// running accumulator and evaluate the dense output from registers (speculatively, as k_step_linear_mfma does).
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int S, bool NEED_MID, bool SC0, bool BIAS>
__device__ __forceinline__ void wt_attempt_pass(const StepArgs& A, const StepPlanes<T, S>& P, const WtCtx<T, D, BIAS>& cx, Acc& acc,
                                                const double* t_out, long long first_tile, long long tile_stride) {
  using G = Wt<T, D>;
  using acc_t = typename G::acc_t;
  using CH = typename G::CH;
  constexpr int NE = G::NE, VEC = G::VEC, NB = G::NB, NQ = NE / VEC;
  constexpr int PS = (S + 2) / 2;                 // k_1..k_{PS-1} are kept; from k_PS on the sums are carried (S = 6: 4, S = 3: 2)
  static_assert(PS >= 2 && PS <= S, "tableau too short for the running-sum schedule");
  constexpr int kLook = (MI_WT_LOOK / VEC) * VEC < NE ? (MI_WT_LOOK / VEC) * VEC : NE - VEC;   // y0 chunks are requested this many k-steps ahead
  const int d = A.dim;
  const long long ntiles = (A.batch + 15) / 16;
  if (first_tile >= ntiles) return;
  T hs_tile = P.hs;

  for (long long tile = first_tile; tile < ntiles; tile += tile_stride) {
    asm volatile("" : "+v"(hs_tile));             // the dt * coefficient products are formed where they are used (see lin_attempt_pass)
    const bool live = tile * 16 + cx.r < A.batch;
    const bool more = tile + tile_stride < ntiles;
    T kst[PS - 1][NE];
    T accs[S - PS > 0 ? S - PS : 1][NE];          // running sums of stages PS+1 .. S
    T errv[NE], midv[NE], y1v[NE];
    CH y0c[S + 1][NQ];                            // y0, chunk q, as requested for evaluation J (SSA values: only a window is ever live)
    acc_t Ca[NB], Cb[NB];
    double mxa = 0.0, mxb = 0.0;
    TileIo<T> iy, i1;
    TileOut<T> o0, o1;
    iy.open(P.y0, tile * 16, A.batch, d); i1.open(P.f0, tile * 16, A.batch, d);
    const int voff = cx.voff_of(tile * 16, A.batch), voff_st = cx.voff_raw();
    tile_load<T, NE, SC0>(i1, voff, kst[0]);      // k_1 = f0
#pragma unroll
    for (int q = 0; q < kLook / VEC; ++q) y0c[1][q] = iy.template load<SC0>(voff + q * 16);

    auto stage = [&](auto j_c) {
      constexpr int J = decltype(j_c)::value;     // stage 1..S: y_J from k_1..k_J, then k_{J+1} = f(y_J)
      acc_t (&Cprev)[NB] = (J & 1) ? Cb : Ca;     // k_J   (J >= 2)
      acc_t (&Cnew)[NB] = (J & 1) ? Ca : Cb;      // k_{J+1}
      // dt * coefficient products of this stage, wave-uniform (SGPRs).  hs is made opaque HERE so that the products of a
      // stage are formed at its start and die at its end (hoisted to the top of the tile they would pin ~70 SGPRs)
      T hs = hs_tile;
      asm volatile("" : "+v"(hs));
      T pb[S + 1][S + 1], pe[S + 2], pm[S + 2];
      if constexpr (J < PS) {
#pragma unroll
        for (int i = 1; i <= J; ++i) pb[J][i] = wt_uniform(hs * (T)A.beta[J - 1][i - 1]);
      } else if constexpr (J == PS) {
#pragma unroll
        for (int sg = J; sg <= S; ++sg)
#pragma unroll
          for (int i = 1; i <= J; ++i) pb[sg][i] = wt_uniform(hs * (T)A.beta[sg - 1][i - 1]);
#pragma unroll
        for (int i = 1; i <= J; ++i) { pe[i] = wt_uniform(hs * (T)A.e[i - 1]); if constexpr (NEED_MID) pm[i] = wt_uniform(hs * (T)A.cmid[i - 1]); }
      } else {
#pragma unroll
        for (int sg = J; sg <= S; ++sg) pb[sg][J] = wt_uniform(hs * (T)A.beta[sg - 1][J - 1]);
        pe[J] = wt_uniform(hs * (T)A.e[J - 1]);
        if constexpr (NEED_MID) pm[J] = wt_uniform(hs * (T)A.cmid[J - 1]);
      }
      if constexpr (J == S) o0.open(P.y1, tile * 16, A.batch, d);
      cx.eval(Cnew, [&](auto s_c) -> T {
        constexpr int s = decltype(s_c)::value;
        // request the y0 chunk that will be needed kLook steps from now (this evaluation's, or the next one's)
        if constexpr (s % VEC == 0) {
          constexpr int t = s + kLook;
          if constexpr (t < NE) y0c[J][t / VEC] = iy.template load<SC0>(voff + (t / VEC) * 16);
          else if constexpr (J < S) y0c[J + 1][(t - NE) / VEC] = iy.template load<SC0>(voff + ((t - NE) / VEC) * 16);
        }
        const T y0e = y0c[J][s / VEC].v[s % VEC];
        T kj;
        if constexpr (J == 1) kj = kst[0][s]; else kj = cx.template k_of<s>(Cprev);
        T ysv;
        if constexpr (J < PS) {
          if constexpr (J > 1) kst[J - 1][s] = kj;
          T a = pb[J][1] * kst[0][s];
#pragma unroll
          for (int i = 2; i <= J; ++i) a = a + pb[J][i] * kst[i - 1][s];
          ysv = y0e + a;
        } else if constexpr (J == PS) {
          T a = pb[J][1] * kst[0][s];
#pragma unroll
          for (int i = 2; i < J; ++i) a = a + pb[J][i] * kst[i - 1][s];
          a = a + pb[J][J] * kj;
          ysv = y0e + a;
#pragma unroll
          for (int sg = J + 1; sg <= S; ++sg) {
            T b = pb[sg][1] * kst[0][s];
#pragma unroll
            for (int i = 2; i < J; ++i) b = b + pb[sg][i] * kst[i - 1][s];
            accs[sg - PS - 1][s] = b + pb[sg][J] * kj;
            wt_pin(accs[sg - PS - 1][s]);
          }
          T er = pe[1] * kst[0][s];
#pragma unroll
          for (int i = 2; i < J; ++i) er = er + pe[i] * kst[i - 1][s];
          errv[s] = er + pe[J] * kj;
          wt_pin(errv[s]);
          if constexpr (NEED_MID) {
            T ym = pm[1] * kst[0][s];
#pragma unroll
            for (int i = 2; i < J; ++i) ym = ym + pm[i] * kst[i - 1][s];
            midv[s] = ym + pm[J] * kj;
            wt_pin(midv[s]);
          }
        } else {
          ysv = y0e + (accs[J - PS - 1][s] + pb[J][J] * kj);
#pragma unroll
          for (int sg = J + 1; sg <= S; ++sg) { accs[sg - PS - 1][s] = accs[sg - PS - 1][s] + pb[sg][J] * kj; wt_pin(accs[sg - PS - 1][s]); }
          errv[s] = errv[s] + pe[J] * kj;
          wt_pin(errv[s]);
          if constexpr (NEED_MID) { midv[s] = midv[s] + pm[J] * kj; wt_pin(midv[s]); }
        }
        if constexpr (J == 1) { mxa = fmax(mxa, (double)fabs(y0e)); wt_pin(mxa); }
        if constexpr (J == S) {                   // y1 = y_S (FSAL, rk_common.py:58): leaves as soon as a 16-byte chunk is complete
          y1v[s] = ysv;
          mxb = fmax(mxb, (double)fabs(ysv));
          wt_pin(mxb);
          if constexpr (s % VEC == VEC - 1) {
            CH c;
#pragma unroll
            for (int e = 0; e < VEC; ++e) c.v[e] = y1v[s - (VEC - 1) + e];
            o0.store(voff_st + (s / VEC) * 16, c);
          }
          if constexpr (s == NE / 2) {            // pull the next tile's y0 / f0 towards this XCD's L2: one dword per 128-byte line
            if (more) {
              long long left = A.batch - (tile + tile_stride) * 16;
              left = left > 16 ? 16 : left;
              const int lines = (int)(left * d * (long long)sizeof(T) / 128);
              const char* n0 = (const char*)(P.y0 + (tile + tile_stride) * 16 * d);
              const char* n1 = (const char*)(P.f0 + (tile + tile_stride) * 16 * d);
#pragma unroll
              for (int q = 0; q < (16 * D * (int)sizeof(T) / 128 + 63) / 64; ++q) {
                const int ln = q * 64 + cx.lane;
                const unsigned off = ln < lines ? ln * 128 : 0;
                const int t0 = *(const int*)(n0 + off);
                const int t1 = *(const int*)(n1 + off);
                asm volatile("" ::"v"(t0), "v"(t1));
              }
            }
          }
        }
        return ysv;
      });
    };
    for_stages<1, S>(stage);

    // f1 = k_{S+1}, err (rk_common.py:60), norms (misc.py:256-263)
    acc_t (&Cfin)[NB] = (S & 1) ? Ca : Cb;
    o1.open(P.f1, tile * 16, A.batch, d);
    T hs = hs_tile;
    asm volatile("" : "+v"(hs));
    const T peK = wt_uniform(hs * (T)A.e[S]);
    double se = 0.0;
    T f1v[NE];
    static_for<0, NE>([&](auto e_c) {
      constexpr int e = decltype(e_c)::value;
      const T k = cx.template k_of<e>(Cfin);
      f1v[e] = k;
      const T err = errv[e] + peK * k;
      se += (double)err * (double)err;
      if constexpr (e % VEC == VEC - 1) {
        CH c;
#pragma unroll
        for (int q = 0; q < VEC; ++q) c.v[q] = f1v[e - (VEC - 1) + q];
        o1.store(voff_st + (e / VEC) * 16, c);
      }
    });
    if (live) { acc.maxa = fmax(acc.maxa, mxa); acc.maxb = fmax(acc.maxb, mxb); acc.suma += se; }
    if constexpr (NEED_MID) {
      if (live) {
        const T pmK = wt_uniform(hs * (T)A.cmid[S]);
        T* out = (T*)A.out;
        const long long idx0 = (tile * 16 + cx.r) * (long long)d + cx.g * NE;
        static_for<0, NQ>([&](auto q_c) {
          constexpr int q = decltype(q_c)::value;
          const CH cy = iy.template load<SC0>(voff + q * 16), cf = i1.template load<SC0>(voff + q * 16);   // y0, f0 again
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            constexpr int e0 = q * VEC;
            const int e = e0 + v;
            const T ymid = cy.v[v] + (midv[e] + pmK * f1v[e]);
            T co[5];
            quartic_from_mid<T>(cy.v[v], y1v[e], ymid, cf.v[v], f1v[e], (T)P.dt64, co);
            for (int j = P.j_lo; j < P.j_hi; ++j)
              out[(long long)j * A.n_plane + idx0 + e] = quartic_eval<T>(co, interp_x<T>(P.t_start, P.t_new, t_out[j]));
          }
        });
      }
    }
  }
}

// before_integrate, first half (lin_f0_pass): f0 = f(t0, y0), sums of (y0/sc)^2 and (f0/sc)^2, the non-finite flag;
// optionally seeds a state plane and solution[0] with y0 in the same pass.
template <typename T, int D, bool SC0, bool BIAS>
__device__ __forceinline__ void wt_f0_pass(const StepArgs& A, const T* y0p, T* f0_out, T* copy_a, T* copy_b, const WtCtx<T, D, BIAS>& cx,
                                           Acc& acc, long long first_tile, long long tile_stride) {
  using G = Wt<T, D>;
  using acc_t = typename G::acc_t;
  constexpr int NE = G::NE;
  const int d = A.dim;
  const long long ntiles = (A.batch + 15) / 16;
  for (long long tile = first_tile; tile < ntiles; tile += tile_stride) {
    const bool live = tile * 16 + cx.r < A.batch;
    T y0[NE], kn[NE];
    TileIo<T> i0;
    i0.open(y0p, tile * 16, A.batch, d);
    const int voff = cx.voff_of(tile * 16, A.batch);
    tile_load<T, NE, SC0>(i0, voff, y0);
    acc_t C[G::NB];
    cx.eval(C, [&](auto s_c) -> T { return y0[decltype(s_c)::value]; });
    static_for<0, NE>([&](auto e_c) { kn[decltype(e_c)::value] = cx.template k_of<decltype(e_c)::value>(C); });
    {
      TileOut<T> o;
      o.open(f0_out, tile * 16, A.batch, d);
      tile_store<T, NE>(o, cx.voff_raw(), kn);
      if (copy_a != nullptr) { o.open(copy_a, tile * 16, A.batch, d); tile_store<T, NE>(o, cx.voff_raw(), y0); }
      if (copy_b != nullptr) { o.open(copy_b, tile * 16, A.batch, d); tile_store<T, NE>(o, cx.voff_raw(), y0); }
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        if (cx.g * NE + e < d) {
          const T sc = (T)A.cp.atol + fabs(y0[e]) * (T)A.cp.rtol;     // misc.py:225
          const double q0 = (double)(y0[e] / sc);
          acc.suma += q0 * q0;
          if (!finite_(y0[e])) acc.flag = 1;
          const double q1 = (double)(kn[e] / sc);
          acc.sumb += q1 * q1;                                         // misc.py:228
        }
      }
    }
  }
}

// before_integrate, second half (lin_initb_pass, misc.py:235-237): f1 = f(t0 + h0, y0 + h0 f0), sum of ((f1 - f0)/sc)^2
template <typename T, int D, bool SC0, bool BIAS>
__device__ __forceinline__ void wt_initb_pass(const StepArgs& A, const T* y0p, const T* f0p, T h0, const WtCtx<T, D, BIAS>& cx, Acc& acc,
                                              long long first_tile, long long tile_stride) {
  using G = Wt<T, D>;
  using acc_t = typename G::acc_t;
  constexpr int NE = G::NE;
  const int d = A.dim;
  const long long ntiles = (A.batch + 15) / 16;
  for (long long tile = first_tile; tile < ntiles; tile += tile_stride) {
    const bool live = tile * 16 + cx.r < A.batch;
    T y0[NE], f0[NE];
    TileIo<T> i0, i1;
    i0.open(y0p, tile * 16, A.batch, d); i1.open(f0p, tile * 16, A.batch, d);
    const int voff = cx.voff_of(tile * 16, A.batch);
    tile_load<T, NE, SC0>(i0, voff, y0);
    tile_load<T, NE, SC0>(i1, voff, f0);
    acc_t C[G::NB];
    cx.eval(C, [&](auto s_c) -> T { constexpr int s = decltype(s_c)::value; return y0[s] + h0 * f0[s]; });   // misc.py:235
    if (live) {
      static_for<0, NE>([&](auto e_c) {
        constexpr int e = decltype(e_c)::value;
        if (cx.g * NE + e < d) {
          const T kn = cx.template k_of<e>(C);
          const T sc = (T)A.cp.atol + fabs(y0[e]) * (T)A.cp.rtol;
          const double q = (double)((kn - f0[e]) / sc);             // misc.py:237
          acc.suma += q * q;
        }
      });
    }
  }
}

// One launch per attempt (fusion = 'step'): grid = min(CUs, ceil(tiles / 4)) workgroups of 4 waves.
template <typename T, int D, int S, bool BIAS>
__global__ __launch_bounds__(256) void k_step_linear_wt(StepArgs A) {
  StepPlanes<T, S> P;
  if (!resolve_step<T, S>(A, P)) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ double red[80];
  WtCtx<T, D, BIAS> cx;
  cx.init(A.rhs, smem_raw, A.dim);
  Acc acc;
  const long long first = (long long)blockIdx.x * Wt<T, D>::kWaves + cx.wave, stride = (long long)gridDim.x * Wt<T, D>::kWaves;
  if (P.j_hi > P.j_lo) wt_attempt_pass<T, D, S, true, false, BIAS>(A, P, cx, acc, A.t_out, first, stride);
  else wt_attempt_pass<T, D, S, false, false, BIAS>(A, P, cx, acc, A.t_out, first, stride);
  finish_attempt(A, acc, red);
}

template <typename T, int D, int PHASE, bool BIAS>
__global__ __launch_bounds__(256) void k_init_linear_wt(InitArgs I) {
  const StepArgs& A = I.s;
  const Ctl* c = A.ctl;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ double red[80];
  WtCtx<T, D, BIAS> cx;
  cx.init(A.rhs, smem_raw, A.dim);
  Acc acc;
  const long long first = (long long)blockIdx.x * Wt<T, D>::kWaves + cx.wave, stride = (long long)gridDim.x * Wt<T, D>::kWaves;
  T* plane_y = (T*)(A.planes + (long long)c->idx_y0 * A.stride);
  T* plane_f = (T*)(A.planes + (long long)c->idx_k[0] * A.stride);
  if constexpr (PHASE == 0) wt_f0_pass<T, D, false, BIAS>(A, (const T*)I.y0, plane_f, plane_y, (T*)I.copy_b, cx, acc, first, stride);
  else wt_initb_pass<T, D, false, BIAS>(A, plane_y, plane_f, (T)c->h0, cx, acc, first, stride);
  block_reduce_store<false>(acc, red, A.partials + (long long)blockIdx.x * kRec);
}

}  // namespace mi
