// Whole-attempt fused kernels: all S stages of one adaptive RK attempt for a tile of trajectories stay on chip.
//
// Legal because every catalogue RHS is trajectory(row)-local: k_2..k_S never have to visit HBM.  Traffic per
// attempt drops from 34 planes (per-stage structure) to
//     reads  y0, f0                       2 planes
//     writes y1, f1                       2 planes
//     (+ one plane per requested output time that falls into (t0, t0+dt])
// Dense output is evaluated HERE, from registers (y0, y1, k_1..k_{S+1} are all live at the end of the attempt), and
// written speculatively: if the attempt is rejected, the same output indices are simply overwritten by the
// accepted step that eventually covers those times (the output cursor only advances on accept).
// The arithmetic is the per-stage kernels' arithmetic, operation for operation (same combine order, same MFMA
// accumulation order), so both paths produce identical bits; only the schedule differs.
// Bound: tiny systems (dim 2-3) stay HBM/latency-bound; the linear RHS at dim 128 becomes fp64-MFMA-bound
// (6 x 256 flop per element per attempt against 40 B).
#pragma once
#include <type_traits>
#include "mi_ode_ctrl_dev.h"
#include "mi_ode_dense.h"
#include "mi_ode_dev.h"
#include "mi_ode_stage_linear.h"
#include "mi_ode_stage_rowlocal.h"

#ifndef MI_ABL
#define MI_ABL 0               // tuning aid (scripts/build_ablations.sh): 1 no MFMAs, 2 one-term stage combinations, 4 no barriers, 8 no plane loads / stores
#endif

namespace mi {

struct StepArgs {
  Ctl* ctl;
  char* planes;
  long long stride;
  long long batch;
  int dim;
  int interp;                  // MI_ODE_INTERP_* of the solver
  void* out;                   // [n_out, n_plane] solution rows of the current advance() call
  const double* t_out;         // device: requested output times
  long long n_plane;           // batch * dim
  double beta[MI_ODE_MAX_STAGES][MI_ODE_MAX_STAGES];
  double alpha[MI_ODE_MAX_STAGES];
  double e[kMaxK];             // c_error
  double csol[kMaxK];          // c_sol (only read for tableaus that are not FSAL shaped)
  double cmid[kMaxK];          // dense-output mid-point weights
  double* partials;
  RhsParams rhs;
  // single-rank runs: the last workgroup to finish runs the step controller itself (one launch per attempt)
  unsigned* ticket;            // device word, zero between launches; nullptr: the controller is a separate launch
  CtrlParams cp;
};

// Epilogue of a whole-attempt kernel.  Every workgroup publishes its reduction record; the LAST one to arrive
// (device-scope ticket) reduces all records and applies the controller.  Inter-workgroup visibility follows the
// gfx950 write-through recipe (cdna_hip_programming.md Guideline 16, R1): the record is stored with agent-scope
// (sc1, write-through) stores, the storing lane drains vmcnt, then takes a relaxed agent-scope ticket; the last
// arriver reads the records with sc1 loads (which bypass its L1).  No release/acquire fence - a per-workgroup
// `buffer_wbl2` costs microseconds here and made this path slower than a separate controller launch.
// No workgroup reads Ctl after the controller modified it: every workgroup takes its ticket at its very end and
// Ctl is only read at kernel start.
__device__ __forceinline__ void finish_attempt(const StepArgs& A, const Acc& acc, double* red) {
  if (A.ticket == nullptr) {
    block_reduce_store<false>(acc, red, A.partials + (long long)blockIdx.x * kRec);
    return;
  }
  block_reduce_store<true>(acc, red, A.partials + (long long)blockIdx.x * kRec);
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!s_last) return;
  __shared__ double rec[kRec];
  reduce_block_records<true>(A.partials, (int)gridDim.x, rec);
  if (threadIdx.x == 0) {
    rec[R_N] = (double)A.cp.n_local;
    controller_apply(A.ctl, rec, PH_ATTEMPT, A.cp);
    __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T, int S>
struct StepPlanes {
  const T* y0;
  const T* f0;
  T* y1;
  T* f1;                       // plane idx_k[S]
  T hs, t0;
  double t_start, t_new, dt64; // attempt interval in float64 (dopri5.py:113-116)
  int j_lo, j_hi;              // requested outputs in (t_start, t_new]
};

template <typename T, int S>
__device__ __forceinline__ bool resolve_step(const StepArgs& A, StepPlanes<T, S>& P) {
  const Ctl* c = A.ctl;
  if (c->done) return false;
  char* base = A.planes;
  P.y0 = (const T*)(base + (long long)c->idx_y0 * A.stride);
  P.y1 = (T*)(base + (long long)c->idx_y1 * A.stride);
  P.f0 = (const T*)(base + (long long)c->idx_k[0] * A.stride);
  P.f1 = (T*)(base + (long long)c->idx_k[S] * A.stride);
  P.hs = (T)c->dt;                               // rk_common.py:46
  P.t0 = (T)c->t1;                               // rk_common.py:45
  P.t_start = c->t1; P.dt64 = c->dt; P.t_new = c->t1 + c->dt;
  int j = c->next_out;
  P.j_lo = j;
  if (A.t_out != nullptr)
    while (j < c->n_out && !(A.t_out[j] > P.t_new)) ++j;     // same test as the controller's output cursor
  P.j_hi = j;
  return true;
}

// stages 1..S at compile time: f(integral_constant<int, sigma>)
template <int SG, int S, class F>
__device__ __forceinline__ void for_stages(F&& f) {
  f(std::integral_constant<int, SG>{});
  if constexpr (SG < S) for_stages<SG + 1, S>(f);
}

// y1 of a tableau that is not FSAL shaped (rk_common.py:55-56): y0 + add_n((dt * c_sol_j) * k_j), all S+1 derivatives
template <typename T, int S>
__device__ __forceinline__ T step_y1_general(T y0, const T* k, T hs, const StepArgs& A) {
  T acc = (hs * (T)A.csol[0]) * k[0];
#pragma unroll
  for (int j = 1; j <= S; ++j) acc = acc + (hs * (T)A.csol[j]) * k[j];
  return y0 + acc;
}

// y_sigma for stage SG (1-based) from y0 and k[0..SG-1]: misc._scaled_dot_product order (rk_common.py:51)
template <typename T, int SG>
__device__ __forceinline__ T step_combine(T y0, const T* k, T hs, const StepArgs& A) {
  T acc = (hs * (T)A.beta[SG - 1][0]) * k[0];
#pragma unroll
  for (int j = 1; j < SG; ++j) acc = acc + (hs * (T)A.beta[SG - 1][j]) * k[j];
  return y0 + acc;
}

// err (rk_common.py:60) and y_mid (dopri5.py:42) from all S+1 stage derivatives.  y_mid only feeds the dense output:
// `need_mid` (wave-uniform: an output time falls into this attempt) skips its 2(S+1) operations otherwise.
template <typename T, int S>
__device__ __forceinline__ void step_finish(T y0, const T* k, T hs, const StepArgs& A, T& err, T& ymid, bool need_mid = true) {
  T er = (hs * (T)A.e[0]) * k[0];
#pragma unroll
  for (int j = 1; j <= S; ++j) er = er + (hs * (T)A.e[j]) * k[j];
  err = er;
  ymid = y0;
  if (need_mid) {
    T ym = (hs * (T)A.cmid[0]) * k[0];
#pragma unroll
    for (int j = 1; j <= S; ++j) ym = ym + (hs * (T)A.cmid[j]) * k[j];
    ymid = y0 + ym;
  }
}

// speculative dense output of one element for every requested time inside the attempt
template <typename T, int S, bool TS>
__device__ __forceinline__ void step_emit(const StepArgs& A, const StepPlanes<T, S>& P, T y0, T y1, const T* k, T ymid,
                                          long long idx, const double* t_out) {
  if (P.j_hi <= P.j_lo) return;
  T* out = (T*)A.out;
  if constexpr (TS) {
    for (int j = P.j_lo; j < P.j_hi; ++j)
      out[(long long)j * A.n_plane + idx] = tsit5_dense<T, S + 1>(y0, k, P.t_start, P.t_new, t_out[j], A.interp);
  } else {
    T co[5];
    quartic_from_mid<T>(y0, y1, ymid, k[0], k[S], (T)P.dt64, co);
    for (int j = P.j_lo; j < P.j_hi; ++j)
      out[(long long)j * A.n_plane + idx] = quartic_eval<T>(co, interp_x<T>(P.t_start, P.t_new, t_out[j]));
  }
}

// ------------------------------------------------------------------------------------------------
// (1) tiny row-local systems: one thread per trajectory, everything in registers, one launch per attempt
// ------------------------------------------------------------------------------------------------
template <typename T, int S, bool TS, class RHS, bool FSAL = true>
__global__ __launch_bounds__(256) void k_step_rowlocal(StepArgs A) {
  constexpr int D = RHS::D;
  using Row = RowVec<T, D>;
  StepPlanes<T, S> P;
  if (!resolve_step<T, S>(A, P)) return;
  const RHS rhs(A.rhs);
  const T sign = (T)A.rhs.sign;
  const T hs = P.hs;
  Acc acc;
  for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < A.batch;
       row += (long long)gridDim.x * blockDim.x) {
    const Row y0 = *(const Row*)(P.y0 + row * D);
    T k[S + 1][D];
    {
      const Row f0 = *(const Row*)(P.f0 + row * D);
#pragma unroll
      for (int d = 0; d < D; ++d) k[0][d] = f0.v[d];
    }
    T ys[D];
    auto stage = [&](auto sg_c) {
      constexpr int SG = decltype(sg_c)::value;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        T kk[SG];
#pragma unroll
        for (int j = 0; j < SG; ++j) kk[j] = k[j][d];
        ys[d] = step_combine<T, SG>(y0.v[d], kk, hs, A);
      }
      T kn[D];
      rhs(sign * (P.t0 + (T)A.alpha[SG - 1] * hs), ys, kn);
#pragma unroll
      for (int d = 0; d < D; ++d) k[SG][d] = sign * kn[d];
    };
    for_stages<1, S>(stage);
    Row y1, f1;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      T kk[S + 1];
#pragma unroll
      for (int j = 0; j <= S; ++j) kk[j] = k[j][d];
      T err, ymid;
      step_finish<T, S>(y0.v[d], kk, hs, A, err, ymid, !TS && P.j_hi > P.j_lo);
      if constexpr (!FSAL) ys[d] = step_y1_general<T, S>(y0.v[d], kk, hs, A);     // rk_common.py:55-56
      y1.v[d] = ys[d];                                       // FSAL: y1 = y_S (rk_common.py:58)
      f1.v[d] = k[S][d];
      acc.maxa = fmax(acc.maxa, (double)fabs(y0.v[d]));
      acc.maxb = fmax(acc.maxb, (double)fabs(ys[d]));
      acc.suma += (double)err * (double)err;
      step_emit<T, S, TS>(A, P, y0.v[d], ys[d], kk, ymid, row * D + d, A.t_out);
    }
    *(Row*)(P.y1 + row * D) = y1;
    *(Row*)(P.f1 + row * D) = f1;
  }
  __shared__ double red[80];
  finish_attempt(A, acc, red);
}

// ------------------------------------------------------------------------------------------------
// (1b) fixed grid (solvers.py:82-104) for the tiny row-local systems: trajectories never interact on a fixed grid,
//      so the WHOLE integration is one launch: a thread carries its trajectory through every grid interval in
//      registers (Euler: fixed_grid.py:6-7, RK4 3/8 rule: rk_common.py:73-81, literally) and streams solution[i+1].
//      Traffic = y0 in + T solution rows out; no k planes, no per-step launches.
// ------------------------------------------------------------------------------------------------
struct FixedArgs {
  const void* y0;
  void* out;                   // [T, batch*D]
  const double* t;             // device: the requested output times, float64 (values already rounded to the state dtype)
  const double* grid;          // device: the time grid the steps are taken on (solvers.py:86: grid_constructor(func, y0, t));
  int M;                       // M intervals; by default the grid IS t (M = T - 1)
  double eps;                  // FixedGridODESolver(eps=...): added to the time the step function evaluates f at (fixed_grid.py:7, 42)
  long long batch;
  int T;
  int rk4;                     // 0: Euler, 1: RK4 (3/8 rule)
  int dim;                     // row length (tile kernels: may be smaller than their instantiated width)
  RhsParams rhs;
  long long* clk;              // pinned host {shader cycles, 10 ns ticks} of workgroup 0's run (mi_ode_stats.clock_mhz), or null
};

// shader clock of a fixed-grid launch: thread 0 of workgroup 0 brackets its whole loop (a one-trajectory call IS that thread)
struct FixedClk {
  long long c0 = 0, w0 = 0;
  __device__ __forceinline__ void begin() {
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = (long long)__builtin_readcyclecounter(); w0 = (long long)wall_clock64(); }
  }
  __device__ __forceinline__ void end(long long* clk) const {
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk != nullptr) {
      __hip_atomic_store(clk + 0, (long long)__builtin_readcyclecounter() - c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(clk + 1, (long long)wall_clock64() - w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
};

template <typename T, class RHS>
__global__ __launch_bounds__(256) void k_fixed_rowlocal(FixedArgs A) {
  constexpr int D = RHS::D;
  using Row = RowVec<T, D>;
  const RHS rhs(A.rhs);
  const T sign = (T)A.rhs.sign;
  const long long n = rhs_is_coop<RHS>::value ? A.batch * A.dim : A.batch * D;      // elements per solution row
  const T* y0p = (const T*)A.y0;
  T* out = (T*)A.out;
  FixedClk clk;
  clk.begin();
  // e0: this thread's first element; live: it exists (cooperative right-hand sides call rhs() from every thread of the workgroup - it
  // contains barriers - so a thread without a trajectory runs the loop too and only skips its loads and stores)
  auto run = [&](long long e0, bool live) {
    Row y;
#pragma unroll
    for (int d = 0; d < D; ++d) y.v[d] = (T)0;
    if (live) {
      y = *(const Row*)(y0p + e0);
      *(Row*)(out + e0) = y;                                   // solution = [y0]
    }
    int j = 1;
    const T eps = (T)A.eps;
    for (int i = 0; i < A.M; ++i) {
      const T t0 = (T)A.grid[i];                             // solvers.py:84: the grid is cast to the STATE dtype
      const T t1 = (T)A.grid[i + 1];
      const T dt = t1 - t0;
      const T te = t0 + eps;                                 // fixed_grid.py:7 / :42: the step function sees t + eps
      T k1[D], k2[D], k3[D], k4[D], ys[D];
      Row yn;
      rhs(sign * te, y.v, k1);
#pragma unroll
      for (int d = 0; d < D; ++d) k1[d] = sign * k1[d];
      if (!A.rk4) {
#pragma unroll
        for (int d = 0; d < D; ++d) yn.v[d] = y.v[d] + dt * k1[d];                       // fixed_grid.py:7
      } else {
#pragma unroll
        for (int d = 0; d < D; ++d) ys[d] = y.v[d] + dt * k1[d] / (T)3;                 // rk_common.py:77
        rhs(sign * (te + dt / (T)3), ys, k2);
#pragma unroll
        for (int d = 0; d < D; ++d) { k2[d] = sign * k2[d]; ys[d] = y.v[d] + dt * (k1[d] / (T)-3 + k2[d]); }   // :78
        rhs(sign * (te + dt * (T)2 / (T)3), ys, k3);
#pragma unroll
        for (int d = 0; d < D; ++d) { k3[d] = sign * k3[d]; ys[d] = y.v[d] + dt * (k1[d] - k2[d] + k3[d]); }   // :79
        rhs(sign * (te + dt), ys, k4);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          k4[d] = sign * k4[d];
          yn.v[d] = y.v[d] + (k1[d] + (T)3 * k2[d] + (T)3 * k3[d] + k4[d]) * (dt / (T)8);                       // :81
        }
      }
      // solvers.py:97-100: every requested time the step reaches, exactly y0 / y1 on a grid hit, else _linear_interp (:106-115)
      while (j < A.T && t1 >= (T)A.t[j]) {
        const T tj = (T)A.t[j];
        Row o;
#pragma unroll
        for (int d = 0; d < D; ++d)
          o.v[d] = (tj == t0) ? y.v[d] : ((tj == t1) ? yn.v[d] : y.v[d] + ((yn.v[d] - y.v[d]) / (t1 - t0)) * (tj - t0));
        if (live) *(Row*)(out + (long long)j * n + e0) = o;
        ++j;
      }
      y = yn;
    }
  };
  if constexpr (rhs_is_coop<RHS>::value) {                   // a thread per state element, tpw trajectories per workgroup (uniform trip count)
    const int tpw = RHS::tpw(A.rhs, A.dim);
    const int slot = (int)threadIdx.x / A.dim, col = (int)threadIdx.x - slot * A.dim;
    for (long long tr0 = (long long)blockIdx.x * tpw; tr0 < A.batch; tr0 += (long long)gridDim.x * tpw) {
      const long long traj = tr0 + slot;
      run(traj * A.dim + col, slot < tpw && traj < A.batch);
    }
  } else {
    for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < A.batch; row += (long long)gridDim.x * blockDim.x) run(row * D, true);
  }
  clk.end(A.clk);
}

// ------------------------------------------------------------------------------------------------
// (2) linear RHS, dim D in {16, 32, 64, 128}: the MFMA tile kernel with the stage loop inside.
//     Tile = 16 rows; each thread owns the 4 accumulator-layout elements (row = acc_row(lane, i), col = 16w + lane&15)
//     of every stage derivative k_1..k_{S+1} in registers; per stage: combine in registers -> y_sigma tile to LDS ->
//     barrier -> 32 MFMA steps against the wave's resident W slice -> k_{sigma+1} (registers) -> barrier.
//     y0 / f0 of the NEXT tile are prefetched into registers during the stages of the current one.
// ------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a workgroup-scope fence, i.e. an
// s_waitcnt vmcnt(0): every global load still in flight (the next tile's prefetch) and every store (the previous
// tile's y1 / f1 / outputs) would have to land before the FIRST stage of a tile may start its MFMA phase - that exposed
// ~2 us of HBM latency per tile.  The tile kernels only exchange data through LDS at these points.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Per-thread context of the 16-row-tile linear kernels: the wavefront's W slice (resident in VGPRs), bias, LDS tile.
//
// D = 256 (round 6, dims 129 .. 256): sixteen wavefronts, four per SIMD, 128 registers each - a wavefront's slice of W (64 values
// per lane, 128 registers in float64) cannot stay resident next to the stage derivatives of the tile.  STREAM: the slice is read
// again for every evaluation, from a copy of W in the order the lanes consume it (k_lin_pack below: chunk m of wavefront w is 64
// consecutive 16-byte pieces, one per lane - every wave instruction reads 1 KB of consecutive addresses), a group of MFMAs ahead of
// the ones that consume them; the last trip of an evaluation loads the FIRST chunks again - W does not depend on the tile, so the
// next evaluation's chain starts on operands that arrived long ago.  The register budget (128) has no room for the resident
// kernels' prefetch of the next tile's y0 / f0 nor for the pre-read combination coefficients: both are read where they are used
// (PF below; measured with the tile prefetch switched on anyway: 166 instead of 117 spilled registers, 6.28 against 5.54 ms in float64,
// 3.01 against 2.85 in float32).  The copy is dim-independent
// (zero padded to 256 x 256: 512 KB in float64) and stays in every XCD's L2; what the schedule needs from L2 is one 16-byte piece per
// lane for every two float64 (four float32) MFMAs: 32 B per clock and CU with the matrix pipe saturated.  Everything else - the
// accumulator-layout ownership of the stage derivatives, the two alternating LDS tiles, one barrier per evaluation - is the resident
// kernels'.  The k-permutation is the resident kernels' with a RUN-TIME group length: lane group g covers k in [g ks, (g + 1) ks),
// ks = 4 VEC trips, trips = ceil(dim / (16 VEC)) - the chain stops at the state's true row length (5 .. 8 trips in float64 instead of
// always 8), and a wavefront whose sixteen columns lie beyond dim skips its chain altogether (the wavefronts of a workgroup are
// dealt to the SIMDs round robin, so the idle ones thin out every SIMD alike): a dim-144 system costs 5 / 8 of the trips on 9 / 16 of
// the wavefronts, not the padded width.  At dim 256 this IS the resident layout (ks = 64: the four lane groups of a ds_read_b128 fall
// on their rows' own banks); other dims take two-way conflicts on the operand reads, which the ablation prices at a few percent.
// (Measured and dropped: interleaved chunks, k = (4 c + g) VEC + v - 5.81 ms at dim 256 against 5.47; the same with a 64-byte row pad
// and an XOR swizzle that makes the reads conflict free on paper - 6.44 ms in float64, 3.02 against 3.26 in float32.)
template <typename T, int D>
struct LinCtx {
  using TR = MfmaTraits<T>;
  using acc_t = typename TR::acc_t;
  static constexpr int VEC = TR::VEC;
  static constexpr int R_ = 16;
  static constexpr int LD = D + VEC;
  static constexpr int KS = D / 4;
  static constexpr int TILE = R_ * LD;                       // one stage tile; the kernels allocate kBufs of them
  static constexpr int kBufs = 2;
  static constexpr bool STREAM = D > 128;                    // W slice streamed per evaluation instead of resident
  static constexpr int NCH = KS / VEC;                       // 16-byte chunks of the slice per lane
  static constexpr bool PF = !STREAM;                        // the passes prefetch the next tile / the next combination's coefficients
  using CH = Chunk<T, VEC>;
  int lane, wave, li, lg, col;
  int d;                                                     // the state's true row length, d <= D: the tile kernels are instantiated
  bool colok;                                                // for D in {16, 32, 64, 128} and run any smaller dim zero padded (columns >= d
  T bf[STREAM ? 1 : KS];                                     // are never loaded or stored; W rows / columns >= d are zero, so the padding
  // STREAM: chunk m of this lane at wp[64 m + wl].  float32 (UNI): wp is the WAVEFRONT's slice, a uniform (scalar-register) base, wl = lane -
  // scalar base + 32-bit lane offset + immediate, no vector address arithmetic in the chain (3.02 -> 2.90 ms at 65536 x 256).  float64:
  // wp is the lane's own first chunk, wl = 0 (the scalar-base form measured 5.72 against 5.53 ms there: ten more spilled registers).
  static constexpr bool UNI = sizeof(T) == 4;
  const CH* wp;
  int wl;
  CH r0, r1;                                                 // STREAM: the two chunks the next evaluation's first MFMA group consumes (in flight or landed)
  int trips;                                                 // STREAM: ceil(dim / (16 VEC)) trips of four chunks cover k < dim
  bool active;                                               // STREAM: this wavefront owns at least one column < dim
  T bias_v, sign;                                            // contributes exact zeros to every product, sum and norm)
  bool has_bias;
  bool plain;                                                // no bias, forward time: k is the accumulator as it is (the bias add and
                                                             // the sign product are exact no-ops then; skipping them saves 16 vector
                                                             // instructions per evaluation that the matrix pipe would wait for)
  T* s_ys;                                                   // [kBufs][R_][LD]: consecutive evaluations alternate between the two tiles,
  int cur = 0;                                               // so ONE barrier per evaluation is enough (see rhs_eval)
#ifdef MI_TRACE
  long long* tr = nullptr;                                   // timeline of rhs_eval (tuning aid): 4 stamps per evaluation
  int tn = 0;
  __device__ __forceinline__ void stamp() { if (tr != nullptr && tn < 256) tr[tn++] = (long long)__builtin_readcyclecounter(); }
#else
  __device__ __forceinline__ void stamp() {}
#endif

  __device__ __forceinline__ void init(const RhsParams& rhs, T* lds, int dim) {
    if constexpr (STREAM) {                                  // (the launcher put the packed copy into rhs.w[0]: mi_lin_pack)
      const int tid = threadIdx.x;
      lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4;
      col = 16 * wave + li;
      d = dim; colok = col < dim;
      if constexpr (UNI) { wp = (const CH*)rhs.w[0] + (long long)__builtin_amdgcn_readfirstlane(wave) * NCH * 64; wl = lane; }
      else { wp = (const CH*)rhs.w[0] + (long long)wave * NCH * 64 + lane; wl = 0; }
      trips = (dim + 16 * VEC - 1) / (16 * VEC);
      active = __builtin_amdgcn_readfirstlane(16 * wave) < dim;
      if (active) { r0 = wp[wl]; r1 = wp[64 + wl]; }
      bf[0] = (T)0;
      const T* bias = (const T*)rhs.b[0];
      has_bias = bias != nullptr;
      bias_v = (has_bias && colok) ? bias[col] : (T)0;
      sign = (T)rhs.sign;
      plain = bias == nullptr && rhs.sign == 1.0;
      s_ys = lds;
    } else {
      init_matrix((const T*)rhs.w[0], (const T*)rhs.b[0], rhs.sign, false, lds, dim);
    }
  }
  // f(y) = sgn (y M + bias) with M = W, or M = W^T when `transposed` (the adjoint system a' = -s a W^T of the linear right-hand side,
  // csrc/mi_ode_linadj.h: the same resident-slice tile kernel with the other operand order of W)
  __device__ __forceinline__ void init_matrix(const T* W, const T* bias, double sgn, bool transposed, T* lds, int dim) {
    static_assert(!STREAM, "streamed W: init() takes the packed copy");
    const int tid = threadIdx.x;
    lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4;
    col = 16 * wave + li;
    d = dim; colok = col < dim;
    // (every lane loads from a valid address and selects afterwards: a load under a per-element condition is a branch and a wait per
    // element - ten microseconds per call where the adjoint kernel switches matrices twice per attempt)
    const int cc = colok ? col : 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kk = lg * KS + s, kc = kk < dim ? kk : 0;
      const T v = transposed ? W[(long long)cc * dim + kc] : W[(long long)kc * dim + cc];
      bf[s] = (colok && kk < dim) ? v : (T)0;
    }
    has_bias = bias != nullptr;
    bias_v = (has_bias && colok) ? bias[col] : (T)0;
    sign = (T)sgn;
    plain = bias == nullptr && sgn == 1.0;
    s_ys = lds;
  }
  // ... from a matrix already zero padded to [D, D] (row stride D): no bounds, every load coalesced over the sixteen lanes of a group
  __device__ __forceinline__ void init_padded(const T* Mpad, const T* bias, double sgn, T* lds, int dim) {
    static_assert(!STREAM, "streamed W: init() takes the packed copy");
    const int tid = threadIdx.x;
    lane = tid & 63; wave = tid >> 6; li = lane & 15; lg = lane >> 4;
    col = 16 * wave + li;
    d = dim; colok = col < dim;
#pragma unroll
    for (int s = 0; s < KS; ++s) bf[s] = Mpad[(lg * KS + s) * D + col];
    has_bias = bias != nullptr;
    bias_v = (has_bias && colok) ? bias[col] : (T)0;
    sign = (T)sgn;
    plain = bias == nullptr && sgn == 1.0;
    s_ys = lds;
  }
  __device__ __forceinline__ int row_of(int i) const { return TR::acc_row(lane, i); }
  // Element i of this thread inside a tile whose first row starts at `base` (a uniform pointer: tile * R_ * d elements into
  // the plane): a 32-bit offset, so the access is "scalar base + vector offset" and costs no 64-bit vector arithmetic.
  __device__ __forceinline__ unsigned off_of(int i) const { return (unsigned)(TR::acc_row(lane, i) * d + col); }
  // rows of the tile that exist (the last tile of a batch may be ragged)
  __device__ __forceinline__ int rows_here(long long tile_i, long long batch) const {
    const long long left = batch - tile_i * R_;
    return left < R_ ? (int)left : R_;
  }

  // f(ys) for the tile: ys (this thread's 4 accumulator-layout elements) -> LDS tile `cur` -> barrier -> KS MFMA steps against
  // the resident W slice -> k (same layout, reversed-time sign applied).  No closing barrier: the next evaluation writes the
  // OTHER tile, and a wavefront that gets there has passed this evaluation's barrier, which every wavefront reaches only after
  // it has finished reading that other tile (its previous evaluation's chain).
  __device__ __forceinline__ void rhs_eval(const T (&ys)[4], T (&kn)[4]) { rhs_eval(ys, kn, [] {}); }
  // `under_chain()` is issued between the first barrier and the MFMA chain: LDS reads that do not depend on the tile (the next
  // combination's coefficients) complete while the matrix pipe works instead of after the closing barrier.
  template <class F>
  __device__ __forceinline__ void rhs_eval(const T (&ys)[4], T (&kn)[4], F&& under_chain) {
    stamp();
    T* tile = s_ys + cur * TILE;
    int aoff = cur * TILE + li * LD + lg * (STREAM ? 4 * VEC * trips : KS);       // this lane's first operand chunk: lane group g covers k in [g ks, (g + 1) ks)
    if constexpr (STREAM) asm volatile("" : "+v"(aoff));     // (in a register BEFORE the barrier: under the 128-register budget the compiler kept it in scratch
                                                             // and reloaded it behind the barrier - the reload's latency in front of every chain, 6 % of the call)
    cur ^= 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[TR::acc_row(lane, i) * LD + col] = ys[i];
    if (!(MI_ABL & 4)) lds_barrier();
    stamp();
    under_chain();
    acc_t c0 = {0, 0, 0, 0};
    const T* ap = s_ys + aoff;
#if (MI_ABL & 1)
    c0[0] = ap[0] * bf[0]; c0[1] = ap[1] * bf[0]; c0[2] = ap[2] * bf[0]; c0[3] = ap[3] * bf[0];
#else
    if constexpr (STREAM) {
      static_assert(NCH % 4 == 0, "two chunk pairs per trip");
      // Two operand sets of two chunks: (r0, r1) - loaded at the end of the previous trip (or evaluation) - and (q0, q1), loaded at the
      // top of the trip.  Each set travels while the other set's MFMA group (of each of the SIMD's four wavefronts) runs, and is
      // loaded in place: no register moves.  (The compiler waits for EVERY outstanding load at a loop's top, so the loads that cross
      // the back edge must be the only ones in flight there: hence this order, pinned by the scheduling barriers.)  The last trip
      // reloads chunks 0, 1: the next evaluation's first operands - W does not depend on the tile.
      const CH* wq = wp;
#ifndef MI_STREAM_ABL
#define MI_STREAM_ABL 0                                      // tuning aid: 1 no W stream inside the chain, 2 no LDS operand reads (both: the matrix pipe + combinations alone)
#endif
#if (MI_STREAM_ABL & 2)
      CH fa = *(const CH*)ap;
#define MI_LDA(p) fa
#else
#define MI_LDA(p) (*(const CH*)(p))
#endif
#if (MI_STREAM_ABL & 1)
#define MI_LDW(p, keep) keep
#else
#define MI_LDW(p, keep) (p)
#endif
      auto chain = [&](auto nt) {                             // nt: the trips of this chain - a constant when the state fills the tile width
#pragma unroll 1
        for (int g = 0; g < nt; ++g) {
          __builtin_amdgcn_sched_barrier(0);
          const CH q0 = MI_LDW(wq[128 + wl], r0), q1 = MI_LDW(wq[192 + wl], r1);
          const CH a0 = MI_LDA(ap + (4 * g) * VEC);
          const CH a1 = MI_LDA(ap + (4 * g + 1) * VEC);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int v = 0; v < VEC; ++v) c0 = TR::mfma(a0.v[v], r0.v[v], c0);
#pragma unroll
          for (int v = 0; v < VEC; ++v) c0 = TR::mfma(a1.v[v], r1.v[v], c0);
          const CH a2 = MI_LDA(ap + (4 * g + 2) * VEC);
          const CH a3 = MI_LDA(ap + (4 * g + 3) * VEC);
          wq = (g + 1 == nt) ? wp : wq + 256;
          __builtin_amdgcn_sched_barrier(0);
          r0 = MI_LDW(wq[wl], r0); r1 = MI_LDW(wq[64 + wl], r1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int v = 0; v < VEC; ++v) c0 = TR::mfma(a2.v[v], q0.v[v], c0);
#pragma unroll
          for (int v = 0; v < VEC; ++v) c0 = TR::mfma(a3.v[v], q1.v[v], c0);
        }
      };
      if (active) {                                           // (a wavefront without columns never loaded r0 / r1: it must not run a chain)
        if (trips == NCH / 4) chain(std::integral_constant<int, NCH / 4>{});    // (the constant trip count is worth 4 % at dim 256: 5.53 against 5.78 ms)
        else chain(trips);
      }
    } else {
#pragma unroll
      for (int m = 0; m < KS / VEC; ++m) {
        const CH a0 = *(const CH*)(ap + m * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) c0 = TR::mfma(a0.v[v], bf[m * VEC + v], c0);
      }
    }
#endif
    stamp();
    if (plain) {
#pragma unroll
      for (int i = 0; i < 4; ++i) kn[i] = c0[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        T k_ = c0[i];
        if (has_bias) k_ = k_ + bias_v;
        kn[i] = sign * k_;
      }
    }
    stamp();
  }
};

// W [dim, dim] (row major) -> the copy the STREAM kernels read (LinCtx<T, D>::wp): piece (w, m, lane) holds
// W[k = (lane >> 4) ks + m VEC + v][16 w + (lane & 15)], v < VEC, ks = 4 VEC ceil(dim / (16 VEC)); zero beyond dim and for m VEC >= ks.  Launched on the stream in front of every
// kernel of the streamed family (the caller may have updated W in place since the last call).
template <typename T, int D>
__global__ __launch_bounds__(256) void k_lin_pack(const T* W, int dim, T* pack) {
  constexpr int VEC = MfmaTraits<T>::VEC, KS = D / 4, NCH = KS / VEC;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < D * D; e += gridDim.x * blockDim.x) {
    const int v = e % VEC, lane = (e / VEC) % 64, m = (e / (VEC * 64)) % NCH, w = e / (VEC * 64 * NCH);
    const int ks = 4 * VEC * ((dim + 16 * VEC - 1) / (16 * VEC));
    const int k = (lane >> 4) * ks + m * VEC + v, c = 16 * w + (lane & 15);
    pack[e] = (m * VEC < ks && k < dim && c < dim) ? W[(long long)k * dim + c] : (T)0;
  }
}

// dt * coefficient products of an attempt, staged in LDS once per pass: (T)dt * (T)c, the very products step_combine /
// step_finish form.  Rows of beta back to back (row s at s(s-1)/2, s = 1..S), then c_error, then c_mid.  Reading one back is
// an LDS instruction; forming it in place is a v_mul_f64 plus the scalar-register traffic of the tableau (the ~35 entries do
// not fit the SGPR file next to everything else: v_readlane reloads) per use and tile - and on this part every vector
// instruction of either wavefront of a SIMD takes its issue time away from the matrix pipe (scripts/micro/mfma_pair.hip).
template <int S>
struct LinCoef {
  static constexpr int kErr = S * (S + 1) / 2, kMid = kErr + S + 1, kCount = kMid + S + 1;
  static constexpr int row(int s) { return s * (s - 1) / 2; }
};
constexpr int kLinCoefMax = 128;                             // >= LinCoef<13>::kCount = 119

template <typename T, int S>
__device__ __forceinline__ void lin_fill_coef(const StepArgs& A, T hs, T* coef) {
  using CF = LinCoef<S>;
  for (int q = threadIdx.x; q < CF::kCount; q += blockDim.x) {
    double c;
    if (q < CF::kErr) {
      int s_ = 1;
      while (CF::row(s_ + 1) <= q) ++s_;
      c = A.beta[s_ - 1][q - CF::row(s_)];
    } else if (q < CF::kMid) c = A.e[q - CF::kErr];
    else c = A.cmid[q - CF::kMid];
    coef[q] = hs * (T)c;
  }
}

// SC0: read the streamed state with workgroup-scope (sc0, L1-bypassing) loads - needed when the kernel outlives an
// attempt (whole-integration kernel: a plane is rewritten and re-read inside one launch)
template <bool SC0, typename T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr (SC0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else return *p;
}

// (Round 4, measured and removed: loading a workgroup's first tile of the NEXT pass before it enters the grid hand-off.  The
// hand-off's barriers carry a vmcnt(0) - the loads were simply waited for there: hand-offs 40 -> 57 us per call, passes unchanged.)
// One adaptive attempt over this workgroup's tiles (tile = blockIdx.x, + gridDim.x, ...).
// `blk` / `nblk`: this workgroup's index and the number of workgroups that share the pass (default: the whole grid; the adjoint
// kernel of the linear system runs two such passes side by side, one per half of its grid).
template <typename T, int D, int S, bool TS, bool SC0>
__device__ __forceinline__ void lin_attempt_pass(const StepArgs& A, const StepPlanes<T, S>& P, LinCtx<T, D>& cx, Acc& acc,
                                                 const double* t_out, T* coef /* LDS, kLinCoefMax */, int blk = (int)blockIdx.x,
                                                 int nblk = (int)gridDim.x) {
  constexpr int R_ = LinCtx<T, D>::R_;
  using CF = LinCoef<S>;
  const long long ntiles = (A.batch + R_ - 1) / R_;
  lin_fill_coef<T, S>(A, P.hs, coef);
  lds_barrier();
  constexpr bool PF = LinCtx<T, D>::PF;                      // (streamed W: no registers for the prefetches - read at the point of use)
  T y0n[4], f0n[4];                                          // prefetched next tile (accumulator layout)
  auto fetch = [&](long long t_i) {
    const T* ty = P.y0 + t_i * R_ * cx.d;
    const T* tf = P.f0 + t_i * R_ * cx.d;
    const int nr = cx.rows_here(t_i, A.batch);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = cx.row_of(i) < nr && cx.colok && !(MI_ABL & 8);
      y0n[i] = ok ? stream_load<SC0>(ty + cx.off_of(i)) : (T)0;
      f0n[i] = ok ? stream_load<SC0>(tf + cx.off_of(i)) : (T)0;
    }
  };
  if (PF && (long long)blk < ntiles) fetch(blk);
  const T c_first = coef[CF::row(1)];                        // dt * beta_{1,0}: the same for every tile
  T cnx[PF ? S + 1 : 1];                                     // the coefficients of the NEXT combination (read under the MFMA chain)

  for (long long tile_i = blk; tile_i < ntiles; tile_i += nblk) {
    const long long row0 = tile_i * R_;
    T y0e[4], k[S + 1][4], ys[4];
    if constexpr (!PF) fetch(tile_i);
#pragma unroll
    for (int i = 0; i < 4; ++i) { y0e[i] = y0n[i]; k[0][i] = f0n[i]; }
    if (PF && tile_i + nblk < ntiles) fetch(tile_i + nblk);

    auto stage = [&](auto sg_c) {
      constexpr int SG = decltype(sg_c)::value;
      T cb[SG];                                              // dt * beta_{SG, j}: step_combine's products, from the table
#pragma unroll
      for (int j = 0; j < SG; ++j) cb[j] = (SG == 1) ? c_first : (PF ? cnx[PF ? j : 0] : coef[CF::row(SG) + j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        T a_ = cb[0] * k[0][i];                              // misc._scaled_dot_product order (rk_common.py:51)
#pragma unroll
        for (int j = 1; j < SG; ++j) a_ = lin_madd(cb[j], k[j][i], a_);
        ys[i] = y0e[i] + a_;
      }
      cx.rhs_eval(ys, k[SG], [&] {                           // next: row SG + 1 of beta, after the last stage c_error
        if constexpr (PF) {
#pragma unroll
          for (int j = 0; j <= SG; ++j) cnx[j] = coef[(SG < S ? CF::row(SG + 1) : CF::kErr) + j];
        }
      });
    };
    for_stages<1, S>(stage);
    const bool need_mid = !TS && P.j_hi > P.j_lo;            // (wave-uniform: an output time falls into this attempt)
    T err4[4], ym4[4];                                       // rk_common.py:60 / dopri5.py:42: step_finish's operations, one table
#pragma unroll                                               // entry at a time (the coefficients are vector registers now)
    for (int j = 0; j <= S; ++j) {
      const T ce = PF ? cnx[PF ? j : 0] : coef[CF::kErr + j];
#pragma unroll
      for (int i = 0; i < 4; ++i) err4[i] = (j == 0) ? ce * k[0][i] : lin_madd(ce, k[j][i], err4[i]);
    }
    if (need_mid) {
#pragma unroll
      for (int j = 0; j <= S; ++j) {
        const T cm = coef[CF::kMid + j];
#pragma unroll
        for (int i = 0; i < 4; ++i) ym4[i] = (j == 0) ? cm * k[0][i] : ym4[i] + cm * k[j][i];   // (two roundings: k_emit's y_mid)
      }
    }
    const int nr = cx.rows_here(tile_i, A.batch);
    T* ty1 = P.y1 + row0 * cx.d;
    T* tf1 = P.f1 + row0 * cx.d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (cx.row_of(i) < nr && cx.colok) {
        T kk[S + 1];
#pragma unroll
        for (int j = 0; j <= S; ++j) kk[j] = k[j][i];
        const T err = err4[i];
        const T ymid = need_mid ? y0e[i] + ym4[i] : y0e[i];
        const long long idx = row0 * cx.d + cx.off_of(i);
        if (!(MI_ABL & 8) || err == (T)123.456) {
          ty1[cx.off_of(i)] = ys[i];
          tf1[cx.off_of(i)] = k[S][i];
        }
        step_emit<T, S, TS>(A, P, y0e[i], ys[i], kk, ymid, idx, t_out);
        acc.maxa = fmax(acc.maxa, (double)fabs(y0e[i]));
        acc.maxb = fmax(acc.maxb, (double)fabs(ys[i]));
        acc.suma += (double)err * (double)err;
      }
    }
  }
}

// before_integrate, first half (dopri5.py:71 + misc.py:225-233): f0 = f(t0, y0), sums of (y0/sc)^2 and (f0/sc)^2, the
// non-finite flag; optionally seeds a state plane and solution[0] with y0 in the same pass.
// DOT (the adjoint kernel of the linear system): also sum f0 . gdot over the tiles into *dot - the time gradient of adjoint.py:134-140
// needs f(t_i, y_i) . grad_output_i, and f(t_i, y_i) is this pass's f0.
template <typename T, int D, bool SC0, bool DOT = false>
__device__ __forceinline__ void lin_f0_pass(const StepArgs& A, const T* y0, T* f0_out, T* copy_a, T* copy_b, LinCtx<T, D>& cx,
                                            Acc& acc, int blk = (int)blockIdx.x, int nblk = (int)gridDim.x, const T* gdot = nullptr,
                                            double* dot = nullptr) {
  constexpr int R_ = LinCtx<T, D>::R_;
  const long long ntiles = (A.batch + R_ - 1) / R_;
  T y0n[4], gn[DOT ? 4 : 1];
  auto fetch = [&](long long t_i) {
    const T* ty = y0 + t_i * R_ * cx.d;
    const int nr = cx.rows_here(t_i, A.batch);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = cx.row_of(i) < nr && cx.colok;
      y0n[i] = ok ? stream_load<SC0>(ty + cx.off_of(i)) : (T)0;
      if constexpr (DOT) gn[i] = ok ? (gdot + t_i * R_ * cx.d)[cx.off_of(i)] : (T)0;
    }
  };
  constexpr bool PF = LinCtx<T, D>::PF;
  if (PF && (long long)blk < ntiles) fetch(blk);
  for (long long tile_i = blk; tile_i < ntiles; tile_i += nblk) {
    T y0e[4], kn[4], ge[DOT ? 4 : 1];
    if constexpr (!PF) fetch(tile_i);
#pragma unroll
    for (int i = 0; i < 4; ++i) { y0e[i] = y0n[i]; if constexpr (DOT) ge[i] = gn[i]; }
    if (PF && tile_i + nblk < ntiles) fetch(tile_i + nblk);
    cx.rhs_eval(y0e, kn);
    const int nr = cx.rows_here(tile_i, A.batch);
    const long long tb = tile_i * R_ * cx.d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (cx.row_of(i) < nr && cx.colok) {
        (f0_out + tb)[cx.off_of(i)] = kn[i];
        if (copy_a != nullptr) (copy_a + tb)[cx.off_of(i)] = y0e[i];
        if (copy_b != nullptr) (copy_b + tb)[cx.off_of(i)] = y0e[i];
        const T sc = (T)A.cp.atol + fabs(y0e[i]) * (T)A.cp.rtol;     // misc.py:225
        const double q0 = (double)(y0e[i] / sc);
        acc.suma += q0 * q0;
        if (!finite_(y0e[i])) acc.flag = 1;
        const double q1 = (double)(kn[i] / sc);
        acc.sumb += q1 * q1;                                         // misc.py:228
        if constexpr (DOT) *dot += (double)kn[i] * (double)ge[i];
      }
    }
  }
}

// before_integrate, second half (misc.py:235-237): f1 = f(t0 + h0, y0 + h0 f0), sum of ((f1 - f0)/sc)^2
template <typename T, int D, bool SC0>
__device__ __forceinline__ void lin_initb_pass(const StepArgs& A, const T* y0, const T* f0, T h0, LinCtx<T, D>& cx, Acc& acc,
                                               int blk = (int)blockIdx.x, int nblk = (int)gridDim.x) {
  constexpr int R_ = LinCtx<T, D>::R_;
  const long long ntiles = (A.batch + R_ - 1) / R_;
  T y0n[4], f0n[4];
  auto fetch = [&](long long t_i) {
    const T* ty = y0 + t_i * R_ * cx.d;
    const T* tf = f0 + t_i * R_ * cx.d;
    const int nr = cx.rows_here(t_i, A.batch);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = cx.row_of(i) < nr && cx.colok;
      y0n[i] = ok ? stream_load<SC0>(ty + cx.off_of(i)) : (T)0;
      f0n[i] = ok ? stream_load<SC0>(tf + cx.off_of(i)) : (T)0;
    }
  };
  if ((long long)blk < ntiles) fetch(blk);
  for (long long tile_i = blk; tile_i < ntiles; tile_i += nblk) {
    T y0e[4], f0e[4], ys[4], kn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { y0e[i] = y0n[i]; f0e[i] = f0n[i]; ys[i] = y0e[i] + h0 * f0e[i]; }   // misc.py:235
    if (tile_i + nblk < ntiles) fetch(tile_i + nblk);
    cx.rhs_eval(ys, kn);
    const int nr = cx.rows_here(tile_i, A.batch);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (cx.row_of(i) < nr && cx.colok) {
        const T sc = (T)A.cp.atol + fabs(y0e[i]) * (T)A.cp.rtol;
        const double q = (double)((kn[i] - f0e[i]) / sc);            // misc.py:237
        acc.suma += q * q;
      }
    }
  }
}

template <typename T, int D, int S, bool TS>
__global__ __launch_bounds__(D * 4) void k_step_linear_mfma(StepArgs A) {
  StepPlanes<T, S> P;
  if (!resolve_step<T, S>(A, P)) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* s_ys = (T*)smem_raw;
  double* red = (double*)(s_ys + LinCtx<T, D>::kBufs * LinCtx<T, D>::TILE);
  LinCtx<T, D> cx;
  cx.init(A.rhs, s_ys, A.dim);
  Acc acc;
  lin_attempt_pass<T, D, S, TS, false>(A, P, cx, acc, A.t_out, (T*)(red + 80));
  finish_attempt(A, acc, red);
}

// before_integrate for the launch-per-attempt schedule: PHASE 0 = f0 pass (reads the caller's y0, seeds the state plane
// and solution[0]), PHASE 1 = second half of misc._select_initial_step.  Each is followed by k_controller.
struct InitArgs {
  StepArgs s;
  const void* y0;              // PHASE 0: the caller's initial state
  void* copy_b;                // PHASE 0: solution[0] (may be null)
};

template <typename T, int D, int PHASE>
__global__ __launch_bounds__(D * 4) void k_init_linear_mfma(InitArgs I) {
  const StepArgs& A = I.s;
  const Ctl* c = A.ctl;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* s_ys = (T*)smem_raw;
  double* red = (double*)(s_ys + LinCtx<T, D>::kBufs * LinCtx<T, D>::TILE);
  LinCtx<T, D> cx;
  cx.init(A.rhs, s_ys, A.dim);
  Acc acc;
  T* plane_y = (T*)(A.planes + (long long)c->idx_y0 * A.stride);
  T* plane_f = (T*)(A.planes + (long long)c->idx_k[0] * A.stride);
  if constexpr (PHASE == 0) lin_f0_pass<T, D, false>(A, (const T*)I.y0, plane_f, plane_y, (T*)I.copy_b, cx, acc);
  else lin_initb_pass<T, D, false>(A, plane_y, plane_f, (T)c->h0, cx, acc);
  block_reduce_store<false>(acc, red, A.partials + (long long)blockIdx.x * kRec);
}

// Fixed grid (solvers.py:82-104) for the linear RHS, dim in {16, 32, 64, 128}: trajectories never interact on a fixed
// grid, so a tile runs through EVERY grid interval with y and k_1..k_4 in registers (Euler: fixed_grid.py:6-7, RK4
// 3/8 rule: rk_common.py:73-81, the arithmetic of the FX_* stage modes) and streams solution[i+1]; one launch per call.
// Traffic = y0 in + T solution rows out; bound: fp64/fp32 MFMA.  Only W (64 VGPRs at D = 128) and a handful of state
// registers are live.  (Forcing two workgroups per CU with a 128-VGPR cap brought nothing: the limit is not phase
// serialisation - see DESIGN.md, "what limits the tile kernels".)
template <typename T, int D>
__global__ __launch_bounds__(D * 4) void k_fixed_linear_mfma(FixedArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LinCtx<T, D> cx;
  cx.init(A.rhs, (T*)smem_raw, A.dim);
  constexpr int R_ = LinCtx<T, D>::R_;
  const long long ntiles = (A.batch + R_ - 1) / R_;
  const long long n = A.batch * A.dim;
  const T* y0p = (const T*)A.y0;
  T* out = (T*)A.out;
  FixedClk clk;
  clk.begin();
  for (long long tile_i = blockIdx.x; tile_i < ntiles; tile_i += gridDim.x) {
    T y[4];
    long long idx[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long row = tile_i * R_ + cx.row_of(i);
      ok[i] = row < A.batch && cx.colok;
      idx[i] = row * cx.d + cx.col;
      y[i] = ok[i] ? y0p[idx[i]] : (T)0;
      if (ok[i]) out[idx[i]] = y[i];                          // solution = [y0]
    }
    int j = 1;
    for (int s = 0; s < A.M; ++s) {
      const T t0 = (T)A.grid[s];                              // solvers.py:84: the grid is cast to the STATE dtype
      const T t1 = (T)A.grid[s + 1];
      const T dt = t1 - t0;                                   // (the linear RHS does not depend on t: eps has no effect here)
      T k1[4], k2[4], k3[4], k4[4], ys[4], yn[4];
      cx.rhs_eval(y, k1);
      if (!A.rk4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) yn[i] = y[i] + dt * k1[i];                          // fixed_grid.py:7
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ys[i] = y[i] + dt * k1[i] / (T)3;                   // rk_common.py:77
        cx.rhs_eval(ys, k2);
#pragma unroll
        for (int i = 0; i < 4; ++i) ys[i] = y[i] + dt * (k1[i] / (T)-3 + k2[i]);        // :78
        cx.rhs_eval(ys, k3);
#pragma unroll
        for (int i = 0; i < 4; ++i) ys[i] = y[i] + dt * (k1[i] - k2[i] + k3[i]);        // :79
        cx.rhs_eval(ys, k4);
#pragma unroll
        for (int i = 0; i < 4; ++i) yn[i] = y[i] + (k1[i] + (T)3 * k2[i] + (T)3 * k3[i] + k4[i]) * (dt / (T)8);   // :81
      }
      while (j < A.T && t1 >= (T)A.t[j]) {                   // solvers.py:97-100, _linear_interp :106-115
        const T tj = (T)A.t[j];
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

// row stride of the stage tiles in LDS (= LinCtx<T, D>::LD) for a tile width known at run time
template <typename T>
constexpr size_t lin_ld(int D) { return (size_t)D + (size_t)MfmaTraits<T>::VEC; }
template <typename T, int D>
constexpr size_t step_linear_lds_bytes() {
  return (size_t)2 * 16 * lin_ld<T>(D) * sizeof(T) + (80 + kLinCoefMax) * sizeof(double);     // two stage tiles | red | coefficient table
}

}  // namespace mi
