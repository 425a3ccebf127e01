// Device-side common definitions for the MI355X (gfx950) ODE engine.
//
// Data layout in HBM: the solver state lives in a workspace of
// (2 + S + 1) contiguous "planes" of N = batch*dim elements (y_a, y_b, k_0..k_S), each
// plane row-major [batch, dim].  Which plane currently holds y0 / f0 / scratch is an
// index table inside the device-resident controller record (Ctl), rotated on accept, so
// an accepted step costs no copy.  Wave-uniform values (dt, tableau row, plane indices)
// travel through the kernarg segment / scalar loads into SGPRs, not LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mi_ode.h"

// The MFMA-linear family (rhs.Linear on the tile kernels: config 4) forms its stage / error / mid-point combinations with fused
// multiply-adds - acc = fma(dt * c_j, k_j, acc), one rounding per term where the reference's add_n((scale * c_j) * k_j) has two.
// On this part every fp64 vector instruction takes its issue time from the fp64 matrix pipe, and these
// combinations are most of the vector work of an attempt: 0 restores the reference's two roundings (measured: config 4 is 2.2 %
// slower per call, 3.2 % per attempt pass).  All schedules of the family (per-stage, whole-attempt, whole-call) follow the same
// setting and stay bit-identical to each other; the family never was bit-identical to the oracle (the matrix product's summation
// order differs), its parity bar is the north star's rtol 1e-5 / atol 1e-6 and, in the tests, 1e-12 per attempt.  Every other
// family (row-local systems, MLP, VALU fallback, plane kernels) keeps the reference's operation order to the bit.
#ifndef MI_LIN_FMA
#define MI_LIN_FMA 1
#endif

namespace mi {

// acc + c * k with two roundings, or with one (FMA)
template <bool FMA, typename T>
__device__ __forceinline__ T madd(T c, T k, T acc) {
  if constexpr (FMA) {
    if constexpr (sizeof(T) == 8) return __builtin_fma(c, k, acc);
    else return __builtin_fmaf(c, k, acc);
  } else {
    return acc + c * k;
  }
}
template <typename T>
__device__ __forceinline__ T lin_madd(T c, T k, T acc) { return madd<(MI_LIN_FMA != 0), T>(c, k, acc); }

constexpr int kMaxK = MI_ODE_MAX_K;       // stage derivative planes (S + 1)
constexpr int kNumPlanes = 2 + kMaxK;     // y_a, y_b, k_0..k_S
constexpr int kMaxBlocks = 2048;          // cap on the grid of every streaming kernel
constexpr int kRec = 8;                   // doubles per partial record
// record slots: {max_a, max_b, sum_a, sum_b, flag, n, -, -}; combine = max, max, sum, sum, max, sum
enum { R_MAXA = 0, R_MAXB = 1, R_SUMA = 2, R_SUMB = 3, R_FLAG = 4, R_N = 5 };

// Device-resident controller record: the reference's _RungeKuttaState scalars (rk_common.py:8-19)
// plus the bookkeeping that lets the host enqueue attempts without synchronising.
struct Ctl {
  double t0, t1, dt;        // rk_state.t0, rk_state.t1, rk_state.dt (next step size)
  double h0, d0, d1;        // _select_initial_step intermediates (misc.py:227-233)
  double ratio;             // last mean_sq_error_ratio
  double emit_t0, emit_t1, emit_dt;   // the accepted step whose dense output is being emitted
  long long n_attempt, n_accept, n_reject, nfe, n_steps_out;
  int idx_y0, idx_y1;       // plane of the current state / plane that receives y1
  int idx_k[kMaxK];         // idx_k[0] = f at the current state, idx_k[s] = k_{s+1} of the attempt
  int emit_y0, emit_y1, emit_k[kMaxK];
  int emit_lo, emit_hi;     // output indices [lo, hi) falling into the accepted step
  int next_out, n_out;
  int done;                 // 1: all requested outputs produced, or a status bit was raised
  unsigned status;          // MI_ODE_ST_*
  int y0_nonfinite;
  int accepted;             // result of the last attempt
  long long prof[4];        // -DMI_PERSIST_PROF: wall_clock64 ticks (10 ns) in stages / reduce+hand-off / controller / emit
  long long clk_cycles, clk_ticks;   // whole-call kernels: shader cycles and 10 ns ticks between entry and the final write-back
};

// Stage kernel flavours.  STAGE / LAST_FSAL follow rk_common.py:49-60 literally (operation
// order of misc._scaled_dot_product, no FMA contraction); F0 / INITB carry the reductions of
// misc._select_initial_step; FX_* are fixed_grid.py:6-7 and rk_common.py:73-81 literally.
enum Mode {
  M_STAGE = 0,      // ys = y0 + sum (hs*a_j) k_j ; k_out = f(ys)
  M_LAST_FSAL = 1,  // + y1 = ys ; err = sum (hs*e_j) k_j + (hs*e_NK) kn ; norms
  M_F0 = 2,         // NK = 0: k_out = f(y0) ; sums (y0/sc)^2, (f0/sc)^2 ; non-finite flag
  M_INITB = 3,      // NK = 1: ys = y0 + h0*f0 ; f1 = f(ys) ; sum ((f1-f0)/sc)^2
  M_FX_EULER = 4,   // NK = 0: y1 = y0 + dt*f(y0)
  M_FX_RK4_2 = 5,   // NK = 1: ys = y + dt*k1/3
  M_FX_RK4_3 = 6,   // NK = 2: ys = y + dt*(k1/-3 + k2)
  M_FX_RK4_4 = 7    // NK = 3: ys = y + dt*(k1-k2+k3) ; y1 = y + (k1+3*k2+3*k3+k4)*(dt/8)
};

struct RhsParams {
  double s[8];
  const void* w[3];
  const void* b[3];
  double sign;
  int hidden;
  int cube;
};

struct StageArgs {
  Ctl* ctl;                  // adaptive engine: planes/dt/t come from here
  char* planes;              // workspace base
  long long stride;          // bytes between planes
  // explicit mode (fixed grid, single-step parity surface): pointers and scalars by value
  const void* x_y0;
  void* x_y1;
  const void* x_k[kMaxK];
  void* x_kout;
  double x_t0, x_dt;
  int explicit_mode;
  int k_out_slot;            // ctl mode: idx_k slot that receives kn (-1: discard)
  long long batch;
  int dim;
  int pad_;
  double a[kMaxK];           // tableau row (beta_sigma)
  double e[kMaxK + 1];       // c_error (LAST_FSAL)
  double alpha;
  double rtol, atol;         // scale of the initial-step norms
  double* partials;          // [gridDim.x][kRec]
  void* copy_a;              // F0 only: also store y0 here (the workspace state plane) ...
  void* copy_b;              // ... and here (solution[0]); saves two plane-sized copy launches
  RhsParams rhs;
};

template <typename T>
struct Resolved {
  const T* y0;
  T* y1;
  const T* k[kMaxK];
  T* k_out;
  T hs;       // dt (or h0 for INITB), cast to the state dtype (rk_common.py:46)
  T ts;       // stage time t0 + alpha*dt in the state dtype (rk_common.py:50)
};

template <typename T, int NK, int MODE>
__device__ __forceinline__ bool resolve(const StageArgs& A, Resolved<T>& R) {
  if (A.explicit_mode) {
    R.y0 = (const T*)A.x_y0;
    R.y1 = (T*)A.x_y1;
#pragma unroll
    for (int j = 0; j < (NK > 0 ? NK : 1); ++j) R.k[j] = (const T*)A.x_k[j];
    R.k_out = (T*)A.x_kout;
    R.hs = (T)A.x_dt;
    R.ts = (T)A.x_t0 + (T)A.alpha * (T)A.x_dt;
    return true;
  }
  const Ctl* c = A.ctl;
  if (c->done) return false;
  const char* base = A.planes;
  R.y0 = (const T*)(base + (long long)c->idx_y0 * A.stride);
  R.y1 = (T*)(base + (long long)c->idx_y1 * A.stride);
#pragma unroll
  for (int j = 0; j < (NK > 0 ? NK : 1); ++j) R.k[j] = (const T*)(base + (long long)c->idx_k[j] * A.stride);
  R.k_out = A.k_out_slot >= 0 ? (T*)(base + (long long)c->idx_k[A.k_out_slot] * A.stride) : nullptr;
  const T t0 = (T)c->t1;
  const T dt = (MODE == M_INITB) ? (T)c->h0 : (T)c->dt;
  R.hs = dt;
  R.ts = t0 + (T)A.alpha * dt;
  return true;
}

// ------------------------------------------------------------------------------------------
// per-element stage math (shared by every kernel shape)
// ------------------------------------------------------------------------------------------
// ys and the auxiliary partial sum, from y0 and the NK loaded stage derivatives.
template <typename T, int NK, int MODE, bool FMA = false>
__device__ __forceinline__ T combine_elem(T y0, const T* k, T hs, const StageArgs& A, T& aux) {
  aux = (T)0;
  if constexpr (MODE == M_FX_RK4_2) {
    return y0 + hs * k[0] / (T)3;                       // rk_common.py:77
  } else if constexpr (MODE == M_FX_RK4_3) {
    return y0 + hs * (k[0] / (T)-3 + k[1]);             // rk_common.py:78
  } else if constexpr (MODE == M_FX_RK4_4) {
    aux = k[0] + (T)3 * k[1] + (T)3 * k[2];             // rk_common.py:81 (k4 added in the epilogue)
    return y0 + hs * (k[0] - k[1] + k[2]);              // rk_common.py:79
  } else if constexpr (MODE == M_INITB) {
    return y0 + hs * k[0];                              // misc.py:235
  } else if constexpr (NK == 0) {
    return y0;
  } else {
    T acc = (hs * (T)A.a[0]) * k[0];                    // misc.py:121: (scale * x) * y, summed in order
#pragma unroll
    for (int j = 1; j < NK; ++j) acc = madd<FMA, T>(hs * (T)A.a[j], k[j], acc);
    if constexpr (MODE == M_LAST_FSAL) {
      T er = (hs * (T)A.e[0]) * k[0];
#pragma unroll
      for (int j = 1; j < NK; ++j) er = madd<FMA, T>(hs * (T)A.e[j], k[j], er);
      aux = er;
    }
    return y0 + acc;                                    // rk_common.py:51
  }
}

struct Acc {        // per-thread running reductions
  double maxa = 0.0, maxb = 0.0, suma = 0.0, sumb = 0.0;
  int flag = 0;
};

template <typename T>
__device__ __forceinline__ bool finite_(T v) {
  return !(isnan(v) || isinf(v));
}

// reductions that only need flat-phase values (y0, y1)
template <typename T, int MODE>
__device__ __forceinline__ void reduce_flat(T y0, T ys, const StageArgs& A, Acc& acc) {
  if constexpr (MODE == M_LAST_FSAL) {
    acc.maxa = fmax(acc.maxa, (double)fabs(y0));
    acc.maxb = fmax(acc.maxb, (double)fabs(ys));
  } else if constexpr (MODE == M_F0) {
    const T sc = (T)A.atol + fabs(y0) * (T)A.rtol;      // misc.py:225
    const double q = (double)(y0 / sc);
    acc.suma += q * q;
    if (!finite_(y0)) acc.flag = 1;
  }
}

// epilogue once kn = f(ts, ys) is known; y0 / k0 are only read for the modes that need them.
// Returns the y1 value to store (LAST_FSAL: ys itself is stored by the flat phase).
template <typename T, int NK, int MODE, bool FMA = false>
__device__ __forceinline__ T epilogue_elem(T y0, T k0, T kn, T aux, T hs, const StageArgs& A, Acc& acc) {
  if constexpr (MODE == M_LAST_FSAL) {
    const T err = madd<FMA, T>(hs * (T)A.e[NK], kn, aux);   // rk_common.py:60, last term of the add_n
    acc.suma += (double)err * (double)err;
    return (T)0;
  } else if constexpr (MODE == M_F0) {
    const T sc = (T)A.atol + fabs(y0) * (T)A.rtol;
    const double q = (double)(kn / sc);
    acc.sumb += q * q;                                  // misc.py:228
    return (T)0;
  } else if constexpr (MODE == M_INITB) {
    const T sc = (T)A.atol + fabs(y0) * (T)A.rtol;
    const double q = (double)((kn - k0) / sc);          // misc.py:237
    acc.suma += q * q;
    return (T)0;
  } else if constexpr (MODE == M_FX_EULER) {
    return y0 + hs * kn;                                // fixed_grid.py:7 + solvers.py:95
  } else if constexpr (MODE == M_FX_RK4_4) {
    return y0 + (aux + kn) * (hs / (T)8);               // rk_common.py:81 + solvers.py:95
  } else {
    return (T)0;
  }
}

constexpr __host__ __device__ bool mode_has_reduction(int m) { return m == M_LAST_FSAL || m == M_F0 || m == M_INITB; }
constexpr __host__ __device__ bool mode_writes_y1(int m) { return m == M_LAST_FSAL || m == M_FX_EULER || m == M_FX_RK4_4; }
constexpr __host__ __device__ bool mode_needs_y0_epi(int m) {
  return m == M_F0 || m == M_INITB || m == M_FX_EULER || m == M_FX_RK4_4;
}

// ------------------------------------------------------------------------------------------
// wavefront (64 lanes) / block reductions: __shfl_down across the wave, LDS across waves,
// fixed order => deterministic per grid size (SURVEY.md section 7 "Reduction determinism")
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// Block-reduce an Acc.  `red` is >= 5*16 doubles of LDS.
// Core: after the call thread 0 holds the block's {max a, max b, sum a, sum b, flag} in r[0..4].
__device__ __forceinline__ void block_reduce_thread0(const Acc& a, double* red, double (&r)[5]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  double v0 = wave_max(a.maxa), v1 = wave_max(a.maxb), v2 = wave_sum(a.suma), v3 = wave_sum(a.sumb);
  double v4 = wave_max((double)a.flag);
  __syncthreads();
  if (lane == 0) {
    red[wave] = v0; red[16 + wave] = v1; red[32 + wave] = v2; red[48 + wave] = v3; red[64 + wave] = v4;
  }
  __syncthreads();
  r[0] = r[1] = r[2] = r[3] = r[4] = 0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < nw; ++w) {
      r[0] = fmax(r[0], red[w]); r[1] = fmax(r[1], red[16 + w]); r[2] += red[32 + w]; r[3] += red[48 + w];
      r[4] = fmax(r[4], red[64 + w]);
    }
  }
}

// ... and let thread 0 write one record.
// SC1: store the record write-through with agent-scope (sc1) stores - the producer side of the in-kernel hand-off
// to the last workgroup (no release fence needed; the caller drains vmcnt before taking its ticket).
template <bool SC1 = false>
__device__ __forceinline__ void block_reduce_store(const Acc& a, double* red, double* rec_out) {
  double r[5];
  block_reduce_thread0(a, red, r);
  if (threadIdx.x == 0) {
    if constexpr (SC1) {
      __hip_atomic_store(rec_out + R_MAXA, r[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rec_out + R_MAXB, r[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rec_out + R_SUMA, r[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rec_out + R_SUMB, r[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(rec_out + R_FLAG, r[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      rec_out[R_MAXA] = r[0]; rec_out[R_MAXB] = r[1]; rec_out[R_SUMA] = r[2]; rec_out[R_SUMB] = r[3];
      rec_out[R_FLAG] = r[4]; rec_out[R_N] = 0; rec_out[6] = 0; rec_out[7] = 0;
    }
  }
}

// ---- parameter blocks of the controller / dense-output kernels (mi_ode_control.h) ----------------
enum Phase { PH_F0 = 0, PH_INITB = 1, PH_ATTEMPT = 2 };

struct CtrlParams {
  double rtol, atol, safety, ifactor, dfactor;
  double inv_ifactor, inv_dfactor;   // 1.0 / ifactor, 1.0 / dfactor (host-computed: the same IEEE quotients)
  long long max_num_steps;
  long long n_local;          // elements held by this rank
  int order, init_order;
  int controller;             // MI_ODE_CTRL_*
  int is_f32;
  int n_stages;
  int auto_first_step;
  const double* t_out;        // device: requested output times (advance)
};

struct InterpParams {
  int kind;                   // MI_ODE_INTERP_*
  int nk;                     // S + 1
  double c_mid[MI_ODE_MAX_LINCOMB];   // up to 14 stage derivatives (dopri8) for the stateless dense output
};

}  // namespace mi
