// Device-side step controller, partial-record reduction and dense-output emission.
// One tiny single-workgroup kernel per attempt replaces the ~10 host synchronisations per attempt
// of the reference's Python controller.
#pragma once
#include "mi_ode_dev.h"
#include "mi_ode_dense.h"
#include "mi_ode_ctrl_dev.h"

namespace mi {

// multi-rank path, step 1: this rank's record (to be all-gathered by the exchange hook)
__global__ __launch_bounds__(256) void k_reduce_partials(const Ctl* ctl, const double* part, int nblocks,
                                                         long long n_local, double* rec_out) {
  if (ctl != nullptr && ctl->done) {
    if (threadIdx.x < kRec) rec_out[threadIdx.x] = 0.0;   // keep the collective well defined
    return;
  }
  __shared__ double rec[kRec];
  reduce_block_records(part, nblocks, rec);
  if (threadIdx.x == 0) {
    rec[R_N] = (double)n_local;
    for (int i = 0; i < kRec; ++i) rec_out[i] = rec[i];
  }
}

// The controller as its own launch: combines the rank records (rank order => G-independent rounding), then applies the
// phase logic.  recs == nullptr: single rank, reduce this rank's block records here.
__global__ __launch_bounds__(256) void k_controller(Ctl* c, const double* part, int nblocks, const double* recs,
                                                    int n_ranks, int phase, CtrlParams P) {
  if (c->done) return;
  __shared__ double rec[kRec];
  if (recs == nullptr) {
    reduce_block_records(part, nblocks, rec);
    if (threadIdx.x == 0) rec[R_N] = (double)P.n_local;
  } else if (threadIdx.x == 0) {
    double m0 = 0, m1 = 0, s0 = 0, s1 = 0, fl = 0, n = 0;
    for (int r = 0; r < n_ranks; ++r) {
      const double* p = recs + (long long)r * kRec;
      m0 = fmax(m0, p[R_MAXA]); m1 = fmax(m1, p[R_MAXB]); s0 += p[R_SUMA]; s1 += p[R_SUMB];
      fl = fmax(fl, p[R_FLAG]); n += p[R_N];
    }
    rec[R_MAXA] = m0; rec[R_MAXB] = m1; rec[R_SUMA] = s0; rec[R_SUMB] = s1; rec[R_FLAG] = fl; rec[R_N] = n;
  }
  if (threadIdx.x != 0) return;
  controller_apply(c, rec, phase, P);
}

// advance(): arm the output cursor (one thread)
__global__ void k_set_outputs(Ctl* c, int n_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  c->next_out = 0; c->n_out = n_out; c->n_steps_out = 0; c->emit_lo = c->emit_hi = 0;
  c->done = 0;
  if (c->status != 0) { c->done = 1; return; }
  if (n_out <= 0) { c->done = 1; return; }
  if (c->y0_nonfinite && c->n_attempt == 0) { c->status |= MI_ODE_ST_NONFINITE; c->done = 1; return; }   // dopri5.py:99-100
  if (!(c->t1 + c->dt > c->t1)) { c->status |= MI_ODE_ST_DT_UNDERFLOW; c->done = 1; }
}

// ------------------------------------------------------------------------------------------------
// dense output
// ------------------------------------------------------------------------------------------------
// Engine emission: all output times in [emit_lo, emit_hi) for the step just accepted.
template <typename T, int NK>
__global__ __launch_bounds__(256) void k_emit(const Ctl* c, const char* planes, long long stride, long long n,
                                              const double* t_out, T* out, InterpParams I) {
  const int lo = c->emit_lo, hi = c->emit_hi;
  if (hi <= lo || c->accepted == 0) return;
  const T* y0p = (const T*)(planes + (long long)c->emit_y0 * stride);
  const T* y1p = (const T*)(planes + (long long)c->emit_y1 * stride);
  const T* kp[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) kp[j] = (const T*)(planes + (long long)c->emit_k[j] * stride);
  const double t0 = c->emit_t0, t1 = c->emit_t1;
  const T dtT = (T)c->emit_dt;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    T k[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) k[j] = kp[j][i];
    const T y0 = y0p[i], y1 = y1p[i];
    if (I.kind == MI_ODE_INTERP_QUARTIC_MID) {
      T co[5];
      quartic_fit<T, NK>(y0, y1, k, dtT, I, co);
      for (int j = lo; j < hi; ++j) out[(long long)j * n + i] = quartic_eval<T>(co, interp_x<T>(t0, t1, t_out[j]));
    } else {
      for (int j = lo; j < hi; ++j) out[(long long)j * n + i] = tsit5_dense<T, NK>(y0, k, t0, t1, t_out[j], I.kind);
    }
  }
}

// Stateless dense output at one time (mi_ode_interp_eval): explicit pointers.
struct InterpPtrs {
  const void* y0;
  const void* y1;
  const void* k[MI_ODE_MAX_LINCOMB];
};
template <typename T, int NK>
__global__ __launch_bounds__(256) void k_interp_eval(InterpPtrs P, long long n, double t0, double t1, double t,
                                                     double dt, T* out, InterpParams I) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    T k[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) k[j] = ((const T*)P.k[j])[i];
    const T y0 = ((const T*)P.y0)[i];
    if (I.kind == MI_ODE_INTERP_QUARTIC_MID) {
      const T y1 = ((const T*)P.y1)[i];
      T co[5];
      quartic_fit<T, NK>(y0, y1, k, (T)dt, I, co);   // the step's own dt (dopri5.py:41)
      out[i] = quartic_eval<T>(co, interp_x<T>(t0, t1, t));
    } else {
      out[i] = tsit5_dense<T, NK>(y0, k, t0, t1, t, I.kind);
    }
  }
}

}  // namespace mi
