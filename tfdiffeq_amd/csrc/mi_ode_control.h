// Device-side step controller, partial-record reduction and dense-output emission.
// One tiny single-workgroup kernel per attempt replaces the ~10 host synchronisations per attempt
// of the reference's Python controller.
#pragma once
#include "mi_ode_dev.h"
#include "mi_ode_dense.h"

namespace mi {

// Reduce per-block records (fixed order: thread i takes blocks i, i+256, ...; then an LDS tree).
// All 256 threads must call it; the result is valid in thread 0.
__device__ __forceinline__ void reduce_block_records(const double* part, int nblocks, double* out /*[kRec]*/) {
  __shared__ double s[5][256];
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const double* p = part + (long long)b * kRec;
    v0 = fmax(v0, p[R_MAXA]); v1 = fmax(v1, p[R_MAXB]); v2 += p[R_SUMA]; v3 += p[R_SUMB]; v4 = fmax(v4, p[R_FLAG]);
  }
  s[0][threadIdx.x] = v0; s[1][threadIdx.x] = v1; s[2][threadIdx.x] = v2; s[3][threadIdx.x] = v3; s[4][threadIdx.x] = v4;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s[0][threadIdx.x] = fmax(s[0][threadIdx.x], s[0][threadIdx.x + off]);
      s[1][threadIdx.x] = fmax(s[1][threadIdx.x], s[1][threadIdx.x + off]);
      s[2][threadIdx.x] += s[2][threadIdx.x + off];
      s[3][threadIdx.x] += s[3][threadIdx.x + off];
      s[4][threadIdx.x] = fmax(s[4][threadIdx.x], s[4][threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[R_MAXA] = s[0][0]; out[R_MAXB] = s[1][0]; out[R_SUMA] = s[2][0]; out[R_SUMB] = s[3][0]; out[R_FLAG] = s[4][0];
    out[R_N] = 0; out[6] = 0; out[7] = 0;
  }
}

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

// NaN-propagating min / max, as tf.reduce_min / tf.reduce_max (and numpy) behave
__device__ __forceinline__ double nan_min(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : fmin(a, b); }
__device__ __forceinline__ double nan_max(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : fmax(a, b); }

// misc._optimal_step_size (misc.py:267-287) / tsit5._optimal_step_size (tsit5.py:53-62)
__device__ __forceinline__ double optimal_step(double last_step, double ratio, const CtrlParams& P) {
  if (ratio == 0.0) return last_step * P.ifactor;
  const double dfac = (ratio < 1.0) ? 1.0 : P.dfactor;
  double er, expo;
  if (P.controller == MI_ODE_CTRL_TSIT5) {
    er = ratio;                                            // no sqrt (tsit5.py:59)
    expo = 1.0 / (double)P.order;                          // true float64 exponent (tsit5.py:60)
  } else {
    er = P.is_f32 ? (double)sqrtf((float)ratio) : sqrt(ratio);   // sqrt in the ratio's dtype (misc.py:277-278)
    expo = (double)(float)(1.0 / (double)P.order);         // float32 detour, F4 (misc.py:281-282)
  }
  const double factor = nan_max(1.0 / P.ifactor, nan_min(pow(er, expo) / P.safety, 1.0 / dfac));
  return last_step / factor;
}

// The controller: combines the rank records (rank order => G-independent rounding), then applies the
// phase logic with exactly the reference's scalar arithmetic.  recs == nullptr: single rank, reduce
// this rank's block records here (saves one launch per attempt).
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
  const double N = rec[R_N];

  if (phase == PH_F0) {                                    // misc.py:227-233
    c->nfe += 1;
    c->y0_nonfinite = rec[R_FLAG] != 0.0;
    double d0, d1, h0;
    if (P.is_f32) {
      const float f0 = sqrtf((float)rec[R_SUMA]) / powf((float)N, 0.5f);
      const float f1 = sqrtf((float)rec[R_SUMB]) / powf((float)N, 0.5f);
      const float h = (f0 < 1e-5f || f1 < 1e-5f) ? 1e-6f : 0.01f * (f0 / f1);
      d0 = f0; d1 = f1; h0 = h;
    } else {
      d0 = sqrt(rec[R_SUMA]) / pow(N, 0.5);
      d1 = sqrt(rec[R_SUMB]) / pow(N, 0.5);
      h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
    }
    c->d0 = d0; c->d1 = d1; c->h0 = h0;
    return;
  }
  if (phase == PH_INITB) {                                 // misc.py:236-245
    c->nfe += 1;
    double first;
    if (P.is_f32) {
      const float h0 = (float)c->h0, d1 = (float)c->d1;
      const float d2 = (sqrtf((float)rec[R_SUMA]) / powf((float)N, 0.5f)) / h0;
      float h1;
      if (d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
      else h1 = powf(0.01f / fmaxf(d1, d2), (float)(1.0 / (double)(P.init_order + 1)));
      first = (double)fminf(100.0f * h0, h1);
    } else {
      const double h0 = c->h0, d1 = c->d1;
      const double d2 = (sqrt(rec[R_SUMA]) / pow(N, 0.5)) / h0;
      double h1;
      if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
      else h1 = pow(0.01 / fmax(d1, d2), 1.0 / (double)(P.init_order + 1));
      first = fmin(100.0 * h0, h1);
    }
    c->dt = first;
    return;
  }

  // ---- PH_ATTEMPT: dopri5.py:103-121 -------------------------------------------------------
  c->n_attempt += 1;
  c->nfe += P.n_stages;
  c->n_steps_out += 1;
  const double dt = c->dt, t_start = c->t1;
  double ratio;
  if (P.is_f32) {                                          // misc.py:256-263 in the state dtype
    const float tol = (float)P.atol + (float)P.rtol * (float)fmax(rec[R_MAXA], rec[R_MAXB]);
    ratio = (double)(float)(rec[R_SUMA] / (N * (double)tol * (double)tol));
  } else {
    const double tol = P.atol + P.rtol * fmax(rec[R_MAXA], rec[R_MAXB]);
    ratio = rec[R_SUMA] / (N * tol * tol);
  }
  c->ratio = ratio;
  const bool accept = ratio <= 1.0;                        // NaN -> rejected (dopri5.py:108)
  const double dt_next = optimal_step(dt, ratio, P);
  c->accepted = accept ? 1 : 0;
  c->emit_lo = c->emit_hi = c->next_out;
  if (accept) {
    c->n_accept += 1;
    const double t_new = t_start + dt;
    // the step's planes, for dense output
    c->emit_y0 = c->idx_y0; c->emit_y1 = c->idx_y1;
    for (int j = 0; j <= P.n_stages; ++j) c->emit_k[j] = c->idx_k[j];
    c->emit_t0 = t_start; c->emit_t1 = t_new; c->emit_dt = dt;
    // rotate: y1 becomes the state, k_S (= f1, FSAL) becomes f0
    const int iy = c->idx_y0; c->idx_y0 = c->idx_y1; c->idx_y1 = iy;
    const int ik = c->idx_k[0]; c->idx_k[0] = c->idx_k[P.n_stages]; c->idx_k[P.n_stages] = ik;
    c->t0 = t_start; c->t1 = t_new;
    // outputs that fall into (t_start, t_new]   (`while next_t > t1` exits, dopri5.py:84)
    int nx = c->next_out;
    while (nx < c->n_out && !(P.t_out[nx] > t_new)) ++nx;
    if (nx > c->next_out) { c->emit_hi = nx; c->next_out = nx; c->n_steps_out = 0; }
  } else {
    c->n_reject += 1;
    c->t0 = t_start;                                       // rejected: rk_state.t0 == rk_state.t1
  }
  c->dt = dt_next;
  if (c->next_out >= c->n_out) { c->done = 1; return; }
  if (c->n_steps_out >= P.max_num_steps) { c->status |= MI_ODE_ST_MAX_STEPS; c->done = 1; return; }   // dopri5.py:85
  if (!(c->t1 + dt_next > c->t1)) { c->status |= MI_ODE_ST_DT_UNDERFLOW; c->done = 1; }               // dopri5.py:98
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
