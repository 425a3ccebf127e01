// Device-side controller arithmetic shared by k_controller (own launch) and the whole-attempt kernels
// (last-workgroup-done epilogue): block-record reduction in a FIXED order and the reference's scalar logic.
#pragma once
#include "mi_ode_dev.h"

namespace mi {

// Reduce per-block records in a fixed order (same bits for every caller).  The result is valid in thread 0 (which
// is also the only thread that reads it afterwards).  SC1: read the records with agent-scope (sc1) loads - the
// consumer side of the write-through hand-off used by the whole-attempt kernels' last workgroup.
template <bool SC1>
__device__ __forceinline__ double rec_load(const double* p) {
  if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

template <bool SC1 = false>
__device__ __forceinline__ void reduce_block_records(const double* part, int nblocks, double* out /*[kRec]*/) {
  // ONE wavefront does it: lane l folds records l, l+64, ... in order, then a fixed-order __shfl_down tree.  No LDS,
  // no workgroup barriers on the per-attempt critical path; the order does not depend on the caller's workgroup size.
  if (threadIdx.x < 64) {
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
      const double* p = part + (long long)b * kRec;
      v0 = fmax(v0, rec_load<SC1>(p + R_MAXA)); v1 = fmax(v1, rec_load<SC1>(p + R_MAXB));
      v2 += rec_load<SC1>(p + R_SUMA); v3 += rec_load<SC1>(p + R_SUMB); v4 = fmax(v4, rec_load<SC1>(p + R_FLAG));
    }
    v0 = wave_max(v0); v1 = wave_max(v1); v2 = wave_sum(v2); v3 = wave_sum(v3); v4 = wave_max(v4);
    if (threadIdx.x == 0) {
      out[R_MAXA] = v0; out[R_MAXB] = v1; out[R_SUMA] = v2; out[R_SUMB] = v3; out[R_FLAG] = v4;
      out[R_N] = 0; out[6] = 0; out[7] = 0;
    }
  }
}

// NaN-propagating min / max, as tf.reduce_min / tf.reduce_max (and numpy) behave
__device__ __forceinline__ double nan_min(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : fmin(a, b); }
__device__ __forceinline__ double nan_max(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : fmax(a, b); }

// misc._optimal_step_size (misc.py:267-287) / tsit5._optimal_step_size (tsit5.py:53-62)
__device__ __forceinline__ double optimal_step(double last_step, double ratio, const CtrlParams& P) {
  if (ratio == 0.0) return last_step * P.ifactor;
  const double inv_dfac = (ratio < 1.0) ? 1.0 : P.inv_dfactor;      // 1 / dfactor, dfactor = 1 when ratio < 1 (misc.py:275-276)
  double er, expo;
  if (P.controller == MI_ODE_CTRL_TSIT5) {
    er = ratio;                                            // no sqrt (tsit5.py:59)
    expo = 1.0 / (double)P.order;                          // true float64 exponent (tsit5.py:60)
  } else {
    er = P.is_f32 ? (double)sqrtf((float)ratio) : sqrt(ratio);   // sqrt in the ratio's dtype (misc.py:277-278)
    expo = (double)(float)(1.0 / (double)P.order);         // float32 detour, F4 (misc.py:281-282)
  }
  const double factor = nan_max(P.inv_ifactor, nan_min(pow(er, expo) / P.safety, inv_dfac));
  return last_step / factor;
}

// The scalars one attempt reads and writes (a plain value type: a caller that keeps it in registers pays no memory
// traffic for the controller - the whole-integration kernel does).
struct AttemptState {
  double t0, t1, dt, ratio, emit_t0, emit_t1, emit_dt;
  long long n_attempt, n_accept, n_reject, nfe, n_steps_out;
  int next_out, n_out, emit_lo, emit_hi, done, accepted;
  unsigned status;
  __device__ __forceinline__ void load(const Ctl& c) {
    t0 = c.t0; t1 = c.t1; dt = c.dt; ratio = c.ratio; emit_t0 = c.emit_t0; emit_t1 = c.emit_t1; emit_dt = c.emit_dt;
    n_attempt = c.n_attempt; n_accept = c.n_accept; n_reject = c.n_reject; nfe = c.nfe; n_steps_out = c.n_steps_out;
    next_out = c.next_out; n_out = c.n_out; emit_lo = c.emit_lo; emit_hi = c.emit_hi; done = c.done; accepted = c.accepted;
    status = c.status;
  }
  __device__ __forceinline__ void store(Ctl& c) const {
    c.t0 = t0; c.t1 = t1; c.dt = dt; c.ratio = ratio; c.emit_t0 = emit_t0; c.emit_t1 = emit_t1; c.emit_dt = emit_dt;
    c.n_attempt = n_attempt; c.n_accept = n_accept; c.n_reject = n_reject; c.nfe = nfe; c.n_steps_out = n_steps_out;
    c.next_out = next_out; c.n_out = n_out; c.emit_lo = emit_lo; c.emit_hi = emit_hi; c.done = done; c.accepted = accepted;
    c.status = status;
  }
};

// misc._compute_error_ratio (misc.py:256-263) of ONE state component from its combined record, in the state dtype
__device__ __forceinline__ double error_ratio(const double* rec, const CtrlParams& P) {
  const double N = rec[R_N];
  if (P.is_f32) {
    const float tol = (float)P.atol + (float)P.rtol * (float)fmax(rec[R_MAXA], rec[R_MAXB]);
    return (double)(float)(rec[R_SUMA] / (N * (double)tol * (double)tol));
  }
  const double tol = P.atol + P.rtol * fmax(rec[R_MAXA], rec[R_MAXB]);
  return rec[R_SUMA] / (N * tol * tol);
}

// dopri5.py:103-121 after the error ratio: accept test, next step size, output cursor,
// termination checks.  Shared by k_controller, the whole-attempt kernels' last workgroup and the whole-integration kernel.
// Takes the ratio that drives the step size (the python max() over the components' ratios, misc.py:270) and the
// accept test (every component's ratio <= 1, dopri5.py:108).  One-component states: ratio, ratio <= 1.
__device__ __forceinline__ void attempt_tail(AttemptState& c, double ratio, bool accept, const CtrlParams& P) {
  c.n_attempt += 1;
  c.nfe += P.n_stages;
  c.n_steps_out += 1;
  const double dt = c.dt, t_start = c.t1;
  c.ratio = ratio;
  const double dt_next = optimal_step(dt, ratio, P);
  c.accepted = accept ? 1 : 0;
  c.emit_lo = c.emit_hi = c.next_out;
  if (accept) {
    c.n_accept += 1;
    const double t_new = t_start + dt;
    c.emit_t0 = t_start; c.emit_t1 = t_new; c.emit_dt = dt;
    c.t0 = t_start; c.t1 = t_new;
    // outputs that fall into (t_start, t_new]   (`while next_t > t1` exits, dopri5.py:84)
    int nx = c.next_out;
    while (nx < c.n_out && !(P.t_out[nx] > t_new)) ++nx;
    if (nx > c.next_out) { c.emit_hi = nx; c.next_out = nx; c.n_steps_out = 0; }
  } else {
    c.n_reject += 1;
    c.t0 = t_start;                                        // rejected: rk_state.t0 == rk_state.t1
  }
  c.dt = dt_next;
  if (c.next_out >= c.n_out) { c.done = 1; return; }
  if (c.n_steps_out >= P.max_num_steps) { c.status |= MI_ODE_ST_MAX_STEPS; c.done = 1; return; }   // dopri5.py:85
  if (!(c.t1 + dt_next > c.t1)) { c.status |= MI_ODE_ST_DT_UNDERFLOW; c.done = 1; }               // dopri5.py:98
}

__device__ __forceinline__ void attempt_core(AttemptState& c, const double* rec, const CtrlParams& P) {
  const double ratio = error_ratio(rec, P);
  attempt_tail(c, ratio, ratio <= 1.0, P);                 // NaN -> rejected (dopri5.py:108)
}

// ---- tuple states: one record per component ("segment") -------------------------------------------------------------
// The reference keeps a tuple of tensors as the state (odeint.py:28-81; adjoint.py:148 builds one).  Every scalar decision
// then runs over the components: _select_initial_step takes python max() over the per-component norms (misc.py:227-245),
// _compute_error_ratio returns one ratio per component (misc.py:250-264), the step is accepted if ALL are <= 1
// (dopri5.py:108) and _optimal_step_size uses max() of them (misc.py:270).  python max(): first maximal element.
constexpr int kMaxSeg = MI_ODE_MAX_SEGMENTS;
struct SegState { double d1[kMaxSeg]; };                     // what the second half of the initial step keeps from the first

__device__ __forceinline__ double py_max(const double* v, int n) {
  double best = v[0];
  for (int i = 1; i < n; ++i)
    if (v[i] > best) best = v[i];
  return best;
}

// rec[k]: the combined record of component k, rec[k][R_N] = its element count
__device__ __forceinline__ void controller_apply_seg(Ctl* c, SegState* ss, const double (*rec)[kRec], int nseg, int phase, const CtrlParams& P) {
  if (phase == PH_F0) {                                      // misc.py:227-233
    c->nfe += 1;
    double d0[kMaxSeg], d1[kMaxSeg], q[kMaxSeg];
    bool nonfinite = false;
    for (int k = 0; k < nseg; ++k) {
      const double N = rec[k][R_N];
      nonfinite = nonfinite || rec[k][R_FLAG] != 0.0;
      if (P.is_f32) {
        d0[k] = (double)(sqrtf((float)rec[k][R_SUMA]) / powf((float)N, 0.5f));
        d1[k] = (double)(sqrtf((float)rec[k][R_SUMB]) / powf((float)N, 0.5f));
        q[k] = (double)((float)d0[k] / (float)d1[k]);
      } else {
        d0[k] = sqrt(rec[k][R_SUMA]) / pow(N, 0.5);
        d1[k] = sqrt(rec[k][R_SUMB]) / pow(N, 0.5);
        q[k] = d0[k] / d1[k];
      }
      ss->d1[k] = d1[k];
    }
    c->y0_nonfinite = nonfinite;
    const double m0 = py_max(d0, nseg), m1 = py_max(d1, nseg);
    double h0;
    if (P.is_f32) h0 = ((float)m0 < 1e-5f || (float)m1 < 1e-5f) ? (double)1e-6f : (double)(0.01f * (float)py_max(q, nseg));
    else h0 = (m0 < 1e-5 || m1 < 1e-5) ? 1e-6 : 0.01 * py_max(q, nseg);
    c->d0 = d0[0]; c->d1 = d1[0]; c->h0 = h0;
    return;
  }
  if (phase == PH_INITB) {                                   // misc.py:236-245
    c->nfe += 1;
    double both[2 * kMaxSeg], d2[kMaxSeg];
    for (int k = 0; k < nseg; ++k) {
      const double N = rec[k][R_N];
      if (P.is_f32) d2[k] = (double)((sqrtf((float)rec[k][R_SUMA]) / powf((float)N, 0.5f)) / (float)c->h0);
      else d2[k] = (sqrt(rec[k][R_SUMA]) / pow(N, 0.5)) / c->h0;
      both[k] = ss->d1[k]; both[nseg + k] = d2[k];
    }
    double first;
    if (P.is_f32) {
      const float h0 = (float)c->h0;
      float h1;
      if ((float)py_max(ss->d1, nseg) <= 1e-15f && (float)py_max(d2, nseg) <= 1e-15f) h1 = (float)nan_max((double)1e-6f, (double)(h0 * 1e-3f));
      else h1 = powf(0.01f / (float)py_max(both, 2 * nseg), (float)(1.0 / (double)(P.init_order + 1)));
      first = nan_min((double)(100.0f * h0), (double)h1);
    } else {
      const double h0 = c->h0;
      double h1;
      if (py_max(ss->d1, nseg) <= 1e-15 && py_max(d2, nseg) <= 1e-15) h1 = nan_max(1e-6, h0 * 1e-3);
      else h1 = pow(0.01 / py_max(both, 2 * nseg), 1.0 / (double)(P.init_order + 1));
      first = nan_min(100.0 * h0, h1);
    }
    c->dt = first;
    return;
  }
}

// dopri5.py:103-121 over the components; tol: {rtol_k, atol_k} per component, or null for P.rtol / P.atol
__device__ __forceinline__ void attempt_core_seg(AttemptState& c, const double (*rec)[kRec], int nseg, const CtrlParams& P,
                                                 const double* seg_rtol, const double* seg_atol) {
  if (P.controller == MI_ODE_CTRL_TSIT5) {                   // tsit5.py:126-138: ONE mean over every element of every component
    double num = 0.0, den = 0.0;
    for (int k = 0; k < nseg; ++k) {
      double tol;
      if (P.is_f32) tol = (double)((float)P.atol + (float)P.rtol * (float)fmax(rec[k][R_MAXA], rec[k][R_MAXB]));
      else tol = P.atol + P.rtol * fmax(rec[k][R_MAXA], rec[k][R_MAXB]);
      num += rec[k][R_SUMA] / (tol * tol);
      den += rec[k][R_N];
    }
    const double ratio = P.is_f32 ? (double)(float)(num / den) : num / den;
    attempt_tail(c, ratio, ratio <= 1.0, P);
    return;
  }
  double ratios[kMaxSeg];
  bool accept = true;
  for (int k = 0; k < nseg; ++k) {
    CtrlParams Pk = P;
    if (seg_rtol != nullptr) { Pk.rtol = seg_rtol[k]; Pk.atol = seg_atol[k]; }
    ratios[k] = error_ratio(rec[k], Pk);
    accept = accept && (ratios[k] <= 1.0);
  }
  attempt_tail(c, py_max(ratios, nseg), accept, P);
}

// One thread: apply the phase logic to the combined record `rec` with exactly the reference's scalar arithmetic.
__device__ __forceinline__ void controller_apply(Ctl* c, const double* rec, int phase, const CtrlParams& P) {
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
  AttemptState a;
  a.load(*c);
  attempt_core(a, rec, P);
  if (a.accepted) {
    // the step's planes, for dense output
    c->emit_y0 = c->idx_y0; c->emit_y1 = c->idx_y1;
    for (int j = 0; j <= P.n_stages; ++j) c->emit_k[j] = c->idx_k[j];
    // rotate: y1 becomes the state, k_S (= f1, FSAL) becomes f0
    const int iy = c->idx_y0; c->idx_y0 = c->idx_y1; c->idx_y1 = iy;
    const int ik = c->idx_k[0]; c->idx_k[0] = c->idx_k[P.n_stages]; c->idx_k[P.n_stages] = ik;
  }
  a.store(*c);
}

}  // namespace mi
