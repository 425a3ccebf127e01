// Family (C) of include/mi_ode.h: adaptive Runge-Kutta with an OPAQUE right-hand side - f(t, y) is whatever the caller
// evaluates between our launches (a Python callable over torch ops: the call shape of every test and example of the
// reference, tests/odeint_tests.py:30-77, examples/ode_demo.py:39,169) - but with everything else of an attempt on the
// device, so that an attempt contains no host decision and can be recorded ONCE as a hipGraph and replayed:
//   * the step size and the stage times live in device memory (mi_ode_lincomb_dev reads dt when it runs; f receives
//     0-d views of the stage-time array);
//   * mi_ode_opq_finish: error estimate (rk_common.py:60) + its norms (misc.py:256-263) per component in one pass, then
//     the controller of the fused engine (csrc/mi_ode_ctrl_dev.h: per-component ratios, accept test, next step size,
//     output cursor, the reference's assertions as status bits) in a one-wavefront kernel that also forms the NEXT
//     attempt's stage times;
//   * mi_ode_opq_commit: what `rk_state = _RungeKuttaState(y1, f1, ...)` and `_interp_evaluate` do on accept
//     (dopri5.py:113-121, :87): dense output of every requested time inside the accepted step, then y0 <- y1, f0 <- f1 in
//     the caller's (static) buffers.  Nothing happens on a rejected attempt, and nothing once `done` is set, so a host that
//     replays attempts blindly in chunks and reads the scalar state back once per chunk is correct.
#include <hip/hip_runtime.h>
#include <string.h>

#include "mi_ode_ctrl_dev.h"
#include "mi_ode_dense.h"
#include "mi_ode_host.h"

using namespace mi;

namespace {

struct OpqComp {               // one state component of one attempt
  const void* y0;
  const void* y1;
  const void* k[MI_ODE_MAX_K];
};

struct OpqErrCoef { double e[MI_ODE_MAX_K]; };

struct OpqCtlArgs {
  int ncomp, S;
  int grid[MI_ODE_MAX_SEGMENTS];
  double n[MI_ODE_MAX_SEGMENTS];
  double rtol[MI_ODE_MAX_SEGMENTS], atol[MI_ODE_MAX_SEGMENTS];
  double alpha[MI_ODE_MAX_STAGES];
  void* ts;                    // [S] stage times of the next attempt, state dtype
};

// t_sigma = t0 + alpha_sigma * dt in the state dtype (rk_common.py:45-50)
__device__ __forceinline__ void opq_stage_times(const Ctl* c, const OpqCtlArgs& A, int is_f32) {
  for (int s = 0; s < A.S; ++s) {
    if (is_f32) ((float*)A.ts)[s] = (float)c->t1 + (float)A.alpha[s] * (float)c->dt;
    else ((double*)A.ts)[s] = c->t1 + A.alpha[s] * c->dt;
  }
}

// the first wavefront of a workgroup: fold the block records of every component, apply the controller, publish the next attempt's
// stage times.  SC1: the records were stored write-through by workgroups of the SAME launch (the norms kernel's last workgroup).
template <bool SC1>
__device__ __forceinline__ void opq_controller_apply(Ctl* c, const double* part, const OpqCtlArgs& A, const CtrlParams& P, double (*rec)[kRec]) {
  for (int k = 0; k < A.ncomp; ++k) reduce_block_records<SC1>(part + (long long)k * kMaxBlocks * kRec, A.grid[k], rec[k]);
  if (threadIdx.x != 0) return;
  bool nonfinite = false;
  for (int k = 0; k < A.ncomp; ++k) { rec[k][R_N] = A.n[k]; nonfinite = nonfinite || rec[k][R_FLAG] != 0.0; }
  if (nonfinite) {                                             // dopri5.py:99-100
    c->status |= MI_ODE_ST_NONFINITE; c->done = 1; c->accepted = 0;
    return;
  }
  AttemptState a;
  a.load(*c);
  attempt_core_seg(a, rec, A.ncomp, P, A.rtol, A.atol);
  a.store(*c);
  opq_stage_times(c, A, P.is_f32);
}

// (the controller as a launch of its own: kept for the function-level surface; mi_ode_opq_finish folds it into the norms kernel)
__global__ __launch_bounds__(64) void k_opq_controller(Ctl* c, const double* part, OpqCtlArgs A, CtrlParams P) {
  __shared__ double rec[MI_ODE_MAX_SEGMENTS][kRec];
  if (c->done) {                                               // a blind replay after the end: nothing to commit any more
    if (threadIdx.x == 0) c->accepted = 0;
    return;
  }
  opq_controller_apply<false>(c, part, A, P, rec);
}


// block records {max|y0|, max|y1|, sum err^2, -, nonfinite(y0)} with err = add_n((dt * c_error_j) * k_j) formed in registers
// (NK known at compile time and two grid-stride elements per trip: all 2 (NK + 2) loads of a trip are in flight together - with a
// run-time stage count and one element per trip a 64 MB component took 165 us, 3.5 TB/s)
// CTRL (round 5: one graph node fewer per attempt): the LAST workgroup to finish - a device-scope ticket, the write-through recipe of
// finish_attempt (mi_ode_step_fused.h) - folds the records of every component and applies the controller itself; launched so for the last
// component only (the earlier components' kernels have completed: same stream).  No workgroup reads Ctl after that: each takes its ticket
// at its very end, Ctl is only read at kernel start.
struct OpqCtrlTail {
  Ctl* c;                        // writable
  const double* part_all;        // records of every component
  unsigned* ticket;              // device word, zero between launches
  OpqCtlArgs A;
  CtrlParams P;
};

template <typename T, int NK, bool CTRL>
__global__ __launch_bounds__(256) void k_opq_norms(const Ctl* c, OpqComp P, OpqErrCoef E, long long n, double* part, OpqCtrlTail Z) {
  if (c->done) {
    if (CTRL && blockIdx.x == 0 && threadIdx.x == 0) Z.c->accepted = 0;   // a blind replay after the end: nothing to commit any more
    return;
  }
  const T hs = (T)c->dt;                                       // rk_common.py:46
  const T* y0 = (const T*)P.y0;
  const T* y1 = (const T*)P.y1;
  T ce[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) ce[j] = hs * (T)E.e[j];
  Acc acc;
  auto fold = [&](T a, T b, const T (&kv)[NK]) {
    T er = ce[0] * kv[0];                                      // misc._scaled_dot_product order (misc.py:121)
#pragma unroll
    for (int j = 1; j < NK; ++j) er = er + ce[j] * kv[j];
    acc.maxa = fmax(acc.maxa, (double)fabs(a));
    acc.maxb = fmax(acc.maxb, (double)fabs(b));
    acc.suma += (double)er * (double)er;
    if (!finite_(a)) acc.flag = 1;
  };
  const long long pitch = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + pitch < n; i += 2 * pitch) {
    T k0[NK], k1[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) { k0[j] = ((const T*)P.k[j])[i]; k1[j] = ((const T*)P.k[j])[i + pitch]; }
    const T a0 = y0[i], b0 = y1[i], a1 = y0[i + pitch], b1 = y1[i + pitch];
    fold(a0, b0, k0);
    fold(a1, b1, k1);
  }
  if (i < n) {
    T k0[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) k0[j] = ((const T*)P.k[j])[i];
    fold(y0[i], y1[i], k0);
  }
  __shared__ double red[80];
  if constexpr (!CTRL) {
    block_reduce_store(acc, red, part + (long long)blockIdx.x * kRec);
  } else {
    block_reduce_store<true>(acc, red, part + (long long)blockIdx.x * kRec);
    __shared__ int s_last;
    if (threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(Z.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    __shared__ double rec[MI_ODE_MAX_SEGMENTS][kRec];
    opq_controller_apply<true>(Z.c, Z.part_all, Z.A, Z.P, rec);
    if (threadIdx.x == 0) __hip_atomic_store(Z.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T, bool CTRL>
static int opq_norms_t(int nk, dim3 g, hipStream_t st, const Ctl* ctl, const OpqComp& P, const OpqErrCoef& E, long long n, double* part, const OpqCtrlTail& Z) {
  switch (nk) {
    case 2: hipLaunchKernelGGL((k_opq_norms<T, 2, CTRL>), g, dim3(256), 0, st, ctl, P, E, n, part, Z); break;
    case 4: hipLaunchKernelGGL((k_opq_norms<T, 4, CTRL>), g, dim3(256), 0, st, ctl, P, E, n, part, Z); break;
    case 7: hipLaunchKernelGGL((k_opq_norms<T, 7, CTRL>), g, dim3(256), 0, st, ctl, P, E, n, part, Z); break;
    case 14: hipLaunchKernelGGL((k_opq_norms<T, 14, CTRL>), g, dim3(256), 0, st, ctl, P, E, n, part, Z); break;
    default: mi_set_error("opq_finish: unsupported stage count"); return MI_ODE_E_INVALID;
  }
  return 0;
}

__global__ void k_opq_times(const Ctl* c, OpqCtlArgs A, int is_f32) {
  if (threadIdx.x == 0 && blockIdx.x == 0) opq_stage_times(c, A, is_f32);
}

struct OpqCommit {
  void* y0;
  void* f0;
  const void* y1;
  const void* k[MI_ODE_MAX_K];   // k[0] is f0's buffer
  void* const* out_tab;          // device table of the components' solution rows [n_out, n], rewritten by every opq_begin: read when the
  int comp;                      // kernel runs, so that a recorded attempt replayed by a LATER call writes that call's rows
};

template <typename T, int NK>
__global__ __launch_bounds__(256) void k_opq_commit(const Ctl* c, OpqCommit P, long long n, const double* t_out, InterpParams I) {
  if (c->accepted == 0) return;
  const int lo = c->emit_lo, hi = c->emit_hi;
  const double t0 = c->emit_t0, t1 = c->emit_t1;
  const T dtT = (T)c->emit_dt;
  T* y0p = (T*)P.y0;
  T* f0p = (T*)P.f0;
  const T* y1p = (const T*)P.y1;
  T* out = hi > lo ? (T*)P.out_tab[P.comp] : nullptr;
  const long long pitch = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (hi <= lo) {                                              // no output time inside the step: a copy, four loads in flight per trip
    const T* f1p = (const T*)P.k[NK - 1];                      // (state buffers may alias nothing the loop still reads: plain copies)
    for (; i + pitch < n; i += 2 * pitch) {
      const T ya = y1p[i], fa = f1p[i], yb = y1p[i + pitch], fb = f1p[i + pitch];
      y0p[i] = ya; f0p[i] = fa; y0p[i + pitch] = yb; f0p[i + pitch] = fb;
    }
    if (i < n) { const T ya = y1p[i], fa = f1p[i]; y0p[i] = ya; f0p[i] = fa; }
    return;
  }
  for (; i < n; i += pitch) {
    const T y1 = y1p[i];
    const T f1 = ((const T*)P.k[NK - 1])[i];
    if (hi > lo) {                                             // dopri5.py:87 / interp.py:6-67 / tsit5.py:33-50
      T k[NK];
#pragma unroll
      for (int j = 0; j < NK; ++j) k[j] = ((const T*)P.k[j])[i];
      const T y0 = y0p[i];
      if (I.kind == MI_ODE_INTERP_QUARTIC_MID) {
        T co[5];
        quartic_fit<T, NK>(y0, y1, k, dtT, I, co);
        for (int j = lo; j < hi; ++j) out[(long long)j * n + i] = quartic_eval<T>(co, interp_x<T>(t0, t1, t_out[j]));
      } else {
        for (int j = lo; j < hi; ++j) out[(long long)j * n + i] = tsit5_dense<T, NK>(y0, k, t0, t1, t_out[j], I.kind);
      }
    }
    y0p[i] = y1;                                               // rk_state.y1 / rk_state.f1 (dopri5.py:113-114)
    f0p[i] = f1;
  }
}

}  // namespace

struct mi_ode_opq {
  mi_ode_opq_desc d;
  int is_f32, S, nk;
  Ctl* ctl;                    // device
  Ctl* ctl_host;               // pinned
  double* partials;            // [n_comp][kMaxBlocks][kRec]
  unsigned* ticket;            // last-workgroup-done counter of the norms kernel
  double* t_out_dev;
  double* t_out_host;          // pinned
  int t_out_cap;
  void* out[MI_ODE_MAX_SEGMENTS];
  void** out_tab_dev;          // device copy of out[] (what the commit kernels read)
  void** out_tab_host;         // pinned staging
  int grid[MI_ODE_MAX_SEGMENTS];
  CtrlParams cp;
  InterpParams ip;
  OpqCtlArgs ca;
  OpqErrCoef ec;
  int begun;
};

extern "C" int mi_ode_opq_destroy(mi_ode_opq_handle h) {
  if (h == nullptr) return 0;
  if (h->ctl) (void)hipFree(h->ctl);
  if (h->partials) (void)hipFree(h->partials);
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->t_out_dev) (void)hipFree(h->t_out_dev);
  if (h->ctl_host) (void)hipHostFree(h->ctl_host);
  if (h->t_out_host) (void)hipHostFree(h->t_out_host);
  if (h->out_tab_dev) (void)hipFree(h->out_tab_dev);
  if (h->out_tab_host) (void)hipHostFree(h->out_tab_host);
  delete h;
  return 0;
}

extern "C" int mi_ode_opq_create(const mi_ode_opq_desc* d, mi_ode_opq_handle* out) {
  if (d == nullptr || out == nullptr) { mi_set_error("opq_create: null argument"); return MI_ODE_E_INVALID; }
  *out = nullptr;
  if (d->dtype != MI_ODE_F32 && d->dtype != MI_ODE_F64) { mi_set_error("opq_create: bad dtype"); return MI_ODE_E_INVALID; }
  if (d->n_comp < 1 || d->n_comp > MI_ODE_MAX_SEGMENTS) { mi_set_error("opq_create: 1 .. %d components", MI_ODE_MAX_SEGMENTS); return MI_ODE_E_INVALID; }
  const int S = d->tableau.n_stages;
  if (S < 1 || S > MI_ODE_MAX_STAGES) { mi_set_error("opq_create: 1 .. %d tableau rows", MI_ODE_MAX_STAGES); return MI_ODE_E_INVALID; }
  const int nk = S + 1;
  if (nk != 2 && nk != 4 && nk != 7 && nk != 14) { mi_set_error("opq_create: dense output is instantiated for 2, 4, 7 and 14 stage derivatives"); return MI_ODE_E_INVALID; }
  if (d->interp != MI_ODE_INTERP_QUARTIC_MID && nk != 7) { mi_set_error("opq_create: the tsit5 dense output needs 7 stage derivatives"); return MI_ODE_E_INVALID; }
  for (int k = 0; k < d->n_comp; ++k)
    if (d->n[k] <= 0) { mi_set_error("opq_create: component %d is empty", k); return MI_ODE_E_INVALID; }
  mi_ode_opq* h = new mi_ode_opq();
  memset((void*)h, 0, sizeof(*h));
  h->d = *d;
  h->is_f32 = d->dtype == MI_ODE_F32;
  h->S = S; h->nk = nk;
  hipError_t e = hipMalloc((void**)&h->ctl, sizeof(Ctl));
  if (e == hipSuccess) e = hipMalloc((void**)&h->partials, (size_t)d->n_comp * kMaxBlocks * kRec * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->ticket, sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(h->ticket, 0, sizeof(unsigned));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->ctl_host, sizeof(Ctl));
  h->t_out_cap = 1024;                           // (allocated up front: a captured graph holds the device address)
  if (e == hipSuccess) e = hipMalloc((void**)&h->t_out_dev, (size_t)h->t_out_cap * sizeof(double));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->t_out_host, (size_t)h->t_out_cap * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->out_tab_dev, MI_ODE_MAX_SEGMENTS * sizeof(void*));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->out_tab_host, MI_ODE_MAX_SEGMENTS * sizeof(void*));
  if (e != hipSuccess) {
    mi_set_error("opq_create: allocation failed: %s", hipGetErrorString(e));
    mi_ode_opq_destroy(h);
    return MI_ODE_E_HIP;
  }
  CtrlParams& P = h->cp;
  memset(&P, 0, sizeof(P));
  P.rtol = d->rtol[0]; P.atol = d->atol[0];
  P.safety = d->safety; P.ifactor = d->ifactor; P.dfactor = d->dfactor;
  P.inv_ifactor = 1.0 / d->ifactor; P.inv_dfactor = 1.0 / d->dfactor;
  P.max_num_steps = d->max_num_steps > 0 ? d->max_num_steps : (1LL << 62);
  P.order = d->order; P.init_order = d->init_order; P.controller = d->controller;
  P.is_f32 = h->is_f32; P.n_stages = S; P.auto_first_step = 0;
  long long n_all = 0;
  OpqCtlArgs& A = h->ca;
  A.ncomp = d->n_comp; A.S = S;
  for (int k = 0; k < d->n_comp; ++k) {
    long long g = (d->n[k] + 255) / 256;
    if (g > kMaxBlocks) g = kMaxBlocks;
    h->grid[k] = A.grid[k] = (int)g;
    A.n[k] = (double)d->n[k];
    A.rtol[k] = d->rtol[k]; A.atol[k] = d->atol[k];
    n_all += d->n[k];
  }
  P.n_local = n_all;
  for (int s = 0; s < S; ++s) A.alpha[s] = d->tableau.alpha[s];
  for (int j = 0; j < nk; ++j) h->ec.e[j] = d->tableau.c_error[j];
  h->ip.kind = d->interp; h->ip.nk = nk;
  for (int j = 0; j < nk && j < MI_ODE_MAX_LINCOMB; ++j) h->ip.c_mid[j] = d->tableau.c_mid[j];
  *out = h;
  return 0;
}

extern "C" const double* mi_ode_opq_dt_dev(mi_ode_opq_handle h) {
  return h == nullptr ? nullptr : &h->ctl->dt;
}

extern "C" int mi_ode_opq_begin(mi_ode_opq_handle h, double t0, double first_dt, const double* t_out_host, int32_t n_out,
                                void* const* out_dev, void* stage_times_dev, void* stream) {
  if (h == nullptr || stage_times_dev == nullptr || (n_out > 0 && (t_out_host == nullptr || out_dev == nullptr))) { mi_set_error("opq_begin: null argument"); return MI_ODE_E_INVALID; }
  for (int i = 1; i < n_out; ++i)
    if (!(t_out_host[i] > t_out_host[i - 1])) { mi_set_error("output times must increase"); return MI_ODE_ST_BAD_T; }
  hipStream_t st = (hipStream_t)stream;
  MI_HIP(hipStreamSynchronize(st));              // the pinned staging buffers may still be in flight from a previous call
  if (n_out > h->t_out_cap) {
    if (h->t_out_dev) (void)hipFree(h->t_out_dev);
    if (h->t_out_host) (void)hipHostFree(h->t_out_host);
    h->t_out_dev = nullptr; h->t_out_host = nullptr; h->t_out_cap = 0;
    const int cap = n_out;                        // (graphs captured against the old buffer must be re-recorded: opq_begin's caller
                                                  //  re-captures whenever it begins with more than 1024 output times)
    MI_HIP(hipMalloc((void**)&h->t_out_dev, (size_t)cap * sizeof(double)));
    MI_HIP(hipHostMalloc((void**)&h->t_out_host, (size_t)cap * sizeof(double)));
    h->t_out_cap = cap;
  }
  if (n_out > 0) {
    memcpy(h->t_out_host, t_out_host, (size_t)n_out * sizeof(double));
    MI_HIP(hipMemcpyAsync(h->t_out_dev, h->t_out_host, (size_t)n_out * sizeof(double), hipMemcpyHostToDevice, st));
  }
  h->cp.t_out = h->t_out_dev;
  for (int k = 0; k < MI_ODE_MAX_SEGMENTS; ++k) h->out_tab_host[k] = h->out[k] = (k < h->d.n_comp && n_out > 0) ? out_dev[k] : nullptr;
  MI_HIP(hipMemcpyAsync(h->out_tab_dev, h->out_tab_host, MI_ODE_MAX_SEGMENTS * sizeof(void*), hipMemcpyHostToDevice, st));
  Ctl* c = h->ctl_host;
  memset(c, 0, sizeof(Ctl));
  c->t0 = c->t1 = t0;
  c->dt = first_dt;
  c->n_out = n_out;
  if (n_out <= 0) c->done = 1;
  else if (!(t0 + first_dt > t0)) { c->status |= MI_ODE_ST_DT_UNDERFLOW; c->done = 1; }      // dopri5.py:98
  MI_HIP(hipMemcpyAsync(h->ctl, c, sizeof(Ctl), hipMemcpyHostToDevice, st));
  h->ca.ts = stage_times_dev;
  hipLaunchKernelGGL(k_opq_times, dim3(1), dim3(1), 0, st, (const Ctl*)h->ctl, h->ca, h->is_f32);
  MI_HIP(hipGetLastError());
  h->begun = 1;
  return 0;
}

extern "C" int mi_ode_opq_finish(mi_ode_opq_handle h, const void* const* y0_dev, const void* const* y1_dev, const void* const* k_dev,
                                 void* stream) {
  if (h == nullptr || !y0_dev || !y1_dev || !k_dev) { mi_set_error("opq_finish: null argument"); return MI_ODE_E_INVALID; }
  if (!h->begun) { mi_set_error("opq_finish before opq_begin"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  for (int c = 0; c < h->d.n_comp; ++c) {
    OpqComp P;
    memset(&P, 0, sizeof(P));
    P.y0 = y0_dev[c]; P.y1 = y1_dev[c];
    for (int j = 0; j < h->nk; ++j) {
      P.k[j] = k_dev[c * h->nk + j];
      if (P.k[j] == nullptr) { mi_set_error("opq_finish: null stage derivative"); return MI_ODE_E_INVALID; }
    }
    if (!P.y0 || !P.y1) { mi_set_error("opq_finish: null state"); return MI_ODE_E_INVALID; }
    double* part = h->partials + (long long)c * kMaxBlocks * kRec;
    OpqCtrlTail Z;
    memset(&Z, 0, sizeof(Z));
    int rcn;
    if (c + 1 < h->d.n_comp) {
      rcn = h->is_f32 ? opq_norms_t<float, false>(h->nk, dim3(h->grid[c]), st, h->ctl, P, h->ec, (long long)h->d.n[c], part, Z)
                      : opq_norms_t<double, false>(h->nk, dim3(h->grid[c]), st, h->ctl, P, h->ec, (long long)h->d.n[c], part, Z);
    } else {                                     // the last component's kernel also runs the controller (its last workgroup)
      Z.c = h->ctl; Z.part_all = h->partials; Z.ticket = h->ticket; Z.A = h->ca; Z.P = h->cp;
      rcn = h->is_f32 ? opq_norms_t<float, true>(h->nk, dim3(h->grid[c]), st, h->ctl, P, h->ec, (long long)h->d.n[c], part, Z)
                      : opq_norms_t<double, true>(h->nk, dim3(h->grid[c]), st, h->ctl, P, h->ec, (long long)h->d.n[c], part, Z);
    }
    if (rcn != 0) return rcn;
  }
  MI_HIP(hipGetLastError());
  return 0;
}

template <typename T>
static int opq_commit_t(mi_ode_opq* h, const OpqCommit& P, int c, hipStream_t st) {
  const long long n = h->d.n[c];
  const dim3 g(h->grid[c]), b(256);
  const Ctl* ctl = h->ctl;
  const double* to = h->t_out_dev;
  switch (h->nk) {
    case 2: hipLaunchKernelGGL((k_opq_commit<T, 2>), g, b, 0, st, ctl, P, n, to, h->ip); break;
    case 4: hipLaunchKernelGGL((k_opq_commit<T, 4>), g, b, 0, st, ctl, P, n, to, h->ip); break;
    case 7: hipLaunchKernelGGL((k_opq_commit<T, 7>), g, b, 0, st, ctl, P, n, to, h->ip); break;
    case 14: hipLaunchKernelGGL((k_opq_commit<T, 14>), g, b, 0, st, ctl, P, n, to, h->ip); break;
    default: mi_set_error("opq_commit: unsupported stage count"); return MI_ODE_E_INVALID;
  }
  return 0;
}

extern "C" int mi_ode_opq_commit(mi_ode_opq_handle h, void* const* y0_dev, void* const* f0_dev, const void* const* y1_dev,
                                 const void* const* k_dev, void* stream) {
  if (h == nullptr || !y0_dev || !f0_dev || !y1_dev || !k_dev) { mi_set_error("opq_commit: null argument"); return MI_ODE_E_INVALID; }
  if (!h->begun) { mi_set_error("opq_commit before opq_begin"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  for (int c = 0; c < h->d.n_comp; ++c) {
    OpqCommit P;
    memset(&P, 0, sizeof(P));
    P.y0 = y0_dev[c]; P.f0 = f0_dev[c]; P.y1 = y1_dev[c]; P.out_tab = h->out_tab_dev; P.comp = c;
    if (!P.y0 || !P.f0 || !P.y1) { mi_set_error("opq_commit: null state"); return MI_ODE_E_INVALID; }
    for (int j = 0; j < h->nk; ++j) P.k[j] = k_dev[c * h->nk + j];
    const int rc = h->is_f32 ? opq_commit_t<float>(h, P, c, st) : opq_commit_t<double>(h, P, c, st);
    if (rc != 0) return rc;
  }
  MI_HIP(hipGetLastError());
  return 0;
}

extern "C" int mi_ode_opq_poll(mi_ode_opq_handle h, mi_ode_stats* stats, int32_t* done, void* stream) {
  if (h == nullptr) { mi_set_error("opq_poll: null handle"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  MI_HIP(hipMemcpyAsync(h->ctl_host, h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, st));
  MI_HIP(hipStreamSynchronize(st));
  const Ctl* c = h->ctl_host;
  if (done) *done = c->done;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->n_attempts = c->n_attempt; stats->n_accepted = c->n_accept; stats->n_rejected = c->n_reject; stats->nfe = c->nfe;
    stats->t = c->t1; stats->dt = c->dt; stats->last_ratio = c->ratio; stats->status = c->status;
  }
  return (int)c->status;
}
