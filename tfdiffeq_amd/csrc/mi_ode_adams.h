// Fixed-grid multistep solvers of the reference in ONE launch for the row-local catalogue systems:
//   'explicit_adams'  AdamsBashforth         (fixed_adams.py:209-212)
//   'fixed_adams'     AdamsBashforthMoulton  (fixed_adams.py:152-207): AB predictor, AM corrector by functional iteration
// One thread owns one trajectory: the state and the history of up to max_order - 1 derivatives (newest first, the reference's
// deque) stay in registers; solution rows are streamed out as the grid walks past the requested times (solvers.py:82-115).
//
// What the reference does per grid interval [t, t + dt] (fixed_adams.py:170-206), mirrored operation for operation:
//   history <- f(t, y) pushed left                                                              (:171-172, _update_history)
//   order = min(len(history), max_order - 1);  order < min_order - 1: one RK4 3/8-rule step with k1 = history[0]    (:176-179)
//   dy = dt * add_n([(1/div * c_j) * f_j])                      Adams-Bashforth predictor, integer table / divisor   (:182-184)
//   implicit: delta = dt * add_n([(1/mdiv * m_j) * f_{j-1}], j >= 1);  up to max_iters times:
//       f = func(t + dt, y + dy);  dy <- dt * (m_0 / mdiv) * f + delta;  stop when _has_converged(dy_old, dy)        (:187-196)
//     not converged: warning, history.pop() (the OLDEST entry);  _update_history(t, f) is a no-op (same t)           (:197-201)
//   y <- y + dy                                                                                  (solvers.py:95)
// `_has_converged` (misc.py:129-134) is ONE decision for the whole state tensor - every trajectory of the batch: the
// only coupling between threads.  Inside a workgroup it is a __syncthreads_or; across workgroups the kernel uses the grid
// hand-off of the whole-integration kernels (mi_ode_persist.h: sequence-numbered records, every workgroup folds the same
// records), so the implicit solver needs a co-resident grid; the explicit one has no coupling at all.
#pragma once
#include "mi_ode_persist.h"

namespace mi {

constexpr int kAdamsMaxOrder = 12;                           // fixed_adams.py:89 (_MAX_ORDER)
constexpr int kAdamsHist = kAdamsMaxOrder - 1;               // deque(maxlen = max_order - 1)

struct AdamsArgs {
  FixedArgs f;                 // y0, out, t, grid, M, eps, batch, T, dim, rhs
  PersistArgs p;               // the hand-off fields only: s.partials, seq_base, spin_*, sleep_*, nseg = 1, world = 1
  int implicit, max_iters, max_order, min_order;
  double rtol, atol;
  const double* tab;           // device: ab[13][12] | am[13][12] | am0[13] (host-computed in Python floats, as the reference forms them)
  long long* result;           // pinned host: [0] steps whose corrector did not converge, [1] status bits (MI_ODE_ST_SYNC_TIMEOUT)
};

template <typename T, class RHS>
__global__ __launch_bounds__(256) void k_fixed_adams_rowlocal(AdamsArgs A) {
  constexpr int D = RHS::D;
  using Row = RowVec<T, D>;
  __shared__ PersistShared sh;
  const RHS rhs(A.f.rhs);
  const T sign = (T)A.f.rhs.sign;
  long long n, off;                                          // elements per solution row; this thread's first element
  bool active;
  rowmap<RHS>(A.f.batch, A.f.dim, A.f.rhs, off, active, n);
  const T* y0p = (const T*)A.f.y0;
  T* out = (T*)A.f.out;
  const double* AB = A.tab;
  const double* AM = A.tab + 13 * 12;
  const double* AM0 = A.tab + 2 * 13 * 12;
  if (threadIdx.x == 0) sh.ok = 1;
  __syncthreads();

  Row y;
#pragma unroll
  for (int d = 0; d < D; ++d) y.v[d] = (T)0;
  if (active) {
    y = *(const Row*)(y0p + off);
    *(Row*)(out + off) = y;                                  // solution = [y0]
  }
  T hist[kAdamsHist][D];                                     // hist[0] = newest
#pragma unroll
  for (int j = 0; j < kAdamsHist; ++j)
#pragma unroll
    for (int d = 0; d < D; ++d) hist[j][d] = (T)0;
  int len = 0, j_out = 1;
  long long n_noconv = 0;
  unsigned gen = 0;
  bool ok = true;
  const T eps = (T)A.f.eps;
  const T rtol = (T)A.rtol, atol = (T)A.atol;

  for (int i = 0; i < A.f.M && ok; ++i) {
    const T t0 = (T)A.f.grid[i], t1 = (T)A.f.grid[i + 1];
    const T dt = t1 - t0;
    const T te = t0 + eps;
    T fn[D], dy[D];
    rhs(sign * te, y.v, fn);
    // deque.appendleft (maxlen max_order - 1: the oldest falls out)
#pragma unroll
    for (int j = kAdamsHist - 1; j > 0; --j)
#pragma unroll
      for (int d = 0; d < D; ++d) hist[j][d] = hist[j - 1][d];
#pragma unroll
    for (int d = 0; d < D; ++d) hist[0][d] = sign * fn[d];
    len = len + 1 < A.max_order - 1 ? len + 1 : A.max_order - 1;
    const int order = len;
    if (order < A.min_order - 1) {                           // start-up: rk_common.py:73-81 with k1 = history[0]
      T k2[D], k3[D], k4[D], ys[D];
#pragma unroll
      for (int d = 0; d < D; ++d) ys[d] = y.v[d] + dt * hist[0][d] / (T)3;
      rhs(sign * (te + dt / (T)3), ys, k2);
#pragma unroll
      for (int d = 0; d < D; ++d) { k2[d] = sign * k2[d]; ys[d] = y.v[d] + dt * (hist[0][d] / (T)-3 + k2[d]); }
      rhs(sign * (te + dt * (T)2 / (T)3), ys, k3);
#pragma unroll
      for (int d = 0; d < D; ++d) { k3[d] = sign * k3[d]; ys[d] = y.v[d] + dt * (hist[0][d] - k2[d] + k3[d]); }
      rhs(sign * (te + dt), ys, k4);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        k4[d] = sign * k4[d];
        dy[d] = (hist[0][d] + (T)3 * k2[d] + (T)3 * k3[d] + k4[d]) * (dt / (T)8);
      }
    } else {
      const double* cb = AB + order * 12;
      {
        T a[D];                                              // misc._scaled_dot_product: add_n([(scale * c_j) * f_j]), in order
#pragma unroll
        for (int d = 0; d < D; ++d) a[d] = (T)cb[0] * hist[0][d];
#pragma unroll
        for (int j = 1; j < kAdamsHist; ++j) {               // (fully unrolled with a uniform guard: the history stays in registers)
          if (j < order) {
            const T c = (T)cb[j];
#pragma unroll
            for (int d = 0; d < D; ++d) a[d] = a[d] + c * hist[j][d];
          }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) dy[d] = dt * a[d];
      }
      if (A.implicit) {
        const double* cm = AM + order * 12;
        const T c0 = dt * (T)AM0[order];                     // dt * (m_0 / mdiv)
        T delta[D];
        {
          T a[D];
#pragma unroll
          for (int d = 0; d < D; ++d) a[d] = (T)cm[0] * hist[0][d];
#pragma unroll
          for (int j = 1; j < kAdamsHist; ++j) {
            if (j < order) {
              const T c = (T)cm[j];
#pragma unroll
              for (int d = 0; d < D; ++d) a[d] = a[d] + c * hist[j][d];
            }
          }
#pragma unroll
          for (int d = 0; d < D; ++d) delta[d] = dt * a[d];
        }
        bool converged = false;
        for (int it = 0; it < A.max_iters; ++it) {
          T ys[D], f[D];
#pragma unroll
          for (int d = 0; d < D; ++d) ys[d] = y.v[d] + dy[d];
          rhs(sign * (te + dt), ys, f);
          int bad = 0;
#pragma unroll
          for (int d = 0; d < D; ++d) {
            const T dn = c0 * (sign * f[d]) + delta[d];
            const T tol = atol + rtol * fmax(fabs(dy[d]), fabs(dn));          // misc.py:131-133
            if (!(fabs(dy[d] - dn) < tol)) bad = 1;
            dy[d] = dn;
          }
          Acc acc;
          acc.flag = active ? bad : 0;
          double r[5];
          ok = grid_reduce_rank(A.p, acc, sh, gen++, r);     // one decision for the whole batch (sh.ok: hand-off time-out)
          if (threadIdx.x == 0) sh.pub.accepted = (ok && !(r[4] > 0.0)) ? 1 : 0;
          __syncthreads();
          converged = sh.pub.accepted != 0;
          __syncthreads();
          if (converged || !ok) break;
        }
        if (!converged && ok) {                              // fixed_adams.py:197-200: warning + prev_f.pop()
          n_noconv += 1;
          len -= 1;
        }
      }
    }
    Row yn;
#pragma unroll
    for (int d = 0; d < D; ++d) yn.v[d] = y.v[d] + dy[d];    // solvers.py:95
    while (j_out < A.f.T && t1 >= (T)A.f.t[j_out]) {         // solvers.py:97-100, _linear_interp :106-115
      const T tj = (T)A.f.t[j_out];
      if (active) {
        Row o;
#pragma unroll
        for (int d = 0; d < D; ++d)
          o.v[d] = (tj == t0) ? y.v[d] : ((tj == t1) ? yn.v[d] : y.v[d] + ((yn.v[d] - y.v[d]) / (t1 - t0)) * (tj - t0));
        *(Row*)(out + (long long)j_out * n + off) = o;
      }
      ++j_out;
    }
    y = yn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && A.result != nullptr) {
    __hip_atomic_store(A.result + 0, n_noconv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.result + 1, (long long)(ok ? 0 : MI_ODE_ST_SYNC_TIMEOUT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace mi
