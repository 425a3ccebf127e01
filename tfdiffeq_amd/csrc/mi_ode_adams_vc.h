// The reference's variable-step, variable-order Adams-Bashforth-Moulton solver ('adams': tfdiffeq/adams.py:66-211, Hairer,
// Norsett, Wanner III.5) in ONE launch for the row-local catalogue systems.
//
// One thread owns one trajectory: the state y and the implicit-phi deque (up to max_order + 1 backward differences of f, newest
// first) stay in registers for the whole integration.  Thread 0 of EVERY workgroup runs the scalar bookkeeping redundantly (the
// same inputs, the same IEEE operations): the deque of previous times, beta and the g vector of adams.py:29-63 - g in a float32
// array, as the reference's tf.Variable (adams.py:34) -, the accept test, the order selection of adams.py:176-199 and the step
// size; what the other threads need travels through LDS.  The only coupling between trajectories are the error ratios of
// misc._compute_error_ratio - scalar tolerance from the max norms of the WHOLE state, then mean((estimate / tolerance)^2) - so an
// attempt has two grid hand-offs (max norms; sums for error_k and error_{k-1}) and an accepted step of order >= 3 a third (sums
// for error_{k-2}, error_{k+1}); mi_ode_persist.h's sequence-numbered records.  The workgroups must be co-resident.
//
// Per attempt (adams.py:134-210), operation for operation (every product written as the plane kernels the host loop used form it:
// mi_ode_lincomb = base + add_n((scale * c_j) * x_j) in the state dtype):
//   next_t clipped to the requested time (:136-137);  dt = next_t - prev_t[0];  g, beta = g_and_explicit_phi(prev_t, next_t, phi, k)
//   explicit_phi_0 = phi_0, explicit_phi_j = beta_j * phi_j                                                       (:49-52)
//   p_next = y + dt * sum_{j < max(1, k-1)} g_j explicit_phi_j                                                    (:146-149)
//   f_p = f(next_t, p_next);  implicit_phi_p: ip_0 = f_p, ip_j = ip_{j-1} - explicit_phi_{j-1}, j <= k              (:66-81)
//   y_next = p_next + dt * g_{k-1} ip_{k-1};  local_error = dt * (g_k - g_{k-1}) ip_k                             (:155-164)
//   tolerance = atol + rtol * max(max|y|, max|y_next|);  error_k = mean((local_error / tolerance)^2);  accept: error_k <= 1
//   rejected: next_t = prev_t[0] + optimal_step_size(dt, error_k, order = k)                                      (:169-172)
//   accepted: f_n = f(next_t, y_next);  phi <- compute_implicit_phi(explicit_phi, f_n, k + 2)   (k + 1 entries);
//     order: len(prev_t) <= 4 or k < 3: min(k + 1, 3, max_order);  else k - 1 if min(error_{k-1}, error_{k-2}) < error_k, else k + 1
//     if k < max_order and error_{k+1} < error_k                                                                 (:176-199)
//     dt_next = dt if the order grew else optimal_step_size(dt, error_k, order = k + 1);  prev_t.appendleft(next_t)
//     the state advances with the PREDICTOR p_next (adams.py:210), next_t += dt_next
// solution[i] is the state when prev_t[0] lands exactly on t[i] (:122-128).
#pragma once
#include "mi_ode_persist.h"

namespace mi {

constexpr int kVcMaxOrder = 12;                              // adams.py:13 (_MAX_ORDER)
constexpr int kVcPhi = kVcMaxOrder + 1;                      // compute_implicit_phi(.., k + 2) keeps k + 1 <= 13 entries

struct AdamsVcArgs {
  FixedArgs f;                 // y0, out, t (device copy of the T requested times), T, batch, dim, rhs
  PersistArgs p;               // the hand-off fields only (s.partials, seq_base, spin_*, sleep_*, nseg = 1, world = 1)
  CtrlParams cp;               // rtol, atol, safety, ifactor, dfactor (and inverses), is_f32, controller = MISC, init_order = 2, n_local
  int max_order;
  long long max_attempts;      // the reference has no bound: a NaN step would loop forever - status MI_ODE_ST_MAX_STEPS instead
  double gamma_star[kVcMaxOrder + 1];
  long long* result;           // pinned host: attempts, accepted, nfe, status bits
};

struct VcShared {
  double prev_t[kVcMaxOrder + 1];                            // newest first (deque(maxlen = max_order + 1), adams.py:101)
  double beta[kVcMaxOrder + 1];
  double cw[kVcMaxOrder + 2];                                // the c vector of g_and_explicit_phi
  float g[kVcMaxOrder + 2];
  double dt, next_t, tol;
  int n_prev, order, accept, done, want_lower, want_higher;
  unsigned status;
};

template <typename T, class RHS>
__global__ __launch_bounds__(256, 2) void k_adams_vc_rowlocal(AdamsVcArgs A) {     // (2 wavefronts per SIMD: 131072 co-resident trajectories)
  constexpr int D = RHS::D;
  using Row = RowVec<T, D>;
  __shared__ PersistShared sh;
  __shared__ VcShared vs;
  const RHS rhs(A.f.rhs);
  const T sign = (T)A.f.rhs.sign;
  long long n, off;                                          // elements per solution row; this thread's first element
  bool live;
  rowmap<RHS>(A.f.batch, A.f.dim, A.f.rhs, off, live, n);
  const CtrlParams cp = A.cp;
  const double n_tot = (double)cp.n_local;
  Ctl& c = sh.c;
  unsigned gen = 0;
  double r[5], rec[kRec];
  T* out = (T*)A.f.out;
  const double t_first = A.f.t[0];
  if (threadIdx.x == 0) { sh.ok = 1; c.nfe = 0; c.h0 = c.d0 = c.d1 = 0.0; c.dt = 0.0; c.y0_nonfinite = 0; }
  __syncthreads();

  Row y;
#pragma unroll
  for (int d = 0; d < D; ++d) y.v[d] = (T)0;
  if (live) {
    y = *(const Row*)((const T*)A.f.y0 + off);
    *(Row*)(out + off) = y;                                  // solution[0] = y0 (solvers.py:30)
  }
  T phi[kVcPhi][D];
#pragma unroll
  for (int j = 0; j < kVcPhi; ++j)
#pragma unroll
    for (int d = 0; d < D; ++d) phi[j][d] = (T)0;

  // ---- before_integrate (adams.py:104-119): f0, first step = misc._select_initial_step(.., order 2, ..) ----
  bool ok;
  {
    Acc acc;
    T f0[D];
    rhs(sign * (T)t_first, y.v, f0);
#pragma unroll
    for (int d = 0; d < D; ++d) { f0[d] = sign * f0[d]; phi[0][d] = f0[d]; }
    if (live) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T sc = (T)cp.atol + fabs(y.v[d]) * (T)cp.rtol;    // misc.py:225
        const double q0 = (double)(y.v[d] / sc), q1 = (double)(f0[d] / sc);
        acc.suma += q0 * q0; acc.sumb += q1 * q1;
        if (!finite_(y.v[d])) acc.flag = 1;
      }
    }
    ok = grid_reduce_rank(A.p, acc, sh, gen++, r);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&c, rec, PH_F0, cp); }
    __syncthreads();
  }
  if (ok) {
    Acc acc;
    const T h0 = (T)uniform_d(c.h0);
    T ys[D], f1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ys[d] = y.v[d] + h0 * phi[0][d];
    rhs(sign * ((T)t_first + (T)1.0 * h0), ys, f1);
    if (live) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T kn = sign * f1[d];
        const T sc = (T)cp.atol + fabs(y.v[d]) * (T)cp.rtol;
        const double q = (double)((kn - phi[0][d]) / sc);    // misc.py:237
        acc.suma += q * q;
      }
    }
    ok = grid_reduce_rank(A.p, acc, sh, gen++, r);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&c, rec, PH_INITB, cp); }
  }
  long long n_attempt = 0, n_accept = 0, nfe = 2;
  if (threadIdx.x == 0) {
    vs.prev_t[0] = t_first; vs.n_prev = 1; vs.order = 1;
    vs.next_t = t_first + c.dt;                              // adams.py:119
    vs.status = ok ? 0u : (unsigned)MI_ODE_ST_SYNC_TIMEOUT;
    vs.done = ok ? 0 : 1;
    vs.accept = 0; vs.want_lower = vs.want_higher = 0;
  }
  __syncthreads();

  for (int i_out = 1; i_out < A.f.T; ++i_out) {
    const double final_t = A.f.t[i_out];
    for (;;) {                                               // advance (adams.py:122-128): while final_t > prev_t[0]
      __syncthreads();                                       // (thread 0's updates of the previous attempt are complete)
      if (uniform_i(vs.done) || !(final_t > uniform_d(vs.prev_t[0]))) break;
      if (threadIdx.x == 0) {                                // ---- g_and_explicit_phi (adams.py:29-63), scalars ----
        double next_t = vs.next_t;
        if (next_t > final_t) next_t = final_t;              // :136-137
        vs.next_t = next_t;
        const int k = vs.order;
        const double curr_t = vs.prev_t[0];
        const double dt = next_t - curr_t;
        vs.dt = dt;
        for (int q = 0; q <= k; ++q) vs.cw[q] = 1.0 / (double)(q + 1);
        int len = k + 1;
        vs.g[0] = 1.0f;
        double beta = 1.0;
        vs.beta[0] = 1.0;
        for (int j = 1; j < k; ++j) {
          beta = (next_t - vs.prev_t[j - 1]) / (curr_t - vs.prev_t[j]) * beta;
          vs.beta[j] = beta;
          if (j == 1) { for (int q = 0; q + 1 < len; ++q) vs.cw[q] = vs.cw[q] - vs.cw[q + 1]; }
          else { const double den = next_t - vs.prev_t[j - 1]; for (int q = 0; q + 1 < len; ++q) vs.cw[q] = vs.cw[q] - vs.cw[q + 1] * dt / den; }
          len -= 1;
          vs.g[j] = (float)vs.cw[0];
        }
        { const double den = next_t - vs.prev_t[k - 1]; for (int q = 0; q + 1 < len; ++q) vs.cw[q] = vs.cw[q] - vs.cw[q + 1] * dt / den; }
        vs.g[k] = (float)vs.cw[0];
        vs.want_lower = !(vs.n_prev <= 4 || k < 3) ? 1 : 0;
        vs.want_higher = (vs.want_lower && k < A.max_order) ? 1 : 0;
      }
      __syncthreads();
      const int order = uniform_i(vs.order);
      const bool want_lower = uniform_i(vs.want_lower) != 0;
      const double next_t = uniform_d(vs.next_t);
      const T dtc = (T)uniform_d(vs.dt);
      const T tn = sign * (T)next_t;
      const int nterm = order - 1 > 1 ? order - 1 : 1;

      // explicit phi is beta_j * phi_j (recomputed where it is used: the products are exact functions of the same operands)
      T p[D], fp[D];
      {
        T a_[D];
        const T c0 = dtc * (T)vs.g[0];
#pragma unroll
        for (int d = 0; d < D; ++d) a_[d] = c0 * phi[0][d];
#pragma unroll
        for (int j = 1; j < kVcMaxOrder; ++j) {
          if (j < nterm) {
            const T cj = dtc * (T)vs.g[j], bj = (T)vs.beta[j];
#pragma unroll
            for (int d = 0; d < D; ++d) a_[d] = a_[d] + cj * (bj * phi[j][d]);
          }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) p[d] = y.v[d] + a_[d];
      }
      rhs(tn, p, fp);
      T ipk[D], ipk1[D], ipk2[D];                            // implicit_phi_p[order], [order - 1], [order - 2]
      {
        T cur[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { cur[d] = sign * fp[d]; ipk[d] = ipk1[d] = ipk2[d] = (T)0; }
        if (order == 1) {
#pragma unroll
          for (int d = 0; d < D; ++d) ipk1[d] = cur[d];
        }
        if (order == 2) {
#pragma unroll
          for (int d = 0; d < D; ++d) ipk2[d] = cur[d];
        }
#pragma unroll
        for (int j = 1; j <= kVcMaxOrder; ++j) {
          if (j <= order) {
            const T bj = (T)vs.beta[j - 1];
#pragma unroll
            for (int d = 0; d < D; ++d) {
              const T pe = (j - 1 == 0) ? phi[0][d] : bj * phi[j - 1][d];
              cur[d] = cur[d] - pe;
              if (j == order) ipk[d] = cur[d];
              if (j == order - 1) ipk1[d] = cur[d];
              if (j == order - 2) ipk2[d] = cur[d];
            }
          }
        }
      }
      T yn[D], le[D];
      {
        const T gk1 = (T)vs.g[order - 1], gk = (T)vs.g[order];
        const T cy = dtc * gk1, ce = dtc * (gk - gk1);
#pragma unroll
        for (int d = 0; d < D; ++d) { yn[d] = p[d] + cy * ipk1[d]; le[d] = ce * ipk[d]; }
      }
      // ---- hand-off A: the max norms behind the scalar tolerance (misc.py:256-259) ----
      {
        Acc acc;
        if (live) {
#pragma unroll
          for (int d = 0; d < D; ++d) { acc.maxa = fmax(acc.maxa, (double)fabs(y.v[d])); acc.maxb = fmax(acc.maxb, (double)fabs(yn[d])); }
        }
        ok = grid_reduce_rank(A.p, acc, sh, gen++, r);
        if (threadIdx.x == 0) {
          const double m = fmax(r[0], r[1]);
          vs.tol = cp.is_f32 ? (double)((float)cp.atol + (float)cp.rtol * (float)m) : cp.atol + cp.rtol * m;
          if (!ok) { vs.status |= MI_ODE_ST_SYNC_TIMEOUT; vs.done = 1; }
        }
        __syncthreads();
        if (!ok) continue;
      }
      const T tol = (T)uniform_d(vs.tol);
      double e_k = 0.0, e_km1 = 0.0, e_km2 = 0.0, e_kp1 = 0.0;
      // ---- hand-off B: sums of (estimate / tolerance)^2 for error_k and error_{k-1} ----
      {
        Acc acc;
        if (live) {
          const T ce1 = want_lower ? dtc * ((T)vs.g[order - 1] - (T)vs.g[order - 2]) : (T)0;
#pragma unroll
          for (int d = 0; d < D; ++d) {
            const double q = (double)(le[d] / tol);
            acc.suma += q * q;
            if (want_lower) { const double q1 = (double)((ce1 * ipk1[d]) / tol); acc.sumb += q1 * q1; }
          }
        }
        ok = grid_reduce_rank(A.p, acc, sh, gen++, r);
        if (threadIdx.x == 0) {
          e_k = cp.is_f32 ? (double)(float)(r[2] / n_tot) : r[2] / n_tot;
          e_km1 = cp.is_f32 ? (double)(float)(r[3] / n_tot) : r[3] / n_tot;
        }
      }
      if (ok && want_lower) {                                // ---- hand-off C: error_{k-2}, error_{k+1} ----
        Acc acc;
        if (live) {
          const T ce2 = dtc * ((T)vs.g[order - 2] - (T)vs.g[order - 3]);
          const T cp1 = dtc * (T)A.gamma_star[order < kVcMaxOrder ? order : kVcMaxOrder];
#pragma unroll
          for (int d = 0; d < D; ++d) {
            const double q2 = (double)((ce2 * ipk2[d]) / tol), q3 = (double)((cp1 * ipk[d]) / tol);
            acc.suma += q2 * q2; acc.sumb += q3 * q3;
          }
        }
        ok = grid_reduce_rank(A.p, acc, sh, gen++, r);
        if (threadIdx.x == 0) {
          e_km2 = cp.is_f32 ? (double)(float)(r[2] / n_tot) : r[2] / n_tot;
          e_kp1 = cp.is_f32 ? (double)(float)(r[3] / n_tot) : r[3] / n_tot;
        }
      }
      // ---- thread 0: accept test, order selection, next step (adams.py:166-210) ----
      if (threadIdx.x == 0) {
        n_attempt += 1; nfe += 1;
        const double dt = vs.dt;
        if (!ok) { vs.status |= MI_ODE_ST_SYNC_TIMEOUT; vs.done = 1; vs.accept = 0; }
        else {
          const bool accept = e_k <= 1.0;
          vs.accept = accept ? 1 : 0;
          CtrlParams cq = cp;
          if (!accept) {
            cq.order = order;
            vs.next_t = vs.prev_t[0] + optimal_step(dt, e_k, cq);
          } else {
            n_accept += 1; nfe += 1;
            int next_order = order;
            if (vs.n_prev <= 4 || order < 3) {
              next_order = order + 1 < 3 ? order + 1 : 3;
              if (next_order > A.max_order) next_order = A.max_order;
            } else {
              const double lo = e_km1 < e_km2 ? e_km1 : e_km2;  // python min() over (error_km1 + error_km2)
              if (lo < e_k) next_order = order - 1;
              else if (order < A.max_order && e_kp1 < e_k) next_order = order + 1;
            }
            cq.order = order + 1;
            const double dt_next = next_order > order ? dt : optimal_step(dt, e_k, cq);
            const int np = vs.n_prev < A.max_order + 1 ? vs.n_prev + 1 : A.max_order + 1;
            for (int j = np - 1; j > 0; --j) vs.prev_t[j] = vs.prev_t[j - 1];
            vs.prev_t[0] = vs.next_t;
            vs.n_prev = np;
            vs.next_t = vs.next_t + dt_next;
            vs.order = next_order;
          }
          if (n_attempt >= A.max_attempts) { vs.status |= MI_ODE_ST_MAX_STEPS; vs.done = 1; }
        }
      }
      __syncthreads();
      if (uniform_i(vs.accept) && !uniform_i((int)(vs.status & MI_ODE_ST_SYNC_TIMEOUT))) {
        // phi <- compute_implicit_phi(explicit_phi, f(next_t, y_next), order + 2): order + 1 entries (adams.py:66-81, :174-175)
        T fnw[D];
        rhs(tn, yn, fnw);
        T prev_new[D], pe_prev[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { prev_new[d] = sign * fnw[d]; pe_prev[d] = phi[0][d]; phi[0][d] = prev_new[d]; }
#pragma unroll
        for (int j = 1; j < kVcPhi; ++j) {
          if (j <= order) {
            const T bj = (T)vs.beta[j < kVcMaxOrder ? j : kVcMaxOrder];
#pragma unroll
            for (int d = 0; d < D; ++d) {
              const T pe_j = bj * phi[j][d];                 // explicit_phi_j of the OLD deque (only read for j < order)
              const T nv = prev_new[d] - pe_prev[d];
              phi[j][d] = nv;
              prev_new[d] = nv;
              pe_prev[d] = pe_j;
            }
          }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) y.v[d] = p[d];           // the PREDICTOR (adams.py:210)
      }
    }
    if (uniform_i(vs.done)) break;
    if (live) *(Row*)(out + (long long)i_out * n + off) = y;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && A.result != nullptr) {
    __hip_atomic_store(A.result + 0, n_attempt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.result + 1, n_accept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.result + 2, nfe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.result + 3, (long long)vs.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace mi
