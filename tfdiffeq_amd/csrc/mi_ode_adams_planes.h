// Plane kernels of the variable-order Adams solver's host loop (tfdiffeq_amd/adams.py; any Python callable, tuple states):
// what adams.py:134-210 does per attempt with one linear combination at a time - ~45 launches at order 12 - as four elementwise
// kernels.  The arithmetic is the plane kernels' (mi_ode_lincomb: base + add_n((scale * c_j) * x_j) in the state dtype, k_error_norms,
// k_scaled_sumsq), operation for operation, so the solver's numbers do not change; mi_ode_adams_vc.h has the derivation.
#pragma once
#include "mi_ode_dev.h"

namespace mi {

constexpr int kAdamsPlanesMax = 13;                          // implicit phi holds up to max_order + 1 planes

struct AdamsPlaneArgs {
  const void* y0;              // the state
  const void* p;               // predictor p_next                                  (correct)
  const void* f;               // f(next_t, p_next) / f(next_t, y_next)             (correct / update)
  const void* phi[kAdamsPlanesMax];      // implicit phi, newest first
  void* out[kAdamsPlanesMax];  // predict: [0] p_next;  correct: [0] y_next, [1] ip_k, [2] ip_{k-1}, [3] ip_{k-2} (or null);  update: new phi
  double g[kAdamsPlanesMax + 1];         // values of the float32 g vector
  double beta[kAdamsPlanesMax];
  double ca, cb;               // error sums: coefficient differences (state dtype values)
  const void* xa; const void* xb;        // error sums: the planes (xb may be null)
  double dt, tol;
  int order;
  long long n;
  double* part;                // [gridDim.x][kRec]
};

// p_next = y + dt * sum_{j < max(1, k - 1)} g_j explicit_phi_j,  explicit_phi_0 = phi_0, explicit_phi_j = beta_j phi_j  (adams.py:49-52, :146-149)
template <typename T>
__global__ __launch_bounds__(256) void k_adams_predict(AdamsPlaneArgs A) {
  const int nterm = A.order - 1 > 1 ? A.order - 1 : 1;
  const T dtc = (T)A.dt;
  T* out = (T*)A.out[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (long long)gridDim.x * blockDim.x) {
    T acc = (dtc * (T)A.g[0]) * ((const T*)A.phi[0])[i];
    for (int j = 1; j < nterm; ++j) acc = acc + (dtc * (T)A.g[j]) * ((T)A.beta[j] * ((const T*)A.phi[j])[i]);
    out[i] = ((const T*)A.y0)[i] + acc;
  }
}

// implicit_phi_p (adams.py:66-81): ip_0 = f_p, ip_j = ip_{j-1} - explicit_phi_{j-1};  y_next = p + dt g_{k-1} ip_{k-1} (:155-158);
// records {max|y|, max|y_next|, sum local_error^2, nonfinite(y)} as k_error_norms(local_error, y, y_next)
template <typename T>
__global__ __launch_bounds__(256) void k_adams_correct(AdamsPlaneArgs A) {
  const int k = A.order;
  const T dtc = (T)A.dt;
  const T cy = dtc * (T)A.g[k - 1], ce = dtc * ((T)A.g[k] - (T)A.g[k - 1]);
  Acc acc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (long long)gridDim.x * blockDim.x) {
    T cur = ((const T*)A.f)[i];
    T ipk1 = cur, ipk2 = cur;                                // (k = 1: ip_{k-1} = ip_0;  k = 2: ip_{k-2} = ip_0)
    for (int j = 1; j <= k; ++j) {
      const T pe = (j == 1) ? ((const T*)A.phi[0])[i] : (T)A.beta[j - 1] * ((const T*)A.phi[j - 1])[i];
      cur = cur - pe;
      if (j == k - 1) ipk1 = cur;
      if (j == k - 2) ipk2 = cur;
    }
    const T y0 = ((const T*)A.y0)[i];
    const T yn = ((const T*)A.p)[i] + cy * ipk1;
    const T e = ce * cur;
    ((T*)A.out[0])[i] = yn;
    ((T*)A.out[1])[i] = cur;
    ((T*)A.out[2])[i] = ipk1;
    if (A.out[3] != nullptr) ((T*)A.out[3])[i] = ipk2;
    acc.maxa = fmax(acc.maxa, (double)fabs(y0));
    acc.maxb = fmax(acc.maxb, (double)fabs(yn));
    acc.suma += (double)e * (double)e;
    if (!finite_(y0)) acc.flag = 1;
  }
  __shared__ double red[80];
  block_reduce_store(acc, red, A.part + (long long)blockIdx.x * kRec);
}

// sums of ((dt * c) x / tol)^2 for one or two estimates: k_scaled_sumsq(lincomb(None, [c], [x], dt), rtol = 0, atol = tol)
template <typename T>
__global__ __launch_bounds__(256) void k_adams_error_sums(AdamsPlaneArgs A) {
  const T dtc = (T)A.dt, tol = (T)A.tol;
  const T ca = dtc * (T)A.ca, cb = dtc * (T)A.cb;
  Acc acc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (long long)gridDim.x * blockDim.x) {
    const T ea = ca * ((const T*)A.xa)[i];
    const T sa = tol + fabs(ea) * (T)0;                      // (the scale k_scaled_sumsq forms: atol + |y0| * rtol with rtol = 0)
    const double qa = (double)(ea / sa);
    acc.suma += qa * qa;
    if (A.xb != nullptr) {
      const T eb = cb * ((const T*)A.xb)[i];
      const T sb = tol + fabs(eb) * (T)0;
      const double qb = (double)(eb / sb);
      acc.sumb += qb * qb;
    }
  }
  __shared__ double red[80];
  block_reduce_store(acc, red, A.part + (long long)blockIdx.x * kRec);
}

// phi <- compute_implicit_phi(explicit_phi, f_n, k + 2): new_0 = f_n, new_j = new_{j-1} - explicit_phi_{j-1}, j <= k   (adams.py:174-175)
template <typename T>
__global__ __launch_bounds__(256) void k_adams_update_phi(AdamsPlaneArgs A) {
  const int k = A.order;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (long long)gridDim.x * blockDim.x) {
    T cur = ((const T*)A.f)[i];
    ((T*)A.out[0])[i] = cur;
    for (int j = 1; j <= k; ++j) {
      const T pe = (j == 1) ? ((const T*)A.phi[0])[i] : (T)A.beta[j - 1] * ((const T*)A.phi[j - 1])[i];
      cur = cur - pe;
      ((T*)A.out[j])[i] = cur;
    }
  }
}

}  // namespace mi
