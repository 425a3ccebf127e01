// Stateless plane kernels (family (B) of include/mi_ode.h): the fused equivalents of the
// reference's eager-op sequences, for right-hand sides that are arbitrary Python callables.
// All are streaming elementwise / reduction kernels => HBM-bound.
#pragma once
#include "mi_ode_dev.h"

namespace mi {

struct LincombArgs {
  const void* base;
  const void* x[MI_ODE_MAX_LINCOMB];
  double coef[MI_ODE_MAX_LINCOMB];
  double scale;
  const double* scale_dev;     // non-null: the scale is read from device memory (graph replay with a new dt)
  long long n;
  void* out;
  int nx;
};

// out = base + add_n([(scale * c_j) * x_j])   -- misc._scaled_dot_product (misc.py:118-121), zeros not skipped
template <typename T>
__global__ __launch_bounds__(256) void k_lincomb(LincombArgs A) {
  const T scale = A.scale_dev != nullptr ? (T)*A.scale_dev : (T)A.scale;
  T* out = (T*)A.out;
  const T* base = (const T*)A.base;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (long long)gridDim.x * blockDim.x) {
    T acc = (scale * (T)A.coef[0]) * ((const T*)A.x[0])[i];
    for (int j = 1; j < A.nx; ++j) acc = acc + (scale * (T)A.coef[j]) * ((const T*)A.x[j])[i];
    out[i] = base != nullptr ? base[i] + acc : acc;
  }
}

// block records {max|y0|, max|y1|, sum err^2, -, nonfinite(y0)}   (misc.py:256-263 + dopri5.py:99-100)
template <typename T>
__global__ __launch_bounds__(256) void k_error_norms(const T* err, const T* y0, const T* y1, long long n, double* part) {
  Acc acc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const T a = y0[i], b = y1[i], e = err[i];
    acc.maxa = fmax(acc.maxa, (double)fabs(a));
    acc.maxb = fmax(acc.maxb, (double)fabs(b));
    acc.suma += (double)e * (double)e;
    if (!finite_(a)) acc.flag = 1;
  }
  __shared__ double red[80];
  block_reduce_store(acc, red, part + (long long)blockIdx.x * kRec);
}

// block records {-, -, sum ((x - xsub) / (atol + |y0| rtol))^2}   (misc.py:225-237)
template <typename T>
__global__ __launch_bounds__(256) void k_scaled_sumsq(const T* x, const T* xsub, const T* y0, long long n, double rtol,
                                                      double atol, double* part) {
  Acc acc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const T sc = (T)atol + fabs(y0[i]) * (T)rtol;
    const T v = xsub != nullptr ? x[i] - xsub[i] : x[i];
    const double q = (double)(v / sc);
    acc.suma += q * q;
  }
  __shared__ double red[80];
  block_reduce_store(acc, red, part + (long long)blockIdx.x * kRec);
}

// block records {-, -, -, -, flag}: flag = some element fails |a - b| < atol + rtol * max(|a|, |b|)   (misc.py:129-134)
template <typename T>
__global__ __launch_bounds__(256) void k_not_converged(const T* a, const T* b, long long n, double rtol, double atol, double* part) {
  Acc acc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const T x = a[i], y = b[i];
    const T tol = (T)atol + (T)rtol * fmax(fabs(x), fabs(y));
    if (!(fabs(x - y) < tol)) acc.flag = 1;
  }
  __shared__ double red[80];
  block_reduce_store(acc, red, part + (long long)blockIdx.x * kRec);
}

}  // namespace mi
