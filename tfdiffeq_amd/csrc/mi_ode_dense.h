// Dense-output arithmetic shared by the emission kernels and the whole-attempt kernels
// (interp.py:6-67 quartic through y0, y_mid, y1; tsit5.py:33-50 seven-weight polynomial).
#pragma once
#include "mi_ode_dev.h"

namespace mi {

// tsit5.py:33-42
__device__ __forceinline__ void tsit5_weights(double t, double* b) {
  const double t2 = t * t;
  b[0] = -1.0530884977290216 * t * (t - 1.3299890189751412) * (t2 - 1.4364028541716351 * t + 0.7139816917074209);
  b[1] = 0.1017 * t2 * (t2 - 2.1966568338249754 * t + 1.2949852507374631);
  b[2] = 2.490627285651252793 * t2 * (t2 - 2.38535645472061657 * t + 1.57803468208092486);
  b[3] = -16.54810288924490272 * (t - 1.21712927295533244) * (t - 0.61620406037800089) * t2;
  b[4] = 47.37952196281928122 * (t - 1.203071208372362603) * (t - 0.658047292653547382) * t2;
  b[5] = -34.87065786149660974 * (t - 1.2) * (t - 0.666666666666666667) * t2;
  b[6] = 2.5 * (t - 1) * (t - 0.6) * t2;
}

// interp._interp_fit (interp.py:6-36) for one element, given y_mid: quartic coefficients a..e
template <typename T>
__device__ __forceinline__ void quartic_from_mid(T y0, T y1, T ym, T f0, T f1, T dt, T* co) {
  // _dot_product = python sum(): ((((0 + c0*f0) + c1*f1) + c2*y0) + c3*y1) + c4*ym
  co[0] = ((((T)-2 * dt) * f0 + ((T)2 * dt) * f1) + (T)-8 * y0 + (T)-8 * y1) + (T)16 * ym;
  co[1] = ((((T)5 * dt) * f0 + ((T)-3 * dt) * f1) + (T)18 * y0 + (T)14 * y1) + (T)-32 * ym;
  co[2] = ((((T)-4 * dt) * f0 + dt * f1) + (T)-11 * y0 + (T)-5 * y1) + (T)16 * ym;
  co[3] = dt * f0;
  co[4] = y0;
}

template <typename T, int NK>
__device__ __forceinline__ void quartic_fit(T y0, T y1, const T* k, T dt, const InterpParams& I, T* co) {
  T ym = (dt * (T)I.c_mid[0]) * k[0];                     // dopri5.py:42 via misc.py:121
#pragma unroll
  for (int j = 1; j < NK; ++j) ym = ym + (dt * (T)I.c_mid[j]) * k[j];
  ym = y0 + ym;
  quartic_from_mid<T>(y0, y1, ym, k[0], k[NK - 1], dt, co);
}

// interp._interp_evaluate (interp.py:39-67): x in the STATE dtype
template <typename T>
__device__ __forceinline__ T quartic_eval(const T* co, T x) {
  const T x2 = x * x, x3 = x2 * x, x4 = x3 * x;
  return (((co[0] * x4 + co[1] * x3) + co[2] * x2) + co[3] * x) + co[4] * (T)1;
}

template <typename T>
__device__ __forceinline__ T interp_x(double t0, double t1, double t) {
  const T a = (T)t0, b = (T)t1, c = (T)t;
  return (T)((c - a) / (b - a));
}

// tsit5._interp_eval_tsit5 (tsit5.py:45-50) for one element; NK == 7
template <typename T, int NK>
__device__ __forceinline__ T tsit5_dense(T y0, const T* k, double t0, double t1, double t, int kind) {
  const double dt = t1 - t0;                               // tsit5.py:46
  double b[7];
  tsit5_weights((t - t0) / dt, b);
  const T base = (kind == MI_ODE_INTERP_TSIT5_REF) ? k[0] : y0;   // tsit5.py:47 starts from k[0] = f0 (defect F6b)
  T s = ((T)(dt * b[0])) * k[0];
#pragma unroll
  for (int q = 1; q < (NK < 7 ? NK : 7); ++q) s = s + ((T)(dt * b[q])) * k[q];
  return base + s;
}

}  // namespace mi
