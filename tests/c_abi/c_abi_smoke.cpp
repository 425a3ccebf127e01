// Standalone user of the C ABI (include/mi_ode.h): no Python, no torch - hipMalloc'ed buffers, plain C calls.
//   1. fixed grid RK4 (3/8 rule) on a Lorenz batch vs a scalar host loop written here   (bit-exact expected)
//   2. adaptive Dopri5 on the same batch vs the fine RK4 solution                        (1e-6)
//   3. the stateless plane kernel mi_ode_lincomb vs a host loop                          (bit-exact expected)
//   3b. tuple state, 4. fused adjoint interval, 5. fixed-grid Adams-Bashforth in one launch vs a host loop, 6. 300000 trajectories in one launch, 7. the variable-order Adams solver in one launch
//   8. family C (ABI 11): f evaluated by a kernel of THIS program between the library's launches, the controller on the device, one
//      attempt captured as a hipGraph and replayed in blind chunks
//   9. (ABI 12) mi_ode_outer_reduce vs a host loop
//  10. (ABI 13) mi_ode_linadj_segment: a backward interval of the linear system's adjoint in one launch vs a fine RK4 solve written here
//  11. (round 5) a float64 MLP through the plain engine entry points vs a fine RK4 solve of the network written here (round 5: the cooperative
//      kernels; since round 6 a network of this size runs on the float64 MFMA tile kernels, csrc/mi_ode_mlp64.h - same calls, same bars)
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I include tests/c_abi/c_abi_smoke.cpp -L tfdiffeq_amd -lmi_ode
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mi_ode.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define MI(x) do { int r_ = (x); if (r_ < 0) { printf("mi_ode error %d: %s (%s)\n", r_, mi_ode_last_error(), #x); return 3; } } while (0)

// the CALLER's right-hand side for section 8 (family C: opaque RHS): a kernel of this program, not of the library
__global__ void k_user_lorenz(const double* y, double* f, long long rows) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const double a = y[3 * r], b = y[3 * r + 1], c = y[3 * r + 2];
  f[3 * r] = 10.0 * (b - a);
  f[3 * r + 1] = a * (28.0 - c) - b;
  f[3 * r + 2] = a * b - (8.0 / 3.0) * c;
}

static void lorenz(const double* y, double* f) {
  f[0] = 10.0 * (y[1] - y[0]);
  f[1] = y[0] * (28.0 - y[2]) - y[1];
  f[2] = y[0] * y[1] - (8.0 / 3.0) * y[2];
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);                     // (a sanitizer abort at process exit must not swallow the report lines)
  const int B = 1000, D = 3, T = 41;
  const long long n = (long long)B * D;
  std::vector<double> y0(n), t(T);
  for (int b = 0; b < B; ++b) { y0[3 * b] = 1.0 + 1e-3 * sin(b); y0[3 * b + 1] = 1.0 + 1e-3 * cos(b); y0[3 * b + 2] = 1.0 + 1e-4 * b / B; }
  for (int i = 0; i < T; ++i) t[i] = 0.0125 * i;
  // host reference: rk_common.rk4_alt_step_func arithmetic
  std::vector<double> ref((size_t)T * n);
  memcpy(ref.data(), y0.data(), n * sizeof(double));
  for (int b = 0; b < B; ++b) {
    double y[3] = {y0[3 * b], y0[3 * b + 1], y0[3 * b + 2]};
    for (int i = 0; i + 1 < T; ++i) {
      const double dt = t[i + 1] - t[i];
      double k1[3], k2[3], k3[3], k4[3], ys[3];
      lorenz(y, k1);
      for (int d = 0; d < 3; ++d) ys[d] = y[d] + dt * k1[d] / 3;
      lorenz(ys, k2);
      for (int d = 0; d < 3; ++d) ys[d] = y[d] + dt * (k1[d] / -3 + k2[d]);
      lorenz(ys, k3);
      for (int d = 0; d < 3; ++d) ys[d] = y[d] + dt * (k1[d] - k2[d] + k3[d]);
      lorenz(ys, k4);
      for (int d = 0; d < 3; ++d) y[d] = y[d] + (k1[d] + 3 * k2[d] + 3 * k3[d] + k4[d]) * (dt / 8);
      for (int d = 0; d < 3; ++d) ref[(size_t)(i + 1) * n + 3 * b + d] = y[d];
    }
  }
  double *d_y0, *d_out;
  CK(hipMalloc(&d_y0, n * sizeof(double)));
  CK(hipMalloc(&d_out, (size_t)T * n * sizeof(double)));
  CK(hipMemcpy(d_y0, y0.data(), n * sizeof(double), hipMemcpyHostToDevice));

  // ---- 1. fixed grid RK4 -------------------------------------------------------------------------
  mi_ode_desc d;
  memset(&d, 0, sizeof(d));
  d.dtype = MI_ODE_F64; d.adaptive = 0; d.batch = B; d.dim = D;
  d.tableau.n_stages = 3;
  d.rhs.kind = MI_ODE_RHS_LORENZ; d.rhs.sign = 1.0; d.rhs.scalars[0] = 10.0; d.rhs.scalars[1] = 8.0 / 3.0; d.rhs.scalars[2] = 28.0;
  d.first_step = NAN;
  mi_ode_handle h = nullptr;
  MI(mi_ode_create(&d, &h));
  mi_ode_stats st;
  MI(mi_ode_fixed_grid_integrate(h, d_y0, t.data(), T, d_out, &st, nullptr));
  std::vector<double> out((size_t)T * n);
  CK(hipMemcpy(out.data(), d_out, (size_t)T * n * sizeof(double), hipMemcpyDeviceToHost));
  double maxdiff = 0;
  for (size_t i = 0; i < out.size(); ++i) maxdiff = fmax(maxdiff, fabs(out[i] - ref[i]));
  printf("rk4 fixed grid: max |gpu - host| = %.3e, nfe %lld, launches %lld\n", maxdiff, (long long)st.nfe, (long long)st.n_launches);
  if (maxdiff != 0.0) { printf("FAIL rk4 not bit-exact\n"); return 1; }
  MI(mi_ode_destroy(h));

  // ---- 2. adaptive dopri5 --------------------------------------------------------------------------
  const double alpha[6] = {1 / 5., 3 / 10., 4 / 5., 8 / 9., 1., 1.};
  const double beta[6][6] = {{1 / 5.}, {3 / 40., 9 / 40.}, {44 / 45., -56 / 15., 32 / 9.},
                             {19372 / 6561., -25360 / 2187., 64448 / 6561., -212 / 729.},
                             {9017 / 3168., -355 / 33., 46732 / 5247., 49 / 176., -5103 / 18656.},
                             {35 / 384., 0, 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84.}};
  const double c_sol[7] = {35 / 384., 0, 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84., 0};
  const double c_err[7] = {35 / 384. - 1951 / 21600., 0, 500 / 1113. - 22642 / 50085., 125 / 192. - 451 / 720.,
                           -2187 / 6784. - -12231 / 42400., 11 / 84. - 649 / 6300., -1. / 60.};
  const double c_mid[7] = {6025192743 / 30085553152. / 2, 0, 51252292925 / 65400821598. / 2, -2691868925 / 45128329728. / 2,
                           187940372067 / 1594534317056. / 2, -1776094331 / 19743644256. / 2, 11237099 / 235043384. / 2};
  d.adaptive = 1;
  d.tableau.n_stages = 6; d.tableau.fsal = 1;
  for (int i = 0; i < 6; ++i) { d.tableau.alpha[i] = alpha[i]; for (int j = 0; j < 6; ++j) d.tableau.beta[i][j] = beta[i][j]; }
  for (int j = 0; j < 7; ++j) { d.tableau.c_sol[j] = c_sol[j]; d.tableau.c_error[j] = c_err[j]; d.tableau.c_mid[j] = c_mid[j]; }
  d.controller = MI_ODE_CTRL_MISC; d.interp = MI_ODE_INTERP_QUARTIC_MID; d.order = 5; d.init_order = 4;
  d.rtol = 1e-9; d.atol = 1e-11; d.safety = (double)0.9f; d.ifactor = 10.0; d.dfactor = (double)0.2f;
  d.max_num_steps = 100000;
  MI(mi_ode_create(&d, &h));
  const double t2[3] = {0.0, 0.25, 0.5};
  int bits = mi_ode_integrate(h, d_y0, t2, 3, d_out, &st, nullptr);
  if (bits != 0) { printf("FAIL dopri5 status %d (%s) %s\n", bits, bits > 0 ? mi_ode_status_string(bits) : "", mi_ode_last_error()); return 1; }
  CK(hipMemcpy(out.data(), d_out, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost));
  // fine host reference: the same RK4 with 32 sub-steps per grid interval (error ~1e-10)
  double md2 = 0;
  for (int b = 0; b < B; ++b) {
    double y[3] = {y0[3 * b], y0[3 * b + 1], y0[3 * b + 2]};
    const int sub = 32;
    for (int i = 0; i < 40 * sub; ++i) {
      const double dt = 0.0125 / sub;
      double k1[3], k2[3], k3[3], k4[3], ys[3];
      lorenz(y, k1);
      for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * k1[q] / 3;
      lorenz(ys, k2);
      for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * (k1[q] / -3 + k2[q]);
      lorenz(ys, k3);
      for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * (k1[q] - k2[q] + k3[q]);
      lorenz(ys, k4);
      for (int q = 0; q < 3; ++q) y[q] = y[q] + (k1[q] + 3 * k2[q] + 3 * k3[q] + k4[q]) * (dt / 8);
      if (i + 1 == 20 * sub) for (int q = 0; q < 3; ++q) md2 = fmax(md2, fabs(out[n + 3 * b + q] - y[q]));
      if (i + 1 == 40 * sub) for (int q = 0; q < 3; ++q) md2 = fmax(md2, fabs(out[2 * n + 3 * b + q] - y[q]));
    }
  }
  printf("dopri5 adaptive: attempts %lld accepted %lld nfe %lld launches %lld polls %d, max |dopri5 - fine rk4| = %.3e\n",
         (long long)st.n_attempts, (long long)st.n_accepted, (long long)st.nfe, (long long)st.n_launches, st.n_polls, md2);
  if (!(md2 < 1e-6)) { printf("FAIL dopri5 accuracy\n"); return 1; }
  MI(mi_ode_destroy(h));

  // ---- 3. plane kernel ------------------------------------------------------------------------------
  double* d_tmp;
  CK(hipMalloc(&d_tmp, n * sizeof(double)));
  const void* xs[2] = {d_y0, d_out};
  const double coef[2] = {0.3, -1.25};
  MI(mi_ode_lincomb(MI_ODE_F64, n, d_y0, xs, coef, 2, 0.0625, d_tmp, nullptr));
  std::vector<double> lc(n), o0(n);
  CK(hipMemcpy(lc.data(), d_tmp, n * sizeof(double), hipMemcpyDeviceToHost));
  CK(hipMemcpy(o0.data(), d_out, n * sizeof(double), hipMemcpyDeviceToHost));
  double md3 = 0;
  for (long long i = 0; i < n; ++i) {
    const double acc = (0.0625 * 0.3) * y0[i] + (0.0625 * -1.25) * o0[i];
    md3 = fmax(md3, fabs(lc[i] - (y0[i] + acc)));
  }
  printf("lincomb: max diff %.3e\n", md3);
  if (md3 != 0.0) { printf("FAIL lincomb not bit-exact\n"); return 1; }
  // ---- 3b. tuple state: two components (the same 1000 rows twice, the second scaled by 1e-3 in its perturbation) -------
  {
    const int AL = MI_ODE_SEGMENT_ALIGN;
    const long long pad = ((long long)B + AL - 1) / AL * AL;           // rows per padded component
    std::vector<double> packed((size_t)2 * pad * D, 0.0);
    memcpy(packed.data(), y0.data(), n * sizeof(double));
    memcpy(packed.data() + pad * D, y0.data(), n * sizeof(double));
    double *d_p, *d_po, *d_single;
    CK(hipMalloc(&d_p, packed.size() * sizeof(double)));
    CK(hipMalloc(&d_po, 2 * packed.size() * sizeof(double)));
    CK(hipMalloc(&d_single, 2 * n * sizeof(double)));
    CK(hipMemcpy(d_p, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice));
    const double t3[2] = {0.0, 0.5};
    mi_ode_desc dt_ = d;                                                // the dopri5 descriptor of section 2
    dt_.batch = 2 * pad; dt_.n_segments = 2; dt_.seg_rows[0] = B; dt_.seg_rows[1] = B;
    mi_ode_handle ht = nullptr, hs = nullptr;
    MI(mi_ode_create(&dt_, &ht));
    bits = mi_ode_integrate(ht, d_p, t3, 2, d_po, &st, nullptr);
    if (bits != 0) { printf("FAIL tuple state status %d %s\n", bits, mi_ode_last_error()); return 1; }
    const long long tuple_attempts = st.n_attempts, tuple_launches = st.n_launches;
    MI(mi_ode_create(&d, &hs));                                          // the single tensor of the same rows
    bits = mi_ode_integrate(hs, d_y0, t3, 2, d_single, &st, nullptr);
    if (bits != 0) { printf("FAIL single state status %d\n", bits); return 1; }
    std::vector<double> a(n), b2(n), c(n);
    CK(hipMemcpy(a.data(), d_po + packed.size(), n * sizeof(double), hipMemcpyDeviceToHost));                 // t = 0.5, component 0
    CK(hipMemcpy(b2.data(), d_po + packed.size() + pad * D, n * sizeof(double), hipMemcpyDeviceToHost));      // component 1
    CK(hipMemcpy(c.data(), d_single + n, n * sizeof(double), hipMemcpyDeviceToHost));
    double md = 0;
    for (long long i = 0; i < n; ++i) md = fmax(md, fmax(fabs(a[i] - c[i]), fabs(b2[i] - c[i])));
    printf("tuple state (2 components, segment table): attempts %lld (single tensor: %lld), launches %lld, max |component - single| = %.3e\n",
           tuple_attempts, (long long)st.n_attempts, tuple_launches, md);
    // two identical components have identical error ratios: the max() over them is the single tensor's ratio - same steps, same bits
    if (md != 0.0 || tuple_attempts != st.n_attempts || tuple_launches != 1) { printf("FAIL tuple state\n"); return 1; }
    MI(mi_ode_destroy(ht)); MI(mi_ode_destroy(hs));
  }

  // ---- 4. fused backward interval of odeint_adjoint (tanh MLP 3 -> 8 -> 8 -> 3, fp32) -----------------------------
  {
    const int AB = 50, AD = 3, AH = 8;
    const int P = AD * AH + AH + AH * AH + AH + AH * AD + AD;
    std::vector<float> W1(AD * AH), b1(AH), W2(AH * AH), b2(AH), W3(AH * AD), b3(AD), ya(AB * AD), aa(AB * AD);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto* v : {&W1, &b1, &W2, &b2, &W3, &b3, &ya, &aa}) for (float& x : *v) x = rnd();
    float *dW1, *db1, *dW2, *db2, *dW3, *db3, *dy, *da, *df, *dvy, *dvp, *dth, *dth1, *dat, *dat1, *da1;
    auto up = [&](float** d, const std::vector<float>& h) { if (hipMalloc(d, h.size() * 4) != hipSuccess) return 1; return (int)hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice); };
    if (up(&dW1, W1) || up(&db1, b1) || up(&dW2, W2) || up(&db2, b2) || up(&dW3, W3) || up(&db3, b3) || up(&dy, ya) || up(&da, aa)) return 1;
    CK(hipMalloc(&df, AB * AD * 4)); CK(hipMalloc(&dvy, AB * AD * 4)); CK(hipMalloc(&da1, AB * AD * 4));
    CK(hipMalloc(&dvp, P * 4)); CK(hipMalloc(&dth, P * 4)); CK(hipMalloc(&dth1, P * 4)); CK(hipMalloc(&dat, 4)); CK(hipMalloc(&dat1, 4));
    CK(hipMemset(dth, 0, P * 4));
    const float at0 = 0.25f;
    CK(hipMemcpy(dat, &at0, 4, hipMemcpyHostToDevice));
    mi_ode_adjoint_desc ad;
    memset(&ad, 0, sizeof(ad));
    ad.batch = AB; ad.dim = AD; ad.hidden = AH; ad.tableau = d.tableau;
    ad.rtol = 1e-4; ad.atol = 1e-4; ad.safety = (double)0.9f; ad.ifactor = 10.0; ad.dfactor = (double)0.2f; ad.order = 5; ad.init_order = 4;
    ad.max_num_steps = 1000;
    mi_ode_rhs r;
    memset(&r, 0, sizeof(r));
    r.kind = MI_ODE_RHS_MLP_TANH; r.hidden = AH; r.sign = 1.0;
    r.w[0] = dW1; r.w[1] = dW2; r.w[2] = dW3; r.b[0] = db1; r.b[1] = db2; r.b[2] = db3;
    mi_ode_adjoint_handle ah = nullptr;
    MI(mi_ode_adjoint_create(&ad, &ah));
    if (mi_ode_adjoint_num_params(ah) != P) { printf("FAIL adjoint parameter count\n"); return 1; }
    MI(mi_ode_adjoint_dynamics(ah, &r, dy, da, df, dvy, dvp, nullptr));
    std::vector<float> f(AB * AD), vy(AB * AD), vp(P);
    CK(hipMemcpy(f.data(), df, f.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(vy.data(), dvy, vy.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(vp.data(), dvp, vp.size() * 4, hipMemcpyDeviceToHost));
    // host: f, -a^T df/dy, -a^T df/dtheta in double
    std::vector<double> rvp(P, 0.0);
    double mf = 0, mv = 0, mp = 0, sp = 0;
    for (int b = 0; b < AB; ++b) {
      double h1[AH], h2[AH], g2[AH], g1[AH];
      for (int j = 0; j < AH; ++j) { double z = b1[j]; for (int i = 0; i < AD; ++i) z += ya[b * AD + i] * W1[i * AH + j]; h1[j] = tanh(z); }
      for (int j = 0; j < AH; ++j) { double z = b2[j]; for (int i = 0; i < AH; ++i) z += h1[i] * W2[i * AH + j]; h2[j] = tanh(z); }
      for (int k = 0; k < AD; ++k) { double z = b3[k]; for (int j = 0; j < AH; ++j) z += h2[j] * W3[j * AD + k]; mf = fmax(mf, fabs(z - f[b * AD + k])); }
      for (int j = 0; j < AH; ++j) { double z = 0; for (int k = 0; k < AD; ++k) z += aa[b * AD + k] * W3[j * AD + k]; g2[j] = z * (1 - h2[j] * h2[j]); }
      for (int i = 0; i < AH; ++i) { double z = 0; for (int j = 0; j < AH; ++j) z += g2[j] * W2[i * AH + j]; g1[i] = z * (1 - h1[i] * h1[i]); }
      for (int i = 0; i < AD; ++i) { double z = 0; for (int j = 0; j < AH; ++j) z += g1[j] * W1[i * AH + j]; mv = fmax(mv, fabs(-z - vy[b * AD + i])); }
      int o = 0;
      for (int i = 0; i < AD; ++i) for (int j = 0; j < AH; ++j) rvp[o++] -= ya[b * AD + i] * g1[j];
      for (int j = 0; j < AH; ++j) rvp[o++] -= g1[j];
      for (int i = 0; i < AH; ++i) for (int j = 0; j < AH; ++j) rvp[o++] -= h1[i] * g2[j];
      for (int j = 0; j < AH; ++j) rvp[o++] -= g2[j];
      for (int j = 0; j < AH; ++j) for (int k = 0; k < AD; ++k) rvp[o++] -= h2[j] * aa[b * AD + k];
      for (int k = 0; k < AD; ++k) rvp[o++] -= aa[b * AD + k];
    }
    for (int i = 0; i < P; ++i) { mp = fmax(mp, fabs(rvp[i] - vp[i])); sp = fmax(sp, fabs(rvp[i])); }
    printf("adjoint dynamics: max |f - host| %.2e, |vjp_y - host| %.2e, |vjp_params - host| %.2e (scale %.2e)\n", mf, mv, mp, sp);
    if (!(mf < 2e-6 && mv < 2e-6 && mp < 1e-5 * fmax(sp, 1.0))) { printf("FAIL adjoint dynamics\n"); return 1; }
    mi_ode_stats ast;
    bits = mi_ode_adjoint_segment(ah, &r, dy, da, dat, dth, 1.0, 0.0, nullptr, da1, dat1, dth1, &ast, nullptr);
    if (bits != 0) { printf("FAIL adjoint interval status %d %s\n", bits, mi_ode_last_error()); return 1; }
    float at1 = 0;
    CK(hipMemcpy(&at1, dat1, 4, hipMemcpyDeviceToHost));
    printf("adjoint interval 1 -> 0: attempts %lld accepted %lld launches %lld, adj_t %g -> %g\n", (long long)ast.n_attempts,
           (long long)ast.n_accepted, (long long)ast.n_launches, at0, at1);
    if (ast.n_launches != 1 || ast.n_accepted < 1 || at1 != at0) { printf("FAIL adjoint interval\n"); return 1; }
    MI(mi_ode_adjoint_destroy(ah));
  }
  // ---- 5. fixed-grid Adams-Bashforth in one launch (mi_ode_desc.multistep) vs a host loop ------------------
  {
    // tables as the reference forms them in Python floats: (1 / divisor) * c_j, orders 1..4 (fixed_adams.py:9-86)
    static double ab[13 * 12], am[13 * 12], am0[13];
    memset(ab, 0, sizeof(ab)); memset(am, 0, sizeof(am)); memset(am0, 0, sizeof(am0));
    const double cab[5][4] = {{0}, {1}, {3, -1}, {23, -16, 5}, {55, -59, 37, -9}};
    const double dab[5] = {0, 1, 2, 12, 24};
    for (int o = 1; o <= 4; ++o) for (int j = 0; j < o; ++j) ab[o * 12 + j] = (1 / dab[o]) * cab[o][j];
    const int MO = 5;                                                   // max_order 5: history of 4 derivatives -> AB4
    mi_ode_desc dm;
    memset(&dm, 0, sizeof(dm));
    dm.dtype = MI_ODE_F64; dm.adaptive = 0; dm.batch = B; dm.dim = D; dm.tableau.n_stages = 3; dm.first_step = NAN;
    dm.rhs = d.rhs;
    dm.multistep = 1; dm.ms_max_order = MO; dm.ms_max_iters = 4; dm.ms_min_order = 4; dm.ms_ab = ab; dm.ms_am = am; dm.ms_am0 = am0;
    dm.rtol = 1e-7; dm.atol = 1e-9;
    mi_ode_handle hm = nullptr;
    MI(mi_ode_create(&dm, &hm));
    MI(mi_ode_fixed_grid_integrate(hm, d_y0, t.data(), T, d_out, &st, nullptr));
    CK(hipMemcpy(out.data(), d_out, (size_t)T * n * sizeof(double), hipMemcpyDeviceToHost));
    double md5 = 0;
    for (int b = 0; b < B; ++b) {
      double y[3] = {y0[3 * b], y0[3 * b + 1], y0[3 * b + 2]}, hist[4][3];
      int len = 0;
      for (int i = 0; i + 1 < T; ++i) {
        const double dt = t[i + 1] - t[i];
        double fn[3], dy[3];
        lorenz(y, fn);
        for (int j = 3; j > 0; --j) memcpy(hist[j], hist[j - 1], sizeof(fn));
        memcpy(hist[0], fn, sizeof(fn));
        len = len + 1 < MO - 1 ? len + 1 : MO - 1;
        if (len < 4 - 1) {                                              // start-up: RK4 3/8 rule with k1 = history[0]
          double k2[3], k3[3], k4[3], ys[3];
          for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * fn[q] / 3;
          lorenz(ys, k2);
          for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * (fn[q] / -3 + k2[q]);
          lorenz(ys, k3);
          for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * (fn[q] - k2[q] + k3[q]);
          lorenz(ys, k4);
          for (int q = 0; q < 3; ++q) dy[q] = (fn[q] + 3 * k2[q] + 3 * k3[q] + k4[q]) * (dt / 8);
        } else {
          for (int q = 0; q < 3; ++q) {
            double a_ = ab[len * 12] * hist[0][q];
            for (int j = 1; j < len; ++j) a_ = a_ + ab[len * 12 + j] * hist[j][q];
            dy[q] = dt * a_;
          }
        }
        for (int q = 0; q < 3; ++q) { y[q] = y[q] + dy[q]; md5 = fmax(md5, fabs(out[(size_t)(i + 1) * n + 3 * b + q] - y[q])); }
      }
    }
    printf("explicit Adams (one launch): max |gpu - host| = %.3e, launches %lld\n", md5, (long long)st.n_launches);
    if (!(md5 <= 1e-12) || st.n_launches != 1) { printf("FAIL multistep\n"); return 1; }
    MI(mi_ode_destroy(hm));
  }

  // ---- 6. more trajectories than one per thread keeps co-resident: the whole call is still one launch (state in HBM planes) --
  {
    const long long BB = 300000, nn = BB * D;
    std::vector<double> yb((size_t)nn);
    for (long long b = 0; b < BB; ++b) for (int q = 0; q < 3; ++q) yb[3 * b + q] = y0[3 * (b % B) + q];      // the 1000 rows, repeated
    double *d_yb, *d_ob;
    CK(hipMalloc(&d_yb, nn * sizeof(double))); CK(hipMalloc(&d_ob, 2 * nn * sizeof(double)));
    CK(hipMemcpy(d_yb, yb.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    mi_ode_desc db_ = d;                                                 // the dopri5 descriptor of section 2
    db_.batch = BB;
    mi_ode_handle hb = nullptr, hs = nullptr;
    MI(mi_ode_create(&db_, &hb));
    const double t6[2] = {0.0, 0.5};
    bits = mi_ode_integrate(hb, d_yb, t6, 2, d_ob, &st, nullptr);
    if (bits != 0) { printf("FAIL large batch status %d %s\n", bits, mi_ode_last_error()); return 1; }
    const long long big_launches = st.n_launches, big_attempts = st.n_attempts;
    MI(mi_ode_create(&d, &hs));
    bits = mi_ode_integrate(hs, d_y0, t6, 2, d_out, &st, nullptr);
    if (bits != 0) { printf("FAIL status %d\n", bits); return 1; }
    std::vector<double> a((size_t)nn), c(n);
    CK(hipMemcpy(a.data(), d_ob + nn, nn * sizeof(double), hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.data(), d_out + n, n * sizeof(double), hipMemcpyDeviceToHost));
    double md6 = 0;                                                      // 300 copies of the same rows: the norms are those of the 1000 -> same steps
    for (long long i = 0; i < nn; ++i) md6 = fmax(md6, fabs(a[i] - c[i % n]));
    printf("300000 trajectories: launches %lld, attempts %lld (1000 rows: %lld), max |row - its copy in the small run| = %.3e\n", big_launches,
           big_attempts, (long long)st.n_attempts, md6);
    if (big_launches != 1 || big_attempts != st.n_attempts || !(md6 <= 1e-12)) { printf("FAIL large batch\n"); return 1; }
    MI(mi_ode_destroy(hb)); MI(mi_ode_destroy(hs));
  }
  // ---- 7. the variable-order Adams solver in one launch (mi_ode_desc.multistep = 3, ABI 10) vs the fine RK4 solution ----------
  {
    const double gamma_star[13] = {1, -0.5, -0.08333333333333333, -0.041666666666666664, -0.02638888888888889, -0.01875, -0.014269179894179895,
                                   -0.01136739417989418, -0.00935653659611993, -0.00789255, -0.00678585, -0.00592406, -0.00523669};   // adams.py:15-18
    mi_ode_desc dv = d;                                                  // the dopri5 descriptor of section 2: rtol 1e-9, atol 1e-11, misc controller
    memset(&dv.tableau, 0, sizeof(dv.tableau));
    dv.multistep = 3; dv.ms_max_order = 12; dv.ms_gamma_star = gamma_star;
    mi_ode_handle hv = nullptr;
    MI(mi_ode_create(&dv, &hv));
    const double t7[3] = {0.0, 0.25, 0.5};
    bits = mi_ode_integrate(hv, d_y0, t7, 3, d_out, &st, nullptr);
    if (bits != 0) { printf("FAIL adams status %d %s\n", bits, mi_ode_last_error()); return 1; }
    CK(hipMemcpy(out.data(), d_out, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost));
    double md7 = 0;
    for (int b = 0; b < B; ++b) {
      double y[3] = {y0[3 * b], y0[3 * b + 1], y0[3 * b + 2]};
      const int sub = 32;
      for (int i = 0; i < 40 * sub; ++i) {
        const double dt = 0.0125 / sub;
        double k1[3], k2[3], k3[3], k4[3], ys[3];
        lorenz(y, k1);
        for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * k1[q] / 3;
        lorenz(ys, k2);
        for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * (k1[q] / -3 + k2[q]);
        lorenz(ys, k3);
        for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * (k1[q] - k2[q] + k3[q]);
        lorenz(ys, k4);
        for (int q = 0; q < 3; ++q) y[q] = y[q] + (k1[q] + 3 * k2[q] + 3 * k3[q] + k4[q]) * (dt / 8);
        if (i + 1 == 20 * sub) for (int q = 0; q < 3; ++q) md7 = fmax(md7, fabs(out[n + 3 * b + q] - y[q]));
        if (i + 1 == 40 * sub) for (int q = 0; q < 3; ++q) md7 = fmax(md7, fabs(out[2 * n + 3 * b + q] - y[q]));
      }
    }
    printf("adams (variable order, one launch): attempts %lld accepted %lld nfe %lld launches %lld, max |adams - fine rk4| = %.3e\n",
           (long long)st.n_attempts, (long long)st.n_accepted, (long long)st.nfe, (long long)st.n_launches, md7);
    if (st.n_launches != 1 || st.n_accepted < 2 || !(md7 < 1e-4)) { printf("FAIL adams\n"); return 1; }
    MI(mi_ode_destroy(hv));
  }
  // ---- 8. family C (ABI 11): opaque right-hand side, controller on the device, one attempt recorded as a hipGraph and replayed ----
  {
    mi_ode_opq_desc od;
    memset(&od, 0, sizeof(od));
    od.dtype = MI_ODE_F64; od.n_comp = 1; od.n[0] = n;
    od.tableau = d.tableau;                                              // the dopri5 tableau of section 2
    od.controller = MI_ODE_CTRL_MISC; od.interp = MI_ODE_INTERP_QUARTIC_MID; od.order = 5; od.init_order = 4;
    od.rtol[0] = 1e-9; od.atol[0] = 1e-11; od.safety = (double)0.9f; od.ifactor = 10.0; od.dfactor = (double)0.2f; od.max_num_steps = 100000;
    mi_ode_opq_handle ho = nullptr;
    MI(mi_ode_opq_create(&od, &ho));
    hipStream_t s8;
    CK(hipStreamCreate(&s8));
    double *Y0, *F0, *YS[6], *K[7], *ts;
    CK(hipMalloc(&Y0, n * sizeof(double))); CK(hipMalloc(&F0, n * sizeof(double))); CK(hipMalloc(&ts, 6 * sizeof(double)));
    for (int i = 0; i < 6; ++i) { CK(hipMalloc(&YS[i], n * sizeof(double))); CK(hipMalloc(&K[i + 1], n * sizeof(double))); }
    K[0] = F0;
    CK(hipMemcpy(Y0, d_y0, n * sizeof(double), hipMemcpyDeviceToDevice));
    const dim3 ug((unsigned)((B + 255) / 256)), ub(256);
    hipLaunchKernelGGL(k_user_lorenz, ug, ub, 0, s8, (const double*)Y0, F0, (long long)B);           // f0 (dopri5.py:71)
    const double t8[2] = {0.25, 0.5};
    void* rows8[1] = {d_out + n};                                        // solution rows 1, 2 (row 0 = y0 is the caller's)
    MI(mi_ode_opq_begin(ho, 0.0, 1e-3, t8, 2, rows8, ts, s8));
    const double* dt_dev = mi_ode_opq_dt_dev(ho);
    auto attempt = [&]() -> int {                                        // rk_common.py:49-60 + dopri5.py:103-121: no host value inside
      for (int sg = 0; sg < 6; ++sg) {
        const void* ks[7];
        for (int j = 0; j <= sg; ++j) ks[j] = K[j];
        MI(mi_ode_lincomb_dev(MI_ODE_F64, n, Y0, ks, beta[sg], sg + 1, dt_dev, YS[sg], s8));
        hipLaunchKernelGGL(k_user_lorenz, ug, ub, 0, s8, (const double*)YS[sg], K[sg + 1], (long long)B);    // (Lorenz ignores ts[sg])
      }
      const void* y0p[1] = {Y0}; const void* y1p[1] = {YS[5]}; const void* kp[7];
      void* y0w[1] = {Y0}; void* f0w[1] = {F0};
      for (int j = 0; j < 7; ++j) kp[j] = K[j];
      MI(mi_ode_opq_finish(ho, y0p, y1p, kp, s8));
      MI(mi_ode_opq_commit(ho, y0w, f0w, y1p, kp, s8));
      return 0;
    };
    mi_ode_stats st8;
    int32_t done8 = 0;
    if (attempt() != 0) return 3;                                        // one eager attempt (warm-up), then record the next one
    MI(mi_ode_opq_poll(ho, &st8, &done8, s8));
    hipGraph_t graph; hipGraphExec_t gexec;
    CK(hipStreamBeginCapture(s8, hipStreamCaptureModeThreadLocal));
    if (attempt() != 0) return 3;
    CK(hipStreamEndCapture(s8, &graph));
    CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    int replays = 0, polls = 1;
    while (!done8 && replays < 100000) {
      for (int c = 0; c < 16; ++c) CK(hipGraphLaunch(gexec, s8));        // blind chunk: attempts after `done` are no-ops
      replays += 16;
      int rc8 = mi_ode_opq_poll(ho, &st8, &done8, s8);
      if (rc8 != 0) { printf("FAIL family C status %d %s\n", rc8, mi_ode_last_error()); return 1; }
      ++polls;
    }
    CK(hipMemcpy(out.data(), d_out, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost));
    double md8 = 0;
    for (int b = 0; b < B; ++b) {
      double y[3] = {y0[3 * b], y0[3 * b + 1], y0[3 * b + 2]};
      const int sub = 32;
      for (int i = 0; i < 40 * sub; ++i) {
        const double dt = 0.0125 / sub;
        double k1[3], k2[3], k3[3], k4[3], ys[3];
        lorenz(y, k1);
        for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * k1[q] / 3;
        lorenz(ys, k2);
        for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * (k1[q] / -3 + k2[q]);
        lorenz(ys, k3);
        for (int q = 0; q < 3; ++q) ys[q] = y[q] + dt * (k1[q] - k2[q] + k3[q]);
        lorenz(ys, k4);
        for (int q = 0; q < 3; ++q) y[q] = y[q] + (k1[q] + 3 * k2[q] + 3 * k3[q] + k4[q]) * (dt / 8);
        if (i + 1 == 20 * sub) for (int q = 0; q < 3; ++q) md8 = fmax(md8, fabs(out[n + 3 * b + q] - y[q]));
        if (i + 1 == 40 * sub) for (int q = 0; q < 3; ++q) md8 = fmax(md8, fabs(out[2 * n + 3 * b + q] - y[q]));
      }
    }
    printf("family C (opaque RHS, device controller, hipGraph replay): attempts %lld accepted %lld, %d replays, %d read-backs, max |dopri5 - fine rk4| = %.3e\n",
           (long long)st8.n_attempts, (long long)st8.n_accepted, replays, polls, md8);
    if (!done8 || st8.n_accepted < 2 || polls * 4 > (int)st8.n_attempts + 8 || !(md8 < 1e-6)) { printf("FAIL family C\n"); return 1; }
    CK(hipGraphExecDestroy(gexec)); CK(hipGraphDestroy(graph));
    MI(mi_ode_opq_destroy(ho));
    CK(hipStreamDestroy(s8));
  }
  // ---- 9. (ABI 12) mi_ode_outer_reduce: -(y^T a) and -sum_rows a of two [batch, dim] planes vs a host loop (ragged slab, padded width) ----
  {
    const int OB = 1003, OD = 20;
    std::vector<double> hy((size_t)OB * OD), ha((size_t)OB * OD), hw((size_t)OD * OD), hb(OD), rw((size_t)OD * OD, 0.0), rb(OD, 0.0);
    for (int r = 0; r < OB; ++r)
      for (int c = 0; c < OD; ++c) { hy[(size_t)r * OD + c] = sin(0.37 * r + c); ha[(size_t)r * OD + c] = cos(0.11 * r - 0.5 * c) / OB; }
    for (int r = 0; r < OB; ++r)
      for (int m = 0; m < OD; ++m) {
        for (int q = 0; q < OD; ++q) rw[(size_t)m * OD + q] -= hy[(size_t)r * OD + m] * ha[(size_t)r * OD + q];
        rb[m] -= ha[(size_t)r * OD + m];
      }
    double *dy9 = nullptr, *da9 = nullptr, *dw9 = nullptr, *db9 = nullptr;
    void* ws9 = nullptr;
    const int64_t wsb = mi_ode_outer_workspace_bytes(MI_ODE_F64, OB, OD);
    if (wsb <= 0) { printf("FAIL outer workspace size\n"); return 1; }
    CK(hipMalloc((void**)&dy9, hy.size() * 8)); CK(hipMalloc((void**)&da9, ha.size() * 8));
    CK(hipMalloc((void**)&dw9, hw.size() * 8)); CK(hipMalloc((void**)&db9, hb.size() * 8)); CK(hipMalloc(&ws9, (size_t)wsb));
    CK(hipMemcpy(dy9, hy.data(), hy.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(da9, ha.data(), ha.size() * 8, hipMemcpyHostToDevice));
    MI(mi_ode_outer_reduce(MI_ODE_F64, OB, OD, dy9, da9, -1.0, dw9, db9, ws9, nullptr));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hw.data(), dw9, hw.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), db9, hb.size() * 8, hipMemcpyDeviceToHost));
    double md9 = 0.0;
    for (size_t i = 0; i < hw.size(); ++i) md9 = fmax(md9, fabs(hw[i] - rw[i]));
    for (int i = 0; i < OD; ++i) md9 = fmax(md9, fabs(hb[i] - rb[i]));
    printf("outer_reduce (1003 x 20, fp64): max |device - host loop| = %.3e\n", md9);
    if (!(md9 < 1e-13)) { printf("FAIL outer_reduce\n"); return 1; }
    if (mi_ode_outer_reduce(MI_ODE_F64, OB, 200, dy9, da9, -1.0, dw9, db9, ws9, nullptr) >= 0) { printf("FAIL outer_reduce accepted dim 200\n"); return 1; }
    CK(hipFree(dy9)); CK(hipFree(da9)); CK(hipFree(dw9)); CK(hipFree(db9)); CK(hipFree(ws9));
  }
  // ---- 10. (ABI 13) mi_ode_linadj_*: one backward interval of odeint_adjoint for f = y W + b in ONE launch, t = 1 -> 0, against a fine RK4
  //          solve of the augmented system (y' = yW + b, a' = -a W^T, dW' = -y^T a, db' = -sum_rows a) written here ----
  {
    const int LB = 70, LD = 8, NP = LD * LD + LD;
    std::vector<double> W((size_t)LD * LD), bb(LD), y1((size_t)LB * LD), a1((size_t)LB * LD), th(NP, 0.0);
    for (int i = 0; i < LD; ++i) {
      bb[i] = 0.05 * cos(1.3 * i);
      for (int j = 0; j < LD; ++j) W[(size_t)i * LD + j] = (i == j ? -0.4 : 0.0) + 0.3 * sin(0.7 * i - 1.1 * j) / sqrt((double)LD);
    }
    for (int r = 0; r < LB; ++r)
      for (int c = 0; c < LD; ++c) { y1[(size_t)r * LD + c] = sin(0.3 * r + 0.9 * c); a1[(size_t)r * LD + c] = cos(0.17 * r - 0.4 * c) / LB; }
    // host reference
    std::vector<double> ry = y1, ra = a1, rth(NP, 0.0);
    auto F = [&](const std::vector<double>& y, const std::vector<double>& a, std::vector<double>& fy, std::vector<double>& fa, std::vector<double>& fth) {
      std::fill(fth.begin(), fth.end(), 0.0);
      for (int r = 0; r < LB; ++r)
        for (int c = 0; c < LD; ++c) {
          double sy = bb[c], sa = 0.0;
          for (int k = 0; k < LD; ++k) { sy += y[(size_t)r * LD + k] * W[(size_t)k * LD + c]; sa -= a[(size_t)r * LD + k] * W[(size_t)c * LD + k]; }
          fy[(size_t)r * LD + c] = sy; fa[(size_t)r * LD + c] = sa;
          fth[LD * LD + c] -= a[(size_t)r * LD + c];
          for (int k = 0; k < LD; ++k) fth[(size_t)k * LD + c] -= y[(size_t)r * LD + k] * a[(size_t)r * LD + c];
        }
    };
    {
      const int NS = 400;
      const double dt = -1.0 / NS;
      const size_t m = (size_t)LB * LD;
      std::vector<double> ky[4], ka[4], kt[4], ty(m), ta(m);
      for (int q = 0; q < 4; ++q) { ky[q].resize(m); ka[q].resize(m); kt[q].resize(NP); }
      for (int sidx = 0; sidx < NS; ++sidx) {
        F(ry, ra, ky[0], ka[0], kt[0]);
        for (size_t i = 0; i < m; ++i) { ty[i] = ry[i] + 0.5 * dt * ky[0][i]; ta[i] = ra[i] + 0.5 * dt * ka[0][i]; }
        F(ty, ta, ky[1], ka[1], kt[1]);
        for (size_t i = 0; i < m; ++i) { ty[i] = ry[i] + 0.5 * dt * ky[1][i]; ta[i] = ra[i] + 0.5 * dt * ka[1][i]; }
        F(ty, ta, ky[2], ka[2], kt[2]);
        for (size_t i = 0; i < m; ++i) { ty[i] = ry[i] + dt * ky[2][i]; ta[i] = ra[i] + dt * ka[2][i]; }
        F(ty, ta, ky[3], ka[3], kt[3]);
        for (size_t i = 0; i < m; ++i) {
          ry[i] += dt / 6 * (ky[0][i] + 2 * ky[1][i] + 2 * ky[2][i] + ky[3][i]);
          ra[i] += dt / 6 * (ka[0][i] + 2 * ka[1][i] + 2 * ka[2][i] + ka[3][i]);
        }
        for (int i = 0; i < NP; ++i) rth[i] += dt / 6 * (kt[0][i] + 2 * kt[1][i] + 2 * kt[2][i] + kt[3][i]);
      }
    }
    mi_ode_linadj_desc ld;
    memset(&ld, 0, sizeof(ld));
    if (mi_ode_sizeof(8) != (int64_t)sizeof(ld)) { printf("FAIL sizeof(mi_ode_linadj_desc)\n"); return 1; }
    ld.batch = LB; ld.dim = LD; ld.dtype = MI_ODE_F64;
    ld.tableau = d.tableau;                                              // the dopri5 tableau of section 2
    ld.rtol = 1e-9; ld.atol = 1e-11; ld.safety = 0.9; ld.ifactor = 10.0; ld.dfactor = 0.2; ld.order = 5; ld.init_order = 4;
    ld.max_num_steps = 1000000;
    mi_ode_linadj_handle lh = nullptr;
    MI(mi_ode_linadj_create(&ld, &lh));
    double *dW = nullptr, *db = nullptr, *dy = nullptr, *da = nullptr, *dat = nullptr, *dth = nullptr, *oa = nullptr, *oat = nullptr, *oth = nullptr;
    CK(hipMalloc((void**)&dW, W.size() * 8)); CK(hipMalloc((void**)&db, bb.size() * 8)); CK(hipMalloc((void**)&dy, y1.size() * 8));
    CK(hipMalloc((void**)&da, a1.size() * 8)); CK(hipMalloc((void**)&dat, 8)); CK(hipMalloc((void**)&dth, NP * 8));
    CK(hipMalloc((void**)&oa, a1.size() * 8)); CK(hipMalloc((void**)&oat, 8)); CK(hipMalloc((void**)&oth, NP * 8));
    CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bb.data(), bb.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, y1.data(), y1.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(da, a1.data(), a1.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dat, 0, 8)); CK(hipMemcpy(dth, th.data(), NP * 8, hipMemcpyHostToDevice));
    mi_ode_stats ls;
    double hs[2] = {0, 0};
    const int rcl = mi_ode_linadj_segment(lh, dW, db, dy, da, dat, dth, nullptr, 1.0, 0.0, oa, oat, oth, nullptr, hs, &ls, nullptr);
    if (rcl != 0) { printf("FAIL linadj segment status %d %s\n", rcl, mi_ode_last_error()); return 1; }
    std::vector<double> ga(a1.size()), gth(NP);
    double gat = 1.0;
    CK(hipMemcpy(ga.data(), oa, ga.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(gth.data(), oth, NP * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&gat, oat, 8, hipMemcpyDeviceToHost));
    double mda = 0.0, mdt = 0.0;
    for (size_t i = 0; i < ga.size(); ++i) mda = fmax(mda, fabs(ga[i] - ra[i]));
    for (int i = 0; i < NP; ++i) mdt = fmax(mdt, fabs(gth[i] - rth[i]));
    printf("linadj (70 x 8, fp64, t 1 -> 0): launches %d attempts %lld accepted %lld, max |adj_y - fine rk4| = %.3e, |adj_params - fine rk4| = %.3e, adj_t %.1e\n",
           (int)ls.n_launches, (long long)ls.n_attempts, (long long)ls.n_accepted, mda, mdt, gat);
    if (ls.n_launches != 1 || ls.n_accepted < 2 || !(mda < 1e-8) || !(mdt < 1e-8) || gat != 0.0) { printf("FAIL linadj\n"); return 1; }
    ld.dim = 200;                                                                  // outside the kernel's box: refused, not mis-run
    mi_ode_linadj_handle bad = nullptr;
    if (mi_ode_linadj_create(&ld, &bad) >= 0) { printf("FAIL linadj accepted dim 200\n"); return 1; }
    MI(mi_ode_linadj_destroy(lh));
    CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dy)); CK(hipFree(da)); CK(hipFree(dat)); CK(hipFree(dth)); CK(hipFree(oa)); CK(hipFree(oat)); CK(hipFree(oth));
  }
  // ---- 11. MI_ODE_RHS_MLP_TANH in float64, 6 -> 24 -> 24 -> 6, tanh, through the plain engine entry points: one launch per call at two
  //          batch sizes and on a fixed grid (rk4), against a fine RK4 solve of the same network written here.  (Round 5: the cooperative
  //          kernel, a thread per state element; round 6: the float64 MFMA tile kernels k_persist_mlp64 / k_fixed_mlp64 take this size.) ----
  {
    const int MD = 6, MH = 24;
    std::vector<double> W1((size_t)MD * MH), W2((size_t)MH * MH), W3((size_t)MH * MD), b1(MH), b2(MH), b3(MD);
    for (int i = 0; i < MD; ++i) for (int j = 0; j < MH; ++j) W1[(size_t)i * MH + j] = 0.6 * sin(1.3 * i + 0.7 * j) / sqrt((double)MD);
    for (int i = 0; i < MH; ++i) for (int j = 0; j < MH; ++j) W2[(size_t)i * MH + j] = 0.6 * cos(0.9 * i - 1.1 * j) / sqrt((double)MH);
    for (int i = 0; i < MH; ++i) for (int j = 0; j < MD; ++j) W3[(size_t)i * MD + j] = 0.6 * sin(0.5 * i + 1.9 * j + 0.3) / sqrt((double)MH);
    for (int j = 0; j < MH; ++j) { b1[j] = 0.1 * cos(j); b2[j] = 0.1 * sin(2.0 * j); }
    for (int j = 0; j < MD; ++j) b3[j] = 0.05 * cos(3.0 * j);
    auto net = [&](const double* y, double* f) {
      double h1[24], h2[24];
      for (int j = 0; j < MH; ++j) { double a = b1[j]; for (int k = 0; k < MD; ++k) a += y[k] * W1[(size_t)k * MH + j]; h1[j] = tanh(a); }
      for (int j = 0; j < MH; ++j) { double a = b2[j]; for (int k = 0; k < MH; ++k) a += h1[k] * W2[(size_t)k * MH + j]; h2[j] = tanh(a); }
      for (int j = 0; j < MD; ++j) { double a = b3[j]; for (int k = 0; k < MH; ++k) a += h2[k] * W3[(size_t)k * MD + j]; f[j] = a; }
    };
    double *dW1, *dW2, *dW3, *db1, *db2, *db3;
    CK(hipMalloc((void**)&dW1, W1.size() * 8)); CK(hipMalloc((void**)&dW2, W2.size() * 8)); CK(hipMalloc((void**)&dW3, W3.size() * 8));
    CK(hipMalloc((void**)&db1, b1.size() * 8)); CK(hipMalloc((void**)&db2, b2.size() * 8)); CK(hipMalloc((void**)&db3, b3.size() * 8));
    CK(hipMemcpy(dW1, W1.data(), W1.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW2, W2.data(), W2.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW3, W3.data(), W3.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, b1.data(), b1.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, b2.data(), b2.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db3, b3.data(), b3.size() * 8, hipMemcpyHostToDevice));
    const long long batches[2] = {50, 120000};                  // 42 trajectories per workgroup: 2 / 2858 workgroups (the latter not co-resident)
    for (int bi = 0; bi < 2; ++bi) {
      const long long MB = batches[bi];
      std::vector<double> my0((size_t)MB * MD), mout((size_t)3 * MB * MD);
      for (long long r = 0; r < MB; ++r) for (int c = 0; c < MD; ++c) my0[(size_t)r * MD + c] = sin(0.37 * (double)(r % 977) + 1.1 * c);
      double *dmy = nullptr, *dmo = nullptr;
      CK(hipMalloc((void**)&dmy, my0.size() * 8)); CK(hipMalloc((void**)&dmo, mout.size() * 8));
      CK(hipMemcpy(dmy, my0.data(), my0.size() * 8, hipMemcpyHostToDevice));
      for (int fixed = 0; fixed < 2; ++fixed) {
        mi_ode_desc md = d;                                      // dopri5, rtol 1e-9 / atol 1e-11 (section 2)
        md.batch = MB; md.dim = MD; md.dtype = MI_ODE_F64;
        memset(&md.rhs, 0, sizeof(md.rhs));
        md.rhs.kind = MI_ODE_RHS_MLP_TANH; md.rhs.sign = 1.0; md.rhs.hidden = MH;
        md.rhs.w[0] = dW1; md.rhs.w[1] = dW2; md.rhs.w[2] = dW3; md.rhs.b[0] = db1; md.rhs.b[1] = db2; md.rhs.b[2] = db3;
        if (fixed) { md.adaptive = 0; md.tableau.n_stages = 3; md.first_step = NAN; }
        mi_ode_handle mh = nullptr;
        MI(mi_ode_create(&md, &mh));
        mi_ode_stats ms;
        double tm[3] = {0.0, 0.5, 1.0};
        std::vector<double> tg(41);
        for (int i = 0; i < 41; ++i) tg[i] = 0.025 * i;
        int mb;
        if (fixed) mb = mi_ode_fixed_grid_integrate_on(mh, dmy, tg.data(), 41, tm, 3, 0.0, dmo, &ms, nullptr);
        else mb = mi_ode_integrate(mh, dmy, tm, 3, dmo, &ms, nullptr);
        if (mb != 0) { printf("FAIL float64 MLP (batch %lld, fixed %d): %d %s\n", MB, fixed, mb, mi_ode_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(mout.data(), dmo, mout.size() * 8, hipMemcpyDeviceToHost));
        double mdm = 0.0;
        for (long long r = 0; r < MB; r += (MB > 1000 ? 997 : 1)) {              // (a sample of the large batch)
          double y[6], k1[6], k2[6], k3[6], k4[6], ys[6];
          for (int c = 0; c < MD; ++c) y[c] = my0[(size_t)r * MD + c];
          const int NS = 400;
          for (int sidx = 0; sidx < NS; ++sidx) {
            const double dt = 1.0 / NS;
            net(y, k1);
            for (int c = 0; c < MD; ++c) ys[c] = y[c] + 0.5 * dt * k1[c];
            net(ys, k2);
            for (int c = 0; c < MD; ++c) ys[c] = y[c] + 0.5 * dt * k2[c];
            net(ys, k3);
            for (int c = 0; c < MD; ++c) ys[c] = y[c] + dt * k3[c];
            net(ys, k4);
            for (int c = 0; c < MD; ++c) y[c] += dt / 6 * (k1[c] + 2 * k2[c] + 2 * k3[c] + k4[c]);
            if (sidx + 1 == NS / 2) for (int c = 0; c < MD; ++c) mdm = fmax(mdm, fabs(mout[(size_t)MB * MD + (size_t)r * MD + c] - y[c]));
          }
          for (int c = 0; c < MD; ++c) mdm = fmax(mdm, fabs(mout[(size_t)2 * MB * MD + (size_t)r * MD + c] - y[c]));
        }
        printf("float64 MLP 6-24-24-6 (MFMA tile kernels), batch %lld, %s: launches %d attempts %lld, max |gpu - fine rk4| = %.3e\n", MB,
               fixed ? "rk4 on a 40-step grid" : "dopri5", (int)ms.n_launches, (long long)ms.n_attempts, mdm);
        if (ms.n_launches != 1 || !(mdm < (fixed ? 1e-6 : 1e-8))) { printf("FAIL float64 MLP\n"); return 1; }
        MI(mi_ode_destroy(mh));
      }
      CK(hipFree(dmy)); CK(hipFree(dmo));
    }
    CK(hipFree(dW1)); CK(hipFree(dW2)); CK(hipFree(dW3)); CK(hipFree(db1)); CK(hipFree(db2)); CK(hipFree(db3));
  }
  // ---- 12. MI_ODE_RHS_LINEAR at dim 200 (float64, bias): outside the resident-W tile kernels' 128 columns - round 6: the 256-wide tile
  //          kernels with W streamed (k_persist_linear_mfma<double, 256, 6>, k_fixed_linear_mfma<double, 256>); one launch per call,
  //          dopri5 and rk4 on a fixed grid against a fine RK4 solve written here; W updated in place between two calls. ----
  {
    const int WD = 200;
    const long long WB = 300;
    std::vector<double> Wm((size_t)WD * WD), wb(WD), wy0((size_t)WB * WD), wout((size_t)3 * WB * WD);
    for (int i = 0; i < WD; ++i) for (int j = 0; j < WD; ++j) Wm[(size_t)i * WD + j] = (i == j ? -0.5 : 0.0) + 0.5 * sin(0.7 * i - 1.3 * j) / sqrt((double)WD);
    for (int j = 0; j < WD; ++j) wb[j] = 0.1 * cos(0.3 * j);
    for (long long r = 0; r < WB; ++r) for (int c = 0; c < WD; ++c) wy0[(size_t)r * WD + c] = sin(0.37 * (double)r + 1.1 * c);
    double *dWm, *dwb, *dwy, *dwo;
    CK(hipMalloc((void**)&dWm, Wm.size() * 8)); CK(hipMalloc((void**)&dwb, wb.size() * 8));
    CK(hipMalloc((void**)&dwy, wy0.size() * 8)); CK(hipMalloc((void**)&dwo, wout.size() * 8));
    CK(hipMemcpy(dwb, wb.data(), wb.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dwy, wy0.data(), wy0.size() * 8, hipMemcpyHostToDevice));
    auto lin = [&](const double* y, double* f) {
      for (int j = 0; j < WD; ++j) { double a = wb[j]; for (int k = 0; k < WD; ++k) a += y[k] * Wm[(size_t)k * WD + j]; f[j] = a; }
    };
    mi_ode_handle wh[2] = {nullptr, nullptr};
    for (int pass = 0; pass < 2; ++pass) {                     // pass 1: the same handles after W was scaled in place (the kernels read a refreshed copy)
      if (pass == 1) for (auto& v : Wm) v *= 0.5;
      CK(hipMemcpy(dWm, Wm.data(), Wm.size() * 8, hipMemcpyHostToDevice));
      for (int fixed = 0; fixed < 2; ++fixed) {
        if (wh[fixed] == nullptr) {
          mi_ode_desc wd = d;                                    // dopri5, rtol 1e-9 / atol 1e-11 (section 2)
          wd.batch = WB; wd.dim = WD; wd.dtype = MI_ODE_F64;
          memset(&wd.rhs, 0, sizeof(wd.rhs));
          wd.rhs.kind = MI_ODE_RHS_LINEAR; wd.rhs.sign = 1.0; wd.rhs.w[0] = dWm; wd.rhs.b[0] = dwb;
          if (fixed) { wd.adaptive = 0; wd.tableau.n_stages = 3; wd.first_step = NAN; }
          MI(mi_ode_create(&wd, &wh[fixed]));
        }
        mi_ode_stats ws;
        double tm[3] = {0.0, 0.5, 1.0};
        std::vector<double> tg(41);
        for (int i = 0; i < 41; ++i) tg[i] = 0.025 * i;
        int wbits;
        if (fixed) wbits = mi_ode_fixed_grid_integrate_on(wh[fixed], dwy, tg.data(), 41, tm, 3, 0.0, dwo, &ws, nullptr);
        else wbits = mi_ode_integrate(wh[fixed], dwy, tm, 3, dwo, &ws, nullptr);
        if (wbits != 0) { printf("FAIL linear dim 200 (fixed %d): %d %s\n", fixed, wbits, mi_ode_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(wout.data(), dwo, wout.size() * 8, hipMemcpyDeviceToHost));
        double wdm = 0.0;
        for (long long r = 0; r < WB; r += 37) {
          std::vector<double> y(WD), k1(WD), k2(WD), k3(WD), k4(WD), ys(WD);
          for (int c = 0; c < WD; ++c) y[c] = wy0[(size_t)r * WD + c];
          const int NS = 400;
          for (int sidx = 0; sidx < NS; ++sidx) {
            const double dt = 1.0 / NS;
            lin(y.data(), k1.data());
            for (int c = 0; c < WD; ++c) ys[c] = y[c] + 0.5 * dt * k1[c];
            lin(ys.data(), k2.data());
            for (int c = 0; c < WD; ++c) ys[c] = y[c] + 0.5 * dt * k2[c];
            lin(ys.data(), k3.data());
            for (int c = 0; c < WD; ++c) ys[c] = y[c] + dt * k3[c];
            lin(ys.data(), k4.data());
            for (int c = 0; c < WD; ++c) y[c] += dt / 6 * (k1[c] + 2 * k2[c] + 2 * k3[c] + k4[c]);
            if (sidx + 1 == NS / 2) for (int c = 0; c < WD; ++c) wdm = fmax(wdm, fabs(wout[(size_t)WB * WD + (size_t)r * WD + c] - y[c]));
          }
          for (int c = 0; c < WD; ++c) wdm = fmax(wdm, fabs(wout[(size_t)2 * WB * WD + (size_t)r * WD + c] - y[c]));
        }
        printf("linear dim 200 float64 (256-wide tile kernels, W streamed)%s, %s: launches %d attempts %lld, max |gpu - fine rk4| = %.3e\n",
               pass ? " after W *= 0.5 in place" : "", fixed ? "rk4 on a 40-step grid" : "dopri5", (int)ws.n_launches, (long long)ws.n_attempts, wdm);
        if (ws.n_launches != 1 || !(wdm < (fixed ? 1e-6 : 1e-8))) { printf("FAIL linear dim 200\n"); return 1; }
      }
    }
    MI(mi_ode_destroy(wh[0])); MI(mi_ode_destroy(wh[1]));
    CK(hipFree(dWm)); CK(hipFree(dwb)); CK(hipFree(dwy)); CK(hipFree(dwo));
  }
  printf("C-ABI OK (abi %d)\n", mi_ode_abi_version());
  return 0;
}
