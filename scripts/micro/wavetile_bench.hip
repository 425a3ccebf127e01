// Wave-tile attempt kernel (csrc/mi_ode_wavetile.h) against the workgroup-tile kernel it replaces (k_step_linear_mfma):
// same inputs, bit-for-bit comparison of y1 / f1 / the reduction records, then timing of both at config-4 size.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I tfdiffeq_amd/csrc -I scripts/micro -o scripts/micro/wavetile_bench scripts/micro/wavetile_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "mi_ode_wavetile.h"   // (next to this file: an experiment, not part of libmi_ode)

using namespace mi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static const double A_[6][6] = {{1 / 5.}, {3 / 40., 9 / 40.}, {44 / 45., -56 / 15., 32 / 9.},
                                {19372 / 6561., -25360 / 2187., 64448 / 6561., -212 / 729.},
                                {9017 / 3168., -355 / 33., 46732 / 5247., 49 / 176., -5103 / 18656.},
                                {35 / 384., 0., 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84.}};
static const double AL_[6] = {1 / 5., 3 / 10., 4 / 5., 8 / 9., 1., 1.};
static const double E_[7] = {35 / 384. - 1951 / 21600., 0., 500 / 1113. - 22642 / 50085., 125 / 192. - 451 / 720.,
                             -2187 / 6784. - -12231 / 42400., 11 / 84. - 649 / 6300., -1. / 60.};
static const double MID_[7] = {6025192743 / 30085553152. / 2, 0, 51252292925 / 65400821598. / 2, -2691868925 / 45128329728. / 2,
                               187940372067 / 1594534317056. / 2, -1776094331 / 19743644256. / 2, 11237099 / 235043384. / 2};

template <typename T>
int run(long long batch, int iters, bool with_out) {
  constexpr int D = 128, S = 6;
  const long long n = batch * D;
  const size_t pb = (size_t)n * sizeof(T);
  std::vector<T> hW((size_t)D * D), hy(n), hf(n);
  srand(1);
  for (auto& v : hW) v = (T)((rand() / (double)RAND_MAX - 0.5) * 0.2);
  for (auto& v : hy) v = (T)(rand() / (double)RAND_MAX - 0.5);
  for (auto& v : hf) v = (T)(rand() / (double)RAND_MAX - 0.5);
  T *W, *planes, *out;
  const int nplanes = 2 + S + 1;
  CK(hipMalloc(&W, hW.size() * sizeof(T)));
  CK(hipMalloc(&planes, pb * nplanes));
  CK(hipMalloc(&out, pb * 2));
  CK(hipMemcpy(W, hW.data(), hW.size() * sizeof(T), hipMemcpyHostToDevice));
  CK(hipMemcpy(planes, hy.data(), pb, hipMemcpyHostToDevice));                       // plane 0 = y0
  CK(hipMemcpy((char*)planes + 2 * pb, hf.data(), pb, hipMemcpyHostToDevice));       // plane 2 = f0
  Ctl hc;
  memset(&hc, 0, sizeof(hc));
  hc.t1 = 0.0; hc.dt = 0.05; hc.idx_y0 = 0; hc.idx_y1 = 1;
  for (int j = 0; j < kMaxK; ++j) hc.idx_k[j] = 2 + j;
  hc.next_out = 0; hc.n_out = with_out ? 1 : 0;
  Ctl* ctl;
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMemcpy(ctl, &hc, sizeof(Ctl), hipMemcpyHostToDevice));
  double *partials, *tout;
  CK(hipMalloc(&partials, kMaxBlocks * kRec * sizeof(double)));
  CK(hipMalloc(&tout, 8 * sizeof(double)));
  const double t_out_h[1] = {0.03};
  CK(hipMemcpy(tout, t_out_h, sizeof(double), hipMemcpyHostToDevice));
  StepArgs A;
  memset(&A, 0, sizeof(A));
  A.ctl = ctl; A.planes = (char*)planes; A.stride = (long long)pb; A.batch = batch; A.dim = D; A.interp = 0;
  A.out = out; A.t_out = with_out ? tout : nullptr; A.n_plane = n;
  for (int i = 0; i < S; ++i) { A.alpha[i] = AL_[i]; for (int j = 0; j <= i; ++j) A.beta[i][j] = A_[i][j]; }
  for (int j = 0; j <= S; ++j) { A.e[j] = E_[j]; A.cmid[j] = MID_[j]; A.csol[j] = j < S ? A_[S - 1][j] : 0.0; }
  A.partials = partials; A.rhs.w[0] = W; A.rhs.sign = 1.0; A.ticket = nullptr;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  // --- old kernel ---
  auto k_old = k_step_linear_mfma<T, D, S, false>;
  const size_t lds_old = step_linear_lds_bytes<T, D>();
  const long long nt = (batch + 15) / 16;
  int g_old = (int)std::min<long long>(cus, nt);
  std::vector<T> y_old(n), f_old(n), o_old(n), y_new(n), f_new(n), o_new(n);
  std::vector<double> p_old((size_t)g_old * kRec), p_new;
  hipLaunchKernelGGL(k_old, dim3(g_old), dim3(D * 4), lds_old, 0, A);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y_old.data(), (char*)planes + pb, pb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(f_old.data(), (char*)planes + (2 + S) * pb, pb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(p_old.data(), partials, p_old.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (with_out) CK(hipMemcpy(o_old.data(), out, pb, hipMemcpyDeviceToHost));
  CK(hipMemset((char*)planes + pb, 0, pb));
  CK(hipMemset((char*)planes + (2 + S) * pb, 0, pb));
  CK(hipMemset(out, 0, pb));
  // --- new kernel ---
  auto k_new = k_step_linear_wt<T, D, S, false>;
  const size_t lds_new = Wt<T, D>::kLdsBytes;
  CK(hipFuncSetAttribute((const void*)k_new, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
  int g_new = (int)std::min<long long>(cus, (nt + 3) / 4);
  p_new.resize((size_t)g_new * kRec);
  hipLaunchKernelGGL(k_new, dim3(g_new), dim3(256), lds_new, 0, A);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y_new.data(), (char*)planes + pb, pb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(f_new.data(), (char*)planes + (2 + S) * pb, pb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(p_new.data(), partials, p_new.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (with_out) CK(hipMemcpy(o_new.data(), out, pb, hipMemcpyDeviceToHost));
  long long bad_y = 0, bad_f = 0, bad_o = 0;
  double worst = 0;
  for (long long i = 0; i < n; ++i) {
    if (memcmp(&y_old[i], &y_new[i], sizeof(T)) != 0) { ++bad_y; worst = std::max(worst, (double)std::fabs(y_old[i] - y_new[i])); }
    if (memcmp(&f_old[i], &f_new[i], sizeof(T)) != 0) { ++bad_f; worst = std::max(worst, (double)std::fabs(f_old[i] - f_new[i])); }
    if (with_out && memcmp(&o_old[i], &o_new[i], sizeof(T)) != 0) { ++bad_o; worst = std::max(worst, (double)std::fabs(o_old[i] - o_new[i])); }
  }
  double ro[3] = {0, 0, 0}, rn[3] = {0, 0, 0};
  for (int b = 0; b < g_old; ++b) { ro[0] = std::max(ro[0], p_old[b * kRec + R_MAXA]); ro[1] = std::max(ro[1], p_old[b * kRec + R_MAXB]); ro[2] += p_old[b * kRec + R_SUMA]; }
  for (int b = 0; b < g_new; ++b) { rn[0] = std::max(rn[0], p_new[b * kRec + R_MAXA]); rn[1] = std::max(rn[1], p_new[b * kRec + R_MAXB]); rn[2] += p_new[b * kRec + R_SUMA]; }
  printf("[%s batch %lld out %d] mismatching elements: y1 %lld  f1 %lld  out %lld  (worst abs diff %.3e)\n", sizeof(T) == 8 ? "f64" : "f32", batch,
         (int)with_out, bad_y, bad_f, bad_o, worst);
  printf("  records old {%.17g %.17g %.17g}\n          new {%.17g %.17g %.17g}  rel diff of sum %.2e\n", ro[0], ro[1], ro[2], rn[0], rn[1], rn[2],
         std::fabs(ro[2] - rn[2]) / std::fabs(ro[2]));
  const int fail = (bad_y || bad_f || bad_o || ro[0] != rn[0] || ro[1] != rn[1] || std::fabs(ro[2] - rn[2]) > 1e-12 * std::fabs(ro[2])) ? 1 : 0;
  if (iters > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 6.0 * 2 * D * (double)n;
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 20; ++i) {     // clock ramp
          if (which == 0) hipLaunchKernelGGL(k_old, dim3(g_old), dim3(D * 4), lds_old, 0, A);
          else hipLaunchKernelGGL(k_new, dim3(g_new), dim3(256), lds_new, 0, A);
        }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) {
          if (which == 0) hipLaunchKernelGGL(k_old, dim3(g_old), dim3(D * 4), lds_old, 0, A);
          else hipLaunchKernelGGL(k_new, dim3(g_new), dim3(256), lds_new, 0, A);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %s kernel: %.4f ms per attempt, %.2f TFLOP/s (launch gaps included)\n", which == 0 ? "workgroup-tile (old)" : "wave-tile (new)     ",
               ms / iters, flops / (ms / iters * 1e-3) / 1e12);
      }
    }
  }
  (void)hipFree(W); (void)hipFree(planes); (void)hipFree(out); (void)hipFree(ctl); (void)hipFree(partials); (void)hipFree(tout);
  return fail;
}

int main(int argc, char** argv) {
  int fail = 0;
  fail |= run<double>(16 * 7 + 5, 0, false);          // ragged last tile, fewer tiles than waves
  fail |= run<double>(16 * 1024 * 3 + 9, 0, true);    // several tiles per wave, dense output inside the attempt
  fail |= run<double>(65536, 50, false);              // config 4
  fail |= run<double>(65536, 20, true);
  printf(fail ? "FAILED\n" : "ALL BIT-IDENTICAL\n");
  return fail;
}
