// Host side of the one-launch backward segment of odeint_adjoint for the linear right-hand side (include/mi_ode.h section A''',
// csrc/mi_ode_linadj.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi_ode_host.h"
#include "mi_ode_linadj.h"

using namespace mi;

struct mi_ode_linadj {
  mi_ode_linadj_desc d;
  int dp;                      // tile width of the kernel instantiation (16 / 32 / 64 / 128 >= dim)
  int is_f32;
  size_t elt;
  int grid, block;
  size_t lds;
  const void* fn;
  long long ntiles;
  char* planes;                // 8 state planes
  long long stride;
  double *pw, *cvec, *g0, *lmat, *mmat, *theta, *ktab;
  void *gpart, *wpad;
  long long* skew;
  double* partials;            // hand-off records (2 parities)
  Ctl* ctl_dev;
  LinAdjResult* res;           // pinned host
  unsigned seq;
  int spin_limit, spin_first, sleep_first, sleep_poll;
  long long n_launches;
};

namespace {

template <typename T>
const void* la_fn(int dp, size_t* lds, int* block) {
  *block = dp * 4;
  switch (dp) {
    case 16: *lds = linadj_lds_bytes<T, 16>(); return (const void*)k_linadj<T, 16>;
    case 32: *lds = linadj_lds_bytes<T, 32>(); return (const void*)k_linadj<T, 32>;
    case 64: *lds = linadj_lds_bytes<T, 64>(); return (const void*)k_linadj<T, 64>;
    default: *lds = linadj_lds_bytes<T, 128>(); return (const void*)k_linadj<T, 128>;
  }
}

int la_pad_dim(int dim) { return dim <= 16 ? 16 : (dim <= 32 ? 32 : (dim <= 64 ? 64 : 128)); }

// pi[sigma][p]: the coefficient of z^p in the stage-input polynomial R_sigma(z) of y' = y z (rk_common.py:44-52):
// R_0 = 1, R_sigma = 1 + z sum_j beta[sigma-1][j] R_j (oracle/linear_adjoint_numpy.py: stage_polynomials)
void stage_polynomials(const mi_ode_tableau& tb, double (*pi)[kLaP]) {
  for (int s = 0; s < kLaP; ++s)
    for (int p = 0; p < kLaP; ++p) pi[s][p] = 0.0;
  pi[0][0] = 1.0;
  for (int s = 1; s <= kLaS; ++s) {
    pi[s][0] = 1.0;
    for (int j = 0; j < s; ++j)
      for (int p = 1; p < kLaP; ++p) pi[s][p] += tb.beta[s - 1][j] * pi[j][p - 1];
  }
}

}  // namespace

extern "C" int mi_ode_linadj_destroy(mi_ode_linadj_handle h) {
  if (h == nullptr) return 0;
  if (h->planes) (void)hipFree(h->planes);
  if (h->pw) (void)hipFree(h->pw);
  if (h->cvec) (void)hipFree(h->cvec);
  if (h->g0) (void)hipFree(h->g0);
  if (h->lmat) (void)hipFree(h->lmat);
  if (h->mmat) (void)hipFree(h->mmat);
  if (h->theta) (void)hipFree(h->theta);
  if (h->ktab) (void)hipFree(h->ktab);
  if (h->gpart) (void)hipFree(h->gpart);
  if (h->wpad) (void)hipFree(h->wpad);
  if (h->skew) (void)hipFree(h->skew);
  if (h->partials) (void)hipFree(h->partials);
  if (h->ctl_dev) (void)hipFree(h->ctl_dev);
  if (h->res) (void)hipHostFree(h->res);
  delete h;
  return 0;
}

extern "C" int mi_ode_linadj_create(const mi_ode_linadj_desc* desc, mi_ode_linadj_handle* out) {
  if (desc == nullptr || out == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  *out = nullptr;
  const mi_ode_tableau& tb = desc->tableau;
  if (desc->batch < 1 || desc->dim < 1 || desc->dim > 128 || (desc->dtype != MI_ODE_F32 && desc->dtype != MI_ODE_F64)) {
    mi_set_error("linear adjoint: batch >= 1, 1 <= dim <= 128, dtype float32 or float64"); return MI_ODE_E_INVALID;
  }
  if (tb.n_stages != kLaS || !tb.fsal) {
    mi_set_error("linear adjoint: a 6-row FSAL-shaped tableau (dopri5) only - other methods take the generic path"); return MI_ODE_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); mi_set_error("no HIP device"); return MI_ODE_E_NODEVICE; }
  mi_ode_linadj* h = new mi_ode_linadj();
  memset(h, 0, sizeof(*h));
  h->d = *desc;
  h->is_f32 = desc->dtype == MI_ODE_F32;
  h->elt = h->is_f32 ? 4 : 8;
  h->dp = la_pad_dim(desc->dim);
  h->fn = h->is_f32 ? la_fn<float>(h->dp, &h->lds, &h->block) : la_fn<double>(h->dp, &h->lds, &h->block);
  h->ntiles = (desc->batch + 15) / 16;
  int dev = 0, cus = 0, per_cu = 0;
  {                                              // (h is released on these early returns too)
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 == hipSuccess) e0 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e0 != hipSuccess) {
      mi_set_error("mi_ode_linadj_create: device query failed: %s", hipGetErrorString(e0));
      (void)hipGetLastError();
      delete h; return MI_ODE_E_HIP;
    }
  }
  if (hipFuncSetAttribute(h->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds) != hipSuccess) (void)hipGetLastError();
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, h->fn, h->block, h->lds) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    mi_set_error("linear adjoint kernel does not fit a compute unit (LDS %zu bytes, %d threads)", h->lds, h->block);
    delete h; return MI_ODE_E_HIP;
  }
  long long g = h->ntiles;                       // every workgroup co-resident (the hand-offs spin): at most one per CU
  if (g > cus) g = cus;
  if (g > kLaMaxG) g = kLaMaxG;
  if (g < 1) g = 1;
  if (const char* eg = getenv("MI_ODE_LINADJ_GRID")) { const int v = atoi(eg); if (v >= 1 && v <= g) g = v; }   // tests: other grids
  h->grid = (int)g;
  const size_t D = (size_t)h->dp, E = D * D + D;
  const size_t n = (size_t)desc->batch * (size_t)desc->dim;
  h->stride = (long long)((n * h->elt + 255) / 256 * 256);
  hipError_t e = hipMalloc((void**)&h->planes, 8 * (size_t)h->stride);
  if (e == hipSuccess) e = hipMalloc((void**)&h->pw, kLaP * D * D * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->cvec, kLaP * D * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->g0, 2 * E * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->lmat, kLaP * D * D * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->mmat, kLaPP * E * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->theta, 2 * E * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->ktab, 4 * kLaPP * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->gpart, (size_t)h->grid * E * h->elt);
  if (e == hipSuccess) e = hipMalloc((void**)&h->wpad, 2 * D * D * h->elt);
  if (e == hipSuccess) e = hipMalloc((void**)&h->partials, (size_t)kMaxBlocks * kRec * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->ctl_dev, sizeof(Ctl));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->res, sizeof(LinAdjResult), hipHostMallocDefault);
  if (e == hipSuccess && getenv("MI_ODE_LINADJ_PROF") != nullptr) e = hipMalloc((void**)&h->skew, 32 * (size_t)kLaMaxG * 2 * sizeof(long long));
  if (e == hipSuccess) e = hipMemset(h->partials, 0, (size_t)kMaxBlocks * kRec * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->mmat, 0, kLaPP * E * sizeof(double));      // the bias rows of M_pq, p > 0, stay zero
  if (e == hipSuccess) e = hipMemset(h->cvec, 0, kLaP * D * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->lmat, 0, kLaP * D * D * sizeof(double));
  // the tableau's combination tables K^c_pq = sum_sigma c_sigma pi[sigma][p] pi[sigma][q]: solution, error estimate, y_mid, last stage
  double pi[kLaP][kLaP], ktab[4][kLaPP];
  stage_polynomials(tb, pi);
  for (int p = 0; p < kLaP; ++p)
    for (int q = 0; q < kLaP; ++q) {
      double ks = 0.0, ke = 0.0, km = 0.0;
      for (int s = 0; s < kLaP; ++s) {
        const double pp = pi[s][p] * pi[s][q];
        ks += tb.c_sol[s] * pp; ke += tb.c_error[s] * pp; km += tb.c_mid[s] * pp;
      }
      ktab[0][p * kLaP + q] = ks; ktab[1][p * kLaP + q] = ke; ktab[2][p * kLaP + q] = km;
      ktab[3][p * kLaP + q] = pi[kLaS][p] * pi[kLaS][q];
    }
  if (e == hipSuccess) e = hipMemcpy(h->ktab, ktab, sizeof(ktab), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    mi_set_error("linear adjoint workspace: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    mi_ode_linadj_destroy(h);
    return MI_ODE_E_HIP;
  }
  memset(h->res, 0, sizeof(LinAdjResult));
  h->seq = 0;
  h->spin_limit = 1 << 20;                       // an attempt of the two systems lasts a fraction of a millisecond: skew bound, not a time-out to hit
  h->spin_first = 1 << 14;                       // residency check (the first hand-off comes after a tile pass and a slab pass)
  if (const char* e3 = getenv("MI_ODE_PERSIST_SPIN_FIRST")) h->spin_first = atoi(e3);
  if (const char* e2 = getenv("MI_ODE_PERSIST_SPIN_LIMIT")) h->spin_limit = atoi(e2);
  h->sleep_first = h->grid <= 32 ? 16 : 32;
  h->sleep_poll = 2;
  *out = h;
  return 0;
}

extern "C" int mi_ode_linadj_segment(mi_ode_linadj_handle h, const void* w_dev, const void* b_dev, const void* y_dev, const void* adj_y_dev,
                                     const void* adj_t_dev, const void* adj_params_dev, const void* grad_out_dev, double t_start, double t_end,
                                     void* adj_y_out_dev, void* adj_t_out_dev, void* adj_params_out_dev, void* dldt_out_dev, double* host_scalars,
                                     mi_ode_stats* stats, void* stream) {
  if (h == nullptr || w_dev == nullptr || y_dev == nullptr || adj_y_dev == nullptr || adj_t_dev == nullptr || adj_params_dev == nullptr ||
      adj_y_out_dev == nullptr || adj_t_out_dev == nullptr || adj_params_out_dev == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (!(t_end != t_start)) {                      // _assert_increasing on the (possibly negated) pair (misc.py:158-159)
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->status = MI_ODE_ST_BAD_T; }
    return MI_ODE_ST_BAD_T;
  }
  hipStream_t st = (hipStream_t)stream;
  LinAdjArgs A;
  memset(&A, 0, sizeof(A));
  StepArgs& S = A.p.s;
  const mi_ode_tableau& tb = h->d.tableau;
  S.ctl = h->ctl_dev; S.batch = h->d.batch; S.dim = h->d.dim; S.n_plane = h->d.batch * (long long)h->d.dim;
  S.interp = MI_ODE_INTERP_QUARTIC_MID;
  for (int i = 0; i < kLaS; ++i) {
    S.alpha[i] = tb.alpha[i];
    for (int j = 0; j <= i; ++j) S.beta[i][j] = tb.beta[i][j];
  }
  for (int j = 0; j <= kLaS; ++j) { S.e[j] = tb.c_error[j]; S.cmid[j] = tb.c_mid[j]; S.csol[j] = tb.c_sol[j]; }
  S.partials = h->partials;
  S.out = adj_y_out_dev;
  const bool reversed = t_end < t_start;          // misc.py:311-321: t <- -t, f <- -f(-t, y)
  S.rhs.w[0] = w_dev; S.rhs.b[0] = b_dev;
  S.rhs.sign = reversed ? -1.0 : 1.0;
  CtrlParams& cp = S.cp;
  cp.rtol = h->d.rtol; cp.atol = h->d.atol; cp.safety = h->d.safety; cp.ifactor = h->d.ifactor; cp.dfactor = h->d.dfactor;
  cp.inv_ifactor = 1.0 / h->d.ifactor; cp.inv_dfactor = 1.0 / h->d.dfactor;
  cp.max_num_steps = h->d.max_num_steps > 0 ? h->d.max_num_steps : 2147483647LL;
  cp.n_local = S.n_plane;
  cp.order = h->d.order; cp.init_order = h->d.init_order;
  cp.controller = MI_ODE_CTRL_MISC; cp.is_f32 = h->is_f32; cp.n_stages = kLaS; cp.auto_first_step = 1;
  A.p.t0 = reversed ? -t_start : t_start;
  A.t_end = reversed ? -t_end : t_end;
  A.p.t_small[0] = A.t_end;
  A.p.n_out = 1;
  A.p.world = 1;
  A.p.nseg = 1;
  A.p.seq_base = h->seq;
  A.p.spin_limit = h->spin_limit;
  A.p.xspin_limit = h->spin_limit;
  A.p.spin_first = h->spin_first < h->spin_limit ? h->spin_first : h->spin_limit;
  A.p.sleep_first = h->sleep_first; A.p.sleep_poll = h->sleep_poll;
  A.y_in = y_dev; A.a_in = adj_y_dev; A.th_in = adj_params_dev; A.adjt_in = adj_t_dev;
  A.th_out = adj_params_out_dev; A.adjt_out = adj_t_out_dev;
  A.grad_in = grad_out_dev; A.dldt_out = dldt_out_dev;
  A.planes = h->planes; A.stride = h->stride;
  A.pw = h->pw; A.cvec = h->cvec; A.gpart = h->gpart; A.g0 = h->g0; A.lmat = h->lmat; A.mmat = h->mmat; A.theta = h->theta; A.ktab = h->ktab; A.wpad = h->wpad;
  A.res = h->res;
  A.has_bias = b_dev != nullptr ? 1 : 0;
  A.skew = h->skew;
  if (h->skew) (void)hipMemsetAsync(h->skew, 0, 32 * (size_t)kLaMaxG * 2 * sizeof(long long), st);
  if (const char* ed = getenv("MI_ODE_LINADJ_DBG")) A.dbg = atoi(ed);
  memset(h->res, 0, sizeof(LinAdjResult));
  h->res->status = MI_ODE_ST_SYNC_TIMEOUT;       // (overwritten by the kernel's result record)
  void* args[] = {(void*)&A};
  hipError_t e = hipLaunchKernel(h->fn, dim3((unsigned)h->grid), dim3((unsigned)h->block), args, h->lds, st);
  if (e != hipSuccess) { mi_set_error("linear adjoint kernel launch failed: %s", hipGetErrorString(e)); (void)hipGetLastError(); return MI_ODE_E_HIP; }
  h->n_launches += 1;
  {
    const hipError_t es = hipStreamSynchronize(st);          // the kernel's last act was the zero-copy store of its result record
    if (es != hipSuccess) {
      // the aborted launch may have written records and flags under this call's stamps: the next segment must not reuse them
      h->seq += 4096u;
      if (h->seq >= 0xE0000000u) h->seq = 0;
      mi_set_error("hipStreamSynchronize after the linear adjoint kernel failed: %s", hipGetErrorString(es));
      (void)hipGetLastError();
      return MI_ODE_E_HIP;
    }
  }
  const LinAdjResult r = *h->res;
  h->seq += (unsigned)(r.handoffs > 0 ? r.handoffs : 64) + 16u;
  if (h->seq >= 0xE0000000u) h->seq = 0;
  if (getenv("MI_ODE_LINADJ_PROF") != nullptr)
    fprintf(stderr, "[linadj prof] attempts %lld (accepted %lld) hand-offs %d  us: tile passes %.1f (matrix loads %.1f, before %.1f, block reduce %.1f)  theta combinations %.1f  "
            "attempt hand-offs %.1f  small products: flag waits %.1f L %.1f M %.1f | prologue: init + first step size + first L_p %.1f f0 passes %.1f power levels %.1f slab %.1f first hand-off + barrier %.1f fold %.1f rest %.1f | epilogue %.1f\n",
            r.n_attempt, r.n_accept, r.handoffs, 0.01 * r.prof[0], 0.01 * r.prof[8], 0.01 * r.prof[7], 0.01 * r.prof[9], 0.01 * r.prof[1], 0.01 * r.prof[2], 0.01 * r.prof[10],
            0.01 * r.prof[12], 0.01 * r.prof[13], 0.01 * r.prof[5], 0.01 * r.prof[14], 0.01 * r.prof[15], 0.01 * r.prof[3], 0.01 * r.prof[4], 0.01 * r.prof[11], 0.0, 0.01 * r.prof[6]);
  if (h->skew != nullptr && getenv("MI_ODE_LINADJ_PROF") != nullptr) {
    static long long stamps[32 * kLaMaxG * 2];
    const int G = h->grid;
    if (hipMemcpy(stamps, h->skew, sizeof(long long) * 32 * G * 2, hipMemcpyDeviceToHost) == hipSuccess) {
      for (int g = 0; g < 32; ++g) {
        long long a_min = (1LL << 62), a_max = 0, l_min = (1LL << 62), l_max = 0; double a_sum = 0; int n = 0, last = -1;
        for (int b = 0; b < G; ++b) {
          const long long a = stamps[(g * G + b) * 2], l = stamps[(g * G + b) * 2 + 1];
          if (a == 0) continue;
          ++n;
          if (a < a_min) a_min = a;
          if (a > a_max) { a_max = a; last = b; }
          if (l < l_min) l_min = l;
          if (l > l_max) l_max = l;
          a_sum += (double)a;
        }
        if (n > 0)
          fprintf(stderr, "[linadj skew] hand-off %d: arrivals spread %.2f us (mean arrives %.2f us before the last, workgroup %d), first leaves %.2f us / last leaves %.2f us after the last arrival\n",
                  g, 0.01 * (a_max - a_min), 0.01 * ((double)a_max - a_sum / n), last, 0.01 * (l_min - a_max), 0.01 * (l_max - a_max));
      }
    }
  }
  if (host_scalars != nullptr) { host_scalars[0] = r.dldt; host_scalars[1] = r.adjt_end; }
  if (stats != nullptr) {
    memset(stats, 0, sizeof(*stats));
    stats->n_attempts = r.n_attempt; stats->n_accepted = r.n_accept; stats->n_rejected = r.n_attempt - r.n_accept;
    stats->nfe = 2 + 6 * r.n_attempt;
    stats->t = reversed ? -r.t1 : r.t1; stats->dt = r.dt; stats->last_ratio = r.ratio; stats->status = r.status;
    stats->n_polls = 1; stats->n_launches = 1;
    stats->clock_mhz = r.clk_ticks > 0 ? 100.0 * (double)r.clk_cycles / (double)r.clk_ticks : 0.0;
  }
  return (int)r.status;
}

extern "C" int mi_ode_linadj_profile(mi_ode_linadj_handle h, double* out8) {
  if (h == nullptr || out8 == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  const LinAdjResult& r = *h->res;
  out8[0] = 0.01 * (double)(r.prof[0] + r.prof[7] + r.prof[8] + r.prof[9]);
  out8[1] = 0.01 * (double)r.prof[1]; out8[2] = 0.01 * (double)r.prof[2]; out8[3] = 0.01 * (double)r.prof[3];
  out8[4] = 0.01 * (double)(r.prof[10] + r.prof[11] + r.prof[12] + r.prof[13]);
  out8[5] = 0.01 * (double)r.prof[5]; out8[6] = 0.01 * (double)r.prof[6];
  out8[7] = (double)r.handoffs;
  return 0;
}
