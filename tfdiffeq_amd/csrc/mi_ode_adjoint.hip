// Host side of the fused backward segment of odeint_adjoint (include/mi_ode.h section A', csrc/mi_ode_adjoint.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi_ode_host.h"
#include "mi_ode_adjoint.h"

using namespace mi;

struct mi_ode_adjoint {
  mi_ode_adjoint_desc d;
  int dp, hp;                  // padded widths of the kernel instantiation
  int P, Ppad, SL;
  int grid, block;
  size_t lds;
  const void* fn[3];           // the kernel for each hidden activation (mi_ode_rhs.scalars[0])
  long long ntiles;
  float *planes, *theta, *act, *wpart;
  double* partials;            // hand-off records (2 parities)
  Ctl* ctl_dev;                // (persist_init_ctl target is LDS; this backs StepArgs.ctl for completeness)
  AdjResult* res;              // pinned host
  AdjArgs* args_host;          // pinned staging of the kernel's argument block ...
  AdjArgs* args_dev;           // ... and its device copy (the kernel and its non-inlined passes read it with scalar loads)
  unsigned seq;
  int spin_limit, spin_first, sleep_first, sleep_poll;
  long long n_launches;
};

namespace {
template <int DP, int HP>
const void* adj_fn(int act, size_t* lds, int* block) {
  *lds = AdjGeom<DP, HP>::lds_bytes();
  *block = 64 * AdjGeom<DP, HP>::NW;
  switch (act) {
    case MLP_ACT_TANH: return (const void*)k_adjoint_mlp<DP, HP, MLP_ACT_TANH, 6>;
    case MLP_ACT_RELU: return (const void*)k_adjoint_mlp<DP, HP, MLP_ACT_RELU, 6>;
    case MLP_ACT_SOFTPLUS: return (const void*)k_adjoint_mlp<DP, HP, MLP_ACT_SOFTPLUS, 6>;
    default: return nullptr;
  }
}
const void* adj_fn_dims(int dp, int hp, int act, size_t* lds, int* block) {
  if (dp == 16 && hp == 16) return adj_fn<16, 16>(act, lds, block);
  if (dp == 16 && hp == 128) return adj_fn<16, 128>(act, lds, block);
  if (dp == 64 && hp == 16) return adj_fn<64, 16>(act, lds, block);
  return adj_fn<64, 128>(act, lds, block);
}
int pad16(int v, int lo, int hi) { return v <= lo ? lo : hi; }
}  // namespace

extern "C" int mi_ode_adjoint_destroy(mi_ode_adjoint_handle h) {
  if (h == nullptr) return 0;
  if (h->planes) (void)hipFree(h->planes);
  if (h->theta) (void)hipFree(h->theta);
  if (h->act) (void)hipFree(h->act);
  if (h->wpart) (void)hipFree(h->wpart);
  if (h->partials) (void)hipFree(h->partials);
  if (h->ctl_dev) (void)hipFree(h->ctl_dev);
  if (h->res) (void)hipHostFree(h->res);
  if (h->args_host) (void)hipHostFree(h->args_host);
  if (h->args_dev) (void)hipFree(h->args_dev);
  delete h;
  return 0;
}

extern "C" int64_t mi_ode_adjoint_num_params(mi_ode_adjoint_handle h) { return h ? (int64_t)h->P : -1; }

extern "C" int mi_ode_adjoint_create(const mi_ode_adjoint_desc* desc, mi_ode_adjoint_handle* out) {
  if (desc == nullptr || out == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  *out = nullptr;
  const mi_ode_tableau& tb = desc->tableau;
  if (desc->batch < 1 || desc->dim < 1 || desc->dim > 64 || desc->hidden < 1 || desc->hidden > 128) {
    mi_set_error("fused adjoint: batch >= 1, 1 <= dim <= 64, 1 <= hidden <= 128"); return MI_ODE_E_INVALID;
  }
  if (tb.n_stages != 6 || !tb.fsal || tb.c_sol[1] != 0.0 || tb.c_error[1] != 0.0 || tb.c_mid[1] != 0.0) {
    mi_set_error("fused adjoint: the dopri5 tableau only (6 rows, FSAL shaped, no weight on stage 2 in c_sol / c_error / c_mid)");
    return MI_ODE_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); mi_set_error("no HIP device"); return MI_ODE_E_NODEVICE; }
  mi_ode_adjoint* h = new mi_ode_adjoint();
  memset(h, 0, sizeof(*h));
  h->d = *desc;
  h->dp = pad16(desc->dim, 16, 64);
  h->hp = pad16(desc->hidden, 16, 128);
  for (int act = 0; act < 3; ++act) h->fn[act] = adj_fn_dims(h->dp, h->hp, act, &h->lds, &h->block);
  const int d = desc->dim, hd = desc->hidden;
  h->P = (desc->time_dependent ? hd : 0) + d * hd + hd + hd * hd + hd + hd * d + d;
  h->Ppad = (h->P + 63) / 64 * 64;
  h->ntiles = (desc->batch + 31) / 32;
  int dev = 0, cus = 0, per_cu = 0;
  MI_HIP(hipGetDevice(&dev));
  MI_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  for (int act = 0; act < 3; ++act)
    if (hipFuncSetAttribute(h->fn[act], hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds) != hipSuccess) (void)hipGetLastError();
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, h->fn[0], h->block, h->lds) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    mi_set_error("fused adjoint kernel does not fit a compute unit (LDS %zu bytes, %d threads)", h->lds, h->block);
    delete h; return MI_ODE_E_HIP;
  }
  long long g = h->ntiles;                       // every workgroup co-resident (the hand-offs spin): at most one per CU
  if (g > cus) g = cus;
  if (g > kPersistMaxGrid) g = kPersistMaxGrid;
  h->grid = (int)g;
  h->SL = (h->P + h->grid - 1) / h->grid;
  const size_t n = (size_t)desc->batch * (size_t)d;
  size_t slot_floats = (size_t)32 * (2 * (size_t)h->dp + 4 * (size_t)h->hp);
  hipError_t e = hipMalloc((void**)&h->planes, 8 * n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&h->theta, 3 * (size_t)h->Ppad * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&h->act, (size_t)h->ntiles * 6 * slot_floats * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&h->wpart, (size_t)h->grid * 3 * (size_t)h->Ppad * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&h->partials, (size_t)kMaxBlocks * kRec * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->ctl_dev, sizeof(Ctl));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->res, sizeof(AdjResult), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->args_host, sizeof(AdjArgs), hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc((void**)&h->args_dev, sizeof(AdjArgs));
  if (e == hipSuccess) e = hipMemset(h->partials, 0, (size_t)kMaxBlocks * kRec * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->wpart, 0, (size_t)h->grid * 3 * (size_t)h->Ppad * sizeof(float));
  if (e == hipSuccess) e = hipMemset(h->theta, 0, 3 * (size_t)h->Ppad * sizeof(float));   // (mi_ode_adjoint_dynamics reads a dummy adj_t from it)
  if (e != hipSuccess) {
    mi_set_error("fused adjoint workspace: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    mi_ode_adjoint_destroy(h);
    return MI_ODE_E_HIP;
  }
  memset(h->res, 0, sizeof(AdjResult));
  h->seq = 0;
  h->spin_limit = 1 << 20;                       // an attempt of the augmented system lasts ~1 ms: skew bound, not a time-out to hit
  h->spin_first = 1 << 14;                       // residency check (first hand-off comes after a whole tile pass)
  if (const char* e3 = getenv("MI_ODE_PERSIST_SPIN_FIRST")) h->spin_first = atoi(e3);
  if (const char* e2 = getenv("MI_ODE_PERSIST_SPIN_LIMIT")) h->spin_limit = atoi(e2);
  h->sleep_first = h->grid <= 32 ? 16 : 32;
  h->sleep_poll = 2;
  *out = h;
  return 0;
}

static int adj_launch(mi_ode_adjoint* h, const mi_ode_rhs* rhs, int mode, const void* y, const void* a, const void* adjt,
                      const void* th, double t_start, double t_end, void* y_out, void* a_out, void* adjt_out, void* th_out,
                      mi_ode_stats* stats, hipStream_t st) {
  if (rhs == nullptr || rhs->kind != MI_ODE_RHS_MLP_TANH || rhs->hidden != h->d.hidden || rhs->w[0] == nullptr || rhs->w[1] == nullptr ||
      rhs->w[2] == nullptr) {
    mi_set_error("fused adjoint: rhs must be the MLP-tanh descriptor the handle was created for"); return MI_ODE_E_INVALID;
  }
  if ((rhs->scalars[1] != 0.0) != (h->d.time_dependent != 0)) {   // (mi_ode_rhs carries no dim: this is the one shape mismatch the descriptor can show)
    mi_set_error("fused adjoint: the handle was created with time_dependent = %d but rhs.scalars[1] says %s (w[0] is [dim + 1, hidden] for the "
                 "time-dependent network)", (int)h->d.time_dependent, rhs->scalars[1] != 0.0 ? "time-dependent" : "time-independent");
    return MI_ODE_E_INVALID;
  }
  MI_HIP(hipStreamSynchronize(st));              // the pinned argument block may still be in flight from a previous call
  AdjArgs& A = *h->args_host;
  memset(&A, 0, sizeof(A));
  StepArgs& S = A.p.s;
  const mi_ode_tableau& tb = h->d.tableau;
  S.ctl = h->ctl_dev; S.batch = h->d.batch; S.dim = h->d.dim; S.n_plane = h->d.batch * (long long)h->d.dim;
  S.interp = MI_ODE_INTERP_QUARTIC_MID;
  for (int i = 0; i < 6; ++i) {
    S.alpha[i] = tb.alpha[i];
    for (int j = 0; j <= i; ++j) S.beta[i][j] = tb.beta[i][j];
  }
  for (int j = 0; j <= 6; ++j) { S.e[j] = tb.c_error[j]; S.cmid[j] = tb.c_mid[j]; S.csol[j] = tb.c_sol[j]; }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) A.cb[i][j] = (float)tb.beta[i][j];
  for (int j = 0; j <= 6; ++j) { A.ce[j] = (float)tb.c_error[j]; A.cm[j] = (float)tb.c_mid[j]; }
  for (int j = 0; j < 6; ++j) A.ca[j] = (float)tb.alpha[j];
  A.td = h->d.time_dependent ? 1 : 0;
  S.partials = h->partials;
  const bool reversed = mode == 0 && t_end < t_start;       // misc.py:311-321: t <- -t, f <- -f(-t, y)
  for (int i = 0; i < 8; ++i) S.rhs.s[i] = rhs->scalars[i];
  for (int i = 0; i < 3; ++i) { S.rhs.w[i] = rhs->w[i]; S.rhs.b[i] = rhs->b[i]; }
  S.rhs.sign = reversed ? -1.0 : 1.0;
  S.rhs.hidden = rhs->hidden;
  CtrlParams& cp = S.cp;
  cp.rtol = h->d.rtol; cp.atol = h->d.atol; cp.safety = h->d.safety; cp.ifactor = h->d.ifactor; cp.dfactor = h->d.dfactor;
  cp.inv_ifactor = 1.0 / h->d.ifactor; cp.inv_dfactor = 1.0 / h->d.dfactor;
  cp.max_num_steps = h->d.max_num_steps > 0 ? h->d.max_num_steps : 2147483647LL;
  cp.n_local = S.n_plane;
  cp.order = h->d.order; cp.init_order = h->d.init_order;
  cp.controller = MI_ODE_CTRL_MISC; cp.is_f32 = 1; cp.n_stages = 6; cp.auto_first_step = 1;
  A.p.t0 = reversed ? -t_start : t_start;
  A.t_end = reversed ? -t_end : t_end;
  A.p.n_out = 1;
  A.p.world = 1;
  A.p.seq_base = h->seq;
  A.p.spin_limit = h->spin_limit;
  A.p.spin_first = h->spin_first < h->spin_limit ? h->spin_first : h->spin_limit;
  A.p.sleep_first = h->sleep_first; A.p.sleep_poll = h->sleep_poll;
  A.y_in = (const float*)y; A.a_in = (const float*)a; A.y_out = (float*)y_out; A.a_out = (float*)a_out;
  A.th_in = (const float*)th; A.th_out = (float*)th_out; A.adjt_in = (const float*)adjt; A.adjt_out = (float*)adjt_out;
  A.planes = h->planes; A.theta = h->theta; A.act = h->act; A.wpart = h->wpart; A.res = h->res;
  A.mode = mode; A.P = h->P; A.Ppad = h->Ppad; A.SL = h->SL;
  if (const char* be = getenv("MI_ODE_ADJOINT_BENCH")) {     // "mode,iterations": segment calls time one pass instead (tuning aid)
    int bm = 0, bi = 0, bf = 0;
    if (mode == 0 && sscanf(be, "%d,%d,%d", &bm, &bi, &bf) >= 2 && (bm == 2 || bm == 3) && bi > 0) { A.mode = bm; A.bench_iters = bi; A.bench_flags = bf; }
  }
  MI_HIP(hipMemcpyAsync(h->args_dev, h->args_host, sizeof(AdjArgs), hipMemcpyHostToDevice, st));
  const AdjArgs* dev_args = h->args_dev;
  void* args[] = {(void*)&dev_args};
  const int act = (int)rhs->scalars[0];
  if (act < 0 || act > 2) { mi_set_error("fused adjoint: unknown activation code %d", act); return MI_ODE_E_INVALID; }
  hipError_t e = hipLaunchKernel(h->fn[act], dim3((unsigned)h->grid), dim3((unsigned)h->block), args, h->lds, st);
  if (e != hipSuccess) { mi_set_error("fused adjoint kernel launch failed: %s", hipGetErrorString(e)); (void)hipGetLastError(); return MI_ODE_E_HIP; }
  h->n_launches += 1;
  MI_HIP(hipStreamSynchronize(st));              // the kernel's last act was the zero-copy store of its result record
  const AdjResult r = *h->res;
  h->seq += (unsigned)r.handoffs + 16u;
  if (h->seq >= 0xE0000000u) h->seq = 0;
  if (A.mode >= 2) {
    fprintf(stderr, "[adjoint bench] %s pass: %.1f us each (%d iterations, grid %d)\n", A.mode == 2 ? "tile (one attempt, 6 evaluations per tile)" : "weight-gradient (6 slots, 2 combinations)",
            0.01 * (double)(r.prof[0] + r.prof[1]) / A.bench_iters, A.bench_iters, h->grid);
    if (stats != nullptr) memset(stats, 0, sizeof(*stats));
    return 0;
  }
  if (getenv("MI_ODE_ADJOINT_PROF") != nullptr && mode == 0)
    fprintf(stderr, "[adjoint prof] attempts %lld  us: tile passes %.1f  weight-gradient passes %.1f  hand-off 1 %.1f  theta slices %.1f  hand-off 2 %.1f  epilogue %.1f\n",
            r.n_attempt, 0.01 * r.prof[0], 0.01 * r.prof[1], 0.01 * r.prof[2], 0.01 * r.prof[3], 0.01 * r.prof[4], 0.01 * r.prof[5]);
  if (stats != nullptr) {
    memset(stats, 0, sizeof(*stats));
    stats->n_attempts = r.n_attempt; stats->n_accepted = r.n_accept; stats->n_rejected = r.n_attempt - r.n_accept;
    stats->nfe = 2 + 6 * r.n_attempt;
    stats->t = reversed ? -r.t1 : r.t1; stats->dt = r.dt; stats->last_ratio = r.ratio; stats->status = r.status;
    stats->n_polls = 1; stats->n_launches = 1;
  }
  return (int)r.status;
}

extern "C" int mi_ode_adjoint_segment(mi_ode_adjoint_handle h, const mi_ode_rhs* rhs, const void* y_dev, const void* adj_y_dev,
                                      const void* adj_t_dev, const void* adj_params_dev, double t_start, double t_end, void* y_out_dev,
                                      void* adj_y_out_dev, void* adj_t_out_dev, void* adj_params_out_dev, mi_ode_stats* stats, void* stream) {
  if (h == nullptr || y_dev == nullptr || adj_y_dev == nullptr || adj_t_dev == nullptr || adj_params_dev == nullptr ||
      adj_y_out_dev == nullptr || adj_t_out_dev == nullptr || adj_params_out_dev == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (!(t_end != t_start)) {                      // _assert_increasing on the (possibly negated) pair (misc.py:158-159)
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->status = MI_ODE_ST_BAD_T; }
    return MI_ODE_ST_BAD_T;
  }
  return adj_launch(h, rhs, 0, y_dev, adj_y_dev, adj_t_dev, adj_params_dev, t_start, t_end, y_out_dev, adj_y_out_dev, adj_t_out_dev,
                    adj_params_out_dev, stats, (hipStream_t)stream);
}

extern "C" int mi_ode_adjoint_dynamics(mi_ode_adjoint_handle h, const mi_ode_rhs* rhs, const void* y_dev, const void* adj_y_dev,
                                       void* f_out_dev, void* vjp_y_out_dev, void* vjp_params_out_dev, void* stream) {
  if (h == nullptr || y_dev == nullptr || adj_y_dev == nullptr || f_out_dev == nullptr || vjp_y_out_dev == nullptr ||
      vjp_params_out_dev == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  // (theta / adj_t inputs are not read by the dynamics; the kernel wants valid pointers)
  return adj_launch(h, rhs, 1, y_dev, adj_y_dev, h->theta, vjp_params_out_dev, 0.0, 1.0, f_out_dev, vjp_y_out_dev, h->theta, vjp_params_out_dev,
                    nullptr, (hipStream_t)stream);
}

extern "C" int mi_ode_adjoint_dynamics_at(mi_ode_adjoint_handle h, const mi_ode_rhs* rhs, double t, const void* y_dev, const void* adj_y_dev,
                                          void* f_out_dev, void* vjp_y_out_dev, void* vjp_params_out_dev, void* stream) {
  if (h == nullptr || y_dev == nullptr || adj_y_dev == nullptr || f_out_dev == nullptr || vjp_y_out_dev == nullptr ||
      vjp_params_out_dev == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  return adj_launch(h, rhs, 1, y_dev, adj_y_dev, h->theta, vjp_params_out_dev, t, t + 1.0, f_out_dev, vjp_y_out_dev, h->theta, vjp_params_out_dev,
                    nullptr, (hipStream_t)stream);
}
