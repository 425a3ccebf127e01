// libmi_ode.so - C ABI (include/mi_ode.h) of the MI355X-native explicit RK engine: handle
// management, the host side of the attempt loop, and the stateless plane entry points.
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi_ode_control.h"
#include "mi_ode_host.h"
#include "mi_ode_plane.h"
#include "mi_ode_step_fused.h"
#include "mi_ode_persist.h"
#include "mi_ode_adams.h"
#include "mi_ode_adams_vc.h"
#include "mi_ode_adams_planes.h"
#include "mi_ode_mlp.h"
#include "mi_ode_plugin.h"

using namespace mi;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void mi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mi_ode_last_error(void) { return g_err; }
extern "C" int mi_ode_abi_version(void) { return MI_ODE_ABI_VERSION; }

extern "C" const char* mi_ode_status_string(uint32_t s) {
  if (s & MI_ODE_ST_BAD_T) return "t must be strictly increasing or decrasing";      // misc.py:159 (sic)
  if (s & MI_ODE_ST_NONFINITE) return "non-finite values in state `y`";               // dopri5.py:100
  if (s & MI_ODE_ST_MAX_STEPS) return "max_num_steps exceeded";                       // dopri5.py:85
  if (s & MI_ODE_ST_DT_UNDERFLOW) return "underflow in dt";                           // dopri5.py:98
  if (s & MI_ODE_ST_SYNC_TIMEOUT) return "engine: grid hand-off timed out (whole-integration kernel)";
  return "ok";
}

extern "C" int64_t mi_ode_reduce_workspace_bytes(void) { return (int64_t)kMaxBlocks * kRec * sizeof(double); }
extern "C" int64_t mi_ode_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(mi_ode_desc);
    case 1: return (int64_t)sizeof(mi_ode_stats);
    case 2: return (int64_t)sizeof(mi_ode_tableau);
    case 3: return (int64_t)sizeof(mi_ode_rhs);
    case 4: return (int64_t)sizeof(mi_ode_solver);          /* what a RHS plugin must have been compiled against */
    case 5: return (int64_t)sizeof(mi_ode_ctrl_params);
    case 6: return (int64_t)sizeof(mi_ode_adjoint_desc);
    case 7: return (int64_t)sizeof(mi_ode_opq_desc);
    case 8: return (int64_t)sizeof(mi_ode_linadj_desc);
    default: return -1;
  }
}

// ---- librccl, resolved at run time (the process usually has PyTorch-ROCm's copy loaded already) -------------------------
typedef int (*nccl_get_unique_id_fn)(void*);
typedef int (*nccl_comm_init_rank_fn)(void**, int, const void*, int);      // (ncclComm_t*, nranks, ncclUniqueId by value, rank)
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_comm_destroy_fn)(void*);
typedef const char* (*nccl_error_string_fn)(int);
struct NcclUniqueId { char internal[MI_ODE_RCCL_ID_BYTES]; };
typedef int (*nccl_comm_init_rank_byval_fn)(void**, int, NcclUniqueId, int);
static struct {
  int tried, ok;
  nccl_get_unique_id_fn get_id;
  nccl_comm_init_rank_byval_fn init_rank;
  nccl_all_gather_fn all_gather;
  nccl_comm_destroy_fn destroy;
  nccl_error_string_fn err;
} g_nccl;

static int load_rccl() {
  if (g_nccl.tried) return g_nccl.ok ? 0 : MI_ODE_E_EXCHANGE;
  g_nccl.tried = 1;
  void* lib = RTLD_DEFAULT;
  if (dlsym(RTLD_DEFAULT, "ncclAllGather") == nullptr) {
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (lib == nullptr) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (lib == nullptr) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (lib == nullptr) { mi_set_error("librccl not found: %s", dlerror()); return MI_ODE_E_EXCHANGE; }
  }
  g_nccl.get_id = (nccl_get_unique_id_fn)dlsym(lib, "ncclGetUniqueId");
  g_nccl.init_rank = (nccl_comm_init_rank_byval_fn)dlsym(lib, "ncclCommInitRank");
  g_nccl.all_gather = (nccl_all_gather_fn)dlsym(lib, "ncclAllGather");
  g_nccl.destroy = (nccl_comm_destroy_fn)dlsym(lib, "ncclCommDestroy");
  g_nccl.err = (nccl_error_string_fn)dlsym(lib, "ncclGetErrorString");
  if (!g_nccl.get_id || !g_nccl.init_rank || !g_nccl.all_gather || !g_nccl.destroy) { mi_set_error("librccl lacks the expected entry points"); return MI_ODE_E_EXCHANGE; }
  g_nccl.ok = 1;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
static inline int launch_stage(mi_ode_solver* h, int mode, int nk, StageArgs& A, hipStream_t st) {
  if (h->family == FAM_PLUGIN) {
    const int rc = h->plugin->launch_init(h, mode, nk, &A, st);
    if (rc != 0) { mi_set_error("plugin stage launch failed (mode %d nk %d): plugins provide F0 / INITB only", mode, nk); return rc; }
    h->n_launches += 1;
    return 0;
  }
  return h->is_f32 ? mi_launch_stage_f32(h, mode, nk, A, st) : mi_launch_stage_f64(h, mode, nk, A, st);
}

static inline int streaming_grid(long long n) {
  long long g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (g > kMaxBlocks) g = kMaxBlocks;
  return (int)g;
}

static void fill_common(mi_ode_solver* h, StageArgs& A) {
  memset(&A, 0, sizeof(A));
  A.ctl = h->ctl;
  A.planes = h->planes;
  A.stride = h->stride;
  A.batch = h->d.batch;
  A.dim = (int)h->d.dim;
  A.rtol = h->d.rtol;
  A.atol = h->d.atol;
  A.partials = h->partials;
  A.rhs = h->rhs;
  A.k_out_slot = -1;
}

// stage sigma (1-based) of an adaptive attempt in controller (device-state) mode
static int enqueue_adaptive_stage(mi_ode_solver* h, int sigma, hipStream_t st) {
  const mi_ode_tableau& tb = h->d.tableau;
  StageArgs A;
  fill_common(h, A);
  const int nk = sigma;
  for (int j = 0; j < nk; ++j) A.a[j] = tb.beta[sigma - 1][j];
  A.alpha = tb.alpha[sigma - 1];
  A.k_out_slot = sigma;
  int mode = M_STAGE;
  if (sigma == h->S) {
    mode = M_LAST_FSAL;
    for (int j = 0; j <= nk; ++j) A.e[j] = tb.c_error[j];
  }
  return launch_stage(h, mode, nk, A, st);
}

static void fill_step_args(mi_ode_solver* h, StepArgs& A);
static void fill_mlp_args(mi_ode_solver* h, MlpArgs& M);

// all S stages of one attempt: six stage launches, or one whole-attempt launch
static int enqueue_attempt_kernels(mi_ode_solver* h, hipStream_t st, hipEvent_t ev_last) {
  if (h->family == FAM_MLP) {
    MlpArgs M;
    fill_mlp_args(h, M);
    if (ev_last) (void)hipEventRecord(ev_last, st);
    return h->is_f32 ? mi_launch_mlp_f32(h, MLP_STEP, M, st) : mi_launch_mlp_f64(h, MLP_STEP, M, st);
  }
  if (h->step_fused) {
    StepArgs A;
    fill_step_args(h, A);
    if (ev_last) (void)hipEventRecord(ev_last, st);
    if (h->family == FAM_PLUGIN) {
      const int rcp = h->plugin->launch_step(h, &A, st);
      if (rcp != 0) { mi_set_error("plugin step kernel launch failed"); return rcp; }
      h->n_launches += 1;
      return 0;
    }
    return h->is_f32 ? mi_launch_step_f32(h, A, st) : mi_launch_step_f64(h, A, st);
  }
  for (int sigma = 1; sigma <= h->S; ++sigma) {
    if (ev_last && sigma == h->S) (void)hipEventRecord(ev_last, st);
    int rc = enqueue_adaptive_stage(h, sigma, st);
    if (rc != 0) return rc;
  }
  return 0;
}

// reduce -> (exchange) -> controller
static int enqueue_controller(mi_ode_solver* h, int phase, hipStream_t st) {
  const int nblocks = ((phase == PH_ATTEMPT && h->step_fused) || h->init_tiles16) ? h->step_grid : h->stage_grid;
  if (h->d.world_size > 1 || h->d.allgather != nullptr || h->nccl_comm != nullptr) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, (const Ctl*)h->ctl, (const double*)h->partials,
                       nblocks, h->n, h->rank_rec);
    if (h->nccl_comm != nullptr) {                 // RCCL straight from here: stream-ordered between the two kernels
      const int nrc = g_nccl.all_gather(h->rank_rec, h->gathered, (size_t)kRec, /*ncclDouble*/ 8, h->nccl_comm, st);
      if (nrc != 0) {
        mi_set_error("ncclAllGather failed: %s", g_nccl.err ? g_nccl.err(nrc) : "?");
        return MI_ODE_E_EXCHANGE;
      }
    } else {
      if (h->d.allgather == nullptr) {
        mi_set_error("world_size > 1 needs mi_ode_rccl_connect or an allgather hook");
        return MI_ODE_E_INVALID;
      }
      const int rc = h->d.allgather(h->d.allgather_user, h->rank_rec, h->gathered, kRec, (void*)st);
      if (rc != 0) {
        mi_set_error("allgather hook returned %d", rc);
        return MI_ODE_E_EXCHANGE;
      }
    }
    hipLaunchKernelGGL(k_controller, dim3(1), dim3(256), 0, st, h->ctl, (const double*)nullptr, 0,
                       (const double*)h->gathered, (int)h->d.world_size, phase, h->cp);
    h->n_launches += 2;
  } else {
    hipLaunchKernelGGL(k_controller, dim3(1), dim3(256), 0, st, h->ctl, (const double*)h->partials, nblocks,
                       (const double*)nullptr, 1, phase, h->cp);
    h->n_launches += 1;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mi_set_error("controller launch failed: %s", hipGetErrorString(e));
    return MI_ODE_E_HIP;
  }
  return 0;
}

template <typename T>
static void launch_emit_t(mi_ode_solver* h, void* out, hipStream_t st) {
  const int g = streaming_grid(h->n);
  if (h->S + 1 == 7)
    hipLaunchKernelGGL((k_emit<T, 7>), dim3(g), dim3(256), 0, st, (const Ctl*)h->ctl, (const char*)h->planes, h->stride,
                       h->n, (const double*)h->t_out_dev, (T*)out, h->ip);
  else
    hipLaunchKernelGGL((k_emit<T, 4>), dim3(g), dim3(256), 0, st, (const Ctl*)h->ctl, (const char*)h->planes, h->stride,
                       h->n, (const double*)h->t_out_dev, (T*)out, h->ip);
}

static int enqueue_emit(mi_ode_solver* h, void* out, hipStream_t st) {
  if (h->step_fused) return 0;                 // the whole-attempt kernels emit from registers
  if (h->is_f32) launch_emit_t<float>(h, out, st);
  else launch_emit_t<double>(h, out, st);
  h->n_launches += 1;
  return 0;
}

static int poll_ctl(mi_ode_solver* h, hipStream_t st) {
  MI_HIP(hipMemcpyAsync(h->ctl_host, h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, st));
  MI_HIP(hipStreamSynchronize(st));
  h->n_polls += 1;
  return 0;
}

// after a poll (stream idle): fold the event pairs of the attempts that really ran into the totals
static void harvest_profile(mi_ode_solver* h) {
  if (!h->d.profile || !h->ev_ready) return;
  const long long real = h->ctl_host->n_attempt;
  for (long long g = h->prof_done; g < h->enq_attempts; ++g) {
    if (g < real) {
      const int i = (int)(g % 64);
      float ms_last = 0.f, ms_all = 0.f;
      if (hipEventElapsedTime(&ms_last, h->ev_b[i], h->ev_c[i]) == hipSuccess &&
          hipEventElapsedTime(&ms_all, h->ev_a[i], h->ev_c[i]) == hipSuccess) {
        h->prof_last_ms += ms_last; h->prof_all_ms += ms_all; h->prof_n += 1;
      }
    }
  }
  h->prof_done = h->enq_attempts;
}

static void fill_stats(mi_ode_solver* h, mi_ode_stats* s) {
  const Ctl* c = h->ctl_host;
  s->n_attempts = c->n_attempt;
  s->n_accepted = c->n_accept;
  s->n_rejected = c->n_reject;
  s->nfe = c->nfe;
  s->t = c->t1;
  s->dt = c->dt;
  s->last_ratio = c->ratio;
  s->status = c->status;
  s->n_polls = h->n_polls;
  s->n_launches = h->n_launches;
  s->clock_mhz = c->clk_ticks > 0 ? 100.0 * (double)c->clk_cycles / (double)c->clk_ticks : 0.0;
  s->handoff_us = (h->family == mi::FAM_LINEAR_MFMA && h->persist) ? 0.01 * (double)c->prof[3] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------
static int pick_family(mi_ode_solver* h) {
  const mi_ode_rhs& r = h->d.rhs;
  const int D = (int)h->d.dim;
  // any dim up to 128: zero padded to the next tile width (16, 32, 64, 128), W resident in registers.  129 .. 256 (round 6): the
  // 256-wide tile kernels with W streamed from L2 - the whole-attempt / whole-call kernels of the three- and six-row FSAL tableaus
  // and the fixed-grid kernel (their per-stage schedule and every other tableau stay with the vector-ALU family).
  const mi_ode_tableau& tbp = h->d.tableau;
  const bool wide_ok = D > 128 && D <= 256 && h->d.multistep == 0 && h->d.fusion != 1 &&
                       (h->d.adaptive ? (tbp.fsal && (tbp.n_stages == 3 || tbp.n_stages == 6)) : (tbp.n_stages == 0 || tbp.n_stages == 3));
  const bool mfma_dim = (D >= 3 && D <= 128) || wide_ok;
  h->lin_dp = D <= 16 ? 16 : (D <= 32 ? 32 : (D <= 64 ? 64 : (D <= 128 ? 128 : 256)));
  h->rhs.cube = 0;
  switch (r.kind) {
    case MI_ODE_RHS_LOTKA_VOLTERRA:
      if (D != 2) { mi_set_error("lotka_volterra needs dim 2"); return MI_ODE_E_INVALID; }
      h->family = FAM_LV; return 0;
    case MI_ODE_RHS_LORENZ:
      if (D != 3) { mi_set_error("lorenz needs dim 3"); return MI_ODE_E_INVALID; }
      h->family = FAM_LORENZ; return 0;
    case MI_ODE_RHS_CUBIC_LINEAR:
    case MI_ODE_RHS_LINEAR: {
      const bool cube = r.kind == MI_ODE_RHS_CUBIC_LINEAR;
      if (r.w[0] == nullptr) { mi_set_error("linear RHS needs W"); return MI_ODE_E_INVALID; }
      if (D == 2 && r.b[0] == nullptr && h->d.linear_variant == 0) {
        // W travels by value (scalars[0..3] = W row-major); the caller fills them for dim 2
        h->family = cube ? FAM_CUBIC2 : FAM_LINEAR2; return 0;
      }
      h->rhs.cube = cube ? 1 : 0;
      if (!cube && mfma_dim && h->d.linear_variant != 1) { h->family = FAM_LINEAR_MFMA; return 0; }
      if (h->d.linear_variant == 2) { mi_set_error("MFMA linear kernel needs 3 <= dim <= 128 (<= 256 for dopri5 / tsit5 / bosh3 / the fixed grid) and no cube"); return MI_ODE_E_INVALID; }
      if (D > 256) { mi_set_error("fused linear RHS supports dim <= 256 (got %d)", D); return MI_ODE_E_INVALID; }
      h->family = FAM_LINEAR_VALU; return 0;
    }
    case MI_ODE_RHS_PLUGIN: {
      const mi_ode_rowlocal_plugin* pl = (const mi_ode_rowlocal_plugin*)r.plugin;
      if (pl == nullptr) { mi_set_error("MI_ODE_RHS_PLUGIN needs mi_ode_rhs.plugin"); return MI_ODE_E_INVALID; }
      if (pl->abi != MI_ODE_PLUGIN_ABI || pl->solver_size != sizeof(mi_ode_solver)) {
        mi_set_error("RHS plugin was built against different headers (abi %d vs %d, handle %zu vs %zu bytes): rebuild it", pl->abi,
                     MI_ODE_PLUGIN_ABI, pl->solver_size, sizeof(mi_ode_solver));
        return MI_ODE_E_INVALID;
      }
      if (pl->dtype != h->d.dtype || pl->dim != D) { mi_set_error("RHS plugin is for dtype %d dim %d, the state is dtype %d dim %d", pl->dtype, pl->dim, h->d.dtype, D); return MI_ODE_E_INVALID; }
      h->plugin = pl;
      if (pl->cooperative) {                       // a thread per state element: whole-call kernel (adaptive) and the multistep kernels only
        const bool adaptive_ok = h->d.adaptive && h->d.multistep == 0 && (h->d.fusion == 0 || h->d.fusion == 4);
        const bool fixed_ok = !h->d.adaptive && h->d.multistep == 0 && h->d.fusion != 1 && (h->d.tableau.n_stages == 0 || h->d.tableau.n_stages == 3);
        if (D < 1 || D > 256 || !(adaptive_ok || fixed_ok || h->d.multistep != 0) || h->d.world_size > 1 || h->d.allgather != nullptr || h->d.n_segments > 1 ||
            pl->persist_fn == nullptr || pl->multistep_fn == nullptr) {
          mi_set_error("cooperative RHS plugin: dim <= 256; an adaptive solver (fusion auto / whole), euler / rk4 on a fixed grid or the Adams family; one rank, one tensor");
          return MI_ODE_E_INVALID;
        }
        h->family = FAM_PLUGIN_COOP; return 0;
      }
      h->family = FAM_PLUGIN; return 0;
    }
    case MI_ODE_RHS_MLP_TANH: {
      const int hd = r.hidden;
      if (h->d.multistep != 0) {                   // the Adams family: a thread per state element, the three layers through LDS (RhsMlpCoop)
        if (D < 1 || D > 256 || hd < 1 || hd > 256 || !r.w[0] || !r.w[1] || !r.w[2]) {
          mi_set_error("multistep kernels for the MLP: dim <= 256, hidden <= 256 (got %d, %d)", D, hd);
          return MI_ODE_E_INVALID;
        }
        h->family = FAM_MLP_COOP; return 0;
      }
      // the MFMA tile kernels: float32 (weights resident in registers, mi_ode_mlp.h) and - round 6 - float64 (weights streamed from a
      // packed copy, mi_ode_mlp64.h: instantiated for the padded geometries 16 x 16 and 64 x 128)
      const bool one_row = h->d.adaptive && h->d.tableau.n_stages == 1;      // adaptive_heun: no tile kernel - the cooperative one
      const bool tile_box = D >= 1 && D <= 64 && hd >= 1 && hd <= 128 && !one_row;
      const bool fixed_rk = !h->d.adaptive && h->d.multistep == 0 && (h->d.tableau.n_stages == 0 || h->d.tableau.n_stages == 3);
      if ((h->d.adaptive || fixed_rk) && !tile_box) {
        // outside the MFMA tile kernels' box (float64, dim > 64, hidden > 128): the cooperative whole-call kernel - a thread per state
        // element, the three layers through LDS (RhsMlpCoop, round 5) - for batches whose workgroups are co-resident; one rank, one tensor
        if (D < 1 || D > 256 || hd < 1 || hd > 256 || !r.w[0] || !r.w[1] || !r.w[2] || h->d.world_size > 1 || h->d.allgather != nullptr ||
            h->d.n_segments > 1 || (h->d.adaptive && h->d.fusion != 0 && h->d.fusion != 4) || (fixed_rk && h->d.fusion == 1)) {
          mi_set_error("MLP outside the tile kernels (float32, dim <= 64, hidden <= 128): the cooperative whole-call kernel takes dim, hidden <= 256, one rank, one tensor, fusion auto / whole (got dim %d, hidden %d)", D, hd);
          return MI_ODE_E_INVALID;
        }
        h->family = FAM_MLP_COOP; return 0;
      }
      if (!h->d.adaptive && (h->d.multistep != 0 || (h->d.tableau.n_stages != 0 && h->d.tableau.n_stages != 3))) {
        mi_set_error("fused MLP kernels on a fixed grid: euler or rk4 (3/8 rule) in one launch (k_fixed_mlp); no multistep kernel");
        return MI_ODE_E_INVALID;
      }
      if (D < 1 || D > 64 || hd < 1 || hd > 128 || !r.w[0] || !r.w[1] || !r.w[2]) {
        mi_set_error("fused MLP kernel supports dim <= 64, hidden <= 128 (got %d, %d)", D, hd);
        return MI_ODE_E_INVALID;
      }
      h->mlp_dp = D <= 16 ? 16 : 64;
      h->mlp_hp = hd <= 16 ? 16 : 128;
      if (!h->is_f32 && h->mlp_dp != (h->mlp_hp == 16 ? 16 : 64)) { h->mlp_dp = 64; h->mlp_hp = 128; }   // float64: two geometries (16 x 16, 64 x 128)
      h->family = FAM_MLP; return 0;
    }
    default:
      mi_set_error("RHS kind %d has no fused kernel yet", r.kind);
      return MI_ODE_E_INVALID;
  }
}

// The one-launch multistep kernels (mi_ode_adams.h, mi_ode_adams_vc.h): a thread per trajectory for the row-local families, a
// thread per state element - floor(256 / dim) trajectories per workgroup - for the matrix families (RhsLinearCoop).
static bool multistep_family(const mi_ode_solver* h) {
  const bool rowlocal = h->family == FAM_CUBIC2 || h->family == FAM_LINEAR2 || h->family == FAM_LV || h->family == FAM_LORENZ ||
                        h->family == FAM_PLUGIN;
  const bool coop = (h->family == FAM_LINEAR_MFMA || h->family == FAM_LINEAR_VALU) && h->d.dim >= 1 && h->d.dim <= 256;
  return rowlocal || coop || h->family == FAM_MLP_COOP || h->family == FAM_PLUGIN_COOP;
}
static long long multistep_grid(const mi_ode_solver* h) {
  if (h->family == FAM_LINEAR_MFMA || h->family == FAM_LINEAR_VALU) {
    const long long tpw = 256 / h->d.dim;
    return (h->d.batch + tpw - 1) / tpw;
  }
  if (h->family == FAM_MLP_COOP) {
    const long long tpw = RhsMlpCoop<float>::tpw(h->rhs, (int)h->d.dim);
    return (h->d.batch + tpw - 1) / tpw;
  }
  if (h->family == FAM_PLUGIN_COOP) {
    const long long tpw = 256 / h->d.dim;            // (RhsUserCoop::tpw, rhs.CustomCoop)
    return (h->d.batch + tpw - 1) / tpw;
  }
  return (h->d.batch + 255) / 256;
}
static void multistep_rhs(const mi_ode_solver* h, RhsParams& r) {
  if (h->family == FAM_LINEAR_MFMA || h->family == FAM_LINEAR_VALU) r.hidden = (int)h->d.dim;   // (RhsLinearCoop reads the row length here)
  if (h->family == FAM_MLP_COOP) r.cube = (int)h->d.dim;                                         // (RhsMlpCoop: the aux field carries dim)
}

static void fill_step_args(mi_ode_solver* h, StepArgs& A) {
  const mi_ode_tableau& tb = h->d.tableau;
  memset(&A, 0, sizeof(A));
  A.ctl = h->ctl; A.planes = h->planes; A.stride = h->stride; A.batch = h->d.batch; A.dim = (int)h->d.dim;
  A.interp = h->d.interp; A.out = h->cur_out; A.t_out = h->t_out_dev; A.n_plane = h->n;
  for (int i = 0; i < h->S; ++i) {
    A.alpha[i] = tb.alpha[i];
    for (int j = 0; j <= i; ++j) A.beta[i][j] = tb.beta[i][j];
  }
  for (int j = 0; j <= h->S; ++j) { A.e[j] = tb.c_error[j]; A.cmid[j] = tb.c_mid[j]; A.csol[j] = tb.c_sol[j]; }
  A.partials = h->partials; A.rhs = h->rhs;
  A.ticket = h->fused_ctl ? h->ticket : nullptr;
  A.cp = h->cp;
}

static void fill_mlp_args(mi_ode_solver* h, MlpArgs& M) {
  memset(&M, 0, sizeof(M));
  fill_step_args(h, M.step);
  M.rtol = h->d.rtol; M.atol = h->d.atol; M.hidden = h->d.rhs.hidden;
}

static void close_peers(mi_ode_solver* h) {
  for (int q = 0; q < h->xpeer_world && q < 64; ++q)
    if (h->xpeer_open[q] != nullptr && h->xpeer_open[q] != h->xpeer_local) (void)hipIpcCloseMemHandle(h->xpeer_open[q]);
  memset(h->xpeer_open, 0, sizeof(h->xpeer_open));
  h->xpeer_world = 0;
  if (h->xpeer_tab_dev) { (void)hipFree(h->xpeer_tab_dev); h->xpeer_tab_dev = nullptr; }
}

extern "C" int mi_ode_destroy(mi_ode_handle h) {
  if (h != nullptr) {
    close_peers(h);
    if (h->xpeer_local) { (void)hipFree(h->xpeer_local); h->xpeer_local = nullptr; }
    if (h->nccl_comm && g_nccl.ok) { (void)g_nccl.destroy(h->nccl_comm); h->nccl_comm = nullptr; }
  }
  if (h != nullptr && h->xrank_registered) { (void)hipHostUnregister(h->d.xrank_host); h->xrank_registered = 0; }
  if (h != nullptr && h->gbuf) { (void)hipFree(h->gbuf); h->gbuf = nullptr; }
  if (h == nullptr) return 0;
  if (h->planes) (void)hipFree(h->planes);
  if (h->partials) (void)hipFree(h->partials);
  if (h->adams_tab) (void)hipFree(h->adams_tab);
  if (h->mlp_pack) (void)hipFree(h->mlp_pack);
  if (h->lin_pack) (void)hipFree(h->lin_pack);
  if (h->adams_res) (void)hipHostFree(h->adams_res);
  if (h->rank_rec && h->own_exchange) (void)hipFree(h->rank_rec);
  if (h->gathered && h->own_exchange) (void)hipFree(h->gathered);
  if (h->ctl) (void)hipFree(h->ctl);
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->t_out_dev) (void)hipFree(h->t_out_dev);
  if (h->ev_ready) for (int i = 0; i < 64; ++i) { (void)hipEventDestroy(h->ev_a[i]); (void)hipEventDestroy(h->ev_b[i]); (void)hipEventDestroy(h->ev_c[i]); }
  if (h->ctl_host) (void)hipHostFree(h->ctl_host);
  if (h->t_out_host) (void)hipHostFree(h->t_out_host);
  delete h;
  return 0;
}

extern "C" int mi_ode_create(const mi_ode_desc* desc, mi_ode_handle* out) {
  if (desc == nullptr || out == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { mi_set_error("no HIP device"); return MI_ODE_E_NODEVICE; }
  if (desc->dtype != MI_ODE_F32 && desc->dtype != MI_ODE_F64) { mi_set_error("bad dtype"); return MI_ODE_E_INVALID; }
  if (desc->batch <= 0 || desc->dim <= 0) { mi_set_error("empty state"); return MI_ODE_E_INVALID; }
  const mi_ode_tableau& tb = desc->tableau;
  if (tb.n_stages < 0 || tb.n_stages > MI_ODE_MAX_STAGES) { mi_set_error("bad n_stages"); return MI_ODE_E_INVALID; }
  if (desc->adaptive) {
    // kernels are instantiated for: 3 and 6 rows, FSAL shaped (bosh3, dopri5, tsit5; every family and schedule);
    // 13 rows FSAL shaped (dopri8): row-local families and the MFMA-linear tile kernels; 1 row not FSAL shaped (adaptive_heun):
    // row-local families; schedules 2-4 only
    const bool classic = tb.fsal && (tb.n_stages == 3 || tb.n_stages == 6);
    const bool wide = (tb.fsal && tb.n_stages == 13) || (!tb.fsal && tb.n_stages == 1);
    if (!classic && !wide && desc->multistep != 3) { mi_set_error("fused adaptive engine: no kernels for a %d-row %s tableau", tb.n_stages, tb.fsal ? "FSAL-shaped" : "non-FSAL"); return MI_ODE_E_INVALID; }
    if (desc->interp != MI_ODE_INTERP_QUARTIC_MID && tb.n_stages != 6) { mi_set_error("tsit5 dense output needs 7 stage derivatives"); return MI_ODE_E_INVALID; }
  }
  mi_ode_solver* h = new mi_ode_solver();
  memset(h, 0, sizeof(*h));
  h->d = *desc;
  if (h->d.world_size < 1) h->d.world_size = 1;
  h->is_f32 = desc->dtype == MI_ODE_F32;
  h->elt = h->is_f32 ? 4 : 8;
  h->n = desc->batch * desc->dim;
  h->stride = ((h->n * (long long)h->elt + 255) / 256) * 256;
  h->S = tb.n_stages;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    mi_set_error("cannot query device");
    mi_ode_destroy(h);
    return MI_ODE_E_HIP;
  }
  h->num_cus = prop.multiProcessorCount;
  // RHS parameters
  memcpy(h->rhs.s, desc->rhs.scalars, sizeof(h->rhs.s));
  for (int i = 0; i < 3; ++i) { h->rhs.w[i] = desc->rhs.w[i]; h->rhs.b[i] = desc->rhs.b[i]; }
  h->rhs.sign = desc->rhs.sign == 0.0 ? 1.0 : desc->rhs.sign;
  h->rhs.hidden = desc->rhs.hidden;
  int rc = pick_family(h);
  if (rc != 0) { mi_ode_destroy(h); return rc; }
  if (h->family == FAM_LINEAR_MFMA && h->lin_dp == 256) {   // streamed W: the copy in consumption order (256 x 256 elements, refreshed before every launch)
    if (hipMalloc(&h->lin_pack, (size_t)256 * 256 * (size_t)h->elt) != hipSuccess) {
      (void)hipGetLastError();
      mi_set_error("linear tile kernels (dim > 128): cannot allocate the copy of W");
      mi_ode_destroy(h); return MI_ODE_E_HIP;
    }
  }
  if (h->family == FAM_MLP && !h->is_f32) {          // float64 tile kernels: the packed copy of the weights (refreshed before every launch)
    const int nd = mi_mlp64_pack_doubles(h->mlp_dp, h->mlp_hp);
    if (nd <= 0 || hipMalloc((void**)&h->mlp_pack, (size_t)nd * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      mi_set_error("float64 MLP kernels: cannot allocate the weight pack (%d doubles)", nd);
      mi_ode_destroy(h); return nd <= 0 ? MI_ODE_E_INVALID : MI_ODE_E_HIP;
    }
  }
  if (desc->adaptive && desc->multistep != 3 && tb.n_stages != 3 && tb.n_stages != 6) {   // (multistep = 3: no tableau at all)
    const bool rowlocal_fam = h->family == FAM_CUBIC2 || h->family == FAM_LINEAR2 || h->family == FAM_LV || h->family == FAM_LORENZ ||
                              h->family == FAM_PLUGIN || h->family == FAM_PLUGIN_COOP || h->family == FAM_MLP_COOP;
    const bool mfma13 = (h->family == FAM_LINEAR_MFMA || h->family == FAM_MLP) && tb.fsal && tb.n_stages == 13;   // dopri8 on the tile kernels
    if (!(rowlocal_fam || mfma13) || desc->fusion == 1) {
      mi_set_error("%d-row tableaus run on the whole-attempt / whole-call kernels of the row-local families and (13 rows) of the MFMA-linear and MLP families only (no per-stage kernels)", tb.n_stages);
      mi_ode_destroy(h); return MI_ODE_E_INVALID;
    }
  }
  h->nseg = desc->n_segments > 1 ? desc->n_segments : 0;
  if (h->nseg > 1) {                               // tuple state (include/mi_ode.h): row-local kernels, whole-call schedule
    const bool rowlocal_fam = h->family == FAM_CUBIC2 || h->family == FAM_LINEAR2 || h->family == FAM_LV || h->family == FAM_LORENZ ||
                              h->family == FAM_PLUGIN;
    long long rows = 0;
    bool ok = h->nseg <= MI_ODE_MAX_SEGMENTS && rowlocal_fam && desc->adaptive &&
              !(desc->controller == MI_ODE_CTRL_TSIT5 && desc->seg_tolerances) &&       // tsit5 takes scalar tolerances (tsit5.py:81-82)
              h->d.world_size <= 1 && desc->allgather == nullptr && (desc->fusion == 0 || desc->fusion == 4);
    for (int k = 0; ok && k < h->nseg; ++k) {
      if (desc->seg_rows[k] < 1) ok = false;
      h->seg_blk[k] = (int)(rows / MI_ODE_SEGMENT_ALIGN);
      rows += (desc->seg_rows[k] + MI_ODE_SEGMENT_ALIGN - 1) / MI_ODE_SEGMENT_ALIGN * MI_ODE_SEGMENT_ALIGN;
    }
    if (ok) h->seg_blk[h->nseg] = (int)(rows / MI_ODE_SEGMENT_ALIGN);
    if (!ok || rows != desc->batch) {
      mi_set_error("tuple states: 2..%d components of >= 1 row each, every component padded to %d rows (batch = the padded total), a row-local RHS, an "
                   "adaptive tableau, one rank, fusion 0 or 4", MI_ODE_MAX_SEGMENTS, MI_ODE_SEGMENT_ALIGN);
      mi_ode_destroy(h); return MI_ODE_E_INVALID;
    }
  }
  rc = h->is_f32 ? mi_stage_geometry_f32(h) : mi_stage_geometry_f64(h);
  if (rc != 0) { mi_ode_destroy(h); return rc; }
  if (desc->multistep == 3) {                      // the variable-order Adams solver in one launch (mi_ode_adams_vc.h)
    const bool rowlocal_cat = multistep_family(h);
    if (!desc->adaptive || !rowlocal_cat || desc->ms_gamma_star == nullptr || desc->ms_max_order < 1 || desc->ms_max_order > kVcMaxOrder ||
        h->d.world_size > 1 || h->nseg > 1 || desc->controller != MI_ODE_CTRL_MISC) {
      mi_set_error("multistep = 3 ('adams'): adaptive = 1, a row-local catalogue system or a matrix RHS of dim <= 256, one rank, one tensor, the misc controller; 1 <= max_order <= %d, gamma_star", kVcMaxOrder);
      mi_ode_destroy(h); return MI_ODE_E_INVALID;
    }
    const long long g = multistep_grid(h);
    if (g > 1) {                                   // every attempt's error ratio crosses the workgroups: they must be co-resident
      const int cap = h->is_f32 ? mi_adams_vc_capacity_f32(h) : mi_adams_vc_capacity_f64(h);
      if (g > cap || g > kPersistMaxGrid) {
        mi_set_error("adams in one launch: %lld workgroups cannot be co-resident on this device (%d)", g, cap);
        mi_ode_destroy(h); return MI_ODE_E_INVALID;
      }
    }
    memcpy(h->adams_gamma_star, desc->ms_gamma_star, sizeof(h->adams_gamma_star));
    hipError_t ea = hipHostMalloc((void**)&h->adams_res, 4 * sizeof(long long), hipHostMallocDefault);
    if (ea != hipSuccess) { mi_set_error("multistep result record: %s", hipGetErrorString(ea)); mi_ode_destroy(h); return MI_ODE_E_HIP; }
    for (int i = 0; i < 4; ++i) h->adams_res[i] = 0;
    h->d.ms_gamma_star = nullptr;                  // (caller-owned host array: not kept)
  } else
  if (desc->multistep != 0) {                      // fixed-grid Adams family in one launch (mi_ode_adams.h)
    const bool rowlocal_cat = multistep_family(h);
    if (desc->adaptive || !rowlocal_cat || (desc->multistep != 1 && desc->multistep != 2) || desc->ms_ab == nullptr || desc->ms_am == nullptr ||
        desc->ms_am0 == nullptr || desc->ms_max_order < 1 || desc->ms_max_order > kAdamsMaxOrder || desc->ms_max_iters < 1 ||
        desc->ms_min_order < 1 || h->d.world_size > 1 || h->nseg > 1) {
      mi_set_error("multistep: fixed grid, a row-local catalogue system or a matrix RHS of dim <= 256, one rank, one tensor; 1 <= max_order <= %d, max_iters >= 1, coefficient tables", kAdamsMaxOrder);
      mi_ode_destroy(h); return MI_ODE_E_INVALID;
    }
    const long long g = multistep_grid(h);
    if (desc->multistep == 2 && g > 1) {           // the corrector's convergence test couples the workgroups: they must be co-resident
      const int cap = h->is_f32 ? mi_adams_capacity_f32(h) : mi_adams_capacity_f64(h);
      if (g > cap || g > kPersistMaxGrid) {
        mi_set_error("fixed_adams in one launch: %lld workgroups cannot be co-resident on this device (%d)", g, cap);
        mi_ode_destroy(h); return MI_ODE_E_INVALID;
      }
    }
    double tab[2 * 13 * 12 + 13];
    memcpy(tab, desc->ms_ab, sizeof(double) * 13 * 12);
    memcpy(tab + 13 * 12, desc->ms_am, sizeof(double) * 13 * 12);
    memcpy(tab + 2 * 13 * 12, desc->ms_am0, sizeof(double) * 13);
    hipError_t ea = hipMalloc((void**)&h->adams_tab, sizeof(tab));
    if (ea == hipSuccess) ea = hipMemcpy(h->adams_tab, tab, sizeof(tab), hipMemcpyHostToDevice);
    if (ea == hipSuccess) ea = hipHostMalloc((void**)&h->adams_res, 2 * sizeof(long long), hipHostMallocDefault);
    if (ea != hipSuccess) { mi_set_error("multistep tables: %s", hipGetErrorString(ea)); mi_ode_destroy(h); return MI_ODE_E_HIP; }
    h->adams_res[0] = h->adams_res[1] = 0;
    h->d.ms_ab = h->d.ms_am = h->d.ms_am0 = nullptr;   // (caller-owned host arrays: not kept)
  }
  {   // whole-attempt fusion: row-local families and the MFMA linear family, adaptive FSAL tableaus
    const bool can = desc->adaptive && (h->family == FAM_CUBIC2 || h->family == FAM_LINEAR2 || h->family == FAM_LV ||
                                        h->family == FAM_LORENZ || h->family == FAM_LINEAR_MFMA || h->family == FAM_MLP ||
                                        h->family == FAM_PLUGIN);
    if (h->family == FAM_PLUGIN && desc->fusion == 1) { mi_set_error("RHS plugins have no per-stage kernels (fusion = 1)"); mi_ode_destroy(h); return MI_ODE_E_INVALID; }
    if (h->family == FAM_MLP && desc->fusion == 1) { mi_set_error("the MLP family only has a whole-attempt kernel"); mi_ode_destroy(h); return MI_ODE_E_INVALID; }
    const bool can_fixed = !desc->adaptive && (h->family == FAM_CUBIC2 || h->family == FAM_LINEAR2 || h->family == FAM_LV ||
                                               h->family == FAM_LORENZ || h->family == FAM_LINEAR_MFMA || h->family == FAM_MLP ||
                                               h->family == FAM_PLUGIN);
    if (desc->fusion == 2 && !can && !can_fixed) { mi_set_error("fusion=2: no whole-attempt kernel for this problem"); mi_ode_destroy(h); return MI_ODE_E_INVALID; }
    h->step_fused = (can && desc->fusion != 1) ? 1 : 0;
    h->ts_dense = (desc->interp != MI_ODE_INTERP_QUARTIC_MID) ? 1 : 0;
    // In-kernel controller (last workgroup): measured equal or better than a separate k_controller launch for the
    // heavy MFMA / MLP kernels and for small grids; for the feather-weight row-local kernels on many workgroups the
    // serial tail (ticket + sc1 record reads) costs more than the extra launch (13.6 vs 15.5 us/attempt at config 3).
    const bool light_many = (h->family != FAM_LINEAR_MFMA && h->family != FAM_MLP) && h->step_grid > 32;
    h->fused_ctl = (h->step_fused && h->d.world_size <= 1 && desc->allgather == nullptr && desc->fusion != 3 &&
                    !(light_many && desc->fusion == 0)) ? 1 : 0;
  }
  {   // whole integration in one launch (mi_ode_persist.h): single rank, every workgroup co-resident (the in-kernel
      // hand-off spins).  Row-local systems: one trajectory per thread; linear MFMA family: the persistent tile grid.
    const bool rowlocal = h->family == FAM_CUBIC2 || h->family == FAM_LINEAR2 || h->family == FAM_LV || h->family == FAM_LORENZ ||
                          h->family == FAM_PLUGIN;
    const bool mfma = h->family == FAM_LINEAR_MFMA && h->step_fused;
    const bool mlp = h->family == FAM_MLP;
    const bool coop = (h->family == FAM_MLP_COOP || h->family == FAM_PLUGIN_COOP) && desc->multistep == 0;
    h->init_tiles16 = mfma ? 1 : 0;
    long long g = (mfma || mlp) ? (long long)h->step_grid : coop ? multistep_grid(h) : (desc->batch + 255) / 256;
    const bool single = h->d.world_size <= 1 && desc->allgather == nullptr;
    if (desc->xrank_host != nullptr) {           // cross-rank hand-off segment: make it visible to this GPU
      if (h->d.world_size > kXMaxWorld || desc->xrank_bytes < mi_ode_xrank_bytes(h->d.world_size)) {
        mi_set_error("xrank_host: world_size <= %d and at least %lld bytes are needed", kXMaxWorld, (long long)mi_ode_xrank_bytes(h->d.world_size));
        mi_ode_destroy(h); return MI_ODE_E_INVALID;
      }
      void* dptr = nullptr;
      hipError_t re = hipHostRegister(desc->xrank_host, (size_t)desc->xrank_bytes, hipHostRegisterMapped | hipHostRegisterPortable);
      if (re == hipSuccess) { h->xrank_registered = 1; re = hipHostGetDevicePointer(&dptr, desc->xrank_host, 0); }
      if (re != hipSuccess) {
        (void)hipGetLastError();
        if (h->xrank_registered) { (void)hipHostUnregister(desc->xrank_host); h->xrank_registered = 0; }
        dptr = nullptr;                            // not fatal: the allgather hook path remains
      }
      h->xrank_dev = (double*)dptr;
    }
    // The tile kernels of the linear family walk their tiles with the grid's stride: ANY co-resident grid serves any batch.  (Round 6: the
    // per-attempt grid - up to 8 workgroups per CU at the narrow widths - used to be taken as it was, and 313 tiles at dim 64 or 4096 at
    // dim 16 fell back to one launch per attempt because THAT grid was not co-resident.)
    // Measured (scripts/bench_linear_grids.py, profiles/r06_linear_grid_clamp.txt): the clamped whole-call grid wins 6 - 18 % over one launch per
    // attempt up to a few million multiply-adds per tile column of work (313 .. 513 tiles at dim 64, 4096 tiles at dim 16, 375 at dim 100) and
    // loses 3 % at 4096 tiles x dim 64 (the per-attempt grid is eight workgroups per CU there): clamp below that size only.
    // (single rank only: ranks that share a GPU - the test configuration of the cross-rank hand-off - would each ask for a full grid)
    const bool clamp_ok = mfma && single && (double)((desc->batch + 15) / 16) * (double)h->lin_dp * (double)h->lin_dp <= 8.0e6;
    if (clamp_ok && g > kPersistMaxGrid) g = kPersistMaxGrid;
    bool capable = desc->adaptive && (rowlocal || mfma || mlp || coop) && g <= kPersistMaxGrid;
    if (capable) {
      const int cap = mlp ? (h->is_f32 ? mi_persist_capacity_mlp_f32(h) : mi_persist_capacity_mlp_f64(h))
                          : (h->is_f32 ? mi_persist_capacity_f32(h) : mi_persist_capacity_f64(h));
      if (clamp_ok && cap > 0 && g > cap) g = cap;
      capable = cap > 0 && g <= cap;
    }
    if (!capable && desc->adaptive && (rowlocal || coop)) {
      // more trajectories than one-per-thread keeps co-resident: the same loop with the state in HBM planes, a co-resident
      // grid walking the batch (k_persist_rowlocal_planes)
      h->persist_planes = 1;
      h->persist_planes_block = 256;                 // (512 measured: +3 % at 1M Lorenz rows, -50 % at 4M)
      if (const char* eb = getenv("MI_ODE_PERSIST_PLANES_BLOCK")) { const int v = atoi(eb); if ((v == 256 || v == 512) && !coop) h->persist_planes_block = v; }
      const int cap = h->is_f32 ? mi_persist_capacity_f32(h) : mi_persist_capacity_f64(h);
      long long gp = cap < kPersistMaxGrid ? cap : kPersistMaxGrid;
      if (gp > h->num_cus && !coop) gp = h->num_cus;         // the hand-off is all-to-all: one (large) workgroup per CU; the cooperative
                                                             // right-hand sides (barriers inside every evaluation) want every co-resident workgroup
      if (const char* eg = getenv("MI_ODE_PERSIST_PLANES_GRID")) { const long long v = atoll(eg); if (v >= 1 && v <= gp) gp = v; }
      const long long gneed = coop ? multistep_grid(h) : (desc->batch + h->persist_planes_block - 1) / h->persist_planes_block;   // (cooperative: tpw trajectories per workgroup)
      if (gp > gneed) gp = gneed;
      if (h->nseg > 1 && gp < h->nseg) gp = 0;               // tuple state: every component needs a workgroup of its own
      if (cap > 0 && gp >= 1) { capable = true; g = gp; }
      else h->persist_planes = 0;
      if (h->persist_planes && h->nseg > 1) {
        // tuple state on the plane-streaming kernel: the grid's workgroups are dealt to the components in proportion to their rows
        // (at least one each); a workgroup walks the rows of ITS component only, so the per-component fold of the hand-off is the
        // register-resident kernel's (seg_blk = first workgroup of a component)
        long long total = 0;
        for (int k = 0; k < h->nseg; ++k) total += desc->seg_rows[k];
        int used = 0;
        for (int k = 0; k < h->nseg; ++k) {
          long long share = (long long)((double)g * (double)desc->seg_rows[k] / (double)total);
          const long long need = (desc->seg_rows[k] + h->persist_planes_block - 1) / h->persist_planes_block;
          if (share > need) share = need;
          if (share < 1) share = 1;
          const int left_for_rest = h->nseg - 1 - k;
          if (used + share > g - left_for_rest) share = g - left_for_rest - used;
          h->seg_blk[k] = used;
          used += (int)share;
        }
        h->seg_blk[h->nseg] = used;
        g = used;
      }
    }
    h->persist_capable = capable ? 1 : 0;          // (a peer mailbox connected later can still switch the one-launch schedule on)
    const bool can = capable && (single || h->xrank_dev != nullptr);
    if (desc->fusion == 4 && !can && single) { mi_set_error("fusion=4: no whole-integration kernel for this problem (row-local or MFMA-linear RHS, single rank, every workgroup co-resident)"); mi_ode_destroy(h); return MI_ODE_E_INVALID; }
    h->persist = (can && (desc->fusion == 4 || desc->fusion == 0)) ? 1 : 0;
    if (coop && desc->adaptive && !h->persist) {
      mi_set_error("cooperative kernel: %lld workgroups cannot be co-resident on this device (the whole-call kernel is its only schedule)", g);
      mi_ode_destroy(h); return MI_ODE_E_INVALID;
    }
    if (h->nseg > 1 && !h->persist) {
      mi_set_error("tuple states run on the whole-call kernel only, and its grid (%lld workgroups) is not co-resident on this device", g);
      mi_ode_destroy(h); return MI_ODE_E_INVALID;
    }
    h->persist_grid = (int)g;
    h->persist_sleep_first = g <= 32 ? 16 : 32;
    h->persist_sleep_poll = 2;
    if (const char* e0 = getenv("MI_ODE_PERSIST_SLEEP0")) h->persist_sleep_first = atoi(e0);      // tuning sweeps (scripts/gpu_persist_sweep.sh)
    if (const char* e1 = getenv("MI_ODE_PERSIST_SLEEP1")) h->persist_sleep_poll = atoi(e1);
    h->persist_spin_limit = 1 << 17;              // ~0.2 s of polling: skew between resident workgroups never gets near it
    h->persist_spin_first = 1 << 12;              // first hand-off of a launch = residency check: ~5 ms (a poll round is ~1 us)
    if (const char* e3 = getenv("MI_ODE_PERSIST_SPIN_FIRST")) h->persist_spin_first = atoi(e3);
    if (const char* e2 = getenv("MI_ODE_PERSIST_SPIN_LIMIT")) h->persist_spin_limit = atoi(e2);  // tests: force the time-out path
    h->persist_xspin_limit = 1 << 22;             // cross-rank part: ~10 s (a poll round is ~2.5 us) - launch skew between the ranks of a job is
                                                  // normal (measured tolerance test: tests/test_gpu_parity.py, 50 ms per rank); a dead peer still ends the wait
    if (const char* e4 = getenv("MI_ODE_PERSIST_XSPIN_LIMIT")) h->persist_xspin_limit = atoi(e4);
    if (getenv("MI_ODE_PERSIST_SPIN_LIMIT") && !getenv("MI_ODE_PERSIST_XSPIN_LIMIT")) h->persist_xspin_limit = h->persist_spin_limit;
  }
  // controller / dense-output parameters
  h->cp.rtol = desc->rtol; h->cp.atol = desc->atol;
  h->cp.safety = desc->safety; h->cp.ifactor = desc->ifactor; h->cp.dfactor = desc->dfactor;
  h->cp.inv_ifactor = 1.0 / desc->ifactor; h->cp.inv_dfactor = 1.0 / desc->dfactor;   // same IEEE quotients as on the device
  h->cp.max_num_steps = desc->max_num_steps > 0 ? desc->max_num_steps : 2147483647LL;
  h->cp.n_local = h->n;
  h->cp.order = desc->order; h->cp.init_order = desc->init_order;
  h->cp.controller = desc->controller;
  h->cp.is_f32 = h->is_f32;
  h->cp.n_stages = h->S;
  h->cp.auto_first_step = isnan(desc->first_step) ? 1 : 0;
  h->ip.kind = desc->interp;
  h->ip.nk = h->S + 1;
  for (int j = 0; j < kMaxK; ++j) h->ip.c_mid[j] = tb.c_mid[j];
  // workspace
  hipError_t e = hipSuccess;
  e = hipMalloc((void**)&h->planes, (size_t)h->stride * (2 + h->S + 1));     // y_a, y_b, k_0..k_S
  if (e == hipSuccess) e = hipMalloc((void**)&h->partials, (size_t)kMaxBlocks * kRec * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->partials, 0, (size_t)kMaxBlocks * kRec * sizeof(double));   // no stale sequence numbers
  if (desc->exchange_send_dev != nullptr && desc->exchange_recv_dev != nullptr) {
    h->rank_rec = desc->exchange_send_dev;
    h->gathered = desc->exchange_recv_dev;
    h->own_exchange = 0;
  } else {
    h->own_exchange = 1;
    if (e == hipSuccess) e = hipMalloc((void**)&h->rank_rec, kRec * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&h->gathered, (size_t)h->d.world_size * kRec * sizeof(double));
  }
  if (e == hipSuccess) e = hipMalloc((void**)&h->ctl, sizeof(Ctl));
  if (e == hipSuccess) e = hipMalloc((void**)&h->gbuf, 2 * kPRec * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->gbuf, 0, 2 * kPRec * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&h->ticket, 64);
  if (e == hipSuccess) e = hipMemset(h->ticket, 0, 64);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->ctl_host, sizeof(Ctl), hipHostMallocDefault);
  if (e != hipSuccess) {
    mi_set_error("workspace allocation failed: %s", hipGetErrorString(e));
    mi_ode_destroy(h);
    return MI_ODE_E_HIP;
  }
  memset(h->ctl_host, 0, sizeof(Ctl));
  if (h->d.profile) {
    for (int i = 0; i < 64 && e == hipSuccess; ++i) {
      e = hipEventCreate(&h->ev_a[i]);
      if (e == hipSuccess) e = hipEventCreate(&h->ev_b[i]);
      if (e == hipSuccess) e = hipEventCreate(&h->ev_c[i]);
    }
    if (e != hipSuccess) { mi_set_error("event creation failed"); mi_ode_destroy(h); return MI_ODE_E_HIP; }
    h->ev_ready = 1;
  }
  *out = h;
  return 0;
}

static int ensure_t_out(mi_ode_solver* h, int n) {
  if (n > h->t_out_cap) {
    if (h->t_out_dev) (void)hipFree(h->t_out_dev);
    h->t_out_dev = nullptr;
    int cap = n < 64 ? 64 : n;
    MI_HIP(hipMalloc((void**)&h->t_out_dev, (size_t)cap * sizeof(double)));
    h->t_out_cap = cap;
  }
  if (n > h->t_out_host_cap) {
    if (h->t_out_host) (void)hipHostFree(h->t_out_host);
    h->t_out_host = nullptr;
    int cap = n < 64 ? 64 : n;
    MI_HIP(hipHostMalloc((void**)&h->t_out_host, (size_t)cap * sizeof(double), hipHostMallocDefault));
    h->t_out_host_cap = cap;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// adaptive engine
// ------------------------------------------------------------------------------------------------
static int begin_impl(mi_ode_handle h, const void* y0_dev, double t0, void* first_out_dev, void* stream);

extern "C" int mi_ode_begin(mi_ode_handle h, const void* y0_dev, double t0, void* stream) {
  return begin_impl(h, y0_dev, t0, nullptr, stream);
}

// before_integrate: the F0 kernel reads y0 straight from the caller's buffer, evaluates f0 and the initial-step
// norms, and in the same pass seeds the workspace state plane (and solution[0] when first_out_dev is given).
static int begin_impl(mi_ode_handle h, const void* y0_dev, double t0, void* first_out_dev, void* stream) {
  if (h == nullptr || y0_dev == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (!h->d.adaptive) { mi_set_error("mi_ode_begin on a fixed-grid handle"); return MI_ODE_E_INVALID; }
  if (h->d.multistep != 0) { mi_set_error("multistep handles run through mi_ode_integrate / mi_ode_fixed_grid_integrate only"); return MI_ODE_E_INVALID; }
  if (h->nseg > 1) { mi_set_error("tuple states: mi_ode_integrate with at least two times only (whole-call kernel)"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  MI_HIP(hipStreamSynchronize(st));            // the pinned staging record may still be in flight from a previous call
  Ctl* c = h->ctl_host;
  memset(c, 0, sizeof(Ctl));
  c->t0 = c->t1 = t0;
  c->dt = h->cp.auto_first_step ? 0.0 : h->d.first_step;
  c->idx_y0 = 0; c->idx_y1 = 1;
  for (int j = 0; j < kMaxK; ++j) c->idx_k[j] = 2 + j;
  h->n_launches = 0; h->n_polls = 0;
  h->enq_attempts = 0; h->prof_done = 0;
  MI_HIP(hipMemcpyAsync(h->ctl, c, sizeof(Ctl), hipMemcpyHostToDevice, st));
  if (h->family == FAM_MLP) {
    MlpArgs M;
    fill_mlp_args(h, M);
    M.x_y0 = y0_dev; M.copy_a = h->planes; M.copy_b = first_out_dev;
    int rcm = h->is_f32 ? mi_launch_mlp_f32(h, MLP_F0, M, st) : mi_launch_mlp_f64(h, MLP_F0, M, st);
    if (rcm != 0) return rcm;
    rcm = enqueue_controller(h, PH_F0, st);
    if (rcm != 0) return rcm;
    if (h->cp.auto_first_step) {
      fill_mlp_args(h, M);
      rcm = h->is_f32 ? mi_launch_mlp_f32(h, MLP_INITB, M, st) : mi_launch_mlp_f64(h, MLP_INITB, M, st);
      if (rcm != 0) return rcm;
      rcm = enqueue_controller(h, PH_INITB, st);
      if (rcm != 0) return rcm;
    }
    h->begun = 1;
    return 0;
  }
  if (h->init_tiles16) {                         // linear MFMA family: the 16-row tile passes (mi_ode_step_fused.h)
    InitArgs I;
    memset(&I, 0, sizeof(I));
    fill_step_args(h, I.s);
    I.y0 = y0_dev; I.copy_b = first_out_dev;
    int rci = h->is_f32 ? mi_launch_init_linear_f32(h, 0, I, st) : mi_launch_init_linear_f64(h, 0, I, st);
    if (rci != 0) return rci;
    rci = enqueue_controller(h, PH_F0, st);
    if (rci != 0) return rci;
    if (h->cp.auto_first_step) {
      rci = h->is_f32 ? mi_launch_init_linear_f32(h, 1, I, st) : mi_launch_init_linear_f64(h, 1, I, st);
      if (rci != 0) return rci;
      rci = enqueue_controller(h, PH_INITB, st);
      if (rci != 0) return rci;
    }
    h->begun = 1;
    return 0;
  }
  // f0 = f(t0, y0) with the norms of misc._select_initial_step riding along (dopri5.py:71-75)
  StageArgs A;
  fill_common(h, A);
  A.explicit_mode = 1; A.ctl = nullptr;
  A.x_y0 = y0_dev; A.x_t0 = t0; A.x_dt = 0.0;
  A.x_kout = h->planes + 2 * h->stride;          // idx_k[0]
  A.copy_a = h->planes;                          // idx_y0
  A.copy_b = first_out_dev;
  int rc = launch_stage(h, M_F0, 0, A, st);
  if (rc != 0) return rc;
  rc = enqueue_controller(h, PH_F0, st);
  if (rc != 0) return rc;
  if (h->cp.auto_first_step) {
    fill_common(h, A);
    A.k_out_slot = 1;                          // f(t0 + h0, y0 + h0 f0) lands in a scratch plane
    A.alpha = 1.0;
    rc = launch_stage(h, M_INITB, 1, A, st);
    if (rc != 0) return rc;
    rc = enqueue_controller(h, PH_INITB, st);
    if (rc != 0) return rc;
  }
  h->begun = 1;
  return 0;
}

extern "C" int mi_ode_advance(mi_ode_handle h, const double* t_out_host, int32_t n_out, void* out_dev, void* stream) {
  if (h == nullptr || (n_out > 0 && (t_out_host == nullptr || out_dev == nullptr))) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (h->d.multistep != 0) { mi_set_error("multistep handles run through mi_ode_integrate / mi_ode_fixed_grid_integrate only"); return MI_ODE_E_INVALID; }
  if (!h->begun) { mi_set_error("mi_ode_advance before mi_ode_begin"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (n_out <= 0) return 0;
  for (int i = 1; i < n_out; ++i)
    if (!(t_out_host[i] > t_out_host[i - 1])) { mi_set_error("output times must increase"); return MI_ODE_ST_BAD_T; }
  int rc = ensure_t_out(h, n_out);
  if (rc != 0) return rc;
  // every advance() ends with a poll (stream idle), so the pinned staging buffer is free unless a fixed-grid call
  // left a copy in flight - no need to wait for the init kernels of mi_ode_begin here
  if (h->t_out_busy) { MI_HIP(hipStreamSynchronize(st)); h->t_out_busy = 0; }
  memcpy(h->t_out_host, t_out_host, (size_t)n_out * sizeof(double));
  MI_HIP(hipMemcpyAsync(h->t_out_dev, h->t_out_host, (size_t)n_out * sizeof(double), hipMemcpyHostToDevice, st));
  h->cp.t_out = h->t_out_dev;
  h->cur_out = out_dev;
  MI_HIP(hipMemsetAsync(h->ticket, 0, 4, st));
  hipLaunchKernelGGL(k_set_outputs, dim3(1), dim3(64), 0, st, h->ctl, (int)n_out);
  h->n_launches += 1;
  const double t_end = t_out_host[n_out - 1];
  // first chunk: what the previous call on this handle needed (ODEBlock-style callers repeat the same problem), else 4
  int chunk = h->d.chunk_attempts > 0 ? h->d.chunk_attempts
                                      : (h->last_call_attempts > 0 && h->last_call_attempts <= 64 ? (int)h->last_call_attempts : 4);
  const long long attempts_before = h->ctl_host->n_attempt;
  for (;;) {
    if (chunk > 64) chunk = 64;
    for (int a = 0; a < chunk; ++a) {
      const int ei = (int)(h->enq_attempts % 64);
      const bool prof = h->d.profile && h->ev_ready;
      if (prof) (void)hipEventRecord(h->ev_a[ei], st);
      rc = enqueue_attempt_kernels(h, st, prof ? h->ev_b[ei] : nullptr);
      if (rc != 0) return rc;
      if (prof) (void)hipEventRecord(h->ev_c[ei], st);
      h->enq_attempts += 1;
      if (!h->fused_ctl) {                       // otherwise the whole-attempt kernel's last workgroup ran it
        rc = enqueue_controller(h, PH_ATTEMPT, st);
        if (rc != 0) return rc;
      }
      rc = enqueue_emit(h, out_dev, st);
      if (rc != 0) return rc;
    }
    rc = poll_ctl(h, st);
    if (rc != 0) return rc;
    harvest_profile(h);
    const Ctl* c = h->ctl_host;
    if (c->done) break;
    if (h->d.chunk_attempts <= 0) {            // adaptive chunking: roughly the attempts still needed at the current dt
      double est = c->dt > 0 ? ceil((t_end - c->t1) / c->dt) : 4.0;
      if (!(est >= 1.0)) est = 1.0;
      if (est > 64.0) est = 64.0;
      chunk = (int)est;
    }
  }
  h->last_call_attempts = h->ctl_host->n_attempt - attempts_before;
  return (int)h->ctl_host->status;
}

// One launch for the whole call (mi_ode_persist.h).  The kernel builds its own scalar state from its arguments, takes
// up to kPersistTSmall output times as arguments too, and stores the final state straight into the pinned host
// record: in the common case the host side of a call is one launch and one stream synchronisation.
static int integrate_persist(mi_ode_solver* h, const void* y0_dev, const double* t_host, int32_t T, void* out_dev,
                             mi_ode_stats* stats, hipStream_t st) {
  const int n_out = T - 1;
  int rc = 0;
  if (n_out > kPersistTSmall) {
    MI_HIP(hipStreamSynchronize(st));          // the pinned staging buffer may still be in flight from a previous call
    rc = ensure_t_out(h, n_out);
    if (rc != 0) return rc;
    memcpy(h->t_out_host, t_host + 1, (size_t)n_out * sizeof(double));
    MI_HIP(hipMemcpyAsync(h->t_out_dev, h->t_out_host, (size_t)n_out * sizeof(double), hipMemcpyHostToDevice, st));
  }
  h->n_launches = 0; h->n_polls = 0; h->enq_attempts = 0; h->prof_done = 0;
  h->cp.t_out = h->t_out_dev;
  h->cur_out = (char*)out_dev + (size_t)h->n * h->elt;
  PersistArgs A;
  memset(&A, 0, sizeof(A));
  fill_step_args(h, A.s);
  if (h->family == FAM_MLP_COOP) multistep_rhs(h, A.s.rhs);   // (RhsMlpCoop reads dim from an aux field)
  A.s.ticket = nullptr;
  A.y0 = y0_dev; A.out0 = out_dev; A.n_out = n_out;
  A.ctl_host = h->ctl_host;
  A.t0 = t_host[0];
  A.first_dt = h->cp.auto_first_step ? 0.0 : h->d.first_step;
  for (int i = 0; i < n_out && i < kPersistTSmall; ++i) A.t_small[i] = t_host[1 + i];
  A.seq_base = h->seq;
  if (h->xrank_on) {
    A.xrank = h->xrank_dev; A.xpeers = h->xpeer_tab_dev; A.gbuf = h->gbuf; A.world = (int)h->d.world_size; A.rank = (int)h->d.rank;
  } else { A.world = 1; }
  A.nseg = h->nseg;
  for (int k = 0; k < h->nseg; ++k) { A.seg_blk[k] = h->seg_blk[k]; A.seg_rows[k] = h->d.seg_rows[k]; }
  if (h->nseg > 1) A.seg_blk[h->nseg] = h->seg_blk[h->nseg];
  A.seg_tol = h->nseg > 1 && h->d.seg_tolerances ? 1 : 0;
  for (int k = 0; k < h->nseg; ++k) { A.seg_rtol[k] = h->d.seg_rtol[k]; A.seg_atol[k] = h->d.seg_atol[k]; }
  A.spin_limit = h->persist_spin_limit;
  A.xspin_limit = h->persist_xspin_limit;
  A.spin_first = h->persist_spin_first < h->persist_spin_limit ? h->persist_spin_first : h->persist_spin_limit;
  // back-off before the first poll (units of 64 clocks): a failed poll round costs G x G record loads on the fabric, so
  // wait about as long as the publish needs to become visible (measured: G=16 best at <= 24, G=256 best at 32)
  A.sleep_first = h->persist_sleep_first; A.sleep_poll = h->persist_sleep_poll;
  const bool prof = h->d.profile && h->ev_ready;
  if (prof) (void)hipEventRecord(h->ev_a[0], st);
  if (h->family == FAM_MLP) rc = h->is_f32 ? mi_launch_persist_mlp_f32(h, A, h->persist_grid, st) : mi_launch_persist_mlp_f64(h, A, h->persist_grid, st);
  else rc = h->is_f32 ? mi_launch_persist_f32(h, A, h->persist_grid, st) : mi_launch_persist_f64(h, A, h->persist_grid, st);
  if (rc != 0) return rc;
  if (prof) (void)hipEventRecord(h->ev_c[0], st);
  MI_HIP(hipStreamSynchronize(st));            // the kernel's last act was the zero-copy store of the final state
  h->n_polls += 1;
  if (prof) {                                    // one launch = the whole call: both profile slots hold its duration
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev_a[0], h->ev_c[0]) == hipSuccess) { h->prof_last_ms += ms; h->prof_all_ms += ms; h->prof_n += 1; }
  }
  h->begun = 1;
  h->last_call_attempts = h->ctl_host->n_attempt;
#ifdef MI_PERSIST_PROF
  {
    const Ctl* cc = h->ctl_host;
    const double na = cc->n_attempt > 0 ? (double)cc->n_attempt : 1.0;
    if (h->family == FAM_LINEAR_MFMA) {
      // per hand-off: spread of the workgroups' arrival times (skew) and how long after the LAST arrival the last workgroup left (latency)
      static long long stamps[8 * kMaxBlocks * 2];
      const int G = h->persist_grid;
      if (G <= kMaxBlocks && hipMemcpy(stamps, h->partials + 8192, sizeof(long long) * 8 * G * 2, hipMemcpyDeviceToHost) == hipSuccess) {
        for (int g = 0; g < 8 && g < (int)cc->n_attempt + 2; ++g) {
          long long a_min = (1LL << 62), a_max = 0, l_min = (1LL << 62), l_max = 0; double a_sum = 0;
          for (int b = 0; b < G; ++b) {
            const long long a = stamps[(g * G + b) * 2], l = stamps[(g * G + b) * 2 + 1];
            if (a < a_min) a_min = a; if (a > a_max) a_max = a; if (l < l_min) l_min = l; if (l > l_max) l_max = l; a_sum += (double)a;
          }
          fprintf(stderr, "[persist skew] hand-off %d: arrivals spread %.2f us (mean arrives %.2f us before the last), first leaves %.2f us / last leaves %.2f us after the last arrival\n",
                  g, 0.01 * (a_max - a_min), 0.01 * ((double)a_max - a_sum / G), 0.01 * (l_min - a_max), 0.01 * (l_max - a_max));
        }
      }
    }
    if (h->family == FAM_LINEAR_MFMA)
      fprintf(stderr, "[persist prof] attempts %lld  us: f0 pass %.1f  initial-step pass %.1f  attempt passes %.1f (%.1f each)  hand-offs %.1f\n",
              cc->n_attempt, 0.01 * cc->prof[0], 0.01 * cc->prof[1], 0.01 * cc->prof[2], 0.01 * cc->prof[2] / na, 0.01 * cc->prof[3]);
    else
    fprintf(stderr, "[persist prof] attempts %lld  ns/attempt: stages %.0f  reduce+handoff %.0f  controller %.0f  emit %.0f\n",
            cc->n_attempt, 10.0 * cc->prof[0] / na, 10.0 * cc->prof[1] / na, 10.0 * cc->prof[2] / na, 10.0 * cc->prof[3] / na);
  }
#endif
  h->seq += (unsigned)h->ctl_host->n_attempt + 16u;            // hand-offs of this call: attempts + 2 (+ margin); identical on
  if (h->seq >= 0xE0000000u) h->seq = 0;                        // every rank.  Wraps below the self-test range; a slot is
                                                                // rewritten every other hand-off, so an old number never matches
  if (stats) fill_stats(h, stats);
  return (int)h->ctl_host->status;
}

// 'adams' (adams.py:66-211) for the row-local catalogue systems: the whole call in one launch (mi_ode_adams_vc.h)
static int integrate_adams_vc(mi_ode_solver* h, const void* y0_dev, const double* t_host, int32_t T, void* out_dev, mi_ode_stats* stats,
                              hipStream_t st) {
  h->n_launches = 0; h->n_polls = 0;
  int rc = ensure_t_out(h, T);
  if (rc != 0) return rc;
  MI_HIP(hipStreamSynchronize(st));              // the pinned staging buffer may still be in flight from a previous call
  memcpy(h->t_out_host, t_host, (size_t)T * sizeof(double));
  MI_HIP(hipMemcpyAsync(h->t_out_dev, h->t_out_host, (size_t)T * sizeof(double), hipMemcpyHostToDevice, st));
  h->t_out_busy = 1;
  AdamsVcArgs A;
  memset(&A, 0, sizeof(A));
  A.f.y0 = y0_dev; A.f.out = out_dev; A.f.t = h->t_out_dev; A.f.batch = h->d.batch; A.f.T = T; A.f.dim = (int)h->d.dim; A.f.rhs = h->rhs;
  A.cp = h->cp;
  A.cp.init_order = 2;                           // adams.py:115-118: the first step is the order-2 heuristic, whatever the solver's order
  A.cp.controller = MI_ODE_CTRL_MISC;
  A.max_order = h->d.ms_max_order;
  A.max_attempts = h->d.max_num_steps > 0 ? h->d.max_num_steps : (1LL << 31);
  memcpy(A.gamma_star, h->adams_gamma_star, sizeof(A.gamma_star));
  A.result = h->adams_res;
  A.p.s.partials = h->partials; A.p.seq_base = h->seq; A.p.nseg = 1; A.p.world = 1;
  A.p.spin_limit = h->persist_spin_limit > 0 ? h->persist_spin_limit : (1 << 17);
  A.p.spin_first = 1 << 12; if (A.p.spin_first > A.p.spin_limit) A.p.spin_first = A.p.spin_limit;
  A.p.xspin_limit = A.p.spin_limit;
  A.p.sleep_first = 16; A.p.sleep_poll = 2;
  multistep_rhs(h, A.f.rhs);
  const long long g = multistep_grid(h);
  rc = h->is_f32 ? mi_launch_adams_vc_f32(h, A, (int)g, st) : mi_launch_adams_vc_f64(h, A, (int)g, st);
  if (rc != 0) return rc;
  MI_HIP(hipStreamSynchronize(st));              // the kernel's last act: {attempts, accepted, nfe, status} into pinned memory
  h->seq += (unsigned)(3 * h->adams_res[0] + 16);  // up to three hand-offs per attempt + the two of before_integrate
  if (h->seq >= 0xE0000000u) h->seq = 0;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->n_attempts = h->adams_res[0]; stats->n_accepted = h->adams_res[1]; stats->n_rejected = h->adams_res[0] - h->adams_res[1];
    stats->nfe = h->adams_res[2];
    stats->status = (uint32_t)h->adams_res[3];
    stats->t = t_host[T - 1];
    stats->n_launches = h->n_launches;
    stats->n_polls = 1;
  }
  return (int)h->adams_res[3];
}

extern "C" int mi_ode_integrate(mi_ode_handle h, const void* y0_dev, const double* t_host, int32_t T, void* out_dev,
                                mi_ode_stats* stats, void* stream) {
  if (h == nullptr || y0_dev == nullptr || t_host == nullptr || out_dev == nullptr || T < 1) { mi_set_error("bad argument"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  for (int i = 1; i < T; ++i)
    if (!(t_host[i] > t_host[i - 1])) {
      if (stats) { memset(stats, 0, sizeof(*stats)); stats->status = MI_ODE_ST_BAD_T; }
      return MI_ODE_ST_BAD_T;                  // _assert_increasing (misc.py:158-159)
    }
  if (h->d.multistep == 3) return integrate_adams_vc(h, y0_dev, t_host, T, out_dev, stats, st);
  const bool multi = h->d.world_size > 1 || h->d.allgather != nullptr || h->nccl_comm != nullptr;
  if (h->persist && T > 1 && h->d.adaptive && (!multi || h->xrank_on)) {
    const int prc = integrate_persist(h, y0_dev, t_host, T, out_dev, stats, st);
    if (prc < 0 || !(prc & MI_ODE_ST_SYNC_TIMEOUT) || h->d.fusion == 4 || multi || h->nseg > 1 || h->family == FAM_MLP_COOP || h->family == FAM_PLUGIN_COOP) return prc;   // (a rank must not change schedule
                                                                                                            // alone; tuple states have no other)
    h->persist = 0;        // the grid hand-off timed out (co-residency lost to another persistent kernel?): this
  }                        // handle goes back to one launch per attempt, starting with this call
  if (h->family == FAM_MLP_COOP || h->family == FAM_PLUGIN_COOP) { mi_set_error("the cooperative kernels have the whole-call schedule only (T > 1)"); return MI_ODE_E_INVALID; }
  int rc = begin_impl(h, y0_dev, t_host[0], out_dev, stream);   // before_integrate runs even when T == 1 (solvers.py:31);
  if (rc != 0) return rc;                                        // solution = [y0] is written by the same kernel
  int status = 0;
  if (T > 1) {
    status = mi_ode_advance(h, t_host + 1, T - 1, (char*)out_dev + (size_t)h->n * h->elt, stream);
    if (status < 0) return status;
  } else {
    rc = poll_ctl(h, st);
    if (rc != 0) return rc;
  }
  if (stats) fill_stats(h, stats);
  return status;
}

extern "C" int64_t mi_ode_xrank_bytes(int32_t world_size) {
  return (int64_t)2 * (world_size < 1 ? 1 : world_size) * kXRec * (int64_t)sizeof(double);
}

extern "C" int mi_ode_xrank_selftest(mi_ode_handle h, void* stream) {
  if (h == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (h->xrank_dev == nullptr && h->xpeer_tab_dev == nullptr) { mi_set_error("no cross-rank mailbox or segment connected"); return 1; }
  hipStream_t st = (hipStream_t)stream;
  PersistArgs A;
  memset(&A, 0, sizeof(A));
  A.xrank = h->xrank_dev; A.xpeers = h->xpeer_tab_dev; A.gbuf = h->gbuf; A.world = (int)h->d.world_size; A.rank = (int)h->d.rank;
  A.spin_limit = A.spin_first = 1 << 20;        // several seconds: covers module-load skew between the ranks
  A.xspin_limit = 1 << 22;
  h->xrank_tests += 64u;
  A.seq_base = 0xF0000000u + (h->xrank_tests & 0x0FFFFFFu);   // disjoint from the numbers of real calls
  int* res_dev = (int*)h->ticket + 8;
  MI_HIP(hipMemsetAsync(res_dev, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_xrank_selftest<0>, dim3(1), dim3(64), 0, st, A, 4, res_dev);
  int res = 0;
  MI_HIP(hipMemcpyAsync(&res, res_dev, sizeof(int), hipMemcpyDeviceToHost, st));
  MI_HIP(hipStreamSynchronize(st));
  return res == 1 ? 0 : 1;
}

extern "C" int mi_ode_xrank_enable(mi_ode_handle h, int32_t on) {
  if (h == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (on && h->xrank_dev == nullptr && h->xpeer_tab_dev == nullptr) { mi_set_error("no cross-rank mailbox or segment connected"); return MI_ODE_E_INVALID; }
  h->xrank_on = on ? 1 : 0;
  if (on && h->persist_capable && (h->d.fusion == 0 || h->d.fusion == 4)) h->persist = 1;   // the sharded run keeps the one-launch schedule
  return 0;
}

// ---- peer-device-memory mailboxes (include/mi_ode.h) -----------------------------------------------------------------
extern "C" int mi_ode_xpeer_prepare(mi_ode_handle h, void* ipc_handle_out) {
  if (h == nullptr || ipc_handle_out == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  static_assert(sizeof(hipIpcMemHandle_t) <= MI_ODE_IPC_HANDLE_BYTES, "MI_ODE_IPC_HANDLE_BYTES too small");
  if (h->d.world_size > kXMaxWorld) { mi_set_error("peer mailboxes: world_size <= %d", kXMaxWorld); return MI_ODE_E_INVALID; }
  if (h->xpeer_local == nullptr) {
    const size_t bytes = (size_t)mi_ode_xrank_bytes(h->d.world_size);
    void* p = nullptr;
    // uncached (MTYPE_UC) device memory: what a peer stores over xGMI must be what the local poll reads, no L2 copy in between
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) { (void)hipGetLastError(); mi_set_error("peer mailbox allocation failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); mi_set_error("peer mailbox clear failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
    h->xpeer_local = p;
  }
  hipIpcMemHandle_t ih;
  hipError_t e = hipIpcGetMemHandle(&ih, h->xpeer_local);
  if (e != hipSuccess) { (void)hipGetLastError(); mi_set_error("hipIpcGetMemHandle failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
  memset(ipc_handle_out, 0, MI_ODE_IPC_HANDLE_BYTES);
  memcpy(ipc_handle_out, &ih, sizeof(ih));
  return 0;
}

extern "C" int mi_ode_xpeer_connect(mi_ode_handle h, const void* all_ipc_handles, int32_t world_size) {
  if (h == nullptr || all_ipc_handles == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (h->xpeer_local == nullptr) { mi_set_error("mi_ode_xpeer_connect before mi_ode_xpeer_prepare"); return MI_ODE_E_INVALID; }
  if (world_size != h->d.world_size || world_size < 1 || world_size > kXMaxWorld) { mi_set_error("peer mailboxes: bad world_size %d", world_size); return MI_ODE_E_INVALID; }
  close_peers(h);
  h->xpeer_world = world_size;
  const char* hs = (const char*)all_ipc_handles;
  for (int q = 0; q < world_size; ++q) {
    if (q == h->d.rank) { h->xpeer_open[q] = h->xpeer_local; continue; }
    hipIpcMemHandle_t ih;
    memcpy(&ih, hs + (size_t)q * MI_ODE_IPC_HANDLE_BYTES, sizeof(ih));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      mi_set_error("hipIpcOpenMemHandle(rank %d) failed: %s", q, hipGetErrorString(e));
      close_peers(h);
      return MI_ODE_E_HIP;
    }
    h->xpeer_open[q] = p;
  }
  hipError_t e = hipMalloc((void**)&h->xpeer_tab_dev, (size_t)kXMaxWorld * sizeof(double*));
  if (e == hipSuccess) e = hipMemcpy(h->xpeer_tab_dev, h->xpeer_open, (size_t)world_size * sizeof(double*), hipMemcpyHostToDevice);
  if (e != hipSuccess) { mi_set_error("peer table upload failed: %s", hipGetErrorString(e)); close_peers(h); return MI_ODE_E_HIP; }
  return 0;
}

extern "C" int mi_ode_rccl_unique_id(void* id_out) {
  if (id_out == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  int rc = load_rccl();
  if (rc != 0) return rc;
  NcclUniqueId id;
  memset(&id, 0, sizeof(id));
  const int nrc = g_nccl.get_id(&id);
  if (nrc != 0) { mi_set_error("ncclGetUniqueId failed: %s", g_nccl.err ? g_nccl.err(nrc) : "?"); return MI_ODE_E_EXCHANGE; }
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int mi_ode_rccl_connect(mi_ode_handle h, const void* id, int32_t world_size, int32_t rank) {
  if (h == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (id == nullptr) {                            // disconnect: back to the allgather hook (all ranks must take the same path)
    if (h->nccl_comm && g_nccl.ok) (void)g_nccl.destroy(h->nccl_comm);
    h->nccl_comm = nullptr;
    return 0;
  }
  if (world_size != h->d.world_size || rank != h->d.rank) { mi_set_error("rccl_connect: world_size / rank differ from the handle's"); return MI_ODE_E_INVALID; }
  int rc = load_rccl();
  if (rc != 0) return rc;
  if (h->nccl_comm) { (void)g_nccl.destroy(h->nccl_comm); h->nccl_comm = nullptr; }
  NcclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  void* comm = nullptr;
  const int nrc = g_nccl.init_rank(&comm, world_size, uid, rank);
  if (nrc != 0 || comm == nullptr) { mi_set_error("ncclCommInitRank failed: %s", g_nccl.err ? g_nccl.err(nrc) : "?"); return MI_ODE_E_EXCHANGE; }
  h->nccl_comm = comm;
  h->fused_ctl = 0;                               // the controller must see the gathered records: it runs as its own launch
  return 0;
}

extern "C" int mi_ode_get_stats(mi_ode_handle h, mi_ode_stats* stats, void* stream) {
  if (h == nullptr || stats == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  int rc = poll_ctl(h, (hipStream_t)stream);
  if (rc != 0) return rc;
  fill_stats(h, stats);
  return 0;
}

extern "C" int mi_ode_get_profile(mi_ode_handle h, double* out4) {
  if (h == nullptr || out4 == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  out4[0] = h->prof_last_ms; out4[1] = (double)h->prof_n; out4[2] = h->prof_all_ms; out4[3] = (double)h->prof_n;
  return 0;
}

extern "C" int mi_ode_get_state(mi_ode_handle h, void* y_dev, void* f_dev, void* stream) {
  if (h == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (h->d.multistep != 0) { mi_set_error("get_state: the one-launch multistep kernels keep their state in registers (the solution rows are the output)"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  int rc = poll_ctl(h, st);
  if (rc != 0) return rc;
  const Ctl* c = h->ctl_host;
  if (y_dev) MI_HIP(hipMemcpyAsync(y_dev, h->planes + (long long)c->idx_y0 * h->stride, (size_t)h->n * h->elt, hipMemcpyDeviceToDevice, st));
  if (f_dev) MI_HIP(hipMemcpyAsync(f_dev, h->planes + (long long)c->idx_k[0] * h->stride, (size_t)h->n * h->elt, hipMemcpyDeviceToDevice, st));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// fixed grid (solvers.py:82-104): everything is known on the host, nothing synchronises
// ------------------------------------------------------------------------------------------------
static int fixed_impl(mi_ode_handle h, const void* y0_dev, const double* grid_host, int32_t G, const double* t_host, int32_t T, double eps,
                      void* out_dev, mi_ode_stats* stats, void* stream);

extern "C" int mi_ode_fixed_grid_integrate(mi_ode_handle h, const void* y0_dev, const double* t_host, int32_t T,
                                           void* out_dev, mi_ode_stats* stats, void* stream) {
  return fixed_impl(h, y0_dev, t_host, T, t_host, T, 0.0, out_dev, stats, stream);
}

extern "C" int mi_ode_fixed_grid_integrate_on(mi_ode_handle h, const void* y0_dev, const double* grid_host, int32_t G,
                                              const double* t_host, int32_t T, double eps, void* out_dev, mi_ode_stats* stats,
                                              void* stream) {
  if (grid_host == nullptr || G < 1) { mi_set_error("bad grid"); return MI_ODE_E_INVALID; }
  return fixed_impl(h, y0_dev, grid_host, G, t_host, T, eps, out_dev, stats, stream);
}

static int fixed_impl(mi_ode_handle h, const void* y0_dev, const double* grid_host, int32_t G, const double* t_host, int32_t T, double eps,
                      void* out_dev, mi_ode_stats* stats, void* stream) {
  if (h == nullptr || y0_dev == nullptr || t_host == nullptr || out_dev == nullptr || T < 1) { mi_set_error("bad argument"); return MI_ODE_E_INVALID; }
  if (h->d.adaptive) { mi_set_error("fixed-grid call on an adaptive handle"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const bool euler = h->S == 0;
  if (!euler && h->S != 3 && h->d.multistep == 0) { mi_set_error("fixed grid: tableau must be euler (0 rows) or rk4 3/8 (3 rows)"); return MI_ODE_E_INVALID; }
  for (int i = 1; i < T; ++i)
    if (!(t_host[i] > t_host[i - 1])) {
      if (stats) { memset(stats, 0, sizeof(*stats)); stats->status = MI_ODE_ST_BAD_T; }
      return MI_ODE_ST_BAD_T;
    }
  const bool own_grid = grid_host != t_host || eps != 0.0;
  if (own_grid) {
    for (int i = 1; i < G; ++i)
      if (!(grid_host[i] > grid_host[i - 1])) { mi_set_error("the time grid must increase"); return MI_ODE_ST_BAD_T; }
    if (!(grid_host[0] == t_host[0] && grid_host[G - 1] == t_host[T - 1])) {               // solvers.py:87
      mi_set_error("the time grid must start at t[0] and end at t[-1]");
      return MI_ODE_ST_BAD_T;
    }
  }
  h->n_launches = 0; h->n_polls = 0;
  const size_t pbytes = (size_t)h->n * h->elt;
  if (h->d.multistep != 0 && (h->d.fusion == 1 || !multistep_family(h))) {     // (never fall through to the Runge-Kutta loop below)
    mi_set_error("multistep handles run on the one-launch kernels only");
    return MI_ODE_E_INVALID;
  }
  if ((h->family == FAM_CUBIC2 || h->family == FAM_LINEAR2 || h->family == FAM_LV || h->family == FAM_LORENZ ||
       h->family == FAM_LINEAR_MFMA || h->family == FAM_MLP || h->family == FAM_PLUGIN ||
       h->family == FAM_MLP_COOP || h->family == FAM_PLUGIN_COOP || (h->family == FAM_LINEAR_VALU && h->d.multistep != 0)) && h->d.fusion != 1) {
    // trajectories never interact on a fixed grid: the whole integration is ONE launch
    // (k_fixed_rowlocal for the tiny row-local systems, k_fixed_linear_mfma for the linear RHS)
    int rcf = ensure_t_out(h, T + (own_grid ? G : 0));
    if (rcf != 0) return rcf;
    MI_HIP(hipStreamSynchronize(st));
    memcpy(h->t_out_host, t_host, (size_t)T * sizeof(double));
    if (own_grid) memcpy(h->t_out_host + T, grid_host, (size_t)G * sizeof(double));
    MI_HIP(hipMemcpyAsync(h->t_out_dev, h->t_out_host, (size_t)(T + (own_grid ? G : 0)) * sizeof(double), hipMemcpyHostToDevice, st));
    h->t_out_busy = 1;
    FixedArgs F;
    memset(&F, 0, sizeof(F));
    F.grid = own_grid ? h->t_out_dev + T : h->t_out_dev; F.M = (own_grid ? G : T) - 1; F.eps = eps;
    F.y0 = y0_dev; F.out = out_dev; F.t = h->t_out_dev; F.batch = h->d.batch; F.T = T; F.rk4 = euler ? 0 : 1; F.dim = (int)h->d.dim; F.rhs = h->rhs;
    // the stream was synchronised above: the PREVIOUS fixed-grid launch of this handle has stored its clock probe (this call's own
    // launch is not waited for - solvers.py:82-104 has no reason to block)
    const double prev_mhz = h->ctl_host->clk_ticks > 0 ? 100.0 * (double)h->ctl_host->clk_cycles / (double)h->ctl_host->clk_ticks : 0.0;
    F.clk = &h->ctl_host->clk_cycles;
    if (h->d.multistep != 0) {                     // Adams-Bashforth(-Moulton): one launch, the history in registers (mi_ode_adams.h)
      AdamsArgs AA;
      memset(&AA, 0, sizeof(AA));
      AA.f = F;
      AA.implicit = h->d.multistep == 2 ? 1 : 0; AA.max_iters = h->d.ms_max_iters; AA.max_order = h->d.ms_max_order; AA.min_order = h->d.ms_min_order;
      AA.rtol = h->d.rtol; AA.atol = h->d.atol; AA.tab = h->adams_tab; AA.result = h->adams_res;
      AA.p.s.partials = h->partials; AA.p.seq_base = h->seq; AA.p.nseg = 1; AA.p.world = 1;
      AA.p.spin_limit = h->persist_spin_limit > 0 ? h->persist_spin_limit : (1 << 17);
      AA.p.spin_first = 1 << 12; if (AA.p.spin_first > AA.p.spin_limit) AA.p.spin_first = AA.p.spin_limit;
      AA.p.xspin_limit = AA.p.spin_limit;
      AA.p.sleep_first = 16; AA.p.sleep_poll = 2;
      multistep_rhs(h, AA.f.rhs);
      const long long g = multistep_grid(h);
      rcf = h->is_f32 ? mi_launch_adams_f32(h, AA, (int)g, st) : mi_launch_adams_f64(h, AA, (int)g, st);
      if (rcf != 0) return rcf;
      MI_HIP(hipStreamSynchronize(st));            // the kernel's last act: {steps without convergence, status} into pinned memory
      h->seq += (unsigned)((long long)F.M * AA.max_iters + 16);
      if (h->seq >= 0xE0000000u) h->seq = 0;
      const long long steps = (own_grid ? G : T) - 1;
      if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->n_attempts = stats->n_accepted = steps;
        stats->n_rejected = h->adams_res[0];
        stats->status = (uint32_t)h->adams_res[1];
        stats->t = t_host[T - 1];
        stats->n_launches = h->n_launches;
        stats->n_polls = 1;
      }
      return (int)h->adams_res[1];
    }
    if (h->family == FAM_MLP_COOP) multistep_rhs(h, F.rhs);                 // (RhsMlpCoop reads dim from an aux field)
    if (h->family == FAM_PLUGIN || h->family == FAM_PLUGIN_COOP) {
      rcf = h->plugin->launch_fixed(h, &F, st);
      if (rcf != 0) { mi_set_error("plugin fixed-grid kernel launch failed"); return rcf; }
      h->n_launches += 1;
    } else if (h->family == FAM_MLP) {
      rcf = h->is_f32 ? mi_launch_fixed_mlp_f32(h, F, st) : mi_launch_fixed_mlp_f64(h, F, st);
    } else {
      rcf = h->is_f32 ? mi_launch_fixed_f32(h, F, st) : mi_launch_fixed_f64(h, F, st);
    }
    if (rcf != 0) return rcf;
    if (stats) {
      memset(stats, 0, sizeof(*stats));
      stats->nfe = (long long)(euler ? 1 : 4) * ((own_grid ? G : T) - 1);
      stats->n_attempts = stats->n_accepted = (own_grid ? G : T) - 1;
      stats->t = t_host[T - 1];
      stats->n_launches = h->n_launches;
      stats->clock_mhz = prev_mhz;               // (of the previous call on this handle; 0 on the first)
    }
    return 0;
  }
  if (own_grid) { mi_set_error("a custom time grid / eps needs the one-launch fixed-grid kernels (row-local or MFMA-linear RHS, fusion != 1)"); return MI_ODE_E_INVALID; }
  MI_HIP(hipMemcpyAsync(out_dev, y0_dev, pbytes, hipMemcpyDeviceToDevice, st));
  char* k0 = h->planes + 2 * h->stride;
  char* k1 = k0 + h->stride;
  char* k2 = k1 + h->stride;
  long long nfe = 0;
  for (int i = 0; i + 1 < T; ++i) {
    double t0, dt;                               // solvers.py:84: time is cast to the STATE dtype first
    if (h->is_f32) { const float a = (float)t_host[i], b = (float)t_host[i + 1]; t0 = a; dt = (double)(b - a); }
    else { t0 = t_host[i]; dt = t_host[i + 1] - t_host[i]; }
    const char* yi = (const char*)out_dev + (size_t)i * pbytes;
    char* yo = (char*)out_dev + (size_t)(i + 1) * pbytes;
    StageArgs A;
    fill_common(h, A);
    A.explicit_mode = 1; A.ctl = nullptr;
    A.x_y0 = yi; A.x_y1 = yo; A.x_t0 = t0; A.x_dt = dt;
    int rc;
    if (euler) {
      A.x_kout = nullptr;
      rc = launch_stage(h, M_FX_EULER, 0, A, st);
      if (rc != 0) return rc;
      nfe += 1;
    } else {
      A.x_kout = k0; A.alpha = 0.0;
      rc = launch_stage(h, M_STAGE, 0, A, st);                    // k1 = f(t, y)
      if (rc != 0) return rc;
      A.x_k[0] = k0; A.x_kout = k1; A.alpha = 1.0 / 3.0;
      rc = launch_stage(h, M_FX_RK4_2, 1, A, st);
      if (rc != 0) return rc;
      A.x_k[1] = k1; A.x_kout = k2; A.alpha = 2.0 / 3.0;
      rc = launch_stage(h, M_FX_RK4_3, 2, A, st);
      if (rc != 0) return rc;
      A.x_k[2] = k2; A.x_kout = nullptr; A.alpha = 1.0;
      rc = launch_stage(h, M_FX_RK4_4, 3, A, st);
      if (rc != 0) return rc;
      nfe += 4;
    }
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->nfe = nfe;
    stats->n_attempts = stats->n_accepted = T - 1;
    stats->t = t_host[T - 1];
    stats->n_launches = h->n_launches;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// single-attempt parity surface (rk_common._runge_kutta_step)
// ------------------------------------------------------------------------------------------------
extern "C" int mi_ode_eval_rhs(mi_ode_handle h, const void* y_dev, double t, void* f_dev, void* stream) {
  if (h == nullptr || y_dev == nullptr || f_dev == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (h->family == FAM_MLP) { mi_set_error("eval_rhs: not available for the MLP family"); return MI_ODE_E_INVALID; }
  StageArgs A;
  fill_common(h, A);
  A.explicit_mode = 1; A.ctl = nullptr;
  A.x_y0 = y_dev; A.x_kout = f_dev; A.x_t0 = t; A.x_dt = 0.0;
  return launch_stage(h, M_STAGE, 0, A, (hipStream_t)stream);
}

extern "C" int mi_ode_rk_step_fused(mi_ode_handle h, const void* y0_dev, const void* f0_dev, double t0, double dt,
                                    void* y1_dev, void* f1_dev, double* err_norms_host, void* k_out_dev, void* stream) {
  if (h == nullptr || y0_dev == nullptr || f0_dev == nullptr) { mi_set_error("null argument"); return MI_ODE_E_INVALID; }
  if (!h->d.adaptive) { mi_set_error("rk_step_fused needs an adaptive handle"); return MI_ODE_E_INVALID; }
  if (h->d.multistep != 0) { mi_set_error("rk_step_fused: a multistep handle has no Runge-Kutta tableau"); return MI_ODE_E_INVALID; }
  if (h->family == FAM_MLP) { mi_set_error("rk_step_fused: not available for the MLP family"); return MI_ODE_E_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const mi_ode_tableau& tb = h->d.tableau;
  const size_t pbytes = (size_t)h->n * h->elt;
  char* kpl[kMaxK];
  for (int j = 0; j < kMaxK; ++j) kpl[j] = h->planes + (long long)(2 + j) * h->stride;
  char* y1w = y1_dev ? (char*)y1_dev : h->planes + h->stride;
  for (int sigma = 1; sigma <= h->S; ++sigma) {
    StageArgs A;
    fill_common(h, A);
    A.explicit_mode = 1; A.ctl = nullptr;
    A.x_y0 = y0_dev; A.x_y1 = y1w; A.x_t0 = t0; A.x_dt = dt;
    A.x_k[0] = f0_dev;
    for (int j = 1; j < sigma; ++j) A.x_k[j] = kpl[j];
    A.x_kout = kpl[sigma];
    for (int j = 0; j < sigma; ++j) A.a[j] = tb.beta[sigma - 1][j];
    A.alpha = tb.alpha[sigma - 1];
    int mode = M_STAGE;
    if (sigma == h->S) {
      mode = M_LAST_FSAL;
      for (int j = 0; j <= sigma; ++j) A.e[j] = tb.c_error[j];
    }
    int rc = launch_stage(h, mode, sigma, A, st);
    if (rc != 0) return rc;
  }
  if (f1_dev) MI_HIP(hipMemcpyAsync(f1_dev, kpl[h->S], pbytes, hipMemcpyDeviceToDevice, st));
  if (k_out_dev) {
    MI_HIP(hipMemcpyAsync(k_out_dev, f0_dev, pbytes, hipMemcpyDeviceToDevice, st));
    for (int j = 1; j <= h->S; ++j)
      MI_HIP(hipMemcpyAsync((char*)k_out_dev + (size_t)j * pbytes, kpl[j], pbytes, hipMemcpyDeviceToDevice, st));
  }
  if (err_norms_host) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, (const Ctl*)nullptr, (const double*)h->partials,
                       h->stage_grid, h->n, h->rank_rec);
    double rec[kRec];
    MI_HIP(hipMemcpyAsync(rec, h->rank_rec, sizeof(rec), hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    err_norms_host[0] = rec[R_MAXA]; err_norms_host[1] = rec[R_MAXB]; err_norms_host[2] = rec[R_SUMA]; err_norms_host[3] = rec[R_FLAG];
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// function-level parity surface of the controller: controller_apply on caller-supplied numbers, one thread per case
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_controller_probe(CtrlParams P, int phase, int n, const double* in, const double* st, double* out,
                                                         const double* far_t) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  Ctl ctl;
  int* w = (int*)&ctl;
  for (int i = 0; i < (int)(sizeof(Ctl) / sizeof(int)); ++i) w[i] = 0;
  ctl.t0 = ctl.t1 = st[c * 4 + 0]; ctl.dt = st[c * 4 + 1]; ctl.h0 = st[c * 4 + 2]; ctl.d1 = st[c * 4 + 3];
  ctl.idx_y1 = 1;
  ctl.n_out = 1;                                   // one output time far ahead: the attempt never finishes the integration
  P.t_out = far_t;
  double rec[kRec];
  for (int i = 0; i < kRec; ++i) rec[i] = in[c * kRec + i];
  controller_apply(&ctl, rec, phase, P);
  double* o = out + c * 8;
  o[0] = ctl.ratio; o[1] = (double)ctl.accepted; o[2] = ctl.dt; o[3] = ctl.t1; o[4] = ctl.t0; o[5] = (double)ctl.status;
  o[6] = ctl.h0; o[7] = phase == PH_F0 ? ctl.d0 : ctl.d1;
  if (phase == PH_F0) o[2] = ctl.d1;
}

extern "C" int mi_ode_controller_update(const mi_ode_ctrl_params* p, int32_t phase, int32_t n_cases, const double* in_host,
                                        const double* st_host, double* out_host, void* stream) {
  if (p == nullptr || in_host == nullptr || st_host == nullptr || out_host == nullptr || n_cases < 1 || phase < 0 || phase > 2) {
    mi_set_error("controller_update: bad argument");
    return MI_ODE_E_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  CtrlParams P;
  memset(&P, 0, sizeof(P));
  P.rtol = p->rtol; P.atol = p->atol; P.safety = p->safety; P.ifactor = p->ifactor; P.dfactor = p->dfactor;
  P.inv_ifactor = 1.0 / p->ifactor; P.inv_dfactor = 1.0 / p->dfactor;
  P.max_num_steps = 2147483647LL; P.n_local = 0; P.order = p->order; P.init_order = p->init_order;
  P.controller = p->controller; P.is_f32 = p->dtype == MI_ODE_F32; P.n_stages = 6; P.auto_first_step = 1;
  double* dev = nullptr;
  const size_t n_in = (size_t)n_cases * kRec, n_st = (size_t)n_cases * 4, n_out = (size_t)n_cases * 8;
  MI_HIP(hipMalloc((void**)&dev, (n_in + n_st + n_out + 1) * sizeof(double)));
  const double far_t = 1e300;
  hipError_t e = hipMemcpyAsync(dev, in_host, n_in * sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dev + n_in, st_host, n_st * sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dev + n_in + n_st + n_out, &far_t, sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_controller_probe, dim3((n_cases + 63) / 64), dim3(64), 0, st, P, (int)phase, (int)n_cases, (const double*)dev,
                       (const double*)(dev + n_in), dev + n_in + n_st, (const double*)(dev + n_in + n_st + n_out));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_host, dev + n_in + n_st, n_out * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(dev);
  if (e != hipSuccess) { mi_set_error("controller_update failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// (B) stateless plane kernels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize_records(const double* part, int nblocks, double* result, int what) {
  __shared__ double rec[kRec];
  reduce_block_records(part, nblocks, rec);
  if (threadIdx.x == 0) {
    if (what == 0) { result[0] = rec[R_MAXA]; result[1] = rec[R_MAXB]; result[2] = rec[R_SUMA]; result[3] = rec[R_FLAG]; }
    else if (what == 2) result[0] = rec[R_FLAG];
    else if (what == 3) { result[0] = rec[R_SUMA]; result[1] = rec[R_SUMB]; }
    else result[0] = rec[R_SUMA];
  }
}

static int lincomb_impl(int32_t dtype, int64_t n, const void* base_dev, const void* const* xs_dev, const double* coef,
                              int32_t nx, double scale, const double* scale_dev, void* out_dev, void* stream) {
  if (n < 0 || nx < 1 || nx > MI_ODE_MAX_LINCOMB || xs_dev == nullptr || coef == nullptr || out_dev == nullptr) {
    mi_set_error("lincomb: bad argument");
    return MI_ODE_E_INVALID;
  }
  if (n == 0) return 0;
  LincombArgs A;
  memset(&A, 0, sizeof(A));
  A.base = base_dev; A.scale = scale; A.scale_dev = scale_dev; A.n = n; A.out = out_dev; A.nx = nx;
  for (int j = 0; j < nx; ++j) { A.x[j] = xs_dev[j]; A.coef[j] = coef[j]; }
  const int g = streaming_grid(n);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_ODE_F64) hipLaunchKernelGGL(k_lincomb<double>, dim3(g), dim3(256), 0, st, A);
  else if (dtype == MI_ODE_F32) hipLaunchKernelGGL(k_lincomb<float>, dim3(g), dim3(256), 0, st, A);
  else { mi_set_error("bad dtype"); return MI_ODE_E_INVALID; }
  MI_HIP(hipGetLastError());
  return 0;
}

extern "C" int mi_ode_lincomb(int32_t dtype, int64_t n, const void* base_dev, const void* const* xs_dev, const double* coef,
                              int32_t nx, double scale, void* out_dev, void* stream) {
  return lincomb_impl(dtype, n, base_dev, xs_dev, coef, nx, scale, nullptr, out_dev, stream);
}

extern "C" int mi_ode_lincomb_dev(int32_t dtype, int64_t n, const void* base_dev, const void* const* xs_dev, const double* coef,
                                  int32_t nx, const double* scale_dev, void* out_dev, void* stream) {
  if (scale_dev == nullptr) { mi_set_error("lincomb_dev: null scale pointer"); return MI_ODE_E_INVALID; }
  return lincomb_impl(dtype, n, base_dev, xs_dev, coef, nx, 0.0, scale_dev, out_dev, stream);
}

extern "C" int mi_ode_rk_stage_combine(int32_t dtype, int64_t n, const void* y0_dev, const void* const* k_dev, const double* beta_row,
                                       int32_t n_k, double dt, void* out_dev, void* stream) {
  return lincomb_impl(dtype, n, y0_dev, k_dev, beta_row, n_k, dt, nullptr, out_dev, stream);
}

extern "C" int mi_ode_error_norms(int32_t dtype, int64_t n, const void* err_dev, const void* y0_dev, const void* y1_dev,
                                  double* result_dev, void* workspace_dev, void* stream);
extern "C" int mi_ode_rk_error_reduce(int32_t dtype, int64_t n, const void* err_dev, const void* y0_dev, const void* y1_dev,
                                      double* result_dev, void* workspace_dev, void* stream) {
  return mi_ode_error_norms(dtype, n, err_dev, y0_dev, y1_dev, result_dev, workspace_dev, stream);
}

extern "C" int mi_ode_error_norms(int32_t dtype, int64_t n, const void* err_dev, const void* y0_dev, const void* y1_dev,
                                  double* result_dev, void* workspace_dev, void* stream) {
  if (n <= 0 || !err_dev || !y0_dev || !y1_dev || !result_dev || !workspace_dev) { mi_set_error("error_norms: bad argument"); return MI_ODE_E_INVALID; }
  const int g = streaming_grid(n);
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace_dev;
  if (dtype == MI_ODE_F64) hipLaunchKernelGGL(k_error_norms<double>, dim3(g), dim3(256), 0, st, (const double*)err_dev, (const double*)y0_dev, (const double*)y1_dev, (long long)n, part);
  else if (dtype == MI_ODE_F32) hipLaunchKernelGGL(k_error_norms<float>, dim3(g), dim3(256), 0, st, (const float*)err_dev, (const float*)y0_dev, (const float*)y1_dev, (long long)n, part);
  else { mi_set_error("bad dtype"); return MI_ODE_E_INVALID; }
  hipLaunchKernelGGL(k_finalize_records, dim3(1), dim3(256), 0, st, (const double*)part, g, result_dev, 0);
  MI_HIP(hipGetLastError());
  return 0;
}

extern "C" int mi_ode_scaled_sumsq(int32_t dtype, int64_t n, const void* x_dev, const void* xsub_dev, const void* y0_dev,
                                   double rtol, double atol, double* result_dev, void* workspace_dev, void* stream) {
  if (n <= 0 || !x_dev || !y0_dev || !result_dev || !workspace_dev) { mi_set_error("scaled_sumsq: bad argument"); return MI_ODE_E_INVALID; }
  const int g = streaming_grid(n);
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace_dev;
  if (dtype == MI_ODE_F64) hipLaunchKernelGGL(k_scaled_sumsq<double>, dim3(g), dim3(256), 0, st, (const double*)x_dev, (const double*)xsub_dev, (const double*)y0_dev, (long long)n, rtol, atol, part);
  else if (dtype == MI_ODE_F32) hipLaunchKernelGGL(k_scaled_sumsq<float>, dim3(g), dim3(256), 0, st, (const float*)x_dev, (const float*)xsub_dev, (const float*)y0_dev, (long long)n, rtol, atol, part);
  else { mi_set_error("bad dtype"); return MI_ODE_E_INVALID; }
  hipLaunchKernelGGL(k_finalize_records, dim3(1), dim3(256), 0, st, (const double*)part, g, result_dev, 1);
  MI_HIP(hipGetLastError());
  return 0;
}

// ---- the variable-order Adams solver's host loop on four plane kernels (mi_ode_adams_planes.h) -------------------------------
namespace {
int adams_planes_common(AdamsPlaneArgs& A, int32_t dtype, int64_t n, const void* const* phi, int32_t order, const double* g, const double* beta,
                        double dt, const char* who) {
  memset(&A, 0, sizeof(A));
  if (n <= 0 || order < 1 || order > kAdamsPlanesMax - 1 || phi == nullptr || (dtype != MI_ODE_F32 && dtype != MI_ODE_F64)) {
    mi_set_error("%s: bad argument (1 <= order <= %d)", who, kAdamsPlanesMax - 1);
    return MI_ODE_E_INVALID;
  }
  A.n = n; A.order = order; A.dt = dt;
  for (int j = 0; j < order; ++j) { if (phi[j] == nullptr) { mi_set_error("%s: phi[%d] is null", who, j); return MI_ODE_E_INVALID; } A.phi[j] = phi[j]; }
  if (g != nullptr) for (int j = 0; j <= order; ++j) A.g[j] = g[j];
  if (beta != nullptr) for (int j = 0; j < order; ++j) A.beta[j] = beta[j];
  return 0;
}
}  // namespace

extern "C" int mi_ode_adams_predict(int32_t dtype, int64_t n, const void* y_dev, const void* const* phi_dev, int32_t order, const double* g,
                                    const double* beta, double dt, void* p_out_dev, void* stream) {
  AdamsPlaneArgs A;
  int rc = adams_planes_common(A, dtype, n, phi_dev, order, g, beta, dt, "adams_predict");
  if (rc != 0) return rc;
  if (!y_dev || !p_out_dev || !g || !beta) { mi_set_error("adams_predict: null argument"); return MI_ODE_E_INVALID; }
  A.y0 = y_dev; A.out[0] = p_out_dev;
  const int gr = streaming_grid(n);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_ODE_F64) hipLaunchKernelGGL(k_adams_predict<double>, dim3(gr), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(k_adams_predict<float>, dim3(gr), dim3(256), 0, st, A);
  MI_HIP(hipGetLastError());
  return 0;
}

extern "C" int mi_ode_adams_correct(int32_t dtype, int64_t n, const void* y_dev, const void* p_dev, const void* f_p_dev, const void* const* phi_dev,
                                    int32_t order, const double* g, const double* beta, double dt, void* y_next_out_dev, void* ip_k_out_dev,
                                    void* ip_k1_out_dev, void* ip_k2_out_dev, double* result_dev, void* workspace_dev, void* stream) {
  AdamsPlaneArgs A;
  int rc = adams_planes_common(A, dtype, n, phi_dev, order, g, beta, dt, "adams_correct");
  if (rc != 0) return rc;
  if (!y_dev || !p_dev || !f_p_dev || !g || !beta || !y_next_out_dev || !ip_k_out_dev || !ip_k1_out_dev || !result_dev || !workspace_dev) {
    mi_set_error("adams_correct: null argument"); return MI_ODE_E_INVALID;
  }
  A.y0 = y_dev; A.p = p_dev; A.f = f_p_dev;
  A.out[0] = y_next_out_dev; A.out[1] = ip_k_out_dev; A.out[2] = ip_k1_out_dev; A.out[3] = ip_k2_out_dev;
  A.part = (double*)workspace_dev;
  const int gr = streaming_grid(n);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_ODE_F64) hipLaunchKernelGGL(k_adams_correct<double>, dim3(gr), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(k_adams_correct<float>, dim3(gr), dim3(256), 0, st, A);
  hipLaunchKernelGGL(k_finalize_records, dim3(1), dim3(256), 0, st, (const double*)A.part, gr, result_dev, 0);
  MI_HIP(hipGetLastError());
  return 0;
}

extern "C" int mi_ode_adams_error_sums(int32_t dtype, int64_t n, const void* xa_dev, double coef_a, const void* xb_dev, double coef_b, double dt,
                                       double tol, double* result_dev, void* workspace_dev, void* stream) {
  if (n <= 0 || !xa_dev || !result_dev || !workspace_dev || (dtype != MI_ODE_F32 && dtype != MI_ODE_F64)) { mi_set_error("adams_error_sums: bad argument"); return MI_ODE_E_INVALID; }
  AdamsPlaneArgs A;
  memset(&A, 0, sizeof(A));
  A.n = n; A.dt = dt; A.tol = tol; A.xa = xa_dev; A.xb = xb_dev; A.ca = coef_a; A.cb = coef_b; A.part = (double*)workspace_dev;
  const int gr = streaming_grid(n);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_ODE_F64) hipLaunchKernelGGL(k_adams_error_sums<double>, dim3(gr), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(k_adams_error_sums<float>, dim3(gr), dim3(256), 0, st, A);
  hipLaunchKernelGGL(k_finalize_records, dim3(1), dim3(256), 0, st, (const double*)A.part, gr, result_dev, 3);
  MI_HIP(hipGetLastError());
  return 0;
}

extern "C" int mi_ode_adams_update_phi(int32_t dtype, int64_t n, const void* f_new_dev, const void* const* phi_dev, int32_t order,
                                       const double* beta, void* const* new_phi_out_dev, void* stream) {
  AdamsPlaneArgs A;
  int rc = adams_planes_common(A, dtype, n, phi_dev, order, nullptr, beta, 0.0, "adams_update_phi");
  if (rc != 0) return rc;
  if (!f_new_dev || !beta || !new_phi_out_dev) { mi_set_error("adams_update_phi: null argument"); return MI_ODE_E_INVALID; }
  A.f = f_new_dev;
  for (int j = 0; j <= order; ++j) { if (new_phi_out_dev[j] == nullptr) { mi_set_error("adams_update_phi: output %d is null", j); return MI_ODE_E_INVALID; } A.out[j] = new_phi_out_dev[j]; }
  const int gr = streaming_grid(n);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_ODE_F64) hipLaunchKernelGGL(k_adams_update_phi<double>, dim3(gr), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(k_adams_update_phi<float>, dim3(gr), dim3(256), 0, st, A);
  MI_HIP(hipGetLastError());
  return 0;
}

extern "C" int mi_ode_not_converged(int32_t dtype, int64_t n, const void* a_dev, const void* b_dev, double rtol, double atol,
                                    double* result_dev, void* workspace_dev, void* stream) {
  if (n <= 0 || !a_dev || !b_dev || !result_dev || !workspace_dev) { mi_set_error("not_converged: bad argument"); return MI_ODE_E_INVALID; }
  const int g = streaming_grid(n);
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace_dev;
  if (dtype == MI_ODE_F64) hipLaunchKernelGGL(k_not_converged<double>, dim3(g), dim3(256), 0, st, (const double*)a_dev, (const double*)b_dev, (long long)n, rtol, atol, part);
  else if (dtype == MI_ODE_F32) hipLaunchKernelGGL(k_not_converged<float>, dim3(g), dim3(256), 0, st, (const float*)a_dev, (const float*)b_dev, (long long)n, rtol, atol, part);
  else { mi_set_error("bad dtype"); return MI_ODE_E_INVALID; }
  hipLaunchKernelGGL(k_finalize_records, dim3(1), dim3(256), 0, st, (const double*)part, g, result_dev, 2);
  MI_HIP(hipGetLastError());
  return 0;
}

template <typename T>
static int interp_eval_t(int32_t interp, int64_t n, const void* y0, const void* y1, const void* const* ks, int32_t nk,
                         const double* c_mid, double dt, double t0, double t1, double t, void* out, hipStream_t st) {
  InterpPtrs P;
  memset(&P, 0, sizeof(P));
  P.y0 = y0; P.y1 = y1;
  for (int j = 0; j < nk; ++j) P.k[j] = ks[j];
  InterpParams I;
  memset(&I, 0, sizeof(I));
  I.kind = interp; I.nk = nk;
  if (c_mid) for (int j = 0; j < nk && j < MI_ODE_MAX_LINCOMB; ++j) I.c_mid[j] = c_mid[j];
  const int g = streaming_grid(n);
  // the quartic fit takes the step's own dt (dopri5.py:41), not t1 - t0; hand it over through t1' = t0 + dt semantics
  if (nk == 7) hipLaunchKernelGGL((k_interp_eval<T, 7>), dim3(g), dim3(256), 0, st, P, (long long)n, t0, t1, t, dt, (T*)out, I);
  else if (nk == 4) hipLaunchKernelGGL((k_interp_eval<T, 4>), dim3(g), dim3(256), 0, st, P, (long long)n, t0, t1, t, dt, (T*)out, I);
  else if (nk == 3) hipLaunchKernelGGL((k_interp_eval<T, 3>), dim3(g), dim3(256), 0, st, P, (long long)n, t0, t1, t, dt, (T*)out, I);
  else if (nk == 2) hipLaunchKernelGGL((k_interp_eval<T, 2>), dim3(g), dim3(256), 0, st, P, (long long)n, t0, t1, t, dt, (T*)out, I);
  else if (nk == 14) hipLaunchKernelGGL((k_interp_eval<T, 14>), dim3(g), dim3(256), 0, st, P, (long long)n, t0, t1, t, dt, (T*)out, I);
  else { mi_set_error("interp_eval: nk must be 2, 3, 4, 7 or 14"); return MI_ODE_E_INVALID; }
  MI_HIP(hipGetLastError());
  return 0;
}

extern "C" int mi_ode_interp_eval(int32_t dtype, int32_t interp, int64_t n, const void* y0_dev, const void* y1_dev,
                                  const void* const* ks_dev, int32_t nk, const double* c_mid, double dt, double t0, double t1,
                                  double t, void* out_dev, void* stream) {
  if (n <= 0 || !y0_dev || !ks_dev || !out_dev) { mi_set_error("interp_eval: bad argument"); return MI_ODE_E_INVALID; }
  if (interp == MI_ODE_INTERP_QUARTIC_MID && (!y1_dev || !c_mid)) { mi_set_error("interp_eval: quartic needs y1 and c_mid"); return MI_ODE_E_INVALID; }
  if (interp != MI_ODE_INTERP_QUARTIC_MID && nk != 7) { mi_set_error("interp_eval: tsit5 needs nk == 7"); return MI_ODE_E_INVALID; }
  if (!(t0 <= t && t <= t1)) { mi_set_error("invalid interpolation, fails `t0 <= t <= t1`"); return MI_ODE_ST_BAD_T; }   // interp.py:59
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_ODE_F64) return interp_eval_t<double>(interp, n, y0_dev, y1_dev, ks_dev, nk, c_mid, dt, t0, t1, t, out_dev, st);
  if (dtype == MI_ODE_F32) return interp_eval_t<float>(interp, n, y0_dev, y1_dev, ks_dev, nk, c_mid, dt, t0, t1, t, out_dev, st);
  mi_set_error("bad dtype");
  return MI_ODE_E_INVALID;
}
