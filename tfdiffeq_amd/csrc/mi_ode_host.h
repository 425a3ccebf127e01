// Host-side solver object shared by the translation units of libmi_ode.so.
#pragma once
#include <hip/hip_runtime.h>
#include "mi_ode_dev.h"
namespace mi { struct AdamsArgs; struct AdamsVcArgs; }

namespace mi {
struct StepArgs;
struct MlpArgs;
struct FixedArgs;
struct PersistArgs;
struct InitArgs;

enum Family {
  FAM_NONE = 0,
  FAM_CUBIC2,        // rowlocal, dim 2: (y**3) @ W
  FAM_LINEAR2,       // rowlocal, dim 2: y @ W
  FAM_LV,            // rowlocal, dim 2
  FAM_LORENZ,        // rowlocal, dim 3
  FAM_LINEAR_VALU,   // any dim <= 256, optional cube / bias
  FAM_LINEAR_MFMA,   // 3 <= dim <= 256 (tile widths 16, 32, 64, 128 with W resident in registers; 256 with W streamed from L2)
  FAM_MLP,           // fp32 dim<=64, hidden<=128: whole-attempt MFMA kernel only
  FAM_PLUGIN,        // row-local user code behind a mi_ode_rowlocal_plugin table (mi_ode_plugin.h)
  FAM_MLP_COOP,      // the ODEFunc network outside the tile kernels' box on the cooperative kernels (a thread per state element): the
                     // whole-call kernel (adaptive 3- / 6-row tableaus) and the one-launch multistep kernels; fp32 / fp64, dim, hidden <= 256
  FAM_PLUGIN_COOP    // user code on the same cooperative kernels (plugin table with cooperative = 1; rhs.CustomCoop): dim <= 256
};

struct LaunchInfo {   // filled per (mode) at create time
  int grid;
  int block;
  size_t lds;
};

}  // namespace mi

struct mi_ode_rowlocal_plugin;

struct mi_ode_solver {
  mi_ode_desc d;
  const mi_ode_rowlocal_plugin* plugin;   // FAM_PLUGIN
  mi::Family family;
  int is_f32;
  size_t elt;                 // sizeof(state dtype)
  long long n;                // batch * dim (this rank)
  long long stride;           // bytes between planes
  int S;                      // tableau rows
  int num_cus;
  // device workspace
  char* planes;
  double* partials;           // [kMaxBlocks][kRec]
  double* rank_rec;           // [kRec]
  double* gathered;           // [world][kRec]
  mi::Ctl* ctl;
  unsigned* ticket;           // last-workgroup-done counter of the whole-attempt kernels
  int fused_ctl;              // 1: the whole-attempt kernel also runs the controller (single rank)
  double* t_out_dev;
  void* cur_out;              // solution rows of the advance() call in progress
  int t_out_cap;
  // pinned host staging
  mi::Ctl* ctl_host;
  double* t_out_host;
  int t_out_host_cap;
  int t_out_busy;             // an un-synchronised H2D copy out of t_out_host may still be in flight
  // launch geometry of the stage kernels
  int stage_grid, stage_block;
  int step_fused;             // 1: whole-attempt kernel in use
  int step_grid, step_block;
  int persist;                // 1: whole integration in one launch (mi_ode_persist.h)
  int persist_planes;         // 1: row-local system too large for one trajectory per thread: k_persist_rowlocal_planes (state in HBM planes)
  int persist_planes_block;   // its workgroup size
  int persist_capable;        // 1: such a kernel exists for this problem and its grid is co-resident (transport aside)
  int persist_grid;
  int persist_sleep_first, persist_sleep_poll;   // hand-off back-off (units of 64 clocks)
  int persist_spin_limit;     // bound on the polls of one hand-off
  int persist_xspin_limit;    // ... of its cross-rank part (seconds: a rank that launches late is normal)
  int persist_spin_first;     // ... of the first hand-off of a launch (residency check)
  int init_tiles16;           // 1: before_integrate runs on the 16-row tile kernels (k_init_linear_mfma), grid = step_grid
  double* xrank_dev;          // device view of desc.xrank_host (registered), or null
  double* gbuf;               // device: the global record WG 0 broadcasts after a cross-rank hand-off (2 parities)
  int xrank_on;               // 1: multi-rank calls use the whole-call kernel with the cross-rank hand-off
  int xrank_registered;
  void* xpeer_local;          // this rank's mailbox in its own HBM (mi_ode_xpeer_prepare), or null
  void* xpeer_open[64];       // peers' mailboxes as mapped here by hipIpcOpenMemHandle (entry `rank` = xpeer_local)
  double** xpeer_tab_dev;     // device copy of that pointer table (what the kernels read), or null
  int xpeer_world;
  void* nccl_comm;            // ncclComm_t created by mi_ode_rccl_connect, or null
  unsigned xrank_tests;       // self-test rounds use their own sequence range
  unsigned seq;               // hand-off sequence numbers already used on this handle (identical on every rank)
  int ts_dense;               // whole-attempt kernels evaluate the tsit5 seven-weight dense output (else the quartic)
  int mlp_dp, mlp_hp;         // padded widths of the MLP kernel instantiation
  double* mlp_pack;           // float64 MLP tile kernels (mi_ode_mlp64.h): the packed, zero-padded weights, refreshed before every launch
  int nseg;                   // tuple state: components packed into the one buffer (mi_ode_desc.n_segments), 0 / 1: a single tensor
  int seg_blk[MI_ODE_MAX_SEGMENTS + 1];   // first workgroup of every component in the whole-call kernel's grid
  double* adams_tab;          // device: the multistep coefficient tables of the descriptor (mi_ode_adams.h), or null
  long long* adams_res;       // pinned host: {steps whose corrector did not converge, status}; multistep = 3: {attempts, accepted, nfe, status}
  double adams_gamma_star[13]; // multistep = 3 (adams.py:15-18)
  int lin_dp;                 // FAM_LINEAR_MFMA: tile width the kernels are instantiated for (16 / 32 / 64 / 128 / 256 >= dim, zero padded)
  void* lin_pack;             // lin_dp = 256 (W streamed, mi_ode_step_fused.h LinCtx::STREAM): the copy of W in consumption order, refreshed before every launch
  // bookkeeping
  long long n_launches;
  int n_polls;
  int begun;
  long long last_call_attempts;  // attempts the previous advance() needed: first-chunk estimate for repeated calls
  int own_exchange;
  // optional event profiling (desc.profile)
  hipEvent_t ev_a[64], ev_b[64], ev_c[64];
  int ev_ready;
  long long enq_attempts;       // attempts enqueued since begin
  long long prof_done;          // attempts already harvested
  double prof_last_ms, prof_all_ms;
  long long prof_n;
  mi::RhsParams rhs;
  mi::CtrlParams cp;
  mi::InterpParams ip;
};

// implemented once per state dtype (mi_ode_launch_f64.hip / mi_ode_launch_f32.hip)
int mi_launch_stage_f64(mi_ode_solver* h, int mode, int nk, mi::StageArgs& A, hipStream_t st);
int mi_launch_stage_f32(mi_ode_solver* h, int mode, int nk, mi::StageArgs& A, hipStream_t st);
int mi_launch_step_f64(mi_ode_solver* h, mi::StepArgs& A, hipStream_t st);
int mi_launch_step_f32(mi_ode_solver* h, mi::StepArgs& A, hipStream_t st);
int mi_launch_fixed_f64(mi_ode_solver* h, mi::FixedArgs& A, hipStream_t st);
int mi_launch_adams_f64(mi_ode_solver* h, mi::AdamsArgs& A, int grid, hipStream_t st);
int mi_launch_adams_f32(mi_ode_solver* h, mi::AdamsArgs& A, int grid, hipStream_t st);
int mi_adams_capacity_f64(mi_ode_solver* h);
int mi_launch_adams_vc_f64(mi_ode_solver* h, mi::AdamsVcArgs& A, int grid, hipStream_t st);
int mi_launch_adams_vc_f32(mi_ode_solver* h, mi::AdamsVcArgs& A, int grid, hipStream_t st);
int mi_adams_vc_capacity_f64(mi_ode_solver* h);
int mi_adams_vc_capacity_f32(mi_ode_solver* h);
int mi_adams_capacity_f32(mi_ode_solver* h);
int mi_launch_fixed_f32(mi_ode_solver* h, mi::FixedArgs& A, hipStream_t st);
int mi_launch_persist_f64(mi_ode_solver* h, mi::PersistArgs& A, int grid, hipStream_t st);
int mi_launch_persist_f32(mi_ode_solver* h, mi::PersistArgs& A, int grid, hipStream_t st);
int mi_launch_init_linear_f64(mi_ode_solver* h, int phase, mi::InitArgs& I, hipStream_t st);
int mi_launch_init_linear_f32(mi_ode_solver* h, int phase, mi::InitArgs& I, hipStream_t st);
int mi_persist_capacity_f64(mi_ode_solver* h);
int mi_persist_capacity_f32(mi_ode_solver* h);
int mi_launch_persist_mlp_f32(mi_ode_solver* h, mi::PersistArgs& A, int grid, hipStream_t st);
int mi_persist_capacity_mlp_f32(mi_ode_solver* h);
int mi_launch_mlp_f32(mi_ode_solver* h, int mode, mi::MlpArgs& M, hipStream_t st);
int mi_launch_fixed_mlp_f32(mi_ode_solver* h, mi::FixedArgs& A, hipStream_t st);
// ... and their float64 counterparts (csrc/mi_ode_launch_mlp64.hip)
int mi_launch_persist_mlp_f64(mi_ode_solver* h, mi::PersistArgs& A, int grid, hipStream_t st);
int mi_persist_capacity_mlp_f64(mi_ode_solver* h);
int mi_launch_mlp_f64(mi_ode_solver* h, int mode, mi::MlpArgs& M, hipStream_t st);
int mi_launch_fixed_mlp_f64(mi_ode_solver* h, mi::FixedArgs& A, hipStream_t st);
int mi_mlp64_pack_doubles(int dp, int hp);
int mi_stage_geometry_f64(mi_ode_solver* h);
int mi_stage_geometry_f32(mi_ode_solver* h);

void mi_set_error(const char* fmt, ...);
#define MI_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      mi_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return MI_ODE_E_HIP;                                                               \
    }                                                                                    \
  } while (0)
