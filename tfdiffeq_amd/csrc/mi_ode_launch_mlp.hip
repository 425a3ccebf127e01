// Launchers of the fused MLP kernels (fp32 only): a translation unit of its own - the kernels are instantiated per padded
// width, tableau and hidden activation, and compile in parallel with the rest of the library.
#include <hip/hip_runtime.h>
#include "mi_ode_host.h"
// ---- fused MLP kernels (fp32 only) ---------------------------------------------------------------------------
#include "mi_ode_mlp.h"

namespace {
int mlp_activation(const mi_ode_solver* h) { return (int)h->rhs.s[0]; }      // mi_ode_rhs.scalars[0]: 0 tanh, 1 relu, 2 softplus

template <int DP, int HP, int ACT>
int launch_mlp_act(mi_ode_solver* h, int mode, mi::MlpArgs& M, hipStream_t st) {
  using G = mi::MlpGeom<DP, HP>;
  const size_t lds = G::lds_bytes();
  const dim3 grid(h->step_grid), block(64 * G::NW);
  const bool s6 = h->S == 6;
  if (mode == mi::MLP_F0) hipLaunchKernelGGL((mi::k_mlp<DP, HP, ACT, mi::MLP_F0, 6, false>), grid, block, lds, st, M);
  else if (mode == mi::MLP_INITB) hipLaunchKernelGGL((mi::k_mlp<DP, HP, ACT, mi::MLP_INITB, 6, false>), grid, block, lds, st, M);
  else if (s6 && !h->ts_dense) hipLaunchKernelGGL((mi::k_mlp<DP, HP, ACT, mi::MLP_STEP, 6, false>), grid, block, lds, st, M);
  else if (s6 && h->ts_dense) hipLaunchKernelGGL((mi::k_mlp<DP, HP, ACT, mi::MLP_STEP, 6, true>), grid, block, lds, st, M);
  else if (!h->ts_dense) hipLaunchKernelGGL((mi::k_mlp<DP, HP, ACT, mi::MLP_STEP, 3, false>), grid, block, lds, st, M);
  else hipLaunchKernelGGL((mi::k_mlp<DP, HP, ACT, mi::MLP_STEP, 3, true>), grid, block, lds, st, M);
  return 0;
}
template <int DP, int HP>
int launch_mlp_dims(mi_ode_solver* h, int mode, mi::MlpArgs& M, hipStream_t st) {
  switch (mlp_activation(h)) {
    case mi::MLP_ACT_TANH: return launch_mlp_act<DP, HP, mi::MLP_ACT_TANH>(h, mode, M, st);
    case mi::MLP_ACT_RELU: return launch_mlp_act<DP, HP, mi::MLP_ACT_RELU>(h, mode, M, st);
    case mi::MLP_ACT_SOFTPLUS: return launch_mlp_act<DP, HP, mi::MLP_ACT_SOFTPLUS>(h, mode, M, st);
    default: mi_set_error("MLP kernels: unknown activation code %d", mlp_activation(h)); return MI_ODE_E_INVALID;
  }
}
}  // namespace

int mi_launch_mlp_f32(mi_ode_solver* h, int mode, mi::MlpArgs& M, hipStream_t st) {
  int rc = MI_ODE_E_INVALID;
  if (h->mlp_dp == 16 && h->mlp_hp == 16) rc = launch_mlp_dims<16, 16>(h, mode, M, st);
  else if (h->mlp_dp == 16 && h->mlp_hp == 128) rc = launch_mlp_dims<16, 128>(h, mode, M, st);
  else if (h->mlp_dp == 64 && h->mlp_hp == 16) rc = launch_mlp_dims<64, 16>(h, mode, M, st);
  else if (h->mlp_dp == 64 && h->mlp_hp == 128) rc = launch_mlp_dims<64, 128>(h, mode, M, st);
  else { mi_set_error("MLP kernel: unsupported padded dims"); return MI_ODE_E_INVALID; }
  if (rc != 0) return rc;
  h->n_launches += 1;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mi_set_error("MLP kernel launch failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
  return 0;
}

// ---- whole call in one launch for the MLP family (k_persist_mlp) -------------------------------------------------
namespace {
template <int DP, int HP, int ACT>
const void* persist_mlp_fn_act(const mi_ode_solver* h) {
  if (h->S == 6) return h->ts_dense ? (const void*)mi::k_persist_mlp<DP, HP, ACT, 6, true> : (const void*)mi::k_persist_mlp<DP, HP, ACT, 6, false>;
  if (h->S == 3 && !h->ts_dense) return (const void*)mi::k_persist_mlp<DP, HP, ACT, 3, false>;
  return nullptr;
}
template <int DP, int HP>
const void* persist_mlp_fn(const mi_ode_solver* h) {
  switch (mlp_activation(h)) {
    case mi::MLP_ACT_TANH: return persist_mlp_fn_act<DP, HP, mi::MLP_ACT_TANH>(h);
    case mi::MLP_ACT_RELU: return persist_mlp_fn_act<DP, HP, mi::MLP_ACT_RELU>(h);
    case mi::MLP_ACT_SOFTPLUS: return persist_mlp_fn_act<DP, HP, mi::MLP_ACT_SOFTPLUS>(h);
    default: return nullptr;
  }
}
const void* persist_mlp_fn_any(const mi_ode_solver* h, size_t* lds, int* block) {
  if (h->mlp_dp == 16 && h->mlp_hp == 16) { *lds = mi::MlpGeom<16, 16>::lds_bytes(); *block = 64 * mi::MlpGeom<16, 16>::NW; return persist_mlp_fn<16, 16>(h); }
  if (h->mlp_dp == 16 && h->mlp_hp == 128) { *lds = mi::MlpGeom<16, 128>::lds_bytes(); *block = 64 * mi::MlpGeom<16, 128>::NW; return persist_mlp_fn<16, 128>(h); }
  if (h->mlp_dp == 64 && h->mlp_hp == 16) { *lds = mi::MlpGeom<64, 16>::lds_bytes(); *block = 64 * mi::MlpGeom<64, 16>::NW; return persist_mlp_fn<64, 16>(h); }
  if (h->mlp_dp == 64 && h->mlp_hp == 128) { *lds = mi::MlpGeom<64, 128>::lds_bytes(); *block = 64 * mi::MlpGeom<64, 128>::NW; return persist_mlp_fn<64, 128>(h); }
  return nullptr;
}
}  // namespace

int mi_persist_capacity_mlp_f32(mi_ode_solver* h) {
  size_t lds = 0; int block = 0;
  const void* fn = persist_mlp_fn_any(h, &lds, &block);
  if (fn == nullptr) return 0;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds) != hipSuccess || per_cu < 1) return 0;
  return h->num_cus * per_cu;
}

int mi_launch_persist_mlp_f32(mi_ode_solver* h, mi::PersistArgs& A, int grid, hipStream_t st) {
  size_t lds = 0; int block = 0;
  const void* fn = persist_mlp_fn_any(h, &lds, &block);
  if (fn == nullptr) { mi_set_error("no whole-call MLP kernel for this problem"); return MI_ODE_E_INVALID; }
  void* args[] = {(void*)&A};
  hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3((unsigned)block), args, lds, st);
  if (e != hipSuccess) { mi_set_error("whole-call MLP kernel launch failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
  h->n_launches += 1;
  return 0;
}

