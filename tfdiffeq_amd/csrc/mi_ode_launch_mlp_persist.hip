// Whole-call launchers of the fused MLP kernels (fp32 only): k_persist_mlp per padded width, tableau and hidden activation - a
// translation unit of its own so that these (the largest kernels of the library) compile in parallel with the per-attempt ones.
#include <hip/hip_runtime.h>
#include "mi_ode_host.h"
#include "mi_ode_mlp.h"

namespace {
int mlp_activation(const mi_ode_solver* h) { return (int)h->rhs.s[0]; }      // mi_ode_rhs.scalars[0]: 0 tanh, 1 relu, 2 softplus
}  // namespace

// ---- whole call in one launch for the MLP family (k_persist_mlp) -------------------------------------------------
namespace {
template <int DP, int HP, int ACT>
const void* persist_mlp_fn_act(const mi_ode_solver* h) {
  if (h->S == 6) return h->ts_dense ? (const void*)mi::k_persist_mlp<DP, HP, ACT, 6, true> : (const void*)mi::k_persist_mlp<DP, HP, ACT, 6, false>;
  if (h->S == 3 && !h->ts_dense) return (const void*)mi::k_persist_mlp<DP, HP, ACT, 3, false>;
  if (h->S == 13 && !h->ts_dense) return (const void*)mi::k_persist_mlp<DP, HP, ACT, 13, false>;     // dopri8
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

