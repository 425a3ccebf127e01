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
  if (mode == mi::MLP_STEP && h->S == 13) {                   // dopri8 (dopri8.py:12-77): FSAL shaped, quartic dense output
    hipLaunchKernelGGL((mi::k_mlp<DP, HP, ACT, mi::MLP_STEP, 13, false>), grid, block, lds, st, M);
    return 0;
  }
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
template <int DP, int HP, int ACT>
void launch_fixed_mlp_act(mi_ode_solver* h, mi::FixedArgs& A, hipStream_t st) {
  using G = mi::MlpGeom<DP, HP>;
  hipLaunchKernelGGL((mi::k_fixed_mlp<DP, HP, ACT>), dim3(h->step_grid), dim3(64 * G::NW), G::lds_bytes(), st, A);
}
template <int DP, int HP>
int launch_fixed_mlp_dims(mi_ode_solver* h, mi::FixedArgs& A, hipStream_t st) {
  switch (mlp_activation(h)) {
    case mi::MLP_ACT_TANH: launch_fixed_mlp_act<DP, HP, mi::MLP_ACT_TANH>(h, A, st); return 0;
    case mi::MLP_ACT_RELU: launch_fixed_mlp_act<DP, HP, mi::MLP_ACT_RELU>(h, A, st); return 0;
    case mi::MLP_ACT_SOFTPLUS: launch_fixed_mlp_act<DP, HP, mi::MLP_ACT_SOFTPLUS>(h, A, st); return 0;
    default: mi_set_error("MLP kernels: unknown activation code %d", mlp_activation(h)); return MI_ODE_E_INVALID;
  }
}
}  // namespace

// Euler / RK4 (3/8 rule) on a fixed grid for the MLP family: the whole integration in one launch (k_fixed_mlp)
int mi_launch_fixed_mlp_f32(mi_ode_solver* h, mi::FixedArgs& A, hipStream_t st) {
  int rc = MI_ODE_E_INVALID;
  if (h->mlp_dp == 16 && h->mlp_hp == 16) rc = launch_fixed_mlp_dims<16, 16>(h, A, st);
  else if (h->mlp_dp == 16 && h->mlp_hp == 128) rc = launch_fixed_mlp_dims<16, 128>(h, A, st);
  else if (h->mlp_dp == 64 && h->mlp_hp == 16) rc = launch_fixed_mlp_dims<64, 16>(h, A, st);
  else if (h->mlp_dp == 64 && h->mlp_hp == 128) rc = launch_fixed_mlp_dims<64, 128>(h, A, st);
  else { mi_set_error("MLP kernel: unsupported padded dims"); return MI_ODE_E_INVALID; }
  if (rc != 0) return rc;
  h->n_launches += 1;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mi_set_error("fixed-grid MLP kernel launch failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
  return 0;
}

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

