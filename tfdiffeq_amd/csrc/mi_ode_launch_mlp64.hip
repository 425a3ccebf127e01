// Launchers of the FLOAT64 MLP tile kernels (mi_ode_mlp64.h, round 6): per padded width, tableau and hidden activation; a translation
// unit of its own.  Every launch is preceded, on the same stream, by k_mlp64_pack: this call's weights -> the handle's packed copy.
#include <hip/hip_runtime.h>
#include "mi_ode_host.h"
#include "mi_ode_mlp64.h"

namespace {
int mlp_activation(const mi_ode_solver* h) { return (int)h->rhs.s[0]; }      // mi_ode_rhs.scalars[0]: 0 tanh, 1 relu, 2 softplus

template <int DP, int HP>
int pack_dims(mi_ode_solver* h, hipStream_t st) {
  hipLaunchKernelGGL((mi::k_mlp64_pack<DP, HP>), dim3((mi::MlpGeom64<DP, HP>::PACK + 255) / 256), dim3(256), 0, st, h->rhs, (int)h->d.dim, h->mlp_pack);
  return hipGetLastError() == hipSuccess ? 0 : MI_ODE_E_HIP;
}
int pack(mi_ode_solver* h, hipStream_t st) {
  if (h->mlp_pack == nullptr) { mi_set_error("float64 MLP kernels: no pack buffer"); return MI_ODE_E_INVALID; }
  if (h->mlp_dp == 16 && h->mlp_hp == 16) return pack_dims<16, 16>(h, st);
  if (h->mlp_dp == 64 && h->mlp_hp == 128) return pack_dims<64, 128>(h, st);
  mi_set_error("float64 MLP kernel: unsupported padded dims");
  return MI_ODE_E_INVALID;
}
template <class Fn>
void allow_lds(Fn fn, size_t lds) {                          // more than 64 KB of dynamic LDS needs the attribute (160 KB per CU on this part)
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) (void)hipGetLastError();
}
#define MI_LAUNCH64(KERNEL, ARG)                                          \
  do {                                                                    \
    auto fn_ = KERNEL;                                                    \
    allow_lds(fn_, lds);                                                  \
    hipLaunchKernelGGL(fn_, grid, block, lds, st, ARG);                   \
  } while (0)

template <int DP, int HP, int ACT>
int launch_mlp_act(mi_ode_solver* h, int mode, mi::MlpArgs& M, hipStream_t st) {
  using G = mi::MlpGeom64<DP, HP>;
  const size_t lds = G::lds_bytes();
  const dim3 grid(h->step_grid), block(64 * G::NW);
  const bool s6 = h->S == 6;
  if (mode == mi::MLP_STEP && h->S == 13) { MI_LAUNCH64((mi::k_mlp64<DP, HP, ACT, mi::MLP_STEP, 13, false>), M); return 0; }
  if (mode == mi::MLP_F0) MI_LAUNCH64((mi::k_mlp64<DP, HP, ACT, mi::MLP_F0, 6, false>), M);
  else if (mode == mi::MLP_INITB) MI_LAUNCH64((mi::k_mlp64<DP, HP, ACT, mi::MLP_INITB, 6, false>), M);
  else if (s6 && !h->ts_dense) MI_LAUNCH64((mi::k_mlp64<DP, HP, ACT, mi::MLP_STEP, 6, false>), M);
  else if (s6 && h->ts_dense) MI_LAUNCH64((mi::k_mlp64<DP, HP, ACT, mi::MLP_STEP, 6, true>), M);
  else if (!h->ts_dense) MI_LAUNCH64((mi::k_mlp64<DP, HP, ACT, mi::MLP_STEP, 3, false>), M);
  else MI_LAUNCH64((mi::k_mlp64<DP, HP, ACT, mi::MLP_STEP, 3, true>), M);
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
  using G = mi::MlpGeom64<DP, HP>;
  const size_t lds = G::lds_bytes();
  const dim3 grid(h->step_grid), block(64 * G::NW);
  MI_LAUNCH64((mi::k_fixed_mlp64<DP, HP, ACT>), A);
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
template <int DP, int HP, int ACT>
const void* persist_mlp_fn_act(const mi_ode_solver* h) {
  if (h->S == 6) return h->ts_dense ? (const void*)mi::k_persist_mlp64<DP, HP, ACT, 6, true> : (const void*)mi::k_persist_mlp64<DP, HP, ACT, 6, false>;
  if (h->S == 3 && !h->ts_dense) return (const void*)mi::k_persist_mlp64<DP, HP, ACT, 3, false>;
  if (h->S == 13 && !h->ts_dense) return (const void*)mi::k_persist_mlp64<DP, HP, ACT, 13, false>;     // dopri8
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
  const void* fn = nullptr;
  if (h->mlp_dp == 16 && h->mlp_hp == 16) { *lds = mi::MlpGeom64<16, 16>::lds_bytes(); *block = 64 * mi::MlpGeom64<16, 16>::NW; fn = persist_mlp_fn<16, 16>(h); }
  else if (h->mlp_dp == 64 && h->mlp_hp == 128) { *lds = mi::MlpGeom64<64, 128>::lds_bytes(); *block = 64 * mi::MlpGeom64<64, 128>::NW; fn = persist_mlp_fn<64, 128>(h); }
  static const void* allowed[64];                            // (the attribute is set once per kernel, not on every launch)
  static int n_allowed = 0;
  bool seen = false;
  for (int i = 0; i < n_allowed; ++i) seen = seen || allowed[i] == fn;
  if (fn != nullptr && !seen && *lds > 64 * 1024) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)*lds) != hipSuccess) (void)hipGetLastError();
    if (n_allowed < 64) allowed[n_allowed++] = fn;
  }
  return fn;
}
// the kernels read the pack through rhs.w[0] (MlpCtx64::init)
void with_pack(const mi_ode_solver* h, mi::RhsParams& r) { r.w[0] = h->mlp_pack; }
}  // namespace

int mi_mlp64_pack_doubles(int dp, int hp) {
  if (dp == 16 && hp == 16) return mi::MlpGeom64<16, 16>::PACK;
  if (dp == 64 && hp == 128) return mi::MlpGeom64<64, 128>::PACK;
  return 0;
}

int mi_launch_fixed_mlp_f64(mi_ode_solver* h, mi::FixedArgs& A, hipStream_t st) {
  int rc = pack(h, st);
  if (rc != 0) return rc;
  mi::FixedArgs F = A;
  with_pack(h, F.rhs);
  if (h->mlp_dp == 16 && h->mlp_hp == 16) rc = launch_fixed_mlp_dims<16, 16>(h, F, st);
  else if (h->mlp_dp == 64 && h->mlp_hp == 128) rc = launch_fixed_mlp_dims<64, 128>(h, F, st);
  else { mi_set_error("float64 MLP kernel: unsupported padded dims"); return MI_ODE_E_INVALID; }
  if (rc != 0) return rc;
  h->n_launches += 1;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mi_set_error("fixed-grid float64 MLP kernel launch failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
  return 0;
}

int mi_launch_mlp_f64(mi_ode_solver* h, int mode, mi::MlpArgs& M, hipStream_t st) {
  int rc = pack(h, st);
  if (rc != 0) return rc;
  mi::MlpArgs M2 = M;
  with_pack(h, M2.step.rhs);
  if (h->mlp_dp == 16 && h->mlp_hp == 16) rc = launch_mlp_dims<16, 16>(h, mode, M2, st);
  else if (h->mlp_dp == 64 && h->mlp_hp == 128) rc = launch_mlp_dims<64, 128>(h, mode, M2, st);
  else { mi_set_error("float64 MLP kernel: unsupported padded dims"); return MI_ODE_E_INVALID; }
  if (rc != 0) return rc;
  h->n_launches += 1;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mi_set_error("float64 MLP kernel launch failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
  return 0;
}

int mi_persist_capacity_mlp_f64(mi_ode_solver* h) {
  size_t lds = 0; int block = 0;
  const void* fn = persist_mlp_fn_any(h, &lds, &block);
  if (fn == nullptr) return 0;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return 0; }
  return h->num_cus * per_cu;
}

int mi_launch_persist_mlp_f64(mi_ode_solver* h, mi::PersistArgs& A, int grid, hipStream_t st) {
  size_t lds = 0; int block = 0;
  const void* fn = persist_mlp_fn_any(h, &lds, &block);
  if (fn == nullptr) { mi_set_error("no whole-call float64 MLP kernel for this problem"); return MI_ODE_E_INVALID; }
  int rc = pack(h, st);
  if (rc != 0) return rc;
  mi::PersistArgs P = A;
  with_pack(h, P.s.rhs);
  void* args[] = {(void*)&P};
  hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3((unsigned)block), args, lds, st);
  if (e != hipSuccess) { mi_set_error("whole-call float64 MLP kernel launch failed: %s", hipGetErrorString(e)); return MI_ODE_E_HIP; }
  h->n_launches += 1;
  return 0;
}
