// Right-hand-side plugins: user-supplied device code for a trajectory-local f(t, y) of small dimension, compiled
// against these headers into its own shared object and handed to libmi_ode through mi_ode_rhs.plugin
// (kind = MI_ODE_RHS_PLUGIN).  The plugin instantiates the SAME row-local kernels the built-in catalogue uses
// (k_persist_rowlocal: the whole adaptive integration in one launch; k_step_rowlocal: one launch per attempt;
// k_stage_rowlocal<F0 / INITB>: before_integrate; k_fixed_rowlocal: Euler / RK4 on a fixed grid; k_persist_rowlocal_planes: any
// batch size; k_fixed_adams_rowlocal / k_adams_vc_rowlocal: the Adams family in one launch) for its functor,
// so a custom system runs on exactly the code path of rhs.Lorenz & co.  (Per-stage kernels are not instantiated:
// fusion = 1 is rejected for plugins.)
//
// A plugin translation unit looks like this (tfdiffeq_amd.rhs.CustomRowLocal generates it):
//
//   #include "mi_ode_plugin.h"
//   namespace mi {
//   template <typename T> struct RhsUser {
//     static constexpr int D = 3;
//     const double* p;                                         // mi_ode_rhs.scalars[0..7]
//     __device__ explicit RhsUser(const RhsParams& r) : p(r.s) {}
//     __device__ __forceinline__ void operator()(T t, const T* y, T* k) const { ... }
//   };
//   }
//   MI_ODE_DEFINE_ROWLOCAL_PLUGIN(mi::RhsUser)
#pragma once
#include "mi_ode_host.h"
#include "mi_ode_persist.h"
#include "mi_ode_adams.h"
#include "mi_ode_adams_vc.h"

#define MI_ODE_PLUGIN_ABI 3

struct mi_ode_rowlocal_plugin {
  int abi;                     // MI_ODE_PLUGIN_ABI
  int dtype;                   // MI_ODE_F32 / MI_ODE_F64
  int dim;
  int cooperative;             // 0: a trajectory per thread (RowLocalPlugin); 1: a state ELEMENT per thread, f evaluated by the trajectory's
                               // threads together (CoopPlugin, round 5): persist_fn, persist_planes_fn, multistep_fn and launch_fixed only, dim <= 256
  size_t solver_size;          // sizeof(mi_ode_solver) the plugin was compiled against
  int (*launch_init)(mi_ode_solver* h, int mode, int nk, mi::StageArgs* A, hipStream_t st);   // M_F0 (nk 0), M_INITB (nk 1)
  int (*launch_step)(mi_ode_solver* h, mi::StepArgs* A, hipStream_t st);
  int (*launch_fixed)(mi_ode_solver* h, mi::FixedArgs* A, hipStream_t st);
  const void* (*persist_fn)(int S, int ts_dense);              // k_persist_rowlocal instantiation, or null
  // plugin ABI 2:
  const void* (*persist_planes_fn)(int S, int ts_dense);       // k_persist_rowlocal_planes (more trajectories than one per thread), or null
  const void* (*multistep_fn)(int kind);                       // 1 / 2: k_fixed_adams_rowlocal, 3: k_adams_vc_rowlocal (mi_ode_desc.multistep)
};

namespace mi {

template <typename T, class RHS>
struct RowLocalPlugin {
  static int launch_init(mi_ode_solver* h, int mode, int nk, StageArgs* A, hipStream_t st) {
    const dim3 grid(h->stage_grid), block(h->stage_block);
    if (mode == M_F0 && nk == 0) hipLaunchKernelGGL((k_stage_rowlocal<T, 0, M_F0, RHS>), grid, block, 0, st, *A);
    else if (mode == M_INITB && nk == 1) hipLaunchKernelGGL((k_stage_rowlocal<T, 1, M_INITB, RHS>), grid, block, 0, st, *A);
    else return MI_ODE_E_INVALID;
    return hipGetLastError() == hipSuccess ? 0 : MI_ODE_E_HIP;
  }
  static int launch_step(mi_ode_solver* h, StepArgs* A, hipStream_t st) {
    const dim3 grid(h->step_grid), block(h->step_block);
    if (h->S == 6 && h->ts_dense) hipLaunchKernelGGL((k_step_rowlocal<T, 6, true, RHS>), grid, block, 0, st, *A);
    else if (h->S == 6) hipLaunchKernelGGL((k_step_rowlocal<T, 6, false, RHS>), grid, block, 0, st, *A);
    else if (h->S == 3 && !h->ts_dense) hipLaunchKernelGGL((k_step_rowlocal<T, 3, false, RHS>), grid, block, 0, st, *A);
    else if (h->S == 13 && h->d.tableau.fsal && !h->ts_dense) hipLaunchKernelGGL((k_step_rowlocal<T, 13, false, RHS, true>), grid, block, 0, st, *A);
    else if (h->S == 1 && !h->d.tableau.fsal && !h->ts_dense) hipLaunchKernelGGL((k_step_rowlocal<T, 1, false, RHS, false>), grid, block, 0, st, *A);
    else return MI_ODE_E_INVALID;
    return hipGetLastError() == hipSuccess ? 0 : MI_ODE_E_HIP;
  }
  static int launch_fixed(mi_ode_solver*, FixedArgs* A, hipStream_t st) {
    long long g = (A->batch + 255) / 256;
    if (g < 1) g = 1;
    if (g > kMaxBlocks) g = kMaxBlocks;
    hipLaunchKernelGGL((k_fixed_rowlocal<T, RHS>), dim3((unsigned)g), dim3(256), 0, st, *A);
    return hipGetLastError() == hipSuccess ? 0 : MI_ODE_E_HIP;
  }
  static const void* persist_fn(int S, int ts_dense) {
    if (S == 6) return ts_dense ? (const void*)k_persist_rowlocal<T, 6, true, RHS> : (const void*)k_persist_rowlocal<T, 6, false, RHS>;
    if (S == 3 && !ts_dense) return (const void*)k_persist_rowlocal<T, 3, false, RHS>;
    if (S == 13 && !ts_dense) return (const void*)k_persist_rowlocal<T, 13, false, RHS, true>;      // dopri8 (FSAL shaped)
    if (S == 1 && !ts_dense) return (const void*)k_persist_rowlocal<T, 1, false, RHS, false>;       // adaptive_heun (not FSAL shaped)
    return nullptr;
  }
  static const void* persist_planes_fn(int S, int ts_dense) {
    if (S == 6) return ts_dense ? (const void*)k_persist_rowlocal_planes<T, 6, true, RHS> : (const void*)k_persist_rowlocal_planes<T, 6, false, RHS>;
    if (S == 3 && !ts_dense) return (const void*)k_persist_rowlocal_planes<T, 3, false, RHS>;
    if (S == 13 && !ts_dense) return (const void*)k_persist_rowlocal_planes<T, 13, false, RHS, true>;
    if (S == 1 && !ts_dense) return (const void*)k_persist_rowlocal_planes<T, 1, false, RHS, false>;
    return nullptr;
  }
  static const void* multistep_fn(int kind) {
    if (kind == 1 || kind == 2) return (const void*)k_fixed_adams_rowlocal<T, RHS>;
    if (kind == 3) return (const void*)k_adams_vc_rowlocal<T, RHS>;
    return nullptr;
  }
  static const mi_ode_rowlocal_plugin* table(int dtype) {
    static const mi_ode_rowlocal_plugin t = {MI_ODE_PLUGIN_ABI, dtype, RHS::D, 0, sizeof(mi_ode_solver),
                                            &launch_init, &launch_step, &launch_fixed, &persist_fn, &persist_planes_fn, &multistep_fn};
    return &t;
  }
};

// Cooperative plugins (round 5): a thread owns ONE state element (RHS::D = 1, RHS::kCoop, RHS::DIM = the system's dimension <= 256,
// RHS::tpw = trajectories per 256-thread workgroup); the functor receives its element, shares the trajectory's state through LDS
// and returns the derivative of its element (tfdiffeq_amd.rhs.CustomCoop generates it).  Same kernels: k_persist_rowlocal (the whole
// adaptive call in one launch - the only adaptive schedule), k_fixed_rowlocal (Euler / RK4 on a fixed grid), k_fixed_adams_rowlocal /
// k_adams_vc_rowlocal.
template <typename T, class RHS>
struct CoopPlugin {
  static const void* persist_fn(int S, int ts_dense) { return RowLocalPlugin<T, RHS>::persist_fn(S, ts_dense); }
  static const void* persist_planes_fn(int S, int ts_dense) { return RowLocalPlugin<T, RHS>::persist_planes_fn(S, ts_dense); }   // any batch size
  static const void* multistep_fn(int kind) { return RowLocalPlugin<T, RHS>::multistep_fn(kind); }
  static int launch_fixed(mi_ode_solver*, FixedArgs* A, hipStream_t st) {       // Euler / RK4 on a fixed grid: trajectories never interact, any batch
    const long long tpw = RHS::tpw(A->rhs, A->dim);
    long long g = (A->batch + tpw - 1) / tpw;
    if (g < 1) g = 1;
    if (g > kMaxBlocks) g = kMaxBlocks;
    hipLaunchKernelGGL((k_fixed_rowlocal<T, RHS>), dim3((unsigned)g), dim3(256), 0, st, *A);
    return hipGetLastError() == hipSuccess ? 0 : MI_ODE_E_HIP;
  }
  static const mi_ode_rowlocal_plugin* table(int dtype) {
    static const mi_ode_rowlocal_plugin t = {MI_ODE_PLUGIN_ABI, dtype, RHS::DIM, 1, sizeof(mi_ode_solver),
                                            nullptr, nullptr, &launch_fixed, &persist_fn, &persist_planes_fn, &multistep_fn};
    return &t;
  }
};

}  // namespace mi

// MI_ODE_PLUGIN_F32 / MI_ODE_PLUGIN_F64 select which state dtypes the plugin is built for (both by default)
#if !defined(MI_ODE_PLUGIN_F32) && !defined(MI_ODE_PLUGIN_F64)
#define MI_ODE_PLUGIN_F32 1
#define MI_ODE_PLUGIN_F64 1
#endif
#ifdef MI_ODE_PLUGIN_F64
#define MI_ODE_PLUGIN_CASE_F64(RHS) if (dtype == MI_ODE_F64) return mi::RowLocalPlugin<double, RHS<double>>::table(MI_ODE_F64);
#else
#define MI_ODE_PLUGIN_CASE_F64(RHS)
#endif
#ifdef MI_ODE_PLUGIN_F32
#define MI_ODE_PLUGIN_CASE_F32(RHS) if (dtype == MI_ODE_F32) return mi::RowLocalPlugin<float, RHS<float>>::table(MI_ODE_F32);
#else
#define MI_ODE_PLUGIN_CASE_F32(RHS)
#endif

#define MI_ODE_DEFINE_ROWLOCAL_PLUGIN(RHS)                                               \
  extern "C" const mi_ode_rowlocal_plugin* mi_ode_plugin_get(int dtype) {                \
    MI_ODE_PLUGIN_CASE_F64(RHS)                                                          \
    MI_ODE_PLUGIN_CASE_F32(RHS)                                                          \
    return nullptr;                                                                      \
  }

#ifdef MI_ODE_PLUGIN_F64
#define MI_ODE_COOP_CASE_F64(RHS) if (dtype == MI_ODE_F64) return mi::CoopPlugin<double, RHS<double>>::table(MI_ODE_F64);
#else
#define MI_ODE_COOP_CASE_F64(RHS)
#endif
#ifdef MI_ODE_PLUGIN_F32
#define MI_ODE_COOP_CASE_F32(RHS) if (dtype == MI_ODE_F32) return mi::CoopPlugin<float, RHS<float>>::table(MI_ODE_F32);
#else
#define MI_ODE_COOP_CASE_F32(RHS)
#endif
#define MI_ODE_DEFINE_COOP_PLUGIN(RHS)                                                   \
  extern "C" const mi_ode_rowlocal_plugin* mi_ode_plugin_get(int dtype) {                \
    MI_ODE_COOP_CASE_F64(RHS)                                                            \
    MI_ODE_COOP_CASE_F32(RHS)                                                            \
    return nullptr;                                                                      \
  }
