// Fused RK stage kernel for trajectory-local right-hand sides of tiny width (dim 2..4):
// one thread owns one trajectory (row), every value of the stage lives in registers.
//   reads  y0, k_0..k_{NK-1}      (NK+1 planes)
//   writes k_out (and y1 for the last stage)
// => (NK + 2) planes of traffic per stage, +1 for the last: the 34-units-per-attempt
// structure of SURVEY.md 8(d).  Bound: HBM (a handful of flops per element).
#pragma once
#include <type_traits>
#include "mi_ode_dev.h"

namespace mi {

// ---- right-hand sides (device catalogue, dim <= 4) -------------------------------------------
// (y**3) @ W with W [2,2] row-major: examples/ode_demo.py:33-35
template <typename T>
struct RhsCubic2 {
  static constexpr int D = 2;
  T w00, w01, w10, w11;
  __device__ explicit RhsCubic2(const RhsParams& p)
      : w00((T)p.s[0]), w01((T)p.s[1]), w10((T)p.s[2]), w11((T)p.s[3]) {}
  __device__ __forceinline__ void operator()(T, const T* y, T* f) const {
    const T c0 = y[0] * y[0] * y[0], c1 = y[1] * y[1] * y[1];
    f[0] = c0 * w00 + c1 * w10;
    f[1] = c0 * w01 + c1 * w11;
  }
};

// y @ W with W [2,2] (2-D linear systems of examples/ode_usage.ipynb)
template <typename T>
struct RhsLinear2 {
  static constexpr int D = 2;
  T w00, w01, w10, w11;
  __device__ explicit RhsLinear2(const RhsParams& p)
      : w00((T)p.s[0]), w01((T)p.s[1]), w10((T)p.s[2]), w11((T)p.s[3]) {}
  __device__ __forceinline__ void operator()(T, const T* y, T* f) const {
    f[0] = y[0] * w00 + y[1] * w10;
    f[1] = y[0] * w01 + y[1] * w11;
  }
};

// sum_k x[k] * W[k * ld + j] + acc0: eight interleaved partial sums (k = 0, 8, 16 .. | 1, 9, .. | ..) so that the loads of eight terms
// are in flight together (the evaluation is a chain of dependent memory round trips otherwise); a fixed order - every launch
// geometry gives the same bits
template <typename T>
__device__ __forceinline__ T coop_dot_col(const T* x, const T* Wc, int n, int ld, T acc0) {
  T a[8] = {acc0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0};
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    T w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = Wc[(long long)(k + i) * ld];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = fma(x[k + i], w[i], a[i]);
  }
  for (; k < n; ++k) a[0] = fma(x[k], Wc[(long long)k * ld], a[0]);
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// y @ W (+ b) / (y ** 3) @ W of ANY dim <= 256 for the one-launch MULTISTEP kernels (mi_ode_adams.h, mi_ode_adams_vc.h): those keep a
// trajectory's state and its history of derivatives in registers, which a dim-51 system (DETEST C4) does not fit into one
// thread's - so here a thread owns ONE state element (D = 1: its history is 13 registers again) and the threads of a trajectory
// evaluate f together: the state goes through LDS, every thread forms its column's dot product (coop_dot_col above).  A 256-thread workgroup holds floor(256 / dim) trajectories; `rowmap` gives a thread its
// element.  The kernels call rhs() from uniform control flow (it contains barriers).
template <typename T>
struct RhsLinearCoop {
  static constexpr int D = 1;
  static constexpr bool kCoop = true;
  const T* W;
  const T* bias;
  int dim, cube;
  __device__ explicit RhsLinearCoop(const RhsParams& p) : W((const T*)p.w[0]), bias((const T*)p.b[0]), dim(p.hidden), cube(p.cube) {}
  static __host__ __device__ int tpw(const RhsParams&, int dim) { return 256 / dim; }       // trajectories per 256-thread workgroup
  __device__ __forceinline__ void operator()(T, const T* y, T* f) const {
    __shared__ T s_y[256];
    const int slot = (int)threadIdx.x / dim, col = (int)threadIdx.x - slot * dim;
    __syncthreads();                                         // the previous evaluation's readers are done
    s_y[threadIdx.x] = cube ? y[0] * y[0] * y[0] : y[0];
    __syncthreads();
    T acc = (T)0;
    if ((slot + 1) * dim <= (int)blockDim.x) {
      acc = coop_dot_col<T>(s_y + slot * dim, W + col, dim, dim, (T)0);       // (round 5: eight partial sums in flight instead of one chain)
      if (bias != nullptr) acc = acc + bias[col];
    }
    f[0] = acc;
  }
};
// The ODEFunc network dim -> hidden -> hidden -> dim (models/dense_odenet.py:41-92; tanh / relu / softplus, optionally time dependent)
// for the same one-launch MULTISTEP kernels, in float32 AND float64 and for any dim, hidden <= 256 (round 5; the MFMA tile kernels of
// mi_ode_mlp.h are float32, dim <= 64, hidden <= 128 and have no multistep schedule).  A thread owns one state element; the dim threads of
// a trajectory's state element; the hidden units of ALL the workgroup's trajectories are dealt to all its threads (a 2-dimensional state
// with 50 hidden units would otherwise leave 25 dependent dot products to each of two threads), the state and both hidden activations go
// through LDS, then every thread forms its own output column.  Weights are read from global
// memory - consecutive threads read consecutive columns, every trajectory of the workgroup the same addresses (L1 / L2 hits).  Vector
// ALU work, latency bound: what it replaces is the host loop with ~6 launches per evaluation.  p.cube carries dim here (aux field).
constexpr int kMlpCoopCap = 2048;        // hidden activations of a workgroup's trajectories per layer: trajectories per workgroup = min(256 / dim, 2048 / hidden)
template <typename T>
struct RhsMlpCoop {
  static constexpr int D = 1;
  static constexpr bool kCoop = true;
  const T *W1, *W2, *W3, *B1, *B2, *B3;
  int dim, hd, act, td, tpw_;
  __device__ explicit RhsMlpCoop(const RhsParams& p)
      : W1((const T*)p.w[0]), W2((const T*)p.w[1]), W3((const T*)p.w[2]), B1((const T*)p.b[0]), B2((const T*)p.b[1]), B3((const T*)p.b[2]),
        dim(p.cube), hd(p.hidden), act((int)p.s[0]), td(p.s[1] != 0.0 ? 1 : 0), tpw_(tpw(p, p.cube)) {}
  static __host__ __device__ int tpw(const RhsParams& p, int dim) {
    const int a = 256 / dim, b = kMlpCoopCap / (p.hidden > 0 ? p.hidden : 1);
    return a < b ? a : b;
  }
  static __device__ __forceinline__ T activation(int act, T x) {
    if constexpr (std::is_same<T, float>::value) {
      return act == 0 ? tanhf(x) : act == 1 ? (x > 0.f ? x : (x != x ? x : 0.f)) : (x > 20.f ? x : log1pf(expf(x)));
    } else {
      return act == 0 ? tanh(x) : act == 1 ? (x > 0.0 ? x : (x != x ? x : 0.0)) : (x > 30.0 ? x : log1p(exp(x)));
    }
  }
  __device__ __forceinline__ void operator()(T t, const T* y, T* f) const {
    __shared__ T s_y[256];
    __shared__ T s_h1[kMlpCoopCap];
    __shared__ T s_h2[kMlpCoopCap];
    const int slot = (int)threadIdx.x / dim, col = (int)threadIdx.x - slot * dim;
    const int units = tpw_ * hd;                             // hidden units of the workgroup's trajectories: dealt to ALL its threads
    __syncthreads();                                         // the previous evaluation's readers are done
    s_y[threadIdx.x] = y[0];
    __syncthreads();
    for (int u = (int)threadIdx.x; u < units; u += (int)blockDim.x) {     // fc1 (+ the time row of concat([t, x]), dense_odenet.py:79-84)
      const int sl = u / hd, j = u - sl * hd;
      T acc = B1 != nullptr ? B1[j] : (T)0;
      if (td) acc = fma(t, W1[j], acc);
      s_h1[u] = activation(act, coop_dot_col<T>(s_y + sl * dim, W1 + (long long)td * hd + j, dim, hd, acc));
    }
    __syncthreads();
    for (int u = (int)threadIdx.x; u < units; u += (int)blockDim.x) {
      const int sl = u / hd, j = u - sl * hd;
      s_h2[u] = activation(act, coop_dot_col<T>(s_h1 + sl * hd, W2 + j, hd, hd, B2 != nullptr ? B2[j] : (T)0));
    }
    __syncthreads();
    T acc = (T)0;
    if (slot < tpw_) acc = coop_dot_col<T>(s_h2 + slot * hd, W3 + col, hd, dim, B3 != nullptr ? B3[col] : (T)0);
    f[0] = acc;
  }
};
template <class R, class = void>
struct rhs_is_coop : std::false_type {};
template <class R>
struct rhs_is_coop<R, std::void_t<decltype(R::kCoop)>> : std::true_type {};

// trajectories per 256-thread workgroup of a cooperative right-hand side (1 for the thread-per-trajectory ones: unused there)
template <class RHS>
__device__ __forceinline__ int coop_tpw(const RhsParams& rp, int dim) {
  if constexpr (rhs_is_coop<RHS>::value) return RHS::tpw(rp, dim);
  else return 1;
}

// which element(s) of the state this thread owns: offset of its first element, whether it exists, elements per plane
template <class RHS>
__device__ __forceinline__ void rowmap(long long batch, int dim, const RhsParams& rp, long long& off, bool& live, long long& n_plane) {
  if constexpr (rhs_is_coop<RHS>::value) {
    const int tpw = RHS::tpw(rp, dim);                       // trajectories per workgroup
    const int slot = (int)threadIdx.x / dim, col = (int)threadIdx.x - slot * dim;
    const long long traj = (long long)blockIdx.x * tpw + slot;
    live = slot < tpw && traj < batch;
    off = traj * dim + col;
    n_plane = batch * dim;
  } else {
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    live = row < batch;
    off = row * RHS::D;
    n_plane = batch * RHS::D;
  }
}

// Lotka-Volterra, examples/ode_usage.ipynb cells 39-42 / README.md:67-82
template <typename T>
struct RhsLotkaVolterra {
  static constexpr int D = 2;
  T a, b, c, d;
  __device__ explicit RhsLotkaVolterra(const RhsParams& p) : a((T)p.s[0]), b((T)p.s[1]), c((T)p.s[2]), d((T)p.s[3]) {}
  __device__ __forceinline__ void operator()(T, const T* y, T* f) const {
    const T u = y[0], v = y[1];
    f[0] = a * u - b * u * v;
    f[1] = -c * v + d * u * v;
  }
};

// Lorenz, examples/lorenz_attractor.py:28-37
template <typename T>
struct RhsLorenz {
  static constexpr int D = 3;
  T sigma, beta, rho;
  __device__ explicit RhsLorenz(const RhsParams& p) : sigma((T)p.s[0]), beta((T)p.s[1]), rho((T)p.s[2]) {}
  __device__ __forceinline__ void operator()(T, const T* y, T* f) const {
    f[0] = sigma * (y[1] - y[0]);
    f[1] = y[0] * (rho - y[2]) - y[1];
    f[2] = y[0] * y[1] - beta * y[2];
  }
};

template <typename T, int D>
struct alignas((D * sizeof(T)) % 16 == 0 ? 16 : ((D * sizeof(T)) % 8 == 0 ? 8 : sizeof(T))) RowVec {
  T v[D];
};

template <typename T, int NK, int MODE, class RHS>
__global__ __launch_bounds__(256) void k_stage_rowlocal(StageArgs A) {
  constexpr int D = RHS::D;
  using Row = RowVec<T, D>;
  Resolved<T> R;
  if (!resolve<T, NK, MODE>(A, R)) return;
  const RHS rhs(A.rhs);
  const T sign = (T)A.rhs.sign;
  Acc acc;
  const long long nrows = A.batch;
  for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < nrows;
       row += (long long)gridDim.x * blockDim.x) {
    const Row y0 = *(const Row*)(R.y0 + row * D);
    Row kj[NK > 0 ? NK : 1];
#pragma unroll
    for (int j = 0; j < NK; ++j) kj[j] = *(const Row*)(R.k[j] + row * D);
    T ys[D], aux[D], kn[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      T kk[NK > 0 ? NK : 1];
#pragma unroll
      for (int j = 0; j < NK; ++j) kk[j] = kj[j].v[d];
      ys[d] = combine_elem<T, NK, MODE>(y0.v[d], kk, R.hs, A, aux[d]);
      reduce_flat<T, MODE>(y0.v[d], ys[d], A, acc);
    }
    if constexpr (MODE == M_F0) {
      if (A.copy_a != nullptr) *(Row*)((T*)A.copy_a + row * D) = y0;
      if (A.copy_b != nullptr) *(Row*)((T*)A.copy_b + row * D) = y0;
    }
    rhs(sign * R.ts, ys, kn);                     // reversed time: f <- -f(-t, y) (misc.py:318-321)
#pragma unroll
    for (int d = 0; d < D; ++d) kn[d] = sign * kn[d];
    if (R.k_out != nullptr) {
      Row o;
#pragma unroll
      for (int d = 0; d < D; ++d) o.v[d] = kn[d];
      *(Row*)(R.k_out + row * D) = o;
    }
    Row y1;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const T k0 = NK > 0 ? kj[0].v[d] : (T)0;
      const T v = epilogue_elem<T, NK, MODE>(y0.v[d], k0, kn[d], aux[d], R.hs, A, acc);
      y1.v[d] = (MODE == M_LAST_FSAL) ? ys[d] : v;
    }
    if constexpr (mode_writes_y1(MODE)) *(Row*)(R.y1 + row * D) = y1;
  }
  if constexpr (mode_has_reduction(MODE)) {
    __shared__ double red[80];
    block_reduce_store(acc, red, A.partials + (long long)blockIdx.x * kRec);
  }
}

}  // namespace mi
