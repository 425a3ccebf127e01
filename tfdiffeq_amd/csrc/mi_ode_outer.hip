// Parameter vector-Jacobian products of the LINEAR right-hand side f(t, y) = y W + b over a batch (include/mi_ode.h section A''):
//     -a^T df/dW = -(y^T a)   [dim, dim] (weights [in, out], the layout of mi_ode_rhs.w)        -a^T df/db = -(column sums of a)
// i.e. what the reference's augmented dynamics (tfdiffeq/adjoint.py:69-105) obtains from the GradientTape for a dense layer without
// activation.  It is a GEMM with M = N = dim <= 128 and K = batch (65536 in BASELINE config 4's shape): the shape vendor BLAS
// libraries serve worst (`y.t() @ a` takes 7.0 ms in rocBLAS fp64 - 0.3 TFLOP/s - against 63 us here).
//
// k_outer_partial: a workgroup of dim/16 wavefronts takes a contiguous slab of rows, 16 at a time through LDS; wavefront w owns the 16
// output columns 16 w .. 16 w + 15 and keeps the dim x 16 block of y^T a in MFMA accumulators (dim/16 x 4 registers); per 4 rows one
// operand of `a`, dim/16 operands of `y` and dim/16 MFMAs.  The slabs' partial blocks go to a workspace; k_outer_fold sums them in a
// fixed order (eight interleaved chains) - no atomics, the same bits every time.
// Bound: HBM - both planes are read once (2 x batch x dim elements); fp64 MFMA time is 2 batch dim^2 flop (27 us at config 4's shape
// against 27 us for the 128 MB at 5 TB/s; measured 48 + 15 us).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mi_ode_host.h"
#include "mi_ode_stage_linear.h"

using namespace mi;

namespace {

constexpr int kOuterMaxSlabs = 256;

// VL: the planes are staged with 16-byte loads (dim a multiple of 16 bytes' worth of elements, 16-byte aligned planes); else element loads
template <typename T, int D, bool VL>
__global__ __launch_bounds__(D * 4) void k_outer_partial(const T* __restrict__ y, const T* __restrict__ a, long long batch, int dim,
                                                         long long rows_per_slab, T* __restrict__ part) {
  using TR = MfmaTraits<T>;
  using acc_t = typename TR::acc_t;
  constexpr int MB = D / 16, NT = D * 4, R = 16, EPT = R * D / NT;   // 16-row tiles (32 KB of LDS at D = 128, fp64; 32 rows: the same 52 us), EPT = 4 elements per thread and plane
  constexpr int LDP = D + 16;                                  // row stride: the four row groups of an MFMA operand read land in different banks
  __shared__ __attribute__((aligned(16))) T sy[R * LDP], sa[R * LDP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lg = lane >> 4;
  acc_t acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) acc[m] = acc_t{0, 0, 0, 0};
  T colsum = (T)0;
  const long long r0 = (long long)blockIdx.x * rows_per_slab;
  long long r1 = r0 + rows_per_slab;
  if (r1 > batch) r1 = batch;
  const int ncol = 16 * w + li;                                // this lane's column of a (B operand: B[k = lg][n = li])
  // A tile of both planes goes through LDS once per workgroup (coalesced element loads, zero padded to D columns / R rows); the
  // wavefronts take their MFMA operands from there.  (Operands straight from global memory - every wavefront reads all of y, eight
  // times the traffic through L1 / L2 - measured 66-73 us per call at 65536 x 128 whatever the prefetch depth; matrix-pipe time
  // 27 us, the two planes at 5 TB/s 27 us.)
  constexpr int VEC = TR::VEC, CPT = EPT / VEC;                 // 16-byte chunks per thread and plane (2 elements fp64, 4 fp32)
  using CH = Chunk<T, VEC>;
  static_assert(EPT % VEC == 0, "tile / workgroup geometry");
  T py[EPT], pa[EPT];
  auto fetch = [&](long long t0) {
    if constexpr (VL) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int idx = (c * NT + tid) * VEC, row = idx / D, col = idx % D;
        const bool ok = t0 + row < r1 && col < dim;                // (dim % VEC == 0: a chunk is inside or outside as a whole)
        CH vy, va;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { vy.v[v] = (T)0; va.v[v] = (T)0; }
        if (ok) { vy = *(const CH*)(y + (t0 + row) * dim + col); va = *(const CH*)(a + (t0 + row) * dim + col); }
#pragma unroll
        for (int v = 0; v < VEC; ++v) { py[c * VEC + v] = vy.v[v]; pa[c * VEC + v] = va.v[v]; }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int idx = e * NT + tid, row = idx / D, col = idx % D;
        const bool ok = t0 + row < r1 && col < dim;
        py[e] = ok ? y[(t0 + row) * dim + col] : (T)0;
        pa[e] = ok ? a[(t0 + row) * dim + col] : (T)0;
      }
    }
  };
  if (r0 < r1) fetch(r0);
  for (long long t0 = r0; t0 < r1; t0 += R) {
    __syncthreads();                                           // (the previous tile's operands have been read)
    if constexpr (VL) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        CH vy, va;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { vy.v[v] = py[c * VEC + v]; va.v[v] = pa[c * VEC + v]; }
        const int idx = (c * NT + tid) * VEC;
        *(CH*)(sy + idx / D * LDP + idx % D) = vy;
        *(CH*)(sa + idx / D * LDP + idx % D) = va;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) { const int idx = e * NT + tid; sy[idx / D * LDP + idx % D] = py[e]; sa[idx / D * LDP + idx % D] = pa[e]; }
    }
    __syncthreads();
    if (t0 + R < r1) fetch(t0 + R);                            // in flight under this tile's MFMAs
    T ay[R / 4][MB], bv[R / 4];                                // every operand of the tile out of LDS first, then the chain (round 5: the scheduler put
#pragma unroll                                                 // an LDS wait in front of every MFMA - 48 -> 3x us per call at 65536 x 128)
    for (int u = 0; u < R / 4; ++u) {
      const int row = 4 * u + lg;
      bv[u] = sa[row * LDP + ncol];
#pragma unroll
      for (int m = 0; m < MB; ++m) ay[u][m] = sy[row * LDP + 16 * m + li];                        // A operand: A[i = li][k = lg] = y[row][16 m + li]
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < R / 4; ++u) {
      colsum += bv[u];
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = TR::mfma(ay[u][m], bv[u], acc[m]);
    }
  }
  // this slab's block: [D x D] (padded) then [D] column sums
  T* out = part + (long long)blockIdx.x * (D * D + D);
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(16 * m + TR::acc_row(lane, i)) * D + ncol] = acc[m][i];
  colsum += __shfl_xor(colsum, 16, 64);
  colsum += __shfl_xor(colsum, 32, 64);
  if (lg == 0) out[D * D + ncol] = colsum;
}

template <typename T, int D>
__global__ __launch_bounds__(256) void k_outer_fold(const T* __restrict__ part, int slabs, int dim, T scale, T* __restrict__ out_w,
                                                    T* __restrict__ out_b) {
  const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= D * D + D) return;
  // eight independent partial sums (a fixed order all the same): one dependent chain over 256 slabs is 256 memory round trips -
  // 64 us per call, as long as the partial kernel itself
  constexpr long long STR = D * D + D;
  T q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int g = 0;
  for (; g + 8 <= slabs; g += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] += part[(long long)(g + u) * STR + e];
  }
  for (; g < slabs; ++g) q[0] += part[(long long)g * STR + e];
  const T s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
  if (e < D * D) {
    const int m = e / D, n = e % D;
    if (m < dim && n < dim) out_w[m * dim + n] = scale * s;
  } else if (out_b != nullptr && e - D * D < dim) {
    out_b[e - D * D] = scale * s;
  }
}

int pad_dim(int dim) { return dim <= 16 ? 16 : (dim <= 32 ? 32 : (dim <= 64 ? 64 : 128)); }

int slabs_for(long long batch) {
  long long g = (batch + 255) / 256;                           // at least 256 rows per slab
  if (g > kOuterMaxSlabs) g = kOuterMaxSlabs;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T, int D>
int outer_launch(long long batch, int dim, const void* y, const void* a, double scale, void* out_w, void* out_b, void* ws, hipStream_t st) {
  const int slabs = slabs_for(batch);
  long long rps = (batch + slabs - 1) / slabs;
  rps = (rps + 3) / 4 * 4;
  const bool vl = dim % MfmaTraits<T>::VEC == 0 && (((uintptr_t)y | (uintptr_t)a) & 15u) == 0;
  if (vl) hipLaunchKernelGGL((k_outer_partial<T, D, true>), dim3(slabs), dim3(D * 4), 0, st, (const T*)y, (const T*)a, batch, dim, rps, (T*)ws);
  else hipLaunchKernelGGL((k_outer_partial<T, D, false>), dim3(slabs), dim3(D * 4), 0, st, (const T*)y, (const T*)a, batch, dim, rps, (T*)ws);
  hipLaunchKernelGGL((k_outer_fold<T, D>), dim3((D * D + D + 255) / 256), dim3(256), 0, st, (const T*)ws, slabs, dim, (T)scale, (T*)out_w, (T*)out_b);
  MI_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int outer_dims(int dp, long long batch, int dim, const void* y, const void* a, double scale, void* out_w, void* out_b, void* ws, hipStream_t st) {
  switch (dp) {
    case 16: return outer_launch<T, 16>(batch, dim, y, a, scale, out_w, out_b, ws, st);
    case 32: return outer_launch<T, 32>(batch, dim, y, a, scale, out_w, out_b, ws, st);
    case 64: return outer_launch<T, 64>(batch, dim, y, a, scale, out_w, out_b, ws, st);
    default: return outer_launch<T, 128>(batch, dim, y, a, scale, out_w, out_b, ws, st);
  }
}

}  // namespace

extern "C" int64_t mi_ode_outer_workspace_bytes(int32_t dtype, int64_t batch, int32_t dim) {
  if (batch < 1 || dim < 1 || dim > 128 || (dtype != MI_ODE_F32 && dtype != MI_ODE_F64)) return -1;
  const int dp = pad_dim(dim);
  return (int64_t)slabs_for(batch) * (dp * dp + dp) * (dtype == MI_ODE_F64 ? 8 : 4);
}

extern "C" int mi_ode_outer_reduce(int32_t dtype, int64_t batch, int32_t dim, const void* y_dev, const void* a_dev, double scale,
                                   void* out_w_dev, void* out_b_dev, void* workspace_dev, void* stream) {
  if (y_dev == nullptr || a_dev == nullptr || out_w_dev == nullptr || workspace_dev == nullptr) { mi_set_error("outer_reduce: null argument"); return MI_ODE_E_INVALID; }
  if (batch < 1 || dim < 1 || dim > 128) { mi_set_error("outer_reduce: batch >= 1, 1 <= dim <= 128"); return MI_ODE_E_INVALID; }
  const int dp = pad_dim(dim);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_ODE_F64) return outer_dims<double>(dp, batch, dim, y_dev, a_dev, scale, out_w_dev, out_b_dev, workspace_dev, st);
  if (dtype == MI_ODE_F32) return outer_dims<float>(dp, batch, dim, y_dev, a_dev, scale, out_w_dev, out_b_dev, workspace_dev, st);
  mi_set_error("outer_reduce: dtype must be MI_ODE_F32 or MI_ODE_F64");
  return MI_ODE_E_INVALID;
}
