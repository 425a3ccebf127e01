// What can the two wavefronts of a SIMD overlap on gfx950?  (round 3, config-4 kernel design)
// One workgroup per CU; wavefront w runs role[w % nroles] for `iters` iterations of an unrolled block and records its own
// duration in shader clocks (s_memtime).  Roles:  M1 = 32 dependent v_mfma_f64_16x16x4_f64 (one chain), M2 = two chains of 16,
// V = 64 independent v_fma_f64 on 8 registers, L = 16 ds_read_b128 + wait, I = idle (exits at once).
// hipcc --offload-arch=gfx950 -O3 -o mfma_pair mfma_pair.hip && ./mfma_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K, int KIND>
__device__ __forceinline__ void blk_mf(d4& c, double a, double b, double (&f)[8], int (&q)[8], int lane) {
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[(i * K + j) & 7]) : "v"(b));
      else if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q[(i * K + j) & 7]) : "v"(lane));
      else { f4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lane * 16)); q[(i * K + j) & 7] += (int)t[0]; }
    }
  }
  if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

enum Role { R_IDLE = 0, R_M1 = 1, R_M2 = 2, R_V = 3, R_L = 4, R_MV = 5 /* 32 MFMA then 64 VALU */, R_VM = 6 /* 64 VALU then 32 MFMA */,
  R_MV16 = 20, R_VM16 = 21, R_MV32 = 22, R_VM32 = 23, R_MV48 = 24, R_VM48 = 25, R_MV96 = 26, R_VM96 = 27,
  R_MF1 = 7, R_MF2 = 8, R_MF4 = 9, R_MI1 = 10, R_MI2 = 11, R_MI4 = 12, R_ML1 = 13 };

__device__ __forceinline__ void blk_m1(d4& c, double a, double b) {
#pragma unroll
  for (int i = 0; i < 32; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void blk_m2(d4& c, d4& e, double a, double b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    e = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, e, 0, 0, 0);
  }
}
__device__ __forceinline__ void blk_v(double (&f)[8], double x) {
#pragma unroll
  for (int i = 0; i < 64; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[i & 7]) : "v"(x));
}
template <int NV>
__device__ __forceinline__ void blk_vn(double (&f)[8], double x) {
#pragma unroll
  for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[i & 7]) : "v"(x));
}
__device__ __forceinline__ void blk_l(f4 (&r)[4], const char* lds, int lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    f4 t;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(lane * 16), "n"(0));
    r[i & 3] = t;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) void k(const int* roles, int nroles, int iters, int sync, long long* out, double* sink) {
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  ((float*)lds)[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const int role = roles[wave % nroles];
  d4 c = {0, 0, 0, 0}, e = c;
  double a = lane * 1e-3, b = 1.0 + lane * 1e-4, f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  f4 r[4];
  int q[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  for (int i = 0; i < 4; ++i) r[i] = f4{0, 0, 0, 0};
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    switch (role) {
      case R_M1: blk_m1(c, a, b); break;
      case R_M2: blk_m2(c, e, a, b); break;
      case R_V: blk_v(f, b); break;
      case R_L: blk_l(r, lds, lane); break;
      case R_MV: blk_m1(c, a, b); blk_v(f, b); break;
      case R_VM: blk_v(f, b); blk_m1(c, a, b); break;
      case R_MV16: blk_m1(c, a, b); blk_vn<16>(f, b); break;
      case R_VM16: blk_vn<16>(f, b); blk_m1(c, a, b); break;
      case R_MV32: blk_m1(c, a, b); blk_vn<32>(f, b); break;
      case R_VM32: blk_vn<32>(f, b); blk_m1(c, a, b); break;
      case R_MV48: blk_m1(c, a, b); blk_vn<48>(f, b); break;
      case R_VM48: blk_vn<48>(f, b); blk_m1(c, a, b); break;
      case R_MV96: blk_m1(c, a, b); blk_vn<96>(f, b); break;
      case R_VM96: blk_vn<96>(f, b); blk_m1(c, a, b); break;
      case R_MF1: blk_mf<1, 0>(c, a, b, f, q, lane); break;
      case R_MF2: blk_mf<2, 0>(c, a, b, f, q, lane); break;
      case R_MF4: blk_mf<4, 0>(c, a, b, f, q, lane); break;
      case R_MI1: blk_mf<1, 1>(c, a, b, f, q, lane); break;
      case R_MI2: blk_mf<2, 1>(c, a, b, f, q, lane); break;
      case R_MI4: blk_mf<4, 1>(c, a, b, f, q, lane); break;
      case R_ML1: blk_mf<1, 2>(c, a, b, f, q, lane); break;
      default: break;
    }
    if (sync) __syncthreads();
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  double s = c[0] + c[1] + c[2] + c[3] + e[0] + e[1] + e[2] + e[3];
  for (int i = 0; i < 8; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += r[i][0];
  for (int i = 0; i < 8; ++i) s += q[i];
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  int* d_roles; long long* d_out; double* d_sink;
  hipMalloc(&d_roles, 64); hipMalloc(&d_out, sizeof(long long) * 16 * cus); hipMalloc(&d_sink, sizeof(double) * 512 * cus);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { const char* name; int nwaves; int nroles; int roles[8]; int sync; };
  const Case cases[] = {
    {"1 wave/SIMD  M1                       ", 4, 1, {R_M1}, 0},
    {"1 wave/SIMD  M2 (two chains)          ", 4, 1, {R_M2}, 0},
    {"2 waves/SIMD M1 + M1                  ", 8, 1, {R_M1}, 0},
    {"2 waves/SIMD M2 + M2                  ", 8, 1, {R_M2}, 0},
    {"2 waves/SIMD M1 + idle                ", 8, 8, {R_M1, R_M1, R_M1, R_M1, R_IDLE, R_IDLE, R_IDLE, R_IDLE}, 0},
    {"2 waves/SIMD M1 + V   (w, w+4 paired) ", 8, 8, {R_M1, R_M1, R_M1, R_M1, R_V, R_V, R_V, R_V}, 0},
    {"2 waves/SIMD M1 + V   (w, w+1 paired) ", 8, 2, {R_M1, R_V}, 0},
    {"2 waves/SIMD V + V                    ", 8, 1, {R_V}, 0},
    {"1 wave/SIMD  V                        ", 4, 1, {R_V}, 0},
    {"2 waves/SIMD M1 + L   (w, w+4)        ", 8, 8, {R_M1, R_M1, R_M1, R_M1, R_L, R_L, R_L, R_L}, 0},
    {"2 waves/SIMD L + L                    ", 8, 1, {R_L}, 0},
    {"2 waves/SIMD MV + MV  barrier/iter    ", 8, 1, {R_MV}, 1},
    {"2 waves/SIMD MV + VM (w,w+4) barrier  ", 8, 8, {R_MV, R_MV, R_MV, R_MV, R_VM, R_VM, R_VM, R_VM}, 1},
    {"2 waves/SIMD MV + VM (w,w+1) barrier  ", 8, 2, {R_MV, R_VM}, 1},
    {"2 waves/SIMD MV + VM (w,w+4) no barr. ", 8, 8, {R_MV, R_MV, R_MV, R_MV, R_VM, R_VM, R_VM, R_VM}, 0},
    {"2 waves/SIMD MV + MV  no barrier      ", 8, 1, {R_MV}, 0},
    {"MV16 + MV16 (lockstep) barrier          ", 8, 1, {R_MV16}, 1},
    {"MV16 + VM16 (w,w+4 opposite) barrier    ", 8, 8, {R_MV16, R_MV16, R_MV16, R_MV16, R_VM16, R_VM16, R_VM16, R_VM16}, 1},
    {"MV32 + MV32 (lockstep) barrier          ", 8, 1, {R_MV32}, 1},
    {"MV32 + VM32 (w,w+4 opposite) barrier    ", 8, 8, {R_MV32, R_MV32, R_MV32, R_MV32, R_VM32, R_VM32, R_VM32, R_VM32}, 1},
    {"MV48 + MV48 (lockstep) barrier          ", 8, 1, {R_MV48}, 1},
    {"MV48 + VM48 (w,w+4 opposite) barrier    ", 8, 8, {R_MV48, R_MV48, R_MV48, R_MV48, R_VM48, R_VM48, R_VM48, R_VM48}, 1},
    {"MV96 + MV96 (lockstep) barrier          ", 8, 1, {R_MV96}, 1},
    {"MV96 + VM96 (w,w+4 opposite) barrier    ", 8, 8, {R_MV96, R_MV96, R_MV96, R_MV96, R_VM96, R_VM96, R_VM96, R_VM96}, 1},
    {"1 wave  M1 + 1 fp64 FMA per MFMA      ", 4, 1, {R_MF1}, 0},
    {"1 wave  M1 + 2 fp64 FMA per MFMA      ", 4, 1, {R_MF2}, 0},
    {"1 wave  M1 + 4 fp64 FMA per MFMA      ", 4, 1, {R_MF4}, 0},
    {"1 wave  M1 + 1 v_add_u32 per MFMA     ", 4, 1, {R_MI1}, 0},
    {"1 wave  M1 + 2 v_add_u32 per MFMA     ", 4, 1, {R_MI2}, 0},
    {"1 wave  M1 + 4 v_add_u32 per MFMA     ", 4, 1, {R_MI4}, 0},
    {"1 wave  M1 + 1 ds_read_b128 per MFMA  ", 4, 1, {R_ML1}, 0},
    {"2 waves M1+2FMA each                  ", 8, 1, {R_MF2}, 0},
    {"2 waves M1+4FMA each                  ", 8, 1, {R_MF4}, 0},
    {"2 waves M1+2 v_add_u32 each           ", 8, 1, {R_MI2}, 0},
    {"2 waves M1+4 v_add_u32 each           ", 8, 1, {R_MI4}, 0},
    {"2 waves M1+1 ds_read each             ", 8, 1, {R_ML1}, 0},
  };
  const int iters = 2000;
  printf("%-42s %10s %12s %12s   (per iteration of the block; MFMA block = 32 x 64 = 2048 pipe cycles)\n", "case", "us/launch", "clk wave0", "clk wave4");
  for (const Case& cs : cases) {
    hipMemcpy(d_roles, cs.roles, sizeof(int) * 8, hipMemcpyHostToDevice);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(cus), dim3(64 * cs.nwaves), 4096, 0, d_roles, cs.nroles, iters, cs.sync, d_out, d_sink);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    long long o[16]; hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
    printf("%-42s %10.1f %12.1f %12.1f   wall clk/iter @2.4GHz %.0f\n", cs.name, ms * 1e3, (double)o[0] / iters, (double)o[cs.nwaves > 4 ? 4 : 1] / iters,
           ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
