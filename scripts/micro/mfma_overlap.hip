// Is v_mfma_f64_16x16x4_f64 safe when vDst overlaps SrcA (the compiler allocates  v[10:17] <- v[16:17] x v[..] + 0 )?
// Every wavefront computes one MFMA both ways (builtin vs hand-placed overlapping registers) many times while its SIMD partner
// keeps the matrix pipe busy; counts mismatching results.   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int iters, int mode, unsigned long long* bad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long nbad = 0;
  d4 bg = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const double a = 1.0 + 0.001 * lane + 0.37 * it, b = 2.0 - 0.003 * lane + 0.11 * it;
    if (wave >= 4 && mode == 1) {            // partner wavefronts: background MFMA chain
#pragma unroll
      for (int i = 0; i < 8; ++i) bg = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, bg, 0, 0, 0);
      continue;
    }
    const d4 z = {0, 0, 0, 0};
    const d4 ref = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 0);
    int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    int o[8];
    asm volatile(
        "v_mov_b32 v16, %8\n\tv_mov_b32 v17, %9\n\tv_mov_b32 v20, %10\n\tv_mov_b32 v21, %11\n\t"
        "s_nop 4\n\t"
        "v_mfma_f64_16x16x4_f64 v[10:17], v[16:17], v[20:21], 0\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        "v_mov_b32 %0, v10\n\tv_mov_b32 %1, v11\n\tv_mov_b32 %2, v12\n\tv_mov_b32 %3, v13\n\t"
        "v_mov_b32 %4, v14\n\tv_mov_b32 %5, v15\n\tv_mov_b32 %6, v16\n\tv_mov_b32 %7, v17\n\t"
        : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7])
        : "v"(alo), "v"(ahi), "v"(blo), "v"(bhi)
        : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double got = __hiloint2double(o[2 * i + 1], o[2 * i]);
      if (got != ref[i]) ++nbad;
    }
  }
  if (bg[0] == 123.456) nbad += 1;
  atomicAdd(bad, nbad);
}

int main() {
  unsigned long long* d; hipMalloc(&d, 8);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, 20000, mode, d);
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%s): mismatching elements %llu\n", mode, mode ? "partner wave busy with MFMAs" : "all waves test", h);
  }
  return 0;
}
