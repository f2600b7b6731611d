// Micro-benchmark: issue cost of fp64 / fp32 / int VALU instructions on one CU (1024-thread workgroup = 4 waves per SIMD, and 256 threads = 1 per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_rate.hip -o tools/micro/bin/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, unsigned long long* cyc, int iters) {
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3 + i;
  double s = 1.0 + threadIdx.x * 1e-9, t = 0.5;
  float f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 1e-3f + i;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 16 independent v_fma_f64
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = fma(a[i], s, t);
    } else if (MODE == 1) {   // 16 independent v_add_f64
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] + s;
    } else if (MODE == 2) {   // 16 independent v_mul_f64
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] * s;
    } else if (MODE == 3) {   // 16 independent v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = fmaf(f[i], (float)s, (float)t);
    } else {                  // the moment update of phi_sort.hpp: 3 mul + 13 fma/add on one point
      double x = a[15] * 1e-3, y = a[14];
      double s2 = x * x, s3 = s2 * x, s4 = s2 * s2;
      a[0] += x; a[1] += s2; a[2] += s3; a[3] += s4;
      a[4] = fma(s4, x, a[4]); a[5] = fma(s4, s2, a[5]); a[6] = fma(s4, s3, a[6]); a[7] = fma(s4, s4, a[7]);
      a[8] += y; a[9] = fma(y, x, a[9]); a[10] = fma(y, s2, a[10]); a[11] = fma(y, s3, a[11]); a[12] = fma(y, s4, a[12]);
    }
    asm volatile("" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  double r = 0; float rf = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { r += a[i]; rf += f[i]; }
  if (r + rf == 1.2345) out[0] = r;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = t2 - t0; }
}
template <int MODE> void run(const char* name, int threads) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64); hipMalloc(&cyc, 256 * 16);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[512];
  hipMemcpy(h, cyc, 256 * 16, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0;
  for (int i = 0; i < 256; ++i) { m0 += h[2 * i]; m1 += h[2 * i + 1]; }
  m0 /= 256; m1 /= 256;
  const int waves_per_simd = threads / 256;
  printf("%-28s %4d threads: wave 0 alone %.1f cyc per instr; all waves done: %.2f cyc per instr per wave (%d waves/SIMD)\n", name, threads,
         m0 / iters / 16.0, m1 / iters / 16.0 / waves_per_simd, waves_per_simd);
}
int main() {
  for (int th : {256, 1024}) {
    run<0>("v_fma_f64 x16", th); run<1>("v_add_f64 x16", th); run<2>("v_mul_f64 x16", th); run<3>("v_fma_f32 x16", th); run<4>("moment update (16 ops)", th);
  }
  return 0;
}
