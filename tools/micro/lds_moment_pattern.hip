// Micro-benchmark: the LDS traffic of the centred-moment Phi kernel in isolation (1 workgroup of 1024 threads per CU):
// per point one ds_add_u32 (count) + 8 ds_add_u64 at plane stride 2048 cells + 5 ds_add_u64 at consecutive addresses (Phi y),
// random cells, with FILL dependent fp64 FMAs per atomic in between (0 = LDS only).  Prints cycles per LDS wave-instruction.
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_moment_pattern.hip -o gpurun_out/lds_moment_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FILL>
__global__ __launch_bounds__(1024) void k(double* out, int iters, unsigned long long* cyc, unsigned long long* wall) {
  extern __shared__ unsigned long long lds[];
  constexpr int CS = 2048;
  unsigned long long* planes = lds;                     // [8][CS]
  unsigned* cnt = reinterpret_cast<unsigned*>(planes + 8 * CS);
  unsigned long long* rhs = reinterpret_cast<unsigned long long*>(cnt + CS);   // [2048]
  for (int i = threadIdx.x; i < (8 * CS * 8 + CS * 4 + 2048 * 8) / 8; i += 1024) lds[i] = 0;
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  double acc = 1.0 + threadIdx.x * 1e-9, x = 0.999999;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const unsigned c = (s >> 8) % 2044u;
    __hip_atomic_fetch_add(cnt + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
      for (int f = 0; f < FILL; ++f) acc = fma(acc, x, 1e-3);
      __hip_atomic_fetch_add(planes + p * CS + c, (unsigned long long)(FILL ? __double_as_longlong(acc) & 0xffff : 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
      for (int f = 0; f < FILL; ++f) acc = fma(acc, x, 1e-3);
      __hip_atomic_fetch_add(rhs + c + 4 - i, (unsigned long long)(FILL ? __double_as_longlong(acc) & 0xffff : 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = r1 - r0; }
  if (acc == 123.456) out[0] = acc;
}
template <int FILL> void run() {
  unsigned long long *cyc, *wall; double* out;
  hipMalloc(&cyc, 256 * 8); hipMalloc(&wall, 256 * 8); hipMalloc(&out, 64);
  const int iters = 600;
  size_t bytes = 8 * 2048 * 8 + 2048 * 4 + 2048 * 8;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<FILL>, dim3(256), dim3(1024), bytes, 0, out, iters, cyc, wall);
  hipDeviceSynchronize();
  unsigned long long h[256], w[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(w, wall, sizeof w, hipMemcpyDeviceToHost);
  double mc = 0, mw = 0; for (int i = 0; i < 256; ++i) { mc += h[i]; mw += w[i]; } mc /= 256; mw /= 256;
  const double wi = 16.0 * iters * 14;   // LDS wave-instructions per CU
  printf("fill %2d fp64 FMAs per atomic: %8.0f shader cycles (clock %.2f GHz) -> %6.2f cycles per LDS wave-instruction, %6.2f per point-wave; VALU alone would be %.0f cycles per SIMD\n",
         FILL, mc, mc / (mw * 10.0) , mc / wi, mc / (16.0 * iters), 4.0 * iters * 13 * FILL * 4.0);
}
int main() { run<0>(); run<2>(); run<4>(); run<8>(); run<12>(); return 0; }
