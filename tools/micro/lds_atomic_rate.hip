// Micro-benchmark: LDS atomic throughput per CU by operand type (random addresses, 1024 threads per workgroup).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomic_rate.hip -o gpurun_out/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T> __device__ __forceinline__ void lds_add(T* p, T v);
template <> __device__ __forceinline__ void lds_add<double>(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <> __device__ __forceinline__ void lds_add<float>(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <> __device__ __forceinline__ void lds_add<unsigned long long>(unsigned long long* p, unsigned long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <> __device__ __forceinline__ void lds_add<unsigned>(unsigned* p, unsigned v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename T, int MODE>
__global__ __launch_bounds__(1024) void k(T* out, int words, int iters, unsigned long long* cyc) {
  extern __shared__ unsigned char raw[];
  T* lds = reinterpret_cast<T*>(raw);
  for (int i = threadIdx.x; i < words; i += 1024) lds[i] = T(0);
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    unsigned a = (s >> 8);
    T v = T(1);
    if (MODE == 0) {   // 20 adds at 20 unrelated offsets (like the band scatter: 5 rows x different planes)
#pragma unroll
      for (int j = 0; j < 20; ++j) lds_add<T>(lds + ((a + j * 2049u) & (unsigned)(words - 1)), v);
    } else if (MODE == 2) {   // random cells, but the 16 lanes of a group hit 16 different (address mod 16): bank-conflict-free
      const unsigned a16 = (a & ~15u) | ((threadIdx.x * 5u + (s >> 28)) & 15u);   // (x5: a permutation of 0..15 across the group)
#pragma unroll
      for (int j = 0; j < 20; ++j) lds_add<T>(lds + ((a16 + j * 2048u) & (unsigned)(words - 1)), v);
    } else if (MODE == 3) {   // random cells, (address mod 32) distinct within each half-wave of 32 lanes
      const unsigned a32 = (a & ~31u) | ((threadIdx.x * 5u + (s >> 27)) & 31u);
#pragma unroll
      for (int j = 0; j < 20; ++j) lds_add<T>(lds + ((a32 + j * 2048u) & (unsigned)(words - 1)), v);
    } else if (MODE == 4) {   // random cells, (address mod 64) distinct over the whole wave
      const unsigned a64 = (a & ~63u) | ((threadIdx.x * 5u + (s >> 26)) & 63u);
#pragma unroll
      for (int j = 0; j < 20; ++j) lds_add<T>(lds + ((a64 + j * 2048u) & (unsigned)(words - 1)), v);
    } else if (MODE == 5) {   // consecutive addresses (lane l -> base + l): the ideal
      const unsigned ab = (a & ~63u) | (threadIdx.x & 63u);
#pragma unroll
      for (int j = 0; j < 20; ++j) lds_add<T>(lds + ((ab + j * 2048u) & (unsigned)(words - 1)), v);
    } else {           // plain stores for reference
#pragma unroll
      for (int j = 0; j < 20; ++j) lds[(a + j * 2049u) & (unsigned)(words - 1)] = v;
    }
  }
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  T acc = T(0);
  for (int i = threadIdx.x; i < words; i += 1024) acc += lds[i];
  if (acc == T(123457)) out[0] = acc;
}

template <typename T, int MODE> void run(const char* name, int words) {
  unsigned long long* cyc; T* out;
  hipMalloc(&cyc, 256 * 8); hipMalloc(&out, 64);
  const int iters = 200;
  size_t bytes = (size_t)words * sizeof(T);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<T, MODE>), dim3(256), dim3(1024), bytes, 0, out, words, iters, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<T, MODE>), dim3(256), dim3(1024), bytes, 0, out, words, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
  double wave_instr = 16.0 * iters * 20;   // per CU
  // s_memtime counts at 100 MHz; use the event time and 2.4 GHz for cycles
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-28s words %6d: %8.1f us  -> %6.2f cycles per wave-instruction (memtime ticks %.0f)\n", name, words, ms * 1e3, cycles / wave_instr, mean);
}

int main() {
  run<double, 0>("ds_add_f64", 8192);
  run<unsigned long long, 0>("ds_add_u64", 8192);
  run<float, 0>("ds_add_f32", 8192);
  run<unsigned, 0>("ds_add_u32", 8192);
  run<unsigned long long, 2>("ds_add_u64 bank-distinct per 16 lanes", 8192);
  run<unsigned long long, 3>("ds_add_u64 distinct mod 32 per 32 lanes", 8192);
  run<unsigned long long, 4>("ds_add_u64 distinct mod 64 per wave", 8192);
  run<unsigned long long, 5>("ds_add_u64 consecutive", 8192);
  run<double, 1>("ds_write_b64", 8192);
  run<float, 1>("ds_write_b32", 8192);
  return 0;
}
