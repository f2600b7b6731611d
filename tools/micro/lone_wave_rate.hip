// Micro-benchmark: how fast ONE wave retires instructions when it has its SIMD (and its CU) to itself - the situation of the single-wave
// recurrences (band_ops.hip) and of the chain workgroups of the ELBO launch - and whether the rest of the chip being busy changes it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lone_wave_rate.hip -o tools/micro/bin/lone_wave_rate && tools/micro/bin/lone_wave_rate
// Modes: 0 = 16 independent v_fma_f64, 1 = one dependent chain of v_fma_f64, 2 = 16 independent v_add_u32, 3 = 16 independent v_fma_f32,
//        4 = broadcast ds_read_b64 + one dependent v_fma_f64 each (the operand pattern of the recurrences).
// Workgroup shapes: 64 threads (one wave), 128 (two waves, two SIMDs), 320 with waves 1..3 idle (waves 0 and 4 share a SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(320) void k(double* out, int iters, unsigned active_mask) {
  __shared__ double lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = 1.0 + threadIdx.x * 1e-6;
  __syncthreads();
  const int wv = threadIdx.x / 64;
  if (!((active_mask >> wv) & 1u)) return;
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 1.0 + i * 1e-3;
  float f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = 1.0f + i * 1e-3f;
  unsigned u[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) u[i] = i + threadIdx.x;
  const double s = 1.0 + 1e-9, t = 1e-12;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(s), "v"(t));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(3u));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"((float)s), "v"((float)t));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double x = *(volatile double*)&lds[(it + i) & 63];     // uniform address: broadcast read
        a[i] = fma(a[i], x, t);
      }
    }
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a[i] + f[i] + u[i];
  if (r == 1.2345) out[0] = r;
}

// keeps the other CUs busy with fp64 work for `iters` iterations
__global__ __launch_bounds__(256) void busy(double* out, int iters) {
  double a = threadIdx.x, b = 1.0 + 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a = fma(a, b, 1e-12);
  }
  if (a == 1.2345) out[1] = a;
}

template <int MODE> void run(const char* name, int instr_per_iter, double* out, hipStream_t sa, hipStream_t sb, bool with_busy) {
  const int iters = 200000;
  struct { int threads; unsigned mask; const char* what; } shapes[] = {
      {64, 1u, "one wave"}, {128, 3u, "two waves, two SIMDs"}, {320, 0x11u, "two waves, ONE SIMD"}, {256, 0xfu, "four waves, four SIMDs"}};
  for (auto& sh : shapes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(sh.threads), 0, sa, out, 1000, sh.mask);
    CK(hipStreamSynchronize(sa));
    if (with_busy) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, sb, out, 1000000);   // one wave per SIMD on every CU, ~50 ms
    CK(hipEventRecord(e0, sa));
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(sh.threads), 0, sa, out, iters, sh.mask);
    CK(hipEventRecord(e1, sa));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipDeviceSynchronize());
    const double ns = ms * 1e6 / ((double)iters * instr_per_iter);
    printf("%-34s %-24s %s: %.2f ns per instruction and wave = %.1f cycles at 2.4 GHz\n", name, sh.what, with_busy ? "chip busy" : "chip idle", ns, ns * 2.4);
  }
}

int main() {
  double* out;
  CK(hipMalloc(&out, 64));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  for (int busy_on = 0; busy_on < 2; ++busy_on) {
    run<0>("16 independent v_fma_f64", 16, out, sa, sb, busy_on);
    run<1>("dependent chain of v_fma_f64", 16, out, sa, sb, busy_on);
    run<2>("16 independent v_add_u32", 16, out, sa, sb, busy_on);
    run<3>("16 independent v_fma_f32", 16, out, sa, sb, busy_on);
    run<4>("8 x (ds_read_b64 + v_fma_f64)", 16, out, sa, sb, busy_on);
  }
  return 0;
}
