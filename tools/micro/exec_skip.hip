// Does a wave64 fp64 VALU instruction cost less when whole 16/32-lane groups are masked off in EXEC?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(1024) void k(int active, int iters, double* out) {
  const int lane = threadIdx.x & 63;
  double a0 = 1.0 + lane, a1 = 2.0, a2 = 3.0, a3 = 4.0, a4 = 0.5, a5 = 0.25, a6 = 0.1, a7 = 0.2, x = 1.0000001;
  if (lane < active) {
    for (int i = 0; i < iters; ++i) {
      a0 = fma(a0, x, 1e-9); a1 = fma(a1, x, 1e-9); a2 = fma(a2, x, 1e-9); a3 = fma(a3, x, 1e-9);
      a4 = fma(a4, x, 1e-9); a5 = fma(a5, x, 1e-9); a6 = fma(a6, x, 1e-9); a7 = fma(a7, x, 1e-9);
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  double* d; CK(hipMalloc(&d, 256 * 1024 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int active : {64, 48, 32, 16, 8, 1}) {
    for (int r = 0; r < 2; ++r) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, active, 4000, d);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) printf("active lanes %2d: %.1f us  (%.2f cycles per wave fp64 FMA per SIMD at 2.4 GHz)\n", active, ms * 1e3, ms * 1e-3 * 2.4e9 / (4000.0 * 8 * 4));
    }
  }
  return 0;
}
