// Probe: operand / result lane layout of v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4 per wave) on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f64_4x4x4_layout.hip -o tools/micro/bin/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {
  const int l = threadIdx.x;
  for (int t = 0; t < 64; ++t) {
    const double a = (double)(l + 1);            // A operand: lane id + 1
    const double b = (l == t) ? 1.0 : 0.0;       // B operand: one-hot at lane t
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[t * 64 + l] = d;
  }
  for (int t = 0; t < 64; ++t) {                 // and the other way round: A one-hot, B = lane id + 1
    const double a = (l == t) ? 1.0 : 0.0;
    const double b = (double)(l + 1);
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[4096 + t * 64 + l] = d;
  }
}
int main() {
  double* d; hipMalloc(&d, 8192 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  static double h[8192];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("B one-hot at lane t: result lanes (value = A lane + 1)\n");
  for (int t = 0; t < 20; ++t) {
    printf("t=%2d:", t);
    for (int l = 0; l < 64; ++l) if (h[t * 64 + l] != 0.0) printf(" D[%d]=A%d", l, (int)h[t * 64 + l] - 1);
    printf("\n");
  }
  printf("A one-hot at lane t: result lanes (value = B lane + 1)\n");
  for (int t = 0; t < 20; ++t) {
    printf("t=%2d:", t);
    for (int l = 0; l < 64; ++l) if (h[4096 + t * 64 + l] != 0.0) printf(" D[%d]=B%d", l, (int)h[4096 + t * 64 + l] - 1);
    printf("\n");
  }
  return 0;
}
