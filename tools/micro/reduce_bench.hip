// Column sums of the Phi pass's per-workgroup partials (G rows of E1 doubles -> E1 sums): what row stride, load width and row
// split make the 23.6 MB read fastest.  Build: hipcc -O3 --offload-arch=gfx950 reduce_bench.hip -o bin/reduce_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void red_a(const double* __restrict__ p, int G, int E1, long stride, double* __restrict__ out) {
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E1) return;
  int per = (G + gridDim.y - 1) / gridDim.y, g0 = blockIdx.y * per, g1 = min(G, g0 + per);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int g = g0;
  for (; g + 3 < g1; g += 4) {
    s0 += __builtin_nontemporal_load(p + (size_t)g * stride + e);
    s1 += __builtin_nontemporal_load(p + (size_t)(g + 1) * stride + e);
    s2 += __builtin_nontemporal_load(p + (size_t)(g + 2) * stride + e);
    s3 += __builtin_nontemporal_load(p + (size_t)(g + 3) * stride + e);
  }
  for (; g < g1; ++g) s0 += p[(size_t)g * stride + e];
  double s = (s0 + s1) + (s2 + s3);
  if (s != 0.0) __hip_atomic_fetch_add(out + e, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef double d2 __attribute__((ext_vector_type(2)));
// two columns per thread (16-byte loads, needs an even stride), ROWS rows per thread all in flight
template <int ROWS, int NT>
__global__ __launch_bounds__(NT) void red_b(const double* __restrict__ p, int G, int E1, long stride, double* __restrict__ out) {
  int e = 2 * (blockIdx.x * NT + threadIdx.x);
  if (e >= E1) return;
  int g0 = blockIdx.y * ROWS;
  d2 v[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    int g = g0 + r;
    v[r] = (g < G) ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + (size_t)g * stride + e)) : d2{0.0, 0.0};
  }
  d2 s = v[0];
#pragma unroll
  for (int r = 1; r < ROWS; ++r) s += v[r];
  if (s.x != 0.0) __hip_atomic_fetch_add(out + e, s.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (e + 1 < E1 && s.y != 0.0) __hip_atomic_fetch_add(out + e + 1, s.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same with the row slices of a workgroup combined through the LDS: one atomic per column and workgroup (blockDim = (64, W))
template <int ROWS, int W>
__global__ __launch_bounds__(64 * W) void red_c(const double* __restrict__ p, int G, int E1, long stride, double* __restrict__ out) {
  __shared__ d2 sh[W][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int e = 2 * (blockIdx.x * 64 + lane);
  int g0 = (blockIdx.y * W + w) * ROWS;
  d2 s = {0.0, 0.0};
  if (e < E1) {
    d2 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int g = g0 + r;
      v[r] = (g < G) ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + (size_t)g * stride + e)) : d2{0.0, 0.0};
    }
    s = v[0];
#pragma unroll
    for (int r = 1; r < ROWS; ++r) s += v[r];
  }
  sh[w][lane] = s;
  __syncthreads();
  if (w == 0 && e < E1) {
#pragma unroll
    for (int k = 1; k < W; ++k) s += sh[k][lane];
    if (s.x != 0.0) __hip_atomic_fetch_add(out + e, s.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e + 1 < E1 && s.y != 0.0) __hip_atomic_fetch_add(out + e + 1, s.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 240, E1 = 12289;
  const long strides[3] = {12289, 12290, 12320};
  double *p, *out;
  CK(hipMalloc(&p, sizeof(double) * 256 * 12320 + 4096));
  CK(hipMalloc(&out, sizeof(double) * 12320));
  std::vector<double> h((size_t)256 * 12320, 1.0);
  CK(hipMemcpy(p, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 5; ++i) { CK(hipMemsetAsync(out, 0, 8 * 12320, 0)); launch(); }
    float best = 1e9, tot = 0;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipMemsetAsync(out, 0, 8 * 12320, 0));
      CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; if (ms < best) best = ms;
    }
    double o0, o1; CK(hipMemcpy(&o0, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&o1, out + E1 - 1, 8, hipMemcpyDeviceToHost));
    printf("%-44s mean %6.2f us  best %6.2f us   (sum[0] %.0f sum[last] %.0f)\n", name, tot / 20 * 1e3, best * 1e3, o0, o1);
  };
  char nm[128];
  for (int gs : {8, 16, 30, 60}) {
    snprintf(nm, 128, "a: 8 B loads, stride 12289, gsplit %d", gs);
    time(nm, [&] { hipLaunchKernelGGL(red_a, dim3((E1 + 255) / 256, gs), dim3(256), 0, 0, p, G, E1, 12289L, out); });
  }
  for (long st : {12290L, 12320L}) {
    snprintf(nm, 128, "b: 16 B loads, stride %ld, 8 rows/thread, 256", st);
    time(nm, [&] { hipLaunchKernelGGL((red_b<8, 256>), dim3((E1 / 2 + 256) / 256, (G + 7) / 8), dim3(256), 0, 0, p, G, E1, st, out); });
    snprintf(nm, 128, "b: 16 B loads, stride %ld, 4 rows/thread, 256", st);
    time(nm, [&] { hipLaunchKernelGGL((red_b<4, 256>), dim3((E1 / 2 + 256) / 256, (G + 3) / 4), dim3(256), 0, 0, p, G, E1, st, out); });
    snprintf(nm, 128, "b: 16 B loads, stride %ld, 16 rows/thread, 256", st);
    time(nm, [&] { hipLaunchKernelGGL((red_b<16, 256>), dim3((E1 / 2 + 256) / 256, (G + 15) / 16), dim3(256), 0, 0, p, G, E1, st, out); });
    snprintf(nm, 128, "c: 16 B, stride %ld, 8 rows x 4 waves + LDS", st);
    time(nm, [&] { hipLaunchKernelGGL((red_c<8, 4>), dim3((E1 / 2 + 64) / 64, (G + 31) / 32), dim3(256), 0, 0, p, G, E1, st, out); });
    snprintf(nm, 128, "c: 16 B, stride %ld, 4 rows x 8 waves + LDS", st);
    time(nm, [&] { hipLaunchKernelGGL((red_c<4, 8>), dim3((E1 / 2 + 64) / 64, (G + 31) / 32), dim3(512), 0, 0, p, G, E1, st, out); });
    snprintf(nm, 128, "c: 16 B, stride %ld, 8 rows x 8 waves + LDS", st);
    time(nm, [&] { hipLaunchKernelGGL((red_c<8, 8>), dim3((E1 / 2 + 64) / 64, (G + 63) / 64), dim3(512), 0, 0, p, G, E1, st, out); });
    snprintf(nm, 128, "c: 16 B, stride %ld, 15 rows x 16 waves + LDS", st);
    time(nm, [&] { hipLaunchKernelGGL((red_c<15, 16>), dim3((E1 / 2 + 64) / 64, (G + 239) / 240), dim3(1024), 0, 0, p, G, E1, st, out); });
  }
  return 0;
}
