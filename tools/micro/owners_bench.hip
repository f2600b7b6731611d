// Micro-benchmark of the owner phase of the tile-sort Phi kernels (phi_sort.hpp ps_own_cell): one 1024-thread workgroup per CU walks
// a synthetic sorted tile REPS times.  Variants separate the VALU work, the LDS reads and the run-length imbalance.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/micro/owners_bench.hip -o tools/micro/bin/owners_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../asvgp_amd/csrc/asvgp_common.hpp"
namespace asvgp {
void set_error(const char*, ...) {}
int check_launch(const char*) { return 0; }
}  // namespace asvgp
#include "../../asvgp_amd/csrc/phi_tables.hpp"
#include "../../asvgp_amd/csrc/phi_moments.hpp"
#include "../../asvgp_amd/csrc/phi_sort.hpp"
using namespace asvgp;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int K = 4;

// MODE 0: ps_own_cell as in the product; 1: no LDS reads (points from registers); 2: LDS reads only (sum of the loaded values);
//      3: one fused loop over both cells (a point of A and a point of B per iteration)
template <int MODE, int T>
__global__ __launch_bounds__(1024) void owners_kernel(const double2* pts, const unsigned* cnt_g, const unsigned* off_g, int reps, double* out) {
  extern __shared__ double lds[];
  double2* buf = reinterpret_cast<double2*>(lds);
  const int tid = threadIdx.x;
  for (int i = tid; i < T; i += 1024) buf[i] = pts[i];
  if (tid == 0) buf[T] = make_double2(0.0, 0.0);
  const unsigned nA = cnt_g[tid], nB = cnt_g[tid + 1024], oA = off_g[tid], oB = off_g[tid + 1024];
  __syncthreads();
  double SA[2 * K], TA[K + 1], SB[2 * K], TB[K + 1];
  for (int p = 0; p < 2 * K; ++p) { SA[p] = 0; SB[p] = 0; }
  for (int p = 0; p <= K; ++p) { TA[p] = 0; TB[p] = 0; }
  for (int r = 0; r < reps; ++r) {
    if constexpr (MODE == 0) {
      ps_own_cell<K, T>(buf, nA, oA, SA, TA);
      ps_own_cell<K, T>(buf, nB, oB, SB, TB);
    } else if constexpr (MODE == 1) {
      const unsigned nmaxA = (ps_wave_max_u32(nA) + 1u) & ~1u, nmaxB = (ps_wave_max_u32(nB) + 1u) & ~1u;
      double2 p = make_double2(1e-3 * tid, 0.5);
      for (unsigned j = 0; j < nmaxA; ++j) { ps_acc<K>(p.x, p.y, SA, TA); p.x += 1e-9; }
      for (unsigned j = 0; j < nmaxB; ++j) { ps_acc<K>(p.x, p.y, SB, TB); p.x += 1e-9; }
    } else if constexpr (MODE == 2) {
      const unsigned nmaxA = (ps_wave_max_u32(nA) + 1u) & ~1u, nmaxB = (ps_wave_max_u32(nB) + 1u) & ~1u;
      for (unsigned j = 0; j < nmaxA; ++j) { const double2 p = buf[j < nA ? oA + j : (unsigned)T]; SA[0] += p.x; TA[0] += p.y; }
      for (unsigned j = 0; j < nmaxB; ++j) { const double2 p = buf[j < nB ? oB + j : (unsigned)T]; SB[0] += p.x; TB[0] += p.y; }
    } else {
      const unsigned nm = ps_wave_max_u32(nA > nB ? nA : nB);
      double2 p = buf[nA > 0 ? oA : (unsigned)T], q = buf[nB > 0 ? oB : (unsigned)T];
      for (unsigned j = 0; j < nm; ++j) {
        const double2 p1 = buf[(j + 1 < nA) ? oA + j + 1 : (unsigned)T], q1 = buf[(j + 1 < nB) ? oB + j + 1 : (unsigned)T];
        ps_acc<K>(p.x, p.y, SA, TA);
        ps_acc<K>(q.x, q.y, SB, TB);
        p = p1; q = q1;
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  double keep = 0;
  for (int p = 0; p < 2 * K; ++p) keep += SA[p] + SB[p];
  for (int p = 0; p <= K; ++p) keep += TA[p] + TB[p];
  out[blockIdx.x * 1024 + tid] = keep;
}

template <typename F> static float time_us(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1000.f / reps;
}

template <int T> static void run(const char* tag, int dist) {
  const int ncell = 2044;
  std::mt19937_64 rng(7);
  std::vector<unsigned> cnt(2048, 0), off(2048, 0);
  if (dist == 0) { std::uniform_int_distribution<int> U(0, ncell - 1); for (int i = 0; i < T; ++i) cnt[U(rng)]++; }
  else { for (int i = 0; i < T; ++i) cnt[i % ncell]++; }
  unsigned acc = 0;
  for (int c = 0; c < 2048; ++c) { off[c] = acc; acc += cnt[c]; }
  std::vector<double2> pts(T);
  std::uniform_real_distribution<double> Us(-0.5, 0.5);
  for (int i = 0; i < T; ++i) pts[i] = make_double2(Us(rng), Us(rng));
  double2* dp; unsigned *dc, *dof; double* dout;
  CK(hipMalloc(&dp, T * 16)); CK(hipMalloc(&dc, 2048 * 4)); CK(hipMalloc(&dof, 2048 * 4)); CK(hipMalloc(&dout, 256 * 1024 * 8));
  CK(hipMemcpy(dp, pts.data(), T * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc, cnt.data(), 2048 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dof, off.data(), 2048 * 4, hipMemcpyHostToDevice));
  const int reps = 64, G = 256;
  const size_t lds = (size_t)(T + 1) * 16;
  float t[4];
  auto go = [&](auto kern, int i) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float t1 = time_us([&] { hipLaunchKernelGGL(kern, dim3(G), dim3(1024), lds, 0, dp, dc, dof, reps, dout); }, 5);
    const float t0 = time_us([&] { hipLaunchKernelGGL(kern, dim3(G), dim3(1024), lds, 0, dp, dc, dof, 0, dout); }, 5);
    t[i] = (t1 - t0) / reps;
  };
  go(owners_kernel<0, T>, 0); go(owners_kernel<1, T>, 1); go(owners_kernel<2, T>, 2); go(owners_kernel<3, T>, 3);
  // per tile: us; scaled to N = 10M over 256 CUs: tiles = 39063 / T
  const double tiles = 39063.0 / T;
  printf("%s T=%d dist=%s: per tile  product %.3f us | VALU only %.3f | LDS reads only %.3f | fused A+B loop %.3f   -> x %.2f tiles = %.1f / %.1f / %.1f / %.1f us per 10M points\n",
         tag, T, dist ? "balanced" : "poisson", t[0], t[1], t[2], t[3], tiles, t[0] * tiles, t[1] * tiles, t[2] * tiles, t[3] * tiles);
  hipFree(dp); hipFree(dc); hipFree(dof); hipFree(dout);
}

int main() {
  run<6144>("owners", 0); run<6144>("owners", 1);
  run<4096>("owners", 0); run<4096>("owners", 1);
  run<8192>("owners", 0); run<8192>("owners", 1);
  return 0;
}
