// Stand-alone harness for the tile-sort Phi kernel (asvgp_amd/csrc/phi_sort.hpp): per-stage ablation timings, in-kernel phase
// stamps, a CPU check of the band / rhs statistics, and the read-only stream ceiling (16 B/point) of the same launch shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/micro/phi_sort_bench.hip -o tools/micro/bin/phi_sort_bench
//   tools/micro/bin/phi_sort_bench [N=10000000] [dist: 0 uniform | 1 sorted | 2 clustered] [check=1]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
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

// read-only stream: DEPTH 16-B loads of x and of y per lane in flight, 1024-thread workgroups, one per CU
template <int DEPTH, bool NT>
__global__ __launch_bounds__(1024) void stream_kernel(const double* x, const double* y, long N, long ppb, double* sink) {
  typedef double v2 __attribute__((ext_vector_type(2)));
  const long beg = (long)blockIdx.x * ppb;
  long end = beg + ppb; if (end > N) end = N;
  const v2* x2 = reinterpret_cast<const v2*>(x) + (beg >> 1);
  const v2* y2 = reinterpret_cast<const v2*>(y) + (beg >> 1);
  const int npair = (int)((end - beg) >> 1);
  double acc = 0.0;
  for (int base = 0; base < npair; base += DEPTH * 1024) {
    v2 xv[DEPTH], yv[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      int u = base + d * 1024 + threadIdx.x;
      u = u < npair - 1 ? u : npair - 1;
      if (NT) { xv[d] = __builtin_nontemporal_load(x2 + u); yv[d] = __builtin_nontemporal_load(y2 + u); }
      else { xv[d] = x2[u]; yv[d] = y2[u]; }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += xv[d].x + xv[d].y + yv[d].x + yv[d].y;
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

template <typename F> static float time_us(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
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

template <int TP, int ABL, int PF = 0> static float run_abl(PsArgs a, int G) {
  auto kern = phi_sort_kernel<K, TP, ABL, PF>;
  size_t lds = ps_lds_bytes<K, TP>();
  if (ps_epilogue_bytes<K>() > lds) lds = ps_epilogue_bytes<K>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return time_us([&] { hipLaunchKernelGGL(kern, dim3(G), dim3(PS_THREADS), lds, 0, a); }, 20);
}

template <int TP, int TS> static float run_ts(PsArgs a, int G) {
  auto kern = phi_sort_kernel<K, TP, 0, 1, TS>;
  size_t lds = ps_lds_bytes<K, TP, TS>();
  if (ps_epilogue_bytes<K>() > lds) lds = ps_epilogue_bytes<K>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return time_us([&] { hipLaunchKernelGGL(kern, dim3(G), dim3(PS_THREADS), lds, 0, a); }, 20);
}

template <int TP> static void run_all(PsArgs a, int G, long N, const char* tag) {
  const float t1 = run_abl<TP, 1>(a, G), t2 = run_abl<TP, 2>(a, G), t3 = run_abl<TP, 3>(a, G), t4 = run_abl<TP, 4>(a, G),
              t5 = run_abl<TP, 5>(a, G), t0 = run_abl<TP, 0>(a, G);
  const float l5 = run_abl<TP, 5, 1>(a, G), l0 = run_abl<TP, 0, 1>(a, G);
  printf("%s TP=%d  loads+search %.1f | +rank %.1f | +scan %.1f | +scatter %.1f | +owners %.1f | full %.1f us  (%.2f TB/s, %.3f of 8 TB/s)   [late prefetch: +owners %.1f full %.1f]\n",
         tag, TP, t1, t2, t3, t4, t5, t0, 16.0 * N / t0 * 1e-6, 16.0 * N / t0 * 1e-6 / 8.0, l5, l0);
}

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 10000000L;
  const int dist = argc > 2 ? atoi(argv[2]) : 0;
  const int check = argc > 3 ? atoi(argv[3]) : 1;
  const int M = 2048, n_mesh = M - K + 1, ncells = n_mesh - 1;
  std::vector<double> mesh(n_mesh), x(N), y(N);
  const double a0 = 0.0, b0 = 1.0;
  const double step = (b0 - a0) / (double)(n_mesh - 1);
  for (int i = 0; i < n_mesh; ++i) { volatile double t = (double)i * step; mesh[i] = t + a0; }
  mesh[n_mesh - 1] = b0;
  const double delta = mesh[1] - mesh[0];
  std::mt19937_64 rng(1234);
  std::uniform_real_distribution<double> U(1e-9, 1.0 - 1e-9);
  std::normal_distribution<double> G01(0.0, 1.0);
  for (long i = 0; i < N; ++i) {
    double v = U(rng);
    if (dist == 2) { v = 0.5 + 0.08 * G01(rng); v = std::min(std::max(v, 1e-9), 1.0 - 1e-9); }
    x[i] = v;
  }
  // a few awkward points: on knots, at the ends (planted before the sort: a time series stays one)
  if (N > 100) { x[5] = mesh[17]; x[6] = mesh[1000]; x[7] = a0; x[8] = b0; x[9] = mesh[n_mesh - 2]; }
  if (dist == 1) std::sort(x.begin(), x.end());
  for (long i = 0; i < N; ++i) y[i] = std::sin(20.0 * x[i]) + 0.1 * G01(rng);

  double *dx, *dy, *dmesh, *dpart, *dsink;
  unsigned long long* dstamps;
  const int G = 256;
  const size_t E1 = (size_t)(K + 2) * M + 1;
  CK(hipMalloc(&dx, N * 8 + 64)); CK(hipMalloc(&dy, N * 8 + 64)); CK(hipMalloc(&dmesh, n_mesh * 8));
  CK(hipMalloc(&dpart, G * E1 * 8)); CK(hipMalloc(&dsink, 64)); CK(hipMalloc(&dstamps, G * 8 * 8));
  CK(hipMemcpy(dx, x.data(), N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dy, y.data(), N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmesh, mesh.data(), n_mesh * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dpart, 0, G * E1 * 8));

  long ppb = (N + G - 1) / G;
  ppb = (ppb + 127) & ~127L;   // whole rows of 64 pairs: only the last workgroup of the grid has an incomplete row
  PsArgs a;
  a.x = dx; a.y = dy; a.N = N; a.mesh_g = dmesh; a.n_mesh = n_mesh; a.inv_delta = 1.0 / delta; a.M = M; a.step = step; a.m0 = mesh[0]; a.m_last = mesh[n_mesh - 1];
  a.smax_fast = 0.5 - (16.0 * DBL_EPSILON * std::max(std::fabs(a0), std::fabs(b0)) / delta + 1e-12);
  a.partials = dpart; a.ppb = ppb; a.zero_ptr = nullptr; a.zero_n = 0; a.stamps = dstamps; a.stamps_wave = 0; a.ranges = nullptr;
  printf("N=%ld dist=%d M=%d ppb=%ld lds(TP=8)=%zu\n", N, dist, M, ppb, ps_lds_bytes<K, 8>());

  // ---- stream ceilings
  {
    const float s1 = time_us([&] { hipLaunchKernelGGL((stream_kernel<1, true>), dim3(G), dim3(1024), 0, 0, dx, dy, N, ppb, dsink); }, 20);
    const float s2 = time_us([&] { hipLaunchKernelGGL((stream_kernel<2, true>), dim3(G), dim3(1024), 0, 0, dx, dy, N, ppb, dsink); }, 20);
    const float s4 = time_us([&] { hipLaunchKernelGGL((stream_kernel<4, true>), dim3(G), dim3(1024), 0, 0, dx, dy, N, ppb, dsink); }, 20);
    const float s8 = time_us([&] { hipLaunchKernelGGL((stream_kernel<8, true>), dim3(G), dim3(1024), 0, 0, dx, dy, N, ppb, dsink); }, 20);
    const float p4 = time_us([&] { hipLaunchKernelGGL((stream_kernel<4, false>), dim3(G), dim3(1024), 0, 0, dx, dy, N, ppb, dsink); }, 20);
    const float p8 = time_us([&] { hipLaunchKernelGGL((stream_kernel<8, false>), dim3(G), dim3(1024), 0, 0, dx, dy, N, ppb, dsink); }, 20);
    const long ppb2 = ((N + 511) / 512 + 1) & ~1L;
    const float w4 = time_us([&] { hipLaunchKernelGGL((stream_kernel<4, true>), dim3(512), dim3(1024), 0, 0, dx, dy, N, ppb2, dsink); }, 20);
    printf("stream ceiling (read-only 16 B/pt, 256 x 1024 threads): nt depth 1/2/4/8 = %.1f / %.1f / %.1f / %.1f us; plain depth 4/8 = %.1f / %.1f us; 512 WGs nt depth 4 = %.1f us\n",
           s1, s2, s4, s8, p4, p8, w4);
    printf("   best = %.2f TB/s\n", 16.0 * N / std::min({s1, s2, s4, s8, p4, p8, w4}) * 1e-6);
  }

  printf("product (late prefetch): TP=6 without / with the time-series front loop: %.1f / %.1f us;  TP=4: %.1f / %.1f us\n",
         run_ts<6, 0>(a, G), run_ts<6, 1>(a, G), run_ts<4, 0>(a, G), run_ts<4, 1>(a, G));
#ifndef PS_QUICK
  run_all<8>(a, G, N, "sort");
  run_all<6>(a, G, N, "sort");
  run_all<4>(a, G, N, "sort");
#else
  printf("quick: TP=8 late full %.1f us, TP=4 early full %.1f us, TP=4 late full %.1f us\n", run_abl<8, 0, 1>(a, G), run_abl<4, 0, 0>(a, G), run_abl<4, 0, 1>(a, G));
#endif

  // ---- phase stamps
  auto stamps = [&](auto kern, size_t lds, const char* tag, int wave) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PsArgs b = a; b.stamps_wave = wave;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(G), dim3(PS_THREADS), lds, 0, b);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(G * 8);
    CK(hipMemcpy(st.data(), dstamps, G * 8 * 8, hipMemcpyDeviceToHost));
    double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = 0; g < G; ++g) for (int i = 0; i < 8; ++i) m[i] += (double)st[g * 8 + i] / G;
    printf("%s wave %2d cycles over all tiles: P1 %.0f (landed+search %.0f, ranks back %.0f, rest = barrier) | P2 scan %.0f | P3 scatter %.0f | P4 owners %.0f (of which heavy-cell pass %.0f) | epilogue %.0f | total %.0f\n",
           tag, wave, m[0], m[5], m[6], m[1], m[2], m[3], m[7], m[4], m[0] + m[1] + m[2] + m[3] + m[4]);
  };
  for (int w : {0, 7, 15}) stamps(phi_sort_kernel<K, 4, 9, 1, 1>, std::max(ps_lds_bytes<K, 4, 1>(), ps_epilogue_bytes<K>()), "TP=4 late TS", w);
  for (int w : {0, 5, 15}) stamps(phi_sort_kernel<K, 4, 9, 0>, std::max(ps_lds_bytes<K, 4>(), ps_epilogue_bytes<K>()), "TP=4 early", w);
#ifndef PS_QUICK
  stamps(phi_sort_kernel<K, 4, 9, 1>, std::max(ps_lds_bytes<K, 4>(), ps_epilogue_bytes<K>()), "TP=4 late ", 0);
  for (int w : {0, 15}) stamps(phi_sort_kernel<K, 8, 9, 1>, std::max(ps_lds_bytes<K, 8>(), ps_epilogue_bytes<K>()), "TP=8 late ", w);
  for (int w : {0, 15}) stamps(phi_sort_kernel<K, 6, 9, 1>, std::max(ps_lds_bytes<K, 6>(), ps_epilogue_bytes<K>()), "TP=6 late ", w);
#else
  for (int w : {0, 15}) stamps(phi_sort_kernel<K, 8, 9, 1>, std::max(ps_lds_bytes<K, 8>(), ps_epilogue_bytes<K>()), "TP=8 late ", w);
#endif

  // ---- correctness of the product kernel against a CPU evaluation
  if (check) {
    auto kern = phi_sort_kernel<K, 6, 0, 1, 1>;
    size_t lds = std::max(ps_lds_bytes<K, 6, 1>(), ps_epilogue_bytes<K>());
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(dpart, 0, G * E1 * 8));
    hipLaunchKernelGGL(kern, dim3(G), dim3(PS_THREADS), lds, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<double> part(G * E1), got(E1, 0.0);
    CK(hipMemcpy(part.data(), dpart, G * E1 * 8, hipMemcpyDeviceToHost));
    for (int g = 0; g < G; ++g) for (size_t e = 0; e < E1; ++e) got[e] += part[g * E1 + e];
    std::vector<long double> ref(E1, 0.0L);
    const double inv_delta = 1.0 / delta;
    double cf[K + 1][K + 1];
    for (int i = 0; i <= K; ++i) for (int p = 0; p <= K; ++p) cf[i][p] = piece_coef<K, 0>(i, p);
    for (long n = 0; n < N; ++n) {
      const double xv = x[n];
      int c = (int)std::floor((xv - mesh[0]) * inv_delta);
      c = std::min(std::max(c, 0), n_mesh - 2);
      while (c > 0 && !(mesh[c] < xv)) --c;
      while (c < n_mesh - 2 && mesh[c + 1] < xv) ++c;
      const double t = (xv - mesh[c]) * inv_delta;
      double v[K + 1];
      for (int i = 0; i <= K; ++i) { double acc = cf[i][K]; for (int p = K - 1; p >= 0; --p) acc = acc * t + cf[i][p]; v[i] = acc; }
      for (int i = 0; i <= K; ++i) {
        ref[(size_t)(K + 1) * M + c + K - i] += (long double)(v[i] * y[n]);
        for (int j = i; j <= K; ++j) ref[(size_t)(j - i) * M + c + K - j] += (long double)(v[i] * v[j]);
      }
      ref[(size_t)(K + 2) * M] += (long double)(y[n] * y[n]);
    }
    double mxb = 0, mxr = 0, eb = 0, er = 0;
    for (size_t e = 0; e < (size_t)(K + 1) * M; ++e) { mxb = std::max(mxb, std::fabs((double)ref[e])); eb = std::max(eb, std::fabs(got[e] - (double)ref[e])); }
    for (size_t e = (size_t)(K + 1) * M; e < (size_t)(K + 2) * M; ++e) { mxr = std::max(mxr, std::fabs((double)ref[e])); er = std::max(er, std::fabs(got[e] - (double)ref[e])); }
    const double yy = got[(size_t)(K + 2) * M], yyr = (double)ref[(size_t)(K + 2) * M];
    printf("check (TP=6, late prefetch, front loop): band max err %.3e (rel to max entry %.3e), rhs %.3e (%.3e), yy rel %.3e  -> %s\n", eb, eb / mxb, er, er / mxr,
           std::fabs(yy - yyr) / yyr, (eb / mxb < 1e-12 && er / mxr < 1e-12 && std::fabs(yy - yyr) / yyr < 1e-12) ? "OK" : "FAIL");
    (void)ncells;
  }
  return 0;
}
