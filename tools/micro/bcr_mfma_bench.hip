// Stand-alone harness for the matrix-core block cyclic reduction (asvgp_amd/csrc/bcr_mfma.hpp) against a long-double banded Cholesky /
// Takahashi evaluation on the host and against bcr.hpp's chain: correctness, kernel time, per-level cycle stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/bcr_mfma_bench.hip -o tools/micro/bin/bcr_mfma_bench
//   tools/micro/bin/bcr_mfma_bench [M=2048] [seed=1]
#include <hip/hip_runtime.h>

#include <algorithm>
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
#include "../../asvgp_amd/csrc/bcr_mfma.hpp"

using namespace asvgp;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(BM_THREADS) void mfma_kernel(const double* band, const double* rhs, int M, double* ws, double* S, double* x, double* logdet,
                                                          int* info, double* stamps) {
  extern __shared__ double lds[];
  bcr_mfma_solve<BandPtr<double>>(BandPtr<double>{band, nullptr}, rhs, M, ws, lds, S, x, logdet, info, 1, stamps);
}
__global__ __launch_bounds__(BCR_THREADS) void old_kernel(const double* band, const double* rhs, int M, double* ws, double* S, double* x, double* logdet,
                                                          int* info) {
  extern __shared__ double lds[];
  bcr_solve<double, 4, 1, BandPtr<double>, false>(BandPtr<double>{band, nullptr}, rhs, M, ws, lds, BandOut<double>{S, nullptr}, x, logdet, info, nullptr, 1);
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

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int M = argc > 1 ? atoi(argv[1]) : 2048;
  const int seed = argc > 2 ? atoi(argv[2]) : 1;
  const int K = 4;
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(-0.5, 0.5), Ud(1.0, 2.0);
  // P = L0 L0^T with a random lower-banded L0: SPD, bandwidth 4
  std::vector<double> L0((size_t)(K + 1) * M, 0.0), band((size_t)(K + 1) * M, 0.0), rhs(M);
  for (int j = 0; j < M; ++j) {
    L0[j] = Ud(rng);
    for (int d = 1; d <= K; ++d) if (j + d < M) L0[(size_t)d * M + j] = U(rng);
    rhs[j] = U(rng) * 10.0;
  }
  auto l0 = [&](int i, int j) -> double { return (i >= j && i - j <= K) ? L0[(size_t)(i - j) * M + j] : 0.0; };
  for (int j = 0; j < M; ++j)
    for (int d = 0; d <= K && j + d < M; ++d) {
      double s = 0.0;
      const int i = j + d;
      for (int p = std::max(0, i - K); p <= j; ++p) s += l0(i, p) * l0(j, p);
      band[(size_t)d * M + j] = s;
    }
  // ---- host reference in long double: band Cholesky, solve, Takahashi selected inverse
  std::vector<long double> Lb((size_t)(K + 1) * M, 0.0L), Sb((size_t)(K + 1) * M, 0.0L), xr(M);
  auto Lr = [&](int i, int j) -> long double& { return Lb[(size_t)(i - j) * M + j]; };
  long double logdet_ref = 0.0L;
  for (int j = 0; j < M; ++j) {
    long double s = band[j];
    for (int p = std::max(0, j - K); p < j; ++p) s -= Lr(j, p) * Lr(j, p);
    const long double ljj = sqrtl(s);
    Lr(j, j) = ljj;
    logdet_ref += 2.0L * logl(ljj);
    for (int i = j + 1; i <= std::min(M - 1, j + K); ++i) {
      long double t = band[(size_t)(i - j) * M + j];
      for (int p = std::max(0, i - K); p < j; ++p) t -= Lr(i, p) * Lr(j, p);
      Lr(i, j) = t / ljj;
    }
  }
  {
    std::vector<long double> z(M);
    for (int i = 0; i < M; ++i) { long double t = rhs[i]; for (int p = std::max(0, i - K); p < i; ++p) t -= Lr(i, p) * z[p]; z[i] = t / Lr(i, i); }
    for (int i = M - 1; i >= 0; --i) { long double t = z[i]; for (int p = i + 1; p <= std::min(M - 1, i + K); ++p) t -= Lr(p, i) * xr[p]; xr[i] = t / Lr(i, i); }
  }
  auto Sg = [&](int i, int j) -> long double { if (i < j) std::swap(i, j); return (i - j <= K) ? Sb[(size_t)(i - j) * M + j] : 0.0L; };
  for (int j = M - 1; j >= 0; --j)
    for (int i = std::min(j + K, M - 1); i >= j; --i) {
      long double t = (i == j) ? 1.0L / Lr(j, j) : 0.0L;
      for (int p = j + 1; p <= std::min(j + K, M - 1); ++p) t -= Lr(p, j) * Sg(p, i);
      Sb[(size_t)(i - j) * M + j] = t / Lr(j, j);
    }

  const long nb = (M + 3) / 4;
  double *dband, *drhs, *dws, *dS, *dx, *dld, *dstamps, *dws_old;
  int* dinfo;
  CK(hipMalloc(&dband, band.size() * 8)); CK(hipMalloc(&drhs, M * 8)); CK(hipMalloc(&dws, bcr_mfma_ws_doubles(nb) * 8));
  CK(hipMalloc(&dS, band.size() * 8)); CK(hipMalloc(&dx, M * 8)); CK(hipMalloc(&dld, 64 * 8)); CK(hipMalloc(&dstamps, 64 * 8)); CK(hipMalloc(&dinfo, 64));
  const size_t ws_old = bcr_ws_doubles<double, 4, 1, false>(nb) + 64;
  CK(hipMalloc(&dws_old, ws_old * 8));
  CK(hipMemcpy(dband, band.data(), band.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(drhs, rhs.data(), M * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dstamps, 0, 64 * 8));

  auto compare = [&](const char* tag) {
    std::vector<double> S(band.size()), x(M), ldv(4);
    int info = -7;
    CK(hipMemcpy(S.data(), dS, S.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x.data(), dx, M * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ldv.data(), dld, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    double es = 0, ms = 0, ex = 0, mx = 0;
    for (size_t k2 = 0; k2 < S.size(); ++k2) { ms = std::max(ms, (double)fabsl(Sb[k2])); es = std::max(es, (double)fabsl((long double)S[k2] - Sb[k2])); }
    for (int i = 0; i < M; ++i) { mx = std::max(mx, (double)fabsl(xr[i])); ex = std::max(ex, (double)fabsl((long double)x[i] - xr[i])); }
    printf("%s: info %d  log-det %.15g (ref %.15Lg, rel err %.2e)  x max err %.2e (rel %.2e)  band of inverse max err %.2e (rel %.2e)\n", tag, info, ldv[0],
           logdet_ref, fabs(ldv[0] - (double)logdet_ref) / fabs((double)logdet_ref), ex, ex / mx, es, es / ms);
  };

  const size_t lds_new = bcr_mfma_lds_doubles(nb) * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
  CK(hipMemset(dS, 0xff, band.size() * 8));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(BM_THREADS), lds_new, 0, dband, drhs, M, dws, dS, dx, dld, dinfo, dstamps);   // (stamps of a warm launch)
  CK(hipDeviceSynchronize());
  compare("mfma");
  {
    std::vector<double> st(64);
    CK(hipMemcpy(st.data(), dstamps, 64 * 8, hipMemcpyDeviceToHost));
    int levels = 0; while ((1 << levels) < nb) ++levels;
    double tot = 0; for (double v : st) tot += v;
    printf("mfma stamps (cycles): pre-pass %.0f | forward levels 0..", st[0]);
    for (int l = 0; l < levels; ++l) printf(" %.0f", st[1 + l]);
    printf(" | root %.0f | backward levels %d..0", st[1 + levels], levels - 1);
    for (int l = 0; l < levels; ++l) printf(" %.0f", st[2 + levels + l]);
    printf(" | outputs %.0f | total %.0f\n", st[2 + 2 * levels], tot);
  }
  const float t_new = time_us([&] { hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(BM_THREADS), lds_new, 0, dband, drhs, M, dws, dS, dx, dld, dinfo, (double*)nullptr); }, 50);

  const size_t lds_old = bcr_lds_doubles<double, 4, 1, false>(nb) * 8;
  float t_old = -1.f;
  if (lds_old <= 160 * 1024) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(old_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
    CK(hipMemset(dS, 0xff, band.size() * 8));
    hipLaunchKernelGGL(old_kernel, dim3(1), dim3(BCR_THREADS), lds_old, 0, dband, drhs, M, dws_old, dS, dx, dld, dinfo);
    CK(hipDeviceSynchronize());
    compare("bcr.hpp");
    t_old = time_us([&] { hipLaunchKernelGGL(old_kernel, dim3(1), dim3(BCR_THREADS), lds_old, 0, dband, drhs, M, dws_old, dS, dx, dld, dinfo); }, 50);
  }
  printf("M = %d: matrix-core chain %.1f us, bcr.hpp chain %.1f us\n", M, t_new, t_old);
  return 0;
}
