// Additive model (GPR_additive, gpr.py:139-236): Phi = vstack(Phi_1 .. Phi_d), so Phi Phi^T has banded diagonal blocks
// Phi_i Phi_i^T (the 1-D Phi pass, phi_pass.hip) and DENSE off-diagonal blocks C_ij = Phi_i Phi_j^T (m_i x m_j).
// This file builds one cross block per launch without materialising Kuf (gpr.py:169-172 vstack + SpGEMM + todense):
// every point adds the (k+1)^2 outer product of its two piece vectors into a workgroup-private LDS image of C_ij
// (64-bit fixed point, s0 + products in [0, 1]: ds_add_u64 at the plain LDS write rate), one flush per workgroup,
// then a cross-workgroup reduce.  Blocks too large for the LDS (m_i * m_j > ~19k) go straight to L2 with fp64 atomics.
//
// HBM roofline: 16 B / point / block (x_i and x_j read once).
#include "asvgp_common.hpp"

namespace asvgp {

constexpr int CROSS_THREADS = 1024;
constexpr size_t CROSS_LDS_BUDGET = 160 * 1024 - 512;

template <int K, bool USE_LDS>
__global__ __launch_bounds__(CROSS_THREADS) void phi_cross_kernel(
    const double* __restrict__ xi, const double* __restrict__ xj, long N, const double* __restrict__ mesh_i_g, int n_i,
    double inv_di, const double* __restrict__ mesh_j_g, int n_j, double inv_dj, int m_i, int m_j,
    double* __restrict__ partials, double* __restrict__ out, long ppb, int s0) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int E = USE_LDS ? m_i * m_j : 0;
  unsigned long long* C = reinterpret_cast<unsigned long long*>(lds);
  double* mesh_i = lds + E;
  double* mesh_j = mesh_i + n_i;
  for (int e = tid; e < E; e += CROSS_THREADS) lds[e] = 0.0;
  for (int e = tid; e < n_i; e += CROSS_THREADS) mesh_i[e] = mesh_i_g[e];
  for (int e = tid; e < n_j; e += CROSS_THREADS) mesh_j[e] = mesh_j_g[e];
  __syncthreads();
  const double mi0 = mesh_i[0], mj0 = mesh_j[0];
  const int chi = ((1075 - s0) << 20) | 0x80000;
  const long beg = (long)blockIdx.x * ppb;
  long end = beg + ppb;
  if (end > N) end = N;
  for (long n = beg + tid; n < end; n += CROSS_THREADS) {
    const double a = xi[n], b = xj[n];
    const int ia = neighbour_index(a, mesh_i, n_i, mi0, inv_di);
    const int ib = neighbour_index(b, mesh_j, n_j, mj0, inv_dj);
    double va[K + 1], vb[K + 1];
    bspline_pieces<K>((a - mesh_i[ia]) * inv_di, va);
    bspline_pieces<K>((b - mesh_j[ib]) * inv_dj, vb);
#pragma unroll
    for (int p = 0; p <= K; ++p)
#pragma unroll
      for (int q = 0; q <= K; ++q) {
        const long o = (long)(ia + K - p) * m_j + (ib + K - q);   // row idx + k - piece (basis.py:72)
        if (USE_LDS) lds_add_u64(C + o, fx_convert(va[p] * vb[q], chi));
        else __hip_atomic_fetch_add(out + o, va[p] * vb[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
  if (USE_LDS) {
    __syncthreads();
    double* dst = partials + (size_t)blockIdx.x * E;
    for (int e = tid; e < E; e += CROSS_THREADS) dst[e] = ldexp((double)(long long)C[e], -s0);
  }
}

__global__ __launch_bounds__(256) void cross_reduce_kernel(const double* __restrict__ partials, int G, long E,
                                                           double* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int per = (G + gridDim.y - 1) / gridDim.y;
  int g0 = blockIdx.y * per, g1 = g0 + per;
  if (g1 > G) g1 = G;
  double s = 0.0;
  for (int g = g0; g < g1; ++g) s += partials[(size_t)g * E + e];
  if (s != 0.0) __hip_atomic_fetch_add(out + e, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int K>
static int launch_cross(const double* xi, const double* xj, long N, const double* mesh_i, long n_i, double di,
                        const double* mesh_j, long n_j, double dj, long m_i, long m_j, double* out, double* ws,
                        size_t ws_bytes, hipStream_t st) {
  const long E = m_i * m_j;
  hipError_t e = hipMemsetAsync(out, 0, (size_t)E * sizeof(double), st);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
  if (N == 0) return ASVGP_OK;
  const size_t lds_full = sizeof(double) * (size_t)(E + n_i + n_j);
  const bool use_lds = lds_full <= CROSS_LDS_BUDGET;
  long nblk = (N + 2 * CROSS_THREADS - 1) / (2 * CROSS_THREADS);
  int G = (int)(nblk < 1 ? 1 : (nblk > 256 ? 256 : nblk));
  if (use_lds && ws_bytes < (size_t)G * E * sizeof(double)) {
    set_error("phi_cross_2d: workspace %zu B < %zu B", ws_bytes, (size_t)G * E * sizeof(double));
    return ASVGP_ERR_WORKSPACE;
  }
  long ppb = (N + G - 1) / G;
  ppb = ((ppb + CROSS_THREADS - 1) / CROSS_THREADS) * CROSS_THREADS;
  int s0 = 50;
  { long c = 2; int lg = 1; while (c < ppb) { c <<= 1; ++lg; } if (62 - lg < s0) s0 = 62 - lg; }
  if (use_lds) {
    auto kern = phi_cross_kernel<K, true>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_full, hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
    hipLaunchKernelGGL(kern, dim3(G), dim3(CROSS_THREADS), lds_full, st, xi, xj, N, mesh_i, (int)n_i, 1.0 / di, mesh_j,
                       (int)n_j, 1.0 / dj, (int)m_i, (int)m_j, ws, out, ppb, s0);
    const int gsplit = G >= 64 ? 8 : 1;
    hipLaunchKernelGGL(cross_reduce_kernel, dim3((unsigned)((E + 255) / 256), gsplit), dim3(256), 0, st, ws, G, E, out);
  } else {
    const size_t lds_small = sizeof(double) * (size_t)(n_i + n_j);
    if (lds_small > CROSS_LDS_BUDGET) { set_error("phi_cross_2d: mesh tables exceed the LDS"); return ASVGP_ERR_LDS_CAPACITY; }
    auto kern = phi_cross_kernel<K, false>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
    hipLaunchKernelGGL(kern, dim3(G), dim3(CROSS_THREADS), lds_small, st, xi, xj, N, mesh_i, (int)n_i, 1.0 / di, mesh_j,
                       (int)n_j, 1.0 / dj, (int)m_i, (int)m_j, ws, out, ppb, s0);
  }
  return check_launch("phi_cross_2d");
}

}  // namespace asvgp

using namespace asvgp;

extern "C" size_t asvgp_phi_cross_workspace_bytes(int64_t m_i, int64_t m_j) {
  if (m_i < 1 || m_j < 1) return 0;
  // the per-workgroup partial images exist only on the LDS path (launch_cross: 8 (E + n_i + n_j) <= CROSS_LDS_BUDGET with
  // n = m - order + 1 >= m - ASVGP_MAX_ORDER + 1); larger blocks accumulate with L2 atomics and never touch the workspace
  const long n_min = (m_i > ASVGP_MAX_ORDER ? m_i - ASVGP_MAX_ORDER + 1 : 2) + (m_j > ASVGP_MAX_ORDER ? m_j - ASVGP_MAX_ORDER + 1 : 2);
  if (sizeof(double) * ((size_t)m_i * (size_t)m_j + (size_t)n_min) > CROSS_LDS_BUDGET) return 8;
  return (size_t)256 * (size_t)m_i * (size_t)m_j * sizeof(double);
}

extern "C" int asvgp_phi_cross_2d(const double* x_i, const double* x_j, int64_t N, const double* mesh_i, int64_t n_mesh_i,
                                  double delta_i, int64_t m_i, const double* mesh_j, int64_t n_mesh_j, double delta_j,
                                  int64_t m_j, int order, double* out, void* workspace, size_t workspace_bytes,
                                  asvgp_stream_t stream) {
  if ((N > 0 && (!x_i || !x_j)) || !mesh_i || !mesh_j || !out || N < 0 || !(delta_i > 0) || !(delta_j > 0) ||
      n_mesh_i != m_i - order + 1 || n_mesh_j != m_j - order + 1 || n_mesh_i < 2 || n_mesh_j < 2) {
    set_error("phi_cross_2d: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  if (order < 1 || order > ASVGP_MAX_ORDER) { set_error("phi_cross_2d: order %d unsupported", order); return ASVGP_ERR_UNSUPPORTED; }
  hipStream_t st = as_stream(stream);
  double* ws = static_cast<double*>(workspace);
#define CROSS_CASE(KK) case KK: return launch_cross<KK>(x_i, x_j, N, mesh_i, n_mesh_i, delta_i, mesh_j, n_mesh_j, delta_j, m_i, m_j, out, ws, workspace_bytes, st);
  switch (order) {
    CROSS_CASE(1) CROSS_CASE(2) CROSS_CASE(3) CROSS_CASE(4) CROSS_CASE(5) CROSS_CASE(6)
  }
#undef CROSS_CASE
  return ASVGP_ERR_UNSUPPORTED;
}
