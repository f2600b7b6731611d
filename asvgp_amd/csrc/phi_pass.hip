// Phi pass: B-spline design matrix evaluation fused with the banded sufficient statistics.
//
// Replaces (reference, HJakeCunningham/ASVGP): basis.py:51-76 evaluate_basis (index search, piece
// polynomials, COO->CSR), gpr.py:41-44 (Kuf@y, Kuf@Kuf.T, sparse_to_band, sum y^2).  Phi is never
// materialised: each point's (k+1) non-zeros are formed in registers and scattered straight into an
// LDS-resident private band [(k+1) x cols | rhs | mesh table]; one flush per workgroup, then a
// cross-workgroup tree/atomic reduce into the packed stats buffer.
//
// HBM roofline: 16 B/point (x and y read once, fp64).  Everything else is on-chip.
#include "asvgp_common.hpp"

namespace asvgp {

constexpr int PHI_THREADS = 1024;
constexpr int PHI_MAX_BLOCKS = 256;           // one 1024-thread workgroup per CU (LDS-limited)
constexpr size_t PHI_LDS_BUDGET = 160 * 1024 - 512;

__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_f64, no return
}

template <int K>
__device__ __forceinline__ void phi_point(double xv, double yv, const double* mesh, int n_mesh, double m0,
                                          double inv_delta, int cell0, int cell1, int ncols, bool do_band,
                                          double* band, double* rhs, double& yy) {
  int idx = neighbour_index(xv, mesh, n_mesh, m0, inv_delta);
  if (idx < cell0 || idx >= cell1) return;
  double t = (xv - mesh[idx]) * inv_delta;
  double v[K + 1];
  bspline_pieces<K>(t, v);
  int cb = idx - cell0;  // local column of basis row `idx`; piece i lives on row idx + K - i
#pragma unroll
  for (int i = 0; i <= K; ++i) lds_add(rhs + cb + K - i, v[i] * yv);
  if (do_band) {
#pragma unroll
    for (int i = 0; i <= K; ++i)
#pragma unroll
      for (int j = i; j <= K; ++j)  // row_i = idx+K-i >= row_j = idx+K-j: sub-diagonal d = j-i, column row_j
        lds_add(band + (j - i) * ncols + cb + K - j, v[i] * v[j]);
  }
  yy = fma(yv, yv, yy);
}

// One workgroup per CU; block b owns points [b*ppb, (b+1)*ppb).  VEC: 16-B loads of (x0,x1),(y0,y1).
template <int K, bool VEC>
__global__ __launch_bounds__(PHI_THREADS) void phi_accumulate_kernel(
    const double* __restrict__ x, const double* __restrict__ y, long y_stride, long N,
    const double* __restrict__ mesh_g, int n_mesh, double inv_delta, int cell0, int cell1, int ncols,
    int do_band, double* __restrict__ partials, long ppb) {
  extern __shared__ double lds[];
  double* band = lds;                      // (K+1) x ncols
  double* rhs = band + (K + 1) * ncols;    // ncols
  double* mesh = rhs + ncols;              // n_mesh
  double* scratch = mesh + n_mesh;         // 16
  const int tid = threadIdx.x;
  const int E = (K + 2) * ncols;
  for (int e = tid; e < E; e += PHI_THREADS) lds[e] = 0.0;
  for (int e = tid; e < n_mesh; e += PHI_THREADS) mesh[e] = mesh_g[e];
  __syncthreads();
  const double m0 = mesh[0];
  const long beg = (long)blockIdx.x * ppb;
  long end = beg + ppb;
  if (end > N) end = N;
  double yy = 0.0;
  if (VEC) {
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const double2* y2 = reinterpret_cast<const double2*>(y);
    const long pend = end >> 1;  // pairs [beg/2, pend)
    long p = (beg >> 1) + tid;
    double2 xa, ya;
    if (p < pend) { xa = x2[p]; ya = y2[p]; }
    while (p < pend) {
      long pn = p + PHI_THREADS;
      double2 xb = xa, yb = ya;
      if (pn < pend) { xa = x2[pn]; ya = y2[pn]; }   // prefetch next pair before the LDS-atomic burst
      phi_point<K>(xb.x, yb.x, mesh, n_mesh, m0, inv_delta, cell0, cell1, ncols, do_band, band, rhs, yy);
      phi_point<K>(xb.y, yb.y, mesh, n_mesh, m0, inv_delta, cell0, cell1, ncols, do_band, band, rhs, yy);
      p = pn;
    }
    if ((end & 1) && tid == 0 && end > beg)  // odd tail point (only the last block can have one)
      phi_point<K>(x[end - 1], y[end - 1], mesh, n_mesh, m0, inv_delta, cell0, cell1, ncols, do_band, band, rhs, yy);
  } else {
    for (long i = beg + tid; i < end; i += PHI_THREADS)
      phi_point<K>(x[i], y[i * y_stride], mesh, n_mesh, m0, inv_delta, cell0, cell1, ncols, do_band, band, rhs, yy);
  }
  double tot = block_sum(yy, scratch);  // contains the barrier that orders the LDS atomics before the flush
  __syncthreads();
  double* out = partials + (size_t)blockIdx.x * (E + 1);
  for (int e = tid; e < E; e += PHI_THREADS) out[e] = lds[e];
  if (tid == 0) out[E] = tot;
}

// Sum the per-workgroup partials into the packed stats buffer (zeroed beforehand).
// grid = (ceil((E+1)/256), gsplit); each thread sums its slice of workgroups, then one fp64 global atomic.
__global__ __launch_bounds__(256) void phi_reduce_kernel(const double* __restrict__ partials, int G, int ncols,
                                                         int K, int col0, long M, long D, int dcol, int do_band,
                                                         double* __restrict__ stats) {
  const int E1 = (K + 2) * ncols + 1;
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E1) return;
  int per = (G + gridDim.y - 1) / gridDim.y;
  int g0 = blockIdx.y * per, g1 = g0 + per;
  if (g1 > G) g1 = G;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int g = g0;
  for (; g + 3 < g1; g += 4) {
    s0 += partials[(size_t)g * E1 + e];
    s1 += partials[(size_t)(g + 1) * E1 + e];
    s2 += partials[(size_t)(g + 2) * E1 + e];
    s3 += partials[(size_t)(g + 3) * E1 + e];
  }
  for (; g < g1; ++g) s0 += partials[(size_t)g * E1 + e];
  double s = (s0 + s1) + (s2 + s3);
  long o;
  if (e < (K + 1) * ncols) {
    if (!do_band) return;
    int d = e / ncols, c = e - d * ncols;
    if (col0 + c >= M) return;
    o = (long)d * M + col0 + c;
  } else if (e < (K + 2) * ncols) {
    int c = e - (K + 1) * ncols;
    if (col0 + c >= M) return;
    o = (long)(K + 1) * M + (long)(col0 + c) * D + dcol;
  } else {
    o = (long)(K + 1) * M + M * D;
  }
  if (s != 0.0) __hip_atomic_fetch_add(stats + o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void phi_index_kernel(const double* __restrict__ x, long N, const double* __restrict__ mesh, int n_mesh,
                                 double inv_delta, long long* __restrict__ idx) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  idx[i] = neighbour_index(x[i], mesh, n_mesh, mesh[0], inv_delta);
}

template <int K, int DERIV>
__global__ void phi_evaluate_kernel(const double* __restrict__ x, long N, const double* __restrict__ mesh,
                                    int n_mesh, double inv_delta, long long* __restrict__ rows,
                                    double* __restrict__ data) {
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double xv = x[n];
  int idx = neighbour_index(xv, mesh, n_mesh, mesh[0], inv_delta);
  double t = (xv - mesh[idx]) * inv_delta;
  double v[K + 1];
  bspline_pieces<K, DERIV>(t, v);
  double sc = 1.0;
#pragma unroll
  for (int d = 0; d < DERIV; ++d) sc *= inv_delta;
#pragma unroll
  for (int i = 0; i <= K; ++i) {
    rows[(long)i * N + n] = idx + K - i;
    data[(long)i * N + n] = v[i] * sc;
  }
}

// Posterior moments per test point: phi* has k+1 contiguous non-zeros, so
// mean = sum_i phi_i alpha[row_i], var = v + sum_ij phi_i phi_j W[|r_i-r_j|][min(r_i,r_j)], W = band(P^-1)-band(Kuu^-1).
// alpha, W and the mesh table are staged in LDS once per workgroup; 8 B in, 16 B out per point.
template <int K>
__global__ __launch_bounds__(1024) void predict_kernel(const double* __restrict__ xnew, long n,
                                                       const double* __restrict__ mesh_g, int n_mesh,
                                                       double inv_delta, int M, const double* __restrict__ alpha_g,
                                                       const double* __restrict__ W_g, double variance, int D,
                                                       int stage, double* __restrict__ mean,
                                                       double* __restrict__ var) {
  extern __shared__ double lds[];
  const double* W = W_g;
  const double* alpha = alpha_g;
  const double* mesh = mesh_g;
  if (stage) {  // stage == 1 implies D == 1
    double* w = lds;
    double* a = w + (K + 1) * M;
    double* ms = a + M;
    for (int e = threadIdx.x; e < (K + 1) * M; e += blockDim.x) w[e] = W_g[e];
    for (int e = threadIdx.x; e < M; e += blockDim.x) a[e] = alpha_g[e];
    for (int e = threadIdx.x; e < n_mesh; e += blockDim.x) ms[e] = mesh_g[e];
    __syncthreads();
    W = w; alpha = a; mesh = ms;
  }
  const double m0 = mesh[0];
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
    double xv = xnew[p];
    int idx = neighbour_index(xv, mesh, n_mesh, m0, inv_delta);
    double t = (xv - mesh[idx]) * inv_delta;
    double v[K + 1];
    bspline_pieces<K>(t, v);
    double q = 0.0;
#pragma unroll
    for (int i = 0; i <= K; ++i) {      // row_i = idx + K - i
      double acc = 0.5 * v[i] * W[idx + K - i];   // diagonal term (halved, doubled below)
#pragma unroll
      for (int j = i + 1; j <= K; ++j)  // row_j < row_i: W[d=j-i][row_j]
        acc = fma(v[j], W[(j - i) * M + idx + K - j], acc);
      q = fma(v[i], acc, q);
    }
    var[p] = fma(2.0, q, variance);
    for (int d = 0; d < D; ++d) {
      double m = 0.0;
#pragma unroll
      for (int i = 0; i <= K; ++i) m = fma(v[i], alpha[(long)(idx + K - i) * D + d], m);
      mean[p * D + d] = m;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Optional in-library timing of the dominant kernel: HIP events recorded on the launch stream right around
// phi_accumulate_kernel (bench.py's roofline figure; must agree with the rocprofv3 kernel-trace average).
constexpr int PROF_RING = 1024;
static bool g_prof_on = false;
static hipEvent_t g_prof_ev[PROF_RING][2];
static bool g_prof_made = false;
static long g_prof_n = 0;

static int phi_max_cols(int K, long n_mesh) {
  long avail = (long)PHI_LDS_BUDGET - (long)n_mesh * 8 - 16 * 8;
  if (avail <= 0) return 0;
  return (int)(avail / (8 * (K + 2)));
}

template <int K>
static int launch_phi(const double* x, const double* y, long N, long D, const double* mesh, long n_mesh,
                      double delta, long M, double* stats, double* partials, hipStream_t st) {
  const int ncells = (int)n_mesh - 1;
  int maxc = phi_max_cols(K, n_mesh);
  if (maxc < 2 * K + 2) {
    set_error("phi_accumulate_1d: mesh table (%ld knots) leaves no LDS for the band", n_mesh);
    return ASVGP_ERR_LDS_CAPACITY;
  }
  int cells_per_chunk = (M <= maxc) ? ncells : (maxc - K);
  long nblk = (N + 2 * PHI_THREADS - 1) / (2 * PHI_THREADS);
  int G = (int)(nblk < 1 ? 1 : (nblk > PHI_MAX_BLOCKS ? PHI_MAX_BLOCKS : nblk));
  long ppb = (N + G - 1) / G;
  ppb = ((ppb + 2 * PHI_THREADS - 1) / (2 * PHI_THREADS)) * (2 * PHI_THREADS);
  const double inv_delta = 1.0 / delta;
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * ((K + 1) * M + M * D + 1), st);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
  for (long dcol = 0; dcol < D; ++dcol) {
    const double* yd = y + dcol;
    bool vec = (D == 1) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(yd) & 15) == 0);
    for (int cell0 = 0; cell0 < ncells; cell0 += cells_per_chunk) {
      int cell1 = cell0 + cells_per_chunk;
      if (cell1 > ncells) cell1 = ncells;
      int ncols = cell1 - cell0 + K;
      size_t lds_bytes = sizeof(double) * ((size_t)(K + 2) * ncols + n_mesh + 16);
      int do_band = (dcol == 0);
      auto kern = vec ? phi_accumulate_kernel<K, true> : phi_accumulate_kernel<K, false>;
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_bytes, hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
      const bool prof = g_prof_on && g_prof_n < PROF_RING;
      if (prof) hipEventRecord(g_prof_ev[g_prof_n][0], st);
      hipLaunchKernelGGL(kern, dim3(G), dim3(PHI_THREADS), lds_bytes, st, x, yd, (long)D, N, mesh, (int)n_mesh,
                         inv_delta, cell0, cell1, ncols, do_band, partials, ppb);
      if (prof) { hipEventRecord(g_prof_ev[g_prof_n][1], st); ++g_prof_n; }
      int E1 = (K + 2) * ncols + 1;
      int gsplit = G >= 64 ? 16 : (G >= 8 ? 4 : 1);
      hipLaunchKernelGGL(phi_reduce_kernel, dim3((E1 + 255) / 256, gsplit), dim3(256), 0, st, partials, G, ncols, K,
                         cell0, M, D, (int)dcol, do_band, stats);
    }
  }
  return check_launch("phi_accumulate_1d");
}

}  // namespace asvgp

using namespace asvgp;

extern "C" int asvgp_profile_enable(int on) {
  if (on && !g_prof_made) {
    for (int i = 0; i < PROF_RING; ++i)
      for (int j = 0; j < 2; ++j)
        if (hipEventCreate(&g_prof_ev[i][j]) != hipSuccess) { set_error("hipEventCreate failed"); return ASVGP_ERR_HIP; }
    g_prof_made = true;
  }
  g_prof_on = on != 0;
  g_prof_n = 0;
  return ASVGP_OK;
}

extern "C" int asvgp_profile_read(double* phi_kernel_ms_sum, int64_t* launches) {
  if (!phi_kernel_ms_sum || !launches) { set_error("profile_read: bad argument"); return ASVGP_ERR_BAD_ARG; }
  double tot = 0.0;
  for (long i = 0; i < g_prof_n; ++i) {
    if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess) { set_error("hipEventSynchronize failed"); return ASVGP_ERR_HIP; }
    float ms = 0.f;
    hipEventElapsedTime(&ms, g_prof_ev[i][0], g_prof_ev[i][1]);
    tot += ms;
  }
  *phi_kernel_ms_sum = tot;
  *launches = g_prof_n;
  g_prof_n = 0;
  return ASVGP_OK;
}

extern "C" size_t asvgp_phi_workspace_bytes(int64_t M, int order, int64_t D) {
  (void)D;
  if (M <= 0 || order < 1 || order > ASVGP_MAX_ORDER) return 0;
  return sizeof(double) * (size_t)PHI_MAX_BLOCKS * ((size_t)(order + 2) * (size_t)M + 1);
}

extern "C" int asvgp_phi_accumulate_1d(const double* x, const double* y, int64_t N, int64_t D, const double* mesh,
                                       int64_t n_mesh, double delta, int order, int64_t M, double* stats,
                                       void* workspace, size_t workspace_bytes, asvgp_stream_t stream) {
  if (((!x || !y) && N > 0) || !mesh || !stats || N < 0 || D < 1 || M < 1 || !(delta > 0.0)) {
    set_error("phi_accumulate_1d: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  if (order < 1 || order > ASVGP_MAX_ORDER) { set_error("phi_accumulate_1d: order %d unsupported", order); return ASVGP_ERR_UNSUPPORTED; }
  if (n_mesh != M - order + 1 || n_mesh < 2) { set_error("phi_accumulate_1d: n_mesh=%ld != M-order+1", (long)n_mesh); return ASVGP_ERR_BAD_ARG; }
  if (M > 0x3fffffff) { set_error("phi_accumulate_1d: M too large"); return ASVGP_ERR_UNSUPPORTED; }
  if (!workspace || workspace_bytes < asvgp_phi_workspace_bytes(M, order, D)) {
    set_error("phi_accumulate_1d: workspace too small (%zu < %zu)", workspace_bytes, asvgp_phi_workspace_bytes(M, order, D));
    return ASVGP_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  double* part = static_cast<double*>(workspace);
  switch (order) {
    case 1: return launch_phi<1>(x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    case 2: return launch_phi<2>(x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    case 3: return launch_phi<3>(x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    case 4: return launch_phi<4>(x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    case 5: return launch_phi<5>(x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    default: return launch_phi<6>(x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
  }
}

extern "C" int asvgp_phi_index_1d(const double* x, int64_t N, const double* mesh, int64_t n_mesh, double delta,
                                  int64_t* idx, asvgp_stream_t stream) {
  if (!x || !mesh || !idx || N < 0 || n_mesh < 2 || !(delta > 0.0)) { set_error("phi_index_1d: bad argument"); return ASVGP_ERR_BAD_ARG; }
  if (N == 0) return ASVGP_OK;
  hipLaunchKernelGGL(phi_index_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), x, (long)N,
                     mesh, (int)n_mesh, 1.0 / delta, reinterpret_cast<long long*>(idx));
  return check_launch("phi_index_1d");
}

template <int K>
static int launch_eval(const double* x, long N, const double* mesh, int n_mesh, double delta, int deriv,
                       long long* rows, double* data, hipStream_t st) {
  dim3 g((unsigned)((N + 255) / 256)), b(256);
  double id = 1.0 / delta;
  switch (deriv) {
    case 0: hipLaunchKernelGGL((phi_evaluate_kernel<K, 0>), g, b, 0, st, x, N, mesh, n_mesh, id, rows, data); break;
    case 1: hipLaunchKernelGGL((phi_evaluate_kernel<K, (K >= 1 ? 1 : 0)>), g, b, 0, st, x, N, mesh, n_mesh, id, rows, data); break;
    case 2: hipLaunchKernelGGL((phi_evaluate_kernel<K, (K >= 2 ? 2 : 0)>), g, b, 0, st, x, N, mesh, n_mesh, id, rows, data); break;
    default: hipLaunchKernelGGL((phi_evaluate_kernel<K, (K >= 3 ? 3 : 0)>), g, b, 0, st, x, N, mesh, n_mesh, id, rows, data); break;
  }
  return check_launch("phi_evaluate_1d");
}

extern "C" int asvgp_phi_evaluate_1d(const double* x, int64_t N, const double* mesh, int64_t n_mesh, double delta,
                                     int order, int deriv, int64_t* rows, double* data, asvgp_stream_t stream) {
  if (!x || !mesh || !rows || !data || N < 0 || n_mesh < 2 || !(delta > 0.0)) { set_error("phi_evaluate_1d: bad argument"); return ASVGP_ERR_BAD_ARG; }
  if (order < 1 || order > ASVGP_MAX_ORDER || deriv < 0 || deriv > 3 || deriv > order) {
    set_error("phi_evaluate_1d: order %d / deriv %d unsupported", order, deriv);
    return ASVGP_ERR_UNSUPPORTED;
  }
  if (N == 0) return ASVGP_OK;
  hipStream_t st = as_stream(stream);
  long long* r = reinterpret_cast<long long*>(rows);
  switch (order) {
    case 1: return launch_eval<1>(x, N, mesh, (int)n_mesh, delta, deriv, r, data, st);
    case 2: return launch_eval<2>(x, N, mesh, (int)n_mesh, delta, deriv, r, data, st);
    case 3: return launch_eval<3>(x, N, mesh, (int)n_mesh, delta, deriv, r, data, st);
    case 4: return launch_eval<4>(x, N, mesh, (int)n_mesh, delta, deriv, r, data, st);
    case 5: return launch_eval<5>(x, N, mesh, (int)n_mesh, delta, deriv, r, data, st);
    default: return launch_eval<6>(x, N, mesh, (int)n_mesh, delta, deriv, r, data, st);
  }
}

template <int K>
static int launch_predict(const double* xnew, long n, const double* mesh, int n_mesh, double delta, int M,
                          const double* alpha, const double* W, double variance, int D, double* mean, double* var,
                          hipStream_t st) {
  size_t lds_bytes = sizeof(double) * ((size_t)(K + 2) * M + n_mesh);
  int stage = (D == 1 && lds_bytes <= PHI_LDS_BUDGET && n >= 65536) ? 1 : 0;
  int threads = stage ? 1024 : 256;
  long blocks = (n + threads - 1) / threads;
  long cap = stage ? 256 : 2048;
  if (blocks > cap) blocks = cap;
  if (stage) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(predict_kernel<K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
  }
  hipLaunchKernelGGL(predict_kernel<K>, dim3((unsigned)blocks), dim3(threads), stage ? lds_bytes : 0, st, xnew, n, mesh,
                     n_mesh, 1.0 / delta, M, alpha, W, variance, D, stage, mean, var);
  return check_launch("predict_1d");
}

extern "C" int asvgp_predict_1d(const double* xnew, int64_t n, const double* mesh, int64_t n_mesh, double delta,
                                int order, int64_t M, const double* alpha, const double* W, double variance,
                                int64_t D, double* mean, double* var, asvgp_stream_t stream) {
  if (!xnew || !mesh || !alpha || !W || !mean || !var || n < 0 || D < 1 || !(delta > 0.0) || n_mesh != M - order + 1) {
    set_error("predict_1d: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  if (order < 1 || order > ASVGP_MAX_ORDER) { set_error("predict_1d: order %d unsupported", order); return ASVGP_ERR_UNSUPPORTED; }
  if (n == 0) return ASVGP_OK;
  hipStream_t st = as_stream(stream);
  switch (order) {
    case 1: return launch_predict<1>(xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    case 2: return launch_predict<2>(xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    case 3: return launch_predict<3>(xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    case 4: return launch_predict<4>(xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    case 5: return launch_predict<5>(xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    default: return launch_predict<6>(xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
  }
}
