// banded_matrices.banded operator replacements + Kuu assembly (C-ABI entry points).
// Reference call sites: gpr.py:56-75, utils.py:7-9,24-57, inducing_features.py:12-44.
#include <math.h>
#include <stdarg.h>

#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "band_sweeps.hpp"

namespace asvgp {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return ASVGP_ERR_HIP;
  }
  return ASVGP_OK;
}

// ---------------------------------------------------------------- kernels
template <int K>
__global__ __launch_bounds__(64) void cholesky_band_kernel(const double* A, double* L, int M, int* info) {
  cholesky_sweep<double, K, false>(BandPtr<double>{A, nullptr}, BandOut<double>{L, nullptr}, M, nullptr, nullptr, info);
}
__device__ __forceinline__ double band_rcp(double x) {       // 1 / x: v_rcp_f64 + two Newton steps
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}
// ---------------------------------------------------------------------------------------------------------
// The two operator recurrences with the band in the LDS (round 4; M (k+1) doubles <= 156 KB - M <= 4000 at k = 4 - in one piece, in segments beyond).
// The register-window sweeps above keep row i on lane i mod (k+1) and pay ~500 instructions per column for v_readlane traffic with
// run-time lane indices and lane-select chains (1.06 / 0.60 ms at M = 2048).  Here the band is staged into the LDS by the whole
// workgroup (coalesced), ONE wave then runs the recurrence with every lane holding the same (k+1) x (k+1) register window - all
// operands are broadcast LDS reads, no cross-lane instruction on the chain, reciprocal square roots by Newton steps on v_rsq_f64 -,
// the result goes back to the LDS and the workgroup stores it coalesced.  What is left per column is the dependent chain itself
// (rsqrt -> scale -> the next pivot's update): ~35 instructions.
// ---------------------------------------------------------------------------------------------------------
// Any M: the columns are walked in SEGMENTS of `seg_blocks` blocks of k+1 columns; a segment's columns (+ the k+1 behind it, which enter
// the window while it is walked) are staged into the LDS by the whole workgroup, the window registers carry over from segment to segment,
// the segment's columns of the factor leave through the LDS.  M (k+1) 8 B <= 160 KB is one segment.
template <int K>
__global__ __launch_bounds__(256) void cholesky_band_lds_kernel(const double* __restrict__ A, double* __restrict__ L, int M, int* __restrict__ info,
                                                                int seg_blocks) {
  extern __shared__ double bs_lds[];                  // [column - c0][K+1], a column's k+1 entries side by side (two-address LDS instructions: 3 per
                                                      // column instead of 5); overwritten by the factor column by column
  constexpr int W = K + 1;
  const int tid = threadIdx.x;
  double w[W][W];                                     // w[c][d]: entry (j + c + d, j + c) of the matrix as updated so far (window of k+1 columns)
#pragma unroll
  for (int c = 0; c <= K; ++c)
#pragma unroll
    for (int d = 0; d <= K; ++d) w[c][d] = (c < M) ? A[(long)d * M + (c < M ? c : 0)] : 0.0;
  // A pivot that is not > 0 (or not finite) makes every later one NaN through the k updates behind it: the sweep carries no test; the
  // first column whose pivot is not > 0 is found when the segments are stored (whole workgroup).
  __shared__ int first_bad;
  if (tid == 0) first_bad = 0x7fffffff;
  const int nblk = (M + W - 1) / W;
  for (int b0 = 0; b0 < nblk; b0 += seg_blocks) {
    const int b1 = b0 + seg_blocks < nblk ? b0 + seg_blocks : nblk;
    const int c0 = b0 * W, c1 = b1 * W < M ? b1 * W : M;
    const int lend = c1 + W < M ? c1 + W : M;         // staged columns [c0, lend)
    for (int d = 0; d <= K; ++d)
      for (int c = c0 + tid; c < lend; c += blockDim.x) bs_lds[(long)(c - c0) * W + d] = A[(long)d * M + c];
    __syncthreads();
    if (tid < 64) {
      // one column; CHECK = false: every row j + d and the column j + k + 1 exist (no bounds tests, no branches in the block of k+1 columns)
      auto column = [&](int j, auto jm_c, auto check_c) __attribute__((always_inline)) {
        constexpr int jm = decltype(jm_c)::value;
        constexpr bool CHECK = decltype(check_c)::value;
        const double piv = w[jm][0];
        double y = __builtin_amdgcn_rsq(piv);         // 1 / sqrt(piv): two Newton steps (<= 2e-16 relative)
        const double hh = 0.5 * piv;
        y = fma(y, fma(-hh * y, y, 0.5), y);
        y = fma(y, fma(-hh * y, y, 0.5), y);
        double l[W];
        l[0] = piv;                                   // (the diagonal entry leaves as the PIVOT: its square root is taken when the segment is stored,
#pragma unroll                                        // by the whole workgroup - the recurrence only needs 1 / sqrt(piv))
        for (int d = 1; d <= K; ++d) l[d] = (!CHECK || j + d < M) ? w[jm][d] * y : 0.0;
#pragma unroll
        for (int c = 1; c <= K; ++c)                  // the k columns behind it
#pragma unroll
          for (int d = 0; d + c <= K; ++d) w[(jm + c) % W][d] = fma(-l[c + d], l[c], w[(jm + c) % W][d]);
#pragma unroll
        for (int d = 0; d <= K; ++d) bs_lds[(long)(j - c0) * W + d] = l[d];     // (every lane: the same value to the same address)
        const int jn = j + K + 1;                     // the window slot takes column j + k + 1 (untouched so far, never the address just written)
#pragma unroll
        for (int d = 0; d <= K; ++d) w[jm][d] = (!CHECK || jn < M) ? bs_lds[(long)((!CHECK || jn < M) ? jn - c0 : 0) * W + d] : 0.0;
      };
      int jb = c0;
      for (; jb < c1 && jb + 2 * K + 1 < M; jb += W) {        // full blocks: unrolled k+1 times, every window index a compile-time constant
        [&]<int... JM>(std::integer_sequence<int, JM...>) { (column(jb + JM, std::integral_constant<int, JM>{}, std::false_type{}), ...); }(std::make_integer_sequence<int, W>{});
      }
      for (; jb < c1; jb += W) {
        [&]<int... JM>(std::integer_sequence<int, JM...>) { ((jb + JM < M ? column(jb + JM, std::integral_constant<int, JM>{}, std::true_type{}) : (void)0), ...); }(std::make_integer_sequence<int, W>{});
      }
    }
    __syncthreads();
    for (int d = 0; d <= K; ++d)
      for (int c = c0 + tid; c < c1; c += blockDim.x) {
        double v = bs_lds[(long)(c - c0) * W + d];
        if (d == 0) {                                 // the pivot: not > 0 (or NaN) = not positive definite at this column
          if (!(v > 0.0)) atomicMin(&first_bad, c + 1);
          v = sqrt(v);
        }
        L[(long)d * M + c] = v;
      }
    __syncthreads();
  }
  if (info && tid == 0) *info = first_bad == 0x7fffffff ? 0 : first_bad;
}

// S = band((L L^T)^-1) backwards, the factor in the LDS and overwritten column by column:
//   S(i, j) = ([i = j] / L_jj - sum_{p = j+1 .. j+k} L(p, j) S(max(p, i), min(p, i))) / L_jj,   i = j+k .. j.
// Every lane holds the symmetric k x k window S(j+1 .. j+k, j+1 .. j+k) in registers (column c in slot c mod k: the new column j takes
// the slot of column j + k, which is read for the last time while column j is formed); the loop is unrolled k times.
// Any M: segments of `seg_blocks` blocks of k columns from the top, a segment's columns of L (+ the one below it, read ahead) in the LDS.
template <int K>
__global__ __launch_bounds__(256) void takahashi_lds_kernel(const double* __restrict__ L, double* __restrict__ S, int M, int seg_blocks) {
  extern __shared__ double bs_lds[];                  // [column - lbase][K+1] (as above)
  constexpr int W = K + 1;
  const int tid = threadIdx.x;
  double sw[K][K];                                    // sw[a][b] = S(row in slot a, column in slot b), both orders kept
#pragma unroll
  for (int a = 0; a < K; ++a)
#pragma unroll
    for (int b = 0; b < K; ++b) sw[a][b] = 0.0;
  // column j - 1 of L and its reciprocal diagonal are fetched while column j is formed (L is only read: nothing on the dependent chain)
  double ln[W];
  int lbase = 0;
  auto fetch = [&](int jn) __attribute__((always_inline)) {      // (issued at the top of a column - the scheduler otherwise sinks the reads to
    const int jc = jn >= 0 ? jn - lbase : 0;                     // their first use and the wave waits out the LDS latency every column)
#pragma unroll
    for (int d = 0; d <= K; ++d) ln[d] = bs_lds[(long)jc * W + d];
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  // a column of L enters the recurrence as  m_0 = 1 / L_jj^2,  m_d = -L(j + d, j) / L_jj:  S(j + d, j) = sum_c m_c S(j + c, j + d),
  // S(j, j) = m_0 + sum_c m_c S(j + c, j) - formed at staging by the whole workgroup (per column 10 instructions off the single wave)
  for (int d = 0; d <= K; ++d) ln[d] = L[(long)d * M + M - 1];  // the first column walked, M - 1
  {
    const double inv = band_rcp(ln[0]);
    ln[0] = inv * inv;
#pragma unroll
    for (int d = 1; d <= K; ++d) ln[d] = -ln[d] * inv;
  }
  const int nblk = (M + K - 1) / K;
  for (int b1 = nblk; b1 > 0; b1 -= seg_blocks) {
    const int b0 = b1 - seg_blocks > 0 ? b1 - seg_blocks : 0;
    const int c0 = b0 * K, c1 = b1 * K < M ? b1 * K : M;
    lbase = c0 - 1 > 0 ? c0 - 1 : 0;
    for (int d = 0; d <= K; ++d)
      for (int c = lbase + tid; c < c1; c += blockDim.x) {
        const double inv = band_rcp(L[c]);
        bs_lds[(long)(c - lbase) * W + d] = d == 0 ? inv * inv : -L[(long)d * M + c] * inv;
      }
    __syncthreads();
    if (tid < 64) {
      auto column = [&](int j, auto jm_c, auto check_c) __attribute__((always_inline)) {
        constexpr int jm = decltype(jm_c)::value;     // j mod K
        constexpr bool CHECK = decltype(check_c)::value;
        double l[W];
#pragma unroll
        for (int d = 0; d <= K; ++d) l[d] = ln[d];
        if (CHECK) {
#pragma unroll
          for (int d = 1; d <= K; ++d) l[d] = (j + d < M) ? l[d] : 0.0;
        }
        fetch(j - 1);
        double sn[W];                                 // the new column: sn[d] = S(j + d, j)
#pragma unroll
        for (int d = K; d >= 1; --d) {                // i = j + d: sum over p = j+1 .. j+k of m_{p-j} S(p, i), all from the window
          double acc = l[1] * sw[(jm + 1) % K][(jm + d) % K];
#pragma unroll
          for (int c = 2; c <= K; ++c) acc = fma(l[c], sw[(jm + c) % K][(jm + d) % K], acc);
          sn[d] = acc;
        }
        {
          double acc = l[0];                          // i = j: 1 / L_jj^2 + sum_p m_{p-j} S(p, j), with the entries just formed
#pragma unroll
          for (int c = 1; c <= K; ++c) acc = fma(l[c], sn[c], acc);
          sn[0] = acc;
        }
#pragma unroll
        for (int d = 0; d <= K; ++d) bs_lds[(long)(j - lbase) * W + d] = (!CHECK || j + d < M) ? sn[d] : 0.0;
        // column j into the window: slot jm = j mod K (it held column j + K: (jm + K) % K), rows j .. j + K - 1
        sw[jm][jm] = sn[0];
#pragma unroll
        for (int d = 1; d < K; ++d) { sw[(jm + d) % K][jm] = sn[d]; sw[jm][(jm + d) % K] = sn[d]; }
      };
      int jb = (b1 - 1) * K;
      for (; jb >= c0 && jb + 2 * K > M; jb -= K) {   // the last blocks: rows beyond the matrix
        [&]<int... JR>(std::integer_sequence<int, JR...>) { ((jb + (K - 1 - JR) < M ? column(jb + (K - 1 - JR), std::integral_constant<int, K - 1 - JR>{}, std::true_type{}) : (void)0), ...); }(std::make_integer_sequence<int, K>{});
      }
      for (; jb >= c0; jb -= K) {
        [&]<int... JR>(std::integer_sequence<int, JR...>) { (column(jb + (K - 1 - JR), std::integral_constant<int, K - 1 - JR>{}, std::false_type{}), ...); }(std::make_integer_sequence<int, K>{});
      }
    }
    __syncthreads();
    for (int d = 0; d <= K; ++d)
      for (int c = c0 + tid; c < c1; c += blockDim.x) S[(long)d * M + c] = bs_lds[(long)(c - lbase) * W + d];
    __syncthreads();
  }
}

template <int K>
__global__ __launch_bounds__(64) void takahashi_kernel(const double* L, double* S, int M) {
  takahashi_sweep<double, K, false>(BandPtr<double>{L, nullptr}, BandOut<double>{S, nullptr}, M, nullptr, nullptr);
}
template <int K>
__global__ __launch_bounds__(64) void trsv_kernel(const double* L, int M, const double* B, double* X, long D, int trans) {
  const long d = blockIdx.x;  // one wave per right-hand-side column
  if (trans) trsv_sweep<K, true>(L, M, B + d, X + d, D);
  else trsv_sweep<K, false>(L, M, B + d, X + d, D);
}

struct KuuCoefs { double c[ASVGP_MAX_KUU_TERMS]; double dc[ASVGP_MAX_KUU_TERMS]; int n; };

// inducing_features.py:16-44: Kuu = sum_t c_t * S_t, accumulated left to right like the reference (no FMA
// contraction, so the band is bit-identical to the numpy/TF evaluation order).
__global__ void kuu_assemble_kernel(const double* __restrict__ S, KuuCoefs cf, long E, double* __restrict__ Kuu,
                                    double* __restrict__ dK) {
  // (HIP's __dmul_rn / __dadd_rn are plain * and +: without the pragma the compiler contracts them into fma and the band is no
  // longer the reference's rounding sequence - found in round 2 through a 1-ulp knot mismatch in the Phi kernel)
#pragma clang fp contract(off)
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double acc = cf.c[0] * S[e];
  double dacc = cf.dc[0] * S[e];
  for (int t = 1; t < cf.n; ++t) {
    double s = S[(long)t * E + e];
    acc = acc + cf.c[t] * s;
    dacc = dacc + cf.dc[t] * s;
  }
  Kuu[e] = acc;
  if (dK) dK[e] = dacc;
}

// out (u,l) band of A^T from the (l,u) band of A: out[(l + j - i), i] = A[i][j]
__global__ void transpose_band_kernel(const double* __restrict__ in, double* __restrict__ out, long M, int l, int u) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)(l + u + 1) * M) return;
  int r = (int)(t / M);      // output row
  long i = t - (long)r * M;  // output column = source row index
  long j = i + r - l;        // r = l + j - i
  double v = 0.0;
  if (j >= 0 && j < M) v = in[(long)(u + i - j) * M + j];
  out[t] = v;
}

__global__ void symmetrise_band_kernel(const double* __restrict__ in, double* __restrict__ out, long M, int l) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)(2 * l + 1) * M) return;
  int r = (int)(t / M);
  long j = t - (long)r * M;
  double v;
  if (r >= l) v = in[(long)(r - l) * M + j];                     // lower part: row offset d = r-l, A[j+d][j]
  else { int d = l - r; long c = j - d; v = (c >= 0) ? in[(long)d * M + c] : 0.0; }  // A[j-d][j] = A[j][j-d]
  out[t] = v;
}

__global__ void unpack_band_kernel(const double* __restrict__ band, double* __restrict__ dense, long M, int l, int u) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * M) return;
  long i = t / M, j = t - i * M;
  long d = i - j;
  dense[t] = (d <= l && -d <= u) ? band[(long)(u + d) * M + j] : 0.0;
}

__global__ void pack_band_kernel(const double* __restrict__ dense, double* __restrict__ band, long M, int l, int u) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)(l + u + 1) * M) return;
  int r = (int)(t / M);
  long j = t - (long)r * M;
  long i = j + r - u;
  band[t] = (i >= 0 && i < M) ? dense[i * M + j] : 0.0;
}

__global__ void product_band_band_kernel(const double* __restrict__ L, const double* __restrict__ R,
                                         double* __restrict__ out, long M, int ll, int lu, int rl, int ru, int ol,
                                         int ou) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)(ol + ou + 1) * M) return;
  int r = (int)(t / M);
  long j = t - (long)r * M;
  long i = j + r - ou;
  double acc = 0.0;
  if (i >= 0 && i < M) {
    long p0 = i - ll; if (j - ru > p0) p0 = j - ru; if (p0 < 0) p0 = 0;
    long p1 = i + lu; if (j + rl < p1) p1 = j + rl; if (p1 > M - 1) p1 = M - 1;
    for (long p = p0; p <= p1; ++p)
      acc = fma(L[(long)(lu + i - p) * M + p], R[(long)(ru + p - j) * M + j], acc);
  }
  out[t] = acc;
}

__global__ __launch_bounds__(1024) void band_trace_sym_kernel(const double* __restrict__ S, const double* __restrict__ A,
                                                              long M, int k, double* __restrict__ out) {
  __shared__ double scratch[16];
  double acc = 0.0;
  for (long j = threadIdx.x; j < M; j += blockDim.x) {
    double a = S[j] * A[j];
    for (int d = 1; d <= k; ++d) a = fma(2.0 * S[(long)d * M + j], A[(long)d * M + j], a);
    acc += a;
  }
  double tot = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[0] = tot;
}

template <template <int> class Launcher, typename... Args>
static int dispatch_k(int k, Args... args) {
  switch (k) {
    case 1: return Launcher<1>::run(args...);
    case 2: return Launcher<2>::run(args...);
    case 3: return Launcher<3>::run(args...);
    case 4: return Launcher<4>::run(args...);
    case 5: return Launcher<5>::run(args...);
    case 6: return Launcher<6>::run(args...);
    case 7: return Launcher<7>::run(args...);
    case 8: return Launcher<8>::run(args...);
    default: set_error("bandwidth %d outside 1..%d", k, (int)ASVGP_MAX_BANDWIDTH); return ASVGP_ERR_UNSUPPORTED;
  }
}
template <int K> struct CholLauncher {
  static int run(const double* A, double* L, int M, int* info, hipStream_t st) {
    constexpr int W = K + 1;
    static const bool lds_off = getenv("ASVGP_BAND_OPS_LDS") && atoi(getenv("ASVGP_BAND_OPS_LDS")) == 0;      // (0: the register-window sweep always)
    static const int seg_env = getenv("ASVGP_BAND_OPS_SEG_BLOCKS") ? atoi(getenv("ASVGP_BAND_OPS_SEG_BLOCKS")) : 0;     // (tests: short segments)
    if (!lds_off && M > K + 1 && M < (1 << 30)) {
      int cap = (156 * 1024) / (8 * W);               // columns the LDS holds
      int seg = M <= cap ? (M + W - 1) / W : (cap - W) / W;     // one segment when the band fits, else cap - (k+1) columns per segment
      if (seg_env >= 1 && seg_env < seg) { seg = seg_env; cap = seg * W + W; }
      else if (M <= cap) cap = M;
      const size_t lds_bytes = sizeof(double) * (size_t)W * (size_t)cap;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(cholesky_band_lds_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess) {
        hipLaunchKernelGGL(cholesky_band_lds_kernel<K>, dim3(1), dim3(256), lds_bytes, st, A, L, M, info, seg);
        return check_launch("cholesky_band (band in the LDS)");
      }
      (void)hipGetLastError();
    }
    hipLaunchKernelGGL(cholesky_band_kernel<K>, dim3(1), dim3(64), 0, st, A, L, M, info);
    return check_launch("cholesky_band");
  }
};
template <int K> struct TakaLauncher {
  static int run(const double* L, double* S, int M, hipStream_t st) {
    constexpr int W = K + 1;
    static const bool lds_off = getenv("ASVGP_BAND_OPS_LDS") && atoi(getenv("ASVGP_BAND_OPS_LDS")) == 0;
    static const int seg_env = getenv("ASVGP_BAND_OPS_SEG_BLOCKS") ? atoi(getenv("ASVGP_BAND_OPS_SEG_BLOCKS")) : 0;     // (tests: short segments)
    if (!lds_off && M > 2 * K + 1 && M < (1 << 30)) {
      int cap = (156 * 1024) / (8 * W);               // columns the LDS holds
      int seg = M <= cap ? (M + K - 1) / K : (cap - 1) / K;     // blocks of k columns: one segment when the band fits, else cap - 1 columns per segment
      if (seg_env >= 1 && seg_env < seg) { seg = seg_env; cap = seg * K + 1; }
      else if (M <= cap) cap = M;
      const size_t lds_bytes = sizeof(double) * (size_t)W * (size_t)cap;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(takahashi_lds_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess) {
        hipLaunchKernelGGL(takahashi_lds_kernel<K>, dim3(1), dim3(256), lds_bytes, st, L, S, M, seg);
        return check_launch("inverse_from_cholesky_band (band in the LDS)");
      }
      (void)hipGetLastError();
    }
    hipLaunchKernelGGL(takahashi_kernel<K>, dim3(1), dim3(64), 0, st, L, S, M);
    return check_launch("inverse_from_cholesky_band");
  }
};
template <int K> struct TrsvLauncher {
  static int run(const double* L, int M, const double* B, double* X, long D, int trans, hipStream_t st) {
    hipLaunchKernelGGL(trsv_kernel<K>, dim3((unsigned)D), dim3(64), 0, st, L, M, B, X, D, trans);
    return check_launch("solve_triang_mat");
  }
};

}  // namespace asvgp

using namespace asvgp;

extern "C" int asvgp_version(void) { return 100; }
extern "C" const char* asvgp_last_error_string(void) { return g_err; }
extern "C" const char* asvgp_status_name(int s) {
  switch (s) {
    case ASVGP_OK: return "ASVGP_OK";
    case ASVGP_ERR_BAD_ARG: return "ASVGP_ERR_BAD_ARG";
    case ASVGP_ERR_UNSUPPORTED: return "ASVGP_ERR_UNSUPPORTED";
    case ASVGP_ERR_LDS_CAPACITY: return "ASVGP_ERR_LDS_CAPACITY";
    case ASVGP_ERR_WORKSPACE: return "ASVGP_ERR_WORKSPACE";
    case ASVGP_ERR_HIP: return "ASVGP_ERR_HIP";
    default: return "ASVGP_ERR_UNKNOWN";
  }
}

extern "C" int asvgp_matern_coeffs(int kind, double v, double l, double* c, double* dc, int* n_terms) {
  if (!c || !dc || !n_terms || !(v > 0.0) || !(l > 0.0)) { set_error("matern_coeffs: bad argument"); return ASVGP_ERR_BAD_ARG; }
  const double s3 = sqrt(3.0), s5 = sqrt(5.0);
  // term order: A, B, C, D, BC, BC_grad, BC_ggrad, BC_ggrad_none, BC_none_ggrad (those the kernel uses)
  // expressions written exactly as inducing_features.py:17-42 writes them (same rounding sequence)
  switch (kind) {
    case ASVGP_MATERN12:
      c[0] = 1 / (2 * l * v);      dc[0] = -1 / (2 * l * l * v);
      c[1] = l / (2 * v);          dc[1] = 1 / (2 * v);
      c[2] = 1 / (2 * v);          dc[2] = 0.0;
      *n_terms = 3;
      return ASVGP_OK;
    case ASVGP_MATERN32:
      c[0] = s3 / (4 * l * v);             dc[0] = -s3 / (4 * l * l * v);
      c[1] = l / (2 * s3 * v);             dc[1] = 1 / (2 * s3 * v);
      c[2] = pow(l, 3) / (12 * s3 * v);     dc[2] = 3 * l * l / (12 * s3 * v);
      c[3] = 1 / (2 * v);                  dc[3] = 0.0;
      c[4] = pow(l, 2) / (2 * v);          dc[4] = l / v;
      *n_terms = 5;
      return ASVGP_OK;
    case ASVGP_MATERN52:
      c[0] = (3 * s5) / (16 * l * v);                  dc[0] = -(3 * s5) / (16 * l * l * v);
      c[1] = (9 * l) / (16 * s5 * v);                  dc[1] = 9 / (16 * s5 * v);
      c[2] = (9 * pow(l, 3)) / (80 * s5 * v);          dc[2] = (27 * l * l) / (80 * s5 * v);
      c[3] = (3 * pow(l, 5)) / (400 * s5 * v); dc[3] = (15 * l * l * l * l) / (400 * s5 * v);
      c[4] = 9 / (16 * v);                             dc[4] = 0.0;
      c[5] = (3 * pow(l, 2)) / (10 * v);                 dc[5] = (6 * l) / (10 * v);
      c[6] = (9 * pow(l, 4)) / (400 * v);         dc[6] = (36 * l * l * l) / (400 * v);
      c[7] = (3 * pow(l, 2)) / (80 * v);                 dc[7] = (6 * l) / (80 * v);
      c[8] = (3 * pow(l, 2)) / (80 * v);                 dc[8] = (6 * l) / (80 * v);
      *n_terms = 9;
      return ASVGP_OK;
    default:
      set_error("matern_coeffs: unknown kernel kind %d", kind);
      return ASVGP_ERR_UNSUPPORTED;
  }
}

extern "C" int asvgp_kuu_assemble(const double* static_bands, int n_terms, const double* coef, const double* dcoef,
                                  int64_t M, int k, double* Kuu, double* dKuu_dl, asvgp_stream_t stream) {
  if (!static_bands || !coef || !Kuu || n_terms < 1 || n_terms > ASVGP_MAX_KUU_TERMS || M < 1 || k < 0) {
    set_error("kuu_assemble: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  KuuCoefs cf;
  cf.n = n_terms;
  for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) {
    cf.c[t] = t < n_terms ? coef[t] : 0.0;
    cf.dc[t] = (t < n_terms && dcoef) ? dcoef[t] : 0.0;
  }
  long E = (long)(k + 1) * M;
  hipLaunchKernelGGL(kuu_assemble_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, as_stream(stream),
                     static_bands, cf, E, Kuu, dKuu_dl);
  return check_launch("kuu_assemble");
}

static int band_args_ok(const void* a, const void* b, int64_t M, int k, const char* who) {
  if (!a || !b || M < 1) { set_error("%s: bad argument", who); return ASVGP_ERR_BAD_ARG; }
  if (M > 0x3fffffff) { set_error("%s: M too large", who); return ASVGP_ERR_UNSUPPORTED; }
  if (k < 1 || k > ASVGP_MAX_BANDWIDTH) { set_error("%s: bandwidth %d outside 1..%d", who, k, (int)ASVGP_MAX_BANDWIDTH); return ASVGP_ERR_UNSUPPORTED; }
  return ASVGP_OK;
}

extern "C" int asvgp_cholesky_band(const double* K, double* L, int64_t M, int k, int* info, asvgp_stream_t stream) {
  int rc = band_args_ok(K, L, M, k, "cholesky_band");
  if (rc) return rc;
  return dispatch_k<CholLauncher>(k, K, L, (int)M, info, as_stream(stream));
}

extern "C" int asvgp_inverse_from_cholesky_band(const double* L, double* S, int64_t M, int k, asvgp_stream_t stream) {
  int rc = band_args_ok(L, S, M, k, "inverse_from_cholesky_band");
  if (rc) return rc;
  return dispatch_k<TakaLauncher>(k, L, S, (int)M, as_stream(stream));
}

extern "C" int asvgp_solve_triang_mat(const double* L, const double* B, double* X, int64_t M, int k, int64_t D,
                                      int transpose_left, asvgp_stream_t stream) {
  int rc = band_args_ok(L, B, M, k, "solve_triang_mat");
  if (rc) return rc;
  if (!X || D < 1 || D > 65535) { set_error("solve_triang_mat: bad argument"); return ASVGP_ERR_BAD_ARG; }
  return dispatch_k<TrsvLauncher>(k, L, (int)M, B, X, (long)D, transpose_left, as_stream(stream));
}

// ---------------------------------------------------------------------------------------------------------
// Reverse mode of the four operators the reference's bound differentiates through (gpr.py:56-75; banded_matrices registers these
// gradients for its TF ops).  The two recurrences are the adjoints of the column loops.  Single-thread sweeps (below) are the general
// fallback (5.6 / 7.9 ms at M = 2048, k = 4); the wave-parallel forms further down take over whenever (k + 1)^2 <= 64 and the two
// arrays the recurrence walks fit the LDS: 0.85 / 1.09 ms; since round 4 the lane-uniform register-window forms at the end of this
// section run first, for any M (0.22 / 0.29 ms; the forward operators: 0.21 / 0.20 ms).  The training path
// of this library is still the fused asvgp_elbo_grad_1d (one launch, analytic gradient); these make a per-op binding usable.
// ---------------------------------------------------------------------------------------------------------
__device__ void chol_vjp_sweep(const double* L, double* Lb, double* Kb, int M, int k) {
  for (int j = M - 1; j >= 0; --j) {
    const int hi = (j + k < M - 1) ? j + k : M - 1;
    const double ljj = L[j];
    for (int i = hi; i >= j; --i) {
      double sb;
      if (i == j) sb = Lb[j] / (2.0 * ljj);
      else {
        const double lb = Lb[(long)(i - j) * M + j];
        sb = lb / ljj;
        Lb[j] -= lb * L[(long)(i - j) * M + j] / ljj;
      }
      Kb[(long)(i - j) * M + j] = sb;
      for (int p = (i - k > 0 ? i - k : 0); p < j; ++p) {
        const double lip = L[(long)(i - p) * M + p], ljp = L[(long)(j - p) * M + p];
        Lb[(long)(i - p) * M + p] -= sb * ljp;
        Lb[(long)(j - p) * M + p] -= sb * lip;
      }
    }
  }
}
__global__ __launch_bounds__(256) void band_cholesky_vjp_kernel(const double* __restrict__ L, const double* __restrict__ Lbar, double* __restrict__ Kbar,
                                                                double* __restrict__ work, int M, int k, int use_lds) {
  extern __shared__ double sh[];
  const long E = (long)(k + 1) * M;
  double* Lw = use_lds ? sh : const_cast<double*>(L);
  double* Lb = use_lds ? sh + E : work;
  for (long e = threadIdx.x; e < E; e += blockDim.x) {
    if (use_lds) Lw[e] = L[e];
    Lb[e] = Lbar[e];
    Kbar[e] = 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) chol_vjp_sweep(Lw, Lb, Kbar, M, k);
}

__device__ void taka_vjp_sweep(const double* L, const double* S, double* Sb, double* Lbo, int M, int k) {
  for (int j = 0; j < M; ++j) {
    const int hi = (j + k < M - 1) ? j + k : M - 1;
    const double ljj = L[j];
    double lb[ASVGP_MAX_BANDWIDTH + 1];
    for (int d = 0; d <= k; ++d) lb[d] = 0.0;
    for (int i = j; i <= hi; ++i) {
      const double sb = Sb[(long)(i - j) * M + j];
      const double accb = sb / ljj;
      lb[0] -= sb * S[(long)(i - j) * M + j] / ljj;
      if (i == j) lb[0] -= accb / (ljj * ljj);
      for (int p = j + 1; p <= hi; ++p) {
        const long o = (p >= i) ? (long)(p - i) * M + i : (long)(i - p) * M + p;     // symmetric in-band entry (p, i)
        lb[p - j] -= accb * S[o];
        Sb[o] -= accb * L[(long)(p - j) * M + j];
      }
    }
    for (int d = 0; d <= k; ++d) Lbo[(long)d * M + j] = (j + d < M) ? lb[d] : 0.0;
  }
}
__global__ __launch_bounds__(256) void band_takahashi_vjp_kernel(const double* __restrict__ L, const double* __restrict__ S, const double* __restrict__ Sbar,
                                                                 double* __restrict__ Lbar, double* __restrict__ work, int M, int k, int use_lds) {
  extern __shared__ double sh[];
  const long E = (long)(k + 1) * M;
  double* Lw = use_lds ? sh : const_cast<double*>(L);
  const double* Sw = S;                               // (read-only: stays in global memory so that L and the adjoint fit the LDS at M = 2048)
  double* Sb = use_lds ? sh + E : work;
  for (long e = threadIdx.x; e < E; e += blockDim.x) {
    if (use_lds) Lw[e] = L[e];
    Sb[e] = Sbar[e];
  }
  __syncthreads();
  if (threadIdx.x == 0) taka_vjp_sweep(Lw, Sw, Sb, Lbar, M, k);
}

// ---- wave-parallel forms: ONE wavefront walks the columns; lane (a, b) = (lane / (K+1), lane % (K+1)) owns one
// (row offset, column offset) pair of the (K+1) x (K+1) window the column touches, so a column costs three dependent LDS round trips
// instead of (K+1) K read-modify-write chains on one thread.  (K+1)^2 <= 64 lanes, i.e. K <= 7; both arrays the recurrence walks sit
// in the LDS (M <= 2048 at K = 4).  Anything else takes the single-thread sweeps above.
__device__ __forceinline__ double vjp_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}
template <int K>
__global__ __launch_bounds__(256) void band_cholesky_vjp_wave_kernel(const double* __restrict__ L, const double* __restrict__ Lbar, double* __restrict__ Kbar, int M) {
  extern __shared__ double sh[];
  const int E = (K + 1) * M;
  double* Ls = sh;                                             // L, read-only; row 0 holds 1 / diag(L) (the recurrence needs the diagonal only as a divisor)
  double* Lb = sh + E;                                         // adjoint of L, updated in place
  for (int e = threadIdx.x; e < E; e += blockDim.x) { const double l = L[e]; Ls[e] = e < M ? vjp_rcp(l) : l; Lb[e] = Lbar[e]; Kbar[e] = 0.0; }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x, a = lane / (K + 1), b = lane % (K + 1);
  const bool active = lane < (K + 1) * (K + 1);
  for (int j = M - 1; j >= 0; --j) {
    double lv[K + 1], lbv[K + 1], sb[K + 1];
#pragma unroll
    for (int c = 0; c <= K; ++c) {                             // (uniform addresses: broadcast reads)
      const bool in = j + c < M;
      lv[c] = Ls[(in ? c : 0) * M + j];
      lbv[c] = in ? Lb[c * M + j] : 0.0;
    }
    // the window entry this lane updates, and the L entries it needs, do not depend on this column's adjoints: requested with the column
    const int p = j - b;
    const bool upd = active && b >= 1 && p >= 0 && (a == 0 || (a + b <= K && j + a < M));
    const int trow = (a == 0) ? b : (a + b <= K ? a + b : 0), pcl = p >= 0 ? p : 0;
    const double cur = Lb[trow * M + pcl];
    double lw[K + 1];
#pragma unroll
    for (int c = 0; c <= K; ++c) lw[c] = Ls[((c + b <= K) ? c + b : 0) * M + pcl];   // L[c + b][p]; lw[0] = L[b][p]
    const double inv = lv[0];
    double diag = lbv[0];
#pragma unroll
    for (int c = K; c >= 1; --c) diag = fma(-lbv[c] * lv[c], inv, diag);
    sb[0] = 0.5 * diag * inv;
#pragma unroll
    for (int c = 1; c <= K; ++c) sb[c] = lbv[c] * inv;
    double sba = sb[0];
#pragma unroll
    for (int c = 1; c <= K; ++c) sba = (a == c) ? sb[c] : sba;
    if (active && b == 0 && j + a < M) Kbar[a * M + j] = sba;
    // one instruction stream for both kinds of lane: a >= 1 subtracts sb_a L[b][p] from (a + b, p); a = 0 collects the column's whole
    // contribution to (b, p): sum_c sb_c L[c + b][p], plus sb_0 L[b][p] once more (the i = j iteration touches (b, p) twice)
    double acc = cur;
#pragma unroll
    for (int c = 0; c <= K; ++c) {
      const bool use = (a == 0) ? (c + b <= K && j + c < M) : (c == 0);
      const double coef = (a == 0) ? sb[c] : sba;
      acc = fma(-(use ? coef : 0.0), lw[c], acc);
    }
    acc = fma(-((a == 0) ? sb[0] : 0.0), lw[0], acc);
    if (upd) Lb[trow * M + pcl] = acc;
  }
}

template <int K>
__global__ __launch_bounds__(256) void band_takahashi_vjp_wave_kernel(const double* __restrict__ L, const double* __restrict__ S, const double* __restrict__ Sbar,
                                                                      double* __restrict__ Lbar, double* __restrict__ work, int M) {
  extern __shared__ double sh[];
  const int E = (K + 1) * M;
  double* Ss = sh;                                             // S, read-only
  double* Sb = sh + E;                                         // adjoint of S, updated in place
  for (int e = threadIdx.x; e < E; e += blockDim.x) { Ss[e] = S[e]; Sb[e] = Sbar[e]; }
  for (int e = threadIdx.x; e < M; e += blockDim.x) work[e] = vjp_rcp(L[e]);    // 1 / diag(L), off the dependent chain
  __syncthreads();                                             // (also drains the stores to work: re-read below by wave 0 of this workgroup)
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x, a = lane / (K + 1), b = lane % (K + 1);
  const bool active = lane < (K + 1) * (K + 1);
  // column j of L: lane c <= K fetches L[c][j], lane K + 1 the reciprocal diagonal, four columns ahead (L stays in global memory: it is
  // read once per column, S and the adjoint are not)
  auto lload = [&](int col) -> double {
    const bool in = lane <= K && col < M && col + lane < M;
    const bool iv = lane == K + 1 && col < M;
    const double* src = iv ? work + col : L + (long)(in ? lane : 0) * M + (in ? col : 0);
    const double v = *src;
    return (in || iv) ? v : 0.0;
  };
  double pf[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) pf[s2] = lload(s2);
  for (int j0 = 0; j0 < M; j0 += 4) {
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int j = j0 + s2;
      if (j >= M) break;
      const double mine = pf[s2];
      pf[s2] = lload(j + 4);
      double lv[K + 1], sbu[K + 1], accb[K + 1], sw[K + 1];
#pragma unroll
      for (int c = 0; c <= K; ++c) {
        const bool in = j + c < M;
        lv[c] = readlane_f64(mine, c);
        sbu[c] = in ? Sb[c * M + j] : 0.0;
        // S(j + a, j + c): what lane (a, 0) needs for Lbar[a][j]
        const bool in2 = in && j + a < M;
        const int d = a > c ? a - c : c - a, lo = a > c ? c : a;
        sw[c] = Ss[(in2 ? d : 0) * M + j + (in2 ? lo : 0)];
        sw[c] = in2 ? sw[c] : 0.0;
      }
      const bool updl = active && b >= 1 && a >= b && j + a < M;
      const int tcol = (j + b < M) ? j + b : j;
      const double cur = Sb[(a >= b ? a - b : 0) * M + tcol];
      const double inv = readlane_f64(mine, K + 1);
      accb[0] = sbu[0] * inv;
#pragma unroll
      for (int c = 1; c <= K; ++c) { sbu[c] = fma(-accb[0], lv[c], sbu[c]); accb[c] = sbu[c] * inv; }   // (i = j first: it updates the rest of column j)
      double accb_a = accb[0], accb_b = accb[0], lv_a = lv[0], lv_b = lv[0];
#pragma unroll
      for (int c = 1; c <= K; ++c) { accb_a = (a == c) ? accb[c] : accb_a; accb_b = (b == c) ? accb[c] : accb_b; lv_a = (a == c) ? lv[c] : lv_a; lv_b = (b == c) ? lv[c] : lv_b; }
      if (active && b == 0) {                                   // Lbar[a][j] = - sum_c accb_c S(j + a, j + c)  [ - accb_0 / l_jj^2 on the diagonal ]
        double out = (a == 0) ? -accb[0] * inv * inv : 0.0;
#pragma unroll
        for (int c = 0; c <= K; ++c) out = fma(-accb[c], sw[c], out);
        Lbar[(long)a * M + j] = (j + a < M) ? out : 0.0;
      }
      if (updl) {                                               // entry (u, v) = (j + a, j + b) of the adjoint of S
        double upd = accb_b * lv_a;
        if (a != b) upd = fma(accb_a, lv_b, upd);
        Sb[(a - b) * M + j + b] = cur - upd;
      }
    }
  }
}
// Adjoint of the band Cholesky in the lane-uniform register-window form of cholesky_band_lds_kernel (round 4): L and the incoming
// adjoint in the LDS (2 (k+1) M doubles), ONE wave, every lane the same window - the k+1 columns j-k .. j of L and of the running
// adjoint, slot = column mod (k+1).  A column's adjoint entries are final when the column is reached, so the running adjoint lives in
// registers only: it is read from the LDS once, when the column enters the window, and never written back.
//   for j = M-1 .. 0:  i = j+k .. j:   sb = Lb(i,j) / L_jj  (i = j: Lb(j,j) / (2 L_jj), after Lb(j,j) -= sum_i Lb(i,j) L(i,j) / L_jj)
//                      Kb(i,j) = sb;   for p = i-k .. j-1:  Lb(i,p) -= sb L(j,p),  Lb(j,p) -= sb L(i,p)
// Any M: the columns are walked in SEGMENTS of `seg_blocks` blocks of k+1 columns; a segment's columns of L and of the incoming adjoint
// (+ the k+1 columns below it, which enter the window while it is walked) are staged into the LDS by the whole workgroup, the window
// registers carry over from segment to segment, the segment's results leave through the LDS.  M (k+1) 16 B <= 160 KB is one segment.
template <int K>
__global__ __launch_bounds__(256) void band_cholesky_vjp_lds_kernel(const double* __restrict__ L, const double* __restrict__ Lbar, double* __restrict__ Kbar, int M,
                                                                    int seg_blocks, int cap_cols) {
  extern __shared__ double sh[];
  constexpr int W = K + 1;
  double* Ls = sh;                                    // L, [column - lbase][K+1]: a column's entries side by side
  double* Bs = sh + (long)cap_cols * W;               // incoming adjoint, overwritten by the result column by column
  const int tid = threadIdx.x;
  double wl[W][W], wb[W][W];                          // slot s = column mod (k+1): wl[s][d] = L(p + d, p), wb[s][d] = running adjoint of it
  const int nblk = (M + W - 1) / W;
  {                                                   // the window of the first column walked: columns M-1 .. M-1-k
    const int jtop = M - 1;
#pragma unroll
    for (int c = 0; c <= K; ++c) {
      const int p = jtop - c;
#pragma unroll
      for (int sl = 0; sl <= K; ++sl)
        if (((p % W) + W) % W == sl) {
#pragma unroll
          for (int d = 0; d <= K; ++d) {
            wl[sl][d] = (p >= 0) ? L[(long)d * M + (p >= 0 ? p : 0)] : 0.0;
            wb[sl][d] = (p >= 0) ? Lbar[(long)d * M + (p >= 0 ? p : 0)] : 0.0;
          }
          wl[sl][0] = (p >= 0) ? band_rcp(wl[sl][0]) : 1.0;       // (entry 0 of a column of L is kept as 1 / L_pp: only ever a divisor)
        }
    }
  }
  for (int b1 = nblk; b1 > 0; b1 -= seg_blocks) {
    const int b0 = b1 - seg_blocks > 0 ? b1 - seg_blocks : 0;
    const int c0 = b0 * W, c1 = b1 * W < M ? b1 * W : M;
    const int lbase = c0 - W > 0 ? c0 - W : 0;        // first staged column
    for (int d = 0; d <= K; ++d)
      for (int c = lbase + tid; c < c1; c += blockDim.x) {     // (the reciprocal diagonal formed here, by the whole workgroup)
        const double v = L[(long)d * M + c];
        Ls[(long)(c - lbase) * W + d] = d == 0 ? band_rcp(v) : v;
        Bs[(long)(c - lbase) * W + d] = Lbar[(long)d * M + c];
      }
    __syncthreads();
    if (tid < 64) {
      auto column = [&](int j, auto jm_c, auto check_c) __attribute__((always_inline)) {
        constexpr int jm = decltype(jm_c)::value;     // j mod (K+1)
        constexpr bool CHECK = decltype(check_c)::value;
        const double inv = wl[jm][0];
        double kb[W];
#pragma unroll
        for (int d = K; d >= 0; --d) {                // i = j + d, descending
          const bool rowok = !CHECK || j + d < M;
          double sb;
          if (d == 0) sb = 0.5 * wb[jm][0] * inv;
          else {
            const double lb = rowok ? wb[jm][d] : 0.0;
            sb = lb * inv;
            wb[jm][0] = fma(-sb, wl[jm][d], wb[jm][0]);
          }
          kb[d] = rowok ? sb : 0.0;
#pragma unroll
          for (int c = 1; c + d <= K; ++c) {          // p = j - c >= i - k
            const int sp = ((jm - c) % W + W) % W;
            if (!CHECK || (j - c >= 0 && rowok)) {
              const double lip = wl[sp][d + c], ljp = wl[sp][c];
              if (d != 0) {
                wb[sp][d + c] = fma(-sb, ljp, wb[sp][d + c]);
                wb[sp][c] = fma(-sb, lip, wb[sp][c]);
              } else {
                wb[sp][c] = fma(-2.0 * sb, ljp, wb[sp][c]);    // i = j: the two updates hit the same entry (j, p)
              }
            }
          }
        }
#pragma unroll
        for (int d = 0; d <= K; ++d) Bs[(long)(j - lbase) * W + d] = kb[d];
        const int pn = j - K - 1;                     // the slot takes column j - k - 1 (its incoming adjoint is still untouched in the LDS;
#pragma unroll
        for (int d = 0; d <= K; ++d) {                // reading it at the top of the column instead was measured: slower)
          wl[jm][d] = (!CHECK || pn >= 0) ? Ls[(long)(pn >= 0 ? pn - lbase : 0) * W + d] : 0.0;
          wb[jm][d] = (!CHECK || pn >= 0) ? Bs[(long)(pn >= 0 ? pn - lbase : 0) * W + d] : 0.0;
        }
      };
      // blocks of k+1 columns, descending; the blocks that touch either end of the matrix carry the bounds tests
      int jb = (b1 - 1) * W;
      for (; jb >= c0 && jb + 2 * K + 1 >= M; jb -= W) {
        [&]<int... JR>(std::integer_sequence<int, JR...>) { ((jb + (K - JR) < M ? column(jb + (K - JR), std::integral_constant<int, K - JR>{}, std::true_type{}) : (void)0), ...); }(std::make_integer_sequence<int, W>{});
      }
      for (; jb >= c0 && jb - K - 1 >= 0; jb -= W) {  // (no bounds tests, no branches)
        [&]<int... JR>(std::integer_sequence<int, JR...>) { (column(jb + (K - JR), std::integral_constant<int, K - JR>{}, std::false_type{}), ...); }(std::make_integer_sequence<int, W>{});
      }
      for (; jb >= c0; jb -= W) {
        [&]<int... JR>(std::integer_sequence<int, JR...>) { ((jb + (K - JR) < M ? column(jb + (K - JR), std::integral_constant<int, K - JR>{}, std::true_type{}) : (void)0), ...); }(std::make_integer_sequence<int, W>{});
      }
    }
    __syncthreads();
    for (int d = 0; d <= K; ++d)
      for (int c = c0 + tid; c < c1; c += blockDim.x) Kbar[(long)d * M + c] = Bs[(long)(c - lbase) * W + d];
    __syncthreads();
  }
}

// Adjoint of the band-restricted inverse in the same form (round 4).  The recurrence only READS its three inputs, in ascending column
// order; its running state is the adjoint of S inside the symmetric (k+1) x (k+1) window a column touches:
//   for j = 0 .. M-1, i = j .. j+k:  accb = Sb(i,j) / L_jj;  lb_0 -= accb S(i,j)  [- accb / L_jj^2 at i = j]
//                                     for p = j+1 .. j+k:  lb_{p-j} -= accb S(p,i),  Sb(p,i) -= accb L(p,j)      (symmetric entries)
// A lone wave issues one fp64 instruction per ~8 cycles (all four operator kernels measure instructions x 8 cycles), so the column's
// 56 operations are split over TWO waves on different SIMDs with a one-directional hand-over: wave 0 carries the state (the window
// of Sb, at [r mod (k+1)][c mod (k+1)]; Sb's incoming rows from the LDS, L from global memory one block of k+1 columns ahead) and
// leaves the k+1 factors accb of every column in an LDS array; wave 1 follows a block behind, holds the same window of S (from global
// memory, a block ahead) and forms the results lb from the factors.  Wave 0 never waits for wave 1; wave 1 polls a progress counter
// once per block (it sits in the one slot of the factor array that belongs to no (row, column): (M - 1 + k, M - 1)).
// Any M: SEGMENTS of `seg_blocks` blocks of k+1 columns, as in the adjoint of the band Cholesky above - the rows of the incoming adjoint
// that enter the window while the segment is walked and the segment's factor / result columns are what the LDS holds; the register
// windows, the blocks fetched ahead and the progress counter carry over.  Both waves and the two idle ones pass the same barriers.
template <int K>
__global__ __launch_bounds__(256) void band_takahashi_vjp_lds_kernel(const double* __restrict__ L, const double* __restrict__ S, const double* __restrict__ Sbar,
                                                                     double* __restrict__ Lbar, double* __restrict__ work, int M, int seg_blocks, int cap_rows,
                                                                     int cap_fq) {
  extern __shared__ double sh[];
  constexpr int W = K + 1;
  double* Bs = sh;                                    // incoming adjoint of S by ROWS: (r, r - d) at [(r - rbase) W + d] (a row enters the window at a time)
  double* Fq = sh + (long)cap_rows * W;               // accb(j + a, j) at [(j - c0) W + a], then the column's results
  int* prog = reinterpret_cast<int*>(Fq + cap_fq - 1);   // columns wave 0 has finished (a slot no column uses: see the launcher)
  const int tid = threadIdx.x;
  const int nblk = (M + W - 1) / W;
  int c0 = 0, c1 = 0, rbase = 0;
  // a segment's rows of the incoming adjoint: those that enter the window while it is walked (the first segment: the first window too)
  auto stage = [&](int b0) __attribute__((always_inline)) {
    const int b1 = b0 + seg_blocks < nblk ? b0 + seg_blocks : nblk;
    c0 = b0 * W;
    c1 = b1 * W < M ? b1 * W : M;
    rbase = b0 == 0 ? 0 : c0 + W;
    const int rend = c1 + W < M ? c1 + W : M;
    for (int d = 0; d <= K; ++d)
      for (int r = rbase + tid; r < rend; r += blockDim.x)
        if (r - d >= 0) Bs[(long)(r - rbase) * W + d] = Sbar[(long)d * M + (r - d)];
  };
  auto copy_out = [&]() __attribute__((always_inline)) {
    for (int d = 0; d <= K; ++d)
      for (int c = c0 + tid; c < c1; c += blockDim.x) Lbar[(long)d * M + c] = (c + d < M) ? Fq[(long)(c - c0) * W + d] : 0.0;
  };
  if (tid == 0) *prog = 0;
  // 1 / L_jj (wave 0) and 1 / L_jj^2 (wave 1) by the whole workgroup, into the caller's scratch: 5-6 instructions per column off each wave
  for (int c = tid; c < M; c += blockDim.x) { const double inv = band_rcp(L[c]); work[c] = inv; work[(long)M + c] = inv * inv; }
  __syncthreads();
  int vz = 0;
  asm volatile("" : "+v"(vz));                        // (an opaque zero: keeps the global loads on the vector memory counter, apart from the LDS reads)
  const double* Lv = L + vz;
  const double* Sv = S + vz;
  const double* Wv = work + vz;
  if (tid < 64) {
    // ---------------- wave 0: the window of Sb, the factors
    double bw[W][W];
    // two buffers of k+1 columns of L (entry [c][d] = L(j + d, j), j = block start + c; [c][0] = 1 / L_jj): one holds the block being walked,
    // the other receives the next one; the roles alternate block by block (two blocks per loop trip: no register copies)
    double lA[W][W], lB[W][W];
    auto fetch = [&](int j0, auto check_c, auto& lq) __attribute__((always_inline)) {
      constexpr bool CHECK = decltype(check_c)::value;
#pragma unroll
      for (int c = 0; c <= K; ++c)
#pragma unroll
        for (int d = 0; d <= K; ++d) {
          const bool in = !CHECK || j0 + c + d < M;
          lq[c][d] = d == 0 ? Wv[in ? j0 + c : 0] : Lv[in ? (long)d * M + j0 + c : 0];   // (entry 0: 1 / L_jj)
          if (!in) lq[c][d] = 0.0;
        }
    };
    auto column = [&](int j, auto jm_c, auto check_c, auto& lc) __attribute__((always_inline)) {
      constexpr int jm = decltype(jm_c)::value;       // j mod (K+1)
      constexpr bool CHECK = decltype(check_c)::value;
      double bn[W];
      const int rn = j + K + 1;                       // the row that enters after this column: entries (rn, j + 1 + b) = band row k - b
#pragma unroll
      for (int b = 0; b <= K; ++b) {
        const bool in = !CHECK || rn < M;
        bn[b] = Bs[in ? (long)(rn - rbase) * W + (K - b) : 0];
        if (!in) bn[b] = 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);              // (reads issued here, not sunk to the end of the column)
      const double inv = lc[jm][0];
#pragma unroll
      for (int a = 0; a <= K; ++a) {                  // i = j + a
        const int rs = (jm + a) % W;
        const bool rowok = !CHECK || j + a < M;
        const double accb = (rowok ? bw[rs][jm] : 0.0) * inv;
        if (rowok) Fq[(long)(j - c0) * W + a] = accb;
#pragma unroll
        for (int c = 1; c <= K; ++c) {                // p = j + c
          const int hs = (jm + (c > a ? c : a)) % W, ls = (jm + (c > a ? a : c)) % W;
          if (!CHECK || j + c < M) bw[hs][ls] = fma(-accb, lc[jm][c], bw[hs][ls]);
        }
      }
#pragma unroll
      for (int b = 0; b <= K; ++b) bw[jm][(jm + 1 + b) % W] = bn[b];
    };
    fetch(0, std::true_type{}, lA);
    for (int b0 = 0; b0 < nblk; b0 += seg_blocks) {
      stage(b0);
      __syncthreads();
      if (b0 == 0) {
#pragma unroll
        for (int a = 0; a <= K; ++a)
#pragma unroll
          for (int b = 0; b <= K; ++b) bw[a][b] = (b <= a && a < M) ? Bs[(long)a * W + (a - b)] : 0.0;
      }
      auto block = [&](int jb, auto check_c, auto& lc, auto& lq) __attribute__((always_inline)) {
        constexpr bool CHECK = decltype(check_c)::value;
        fetch(jb + W, check_c, lq);
        if constexpr (CHECK) {
          [&]<int... JR>(std::integer_sequence<int, JR...>) { ((jb + JR < M ? column(jb + JR, std::integral_constant<int, JR>{}, std::true_type{}, lc) : (void)0), ...); }(std::make_integer_sequence<int, W>{});
        } else {
          [&]<int... JR>(std::integer_sequence<int, JR...>) { (column(jb + JR, std::integral_constant<int, JR>{}, std::false_type{}, lc), ...); }(std::make_integer_sequence<int, W>{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(prog, jb + W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      // lA holds the block about to be walked whenever a loop is entered; an odd block at the end of a loop moves lB back into lA
      auto walk = [&](int& jb, auto check_c, auto more) __attribute__((always_inline)) {
        while (more(jb) && more(jb + W)) { block(jb, check_c, lA, lB); block(jb + W, check_c, lB, lA); jb += 2 * W; }
        if (more(jb)) {
          block(jb, check_c, lA, lB);
          jb += W;
#pragma unroll
          for (int c = 0; c <= K; ++c)
#pragma unroll
            for (int d = 0; d <= K; ++d) lA[c][d] = lB[c][d];
        }
      };
      int jb = c0;
      walk(jb, std::false_type{}, [&](int j) { return j < c1 && j + 3 * K + 3 < M; });   // (rows up to jb + 3k + 2 are touched by the next block's loads)
      walk(jb, std::true_type{}, [&](int j) { return j < c1; });
      __syncthreads();
      copy_out();
      __syncthreads();
    }
  } else if (tid < 128) {
    // ---------------- wave 1: the window of S, the results (they take the place of the column's factors in the LDS)
    double sw[W][W];
#pragma unroll
    for (int a = 0; a <= K; ++a)
#pragma unroll
      for (int b = 0; b <= K; ++b) {
        const bool in = b <= a && a < M;
        sw[a][b] = Sv[in ? (long)(a - b) * M + b : 0];
        if (!in) sw[a][b] = 0.0;
      }
    // two buffers of a block's entering rows and reciprocal diagonals ([c][b] = S(j + k + 1, j + 1 + b), [c] = 1 / L(j, j)^2, j = block start + c),
    // alternating like wave 0's
    double sA[W][W], sB[W][W], dA[W], dB[W];
    auto fetch = [&](int j0, auto check_c, auto& sq, auto& dq) __attribute__((always_inline)) {
      constexpr bool CHECK = decltype(check_c)::value;
#pragma unroll
      for (int c = 0; c <= K; ++c) {
        const bool inr = !CHECK || j0 + c + K + 1 < M;
#pragma unroll
        for (int b = 0; b <= K; ++b) {
          sq[c][b] = Sv[inr ? (long)(K - b) * M + j0 + c + 1 + b : 0];
          if (!inr) sq[c][b] = 0.0;
        }
        const bool ind = !CHECK || j0 + c < M;
        dq[c] = Wv[(long)M + (ind ? j0 + c : 0)];
      }
    };
    auto column = [&](int j, auto jm_c, auto check_c, auto& sc, auto& dc) __attribute__((always_inline)) {
      constexpr int jm = decltype(jm_c)::value;
      constexpr bool CHECK = decltype(check_c)::value;
      double accb[W];
#pragma unroll
      for (int a = 0; a <= K; ++a) {
        const bool rowok = !CHECK || j + a < M;
        accb[a] = Fq[rowok ? (long)(j - c0) * W + a : 0];
        if (!rowok) accb[a] = 0.0;
      }
      const double inv2 = dc[jm];                     // 1 / L_jj^2
      double lb[W];
#pragma unroll
      for (int d = 0; d <= K; ++d) lb[d] = 0.0;
#pragma unroll
      for (int a = 0; a <= K; ++a) {
        const int rs = (jm + a) % W;
        lb[0] = fma(-accb[a], sw[rs][jm], lb[0]);
        if (a == 0) lb[0] = fma(-accb[a], inv2, lb[0]);
#pragma unroll
        for (int c = 1; c <= K; ++c) {
          const int hs = (jm + (c > a ? c : a)) % W, ls = (jm + (c > a ? a : c)) % W;
          if (!CHECK || j + c < M) lb[c] = fma(-accb[a], sw[hs][ls], lb[c]);
        }
      }
#pragma unroll
      for (int d = 0; d <= K; ++d)
        if (!CHECK || j + d < M) Fq[(long)(j - c0) * W + d] = lb[d];   // (every lane: the same value to the same address)
#pragma unroll
      for (int b = 0; b <= K; ++b) sw[jm][(jm + 1 + b) % W] = sc[jm][b];
    };
    fetch(0, std::true_type{}, sA, dA);
    for (int b0 = 0; b0 < nblk; b0 += seg_blocks) {
      stage(b0);
      __syncthreads();
      auto block = [&](int jb, auto check_c, auto& sc, auto& dc, auto& sq, auto& dq) __attribute__((always_inline)) {
        constexpr bool CHECK = decltype(check_c)::value;
        fetch(jb + W, check_c, sq, dq);
        while (__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < jb + W) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if constexpr (CHECK) {
          [&]<int... JR>(std::integer_sequence<int, JR...>) { ((jb + JR < M ? column(jb + JR, std::integral_constant<int, JR>{}, std::true_type{}, sc, dc) : (void)0), ...); }(std::make_integer_sequence<int, W>{});
        } else {
          [&]<int... JR>(std::integer_sequence<int, JR...>) { (column(jb + JR, std::integral_constant<int, JR>{}, std::false_type{}, sc, dc), ...); }(std::make_integer_sequence<int, W>{});
        }
      };
      auto walk = [&](int& jb, auto check_c, auto more) __attribute__((always_inline)) {
        while (more(jb) && more(jb + W)) { block(jb, check_c, sA, dA, sB, dB); block(jb + W, check_c, sB, dB, sA, dA); jb += 2 * W; }
        if (more(jb)) {
          block(jb, check_c, sA, dA, sB, dB);
          jb += W;
#pragma unroll
          for (int c = 0; c <= K; ++c) {
            dA[c] = dB[c];
#pragma unroll
            for (int b = 0; b <= K; ++b) sA[c][b] = sB[c][b];
          }
        }
      };
      int jb = c0;
      walk(jb, std::false_type{}, [&](int j) { return j < c1 && j + 3 * K + 3 < M; });
      walk(jb, std::true_type{}, [&](int j) { return j < c1; });
      __syncthreads();
      copy_out();
      __syncthreads();
    }
  } else {
    for (int b0 = 0; b0 < nblk; b0 += seg_blocks) {   // the two idle waves stage and store with the others
      stage(b0);
      __syncthreads();
      __syncthreads();
      copy_out();
      __syncthreads();
    }
  }
}

template <int K> struct TakaVjpLdsLauncher {
  static int run(const double* L, const double* S, const double* Sbar, double* Lbar, double* work, int M, hipStream_t st) {
    if constexpr (K <= 6) {                           // (k = 7, 8: the windows and the blocks fetched ahead no longer fit the register file)
      constexpr int W = K + 1;
      const int nblk = (M + W - 1) / W;
      // one segment: rows and factor columns of the whole matrix, the counter in the slot of (row M - 1 + k, column M - 1), which no column
      // uses; several: seg W + W rows, seg W columns and one more double for the counter
      int seg, cap_rows, cap_fq;
      static const int seg_env = getenv("ASVGP_BAND_OPS_SEG_BLOCKS") ? atoi(getenv("ASVGP_BAND_OPS_SEG_BLOCKS")) : 0;   // (tests: short segments)
      if ((size_t)2 * M * W * sizeof(double) <= 160 * 1024 && !(seg_env >= 2 && seg_env < nblk)) { seg = nblk; cap_rows = M; cap_fq = M * W; }
      else {
        const int doubles = 160 * 1024 / (int)sizeof(double);
        seg = ((doubles - 1) / W - W) / (2 * W);
        if (seg_env >= 2 && seg_env < seg) seg = seg_env;
        cap_rows = seg * W + W;
        cap_fq = seg * W * W + 1;
      }
      if (seg < 2) return 1;
      const size_t bytes = sizeof(double) * ((size_t)cap_rows * W + (size_t)cap_fq);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_takahashi_vjp_lds_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
      hipLaunchKernelGGL(band_takahashi_vjp_lds_kernel<K>, dim3(1), dim3(256), bytes, st, L, S, Sbar, Lbar, work, M, seg, cap_rows, cap_fq);
      return check_launch("inverse_from_cholesky_band_vjp (register window)");
    } else {
      return 1;
    }
  }
};
template <int K> struct CholVjpLdsLauncher {
  static int run(const double* L, const double* Lbar, double* Kbar, int M, hipStream_t st) {
    constexpr int W = K + 1;
    int cap = (160 * 1024) / (16 * W);                // columns of both arrays the LDS holds
    int seg = M <= cap ? (M + W - 1) / W : (cap - W) / W;       // one segment when everything fits, else cap - (k+1) columns per segment
    static const int seg_env = getenv("ASVGP_BAND_OPS_SEG_BLOCKS") ? atoi(getenv("ASVGP_BAND_OPS_SEG_BLOCKS")) : 0;     // (tests: short segments)
    if (seg_env >= 2 && seg_env < seg) { seg = seg_env; cap = seg * W + W; }
    else if (M <= cap) cap = M;
    if (seg < 2) return 1;
    const size_t bytes = sizeof(double) * 2 * (size_t)cap * W;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_cholesky_vjp_lds_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
    hipLaunchKernelGGL(band_cholesky_vjp_lds_kernel<K>, dim3(1), dim3(256), bytes, st, L, Lbar, Kbar, M, seg, cap);
    return check_launch("cholesky_band_vjp (register window)");
  }
};
template <int K> struct CholVjpWaveLauncher {
  static int run(const double* L, const double* Lbar, double* Kbar, int M, size_t bytes, hipStream_t st) {
    if constexpr ((K + 1) * (K + 1) <= 64) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_cholesky_vjp_wave_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
      hipLaunchKernelGGL(band_cholesky_vjp_wave_kernel<K>, dim3(1), dim3(256), bytes, st, L, Lbar, Kbar, M);
      return check_launch("cholesky_band_vjp");
    } else {
      return 1;
    }
  }
};
template <int K> struct TakaVjpWaveLauncher {
  static int run(const double* L, const double* S, const double* Sbar, double* Lbar, double* work, int M, size_t bytes, hipStream_t st) {
    if constexpr ((K + 1) * (K + 1) <= 64) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_takahashi_vjp_wave_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
      hipLaunchKernelGGL(band_takahashi_vjp_wave_kernel<K>, dim3(1), dim3(256), bytes, st, L, S, Sbar, Lbar, work, M);
      return check_launch("inverse_from_cholesky_band_vjp");
    } else {
      return 1;
    }
  }
};

// out[d, j] = sign * sum_c U[j + d, c] V[j, c]   (d = 0..k): the lower band of sign * U V^T  (Lbar of the triangular solves)
__global__ void band_outer_kernel(const double* __restrict__ U, const double* __restrict__ V, long M, long D, int k, double sign, double* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)(k + 1) * M) return;
  const long d = e / M, j = e - d * M;
  double acc = 0.0;
  if (j + d < M)
    for (long c = 0; c < D; ++c) acc = fma(U[(j + d) * D + c], V[j * D + c], acc);
  out[e] = sign * acc;
}

extern "C" int asvgp_cholesky_band_vjp(const double* L, const double* Lbar, double* Kbar, double* work, int64_t M, int k, asvgp_stream_t stream) {
  int rc = band_args_ok(L, Lbar, M, k, "cholesky_band_vjp");
  if (rc) return rc;
  if (!Kbar || !work) { set_error("cholesky_band_vjp: bad argument"); return ASVGP_ERR_BAD_ARG; }
  const size_t bytes = sizeof(double) * 2 * (size_t)(k + 1) * (size_t)M;
  const int use_lds = bytes <= 160 * 1024;
  static const bool lds_off = getenv("ASVGP_BAND_OPS_LDS") && atoi(getenv("ASVGP_BAND_OPS_LDS")) == 0;
  if (!lds_off && M > 2 * (k + 1) && M < (1 << 30)) {           // lane-uniform register window (round 4), any M (segments)
    const int rcl = dispatch_k<CholVjpLdsLauncher>(k, L, Lbar, Kbar, (int)M, as_stream(stream));
    if (rcl != 1) return rcl;
  }
  if (use_lds && (k + 1) * (k + 1) <= 64) {                     // wave-parallel form
    const int rcw = dispatch_k<CholVjpWaveLauncher>(k, L, Lbar, Kbar, (int)M, bytes, as_stream(stream));
    if (rcw != 1) return rcw;
  }
  if (use_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_cholesky_vjp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
  }
  hipLaunchKernelGGL(band_cholesky_vjp_kernel, dim3(1), dim3(256), use_lds ? bytes : 0, as_stream(stream), L, Lbar, Kbar, work, (int)M, k, use_lds);
  return check_launch("cholesky_band_vjp");
}

extern "C" int asvgp_inverse_from_cholesky_band_vjp(const double* L, const double* S, const double* Sbar, double* Lbar, double* work, int64_t M,
                                                    int k, asvgp_stream_t stream) {
  int rc = band_args_ok(L, S, M, k, "inverse_from_cholesky_band_vjp");
  if (rc) return rc;
  if (!Sbar || !Lbar || !work) { set_error("inverse_from_cholesky_band_vjp: bad argument"); return ASVGP_ERR_BAD_ARG; }
  const size_t bytes = sizeof(double) * 2 * (size_t)(k + 1) * (size_t)M;
  const int use_lds = bytes <= 160 * 1024;
  static const bool lds_off = getenv("ASVGP_BAND_OPS_LDS") && atoi(getenv("ASVGP_BAND_OPS_LDS")) == 0;
  if (!lds_off && M > 2 * (k + 1) && M < (1 << 30)) {           // lane-uniform register windows on two waves (round 4), any M (segments)
    const int rcl = dispatch_k<TakaVjpLdsLauncher>(k, L, S, Sbar, Lbar, work, (int)M, as_stream(stream));
    if (rcl != 1) return rcl;
  }
  if (use_lds && (k + 1) * (k + 1) <= 64) {                     // wave-parallel form
    const int rcw = dispatch_k<TakaVjpWaveLauncher>(k, L, S, Sbar, Lbar, work, (int)M, bytes, as_stream(stream));
    if (rcw != 1) return rcw;
  }
  if (use_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_takahashi_vjp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
  }
  hipLaunchKernelGGL(band_takahashi_vjp_kernel, dim3(1), dim3(256), use_lds ? bytes : 0, as_stream(stream), L, S, Sbar, Lbar, work, (int)M, k, use_lds);
  return check_launch("inverse_from_cholesky_band_vjp");
}

extern "C" int asvgp_band_outer_product(const double* U, const double* V, int64_t M, int64_t D, int k, double sign, double* out, asvgp_stream_t stream) {
  if (!U || !V || !out || M < 1 || D < 1 || k < 0 || k > ASVGP_MAX_BANDWIDTH) { set_error("band_outer_product: bad argument"); return ASVGP_ERR_BAD_ARG; }
  const long total = (long)(k + 1) * M;
  hipLaunchKernelGGL(band_outer_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), U, V, (long)M, (long)D, k, sign, out);
  return check_launch("band_outer_product");
}

#define EW_LAUNCH(kern, total, ...)                                                                        \
  hipLaunchKernelGGL(kern, dim3((unsigned)(((total) + 255) / 256)), dim3(256), 0, as_stream(stream), __VA_ARGS__)

extern "C" int asvgp_product_band_band(const double* left, const double* right, double* out, int64_t M, int ll,
                                       int lu, int rl, int ru, int ol, int ou, asvgp_stream_t stream) {
  if (!left || !right || !out || M < 1 || ll < 0 || lu < 0 || rl < 0 || ru < 0 || ol < 0 || ou < 0) {
    set_error("product_band_band: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  EW_LAUNCH(product_band_band_kernel, (long)(ol + ou + 1) * M, left, right, out, (long)M, ll, lu, rl, ru, ol, ou);
  return check_launch("product_band_band");
}

extern "C" int asvgp_transpose_band(const double* in, double* out, int64_t M, int lower, int upper,
                                    asvgp_stream_t stream) {
  if (!in || !out || in == out || M < 1 || lower < 0 || upper < 0) { set_error("transpose_band: bad argument"); return ASVGP_ERR_BAD_ARG; }
  EW_LAUNCH(transpose_band_kernel, (long)(lower + upper + 1) * M, in, out, (long)M, lower, upper);
  return check_launch("transpose_band");
}

extern "C" int asvgp_symmetrise_band(const double* in, double* out, int64_t M, int lower, asvgp_stream_t stream) {
  if (!in || !out || in == out || M < 1 || lower < 0) { set_error("symmetrise_band: bad argument"); return ASVGP_ERR_BAD_ARG; }
  EW_LAUNCH(symmetrise_band_kernel, (long)(2 * lower + 1) * M, in, out, (long)M, lower);
  return check_launch("symmetrise_band");
}

extern "C" int asvgp_unpack_banded_matrix_to_dense(const double* band, double* dense, int64_t M, int lower, int upper,
                                                   asvgp_stream_t stream) {
  if (!band || !dense || M < 1 || lower < 0 || upper < 0) { set_error("unpack_banded_matrix_to_dense: bad argument"); return ASVGP_ERR_BAD_ARG; }
  EW_LAUNCH(unpack_band_kernel, (long)M * M, band, dense, (long)M, lower, upper);
  return check_launch("unpack_banded_matrix_to_dense");
}

extern "C" int asvgp_pack_dense_matrix_to_banded(const double* dense, double* band, int64_t M, int lower, int upper,
                                                 asvgp_stream_t stream) {
  if (!band || !dense || M < 1 || lower < 0 || upper < 0) { set_error("pack_dense_matrix_to_banded: bad argument"); return ASVGP_ERR_BAD_ARG; }
  EW_LAUNCH(pack_band_kernel, (long)(lower + upper + 1) * M, dense, band, (long)M, lower, upper);
  return check_launch("pack_dense_matrix_to_banded");
}

extern "C" int asvgp_band_trace_sym(const double* S, const double* A, int64_t M, int k, double* out,
                                    asvgp_stream_t stream) {
  if (!S || !A || !out || M < 1 || k < 0) { set_error("band_trace_sym: bad argument"); return ASVGP_ERR_BAD_ARG; }
  hipLaunchKernelGGL(band_trace_sym_kernel, dim3(1), dim3(1024), 0, as_stream(stream), S, A, (long)M, k, out);
  return check_launch("band_trace_sym");
}
