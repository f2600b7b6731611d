// Phi pass: B-spline design matrix evaluation fused with the banded sufficient statistics.
//
// Replaces (reference, HJakeCunningham/ASVGP): basis.py:51-76 evaluate_basis (index search, piece
// polynomials, COO->CSR), gpr.py:41-44 (Kuf@y, Kuf@Kuf.T, sparse_to_band, sum y^2).  Phi is never
// materialised: each point's (k+1) non-zeros are formed in registers and scattered straight into an
// LDS-resident private band [(k+1) x cols | rhs | mesh table]; one flush per workgroup, then a
// cross-workgroup tree/atomic reduce into the packed stats buffer.
//
// HBM roofline: 16 B/point (x and y read once, fp64).  Everything else is on-chip.
#include <stdlib.h>

#include "handle.hpp"

namespace asvgp {

constexpr int PHI_THREADS = 1024;
constexpr int PHI_CH = 32;                    // iterations a wavefront stays on one contiguous slice (see phi_accumulate_kernel)
constexpr int PHI_MAX_BLOCKS = 256;           // one 1024-thread workgroup per CU (LDS-limited)
constexpr size_t PHI_LDS_BUDGET = 160 * 1024 - 512;

// ---- fixed-point band accumulation (phi algorithm 3) -------------------------------------------------------------
// ds_add_f64 retires one wave-instruction per ~21 cycles under random addresses, ds_add_u64 per ~11 (the plain
// ds_write_b64 rate: tools/micro/lds_atomic_rate.hip).  The band products v_i v_j are non-negative and bounded by a
// compile-time constant per sub-diagonal, so they are accumulated as 64-bit integers: product * 2^(s0 + g_d), where
// 2^-g_d bounds the products of diagonal d and s0 = 62 - ceil(log2(points per workgroup)) keeps the per-workgroup sum
// below 2^62.  Rounding error per addend <= 2^-(s0+1) of the diagonal's largest product (s0 >= 42 for N <= 2^20 per
// workgroup, 47 at the north-star size), unbiased, and the sums are order-independent (bit-reproducible per workgroup).
// Conversion is one v_add_f64 with the magic constant C_d = 1.5 * 2^(52 - s0 - g_d): the low mantissa bits of
// (p + C_d) are round-to-nearest(p * 2^(s0+g_d)); C_d has a zero low word, so only the high words are subtracted.
template <int K> struct FxTab {
  int g[K + 1];
  constexpr FxTab() : g{} {
    double cf[K + 1][K + 1] = {};
    for (int i = 0; i <= K; ++i)
      for (int p = 0; p <= K; ++p) cf[i][p] = piece_coef<K, 0>(i, p);
    for (int d = 0; d <= K; ++d) {
      double mx = 0.0;
      for (int i = 0; i + d <= K; ++i)
        for (int q = 0; q <= 32; ++q) {
          const double t = q / 32.0;
          double a = cf[i][K], b = cf[i + d][K];
          for (int p = K - 1; p >= 0; --p) { a = a * t + cf[i][p]; b = b * t + cf[i + d][p]; }
          if (a * b > mx) mx = a * b;
        }
      double bound = mx * 1.25;    // grid maximum + margin; one further spare bit is kept in s0
      int gd = 0;
      while (bound * 2.0 <= 1.0 && gd < 40) { bound *= 2.0; ++gd; }
      g[d] = gd;
    }
  }
};
template <int K> struct FxCoef { static constexpr FxTab<K> tab{}; };

template <int K> __device__ __forceinline__ int fx_chi(int s0, int d) {   // wave-uniform (SALU)
  return ((1075 - s0 - FxCoef<K>::tab.g[d]) << 20) | 0x80000;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_step_add_u64(unsigned long long v) {
  unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, true);
  unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, true);
  return v + (((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned long long wave_sum_dpp_u64(unsigned long long v) {
  v = dpp_step_add_u64<0x111, 0xf>(v);
  v = dpp_step_add_u64<0x112, 0xf>(v);
  v = dpp_step_add_u64<0x114, 0xf>(v);
  v = dpp_step_add_u64<0x118, 0xf>(v);
  v = dpp_step_add_u64<0x142, 0xa>(v);
  v = dpp_step_add_u64<0x143, 0xc>(v);
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// Phi y in fixed point too: the scale follows the data - y0 = 2^E bounds |y| over the workgroup's first tile (x4 margin);
// a later point with |y| > y0 (or NaN) takes the fp64 atomic path into a second rhs plane, so the result never depends
// on the guess.  chi_r: magic-constant high word for the scale 2^(s0 - E).
struct FxParams { int s0; int chi_r; double y0; };

template <int K, bool FX>
__device__ __forceinline__ void phi_scatter(double t, double yv, int cb, int ncols, bool do_band, double* band, double* rhs, FxParams fx) {
  const int s0 = fx.s0;
  double v[K + 1];
  bspline_pieces<K>(t, v);
  if (FX) {
    if (fabs(yv) <= fx.y0) {
#pragma unroll
      for (int i = 0; i <= K; ++i)
        lds_add_u64(reinterpret_cast<unsigned long long*>(rhs) + cb + K - i, fx_convert(v[i] * yv, fx.chi_r));
    } else {
#pragma unroll
      for (int i = 0; i <= K; ++i) lds_add(rhs + ncols + cb + K - i, v[i] * yv);
    }
  } else {
#pragma unroll
    for (int i = 0; i <= K; ++i) lds_add(rhs + cb + K - i, v[i] * yv);
  }
  if (do_band) {
#pragma unroll
    for (int i = 0; i <= K; ++i)
#pragma unroll
      for (int j = i; j <= K; ++j) {  // row_i = idx+K-i >= row_j = idx+K-j: sub-diagonal d = j-i, column row_j
        if (FX) lds_add_u64(reinterpret_cast<unsigned long long*>(band) + (j - i) * ncols + cb + K - j,
                            fx_convert(v[i] * v[j], fx_chi<K>(s0, j - i)));
        else lds_add(band + (j - i) * ncols + cb + K - j, v[i] * v[j]);
      }
  }
}

// Two points per lane (NP = 2, a 16-B pair) or one (NP = 1).  Must be called by whole wavefronts (wave-wide votes).
// Time-series / sorted inputs put a whole wavefront into ONE cell: 64 same-address LDS atomics would serialise
// (1.0 ms for the sorted N = 10M case).  When every point of the wave sits in the same cell, the 20 products are
// summed in-lane over the pair, reduced across the wave on the VALU (DPP) and committed by one lane.
// Run-length accumulator of a wavefront that stays inside ONE cell (sorted / time-series input): products are summed
// in-lane across iterations and reduced across the wave only when the cell changes (run_flush).
template <int K> struct RunAcc {
  int cell;   // wave-uniform; -1 = empty
  double r[K + 1];
  double b[(K + 1) * (K + 2) / 2];
};
template <int K> __device__ __forceinline__ void run_clear(RunAcc<K>& A) {
  A.cell = -1;
#pragma unroll
  for (int i = 0; i <= K; ++i) A.r[i] = 0.0;
#pragma unroll
  for (int e = 0; e < (K + 1) * (K + 2) / 2; ++e) A.b[e] = 0.0;
}
// wave-uniform call.  The band sums leave as fixed point through a full double -> int64 conversion (a run's sum exceeds
// the magic-constant range of fx_convert; the conversion runs once per run, on one lane's worth of values).
template <int K, bool FX>
__device__ __forceinline__ void run_flush(RunAcc<K>& A, int cell0, int ncols, bool do_band, double* band, double* rhs, FxParams fx) {
  const int c = __builtin_amdgcn_readfirstlane(A.cell);
  if (c < 0) return;
  const int cb = c - cell0;
  const bool commit = (threadIdx.x & 63) == 0;
  int e = 0;
#pragma unroll
  for (int i = 0; i <= K; ++i) {
    const double r = wave_sum_dpp(A.r[i]);
    if (commit) lds_add(rhs + (FX ? ncols : 0) + cb + K - i, r);   // (fp64 plane: a run's sum is not bounded by y0)
    A.r[i] = 0.0;
#pragma unroll
    for (int j = i; j <= K; ++j) {
      if (do_band) {
        const double t = wave_sum_dpp(A.b[e]);
        if (commit) {
          if (FX) lds_add_u64(reinterpret_cast<unsigned long long*>(band) + (j - i) * ncols + cb + K - j,
                              (unsigned long long)__double2ll_rn(ldexp(t, fx.s0 + FxCoef<K>::tab.g[j - i])));
          else lds_add(band + (j - i) * ncols + cb + K - j, t);
        }
      }
      A.b[e] = 0.0;
      ++e;
    }
  }
  A.cell = -1;
}

// One batch = NP points per lane (a 16-B pair or a single point).  Whole wavefronts only (wave-wide votes).
template <int NP> struct Batch {
  double x[NP], y[NP];
  int idx[NP];
  bool in[NP];
  int idx0;       // wave-uniform: cell of lane 0's first point
  bool uniform;   // wave-uniform: every point of the wave lies in cell idx0 (and inside this column chunk)
};
template <int NP>
__device__ __forceinline__ void classify(Batch<NP>& B, bool valid, const double* mesh, int n_mesh, double m0, double inv_delta,
                                         int cell0, int cell1, double& yy) {
  bool all_in = true;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    B.idx[q] = valid ? neighbour_index(B.x[q], mesh, n_mesh, m0, inv_delta) : -1;
    // a point outside the mesh (cell coordinate beyond [0, 1] by more than the 2 % that float32-linspace meshes wobble) or NaN would
    // leave the fixed-point products' compile-time bounds and wrap the int64 image: it is not accumulated and reported loudly
    // (NaN y^T y), like the moment kernel does - the models check (a, b) first, a direct C-ABI caller gets this
    const double tq = valid ? (B.x[q] - mesh[B.idx[q] < 0 ? 0 : B.idx[q]]) * inv_delta : 0.5;
    const bool inside = tq >= -0.02 && tq <= 1.02;
    yy = inside ? yy : __builtin_nan("");
    B.in[q] = valid && inside && B.idx[q] >= cell0 && B.idx[q] < cell1;
    all_in = all_in && B.in[q];
  }
  B.idx0 = __builtin_amdgcn_readfirstlane(B.idx[0]);
  bool same = all_in;
#pragma unroll
  for (int q = 0; q < NP; ++q) same = same && (B.idx[q] == B.idx0);
  B.uniform = __all(same);
}
template <int K, int NP, bool FX>
__device__ __forceinline__ void scatter_batch(const Batch<NP>& B, const double* mesh, double inv_delta, int cell0, int ncols,
                                              bool do_band, double* band, double* rhs, double& yy, FxParams fx) {
#pragma unroll
  for (int q = 0; q < NP; ++q)
    if (B.in[q]) {
      phi_scatter<K, FX>((B.x[q] - mesh[B.idx[q]]) * inv_delta, B.y[q], B.idx[q] - cell0, ncols, do_band, band, rhs, fx);
      yy = fma(B.y[q], B.y[q], yy);
    }
}
// uniform batch -> run accumulator (flushing first when the cell changes)
template <int K, int NP, bool FX>
__device__ __forceinline__ void run_batch(const Batch<NP>& B, RunAcc<K>& run, const double* mesh, double inv_delta, int cell0,
                                          int ncols, bool do_band, double* band, double* rhs, double& yy, FxParams fx) {
  if (B.idx0 != __builtin_amdgcn_readfirstlane(run.cell)) {
    run_flush<K, FX>(run, cell0, ncols, do_band, band, rhs, fx);
    run.cell = B.idx0;
  }
  const double u = mesh[B.idx0];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    double v[K + 1];
    bspline_pieces<K>((B.x[q] - u) * inv_delta, v);
    int e = 0;
#pragma unroll
    for (int i = 0; i <= K; ++i) {
      run.r[i] = fma(v[i], B.y[q], run.r[i]);
#pragma unroll
      for (int j = i; j <= K; ++j) { run.b[e] = fma(v[i], v[j], run.b[e]); ++e; }
    }
    yy = fma(B.y[q], B.y[q], yy);
  }
}

// workgroup-wide y scale from each thread's first value(s): y0 = 2^E >= 4 max|y| (2^-900 when the tile is all zero, NaN or inf)
__device__ __forceinline__ FxParams fx_scale(double my_abs, int s0, double* scratch) {
  double m = (my_abs == my_abs && my_abs < 1e300) ? my_abs : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmax(t, scratch[w]);
  __syncthreads();
  int E = (t > 0.0) ? ilogb(t) + 3 : -900;   // nothing to go by: every later non-zero y takes the exact fp64 plane
  if (E > 900) E = 900;
  if (E < -900) E = -900;
  FxParams fx;
  fx.s0 = s0;
  fx.chi_r = ((1075 - (s0 - E)) << 20) | 0x80000;
  fx.y0 = ldexp(1.0, E);
  return fx;
}

// One workgroup per CU; block b owns points [b*ppb, (b+1)*ppb).  VEC: 16-B loads of (x0,x1),(y0,y1).
template <int K, bool VEC, bool FX>
__global__ __launch_bounds__(PHI_THREADS) void phi_accumulate_kernel(
    const double* __restrict__ x, const double* __restrict__ y, long y_stride, long N,
    const double* __restrict__ mesh_g, int n_mesh, double inv_delta, int cell0, int cell1, int ncols,
    int do_band, double* __restrict__ partials, long ppb, double* __restrict__ zero_ptr, long zero_n, int s0) {
  extern __shared__ double lds[];
  // the packed stats buffer is zeroed here (it is only touched again by phi_reduce_kernel, after this kernel)
  if (zero_ptr) for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < zero_n; e += (long)gridDim.x * blockDim.x) zero_ptr[e] = 0.0;
  double* band = lds;                      // (K+1) x ncols
  double* rhs = band + (K + 1) * ncols;    // ncols (FX: fixed-point plane, followed by an fp64 plane for out-of-scale y)
  double* mesh = rhs + (FX ? 2 : 1) * ncols;   // n_mesh
  double* scratch = mesh + n_mesh;         // 16
  const int tid = threadIdx.x;
  const int E = (K + 2) * ncols;
  for (int e = tid; e < E + (FX ? ncols : 0); e += PHI_THREADS) lds[e] = 0.0;
  for (int e = tid; e < n_mesh; e += PHI_THREADS) mesh[e] = mesh_g[e];
  __syncthreads();
  const double m0 = mesh[0];
  const long beg = (long)blockIdx.x * ppb;
  long end = beg + ppb;
  if (end > N) end = N;
  double yy = 0.0;
  FxParams fx{s0, 0, 0.0};
  // Point -> wave mapping: the workgroup walks its range in super-tiles of 16 waves x PHI_CH iterations x 128 points; inside
  // a super-tile every wavefront streams a contiguous slice (PHI_CH * 128 points), so a sorted input keeps a wave inside
  // one cell for up to PHI_CH iterations.  (8 -> 32 iterations: sorted input 87 -> 79 us, unsorted unchanged; a two-deep
  // register prefetch was tried and gave nothing: the hand-over copy forces the older load to complete anyway.)
  const int lane = tid & 63, wv = tid >> 6;
  constexpr int NP = VEC ? 2 : 1;
  const long n_it = (ppb / (2 * PHI_THREADS)) * (VEC ? 1 : 2);   // rows of 64 lanes per wave (ppb is a multiple of 2 * PHI_THREADS)
  auto unit_of = [&](long it) -> long {   // row visited by this wave at iteration `it`
    const long nfull = n_it / PHI_CH, rem = n_it - nfull * PHI_CH;
    const long sup = it / PHI_CH, r = it - sup * PHI_CH;
    return (sup < nfull) ? (sup * (PHI_THREADS / 64) + wv) * PHI_CH + r     // full super-tile: slices of PHI_CH rows
                         : nfull * (PHI_THREADS / 64) * PHI_CH + wv * rem + r;   // last, shorter super-tile: slices of `rem` rows
  };
  const double2* x2 = reinterpret_cast<const double2*>(x);
  const double2* y2 = reinterpret_cast<const double2*>(y);
  const long ubeg = VEC ? (beg >> 1) : beg, uend = (end > beg) ? (VEC ? (end >> 1) : end) : 0;   // units: pairs or points
  double2 xa = make_double2(0.0, 0.0), ya = xa;   // the prefetched unit (VEC: pair; scalar: .x only)
  auto fetch = [&](long it) {
    const long u = ubeg + unit_of(it) * 64 + lane;
    if (it < n_it && u < uend) {
      if (VEC) { xa = x2[u]; ya = y2[u]; }
      else { xa.x = x[u]; ya.x = y[u * y_stride]; }
    }
  };
  fetch(0);
  if (FX) {
    double mine = (n_it > 0 && ubeg + unit_of(0) * 64 + lane < uend) ? fmax(fabs(ya.x), VEC ? fabs(ya.y) : 0.0) : 0.0;
    if (VEC && tid == 0 && (end & 1) && end > beg) mine = fmax(mine, fabs(y[end - 1]));   // the odd tail point counts too (N = 1!)
    fx = fx_scale(mine, s0, scratch);
  }
  Batch<NP> B;
  long it = 0;
  // take the prefetched unit as the current batch, start the next prefetch (before the LDS burst), classify
  auto advance = [&]() {
    const bool have = (ubeg + unit_of(it) * 64 + lane) < uend;
    B.x[0] = xa.x; B.y[0] = ya.x;
    if (VEC) { B.x[NP - 1] = xa.y; B.y[NP - 1] = ya.y; }
    fetch(it + 1);
    classify<NP>(B, have, mesh, n_mesh, m0, inv_delta, cell0, cell1, yy);
  };
  // Two loops so that the run accumulator (40 VGPRs) is live only while a wave actually is inside a run: scatter mode is the
  // steady state of unsorted input, run mode of sorted / time-series input; every batch is classified, so a wrong mode
  // costs time, never correctness.
  while (it < n_it) {   // wave-convergent throughout (votes, DPP)
    bool pending = false;
    for (; it < n_it; ++it) {   // ---- scatter mode
      advance();
      if (B.uniform) { pending = true; break; }
      scatter_batch<K, NP, FX>(B, mesh, inv_delta, cell0, ncols, do_band, band, rhs, yy, fx);
    }
    if (!pending) break;
    RunAcc<K> run;              // ---- run mode: the pending uniform batch starts a run
    run_clear<K>(run);
    for (;;) {
      run_batch<K, NP, FX>(B, run, mesh, inv_delta, cell0, ncols, do_band, band, rhs, yy, fx);
      ++it;
      if (it >= n_it) { pending = false; break; }
      advance();
      if (!B.uniform) break;    // (pending stays true: B holds a classified non-uniform batch)
    }
    run_flush<K, FX>(run, cell0, ncols, do_band, band, rhs, fx);
    if (pending) {
      scatter_batch<K, NP, FX>(B, mesh, inv_delta, cell0, ncols, do_band, band, rhs, yy, fx);
      ++it;
    }
  }
  if (VEC) {  // odd tail point (only the last block can have one); whole wave 0 enters, one lane is valid
    const bool tail = (end & 1) && end > beg;
    if (tail && tid < 64) {
      Batch<1> T1;
      T1.x[0] = (tid == 0) ? x[end - 1] : 0.0;
      T1.y[0] = (tid == 0) ? y[end - 1] : 0.0;
      classify<1>(T1, tid == 0, mesh, n_mesh, m0, inv_delta, cell0, cell1, yy);
      scatter_batch<K, 1, FX>(T1, mesh, inv_delta, cell0, ncols, do_band, band, rhs, yy, fx);
    }
  }
  double tot = block_sum(yy, scratch);  // contains the barrier that orders the LDS atomics before the flush
  __syncthreads();
  double* out = partials + (size_t)blockIdx.x * (E + 1);
  if (FX) {
    const int nb = (K + 1) * ncols;   // integer band image -> double (exact to 53 bits), then the power-of-two unscale
    for (int e = tid; e < E; e += PHI_THREADS) {
      double v = lds[e];
      if (e < nb) {
        const int d = e / ncols;
        int gd = FxCoef<K>::tab.g[0];
#pragma unroll
        for (int q = 1; q <= K; ++q) gd = (d == q) ? FxCoef<K>::tab.g[q] : gd;
        v = ldexp((double)(long long)reinterpret_cast<const unsigned long long*>(lds)[e], -(s0 + gd));
      } else {   // Phi y: fixed-point plane (scale 2^(s0 - E), E from chi_r) + fp64 plane
        const int sr = 1075 - (fx.chi_r >> 20);
        v = ldexp((double)(long long)reinterpret_cast<const unsigned long long*>(lds)[e], -sr) + lds[e + ncols];
      }
      out[e] = v;
    }
  } else {
    for (int e = tid; e < E; e += PHI_THREADS) out[e] = lds[e];
  }
  if (tid == 0) out[E] = tot;
}


}  // namespace asvgp
#include "phi_tables.hpp"
#include "phi_moments.hpp"
#include "phi_sort.hpp"
namespace asvgp {


// Sum the per-workgroup partials into the packed stats buffer (zeroed beforehand).
// grid = (ceil((E+1)/256), gsplit); each thread sums its slice of workgroups, then one fp64 global atomic.
// ranges != NULL: [workgroup][2] = first / last column partial g holds (the tile-sort kernels write only those; a time series leaves a
// workgroup a handful of columns) - entries outside are not read.
__global__ __launch_bounds__(256) void phi_reduce_kernel(const double* __restrict__ partials, int G, int ncols,
                                                         int K, int col0, long M, long D, int dcol, int do_band,
                                                         double* __restrict__ stats, const int* __restrict__ ranges = nullptr) {
  const int E1 = (K + 2) * ncols + 1;
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E1) return;
  int per = (G + gridDim.y - 1) / gridDim.y;
  int g0 = blockIdx.y * per, g1 = g0 + per;
  if (g1 > G) g1 = G;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int g = g0;
  if (ranges) {
    const int colr = e < (K + 2) * ncols ? e % ncols : -1;          // (-1: y^T y, which every workgroup writes)
    auto take = [&](int gg) -> double {                             // (the range words are wave-uniform: scalar loads)
      const int lo = ranges[2 * gg], hi = ranges[2 * gg + 1];
      return (colr < 0 || (colr >= lo && colr <= hi)) ? __builtin_nontemporal_load(partials + (size_t)gg * E1 + e) : 0.0;
    };
    for (; g + 3 < g1; g += 4) {
      const double v0 = take(g), v1 = take(g + 1), v2 = take(g + 2), v3 = take(g + 3);
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; g < g1; ++g) s0 += take(g);
  }
  for (; g + 3 < g1; g += 4) {
    s0 += __builtin_nontemporal_load(partials + (size_t)g * E1 + e);       // (read once: keep the 29 MB out of the L2's LRU order)
    s1 += __builtin_nontemporal_load(partials + (size_t)(g + 1) * E1 + e);
    s2 += __builtin_nontemporal_load(partials + (size_t)(g + 2) * E1 + e);
    s3 += __builtin_nontemporal_load(partials + (size_t)(g + 3) * E1 + e);
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
// posterior of one test point from the staged tables: mean = phi*^T alpha, var = v + phi*^T W phi*  (W = band(P^-1) - band(Kuu^-1))
template <int K>
__device__ __forceinline__ void predict_point(double xv, const double* mesh, int n_mesh, double m0, double inv_delta, int M,
                                              const double* alpha, const double* W, double variance, int D, long p,
                                              double* __restrict__ mean, double& var_out, double& mean0_out) {
  const int idx = neighbour_index(xv, mesh, n_mesh, m0, inv_delta);
  const double t = (xv - mesh[idx]) * inv_delta;
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
  var_out = fma(2.0, q, variance);
  if (D == 1) {
    double m = 0.0;
#pragma unroll
    for (int i = 0; i <= K; ++i) m = fma(v[i], alpha[idx + K - i], m);
    mean0_out = m;
  } else {
    for (int d = 0; d < D; ++d) {
      double m = 0.0;
#pragma unroll
      for (int i = 0; i <= K; ++i) m = fma(v[i], alpha[(long)(idx + K - i) * D + d], m);
      mean[p * D + d] = m;
    }
  }
}

// VEC4 (D == 1, 16-B aligned xnew / mean / var): two consecutive points per lane and iteration - one 16-B load, two
// 16-B stores - with the next iteration's load issued before the LDS work.  The scalar form kept one 8-B load per lane in
// flight (8 KB per CU): 0.6 TB/s of reads, 126 us for 10M points; latency-bound, not LDS-bound.
template <int K, bool VEC4, bool STAGE>
__global__ __launch_bounds__(1024) void predict_kernel(const double* __restrict__ xnew, long n,
                                                       const double* __restrict__ mesh_g, int n_mesh,
                                                       double inv_delta, int M, const double* __restrict__ alpha_g,
                                                       const double* __restrict__ W_g, double variance, int D,
                                                       int stage, double* __restrict__ mean,
                                                       double* __restrict__ var) {
  extern __shared__ double lds[];
  // STAGE is a template parameter so that the table pointers are LDS pointers at compile time: a run-time choice between the
  // LDS copy and the global arrays makes them generic and every table read a flat_load (no broadcast, aperture check).
  (void)stage;
  const double* W = STAGE ? lds : W_g;
  const double* alpha = STAGE ? lds + (K + 1) * M : alpha_g;
  const double* mesh = STAGE ? lds + (K + 2) * M : mesh_g;
  if (STAGE) {  // implies D == 1
    double* w = lds;
    double* a = w + (K + 1) * M;
    double* ms = a + M;
#pragma unroll 4
    for (int e = threadIdx.x; e < (K + 1) * M; e += blockDim.x) w[e] = W_g[e];
#pragma unroll 2
    for (int e = threadIdx.x; e < M; e += blockDim.x) a[e] = alpha_g[e];
#pragma unroll 2
    for (int e = threadIdx.x; e < n_mesh; e += blockDim.x) ms[e] = mesh_g[e];
    __syncthreads();
  }
  const double m0 = mesh[0];
  if (VEC4) {   // (name kept: vector path; two points per lane - four spilled 49 VGPRs and ran at half the speed)
    const double2* x2 = reinterpret_cast<const double2*>(xnew);
    double2* mean2 = reinterpret_cast<double2*>(mean);
    double2* var2 = reinterpret_cast<double2*>(var);
    const long nq = n >> 1;   // full pairs
    const long stride = (long)gridDim.x * blockDim.x;
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double2 xa = make_double2(0.0, 0.0);
    if (q < nq) xa = x2[q];
    for (; q < nq; q += stride) {
      const double xs[2] = {xa.x, xa.y};
      const long qn = q + stride;
      if (qn < nq) xa = x2[qn];   // next pair in flight under the LDS work
      double vo[2], mo[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) predict_point<K>(xs[u], mesh, n_mesh, m0, inv_delta, M, alpha, W, variance, 1, 0, nullptr, vo[u], mo[u]);
      var2[q] = make_double2(vo[0], vo[1]);
      mean2[q] = make_double2(mo[0], mo[1]);
    }
    const long p = 2 * nq + (long)blockIdx.x * blockDim.x + threadIdx.x;   // the odd last point
    if (p < n) {
      double vo, mo;
      predict_point<K>(xnew[p], mesh, n_mesh, m0, inv_delta, M, alpha, W, variance, 1, p, mean, vo, mo);
      var[p] = vo;
      mean[p] = mo;
    }
  } else {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
      double vo, mo = 0.0;
      predict_point<K>(xnew[p], mesh, n_mesh, m0, inv_delta, M, alpha, W, variance, D, p, mean, vo, mo);
      var[p] = vo;
      if (D == 1) mean[p] = mo;
    }
  }
}

// Posterior moments through a per-CELL polynomial table (D == 1, large n*, a mesh that is an exact numpy.linspace).
// Inside cell c the variance is a polynomial of degree 2k in the centred coordinate s = t - 1/2:
//   var(s) = v + sum_ij v_i(s) v_j(s) W_ij = sum_p Q_c[p] s^p,   Q_c[p] = [p = 0] v + sum_i pair[i][i][p] W_ii + 2 sum_{i<j} pair[i][j][p] W_ij
// (pair[i][j][.]: the exact integer-ratio product tables of the Phi pass, phi_tables.hpp).  Every workgroup builds the table of all cells
// into its LDS straight from band(W) - 15 (k = 4) coalesced reads and 9 x 15 FMAs per cell, ~2 us - next to alpha; a point then costs
// 2k+1 CONTIGUOUS table doubles (ds_read2_b64) + k+1 of alpha = 8 LDS instructions and 112 B, instead of the 23 scattered 8-byte reads
// (3 mesh, 15 W, 5 alpha; 184 B) of predict_kernel, and a Horner chain instead of the 15 products.  The cell comes from arithmetic on the
// verified linspace (no mesh table: at k = 4, M = 2048 the table and alpha fill 163 552 of the 163 840 B); a point within rounding of a
// knot, outside the mesh, or any point of a mesh that is not that linspace takes the exact table rule against the global mesh.
// The mean is the sum predict_kernel forms, from the same t.  8 B in, 16 B out per point.
template <int K> constexpr size_t predict_poly_lds_bytes(int n_mesh, int M) {
  return sizeof(double) * ((size_t)(n_mesh - 1) * (2 * K + 1) + (size_t)M + 1);
}
template <int K>
__global__ __launch_bounds__(1024) void predict_poly_kernel(const double* __restrict__ xnew, long n, const double* __restrict__ mesh_g, int n_mesh,
                                                            double inv_delta, int M, const double* __restrict__ alpha_g,
                                                            const double* __restrict__ W_g, double variance, int pairs_ok,
                                                            double* __restrict__ mean, double* __restrict__ var) {
  extern __shared__ double lds[];
  constexpr int NQ = 2 * K + 1;
  const int ncells = n_mesh - 1, tid = threadIdx.x;
  double* Q = lds;                                   // [ncells][NQ]
  double* alpha = lds + (size_t)ncells * NQ;         // [M]
  int* flag = reinterpret_cast<int*>(alpha + M);     // != 0: the mesh is not the linspace of its end points
  if (tid == 0) *flag = 0;
  typedef double pp_d2 __attribute__((ext_vector_type(2)));
  const long stride = (long)gridDim.x * blockDim.x;
  const long nq = n >> 1;
  long qi = (long)blockIdx.x * blockDim.x + tid;
  pp_d2 xa = {0.0, 0.0};
  if (pairs_ok && qi < nq) xa = __builtin_nontemporal_load(reinterpret_cast<const pp_d2*>(xnew) + qi);   // the first pair travels under the table build
  const double m0 = mesh_g[0], m_last = mesh_g[n_mesh - 1];
  const double step = (m_last - m0) / (double)(n_mesh - 1);
  __syncthreads();
  {
    bool okm = step > 0.0;
    for (int i = tid; i < n_mesh - 1; i += blockDim.x) okm = okm && (mesh_g[i] == mq_linspace_knot(i, step, m0));
    if (!okm) atomicOr(flag, 1);
  }
  for (int c = tid; c < ncells; c += blockDim.x) {
    double w[K + 1][K + 1];
#pragma unroll
    for (int i = 0; i <= K; ++i)
#pragma unroll
      for (int j = i; j <= K; ++j) w[i][j] = ((i == j) ? 1.0 : 2.0) * W_g[(size_t)(j - i) * M + c + K - j];   // rows c+K-i, c+K-j: W[d = j-i][the smaller row]
#pragma unroll
    for (int p = 0; p < NQ; ++p) {
      double q = (p == 0) ? variance : 0.0;
#pragma unroll
      for (int i = 0; i <= K; ++i)
#pragma unroll
        for (int j = i; j <= K; ++j) q = fma(MomCoef<K>::tab.pair[i][j][p], w[i][j], q);
      Q[(size_t)c * NQ + p] = q;
    }
  }
  for (int e = tid; e < M; e += blockDim.x) alpha[e] = alpha_g[e];
  __syncthreads();
  const double amax = fabs(m0) > fabs(m_last) ? fabs(m0) : fabs(m_last);
  const double margin = 16.0 * 2.220446049250313e-16 * amax * inv_delta + 1e-12;     // knot rounding over delta (as launch_phi_sort)
  const bool regular = (*flag == 0) && margin < 0.125;
  const double smax_fast = regular ? 0.5 - margin : -1.0;                            // (-1: every point takes the table rule)
  auto point = [&](double x, double& mo, double& vo) __attribute__((always_inline)) {
    const double g = floor((x - m0) * inv_delta);
    int c = __double2int_rz(g);                                                      // (saturating; NaN -> 0)
    double u0;
    {
#pragma clang fp contract(off)
      const double tk = g * step;                                                    // numpy.linspace's knot: i * step rounded, THEN + start rounded
      u0 = tk + m0;
    }
    double t = (x - u0) * inv_delta;
    if (!((unsigned)c < (unsigned)ncells && fabs(t - 0.5) <= smax_fast)) {           // rare: the exact table rule (basis.py:58-59), clamped as predict_kernel
      c = neighbour_index(x, mesh_g, n_mesh, m0, inv_delta);
      t = (x - mesh_g[c]) * inv_delta;
    }
    const double sc = t - 0.5;
    const double* __restrict__ q = Q + (size_t)c * NQ;
    double qv[NQ], av[K + 1];
#pragma unroll
    for (int p = 0; p < NQ; ++p) qv[p] = q[p];
#pragma unroll
    for (int i = 0; i <= K; ++i) av[i] = alpha[c + i];
    double acc = qv[NQ - 1];
#pragma unroll
    for (int p = NQ - 2; p >= 0; --p) acc = fma(acc, sc, qv[p]);
    vo = acc;
    double v[K + 1];
    bspline_pieces<K>(t, v);
    double m = 0.0;
#pragma unroll
    for (int i = 0; i <= K; ++i) m = fma(v[i], av[K - i], m);                        // row_i = c + K - i, in predict_kernel's order
    mo = m;
  };
  if (pairs_ok) {                                    // 16-byte aligned xnew / mean / var: two points per lane and iteration, the next pair in flight
    const pp_d2* x2 = reinterpret_cast<const pp_d2*>(xnew);
    pp_d2* mean2 = reinterpret_cast<pp_d2*>(mean);
    pp_d2* var2 = reinterpret_cast<pp_d2*>(var);
    // Two pairs in flight behind the one being worked on, loaded UNCONDITIONALLY at clamped indices: behind a branch the loaded value is
    // a phi node that hipcc materialises right behind the load (s_waitcnt vmcnt(0) in the same iteration - the "prefetch" of the first
    // version overlapped nothing, 16 KB of reads in flight per CU).
    const long qlast = nq > 0 ? nq - 1 : 0;
    pp_d2 xb = __builtin_nontemporal_load(x2 + (qi + stride < nq ? qi + stride : qlast));
    for (; qi < nq; qi += stride) {
      const long q2 = qi + 2 * stride;
      const pp_d2 xc = __builtin_nontemporal_load(x2 + (q2 < nq ? q2 : qlast));
      const double xs0 = xa.x, xs1 = xa.y;
      double m0v, v0v, m1v, v1v;
      point(xs0, m0v, v0v);
      point(xs1, m1v, v1v);
      __builtin_nontemporal_store(pp_d2{v0v, v1v}, var2 + qi);
      __builtin_nontemporal_store(pp_d2{m0v, m1v}, mean2 + qi);
      xa = xb; xb = xc;
    }
    const long pl = 2 * nq + (long)blockIdx.x * blockDim.x + tid;                    // the odd last point
    if (pl < n) { double mo, vo; point(xnew[pl], mo, vo); var[pl] = vo; mean[pl] = mo; }
  } else {
    for (long pi = (long)blockIdx.x * blockDim.x + tid; pi < n; pi += stride) {
      double mo, vo;
      point(xnew[pi], mo, vo);
      var[pi] = vo;
      mean[pi] = mo;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int phi_max_cols(int K, long n_mesh, bool fx) {
  long avail = (long)PHI_LDS_BUDGET - (long)n_mesh * 8 - 16 * 8;
  if (avail <= 0) return 0;
  return (int)(avail / (8 * (K + 2 + (fx ? 1 : 0))));
}

// Centred-moment Phi pass (algorithm 5, phi_moments.hpp).  Returns 1 when it does not apply (D != 1, unaligned inputs, or the
// image does not fit the LDS at this M) and the caller falls back to the band-scatter kernel.
template <int K, int CS>
static int launch_phi_moments_cs(Handle* h, const double* x, const double* y, long N, const double* mesh, long n_mesh, double delta,
                                 long M, double* stats, double* ws, hipStream_t st, bool regular, double step) {
  const int ncells = (int)n_mesh - 1;
  const size_t lds_bytes = mq_lds_bytes<K, CS>(M);
  if (ncells > CS || lds_bytes > PHI_LDS_BUDGET) return 1;
  long nblk = (N + 2 * MQ_THREADS - 1) / (2 * MQ_THREADS);
  const long gmax = (h->phi_blocks > 0 && h->phi_blocks < PHI_MAX_BLOCKS) ? h->phi_blocks : PHI_MAX_BLOCKS;
  const int G = (int)(nblk < 1 ? 1 : (nblk > gmax ? gmax : nblk));
  long ppb = (N + G - 1) / G;
  ppb = ((ppb + 2 * MQ_THREADS - 1) / (2 * MQ_THREADS)) * (2 * MQ_THREADS);
  int s0 = 50;   // 62 - ceil(log2(points per workgroup)), at most 50 (magic-constant conversion range)
  { long c = 2; int lg = 1; while (c < ppb) { c <<= 1; ++lg; } if (62 - lg < s0) s0 = 62 - lg; }
  MqArgs a;
  a.x = x; a.y = y; a.N = N; a.mesh_g = mesh; a.n_mesh = (int)n_mesh; a.inv_delta = 1.0 / delta; a.M = (int)M; a.step = step;
  a.partials = ws;
  a.ov = ws + (size_t)PHI_MAX_BLOCKS * ((size_t)(K + 2) * M + 1);
  a.ppb = ppb; a.zero_ptr = stats; a.zero_n = (K + 2) * M + 1; a.s0 = s0;
  auto kern = regular ? phi_moment_kernel<K, CS, true> : phi_moment_kernel<K, CS, false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_bytes, hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
  const bool prof = h->prof_on && h->prof_n < PROF_RING && (h->prof_calls++ % h->prof_every == 0);
  if (prof) (void)hipEventRecord(h->prof_ev[h->prof_n][0], st);
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(MQ_THREADS), lds_bytes, st, a);
  if (prof) { (void)hipEventRecord(h->prof_ev[h->prof_n][1], st); ++h->prof_n; }
  const int E1 = (int)((K + 2) * M + 1);
  const int gsplit = G >= 64 ? 16 : (G >= 8 ? 4 : 1);
  if (h->phi_defer) {   // the caller enqueues the reduce itself (asvgp_phi_reduce_1d), e.g. on the stream that consumes the statistics
    h->pend = Handle::PendingReduce{a.partials, G, (int)M, K, stats, true, nullptr};
    return check_launch("phi_accumulate_1d (moments, reduce deferred)");
  }
  hipLaunchKernelGGL(phi_reduce_kernel, dim3((E1 + 255) / 256, gsplit), dim3(256), 0, st, a.partials, G, (int)M, K, 0, M, 1L, 0, 1, stats, (const int*)nullptr);
  return check_launch("phi_accumulate_1d (moments)");
}

template <int K>
static int launch_phi_moments(Handle* h, const double* x, const double* y, long N, long D, const double* mesh, long n_mesh, double delta,
                              long M, double* stats, double* ws, hipStream_t st) {
  if (D != 1 || N < 1 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) != 0) return 1;
  double step = 0.0;
  const bool regular = handle_mesh_is_linspace(h, mesh, n_mesh, st, &step, nullptr, nullptr);
  int rc = launch_phi_moments_cs<K, 512>(h, x, y, N, mesh, n_mesh, delta, M, stats, ws, st, regular, step);     // plane stride: smallest that holds the cells
  if (rc == 1) rc = launch_phi_moments_cs<K, 1024>(h, x, y, N, mesh, n_mesh, delta, M, stats, ws, st, regular, step);
  if (rc == 1) rc = launch_phi_moments_cs<K, 2048>(h, x, y, N, mesh, n_mesh, delta, M, stats, ws, st, regular, step);
  return rc;
}

// Input-order probe: rows of 128 consecutive points, one per lane pair as the kernels read them; a row conforms when its points lie in
// at most two mesh cells (the arithmetic cell guess, spread <= 2 as in the kernel's own first vote).  count += conforming rows.
__global__ __launch_bounds__(64) void phi_order_probe_kernel(const double* __restrict__ x, long N, long row_stride, int nrows_probe, double m0, double inv_delta,
                                                             int n_mesh, int* __restrict__ count) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (r >= nrows_probe) return;
  const long base = ((long)r * row_stride) & ~1L;
  if (base + 128 > N) return;
  const double xa = x[base + 2 * lane], xb = x[base + 2 * lane + 1];
  const int ca = mq_guess(xa, m0, inv_delta, n_mesh), cb = mq_guess(xb, m0, inv_delta, n_mesh);
  const int c0 = __builtin_amdgcn_readfirstlane(ca);
  bool good = true;
  if (!__all(ca == c0 && cb == c0)) {
    const unsigned long long d0 = __ballot(ca != c0), d1 = __ballot(cb != c0);
    const int c1 = d0 ? __builtin_amdgcn_readlane(ca, (int)__builtin_ctzll(d0)) : __builtin_amdgcn_readlane(cb, (int)__builtin_ctzll(d1 | (1ull << 63)));
    good = __all((ca - c0 <= 2 && c0 - ca <= 2) || (ca - c1 <= 2 && c1 - ca <= 2)) && __all((cb - c0 <= 2 && c0 - cb <= 2) || (cb - c1 <= 2 && c1 - cb <= 2));
  }
  if (lane == 0 && good) atomicAdd(count, 1);
}

// Is x a time series (sorted / locally sorted)?  Decided once per (pointer, N) from 512 sampled rows; a stale verdict - the caller has
// overwritten the buffer with data of another order - costs speed, never correctness.
static bool handle_phi_is_series(Handle* h, const double* x, long N, double m0, double inv_delta, long n_mesh, hipStream_t st) {
  if (h->phi_order == 1) return false;
  if (h->phi_order == 2) return true;
  for (int i = 0; i < h->n_order_seen; ++i)
    if (h->order_seen[i].ptr == x && h->order_seen[i].n == N) return h->order_seen[i].series != 0;
  int series = 0;
  const int nprobe = 512;
  if (N >= 128L * nprobe) {
    if (!h->order_dev && hipMalloc(&h->order_dev, sizeof(int)) != hipSuccess) h->order_dev = nullptr;
    int cnt = 0;
    if (h->order_dev && hipMemsetAsync(h->order_dev, 0, sizeof(int), st) == hipSuccess) {
      hipLaunchKernelGGL(phi_order_probe_kernel, dim3(nprobe), dim3(64), 0, st, x, N, N / nprobe, nprobe, m0, inv_delta, (int)n_mesh, h->order_dev);
      if (hipMemcpyAsync(&cnt, h->order_dev, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
        series = cnt * 10 >= nprobe * 9 ? 1 : 0;                 // nine rows in ten: the front loop will carry the pass
    }
  }
  if (h->n_order_seen < 8) h->order_seen[h->n_order_seen++] = Handle::OrderSeen{x, N, series};
  return series != 0;
}

// Tile-sort Phi pass (algorithm 6, phi_sort.hpp).  Returns 1 when it does not apply - D != 1, unaligned inputs, fewer than two points,
// more than 2048 columns, a mesh that is not an exact numpy.linspace, or knots so large against delta that the arithmetic cell guess
// has no safe margin - and the caller falls back to the moment scatter (5), then the band scatter (3).
// Two instantiations: TP points per thread and tile without the time-series front loop (fastest on i.i.d. points), and 4 points per
// thread and tile with it (TS; the run sums and the second register set of its prefetch need the room) for inputs the order probe
// - or the caller, asvgp_set_phi_input_order - calls a time series.
template <int K, int TP, int TS>
static int launch_phi_sort_as(Handle* h, const double* x, const double* y, long N, const double* mesh, long n_mesh, double delta,
                              long M, double* stats, double* ws, hipStream_t st, double step, double m0, double m_last, double margin) {
  size_t lds_bytes = ps_lds_bytes<K, TP, TS>();
  if (ps_epilogue_bytes<K>() > lds_bytes) lds_bytes = ps_epilogue_bytes<K>();
  if (lds_bytes > 160 * 1024) return 1;
  long nblk = (N + TP * PS_THREADS - 1) / (TP * PS_THREADS);   // at least one tile per workgroup
  const long gmax = (h->phi_blocks > 0 && h->phi_blocks < PHI_MAX_BLOCKS) ? h->phi_blocks : PHI_MAX_BLOCKS;
  const int G = (int)(nblk < 1 ? 1 : (nblk > gmax ? gmax : nblk));
  long ppb = (N + G - 1) / G;
  ppb = TS ? ((ppb + 127) & ~127L) : ((ppb + 1) & ~1L);       // (TS: whole rows of 64 pairs - only the grid's last workgroup has an incomplete one)
  if (ppb > 0x3fffffffL) return 1;                              // (32-bit pair indices inside a workgroup)
  PsArgs a;
  a.x = x; a.y = y; a.N = N; a.mesh_g = mesh; a.n_mesh = (int)n_mesh; a.inv_delta = 1.0 / delta; a.M = (int)M;
  a.m0 = m0; a.m_last = m_last; a.step = step; a.smax_fast = 0.5 - margin;
  a.partials = ws; a.ppb = ppb; a.zero_ptr = stats; a.zero_n = (K + 2) * M + 1; a.stamps = nullptr; a.stamps_wave = 0;
  a.ranges = reinterpret_cast<int*>(ws + (size_t)PHI_MAX_BLOCKS * ((size_t)(K + 2) * M + 1));   // (behind the 256 partials: the moment kernel's overflow planes' room)
  auto kern = phi_sort_kernel<K, TP, 0, 1, TS>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_bytes, hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
  const bool prof = h->prof_on && h->prof_n < PROF_RING && (h->prof_calls++ % h->prof_every == 0);
  if (prof) (void)hipEventRecord(h->prof_ev[h->prof_n][0], st);
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(PS_THREADS), lds_bytes, st, a);
  if (prof) { (void)hipEventRecord(h->prof_ev[h->prof_n][1], st); ++h->prof_n; }
  h->phi_last_series = TS;
  const int E1 = (int)((K + 2) * M + 1);
  const int gsplit = G >= 64 ? 16 : (G >= 8 ? 4 : 1);
  if (h->phi_defer) {   // the caller enqueues the reduce itself (asvgp_phi_reduce_1d), e.g. on the stream that consumes the statistics
    h->pend = Handle::PendingReduce{a.partials, G, (int)M, K, stats, true, a.ranges};
    return check_launch("phi_accumulate_1d (tile sort, reduce deferred)");
  }
  hipLaunchKernelGGL(phi_reduce_kernel, dim3((E1 + 255) / 256, gsplit), dim3(256), 0, st, a.partials, G, (int)M, K, 0, M, 1L, 0, 1, stats, (const int*)a.ranges);
  return check_launch("phi_accumulate_1d (tile sort)");
}

template <int K>
static int launch_phi_sort(Handle* h, const double* x, const double* y, long N, long D, const double* mesh, long n_mesh, double delta,
                           long M, double* stats, double* ws, hipStream_t st) {
  if (D != 1 || N < 2 || M > PS_NCELL || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) != 0) return 1;
  double step = 0.0, m0 = 0.0, m_last = 0.0;
  if (!handle_mesh_is_linspace(h, mesh, n_mesh, st, &step, &m0, &m_last)) return 1;
  const double amax = fabs(m0) > fabs(m_last) ? fabs(m0) : fabs(m_last);
  const double margin = 16.0 * 2.220446049250313e-16 * amax / delta + 1e-12;   // knot rounding (two roundings <= ulp(|knot|)) over delta
  if (!(margin < 0.125)) return 1;
  constexpr int TP = ps_tile_points<K>();
  constexpr int TPS = TP < 4 ? TP : 4;
  if (handle_phi_is_series(h, x, N, m0, 1.0 / delta, n_mesh, st)) {
    const int rc = launch_phi_sort_as<K, TPS, 1>(h, x, y, N, mesh, n_mesh, delta, M, stats, ws, st, step, m0, m_last, margin);
    if (rc != 1) return rc;
  }
  return launch_phi_sort_as<K, TP, 0>(h, x, y, N, mesh, n_mesh, delta, M, stats, ws, st, step, m0, m_last, margin);
}

// Band-scatter Phi pass (algorithms 1 and 3): the (k+1)(k+2)/2 + (k+1) products of every point go straight into the
// workgroup's LDS band image.  Timing events (Handle::prof_*) are recorded on the launch stream right around the kernel.
template <int K>
static int launch_phi(Handle* h, const double* x, const double* y, long N, long D, const double* mesh, long n_mesh,
                      double delta, long M, double* stats, double* partials, hipStream_t st) {
  // a reduce still parked on this handle (deferred mode, e.g. the per-dimension calls of an additive model share one handle and one
  // partials workspace) goes out now, stream-ordered before this pass overwrites the partials
  { const int rcf = handle_flush_phi_reduce(h, nullptr, st); if (rcf) return rcf; }
  const int ncells = (int)n_mesh - 1;
  if (h->phi_algo == 0 || h->phi_algo == 6) {
    const int rc = launch_phi_sort<K>(h, x, y, N, D, mesh, n_mesh, delta, M, stats, partials, st);
    if (rc != 1) { h->phi_last = 6; return rc; }
    if (h->phi_algo == 6) { set_error("phi algorithm 6 (tile sort) needs D == 1, N >= 2, M <= 2048, 16-byte aligned x / y and a mesh that is an exact linspace"); return ASVGP_ERR_UNSUPPORTED; }
  }
  if (h->phi_algo == 0 || h->phi_algo == 5) {
    const int rc = launch_phi_moments<K>(h, x, y, N, D, mesh, n_mesh, delta, M, stats, partials, st);
    if (rc != 1) { h->phi_last = 5; return rc; }
    if (h->phi_algo == 5) { set_error("phi algorithm 5 (centred moments) needs D == 1, 16-byte aligned x / y and M small enough for the LDS split"); return ASVGP_ERR_UNSUPPORTED; }
  }
  const bool fx = (h->phi_algo != 1);
  h->phi_last = fx ? 3 : 1;
  int maxc = phi_max_cols(K, n_mesh, fx);
  if (maxc < 2 * K + 2) {
    set_error("phi_accumulate_1d: mesh table (%ld knots) leaves no LDS for the band", n_mesh);
    return ASVGP_ERR_LDS_CAPACITY;
  }
  int cells_per_chunk = (M <= maxc) ? ncells : (maxc - K);
  long nblk = (N + 2 * PHI_THREADS - 1) / (2 * PHI_THREADS);
  const long gmax = (h->phi_blocks > 0 && h->phi_blocks < PHI_MAX_BLOCKS) ? h->phi_blocks : PHI_MAX_BLOCKS;
  int G = (int)(nblk < 1 ? 1 : (nblk > gmax ? gmax : nblk));
  long ppb = (N + G - 1) / G;
  ppb = ((ppb + 2 * PHI_THREADS - 1) / (2 * PHI_THREADS)) * (2 * PHI_THREADS);
  const double inv_delta = 1.0 / delta;
  hipError_t e = hipSuccess;
  const long zero_n = (K + 1) * M + M * D + 1;
  bool zeroed = false;   // the first launched kernel zeroes the stats buffer
  for (long dcol = 0; dcol < D; ++dcol) {
    const double* yd = y + dcol;
    bool vec = (D == 1) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(yd) & 15) == 0);
    for (int cell0 = 0; cell0 < ncells; cell0 += cells_per_chunk) {
      int cell1 = cell0 + cells_per_chunk;
      if (cell1 > ncells) cell1 = ncells;
      int ncols = cell1 - cell0 + K;
      size_t lds_bytes = sizeof(double) * ((size_t)(K + 2 + (fx ? 1 : 0)) * ncols + n_mesh + 16);
      int do_band = (dcol == 0);
      auto kern = fx ? (vec ? phi_accumulate_kernel<K, true, true> : phi_accumulate_kernel<K, false, true>)
                     : (vec ? phi_accumulate_kernel<K, true, false> : phi_accumulate_kernel<K, false, false>);
      int s0 = 50;   // 62 - ceil(log2(points per workgroup)), at most 50 (magic-constant conversion range)
      { long c = 2; int lg = 1; while (c < ppb) { c <<= 1; ++lg; } if (62 - lg < s0) s0 = 62 - lg; }
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_bytes, hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
      const bool prof = h->prof_on && h->prof_n < PROF_RING && (h->prof_calls++ % h->prof_every == 0);
      if (prof) (void)hipEventRecord(h->prof_ev[h->prof_n][0], st);
      hipLaunchKernelGGL(kern, dim3(G), dim3(PHI_THREADS), lds_bytes, st, x, yd, (long)D, N, mesh, (int)n_mesh,
                         inv_delta, cell0, cell1, ncols, do_band, partials, ppb, zeroed ? (double*)nullptr : stats, zero_n, s0);
      zeroed = true;
      if (prof) { (void)hipEventRecord(h->prof_ev[h->prof_n][1], st); ++h->prof_n; }
      int E1 = (K + 2) * ncols + 1;
      int gsplit = G >= 64 ? 16 : (G >= 8 ? 4 : 1);
      hipLaunchKernelGGL(phi_reduce_kernel, dim3((E1 + 255) / 256, gsplit), dim3(256), 0, st, partials, G, ncols, K,
                         cell0, M, D, (int)dcol, do_band, stats, (const int*)nullptr);
    }
  }
  return check_launch("phi_accumulate_1d");
}

}  // namespace asvgp

using namespace asvgp;

extern "C" size_t asvgp_phi_workspace_bytes(int64_t M, int order, int64_t D) {
  (void)D;
  if (M <= 0 || order < 1 || order > ASVGP_MAX_ORDER) return 0;
  // 256 partial [band | Phi y | y^T y] images; the centred-moment kernel adds a per-workgroup fp64 plane for out-of-scale y
  const size_t band = (size_t)PHI_MAX_BLOCKS * ((size_t)(order + 2) * (size_t)M + 1);
  const size_t mom = band + (size_t)PHI_MAX_BLOCKS * (size_t)M;   // + the per-workgroup fp64 plane for out-of-scale y
  return sizeof(double) * (band > mom ? band : mom);
}

extern "C" int asvgp_phi_accumulate_1d(asvgp_handle_t handle, const double* x, const double* y, int64_t N, int64_t D, const double* mesh,
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
  Handle* h = as_handle(handle);
  double* part = static_cast<double*>(workspace);
  switch (order) {
    case 1: return launch_phi<1>(h, x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    case 2: return launch_phi<2>(h, x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    case 3: return launch_phi<3>(h, x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    case 4: return launch_phi<4>(h, x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    case 5: return launch_phi<5>(h, x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
    default: return launch_phi<6>(h, x, y, N, D, mesh, n_mesh, delta, M, stats, part, st);
  }
}

extern "C" int asvgp_phi_last_algorithm(asvgp_handle_t handle) { return as_handle(handle)->phi_last; }

// Read-only stream of x and y with the Phi pass's launch shape (one 1024-thread workgroup per CU, 16-byte loads, two per array in
// flight per lane): the measured ceiling the Phi kernel's roofline fraction is quoted beside.  sink: >= 8 bytes, never written
// for finite data.
__global__ __launch_bounds__(1024) void stream_probe_kernel(const double* __restrict__ x, const double* __restrict__ y, long N, long ppb,
                                                            double* __restrict__ sink) {
  typedef double v2 __attribute__((ext_vector_type(2)));
  const long beg = (long)blockIdx.x * ppb;
  long end = beg + ppb;
  if (end > N) end = N;
  if (end <= beg) return;
  const v2* x2 = reinterpret_cast<const v2*>(x) + (beg >> 1);
  const v2* y2 = reinterpret_cast<const v2*>(y) + (beg >> 1);
  const int npair = (int)((end - beg) >> 1);
  double acc = 0.0;
  for (int base = 0; base < npair; base += 2 * 1024) {
    v2 xv[2], yv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      int u = base + d * 1024 + (int)threadIdx.x;
      u = u < npair - 1 ? u : npair - 1;
      xv[d] = __builtin_nontemporal_load(x2 + u);
      yv[d] = __builtin_nontemporal_load(y2 + u);
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) acc += xv[d].x + xv[d].y + yv[d].x + yv[d].y;
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

extern "C" int asvgp_stream_probe(const double* x, const double* y, int64_t N, double* sink, asvgp_stream_t stream) {
  if (!x || !y || !sink || N < 2 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) != 0) {
    set_error("stream_probe: bad argument (N >= 2, 16-byte aligned x / y)");
    return ASVGP_ERR_BAD_ARG;
  }
  const int G = 256;
  long ppb = ((long)N + G - 1) / G;
  ppb = (ppb + 1) & ~1L;
  hipLaunchKernelGGL(stream_probe_kernel, dim3(G), dim3(1024), 0, as_stream(stream), x, y, (long)N, ppb, sink);
  return check_launch("stream_probe");
}

namespace asvgp {
int handle_flush_phi_reduce(Handle* h, const double* stats, hipStream_t st) {
  if (!h->pend.valid) return ASVGP_OK;                 // nothing deferred (the accumulate call has reduced already)
  if (stats && stats != h->pend.stats) return ASVGP_OK;
  const Handle::PendingReduce p = h->pend;
  h->pend.valid = false;
  const int E1 = (p.K + 2) * p.M + 1;
  const int gsplit = p.G >= 64 ? 16 : (p.G >= 8 ? 4 : 1);
  hipLaunchKernelGGL(phi_reduce_kernel, dim3((E1 + 255) / 256, gsplit), dim3(256), 0, st, p.partials, p.G, p.M, p.K, 0, (long)p.M, 1L, 0, 1, p.stats, p.ranges);
  return check_launch("phi_reduce_1d");
}
}  // namespace asvgp

extern "C" int asvgp_phi_reduce_1d(asvgp_handle_t handle, asvgp_stream_t stream) {
  return handle_flush_phi_reduce(as_handle(handle), nullptr, as_stream(stream));
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
static int launch_predict(Handle* h, const double* xnew, long n, const double* mesh, int n_mesh, double delta, int M,
                          const double* alpha, const double* W, double variance, int D, double* mean, double* var,
                          hipStream_t st) {
  size_t lds_bytes = sizeof(double) * ((size_t)(K + 2) * M + n_mesh);
  const bool vec4 = (D == 1) && n >= 4 && (((reinterpret_cast<uintptr_t>(xnew) | reinterpret_cast<uintptr_t>(mean) |
                                            reinterpret_cast<uintptr_t>(var)) & 15) == 0);
  static const int poly_mode = getenv("ASVGP_PREDICT_POLY") ? atoi(getenv("ASVGP_PREDICT_POLY")) : 1;   // (0: the table-read kernel always - diagnostic)
  const size_t poly_bytes = predict_poly_lds_bytes<K>(n_mesh, M);
  double lin_step = 0.0;
  if (h && poly_mode && D == 1 && n >= 262144 && poly_bytes <= 160 * 1024 - 64 && n_mesh >= 2 &&
      handle_mesh_is_linspace(h, mesh, n_mesh, st, &lin_step, nullptr, nullptr)) {      // (verdict cached per mesh pointer; the kernel re-checks)
    auto pk = predict_poly_kernel<K>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)poly_bytes);
    if (e == hipSuccess) {
      long blocks = ((vec4 ? (n + 1) / 2 : n) + 1023) / 1024;
      if (blocks > 256) blocks = 256;
      hipLaunchKernelGGL(pk, dim3((unsigned)blocks), dim3(1024), poly_bytes, st, xnew, n, mesh, n_mesh, 1.0 / delta, M, alpha, W, variance,
                         vec4 ? 1 : 0, mean, var);
      return check_launch("predict_1d (cell polynomials)");
    }
    (void)hipGetLastError();
  }
  int stage = (D == 1 && lds_bytes <= PHI_LDS_BUDGET && n >= 65536) ? 1 : 0;
  int threads = stage ? 1024 : 256;
  long blocks = ((vec4 ? (n + 1) / 2 : n) + threads - 1) / threads;
  long cap = stage ? 256 : 2048;
  if (blocks > cap) blocks = cap;
  auto kern = stage ? (vec4 ? predict_kernel<K, true, true> : predict_kernel<K, false, true>)
                    : (vec4 ? predict_kernel<K, true, false> : predict_kernel<K, false, false>);
  if (stage) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), stage ? lds_bytes : 0, st, xnew, n, mesh,
                     n_mesh, 1.0 / delta, M, alpha, W, variance, D, stage, mean, var);
  return check_launch("predict_1d");
}

static int predict_1d_entry(Handle* h, const double* xnew, int64_t n, const double* mesh, int64_t n_mesh, double delta,
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
    case 1: return launch_predict<1>(h, xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    case 2: return launch_predict<2>(h, xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    case 3: return launch_predict<3>(h, xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    case 4: return launch_predict<4>(h, xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    case 5: return launch_predict<5>(h, xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
    default: return launch_predict<6>(h, xnew, n, mesh, (int)n_mesh, delta, (int)M, alpha, W, variance, (int)D, mean, var, st);
  }
}

extern "C" int asvgp_predict_1d(const double* xnew, int64_t n, const double* mesh, int64_t n_mesh, double delta,
                                int order, int64_t M, const double* alpha, const double* W, double variance,
                                int64_t D, double* mean, double* var, asvgp_stream_t stream) {
  return predict_1d_entry(nullptr, xnew, n, mesh, n_mesh, delta, order, M, alpha, W, variance, D, mean, var, stream);
}

// The same with the model's handle: the handle remembers whether the mesh is an exact numpy.linspace, which lets large D = 1 batches
// take the cell-polynomial kernel (predict_poly_kernel).
extern "C" int asvgp_predict_1d_h(asvgp_handle_t handle, const double* xnew, int64_t n, const double* mesh, int64_t n_mesh, double delta,
                                  int order, int64_t M, const double* alpha, const double* W, double variance,
                                  int64_t D, double* mean, double* var, asvgp_stream_t stream) {
  Handle* h = as_handle(handle);
  if (!h) { set_error("predict_1d_h: bad handle"); return ASVGP_ERR_BAD_ARG; }
  return predict_1d_entry(h, xnew, n, mesh, n_mesh, delta, order, M, alpha, W, variance, D, mean, var, stream);
}
