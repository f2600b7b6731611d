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

#include "asvgp_common.hpp"

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
                                         int cell0, int cell1) {
  bool all_in = true;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    B.idx[q] = valid ? neighbour_index(B.x[q], mesh, n_mesh, m0, inv_delta) : -1;
    B.in[q] = valid && B.idx[q] >= cell0 && B.idx[q] < cell1;
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
    classify<NP>(B, have, mesh, n_mesh, m0, inv_delta, cell0, cell1);
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
      classify<1>(T1, tid == 0, mesh, n_mesh, m0, inv_delta, cell0, cell1);
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


// =================================================================================================
// Phi pass v2: tile-local counting sort + per-cell moment accumulation in registers.
//
// The per-point LDS fp64 atomics of v1 (20 per point, ~30 cycles per wave-instruction under random
// addresses) are replaced by:  (1) one returning u32 LDS atomic per point (rank inside its cell),
// (2) an exclusive scan of the per-cell counts, (3) one 16-B LDS write of (s, y), s = t - 1/2,
// (4) the thread that OWNS the cell (thread tau owns cells tau and tau+1024 of the chunk) reads its
// points back and accumulates the 3k+2 sufficient statistics  S_p = sum s^p (p<=2k), T_p = sum y s^p (p<=k)
// in registers - no fp64 atomics in the streaming loop.  After the last tile the moments are converted
// once per workgroup into band / rhs entries (products of the piece polynomials expanded in s with exact
// integer coefficients), flushed, and reduced across workgroups exactly like v1.
// Cells holding more than HEAVY points of a tile (sorted / clustered inputs) are accumulated by the
// whole wavefront cooperatively, so time-series order is not a worst case.
// =================================================================================================
// Workgroup barrier that orders LDS traffic only: global loads issued before it (the next tile's prefetch) stay
// in flight across it.  __syncthreads() would make hipcc drain them with s_waitcnt vmcnt(0) (cdna guide, "Pipelining
// across barriers"), serialising HBM time with compute.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int MOM_THREADS = 1024;
constexpr int MOM_CELLS = 2048;   // cells per pass (2 per thread)
constexpr int MOM_HEAVY = 40;

using i128 = __int128;
// Compile-time coefficient tables (constexpr constructor => guaranteed constant evaluation; every use below has
// static indices after unrolling, so the entries fold into instruction literals).
template <int K> struct MomTab {
  double single[K + 1][K + 1];             // v_i(s + 1/2) = sum_p single[i][p] s^p
  double pair[K + 1][K + 1][2 * K + 1];    // v_i v_j      = sum_p pair[i][j][p] s^p   (j >= i)
  static constexpr long long pnum(int i, int q) {  // integer numerator of the t^q coefficient of piece i times K!
    long long num = 0;
    for (int j = 0; j <= i; ++j) {
      long long term = binom(K + 1, j) * binom(K, q) * ipow(i - j, K - q);
      num += (j & 1) ? -term : term;
    }
    return num;
  }
  constexpr MomTab() : single{}, pair{} {
    long long m[K + 1][K + 1] = {};        // v_i(s + 1/2) = (1 / (K! 2^K)) sum_r m[i][r] s^r
    for (int i = 0; i <= K; ++i)
      for (int r = 0; r <= K; ++r) {
        long long acc = 0;
        for (int q = r; q <= K; ++q) acc += pnum(i, q) * binom(q, r) * (1LL << (K - q + r));
        m[i][r] = acc;
      }
    const double d1 = (double)fact(K) * (double)(1LL << K);
    const double d2 = (double)(fact(K) * fact(K)) * (double)(1LL << (2 * K));
    for (int i = 0; i <= K; ++i) {
      for (int p = 0; p <= K; ++p) single[i][p] = (double)m[i][p] / d1;
      for (int j = i; j <= K; ++j)
        for (int p = 0; p <= 2 * K; ++p) {
          i128 acc = 0;
          for (int r = 0; r <= K; ++r) {
            int r2 = p - r;
            if (r2 < 0 || r2 > K) continue;
            acc += (i128)m[i][r] * (i128)m[j][r2];
          }
          pair[i][j][p] = (double)acc / d2;
        }
    }
  }
};
template <int K> struct MomCoef {
  static constexpr MomTab<K> tab{};
};

template <int K>
__device__ __forceinline__ void mom_accumulate(double s, double y, double (&S)[2 * K + 1], double (&T)[K + 1]) {
  double pw = 1.0;
  S[0] += 1.0;
  T[0] += y;
#pragma unroll
  for (int p = 1; p <= 2 * K; ++p) {
    pw *= s;
    S[p] += pw;
    if (p <= K) T[p] = fma(y, pw, T[p]);
  }
}


// Heavy cells of a wave's 128 owned cells (sorted / clustered input): the whole wavefront walks the cell's points, in TWO
// passes (S moments, then T moments) so that at most 2k+1 temporaries are live next to the 2 (3k+2) owner accumulators - the
// one-pass version cost 64 spilled VGPRs in the streaming loop - and reduces on the VALU (DPP).  Wave-uniform control flow.
template <int K>
__device__ __forceinline__ void mom_heavy_cells(const double2* buf, unsigned nA, unsigned oA, bool hvA, unsigned nB, unsigned oB,
                                                bool hvB, int lane, double (&SA)[2 * K + 1], double (&TA)[K + 1],
                                                double (&SB)[2 * K + 1], double (&TB)[K + 1]) {
  unsigned long long ma = __ballot(hvA), mb = __ballot(hvB);
  while (ma | mb) {
    const bool isA = ma != 0ull;
    const unsigned long long m = isA ? ma : mb;
    const int h = __ffsll((long long)m) - 1;
    if (isA) ma &= ma - 1; else mb &= mb - 1;
    const unsigned nh = (unsigned)__builtin_amdgcn_readlane((int)(isA ? nA : nB), h);
    const unsigned oh = (unsigned)__builtin_amdgcn_readlane((int)(isA ? oA : oB), h);
    const bool meA = isA && lane == h, meB = !isA && lane == h;
    {
      double S2[2 * K + 1];
#pragma unroll
      for (int p = 0; p <= 2 * K; ++p) S2[p] = 0.0;
      for (unsigned j = lane; j < nh; j += 64) {
        const double sv = buf[oh + j].x;
        double pw = 1.0;
        S2[0] += 1.0;
#pragma unroll
        for (int p = 1; p <= 2 * K; ++p) { pw *= sv; S2[p] += pw; }
      }
#pragma unroll
      for (int p = 0; p <= 2 * K; ++p) {
        const double t = wave_sum_dpp(S2[p]);
        SA[p] += meA ? t : 0.0;
        SB[p] += meB ? t : 0.0;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      double T2[K + 1];
#pragma unroll
      for (int p = 0; p <= K; ++p) T2[p] = 0.0;
      for (unsigned j = lane; j < nh; j += 64) {
        const double2 pt = buf[oh + j];
        double pw = 1.0;
        T2[0] += pt.y;
#pragma unroll
        for (int p = 1; p <= K; ++p) { pw *= pt.x; T2[p] = fma(pt.y, pw, T2[p]); }
      }
#pragma unroll
      for (int p = 0; p <= K; ++p) {
        const double t = wave_sum_dpp(T2[p]);
        TA[p] += meA ? t : 0.0;
        TB[p] += meB ? t : 0.0;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// both cells of one owner lane in ONE loop (trip = max(nA, nB) instead of nA + nB, two independent FMA streams);
// a missing point is fed as (s, y) = (0, 0), which only touches S_0 - masked explicitly.
template <int K>
__device__ __forceinline__ void mom_own_two_cells(const double2* buf, unsigned nA, unsigned oA, unsigned nB, unsigned oB,
                                                  int lane, double (&SA)[2 * K + 1], double (&TA)[K + 1],
                                                  double (&SB)[2 * K + 1], double (&TB)[K + 1]) {
  const bool hvA = nA > MOM_HEAVY, hvB = nB > MOM_HEAVY;
  const unsigned la = hvA ? 0u : nA, lb = hvB ? 0u : nB;
  const unsigned n = la > lb ? la : lb;
  for (unsigned j = 0; j < n; ++j) {
    const bool a = j < la, b = j < lb;
    double2 pa = a ? buf[oA + j] : make_double2(0.0, 0.0);
    double2 pb = b ? buf[oB + j] : make_double2(0.0, 0.0);
    double wa = pa.x, wb = pb.x;
    SA[0] += a ? 1.0 : 0.0;
    SB[0] += b ? 1.0 : 0.0;
    TA[0] += pa.y;
    TB[0] += pb.y;
    SA[1] += wa; SB[1] += wb;
    TA[1] = fma(pa.y, wa, TA[1]); TB[1] = fma(pb.y, wb, TB[1]);
#pragma unroll
    for (int p = 2; p <= 2 * K; ++p) {
      wa *= pa.x; wb *= pb.x;
      SA[p] += wa; SB[p] += wb;
      if (p <= K) { TA[p] = fma(pa.y, wa, TA[p]); TB[p] = fma(pb.y, wb, TB[p]); }
    }
  }
  if (__any(hvA || hvB)) mom_heavy_cells<K>(buf, nA, oA, hvA, nB, oB, hvB, lane, SA, TA, SB, TB);
}

// moments of cell c -> band / rhs contributions (exact integer-ratio coefficients, centred monomials)
template <int K>
__device__ __forceinline__ void mom_to_band(const double (&S)[2 * K + 1], const double (&T)[K + 1], int c, int ncols,
                                            int do_band, double* band, double* rhs) {
  if (S[0] == 0.0) return;
#pragma unroll
  for (int i = 0; i <= K; ++i) {
    double r = 0.0;
#pragma unroll
    for (int p = 0; p <= K; ++p) r = fma(MomCoef<K>::tab.single[i][p], T[p], r);
    lds_add(rhs + c + K - i, r);
    if (do_band) {
#pragma unroll
      for (int j = i; j <= K; ++j) {
        double b = 0.0;
#pragma unroll
        for (int p = 0; p <= 2 * K; ++p) b = fma(MomCoef<K>::tab.pair[i][j][p], S[p], b);
        lds_add(band + (j - i) * ncols + c + K - j, b);
      }
    }
  }
}

template <int K, int TP, bool VEC, int ablate = 0>
__global__ __launch_bounds__(MOM_THREADS) void phi_moments_kernel(
    const double* __restrict__ x, const double* __restrict__ y, long y_stride, long N,
    const double* __restrict__ mesh_g, int n_mesh, double inv_delta, int cell0, int cell1, int ncols,
    int do_band, double* __restrict__ partials, long ppb, double* __restrict__ zero_ptr, long zero_n) {
  extern __shared__ double lds[];
  if (zero_ptr) for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < zero_n; e += (long)gridDim.x * blockDim.x) zero_ptr[e] = 0.0;
  constexpr int T = TP * MOM_THREADS;
  double2* buf = reinterpret_cast<double2*>(lds);                      // T sorted (s, y)
  unsigned* cnt = reinterpret_cast<unsigned*>(lds + 2 * T);            // MOM_CELLS
  unsigned* off = cnt + MOM_CELLS;                                     // MOM_CELLS + 1
  unsigned* wtot = off + MOM_CELLS + 1;                                // 16 wave totals (+pad)
  double* red = reinterpret_cast<double*>(wtot + 32);                  // 16 doubles
  int* flag = reinterpret_cast<int*>(red + 16);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int NC = cell1 - cell0;

  // ---- is the mesh table bit-identical to numpy's linspace arithmetic  i*step + a ?  (then no table lookups)
  const double m0 = mesh_g[0];
  const double step = (mesh_g[n_mesh - 1] - m0) / (double)(n_mesh - 1);
  if (tid == 0) *flag = 1;
  for (int e = tid; e < 2 * MOM_CELLS; e += MOM_THREADS) cnt[e] = 0;   // cnt and off
  __syncthreads();
  {
    bool ok = true;
    for (int e = tid; e < n_mesh - 1; e += MOM_THREADS) ok = ok && (__dadd_rn(__dmul_rn((double)e, step), m0) == mesh_g[e]);
    if (!ok) *flag = 0;
  }
  __syncthreads();
  const bool arith = (*flag != 0);
  auto knot = [&](int i) -> double { return arith ? __dadd_rn(__dmul_rn((double)i, step), m0) : mesh_g[i]; };

  // diagnostic variant 9: per-phase cycle stamps of thread 0 (written over this block's partial after the flush)
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
  auto stamp = [&](int i) {
    if constexpr (ablate == 9) {
      unsigned long long t = __builtin_amdgcn_s_memtime();
      if (i >= 0) ph[i] += t - tprev;
      tprev = t;
    }
  };
  double SA[2 * K + 1], TA[K + 1], SB[2 * K + 1], TB[K + 1];
#pragma unroll
  for (int p = 0; p <= 2 * K; ++p) { SA[p] = 0.0; SB[p] = 0.0; }
#pragma unroll
  for (int p = 0; p <= K; ++p) { TA[p] = 0.0; TB[p] = 0.0; }
  double yy = 0.0;

  const long beg = (long)blockIdx.x * ppb;
  long end = beg + ppb;
  if (end > N) end = N;

  double xs[TP], ys[TP];
  // point q of this thread: VEC -> pairs (base + 2*(q2*1024+tid) + {0,1}).  The tile is fetched in TP/2 slices so that
  // no phase issues more than ~32 KB per CU at once (a whole tile exceeds what a CU keeps in flight and the issue blocks).
  auto load_slice = [&](long base, int q0, int q1) {
#pragma unroll
    for (int q = 0; q < TP; q += 2) {
      if (q < q0 || q >= q1) continue;
      if (VEC) {
        long i = base + 2 * ((long)(q >> 1) * MOM_THREADS + tid);
        if (i + 1 < end) {
          double2 xv = *reinterpret_cast<const double2*>(x + i);
          double2 yv = *reinterpret_cast<const double2*>(y + i);
          xs[q] = xv.x; xs[q + 1] = xv.y; ys[q] = yv.x; ys[q + 1] = yv.y;
        } else {
          xs[q] = (i < end) ? x[i] : __builtin_nan("");
          ys[q] = (i < end) ? y[i] : 0.0;
          xs[q + 1] = __builtin_nan(""); ys[q + 1] = 0.0;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          long i = base + (long)(q + u) * MOM_THREADS + tid;
          xs[q + u] = (i < end) ? x[i] : __builtin_nan("");
          ys[q + u] = (i < end) ? y[i * y_stride] : 0.0;
        }
      }
    }
  };

  auto load_tile = [&](long base) { load_slice(base, 0, TP); };
  constexpr int SL = (TP / 2 + 2) / 3 * 2;  // points per slice (3 slices)
  if (beg < end) load_tile(beg);
  for (long base = beg; base < end; base += T) {
    stamp(-1);
    // ---- P1: cell, centred coordinate, rank within the cell
    double sv[TP], yv[TP];
    int cr[TP];  // cell << 13 | rank   (rank < 8192), -1 = not in this chunk / out of range
    {
      int ci[TP];
      bool slow = false;
#pragma unroll
      for (int q = 0; q < TP; ++q) {
        const double xv = xs[q];
        yv[q] = ys[q];
        double g = floor((xv - m0) * inv_delta);
        int i = (g < 0.0) ? 0 : ((g > (double)(n_mesh - 2)) ? (n_mesh - 2) : (int)g);   // NaN -> 0
        // branch-free +-1 fix-up against the knots (the floor guess is off by at most one on a monotone mesh) ...
        i -= (i > 0 && !(knot(i) < xv)) ? 1 : 0;
        i += (i < n_mesh - 2 && knot(i + 1) < xv) ? 1 : 0;
        // ... verified; anything else (wildly non-uniform table) takes the exact search below
        const double lo = knot(i), hi = knot(i + 1);
        const bool good = (lo < xv || i == 0) && (!(hi < xv) || i == n_mesh - 2);
        slow = slow || (!good && xv == xv);
        ci[q] = i;
        sv[q] = fma(xv - lo, inv_delta, -0.5);
      }
      if (__any(slow)) {  // rare: exact searchsorted semantics by linear walk
#pragma unroll
        for (int q = 0; q < TP; ++q) {
          const double xv = xs[q];
          int i = ci[q];
          while (i > 0 && !(knot(i) < xv)) --i;
          while (i < n_mesh - 2 && knot(i + 1) < xv) ++i;
          ci[q] = i;
          sv[q] = fma(xv - knot(i), inv_delta, -0.5);
        }
      }
      if constexpr (ablate == 1 || ablate == 2) {
#pragma unroll
        for (int q = 0; q < TP; ++q) yy += sv[q] + (double)ci[q] + yv[q];
      } else {
        unsigned rk[TP];
#pragma unroll
        for (int q = 0; q < TP; ++q) {  // all rank atomics of the tile in flight together
          const int c = ci[q] - cell0;
          const bool ok = (xs[q] == xs[q]) && c >= 0 && c < NC;
          cr[q] = ok ? c : -1;
          rk[q] = ok ? atomicAdd(&cnt[c], 1u) : 0u;
          yy = ok ? fma(yv[q], yv[q], yy) : yy;
        }
#pragma unroll
        for (int q = 0; q < TP; ++q) cr[q] = (cr[q] >= 0) ? ((cr[q] << 13) | (int)rk[q]) : -1;
      }
    }
    if constexpr (ablate == 1 || ablate == 2) { if (base + T < end) load_tile(base + T); continue; }
    lds_barrier();
    stamp(0);
    const bool more = base + T < end;
    if constexpr (ablate == 3 || ablate == 4) { if (more) load_tile(base + T); }
    if constexpr (ablate != 3 && ablate != 4) if (more) load_slice(base + T, 0, SL);  // prefetch the next tile in slices under the sort / owner phases
    if constexpr (ablate == 3) {
      lds_barrier();
      cnt[tid] = 0; cnt[tid + MOM_THREADS] = 0;
      lds_barrier();
#pragma unroll
      for (int q = 0; q < TP; ++q) yy += (double)(cr[q] & 8191);
      continue;
    }
    // ---- P2: exclusive scan of cnt -> off   (thread t scans cells 2t, 2t+1)
    {
      unsigned c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
      unsigned v = c0 + c1, inc = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        unsigned o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
      }
      if (lane == 63) wtot[wv] = inc;
      lds_barrier();
      unsigned basew = 0;
      for (int w = 0; w < wv; ++w) basew += wtot[w];
      unsigned ex = basew + inc - v;
      off[2 * tid] = ex;
      off[2 * tid + 1] = ex + c0;
    }
    lds_barrier();
    stamp(1);
    if (more) load_slice(base + T, SL, 2 * SL);
    // ---- P3: scatter (s, y) into cell order
#pragma unroll
    for (int q = 0; q < TP; ++q)
      if (cr[q] >= 0) buf[off[cr[q] >> 13] + (cr[q] & 8191)] = make_double2(sv[q], yv[q]);
    lds_barrier();
    stamp(2);
    if constexpr (ablate == 4) {
      cnt[tid] = 0; cnt[tid + MOM_THREADS] = 0;
      lds_barrier();
      continue;
    }
    // ---- P4: owners accumulate their cells' moments
    if (more) load_slice(base + T, 2 * SL, TP);
    {
      const unsigned nA = cnt[tid], oA = off[tid];
      const unsigned nB = cnt[tid + MOM_THREADS], oB = off[tid + MOM_THREADS];
      mom_own_two_cells<K>(buf, nA, oA, nB, oB, lane, SA, TA, SB, TB);
      cnt[tid] = 0;
      cnt[tid + MOM_THREADS] = 0;
    }
    stamp(3);
    lds_barrier();
    stamp(4);
  }

  // ---- moments -> band / rhs (LDS image aliases the sort buffers), then flush like v1
  double tot = block_sum(yy, red);
  __syncthreads();
  const int E = (K + 2) * ncols;
  for (int e = tid; e < E; e += MOM_THREADS) lds[e] = 0.0;
  __syncthreads();
  double* band = lds;
  double* rhs = band + (K + 1) * ncols;
  if constexpr (ablate == 5) {  // diagnostic: keep the moments alive, skip the conversion
    double keep = 0.0;
#pragma unroll
    for (int p = 0; p <= 2 * K; ++p) keep += SA[p] + SB[p];
#pragma unroll
    for (int p = 0; p <= K; ++p) keep += TA[p] + TB[p];
    if (tid < ncols) rhs[tid] = keep;
  } else {
    if (tid < NC) mom_to_band<K>(SA, TA, tid, ncols, do_band, band, rhs);
    if (tid + MOM_THREADS < NC) mom_to_band<K>(SB, TB, tid + MOM_THREADS, ncols, do_band, band, rhs);
  }
  __syncthreads();
  double* out = partials + (size_t)blockIdx.x * (E + 1);
  for (int e = tid; e < E; e += MOM_THREADS) out[e] = lds[e];
  if (tid == 0) out[E] = tot;
  if constexpr (ablate == 9) {
    __syncthreads();
    if (tid == 0)
      for (int i = 0; i < 6; ++i) out[i] = (double)ph[i];
  }
}

// =================================================================================================
// Phi pass v4 ("buckets"): per-cell point lists instead of a sort, one barrier per tile.
//
// Tile = 2048 points (one 16-B pair per thread), double-buffered in the LDS as raw (x, y).  Rank phase: every thread
// finds its two cells and takes a slot in the cell's list with ONE returning u32 LDS atomic (count), then stores its
// 11-bit point index in the slot (6 slots per cell; mean occupancy is 1 at M = 2048).  Points beyond 6 go to a
// tile-wide overflow list that can hold the whole tile.  Owner phase (next barrier interval, overlapping the rank
// phase of the following tile): thread tau owns cells 2tau and 2tau+1, reads count + slot words (4 conflict-free
// ds_read_b64), fetches its points from the raw tile and accumulates the 3k+2 centred moments in registers exactly like
// v2 - no fp64 atomics, no scan, no scatter of 16-B records.  A wavefront whose 128 points fall into ONE cell (sorted /
// time-series input) reduces its moments on the VALU (DPP) and posts a single aggregate record instead.
// =================================================================================================
constexpr int BK_THREADS = 1024;
constexpr int BK_T = 2 * BK_THREADS;   // points per tile
constexpr int BK_CELLS = 2048;         // cells per pass (2 per thread)
constexpr int BK_SLOTS = 6;
constexpr int BK_AGG = 16;             // at most one aggregate per wave and tile

template <int K> constexpr int bk_agg_doubles() { return 3 * K + 2 + 2; }   // cell, S[2K+1], T[K+1], pad

template <int K>
__host__ __device__ constexpr size_t bk_lds_bytes() {
  return (size_t)2 * BK_T * 16            // raw tiles
       + (size_t)2 * BK_CELLS * 4         // counts
       + (size_t)2 * 3 * BK_CELLS * 4     // slot words (3 dwords = 6 u16 per cell), dword-major
       + (size_t)3 * BK_T * 4             // overflow lists (triple-buffered with their counters)
       + (size_t)3 * BK_AGG * bk_agg_doubles<K>() * 8
       + 64 * 8;                          // counters, reduction scratch, flags
}

template <int K, bool VEC>
__global__ __launch_bounds__(BK_THREADS) void phi_bucket_kernel(
    const double* __restrict__ x, const double* __restrict__ y, long y_stride, long N,
    const double* __restrict__ mesh_g, int n_mesh, double inv_delta, int cell0, int cell1, int ncols,
    int do_band, double* __restrict__ partials, long ppb, double* __restrict__ zero_ptr, long zero_n) {
  extern __shared__ double lds[];
  if (zero_ptr) for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < zero_n; e += (long)gridDim.x * blockDim.x) zero_ptr[e] = 0.0;
  constexpr int NA = bk_agg_doubles<K>();
  double2* raw = reinterpret_cast<double2*>(lds);                                    // [2][BK_T]
  unsigned* cnt = reinterpret_cast<unsigned*>(raw + 2 * BK_T);                      // [2][BK_CELLS]
  unsigned* slotw = cnt + 2 * BK_CELLS;                                             // [2][3][BK_CELLS]
  unsigned* ovf = slotw + 2 * 3 * BK_CELLS;                                         // [3][BK_T]   cell << 16 | point
  double* agg = reinterpret_cast<double*>(ovf + 3 * BK_T);                          // [3][BK_AGG][NA]
  double* red = agg + 3 * BK_AGG * NA;                                              // 16 doubles
  unsigned* ctr = reinterpret_cast<unsigned*>(red + 16);                            // [3] overflow counts, [3] aggregate counts
  int* flag = reinterpret_cast<int*>(ctr + 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int NC = cell1 - cell0;

  const double m0 = mesh_g[0];
  const double step = (mesh_g[n_mesh - 1] - m0) / (double)(n_mesh - 1);
  if (tid == 0) *flag = 1;
  for (int e = tid; e < 2 * BK_CELLS; e += BK_THREADS) cnt[e] = 0;
  if (tid < 8) ctr[tid] = 0;
  __syncthreads();
  {
    bool ok = true;
    for (int e = tid; e < n_mesh - 1; e += BK_THREADS) ok = ok && (__dadd_rn(__dmul_rn((double)e, step), m0) == mesh_g[e]);
    if (!ok) *flag = 0;
  }
  __syncthreads();
  const bool arith = (*flag != 0);
  auto knot = [&](int i) -> double { return arith ? __dadd_rn(__dmul_rn((double)i, step), m0) : mesh_g[i]; };

  double SA[2 * K + 1], TA[K + 1], SB[2 * K + 1], TB[K + 1];
#pragma unroll
  for (int p = 0; p <= 2 * K; ++p) { SA[p] = 0.0; SB[p] = 0.0; }
#pragma unroll
  for (int p = 0; p <= K; ++p) { TA[p] = 0.0; TB[p] = 0.0; }
  double yy = 0.0;
  const int cA = 2 * tid, cB = 2 * tid + 1;                 // owned cells (chunk-local)
  const double uA = knot(min(cell0 + cA, n_mesh - 2)), uB = knot(min(cell0 + cB, n_mesh - 2));

  const long beg = (long)blockIdx.x * ppb;
  long end = beg + ppb;
  if (end > N) end = N;
  const long ntiles = (end > beg) ? (end - beg + BK_T - 1) / BK_T : 0;

  double2 px, py;   // the pair in flight (x0, x1), (y0, y1); NaN x = no point
  auto load_pair = [&](long t) {
    const long i = beg + t * BK_T + 2 * tid;
    if (VEC) {
      if (i + 1 < end) { px = *reinterpret_cast<const double2*>(x + i); py = *reinterpret_cast<const double2*>(y + i); return; }
    }
    px.x = (i < end) ? x[i] : __builtin_nan("");
    py.x = (i < end) ? y[i * y_stride] : 0.0;
    px.y = (i + 1 < end) ? x[i + 1] : __builtin_nan("");
    py.y = (i + 1 < end) ? y[(i + 1) * y_stride] : 0.0;
  };

  auto find_cell = [&](double xv) -> int {
    double g = floor((xv - m0) * inv_delta);
    int i = (g < 0.0) ? 0 : ((g > (double)(n_mesh - 2)) ? (n_mesh - 2) : (int)g);   // NaN -> 0
    i -= (i > 0 && !(knot(i) < xv)) ? 1 : 0;
    i += (i < n_mesh - 2 && knot(i + 1) < xv) ? 1 : 0;
    const double lo = knot(i), hi = knot(i + 1);
    const bool good = (lo < xv || i == 0) && (!(hi < xv) || i == n_mesh - 2);
    if (!good && xv == xv) {   // rare: exact searchsorted semantics by linear walk
      while (i > 0 && !(knot(i) < xv)) --i;
      while (i < n_mesh - 2 && knot(i + 1) < xv) ++i;
    }
    return i;
  };

  // ---- rank phase of tile t into buffer t & 1
  auto rank_tile = [&](long t) {
    const int b = (int)(t & 1), o3 = (int)(t % 3);
    double2* rw = raw + b * BK_T;
    unsigned* cn = cnt + b * BK_CELLS;
    unsigned* sw = slotw + b * 3 * BK_CELLS;
    const double xv[2] = {px.x, px.y}, yv[2] = {py.x, py.y};
    rw[2 * tid] = make_double2(xv[0], yv[0]);
    rw[2 * tid + 1] = make_double2(xv[1], yv[1]);
    int c[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ci = find_cell(xv[u]) - cell0;
      ok[u] = (xv[u] == xv[u]) && ci >= 0 && ci < NC;
      c[u] = ok[u] ? ci : -1;
      yy = ok[u] ? fma(yv[u], yv[u], yy) : yy;
    }
    if (t + 1 < ntiles) load_pair(t + 1);   // next pair in flight under the LDS work (raw barrier keeps it there)
    const int c0 = __builtin_amdgcn_readfirstlane(c[0]);
    if (__all(ok[0] && ok[1] && c[0] == c0 && c[1] == c0)) {   // wave-uniform: one aggregate record
      const double u = knot(cell0 + c0);
      double S2[2 * K + 1], T2[K + 1];
#pragma unroll
      for (int p = 0; p <= 2 * K; ++p) S2[p] = 0.0;
#pragma unroll
      for (int p = 0; p <= K; ++p) T2[p] = 0.0;
#pragma unroll
      for (int q = 0; q < 2; ++q) mom_accumulate<K>(fma(xv[q] - u, inv_delta, -0.5), yv[q], S2, T2);
      unsigned pos = 0;
      if (lane == 0) pos = atomicAdd(&ctr[3 + o3], 1u);
      pos = __builtin_amdgcn_readfirstlane(pos);
      double* rec = agg + ((size_t)o3 * BK_AGG + pos) * NA;
#pragma unroll
      for (int p = 0; p <= 2 * K; ++p) { double r = wave_sum_dpp(S2[p]); if (lane == 0) rec[1 + p] = r; }
#pragma unroll
      for (int p = 0; p <= K; ++p) { double r = wave_sum_dpp(T2[p]); if (lane == 0) rec[2 + 2 * K + p] = r; }
      if (lane == 0) rec[0] = (double)c0;
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (ok[u]) {
        const unsigned r = atomicAdd(&cn[c[u]], 1u);
        const unsigned pidx = 2 * tid + u;
        if (r < BK_SLOTS) {
          reinterpret_cast<unsigned short*>(sw + (r >> 1) * BK_CELLS + c[u])[r & 1] = (unsigned short)pidx;
        } else {
          const unsigned pos = atomicAdd(&ctr[o3], 1u);
          ovf[o3 * BK_T + pos] = ((unsigned)c[u] << 16) | pidx;
        }
      }
    }
  };

  // ---- owner phase of tile t
  auto own_tile = [&](long t) {
    const int b = (int)(t & 1), o3 = (int)(t % 3);
    const double2* rw = raw + b * BK_T;
    unsigned* cn = cnt + b * BK_CELLS;
    const unsigned* sw = slotw + b * 3 * BK_CELLS;
    const uint2 n2 = *reinterpret_cast<const uint2*>(cn + cA);
    const unsigned nA = min(n2.x, (unsigned)BK_SLOTS), nB = min(n2.y, (unsigned)BK_SLOTS);
    const unsigned nmax = max(nA, nB);
    if (nmax) {
      uint2 w[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) w[q] = *reinterpret_cast<const uint2*>(sw + q * BK_CELLS + cA);
      if (n2.x | n2.y) *reinterpret_cast<uint2*>(cn + cA) = make_uint2(0u, 0u);
#pragma unroll
      for (int j = 0; j < BK_SLOTS; ++j) {
        if (j < (int)nmax) {
          const bool a = j < (int)nA, bb = j < (int)nB;
          const unsigned ia = (w[j >> 1].x >> (16 * (j & 1))) & 0xffffu, ib = (w[j >> 1].y >> (16 * (j & 1))) & 0xffffu;
          const double2 pa = a ? rw[ia] : make_double2(0.0, 0.0);
          const double2 pb = bb ? rw[ib] : make_double2(0.0, 0.0);
          const double sa = a ? fma(pa.x - uA, inv_delta, -0.5) : 0.0, sb = bb ? fma(pb.x - uB, inv_delta, -0.5) : 0.0;
          double wa = sa, wb = sb;
          SA[0] += a ? 1.0 : 0.0;
          SB[0] += bb ? 1.0 : 0.0;
          TA[0] += pa.y;
          TB[0] += pb.y;
          SA[1] += wa; SB[1] += wb;
          TA[1] = fma(pa.y, wa, TA[1]); TB[1] = fma(pb.y, wb, TB[1]);
#pragma unroll
          for (int p = 2; p <= 2 * K; ++p) {
            wa *= sa; wb *= sb;
            SA[p] += wa; SB[p] += wb;
            if (p <= K) { TA[p] = fma(pa.y, wa, TA[p]); TB[p] = fma(pb.y, wb, TB[p]); }
          }
        }
      }
    }
    // overflow list of this tile (cells with more than BK_SLOTS points) and wave aggregates: usually both empty
    const unsigned nov = ctr[o3], nag = ctr[3 + o3];
    for (unsigned e = 0; e < nov; ++e) {
      const unsigned ent = ovf[o3 * BK_T + e];
      const int cc = (int)(ent >> 16);
      if (cc == cA || cc == cB) {
        const double2 pt = rw[ent & 0xffffu];
        if (cc == cA) mom_accumulate<K>(fma(pt.x - uA, inv_delta, -0.5), pt.y, SA, TA);
        else mom_accumulate<K>(fma(pt.x - uB, inv_delta, -0.5), pt.y, SB, TB);
      }
    }
    for (unsigned e = 0; e < nag; ++e) {
      const double* rec = agg + ((size_t)o3 * BK_AGG + e) * NA;
      const int cc = (int)rec[0];
      if (cc == cA) {
#pragma unroll
        for (int p = 0; p <= 2 * K; ++p) SA[p] += rec[1 + p];
#pragma unroll
        for (int p = 0; p <= K; ++p) TA[p] += rec[2 + 2 * K + p];
      } else if (cc == cB) {
#pragma unroll
        for (int p = 0; p <= 2 * K; ++p) SB[p] += rec[1 + p];
#pragma unroll
        for (int p = 0; p <= K; ++p) TB[p] += rec[2 + 2 * K + p];
      }
    }
  };

  if (ntiles > 0) {
    load_pair(0);
    rank_tile(0);
  }
  lds_barrier();
  for (long t = 0; t < ntiles; ++t) {
    // counters of tile t + 2 were last read by the owner phase of tile t - 1, which the barrier above has retired
    if (tid == 0) { ctr[(t + 2) % 3] = 0; ctr[3 + (t + 2) % 3] = 0; }
    if (t + 1 < ntiles) rank_tile(t + 1);
    own_tile(t);
    lds_barrier();
  }

  // ---- moments -> band / rhs (LDS image aliases the tile buffers), then flush like v1
  double tot = block_sum(yy, red);
  __syncthreads();
  const int E = (K + 2) * ncols;
  for (int e = tid; e < E; e += BK_THREADS) lds[e] = 0.0;
  __syncthreads();
  double* band = lds;
  double* rhs = band + (K + 1) * ncols;
  if (cA < NC) mom_to_band<K>(SA, TA, cA, ncols, do_band, band, rhs);
  if (cB < NC) mom_to_band<K>(SB, TB, cB, ncols, do_band, band, rhs);
  __syncthreads();
  double* out = partials + (size_t)blockIdx.x * (E + 1);
  for (int e = tid; e < E; e += BK_THREADS) out[e] = lds[e];
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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Optional in-library timing of the dominant kernel: HIP events recorded on the launch stream right around
// phi_accumulate_kernel (bench.py's roofline figure; must agree with the rocprofv3 kernel-trace average).
constexpr int PROF_RING = 1024;
static int g_phi_blocks = 0;   // 0 = PHI_MAX_BLOCKS; fewer leaves CUs free for a concurrent prior-chain kernel
static int g_phi_ablate = 0;   // diagnostic only (ASVGP_PHI_ABLATE): 1 loads, 2 +cell, 3 +rank atomics, 4 +scan/scatter
static int g_phi_algo = 0;  // 0 auto (= 3: 110 us at N=10M, vs 158 us for 1 and 166 us for 2), 1 = fp64 LDS atomics, 2 = moments, 3 = fixed-point band
static bool g_prof_on = false;
static int g_prof_every = 1;     // instrument every n-th Phi launch (events perturb the stream: keep them sparse)
static long g_prof_calls = 0;
static hipEvent_t g_prof_ev[PROF_RING][2];
static bool g_prof_made = false;
static long g_prof_n = 0;

static int phi_max_cols(int K, long n_mesh, bool fx) {
  long avail = (long)PHI_LDS_BUDGET - (long)n_mesh * 8 - 16 * 8;
  if (avail <= 0) return 0;
  return (int)(avail / (8 * (K + 2 + (fx ? 1 : 0))));
}

template <int K>
static int launch_phi(const double* x, const double* y, long N, long D, const double* mesh, long n_mesh,
                      double delta, long M, double* stats, double* partials, hipStream_t st) {
  const int ncells = (int)n_mesh - 1;
  const bool fx = (g_phi_algo == 3 || g_phi_algo == 0 || (g_phi_algo == 4 && bk_lds_bytes<K>() > PHI_LDS_BUDGET));
  int maxc = phi_max_cols(K, n_mesh, fx);
  if (maxc < 2 * K + 2) {
    set_error("phi_accumulate_1d: mesh table (%ld knots) leaves no LDS for the band", n_mesh);
    return ASVGP_ERR_LDS_CAPACITY;
  }
  int cells_per_chunk = (M <= maxc) ? ncells : (maxc - K);
  const bool v2 = (g_phi_algo == 2);
  const bool v4 = (g_phi_algo == 4) && bk_lds_bytes<K>() <= PHI_LDS_BUDGET;   // (k = 6: the bucket buffers exceed the LDS -> algorithm 3)
  constexpr int TP = 4;
  if ((v2 || v4) && cells_per_chunk > MOM_CELLS) cells_per_chunk = MOM_CELLS;
  long nblk = (N + 2 * PHI_THREADS - 1) / (2 * PHI_THREADS);
  const long gmax = (g_phi_blocks > 0 && g_phi_blocks < PHI_MAX_BLOCKS) ? g_phi_blocks : PHI_MAX_BLOCKS;
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
      if (v2) {
        size_t sort_bytes = (size_t)TP * MOM_THREADS * 16 + (size_t)(2 * MOM_CELLS + 1 + 32) * 4 + 16 * 8 + 64;
        size_t band_bytes = sizeof(double) * (size_t)(K + 2) * ncols;
        lds_bytes = sort_bytes > band_bytes ? sort_bytes : band_bytes;
        auto k2 = vec ? phi_moments_kernel<K, TP, true> : phi_moments_kernel<K, TP, false>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_bytes, hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
      }
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_bytes, hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
      const bool prof = g_prof_on && g_prof_n < PROF_RING && (g_prof_calls++ % g_prof_every == 0);
      if (prof) (void)hipEventRecord(g_prof_ev[g_prof_n][0], st);
      if (v4) {
        auto k4 = vec ? phi_bucket_kernel<K, true> : phi_bucket_kernel<K, false>;
        size_t band_bytes = sizeof(double) * (size_t)(K + 2) * ncols;
        size_t l4 = bk_lds_bytes<K>() > band_bytes ? bk_lds_bytes<K>() : band_bytes;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", l4, hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
        hipLaunchKernelGGL(k4, dim3(G), dim3(BK_THREADS), l4, st, x, yd, (long)D, N, mesh, (int)n_mesh, inv_delta,
                           cell0, cell1, ncols, do_band, partials, ppb, zeroed ? (double*)nullptr : stats, zero_n);
      } else if (v2) {
        auto k2 = vec ? phi_moments_kernel<K, TP, true> : phi_moments_kernel<K, TP, false>;
        if (K == 4 && vec && g_phi_ablate >= 1 && g_phi_ablate <= 9) {  // diagnostic builds (tools/phi_ablate.py)
          if (g_phi_ablate == 1) k2 = phi_moments_kernel<4, TP, true, 1>;
          if (g_phi_ablate == 2) k2 = phi_moments_kernel<4, TP, true, 2>;
          if (g_phi_ablate == 3) k2 = phi_moments_kernel<4, TP, true, 3>;
          if (g_phi_ablate == 4) k2 = phi_moments_kernel<4, TP, true, 4>;
          if (g_phi_ablate == 5) k2 = phi_moments_kernel<4, TP, true, 5>;
          if (g_phi_ablate == 9) k2 = phi_moments_kernel<4, TP, true, 9>;
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        }
        hipLaunchKernelGGL(k2, dim3(G), dim3(MOM_THREADS), lds_bytes, st, x, yd, (long)D, N, mesh, (int)n_mesh, inv_delta,
                           cell0, cell1, ncols, do_band, partials, ppb, zeroed ? (double*)nullptr : stats, zero_n);
      } else {
        hipLaunchKernelGGL(kern, dim3(G), dim3(PHI_THREADS), lds_bytes, st, x, yd, (long)D, N, mesh, (int)n_mesh,
                           inv_delta, cell0, cell1, ncols, do_band, partials, ppb, zeroed ? (double*)nullptr : stats, zero_n, s0);
      }
      zeroed = true;
      if (prof) { (void)hipEventRecord(g_prof_ev[g_prof_n][1], st); ++g_prof_n; }
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

extern "C" int asvgp_set_phi_algorithm(int algo) {
  if (algo < 0 || algo > 4) { set_error("set_phi_algorithm: 0 auto, 1 fp64 LDS-atomic scatter, 2 counting-sort + moments, 3 fixed-point band scatter, 4 per-cell buckets + moments"); return ASVGP_ERR_BAD_ARG; }
  g_phi_algo = algo;
  const char* ab = getenv("ASVGP_PHI_ABLATE");
  g_phi_ablate = ab ? atoi(ab) : 0;
  return ASVGP_OK;
}

extern "C" int asvgp_set_phi_workgroups(int n) {
  if (n < 0 || n > PHI_MAX_BLOCKS) { set_error("set_phi_workgroups: 0 (default, one per CU) .. %d", PHI_MAX_BLOCKS); return ASVGP_ERR_BAD_ARG; }
  g_phi_blocks = n;
  return ASVGP_OK;
}

extern "C" int asvgp_profile_enable(int on) {
  if (on && !g_prof_made) {
    for (int i = 0; i < PROF_RING; ++i)
      for (int j = 0; j < 2; ++j)
        if (hipEventCreateWithFlags(&g_prof_ev[i][j], hipEventReleaseToDevice) != hipSuccess) { set_error("hipEventCreate failed"); return ASVGP_ERR_HIP; }
    g_prof_made = true;
  }
  g_prof_on = on != 0;
  g_prof_every = on > 1 ? on : 1;   // on = n > 1: every n-th launch
  g_prof_calls = 0;
  g_prof_n = 0;
  return ASVGP_OK;
}

extern "C" int asvgp_profile_read(double* phi_kernel_ms_sum, int64_t* launches) {
  if (!phi_kernel_ms_sum || !launches) { set_error("profile_read: bad argument"); return ASVGP_ERR_BAD_ARG; }
  double tot = 0.0;
  for (long i = 0; i < g_prof_n; ++i) {
    if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess) { set_error("hipEventSynchronize failed"); return ASVGP_ERR_HIP; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g_prof_ev[i][0], g_prof_ev[i][1]);
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
  const bool vec4 = (D == 1) && n >= 4 && (((reinterpret_cast<uintptr_t>(xnew) | reinterpret_cast<uintptr_t>(mean) |
                                            reinterpret_cast<uintptr_t>(var)) & 15) == 0);
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
