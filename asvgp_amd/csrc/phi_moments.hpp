// Phi pass, algorithm 5 (default): per-CELL centred moments, accumulated in 64-bit fixed point with LDS integer atomics.
//
// Replaces (reference): basis.py:51-76 evaluate_basis + gpr.py:41-44 (Kuf@y, Kuf@Kuf.T, sparse_to_band, sum y^2), D = 1.
// Inside one mesh cell every entry of phi phi^T is a polynomial of degree 2k in the local coordinate and every entry of
// phi y is y times a polynomial of degree k, so the sufficient statistics of a cell are the 3k+2 moments
//     S_p = sum s^p (p = 0..2k),   T_p = sum y s^p (p = 0..k),   s = t - 1/2 in [-1/2, 1/2]
// - 14 numbers at k = 4 against the 20 band / rhs products of the band-scatter kernel (algorithms 1, 3), and S_0 is a plain
// count.  The LDS atomic pipe is what bounds this kernel (one random-address ds_add_u64 wave-instruction per ~11 cycles,
// tools/micro/lds_atomic_rate.hip), so the cost per point drops from 20 to 13 64-bit atomics + one 32-bit one.
//
// The moment image of ALL cells (14 x 2044 x 8 B = 229 KB at M = 2048) does not fit one CU's LDS, so the planes are split
// over G workgroups ("roles") that stream the SAME point range: role g accumulates its share of the planes for every point
// (all lanes busy - a split by cells would idle half of them).  The roles of a range sit on one XCD (blockIdx b and b + 8
// share an XCD under the observed round-robin dispatch; speed only), so the second read of a range is an L2 hit and HBM
// sees every byte once; S-only roles never load y.
//
// Fixed point: |s^p| <= 2^-p, so plane S_p carries round(s^p * 2^(s0 + p)) and T_p carries round(y s^p * 2^(s0 - E + p)) with
// |y| <= y0 = 2^E (a per-workgroup scale from a strided sample of the range; points outside it, or NaN, go to a per-workgroup
// fp64 plane in global memory with L2 atomics - never wrong, only slow, and rare by construction).  s0 = min(50, 62 -
// ceil(log2(points per range))): the rounding error per addend is <= 2^-(s0+1) of the plane's bound, unbiased, and the
// sums inside a workgroup are order-independent.  Signed values ride as two's complement through the same magic-constant
// conversion (one v_add_f64, asvgp_common.hpp fx_convert).
// Sorted / time-series input: a wave whose whole batch sits in ONE cell sums its moments in registers across iterations and
// commits once per run (DPP wave reduction, one lane, full double -> int64 conversion) instead of 64 same-address atomics.
//
// After the kernel: phi_moment_reduce_kernel sums the per-range images (fp64 atomics into a 14 x ncells total), and
// phi_moment_convert_kernel turns moments into the packed [band | Phi y | y^T y] buffer through the exact integer-ratio
// tables MomTab (centred monomials: <= 2e-14 of the largest entry up to k = 6).
#pragma once

namespace asvgp {

constexpr int MQ_THREADS = 1024;
constexpr int MQ_CH = 32;           // iterations a wavefront stays on one contiguous slice (sorted input: long runs)

// planes of a role: the 3K+1 64-bit planes [S_1 .. S_2K | T_0 .. T_K] cut into G contiguous shares; role 0 also owns the
// 32-bit count plane S_0.
template <int K, int G, int ROLE> struct MomSplit {
  static constexpr int NPL = 3 * K + 1;
  static constexpr int lo = (NPL * ROLE) / G, hi = (NPL * (ROLE + 1)) / G;
  static constexpr int n = hi - lo;
  static constexpr bool has_cnt = (ROLE == 0);
  static constexpr bool needs_y = hi > 2 * K;
  static constexpr bool owns_yy = (lo <= 2 * K) && (2 * K < hi);     // the role that owns T_0 also sums y^2
  static constexpr int n_t = needs_y ? hi - (lo > 2 * K ? lo : 2 * K) : 0;   // T planes of this role
  static constexpr int t_lo = needs_y ? ((lo > 2 * K ? lo : 2 * K) - 2 * K) : 0;   // first T power
  __host__ __device__ static constexpr int max_pow() {   // highest power of s this role needs
    int m = 0;
    for (int q = lo; q < hi; ++q) { const int p = (q < 2 * K) ? q + 1 : q - 2 * K; m = p > m ? p : m; }
    return m;
  }
};
template <int K> __host__ __device__ constexpr int mq_planes() { return 3 * K + 2; }   // count + 64-bit planes, as stored in partials / totals

// bytes of LDS the largest role needs: its planes, the mesh table, scratch
template <int K, int G> __host__ inline size_t mq_lds_bytes(int ncells, int n_mesh) {
  size_t worst = 0;
  for (int role = 0; role < G; ++role) {
    const int NPL = 3 * K + 1;
    const int n = (NPL * (role + 1)) / G - (NPL * role) / G;
    size_t b = (size_t)n * ncells * 8 + (role == 0 ? (size_t)ncells * 4 : 0);
    worst = b > worst ? b : worst;
  }
  return ((worst + 15) & ~(size_t)15) + (size_t)n_mesh * 8 + 64 * 8;
}

struct MqArgs {
  const double* x; const double* y; long N;
  const double* mesh_g; int n_mesh; double inv_delta;
  double* partials;        // [range][mq_planes][ncells] doubles
  double* scal;            // [range][2]: y^T y, bad-point count
  double* ov;              // [range][K+1][ncells] fp64 planes for out-of-scale y (written and read by the range's T roles only)
  long ppr; int n_ranges;  // points per range (a multiple of 2 * MQ_THREADS), number of ranges
  double* tot; long tot_n; // totals image to zero (phi_moment_reduce_kernel adds into it afterwards)
  int s0;
};

// cell of x and its left knot (table search against the LDS copy of the mesh: exact searchsorted semantics, basis.py:58-59)
__device__ __forceinline__ int mq_cell(double x, const double* mesh, int n_mesh, double m0, double inv_delta, double& u) {
  const int i = neighbour_index(x, mesh, n_mesh, m0, inv_delta);
  u = mesh[i];
  return i;
}

template <int K, int G, int ROLE>
__device__ __forceinline__ void phi_moment_body(const MqArgs& a, int range, double* lds) {
  using SP = MomSplit<K, G, ROLE>;
  constexpr int NQ = SP::n;
  const int ncells = a.n_mesh - 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long* planes = reinterpret_cast<unsigned long long*>(lds);            // [NQ][ncells]
  unsigned* cnt = reinterpret_cast<unsigned*>(planes + (size_t)NQ * ncells);          // [ncells] (role 0)
  const size_t img_bytes = (((size_t)NQ * ncells * 8 + (SP::has_cnt ? (size_t)ncells * 4 : 0)) + 15) & ~(size_t)15;
  double* mesh = reinterpret_cast<double*>(reinterpret_cast<char*>(lds) + img_bytes);  // [n_mesh]
  double* scratch = mesh + a.n_mesh;                                                   // 64 doubles
  {
    double2* z = reinterpret_cast<double2*>(lds);
    for (int e = tid; e < (int)(img_bytes / 16); e += MQ_THREADS) z[e] = make_double2(0.0, 0.0);
    for (int e = tid; e < a.n_mesh; e += MQ_THREADS) mesh[e] = a.mesh_g[e];
  }
  double* ovr = a.ov + (size_t)range * (K + 1) * ncells;   // this range's fp64 overflow planes (T roles zero their own share)
  if (SP::needs_y)
    for (int e = tid; e < SP::n_t * ncells; e += MQ_THREADS) ovr[(size_t)SP::t_lo * ncells + e] = 0.0;
  __syncthreads();
  const long beg = (long)range * a.ppr;
  long end = beg + a.ppr;
  if (end > a.N) end = a.N;
  const long n_it = a.ppr / (2 * MQ_THREADS);              // rows of 64 lanes x 2 points per wave
  auto unit_of = [&](long it) -> long {                    // super-tiles: each wave streams MQ_CH contiguous rows
    const long nfull = n_it / MQ_CH, rem = n_it - nfull * MQ_CH;
    const long sup = it / MQ_CH, r = it - sup * MQ_CH;
    return (sup < nfull) ? (sup * (MQ_THREADS / 64) + wv) * MQ_CH + r : nfull * (MQ_THREADS / 64) * MQ_CH + wv * rem + r;
  };
  const double2* x2 = reinterpret_cast<const double2*>(a.x);
  const double2* y2 = reinterpret_cast<const double2*>(a.y);
  const long ubeg = beg >> 1, uend = (end > beg) ? (end >> 1) : 0;   // pairs
  // ---- y scale of this workgroup: 2^E >= 4 max |y| over a strided sample of the range (4 x 1024 pairs)
  int E = 0;
  double y0 = 0.0;
  if (SP::needs_y) {
    double m = 0.0;
    const long npairs = uend > ubeg ? uend - ubeg : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long u = ubeg + (npairs * q) / 4 + tid;
      if (u < uend) { const double2 v = y2[u]; const double t = fmax(fabs(v.x), fabs(v.y)); m = (t == t && t < 1e300) ? fmax(m, t) : m; }
    }
    if (tid == 0 && (end & 1) && end > beg) { const double t = fabs(a.y[end - 1]); m = (t == t && t < 1e300) ? fmax(m, t) : m; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if (lane == 0) scratch[wv] = m;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < MQ_THREADS / 64; ++w) t = fmax(t, scratch[w]);
    __syncthreads();
    E = (t > 0.0) ? ilogb(t) + 3 : -900;   // nothing to go by: every non-zero y takes the exact fp64 plane
    E = E > 900 ? 900 : (E < -900 ? -900 : E);
    y0 = ldexp(1.0, E);
  }
  const int s0 = a.s0;
  const int chiS = ((1075 - s0) << 20) | 0x80000;          // magic-constant high word for scale 2^s0; power p: - (p << 20)
  const int chiT = ((1075 - (s0 - E)) << 20) | 0x80000;
  const double m0 = mesh[0];
  double yy = 0.0;
  unsigned nbad = 0;

  // one point (cell c, centred local coordinate s) -> the role's planes
  auto scatter = [&](int c, double s, double yv) {
    if (!(fabs(s) <= 0.50001)) { ++nbad; return; }         // outside the mesh (or NaN): reported, never accumulated
    if (SP::has_cnt) __hip_atomic_fetch_add(cnt + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (SP::owns_yy) yy = fma(yv, yv, yy);
    const bool y_in = !SP::needs_y || (fabs(yv) <= y0);
    double pw = 1.0;
#pragma unroll
    for (int p = 0; p <= SP::max_pow(); ++p) {
      if (p > 0) pw *= s;
      const int qs = p - 1;                                // S plane index of power p (p >= 1)
      if (p >= 1 && qs >= SP::lo && qs < SP::hi && qs < 2 * K)
        lds_add_u64(planes + (size_t)(qs - SP::lo) * ncells + c, fx_convert(pw, chiS - (p << 20)));
      const int qt = 2 * K + p;                            // T plane index of power p
      if (p <= K && qt >= SP::lo && qt < SP::hi) {
        const double v = yv * pw;
        if (y_in) lds_add_u64(planes + (size_t)(qt - SP::lo) * ncells + c, fx_convert(v, chiT - (p << 20)));
        else __hip_atomic_fetch_add(ovr + (size_t)p * ncells + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };

  // run accumulator (a wave whose whole batch lies in one cell): in-lane sums, one commit per run
  double racc[NQ];
  double rcnt = 0.0;
  int rcell = -1;
  auto run_flush = [&]() {
    if (rcell < 0) return;
    const double nc = wave_sum_dpp(rcnt);
    if (SP::has_cnt && lane == 0) __hip_atomic_fetch_add(cnt + rcell, (unsigned)nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = SP::lo + j;
      const double t = wave_sum_dpp(racc[j]);
      const int sc = (q < 2 * K) ? (s0 + q + 1) : (s0 - E + (q - 2 * K));
      if (lane == 0) lds_add_u64(planes + (size_t)j * ncells + rcell, (unsigned long long)__double2ll_rn(ldexp(t, sc)));
      racc[j] = 0.0;
    }
    rcnt = 0.0;
    rcell = -1;
  };
#pragma unroll
  for (int j = 0; j < NQ; ++j) racc[j] = 0.0;

  // ---- streaming loop: two-deep register prefetch of 16-B pairs
  constexpr int DEPTH = 2;
  double2 xb[DEPTH], yb[DEPTH];
  auto fetch = [&](long it, double2& xo, double2& yo) {
    const long u = ubeg + unit_of(it) * 64 + lane;
    if (it < n_it && u < uend) {
      xo = x2[u];
      if (SP::needs_y) yo = y2[u];
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) { xb[d] = make_double2(0.0, 0.0); yb[d] = make_double2(0.0, 0.0); fetch(d, xb[d], yb[d]); }
  for (long it0 = 0; it0 < n_it; it0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long it = it0 + d;
      if (it < n_it) {                                     // wave-uniform
        const bool have = (ubeg + unit_of(it) * 64 + lane) < uend;
        const double2 xc = xb[d], yc = yb[d];
        fetch(it + DEPTH, xb[d], yb[d]);
        // classify: whole wave in one cell, every y inside the scale -> run mode
        double u0 = 0.0, u1 = 0.0;
        const int c0 = have ? mq_cell(xc.x, mesh, a.n_mesh, m0, a.inv_delta, u0) : -1;
        const int c1 = have ? mq_cell(xc.y, mesh, a.n_mesh, m0, a.inv_delta, u1) : -2;
        const int cw = __builtin_amdgcn_readfirstlane(c0);
        const double sa = (xc.x - u0) * a.inv_delta - 0.5, sb = (xc.y - u1) * a.inv_delta - 0.5;
        const bool same = have && c0 == cw && c1 == cw && fabs(sa) <= 0.50001 && fabs(sb) <= 0.50001 &&
                          (!SP::needs_y || (fabs(yc.x) <= y0 && fabs(yc.y) <= y0));
        if (__all(same)) {
          if (cw != rcell) { run_flush(); rcell = cw; }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const double s = h ? sb : sa, yv = h ? yc.y : yc.x;
            if (SP::owns_yy) yy = fma(yv, yv, yy);
            rcnt += 1.0;
            double pw = 1.0;
#pragma unroll
            for (int p = 0; p <= SP::max_pow(); ++p) {
              if (p > 0) pw *= s;
              const int qs = p - 1, qt = 2 * K + p;
              if (p >= 1 && qs >= SP::lo && qs < SP::hi && qs < 2 * K) racc[qs - SP::lo] += pw;
              if (p <= K && qt >= SP::lo && qt < SP::hi) racc[qt - SP::lo] = fma(yv, pw, racc[qt - SP::lo]);
            }
          }
        } else {
          run_flush();
          if (have) { scatter(c0, sa, yc.x); scatter(c1, sb, yc.y); }
        }
      }
    }
  }
  run_flush();
  if ((end & 1) && end > beg && tid == 0) {                // odd tail point of the last range
    double u;
    const double xv = a.x[end - 1];
    const int c = mq_cell(xv, mesh, a.n_mesh, m0, a.inv_delta, u);
    scatter(c, (xv - u) * a.inv_delta - 0.5, SP::needs_y ? a.y[end - 1] : 0.0);
  }
  // ---- flush: fixed point -> double (exact to 53 bits), power-of-two unscale, + the fp64 overflow planes
  __syncthreads();
  double* out = a.partials + (size_t)range * mq_planes<K>() * ncells;
  if (SP::has_cnt) for (int e = tid; e < ncells; e += MQ_THREADS) out[e] = (double)cnt[e];
  for (int e = tid; e < NQ * ncells; e += MQ_THREADS) {
    const int j = e / ncells, cc = e - j * ncells, q = SP::lo + j;
    const double v = (double)(long long)planes[e];
    double r;
    if (q < 2 * K) r = ldexp(v, -(s0 + q + 1));
    else { const int p = q - 2 * K; r = ldexp(v, -(s0 - E + p)) + __hip_atomic_load(ovr + (size_t)p * ncells + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    out[(size_t)(1 + q) * ncells + cc] = r;
  }
  double tot = 0.0;
  if (SP::owns_yy) tot = block_sum(yy, scratch);
  double badf = block_sum((double)nbad, scratch + 32);
  if (tid == 0) {
    if (SP::owns_yy) a.scal[2 * range] = tot;
    if (ROLE == 0) a.scal[2 * range + 1] = badf;
  }
}

// grid = n_ranges * G workgroups.  Role and range from the block index: with n_ranges a multiple of 8 the G roles of a range
// have equal blockIdx % 8 (one XCD under round-robin dispatch).
template <int K, int G>
__global__ __launch_bounds__(MQ_THREADS) void phi_moment_kernel(MqArgs a) {
  extern __shared__ double lds[];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < a.tot_n; e += (long)gridDim.x * blockDim.x) a.tot[e] = 0.0;
  int range, role;
  if ((a.n_ranges & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    role = j % G;
    range = (j / G) * 8 + xcd;
  } else {
    role = blockIdx.x % G;
    range = blockIdx.x / G;
  }
  if constexpr (G == 1) phi_moment_body<K, 1, 0>(a, range, lds);
  else if constexpr (G == 2) {
    if (role == 0) phi_moment_body<K, 2, 0>(a, range, lds);
    else phi_moment_body<K, 2, 1>(a, range, lds);
  } else {
    if (role == 0) phi_moment_body<K, 4, 0>(a, range, lds);
    else if (role == 1) phi_moment_body<K, 4, 1>(a, range, lds);
    else if (role == 2) phi_moment_body<K, 4, 2>(a, range, lds);
    else phi_moment_body<K, 4, 3>(a, range, lds);
  }
}

// totals[e] += sum over a slice of ranges of partials[range][e];  e over mq_planes * ncells;  grid (ceil(E / 256), split)
__global__ __launch_bounds__(256) void phi_moment_reduce_kernel(const double* __restrict__ partials, int n_ranges, long E,
                                                                double* __restrict__ tot) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int per = (n_ranges + gridDim.y - 1) / gridDim.y;
  int g0 = blockIdx.y * per, g1 = g0 + per;
  if (g1 > n_ranges) g1 = n_ranges;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int g = g0;
  for (; g + 3 < g1; g += 4) {
    s0 += partials[(size_t)g * E + e];
    s1 += partials[(size_t)(g + 1) * E + e];
    s2 += partials[(size_t)(g + 2) * E + e];
    s3 += partials[(size_t)(g + 3) * E + e];
  }
  for (; g < g1; ++g) s0 += partials[(size_t)g * E + e];
  const double s = (s0 + s1) + (s2 + s3);
  if (s != 0.0) __hip_atomic_fetch_add(tot + e, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Moments -> packed statistics.  One thread per band column j: rows of cell c are c .. c+K (row = c + K - i for piece i), so
// band[d][j] = A[j+d, j] = sum over cells c in [max(0, j+d-K), min(ncells-1, j)] of sum_p pair[i][i+d][p] S_p(c), i = c+K-j-d,
// rhs[j] = sum_c sum_p single[c+K-j][p] T_p(c);  y^T y and the bad-point count are summed over the ranges by block 0.
template <int K>
__global__ __launch_bounds__(256) void phi_moment_convert_kernel(const double* __restrict__ tot, int ncells, long M,
                                                                 const double* __restrict__ scal, int n_ranges,
                                                                 double* __restrict__ stats) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  __shared__ double red[16];
  if (blockIdx.x == 0) {
    double yy = 0.0, bad = 0.0;
    for (int r = threadIdx.x; r < n_ranges; r += 256) { yy += scal[2 * r]; bad += scal[2 * r + 1]; }
    yy = block_sum(yy, red);
    bad = block_sum(bad, red + 8);
    if (threadIdx.x == 0) stats[(long)(K + 2) * M] = (bad > 0.0) ? __builtin_nan("") : yy;   // a point outside the mesh: loud
  }
  if (j >= M) return;
  double band[K + 1], rhs = 0.0;
#pragma unroll
  for (int d = 0; d <= K; ++d) band[d] = 0.0;
#pragma unroll
  for (int i2 = 0; i2 <= K; ++i2) {           // piece of row j in cell c = j - K + i2
    const long c = j - K + i2;
    if (c < 0 || c >= ncells) continue;
    double S[2 * K + 1], T[K + 1];
#pragma unroll
    for (int p = 0; p <= 2 * K; ++p) S[p] = tot[(size_t)p * ncells + c];
#pragma unroll
    for (int p = 0; p <= K; ++p) T[p] = tot[(size_t)(2 * K + 1 + p) * ncells + c];
    double r = 0.0;
#pragma unroll
    for (int p = 0; p <= K; ++p) r = fma(MomCoef<K>::tab.single[i2][p], T[p], r);
    rhs += r;
#pragma unroll
    for (int d = 0; d <= K; ++d) {
      const int i = i2 - d;                   // piece of row j + d in the same cell
      if (i < 0) continue;
      double v = 0.0;
#pragma unroll
      for (int p = 0; p <= 2 * K; ++p) v = fma(MomCoef<K>::tab.pair[i][i2][p], S[p], v);
      band[d] += v;
    }
  }
#pragma unroll
  for (int d = 0; d <= K; ++d) stats[(long)d * M + j] = (j + d < M) ? band[d] : 0.0;
  stats[(long)(K + 1) * M + j] = rhs;
}

}  // namespace asvgp
