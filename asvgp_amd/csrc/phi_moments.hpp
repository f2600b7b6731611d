// Phi pass, algorithm 5 (default where it fits): per-CELL centred moments for Phi Phi^T + direct scatter for Phi y, both
// accumulated in 64-bit fixed point with LDS integer atomics.
//
// Replaces (reference): basis.py:51-76 evaluate_basis + gpr.py:41-44 (Kuf@y, Kuf@Kuf.T, sparse_to_band, sum y^2), D = 1.
// Inside one mesh cell every entry of phi phi^T is a polynomial of degree 2k in the local coordinate, so the band
// statistics of a cell are the 2k+1 moments  S_p = sum s^p (p = 0..2k),  s = t - 1/2 in [-1/2, 1/2]  - 9 numbers at k = 4
// against the 15 products v_i v_j of the band-scatter kernel (algorithms 1, 3), and S_0 is a plain count.  Phi y stays a
// direct scatter of the k+1 products v_i y (its per-cell moment form would need k+1 planes of ncells entries; the direct form
// needs ONE plane of M entries, which is what lets the whole image fit a CU: 2k planes of ncells u64 + counts + M u64 =
// 156 KB at M = 2048, k = 4).  The LDS atomic pipe bounds the kernel (one random-address ds_add_u64 wave-instruction per
// ~11 cycles, tools/micro/lds_atomic_rate.hip): 13 64-bit atomics + one 32-bit one per point instead of 20.
// (A split of ALL 3k+2 moment planes over two workgroup roles reading the same point range was built first: the second
// read of x does not stay an L2 hit - the faster role runs ahead - so it costs 1.5x the memory traffic: 61 us with the
// atomics switched off against 35 us for one read.)
//
// Fixed point: |s^p| <= 2^-p, so plane S_p carries round(s^p * 2^(s0 + p)); Phi y carries round(v_i y * 2^(s0 - E)) with
// |y| <= y0 = 2^E (a per-workgroup scale from a strided sample of its range; points outside it, or NaN, go to a per-workgroup
// fp64 plane in global memory with L2 atomics - never wrong, only slow, and rare by construction).  s0 = min(50, 62 -
// ceil(log2(points per workgroup))): rounding error per addend <= 2^-(s0+1) of the plane's bound, unbiased, and the sums
// inside a workgroup are order-independent.  Signed values ride as two's complement through the same magic-constant
// conversion (one v_add_f64, asvgp_common.hpp fx_convert).
// Cell search (basis.py:58-59 is a TABLE search): the kernel first checks that the mesh is the fp64 linspace numpy makes
// (knot i = i * step + start); then the knots are generated on the VALU, bit for bit the table's, and the search costs no
// memory access - a dependent ds_read would queue behind the 16 waves' outstanding LDS atomics, and a dependent global gather
// adds a second memory round trip per batch (measured: 57 us with the atomics switched off against 35 us).  Other meshes
// (the float32 linspace of Python-float endpoints) take the table from global memory.
// Sorted / time-series input: a wave whose whole batch sits in ONE cell sums its moments per lane, across batches, and commits a
// run when the cell changes (DPP wave reduction, one lane, full double -> int64 conversion) instead of 64 same-address atomics.
// What bounds it (rocprofv3 PMC + compile-time ablations, profiles/r02_phi_*): 83 us = 8 (prologue / epilogue) + 75; with the
// streaming loads switched off it still takes 82 us, with the atomics switched off 45 us - the memory stream (36 us at 4.4 TB/s)
// is fully hidden, and what remains is 45-50 us of LDS-pipe time (13 cycles per random-address ds_add_u64, 64 % of them
// bank-conflict cycles) plus 43 us of VALU time (168 instructions per point at ~4 cycles) that overlap only by a quarter.
// A software-pipelined form (the next batch's cell search interleaved with the current batch's atomics in one fenced basic
// block) and a half-iteration stagger of the second half of the waves were built and measured: no change (84 us), removed.
// Epilogue: every workgroup turns its moments into band entries through the exact integer-ratio tables MomTab (centred
// monomials: <= 2e-14 of the largest entry up to k = 6) and flushes [band | Phi y | y^T y] like the band-scatter kernel, so
// phi_reduce_kernel sums the partials unchanged.
#pragma once

namespace asvgp {

constexpr int MQ_THREADS = 1024;
constexpr int MQ_CH = 32;           // iterations a wavefront stays on one contiguous slice (sorted input: long runs)

// bytes of LDS: 2K moment planes + counts with a COMPILE-TIME plane stride of CS cells (so that the plane offsets fold into the
// ds_add offset field instead of costing an address add - and a spilled SGPR read - per atomic), the Phi y plane, scratch
template <int K, int CS> __host__ __device__ constexpr size_t mq_img_bytes(long M) {
  return ((((size_t)CS * (16 * K + 4) + (size_t)M * 8) + 15) & ~(size_t)15);
}
template <int K, int CS> __host__ inline size_t mq_lds_bytes(long M) { return mq_img_bytes<K, CS>(M) + 64 * 8; }

struct MqArgs {
  const double* x; const double* y; long N;
  const double* mesh_g; int n_mesh; double inv_delta; int M;
  double step;             // REG: (last knot - first knot) / (n_mesh - 1) as the HOST rounds it (numpy.linspace's step)
  double* partials;        // [workgroup][(K+2) M + 1] doubles: band | Phi y | y^T y  (the layout phi_reduce_kernel sums)
  double* ov;              // [workgroup][M] fp64 plane for out-of-scale y (written and read by that workgroup only)
  long ppb;                // points per workgroup (a multiple of 2 * MQ_THREADS)
  double* zero_ptr; long zero_n;   // packed stats buffer to zero (phi_reduce_kernel adds into it afterwards)
  int s0;
};

// |s| = |t - 1/2| bound of an accepted point.  Exactly 1/2 on an exact linspace; the float32-linspace meshes of basis.py:17 wobble by
// ulp32(|knot|) / delta (3e-4 of a cell at [-3.5, 10.5], M = 5000), and the reference extrapolates its pieces there just the same.
// The fixed-point headroom (scale 2^(s0 + p) for s^p, s0 = 62 - ceil(log2(points per workgroup))) needs (2 |s|)^(2K) < 2.
#define MQ_SMAX 0.52

// knot i of numpy.linspace: i * step rounded, THEN + start rounded.  (HIP's __dmul_rn / __dadd_rn are plain * and + and may be
// contracted into an fma - one rounding, a different knot in ~40 % of the cases when start != 0; the pragma forbids it.)
__device__ __forceinline__ double mq_linspace_knot(int i, double step, double m0) {
#pragma clang fp contract(off)
  const double t = (double)i * step;
  return t + m0;
}

// Cell search (basis.py:58-59: idx = max(#{mesh < x} - 1, 0), a TABLE search, not arithmetic).
// mq_guess: arithmetic guess; the two knots around it are fetched one pipeline stage ahead; mq_resolve fixes the guess
// up against the table (rare: x on a knot, or the float32-linspace meshes of basis.py:17 whose spacing wobbles).
typedef double mq_d2u __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ int mq_guess(double x, double m0, double inv_delta, int n_mesh) {
  const double g = floor((x - m0) * inv_delta);
  return (g < 0.0) ? 0 : ((g > (double)(n_mesh - 2)) ? (n_mesh - 2) : (int)g);   // NaN -> 0
}
__device__ __forceinline__ int mq_resolve(double x, int g, double lo, double hi, const double* __restrict__ mesh, int n_mesh, double& u) {
  int i = g;
  if (!(lo < x)) {                                  // x on knot j > 0 belongs to interval j - 1
    while (i > 0 && !(mesh[i] < x)) --i;
    lo = mesh[i];
  } else if (hi < x) {
    while (i < n_mesh - 2 && mesh[i + 1] < x) ++i;
    lo = mesh[i];
  }
  u = lo;
  return i;
}

template <int K, int CS, bool REG>
__global__ __launch_bounds__(MQ_THREADS) void phi_moment_kernel(MqArgs a) {
  extern __shared__ double lds[];
  if (a.zero_ptr) for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < a.zero_n; e += (long)gridDim.x * blockDim.x) a.zero_ptr[e] = 0.0;
  constexpr int NS = 2 * K;                                // planes S_1 .. S_2K
  const int ncells = a.n_mesh - 1, M = a.M;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long* planes = reinterpret_cast<unsigned long long*>(lds);            // [NS][CS]
  unsigned* cnt = reinterpret_cast<unsigned*>(planes + (size_t)NS * CS);               // [CS]
  unsigned long long* rhs = reinterpret_cast<unsigned long long*>(cnt + CS);           // [M]
  const size_t img_bytes = mq_img_bytes<K, CS>(M);
  double* scratch = reinterpret_cast<double*>(reinterpret_cast<char*>(lds) + img_bytes);   // 64 doubles
  {
    double2* z = reinterpret_cast<double2*>(lds);
    for (int e = tid; e < (int)(img_bytes / 16); e += MQ_THREADS) z[e] = make_double2(0.0, 0.0);
  }
  double* ovr = a.ov + (size_t)blockIdx.x * M;             // this workgroup's fp64 overflow plane
  for (int e = tid; e < M; e += MQ_THREADS) ovr[e] = 0.0;
  __syncthreads();
  const double* __restrict__ mesh = a.mesh_g;
  const int n_mesh = a.n_mesh;
  const double inv_delta = a.inv_delta;
  const long beg = (long)blockIdx.x * a.ppb;
  long end = beg + a.ppb;
  const bool full = end <= a.N;                            // workgroup-uniform: every batch of this workgroup is complete
  if (end > a.N) end = a.N;
  const int n_it = (int)(a.ppb / (2 * MQ_THREADS));        // rows of 64 lanes x 2 points per wave
  // super-tiles: each wave streams MQ_CH contiguous rows (sorted input then keeps a wave inside one cell for long)
  const int nfull = n_it / MQ_CH, rem = n_it - nfull * MQ_CH;
  auto unit_of = [&](int it) __attribute__((always_inline)) -> int {
    const int sup = it / MQ_CH, r = it - sup * MQ_CH;
    return (sup < nfull) ? (sup * (MQ_THREADS / 64) + wv) * MQ_CH + r : nfull * (MQ_THREADS / 64) * MQ_CH + wv * rem + r;
  };
  const long ubeg = beg >> 1, uend = (end > beg) ? (end >> 1) : 0;   // pairs
  const double2* x2 = reinterpret_cast<const double2*>(a.x) + ubeg;  // workgroup-uniform bases: per-lane offsets stay 32-bit
  const double2* y2 = reinterpret_cast<const double2*>(a.y) + ubeg;
  const int npair = (int)(uend > ubeg ? uend - ubeg : 0);
  // ---- y scale of this workgroup: 2^E >= 4 max |y| over a strided sample of its range (4 x 1024 pairs)
  int E;
  double y0;
  {
    double m = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = (int)(((long)npair * q) / 4) + tid;
      if (u < npair) { const double2 v = y2[u]; const double t = fmax(fabs(v.x), fabs(v.y)); m = (t == t && t < 1e300) ? fmax(m, t) : m; }
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
  const int chiR = ((1075 - (s0 - E)) << 20) | 0x80000;
  const double m0 = mesh[0];
  const double m_last = mesh[n_mesh - 1];
  const double step = a.step;                              // numpy.linspace: delta / div, divided on the host
  double yy = 0.0;
  unsigned nbad = 0;
  if (REG) {   // the host chose this instantiation from its copy of the mesh; a mesh that is NOT that linspace here is reported loudly
    bool okm = true;
    for (int i = tid; i < n_mesh - 1; i += MQ_THREADS) okm = okm && (mesh[i] == mq_linspace_knot(i, step, m0));
    if (!okm) ++nbad;
  }
  auto knot = [&](int i) __attribute__((always_inline)) -> double { return (i == n_mesh - 1) ? m_last : mq_linspace_knot(i, step, m0); };

  // ---- general code for one point: exact table search (basis.py:58-59), every special case handled per lane
  auto cell_of = [&](double x, double& u) __attribute__((always_inline)) -> int {
    int i = mq_guess(x, m0, inv_delta, n_mesh);
    if (REG) {
      double lo = knot(i);
      if (!(lo < x)) { while (i > 0 && !(knot(i) < x)) --i; lo = knot(i); }
      else if (knot(i + 1) < x) { while (i < n_mesh - 2 && knot(i + 1) < x) ++i; lo = knot(i); }
      u = lo;
      return i;
    }
    return mq_resolve(x, i, mesh[i], mesh[i + 1], mesh, n_mesh, u);
  };
  auto scatter = [&](int c, double s, double yv) __attribute__((always_inline)) {
    if (!(fabs(s) <= MQ_SMAX)) { ++nbad; return; }           // outside the mesh (or NaN): reported, never accumulated
    yy = fma(yv, yv, yy);
    double pw = s;
    __hip_atomic_fetch_add(cnt + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int p = 1; p <= NS; ++p) {
      lds_add_u64(planes + (size_t)(p - 1) * CS + c, fx_convert(pw, chiS - (p << 20)));
      pw *= s;
    }
    double v[K + 1];
    bspline_pieces<K>(s + 0.5, v);
    if (fabs(yv) <= y0) {
#pragma unroll
      for (int i = 0; i <= K; ++i) lds_add_u64(rhs + c + K - i, fx_convert(v[i] * yv, chiR));
    } else {                                               // out of scale (or NaN): exact fp64, L2 atomics
#pragma unroll
      for (int i = 0; i <= K; ++i) __hip_atomic_fetch_add(ovr + c + K - i, v[i] * yv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // the whole wave in ONE cell (sorted / time-series input: a wave stays in a cell for dozens of batches): the moments are summed
  // per lane across batches and leave through ONE wave reduction (DPP, VALU only) + single-lane commits when the cell changes
  double raS[NS], raR[K + 1];
  int rcell = -1, rn = 0;                                    // wave-uniform: cell of the open run, batches in it
#pragma unroll
  for (int p = 0; p < NS; ++p) raS[p] = 0.0;
#pragma unroll
  for (int i = 0; i <= K; ++i) raR[i] = 0.0;
  auto run_flush = [&]() __attribute__((always_inline)) {
    if (rcell < 0) return;
    if (lane == 0) __hip_atomic_fetch_add(cnt + rcell, 128u * (unsigned)rn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int p = 1; p <= NS; ++p) {
      const double t = wave_sum_dpp(raS[p - 1]);
      if (lane == 0) lds_add_u64(planes + (size_t)(p - 1) * CS + rcell, (unsigned long long)__double2ll_rn(ldexp(t, s0 + p)));
      raS[p - 1] = 0.0;
    }
#pragma unroll
    for (int i = 0; i <= K; ++i) {
      const double t = wave_sum_dpp(raR[i]);
      if (lane == 0) lds_add_u64(rhs + rcell + K - i, (unsigned long long)__double2ll_rn(ldexp(t, s0 - E)));
      raR[i] = 0.0;
    }
    rcell = -1;
    rn = 0;
  };
  auto wave_cell = [&](int cw, double sa, double sb, double ya, double yb) __attribute__((always_inline)) {
    if (cw != rcell) { run_flush(); rcell = cw; }
    ++rn;
    yy = fma(ya, ya, fma(yb, yb, yy));
    double pa = sa, pb = sb;
#pragma unroll
    for (int p = 1; p <= NS; ++p) { raS[p - 1] += pa + pb; pa *= sa; pb *= sb; }
    double va[K + 1], vb[K + 1];
    bspline_pieces<K>(sa + 0.5, va);
    bspline_pieces<K>(sb + 0.5, vb);
#pragma unroll
    for (int i = 0; i <= K; ++i) raR[i] += fma(va[i], ya, vb[i] * yb);
  };

  // ---- streaming loop.  One batch = a 16-B pair (x0, x1), (y0, y1) per lane; two batches in flight per wave (ping-pong
  // register sets, the loop statically unrolled by two; loads unconditional with clamped offsets, so that no branch costs the
  // compiler its count of what is in flight).  Every batch is classified by wave votes:
  //   common - every lane holds two in-range, in-scale points whose guessed cell needs at most one correction step: a branch-free
  //            block (26 ds_add_u64 + 2 ds_add_u32 per lane pair, no exec-mask traffic);
  //   whole wave in ONE cell: wave_cell;   anything else (last partial batch, outliers, far-off guesses): the general code.
  if (npair > 0) {
    const int ulast = npair - 1;
    double2 xA, yA, xB, yB;
    auto fetch = [&](int it, double2& xo, double2& yo) __attribute__((always_inline)) {
      int u = unit_of(it < n_it ? it : n_it - 1) * 64 + lane;
      u = u < ulast ? u : ulast;
      // streaming loads marked non-temporal: x, y are read exactly once, and in the steps-in-flight schedule the band chains of the
      // previous step run beside this kernel out of an L2-resident workspace that the 160 MB stream would otherwise evict
      typedef double mq_nt2 __attribute__((ext_vector_type(2)));
      const mq_nt2 xv = __builtin_nontemporal_load(reinterpret_cast<const mq_nt2*>(x2 + u));
      const mq_nt2 yv = __builtin_nontemporal_load(reinterpret_cast<const mq_nt2*>(y2 + u));
      xo = make_double2(xv.x, xv.y); yo = make_double2(yv.x, yv.y);
    };
    auto batch = [&](int it, double2& xbuf, double2& ybuf) __attribute__((always_inline)) {
      const double2 xc = xbuf, yc = ybuf;
      fetch(it + 2, xbuf, ybuf);
      bool fast = false;
      int c0 = 0, c1 = 0;
      double sa = 0.0, sb = 0.0;
      if (REG && full) {                                   // (uniform) branch-free classification
        // the arithmetic guess IS the cell unless x sits within rounding of a knot: validated against the two knots around it
        // (basis.py:58-59: mesh[c] < x <= mesh[c+1], open ends at the boundary cells); a miss anywhere in the wave -> general code
        c0 = mq_guess(xc.x, m0, inv_delta, n_mesh);
        c1 = mq_guess(xc.y, m0, inv_delta, n_mesh);
        const double u0 = knot(c0), u0n = knot(c0 + 1), u1 = knot(c1), u1n = knot(c1 + 1);
        const bool v0 = (c0 == 0 || u0 < xc.x) && (c0 == n_mesh - 2 || !(u0n < xc.x));
        const bool v1 = (c1 == 0 || u1 < xc.y) && (c1 == n_mesh - 2 || !(u1n < xc.y));
        sa = (xc.x - u0) * inv_delta - 0.5;
        sb = (xc.y - u1) * inv_delta - 0.5;
        const bool good = v0 && v1 && fabs(sa) <= MQ_SMAX && fabs(sb) <= MQ_SMAX && fabs(yc.x) <= y0 && fabs(yc.y) <= y0;
        fast = __all(good);
      }
      if (fast) {
        const int cw = __builtin_amdgcn_readfirstlane(c0);
        if (__all(c0 == cw && c1 == cw)) {
          wave_cell(cw, sa, sb, yc.x, yc.y);
        } else {
          run_flush();
          yy = fma(yc.x, yc.x, fma(yc.y, yc.y, yy));
          __hip_atomic_fetch_add(cnt + c0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(cnt + c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          double pa = sa, pb = sb;
#pragma unroll
          for (int p = 1; p <= NS; ++p) {
            lds_add_u64(planes + (size_t)(p - 1) * CS + c0, fx_convert(pa, chiS - (p << 20)));
            lds_add_u64(planes + (size_t)(p - 1) * CS + c1, fx_convert(pb, chiS - (p << 20)));
            pa *= sa; pb *= sb;
          }
          double va[K + 1], vb[K + 1];
          bspline_pieces<K>(sa + 0.5, va);
          bspline_pieces<K>(sb + 0.5, vb);
#pragma unroll
          for (int i = 0; i <= K; ++i) {
            lds_add_u64(rhs + c0 + K - i, fx_convert(va[i] * yc.x, chiR));
            lds_add_u64(rhs + c1 + K - i, fx_convert(vb[i] * yc.y, chiR));
          }
        }
      } else {                                             // ---- general code
        const bool have = (unit_of(it) * 64 + lane) < npair;
        double u0, u1;
        c0 = cell_of(xc.x, u0); c1 = cell_of(xc.y, u1);
        sa = (xc.x - u0) * inv_delta - 0.5; sb = (xc.y - u1) * inv_delta - 0.5;
        const int cw = __builtin_amdgcn_readfirstlane(c0);
        const bool same = have && c0 == cw && c1 == cw && fabs(sa) <= MQ_SMAX && fabs(sb) <= MQ_SMAX &&
                          fabs(yc.x) <= y0 && fabs(yc.y) <= y0;
        if (__all(same)) wave_cell(cw, sa, sb, yc.x, yc.y);
        else {
          run_flush();
          if (have) { scatter(c0, sa, yc.x); scatter(c1, sb, yc.y); }
        }
      }
    };
    fetch(0, xA, yA);
    fetch(1, xB, yB);
    for (int it = 0; it < n_it; it += 2) {                 // wave-convergent throughout (votes, DPP)
      batch(it, xA, yA);
      if (it + 1 < n_it) batch(it + 1, xB, yB);
    }
    run_flush();
  }
  if ((end & 1) && end > beg && tid == 0) {                // odd tail point (only the last workgroup with points can have one)
    double u;
    const double xv = a.x[end - 1];
    const int c = cell_of(xv, u);
    scatter(c, (xv - u) * inv_delta - 0.5, a.y[end - 1]);
  }
  double tot = block_sum(yy, scratch);                     // (contains the barrier that orders the LDS atomics before the read-out)
  const double badf = block_sum((double)nbad, scratch + 32);
  __syncthreads();
  // ---- epilogue: moments -> band entries of this workgroup, fixed point -> double, one coalesced flush.
  // Rows of cell c are c .. c+K (row = c + K - i for piece i):  band[d][j] = A[j+d, j] = sum over the cells c = j-K+i2 (i2 = piece of
  // row j) of sum_p pair[i2-d][i2][p] S_p(c);  Phi y[j] = fixed-point plane + fp64 overflow plane.
  double* out = a.partials + (size_t)blockIdx.x * ((size_t)(K + 2) * M + 1);
  for (int j = tid; j < M; j += MQ_THREADS) {
    double band[K + 1];
#pragma unroll
    for (int d = 0; d <= K; ++d) band[d] = 0.0;
#pragma unroll
    for (int i2 = 0; i2 <= K; ++i2) {
      const int c = j - K + i2;
      if (c < 0 || c >= ncells) continue;
      double S[NS + 1];
      S[0] = (double)cnt[c];
#pragma unroll
      for (int p = 1; p <= NS; ++p) S[p] = ldexp((double)(long long)planes[(size_t)(p - 1) * CS + c], -(s0 + p));
#pragma unroll
      for (int d = 0; d <= i2; ++d) {
        double v = 0.0;
#pragma unroll
        for (int p = 0; p <= NS; ++p) v = fma(MomCoef<K>::tab.pair[i2 - d][i2][p], S[p], v);
        band[d] += v;
      }
    }
#pragma unroll
    for (int d = 0; d <= K; ++d) __builtin_nontemporal_store((j + d < M) ? band[d] : 0.0, out + (size_t)d * M + j);   // written once, read once by the reduce
    out[(size_t)(K + 1) * M + j] = ldexp((double)(long long)rhs[j], -(s0 - E)) +
                                   __hip_atomic_load(ovr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) out[(size_t)(K + 2) * M] = (badf > 0.0) ? __builtin_nan("") : tot;   // a point outside the mesh: loud (NaN y^T y)
}

}  // namespace asvgp
