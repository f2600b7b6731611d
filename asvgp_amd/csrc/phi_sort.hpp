// Phi pass, algorithm 6 ("tile sort", the default where it applies): per-cell centred moments accumulated in REGISTERS.
//
// Replaces (reference): basis.py:51-76 evaluate_basis + gpr.py:41-44 (Kuf@y, Kuf@Kuf.T, sparse_to_band, sum y^2), D = 1.
// Same sufficient statistics as algorithm 5 (phi_moments.hpp): inside one mesh cell every entry of phi phi^T is a polynomial of
// degree 2k in the centred local coordinate s = t - 1/2, so a cell's share of Phi Phi^T is the 2k+1 power sums S_p = sum s^p and
// its share of Phi y the k+1 sums T_p = sum y s^p.  Algorithm 5 pays 3k+2 random-address LDS atomics and as many double -> fixed
// point conversions per point (LDS pipe 39 us + VALU 42 us at N = 10M); here no statistic ever leaves the register file:
//   thread t of the ONE 1024-thread workgroup per CU owns cells t and t + 1024 for the whole kernel (2 (3k+1) fp64 accumulators);
//   the workgroup streams its points in tiles of TP x 1024; per tile
//     P1  every lane finds cell and s of its TP points and takes a rank inside the cell with ONE returning ds_add_u32,
//     P2  exclusive scan of the 2048 per-cell counts (DPP inside a wave, 16 wave totals through the LDS),
//     P3  (s, y) scattered into cell order: one ds_write_b128 per point,
//     P4  owners walk their two cells' runs: one ds_read_b128 + 4k fp64 operations per point (powers of s shared by S and T),
//         wave-uniform trip count, lanes that have run out read a (0, 0) point which adds nothing;
//   the next tile's loads are in flight during P4; cells with more than PS_HEAVY points in a tile (sorted / clustered input) are
//   summed by the whole wavefront (DPP reduction), so time-series order is not a worst case.
// Per point: 1 LDS atomic + 1 16-B write + 1 16-B read instead of 14 atomics, ~60 VALU instructions instead of ~157.
// The sums inside a cell follow the arrival order of the rank atomics: results are reproducible to rounding (a few ulp of the
// per-cell sums), not bit for bit - algorithm 5 (fixed point, order-independent) stays selectable for that.
// What bounds it (tools/micro/phi_sort_bench.hip: per-stage ablations and in-kernel phase stamps; N = 10M, M = 2048, k = 4, TP = 6):
// 118 K cycles per workgroup ~ 58-61 us = P1 15 K (search 6.5, ranks 5, barrier) + scan 9.5 K (three barriers) + scatter 18 K (LDS: a
// random-address ds_write_b128 per point) + owners 62 K (VALU: 16 fp64 operations per point at 4.5 cycles with four waves per SIMD,
// but only 1 in 2.8 lane-iterations carries a point: the loops run as long as the longest of 64 Poisson(3) runs) + epilogue 13 K.
// The memory stream is hidden (the tile's loads have landed when P1 starts); a read-only kernel of the same launch shape takes
// 23-25 us (tools/micro/phi_sort_bench.hip `stream ceiling`).  Measured and NOT adopted: tiles of 8 points per thread (longer runs,
// 13 % fewer owner iterations, but 11 spilled registers whose reloads wait on the prefetched tile: +8 us); issuing the next tile's
// loads before the scan instead of behind the scatter (no change: the stream is not the bound); a run-length cap per cell and tile
// with the surplus points carried to the next tile through an LDS queue (owner loops 62 K -> 57 K cycles, but scatter, scan and
// rank phases +20 K: 8 us slower); the in-wave heavy-cell path of the first version (sorted input: one wave walks 8192 points).
// Applies to meshes that are an exact numpy.linspace (the host decides, the kernel re-checks the table), D = 1, 16-B aligned
// x / y, at most 2048 cells.  Epilogue: moments -> band entries through the exact integer-ratio tables MomTab, one partial
// [band | Phi y | y^T y] per workgroup, summed by phi_reduce_kernel exactly like the other algorithms.
#pragma once

namespace asvgp {

constexpr int PS_THREADS = 1024;
constexpr int PS_NCELL = 2 * PS_THREADS;   // cells an image holds (two per thread)
constexpr int PS_HEAVY = 48;               // points of one cell in one tile beyond which the whole workgroup sums the cell
constexpr int PS_HROUND = 16;              // heavy cells summed per round (slots of the LDS hand-over table)
constexpr int PS_HLIST = 8192 / (PS_HEAVY + 1) + 1;   // most heavy cells a tile can have

struct PsArgs {
  const double* x; const double* y; long N;
  const double* mesh_g; int n_mesh; double inv_delta; int M;
  double m0, m_last;       // first / last knot (the host's copy; the kernel checks the table against them)
  double step;             // (last knot - first knot) / (n_mesh - 1) as the HOST rounds it (numpy.linspace's step)
  double smax_fast;        // |s| below which the arithmetic cell guess is certainly the table's cell (1/2 - knot rounding margin)
  double* partials;        // [workgroup][(K+2) M + 1] doubles: band | Phi y | y^T y  (the layout phi_reduce_kernel sums)
  long ppb;                // points per workgroup (even)
  double* zero_ptr; long zero_n;   // packed stats buffer to zero (phi_reduce_kernel adds into it afterwards)
  unsigned long long* stamps;      // diagnostics: per-phase cycles of one wave's lane 0 (ABL == 9), else unused
  int stamps_wave;
  int* ranges;                     // [workgroup][2]: first / last COLUMN this workgroup's partial holds (the rest of it is not written, and
                                   // phi_reduce_kernel does not read it); NULL: every partial is written in full
};

// points per thread and tile: as many as leave the 2 (3k+1) owner accumulators, the tile in flight and the current tile's (s, y, rank)
// inside 128 VGPRs without a spill (a scratch reload counts in vmcnt and stalls on the prefetched tile)
template <int K> constexpr int ps_tile_points() { return K <= 3 ? 8 : (K == 4 ? 6 : (K == 5 ? 4 : 2)); }
#ifndef PS_GW
#define PS_GW 4
#endif
constexpr int PS_TS_SLOTS = 256;           // cells the time-series table of a workgroup holds (open addressing)
template <int K> constexpr size_t ps_ts_bytes() { return (size_t)PS_TS_SLOTS * (3 * K + 2) * 8 + (size_t)2 * PS_TS_SLOTS * 4; }
template <int K, int TP, int TS = 0> constexpr size_t ps_lds_bytes() {
  return (size_t)(TP * PS_THREADS + 1) * 16 + (size_t)3 * PS_NCELL * 4 + 64 * 4 + 64 * 8 +
         (size_t)PS_HROUND * (3 * K + 1) * 8 + (size_t)PS_HLIST * 12 + (TS ? ps_ts_bytes<K>() : 0);
}
// LDS of the epilogue's largest round: the Q planes of one half (all (K+1)(K+2)/2 for K <= 4, else those of the sub-diagonals d >= 2 or
// d < 2, whichever are more) or the 2 (K+1) rhs planes
template <int K> constexpr size_t ps_epilogue_bytes() {
  int all = (K + 1) * (K + 2) / 2, lo = (K + 1) + K, hi = all - lo;
  int planes = K <= 4 ? all : (lo > hi ? lo : hi);
  if (planes < 2 * (K + 1)) planes = 2 * (K + 1);
  return (size_t)planes * (PS_THREADS + K) * 8;
}

// Workgroup barrier that orders LDS traffic only: global loads issued before it (the next tile's prefetch) stay in flight across
// it (__syncthreads() makes hipcc drain them with s_waitcnt vmcnt(0)).
__device__ __forceinline__ void ps_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave64 inclusive scan on the VALU (DPP row shifts + row broadcasts)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned ps_dpp_add_u32(unsigned v) {
  return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ unsigned ps_wave_scan_incl(unsigned v) {
  v = ps_dpp_add_u32<0x111, 0xf>(v);   // row_shr:1
  v = ps_dpp_add_u32<0x112, 0xf>(v);   // row_shr:2
  v = ps_dpp_add_u32<0x114, 0xf>(v);   // row_shr:4
  v = ps_dpp_add_u32<0x118, 0xf>(v);   // row_shr:8   -> inclusive scan inside every row of 16
  v = ps_dpp_add_u32<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = ps_dpp_add_u32<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}

// wave64 maximum, returned wave-uniform (SGPR)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned ps_dpp_max_u32(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
  return v > o ? v : o;
}
__device__ __forceinline__ unsigned ps_wave_max_u32(unsigned v) {
  v = ps_dpp_max_u32<0x111, 0xf>(v);
  v = ps_dpp_max_u32<0x112, 0xf>(v);
  v = ps_dpp_max_u32<0x114, 0xf>(v);
  v = ps_dpp_max_u32<0x118, 0xf>(v);
  v = ps_dpp_max_u32<0x142, 0xa>(v);
  v = ps_dpp_max_u32<0x143, 0xc>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// wave64 minimum of values < 2^31, wave-uniform
__device__ __forceinline__ unsigned ps_wave_min_u32(unsigned v) { return 0x7fffffffu - ps_wave_max_u32(0x7fffffffu - v); }

// one point into a cell's moments: powers s^1..s^K once (shortest chains), S_p = sum s^p (p = 1..2K), T_p = sum y s^p (p = 0..K)
template <int K>
__device__ __forceinline__ void ps_acc(double s, double y, double (&S)[2 * K], double (&Tm)[K + 1]) {
  double pw[K + 1];
  pw[0] = 1.0;
  pw[1] = s;
#pragma unroll
  for (int p = 2; p <= K; ++p) pw[p] = pw[p / 2] * pw[p - p / 2];
#pragma unroll
  for (int p = 1; p <= K; ++p) S[p - 1] += pw[p];
#pragma unroll
  for (int p = K + 1; p <= 2 * K; ++p) S[p - 1] = fma(pw[K], pw[p - K], S[p - 1]);
  Tm[0] += y;
#pragma unroll
  for (int p = 1; p <= K; ++p) Tm[p] = fma(y, pw[p], Tm[p]);
}

// Owner lane: the run of l points at buf[o..] of ONE cell into its moments.  Wave-uniform trip count (the longest run of the wave's 64
// cells, in an SGPR: a loop on a wave vote makes hipcc copy all 3k+1 accumulators every iteration); a lane that has run out reads the
// (0, 0) point at buf[ZS], which adds nothing.  Two points per iteration in two register sets, each load issued one point ahead of its
// use.  (Heavy cells arrive here with l = 0: ps_heavy_cells has summed them.)
template <int K, int ZS>
__device__ __forceinline__ void ps_own_cell(const double2* buf, unsigned l, unsigned o, double (&S)[2 * K], double (&Tm)[K + 1]) {
  const unsigned nmax = (ps_wave_max_u32(l) + 1u) & ~1u;
  double2 p = buf[l > 0 ? o : (unsigned)ZS];
  for (unsigned j = 0; j < nmax; j += 2) {
    const double2 q = buf[(j + 1 < l) ? o + j + 1 : (unsigned)ZS];
    __builtin_amdgcn_sched_barrier(0);
    ps_acc<K>(p.x, p.y, S, Tm);
    __builtin_amdgcn_sched_barrier(0);
    p = buf[(j + 2 < l) ? o + j + 2 : (unsigned)ZS];
    __builtin_amdgcn_sched_barrier(0);
    ps_acc<K>(q.x, q.y, S, Tm);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Heavy cells: more than PS_HEAVY points of one tile in one cell (sorted / time-series / clustered input - an owner lane would walk
// them alone while its wavefront, or the whole workgroup, waits).  The scan phase lists them (cell, count, offset); here FOUR waves per
// cell (wave numbers equal mod 4 share a SIMD under the cyclic placement; the group rotates with the cell) take slices of 64 consecutive
// points of the listed run, sum them lane-parallel - in TWO passes, S then T, so that at most 2k temporaries are live next to the
// owner accumulators -, reduce on the VALU (DPP: ~700 cycles per wave and cell, which is why not all 16 waves take part) and add
// their 3k+1 partial moments to the cell's slot of a small LDS table with ds_add_f64; the owner lane then takes the slot.  Rounds of
// PS_HROUND cells.
// Sum each of up to 8 per-lane values over the 64 lanes as a reduce-scatter butterfly: after the exchanges with lanes ^1, ^2, ^4 a lane
// carries ONE of the values summed over its group of 8, three more exchanges finish it - 10 shuffles instead of 8 x 6 DPP steps (the
// separate wave reductions were 83 K of the 164 K cycles a workgroup spends on sorted input).  Returns the total of value `idx`.
// (the exchanges with lanes ^ 1, ^ 2, ^ 4, ^ 8 are DPP moves on the VALU - asvgp_common.hpp dpp_move_f64 -: as ds_bpermute-based __shfl_xor the
// two butterflies of a slice group were ~5 us of a sorted input's pass)
template <int NV>
__device__ __forceinline__ double ps_reduce_scatter8(const double (&v)[NV], int lane, int& idx) {
  static_assert(NV <= 8, "at most eight values");
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
  double a8[8], s4[4], s2[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) a8[i] = i < NV ? v[i] : 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double keep = b0 ? a8[4 + i] : a8[i], send = b0 ? a8[i] : a8[4 + i]; s4[i] = keep + dpp_move_f64<0xB1>(send); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { const double keep = b1 ? s4[2 + i] : s4[i], send = b1 ? s4[i] : s4[2 + i]; s2[i] = keep + dpp_move_f64<0x4E>(send); }
  // lane ^ 4 = 4 lanes up (bit 2 clear: row_ror:12) or down (set: row_ror:4) inside the row of 16; lane ^ 8 = row_ror:8
  double s1 = (b2 ? s2[1] : s2[0]) + dpp_xor4_f64(b2 ? s2[0] : s2[1], b2);
  s1 += dpp_move_f64<0x128>(s1);
  s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
  idx = (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0);
  return s1;
}

template <int K>
__device__ __forceinline__ void ps_heavy_slices(const double2* buf, unsigned hn, unsigned ho, int first, int stride, int lane, double* slot) {
  constexpr int NS = 2 * K;
  const unsigned nsl = (hn + 63u) >> 6;
  // ONE pass: every point read once, powers shared by the S and the T sums (two passes - S then T - re-read the run and re-formed the
  // powers: ~25 % more instructions in what is half of a sorted input's kernel time)
  double S2[NS], T2[K + 1];
#pragma unroll
  for (int q = 0; q < NS; ++q) S2[q] = 0.0;
#pragma unroll
  for (int q = 0; q <= K; ++q) T2[q] = 0.0;
  for (unsigned sl = (unsigned)first; sl < nsl; sl += (unsigned)stride) {
    const unsigned j = sl * 64 + lane;
    if (j < hn) {
      const double2 pt = buf[ho + j];
      ps_acc<K>(pt.x, pt.y, S2, T2);
    }
  }
  if constexpr (NS <= 8) {
    int idx;
    const double t = ps_reduce_scatter8<NS>(S2, lane, idx);
    if (lane < 8 && idx < NS) lds_add(slot + idx, t);
  } else {
#pragma unroll
    for (int q = 0; q < NS; ++q) { const double t = wave_sum_dpp(S2[q]); if (lane == 0) lds_add(slot + q, t); }
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    int idx;
    const double t = ps_reduce_scatter8<K + 1>(T2, lane, idx);
    if (lane < 8 && idx <= K) lds_add(slot + NS + idx, t);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// ABL: 0 product; 1 loads + cell search only; 2 + rank atomics; 3 + scan; 4 + scatter; 5 + owners (no epilogue conversion);
//      9 product + per-phase cycle stamps of thread 0
//      TS: 1 = the time-series front loop is compiled in (see below)
template <int K, int TP, int ABL = 0, int PF = 0, int TS = 0>
__global__ __launch_bounds__(PS_THREADS) void phi_sort_kernel(PsArgs a) {
  extern __shared__ double lds[];
  static_assert(TP % 2 == 0 && TP * PS_THREADS <= 8192, "tile: rank field is 13 bits");
  constexpr int T = TP * PS_THREADS;
  constexpr int NS = 2 * K;
  double2* buf = reinterpret_cast<double2*>(lds);                 // T sorted (s, y) + slot T = (0, 0)
  unsigned* cnt = reinterpret_cast<unsigned*>(buf + T + 1);       // [2][PS_NCELL] per-tile histogram, double-buffered
  unsigned* off = cnt + 2 * PS_NCELL;                             // [PS_NCELL]
  unsigned* wtot = off + PS_NCELL;                                // [16] wave totals of the scan (+ pad)
  double* scratch = reinterpret_cast<double*>(wtot + 64);         // 64 doubles
  constexpr int NSTAT = 3 * K + 1;
  double* hacc = scratch + 64;                                    // [PS_HROUND][NSTAT] hand-over table of the heavy cells
  unsigned* hlist = reinterpret_cast<unsigned*>(hacc + PS_HROUND * NSTAT);   // [PS_HLIST] x (cell, count, offset)
  unsigned* nheavy_p = wtot + 32;                                 // heavy cells of the current tile
  // time-series table: the runs the front loop hands over, keyed by cell (open addressing), read by the owners before the epilogue
  constexpr int TSW = NSTAT + 1;                                  // 3k+1 sums + the count
  double* tstab = reinterpret_cast<double*>(hlist + 3 * PS_HLIST);   // [PS_TS_SLOTS][TSW]
  int* tskey = reinterpret_cast<int*>(tstab + PS_TS_SLOTS * TSW);    // [PS_TS_SLOTS] cell of each slot, -1 = free
  int* tslist = tskey + PS_TS_SLOTS;                              // [PS_TS_SLOTS] occupied slots in order of allocation
  unsigned* ts_ctl = wtot + 40;                                   // [0] slots in use, [1] "leave the front loop"
  if (a.zero_ptr) for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < a.zero_n; e += (long)gridDim.x * blockDim.x) a.zero_ptr[e] = 0.0;
  const int tid0 = threadIdx.x;
  int tid = tid0, lane = tid0 & 63, wv = tid0 >> 6;
  const int n_mesh = a.n_mesh, ncells = n_mesh - 1, M = a.M;
  const double* __restrict__ mesh = a.mesh_g;
  const double inv_delta = a.inv_delta, step = a.step, smax_fast = a.smax_fast;
  const double m0 = a.m0, m_last = a.m_last;
  {
    uint4* z = reinterpret_cast<uint4*>(cnt);
    z[tid] = make_uint4(0u, 0u, 0u, 0u);                          // 2 x 2048 counters = 1024 x 16 B
    if (tid == 0) { buf[T] = make_double2(0.0, 0.0); ts_ctl[0] = 0; ts_ctl[1] = 0; }
    if constexpr (TS == 1) {
      if (tid < PS_TS_SLOTS) tskey[tid] = -1;
      for (int e = tid; e < PS_TS_SLOTS * TSW; e += PS_THREADS) tstab[e] = 0.0;
    }
  }
  unsigned nbad = 0;
  auto knot = [&](int i) __attribute__((always_inline)) -> double { return (i == n_mesh - 1) ? m_last : mq_linspace_knot(i, step, m0); };
  // Exact table rule (basis.py:58-59: idx = max(#{mesh < x} - 1, 0)) for the rare point within rounding of a knot or outside the mesh.
  // On a linspace the arithmetic guess is off by at most one cell for a point of [a, b]: one step either way, then the rule is
  // VERIFIED; what does not verify (or lies more than MQ_SMAX - 1/2 of a cell outside) is not accumulated and reported.
  auto cell_slow = [&](double x, double& s_out, bool& ok) __attribute__((always_inline)) -> int {
    int i = mq_guess(x, m0, inv_delta, n_mesh);                   // clamped to [0, n_mesh - 2]; NaN -> 0
    const double k0 = knot(i);
    const bool down = !(k0 < x) && i > 0;
    const bool up = !down && i < n_mesh - 2 && knot(i + 1) < x;
    i += up ? 1 : (down ? -1 : 0);
    const double lo = knot(i), hi = knot(i + 1);
    const double s = (x - lo) * inv_delta - 0.5;
    ok = (i == 0 || lo < x) && (i == n_mesh - 2 || !(hi < x)) && fabs(s) <= MQ_SMAX;
    s_out = s;
    return i;
  };

  const long beg = (long)blockIdx.x * a.ppb;
  long end = beg + a.ppb;
  if (end > a.N) end = a.N;
  if (end < beg) end = beg;
  const long ubeg = beg >> 1;
  const int npair = (int)((end >> 1) - ubeg);                     // full pairs of this workgroup (beg is even)
  const bool tail = (end & 1) != 0;                               // one odd last point (only the workgroup that reaches N)
  const int nunit = npair + (tail ? 1 : 0);
  // Point -> thread mapping.  TS = 0: row q2 of tile t is the 1024 pairs t T/2 + q2 1024 + tid.  TS = 1: the workgroup's pairs form rows of
  // 64; the 16 waves form 16 / PS_GW groups, every group streams its own CONTIGUOUS share of the rows, its PS_GW waves taking
  // consecutive rows in turn (wave w' of the group: the group's rows w', w' + PS_GW, ...), TP / 2 rows per wave and tile.  PS_GW = 16 is
  // the TS = 0 order up to a permutation inside the tile; the smaller PS_GW, the longer a sorted / time-series input keeps a wave
  // inside one cell (a wave's consecutive rows are PS_GW rows apart) and the shorter the contiguous piece a load instruction of the
  // workgroup reads (PS_GW KiB).
  constexpr int GW = PS_GW, NG = (PS_THREADS / 64) / GW;
  const int nrows = (nunit + 63) >> 6;
  const int rpg = (nrows + NG - 1) / NG;                          // rows per group
  const int rpw = (rpg + GW - 1) / GW;                            // rows per wave
  const int n_tiles = TS == 1 ? (rpw + TP / 2 - 1) / (TP / 2) : (nunit + T / 2 - 1) / (T / 2);
  auto unit_of = [&](int tile, int q2, bool& inside) __attribute__((always_inline)) -> int {
    if constexpr (TS == 1) {
      const int w = tid >> 6, g = w / GW, wi = w - g * GW;
      const int gr = (tile * (TP / 2) + q2) * GW + wi;             // row inside the group
      const int row = g * rpg + gr;
      inside = gr < rpg && row < nrows;
      return (row << 6) + (tid & 63);
    } else {
      inside = true;
      return tile * (T / 2) + q2 * PS_THREADS + tid;
    }
  };
  typedef double ps_nt2 __attribute__((ext_vector_type(2)));
  // (a workgroup without a full pair - only possible at the very end of the data - aims its unconditional loads at pair 0 of the
  // arrays: the host sends N < 2 to another algorithm)
  const ps_nt2* x2 = reinterpret_cast<const ps_nt2*>(a.x) + (npair > 0 ? ubeg : 0);
  const ps_nt2* y2 = reinterpret_cast<const ps_nt2*>(a.y) + (npair > 0 ? ubeg : 0);
  const int ulast = npair > 0 ? npair - 1 : 0;

  double xs[TP], ys[TP];
  // unconditional, clamped loads and NO branch around them: a branch makes the loaded values phi nodes, which hipcc copies right
  // behind the loads (s_waitcnt vmcnt(1)) - the prefetch then overlaps nothing
  auto load_into = [&](int tile, double (&X)[TP], double (&Y)[TP]) __attribute__((always_inline)) {
#pragma unroll
    for (int q2 = 0; q2 < TP / 2; ++q2) {
      bool inside;
      int u = unit_of(tile, q2, inside);
      u = u < ulast ? u : ulast;
      const ps_nt2 xv = __builtin_nontemporal_load(x2 + u);     // read exactly once: keep the stream out of the L2's LRU order
      const ps_nt2 yv = __builtin_nontemporal_load(y2 + u);
      X[2 * q2] = xv.x; X[2 * q2 + 1] = xv.y; Y[2 * q2] = yv.x; Y[2 * q2 + 1] = yv.y;
    }
  };
  auto load_tile = [&](int tile) __attribute__((always_inline)) { load_into(tile, xs, ys); };

  double SA[NS], TA[K + 1], SB[NS], TB[K + 1];
#pragma unroll
  for (int p = 0; p < NS; ++p) { SA[p] = 0.0; SB[p] = 0.0; }
#pragma unroll
  for (int p = 0; p <= K; ++p) { TA[p] = 0.0; TB[p] = 0.0; }
  unsigned n0A = 0, n0B = 0;
  double yy = 0.0;

  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  auto stamp = [&](int i) __attribute__((always_inline)) {
    if constexpr (ABL == 9) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (i >= 0) ph[i] += t - tprev;
      tprev = t;
    }
  };

  // ---- cell and centred coordinate of the tile held in (xs, ys): the first half of P1, shared by the front loop and the sort loop.
  // One pair at a time (sched_barrier): the scheduler otherwise interleaves all TP points and needs ~120 registers for this block
  // alone, next to the 2 (3k+1) owner accumulators.  cr = the cell; bit q of valm: point q exists and lies inside the mesh.
  double sv[TP], yv[TP];
  int cr[TP];
  unsigned valm = 0;
  int t_done = 0;                                                 // tiles whose rows THIS wave has summed in the front loop (the sort loop skips them)
  // (tile_bad: where the tile's number of points outside the mesh is added)
  auto search_from = [&](int tile, const double (&X)[TP], const double (&Y)[TP], unsigned& tile_bad) __attribute__((always_inline)) {
    valm = 0;
#pragma unroll
    for (int q2 = 0; q2 < TP / 2; ++q2) {
      bool inside;
      const int u = unit_of(tile, q2, inside);
      double xv[2] = {X[2 * q2], X[2 * q2 + 1]};
      bool val[2] = {inside && u < npair, inside && u < npair};
      yv[2 * q2] = Y[2 * q2]; yv[2 * q2 + 1] = Y[2 * q2 + 1];
      if (tail && inside && u == npair) {                                   // the odd last point: a scalar reload by ONE lane of the kernel
        xv[0] = a.x[end - 1]; yv[2 * q2] = a.y[end - 1]; val[0] = true;
      }
      bool slow = false;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double g = floor((xv[e] - m0) * inv_delta);
        const int c = __double2int_rz(g);                         // (v_cvt_i32_f64: saturating; NaN -> 0)
        double u0;
        {
#pragma clang fp contract(off)
          const double t = g * step;                              // numpy.linspace's knot: i * step rounded, THEN + start rounded
          u0 = t + m0;
        }
        const double s = (xv[e] - u0) * inv_delta - 0.5;
        const bool fast = (unsigned)c < (unsigned)ncells && fabs(s) <= smax_fast;
        slow = slow || (val[e] && !fast);
        cr[2 * q2 + e] = c;
        sv[2 * q2 + e] = s;
      }
      if (__any(slow)) {                                          // rare: the exact table rule, per lane
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          bool ok;
          double sq;
          const int c = cell_slow(xv[e], sq, ok);
          cr[2 * q2 + e] = c;
          sv[2 * q2 + e] = sq;
          if (val[e] && !ok) { ++tile_bad; val[e] = false; }        // outside the mesh (or NaN): reported, never accumulated
        }
      }
      valm |= (val[0] ? 1u : 0u) << (2 * q2) | (val[1] ? 2u : 0u) << (2 * q2);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  load_tile(0);
  {   // the host chose this kernel from its copy of the mesh; a table that is NOT that linspace here is reported loudly.  (Behind the first
      // tile's loads: in front of them its two dependent reads of the table were two memory round trips before anything streamed.)
    int okm = 1;                                                  // (no short circuit, no branch: a load must not wait for a comparison before it)
    const double last_v = mesh[n_mesh - 1];
    const int i0 = tid < n_mesh - 1 ? tid : 0, i1 = tid + PS_THREADS < n_mesh - 1 ? tid + PS_THREADS : 0;
    const double v0 = mesh[i0], v1 = mesh[i1];
    okm &= (v0 == mq_linspace_knot(i0, step, m0)) ? 1 : 0;
    okm &= (v1 == mq_linspace_knot(i1, step, m0)) ? 1 : 0;
    okm &= (last_v == m_last) ? 1 : 0;
    for (int i = tid + 2 * PS_THREADS; i < n_mesh - 1; i += PS_THREADS) okm &= (mesh[i] == mq_linspace_knot(i, step, m0)) ? 1 : 0;   // (more than 2049 knots: not with M <= 2048)
    if (!okm) ++nbad;
  }
  __syncthreads();
  int tile = 0;
  // ---- TIME-SERIES FRONT LOOP (sorted / time-series input; SURVEY 8d's secondary case, the order of the reference's own large 1-D data,
  // experiments/large_regression/electricity.py:31-32).  While every pair row of every wave of the tile lies in at most TWO cells, the
  // tile needs no sort: the lanes add their points into a wave-private set of 3k+1 run sums (lane-parallel, every lane busy: 4k
  // operations per point at full utilisation against one lane in ~3 in the owner loops), and when a wave's cell changes the run is
  // summed over the wave and added to the cell's slot of a small LDS table (open addressing, PS_TS_SLOTS cells per workgroup) that
  // the owners read once, before the epilogue.  The owner accumulators are not live in this loop - which is what makes room for the
  // run sums: in ONE loop with the sort phases hipcc kept per-tile state in scratch - and the loop is left for good, at a tile
  // boundary and by the whole workgroup, the first time a tile does not conform (unsorted input: at tile 0, for the price of the votes).
  if constexpr (TS == 1 && (ABL == 0 || ABL == 9)) {
    // does row 0 of every wave conform?  Asked as soon as the row's own loads have landed, before anything is prefetched: an unsorted
    // input leaves here for the price of a few votes and one barrier, with nothing fetched twice.
    {
      bool inside;
      const int u = unit_of(0, 0, inside);
      const bool v = inside && u < npair;
      const int ca = mq_guess(xs[0], m0, inv_delta, n_mesh), cb = mq_guess(xs[1], m0, inv_delta, n_mesh);   // (the guess: within a cell of the rule)
      const int c0 = __builtin_amdgcn_readfirstlane(ca);
      bool good = !__any(v) || __all(v);
      if (good && __any(v) && !__all(ca == c0 && cb == c0)) {
        const unsigned long long d0 = __ballot(ca != c0), d1 = __ballot(cb != c0);
        const int c1 = d0 ? __builtin_amdgcn_readlane(ca, (int)__builtin_ctzll(d0)) : __builtin_amdgcn_readlane(cb, (int)__builtin_ctzll(d1 | (1ull << 63)));
        // (a sorted row that crosses a knot may show a third, adjacent cell in the arithmetic guess: allow a spread of two)
        good = __all((ca - c0 <= 2 && c0 - ca <= 2) || (ca - c1 <= 2 && c1 - ca <= 2)) && __all((cb - c0 <= 2 && c0 - cb <= 2) || (cb - c1 <= 2 && c1 - cb <= 2));
      }
      if (lane == 0 && !good) ts_ctl[1] = 1u;
      ps_lds_barrier();
    }
    if (ts_ctl[1] == 0u) {
      double RS[NS], RT[K + 1];
#pragma unroll
      for (int p = 0; p < NS; ++p) RS[p] = 0.0;
#pragma unroll
      for (int p = 0; p <= K; ++p) RT[p] = 0.0;
      int run_c = -1;
      unsigned run_n = 0;
      // hand the open run to its cell's slot (lane 0 finds or claims the slot; the sums are wave totals, reduce-scatter butterflies:
      // with the chunked mapping a wave changes cell about once per tile).  A slot is always free: the verdict of every tile keeps
      // 16 TP + 16 entries in reserve, the most one tile and the closing hand-overs can claim.
      auto ts_flush = [&]() __attribute__((always_inline)) {
        int h = 0;
        if (lane == 0) {
          h = (int)(((unsigned)run_c * 40503u) & (unsigned)(PS_TS_SLOTS - 1));
          for (int probe = 0; probe < PS_TS_SLOTS; ++probe) {
            int expect = -1;
            const bool won = __hip_atomic_compare_exchange_strong(tskey + h, &expect, run_c, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (won) { const unsigned pos = __hip_atomic_fetch_add(ts_ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); tslist[pos] = h; }
            if (won || expect == run_c) break;
            h = (h + 1) & (PS_TS_SLOTS - 1);
          }
        }
        h = __builtin_amdgcn_readfirstlane(h);
        double* sl = tstab + (size_t)h * TSW;
        if constexpr (NS <= 8 && TP <= 4) {                       // (the butterflies' 28 temporaries fit beside two tiles of 4 points)
          int idx;
          const double t = ps_reduce_scatter8<NS>(RS, lane, idx);
          if (lane < 8 && idx < NS) lds_add(sl + idx, t);
          __builtin_amdgcn_sched_barrier(0);
          const double t2 = ps_reduce_scatter8<K + 1>(RT, lane, idx);
          if (lane < 8 && idx <= K) lds_add(sl + NS + idx, t2);
        } else {
#pragma unroll
          for (int q = 0; q < NS; ++q) { const double t = wave_sum_dpp(RS[q]); if (lane == 0) lds_add(sl + q, t); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
          for (int q = 0; q <= K; ++q) { const double t = wave_sum_dpp(RT[q]); if (lane == 0) lds_add(sl + NS + q, t); __builtin_amdgcn_sched_barrier(0); }
        }
        if (lane == 0) lds_add(sl + NSTAT, (double)run_n);
#pragma unroll
        for (int p = 0; p < NS; ++p) RS[p] = 0.0;
#pragma unroll
        for (int p = 0; p <= K; ++p) RT[p] = 0.0;
        run_c = -1;
        run_n = 0;
      };
      // one tile per iteration, its points in (xs, ys); the NEXT tile's loads go into a second register set (xb, yb) before anything
      // else - a whole tile ahead of their use -, so (xs, ys) are intact should the wave have to stop at this tile.  No barrier: a wave
      // that cannot take its rows (or finds the table short of its reserve) raises the flag and stops; the others see the flag at their
      // next tile and stop where THEY are; the sort loop then starts at the earliest of those tiles and every wave skips the rows it
      // has already summed.
      double xb[TP], yb[TP];
      while (t_done < n_tiles) {
        stamp(-1);
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        load_into(t_done + 1, xb, yb);                            // (clamped: past the end it re-reads the last pair)
        unsigned tile_bad = 0;
        search_from(t_done, xs, ys, tile_bad);
        stamp(0);                                                 // (front loop stamps: 0 loads landed + search, 1 votes, 5 sums)
        // verdict of this wave: every row either empty (beyond the data) or complete and inside at most two cells
        int clo[TP / 2], chi[TP / 2];
        bool good = true;
#pragma unroll
        for (int q2 = 0; q2 < TP / 2; ++q2) {
          const bool va = (valm >> (2 * q2)) & 1u, vb = (valm >> (2 * q2 + 1)) & 1u;
          const int c0 = __builtin_amdgcn_readfirstlane(cr[2 * q2]);
          clo[q2] = c0; chi[q2] = c0;
          if (!__any(va || vb)) { clo[q2] = -1; chi[q2] = -1; continue; }
          if (__all(va && vb && cr[2 * q2] == c0 && cr[2 * q2 + 1] == c0)) continue;
          if (!good) continue;                                     // (wave-uniform) the verdict is in
          // not one cell: the cell of lane 0 and the first cell that differs from it - are those all?
          const unsigned long long d0 = __ballot(cr[2 * q2] != c0), d1 = __ballot(cr[2 * q2 + 1] != c0);
          const int c1 = d0 ? __builtin_amdgcn_readlane(cr[2 * q2], (int)__builtin_ctzll(d0))
                            : __builtin_amdgcn_readlane(cr[2 * q2 + 1], (int)__builtin_ctzll(d1 | (1ull << 63)));
          clo[q2] = c0 < c1 ? c0 : c1; chi[q2] = c0 < c1 ? c1 : c0;
          good = __all(va && vb && (cr[2 * q2] == c0 || cr[2 * q2] == c1) && (cr[2 * q2 + 1] == c0 || cr[2 * q2 + 1] == c1));
        }
        if (__any(tile_bad != 0u)) good = false;                  // (a point outside the mesh: the sort loop reports it)
        const unsigned stop = __builtin_amdgcn_readfirstlane((int)(ts_ctl[1] | (ts_ctl[0] > (unsigned)(PS_TS_SLOTS - 16 * TP - 16) ? 1u : 0u)));
        if (!good || stop != 0u) { if (lane == 0) ts_ctl[1] = 1u; break; }
        stamp(1);
        // the sums: one rolled loop over the rows (one copy of the code), a second segment only for a row that crosses a knot
        for (int q2 = 0; q2 < TP / 2; ++q2) {
          int lo = clo[0], hi = chi[0];
          double s0 = sv[0], s1 = sv[1], y0 = ys[0], y1 = ys[1];
          int c0r = cr[0], c1r = cr[1];
#pragma unroll
          for (int r = 1; r < TP / 2; ++r)
            if (q2 == r) { lo = clo[r]; hi = chi[r]; s0 = sv[2 * r]; s1 = sv[2 * r + 1]; y0 = ys[2 * r]; y1 = ys[2 * r + 1]; c0r = cr[2 * r]; c1r = cr[2 * r + 1]; }
          if (lo < 0) continue;                                    // an empty row
          yy = fma(y0, y0, yy);                                   // (a row that is not empty is complete)
          yy = fma(y1, y1, yy);
          if (run_c != lo) { if (run_c >= 0) ts_flush(); run_c = lo; }
          if (lo == hi) {
            run_n += 128u;
            ps_acc<K>(s0, y0, RS, RT);
            __builtin_amdgcn_sched_barrier(0);
            ps_acc<K>(s1, y1, RS, RT);
          } else {   // the row crosses a knot: the other cell's points count as the point (0, 0), which adds nothing
            const bool a0 = c0r == lo, a1 = c1r == lo;
            run_n += (unsigned)__popcll(__ballot(a0)) + (unsigned)__popcll(__ballot(a1));
            ps_acc<K>(a0 ? s0 : 0.0, a0 ? y0 : 0.0, RS, RT);
            __builtin_amdgcn_sched_barrier(0);
            ps_acc<K>(a1 ? s1 : 0.0, a1 ? y1 : 0.0, RS, RT);
            __builtin_amdgcn_sched_barrier(0);
            ts_flush();
            run_c = hi;
            run_n = (unsigned)__popcll(__ballot(!a0)) + (unsigned)__popcll(__ballot(!a1));
            ps_acc<K>(a0 ? 0.0 : s0, a0 ? 0.0 : y0, RS, RT);
            __builtin_amdgcn_sched_barrier(0);
            ps_acc<K>(a1 ? 0.0 : s1, a1 ? 0.0 : y1, RS, RT);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        stamp(5);
        ++t_done;
#pragma unroll
        for (int q = 0; q < TP; ++q) { xs[q] = xb[q]; ys[q] = yb[q]; }   // (waits for the prefetched tile: it had the whole iteration to land)
      }
      if (run_c >= 0) ts_flush();                                 // (the table is read behind the barriers that end the sort loop)
      // where does the sort loop start?  At the earliest tile a wave stopped at.
      if (lane == 0) wtot[48 + wv] = (unsigned)t_done;
      __syncthreads();
      {
        unsigned m = wtot[48 + (lane & 15)];
        m = 0x7fffffffu - ps_wave_max_u32(0x7fffffffu - m);
        tile = (int)m;
      }
      if (tile < n_tiles) load_tile(tile);                        // (only an input that stops being a time series gets here: one exposed round trip)
    }
  }
  for (; tile < n_tiles; ++tile) {
    unsigned* cntb = cnt + (tile & 1) * PS_NCELL;
    stamp(-1);
    // the thread index is re-read per tile behind an opaque barrier: LDS addresses derived from it are then recomputed where they are
    // used (one or two instructions) instead of being hoisted out of the tile loop and spilled - a reload from scratch counts in
    // vmcnt and would wait for the prefetched tile
    tid = tid0;
    asm volatile("" : "+v"(tid));
    lane = tid & 63;
    wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- P1: cell, centred coordinate, rank inside the cell
    search_from(tile, xs, ys, nbad);
    if constexpr (TS == 1) { if (tile < t_done) valm = 0; }       // (wave-uniform) rows this wave has summed in the front loop
    {
      if constexpr (ABL == 9) { __builtin_amdgcn_sched_barrier(0); ph[5] += __builtin_amdgcn_s_memtime() - tprev; __builtin_amdgcn_sched_barrier(0); }   // loads landed, cells known
      if constexpr (ABL == 1) {
#pragma unroll
        for (int q = 0; q < TP; ++q) yy += sv[q] + (double)cr[q] + yv[q];
      } else {
        unsigned rk[TP];
#pragma unroll
        for (int q2 = 0; q2 < TP / 2; ++q2) {                     // all rank atomics of the tile in flight together
          const bool va = (valm >> (2 * q2)) & 1u, vb = (valm >> (2 * q2 + 1)) & 1u;
          const int c0 = __builtin_amdgcn_readfirstlane(cr[2 * q2]);
          if (__all(va && vb && cr[2 * q2] == c0 && cr[2 * q2 + 1] == c0)) {
            // all 128 points of the wave's pair row in ONE cell - one atomic instead of 128 same-address ones
            unsigned base = 0;
            if (lane == 0) base = __hip_atomic_fetch_add(cntb + c0, 128u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            rk[2 * q2] = base + 2u * (unsigned)lane;
            rk[2 * q2 + 1] = base + 2u * (unsigned)lane + 1u;
          } else {
            rk[2 * q2] = va ? __hip_atomic_fetch_add(cntb + cr[2 * q2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
            rk[2 * q2 + 1] = vb ? __hip_atomic_fetch_add(cntb + cr[2 * q2 + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
          }
          yy = va ? fma(yv[2 * q2], yv[2 * q2], yy) : yy;
          yy = vb ? fma(yv[2 * q2 + 1], yv[2 * q2 + 1], yy) : yy;
        }
#pragma unroll
        for (int q = 0; q < TP; ++q) cr[q] = ((valm >> q) & 1u) ? ((cr[q] << 13) | (int)rk[q]) : -1;
      }
    }
    if constexpr (ABL == 1) { load_tile(tile + 1); continue; }
    if constexpr (ABL == 9) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ph[6] += __builtin_amdgcn_s_memtime() - tprev; }   // ranks returned
    if constexpr (PF == 0 && (ABL == 0 || ABL == 9 || ABL == 5)) load_tile(tile + 1);   // the next tile's loads fly under scan, scatter and owners
    ps_lds_barrier();
    stamp(0);
    if constexpr (ABL == 2) {
      load_tile(tile + 1);
      cntb[tid] = 0; cntb[tid + PS_THREADS] = 0;
#pragma unroll
      for (int q = 0; q < TP; ++q) yy += (double)(cr[q] & 8191) + sv[q];
      ps_lds_barrier();
      continue;
    }
    // ---- P2: exclusive scan of the counts -> off   (thread t scans cells 2t, 2t+1)
    {
      const uint2 c2 = reinterpret_cast<const uint2*>(cntb)[tid];
      const unsigned v = c2.x + c2.y;
      const unsigned inc = ps_wave_scan_incl(v);
      if (lane == 63) wtot[wv] = inc;
      if (tid == 0) *nheavy_p = 0;
      ps_lds_barrier();
      unsigned w = wtot[lane & 15];
      w = ps_dpp_add_u32<0x111, 0xf>(w);
      w = ps_dpp_add_u32<0x112, 0xf>(w);
      w = ps_dpp_add_u32<0x114, 0xf>(w);
      w = ps_dpp_add_u32<0x118, 0xf>(w);                          // lane i < 16: wtot[0] + .. + wtot[i]
      const unsigned basew = (wv == 0) ? 0u : (unsigned)__builtin_amdgcn_readlane((int)w, wv > 0 ? wv - 1 : 0);
      const unsigned ex = basew + inc - v;
      reinterpret_cast<uint2*>(off)[tid] = make_uint2(ex, ex + c2.x);
      if (c2.x > PS_HEAVY || c2.y > PS_HEAVY) {                   // rare: list the heavy cells, flag them for their owners
        if (c2.x > PS_HEAVY) {
          const unsigned hs = __hip_atomic_fetch_add(nheavy_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          hlist[3 * hs] = 2 * tid; hlist[3 * hs + 1] = c2.x; hlist[3 * hs + 2] = ex;
          cntb[2 * tid] = 0x80000000u | hs;
        }
        if (c2.y > PS_HEAVY) {
          const unsigned hs = __hip_atomic_fetch_add(nheavy_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          hlist[3 * hs] = 2 * tid + 1; hlist[3 * hs + 1] = c2.y; hlist[3 * hs + 2] = ex + c2.x;
          cntb[2 * tid + 1] = 0x80000000u | hs;
        }
      }
    }
    ps_lds_barrier();
    stamp(1);
    if constexpr (ABL == 3) {
      load_tile(tile + 1);
      cntb[tid] = 0; cntb[tid + PS_THREADS] = 0;
#pragma unroll
      for (int q = 0; q < TP; ++q) yy += (double)(cr[q] & 8191) + sv[q] + (double)off[(cr[q] >> 13) & (PS_NCELL - 1)];
      ps_lds_barrier();
      continue;
    }
    // ---- P3: (s, y) into cell order
    {
      unsigned pos[TP];
#pragma unroll
      for (int q = 0; q < TP; ++q) pos[q] = off[(cr[q] >> 13) & (PS_NCELL - 1)] + (unsigned)(cr[q] & 8191);
#pragma unroll
      for (int q = 0; q < TP; ++q)
        if (cr[q] >= 0) buf[pos[q]] = make_double2(sv[q], yv[q]);
    }
    ps_lds_barrier();
    stamp(2);
    if constexpr (ABL == 4) {
      load_tile(tile + 1);
      cntb[tid] = 0; cntb[tid + PS_THREADS] = 0;
      const double2 p = buf[tid];
      yy += p.x + p.y;
      ps_lds_barrier();
      continue;
    }
    // ---- P4: owners accumulate their two cells' moments (one cell after the other: half the loop temporaries of a fused loop)
    {
      unsigned nA = cntb[tid], nB = cntb[tid + PS_THREADS];
      const unsigned oA = off[tid], oB = off[tid + PS_THREADS];
      const unsigned nheavy = *nheavy_p;
      cntb[tid] = 0; cntb[tid + PS_THREADS] = 0;                  // (owner-exclusive; this buffer is next used two tiles on)
      if (nheavy > 0) {                                           // (workgroup-uniform) sorted / clustered input
        const bool hvA = (nA & 0x80000000u) != 0, hvB = (nB & 0x80000000u) != 0;
        const unsigned slA = nA & 0x7fffffffu, slB = nB & 0x7fffffffu;
        if (hvA) { n0A += hlist[3 * slA + 1]; nA = 0; }
        if (hvB) { n0B += hlist[3 * slB + 1]; nB = 0; }
        for (unsigned r0 = 0; r0 < nheavy; r0 += PS_HROUND) {
          if (tid < PS_HROUND * NSTAT) hacc[tid] = 0.0;
          ps_lds_barrier();
          const unsigned rn = nheavy - r0 < (unsigned)PS_HROUND ? nheavy - r0 : (unsigned)PS_HROUND;
          // waves per cell: all 16 on a single heavy cell (sorted input: one or two cells hold the whole tile), 8 each on two, else groups
          // of four waves (group g takes the cells hh = g mod 4); a wave takes slices first, first + per, ...
          const int per = rn == 1 ? 16 : (rn == 2 ? 8 : 4), sh_per = rn == 1 ? 4 : (rn == 2 ? 3 : 2);
          for (unsigned hh = 0; hh < rn; ++hh) {
            const unsigned hn = hlist[3 * (r0 + hh) + 1], ho = hlist[3 * (r0 + hh) + 2];
            const bool mine = rn <= 2 ? (wv >> sh_per) == (int)hh : (wv >> 2) == (int)(hh & 3u);
            if (mine && (unsigned)(wv & (per - 1)) < ((hn + 63u) >> 6))
              ps_heavy_slices<K>(buf, hn, ho, wv & (per - 1), per, lane, hacc + hh * NSTAT);
          }
          ps_lds_barrier();
          if (hvA && slA >= r0 && slA < r0 + PS_HROUND) {
#pragma unroll
            for (int p = 0; p < NS; ++p) SA[p] += hacc[(slA - r0) * NSTAT + p];
#pragma unroll
            for (int p = 0; p <= K; ++p) TA[p] += hacc[(slA - r0) * NSTAT + NS + p];
          }
          if (hvB && slB >= r0 && slB < r0 + PS_HROUND) {
#pragma unroll
            for (int p = 0; p < NS; ++p) SB[p] += hacc[(slB - r0) * NSTAT + p];
#pragma unroll
            for (int p = 0; p <= K; ++p) TB[p] += hacc[(slB - r0) * NSTAT + NS + p];
          }
          if (r0 + PS_HROUND < nheavy) ps_lds_barrier();
        }
      }
      n0A += nA; n0B += nB;
      if constexpr (ABL == 9) { __builtin_amdgcn_sched_barrier(0); ph[7] += __builtin_amdgcn_s_memtime() - tprev; __builtin_amdgcn_sched_barrier(0); }   // heavy-cell pass
      // ---- the next tile's loads fly under the owner loops (issued behind the heavy-cell pass: its temporaries and the 4 TP prefetch
      // registers do not fit the register file together)
      if constexpr (PF == 1) load_tile(tile + 1);                 // (clamped: the last one re-reads the final pair)
      ps_own_cell<K, T>(buf, nA, oA, SA, TA);
      ps_own_cell<K, T>(buf, nB, oB, SB, TB);
    }
    stamp(3);
  }
  stamp(-1);

  // ---- the runs of the front loop go to their cells' owners
  if constexpr (TS == 1 && (ABL == 0 || ABL == 9)) {
    __syncthreads();
    const unsigned nts = ts_ctl[0];
    for (unsigned e = 0; e < nts; ++e) {
      const int h = tslist[e];
      const int c = tskey[h];
      const double* sl = tstab + (size_t)h * TSW;
      if (c == tid) {
#pragma unroll
        for (int p = 0; p < NS; ++p) SA[p] += sl[p];
#pragma unroll
        for (int p = 0; p <= K; ++p) TA[p] += sl[NS + p];
        n0A += (unsigned)sl[NSTAT];
      } else if (c == tid + PS_THREADS) {
#pragma unroll
        for (int p = 0; p < NS; ++p) SB[p] += sl[p];
#pragma unroll
        for (int p = 0; p <= K; ++p) TB[p] += sl[NS + p];
        n0B += (unsigned)sl[NSTAT];
      }
    }
  }
  // ---- epilogue: moments -> band / rhs entries of this workgroup (the LDS image aliases the sort buffers)
  double tot = block_sum(yy, scratch);                            // (its barriers also end the last owner phase)
  const double badf = block_sum((double)nbad, scratch + 32);
  __syncthreads();
  double* out = a.partials + (size_t)blockIdx.x * ((size_t)(K + 2) * M + 1);
  // The columns this workgroup has anything for: cells [clo, chi] with points -> columns [clo, chi + K].  A time series leaves a
  // workgroup with a handful of cells: it then writes (and the reduce reads) a few hundred bytes instead of the whole 112 KB partial.
  int col_lo = 0, col_hi = M - 1;
  if (a.ranges) {
    const unsigned mn = n0A ? (unsigned)tid : (n0B ? (unsigned)(tid + PS_THREADS) : 0x7fffffu);
    const unsigned mx = n0B ? (unsigned)(tid + PS_THREADS) + 1u : (n0A ? (unsigned)tid + 1u : 0u);   // (+1: 0 = no cell)
    const unsigned wmn = ps_wave_min_u32(mn), wmx = ps_wave_max_u32(mx);
    if (lane == 0) { wtot[wv] = wmn; wtot[16 + wv] = wmx; }
    __syncthreads();
    const unsigned bmn = ps_wave_min_u32(wtot[lane & 15]), bmx = ps_wave_max_u32(wtot[16 + (lane & 15)]);
    col_lo = bmx ? (int)bmn : 1;
    col_hi = bmx ? (int)bmx - 1 + K : 0;
    if (col_hi > M - 1) col_hi = M - 1;
    if (tid == 0) { a.ranges[2 * blockIdx.x] = col_lo; a.ranges[2 * blockIdx.x + 1] = col_hi; }
    __syncthreads();
  }
  if constexpr (ABL == 5) {   // diagnostic: keep the moments alive, skip the conversion
    double keep = (double)(n0A + n0B);
#pragma unroll
    for (int p = 0; p < NS; ++p) keep += SA[p] + SB[p];
#pragma unroll
    for (int p = 0; p <= K; ++p) keep += TA[p] + TB[p];
    out[tid] = keep;
  } else if constexpr (ABL == 0 || ABL == 9) {
    // Rows of cell c are c .. c+K (row = c + K - i for piece i).  Every thread turns the moments of its OWN cell into the cell's
    // (K+1)(K+2)/2 band contributions Q_ij = sum_p pair[i][j][p] S_p (i <= j: sub-diagonal d = j - i, column c + K - j) straight from
    // its registers, and into the K+1 rhs contributions R_i = sum_p single[i][p] T_p; the mirror symmetry of the pieces,
    // v_i(s) = v_{K-i}(-s), makes Q_{K-j,K-i} / R_{K-i} the same sums with the odd powers negated, so a mirror pair costs one set of
    // multiplies.  The contributions go through plane-major LDS images (conflict-free both ways); column col then adds
    //   band[d][col] = sum_{j=d..K} Q_{j-d,j}(col - K + j),     Phi y[col] = sum_i R_i(col - K + i).
    // Two halves of 1024 columns (an image holds the half's cells plus the K cells below it); the Q planes of a half in one round
    // for K <= 4, in two rounds (d < 2, d >= 2) above, so that a round's planes fit the LDS.
    constexpr int IW = PS_THREADS + K;
    double* img = lds;
    auto q_planes = [&](const double (&S)[NS], unsigned n0, int slot, int d0, int d1) __attribute__((always_inline)) {
      const double s0v = (double)n0;
      int pid = 0;
      if (!__any(n0 != 0u)) {                                      // (a time series: most waves of a workgroup own no cell with points)
#pragma unroll
        for (int d = d0; d < d1; ++d)
#pragma unroll
          for (int i = 0; i + d <= K; ++i) img[(pid++) * IW + slot] = 0.0;
        return;
      }
#pragma unroll
      for (int d = d0; d < d1; ++d) {
#pragma unroll
        for (int i = 0; i + d <= K; ++i) {
          const int j = i + d, mi = K - j, mj = K - i;             // (mi, mj): the mirror pair, same sub-diagonal
          if (mi < i) continue;                                    // written together with its mirror
          double e = MomCoef<K>::tab.pair[i][j][0] * s0v, o = 0.0;
#pragma unroll
          for (int p = 2; p <= NS; p += 2) e = fma(MomCoef<K>::tab.pair[i][j][p], S[p - 1], e);
          if (mi != i) {
#pragma unroll
            for (int p = 1; p <= NS; p += 2) o = fma(MomCoef<K>::tab.pair[i][j][p], S[p - 1], o);
            img[(pid + mi) * IW + slot] = e - o;
          }
          img[(pid + i) * IW + slot] = e + o;
        }
        pid += K + 1 - d;
      }
    };
    auto r_planes = [&](const double (&Tm)[K + 1], int slot, int plane0) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; 2 * i <= K; ++i) {
        double e = 0.0, o = 0.0;
#pragma unroll
        for (int p = 0; p <= K; p += 2) e = fma(MomCoef<K>::tab.single[i][p], Tm[p], e);
        if (2 * i != K) {
#pragma unroll
          for (int p = 1; p <= K; p += 2) o = fma(MomCoef<K>::tab.single[i][p], Tm[p], o);
          img[(plane0 + K - i) * IW + slot] = e - o;
        }
        img[(plane0 + i) * IW + slot] = e + o;
      }
    };
    constexpr int DSPLIT = (K <= 4) ? K + 1 : 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int j = half * PS_THREADS + tid;                       // this thread's column (= its cell) in the half
#pragma unroll
      for (int rnd = 0; rnd < (DSPLIT <= K ? 2 : 1); ++rnd) {
        const int d0 = rnd == 0 ? 0 : DSPLIT, d1 = rnd == 0 ? DSPLIT : K + 1;
        if (half == 0) q_planes(SA, n0A, K + tid, d0, d1); else q_planes(SB, n0B, K + tid, d0, d1);
        if (tid >= PS_THREADS - K) {                               // the K cells below the half: none (zeros), or the top cells of half A
          const int hs = tid - (PS_THREADS - K);
          if (half == 0) {
            int np = 0;
#pragma unroll
            for (int d = d0; d < d1; ++d) np += K + 1 - d;
            for (int pl = 0; pl < np; ++pl) img[pl * IW + hs] = 0.0;
          } else {
            q_planes(SA, n0A, hs, d0, d1);
          }
        }
        ps_lds_barrier();                                       // (LDS only: the image is complete)
        if (j < M && j >= col_lo && j <= col_hi) {
          int pid = 0;
#pragma unroll
          for (int d = d0; d < d1; ++d) {
            double v = 0.0;
#pragma unroll
            for (int jj = d; jj <= K; ++jj) v += img[(pid + jj - d) * IW + tid + jj];
            __builtin_nontemporal_store((j + d < M) ? v : 0.0, out + (size_t)d * M + j);   // written once, read once by the reduce
            pid += K + 1 - d;
          }
        }
        ps_lds_barrier();                                       // (LDS only: __syncthreads would also wait for the round's partial stores to be acknowledged before the next round computes)
      }
    }
    // rhs: both halves in one round (2 (K+1) planes)
    r_planes(TA, K + tid, 0);
    r_planes(TB, K + tid, K + 1);
    if (tid >= PS_THREADS - K) {
      const int hs = tid - (PS_THREADS - K);
#pragma unroll
      for (int i = 0; i <= K; ++i) img[i * IW + hs] = 0.0;
      r_planes(TA, hs, K + 1);
    }
    ps_lds_barrier();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int j = half * PS_THREADS + tid;
      if (j < M && j >= col_lo && j <= col_hi) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i <= K; ++i) r += img[(half * (K + 1) + i) * IW + tid + i];
        __builtin_nontemporal_store(r, out + (size_t)(K + 1) * M + j);
      }
    }
    if (tid == 0) out[(size_t)(K + 2) * M] = (badf > 0.0) ? __builtin_nan("") : tot;   // a point outside the mesh: loud (NaN y^T y)
  } else {
    if (tid == 0) out[0] = tot + badf;
  }
  stamp(4);
  if constexpr (ABL == 9) {
    if (tid == (int)(a.stamps_wave * 64) && a.stamps)
      for (int i = 0; i < 8; ++i) a.stamps[(size_t)blockIdx.x * 8 + i] = ph[i];
  }
}

}  // namespace asvgp
