// Block cyclic reduction on the matrix cores: the P chain of the bound (bandwidth 4, one right-hand side) with every 4 x 4 block
// product on v_mfma_f64_4x4x4f64.
//
// Replaces (reference call sites gpr.py:72-75 and the TF gradients through them): banded.cholesky_band(P), solve_triang_mat,
// inverse_from_cholesky_band(L_P) - the same odd-even (nested dissection) elimination as bcr.hpp: log|P|, x = P^-1 b and the band of
// P^-1, identical up to fp64 rounding.  What changes is who does the arithmetic.  bcr.hpp gives a node to ONE thread (270 fp64
// instructions forward, ~900 backward, and a lone wave issues one fp64 instruction per ~10 cycles) or to 16 lanes that trade
// entries through ds_bpermute (~300 LDS-crossbar round trips per node): 4-17 K cycles per level, 18 dependent levels, 157 K cycles.
// Here a node is 16 lanes of a wave in the instruction's own operand layout - v_mfma_f64_4x4x4f64 holds FOUR independent 4 x 4
// problems, block q in lanes 16 r + 4 q + c (probed: tools/micro/mfma_f64_4x4x4_layout.hip):
//     A operand: lane 16 k + 4 q + i = A[i][k],   B operand: lane 16 k + 4 q + j = B[k][j],   result: lane 16 i + 4 q + j = D[i][j]
// so with every matrix X kept "natural" (lane (r, c) holds X[r][c]) one instruction computes  C + X^T Y  for four nodes, and no
// product ever needs a cross-lane move: transposes are chosen when an operand is LOADED (any lane can read entry (c, r) instead of
// (r, c)) or by computing the transposed product directly ((X Y)^T = Y^T X^T).  The only per-lane scalar work is the Cholesky of
// the node's 4 x 4 pivot block, which every lane repeats for itself (45 instructions), and ONE forward substitution that yields the
// lane's entry of L^-1 (lower triangular: lane (r, c) needs L^-1[max][min], its mirror entry is zero).
//   forward, node i (a = i - h, b = i + h), per lane ~110 instructions, 12 of them MFMA:
//     Ua = L^-1 A[i,a], Ub = L^-1 A[i,b], z = L^-1 y_i;   D_a -= Ua^T Ua, A'[b,a] = -Ub^T Ua, y_a -= Ua^T z  | barrier |
//     D_b -= Ub^T Ub, y_b -= Ub^T z;   record: Ga^T = Ua^T L^-1, Gb^T = Ub^T L^-1, D^-1 = L^-T L^-1, w = L^-T z
//   backward, node i, ~50 instructions, 8 MFMA:
//     -Ca^T = S_aa Ga^T + S_ab Gb^T,  -Cb^T = S_ba Ga^T + S_bb Gb^T,  S_ii = D^-1 + (-Ca^T)^T Ga^T + (-Cb^T)^T Gb^T,
//     x_i = w - Ga x_a - Gb x_b                                          (S_ia = Ca, S_ib = Cb: the selected inverse)
// 1024 threads = 16 waves x 4 nodes per round; the records live in an L2-resident workspace (one 128-byte line per matrix and node).
#pragma once
#include "bcr.hpp"
#include "prior_plan.hpp"

namespace asvgp {

constexpr int BM_THREADS = 1024;
constexpr int BM_B = 4;
// record of a node (doubles): G_a^T, G_b^T, D^-1, Sigma_ii, C_a^T, C_b^T (16 each), w (4), diag(L) (4)
constexpr int BM_GAT = 0, BM_GBT = 16, BM_DINV = 32, BM_SD = 48, BM_CAT = 64, BM_CBT = 80, BM_W = 96, BM_DG = 100, BM_REC = 104;

__host__ __device__ inline size_t bcr_mfma_ws_doubles(long nb) { return (size_t)nb * BM_REC + 64; }
// LDS: D and E images of the even nodes (slot = node / 2), the rhs / solution vector, scratch
// (+ the level-0 band slabs and the forward records of the nodes eliminated at levels >= 3; the backward records of the levels >= 2
// reuse the D / E images, which are dead after the root)
__host__ __device__ inline size_t bcr_mfma_lds_doubles(long nb) {
  return (size_t)((nb + 1) / 2 + 1) * 32 + (size_t)nb * 4 + 96 + (size_t)(BM_THREADS / 64) * 4 * 40 + (size_t)(nb / 8 + 2) * 52;
}

__device__ __forceinline__ double bm_mfma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// Cholesky of the symmetric 4 x 4 block d (lower part used) in registers; returns this lane's entry t = L^-1[hi][lo]
// (hi = max(r, c), lo = min(r, c)), the diagonal product of L in dprod, and flags a non-positive pivot.
__device__ __forceinline__ double bm_chol_linv(const double (&d)[10], int hi, int lo, double& dprod, int& badj) {
  // d: 00 10 11 20 21 22 30 31 32 33
  double l00, i0, l10, l20, l30, l11, i1, l21, l31, l22, i2, l32, l33, i3;
  Num<double>::sqrt_inv(d[0], l00, i0);
  badj = !(d[0] > 0.0) ? 1 : 0;
  l10 = d[1] * i0; l20 = d[3] * i0; l30 = d[6] * i0;
  double s = fma(-l10, l10, d[2]);
  badj = (!(s > 0.0) && !badj) ? 2 : badj;
  Num<double>::sqrt_inv(s, l11, i1);
  l21 = fma(-l20, l10, d[4]) * i1;
  l31 = fma(-l30, l10, d[7]) * i1;
  s = fma(-l21, l21, fma(-l20, l20, d[5]));
  badj = (!(s > 0.0) && !badj) ? 3 : badj;
  Num<double>::sqrt_inv(s, l22, i2);
  l32 = fma(-l31, l21, fma(-l30, l20, d[8])) * i2;
  s = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d[9])));
  badj = (!(s > 0.0) && !badj) ? 4 : badj;
  Num<double>::sqrt_inv(s, l33, i3);
  dprod = (l00 * l11) * (l22 * l33);
  // column lo of L^-1: L z = e_lo
  const double z0 = (lo == 0) ? i0 : 0.0;
  const double z1 = fma(-l10, z0, (lo == 1) ? 1.0 : 0.0) * i1;
  const double z2 = fma(-l21, z1, fma(-l20, z0, (lo == 2) ? 1.0 : 0.0)) * i2;
  const double z3 = fma(-l32, z2, fma(-l31, z1, fma(-l30, z0, (lo == 3) ? 1.0 : 0.0))) * i3;
  return hi == 0 ? z0 : (hi == 1 ? z1 : (hi == 2 ? z2 : z3));
}

// Band source access by (diagonal, column): sources that know the matrix structure (elbo.hip's BandSumToep: Kuu constant per diagonal on
// its Toeplitz interior) provide load_dc; the plain ones are read through load(offset).
template <typename Src> __device__ __forceinline__ auto bm_src_load(const Src& A, int dd, long col, long M, int) -> decltype(A.load_dc(dd, col, M)) {
  return A.load_dc(dd, col, M);
}
template <typename Src> __device__ __forceinline__ double bm_src_load(const Src& A, int dd, long col, long M, long) {
  return A.load((long)dd * M + col, true);
}
// running log of a product without a logarithm on the dependent chain: mantissa in [1, 2) and a separate exponent sum
struct BmLog {
  double m; int e;
  __device__ __forceinline__ void mul(double p) {
    m *= p;
    const int hi = __double2hiint(m);
    e += ((hi >> 20) & 0x7ff) - 1023;
    m = __hiloint2double((hi & 0x800fffff) | 0x3ff00000, __double2loint(m));
  }
};

// One-trip messages between workgroups in different XCDs: n <= 62 payload words and ONE check word = tag ^ (xor of the payload bits), all
// agent-scope atomic accesses, written in any order with no drain and no separate flag; the reader (one wavefront) polls the WHOLE message
// and accepts it when the check word matches - a message an earlier launch left in the box carries another tag and never does.  (A flag
// behind drained stores costs the writer a vmcnt drain and the reader a second round trip for the payload: ~1.4 us per hand-over.)
__device__ __forceinline__ unsigned long long bm_wave_xor(unsigned long long x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x ^= (unsigned long long)__shfl_xor((long long)x, off, 64);
  return x;
}
// called by all 64 lanes of ONE wave; lane < n carries payload word v
__device__ __forceinline__ void bm_msg_send(double* box, int n, double v, unsigned long long tag, int lane) {
  unsigned long long* b = reinterpret_cast<unsigned long long*>(box);
  const unsigned long long w = lane < n ? (unsigned long long)__double_as_longlong(v) : 0ull;
  const unsigned long long x = bm_wave_xor(w) ^ tag;
  if (lane < n) __hip_atomic_store(b + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane == n) __hip_atomic_store(b + n, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// called by all 64 lanes of ONE wave; returns false when nothing valid arrived within the spin limit; lane < n receives word `lane`
__device__ __forceinline__ bool bm_msg_recv(const double* box, int n, double& v, unsigned long long tag, int lane, long spin_limit) {
  const unsigned long long* b = reinterpret_cast<const unsigned long long*>(box);
  for (long spins = 0; spins <= spin_limit; ++spins) {
    const unsigned long long w = lane <= n ? __hip_atomic_load(b + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const unsigned long long x = bm_wave_xor(lane < n ? w : 0ull) ^ tag;
    const unsigned long long chk = (unsigned long long)__shfl((long long)w, n, 64);
    if (x == chk) { v = __longlong_as_double((long long)w); return true; }
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// TWO workgroups on one matrix (the fused ELBO launch at nb >= 128).  The last level's node ht = 2^(levels - 1) separates the elimination
// tree into the nodes below it (left) and above it (right), which meet nowhere else: workgroup `half` = 0 eliminates the left nodes of
// the levels below the top, workgroup 1 the right ones - half the rounds on the wide, throughput-bound levels 0..2 -, each keeping its
// own share of the separator's pivot block and right-hand side (left: minus the updates from its side, starting from zero; right: the
// band's block minus the updates from its side).  Right hands its share over (22 doubles, with its log-det part and first bad column),
// left adds it, eliminates the separator and the root, runs the top backward level and hands the separator's backward record and
// solution back (52 doubles); both then walk their halves down.  Same arithmetic as one workgroup, node by node.  Payloads and flags
// are agent-scope accesses (no fences: the two workgroups sit in different XCDs); the waits are bounded (*gave_up, then return).
struct BmSplit {
  int half = -1;                 // -1: the whole matrix in this workgroup
  double* xchg = nullptr;        // BM_XCHG doubles: three message boxes (right -> left 21 words at 0, its log-det part 3 words at 24, left -> right 53 at 32)
  long spin_limit = 0;
  int* gave_up = nullptr;        // LDS word of the caller
  unsigned long long tag = 0;    // this launch's message tag (sequence number << 8; the boxes add their own low byte)
};
constexpr int BM_XCHG = 96;

// The whole solve for one matrix, called by all BM_THREADS threads of one workgroup.
//   A: lower band (5, M) source (BandPtr<double> or BandSumP);  rhs: (M) with stride rhs_stride;  ws: bcr_mfma_ws_doubles;
//   lds: bcr_mfma_lds_doubles.  Out: S lower band of A^-1 (5, M), x = A^-1 rhs, logdet[0], info (first bad column + 1).
template <typename Src>
__device__ __attribute__((always_inline)) void bcr_mfma_solve(Src A, const double* rhs, int M, double* ws, double* lds, double* Sband, double* x,
                                                              double* logdet, int* info, int rhs_stride = 1, double* stamps = nullptr,
                                                              BmSplit sp = BmSplit{}) {
  constexpr int B = BM_B;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane >> 4, q = (lane >> 2) & 3, c = lane & 3;
  const int e = r * 4 + c, et = c * 4 + r;                        // entry (r, c) and its transpose in a row-major 4 x 4
  const int hi = r > c ? r : c, lo = r > c ? c : r;
  const int nb = (M + B - 1) / B;
  const int nslot = (nb + 1) / 2;
  double* Dl = lds;                                               // [slot][16]
  double* El = lds + (size_t)(nslot + 1) * 16;                    // [slot][16]   E(n)[r][c] = A[(n + h) B + r, n B + c]
  double* xs = El + (size_t)(nslot + 1) * 16;                     // rhs -> z / w -> x, per row
  double* red = xs + (size_t)nb * B;                              // 96 doubles scratch
  double* slab = red + 96 + (size_t)(wv * 4 + q) * 40;            // level 0: the band slab of this lane's node pair (wave-private)
  double* Fl = red + 96 + (size_t)(BM_THREADS / 64) * 4 * 40;     // forward records (G_a^T, G_b^T, D^-1, w) of nodes i = 8 j (levels >= 3): [j][52]
  double* Sl = lds;                                               // backward records (Sigma_ii, C_a^T, C_b^T) of nodes i = 4 j (levels >= 2): [j][48]
  int levels = 0;
  while ((1 << levels) < nb) ++levels;
  const bool split = sp.half >= 0;                                // (the caller splits only trees of >= 7 levels)
  const int top = levels - 1, ht = 1 << (top > 0 ? top : 0);      // the separator: the last level's only node
  double s_other = 0.0;                                           // left: the right workgroup's log-det part and first bad column
  int bad_other = 0x7fffffff;
  // members of the level l for this workgroup: nodes i = h + (m0 + m) 2 h, m < ne_h
  auto level_share = [&](int l, int ne, int& m0, int& ne_h) {
    m0 = 0; ne_h = ne;
    if (!split) return;
    if (l < top) {
      const int mL = ht >> (l + 1);
      if (sp.half == 0) ne_h = ne < mL ? ne : mL;
      else { m0 = mL; ne_h = ne > mL ? ne - mL : 0; }
    } else {
      ne_h = sp.half == 0 ? ne : 0;
    }
  };
  int bad = 0;
  BmLog ld{1.0, 0};
  unsigned long long t_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  int nst = 0;
  auto stamp = [&]() {
    if (stamps && tid == 0) { const unsigned long long t = __builtin_amdgcn_s_memtime(); stamps[nst++] = (double)(t - t_prev); t_prev = t; }
  };
  auto rec = [&](int node) -> double* { return ws + (size_t)node * BM_REC; };
  // band entry helpers (identity padding beyond M, as bcr.hpp)
  auto bandD = [&](int n, int rr, int cc) -> double {             // D_n[rr][cc], rr >= cc
    const int col = n * B + cc, row = n * B + rr;
    const bool pad = row >= M;
    const double v = bm_src_load(A, rr - cc, pad ? 0 : col, (long)M, 0);
    return pad ? ((rr == cc) ? 1.0 : 0.0) : v;
  };
  auto bandE = [&](int n, int rr, int cc) -> double {             // A[(n+1) B + rr, n B + cc]  (upper-triangular block)
    if (rr > cc) return 0.0;
    const int col = n * B + cc, row = (n + 1) * B + rr;
    const bool pad = row >= M;
    const double v = bm_src_load(A, B + rr - cc, pad ? 0 : col, (long)M, 0);
    return pad ? 0.0 : v;
  };
  // ---- pre-pass: rhs -> xs.  (The even nodes' D blocks reach the LDS in level 0, from the band slab of their odd neighbour; only a last
  // even node without one - nb odd - is fetched here.)
  if ((nb & 1) && tid < 16 && !(split && sp.half == 0)) {
    const int rr = tid >> 2, cc = tid & 3;
    Dl[(size_t)((nb - 1) >> 1) * 16 + tid] = bandD(nb - 1, rr > cc ? rr : cc, rr > cc ? cc : rr);
  }
  if (split && sp.half == 0 && tid < 16) Dl[(size_t)(ht >> 1) * 16 + tid] = 0.0;   // left's share of the separator: only the updates
  for (int row = tid; row < nb * B; row += BM_THREADS) {
    const bool sep_left = split && sp.half == 0 && row >= ht * B && row < (ht + 1) * B;
    xs[row] = (row < M && !sep_left) ? rhs[(long)row * rhs_stride] : 0.0;
  }
  __syncthreads();
  stamp();

  constexpr int RMAXR = 4;                                        // rounds of 64 nodes per level (nb <= 512)
  // ---------------- forward elimination ----------------
  for (int l = 0; l < levels; ++l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    int m0, ne_h;
    level_share(l, ne, m0, ne_h);
    if (split && l == top) {
      // ---- right -> left: the separator's pivot block and right-hand side as the right half leaves them, its log-det part, its bad column
      if (sp.half == 1) {
        if (wv == 0) {
          const double v = lane < 16 ? Dl[(size_t)(ht >> 1) * 16 + lane] : (lane < 20 ? xs[ht * B + lane - 16] : 0.0);
          bm_msg_send(sp.xchg, 20, v, sp.tag | 1ull, lane);
        }
        // behind the message, off the left workgroup's critical path (it reads them at its very end): log-det part, bad column
        const double mine = (r == 0 && c == 0) ? (log(ld.m) + (double)ld.e * 0.6931471805599453094) : 0.0;
        const double tot = wave_sum_dpp(mine);
        int bm = bad ? bad : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(bm, off, 64); bm = o < bm ? o : bm; }
        int* sbad = reinterpret_cast<int*>(red + 32);
        if (lane == 0) { red[wv] = tot; sbad[wv] = bm; }
        __syncthreads();
        if (wv == 0) {
          double sv = 0.0;
          int bmin = 0x7fffffff;
          for (int w2 = 0; w2 < BM_THREADS / 64; ++w2) { sv += red[w2]; bmin = sbad[w2] < bmin ? sbad[w2] : bmin; }
          bm_msg_send(sp.xchg + 24, 2, lane == 0 ? sv : (double)bmin, sp.tag | 2ull, lane);
        }
      } else {
        if (wv == 0) {
          double v = 0.0;
          if (!bm_msg_recv(sp.xchg, 20, v, sp.tag | 1ull, lane, sp.spin_limit)) { if (lane == 0) *sp.gave_up = 1; }
          else {
            if (lane < 16) Dl[(size_t)(ht >> 1) * 16 + lane] += v;
            else if (lane < 20) xs[ht * B + lane - 16] += v;
          }
        }
        __syncthreads();
        if (*sp.gave_up) return;
      }
      stamp();
    }
    double updb[RMAXR], ybu[RMAXR];
    int qv = q, ev = e;                                           // (opaque per level: the slab addresses are then recomputed here instead of being
    asm volatile("" : "+v"(qv), "+v"(ev));                        //  hoisted out of the level loop as 64-bit pointers and spilled)
    // level 0: the band slabs of two rounds are requested together (3 entries per lane and round: 5 diagonals x 8 columns of the node
    // pair (a, i) - 64-byte pieces, D_a, E(a), D_i, E(i) complete)
    double sl[2][3];
    auto slab_loads = [&](int rd0) __attribute__((always_inline)) {   // rounds rd0, rd0 + 1
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
          sl[u][t3] = 0.0;
          int m = (rd0 + u) * 64 + wv * 4 + qv;
          m = m0 + (m < ne_h ? m : ne_h - 1);
          const int kk = ev + 16 * t3, dd = kk >> 3, col = m * 2 * B + (kk & 7);
          const bool in = kk < 40 && col + dd < M;
          const double v = bm_src_load(A, in ? dd : 0, in ? col : 0, (long)M, 0);   // (unconditional, clamped)
          sl[u][t3] = in ? v : ((dd == 0) ? 1.0 : 0.0);
        }
      }
    };
#pragma unroll
    for (int rd = 0; rd < RMAXR; ++rd) {
      __builtin_amdgcn_sched_barrier(0);                          // (rounds one after the other: interleaved they do not fit 128 registers)
      if (l == 0 && (rd & 1) == 0) slab_loads(rd);                // two rounds' slabs per memory round trip
      updb[rd] = 0.0; ybu[rd] = 0.0;
      if (rd * 64 + wv * 4 >= ne_h) continue;                     // (wave-uniform) a wave without a node goes straight to the barrier:
                                                                  // on dummy data it would take three of four issue slots from the working wave of its SIMD
      const int ml = rd * 64 + wv * 4 + q;
      const bool act = ml < ne_h;
      // a slot without a node repeats the level's last node (unconditional loads, no per-value predication - as branches around every
      // load they cost more scalar instructions than the arithmetic) and keeps its results to itself
      const int i = h + (m0 + (act ? ml : ne_h - 1)) * 2 * h, a = i - h, b = i + h;
      const bool hasb = b < nb;
      const int bsafe = hasb ? b : a;
      // every lane of the node: the whole pivot block
      double d[10];
      double ea, eb, da;
      if (l == 0) {
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) if (e + 16 * t3 < 40) slab[e + 16 * t3] = sl[rd & 1][t3];   // (same wave reads it back: LDS keeps program order)
        int kk = 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int cc = 0; cc <= rr; ++cc) d[kk++] = slab[(rr - cc) * 8 + 4 + cc];
        ea = slab[(4 + lo - hi) * 8 + hi];                         // E(a)[r][c] = A[i B + r, a B + c] (r <= c)
        ea = (r <= c) ? ea : 0.0;
        eb = slab[(4 + lo - hi) * 8 + 4 + hi];                     // A[i,b] = E(i)^T: entry (c, r), c <= r
        eb = (hasb && c <= r) ? eb : 0.0;
        da = slab[(hi - lo) * 8 + lo];                             // D_a (symmetric image)
      } else {
        const double* Di = Dl + (size_t)(i >> 1) * 16;
        int kk = 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int cc = 0; cc <= rr; ++cc) d[kk++] = Di[rr * 4 + cc];
        ea = El[(size_t)(a >> 1) * 16 + e];
        eb = El[(size_t)(i >> 1) * 16 + et];
        eb = hasb ? eb : 0.0;
        da = Dl[(size_t)(a >> 1) * 16 + e];
      }
      double yv = xs[i * B + r], ya = xs[a * B + r];
      yv = (c == 0) ? yv : 0.0;
      ya = (c == 0) ? ya : 0.0;
      (void)bsafe;
      double dprod;
      int badj;
      const double t = bm_chol_linv(d, hi, lo, dprod, badj);
      if (act && badj && !bad) bad = i * B + badj;
      const double linv = (r >= c) ? t : 0.0;                      // natural L^-1
      const double linvT = (c >= r) ? t : 0.0;                     // natural L^-T
      if (act) ld.mul(dprod);
      const double ua = bm_mfma(linvT, ea, 0.0);                   // L^-1 A[i,a]
      const double ub = bm_mfma(linvT, eb, 0.0);
      const double zz = bm_mfma(linvT, yv, 0.0);                   // column 0: L^-1 y_i
      const double nua = -ua, nub = -ub;
      const double da_new = bm_mfma(nua, ua, da);                  // D_a - Ua^T Ua
      const double e_new = bm_mfma(nub, ua, 0.0);                  // A'[b,a] = -Ub^T Ua
      const double ya_new = bm_mfma(nua, zz, ya);
      updb[rd] = bm_mfma(ub, ub, 0.0);
      ybu[rd] = bm_mfma(ub, zz, 0.0);
      updb[rd] = act ? updb[rd] : 0.0;                             // (a repeated node must not subtract twice in phase B)
      ybu[rd] = act ? ybu[rd] : 0.0;
      const double gat = bm_mfma(ua, linv, 0.0);                   // Ua^T L^-1 = (L^-T Ua)^T
      const double gbt = bm_mfma(ub, linv, 0.0);
      const double dinv = bm_mfma(linv, linv, 0.0);                // L^-T L^-1
      const double w = bm_mfma(linv, zz, 0.0);                     // L^-T z
      if (act) {
        if (l >= 3) {                                              // narrow levels: the record stays on the CU
          double* R = Fl + (size_t)(i >> 3) * 52;
          R[e] = gat; R[16 + e] = gbt; R[32 + e] = dinv;
          if (c == 0) R[48 + r] = w;
        } else {
          double* R = rec(i);
          R[BM_GAT + e] = gat; R[BM_GBT + e] = gbt; R[BM_DINV + e] = dinv;
          if (c == 0) R[BM_W + r] = w;
        }
        Dl[(size_t)(a >> 1) * 16 + e] = da_new;                    // phase A: left neighbour
        El[(size_t)(a >> 1) * 16 + e] = e_new;
        if (c == 0) xs[a * B + r] = ya_new;
      }
    }
    bcr_lds_barrier();                                            // (LDS traffic only: the record stores are re-read by the same lanes, much later)
#pragma unroll
    for (int rd = 0; rd < RMAXR; ++rd) {                           // phase B: right neighbour
      if (rd * 64 + wv * 4 >= ne_h) continue;
      const int ml = rd * 64 + wv * 4 + q;
      const int i = h + (m0 + ml) * 2 * h, b = i + h;
      if (ml < ne_h && b < nb) {
        Dl[(size_t)(b >> 1) * 16 + e] -= updb[rd];
        if (c == 0) xs[b * B + r] -= ybu[rd];
      }
    }
    bcr_lds_barrier();
    stamp();
  }
  // ---------------- root (node 0) ----------------
  if (wv == 0 && !(split && sp.half == 1)) {
    const bool act = q == 0;
    double d[10];
    int kk = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int cc = 0; cc <= rr; ++cc) { const double v = Dl[rr * 4 + cc]; d[kk++] = act ? v : ((rr == cc) ? 1.0 : 0.0); }
    const double yv = (act && c == 0) ? xs[r] : 0.0;
    double dprod;
    int badj;
    const double t = bm_chol_linv(d, hi, lo, dprod, badj);
    if (act && badj && !bad) bad = badj;
    const double linv = (r >= c) ? t : 0.0, linvT = (c >= r) ? t : 0.0;
    if (act) ld.mul(dprod);
    const double s00 = bm_mfma(linv, linv, 0.0);
    const double zz = bm_mfma(linvT, yv, 0.0);
    const double x0 = bm_mfma(linv, zz, 0.0);
    if (act) {
      rec(0)[BM_SD + e] = s00;
      Sl[e] = s00;                                                // (aliases D_0, which this wave has consumed)
      if (r >= c && r < M) Sband[(long)(r - c) * M + c] = s00;
      if (c == 0) xs[r] = x0;
    }
  }
  __syncthreads();
  stamp();
  // ---------------- backward: solve + selected inverse ----------------
  unsigned long long lw = 0ull;
  for (int l = levels - 1; l >= 0; --l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    int m0, ne_h;
    level_share(l, ne, m0, ne_h);
    if (split && sp.half == 0 && wv == BM_THREADS / 64 - 1) {
      // the right workgroup's log-det part and bad column (sent ~10 us ago): a wave without nodes on the narrow levels requests the message
      // at the top of the backward pass and validates it a few levels later - at the very end the round trip would sit on the finisher's
      // critical path (not there yet / torn: a blocking receive, here, still off that path)
      if (l == top - 1 && lane <= 2)
        lw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(sp.xchg + 24) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (l == 2) {
        const unsigned long long x = bm_wave_xor(lane < 2 ? lw : 0ull) ^ (sp.tag | 2ull);
        const unsigned long long chk = (unsigned long long)__shfl((long long)lw, 2, 64);
        double v = __longlong_as_double((long long)lw);
        bool ok = x == chk;
        if (!ok) ok = bm_msg_recv(sp.xchg + 24, 2, v, sp.tag | 2ull, lane, sp.spin_limit);
        if (!ok) { if (lane == 0) *sp.gave_up = 1; }
        else if (lane < 2) red[48 + lane] = v;
      }
    }
#pragma unroll
    for (int rd = 0; rd < RMAXR; ++rd) {
      __builtin_amdgcn_sched_barrier(0);
      if (rd * 64 + wv * 4 >= ne_h) continue;
      const int ml = rd * 64 + wv * 4 + q;
      const bool act = ml < ne_h;
      const int i = h + (m0 + (act ? ml : ne_h - 1)) * 2 * h, a = i - h, b = i + h;   // (a slot without a node repeats the last node, as in the forward pass)
      const bool hasb = b < nb;
      const int bs = hasb ? b : a;
      const bool e_is_a = ((a / (2 * h)) & 1) != 0;               // which of a, b was eliminated at level l + 1
      // own forward record: LDS for the narrow levels; neighbours' backward records: LDS for l >= 1 (a, b are multiples of 4), global at level 0
      double gat, gbt, dinv, w, saa, sbb, sba, sab;
      if (l >= 3) {
        const double* R = Fl + (size_t)(i >> 3) * 52;
        gat = R[e]; gbt = R[16 + e]; dinv = R[32 + e]; w = R[48 + r];
      } else {
        const double* R = rec(i);
        gat = R[BM_GAT + e]; gbt = R[BM_GBT + e]; dinv = R[BM_DINV + e]; w = R[BM_W + r];
      }
      // Sigma_ba[r][c] = e_is_a ? C_b(a)[c][r] : C_a(b)[r][c];  stored transposed: CBT(a)[r][c] = C_b(a)[c][r], CAT(b)[c][r] = C_a(b)[r][c]
      if (l >= 1) {
        const double* Ra = Sl + (size_t)(a >> 2) * 48;
        const double* Rb = Sl + (size_t)(bs >> 2) * 48;
        saa = Ra[e];
        sbb = Rb[e];
        sba = e_is_a ? Ra[32 + e] : Rb[16 + et];
        sab = e_is_a ? Ra[32 + et] : Rb[16 + e];
      } else {
        const double* Ra = rec(a);
        const double* Rb = rec(bs);
        saa = Ra[BM_SD + e];
        sbb = Rb[BM_SD + e];
        sba = e_is_a ? Ra[BM_CBT + e] : Rb[BM_CAT + et];
        sab = e_is_a ? Ra[BM_CBT + et] : Rb[BM_CAT + e];
      }
      sbb = hasb ? sbb : 0.0; sba = hasb ? sba : 0.0; sab = hasb ? sab : 0.0;
      double xa = xs[a * B + r], xb = xs[bs * B + r];
      xa = (c == 0) ? xa : 0.0;
      xb = (hasb && c == 0) ? xb : 0.0;
      w = (c == 0) ? w : 0.0;
      double ncat = bm_mfma(saa, gat, 0.0);                        // S_aa Ga^T + S_ab Gb^T = -Ca^T
      ncat = bm_mfma(sba, gbt, ncat);
      double ncbt = bm_mfma(sab, gat, 0.0);                        // S_ba Ga^T + S_bb Gb^T = -Cb^T
      ncbt = bm_mfma(sbb, gbt, ncbt);
      double sii = bm_mfma(ncat, gat, dinv);                       // D^-1 - Ca Ga^T - Cb Gb^T
      sii = bm_mfma(ncbt, gbt, sii);
      double gx = bm_mfma(gat, xa, 0.0);                           // Ga x_a + Gb x_b
      gx = bm_mfma(gbt, xb, gx);
      if (act) {
        if (c == 0) xs[i * B + r] = w - gx;
        if (l >= 2) {                                              // read by every level below: LDS ...
          double* Rw = Sl + (size_t)(i >> 2) * 48;
          Rw[e] = sii; Rw[16 + e] = -ncat; Rw[32 + e] = -ncbt;
        }
        if (l > 0) {                                               // ... and the workspace, for level 0 (whose other neighbours are level-1 nodes)
          double* Rw = rec(i);
          Rw[BM_SD + e] = sii; Rw[BM_CAT + e] = -ncat; Rw[BM_CBT + e] = -ncbt;
        }
        if (r >= c && i * B + r < M) Sband[(long)(r - c) * M + i * B + c] = sii;
        if (l == 0) {
          // lane (r, c) holds Ca[c][r] and Cb[c][r]:  Sigma[i B + c, a B + r] (c <= r)  and  Sigma[b B + r, i B + c] (r <= c)
          if (c <= r && i * B + c < M) Sband[(long)(B + c - r) * M + a * B + r] = -ncat;
          if (hasb && r <= c && b * B + r < M) Sband[(long)(B + r - c) * M + i * B + c] = -ncbt;
        }
      }
    }
    if (l == 1) __syncthreads(); else bcr_lds_barrier();           // level 0 reads the workspace records of the levels above: drained here
    if (split && l == top) {
      // ---- left -> right: the separator's backward record (Sigma_ii, C_a^T, C_b^T) and solution
      double* Rs = Sl + (size_t)(ht >> 2) * 48;
      if (sp.half == 0) {
        if (wv == 0) bm_msg_send(sp.xchg + 32, 52, lane < 48 ? Rs[lane < 48 ? lane : 0] : (lane < 52 ? xs[ht * B + lane - 48] : 0.0), sp.tag | 3ull, lane);
      } else {
        if (wv == 0) {
          double v = 0.0;
          if (!bm_msg_recv(sp.xchg + 32, 52, v, sp.tag | 3ull, lane, sp.spin_limit)) { if (lane == 0) *sp.gave_up = 1; }
          else if (lane < 48) {
            Rs[lane] = v;
            rec(ht)[BM_SD + lane] = v;                            // (its level-0 neighbour reads the workspace record: this workgroup's own copy)
            if (lane < 16) {                                      // ... and this workgroup's traces read the separator's diagonal block of the band
              const int rr = lane >> 2, cc = lane & 3;
              if (rr >= cc && ht * B + rr < M) Sband[(long)(rr - cc) * M + ht * B + cc] = v;
            }
          } else if (lane < 52) {
            xs[ht * B + lane - 48] = v;
          }
        }
        __syncthreads();
        if (*sp.gave_up) return;
      }
    }
    stamp();
  }
  // ---------------- outputs ----------------
  if (!(split && sp.half == 0))
    for (int col = M - B + tid; col < M; col += BM_THREADS)
      if (col >= 0)
#pragma unroll
        for (int dd = 1; dd <= B; ++dd)
          if (col + dd >= M) Sband[(long)dd * M + col] = 0.0;
  {
    const int row0 = (split && sp.half == 1) ? (ht + 1) * B : 0, row1 = (split && sp.half == 0) ? (ht + 1) * B : M;
    for (int row = row0 + tid; row < row1 && row < M; row += BM_THREADS) x[(long)row * rhs_stride] = xs[row];
  }
  if (!(split && sp.half == 1)) {
    // log|A| = 2 sum log diag(L): every node's lanes carried the same product; lane (0, q, 0) of each wave speaks for its nodes
    const double mine = (r == 0 && c == 0) ? (log(ld.m) + (double)ld.e * 0.6931471805599453094) : 0.0;
    const double tot = wave_sum_dpp(mine);
    int bm = bad ? bad : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(bm, off, 64); bm = o < bm ? o : bm; }
    int* sbad = reinterpret_cast<int*>(red + 32);
    if (lane == 0) { red[wv] = tot; sbad[wv] = bm; }
    __syncthreads();
    if (split && tid == 0) { s_other = red[48]; bad_other = (int)red[49]; }   // (fetched during the backward pass)
    if (tid == 0) {
      double s = 0.0;
      int bmin = 0x7fffffff;
      for (int w2 = 0; w2 < BM_THREADS / 64; ++w2) { s += red[w2]; bmin = sbad[w2] < bmin ? sbad[w2] : bmin; }
      s += s_other;                                               // (the right workgroup's nodes)
      bmin = bad_other < bmin ? bad_other : bmin;
      // (agent-scope stores: the fused launch's last ticket may read them from another XCD, where a plain store would still be a dirty L2 line)
      __hip_atomic_store(logdet + 0, 2.0 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(logdet + 1, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(info, (bmin == 0x7fffffff) ? 0 : bmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  stamp();
}


// ------------------------------------------------------------------------------------------------------------
// Backward half for the PRIOR chain on the matrix cores (the counterpart of bcr_pre.hpp): band(Kuu^-1) and its d/d-lengthscale
// tangent from the host planner's per-class factor table, which carries G_a^T, G_b^T and D^-1 (value and tangent) ready-made.
// Same recurrences as bcr_mfma_solve's backward pass on dual numbers: a product X^T Y is one MFMA for the value and two for the
// tangent (dX^T Y + X^T dY).  1024 threads; the records of the levels >= 2 stay in the LDS (value and tangent planes).
//   tab: table of prior_plan_eval (device-visible);  node_rec: nb ints;  ws: 96 doubles per node;  lds: bcr_mfma_pre_lds_doubles.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t bcr_mfma_pre_ws_doubles(long nb) { return (size_t)nb * 96 + 64; }
__host__ __device__ inline size_t bcr_mfma_pre_lds_doubles(long nb, int n_rec) {
  return PRIOR_TAB_HEADER + (size_t)2 * n_rec * prior_rec_fields(BM_B) + (size_t)(nb / 4 + 2) * 96 + (size_t)(nb + 1) / 2 + 64;
}

struct BmDual { double v, d; };
__device__ __forceinline__ BmDual bm_dmfma(BmDual x, BmDual y, BmDual c) {   // c + x^T y
  BmDual r;
  r.v = bm_mfma(x.v, y.v, c.v);
  r.d = bm_mfma(x.d, y.v, c.d);
  r.d = bm_mfma(x.v, y.d, r.d);
  return r;
}

__device__ __attribute__((always_inline)) void bcr_mfma_backward_pre(const double* __restrict__ tab, int n_rec, const int* __restrict__ node_rec, int M,
                                                                      double* ws, double* lds, double* Sv, double* Sd, double* logdet, int* info,
                                                                      unsigned long long* done_flag, unsigned long long seq,
                                                                      const unsigned long long* ready_flag = nullptr, long spin_limit = 0, int* gave_up = nullptr) {
  constexpr int B = BM_B, W = prior_rec_fields(B);
  const int tid = threadIdx.x;
  if (ready_flag) {
    // The launch was issued BEFORE the host ran this theta's forward pass (elbo.hip, run_chains): wait until the host has published table
    // `seq` in this slot (pinned memory, one 8-byte system-scope load per poll).  Bounded: *gave_up (LDS) is set and the caller leaves.
    if (tid == 0) {
      long spins = 0;
      while (__hip_atomic_load(ready_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {   // (relaxed polls; the acquire fence follows the loop)
        __builtin_amdgcn_s_sleep(8);
        if (++spins > spin_limit) { *gave_up = 1; break; }
      }
    }
    __syncthreads();
    if (*gave_up) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane >> 4, q = (lane >> 2) & 3, c = lane & 3;
  const int e = r * 4 + c, et = c * 4 + r;
  const int nb = (M + B - 1) / B;
  const int n_tab = PRIOR_TAB_HEADER + 2 * n_rec * W;
  double* tl = lds;                                               // the factor table
  double* Sl = lds + n_tab;                                       // [node / 4][96]: Sigma_ii, C_a^T, C_b^T value | tangent
  int* nrl = reinterpret_cast<int*>(Sl + (size_t)(nb / 4 + 2) * 96);   // node -> record
  {
    const double2* src = reinterpret_cast<const double2*>(tab);
    double2* dst = reinterpret_cast<double2*>(tl);
    for (int i2 = tid; i2 < n_tab / 2; i2 += BM_THREADS) dst[i2] = src[i2];   // (16-B loads, all in flight: the source may sit behind PCIe)
    for (int i2 = tid; i2 < nb; i2 += BM_THREADS) nrl[i2] = node_rec[i2];
  }
  __syncthreads();
  if (done_flag && tid == 0) __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const double* tv = tl + PRIOR_TAB_HEADER;
  const double* td = tv + (size_t)n_rec * W;
  int levels = 0;
  while ((1 << levels) < nb) ++levels;
  // ---- root: Sigma_00 rides in the U_a slot of the last record
  if (tid < 16) {
    const int r0 = nrl[0];
    const double xv = tv[(size_t)r0 * W + prior_f_UA(B) + tid], xd = td[(size_t)r0 * W + prior_f_UA(B) + tid];
    Sl[tid] = xv; Sl[48 + tid] = xd;
    ws[tid] = xv; ws[48 + tid] = xd;
    const int rr = tid >> 2, cc = tid & 3;
    if (rr >= cc && rr < M) { Sv[(long)(rr - cc) * M + cc] = xv; Sd[(long)(rr - cc) * M + cc] = xd; }
  }
  __syncthreads();
  constexpr int RMAXR = 4;
  for (int l = levels - 1; l >= 0; --l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
#pragma unroll
    for (int rd = 0; rd < RMAXR; ++rd) {
      __builtin_amdgcn_sched_barrier(0);
      if (rd * 64 + wv * 4 >= ne) continue;                       // (wave-uniform)
      const int m = rd * 64 + wv * 4 + q;
      const bool act = m < ne;
      const int i = h + (act ? m : ne - 1) * 2 * h, a = i - h, b = i + h;   // (a slot without a node repeats the level's last node)
      const bool hasb = b < nb;
      const int bs = hasb ? b : a;
      const bool e_is_a = ((a / (2 * h)) & 1) != 0;
      const int ri = nrl[i];
      const double* fv = tv + (size_t)ri * W;
      const double* fd = td + (size_t)ri * W;
      const BmDual gat{fv[prior_f_GAT(B) + e], fd[prior_f_GAT(B) + e]};
      const BmDual gbt{fv[prior_f_GBT(B) + e], fd[prior_f_GBT(B) + e]};
      const BmDual dinv{fv[prior_f_DINV(B) + e], fd[prior_f_DINV(B) + e]};
      BmDual saa, sbb, sba, sab;
      if (l >= 1) {                                                // (two code paths: LDS and global pointers must not merge into flat accesses)
        const double* Ra = Sl + (size_t)(a >> 2) * 96;
        const double* Rb = Sl + (size_t)(bs >> 2) * 96;
        saa = BmDual{Ra[e], Ra[48 + e]};
        sbb = BmDual{Rb[e], Rb[48 + e]};
        sba = e_is_a ? BmDual{Ra[32 + e], Ra[80 + e]} : BmDual{Rb[16 + et], Rb[64 + et]};
        sab = e_is_a ? BmDual{Ra[32 + et], Ra[80 + et]} : BmDual{Rb[16 + e], Rb[64 + e]};
      } else {
        const double* Ga = ws + (size_t)a * 96;
        const double* Gb = ws + (size_t)bs * 96;
        saa = BmDual{Ga[e], Ga[48 + e]};
        sbb = BmDual{Gb[e], Gb[48 + e]};
        sba = e_is_a ? BmDual{Ga[32 + e], Ga[80 + e]} : BmDual{Gb[16 + et], Gb[64 + et]};
        sab = e_is_a ? BmDual{Ga[32 + et], Ga[80 + et]} : BmDual{Gb[16 + e], Gb[64 + e]};
      }
      if (!hasb) { sbb = BmDual{0.0, 0.0}; sba = sbb; sab = sbb; }
      const BmDual zero{0.0, 0.0};
      BmDual ncat = bm_dmfma(saa, gat, zero);                      // S_aa Ga^T + S_ab Gb^T = -Ca^T
      ncat = bm_dmfma(sba, gbt, ncat);
      BmDual ncbt = bm_dmfma(sab, gat, zero);                      // S_ba Ga^T + S_bb Gb^T = -Cb^T
      ncbt = bm_dmfma(sbb, gbt, ncbt);
      BmDual sii = bm_dmfma(ncat, gat, dinv);                      // D^-1 - Ca Ga^T - Cb Gb^T
      sii = bm_dmfma(ncbt, gbt, sii);
      if (act) {
        if (l >= 2) {
          double* Rw = Sl + (size_t)(i >> 2) * 96;
          Rw[e] = sii.v; Rw[16 + e] = -ncat.v; Rw[32 + e] = -ncbt.v;
          Rw[48 + e] = sii.d; Rw[64 + e] = -ncat.d; Rw[80 + e] = -ncbt.d;
        }
        if (l > 0) {
          double* Rw = ws + (size_t)i * 96;
          Rw[e] = sii.v; Rw[16 + e] = -ncat.v; Rw[32 + e] = -ncbt.v;
          Rw[48 + e] = sii.d; Rw[64 + e] = -ncat.d; Rw[80 + e] = -ncbt.d;
        }
        if (r >= c && i * B + r < M) { const long o = (long)(r - c) * M + i * B + c; Sv[o] = sii.v; Sd[o] = sii.d; }
        if (l == 0) {
          if (c <= r && i * B + c < M) { const long o = (long)(B + c - r) * M + a * B + r; Sv[o] = -ncat.v; Sd[o] = -ncat.d; }
          if (hasb && r <= c && b * B + r < M) { const long o = (long)(B + r - c) * M + i * B + c; Sv[o] = -ncbt.v; Sd[o] = -ncbt.d; }
        }
      }
    }
    if (l == 1) __syncthreads(); else bcr_lds_barrier();
  }
  for (int col = M - B + tid; col < M; col += BM_THREADS)
    if (col >= 0)
#pragma unroll
      for (int dd = 1; dd <= B; ++dd)
        if (col + dd >= M) { Sv[(long)dd * M + col] = 0.0; Sd[(long)dd * M + col] = 0.0; }
  if (tid == 0) {
    __hip_atomic_store(logdet + 0, tl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(logdet + 1, tl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(info, (int)tl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace asvgp
