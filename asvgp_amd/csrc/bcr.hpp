// Block cyclic reduction (BCR) for a symmetric positive definite band matrix with lower bandwidth B, viewed as
// block-tridiagonal with B x B blocks: log-determinant, solve and the band of the inverse ("selected inverse") in
// O(log M) dependent levels instead of the O(M) column recurrences of banded_matrices' cholesky_band /
// inverse_from_cholesky_band / solve_triang_mat (reference call sites gpr.py:56-75).  It is the same factorisation
// under a nested-dissection (odd-even) elimination order, so every quantity the bound needs - log|.|, band(.^-1),
// P^-1 b - is identical up to fp64 rounding; the Cholesky factor itself (operator API) still comes from the
// sequential sweep in band_sweeps.hpp.
//
// One workgroup (256 threads, one wave per SIMD so each lane may use the whole register file) per matrix:
//   forward level l (h = 2^l): every node i = h + m 2h is eliminated by one thread:
//       D_i = L L^T,  U_a = L^-1 A[i,a],  U_b = L^-1 A[i,b],  z = L^-1 y_i          (a = i-h, b = i+h)
//       D_a -= U_a^T U_a, y_a -= U_a^T z, A'[b,a] = -U_b^T U_a   | barrier |   D_b -= U_b^T U_b, y_b -= U_b^T z
//     survivors live in LDS (struct-of-arrays, conflict-free 8-B lanes); factors go to an L2-resident workspace.
//   backward level l: x_i = L^-T (z - U_a x_a - U_b x_b),  G = L^-T [U_a U_b],
//       S_ia = -(G_a S_aa + G_b S_ba),  S_ib = -(G_a S_ab + G_b S_bb),  S_ii = D_i^-1 - S_ia G_a^T - S_ib G_b^T
//   T = double, or Dual for the forward-mode tangent (d/d lengthscale of band(Kuu^-1)).
#pragma once
#include <type_traits>
#include "band_sweeps.hpp"
#include "bcr_lane.hpp"

namespace asvgp {

constexpr int BCR_THREADS = 256;

// barrier ordering LDS only: the forward pass' factor stores to the L2 workspace are re-read by the SAME thread in the
// backward pass, so they need not be drained (s_waitcnt vmcnt(0)) at every level like __syncthreads() would.
__device__ __forceinline__ void bcr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- small dense helpers (registers, fully unrolled) -------------------------------------------------------
template <typename T, int B>
__device__ __forceinline__ void blk_chol(T (&D)[B][B], T (&invd)[B], int& bad, int col0) {
  using N = Num<T>;
#pragma unroll
  for (int j = 0; j < B; ++j) {
    T s = D[j][j];
#pragma unroll
    for (int p = 0; p < j; ++p) s = N::nfma(D[j][p], D[j][p], s);
    if (!(N::val(s) > 0.0) && !bad) bad = col0 + j + 1;
    T ljj, inv;
    N::sqrt_inv(s, ljj, inv);
    D[j][j] = ljj;
    invd[j] = inv;
#pragma unroll
    for (int i = j + 1; i < B; ++i) {
      T t = D[i][j];
#pragma unroll
      for (int p = 0; p < j; ++p) t = N::nfma(D[i][p], D[j][p], t);
      D[i][j] = t * inv;
    }
  }
}
// X <- L^-1 X  (L lower, invd = 1/diag)
template <typename T, int B, int C>
__device__ __forceinline__ void blk_solve_L(const T (&L)[B][B], const T (&invd)[B], T (&X)[B][C]) {
  using N = Num<T>;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = 0; i < B; ++i) {
      T t = X[i][c];
#pragma unroll
      for (int p = 0; p < i; ++p) t = N::nfma(L[i][p], X[p][c], t);
      X[i][c] = t * invd[i];
    }
}
// X <- L^-T X
template <typename T, int B, int C>
__device__ __forceinline__ void blk_solve_LT(const T (&L)[B][B], const T (&invd)[B], T (&X)[B][C]) {
  using N = Num<T>;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = B - 1; i >= 0; --i) {
      T t = X[i][c];
#pragma unroll
      for (int p = i + 1; p < B; ++p) t = N::nfma(L[p][i], X[p][c], t);
      X[i][c] = t * invd[i];
    }
}

// ---- storage views ----------------------------------------------------------------------------------------
// LDS: struct-of-arrays with a COMPILE-TIME slot count NS, so (field * NS + slot) * 8 folds into the ds_read/ds_write
// immediate offset; Dual keeps value and tangent planes apart (every access 8 B / lane, conflict-free).
template <typename T, int NS> struct Soa;
// D fields (diagonal blocks) always live in the LDS; the E fields (couplings) do too, except in the BIG layout (twice
// the node count: M up to 4096 at k = 4) where they move to an L2-resident global plane and every barrier becomes a
// full __syncthreads().  Field indices are relative to the D / E plane.
template <int NS> struct Soa<double, NS> {
  double* p; double* pe;
  __device__ __forceinline__ double getD(int f, int s) const { return p[f * NS + s]; }
  __device__ __forceinline__ void setD(int f, int s, double v) const { p[f * NS + s] = v; }
  __device__ __forceinline__ double getE(int f, int s) const { return pe[f * NS + s]; }
  __device__ __forceinline__ void setE(int f, int s, double v) const { pe[f * NS + s] = v; }
};
template <int NS> struct Soa<Dual, NS> {
  double* p; double* q; double* pe; double* qe;  // q = tangent plane
  __device__ __forceinline__ Dual getD(int f, int s) const { return {p[f * NS + s], q[f * NS + s]}; }
  __device__ __forceinline__ void setD(int f, int s, Dual v) const { p[f * NS + s] = v.v; q[f * NS + s] = v.d; }
  __device__ __forceinline__ Dual getE(int f, int s) const { return {pe[f * NS + s], qe[f * NS + s]}; }
  __device__ __forceinline__ void setE(int f, int s, Dual v) const { pe[f * NS + s] = v.v; qe[f * NS + s] = v.d; }
};
// global workspace: struct-of-arrays over nodes with a compile-time node stride (coalesced across the lanes of a level,
// field offsets are constants added on the scalar unit); `Rec` is the view of one node.
template <typename T, int NN> struct Rec;
template <int NN> struct Rec<double, NN> {
  double* r;  // &plane0[node]
  __device__ __forceinline__ double get(int f) const { return r[(long)f * NN]; }
  __device__ __forceinline__ void set(int f, double v) const { r[(long)f * NN] = v; }
};
template <int NN> struct Rec<Dual, NN> {
  double* r; double* q;
  __device__ __forceinline__ Dual get(int f) const { return {r[(long)f * NN], q[(long)f * NN]}; }
  __device__ __forceinline__ void set(int f, Dual v) const { r[(long)f * NN] = v.v; q[(long)f * NN] = v.d; }
};


template <int B, int NRHS> struct BcrLayout {
  // forward LDS fields per survivor slot
  static constexpr int F_D = 0, F_E = B * B, F_N = 2 * B * B;   // (y / z / x live in a row-indexed LDS vector)
  // factor workspace fields per node
  static constexpr int W_L = 0, W_I = B * B, W_UA = B * B + B, W_UB = 2 * B * B + B,
                       W_SD = 3 * B * B + B,                 // Sigma_ii
                       W_CA = W_SD + B * B, W_CB = W_CA + B * B, W_N = W_CB + B * B;
};
template <typename T> __host__ __device__ constexpr int planes_of() { return sizeof(T) / sizeof(double); }

// slots held in LDS: the largest power of two whose survivor image (+ the rhs vector) fits in ~150 KB; the BIG layout
// doubles it by keeping only the D fields there.
template <typename T, int B, bool BIG = false> __host__ __device__ constexpr int bcr_ns() {
  int ns = 1024;
  while ((long)planes_of<T>() * 2 * B * B * ns * 8 + (long)2 * ns * B * 8 + 1024 > 150 * 1024) ns >>= 1;
  return BIG ? 2 * ns : ns;
}
template <typename T, int B, int NRHS, bool BIG = false>
__host__ __device__ inline size_t bcr_lds_doubles(long nb) {
  if ((nb + 1) / 2 > bcr_ns<T, B, BIG>()) return (size_t)1 << 40;  // does not fit: callers fall back
  const size_t fields = BIG ? (size_t)B * B : (size_t)BcrLayout<B, NRHS>::F_N;
  return (size_t)planes_of<T>() * fields * bcr_ns<T, B, BIG>() + (NRHS || !BIG ? (size_t)nb * B : 0) + 64;
}
template <typename T, int B, int NRHS, bool BIG = false>
__host__ __device__ inline size_t bcr_ws_doubles(long) {
  return (size_t)planes_of<T>() * BcrLayout<B, NRHS>::W_N * 2 * bcr_ns<T, B, BIG>() +
         (BIG ? (size_t)planes_of<T>() * B * B * bcr_ns<T, B, BIG>() : 0);   // + the global E plane
}
template <bool BIG> __device__ __forceinline__ void bcr_barrier() {
  if constexpr (BIG) __syncthreads(); else bcr_lds_barrier();
}

// block extraction from the lower band (B+1, M): D_n (lower part) and E(n) = A[n+1, n] (upper-triangular block)
// P = A / s + Kuu (gpr.py:72) formed on the fly: lets the data chain skip its
// elementwise prepare kernel (Kuu is already in the workspace from the prior chain).
struct BandSumP {
  const double* A; const double* Kuu; double inv_s;   // A * (1/s): within 1 ulp of the reference's A / s (gpr.py:72); 40 fp64
                                                      // divisions per node on the prepass / level-0 chain were ~4 % of the P chain
  __device__ __forceinline__ double load(long off, bool ok) const {
#pragma clang fp contract(off)
    const double t = A[off] * inv_s;
    return ok ? t + Kuu[off] : 0.0;
  }
  __device__ __forceinline__ void load8(long off, double (&o)[8]) const {
#pragma clang fp contract(off)
    const double2* p = reinterpret_cast<const double2*>(A + off);
    const double2* q2 = reinterpret_cast<const double2*>(Kuu + off);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 t = p[q], u = q2[q];
      const double a0 = t.x * inv_s, a1 = t.y * inv_s;   // (plain operators: they are under the pragma, inlined helpers are not)
      o[2 * q] = a0 + u.x;
      o[2 * q + 1] = a1 + u.y;
    }
  }
  __device__ __forceinline__ bool aligned16() const { return ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Kuu)) & 15) == 0; }
};

template <typename T, int B, typename Src>
__device__ __forceinline__ T band_D(const Src& A, int M, int n, int r, int c) {  // r >= c
  int col = n * B + c, row = n * B + r;
  const bool pad = row >= M;  // identity padding; the load itself is unconditional (clamped) so gathers batch up
  T v = A.load((long)(r - c) * M + (pad ? 0 : col), true);
  return pad ? ((r == c) ? Num<T>::make(1.0, 0.0) : Num<T>::zero()) : v;
}
template <typename T, int B, typename Src>
__device__ __forceinline__ T band_E(const Src& A, int M, int n, int r, int c) {  // A[(n+1)B + r, nB + c]
  if (r > c) return Num<T>::zero();   // compile-time after unrolling
  int col = n * B + c, row = (n + 1) * B + r;
  const bool pad = row >= M;
  T v = A.load((long)(B + r - c) * M + (pad ? 0 : col), true);
  return pad ? Num<T>::zero() : v;
}

// ------------------------------------------------------------------------------------------------------------
// The whole solve for one matrix.  Called by all BCR_THREADS threads of one workgroup.
//   A: lower band (B+1, M);  rhs: (M) or null (NRHS = 0);  ws: bcr_ws_doubles;  lds: bcr_lds_doubles
//   out: S lower band of A^-1 (B+1, M), x = A^-1 rhs (M), logdet (1), info (first bad column + 1)
// ------------------------------------------------------------------------------------------------------------
template <typename T, int B, int NRHS, typename Src = BandPtr<T>, bool BIG = false>
__device__ __attribute__((always_inline)) void bcr_solve(Src A, const double* rhs, int M, double* ws, double* lds, BandOut<T> S, double* x,
                          double* logdet, int* info, double* stamps = nullptr, int rhs_stride = 1) {
  using N = Num<T>;
  using Lay = BcrLayout<B, NRHS>;
  const int tid = threadIdx.x;
  const int nb = (M + B - 1) / B;
  constexpr int NS = bcr_ns<T, B, BIG>();
  constexpr int NN0 = 2 * NS;
  Soa<T, NS> F;                                                               // survivors: D | E
  F.p = lds;
  double* xs;                                                                 // y / z / x per row in LDS
  if constexpr (BIG) {
    double* eg = ws + (size_t)planes_of<T>() * Lay::W_N * NN0;                // global E plane(s) behind the factor records
    F.pe = eg;
    if constexpr (planes_of<T>() == 2) { F.q = lds + (size_t)B * B * NS; F.qe = eg + (size_t)B * B * NS; }
    xs = lds + (size_t)planes_of<T>() * B * B * NS;
  } else {
    F.pe = lds + (size_t)Lay::F_E * NS;
    if constexpr (planes_of<T>() == 2) { F.q = lds + (size_t)Lay::F_N * NS; F.qe = F.q + (size_t)Lay::F_E * NS; }
    xs = lds + (size_t)planes_of<T>() * Lay::F_N * NS;
  }
  double* red = xs + ((NRHS || !BIG) ? (size_t)nb * B : 0);                   // 64 doubles scratch
  constexpr int NN = 2 * NS;                                                  // node stride of the workspace arrays
  auto Wn = [&](int node) -> Rec<T, NN> {
    Rec<T, NN> rec;
    rec.r = ws + node;
    if constexpr (planes_of<T>() == 2) rec.q = ws + (size_t)Lay::W_N * NN + node;
    return rec;
  };
  int bad = 0;
  double ld_acc = 0.0, dld_acc = 0.0;
  int levels = 0;
  while ((1 << levels) < nb) ++levels;
  // lane-distributed mode for the narrow levels (bcr_lane.hpp): group of GS lanes per node, entry (r, c) per lane
  constexpr int GS = GroupSize<B>::v;
  constexpr int NG = BCR_THREADS / GS;
  constexpr int BB = B * B;
  constexpr int LANE_MAX_NODES = NG;       // one round per level in lane mode (a second round costs more than the thread-per-node form: 12.2K vs 8.6K cycles forward, 15.9K vs 12K backward at 32 nodes)
  const int grp = tid / GS, e = tid % GS;
  const bool lane_on = e < BB;
  const int r = lane_on ? e / B : 0, c = lane_on ? e % B : 0;
  int nst = 0;
  unsigned long long t_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  auto stamp = [&]() {  // diagnostic: cycles since the previous stamp (thread 0)
    if (stamps && tid == 0) {
      unsigned long long t = __builtin_amdgcn_s_memtime();
      stamps[nst++] = (double)(t - t_prev);
      t_prev = t;
    }
  };

  // FAST0 (k = 4, M a multiple of 8, 16-B aligned bands, level 0 in one-thread-per-node form, one round): thread m reads the
  // band slab of its node pair (2m, 2m+1) - 5 diagonals x 64 contiguous bytes, 16-B loads, coalesced across the wave - inside
  // level 0 and writes the updated D_2m straight to the LDS; the pre-pass gather (8-B loads at a 64-B stride) is skipped.
  // (One round only: a second round's D_2m would not be in the LDS yet when the first round's phase B updates it.)
  const int ne0 = (nb > 1) ? nb / 2 : 0;
  const bool fast0 = (B == 4) && ((M & 7) == 0) && A.aligned16() && (ne0 > LANE_MAX_NODES) && (ne0 <= BCR_THREADS);
  // pre-pass: even nodes -> LDS slots (D lower part + mirrored upper); all gathers issued before the first LDS store
  for (int n = 2 * tid; n < nb && !fast0; n += 2 * BCR_THREADS) {
    const int s = n >> 1;
    T tmp[B * (B + 1) / 2];
    int e = 0;
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) tmp[e++] = band_D<T, B, Src>(A, M, n, r, c);
    e = 0;
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        F.setD(r * B + c, s, tmp[e]);
        if (c != r) F.setD(c * B + r, s, tmp[e]);
        ++e;
      }
  }
  // NRHS <= 1: y lives in xs[] indexed by row and is updated in place (y -> z -> x)
  if (NRHS) {   // all loads of the copy in flight together (the serial form paid one L2/HBM round trip per 256 rows: ~16K cycles)
    constexpr int RMAX = (2 * NS * B + BCR_THREADS - 1) / BCR_THREADS;
    double rv[RMAX];
#pragma unroll
    for (int q = 0; q < RMAX; ++q) { const int r = tid + q * BCR_THREADS; rv[q] = (r < M) ? rhs[(long)r * rhs_stride] : 0.0; }
#pragma unroll
    for (int q = 0; q < RMAX; ++q) { const int r = tid + q * BCR_THREADS; if (r < nb * B) xs[r] = rv[q]; }
  }
  __syncthreads();

  // ---------------- forward elimination ----------------
  for (int l = 0; l < levels; ++l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    if (ne <= LANE_MAX_NODES) {
      for (int m0 = 0; m0 < ne; m0 += NG) {
      const int m = m0 + grp;
      const bool act = m < ne;
      const int i = h + m * 2 * h, a = i - h, b = i + h;
      const bool hasb = act && (b < nb);
      T d = N::zero(), ua = N::zero(), ub = N::zero();
      T invd[B];
      double z = 0.0;
      T upd_b = N::zero();
      double yb_upd = 0.0;
      if (act) {
        if (lane_on) {
          if (l == 0) {
            const int rr = r >= c ? r : c, cc = r >= c ? c : r;
            d = band_D<T, B, Src>(A, M, i, rr, cc);
            ua = band_E<T, B, Src>(A, M, a, r, c);
            ub = hasb ? band_E<T, B, Src>(A, M, i, c, r) : N::zero();   // A[i,b] = E(i)^T
          } else {
            d = F.getD(e, i >> 1);
            ua = F.getE(e, a >> 1);
            ub = hasb ? F.getE(c * B + r, i >> 1) : N::zero();
          }
          if (NRHS && c == 0) z = xs[i * B + r];
        }
        // --- Cholesky of D_i inside the group (lower part of d becomes L)
#pragma unroll
        for (int j = 0; j < B; ++j) {
          T pj = gshfl<T>(d, j * B + j, GS);
          if (!(N::val(pj) > 0.0) && !bad) bad = i * B + j + 1;
          T ljj, inv;
          N::sqrt_inv(pj, ljj, inv);
          invd[j] = inv;
          if (c == j) d = (r == j) ? ljj : ((r > j) ? d * inv : d);
          T lrj = gshfl<T>(d, r * B + j, GS), lcj = gshfl<T>(d, c * B + j, GS);
          if (c > j && r >= c) d = N::nfma(lrj, lcj, d);
        }
        // --- U_a = L^-1 A[i,a], U_b = L^-1 A[i,b], z = L^-1 y  (row rr finalised at step rr)
#pragma unroll
        for (int rr = 0; rr < B; ++rr) {
#pragma unroll
          for (int p = 0; p < rr; ++p) {
            T lv = gshfl<T>(d, rr * B + p, GS);
            T uap = gshfl<T>(ua, p * B + c, GS), ubp = gshfl<T>(ub, p * B + c, GS);
            double zp = NRHS ? __shfl(z, p * B, GS) : 0.0;
            if (r == rr) {
              ua = N::nfma(lv, uap, ua);
              ub = N::nfma(lv, ubp, ub);
              z = fma(-N::val(lv), zp, z);
            }
          }
          if (r == rr) { ua = ua * invd[rr]; ub = ub * invd[rr]; z = z * N::val(invd[rr]); }
        }
        // --- products
        T upd_a = N::zero(), enew = N::zero();
        double ya_upd = 0.0;
#pragma unroll
        for (int p = 0; p < B; ++p) {
          T uar = gshfl<T>(ua, p * B + r, GS), uac = gshfl<T>(ua, p * B + c, GS);
          T ubr = gshfl<T>(ub, p * B + r, GS), ubc = gshfl<T>(ub, p * B + c, GS);
          upd_a = upd_a + uar * uac;
          upd_b = upd_b + ubr * ubc;
          enew = N::nfma(ubr, uac, enew);
          if (NRHS) {
            double zp = __shfl(z, p * B, GS);
            ya_upd = fma(N::val(uar), zp, ya_upd);   // valid on lanes c == 0 (uar = ua[p][r])
            yb_upd = fma(N::val(ubr), zp, yb_upd);
          }
        }
        if (lane_on) {
          // factors -> workspace record (128 contiguous bytes per matrix per group)
          Wn(i).set(Lay::W_L + e, (r >= c) ? d : N::zero());
          Wn(i).set(Lay::W_UA + e, ua);
          Wn(i).set(Lay::W_UB + e, ub);
          if (e < B) {
            T iv = invd[0];
#pragma unroll
            for (int q = 1; q < B; ++q) iv = (e == q) ? invd[q] : iv;
            Wn(i).set(Lay::W_I + e, iv);
          }
          if (NRHS && c == 0) xs[i * B + r] = z;
          // phase A: left neighbour
          const int sa = a >> 1;
          F.setD(e, sa, F.getD(e, sa) - upd_a);
          F.setE(e, sa, enew);
          if (NRHS && c == 0) xs[a * B + r] -= ya_upd;
        }
      }
      bcr_barrier<BIG>();
      if (hasb && lane_on) {  // phase B: right neighbour
        const int sb = b >> 1;
        F.setD(e, sb, F.getD(e, sb) - upd_b);
        if (NRHS && c == 0) xs[b * B + r] -= yb_upd;
      }
      bcr_barrier<BIG>();
      }
      stamp();
      continue;
    }
    // level 0 and the deeper levels get separate copies of the one-thread-per-node body (own register allocation, no band
    // gather code in the deeper levels)
    auto fwd_thread = [&](auto is0) {
      constexpr bool IS0 = decltype(is0)::value;
      for (int m0 = 0; m0 < ne; m0 += BCR_THREADS) {   // more eliminated nodes than threads: several rounds
        const int m = m0 + tid;
        const bool act = m < ne;
        const int i = h + m * 2 * h, a = i - h, b = i + h;
        const bool hasb = act && (b < nb);
        T Ua[B][B], Ub[B][B], D[B][B], Da[B][B], invd[B];
        double z[B];
        if (act) {
          // load D_i, A[i,a] (= E(a)), A[b,i]^T (= E(i)^T), y_i
#pragma unroll
          for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c < B; ++c) {
              if constexpr (IS0) {
                if (!fast0) {
                  D[r][c] = (c <= r) ? band_D<T, B, Src>(A, M, i, r, c) : N::zero();
                  Ua[r][c] = band_E<T, B, Src>(A, M, a, r, c);
                  Ub[r][c] = hasb ? band_E<T, B, Src>(A, M, i, c, r) : N::zero();  // transpose: A[i,b] = A[b,i]^T
                }
              } else {
                D[r][c] = (c <= r) ? F.getD(r * B + c, i >> 1) : N::zero();
                Ua[r][c] = F.getE(r * B + c, a >> 1);
                Ub[r][c] = hasb ? F.getE(c * B + r, i >> 1) : N::zero();
              }
            }
          if constexpr (IS0 && B == 4) {
            if (fast0) {   // slab[d][0..3] = node a = 2m, slab[d][4..7] = node i = 2m + 1  (band[d][8m .. 8m+7])
              T slab[B + 1][2 * B];
#pragma unroll
              for (int d = 0; d <= B; ++d) {
                T row8[8];   // (fixed size keeps the other bandwidths' instantiations well-formed; fast0 implies B == 4)
                A.load8((long)d * M + (long)a * B, row8);
#pragma unroll
                for (int q = 0; q < 2 * B && q < 8; ++q) slab[d][q] = row8[q];
              }
#pragma unroll
              for (int r = 0; r < B; ++r)
#pragma unroll
                for (int c = 0; c < B; ++c) {
                  D[r][c] = (c <= r) ? slab[r - c][B + c] : N::zero();                       // D_i
                  Da[r][c] = (c <= r) ? slab[r - c][c] : N::zero();                          // D_a (goes to the LDS in phase A)
                  Ua[r][c] = (r <= c) ? slab[B + r - c][c] : N::zero();                      // E(a)[r][c] = A[iB + r, aB + c]
                  Ub[r][c] = (hasb && c <= r) ? slab[B + c - r][B + r] : N::zero();          // E(i)^T[r][c] = A[bB + c, iB + r]
                }
            }
          }
          blk_chol<T, B>(D, invd, bad, i * B);
          __builtin_amdgcn_sched_barrier(0);   // (phase fences: keep the scheduler from overlapping the phases' live ranges - Dual blocks spill otherwise)
          blk_solve_L<T, B, B>(D, invd, Ua);
          __builtin_amdgcn_sched_barrier(0);
          blk_solve_L<T, B, B>(D, invd, Ub);
          __builtin_amdgcn_sched_barrier(0);
          if (NRHS) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
              double t = xs[i * B + r];
#pragma unroll
              for (int p = 0; p < r; ++p) t = fma(-N::val(D[r][p]), z[p], t);
              z[r] = t * N::val(invd[r]);
            }
          }
          // factors -> workspace (log-determinant terms are summed from the stored diagonals after the solve)
#pragma unroll
          for (int r = 0; r < B; ++r) {
            Wn(i).set(Lay::W_I + r, invd[r]);
#pragma unroll
            for (int c = 0; c < B; ++c) {
              Wn(i).set(Lay::W_L + r * B + c, (c <= r) ? D[r][c] : N::zero());
              Wn(i).set(Lay::W_UA + r * B + c, Ua[r][c]);
              Wn(i).set(Lay::W_UB + r * B + c, Ub[r][c]);
            }
          }
          if (NRHS) {
#pragma unroll
            for (int r = 0; r < B; ++r) xs[i * B + r] = z[r];  // z_i overwrites y_i (read back in the backward pass)
          }
          // phase A: left neighbour a
          const int sa = a >> 1;
#pragma unroll
          for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {  // D_a -= Ua^T Ua (symmetric, keep both halves)
              T t = (IS0 && fast0) ? Da[r][c] : F.getD(r * B + c, sa);
#pragma unroll
              for (int p = 0; p < B; ++p) t = N::nfma(Ua[p][r], Ua[p][c], t);
              F.setD(r * B + c, sa, t);
              if (c != r) F.setD(c * B + r, sa, t);
            }
#pragma unroll
          for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c < B; ++c) {  // E(a) := A'[b,a] = -Ub^T Ua
              T t = N::zero();
#pragma unroll
              for (int p = 0; p < B; ++p) t = N::nfma(Ub[p][r], Ua[p][c], t);
              F.setE(r * B + c, sa, t);
            }
          if (NRHS) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
              double t = xs[a * B + r];
#pragma unroll
              for (int p = 0; p < B; ++p) t = fma(-N::val(Ua[p][r]), z[p], t);
              xs[a * B + r] = t;
            }
          }
        }
        bcr_barrier<BIG>();
        if (hasb) {  // phase B: right neighbour b
          const int sb = b >> 1;
#pragma unroll
          for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
              T t = F.getD(r * B + c, sb);
#pragma unroll
              for (int p = 0; p < B; ++p) t = N::nfma(Ub[p][r], Ub[p][c], t);
              F.setD(r * B + c, sb, t);
              if (c != r) F.setD(c * B + r, sb, t);
            }
          if (NRHS) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
              double t = xs[b * B + r];
#pragma unroll
              for (int p = 0; p < B; ++p) t = fma(-N::val(Ub[p][r]), z[p], t);
              xs[b * B + r] = t;
            }
          }
        }
        bcr_barrier<BIG>();
      }
    };
    if (l == 0) fwd_thread(std::true_type{});
    else fwd_thread(std::false_type{});
    stamp();
  }

  // ---------------- root (node 0): lane-distributed, group 0 ----------------
  if (grp == 0) {
    T d = lane_on ? F.getD(e, 0) : N::zero();
    T invd[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
      T pj = gshfl<T>(d, j * B + j, GS);
      if (!(N::val(pj) > 0.0) && !bad) bad = j + 1;
      T ljj, inv;
      N::sqrt_inv(pj, ljj, inv);
      invd[j] = inv;
      if (c == j) d = (r == j) ? ljj : ((r > j) ? d * inv : d);
      T lrj = gshfl<T>(d, r * B + j, GS), lcj = gshfl<T>(d, c * B + j, GS);
      if (c > j && r >= c) d = N::nfma(lrj, lcj, d);
    }
    // X = L^-1 (forward, identity rhs), then Sigma_00 = L^-T X ; x_0 = L^-T L^-1 y_0
    T xi = (r == c) ? N::make(1.0, 0.0) : N::zero();
    double z = (NRHS && lane_on && c == 0) ? xs[r] : 0.0;
#pragma unroll
    for (int rr = 0; rr < B; ++rr) {
#pragma unroll
      for (int p = 0; p < rr; ++p) {
        T lv = gshfl<T>(d, rr * B + p, GS);
        T xp = gshfl<T>(xi, p * B + c, GS);
        double zp = NRHS ? __shfl(z, p * B, GS) : 0.0;
        if (r == rr) { xi = N::nfma(lv, xp, xi); z = fma(-N::val(lv), zp, z); }
      }
      if (r == rr) { xi = xi * invd[rr]; z = z * N::val(invd[rr]); }
    }
#pragma unroll
    for (int rr = B - 1; rr >= 0; --rr) {
#pragma unroll
      for (int p = rr + 1; p < B; ++p) {
        T lv = gshfl<T>(d, p * B + rr, GS);   // L[p][rr]
        T xp = gshfl<T>(xi, p * B + c, GS);
        double zp = NRHS ? __shfl(z, p * B, GS) : 0.0;
        if (r == rr) { xi = N::nfma(lv, xp, xi); z = fma(-N::val(lv), zp, z); }
      }
      if (r == rr) { xi = xi * invd[rr]; z = z * N::val(invd[rr]); }
    }
    if (lane_on) {
      Wn(0).set(Lay::W_SD + e, xi);
      if (r >= c && r < M) S.store((long)(r - c) * M + c, xi);
      Wn(0).set(Lay::W_L + e, (r >= c) ? d : N::zero());
      if (NRHS && c == 0) xs[r] = z;
    }
  }
  __syncthreads();
  stamp();

  // ---------------- backward: solve + selected inverse ----------------
  for (int l = levels - 1; l >= 0; --l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    if (ne <= LANE_MAX_NODES) {
      for (int m0 = 0; m0 < ne; m0 += NG) {
      const int m = m0 + grp;
      if (m < ne) {
        const int i = h + m * 2 * h, a = i - h, b = i + h;
        const bool hasb = b < nb;
        T d = N::zero(), ga = N::zero(), gb = N::zero(), saa = N::zero(), sbb = N::zero(), sba = N::zero();
        T invd[B];
#pragma unroll
        for (int q = 0; q < B; ++q) invd[q] = Wn(i).get(Lay::W_I + q);
        const bool e_is_a = ((a / (2 * h)) & 1) != 0;
        if (lane_on) {
          d = Wn(i).get(Lay::W_L + e);
          ga = Wn(i).get(Lay::W_UA + e);
          gb = Wn(i).get(Lay::W_UB + e);
          saa = Wn(a).get(Lay::W_SD + e);
          if (hasb) {
            sbb = Wn(b).get(Lay::W_SD + e);
            sba = e_is_a ? Wn(a).get(Lay::W_CB + c * B + r) : Wn(b).get(Lay::W_CA + e);   // Sigma_ba[r][c]
          }
        }
        if (NRHS) {  // t = z - U_a x_a - U_b x_b on lanes (r, 0), then x_i = L^-T t
          double t = (lane_on && c == 0) ? xs[i * B + r] : 0.0;
#pragma unroll
          for (int p = 0; p < B; ++p) {
            double uarp = N::val(gshfl<T>(ga, r * B + p, GS)), ubrp = N::val(gshfl<T>(gb, r * B + p, GS));
            t = fma(-uarp, xs[a * B + p], t);
            if (hasb) t = fma(-ubrp, xs[b * B + p], t);
          }
#pragma unroll
          for (int rr = B - 1; rr >= 0; --rr) {
#pragma unroll
            for (int p = rr + 1; p < B; ++p) {
              double lv = N::val(gshfl<T>(d, p * B + rr, GS));
              double tp = __shfl(t, p * B, GS);
              if (r == rr) t = fma(-lv, tp, t);
            }
            if (r == rr) t = t * N::val(invd[rr]);
          }
          if (lane_on && c == 0) xs[i * B + r] = t;
        }
        // G_a = L^-T U_a, G_b = L^-T U_b ; Dinv = L^-T L^-1
        T xi = (r == c) ? N::make(1.0, 0.0) : N::zero();
#pragma unroll
        for (int rr = 0; rr < B; ++rr) {   // xi <- L^-1 I
#pragma unroll
          for (int p = 0; p < rr; ++p) {
            T lv = gshfl<T>(d, rr * B + p, GS);
            T xp = gshfl<T>(xi, p * B + c, GS);
            if (r == rr) xi = N::nfma(lv, xp, xi);
          }
          if (r == rr) xi = xi * invd[rr];
        }
#pragma unroll
        for (int rr = B - 1; rr >= 0; --rr) {
#pragma unroll
          for (int p = rr + 1; p < B; ++p) {
            T lv = gshfl<T>(d, p * B + rr, GS);
            T gap = gshfl<T>(ga, p * B + c, GS), gbp = gshfl<T>(gb, p * B + c, GS), xp = gshfl<T>(xi, p * B + c, GS);
            if (r == rr) { ga = N::nfma(lv, gap, ga); gb = N::nfma(lv, gbp, gb); xi = N::nfma(lv, xp, xi); }
          }
          if (r == rr) { ga = ga * invd[rr]; gb = gb * invd[rr]; xi = xi * invd[rr]; }
        }
        // C_a = -(G_a S_aa + G_b S_ba), C_b = -(G_a S_ab + G_b S_bb)
        T ca = N::zero(), cb = N::zero();
#pragma unroll
        for (int p = 0; p < B; ++p) {
          T garp = gshfl<T>(ga, r * B + p, GS), gbrp = gshfl<T>(gb, r * B + p, GS);
          T saapc = gshfl<T>(saa, p * B + c, GS), sbapc = gshfl<T>(sba, p * B + c, GS);
          T sbacp = gshfl<T>(sba, c * B + p, GS), sbbpc = gshfl<T>(sbb, p * B + c, GS);
          ca = N::nfma(garp, saapc, ca);
          ca = N::nfma(gbrp, sbapc, ca);
          cb = N::nfma(garp, sbacp, cb);
          cb = N::nfma(gbrp, sbbpc, cb);
        }
        // S_ii = Dinv - C_a G_a^T - C_b G_b^T
        T sii = xi;
#pragma unroll
        for (int p = 0; p < B; ++p) {
          T carp = gshfl<T>(ca, r * B + p, GS), gacp = gshfl<T>(ga, c * B + p, GS);
          T cbrp = gshfl<T>(cb, r * B + p, GS), gbcp = gshfl<T>(gb, c * B + p, GS);
          sii = N::nfma(carp, gacp, sii);
          sii = N::nfma(cbrp, gbcp, sii);
        }
        if (lane_on) {
          if (l > 0) {   // blocks a deeper level will read (nothing reads a level-0 node again)
            Wn(i).set(Lay::W_SD + e, sii);
            Wn(i).set(Lay::W_CA + e, ca);
            Wn(i).set(Lay::W_CB + e, cb);
          }
          // the band of the inverse leaves from here: Sigma_ii (lower part) and, at level 0, the couplings to both neighbours
          if (r >= c && i * B + r < M) S.store((long)(r - c) * M + i * B + c, sii);
          if (l == 0) {
            // Sigma[iB + r, aB + c] = C_a[r][c] (r <= c);  Sigma[bB + r, iB + c] = C_b[c][r] (r <= c): lane (r, c) holds C_b[r][c]
            if (r <= c && i * B + r < M) S.store((long)(B + r - c) * M + a * B + c, ca);
            if (hasb && c <= r && b * B + c < M) S.store((long)(B + c - r) * M + i * B + r, cb);
          }
        }
      }
      __syncthreads();
      }
      stamp();
      continue;
    }
    // level 0 and the deeper levels get separate copies of the one-thread-per-node body (own register allocation, no band
    // gather / output code where it is not needed)
    auto bwd_thread = [&](auto is0) {
      constexpr bool IS0 = decltype(is0)::value;
      for (int m0 = 0; m0 < ne; m0 += BCR_THREADS) {
        const int m = m0 + tid;
        if (m < ne) {
          const int i = h + m * 2 * h, a = i - h, b = i + h;
          const bool hasb = b < nb;
          T L[B][B], invd[B], Ga[B][B], Gb[B][B];
#pragma unroll
          for (int r = 0; r < B; ++r) {
            invd[r] = Wn(i).get(Lay::W_I + r);
#pragma unroll
            for (int c = 0; c < B; ++c) {
              L[r][c] = Wn(i).get(Lay::W_L + r * B + c);
              Ga[r][c] = Wn(i).get(Lay::W_UA + r * B + c);
              Gb[r][c] = Wn(i).get(Lay::W_UB + r * B + c);
            }
          }
          if (NRHS) {  // x_i = L^-T (z - Ua x_a - Ub x_b)
            double t[B];
#pragma unroll
            for (int r = 0; r < B; ++r) {
              double v = xs[i * B + r];
#pragma unroll
              for (int p = 0; p < B; ++p) {
                v = fma(-N::val(Ga[r][p]), xs[a * B + p], v);
                if (hasb) v = fma(-N::val(Gb[r][p]), xs[b * B + p], v);
              }
              t[r] = v;
            }
#pragma unroll
            for (int r = B - 1; r >= 0; --r) {
              double v = t[r];
#pragma unroll
              for (int p = r + 1; p < B; ++p) v = fma(-N::val(L[p][r]), t[p], v);
              t[r] = v * N::val(invd[r]);
            }
#pragma unroll
            for (int r = 0; r < B; ++r) xs[i * B + r] = t[r];
          }
          __builtin_amdgcn_sched_barrier(0);
          blk_solve_LT<T, B, B>(L, invd, Ga);  // G_a = L^-T U_a = D^-1 A[i,a]
          __builtin_amdgcn_sched_barrier(0);
          blk_solve_LT<T, B, B>(L, invd, Gb);
          __builtin_amdgcn_sched_barrier(0);
          // neighbour blocks of the inverse
          T Saa[B][B], Sbb[B][B], Sba[B][B];
          const bool e_is_a = ((a / (2 * h)) & 1) != 0;  // which of a,b was eliminated at level l+1
#pragma unroll
          for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c < B; ++c) {
              Saa[r][c] = Wn(a).get(Lay::W_SD + r * B + c);
              if (hasb) {
                Sbb[r][c] = Wn(b).get(Lay::W_SD + r * B + c);
                // Sigma_ba: e = a -> (C_a^b)^T ; e = b -> C_b^a
                Sba[r][c] = e_is_a ? Wn(a).get(Lay::W_CB + c * B + r) : Wn(b).get(Lay::W_CA + r * B + c);
              } else {
                Sbb[r][c] = N::zero();
                Sba[r][c] = N::zero();
              }
            }
          T Ca[B][B], Cb[B][B], Sii[B][B];
#pragma unroll
          for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c < B; ++c) {
              T ta = N::zero(), tb = N::zero();
#pragma unroll
              for (int p = 0; p < B; ++p) {
                ta = N::nfma(Ga[r][p], Saa[p][c], ta);       // -(Ga Saa)
                ta = N::nfma(Gb[r][p], Sba[p][c], ta);       // -(Gb Sba)
                tb = N::nfma(Ga[r][p], Sba[c][p], tb);       // -(Ga Sab), Sab = Sba^T
                tb = N::nfma(Gb[r][p], Sbb[p][c], tb);       // -(Gb Sbb)
              }
              Ca[r][c] = ta;
              Cb[r][c] = tb;
            }
          __builtin_amdgcn_sched_barrier(0);
          // D_i^-1 = L^-T L^-1
#pragma unroll
          for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c < B; ++c) Sii[r][c] = (r == c) ? N::make(1.0, 0.0) : N::zero();
          blk_solve_L<T, B, B>(L, invd, Sii);
          blk_solve_LT<T, B, B>(L, invd, Sii);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < B; ++r)
#pragma unroll
            for (int c = 0; c < B; ++c) {
              T t = Sii[r][c];
#pragma unroll
              for (int p = 0; p < B; ++p) {
                t = N::nfma(Ca[r][p], Ga[c][p], t);
                t = N::nfma(Cb[r][p], Gb[c][p], t);
              }
              if constexpr (!IS0) {   // blocks a deeper level will read (nothing reads a level-0 node again)
                Wn(i).set(Lay::W_SD + r * B + c, t);
                Wn(i).set(Lay::W_CA + r * B + c, Ca[r][c]);
                Wn(i).set(Lay::W_CB + r * B + c, Cb[r][c]);
              }
              // the band of the inverse leaves from here (see the lane-distributed branch)
              if (c <= r && i * B + r < M) S.store((long)(r - c) * M + i * B + c, t);
              if constexpr (IS0) {
                if (r <= c && i * B + r < M) S.store((long)(B + r - c) * M + a * B + c, Ca[r][c]);
                if (hasb && c <= r && b * B + c < M) S.store((long)(B + c - r) * M + i * B + r, Cb[r][c]);
              }
            }
        }
        __syncthreads();
      }
    };
    if (l == 0) bwd_thread(std::true_type{});
    else bwd_thread(std::false_type{});
    stamp();
  }

  // ---------------- outputs: x, logdet, info (the band of the inverse was written level by level) ----------------
  // right padding of the band rows (structural zeros): the last B columns
  for (int col = M - B + tid; col < M; col += BCR_THREADS)
    if (col >= 0)
#pragma unroll
      for (int d = 1; d <= B; ++d)
        if (col + d >= M) S.store((long)d * M + col, N::zero());
  if (NRHS)
    for (int r = tid; r < M; r += BCR_THREADS) x[(long)r * rhs_stride] = xs[r];
  // log|A| = 2 sum log diag(L_i) over all nodes, off the dependent chain (padding rows have L = 1)
  {
    constexpr int RMAX = (2 * NS * B + BCR_THREADS - 1) / BCR_THREADS;
    T dv[RMAX];
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {   // loads first ...
      const int e = tid + q * BCR_THREADS;
      dv[q] = (e < nb * B) ? Wn(e / B).get(Lay::W_L + (e % B) * B + (e % B)) : N::make(1.0, 0.0);
    }
    double prod = 1.0;   // ... then ONE log of the product (<= 16 factors between ~1e-4 and ~1e4: no range problem) instead of RMAX logs
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      prod *= N::val(dv[q]);
      dld_acc += 2.0 * N::tan(dv[q]) / N::val(dv[q]);
    }
    ld_acc = 2.0 * log(prod);
  }
  // one combined reduction (log-det, its tangent, first failing column) with a single barrier
  {
    const double a = wave_sum_dpp(ld_acc), b = wave_sum_dpp(dld_acc);
    int bm = bad ? bad : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(bm, off, 64); bm = o < bm ? o : bm; }
    int* sbad = reinterpret_cast<int*>(red + 32);
    const int lane_ = tid & 63, w_ = tid >> 6;
    if (lane_ == 0) { red[w_] = a; red[16 + w_] = b; sbad[w_] = bm; }
    __syncthreads();
    stamp();
    if (tid == 0) {
      double tot = 0.0, dtot = 0.0;
      int bmin = 0x7fffffff;
      for (int w = 0; w < BCR_THREADS / 64; ++w) { tot += red[w]; dtot += red[16 + w]; bmin = sbad[w] < bmin ? sbad[w] : bmin; }
      logdet[0] = tot;
      logdet[1] = dtot;
      *info = (bmin == 0x7fffffff) ? 0 : bmin;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Further right-hand-side columns (y with D > 1 outputs, gpr.py:75,80-82: solve_triang_mat with an M x D rhs) from the factor
// records bcr_solve<double, B, 1> left in `ws`: the same forward / backward substitution over the elimination tree, one thread
// per (node, column) pair per level, columns d0 .. d0 + nd - 1 of the row-major (M, D) arrays rhs / x.
// lds: nb * B * nd doubles (the survivor image of bcr_solve is dead by now).  Called by the whole workgroup.
// ------------------------------------------------------------------------------------------------------------
template <int B, bool BIG = false>
__device__ __attribute__((noinline)) void bcr_solve_more(const double* rhs, double* x, int D, int d0, int nd, int M, double* ws, double* lds) {
  using Lay = BcrLayout<B, 1>;
  const int tid = threadIdx.x;
  const int nb = (M + B - 1) / B;
  constexpr int NN = 2 * bcr_ns<double, B, BIG>();
  auto W = [&](int node, int f) -> double { return ws[(long)f * NN + node]; };
  double* xs = lds;                                          // [row][nd]
  for (int e = tid; e < nb * B * nd; e += BCR_THREADS) {
    const int row = e / nd, d = e - row * nd;
    xs[e] = (row < M) ? rhs[(long)row * D + d0 + d] : 0.0;
  }
  __syncthreads();
  int levels = 0;
  while ((1 << levels) < nb) ++levels;
  for (int l = 0; l < levels; ++l) {                         // forward: z_i = L^-1 y_i, y_a -= U_a^T z_i | y_b -= U_b^T z_i
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    for (int w = tid; w < ne * nd; w += BCR_THREADS) {
      const int m = w / nd, d = w - m * nd;
      const int i = h + m * 2 * h, a = i - h;
      double z[B];
#pragma unroll
      for (int r = 0; r < B; ++r) {
        double t = xs[(i * B + r) * nd + d];
#pragma unroll
        for (int p = 0; p < r; ++p) t = fma(-W(i, Lay::W_L + r * B + p), z[p], t);
        z[r] = t * W(i, Lay::W_I + r);
      }
#pragma unroll
      for (int r = 0; r < B; ++r) {
        xs[(i * B + r) * nd + d] = z[r];
        double t = xs[(a * B + r) * nd + d];
#pragma unroll
        for (int p = 0; p < B; ++p) t = fma(-W(i, Lay::W_UA + p * B + r), z[p], t);
        xs[(a * B + r) * nd + d] = t;
      }
    }
    __syncthreads();
    for (int w = tid; w < ne * nd; w += BCR_THREADS) {
      const int m = w / nd, d = w - m * nd;
      const int i = h + m * 2 * h, b = i + h;
      if (b < nb) {
#pragma unroll
        for (int r = 0; r < B; ++r) {
          double t = xs[(b * B + r) * nd + d];
#pragma unroll
          for (int p = 0; p < B; ++p) t = fma(-W(i, Lay::W_UB + p * B + r), xs[(i * B + p) * nd + d], t);
          xs[(b * B + r) * nd + d] = t;
        }
      }
    }
    __syncthreads();
  }
  for (int d = tid; d < nd; d += BCR_THREADS) {              // root: x_0 = L^-T L^-1 y_0  (only L of the root record is stored)
    double t[B];
#pragma unroll
    for (int r = 0; r < B; ++r) {
      double v = xs[r * nd + d];
#pragma unroll
      for (int p = 0; p < r; ++p) v = fma(-W(0, Lay::W_L + r * B + p), t[p], v);
      t[r] = v / W(0, Lay::W_L + r * B + r);
    }
#pragma unroll
    for (int r = B - 1; r >= 0; --r) {
      double v = t[r];
#pragma unroll
      for (int p = r + 1; p < B; ++p) v = fma(-W(0, Lay::W_L + p * B + r), t[p], v);
      t[r] = v / W(0, Lay::W_L + r * B + r);
    }
#pragma unroll
    for (int r = 0; r < B; ++r) xs[r * nd + d] = t[r];
  }
  __syncthreads();
  for (int l = levels - 1; l >= 0; --l) {                    // backward: x_i = L^-T (z_i - U_a x_a - U_b x_b)
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    for (int w = tid; w < ne * nd; w += BCR_THREADS) {
      const int m = w / nd, d = w - m * nd;
      const int i = h + m * 2 * h, a = i - h, b = i + h;
      const bool hasb = b < nb;
      double t[B];
#pragma unroll
      for (int r = 0; r < B; ++r) {
        double v = xs[(i * B + r) * nd + d];
#pragma unroll
        for (int p = 0; p < B; ++p) {
          v = fma(-W(i, Lay::W_UA + r * B + p), xs[(a * B + p) * nd + d], v);
          if (hasb) v = fma(-W(i, Lay::W_UB + r * B + p), xs[(b * B + p) * nd + d], v);
        }
        t[r] = v;
      }
#pragma unroll
      for (int r = B - 1; r >= 0; --r) {
        double v = t[r];
#pragma unroll
        for (int p = r + 1; p < B; ++p) v = fma(-W(i, Lay::W_L + p * B + r), t[p], v);
        t[r] = v * W(i, Lay::W_I + r);
      }
#pragma unroll
      for (int r = 0; r < B; ++r) xs[(i * B + r) * nd + d] = t[r];
    }
    __syncthreads();
  }
  for (int e = tid; e < M * nd; e += BCR_THREADS) {
    const int row = e / nd, d = e - row * nd;
    x[(long)row * D + d0 + d] = xs[e];
  }
  __syncthreads();
}

}  // namespace asvgp
