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
#include "band_sweeps.hpp"

namespace asvgp {

constexpr int BCR_THREADS = 256;

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
// SoA array of T: element (field, slot) ; Dual keeps value and tangent planes apart so every access is 8 B / lane.
template <typename T> struct Soa;
template <> struct Soa<double> {
  double* p; long n;  // n = slots
  __device__ __forceinline__ double get(int f, long s) const { return p[(long)f * n + s]; }
  __device__ __forceinline__ void set(int f, long s, double v) const { p[(long)f * n + s] = v; }
  static __host__ __device__ constexpr int planes() { return 1; }
};
template <> struct Soa<Dual> {
  double* p; long n; long plane;  // plane = offset (in doubles) of the tangent plane
  __device__ __forceinline__ Dual get(int f, long s) const { long o = (long)f * n + s; return {p[o], p[plane + o]}; }
  __device__ __forceinline__ void set(int f, long s, Dual v) const { long o = (long)f * n + s; p[o] = v.v; p[plane + o] = v.d; }
  static __host__ __device__ constexpr int planes() { return 2; }
};
template <typename T> __device__ __forceinline__ Soa<T> make_soa(double* p, long slots, int fields);
template <> __device__ __forceinline__ Soa<double> make_soa<double>(double* p, long slots, int) { return {p, slots}; }
template <> __device__ __forceinline__ Soa<Dual> make_soa<Dual>(double* p, long slots, int fields) { return {p, slots, (long)fields * slots}; }

template <int B, int NRHS> struct BcrLayout {
  // forward LDS fields per survivor slot
  static constexpr int F_D = 0, F_E = B * B, F_N = 2 * B * B;   // (y / z / x live in a row-indexed LDS vector)
  // factor workspace fields per node
  static constexpr int W_L = 0, W_I = B * B, W_UA = B * B + B, W_UB = 2 * B * B + B,
                       W_SD = 3 * B * B + B,                 // Sigma_ii
                       W_CA = W_SD + B * B, W_CB = W_CA + B * B, W_N = W_CB + B * B;
};
template <typename T> __host__ __device__ constexpr int planes_of() { return sizeof(T) / sizeof(double); }

template <typename T, int B, int NRHS>
__host__ __device__ inline size_t bcr_lds_doubles(long nb) {
  long slots = (nb + 1) / 2;
  return (size_t)planes_of<T>() * BcrLayout<B, NRHS>::F_N * slots + (size_t)nb * B + 64;
}
template <typename T, int B, int NRHS>
__host__ __device__ inline size_t bcr_ws_doubles(long nb) {
  return (size_t)planes_of<T>() * BcrLayout<B, NRHS>::W_N * nb;
}

// block extraction from the lower band (B+1, M): D_n (lower part) and E(n) = A[n+1, n] (upper-triangular block)
template <typename T, int B>
__device__ __forceinline__ T band_D(const BandPtr<T>& A, int M, int n, int r, int c) {  // r >= c
  int col = n * B + c, row = n * B + r;
  if (row >= M) return (r == c) ? Num<T>::make(1.0, 0.0) : Num<T>::zero();  // identity padding
  return A.load((long)(r - c) * M + col, true);
}
template <typename T, int B>
__device__ __forceinline__ T band_E(const BandPtr<T>& A, int M, int n, int r, int c) {  // A[(n+1)B + r, nB + c]
  if (r > c) return Num<T>::zero();
  int col = n * B + c, row = (n + 1) * B + r;
  if (row >= M) return Num<T>::zero();
  return A.load((long)(B + r - c) * M + col, true);
}

// ------------------------------------------------------------------------------------------------------------
// The whole solve for one matrix.  Called by all BCR_THREADS threads of one workgroup.
//   A: lower band (B+1, M);  rhs: (M) or null (NRHS = 0);  ws: bcr_ws_doubles;  lds: bcr_lds_doubles
//   out: S lower band of A^-1 (B+1, M), x = A^-1 rhs (M), logdet (1), info (first bad column + 1)
// ------------------------------------------------------------------------------------------------------------
template <typename T, int B, int NRHS>
__device__ void bcr_solve(BandPtr<T> A, const double* rhs, int M, double* ws, double* lds, BandOut<T> S, double* x,
                          double* logdet, int* info) {
  using N = Num<T>;
  using Lay = BcrLayout<B, NRHS>;
  const int tid = threadIdx.x;
  const int nb = (M + B - 1) / B;
  const long slots = (nb + 1) / 2;
  Soa<T> F = make_soa<T>(lds, slots, Lay::F_N);                               // survivors: D | E | y
  double* xs = lds + (size_t)planes_of<T>() * Lay::F_N * slots;               // x per node (NRHS = 1) in LDS
  double* red = xs + (size_t)nb * B;                                          // 64 doubles scratch
  Soa<T> W = make_soa<T>(ws, nb, Lay::W_N);
  int bad = 0;
  double ld_acc = 0.0, dld_acc = 0.0;
  int levels = 0;
  while ((1 << levels) < nb) ++levels;

  // pre-pass: even nodes -> LDS slots (D lower part + mirrored upper, y)
  for (int n = 2 * tid; n < nb; n += 2 * BCR_THREADS) {
    const long s = n >> 1;
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        T v = band_D<T, B>(A, M, n, r, c);
        F.set(Lay::F_D + r * B + c, s, v);
        if (c != r) F.set(Lay::F_D + c * B + r, s, v);
      }
  }
  // NRHS <= 1: y lives in xs[] indexed by row and is updated in place (y -> z -> x)
  if (NRHS)
    for (int r = tid; r < nb * B; r += BCR_THREADS) xs[r] = (r < M) ? rhs[r] : 0.0;
  __syncthreads();

  // ---------------- forward elimination ----------------
  for (int l = 0; l < levels; ++l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    for (int m0 = 0; m0 < ne; m0 += BCR_THREADS) {   // more eliminated nodes than threads: several rounds
      const int m = m0 + tid;
      const bool act = m < ne;
      const int i = h + m * 2 * h, a = i - h, b = i + h;
      const bool hasb = act && (b < nb);
      T Ua[B][B], Ub[B][B], D[B][B], invd[B];
      double z[B];
      if (act) {
        // load D_i, A[i,a] (= E(a)), A[b,i]^T (= E(i)^T), y_i
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
          for (int c = 0; c < B; ++c) {
            if (l == 0) {
              D[r][c] = (c <= r) ? band_D<T, B>(A, M, i, r, c) : N::zero();
              Ua[r][c] = band_E<T, B>(A, M, a, r, c);
              Ub[r][c] = hasb ? band_E<T, B>(A, M, i, c, r) : N::zero();  // transpose: A[i,b] = A[b,i]^T
            } else {
              D[r][c] = (c <= r) ? F.get(Lay::F_D + r * B + c, i >> 1) : N::zero();
              Ua[r][c] = F.get(Lay::F_E + r * B + c, a >> 1);
              Ub[r][c] = hasb ? F.get(Lay::F_E + c * B + r, i >> 1) : N::zero();
            }
          }
        blk_chol<T, B>(D, invd, bad, i * B);
        blk_solve_L<T, B, B>(D, invd, Ua);
        blk_solve_L<T, B, B>(D, invd, Ub);
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
          W.set(Lay::W_I + r, i, invd[r]);
#pragma unroll
          for (int c = 0; c < B; ++c) {
            W.set(Lay::W_L + r * B + c, i, (c <= r) ? D[r][c] : N::zero());
            W.set(Lay::W_UA + r * B + c, i, Ua[r][c]);
            W.set(Lay::W_UB + r * B + c, i, Ub[r][c]);
          }
        }
        if (NRHS) {
#pragma unroll
          for (int r = 0; r < B; ++r) xs[i * B + r] = z[r];  // z_i overwrites y_i (read back in the backward pass)
        }
        // phase A: left neighbour a
        const long sa = a >> 1;
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) {  // D_a -= Ua^T Ua (symmetric, keep both halves)
            T t = F.get(Lay::F_D + r * B + c, sa);
#pragma unroll
            for (int p = 0; p < B; ++p) t = N::nfma(Ua[p][r], Ua[p][c], t);
            F.set(Lay::F_D + r * B + c, sa, t);
            if (c != r) F.set(Lay::F_D + c * B + r, sa, t);
          }
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
          for (int c = 0; c < B; ++c) {  // E(a) := A'[b,a] = -Ub^T Ua
            T t = N::zero();
#pragma unroll
            for (int p = 0; p < B; ++p) t = N::nfma(Ub[p][r], Ua[p][c], t);
            F.set(Lay::F_E + r * B + c, sa, t);
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
      __syncthreads();
      if (hasb) {  // phase B: right neighbour b
        const long sb = b >> 1;
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) {
            T t = F.get(Lay::F_D + r * B + c, sb);
#pragma unroll
            for (int p = 0; p < B; ++p) t = N::nfma(Ub[p][r], Ub[p][c], t);
            F.set(Lay::F_D + r * B + c, sb, t);
            if (c != r) F.set(Lay::F_D + c * B + r, sb, t);
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
      __syncthreads();
    }
  }

  // ---------------- root (node 0) ----------------
  if (tid == 0) {
    T D[B][B], invd[B], Li[B][B];
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int c = 0; c < B; ++c) D[r][c] = (c <= r) ? F.get(Lay::F_D + r * B + c, 0) : N::zero();
    blk_chol<T, B>(D, invd, bad, 0);
#pragma unroll
    for (int r = 0; r < B; ++r) W.set(Lay::W_L + r * B + r, 0, D[r][r]);
    // Sigma_00 = L^-T L^-1
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int c = 0; c < B; ++c) Li[r][c] = (r == c) ? N::make(1.0, 0.0) : N::zero();
    blk_solve_L<T, B, B>(D, invd, Li);
    blk_solve_LT<T, B, B>(D, invd, Li);
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int c = 0; c < B; ++c) W.set(Lay::W_SD + r * B + c, 0, Li[r][c]);
    if (NRHS) {
      double z[B];
#pragma unroll
      for (int r = 0; r < B; ++r) {
        double t = xs[r];
#pragma unroll
        for (int p = 0; p < r; ++p) t = fma(-N::val(D[r][p]), z[p], t);
        z[r] = t * N::val(invd[r]);
      }
#pragma unroll
      for (int r = B - 1; r >= 0; --r) {
        double t = z[r];
#pragma unroll
        for (int p = r + 1; p < B; ++p) t = fma(-N::val(D[p][r]), z[p], t);
        z[r] = t * N::val(invd[r]);
      }
#pragma unroll
      for (int r = 0; r < B; ++r) xs[r] = z[r];
    }
  }
  __syncthreads();

  // ---------------- backward: solve + selected inverse ----------------
  for (int l = levels - 1; l >= 0; --l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    for (int m0 = 0; m0 < ne; m0 += BCR_THREADS) {
      const int m = m0 + tid;
      if (m < ne) {
        const int i = h + m * 2 * h, a = i - h, b = i + h;
        const bool hasb = b < nb;
        T L[B][B], invd[B], Ga[B][B], Gb[B][B];
#pragma unroll
        for (int r = 0; r < B; ++r) {
          invd[r] = W.get(Lay::W_I + r, i);
#pragma unroll
          for (int c = 0; c < B; ++c) {
            L[r][c] = W.get(Lay::W_L + r * B + c, i);
            Ga[r][c] = W.get(Lay::W_UA + r * B + c, i);
            Gb[r][c] = W.get(Lay::W_UB + r * B + c, i);
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
        blk_solve_LT<T, B, B>(L, invd, Ga);  // G_a = L^-T U_a = D^-1 A[i,a]
        blk_solve_LT<T, B, B>(L, invd, Gb);
        // neighbour blocks of the inverse
        T Saa[B][B], Sbb[B][B], Sba[B][B];
        const bool e_is_a = ((a / (2 * h)) & 1) != 0;  // which of a,b was eliminated at level l+1
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
          for (int c = 0; c < B; ++c) {
            Saa[r][c] = W.get(Lay::W_SD + r * B + c, a);
            if (hasb) {
              Sbb[r][c] = W.get(Lay::W_SD + r * B + c, b);
              // Sigma_ba: e = a -> (C_a^b)^T ; e = b -> C_b^a
              Sba[r][c] = e_is_a ? W.get(Lay::W_CB + c * B + r, a) : W.get(Lay::W_CA + r * B + c, b);
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
        // D_i^-1 = L^-T L^-1
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
          for (int c = 0; c < B; ++c) Sii[r][c] = (r == c) ? N::make(1.0, 0.0) : N::zero();
        blk_solve_L<T, B, B>(L, invd, Sii);
        blk_solve_LT<T, B, B>(L, invd, Sii);
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
            W.set(Lay::W_SD + r * B + c, i, t);
            W.set(Lay::W_CA + r * B + c, i, Ca[r][c]);
            W.set(Lay::W_CB + r * B + c, i, Cb[r][c]);
          }
      }
      __syncthreads();
    }
  }

  // ---------------- outputs: band of the inverse, x, logdet, info ----------------
  for (int n = tid; n < nb; n += BCR_THREADS) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        int row = n * B + r, col = n * B + c;
        if (row < M) S.store((long)(r - c) * M + col, W.get(Lay::W_SD + r * B + c, n));
      }
    if (n + 1 < nb) {  // Sigma[(n+1)B + r, nB + c], r <= c : from the odd member of the pair
#pragma unroll
      for (int r = 0; r < B; ++r)
#pragma unroll
        for (int c = r; c < B; ++c) {
          int row = (n + 1) * B + r, col = n * B + c;
          T v = (n & 1) ? W.get(Lay::W_CB + c * B + r, n) : W.get(Lay::W_CA + r * B + c, n + 1);
          if (row < M) S.store((long)(B + r - c) * M + col, v);
        }
    }
    // right padding of the band rows (structural zeros)
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int d = 1; d <= B; ++d) {
        int col = n * B + r;
        if (col < M && col + d >= M) S.store((long)d * M + col, N::zero());
      }
  }
  if (NRHS)
    for (int r = tid; r < M; r += BCR_THREADS) x[r] = xs[r];
  // log|A| = 2 sum log diag(L_i) over all nodes, off the dependent chain (padding rows have L = 1)
  for (int e = tid; e < nb * B; e += BCR_THREADS) {
    T d = W.get(Lay::W_L + (e % B) * B + (e % B), e / B);
    ld_acc += 2.0 * log(N::val(d));
    dld_acc += 2.0 * N::tan(d) / N::val(d);
  }
  // reductions
  double tot = block_sum(ld_acc, red);
  double dtot = block_sum(dld_acc, red);
  if (tid == 0) {
    logdet[0] = tot;
    logdet[1] = dtot;
  }
  int* sbad = reinterpret_cast<int*>(red + 32);  // smallest failing column + 1 over the workgroup
  __syncthreads();
  if (tid == 0) *sbad = 0x7fffffff;
  __syncthreads();
  if (bad) atomicMin(sbad, bad);
  __syncthreads();
  if (tid == 0) *info = (*sbad == 0x7fffffff) ? 0 : *sbad;
}

}  // namespace asvgp
