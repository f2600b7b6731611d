// Narrow-band serial recurrences on ONE wavefront (wave64), shared by the operator entry points and the
// fused ELBO driver.  Replaces banded_matrices' cholesky_band / inverse_from_cholesky_band / solve_triang_mat
// C++/Eigen TF ops (reference call sites gpr.py:56,59,73,75).
//
// Layout idea (no LDS, no cross-lane data shifts on the critical chain):
//   * matrix row i lives on lane  i mod (K+1); the window column c lives in register  c mod (K+1).
//     The loop is unrolled K+1 times so every lane/register index is a compile-time constant.
//   * the band streams through registers in 64-column tiles: lane l of tile register d holds band[d][j0+l]
//     (one coalesced 512-B load per diagonal per 64 columns, prefetched one tile ahead); single entries are
//     fetched with v_readlane (wave-uniform lane index); each lane stores the entry it owns (8-B masked store).
//   * one optional right-hand side rides along on lane RHS_LANE: the forward substitution c = L^-1 b is the
//     Cholesky update applied to one more "row", the backward substitution x = L^-T c is the Takahashi
//     mat-vec applied to one more "row" - no extra dependent chain.
//   * T = double, or Dual for the forward-mode tangent (d/d lengthscale) through both recurrences.
#pragma once
#include "asvgp_common.hpp"

namespace asvgp {

constexpr int RHS_LANE = 16;

template <typename T> struct BandPtr;  // read-only band (value [+ tangent]) rows of length M
template <> struct BandPtr<double> {
  const double* v; const double* d;
  __device__ __forceinline__ double load(long off, bool ok) const { return ok ? v[off] : 0.0; }
  // eight consecutive band entries with 16-B loads (off a multiple of 2, v 16-B aligned)
  __device__ __forceinline__ void load8(long off, double (&o)[8]) const {
    const double2* p = reinterpret_cast<const double2*>(v + off);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double2 t = p[q]; o[2 * q] = t.x; o[2 * q + 1] = t.y; }
  }
  __device__ __forceinline__ bool aligned16() const { return (reinterpret_cast<uintptr_t>(v) & 15) == 0; }
};
template <> struct BandPtr<Dual> {
  const double* v; const double* d;
  __device__ __forceinline__ Dual load(long off, bool ok) const { return ok ? Dual{v[off], d[off]} : Dual{0.0, 0.0}; }
  __device__ __forceinline__ void load8(long off, Dual (&o)[8]) const {
    const double2* p = reinterpret_cast<const double2*>(v + off);
    const double2* q2 = reinterpret_cast<const double2*>(d + off);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 t = p[q], u = q2[q];
      o[2 * q] = Dual{t.x, u.x};
      o[2 * q + 1] = Dual{t.y, u.y};
    }
  }
  __device__ __forceinline__ bool aligned16() const { return ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(d)) & 15) == 0; }
};
template <typename T> struct BandOut;
template <> struct BandOut<double> {
  double* v; double* d;
  __device__ __forceinline__ void store(long off, double x) const { v[off] = x; }
};
template <> struct BandOut<Dual> {
  double* v; double* d;
  __device__ __forceinline__ void store(long off, Dual x) const { v[off] = x.v; d[off] = x.d; }
};

// ------------------------------------------------------------------------------------------------
// Cholesky sweep (right-looking, register window).  A: lower band (K+1, M).  L out.  Optional rhs b (M) -> c = L^-1 b.
// info: first failing column + 1 (0 = positive definite).  Must be called by all 64 lanes of one wave.
// ------------------------------------------------------------------------------------------------
template <typename T, int K, bool HAS_RHS>
__device__ void cholesky_sweep(BandPtr<T> A, BandOut<T> L, int M, const double* b, double* c, int* info) {
  using N = Num<T>;
  const int lane = threadIdx.x & 63;
  T cur[K + 1], nxt[K + 1], w[K + 1];
  double bcur = 0.0, bnxt = 0.0;
  int j0 = 0, bad = 0;
#pragma unroll
  for (int d = 0; d <= K; ++d) {
    cur[d] = A.load((long)d * M + lane, lane < M);
    nxt[d] = A.load((long)d * M + 64 + lane, 64 + lane < M);
    w[d] = N::zero();
  }
  if (HAS_RHS) {
    bcur = (lane < M) ? b[lane] : 0.0;
    bnxt = (64 + lane < M) ? b[64 + lane] : 0.0;
  }
  auto tile = [&](int d, int col) -> T {  // band[d][col], col in [j0, j0+128)
    int r = col - j0;
    return (r < 64) ? N::rl(cur[d], r) : N::rl(nxt[d], r - 64);
  };
  auto btile = [&](int col) -> double {
    int r = col - j0;
    return (r < 64) ? readlane_f64(bcur, r) : readlane_f64(bnxt, r - 64);
  };
  // initial window: lane p <- row p, columns 0..p ; RHS lane <- b[0..K]
#pragma unroll
  for (int p = 0; p <= K; ++p)
#pragma unroll
    for (int cc = 0; cc <= p; ++cc) {
      T v = (p < M) ? tile(p - cc, cc) : N::zero();
      w[cc] = N::sel(lane == p, v, w[cc]);
    }
  if (HAS_RHS) {
#pragma unroll
    for (int cc = 0; cc <= K; ++cc) {
      double v = (cc < M) ? btile(cc) : 0.0;
      w[cc] = N::sel(lane == RHS_LANE, N::make(v, 0.0), w[cc]);
    }
  }
  for (int jb = 0; jb < M; jb += K + 1) {
#pragma unroll
    for (int jm = 0; jm <= K; ++jm) {
      const int j = jb + jm;
      if (j < M) {
      const int jj = j - j0;
      T piv = N::rl(w[jm], jm);
      if (!(N::val(piv) > 0.0) && !bad) bad = j + 1;
      T ljj = N::sqrt_(piv);
      T inv = N::inv(ljj);
      T lr = N::sel(lane == jm, ljj, w[jm] * inv);
      T lc[K + 1];
#pragma unroll
      for (int cc = 0; cc <= K; ++cc) {
        lc[cc] = N::rl(lr, (jm + cc) % (K + 1));
        if (j + cc >= M) lc[cc] = N::zero();  // structural zero of the right-padded band
      }
      {  // lane (jm+cc)%(K+1) holds L[j+cc][j]: one masked 8-B store per lane straight into the band
        int ccl = lane - jm;
        if (ccl < 0) ccl += K + 1;
        if (lane <= K) L.store((long)ccl * M + j, (j + ccl < M) ? lr : N::zero());
      }
#pragma unroll
      for (int cc = 1; cc <= K; ++cc) {
        const int q = (jm + cc) % (K + 1);
        w[q] = N::nfma(lr, lc[cc], w[q]);
      }
      if (HAS_RHS && lane == RHS_LANE) c[j] = N::val(lr);
      // lane jm now takes row j+K+1 (columns j+1 .. j+K+1); RHS lane takes b[j+K+1] into slot jm
      const bool have = (j + K + 1 < M);
#pragma unroll
      for (int cc = 0; cc <= K; ++cc) {
        const int q = (jm + 1 + cc) % (K + 1);
        T v = have ? tile(K - cc, j + 1 + cc) : N::zero();
        w[q] = N::sel(lane == jm, v, w[q]);
      }
      if (HAS_RHS) {
        double bv = have ? btile(j + K + 1) : 0.0;
        w[jm] = N::sel(lane == RHS_LANE, N::make(bv, 0.0), w[jm]);
      }
      if (jj == 63) {  // tile finished: advance the prefetch ring
#pragma unroll
        for (int d = 0; d <= K; ++d) {
          cur[d] = nxt[d];
          nxt[d] = A.load((long)d * M + j0 + 128 + lane, j0 + 128 + lane < M);
        }
        if (HAS_RHS) {
          bcur = bnxt;
          bnxt = (j0 + 128 + lane < M) ? b[j0 + 128 + lane] : 0.0;
        }
        j0 += 64;
      }
      }
    }
  }
  if (info && lane == 0) *info = bad;
}

// ------------------------------------------------------------------------------------------------
// Takahashi sweep: S = band((L L^T)^-1) from the lower band of L, backwards.
// Optional rhs c (M) -> x = L^-T c  (so x = (L L^T)^-1 b when c = L^-1 b).
// ------------------------------------------------------------------------------------------------
template <typename T, int K, bool HAS_RHS>
__device__ void takahashi_sweep(BandPtr<T> L, BandOut<T> S, int M, const double* c, double* x) {
  using N = Num<T>;
  const int lane = threadIdx.x & 63;
  T cur[K + 1], nxt[K + 1], s[K + 1], icur, inxt;
  double ccur = 0.0, cnxt = 0.0;
  int j0 = ((M - 1) / 64) * 64;
  auto load_tile = [&](T (&t)[K + 1], T& it, double& ct, int base) {
    const bool ok = (base >= 0) && (base + lane < M);
#pragma unroll
    for (int d = 0; d <= K; ++d) t[d] = L.load((long)d * M + base + lane, ok);
    it = ok ? N::inv(t[0]) : N::zero();
    if (HAS_RHS) ct = ok ? c[base + lane] : 0.0;
  };
  load_tile(cur, icur, ccur, j0);
  load_tile(nxt, inxt, cnxt, j0 - 64);
#pragma unroll
  for (int d = 0; d <= K; ++d) s[d] = N::zero();
  const int top = ((M + K) / (K + 1)) * (K + 1);  // first multiple of K+1 above M-1
  for (int jb = top - (K + 1); jb >= 0; jb -= K + 1) {
#pragma unroll
    for (int jm = K; jm >= 0; --jm) {
      const int j = jb + jm;
      if (j >= M) continue;
      const int jj = j - j0;
      T l[K + 1];
#pragma unroll
      for (int cc = 1; cc <= K; ++cc) l[cc] = (j + cc < M) ? N::rl(cur[cc], jj) : N::zero();  // L[j+cc][j]
      T inv = N::rl(icur, jj);
      T u = N::zero();
#pragma unroll
      for (int cc = 1; cc <= K; ++cc) u = u + s[(jm + cc) % (K + 1)] * l[cc];
      T cj = N::zero();
      if (HAS_RHS) cj = N::sel(lane == RHS_LANE, N::make(readlane_f64(ccur, jj), 0.0), cj);
      T sig = (cj - u) * inv;  // matrix lanes: Sigma[row, j];  RHS lane: x_j
      T sc[K + 1];
      T dot = N::zero();
#pragma unroll
      for (int cc = 1; cc <= K; ++cc) {
        sc[cc] = N::rl(sig, (jm + cc) % (K + 1));
        if (j + cc >= M) sc[cc] = N::zero();
        dot = dot + l[cc] * sc[cc];
      }
      sc[0] = (inv - dot) * inv;  // Sigma[j, j]
      s[jm] = sig;
#pragma unroll
      for (int cc = 0; cc <= K; ++cc) {
        const int q = (jm + cc) % (K + 1);
        s[q] = N::sel(lane == jm, sc[cc], s[q]);  // lane jm becomes row j
      }
      {  // slot jm now holds column j of Sigma: lane (jm+cc)%(K+1) has Sigma[j+cc][j]
        int ccl = lane - jm;
        if (ccl < 0) ccl += K + 1;
        if (lane <= K) S.store((long)ccl * M + j, (j + ccl < M) ? s[jm] : N::zero());
      }
      if (HAS_RHS && lane == RHS_LANE) x[j] = N::val(sig);
      if (jj == 0) {
#pragma unroll
        for (int d = 0; d <= K; ++d) cur[d] = nxt[d];
        icur = inxt;
        if (HAS_RHS) ccur = cnxt;
        j0 -= 64;
        load_tile(nxt, inxt, cnxt, j0 - 64);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Banded triangular solve with one right-hand side column (stride ldb): x = L^-1 b or L^-T b.
// ------------------------------------------------------------------------------------------------
template <int K, bool TRANS>
__device__ void trsv_sweep(const double* Lb, int M, const double* b, double* x, long ld) {
  const int lane = threadIdx.x & 63;
  double cur[K + 1], oth[K + 1], icur, bcur, xout = 0.0, h[K + 1];  // xout: lane jj keeps x[j0+jj]
#pragma unroll
  for (int d = 0; d <= K; ++d) h[d] = 0.0;
  if (!TRANS) {
    // forward: x_i = (b_i - sum_c L[i][i-c] h_c) / L_ii,  L[i][i-c] = band[c][i-c] (may sit in the previous tile)
    for (int j0 = 0; j0 < M; j0 += 64) {
      const bool ok = j0 + lane < M;
#pragma unroll
      for (int d = 0; d <= K; ++d) { oth[d] = (j0 == 0) ? 0.0 : cur[d]; cur[d] = ok ? Lb[(long)d * M + j0 + lane] : 0.0; }
      icur = ok ? 1.0 / cur[0] : 0.0;
      bcur = ok ? b[(long)(j0 + lane) * ld] : 0.0;
      const int n = (M - j0 < 64) ? (M - j0) : 64;
      for (int jj = 0; jj < n; ++jj) {
        double acc = readlane_f64(bcur, jj);
#pragma unroll
        for (int cc = 1; cc <= K; ++cc) {
          int r = jj - cc;  // column i-c relative to the tile
          double lv = (r >= 0) ? readlane_f64(cur[cc], r) : readlane_f64(oth[cc], r + 64);
          acc = fma(-lv, h[cc], acc);
        }
        double xi = acc * readlane_f64(icur, jj);
#pragma unroll
        for (int cc = K; cc >= 2; --cc) h[cc] = h[cc - 1];
        h[1] = xi;
        xout = (lane == jj) ? xi : xout;
      }
      if (ok) x[(long)(j0 + lane) * ld] = xout;
    }
  } else {
    for (int j0 = ((M - 1) / 64) * 64; j0 >= 0; j0 -= 64) {
      const bool ok = j0 + lane < M;
#pragma unroll
      for (int d = 0; d <= K; ++d) cur[d] = ok ? Lb[(long)d * M + j0 + lane] : 0.0;
      icur = ok ? 1.0 / cur[0] : 0.0;
      bcur = ok ? b[(long)(j0 + lane) * ld] : 0.0;
      const int n = (M - j0 < 64) ? (M - j0) : 64;
      for (int jj = n - 1; jj >= 0; --jj) {
        double acc = readlane_f64(bcur, jj);
#pragma unroll
        for (int cc = 1; cc <= K; ++cc) acc = fma(-readlane_f64(cur[cc], jj), h[cc], acc);  // L[i+c][i] = band[c][i]
        double xi = acc * readlane_f64(icur, jj);
#pragma unroll
        for (int cc = K; cc >= 2; --cc) h[cc] = h[cc - 1];
        h[1] = xi;
        xout = (lane == jj) ? xi : xout;
      }
      if (ok) x[(long)(j0 + lane) * ld] = xout;
    }
  }
}

}  // namespace asvgp
