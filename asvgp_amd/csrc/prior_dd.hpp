// Forward pass of the prior (Kuu) chain ON THE GPU in double-double arithmetic: the device counterpart of prior_plan_eval
// (prior_plan.cpp), selected with asvgp_set_prior_forward(handle, 1).  Device code only; launchers in prior_dd.hip and elbo.hip.
//
// Replaces (reference): the factorisation half of gpr.py:56-59 (banded.cholesky_band(Kuu) feeding inverse_from_cholesky_band) - in the
// block-cyclic-reduction order of bcr.hpp, on the O(log M) distinct node classes the planner finds (Kuu is Toeplitz away from the
// boundary).  Why not plain fp64: at the headline size (cond(Kuu) = 3.5e7) an fp64 forward elimination leaves the bound 0.4 - 0.9 away
// from an 80-bit evaluation; the host path therefore runs it in x87 long double (64-bit mantissa).  Here every VALUE is a
// pair of doubles (hi, lo) with error-free transforms (Knuth two-sum, FMA two-product): ~104 bits, more than the x87 format, no host in
// the loop, no dependence on the host's long double.  TANGENTS (d / d lengthscale; the gradient is gated at 1e-6) are plain doubles
// formed from the rounded values, exactly as on the host.  Output: the table prior_plan_eval writes (prior_plan.hpp), consumed by
// bcr_backward_pre / bcr_mfma_backward_pre.
//
// Work split.  The classes of a level are independent: wave w takes classes w, w + NW, ...; inside a class lane (r, c) owns entry (r, c)
// of the B x B blocks: Cholesky column by column (every lane of column j recomputes the pivot itself, so a column is one LDS round
// trip), three triangular solves on 3 B lanes (U_a, U_b, L^-1: one column each), then the six B x B products of a node spread over
// 64 / B^2 lane groups.  A class lives in ONE wave: its phases are ordered by wave-level fences (the LDS serves a wave's requests in
// order), workgroup barriers only separate the levels.  Latency-bound: ~10 levels x ~450 dependent fp64 operations.
// Everything below is written with FMA contraction OFF (an error-free transform must not be fused or re-associated); the FMAs it needs
// are spelled __builtin_fma.
#pragma once
#include "prior_plan.hpp"

namespace asvgp {

#pragma clang fp contract(off)

constexpr int PD_MC = PRIOR_MAX_CLASSES;

struct pdd { double hi, lo; };
__device__ __forceinline__ pdd pd_two_sum(double a, double b) {
#pragma clang fp contract(off)
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ pdd pd_fast_two_sum(double a, double b) {
#pragma clang fp contract(off)
  const double s = a + b;
  return {s, b - (s - a)};
}
__device__ __forceinline__ pdd pd_two_prod(double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  return {p, __builtin_fma(a, b, -p)};
}
// a + b with an error of ~2^-104 (|a| + |b|): the x87 pass this replaces rounds at 2^-64 of the operands
__device__ __forceinline__ pdd pd_add(pdd a, pdd b) {
#pragma clang fp contract(off)
  pdd s = pd_two_sum(a.hi, b.hi);
  s.lo += a.lo + b.lo;
  return pd_fast_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ pdd pd_sub(pdd a, pdd b) { return pd_add(a, pdd{-b.hi, -b.lo}); }
__device__ __forceinline__ pdd pd_mul(pdd a, pdd b) {
#pragma clang fp contract(off)
  pdd p = pd_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return pd_fast_two_sum(p.hi, p.lo);
}
// 1 / sqrt(s) and sqrt(s): v_rsq_f64, two fp64 Newton steps (-> 2^-52), ONE double-double Newton step y (1 + (1 - s y^2) / 2) (-> 2^-104),
// root = s * inverse.  No division and no sqrt instruction sequence (4 + 1 of them in a long-division formulation: ~175 dependent
// instructions per pivot).
__device__ __forceinline__ void pd_rsqrt(pdd s, pdd& inv, pdd& root) {
#pragma clang fp contract(off)
  double y = __builtin_amdgcn_rsq(s.hi);
  const double h = 0.5 * s.hi;
  double e = __builtin_fma(-h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  const pdd r = pd_sub(pdd{1.0, 0.0}, pd_mul(s, pd_two_prod(y, y)));
  inv = pd_fast_two_sum(y, y * r.hi * 0.5);
  root = pd_mul(s, inv);
}

struct PdCoefs { double c[ASVGP_MAX_KUU_TERMS]; double dc[ASVGP_MAX_KUU_TERMS]; };

// entry e of a block stored as (hi, lo, tangent) triples
struct PdEnt { double hi, lo, d; };
__device__ __forceinline__ PdEnt pd_ld(const double* m, int e) { return {m[3 * e], m[3 * e + 1], m[3 * e + 2]}; }
__device__ __forceinline__ void pd_st(double* m, int e, PdEnt v) { m[3 * e] = v.hi; m[3 * e + 1] = v.lo; m[3 * e + 2] = v.d; }
__device__ __forceinline__ pdd pd_ld2(const double* m, int e) { return {m[3 * e], m[3 * e + 1]}; }

template <int B> __host__ __device__ constexpr int pd_waves() { return B <= 4 ? 8 : 4; }
template <int B> __host__ __device__ constexpr size_t pd_lds_doubles(int n_rec, int n_img_lds) {
  return (size_t)6 * PD_MC * B * B * 3 + (size_t)pd_waves<B>() * (4 * B * B * 3 + 3 * B) + 4 * 16 + 8 + 2 * ((n_rec + 1) & ~1) + (n_img_lds + 1) / 2;
}

// phases of ONE wave's class: its LDS writes before, its LDS reads after (a wave's LDS requests are served in order)
__device__ __forceinline__ void pd_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cholesky of the B x B block at L (lower triangle in, factor out) by the lanes (r, c = j) of the calling wave, value in double-double,
// tangent in double from the rounded factor (prior_plan.cpp `chol`).  inv = [B] x (hi, lo) of 1 / diag, dinv = [B] tangents.
template <int B>
__device__ __forceinline__ void pd_chol(double* L, double* inv, double* dinv, int r, int c, int col0, unsigned long long order,
                                        unsigned long long* bad_key) {
#pragma clang fp contract(off)
#pragma unroll
  for (int j = 0; j < B; ++j) {
    if (c == j && r >= j && r < B) {
      pdd s = pd_ld2(L, j * B + j);                                // the pivot, recomputed by every lane of the column
      double ds = L[3 * (j * B + j) + 2];
      pdd t = pd_ld2(L, r * B + j);
      double dt = L[3 * (r * B + j) + 2];
#pragma unroll
      for (int q = 0; q < j; ++q) {
        const PdEnt ljq = pd_ld(L, j * B + q), lrq = pd_ld(L, r * B + q);
        s = pd_sub(s, pd_mul(pdd{ljq.hi, ljq.lo}, pdd{ljq.hi, ljq.lo}));
        ds -= 2.0 * ljq.d * ljq.hi;
        t = pd_sub(t, pd_mul(pdd{lrq.hi, lrq.lo}, pdd{ljq.hi, ljq.lo}));
        dt -= lrq.d * ljq.hi + lrq.hi * ljq.d;
      }
      const bool pos = s.hi > 0.0;
      if (!pos && r == j) atomicMin(bad_key, (order << 32) | (unsigned long long)(unsigned)(col0 + j + 1));
      pdd l, li;
      pd_rsqrt(pos ? s : pdd{1.0, 0.0}, li, l);
      const double invd = li.hi, dl = 0.5 * ds * invd;
      if (r == j) {
        pd_st(L, j * B + j, PdEnt{l.hi, l.lo, dl});
        inv[2 * j] = li.hi; inv[2 * j + 1] = li.lo;
        dinv[j] = -dl * invd * invd;
      } else {
        const pdd v = pd_mul(t, li);
        pd_st(L, r * B + j, PdEnt{v.hi, v.lo, (dt - v.hi * dl) * invd});
      }
    }
    pd_wave_sync();
  }
}

// X <- L^-1 X for ONE column held in registers (value double-double, tangent double): prior_plan.cpp `solveL`
template <int B>
__device__ __forceinline__ void pd_solve_col(const double* L, const double* inv, pdd (&X)[B], double (&Xd)[B]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < B; ++i) {
    pdd t = X[i];
#pragma unroll
    for (int q = 0; q < i; ++q) t = pd_sub(t, pd_mul(pd_ld2(L, i * B + q), X[q]));
    X[i] = pd_mul(t, pdd{inv[2 * i], inv[2 * i + 1]});
  }
#pragma unroll
  for (int i = 0; i < B; ++i) {
    double dt = Xd[i];
#pragma unroll
    for (int q = 0; q < i; ++q) dt -= L[3 * (i * B + q) + 2] * X[q].hi + L[3 * (i * B + q)] * Xd[q];
    Xd[i] = (dt - L[3 * (i * B + i) + 2] * X[i].hi) * inv[2 * i];
  }
}

// sum_q X[q][pr] * Y[q][pc] (value double-double, pairwise; tangent double)
template <int B>
__device__ __forceinline__ void pd_dot(const double* X, const double* Y, int pr, int pc, pdd& acc, double& dacc) {
#pragma clang fp contract(off)
  pdd term[B];
  dacc = 0.0;
#pragma unroll
  for (int q = 0; q < B; ++q) {
    const PdEnt x = pd_ld(X, q * B + pr), y = pd_ld(Y, q * B + pc);
    term[q] = pd_mul(pdd{x.hi, x.lo}, pdd{y.hi, y.lo});
    dacc += x.d * y.hi + x.hi * y.d;
  }
#pragma unroll
  for (int w = 1; w < B; w *= 2)
#pragma unroll
    for (int q = 0; q + w < B; q += 2 * w) term[q] = pd_add(term[q], term[q + w]);
  acc = term[0];
}

// The forward pass.  Called by ALL `nthreads` threads of a workgroup (multiple of 64, at least 64 * pd_waves<B>()); lds: pd_lds_doubles<B>(n_rec, n_img_lds).
// img / stat: the plan's device image (prior_plan_image); tab: the table (global memory).  Ends with a workgroup barrier; the table's
// stores are NOT yet fenced for other workgroups.
template <int B>
__device__ __attribute__((always_inline)) void prior_forward_dd(const int* __restrict__ img_g, int n_img_lds, const double* __restrict__ stat, const PdCoefs& cf,
                                                                double* __restrict__ tab, double* lds, int tid, int nthreads,
                                                                unsigned long long* stamps = nullptr) {
#pragma clang fp contract(off)
  int n_stamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) {               // diagnostics (tools/prior_dd_probe.py): cycle stamps of thread 0
    if (stamps && tid == 0 && n_stamp < 60) stamps[n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  constexpr int BB = B * B, NW = pd_waves<B>(), W = prior_rec_fields(B), BLK = BB * 3;
  constexpr int WS = 4 * BLK + 3 * B;                              // per-wave scratch: L, U_a, U_b, L^-1 | inv (hi, lo) | dinv
  double* Dv = lds;
  double* Dn = Dv + PD_MC * BLK;
  double* Ev = Dn + PD_MC * BLK;
  double* En = Ev + PD_MC * BLK;
  double* uA = En + PD_MC * BLK;
  double* uB = uA + PD_MC * BLK;
  double* wsc = uB + PD_MC * BLK;
  double* red = wsc + NW * WS;                                      // [NW] x (logdet hi, lo, dlogdet, pad)
  unsigned long long* bad_key = reinterpret_cast<unsigned long long*>(red + 4 * 16);
  double* lprod = red + 4 * 16 + 8 + (n_img_lds + 1) / 2;           // [n_rec] product of the record's pivots, [n_rec] node count (log-det weights)
  // the plan's class maps are read level by level, one dependent round trip each: from the LDS when they fit (n_img_lds = their length)
  const int* img = img_g;
  if (n_img_lds > 0) {
    int* il = reinterpret_cast<int*>(red + 4 * 16 + 8);
    for (int i = tid; i < n_img_lds; i += nthreads) il[i] = img_g[i];
    img = il;
    __syncthreads();
  }
  stamp();
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane >> 3, c = lane & 7;
  const bool act = r < B && c < B;
  const int e = r * B + c;
  const int nt = img[1], levels = img[2], n_rec = img[3], nd0 = img[4], ne0 = img[5];
  const int n_rec_e = (n_rec + 1) & ~1;
  double* val = tab + PRIOR_TAB_HEADER;
  double* tan = val + (size_t)n_rec * W;
  double* L = wsc + (w < NW ? w : 0) * WS;
  double* Ua = L + BLK;
  double* Ub = Ua + BLK;
  double* Li = Ub + BLK;
  double* inv = Li + BLK;
  double* dinv = inv + 2 * B;
  if (tid == 0) *bad_key = ~0ull;
  // ---- level-0 blocks of the class representatives, with the reference's rounding sequence (inducing_features.py:16-44)
  for (int idx = tid; idx < (nd0 + ne0) * BB; idx += nthreads) {
    const bool isE = idx >= nd0 * BB;
    const int q = (isE ? idx - nd0 * BB : idx) / BB, ee = (isE ? idx - nd0 * BB : idx) % BB;
    int rr = ee / B, cc = ee % B;
    if (!isE && cc > rr) { const int t = rr; rr = cc; cc = t; }  // D is symmetric: the stored entry is the lower one
    const int src = (isE ? nd0 * BB : 0) + q * BB + rr * B + cc;
    const int code = img[img[8] + src];
    double kv = code == 1 ? 1.0 : 0.0, dv = 0.0;
    if (code == 0) {
      const double* sp = stat + (size_t)src * nt;
      double acc = cf.c[0] * sp[0], dacc = cf.dc[0] * sp[0];
      for (int t = 1; t < nt; ++t) {
        const double a = cf.c[t] * sp[t], b = cf.dc[t] * sp[t];
        acc = acc + a;
        dacc = dacc + b;
      }
      kv = acc; dv = dacc;
    }
    pd_st((isE ? Ev : Dv) + q * BLK, ee, PdEnt{kv, 0.0, dv});
  }
  __syncthreads();
  stamp();
  double dlogdet = 0.0;
  for (int l = 0; l < levels; ++l) {
    const int* h = img + img[6] + 8 * l;
    const int nq = h[0], ndn = h[1], rec0 = h[5];
    const int* node_in = img + h[2];
    const int* node_rep = img + h[3];
    const int* d_next = img + h[4];
    if (w < NW)
      for (int q = w; q < nq; q += NW) {                            // (wave-uniform: no workgroup barrier inside)
        const int in0 = node_in[4 * q], in1 = node_in[4 * q + 1], in2 = node_in[4 * q + 2], count = node_in[4 * q + 3];
        const bool hasb = in2 >= 0;
        if (act) {
          PdEnt dv = pd_ld(Dv + in0 * BLK, e);
          if (c > r) dv = PdEnt{0.0, 0.0, 0.0};                     // (the factor's upper triangle)
          pd_st(L, e, dv);
          pd_st(Ua, e, pd_ld(Ev + in1 * BLK, e));                   // A[i, a] = E(a)
          pd_st(Ub, e, hasb ? pd_ld(Ev + in2 * BLK, c * B + r) : PdEnt{0.0, 0.0, 0.0});   // A[i, b] = E(i)^T
        }
        pd_wave_sync();
        if (l == 1) stamp();
        pd_chol<B>(L, inv, dinv, r, c, node_rep[q] * B, (unsigned long long)(l * 64 + q), bad_key);
        if (l == 1) stamp();
        // ---- U_a = L^-1 A[i,a], U_b = L^-1 A[i,b], L^-1: one column per lane (rows r = 0, 1, 2 of the lane grid)
        if (c < B && r < 3) {
          pdd X[B];
          double Xd[B];
          double* src = r == 0 ? Ua : (r == 1 ? Ub : Li);
#pragma unroll
          for (int i = 0; i < B; ++i) {
            if (r < 2) { const PdEnt v = pd_ld(src, i * B + c); X[i] = pdd{v.hi, v.lo}; Xd[i] = v.d; }
            else { X[i] = pdd{i == c ? 1.0 : 0.0, 0.0}; Xd[i] = 0.0; }
          }
          pd_solve_col<B>(L, inv, X, Xd);
#pragma unroll
          for (int i = 0; i < B; ++i) pd_st(src, i * B + c, PdEnt{X[i].hi, X[i].lo, Xd[i]});
        }
        pd_wave_sync();
        if (l == 1) stamp();
        // ---- the node's record and the Schur updates.  Products out = sign * X^T Y over lane groups of B^2
        const int rec = rec0 + q;
        if (act) {
          const PdEnt lv = pd_ld(L, e), ua = pd_ld(Ua, e), ub = pd_ld(Ub, e);
          val[(size_t)rec * W + prior_f_L(B) + e] = lv.hi;  tan[(size_t)rec * W + prior_f_L(B) + e] = lv.d;
          val[(size_t)rec * W + prior_f_UA(B) + e] = ua.hi; tan[(size_t)rec * W + prior_f_UA(B) + e] = ua.d;
          val[(size_t)rec * W + prior_f_UB(B) + e] = ub.hi; tan[(size_t)rec * W + prior_f_UB(B) + e] = ub.d;
          if (r == 0) { val[(size_t)rec * W + prior_f_I(B) + c] = inv[2 * c]; tan[(size_t)rec * W + prior_f_I(B) + c] = dinv[c]; }
        }
        {
          constexpr int NSLOT = 64 / BB > 6 ? 6 : 64 / BB;
          const int slot = lane / BB, pe = lane % BB, pr = pe / B, pc = pe % B;
          if (slot < NSLOT) {
            for (int p = slot; p < 6; p += NSLOT) {
              const double* X = (p == 0 || p == 3) ? Ua : (p == 2 ? Li : Ub);
              const double* Y = p <= 2 ? Li : (p == 4 ? Ub : Ua);
              pdd acc = {0.0, 0.0};
              double dacc = 0.0;
              if (hasb || p == 0 || p == 2 || p == 3) pd_dot<B>(X, Y, pr, pc, acc, dacc);
              if (p == 5) { acc.hi = -acc.hi; acc.lo = -acc.lo; dacc = -dacc; }   // A'[b, a] = -U_b^T U_a
              if (p <= 2) {
                const int f = p == 0 ? prior_f_GAT(B) : (p == 1 ? prior_f_GBT(B) : prior_f_DINV(B));
                val[(size_t)rec * W + f + pe] = acc.hi; tan[(size_t)rec * W + f + pe] = dacc;
              } else {
                pd_st((p == 3 ? uA : (p == 4 ? uB : En)) + q * BLK, pe, PdEnt{acc.hi, acc.lo, dacc});
              }
            }
          }
        }
        if (lane == 0) {
          double prod = 1.0, dsum = 0.0;
#pragma unroll
          for (int i = 0; i < B; ++i) { prod *= L[3 * (i * B + i)]; dsum += L[3 * (i * B + i) + 2] * inv[2 * i]; }
          lprod[rec] = prod; lprod[n_rec_e + rec] = (double)count;    // (the logarithms are taken by parallel lanes at the end)
          dlogdet += (double)count * 2.0 * dsum;
        }
        pd_wave_sync();                                             // (the scratch is reused by the wave's next class)
        if (l == 1) stamp();
      }
    stamp();
    __syncthreads();
    stamp();
    // ---- diagonal blocks of the next level: D - (update from the node on the left) - (update from the node on the right)
    for (int idx = tid; idx < ndn * BB; idx += nthreads) {
      const int q = idx / BB, ee = idx % BB;
      const int i0 = d_next[3 * q], i1 = d_next[3 * q + 1], i2 = d_next[3 * q + 2];
      PdEnt dv = pd_ld(Dv + i0 * BLK, ee);
      pdd v = {dv.hi, dv.lo};
      if (i1 >= 0) { const PdEnt u = pd_ld(uB + i1 * BLK, ee); v = pd_sub(v, pdd{u.hi, u.lo}); dv.d -= u.d; }
      if (i2 >= 0) { const PdEnt u = pd_ld(uA + i2 * BLK, ee); v = pd_sub(v, pdd{u.hi, u.lo}); dv.d -= u.d; }
      pd_st(Dn + q * BLK, ee, PdEnt{v.hi, v.lo, dv.d});
    }
    __syncthreads();
    stamp();
    { double* t = Dv; Dv = Dn; Dn = t; t = Ev; Ev = En; En = t; }
  }
  // ---- root: L_0 and Sigma_00 = D_0^-1 (in the U_a slot)
  if (w == 0) {
    if (act) { PdEnt dv = pd_ld(Dv, e); if (c > r) dv = PdEnt{0.0, 0.0, 0.0}; pd_st(L, e, dv); }
    pd_wave_sync();
    pd_chol<B>(L, inv, dinv, r, c, 0, (unsigned long long)(levels * 64), bad_key);
    const int rec = n_rec - 1;
    if (r == 0 && c < B) {
      pdd X[B];
      double Xd[B];
#pragma unroll
      for (int i = 0; i < B; ++i) { X[i] = pdd{i == c ? 1.0 : 0.0, 0.0}; Xd[i] = 0.0; }
      pd_solve_col<B>(L, inv, X, Xd);
#pragma unroll
      for (int i = B - 1; i >= 0; --i) {                            // X <- L^-T X
        pdd t = X[i];
#pragma unroll
        for (int q = i + 1; q < B; ++q) t = pd_sub(t, pd_mul(pd_ld2(L, q * B + i), X[q]));
        X[i] = pd_mul(t, pdd{inv[2 * i], inv[2 * i + 1]});
      }
#pragma unroll
      for (int i = B - 1; i >= 0; --i) {
        double dt = Xd[i];
#pragma unroll
        for (int q = i + 1; q < B; ++q) dt -= L[3 * (q * B + i) + 2] * X[q].hi + L[3 * (q * B + i)] * Xd[q];
        Xd[i] = (dt - L[3 * (i * B + i) + 2] * X[i].hi) * inv[2 * i];
      }
#pragma unroll
      for (int i = 0; i < B; ++i) {
        val[(size_t)rec * W + prior_f_UA(B) + i * B + c] = X[i].hi; tan[(size_t)rec * W + prior_f_UA(B) + i * B + c] = Xd[i];
        val[(size_t)rec * W + prior_f_UB(B) + i * B + c] = 0.0;      tan[(size_t)rec * W + prior_f_UB(B) + i * B + c] = 0.0;
      }
    }
    if (act) {
      const PdEnt lv = pd_ld(L, e);
      val[(size_t)rec * W + prior_f_L(B) + e] = lv.hi; tan[(size_t)rec * W + prior_f_L(B) + e] = lv.d;
      if (r == 0) { val[(size_t)rec * W + prior_f_I(B) + c] = inv[2 * c]; tan[(size_t)rec * W + prior_f_I(B) + c] = dinv[c]; }
    }
    if (lane == 0) {
      double prod = 1.0, dsum = 0.0;
#pragma unroll
      for (int i = 0; i < B; ++i) { prod *= L[3 * (i * B + i)]; dsum += L[3 * (i * B + i) + 2] * inv[2 * i]; }
      lprod[rec] = prod; lprod[n_rec_e + rec] = 1.0;
      dlogdet += 2.0 * dsum;
    }
  }
  if (lane == 0 && w < NW) red[4 * w + 2] = dlogdet;
  stamp();
  __syncthreads();
  // log|Kuu| = sum over the records of 2 count log(prod of the pivots): log in fp64 (|log| ~ 10 at 1e-16, as on the host), sum in double-double
  if (w == 0) {
    pdd part = {0.0, 0.0};
    for (int i = lane; i < n_rec; i += 64) part = pd_add(part, pd_two_prod(2.0 * lprod[n_rec_e + i], log(lprod[i])));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part = pd_add(part, pdd{__shfl_xor(part.hi, m, 64), __shfl_xor(part.lo, m, 64)});
    if (lane == 0) { red[0] = part.hi; red[1] = part.lo; }
  }
  __syncthreads();
  if (tid == 0) {
    const pdd ls = {red[0], red[1]};
    double dls = 0.0;
    for (int i = 0; i < NW; ++i) dls += red[4 * i + 2];
    const unsigned long long bk = *bad_key;
    tab[0] = ls.hi; tab[1] = dls; tab[2] = bk == ~0ull ? 0.0 : (double)(unsigned)(bk & 0xffffffffull); tab[3] = (double)n_rec;
    tab[4] = tab[5] = tab[6] = tab[7] = 0.0;
  }
  __syncthreads();
  stamp();
  if (stamps && tid == 0) stamps[63] = (unsigned long long)n_stamp;
}

#pragma clang fp contract(fast)

}  // namespace asvgp
