// Forward pass of the prior (Kuu) chain ON THE GPU in double-double arithmetic: the device counterpart of prior_plan_eval
// (prior_plan.cpp), selected with asvgp_set_prior_forward(handle, 1).
//
// Replaces (reference): the factorisation half of gpr.py:56-59 (banded.cholesky_band(Kuu) feeding inverse_from_cholesky_band) - in the
// block-cyclic-reduction order of bcr.hpp, on the O(log M) distinct node classes the planner finds (Kuu is Toeplitz away from the
// boundary).  Why not plain fp64: at the headline size (cond(Kuu) = 3.5e7) an fp64 forward elimination leaves the bound 0.4 - 0.9 away
// from an 80-bit evaluation; the host path therefore runs it in x87 long double (64-bit mantissa).  Here every VALUE is a
// pair of doubles (hi, lo) with error-free transforms (Dekker / Knuth two-sum, FMA two-product): ~106 bits, more than the x87 format, no
// host in the loop, no dependence on the host's long double.  TANGENTS (d / d lengthscale; the gradient is gated at 1e-6) are plain
// doubles formed from the rounded values, exactly as on the host.  Output: the same table prior_plan_eval writes (prior_plan.hpp), in
// device memory, consumed by bcr_backward_pre / bcr_mfma_backward_pre.
//
// One workgroup.  The classes of a level are independent: wave w takes classes w, w + NW, ...; inside a class lane (r, c) owns entry
// (r, c) of the B x B blocks (Cholesky column by column - every lane of column j recomputes the pivot itself, so a column costs one
// barrier -, three triangular solves on 3 B lanes (U_a, U_b, L^-1: one column each), then the six B x B products of a node spread over
// 64 / B^2 lane groups).  Latency-bound: ~9 levels x ~1 600 dependent fp64 operations.
// This translation unit is compiled with -ffp-contract=off (asvgp_amd/build.py): an error-free transform must not be re-associated or
// fused; the FMAs it needs are written as __builtin_fma.
#include <hip/hip_runtime.h>

#include <vector>

#include "handle.hpp"

namespace asvgp {

namespace {

constexpr int PD_MC = PRIOR_MAX_CLASSES;

struct dd { double hi, lo; };
__device__ __forceinline__ dd two_sum(double a, double b) { const double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)}; }
__device__ __forceinline__ dd fast_two_sum(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
__device__ __forceinline__ dd two_prod(double a, double b) { const double p = a * b; return {p, __builtin_fma(a, b, -p)}; }
__device__ __forceinline__ dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = fast_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return fast_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_sub(dd a, dd b) { return dd_add(a, dd{-b.hi, -b.lo}); }
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return fast_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
  dd p = two_prod(a.hi, b);
  p.lo += a.lo * b;
  return fast_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_div(dd a, dd b) {   // three quotient digits (long division on the leading doubles)
  const double q1 = a.hi / b.hi;
  dd r = dd_sub(a, dd_mul_d(b, q1));
  const double q2 = r.hi / b.hi;
  r = dd_sub(r, dd_mul_d(b, q2));
  const double q3 = r.hi / b.hi;
  return dd_add(fast_two_sum(q1, q2), dd{q3, 0.0});
}
__device__ __forceinline__ dd dd_sqrt(dd a) {        // Karp / Markstein: sqrt(a) = a x + (a - (a x)^2) x / 2 with x ~ 1 / sqrt(a)
  const double x = 1.0 / sqrt(a.hi), ax = a.hi * x;
  const dd diff = dd_sub(a, two_prod(ax, ax));
  return fast_two_sum(ax, diff.hi * (x * 0.5));
}

struct PdCoefs { double c[ASVGP_MAX_KUU_TERMS]; double dc[ASVGP_MAX_KUU_TERMS]; };

// entry e of a block stored as (hi, lo, tangent) triples
struct Ent { double hi, lo, d; };
__device__ __forceinline__ Ent ld_ent(const double* m, int e) { return {m[3 * e], m[3 * e + 1], m[3 * e + 2]}; }
__device__ __forceinline__ void st_ent(double* m, int e, Ent v) { m[3 * e] = v.hi; m[3 * e + 1] = v.lo; m[3 * e + 2] = v.d; }
__device__ __forceinline__ dd ld_dd(const double* m, int e) { return {m[3 * e], m[3 * e + 1]}; }

template <int B> __host__ __device__ constexpr int pd_waves() { return B <= 4 ? 8 : 4; }
template <int B> __host__ __device__ constexpr size_t pd_lds_doubles() {
  return (size_t)6 * PD_MC * B * B * 3 + (size_t)pd_waves<B>() * (4 * B * B * 3 + 3 * B) + 4 * 16 + 8;
}

// Cholesky of the B x B block at L (lower triangle in, factor out; upper triangle untouched) by the lanes (r, c = j) of ONE wave, value in
// double-double, tangent in double from the rounded factor (prior_plan.cpp `chol`).  inv = [B] x (hi, lo) of 1 / diag, dinv = [B] tangents.
// All threads of the workgroup call this (barriers inside); only `live` lanes work.
template <int B>
__device__ __forceinline__ void pd_chol(double* L, double* inv, double* dinv, bool live, int r, int c, int col0, unsigned long long order,
                                        unsigned long long* bad_key) {
#pragma unroll
  for (int j = 0; j < B; ++j) {
    if (live && c == j && r >= j && r < B) {
      dd s = ld_dd(L, j * B + j);                                  // the pivot, recomputed by every lane of the column
      double ds = L[3 * (j * B + j) + 2];
      dd t = ld_dd(L, r * B + j);
      double dt = L[3 * (r * B + j) + 2];
#pragma unroll
      for (int q = 0; q < j; ++q) {
        const Ent ljq = ld_ent(L, j * B + q), lrq = ld_ent(L, r * B + q);
        s = dd_sub(s, dd_mul(dd{ljq.hi, ljq.lo}, dd{ljq.hi, ljq.lo}));
        ds -= 2.0 * ljq.d * ljq.hi;
        t = dd_sub(t, dd_mul(dd{lrq.hi, lrq.lo}, dd{ljq.hi, ljq.lo}));
        dt -= lrq.d * ljq.hi + lrq.hi * ljq.d;
      }
      const bool pos = s.hi > 0.0;
      if (!pos && r == j) atomicMin(bad_key, (order << 32) | (unsigned long long)(unsigned)(col0 + j + 1));
      const dd l = dd_sqrt(pos ? s : dd{1.0, 0.0});
      const dd li = dd_div(dd{1.0, 0.0}, l);
      const double invd = li.hi, dl = 0.5 * ds * invd;
      if (r == j) {
        st_ent(L, j * B + j, Ent{l.hi, l.lo, dl});
        inv[2 * j] = li.hi; inv[2 * j + 1] = li.lo;
        dinv[j] = -dl * invd * invd;
      } else {
        const dd v = dd_mul(t, li);
        st_ent(L, r * B + j, Ent{v.hi, v.lo, (dt - v.hi * dl) * invd});
      }
    }
    __syncthreads();
  }
}

// X <- L^-1 X for ONE column held in registers (value double-double, tangent double): prior_plan.cpp `solveL`
template <int B>
__device__ __forceinline__ void pd_solve_col(const double* L, const double* inv, dd (&X)[B], double (&Xd)[B]) {
#pragma unroll
  for (int i = 0; i < B; ++i) {
    dd t = X[i];
#pragma unroll
    for (int q = 0; q < i; ++q) t = dd_sub(t, dd_mul(ld_dd(L, i * B + q), X[q]));
    X[i] = dd_mul(t, dd{inv[2 * i], inv[2 * i + 1]});
  }
#pragma unroll
  for (int i = 0; i < B; ++i) {
    double dt = Xd[i];
#pragma unroll
    for (int q = 0; q < i; ++q) dt -= L[3 * (i * B + q) + 2] * X[q].hi + L[3 * (i * B + q)] * Xd[q];
    Xd[i] = (dt - L[3 * (i * B + i) + 2] * X[i].hi) * inv[2 * i];
  }
}

template <int B>
__global__ __launch_bounds__(64 * pd_waves<B>()) void prior_forward_dd_kernel(const int* __restrict__ img, const double* __restrict__ stat,
                                                                              PdCoefs cf, double* __restrict__ tab) {
  extern __shared__ double lds[];
  constexpr int BB = B * B, NW = pd_waves<B>(), NT = 64 * NW, W = prior_rec_fields(B), BLK = BB * 3;
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
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane >> 3, c = lane & 7;
  const bool act = r < B && c < B;
  const int e = r * B + c;
  const int nt = img[1], levels = img[2], n_rec = img[3], nd0 = img[4], ne0 = img[5];
  double* val = tab + PRIOR_TAB_HEADER;
  double* tan = val + (size_t)n_rec * W;
  double* L = wsc + w * WS;
  double* Ua = L + BLK;
  double* Ub = Ua + BLK;
  double* Li = Ub + BLK;
  double* inv = Li + BLK;
  double* dinv = inv + 2 * B;
  if (tid == 0) *bad_key = ~0ull;
  // ---- level-0 blocks of the class representatives, with the reference's rounding sequence (inducing_features.py:16-44)
  for (int idx = tid; idx < (nd0 + ne0) * BB; idx += NT) {
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
    st_ent((isE ? Ev : Dv) + q * BLK, ee, Ent{kv, 0.0, dv});
  }
  __syncthreads();
  dd logdet = {0.0, 0.0};
  double dlogdet = 0.0;
  for (int l = 0; l < levels; ++l) {
    const int* h = img + img[6] + 8 * l;
    const int nq = h[0], ndn = h[1], rec0 = h[5];
    const int* node_in = img + h[2];
    const int* node_rep = img + h[3];
    const int* d_next = img + h[4];
    for (int q0 = 0; q0 < nq; q0 += NW) {
      const int q = q0 + w;
      const bool live = q < nq;
      const int in0 = live ? node_in[4 * q] : 0, in1 = live ? node_in[4 * q + 1] : 0, in2 = live ? node_in[4 * q + 2] : -1;
      const int count = live ? node_in[4 * q + 3] : 0;
      const bool hasb = in2 >= 0;
      if (live && act) {
        st_ent(L, e, ld_ent(Dv + in0 * BLK, e));
        st_ent(Ua, e, ld_ent(Ev + in1 * BLK, e));                   // A[i, a] = E(a)
        st_ent(Ub, e, hasb ? ld_ent(Ev + in2 * BLK, c * B + r) : Ent{0.0, 0.0, 0.0});   // A[i, b] = E(i)^T
      }
      __syncthreads();
      pd_chol<B>(L, inv, dinv, live, r, c, live ? node_rep[q] * B : 0, (unsigned long long)(l * 64 + (live ? q : 0)), bad_key);
      if (live && act && c > r) st_ent(L, e, Ent{0.0, 0.0, 0.0});
      __syncthreads();
      // ---- U_a = L^-1 A[i,a], U_b = L^-1 A[i,b], L^-1: one column per lane (rows r = 0, 1, 2 of the lane grid)
      if (live && c < B && r < 3) {
        dd X[B];
        double Xd[B];
        double* src = r == 0 ? Ua : (r == 1 ? Ub : Li);
#pragma unroll
        for (int i = 0; i < B; ++i) {
          if (r < 2) { const Ent v = ld_ent(src, i * B + c); X[i] = dd{v.hi, v.lo}; Xd[i] = v.d; }
          else { X[i] = dd{i == c ? 1.0 : 0.0, 0.0}; Xd[i] = 0.0; }
        }
        pd_solve_col<B>(L, inv, X, Xd);
#pragma unroll
        for (int i = 0; i < B; ++i) st_ent(src, i * B + c, Ent{X[i].hi, X[i].lo, Xd[i]});
      }
      __syncthreads();
      // ---- the node's record and the Schur updates.  Products out = sign * X^T Y over lane groups of B^2
      const int rec = rec0 + q;
      if (live && act) {
        const Ent lv = ld_ent(L, e), ua = ld_ent(Ua, e), ub = ld_ent(Ub, e);
        val[(size_t)rec * W + prior_f_L(B) + e] = lv.hi;  tan[(size_t)rec * W + prior_f_L(B) + e] = lv.d;
        val[(size_t)rec * W + prior_f_UA(B) + e] = ua.hi; tan[(size_t)rec * W + prior_f_UA(B) + e] = ua.d;
        val[(size_t)rec * W + prior_f_UB(B) + e] = ub.hi; tan[(size_t)rec * W + prior_f_UB(B) + e] = ub.d;
        if (r == 0) { val[(size_t)rec * W + prior_f_I(B) + c] = inv[2 * c]; tan[(size_t)rec * W + prior_f_I(B) + c] = dinv[c]; }
      }
      {
        constexpr int NSLOT = 64 / BB > 6 ? 6 : 64 / BB;
        const int slot = lane / BB, pe = lane % BB, pr = pe / B, pc = pe % B;
        if (live && slot < NSLOT) {
          for (int p = slot; p < 6; p += NSLOT) {
            const double* X = (p == 0 || p == 3) ? Ua : (p == 2 ? Li : Ub);
            const double* Y = p <= 2 ? Li : (p == 4 ? Ub : Ua);
            dd acc = {0.0, 0.0};
            double dacc = 0.0;
            if (hasb || p == 0 || p == 2 || p == 3) {
#pragma unroll
              for (int q2 = 0; q2 < B; ++q2) {
                const Ent x = ld_ent(X, q2 * B + pr), y = ld_ent(Y, q2 * B + pc);
                acc = dd_add(acc, dd_mul(dd{x.hi, x.lo}, dd{y.hi, y.lo}));
                dacc += x.d * y.hi + x.hi * y.d;
              }
            }
            if (p == 5) { acc.hi = -acc.hi; acc.lo = -acc.lo; dacc = -dacc; }   // A'[b, a] = -U_b^T U_a
            if (p <= 2) {
              const int f = p == 0 ? prior_f_GAT(B) : (p == 1 ? prior_f_GBT(B) : prior_f_DINV(B));
              val[(size_t)rec * W + f + pe] = acc.hi; tan[(size_t)rec * W + f + pe] = dacc;
            } else {
              st_ent((p == 3 ? uA : (p == 4 ? uB : En)) + q * BLK, pe, Ent{acc.hi, acc.lo, dacc});
            }
          }
        }
      }
      if (live && lane == 0) {
        double prod = 1.0, dsum = 0.0;
#pragma unroll
        for (int i = 0; i < B; ++i) { prod *= L[3 * (i * B + i)]; dsum += L[3 * (i * B + i) + 2] * inv[2 * i]; }
        logdet = dd_add(logdet, two_prod(2.0 * (double)count, log(prod)));
        dlogdet += (double)count * 2.0 * dsum;
      }
      __syncthreads();
    }
    // ---- diagonal blocks of the next level: D - (update from the node on the left) - (update from the node on the right)
    for (int idx = tid; idx < ndn * BB; idx += NT) {
      const int q = idx / BB, ee = idx % BB;
      const int i0 = d_next[3 * q], i1 = d_next[3 * q + 1], i2 = d_next[3 * q + 2];
      Ent dv = ld_ent(Dv + i0 * BLK, ee);
      dd v = {dv.hi, dv.lo};
      if (i1 >= 0) { const Ent u = ld_ent(uB + i1 * BLK, ee); v = dd_sub(v, dd{u.hi, u.lo}); dv.d -= u.d; }
      if (i2 >= 0) { const Ent u = ld_ent(uA + i2 * BLK, ee); v = dd_sub(v, dd{u.hi, u.lo}); dv.d -= u.d; }
      st_ent(Dn + q * BLK, ee, Ent{v.hi, v.lo, dv.d});
    }
    __syncthreads();
    { double* t = Dv; Dv = Dn; Dn = t; t = Ev; Ev = En; En = t; }
  }
  // ---- root: L_0 and Sigma_00 = D_0^-1 (in the U_a slot)
  {
    const bool live = w == 0;
    if (live && act) st_ent(L, e, ld_ent(Dv, e));
    __syncthreads();
    pd_chol<B>(L, inv, dinv, live, r, c, 0, (unsigned long long)(levels * 64), bad_key);
    if (live && act && c > r) st_ent(L, e, Ent{0.0, 0.0, 0.0});
    __syncthreads();
    const int rec = n_rec - 1;
    if (live && r == 0 && c < B) {
      dd X[B];
      double Xd[B];
#pragma unroll
      for (int i = 0; i < B; ++i) { X[i] = dd{i == c ? 1.0 : 0.0, 0.0}; Xd[i] = 0.0; }
      pd_solve_col<B>(L, inv, X, Xd);
#pragma unroll
      for (int i = B - 1; i >= 0; --i) {                            // X <- L^-T X
        dd t = X[i];
#pragma unroll
        for (int q = i + 1; q < B; ++q) t = dd_sub(t, dd_mul(ld_dd(L, q * B + i), X[q]));
        X[i] = dd_mul(t, dd{inv[2 * i], inv[2 * i + 1]});
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
    if (live && act) {
      const Ent lv = ld_ent(L, e);
      val[(size_t)rec * W + prior_f_L(B) + e] = lv.hi; tan[(size_t)rec * W + prior_f_L(B) + e] = lv.d;
      if (r == 0) { val[(size_t)rec * W + prior_f_I(B) + c] = inv[2 * c]; tan[(size_t)rec * W + prior_f_I(B) + c] = dinv[c]; }
    }
    if (live && lane == 0) {
      double prod = 1.0, dsum = 0.0;
#pragma unroll
      for (int i = 0; i < B; ++i) { prod *= L[3 * (i * B + i)]; dsum += L[3 * (i * B + i) + 2] * inv[2 * i]; }
      logdet = dd_add(logdet, two_prod(2.0, log(prod)));
      dlogdet += 2.0 * dsum;
    }
  }
  if (lane == 0) { red[4 * w] = logdet.hi; red[4 * w + 1] = logdet.lo; red[4 * w + 2] = dlogdet; }
  __syncthreads();
  if (tid == 0) {
    dd ls = {0.0, 0.0};
    double dls = 0.0;
    for (int i = 0; i < NW; ++i) { ls = dd_add(ls, dd{red[4 * i], red[4 * i + 1]}); dls += red[4 * i + 2]; }
    const unsigned long long bk = *bad_key;
    tab[0] = ls.hi; tab[1] = dls; tab[2] = bk == ~0ull ? 0.0 : (double)(unsigned)(bk & 0xffffffffull); tab[3] = (double)n_rec;
    tab[4] = tab[5] = tab[6] = tab[7] = 0.0;
  }
}

template <int B>
int pd_launch(const int* img_i, const double* img_d, const PdCoefs& cf, double* tab, hipStream_t st) {
  const size_t lds_bytes = sizeof(double) * pd_lds_doubles<B>();
  auto kern = prior_forward_dd_kernel<B>;
  static bool granted = false;
  if (!granted && lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
    granted = true;
  }
  hipLaunchKernelGGL(kern, dim3(1), dim3(64 * pd_waves<B>()), lds_bytes, st, img_i, img_d, cf, tab);
  return ASVGP_OK;
}

}  // namespace

// the device image of the handle's plan (uploaded once per plan) and a device table ring of the pinned ring's shape
int handle_prior_dd_prepare(Handle* h) {
  if (!h->plan) { set_error("prior forward pass on the GPU: the handle holds no plan (asvgp_prior_plan_1d)"); return ASVGP_ERR_BAD_ARG; }
  if (h->dd_img_i) return ASVGP_OK;
  const size_t ni = prior_plan_image_ints(h->plan), nd = prior_plan_image_doubles(h->plan);
  std::vector<int> ints(ni);
  std::vector<double> dbls(nd ? nd : 1);
  prior_plan_image(h->plan, ints.data(), dbls.data());
  bool ok = hipMalloc(reinterpret_cast<void**>(&h->dd_img_i), sizeof(int) * ni) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&h->dd_img_d), sizeof(double) * (nd ? nd : 1)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&h->dd_tab), sizeof(double) * h->slot_doubles * TAB_SLOTS) == hipSuccess &&
            hipMemcpy(h->dd_img_i, ints.data(), sizeof(int) * ni, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(h->dd_img_d, dbls.data(), sizeof(double) * (nd ? nd : 1), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    handle_prior_dd_release(h);
    set_error("prior forward pass on the GPU: device allocation failed: %s", hipGetErrorString(hipGetLastError()));
    return ASVGP_ERR_HIP;
  }
  return ASVGP_OK;
}

void handle_prior_dd_release(Handle* h) {
  if (h->dd_img_i) { (void)hipFree(h->dd_img_i); h->dd_img_i = nullptr; }
  if (h->dd_img_d) { (void)hipFree(h->dd_img_d); h->dd_img_d = nullptr; }
  if (h->dd_tab) { (void)hipFree(h->dd_tab); h->dd_tab = nullptr; }
}

// enqueue the forward pass for one theta on `st`; the table lands in slot `slot` of the handle's DEVICE ring (returned)
int handle_prior_dd_forward(Handle* h, const double* coef, const double* dcoef, int slot, hipStream_t st, double** tab_out) {
  int rc = handle_prior_dd_prepare(h);
  if (rc) return rc;
  PdCoefs cf;
  for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) { cf.c[t] = coef[t]; cf.dc[t] = dcoef[t]; }
  double* tab = h->dd_tab + (size_t)slot * h->slot_doubles;
  switch (prior_plan_k(h->plan)) {
    case 1: rc = pd_launch<1>(h->dd_img_i, h->dd_img_d, cf, tab, st); break;
    case 2: rc = pd_launch<2>(h->dd_img_i, h->dd_img_d, cf, tab, st); break;
    case 3: rc = pd_launch<3>(h->dd_img_i, h->dd_img_d, cf, tab, st); break;
    case 4: rc = pd_launch<4>(h->dd_img_i, h->dd_img_d, cf, tab, st); break;
    case 5: rc = pd_launch<5>(h->dd_img_i, h->dd_img_d, cf, tab, st); break;
    case 6: rc = pd_launch<6>(h->dd_img_i, h->dd_img_d, cf, tab, st); break;
    default: set_error("prior forward pass on the GPU: bandwidth 1..6"); return ASVGP_ERR_BAD_ARG;
  }
  if (rc) return rc;
  *tab_out = tab;
  return check_launch("prior forward pass (double-double)");
}

}  // namespace asvgp

using namespace asvgp;

extern "C" int asvgp_set_prior_forward(asvgp_handle_t handle, int mode) {
  if (mode != 0 && mode != 1) { set_error("set_prior_forward: 0 host (long double), 1 GPU (double-double)"); return ASVGP_ERR_BAD_ARG; }
  as_handle(handle)->prior_forward_gpu = mode == 1;
  return ASVGP_OK;
}

// The GPU forward pass alone: the table for one theta copied back to the host (tests; compare with asvgp_prior_forward_host).
extern "C" int asvgp_prior_forward_device(asvgp_handle_t handle, const double* coef_host, const double* dcoef_dl_host, double* table_host,
                                          size_t table_doubles, void* stream) {
  Handle* h = as_handle(handle);
  if (!coef_host || !dcoef_dl_host || !table_host) { set_error("prior_forward_device: bad argument"); return ASVGP_ERR_BAD_ARG; }
  if (!h->plan) { set_error("prior_forward_device: the handle holds no plan (asvgp_prior_plan_1d)"); return ASVGP_ERR_BAD_ARG; }
  const size_t n = prior_plan_table_doubles(h->plan);
  if (table_doubles < n) { set_error("prior_forward_device: table too small"); return ASVGP_ERR_WORKSPACE; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned long long seq = 0;
  int slot = 0;
  (void)handle_table_acquire(h, &seq, &slot);
  double* tab = nullptr;
  int rc = handle_prior_dd_forward(h, coef_host, dcoef_dl_host, slot, st, &tab);
  if (rc) return rc;
  if (hipMemcpyAsync(table_host, tab, sizeof(double) * n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    set_error("prior_forward_device: copy failed: %s", hipGetErrorString(hipGetLastError()));
    return ASVGP_ERR_HIP;
  }
  __atomic_store_n(h->done_host + slot, seq, __ATOMIC_RELEASE);   // (no kernel consumes this slot)
  return ASVGP_OK;
}
