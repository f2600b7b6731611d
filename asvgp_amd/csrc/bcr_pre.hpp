// Backward half of the block cyclic reduction of bcr.hpp for the PRIOR chain, with the forward factors supplied by the
// host planner (prior_plan.cpp): band(Kuu^-1) and its d/d-lengthscale tangent from the per-class factor table.
//
// Same recurrences as bcr.hpp's backward pass (selected inverse, gpr.py:59 inverse_from_cholesky_band):
//   G = L^-T [U_a U_b],  S_ia = -(G_a S_aa + G_b S_ba),  S_ib = -(G_a S_ab + G_b S_bb),  S_ii = D_i^-1 - S_ia G_a^T - S_ib G_b^T
// but L, 1/diag(L), U_a, U_b of node i come from record node_rec[i] of the table (staged in the LDS: interior nodes of a
// level share ONE record, so the fetch is a broadcast), not from a per-node workspace written by a forward pass.
// One 256-thread workgroup; wide levels one thread per node, narrow levels one lane per block entry (as bcr.hpp).
#pragma once
#include "bcr.hpp"
#include "prior_plan.hpp"

namespace asvgp {

template <int B> __host__ __device__ constexpr int bcr_pre_ws_fields() { return 3 * B * B; }   // Sigma_ii, C_a, C_b per node
__host__ __device__ inline size_t bcr_pre_ws_doubles(int B, long nb) { return (size_t)2 * 3 * B * B * (size_t)((nb + 63) / 64 * 64); }
__host__ __device__ inline size_t bcr_pre_lds_doubles(int B, int n_rec) { return PRIOR_TAB_HEADER + (size_t)2 * n_rec * prior_rec_fields(B) + 64; }

// tab: table of prior_plan_eval in device-visible memory (pinned host memory or device memory); node_rec: nb ints (device).
// ws: bcr_pre_ws_doubles.  lds: bcr_pre_lds_doubles.  S: lower band of Kuu^-1 (value + tangent planes).
// logdet[0..1] = log|Kuu| and its tangent; info = first failing column + 1 as found by the host factorisation.
// done_flag (may be null): host-visible word that receives `seq` once the table has been copied out of `tab`.
template <int B>
__device__ __attribute__((always_inline)) void bcr_backward_pre(const double* __restrict__ tab, int n_rec, const int* __restrict__ node_rec,
                                                                 int M, double* ws, double* lds, BandOut<Dual> S, double* logdet, int* info,
                                                                 unsigned long long* done_flag, unsigned long long seq) {
  using T = Dual;
  using N = Num<T>;
  const int tid = threadIdx.x;
  const int nb = (M + B - 1) / B;
  constexpr int W = prior_rec_fields(B);
  constexpr int BB = B * B;
  constexpr int F_SD = 0, F_CA = BB, F_CB = 2 * BB, F_N = 3 * BB;
  // ---- stage the factor table: 16-B loads, everything in flight at once (the source may be host memory behind PCIe)
  const int n_tab = PRIOR_TAB_HEADER + 2 * n_rec * W;     // even: header 8, W * 2
  {
    const double2* src = reinterpret_cast<const double2*>(tab);
    double2* dst = reinterpret_cast<double2*>(lds);
    for (int e = tid; e < n_tab / 2; e += BCR_THREADS) dst[e] = src[e];
  }
  __syncthreads();
  if (done_flag && tid == 0) __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const double* tv = lds + PRIOR_TAB_HEADER;
  const double* td = tv + (size_t)n_rec * W;
  struct Fac {
    const double* v; const double* d;
    __device__ __forceinline__ Dual get(int f) const { return {v[f], d[f]}; }
  };
  auto Fc = [&](int node) -> Fac { const int r = node_rec[node]; return Fac{tv + r * W, td + r * W}; };
  const int NNr = (nb + 63) / 64 * 64;
  double* wsd = ws + (size_t)F_N * NNr;
  struct Rc {
    double* v; double* d; int stride;
    __device__ __forceinline__ Dual get(int f) const { return {v[(size_t)f * stride], d[(size_t)f * stride]}; }
    __device__ __forceinline__ void set(int f, Dual x) const { v[(size_t)f * stride] = x.v; d[(size_t)f * stride] = x.d; }
  };
  auto Wn = [&](int node) -> Rc { return Rc{ws + node, wsd + node, NNr}; };
  int levels = 0;
  while ((1 << levels) < nb) ++levels;
  constexpr int GS = GroupSize<B>::v;
  constexpr int NG = BCR_THREADS / GS;
  constexpr int LANE_MAX_NODES = NG;
  const int grp = tid / GS, e = tid % GS;
  const bool lane_on = e < BB;
  const int r = lane_on ? e / B : 0, c = lane_on ? e % B : 0;

  // ---- root: Sigma_00 from the table (U_a slot of the last record)
  if (tid < BB) {
    const Fac f0 = Fc(0);
    const T x = f0.get(prior_f_UA(B) + tid);
    Wn(0).set(F_SD + tid, x);
    const int rr = tid / B, cc = tid % B;
    if (rr >= cc && rr < M) S.store((long)(rr - cc) * M + cc, x);
  }
  __syncthreads();

  for (int l = levels - 1; l >= 0; --l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    if (ne <= LANE_MAX_NODES) {
      for (int m0 = 0; m0 < ne; m0 += NG) {
        const int m = m0 + grp;
        if (m < ne) {
          const int i = h + m * 2 * h, a = i - h, b = i + h;
          const bool hasb = b < nb;
          const Fac fi = Fc(i);
          T d = N::zero(), ga = N::zero(), gb = N::zero(), saa = N::zero(), sbb = N::zero(), sba = N::zero();
          T invd[B];
#pragma unroll
          for (int q = 0; q < B; ++q) invd[q] = fi.get(prior_f_I(B) + q);
          const bool e_is_a = ((a / (2 * h)) & 1) != 0;
          if (lane_on) {
            d = fi.get(prior_f_L(B) + e);
            ga = fi.get(prior_f_UA(B) + e);
            gb = fi.get(prior_f_UB(B) + e);
            saa = Wn(a).get(F_SD + e);
            if (hasb) {
              sbb = Wn(b).get(F_SD + e);
              sba = e_is_a ? Wn(a).get(F_CB + c * B + r) : Wn(b).get(F_CA + e);   // Sigma_ba[r][c]
            }
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
          T sii = xi;
#pragma unroll
          for (int p = 0; p < B; ++p) {
            T carp = gshfl<T>(ca, r * B + p, GS), gacp = gshfl<T>(ga, c * B + p, GS);
            T cbrp = gshfl<T>(cb, r * B + p, GS), gbcp = gshfl<T>(gb, c * B + p, GS);
            sii = N::nfma(carp, gacp, sii);
            sii = N::nfma(cbrp, gbcp, sii);
          }
          if (lane_on) {
            if (l > 0) {
              Wn(i).set(F_SD + e, sii);
              Wn(i).set(F_CA + e, ca);
              Wn(i).set(F_CB + e, cb);
            }
            if (r >= c && i * B + r < M) S.store((long)(r - c) * M + i * B + c, sii);
            if (l == 0) {
              if (r <= c && i * B + r < M) S.store((long)(B + r - c) * M + a * B + c, ca);
              if (hasb && c <= r && b * B + c < M) S.store((long)(B + c - r) * M + i * B + r, cb);
            }
          }
        }
        __syncthreads();
      }
      continue;
    }
    auto bwd_thread = [&](auto is0) {
      constexpr bool IS0 = decltype(is0)::value;
      for (int m0 = 0; m0 < ne; m0 += BCR_THREADS) {
        const int m = m0 + tid;
        if (m < ne) {
          const int i = h + m * 2 * h, a = i - h, b = i + h;
          const bool hasb = b < nb;
          const Fac fi = Fc(i);
          T L[B][B], invd[B], Ga[B][B], Gb[B][B];
#pragma unroll
          for (int rr = 0; rr < B; ++rr) {
            invd[rr] = fi.get(prior_f_I(B) + rr);
#pragma unroll
            for (int cc = 0; cc < B; ++cc) {
              L[rr][cc] = fi.get(prior_f_L(B) + rr * B + cc);
              Ga[rr][cc] = fi.get(prior_f_UA(B) + rr * B + cc);
              Gb[rr][cc] = fi.get(prior_f_UB(B) + rr * B + cc);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          blk_solve_LT<T, B, B>(L, invd, Ga);
          __builtin_amdgcn_sched_barrier(0);
          blk_solve_LT<T, B, B>(L, invd, Gb);
          __builtin_amdgcn_sched_barrier(0);
          T Saa[B][B], Sbb[B][B], Sba[B][B];
          const bool e_is_a = ((a / (2 * h)) & 1) != 0;
#pragma unroll
          for (int rr = 0; rr < B; ++rr)
#pragma unroll
            for (int cc = 0; cc < B; ++cc) {
              Saa[rr][cc] = Wn(a).get(F_SD + rr * B + cc);
              if (hasb) {
                Sbb[rr][cc] = Wn(b).get(F_SD + rr * B + cc);
                Sba[rr][cc] = e_is_a ? Wn(a).get(F_CB + cc * B + rr) : Wn(b).get(F_CA + rr * B + cc);
              } else {
                Sbb[rr][cc] = N::zero();
                Sba[rr][cc] = N::zero();
              }
            }
          T Ca[B][B], Cb[B][B], Sii[B][B];
#pragma unroll
          for (int rr = 0; rr < B; ++rr)
#pragma unroll
            for (int cc = 0; cc < B; ++cc) {
              T ta = N::zero(), tb = N::zero();
#pragma unroll
              for (int p = 0; p < B; ++p) {
                ta = N::nfma(Ga[rr][p], Saa[p][cc], ta);
                ta = N::nfma(Gb[rr][p], Sba[p][cc], ta);
                tb = N::nfma(Ga[rr][p], Sba[cc][p], tb);
                tb = N::nfma(Gb[rr][p], Sbb[p][cc], tb);
              }
              Ca[rr][cc] = ta;
              Cb[rr][cc] = tb;
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rr = 0; rr < B; ++rr)
#pragma unroll
            for (int cc = 0; cc < B; ++cc) Sii[rr][cc] = (rr == cc) ? N::make(1.0, 0.0) : N::zero();
          blk_solve_L<T, B, B>(L, invd, Sii);
          blk_solve_LT<T, B, B>(L, invd, Sii);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rr = 0; rr < B; ++rr)
#pragma unroll
            for (int cc = 0; cc < B; ++cc) {
              T t = Sii[rr][cc];
#pragma unroll
              for (int p = 0; p < B; ++p) {
                t = N::nfma(Ca[rr][p], Ga[cc][p], t);
                t = N::nfma(Cb[rr][p], Gb[cc][p], t);
              }
              if constexpr (!IS0) {
                Wn(i).set(F_SD + rr * B + cc, t);
                Wn(i).set(F_CA + rr * B + cc, Ca[rr][cc]);
                Wn(i).set(F_CB + rr * B + cc, Cb[rr][cc]);
              }
              if (cc <= rr && i * B + rr < M) S.store((long)(rr - cc) * M + i * B + cc, t);
              if constexpr (IS0) {
                if (rr <= cc && i * B + rr < M) S.store((long)(B + rr - cc) * M + a * B + cc, Ca[rr][cc]);
                if (hasb && cc <= rr && b * B + cc < M) S.store((long)(B + cc - rr) * M + i * B + rr, Cb[rr][cc]);
              }
            }
        }
        __syncthreads();
      }
    };
    if (l == 0) bwd_thread(std::true_type{});
    else bwd_thread(std::false_type{});
  }
  // right padding of the band rows (structural zeros): the last B columns
  for (int col = M - B + tid; col < M; col += BCR_THREADS)
    if (col >= 0)
#pragma unroll
      for (int d = 1; d <= B; ++d)
        if (col + d >= M) S.store((long)d * M + col, N::zero());
  if (tid == 0) {
    logdet[0] = lds[0];
    logdet[1] = lds[1];
    *info = (int)lds[2];
  }
}

}  // namespace asvgp
