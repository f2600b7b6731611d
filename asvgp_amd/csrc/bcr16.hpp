// Lane-distributed block cyclic reduction: the same elimination as bcr.hpp, but every B x B block operation of a node
// is spread over a group of GS >= B*B lanes (lane e <-> block entry (r, c) = (e / B, e % B)); products, triangular solves
// and the small Cholesky run through intra-group shuffles, so the dependent chain per level is a few hundred
// instructions instead of the ~2-3 thousand of one-thread-per-node.  1024 threads = 64 groups (B <= 4) work on the
// eliminated nodes of a level in rounds; survivors live in LDS as array-of-structs (a group's GS lanes touch
// consecutive 8-B words), factors in an L2-resident array-of-structs workspace (128 contiguous bytes per group access).
//
//   forward  (node i, neighbours a = i-h, b = i+h):  D_i = L L^T, U_a = L^-1 A[i,a], U_b = L^-1 A[i,b], z = L^-1 y_i,
//            D_a -= U_a^T U_a, y_a -= U_a^T z, A'[b,a] = -U_b^T U_a | barrier | D_b -= U_b^T U_b, y_b -= U_b^T z
//   backward: x_i = L^-T (z - U_a x_a - U_b x_b), G = L^-T [U_a U_b], S_ia = -(G_a S_aa + G_b S_ba),
//            S_ib = -(G_a S_ab + G_b S_bb), S_ii = D_i^-1 - S_ia G_a^T - S_ib G_b^T
#pragma once
#include "bcr.hpp"

namespace asvgp {

constexpr int BCR16_THREADS = 1024;


template <int B> struct Bcr16Layout {
  static constexpr int BB = B * B;
  static constexpr int F_D = 0, F_E = BB, F_N = 2 * BB;                       // LDS record per survivor slot
  static constexpr int W_L = 0, W_UA = BB, W_UB = 2 * BB, W_SD = 3 * BB, W_CA = 4 * BB, W_CB = 5 * BB, W_I = 6 * BB,
                       W_N = 6 * BB + B;                                      // workspace record per node
};

template <typename T, int B> __host__ __device__ constexpr int bcr16_ns() { return bcr_ns<T, B>(); }
template <typename T, int B>
__host__ __device__ inline size_t bcr16_lds_doubles(long nb) {
  if ((nb + 1) / 2 > bcr16_ns<T, B>()) return (size_t)1 << 40;
  return (size_t)planes_of<T>() * Bcr16Layout<B>::F_N * bcr16_ns<T, B>() + (size_t)nb * B + 64;
}

template <typename T, int B, int NRHS>
__device__ void bcr16_solve(BandPtr<T> A, const double* rhs, int M, double* ws, double* lds, BandOut<T> S, double* x,
                            double* logdet, int* info, double* stamps = nullptr) {
  using N = Num<T>;
  using Lay = Bcr16Layout<B>;
  constexpr int GS = GroupSize<B>::v;
  constexpr int NG = BCR16_THREADS / GS;   // groups per workgroup
  constexpr int BB = B * B;
  constexpr int NS = bcr16_ns<T, B>();
  constexpr int P2 = planes_of<T>();
  const int tid = threadIdx.x;
  const int grp = tid / GS, e = tid % GS;
  const bool lane_on = e < BB;
  const int r = lane_on ? e / B : 0, c = lane_on ? e % B : 0;
  const int nb = (M + B - 1) / B;
  // LDS: survivors (array of records), planes apart;  xs: y / z / x per row;  red: scratch
  double* Fp = lds;
  double* Fq = lds + (size_t)Lay::F_N * NS;
  double* xs = lds + (size_t)P2 * Lay::F_N * NS;
  double* red = xs + (size_t)nb * B;
  auto fget = [&](int slot, int f) -> T {
    if constexpr (P2 == 2) return N::make(Fp[slot * Lay::F_N + f], Fq[slot * Lay::F_N + f]);
    else return N::make(Fp[slot * Lay::F_N + f], 0.0);
  };
  auto fset = [&](int slot, int f, T v) {
    Fp[slot * Lay::F_N + f] = N::val(v);
    if constexpr (P2 == 2) Fq[slot * Lay::F_N + f] = N::tan(v);
  };
  // workspace records
  double* Wp = ws;
  double* Wq = ws + (size_t)Lay::W_N * 2 * NS;
  auto wget = [&](int node, int f) -> T {
    if constexpr (P2 == 2) return N::make(Wp[(long)node * Lay::W_N + f], Wq[(long)node * Lay::W_N + f]);
    else return N::make(Wp[(long)node * Lay::W_N + f], 0.0);
  };
  auto wset = [&](int node, int f, T v) {
    Wp[(long)node * Lay::W_N + f] = N::val(v);
    if constexpr (P2 == 2) Wq[(long)node * Lay::W_N + f] = N::tan(v);
  };
  int nst = 0;
  unsigned long long t_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  auto stamp = [&]() {
    if (stamps && tid == 0) {
      unsigned long long t = __builtin_amdgcn_s_memtime();
      stamps[nst++] = (double)(t - t_prev);
      t_prev = t;
    }
  };
  int bad = 0;
  int levels = 0;
  while ((1 << levels) < nb) ++levels;

  // pre-pass: even nodes' diagonal blocks -> LDS records (full symmetric), rhs -> xs
  for (int n = 2 * grp; n < nb; n += 2 * NG) {
    if (lane_on) {
      const int rr = r >= c ? r : c, cc = r >= c ? c : r;
      fset(n >> 1, Lay::F_D + e, band_D<T, B, BandPtr<T>>(A, M, n, rr, cc));
    }
  }
  if (NRHS)
    for (int q = tid; q < nb * B; q += BCR16_THREADS) xs[q] = (q < M) ? rhs[q] : 0.0;
  __syncthreads();
  stamp();

  // ---------------- forward elimination ----------------
  for (int l = 0; l < levels; ++l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
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
            d = band_D<T, B, BandPtr<T>>(A, M, i, rr, cc);
            ua = band_E<T, B, BandPtr<T>>(A, M, a, r, c);
            ub = hasb ? band_E<T, B, BandPtr<T>>(A, M, i, c, r) : N::zero();   // A[i,b] = E(i)^T
          } else {
            d = fget(i >> 1, Lay::F_D + e);
            ua = fget(a >> 1, Lay::F_E + e);
            ub = hasb ? fget(i >> 1, Lay::F_E + c * B + r) : N::zero();
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
          wset(i, Lay::W_L + e, (r >= c) ? d : N::zero());
          wset(i, Lay::W_UA + e, ua);
          wset(i, Lay::W_UB + e, ub);
          if (e < B) {
            T iv = invd[0];
#pragma unroll
            for (int q = 1; q < B; ++q) iv = (e == q) ? invd[q] : iv;
            wset(i, Lay::W_I + e, iv);
          }
          if (NRHS && c == 0) xs[i * B + r] = z;
          // phase A: left neighbour
          const int sa = a >> 1;
          fset(sa, Lay::F_D + e, fget(sa, Lay::F_D + e) - upd_a);
          fset(sa, Lay::F_E + e, enew);
          if (NRHS && c == 0) xs[a * B + r] -= ya_upd;
        }
      }
      bcr_lds_barrier();
      if (hasb && lane_on) {  // phase B: right neighbour
        const int sb = b >> 1;
        fset(sb, Lay::F_D + e, fget(sb, Lay::F_D + e) - upd_b);
        if (NRHS && c == 0) xs[b * B + r] -= yb_upd;
      }
      bcr_lds_barrier();
    }
    stamp();
  }

  // ---------------- root (node 0): group 0 ----------------
  if (grp == 0) {
    T d = lane_on ? fget(0, Lay::F_D + e) : N::zero();
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
      wset(0, Lay::W_SD + e, xi);
      wset(0, Lay::W_L + e, (r >= c) ? d : N::zero());
      if (NRHS && c == 0) xs[r] = z;
    }
  }
  __syncthreads();
  stamp();

  // ---------------- backward: solve + selected inverse ----------------
  for (int l = levels - 1; l >= 0; --l) {
    const int h = 1 << l;
    const int ne = (nb > h) ? (nb - h + 2 * h - 1) / (2 * h) : 0;
    for (int m0 = 0; m0 < ne; m0 += NG) {
      const int m = m0 + grp;
      if (m < ne) {
        const int i = h + m * 2 * h, a = i - h, b = i + h;
        const bool hasb = b < nb;
        T d = N::zero(), ga = N::zero(), gb = N::zero(), saa = N::zero(), sbb = N::zero(), sba = N::zero();
        T invd[B];
#pragma unroll
        for (int q = 0; q < B; ++q) invd[q] = wget(i, Lay::W_I + q);
        const bool e_is_a = ((a / (2 * h)) & 1) != 0;
        if (lane_on) {
          d = wget(i, Lay::W_L + e);
          ga = wget(i, Lay::W_UA + e);
          gb = wget(i, Lay::W_UB + e);
          saa = wget(a, Lay::W_SD + e);
          if (hasb) {
            sbb = wget(b, Lay::W_SD + e);
            sba = e_is_a ? wget(a, Lay::W_CB + c * B + r) : wget(b, Lay::W_CA + e);   // Sigma_ba[r][c]
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
          wset(i, Lay::W_SD + e, sii);
          wset(i, Lay::W_CA + e, ca);
          wset(i, Lay::W_CB + e, cb);
        }
      }
      __syncthreads();
    }
    stamp();
  }

  // ---------------- outputs ----------------
  for (int n = grp; n < nb; n += NG) {
    if (lane_on) {
      const int row = n * B + r, col = n * B + c;
      if (r >= c && row < M) S.store((long)(r - c) * M + col, wget(n, Lay::W_SD + e));
      if (n + 1 < nb && r <= c) {   // Sigma[(n+1)B + r, nB + c]
        const int row2 = (n + 1) * B + r;
        T v = (n & 1) ? wget(n, Lay::W_CB + c * B + r) : wget(n + 1, Lay::W_CA + e);
        if (row2 < M) S.store((long)(B + r - c) * M + col, v);
      }
      if (c == 0) {  // right padding of the band rows
        const int cl = n * B + r;
#pragma unroll
        for (int dd = 1; dd <= B; ++dd)
          if (cl < M && cl + dd >= M) S.store((long)dd * M + cl, N::zero());
      }
    }
  }
  if (NRHS)
    for (int q = tid; q < M; q += BCR16_THREADS) x[q] = xs[q];
  double ld_acc = 0.0, dld_acc = 0.0;
  for (int q = tid; q < nb * B; q += BCR16_THREADS) {
    T dv = wget(q / B, Lay::W_L + (q % B) * B + (q % B));
    ld_acc += 2.0 * log(N::val(dv));
    dld_acc += 2.0 * N::tan(dv) / N::val(dv);
  }
  double tot = block_sum(ld_acc, red);
  double dtot = block_sum(dld_acc, red);
  stamp();
  if (tid == 0) {
    logdet[0] = tot;
    logdet[1] = dtot;
  }
  int* sbad = reinterpret_cast<int*>(red + 32);
  __syncthreads();
  if (tid == 0) *sbad = 0x7fffffff;
  __syncthreads();
  if (bad) atomicMin(sbad, bad);
  __syncthreads();
  if (tid == 0) *info = (*sbad == 0x7fffffff) ? 0 : *sbad;
}

}  // namespace asvgp
