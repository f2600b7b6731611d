// Centred-moment coefficient tables shared by the moment-based Phi kernels (phi_moments.hpp, phi_sort.hpp).
#pragma once
#include "asvgp_common.hpp"

namespace asvgp {

// Centred-moment coefficient tables: v_i(s + 1/2) and v_i v_j as polynomials in s = t - 1/2 with exact integer-ratio
// coefficients (used by the centred-moment Phi pass to turn per-cell moments into band / rhs entries).
using i128 = __int128;
// Compile-time coefficient tables (constexpr constructor => guaranteed constant evaluation; every use below has
// static indices after unrolling, so the entries fold into instruction literals).
template <int K> struct MomTab {
  double single[K + 1][K + 1];             // v_i(s + 1/2) = sum_p single[i][p] s^p
  double pair[K + 1][K + 1][2 * K + 1];    // v_i v_j      = sum_p pair[i][j][p] s^p   (j >= i)
  static constexpr long long pnum(int i, int q) {  // integer numerator of the t^q coefficient of piece i times K!
    long long num = 0;
    for (int j = 0; j <= i; ++j) {
      long long term = binom(K + 1, j) * binom(K, q) * ipow(i - j, K - q);
      num += (j & 1) ? -term : term;
    }
    return num;
  }
  constexpr MomTab() : single{}, pair{} {
    long long m[K + 1][K + 1] = {};        // v_i(s + 1/2) = (1 / (K! 2^K)) sum_r m[i][r] s^r
    for (int i = 0; i <= K; ++i)
      for (int r = 0; r <= K; ++r) {
        long long acc = 0;
        for (int q = r; q <= K; ++q) acc += pnum(i, q) * binom(q, r) * (1LL << (K - q + r));
        m[i][r] = acc;
      }
    const double d1 = (double)fact(K) * (double)(1LL << K);
    const double d2 = (double)(fact(K) * fact(K)) * (double)(1LL << (2 * K));
    for (int i = 0; i <= K; ++i) {
      for (int p = 0; p <= K; ++p) single[i][p] = (double)m[i][p] / d1;
      for (int j = i; j <= K; ++j)
        for (int p = 0; p <= 2 * K; ++p) {
          i128 acc = 0;
          for (int r = 0; r <= K; ++r) {
            int r2 = p - r;
            if (r2 < 0 || r2 > K) continue;
            acc += (i128)m[i][r] * (i128)m[j][r2];
          }
          pair[i][j][p] = (double)acc / d2;
        }
    }
  }
};
template <int K> struct MomCoef {
  static constexpr MomTab<K> tab{};
};

}  // namespace asvgp
