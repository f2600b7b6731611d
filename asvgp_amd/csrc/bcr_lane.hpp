// Lane-distributed node operations for block cyclic reduction (used by bcr.hpp for the NARROW levels of the tree).
// One B x B block entry per lane: lane e of a group of GS >= B*B lanes holds entry (r, c) = (e / B, e % B); block
// products, triangular solves and the small Cholesky go through intra-group shuffles, so a node costs a few hundred
// dependent instructions instead of the 2-3 thousand of the one-thread-per-node form.  Measured on MI355X (M = 2048,
// k = 4): ~6K cycles per level against 10-20K - but only while few waves shuffle at once: with all waves of a
// 1024-thread workgroup active the ds_bpermute traffic saturates the LDS crossbar (96K cycles at 256 nodes), which is
// why the wide levels keep one thread per node.  Storage is reached through the caller's accessors (same SoA layouts).
#pragma once
#include "asvgp_common.hpp"

namespace asvgp {

template <int B> struct GroupSize {
  static constexpr int v = (B == 1) ? 1 : (B == 2) ? 4 : (B <= 4) ? 16 : (B == 5) ? 32 : 64;
};
template <typename T> __device__ __forceinline__ T gshfl(T v, int src, int gs);
template <> __device__ __forceinline__ double gshfl<double>(double v, int src, int gs) { return __shfl(v, src, gs); }
template <> __device__ __forceinline__ Dual gshfl<Dual>(Dual v, int src, int gs) { return {__shfl(v.v, src, gs), __shfl(v.d, src, gs)}; }

// in-group Cholesky of the symmetric block held one entry per lane (lower part of d becomes L); invd = 1 / diag(L)
template <typename T, int B>
__device__ __forceinline__ void lane_chol(T& d, T (&invd)[B], int r, int c, int& bad, int col0) {
  using N = Num<T>;
  constexpr int GS = GroupSize<B>::v;
#pragma unroll
  for (int j = 0; j < B; ++j) {
    T pj = gshfl<T>(d, j * B + j, GS);
    if (!(N::val(pj) > 0.0) && !bad) bad = col0 + j + 1;
    T ljj, inv;
    N::sqrt_inv(pj, ljj, inv);
    invd[j] = inv;
    if (c == j) d = (r == j) ? ljj : ((r > j) ? d * inv : d);
    T lrj = gshfl<T>(d, r * B + j, GS), lcj = gshfl<T>(d, c * B + j, GS);
    if (c > j && r >= c) d = N::nfma(lrj, lcj, d);
  }
}

}  // namespace asvgp
