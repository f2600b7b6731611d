// Fused ELBO + gradient and posterior-preparation drivers (stream-ordered, no host sync).
//
// Replaces GPR_1d.elbo (gpr.py:49-89) + the TF reverse-mode pass through banded_matrices' op gradients, and
// the CHOLMOD factor/solve part of GPR_1d.predict_f (gpr.py:96-108).  Everything is O(M k^2) on (k+1) x M bands;
// the two independent chains (Kuu with its d/d-lengthscale tangent, and P = Kuu + A/s with the rhs riding along)
// run concurrently as two single-wave workgroups per phase.
//
//   phase 0  prepare  : Kuu, dKuu/dl (inducing_features.py:16-44), P = A/s + Kuu (gpr.py:72)          [elementwise]
//   phase 1  factor   : block0 dual Cholesky(Kuu) (gpr.py:56) | block1 Cholesky(P) + c = L_P^-1 b (gpr.py:73-75)
//   phase 2  inverse  : block0 dual Takahashi(Kuu) (gpr.py:59) | block1 Takahashi(P) + alpha = L_P^-T c
//   phase 3  finalize : log-dets (gpr.py:57,74), band traces (gpr.py:60-70), 7-term bound (gpr.py:78-87), gradient
#include <stdlib.h>
#include <time.h>
#include <string.h>

#include "bcr_pre.hpp"
#include "bcr_mfma.hpp"
#include "handle.hpp"

namespace asvgp {

// (elbo.hip is compiled once per bandwidth with -DASVGP_ELBO_ONLY_K=k - those units hold the template instantiations -
// and once without: the C entry points.  asvgp_amd/build.py runs the units in parallel.)
// Band algorithm (Handle::band_algo): 0 auto, 1 sequential single-wave sweeps (the reference's elimination order),
// 2 block cyclic reduction, both chains on the GPU, 3 block cyclic reduction with the PLANNED prior chain: forward pass of
// the Kuu chain on the host in long double over the distinct nodes (prior_plan.cpp), backward pass on the GPU (bcr_pre.hpp),
// 4 the planned chains on the matrix cores (bcr_mfma.hpp: k = 4, D = 1, M <= 2048).
// Auto = 4 where it applies, else 3 when the handle holds a plan (asvgp_prior_plan_1d) for this (M, k, kind), else 2, else 1 (D > 1 or LDS).
// With asvgp_elbo_chain_sync the prior chain (stream A) and the data chain (stream B) of algorithm 2 are ordered by the
// handle's own events: evK = Kuu assembled (the P chain may start), evP = prior chain complete (the finalize may start).

struct Ws {
  double *Kuu, *dK, *P, *LK, *dLK, *LP, *SK, *dSK, *SP, *c, *alpha, *logdets, *fin, *bcrK, *bcrP;
};
// factor workspaces of the two BCR chains (bcr_ws_doubles, normal or BIG layout - whichever is larger)
static size_t bcr_ws_chain(int planes, int k, bool big) {
  long ns = 1024;
  while ((long)planes * 2 * k * k * ns * 8 + 2 * ns * k * 8 + 1024 > 150 * 1024) ns >>= 1;
  if (big) ns *= 2;
  return (size_t)planes * (7 * k * k + k) * 2 * ns + (big ? (size_t)planes * k * k * ns : 0);
}
static size_t bcr_ws_K(int k, long M) {   // the all-GPU Dual chain's factor records, or the planned chain's Sigma records (bcr_pre.hpp)
  size_t a = bcr_ws_chain(2, k, false), b = bcr_ws_chain(2, k, true), c = bcr_pre_ws_doubles(k, (M + k - 1) / k);
  a = a > b ? a : b;
  return (a > c ? a : c) + 64;
}
static size_t bcr_ws_P(int k) { size_t a = bcr_ws_chain(1, k, false), b = bcr_ws_chain(1, k, true); return (a > b ? a : b) + 64; }
static size_t bcr_ws_total(long M, int k) { return bcr_ws_K(k, M) + bcr_ws_P(k); }
static size_t ws_doubles(long M, int k, long D) {
  return (size_t)9 * (k + 1) * M + (size_t)2 * M * D + 64 + 32 + bcr_ws_total(M, k);
}
static Ws carve(void* ws, long M, int k, long D) {
  double* p = static_cast<double*>(ws);
  size_t E = (size_t)(k + 1) * M;
  Ws w;
  w.Kuu = p; p += E; w.dK = p; p += E; w.P = p; p += E; w.LK = p; p += E; w.dLK = p; p += E;
  w.LP = p; p += E; w.SK = p; p += E; w.dSK = p; p += E; w.SP = p; p += E;
  w.c = p; p += (size_t)M * D; w.alpha = p; p += (size_t)M * D;
  w.logdets = p; p += 64;
  w.fin = p; p += 32;   // 14 partial-sum slots + the arrival ticket of elbo_finalize_kernel (zero between calls)
  w.bcrK = p; p += bcr_ws_K(k, M);
  w.bcrP = p;
  return w;
}

// launchers: declared for every unit, defined (and explicitly instantiated) only in the per-bandwidth units
template <int K> struct ElboLauncher {
  static int run(Handle* h, const double* stats, const double* S, int kind, double v, double l, double s, long N, long M, long D,
                 double* out, int* info, void* ws, hipStream_t st, int part);
};
template <int K> struct KuuInvLauncher {   // band(Kuu^-1) and its d/d-lengthscale tangent for one 1-D Kuu (the Kronecker factors)
  static int run(Handle* h, const double* S, int kind, double v, double l, long M, double* Kuu, double* dK, double* SK, double* dSK,
                 double* logdet2, int* info, void* ws, hipStream_t st);
};
template <int K> struct PostLauncher {
  static int run(Handle* h, const double* stats, const double* S, int kind, double v, double l, double s, long M, long D,
                 double* alpha, double* W, int* info, void* ws, hipStream_t st);
};

#ifdef ASVGP_ELBO_ONLY_K
struct KuuCoefs2 { double c[ASVGP_MAX_KUU_TERMS]; double dc[ASVGP_MAX_KUU_TERMS]; int n; };

// P = A / s + Kuu for the matrix-core P chain WITHOUT waiting for the helper workgroups' Kuu (~4.5 us at the head of the launch): on the
// Toeplitz interior of the static bands (prior_plan.cpp: columns [lo, hi), found by exact comparison) diagonal d of Kuu is ONE number,
// and the few boundary columns' entries are a 2 x 8 x 16 table; both are formed on the host with the reference's rounding sequence and
// travel as the FIRST kernel argument.  Branch-free: every lane loads its A entry and one table entry (index 0 when interior) - a branch
// around the boundary case made the compiler wait for every A load separately (12 dependent round trips in level 0, +4 us).
// dk / dkb: the same for dKuu / dl (the P chain's own traces): interior values and a 2 x (k + 1 = 5) x 8 table of the boundary columns
// (dk_tab = 1; wider boundaries: dk_tab = 0 and the kernel sums the static bands for those columns) - kernel arguments end at 4 KB
constexpr int KI_DKB = 8;
struct KuuInterior { double k[8]; long lo, hi; double bnd[2 * PRIOR_BND_DIAGS * PRIOR_BND]; double dk[8]; double dkb[2 * 5 * KI_DKB]; int dk_tab; };
// Launch-ahead (asvgp_elbo_grad_ahead_1d / asvgp_elbo_publish_theta): everything of the matrix-core launch that depends on theta, handed
// over through pinned host memory AFTER the launch - the kernel is resident (its ~8 us of launch path and dispatch latency already spent)
// when the optimiser has its next theta.  seq is stored last (release); ~0: the host has withdrawn the launch.
struct ThetaBox {
  unsigned long long seq;
  double s, alpha_scale, v, l, N;
  double k[8], dk[8];
  double bnd[2 * PRIOR_BND_DIAGS * PRIOR_BND];
  double dkb[2 * 5 * KI_DKB];
  double dc[ASVGP_MAX_KUU_TERMS];
};
typedef const __attribute__((address_space(3))) double* lds_cdouble_ptr;
struct BandSumToep {
  const double* A; double inv_s;
  lds_cdouble_ptr kl;                     // the interior diagonal values, in the LDS (a five-way select between struct members by a run-time
  lds_cdouble_ptr bnd;                    // diagonal was turned into address arithmetic on a scratch copy of the struct); bnd: the boundary table,
                                          // copied from the kernel-argument segment to the LDS once (a vector load from that segment in front of
                                          // level 0 cost ~3 us: it is not device memory)
  long lo, hi;
  __device__ __forceinline__ double load_dc(int dd, long col, long M) const {
#pragma clang fp contract(off)
    const double t2 = A[(long)dd * M + col] * inv_s;
    const bool left = col < lo, right = col >= hi;
    const long bi = left ? (long)dd * PRIOR_BND + col : (right ? (long)(PRIOR_BND_DIAGS + dd) * PRIOR_BND + (col - hi) : 0);
    const double bv = bnd[bi];
    const double kv = kl[dd];
    return t2 + ((left || right) ? bv : kv);
  }
};

static __global__ void elbo_prepare_kernel(const double* __restrict__ S, KuuCoefs2 cf, long E, const double* __restrict__ A,
                                    double s, double* __restrict__ Kuu, double* __restrict__ dK,
                                    double* __restrict__ P) {
  // (HIP's __dmul_rn / __dadd_rn are plain * and +: without the pragma the compiler contracts them into fma and the band is no
  // longer the reference's rounding sequence - found in round 2 through a 1-ulp knot mismatch in the Phi kernel)
#pragma clang fp contract(off)
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double acc = cf.c[0] * S[e];
  double dacc = cf.dc[0] * S[e];
  for (int t = 1; t < cf.n; ++t) {
    double sv = S[(long)t * E + e];
    acc = acc + cf.c[t] * sv;
    dacc = dacc + cf.dc[t] * sv;
  }
  Kuu[e] = acc;
  if (dK) dK[e] = dacc;
  if (P) P[e] = A[e] / s + acc;  // gpr.py:72  KufKfu / sigma2 + Kuu
}

template <int K, bool TANGENT, bool RHS>
__global__ __launch_bounds__(64) void elbo_factor_kernel(const double* Kuu, const double* dK, const double* P,
                                                         double* LK, double* dLK, double* LP, const double* b,
                                                         double* c, int M, int* info) {
  if (blockIdx.x == 0) {
    if (TANGENT) cholesky_sweep<Dual, K, false>(BandPtr<Dual>{Kuu, dK}, BandOut<Dual>{LK, dLK}, M, nullptr, nullptr, info);
    else cholesky_sweep<double, K, false>(BandPtr<double>{Kuu, nullptr}, BandOut<double>{LK, nullptr}, M, nullptr, nullptr, info);
  } else {
    cholesky_sweep<double, K, RHS>(BandPtr<double>{P, nullptr}, BandOut<double>{LP, nullptr}, M, b, c, info + 1);
  }
}

template <int K, bool TANGENT, bool RHS>
__global__ __launch_bounds__(64) void elbo_inverse_kernel(const double* LK, const double* dLK, const double* LP,
                                                          double* SK, double* dSK, double* SP, const double* c,
                                                          double* alpha, int M) {
  if (blockIdx.x == 0) {
    if (TANGENT) takahashi_sweep<Dual, K, false>(BandPtr<Dual>{LK, dLK}, BandOut<Dual>{SK, dSK}, M, nullptr, nullptr);
    else takahashi_sweep<double, K, false>(BandPtr<double>{LK, nullptr}, BandOut<double>{SK, nullptr}, M, nullptr, nullptr);
  } else {
    takahashi_sweep<double, K, RHS>(BandPtr<double>{LP, nullptr}, BandOut<double>{SP, nullptr}, M, c, alpha);
  }
}

template <int K>
__global__ __launch_bounds__(64) void elbo_trsv_kernel(const double* L, int M, const double* B, double* X, long D,
                                                       int trans, double scale) {
  const long d = blockIdx.x;
  if (trans) trsv_sweep<K, true>(L, M, B + d, X + d, D);
  else trsv_sweep<K, false>(L, M, B + d, X + d, D);
  (void)scale;
}

// columns 1 .. D-1 of a multi-output rhs, in chunks of as many columns as the (now free) LDS image holds
template <int K, bool BIG>
__device__ __forceinline__ void more_columns(const double* b, double* x, int D, int M, double* wsP, double* lds, int lds_doubles) {
  if (D <= 1) return;
  __syncthreads();
  const int nb = (M + K - 1) / K;
  int chunk = lds_doubles / (nb * K);
  if (chunk < 1) chunk = 1;
  for (int d0 = 1; d0 < D; d0 += chunk) bcr_solve_more<K, BIG>(b, x, D, d0, (D - d0 < chunk) ? D - d0 : chunk, M, wsP, lds);
}

// Both chains by block cyclic reduction, one 256-thread workgroup each (bcr.hpp).
template <int K, bool TANGENT, bool BIG>
__global__ __launch_bounds__(BCR_THREADS) void elbo_bcr_kernel(const double* Kuu, const double* dK, const double* P,
                                                               const double* b, int M, double* wsK, double* wsP,
                                                               double* SK, double* dSK, double* SP, double* x,
                                                               double* logdets, int* info, int do_stamps,
                                                               int first_chain, int D, int lds_doubles) {
  extern __shared__ double lds[];
  double* st = do_stamps ? logdets + 8 : nullptr;  // diagnostic: 24 stamps per chain after the 4 log-det slots
  // do_stamps == 2 (diagnostic): run the solve twice and stamp the second, warm, pass (instruction cache / TLB effects)
  for (int rep = (do_stamps == 2) ? 0 : 1; rep < 2; ++rep) {
    if (blockIdx.x + first_chain == 0) {
      if (TANGENT) bcr_solve<Dual, K, 0, BandPtr<Dual>, BIG>(BandPtr<Dual>{Kuu, dK}, nullptr, M, wsK, lds, BandOut<Dual>{SK, dSK}, nullptr, logdets, info, st);
      else bcr_solve<double, K, 0, BandPtr<double>, BIG>(BandPtr<double>{Kuu, nullptr}, nullptr, M, wsK, lds, BandOut<double>{SK, nullptr}, nullptr, logdets, info, st);
    } else {   // (the split data chain, P formed in the gathers, has its own kernel: elbo_bcr_data_kernel)
      bcr_solve<double, K, 1, BandPtr<double>, BIG>(BandPtr<double>{P, nullptr}, b, M, wsP, lds, BandOut<double>{SP, nullptr}, x, logdets + 2, info + 1, st ? st + 24 : nullptr, D);
      more_columns<K, BIG>(b, x, D, M, wsP, lds, lds_doubles);
    }
    __syncthreads();
  }
}

// Prior chain alone (asvgp_elbo_prior_chain_1d): the Dual (value + tangent) Kuu chain in a kernel of its own - 35 spilled
// VGPRs instead of the 135 of the combined kernel.
template <int K, bool BIG>
__global__ __launch_bounds__(BCR_THREADS) void elbo_bcr_prior_kernel(const double* Kuu, const double* dK, int M, double* wsK,
                                                                     double* SK, double* dSK, double* logdets, int* info,
                                                                     int do_stamps) {
  extern __shared__ double lds[];
  double* st = do_stamps ? logdets + 8 : nullptr;
  bcr_solve<Dual, K, 0, BandPtr<Dual>, BIG>(BandPtr<Dual>{Kuu, dK}, nullptr, M, wsK, lds, BandOut<Dual>{SK, dSK}, nullptr, logdets, info, st);
}

// Data chain alone (asvgp_elbo_data_chain_1d): the P chain with P = A/s + Kuu formed in its gathers.  A kernel of its own
// so that its register allocation is not shared with the Dual (tangent) chain: 256 VGPRs, no spills (the combined kernel
// above spills 135 VGPRs), and it is the one on the critical path of a step.
template <int K, bool BIG>
__global__ __launch_bounds__(BCR_THREADS) void elbo_bcr_data_kernel(const double* Kuu, const double* A, const double* b, int M,
                                                                    double* wsP, double* SP, double* x, double* logdets,
                                                                    int* info, int do_stamps, double s, int D, int lds_doubles) {
  extern __shared__ double lds[];
  double* st = do_stamps ? logdets + 8 : nullptr;
  bcr_solve<double, K, 1, BandSumP, BIG>(BandSumP{A, Kuu, 1.0 / s}, b, M, wsP, lds, BandOut<double>{SP, nullptr}, x, logdets + 2, info + 1,
                                         st ? st + 24 : nullptr, D);
  more_columns<K, BIG>(b, x, D, M, wsP, lds, lds_doubles);
}

// sym-band quadratic form helper: x^T sym(S) x over columns handled by this thread
// (branch-free: out-of-range neighbours are clamped and their band entry is a structural zero of the right-padded band)
template <int KT>
__device__ __forceinline__ double quad_col(const double* S, long M, long j, const double* x, long D, long d) {
  double xj = x[j * D + d];
  double a = S[j] * xj * xj;
#pragma unroll
  for (int r = 1; r <= KT; ++r) {
    long jr = (j + r < M) ? j + r : M - 1;
    double sv = (j + r < M) ? S[(long)r * M + j] : 0.0;
    a = fma(2.0 * sv * xj, x[jr * D + d], a);
  }
  return a;
}

struct ElboScalars { double v, l, s, N; };

// The 14 band traces / quadratic forms of the bound and its gradient over `nblk` cooperating workgroups (this one is `bid`);
// partial sums meet in `gacc` through device-scope atomics, the last workgroup to arrive writes the result and re-arms gacc /
// ticket (and `rearm`, the chain-arrival counter of the fused launch) for the next call.  `part`: 16 * 14 doubles of LDS.
template <int KT>
__device__ __forceinline__ void elbo_finalize_body(
    int bid, int nblk, double* part_lds, const double* __restrict__ stats, const double* __restrict__ Kuu, const double* __restrict__ dK,
    const double* __restrict__ LK, const double* __restrict__ LP, const double* __restrict__ SK,
    const double* __restrict__ dSK, const double* __restrict__ SP, const double* __restrict__ c,
    const double* __restrict__ alpha, const double* __restrict__ logdets, long M, long D, ElboScalars th,
    double alpha_scale, double* __restrict__ gacc, unsigned* __restrict__ ticket, unsigned* __restrict__ rearm, double* __restrict__ out) {
  // (a single workgroup is limited by one CU's L1 bandwidth: ~740 KB of bands at M = 2048)
  constexpr int k = KT;   // compile-time bandwidth: the per-column loops unroll and their loads issue together
  const double* A = stats;
  const double* b = stats + (long)(k + 1) * M;
  const double yy = stats[(long)(k + 1) * M + M * D];
  enum { LOGK, LOGP, TRKA, DTRKA, SKDK, SPDK, SKK, SPK, SPA, CC, AKA, ADKA, AAA, BA, NACC };
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  // (contiguous column chunks, one per workgroup whatever its size: the ~740 KB of bands come from memory / the other XCDs' write-backs at
  //  ~25 GB/s per CU, so the traffic is spread over all nblk CUs - with 1024-thread workgroups a grid-stride loop left it to two of them)
  const long chunk = (M + nblk - 1) / nblk;
  const long j_end = ((long)(bid + 1) * chunk < M) ? (long)(bid + 1) * chunk : M;
  for (long j = (long)bid * chunk + threadIdx.x; j < j_end; j += blockDim.x) {
    if (!logdets) {
      double lk = LK[j], lp = LP[j];
      acc[LOGK] += log(lk * lk);  // gpr.py:57  log(square(L[0,:]))
      acc[LOGP] += log(lp * lp);  // gpr.py:74
    }
#pragma unroll
    for (int r = 0; r <= k; ++r) {
      long o = (long)r * M + j;
      double w = (r == 0) ? 1.0 : 2.0;
      double sk = SK[o], sp = SP[o], a = A[o], kk = Kuu[o], dk = dK[o];
      acc[TRKA] = fma(w * sk, a, acc[TRKA]);
      acc[DTRKA] = fma(w * dSK[o], a, acc[DTRKA]);
      acc[SKDK] = fma(w * sk, dk, acc[SKDK]);
      acc[SPDK] = fma(w * sp, dk, acc[SPDK]);
      acc[SKK] = fma(w * sk, kk, acc[SKK]);
      acc[SPK] = fma(w * sp, kk, acc[SPK]);
      acc[SPA] = fma(w * sp, a, acc[SPA]);
    }
    for (long d = 0; d < D; ++d) {
      if (!logdets) {
        double cv = c[j * D + d];
        acc[CC] = fma(cv, cv, acc[CC]);
      }
      acc[AKA] += quad_col<KT>(Kuu, M, j, alpha, D, d);
      acc[ADKA] += quad_col<KT>(dK, M, j, alpha, D, d);
      acc[AAA] += quad_col<KT>(A, M, j, alpha, D, d);
      acc[BA] = fma(b[j * D + d], alpha[j * D + d], acc[BA]);
    }
  }
  // one shuffle tree per accumulator, ONE barrier pair for all of them
  double (*part)[NACC] = reinterpret_cast<double (*)[NACC]>(part_lds);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    double v = wave_sum_dpp(acc[i]);
    if (lane == 0) part[wv][i] = v;
  }
  __syncthreads();
  double tot[NACC];
  if (threadIdx.x < 64) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) tot[i] = wave_sum_dpp(lane < nw ? part[lane][i] : 0.0);
  }
  int* is_last_p = reinterpret_cast<int*>(part_lds + 16 * NACC);
#define is_last (*is_last_p)
  if (threadIdx.x == 0) {
    // Every word the workgroups share here (partial sums, ticket) is touched by agent-scope atomics only, on both sides: they are performed at
    // the memory side, so no cache write-back / invalidate is needed - two __threadfence() calls here cost ~7 us of the launch's tail.
    // The adds are drained (vmcnt) before the ticket is taken, so the workgroup that draws the last ticket reads complete sums.
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      if (tot[i] != 0.0) __hip_atomic_fetch_add(gacc + i, tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == (unsigned)nblk - 1);
  }
  __syncthreads();
  if (!is_last) return;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) tot[i] = __hip_atomic_load(gacc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < NACC; ++i) __hip_atomic_store(gacc + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rearm) __hip_atomic_store(rearm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double v = th.v, s = th.s, N = th.N, Dd = (double)D;
    // `alpha` may still be the unscaled solve x = P^-1 b (BCR path): alpha = x * alpha_scale; quadratic forms scale^2
    tot[AKA] *= alpha_scale * alpha_scale; tot[ADKA] *= alpha_scale * alpha_scale; tot[AAA] *= alpha_scale * alpha_scale;
    tot[BA] *= alpha_scale;
    if (logdets) {  // BCR path: log-dets come from the elimination, |c|^2 = b^T P^-1 b / s^2 = (b . alpha) / s
      tot[LOGK] = logdets[0];
      tot[LOGP] = logdets[2];
      tot[CC] = tot[BA] / s;
    }
    const double two_pi = 6.283185307179586476925286766559;
    double elbo = -0.5 * N * Dd * log(two_pi * s);
    elbo -= 0.5 * Dd * tot[LOGP];
    elbo += 0.5 * Dd * tot[LOGK];
    elbo -= 0.5 * yy / s;
    elbo += 0.5 * tot[CC];
    elbo -= 0.5 * N * v / s;
    elbo += 0.5 * tot[TRKA] / s;
    // G = 1/2 (D Kuu^-1 - D P^-1 - alpha alpha^T - Kuu^-1 A Kuu^-1 / s)   
    double d_l = 0.5 * (Dd * tot[SKDK] - Dd * tot[SPDK] - tot[ADKA] + tot[DTRKA] / s);
    double d_v = 0.5 * (-Dd * tot[SKK] / v + Dd * tot[SPK] / v + tot[AKA] / v + tot[TRKA] / (v * s)) - 0.5 * N / s;
    double s2 = s * s;
    double d_s = -0.5 * N * Dd / s + 0.5 * Dd * tot[SPA] / s2 + 0.5 * yy / s2 + 0.5 * tot[AAA] / s2 - tot[BA] / s2 +
                 0.5 * N * v / s2 - 0.5 * tot[TRKA] / s2;
    out[0] = elbo; out[1] = d_v; out[2] = d_l; out[3] = d_s;
    out[4] = tot[LOGK]; out[5] = tot[LOGP]; out[6] = tot[TRKA]; out[7] = tot[CC];
  }
#undef is_last
}

constexpr int FIN_LDS_DOUBLES = 16 * 14 + 2;

template <int KT>
__global__ __launch_bounds__(256) void elbo_finalize_kernel(
    const double* __restrict__ stats, const double* __restrict__ Kuu, const double* __restrict__ dK,
    const double* __restrict__ LK, const double* __restrict__ LP, const double* __restrict__ SK,
    const double* __restrict__ dSK, const double* __restrict__ SP, const double* __restrict__ c,
    const double* __restrict__ alpha, const double* __restrict__ logdets, long M, int k_rt, long D, ElboScalars th,
    double alpha_scale, double* __restrict__ gacc, unsigned* __restrict__ ticket, double* __restrict__ out) {
  __shared__ double part[FIN_LDS_DOUBLES];
  (void)k_rt;
  elbo_finalize_body<KT>((int)blockIdx.x, (int)gridDim.x, part, stats, Kuu, dK, LK, LP, SK, dSK, SP, c, alpha, logdets, M, D, th, alpha_scale,
                         gacc, ticket, nullptr, out);
}

// Fused launch of the planned path (band algorithms 0 / 3): ONE kernel per ELBO + gradient evaluation.
//   workgroup 0            the P chain (as elbo_bcr_data_kernel): P = A/s + Kuu formed in its gathers
//   workgroup 1            the backward pass of the Kuu chain from the host's factor table (neither chain reads the other's output)
//   workgroups 2 .. 2+H-1  helpers: first assemble Kuu and dKuu/dl (theta-only, one load round trip spread over H workgroups - a
//                          prepare kernel in front of the launch cost 5.5 us of critical path, the same loop inside workgroup 0 13 us),
//                          then (ELBO call) sleep on the chains' arrival counter and run the finalize reductions.
// Waits: workgroup 0 waits for the helpers' assembly (which depends on nothing), the helpers wait for the two chains: no cycle.
// The helpers need no LDS and are dispatched after workgroups 0 and 1.
struct FusedFin {
  const double* stats; ElboScalars th; double alpha_scale; double* gacc; unsigned* ticket; unsigned* arrived; unsigned* assembled; double* out;
  long D; int n_helpers; int finalize; int debug_no_assembly; int debug_stamps;
  double* mirror; unsigned long long mirror_seq;   // asvgp_result_mirror (matrix-core launch only)
  // matrix-core launch, P chain on TWO workgroups (bcr_mfma.hpp BmSplit): workgroup 2 is the right half, the helpers start at 3
  int split; double* xchg;                        // xchg: BM_XCHG doubles of message boxes
};
constexpr int FIN_GACC_PR = 21;                    // the right P workgroup's seven partial sums: gacc[21..27]
__device__ __forceinline__ void assemble_band_slice(const double* __restrict__ S, const double* __restrict__ coef, const double* __restrict__ dcoef, int n_terms,
                                                    long E, long e, double* __restrict__ Kuu, double* __restrict__ dK) {
#pragma clang fp contract(off)
  double sv[ASVGP_MAX_KUU_TERMS];
#pragma unroll
  for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) sv[t] = (t < n_terms) ? S[(long)t * E + e] : 0.0;
  double acc = coef[0] * sv[0], dacc = dcoef[0] * sv[0];
#pragma unroll
  for (int t = 1; t < ASVGP_MAX_KUU_TERMS; ++t)
    if (t < n_terms) { acc = acc + coef[t] * sv[t]; dacc = dacc + dcoef[t] * sv[t]; }   // inducing_features.py:12-44 rounding sequence
  Kuu[e] = acc;
  if (dK) dK[e] = dacc;
}
template <int K, bool BIG>
__global__ __launch_bounds__(BCR_THREADS) void elbo_chains_kernel(const double* S_static, KuuCoefs2 cf, double* Kuu, double* dK, const double* A, const double* b, int M,
                                                                  double* wsP, double* SP, double* x, double* logdets, int* info,
                                                                  double s, const double* tab, int n_rec, const int* node_rec,
                                                                  double* wsK, double* SK, double* dSK,
                                                                  unsigned long long* done_flag, unsigned long long seq, int D, int lds_doubles, FusedFin fin) {
  extern __shared__ double lds[];
  const long E = (long)(K + 1) * M;
  __shared__ int gave_up;
  if (threadIdx.x == 0) gave_up = 0;
  __syncthreads();
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      long spins = 0;                                 // bounded: helpers that never became resident must not hang the device
      while (__hip_atomic_load(fin.assembled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)fin.n_helpers) {   // (relaxed polls: ONE acquire fence behind the loop - an acquire per poll is a cache invalidation per poll)
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1L << 25)) { gave_up = 1; break; }
      }
      if (!gave_up) __hip_atomic_store(fin.assembled, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch (nobody else reads it)
    }
    __syncthreads();
    // gave up waiting: reported like a failed factorisation (negative), and STICKY - this workgroup neither solves (bcr_solve's epilogue
    // would overwrite the flag) nor signals its arrival; the host re-arms the workspace before the next launch (gpr.py)
    if (gave_up) { if (threadIdx.x == 0) atomicExch(info + 1, -1); return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    bcr_solve<double, K, 1, BandSumP, BIG>(BandSumP{A, Kuu, 1.0 / s}, b, M, wsP, lds, BandOut<double>{SP, nullptr}, x, logdets + 2, info + 1, nullptr, D);
    more_columns<K, BIG>(b, x, D, M, wsP, lds, lds_doubles);
  } else if (blockIdx.x == 1) {
    bcr_backward_pre<K>(tab, n_rec, node_rec, M, wsK, lds, BandOut<Dual>{SK, dSK}, logdets, info, done_flag, seq);
  } else {
    for (long e = (long)(blockIdx.x - 2) * blockDim.x + threadIdx.x; e < E; e += (long)fin.n_helpers * blockDim.x)
      assemble_band_slice(S_static, cf.c, cf.dc, cf.n, E, e, Kuu, dK);
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      __hip_atomic_fetch_add(fin.assembled, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!fin.finalize) return;
  if (blockIdx.x < 2) {                                       // chains: publish the bands, count the arrival
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      __hip_atomic_fetch_add(fin.arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (threadIdx.x == 0) {
    long spins = 0;                                   // bounded like the wait above; the bound is minutes of chain time
    while (__hip_atomic_load(fin.arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1L << 25)) { gave_up = 1; break; }
    }
  }
  __syncthreads();
  if (gave_up) { if (threadIdx.x == 0) atomicExch(info + 1, -1); return; }   // (no finalize from incomplete bands; the chains wrote info before)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  // (Kuu / dK slices of the OTHER helpers: each helper's release precedes its chain-arrival wait only in program order of that helper;
  //  the chains arrive tens of microseconds after every helper has published, and workgroup 0 has acquired all of them before it arrives)
  elbo_finalize_body<K>((int)blockIdx.x - 2, fin.n_helpers, lds, fin.stats, Kuu, dK, nullptr, nullptr, SK, dSK, SP, nullptr, x, logdets, (long)M, fin.D,
                        fin.th, fin.alpha_scale, fin.gacc, fin.ticket, fin.arrived, fin.out);
}

// The same fused launch with both chains on the matrix cores (bcr_mfma.hpp; k = 4, D = 1): 1024-thread workgroups, workgroup 0 the P chain
// (with fin.split: its left half + separator + root, workgroup 2 its right half - bcr_mfma.hpp BmSplit), workgroup 1 the planned Kuu
// backward pass.  NO helper workgroups: both chains form Kuu and dKuu / dl in closed form (Toeplitz interior values + boundary tables from
// the kernel arguments, copied to the LDS once), for the P chain's level-0 loads and for the traces - nobody assembles or waits for a band.
// The finalize is done by the chains themselves: each sums the traces / quadratic forms of ITS bands (the P workgroups: of their own
// columns, x from the solve's LDS image), stores them in its own slots with agent-scope stores and draws a ticket; whoever draws the last
// one (normally the left P workgroup) evaluates the bound and re-arms the slots.  The ticket word and the hand-over flags carry the
// launch's sequence number, so what an aborted launch left behind is never mistaken for this launch's.
// A wait that gives up (the other half never resident, the factor table never published) is sticky: the waiting workgroup sets
// info[1] = -1 and does NOT draw a ticket - no bound is written - and the host re-arms the workspace and re-issues the step through the
// multi-launch sweeps (gpr.py).
template <int K>
__global__ __launch_bounds__(BM_THREADS) void elbo_chains_mfma_kernel(KuuInterior ki, const double* S_static, KuuCoefs2 cf, double* Kuu, double* dK, const double* A, const double* b, int M,
                                                                      double* wsP, double* SP, double* x, double* logdets, int* info,
                                                                      double s, const double* tab, int n_rec, const int* node_rec,
                                                                      double* wsK, double* SK, double* dSK,
                                                                      unsigned long long* done_flag, unsigned long long seq,
                                                                      const unsigned long long* ready_flag, long spin_limit, FusedFin fin, const ThetaBox* box) {
  extern __shared__ double lds[];
  static_assert(K == BM_B, "matrix-core chains: bandwidth 4");
  const long E = (long)(K + 1) * M;
  // diagnostic (ASVGP_CHAIN_STAMPS): 100 MHz wall-clock stamps per role into logdets[8 + 4 * min(block, 2) ..]: start, chain done, end
  const unsigned long long t_start = fin.debug_stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
  auto rstamp = [&](int slot) {
    if (fin.debug_stamps && threadIdx.x == 0 && blockIdx.x <= 2)   // (slots 4..7: extra marks of the P workgroup's tail)
      logdets[(slot < 4 ? 8 + 4 * (int)blockIdx.x : 24) + slot] = (slot == 0) ? (double)(t_start & 0xffffffffull) : (double)((__builtin_amdgcn_s_memrealtime() - t_start) & 0xffffffffull);
  };
  rstamp(0);
  __shared__ int gave_up, kuu_ok;
  __shared__ double kuu_pref[4];
  if (threadIdx.x == 0) { gave_up = 0; kuu_ok = 0; }
  __syncthreads();
  // theta and what the host derives from it: kernel arguments, or - launch-ahead - the handle's pinned box once the host has filled it
  double s_v = s, asc_v = fin.alpha_scale;
  ElboScalars th_v = fin.th;
  if (box) {
    if (threadIdx.x == 0) {
      long spins = 0;
      for (;;) {
        const unsigned long long got = __hip_atomic_load(&box->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (got == seq) break;
        if (got == ~0ull || ++spins > spin_limit) { gave_up = 1; break; }   // (withdrawn, or the host never came: bounded like every wait here)
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    if (gave_up) { if (threadIdx.x == 0) atomicExch(info + 1, -1); return; }
    s_v = box->s; asc_v = box->alpha_scale;
    th_v.v = box->v; th_v.l = box->l; th_v.s = box->s; th_v.N = box->N;
  }
  const double* stats = fin.stats;
  enum { LOGK, LOGP, TRKA, DTRKA, SKDK, SPDK, SKK, SPK, SPA, CC, AKA, ADKA, AAA, BA, NACC };
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  const bool isP = blockIdx.x == 0 || (fin.split && blockIdx.x == 2);
  const bool isPR = fin.split && blockIdx.x == 2;
  // Kuu and dKuu / dl in closed form for BOTH chains' own traces (and the P chain's level-0 loads): interior values kdl, boundary tables
  __shared__ double kdl[16];
  __shared__ double bnd[2 * PRIOR_BND_DIAGS * PRIOR_BND], dkb[2 * 5 * KI_DKB];
  {
    if (isP) rstamp(1);
    if (threadIdx.x == 0) {
      if (box) {
#pragma unroll
        for (int i = 0; i < 5; ++i) { kdl[i] = box->k[i]; kdl[8 + i] = box->dk[i]; }
      } else {
        kdl[0] = ki.k[0]; kdl[1] = ki.k[1]; kdl[2] = ki.k[2]; kdl[3] = ki.k[3]; kdl[4] = ki.k[4];
        kdl[8] = ki.dk[0]; kdl[9] = ki.dk[1]; kdl[10] = ki.dk[2]; kdl[11] = ki.dk[3]; kdl[12] = ki.dk[4];
      }
    }
    {   // (ki is kernel argument 0: its tables are read in place, from the kernel-argument segment, ONCE - or from the box)
      const double* bnd_g = box ? box->bnd : reinterpret_cast<const double*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KuuInterior, bnd));
      const double* dkb_g = box ? box->dkb : reinterpret_cast<const double*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KuuInterior, dkb));
      if (threadIdx.x < 2 * PRIOR_BND_DIAGS * PRIOR_BND) bnd[threadIdx.x] = bnd_g[threadIdx.x];
      else if (threadIdx.x < 2 * PRIOR_BND_DIAGS * PRIOR_BND + 2 * 5 * KI_DKB) dkb[threadIdx.x - 2 * PRIOR_BND_DIAGS * PRIOR_BND] = dkb_g[threadIdx.x - 2 * PRIOR_BND_DIAGS * PRIOR_BND];
    }
    __syncthreads();
  }
  if (isP) {
    BmSplit bsp;
    if (fin.split) { bsp.half = isPR ? 1 : 0; bsp.xchg = fin.xchg; bsp.spin_limit = spin_limit; bsp.gave_up = &gave_up; bsp.tag = seq << 8; }
    bcr_mfma_solve<BandSumToep>(BandSumToep{A, 1.0 / s_v, (lds_cdouble_ptr)kdl, (lds_cdouble_ptr)bnd, ki.lo, ki.hi}, b, M, wsP, lds, SP, x, logdets + 2, info + 1, 1,
                                fin.debug_stamps ? (isPR ? fin.xchg + 128 : logdets + 32) : (double*)nullptr, bsp);   // (per-level cycle stamps: tools/mside_probe.py)
    rstamp(3);
    __syncthreads();                                           // (orders SP, the solve's last global stores, for the loop below)
    if (gave_up) { if (threadIdx.x == 0) atomicExch(info + 1, -1); return; }   // (the other half never answered: sticky, like the helpers' case)
    rstamp(4);
    if (fin.finalize) {
      // The P chain's traces and quadratic forms WITHOUT the helpers' bands (no wait, no loads from another XCD's L2): Kuu and dKuu / dl in
      // closed form (interior: one number per diagonal, kdl; the few boundary columns: the kernel-argument table for Kuu, a sum over the
      // static bands for dKuu), x from the solve's LDS image.  Loads left per column: SP (5), the A band (5), b (1).
      const double* xs_l = lds + (size_t)2 * ((((M + K - 1) / K) + 1) / 2 + 1) * 16;   // bcr_mfma_solve's xs (rows beyond M: 0)
      // (M <= 2048: two columns per thread, tid and tid + 1024; ALL their loads are issued before the first use - as a plain loop the
      //  second column's round trip started after the first column's arithmetic)
      static_assert(BM_THREADS == 1024, "two columns per thread");
      // two P workgroups: the left one takes the columns below the separator node's (it holds x up to the separator), the right one the rest
      int nbl = (M + K - 1) / K, lv = 0;
      while ((1 << lv) < nbl) ++lv;
      const long jsep = (long)(1 << (lv > 0 ? lv - 1 : 0)) * K;
      const long j0 = isPR ? jsep : 0, j1 = (fin.split && !isPR) ? jsep : (long)M;
      // the Kuu workgroup's sums (sent ~10 us ago): wave 1 of the finisher requests the message HERE and validates it behind the loop - at
      // the very end the round trip would sit on the critical path (not there yet / torn: the finisher polls as before)
      unsigned long long kw = 0ull;
      const bool kuu_reader = !isPR && (threadIdx.x >> 6) == 1;
      if (kuu_reader && (threadIdx.x & 63) <= 4)
        kw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(fin.gacc + 8) + (threadIdx.x & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      double sp[2][K + 1], av[2][K + 1], bj[2];
#pragma unroll
      for (int cI = 0; cI < 2; ++cI) {
        const long j = j0 + threadIdx.x + cI * BM_THREADS, jc = j < j1 ? j : j1 - 1;
#pragma unroll
        for (int r = 0; r <= K; ++r) { sp[cI][r] = SP[(long)r * M + jc]; av[cI][r] = stats[(long)r * M + jc]; }
        bj[cI] = stats[(long)(K + 1) * M + jc];
      }
#pragma unroll
      for (int cI = 0; cI < 2; ++cI) {
        const long j = j0 + threadIdx.x + cI * BM_THREADS;
        if (j >= j1) continue;
        const bool left = j < ki.lo, right = j >= ki.hi;
        const double xj = xs_l[j];
#pragma unroll
        for (int r = 0; r <= K; ++r) {
          const long o = (long)r * M + j;
          const double w2 = (r == 0) ? 1.0 : 2.0;
          double kv = kdl[r], dkv = kdl[8 + r];
          if (left || right) {                                 // (rare: at most PRIOR_BND columns on either side)
            kv = bnd[left ? (long)r * PRIOR_BND + j : (long)(PRIOR_BND_DIAGS + r) * PRIOR_BND + (j - ki.hi)];
            if (ki.dk_tab) {
              dkv = dkb[left ? (long)r * KI_DKB + j : (long)(5 + r) * KI_DKB + (j - ki.hi)];
            } else {                                           // (five dependent trips to the static bands: ~5 us on the launch's tail)
              dkv = 0.0;
              for (int t = 0; t < cf.n; ++t) dkv = fma(box ? box->dc[t] : cf.dc[t], S_static[(long)t * E + o], dkv);
            }
          }
          const double xx = (j + r < M) ? w2 * xj * xs_l[j + r] : 0.0;   // (below the matrix: the band entries there are 0 as well)
          acc[SPDK] = fma(w2 * sp[cI][r], dkv, acc[SPDK]);
          acc[SPK] = fma(w2 * sp[cI][r], kv, acc[SPK]);
          acc[SPA] = fma(w2 * sp[cI][r], av[cI][r], acc[SPA]);
          acc[AKA] = fma(kv, xx, acc[AKA]);
          acc[ADKA] = fma(dkv, xx, acc[ADKA]);
          acc[AAA] = fma(av[cI][r], xx, acc[AAA]);
        }
        acc[BA] = fma(bj[cI], xj, acc[BA]);
      }
      if (kuu_reader) {
        const int ln = threadIdx.x & 63;
        const unsigned long long x = bm_wave_xor(ln < 4 ? kw : 0ull) ^ ((seq << 8) | 4ull);
        const unsigned long long chk = (unsigned long long)__shfl((long long)kw, 4, 64);
        if (x == chk) { if (ln < 4) kuu_pref[ln] = __longlong_as_double((long long)kw); if (ln == 0) kuu_ok = 1; }
      }
    }
  } else {
    if (fin.debug_no_assembly) {   // test hook (ASVGP_DEBUG_NO_ASSEMBLY): this workgroup behaves as if its factor table never arrived
      if (threadIdx.x == 0) { for (long sp_ = 0; sp_ < spin_limit; ++sp_) __builtin_amdgcn_s_sleep(8); atomicExch(info + 1, -1); }
      return;
    }
    bcr_mfma_backward_pre(tab, n_rec, node_rec, M, wsK, lds, SK, dSK, logdets, info, done_flag, seq, ready_flag,
                          spin_limit < (1L << 20) ? spin_limit : (1L << 20), &gave_up);
    if (gave_up) { if (threadIdx.x == 0) atomicExch(info + 1, -1); return; }   // (the host never published the table: sticky, like the helpers' case)
    rstamp(1);
    if (fin.finalize) {
      // like the P chain's: no wait for the helpers' bands (the LAST ticket waits for them before it re-arms their counter), Kuu and
      // dKuu / dl in closed form, both columns' loads in flight together
      __syncthreads();                                         // (orders SK, dSK: this workgroup's own global stores)
      double sk[2][K + 1], dsk[2][K + 1], av[2][K + 1];
#pragma unroll
      for (int cI = 0; cI < 2; ++cI) {
        const long j = (long)threadIdx.x + cI * BM_THREADS, jc = j < M ? j : M - 1;
#pragma unroll
        for (int r = 0; r <= K; ++r) { sk[cI][r] = SK[(long)r * M + jc]; dsk[cI][r] = dSK[(long)r * M + jc]; av[cI][r] = stats[(long)r * M + jc]; }
      }
#pragma unroll
      for (int cI = 0; cI < 2; ++cI) {
        const long j = (long)threadIdx.x + cI * BM_THREADS;
        if (j >= M) continue;
        const bool left = j < ki.lo, right = j >= ki.hi;
#pragma unroll
        for (int r = 0; r <= K; ++r) {
          const long o = (long)r * M + j;
          const double w2 = (r == 0) ? 1.0 : 2.0;
          double kv = kdl[r], dkv = kdl[8 + r];
          if (left || right) {
            kv = bnd[left ? (long)r * PRIOR_BND + j : (long)(PRIOR_BND_DIAGS + r) * PRIOR_BND + (j - ki.hi)];
            if (ki.dk_tab) {
              dkv = dkb[left ? (long)r * KI_DKB + j : (long)(5 + r) * KI_DKB + (j - ki.hi)];
            } else {
              dkv = 0.0;
              for (int t = 0; t < cf.n; ++t) dkv = fma(box ? box->dc[t] : cf.dc[t], S_static[(long)t * E + o], dkv);
            }
          }
          acc[TRKA] = fma(w2 * sk[cI][r], av[cI][r], acc[TRKA]);
          acc[DTRKA] = fma(w2 * dsk[cI][r], av[cI][r], acc[DTRKA]);
          acc[SKDK] = fma(w2 * sk[cI][r], dkv, acc[SKDK]);
          acc[SKK] = fma(w2 * sk[cI][r], kv, acc[SKK]);
        }
      }
    }
  }
  if (!fin.finalize) { rstamp(2); return; }
  if (isP) rstamp(5);
  // ---- workgroup sums -> this chain's own slots (agent-scope stores: two writers, disjoint slots, no atomic adds) -> ticket; the last
  // ticket evaluates the bound (elbo_finalize_body's formulas).  Only the accumulators this chain owns are reduced.
  double* part0 = lds;                                         // [wave][8] wave totals
  const int lane = threadIdx.x & 63;
  double mine[8];
  mine[0] = isP ? acc[SPDK] : acc[TRKA];
  mine[1] = isP ? acc[SPK] : acc[DTRKA];
  mine[2] = isP ? acc[SPA] : acc[SKDK];
  mine[3] = isP ? acc[AKA] : acc[SKK];
  mine[4] = acc[ADKA]; mine[5] = acc[AAA]; mine[6] = acc[BA]; mine[7] = 0.0;
  const int nmine = isP ? 7 : 4;                               // (workgroup-uniform)
  __syncthreads();                                             // (the chains' LDS images are dead)
  // Eight sums over 64 lanes as a reduce-scatter butterfly: each exchange halves the values a lane still carries (4 + 2 + 1 exchanges),
  // three more finish the one that is left - 10 shuffles instead of 8 x 6 (the separate wave reductions were 3.9 us of the launch's tail).
  // The 16 wave results meet in a fixed order (no atomics): the same inputs give the same bits.
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
  double s4[4], s2[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double keep = b0 ? mine[4 + i] : mine[i], send = b0 ? mine[i] : mine[4 + i]; s4[i] = keep + dpp_move_f64<0xB1>(send); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { const double keep = b1 ? s4[2 + i] : s4[i], send = b1 ? s4[i] : s4[2 + i]; s2[i] = keep + dpp_move_f64<0x4E>(send); }
  double s1 = (b2 ? s2[1] : s2[0]) + dpp_xor4_f64(b2 ? s2[0] : s2[1], b2);      // (lanes ^ 1, ^ 2, ^ 4, ^ 8: DPP moves, no LDS crossbar)
  s1 += dpp_move_f64<0x128>(s1); s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
  // lane l < 8 now holds the wave total of accumulator idx(l) = 4 bit0 + 2 bit1 + bit2; it files it under its OWN lane number
  const int wv = threadIdx.x >> 6;
  if (lane < 8) part0[wv * 8 + lane] = s1;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  double t2 = part0[(2 * (lane >> 3)) * 8 + (lane & 7)] + part0[(2 * (lane >> 3) + 1) * 8 + (lane & 7)];   // waves 2j, 2j + 1
  t2 += __shfl_xor(t2, 8, 64); t2 += __shfl_xor(t2, 16, 64); t2 += __shfl_xor(t2, 32, 64);
  double red[7];                                               // accumulator i sits in lane rev3(i): 0, 4, 2, 6, 1, 5, 3
  red[0] = __shfl(t2, 0, 64); red[1] = __shfl(t2, 4, 64); red[2] = __shfl(t2, 2, 64); red[3] = __shfl(t2, 6, 64);
  red[4] = __shfl(t2, 1, 64); red[5] = __shfl(t2, 5, 64); red[6] = __shfl(t2, 3, 64);
  (void)nmine;
  double tot[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) tot[i] = 0.0;
  if (isP && threadIdx.x == 0) rstamp(6);
  // Who finishes is fixed: the left (or only) P workgroup.  The Kuu workgroup and the right P workgroup send their sums as one-trip messages
  // (bcr_mfma.hpp bm_msg_*: payload + a check word that carries this launch's sequence number - nothing an aborted launch left behind is
  // accepted); the finisher adds its own sums from registers.  No ticket, no flag, no drain.
  const unsigned long long mtag = seq << 8;
  double* box_kuu = fin.gacc + 8;                              // 4 sums + check word: gacc[8..12]
  double* box_pr = fin.gacc + FIN_GACC_PR;                     // 7 sums + check word: gacc[21..28]
  if (!isP || isPR) {
    double v = red[0];
#pragma unroll
    for (int i = 1; i < 7; ++i) v = (lane == i) ? red[i] : v;
    bm_msg_send(isPR ? box_pr : box_kuu, nmine, v, mtag | (isPR ? 5ull : 4ull), lane);
    if (threadIdx.x == 0) rstamp(2);
    return;
  }
  double kuu_v = 0.0, pr_v = 0.0;
  {
    bool ok = true;
    if (kuu_ok) kuu_v = lane < 4 ? kuu_pref[lane] : 0.0;         // (validated behind the trace loop)
    else ok = bm_msg_recv(box_kuu, 4, kuu_v, mtag | 4ull, lane, spin_limit);
    if (ok && fin.split) ok = bm_msg_recv(box_pr, 7, pr_v, mtag | 5ull, lane, spin_limit);
    if (!ok) { if (lane == 0) atomicExch(info + 1, -1); return; }   // (a chain that gave up never reports: no bound is written)
  }
  {
    const int pidx[7] = {SPDK, SPK, SPA, AKA, ADKA, AAA, BA};
#pragma unroll
    for (int i = 0; i < 7; ++i) tot[pidx[i]] = red[i] + (fin.split ? __shfl(pr_v, i, 64) : 0.0);
  }
  tot[TRKA] = __shfl(kuu_v, 0, 64); tot[DTRKA] = __shfl(kuu_v, 1, 64); tot[SKDK] = __shfl(kuu_v, 2, 64); tot[SKK] = __shfl(kuu_v, 3, 64);
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int i = 0; i < 14; ++i) __hip_atomic_store(fin.gacc + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm (the older kernels ADD into these slots; the message has been read)
  {
    const double v = th_v.v, sn = th_v.s, N = th_v.N;
    const double yy = stats[(long)(K + 1) * M + M];
    const double asc = asc_v;                                  // x = P^-1 b unscaled: alpha = x / s
    tot[AKA] *= asc * asc; tot[ADKA] *= asc * asc; tot[AAA] *= asc * asc; tot[BA] *= asc;
    // the log-determinants were written by the two chains' lane 0 before their atomics (same lanes: program order + the drain above)
    tot[LOGK] = __hip_atomic_load(logdets + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot[LOGP] = __hip_atomic_load(logdets + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (two P workgroups: the left one has added the right one's part)
    tot[CC] = tot[BA] / sn;
    const double two_pi = 6.283185307179586476925286766559;
    double elbo = -0.5 * N * log(two_pi * sn);
    elbo -= 0.5 * tot[LOGP];
    elbo += 0.5 * tot[LOGK];
    elbo -= 0.5 * yy / sn;
    elbo += 0.5 * tot[CC];
    elbo -= 0.5 * N * v / sn;
    elbo += 0.5 * tot[TRKA] / sn;
    const double d_l = 0.5 * (tot[SKDK] - tot[SPDK] - tot[ADKA] + tot[DTRKA] / sn);
    const double d_v = 0.5 * (-tot[SKK] / v + tot[SPK] / v + tot[AKA] / v + tot[TRKA] / (v * sn)) - 0.5 * N / sn;
    const double s2 = sn * sn;
    const double d_s = -0.5 * N / sn + 0.5 * tot[SPA] / s2 + 0.5 * yy / s2 + 0.5 * tot[AAA] / s2 - tot[BA] / s2 + 0.5 * N * v / s2 - 0.5 * tot[TRKA] / s2;
    double* out = fin.out;
    out[0] = elbo; out[1] = d_v; out[2] = d_l; out[3] = d_s;
    out[4] = tot[LOGK]; out[5] = tot[LOGP]; out[6] = tot[TRKA]; out[7] = tot[CC];
    if (fin.mirror) {
      // Pinned host memory: ten values, their checksum, the sequence number.  (No system-scope RELEASE here: that would write the
      // whole L2 back first, ~10 us.  System-scope stores go straight through, in no guaranteed order - so the checksum is BOUND to
      // the launch: sequence number + the ten values, added left to right.  A host that sees the new sequence number beside the
      // previous launch's values and checksum - all of them stale, hence consistent with each other - rejects them.)
      double* m = fin.mirror;
      const double mv[10] = {elbo, d_v, d_l, d_s, tot[LOGK], tot[LOGP], tot[TRKA], tot[CC],
                             (double)__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             (double)__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
      double sum = (double)fin.mirror_seq;
#pragma unroll
      for (int i = 0; i < 10; ++i) { __hip_atomic_store(m + i, mv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); sum += mv[i]; }
      __hip_atomic_store(m + 11, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(m + 10, (double)fin.mirror_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (no wait for the stores above: the
                                                               //  host accepts a sequence number only together with a matching sum, else polls on)
    }
  }
  rstamp(2);
}

static __global__ void scale_sub_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o,
                                 long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] - b[i];
}
static __global__ void scale_kernel(double* __restrict__ x, double f, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * f;
}

template <int K, bool TANGENT>
static int run_chains(Handle* h, const double* stats, const double* S, int kind, double v, double l, double s, long M, long D,
                      Ws w, int* info, hipStream_t st, bool& use_bcr, int part = 0, bool scale_alpha = true, FusedFin* fin = nullptr) {
  double t_enter = 0.0;
  if (debug_env().host_times) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); t_enter = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
  KuuCoefs2 cf;
  for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) cf.c[t] = cf.dc[t] = 0.0;
  int rc = asvgp_matern_coeffs(kind, v, l, cf.c, cf.dc, &cf.n);
  if (rc) return rc;
  const long E = (long)(K + 1) * M;
  const double* A = stats;
  const double* b = stats ? stats + E : nullptr;
  const long nb = (M + K - 1) / K;
  const int algo = h->band_algo;
  // the planned prior chain needs the LDS only for the P chain and the factor table
  const bool have_plan = h->plan && prior_plan_M(h->plan) == M && prior_plan_k(h->plan) == K && prior_plan_terms(h->plan) == cf.n;
  if ((algo == 3 || algo == 4) && !have_plan) { set_error("band algorithms 3 and 4 need asvgp_prior_plan_1d for this (M, k, kernel)"); return ASVGP_ERR_BAD_ARG; }
  size_t ldsK = sizeof(double) * (TANGENT ? bcr_lds_doubles<Dual, K, 0>(nb) : bcr_lds_doubles<double, K, 0>(nb));
  size_t ldsP = sizeof(double) * bcr_lds_doubles<double, K, 1>(nb);
  bool planned = have_plan && (algo == 0 || algo == 3 || algo == 4);
  size_t lds_bytes = planned ? ldsP : (ldsK > ldsP ? ldsK : ldsP);
  bool fits = lds_bytes <= 160 * 1024 - 256;
  bool big = false;   // BIG layout (bcr.hpp): twice the nodes, couplings in an L2-resident plane - M up to 4096 at k = 4
  constexpr bool HAS_BIG = (K <= 5);   // (the k = 6 BIG instantiation alone costs ~5 minutes of compile time)
  if (HAS_BIG && !fits) {
    size_t bK = sizeof(double) * (TANGENT ? bcr_lds_doubles<Dual, K, 0, true>(nb) : bcr_lds_doubles<double, K, 0, true>(nb));
    size_t bP = sizeof(double) * bcr_lds_doubles<double, K, 1, true>(nb);
    size_t bb = planned ? bP : (bK > bP ? bK : bP);
    if (bb <= 160 * 1024 - 256) { big = true; fits = true; lds_bytes = bb; }
  }
  use_bcr = (algo == 2 || algo == 3 || algo == 4 || (algo == 0 && fits));   // (D > 1: column 0 rides through the levels, the others replay the factors)
  planned = planned && use_bcr;
  if (part != 0 && (!use_bcr || planned)) {
    // split scheduling exists for the all-GPU BCR path only: the sweeps and the planned chain run as one unit in the data call
    if (part == 1) return ASVGP_OK;
    part = 0;
  }
  if (planned) {
    // ---- host: forward pass of the Kuu chain for this theta (prior_plan.cpp, ~20 us) into the next slot of the pinned ring.
    // Kuu / dKuu (finalize traces, P = A/s + Kuu in the P chain's gathers) are assembled by the chain workgroups themselves.
    // matrix-core chains: k = 4, one output column, the whole tree in one workgroup's LDS, a Toeplitz interior for the closed-form Kuu;
    // decided BEFORE a table slot is taken (a refused call must not leave a slot whose kernel never reports)
    const int n_rec = prior_plan_nrec(h->plan);
    bool use_mfma = false;
    KuuInterior ki;
    if constexpr (K == BM_B) {
      use_mfma = (algo == 0 || algo == 4) && D == 1 && nb <= 512 && TANGENT && fin != nullptr &&
                 sizeof(double) * bcr_mfma_lds_doubles(nb) <= 160 * 1024 && sizeof(double) * bcr_mfma_pre_lds_doubles(nb, n_rec) <= 160 * 1024;
      if (use_mfma) {
        prior_plan_interior_kuu(h->plan, cf.c, ki.k, &ki.lo, &ki.hi, ki.bnd);   // Kuu in closed form for the P chain's level-0 loads
        use_mfma = ki.hi > ki.lo;                                                // (none: the older kernel, which waits for the assembled band)
        if (use_mfma) {                                                          // dKuu / dl on the interior, for the P chain's own traces
          double bnd2[2 * PRIOR_BND_DIAGS * PRIOR_BND];
          long lo2 = 0, hi2 = 0;
          prior_plan_interior_kuu(h->plan, cf.dc, ki.dk, &lo2, &hi2, bnd2);
          ki.dk_tab = (ki.lo <= KI_DKB && M - ki.hi <= KI_DKB) ? 1 : 0;
          for (int sd = 0; sd < 2; ++sd)
            for (int d = 0; d <= K; ++d)
              for (int cI = 0; cI < KI_DKB; ++cI)
                ki.dkb[(sd * 5 + d) * KI_DKB + cI] = bnd2[(size_t)(sd * PRIOR_BND_DIAGS + d) * PRIOR_BND + cI];
        }
      }
    }
    if (algo == 4 && !use_mfma) {
      set_error("band algorithm 4 (matrix-core chains) needs k = 4, D = 1, M <= 2048, Toeplitz static bands and the ELBO + gradient entry point");
      return ASVGP_ERR_UNSUPPORTED;
    }
    if (h->ahead_req && !use_mfma) { set_error("launch-ahead exists for the matrix-core launch only (k = 4, D = 1, M <= 2048, planned prior chain)"); return ASVGP_ERR_UNSUPPORTED; }
    if (!use_mfma && !fits) { set_error("BCR needs %zu B of LDS", lds_bytes); return ASVGP_ERR_LDS_CAPACITY; }
    unsigned long long seq = 0;
    int slot = 0;
    double t_acq0 = 0.0, t_acq1 = 0.0;
    if (debug_env().host_times) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); t_acq0 = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
    double* tab = handle_table_acquire(h, &seq, &slot);
    if (debug_env().host_times) { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); t_acq1 = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
    // asvgp_set_prior_forward(h, 1): the forward pass runs on the GPU in double-double (prior_dd.hip), enqueued here in front of the
    // chains' launch; the table then sits in the handle's device ring and no launch waits for the host
    const bool gpu_fwd = h->prior_forward_gpu;
    const double* tab_k = h->tab_dev + (size_t)slot * h->slot_doubles;
    const unsigned long long* dd_ready = nullptr;
    if (gpu_fwd) {
      // the matrix-core launch: the pass on the handle's own stream BESIDE the chains' launch (its Kuu workgroup waits for the ready word as
      // it does for the host's - the P chain does not wait at all); the other launches: in front, on the same stream
      double* t = nullptr;
      rc = handle_prior_dd_forward(h, cf.c, cf.dc, slot, st, &t, nullptr, use_mfma ? seq : 0ull, use_mfma ? &dd_ready : nullptr);
      if (rc) return rc;
      tab_k = t;
    }
    size_t lds_pre = sizeof(double) * bcr_pre_lds_doubles(K, n_rec);
    if (lds_pre > lds_bytes) lds_bytes = lds_pre;
    FusedFin ff{};
    if (fin) { ff = *fin; ff.finalize = 1; fin->finalize = -1; }   // (-1: tells the caller that the finalize rode along)
    // 2 + 6 = 8 workgroups: one per XCD (workgroups are dealt to the XCDs round-robin).  Every workgroup of the launch reserves the
    // chains' ~100 KB of LDS, i.e. a CU of its own; with 10 workgroups two of them shared an XCD, and the second ELBO launch of the
    // in-flight schedule found no free CU there for a helper - its P chain then waited for the whole previous launch to finish
    ff.n_helpers = (int)((M + 255) / 256 < 6 ? (M + 255) / 256 : 6);
    ff.split = 0; ff.xchg = nullptr;
    ff.assembled = reinterpret_cast<unsigned*>(w.fin + 20);
    ff.debug_no_assembly = debug_env().no_assembly;
    ff.debug_stamps = debug_env().chain_stamps;   // test hook: the helpers never report -> the P chain gives up waiting
    const long spin_limit = debug_env().spin_limit;
    if constexpr (K == BM_B) {
      if (use_mfma) {
        size_t lb = sizeof(double) * bcr_mfma_lds_doubles(nb), lk = sizeof(double) * bcr_mfma_pre_lds_doubles(nb, n_rec);
        if (lk > lb) lb = lk;
        if (sizeof(double) * FIN_LDS_DOUBLES > lb) lb = sizeof(double) * FIN_LDS_DOUBLES;
        auto kern = elbo_chains_mfma_kernel<K>;
        static size_t lds_granted[16] = {0};                     // per device: the dynamic-LDS size already granted to this kernel
        size_t& granted = lds_granted[h->device & 15];
        if (lb > granted) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
          if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
          granted = lb;
        }
        if (h->mirror_dev) { ff.mirror = h->mirror_dev; ff.mirror_seq = ++h->mirror_seq; h->mirror_pending = ff.mirror_seq; }
        // the P chain on two workgroups where its wide levels are throughput-bound (bcr_mfma.hpp BmSplit); 8 workgroups in all, one per XCD
        if (nb >= 128 && !debug_env().no_split) {
          ff.split = 1;
          ff.xchg = w.LP;                                      // (the sequential sweeps' factor band: unused by this launch)
        }
        ff.n_helpers = 0;                                       // (both chains form Kuu / dKuu in closed form: nobody assembles the bands here)
        // The launch goes out FIRST: its ~8 us of dispatch latency, the helpers' assembly and the P chain (which needs only Kuu, not its
        // factors) run while this thread does the forward pass below; the Kuu workgroup waits on ready[slot] (bcr_mfma_backward_pre).
        const bool plan_first = debug_env().plan_first != 0;   // (measurement aid: the round-2 order, forward pass then launch)
        const bool host_times = debug_env().host_times != 0;   // (measurement aid: where this call's host time goes)
        auto now_us = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
        const double t0 = host_times ? now_us() : 0.0;
        if (plan_first && !gpu_fwd) (void)prior_plan_eval(h->plan, cf.c, cf.dc, tab);
        // launch-ahead: theta comes later, through the handle's pinned box ring (asvgp_elbo_publish_theta); nothing below runs the forward pass
        const ThetaBox* box_k = nullptr;
        if (h->ahead_req) {
          if (gpu_fwd || plan_first) { set_error("launch-ahead needs the host forward pass (asvgp_set_prior_forward(h, 0))"); return ASVGP_ERR_UNSUPPORTED; }
          static_assert(sizeof(ThetaBox) <= BOX_BYTES, "theta box");
          if (!h->box_host) {
            if (hipHostMalloc(&h->box_host, BOX_BYTES * TAB_SLOTS, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess ||
                hipHostGetDevicePointer(&h->box_dev, h->box_host, 0) != hipSuccess) {
              if (h->box_host) { (void)hipHostFree(h->box_host); h->box_host = nullptr; }
              set_error("launch-ahead: pinned allocation failed: %s", hipGetErrorString(hipGetLastError()));
              return ASVGP_ERR_HIP;
            }
            memset(h->box_host, 0, BOX_BYTES * TAB_SLOTS);
          }
          ThetaBox* bh = reinterpret_cast<ThetaBox*>(static_cast<char*>(h->box_host) + (size_t)slot * BOX_BYTES);
          __atomic_store_n(&bh->seq, 0ull, __ATOMIC_RELEASE);     // (not this launch's theta yet)
          box_k = reinterpret_cast<const ThetaBox*>(static_cast<const char*>(h->box_dev) + (size_t)slot * BOX_BYTES);
        }
        const bool to_worker = !plan_first && !gpu_fwd && h->fwd_worker && !h->ahead_req;
        if (to_worker) {                                       // the handle's worker thread starts on the table NOW, beside the launch call
          for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) { h->fwd.coef[t] = cf.c[t]; h->fwd.dcoef[t] = cf.dc[t]; }
          h->fwd.tab = tab; h->fwd.slot = slot; h->fwd.seq = seq;
          handle_post_forward(h);
        }
        hipLaunchKernelGGL(kern, dim3(2 + ff.split + ff.n_helpers), dim3(BM_THREADS), lb, st, ki, S, cf, w.Kuu, w.dK, A, b, (int)M, w.bcrP, w.SP, w.alpha, w.logdets, info, s,
                           tab_k, n_rec, h->node_rec_dev, w.bcrK, w.SK, w.dSK, h->done_dev + slot, seq,
                           gpu_fwd ? dd_ready : (plan_first ? (const unsigned long long*)nullptr : h->ready_dev + slot), spin_limit, ff, box_k);
        if (h->ahead_req) {
          h->ahead = Handle::PendingAhead{true, slot, seq, tab, kind, (long)ff.th.N};
          return check_launch("elbo chains (matrix cores, launched ahead of theta)");
        }
        if (gpu_fwd) return check_launch("elbo chains (matrix cores, forward pass on the GPU)");
        if (to_worker) return check_launch("elbo chains (matrix cores, forward pass on the worker thread)");
        const double t1 = host_times ? now_us() : 0.0;
        if (!plan_first && h->defer_forward) {                  // the caller runs the forward pass later (asvgp_prior_publish): e.g. after
          h->fwd.valid = true;                                   // enqueueing other work that should not wait 19 us behind it
          for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) { h->fwd.coef[t] = cf.c[t]; h->fwd.dcoef[t] = cf.dc[t]; }
          h->fwd.tab = tab; h->fwd.slot = slot; h->fwd.seq = seq;
          return check_launch("elbo chains (matrix cores, forward pass deferred)");
        }
        if (!plan_first) (void)prior_plan_eval(h->plan, cf.c, cf.dc, tab);   // a non-positive pivot is reported through `info` by the kernel
        if (host_times) {
          static double acc_launch = 0.0, acc_plan = 0.0, acc_pre = 0.0, acc_acq = 0.0, acc_a = 0.0;
          acc_acq += t_acq1 - t_acq0; acc_a += t_acq0 - t_enter;
          static long calls = 0;
          const double t2 = now_us();
          acc_pre += t0 - t_enter; acc_launch += t1 - t0; acc_plan += t2 - t1;
          if (++calls % 200 == 0) {
            fprintf(stderr, "[entry -> acquire %.1f us, table acquire %.1f us] ", acc_a / 200, acc_acq / 200);
            acc_a = acc_acq = 0.0;
            fprintf(stderr, "[asvgp host times, mean of 200 calls] entry -> launch call %.1f us | launch call (%s) %.1f us | %s %.1f us\n", acc_pre / 200,
                    plan_first ? "forward pass + hipLaunchKernel" : "hipLaunchKernel", acc_launch / 200, plan_first ? "-" : "forward pass", acc_plan / 200);
            acc_launch = acc_plan = acc_pre = 0.0;
          }
        }
        __atomic_store_n(h->ready_host + slot, seq, __ATOMIC_RELEASE);
      }
    }
    if (!use_mfma) {
    if (!gpu_fwd) {
      (void)prior_plan_eval(h->plan, cf.c, cf.dc, tab);   // a non-positive pivot is reported through `info` by the kernel
      __atomic_store_n(h->ready_host + slot, seq, __ATOMIC_RELEASE);
    }
    auto kern = big ? elbo_chains_kernel<K, HAS_BIG> : elbo_chains_kernel<K, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
    hipLaunchKernelGGL(kern, dim3(2 + ff.n_helpers), dim3(BCR_THREADS), lds_bytes, st, S, cf, w.Kuu, TANGENT ? w.dK : (double*)nullptr, A, b,
                       (int)M, w.bcrP, w.SP, w.alpha, w.logdets, info, s,
                       tab_k, n_rec, h->node_rec_dev, w.bcrK, w.SK, w.dSK,
                       h->done_dev + slot, seq, (int)D, (int)(lds_bytes / sizeof(double)), ff);
    }
    if (scale_alpha) {
      long n = M * D;
      hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.alpha, 1.0 / s, n);
    }
    return check_launch("elbo chains (planned prior)");
  }
  // Kuu/dKuu (theta only) are written by the prior part, P = A/s + Kuu by the data part (which re-forms Kuu in registers)
  const bool pfly = (part == 2) && use_bcr;   // data chain of the split call: P formed inside the BCR gathers
  if (!pfly)
    hipLaunchKernelGGL(elbo_prepare_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, S, cf, E, A, s,
                       part == 2 ? w.LK : w.Kuu, part == 2 ? (double*)nullptr : w.dK, part == 1 ? (double*)nullptr : w.P);
  if (h->sync_on && part == 1) (void)hipEventRecord(h->evK, st);
  if (h->sync_on && part == 2) (void)hipStreamWaitEvent(st, h->evK, 0);
  if (use_bcr) {
    if (!fits) { set_error("BCR forced but needs %zu B of LDS", lds_bytes); return ASVGP_ERR_LDS_CAPACITY; }
    hipError_t e = hipFuncSetAttribute(big ? reinterpret_cast<const void*>(elbo_bcr_kernel<K, TANGENT, HAS_BIG>) : reinterpret_cast<const void*>(elbo_bcr_kernel<K, TANGENT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
    if (part == 1 && TANGENT) {
      if constexpr (TANGENT) {
        auto kern = big ? elbo_bcr_prior_kernel<K, HAS_BIG> : elbo_bcr_prior_kernel<K, false>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
        hipLaunchKernelGGL(kern, dim3(1), dim3(BCR_THREADS), lds_bytes, st, w.Kuu, w.dK, (int)M, w.bcrK, w.SK, w.dSK, w.logdets, info,
                           debug_env().bcr_stamps);
      }
    } else if (pfly) {
      auto kern = big ? elbo_bcr_data_kernel<K, HAS_BIG> : elbo_bcr_data_kernel<K, false>;
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
      hipLaunchKernelGGL(kern, dim3(1), dim3(BCR_THREADS), lds_bytes, st, w.Kuu, A, b, (int)M, w.bcrP, w.SP, w.alpha, w.logdets, info,
                         debug_env().bcr_stamps, s, (int)D, (int)(lds_bytes / sizeof(double)));
    } else {
      auto kern = big ? elbo_bcr_kernel<K, TANGENT, HAS_BIG> : elbo_bcr_kernel<K, TANGENT, false>;
      hipLaunchKernelGGL(kern, dim3(part == 0 ? 2 : 1), dim3(BCR_THREADS), lds_bytes, st, w.Kuu, w.dK,
                         w.P, b, (int)M, w.bcrK, w.bcrP, w.SK, w.dSK, w.SP, w.alpha, w.logdets, info,
                         debug_env().bcr_stamps, part == 2 ? 1 : 0, (int)D, (int)(lds_bytes / sizeof(double)));
    }
    if (part == 1) {
      if (h->sync_on) (void)hipEventRecord(h->evP, st);
      return check_launch("elbo prior chain");
    }
  } else if (D == 1) {
    hipLaunchKernelGGL((elbo_factor_kernel<K, TANGENT, true>), dim3(2), dim3(64), 0, st, w.Kuu, w.dK, w.P, w.LK, w.dLK,
                       w.LP, b, w.c, (int)M, info);
    hipLaunchKernelGGL((elbo_inverse_kernel<K, TANGENT, true>), dim3(2), dim3(64), 0, st, w.LK, w.dLK, w.LP, w.SK,
                       w.dSK, w.SP, w.c, w.alpha, (int)M);
  } else {
    hipLaunchKernelGGL((elbo_factor_kernel<K, TANGENT, false>), dim3(2), dim3(64), 0, st, w.Kuu, w.dK, w.P, w.LK,
                       w.dLK, w.LP, b, w.c, (int)M, info);
    hipLaunchKernelGGL(elbo_trsv_kernel<K>, dim3((unsigned)D), dim3(64), 0, st, w.LP, (int)M, b, w.c, D, 0, 1.0);
    hipLaunchKernelGGL((elbo_inverse_kernel<K, TANGENT, false>), dim3(2), dim3(64), 0, st, w.LK, w.dLK, w.LP, w.SK,
                       w.dSK, w.SP, w.c, w.alpha, (int)M);
    hipLaunchKernelGGL(elbo_trsv_kernel<K>, dim3((unsigned)D), dim3(64), 0, st, w.LP, (int)M, w.c, w.alpha, D, 1, 1.0);
  }
  // c = L_P^-1 b / s (gpr.py:75), alpha = P^-1 b / s
  long n = M * D;
  if (!use_bcr) hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.c, 1.0 / s, n);
  if (!use_bcr || scale_alpha)   // (the BCR + ELBO path leaves alpha unscaled and lets the finalize apply 1/s)
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.alpha, 1.0 / s, n);
  return check_launch("elbo chains");
}

template <int K>
int ElboLauncher<K>::run(Handle* h, const double* stats, const double* S, int kind, double v, double l, double s, long N, long M, long D,
                         double* out, int* info, void* ws, hipStream_t st, int part) {
  {
    Ws w = carve(ws, M, K, D);
    bool bcr = false;
    h->mirror_pending = 0;                                    // (set again by the one launch that writes the mirror)
    ElboScalars th{v, l, s, (double)N};
    const int fin_blocks = (int)((M + 255) / 256 < 64 ? (M + 255) / 256 : 64);
    static_assert(BCR_THREADS == 256, "the fused finalize shares the chain kernel's workgroup size");
    FusedFin ff{stats, th, 1.0 / s, w.fin, reinterpret_cast<unsigned*>(w.fin + 16), reinterpret_cast<unsigned*>(w.fin + 18), nullptr, out, D, 0, 0};
    int rc = run_chains<K, true>(h, stats, S, kind, v, l, s, M, D, w, info, st, bcr, part, false, part == 1 ? nullptr : &ff);
    if (rc || (part == 1)) return rc;
    if (ff.finalize < 0) return check_launch("elbo_grad_1d (fused)");
    if (h->sync_on && part == 2 && bcr && h->evP) (void)hipStreamWaitEvent(st, h->evP, 0);
    hipLaunchKernelGGL(elbo_finalize_kernel<K>, dim3(fin_blocks), dim3(256), 0, st, stats, w.Kuu, w.dK, w.LK, w.LP, w.SK,
                       w.dSK, w.SP, w.c, w.alpha, bcr ? w.logdets : (const double*)nullptr, M, K, D, th,
                       bcr ? 1.0 / s : 1.0, w.fin, reinterpret_cast<unsigned*>(w.fin + 16), out);
    return check_launch("elbo_grad_1d");
  }
}
template <int K>
int PostLauncher<K>::run(Handle* h, const double* stats, const double* S, int kind, double v, double l, double s, long M, long D,
                         double* alpha, double* W, int* info, void* ws, hipStream_t st) {
  {
    Ws w = carve(ws, M, K, D);
    bool bcr = false;
    int rc = run_chains<K, false>(h, stats, S, kind, v, l, s, M, D, w, info, st, bcr);
    if (rc) return rc;
    long E = (long)(K + 1) * M;
    hipLaunchKernelGGL(scale_sub_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, w.SP, w.SK, W, E);
    hipError_t e = hipMemcpyAsync(alpha, w.alpha, sizeof(double) * M * D, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { set_error("hipMemcpyAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
    return check_launch("posterior_prepare_1d");
  }
}


// ---- band(Kuu^-1) with its lengthscale tangent as an operator of its own (asvgp_kuu_inverse_band_1d): the planned chain when the
// handle holds a plan, else the all-GPU Dual chain (block cyclic reduction when it fits the LDS, sequential sweeps otherwise)
template <int K>
__global__ __launch_bounds__(BCR_THREADS) void kuu_inverse_pre_kernel(const double* tab, int n_rec, const int* node_rec, int M, double* wsK,
                                                                      double* SK, double* dSK, double* logdet2, int* info,
                                                                      unsigned long long* done_flag, unsigned long long seq) {
  extern __shared__ double lds[];
  bcr_backward_pre<K>(tab, n_rec, node_rec, M, wsK, lds, BandOut<Dual>{SK, dSK}, logdet2, info, done_flag, seq);
}
template <int K>
__global__ __launch_bounds__(64) void kuu_inverse_sweep_kernel(const double* Kuu, const double* dK, double* LK, double* dLK, double* SK,
                                                               double* dSK, int M, int* info) {
  cholesky_sweep<Dual, K, false>(BandPtr<Dual>{Kuu, dK}, BandOut<Dual>{LK, dLK}, M, nullptr, nullptr, info);
  __syncthreads();
  takahashi_sweep<Dual, K, false>(BandPtr<Dual>{LK, dLK}, BandOut<Dual>{SK, dSK}, M, nullptr, nullptr);
}
static __global__ void kuu_logdet_kernel(const double* __restrict__ LK, const double* __restrict__ dLK, long M, double* __restrict__ out) {
  __shared__ double scratch[32];
  double a = 0.0, b = 0.0;
  for (long j = threadIdx.x; j < M; j += blockDim.x) { const double l = LK[j]; a += log(l * l); b += 2.0 * dLK[j] / l; }
  a = block_sum(a, scratch);
  b = block_sum(b, scratch + 16);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

template <int K>
int KuuInvLauncher<K>::run(Handle* h, const double* S, int kind, double v, double l, long M, double* Kuu, double* dK, double* SK,
                           double* dSK, double* logdet2, int* info, void* ws, hipStream_t st) {
  KuuCoefs2 cf;
  for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) cf.c[t] = cf.dc[t] = 0.0;
  int rc = asvgp_matern_coeffs(kind, v, l, cf.c, cf.dc, &cf.n);
  if (rc) return rc;
  Ws w = carve(ws, M, K, 1);
  const long E = (long)(K + 1) * M, nb = (M + K - 1) / K;
  hipLaunchKernelGGL(elbo_prepare_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, S, cf, E, (const double*)nullptr, 1.0, Kuu, dK,
                     (double*)nullptr);
  const bool have_plan = h->plan && prior_plan_M(h->plan) == M && prior_plan_k(h->plan) == K && prior_plan_terms(h->plan) == cf.n;
  if (have_plan && h->band_algo != 1 && h->band_algo != 2) {
    unsigned long long seq = 0;
    int slot = 0;
    double* tab = handle_table_acquire(h, &seq, &slot);
    const double* tab_k = h->tab_dev + (size_t)slot * h->slot_doubles;
    if (h->prior_forward_gpu) {
      double* t = nullptr;
      rc = handle_prior_dd_forward(h, cf.c, cf.dc, slot, st, &t);
      if (rc) return rc;
      tab_k = t;
    } else {
      (void)prior_plan_eval(h->plan, cf.c, cf.dc, tab);
    }
    const int n_rec = prior_plan_nrec(h->plan);
    const size_t lds_bytes = sizeof(double) * bcr_pre_lds_doubles(K, n_rec);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kuu_inverse_pre_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
    hipLaunchKernelGGL(kuu_inverse_pre_kernel<K>, dim3(1), dim3(BCR_THREADS), lds_bytes, st, tab_k, n_rec,
                       h->node_rec_dev, (int)M, w.bcrK, SK, dSK, logdet2, info, h->done_dev + slot, seq);
    return check_launch("kuu_inverse_band_1d (planned)");
  }
  const size_t ldsK = sizeof(double) * bcr_lds_doubles<Dual, K, 0>(nb);
  if (ldsK <= 160 * 1024 - 256 && h->band_algo != 1) {
    auto kern = elbo_bcr_prior_kernel<K, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsK);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return ASVGP_ERR_LDS_CAPACITY; }
    hipLaunchKernelGGL(kern, dim3(1), dim3(BCR_THREADS), ldsK, st, Kuu, dK, (int)M, w.bcrK, SK, dSK, logdet2, info, 0);
    return check_launch("kuu_inverse_band_1d (BCR)");
  }
  hipLaunchKernelGGL(kuu_inverse_sweep_kernel<K>, dim3(1), dim3(64), 0, st, Kuu, dK, w.LK, w.dLK, SK, dSK, (int)M, info);
  hipLaunchKernelGGL(kuu_logdet_kernel, dim3(1), dim3(256), 0, st, w.LK, w.dLK, M, logdet2);
  return check_launch("kuu_inverse_band_1d (sweeps)");
}

#if ASVGP_ELBO_ONLY_K == 4 && (!defined(ASVGP_ELBO_PART) || ASVGP_ELBO_PART == 1)
// Launch-ahead, second half: the theta of the launch that is out.  Coefficients, Kuu / dKuu in closed form (the reference's rounding
// sequence, as in run_chains), the scalars of the bound -> the slot's box, sequence number LAST; then the host forward pass of the Kuu
// chain and its ready word, exactly as the ordinary launch does behind its kernel launch.
int elbo_publish_theta(Handle* h, double v, double l, double s, bool withdraw) {
  if (!h->ahead.valid) { set_error("publish_theta: no launch is waiting for its theta"); return ASVGP_ERR_BAD_ARG; }
  const Handle::PendingAhead a = h->ahead;
  h->ahead.valid = false;
  ThetaBox* bh = reinterpret_cast<ThetaBox*>(static_cast<char*>(h->box_host) + (size_t)a.slot * BOX_BYTES);
  if (withdraw || !h->plan) {
    __atomic_store_n(&bh->seq, ~0ull, __ATOMIC_RELEASE);
    __atomic_store_n(h->ready_host + a.slot, a.seq, __ATOMIC_RELEASE);   // (nobody reads the table: the launch gives up at its first wait)
    return ASVGP_OK;
  }
  KuuCoefs2 cf;
  for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) cf.c[t] = cf.dc[t] = 0.0;
  const int rc = asvgp_matern_coeffs(a.kind, v, l, cf.c, cf.dc, &cf.n);
  if (rc) { __atomic_store_n(&bh->seq, ~0ull, __ATOMIC_RELEASE); return rc; }
  constexpr int K = 4;
  long lo = 0, hi = 0, lo2 = 0, hi2 = 0;
  double bnd2[2 * PRIOR_BND_DIAGS * PRIOR_BND];
  for (int i = 0; i < 8; ++i) { bh->k[i] = 0.0; bh->dk[i] = 0.0; }
  prior_plan_interior_kuu(h->plan, cf.c, bh->k, &lo, &hi, bh->bnd);
  prior_plan_interior_kuu(h->plan, cf.dc, bh->dk, &lo2, &hi2, bnd2);
  for (int sd = 0; sd < 2; ++sd)
    for (int d = 0; d <= K; ++d)
      for (int cI = 0; cI < KI_DKB; ++cI)
        bh->dkb[(sd * 5 + d) * KI_DKB + cI] = bnd2[(size_t)(sd * PRIOR_BND_DIAGS + d) * PRIOR_BND + cI];
  for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) bh->dc[t] = cf.dc[t];
  bh->s = s; bh->alpha_scale = 1.0 / s; bh->v = v; bh->l = l; bh->N = (double)a.N;
  __atomic_store_n(&bh->seq, a.seq, __ATOMIC_RELEASE);          // the kernel proceeds: P chain at once, the Kuu workgroup waits for the table
  if (h->defer_forward) {                                       // (asvgp_set_deferred_forward_pass(h, 1): the caller enqueues other work first, then asvgp_prior_publish)
    h->fwd.valid = true;
    for (int t = 0; t < ASVGP_MAX_KUU_TERMS; ++t) { h->fwd.coef[t] = cf.c[t]; h->fwd.dcoef[t] = cf.dc[t]; }
    h->fwd.tab = a.tab; h->fwd.slot = a.slot; h->fwd.seq = a.seq;
    return ASVGP_OK;
  }
  (void)prior_plan_eval(h->plan, cf.c, cf.dc, a.tab);           // a non-positive pivot is reported through `info` by the kernel
  __atomic_store_n(h->ready_host + a.slot, a.seq, __ATOMIC_RELEASE);
  return ASVGP_OK;
}
#endif

// ASVGP_ELBO_PART: 1 = ELBO + gradient launcher (tangent chains), 2 = posterior launcher, unset = both
#if !defined(ASVGP_ELBO_PART) || ASVGP_ELBO_PART == 1
template struct ElboLauncher<ASVGP_ELBO_ONLY_K>;
template struct KuuInvLauncher<ASVGP_ELBO_ONLY_K>;
#endif
#if !defined(ASVGP_ELBO_PART) || ASVGP_ELBO_PART == 2
template struct PostLauncher<ASVGP_ELBO_ONLY_K>;
#endif
#endif  // ASVGP_ELBO_ONLY_K

}  // namespace asvgp

#ifndef ASVGP_ELBO_ONLY_K
using namespace asvgp;

extern "C" size_t asvgp_elbo_workspace_bytes(int64_t M, int k, int64_t D) {
  if (M < 1 || k < 1 || D < 1) return 0;
  return sizeof(double) * ws_doubles(M, k, D);
}

static int elbo_args_ok(const void* stats, const void* S, const void* out, int64_t M, int k, int64_t D, double v,
                        double l, double s, void* ws, size_t wsb, int* info, const char* who) {
  if (!stats || !S || !out || !info || M < 1 || D < 1 || D > 65535 || !(v > 0.0) || !(l > 0.0) || !(s > 0.0)) {
    set_error("%s: bad argument", who);
    return ASVGP_ERR_BAD_ARG;
  }
  if (k < 1 || k > ASVGP_MAX_ORDER) { set_error("%s: bandwidth %d outside 1..%d", who, k, (int)ASVGP_MAX_ORDER); return ASVGP_ERR_UNSUPPORTED; }
  if (M > 0x3fffffff) { set_error("%s: M too large", who); return ASVGP_ERR_UNSUPPORTED; }
  if (!ws || wsb < asvgp_elbo_workspace_bytes(M, k, D)) { set_error("%s: workspace too small", who); return ASVGP_ERR_WORKSPACE; }
  return ASVGP_OK;
}

// asvgp_elbo_grad_1d + the host's read of its result in ONE call (an optimiser's evaluation, example.py:31-32): launch with the result
// mirror armed, then poll the mirror from C.  result[0..9] = [out[0..7], info[0], info[1]].
extern "C" int asvgp_elbo_grad_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                                  double lengthscale, double noise_variance, int64_t N, int64_t M, int k, int64_t D,
                                  double* out, int* info, void* workspace, size_t workspace_bytes, asvgp_stream_t stream);
extern "C" int asvgp_result_mirror_read(asvgp_handle_t handle, uint64_t token, double* result10, double timeout_seconds) {
  Handle* h = as_handle(handle);
  if (!result10) { set_error("result_mirror_read: bad argument"); return ASVGP_ERR_BAD_ARG; }
  handle_publish_forward(h);                 // (deferred-forward-pass mode: the launch must not be left waiting for its table)
  if (token == 0 || !h->mirror_host) return 1;                    // this launch does not write the mirror: read through the stream
  volatile const double* m = h->mirror_host;
  const double want = (double)token;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned long spins = 0;; ++spins) {
    if (m[10] == want) {
      double r[12];
      for (int i = 0; i < 12; ++i) r[i] = m[i];
      double sum = want;
      for (int i = 0; i < 10; ++i) sum += r[i];                   // (left to right, as the kernel adds)
      if (r[10] == want && (sum == r[11] || r[11] != r[11])) {     // (NaN results pass through)
        for (int i = 0; i < 10; ++i) result10[i] = r[i];
        return ASVGP_OK;
      }
    }
    __builtin_ia32_pause();
    if ((spins & 1023) == 1023) {
      struct timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_seconds) return 1;   // (a launch that gave up never writes it)
    }
  }
}
extern "C" int asvgp_elbo_grad_ahead_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, int64_t N, int64_t M,
                                        int k, int64_t D, double* out, int* info, void* workspace, size_t workspace_bytes, asvgp_stream_t stream) {
  Handle* h = as_handle(handle);
  if (h->ahead.valid) { set_error("elbo_grad_ahead_1d: the previous launch still waits for its theta (asvgp_elbo_publish_theta)"); return ASVGP_ERR_BAD_ARG; }
  if (!h->mirror_host) { const double* p = nullptr; const int rcm = asvgp_result_mirror(handle, 1, &p); if (rcm) return rcm; }
  h->ahead_req = true;
  const int rc = asvgp_elbo_grad_1d(handle, stats, static_bands, kind, 1.0, 1.0, 1.0, N, M, k, D, out, info, workspace, workspace_bytes, stream);
  h->ahead_req = false;
  return rc;
}
extern "C" int asvgp_elbo_publish_theta(asvgp_handle_t handle, double variance, double lengthscale, double noise_variance) {
  if (!(variance > 0.0) || !(lengthscale > 0.0) || !(noise_variance > 0.0)) { set_error("elbo_publish_theta: bad argument"); return ASVGP_ERR_BAD_ARG; }
  return elbo_publish_theta(as_handle(handle), variance, lengthscale, noise_variance, false);
}
extern "C" int asvgp_elbo_grad_host_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                                       double lengthscale, double noise_variance, int64_t N, int64_t M, int k, int64_t D,
                                       double* out, int* info, void* workspace, size_t workspace_bytes, asvgp_stream_t stream,
                                       double* result10, double timeout_seconds) {
  Handle* h = as_handle(handle);
  if (!h->mirror_host) { const double* p = nullptr; const int rcm = asvgp_result_mirror(handle, 1, &p); if (rcm) return rcm; }
  const int rc = asvgp_elbo_grad_1d(handle, stats, static_bands, kind, variance, lengthscale, noise_variance, N, M, k, D, out, info, workspace,
                                    workspace_bytes, stream);
  if (rc) return rc;
  return asvgp_result_mirror_read(handle, h->mirror_pending, result10, timeout_seconds);
}

extern "C" int asvgp_elbo_grad_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                                  double lengthscale, double noise_variance, int64_t N, int64_t M, int k, int64_t D,
                                  double* out, int* info, void* workspace, size_t workspace_bytes,
                                  asvgp_stream_t stream) {
  int rc = elbo_args_ok(stats, static_bands, out, M, k, D, variance, lengthscale, noise_variance, workspace,
                        workspace_bytes, info, "elbo_grad_1d");
  if (rc) return rc;
  Handle* h = as_handle(handle);
  { const int rcf = handle_flush_phi_reduce(h, stats, as_stream(stream)); if (rcf) return rcf; }   // (a deferred Phi reduce into THIS buffer goes first)
  hipStream_t st = as_stream(stream);
#define ELBO_CASE(KK) case KK: return ElboLauncher<KK>::run(h, stats, static_bands, kind, variance, lengthscale, noise_variance, (long)N, (long)M, (long)D, out, info, workspace, st, 0);
  switch (k) { ELBO_CASE(1) ELBO_CASE(2) ELBO_CASE(3) ELBO_CASE(4) ELBO_CASE(5) ELBO_CASE(6) }
#undef ELBO_CASE
  return ASVGP_ERR_UNSUPPORTED;
}

extern "C" int asvgp_elbo_prior_chain_1d(asvgp_handle_t handle, const double* static_bands, int kind, double variance, double lengthscale,
                                         double noise_variance, int64_t M, int k, int64_t D, int* info, void* workspace,
                                         size_t workspace_bytes, asvgp_stream_t stream) {
  int rc = elbo_args_ok(static_bands, static_bands, static_bands, M, k, D, variance, lengthscale, noise_variance, workspace,
                        workspace_bytes, info, "elbo_prior_chain_1d");
  if (rc) return rc;
  Handle* h = as_handle(handle);
  hipStream_t st = as_stream(stream);
#define ELBO_CASE(KK) case KK: return ElboLauncher<KK>::run(h, nullptr, static_bands, kind, variance, lengthscale, noise_variance, 0, (long)M, (long)D, nullptr, info, workspace, st, 1);
  switch (k) { ELBO_CASE(1) ELBO_CASE(2) ELBO_CASE(3) ELBO_CASE(4) ELBO_CASE(5) ELBO_CASE(6) }
#undef ELBO_CASE
  return ASVGP_ERR_UNSUPPORTED;
}

extern "C" int asvgp_elbo_data_chain_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                                        double lengthscale, double noise_variance, int64_t N, int64_t M, int k, int64_t D,
                                        double* out, int* info, void* workspace, size_t workspace_bytes,
                                        asvgp_stream_t stream) {
  int rc = elbo_args_ok(stats, static_bands, out, M, k, D, variance, lengthscale, noise_variance, workspace,
                        workspace_bytes, info, "elbo_data_chain_1d");
  if (rc) return rc;
  Handle* h = as_handle(handle);
  { const int rcf = handle_flush_phi_reduce(h, stats, as_stream(stream)); if (rcf) return rcf; }   // (a deferred Phi reduce into THIS buffer goes first)
  hipStream_t st = as_stream(stream);
#define ELBO_CASE(KK) case KK: return ElboLauncher<KK>::run(h, stats, static_bands, kind, variance, lengthscale, noise_variance, (long)N, (long)M, (long)D, out, info, workspace, st, 2);
  switch (k) { ELBO_CASE(1) ELBO_CASE(2) ELBO_CASE(3) ELBO_CASE(4) ELBO_CASE(5) ELBO_CASE(6) }
#undef ELBO_CASE
  return ASVGP_ERR_UNSUPPORTED;
}

extern "C" int asvgp_kuu_inverse_band_1d(asvgp_handle_t handle, const double* static_bands, int kind, double variance, double lengthscale,
                                         int64_t M, int k, double* Kuu, double* dKuu_dl, double* S, double* dS_dl, double* logdet2, int* info,
                                         void* workspace, size_t workspace_bytes, asvgp_stream_t stream) {
  if (!static_bands || !Kuu || !dKuu_dl || !S || !dS_dl || !logdet2 || !info || M < 1 || !(variance > 0.0) || !(lengthscale > 0.0)) {
    set_error("kuu_inverse_band_1d: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  if (k < 1 || k > ASVGP_MAX_ORDER) { set_error("kuu_inverse_band_1d: bandwidth %d outside 1..%d", k, (int)ASVGP_MAX_ORDER); return ASVGP_ERR_UNSUPPORTED; }
  if (!workspace || workspace_bytes < asvgp_elbo_workspace_bytes(M, k, 1)) { set_error("kuu_inverse_band_1d: workspace too small"); return ASVGP_ERR_WORKSPACE; }
  Handle* h = as_handle(handle);
  hipStream_t st = as_stream(stream);
#define KINV_CASE(KK) case KK: return KuuInvLauncher<KK>::run(h, static_bands, kind, variance, lengthscale, (long)M, Kuu, dKuu_dl, S, dS_dl, logdet2, info, workspace, st);
  switch (k) { KINV_CASE(1) KINV_CASE(2) KINV_CASE(3) KINV_CASE(4) KINV_CASE(5) KINV_CASE(6) }
#undef KINV_CASE
  return ASVGP_ERR_UNSUPPORTED;
}

extern "C" int asvgp_posterior_prepare_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                                          double lengthscale, double noise_variance, int64_t M, int k, int64_t D,
                                          double* alpha, double* W, int* info, void* workspace, size_t workspace_bytes,
                                          asvgp_stream_t stream) {
  int rc = elbo_args_ok(stats, static_bands, alpha, M, k, D, variance, lengthscale, noise_variance, workspace,
                        workspace_bytes, info, "posterior_prepare_1d");
  if (rc) return rc;
  if (!W) { set_error("posterior_prepare_1d: bad argument"); return ASVGP_ERR_BAD_ARG; }
  Handle* h = as_handle(handle);
  { const int rcf = handle_flush_phi_reduce(h, stats, as_stream(stream)); if (rcf) return rcf; }   // (a deferred Phi reduce into THIS buffer goes first)
  hipStream_t st = as_stream(stream);
#define POST_CASE(KK) case KK: return PostLauncher<KK>::run(h, stats, static_bands, kind, variance, lengthscale, noise_variance, (long)M, (long)D, alpha, W, info, workspace, st);
  switch (k) { POST_CASE(1) POST_CASE(2) POST_CASE(3) POST_CASE(4) POST_CASE(5) POST_CASE(6) }
#undef POST_CASE
  return ASVGP_ERR_UNSUPPORTED;
}

#endif  // !ASVGP_ELBO_ONLY_K
