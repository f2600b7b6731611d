// Shared device/host helpers for the gfx950 ASVGP kernels (wave64, fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/asvgp_hip.h"

namespace asvgp {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

static inline hipStream_t as_stream(asvgp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ------------------------------------------------------------------------------------------------
// Cardinal B-spline pieces N_k(t+i), t in [0,1], i = 0..k, as monomial coefficients (exact integer
// numerators / k!).  Same mathematical objects as b1..b_{k+1} of basis.py:133-136 ... 668-676.
// ------------------------------------------------------------------------------------------------
constexpr long long ipow(long long b, int e) { long long r = 1; for (int i = 0; i < e; ++i) r *= b; return r; }
constexpr long long binom(int n, int r) {
  if (r < 0 || r > n) return 0;
  long long v = 1;
  for (int i = 1; i <= r; ++i) v = v * (n - r + i) / i;
  return v;
}
constexpr long long fact(int n) { long long v = 1; for (int i = 2; i <= n; ++i) v *= i; return v; }

// coefficient of t^p in the DERIV-th t-derivative of piece i of the order-K cardinal B-spline
template <int K, int DERIV>
constexpr double piece_coef(int i, int p) {
  // base polynomial coefficient of t^(p+DERIV), times falling factorial
  int q = p + DERIV;
  if (q > K) return 0.0;
  long long num = 0;
  for (int j = 0; j <= i; ++j) {
    long long term = binom(K + 1, j) * binom(K, q) * ipow(i - j, K - q);
    num += (j & 1) ? -term : term;
  }
  long long ff = 1;
  for (int d = 0; d < DERIV; ++d) ff *= (q - d);
  return (double)(num * ff) / (double)fact(K);
}

// vals[i] = N_K^(DERIV)(t + i), Horner in t (t-derivative: caller scales by delta^-DERIV)
template <int K, int DERIV = 0>
__device__ __forceinline__ void bspline_pieces(double t, double (&vals)[K + 1]) {
#pragma unroll
  for (int i = 0; i <= K; ++i) {
    double acc = piece_coef<K, DERIV>(i, K - DERIV);
#pragma unroll
    for (int p = K - DERIV - 1; p >= 0; --p) acc = fma(acc, t, piece_coef<K, DERIV>(i, p));
    vals[i] = acc;
  }
}

// basis.py:58-59: idx = max(#{mesh < x} - 1, 0), a table search (floor guess + fix-up against the table).
// Clamped to n_mesh-2 so rows idx..idx+k stay inside [0, M) for x beyond b (the reference asserts a < x < b).
__device__ __forceinline__ int neighbour_index(double x, const double* mesh, int n_mesh, double m0, double inv_delta) {
  double g = floor((x - m0) * inv_delta);
  int i = (g < 0.0) ? 0 : ((g > (double)(n_mesh - 2)) ? (n_mesh - 2) : (int)g);  // NaN -> 0
  while (i > 0 && !(mesh[i] < x)) --i;                 // x on knot j>0 belongs to interval j-1
  while (i < n_mesh - 2 && mesh[i + 1] < x) ++i;
  return i;
}

// ------------------------------------------------------------------------------------------------
// wave64 lane helpers for fp64 values (two 32-bit halves)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane) {  // lane: wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_f64, no return
}

__device__ __forceinline__ void lds_add_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u64, no return
}
// fixed-point conversion (see phi_pass.hip): p >= ~0 (tiny negative rounding noise is fine), p * 2^(s0+g) < 2^51;  chi = high word of C
__device__ __forceinline__ unsigned long long fx_convert(double p, int chi) {
  const double q = p + __hiloint2double(chi, 0);
  return ((unsigned long long)(unsigned)(__double2hiint(q) - chi) << 32) | (unsigned)__double2loint(q);
}

// wave64 sum on the VALU only (DPP row shifts + row broadcasts, no LDS crossbar): every lane gets the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_step_add(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi, lo);
}
// One DPP move of a double (two 32-bit halves): lane exchanges on the VALU instead of the LDS crossbar (__shfl_xor = ds_bpermute: ~100
// cycles of latency per step of a dependent chain).  quad_perm 0xB1 = lane ^ 1, 0x4E = lane ^ 2; row_ror:8 (0x128) = lane ^ 8; lane ^ 4 is
// row_ror:12 (0x12C) for lanes with bit 2 clear and row_ror:4 (0x124) for the others.
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_xor4_f64(double x, bool bit2) {
  const double up = dpp_move_f64<0x12C>(x), dn = dpp_move_f64<0x124>(x);
  return bit2 ? dn : up;
}

__device__ __forceinline__ double wave_sum_dpp(double v) {
  v = dpp_step_add<0x111, 0xf>(v);   // row_shr:1
  v = dpp_step_add<0x112, 0xf>(v);   // row_shr:2
  v = dpp_step_add<0x114, 0xf>(v);   // row_shr:4
  v = dpp_step_add<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of each row holds the row sum
  v = dpp_step_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_step_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return readlane_f64(v, 63);
}

// block-wide sum of one double per thread; result valid in thread 0.  scratch: >= blockDim/64 doubles of LDS
// (every thread of the workgroup calls it - the barriers demand that anyway - so the wavefronts are full and the
// VALU-only DPP reduction applies; it is ~3x cheaper than the ds_bpermute tree)
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum_dpp(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double r = 0.0;
  if (w == 0) {
    r = (lane < nw) ? scratch[lane] : 0.0;
    r = wave_sum_dpp(r);
  }
  return r;
}

// ------------------------------------------------------------------------------------------------
// forward-mode dual number (value, tangent) - one direction (d/d lengthscale), SURVEY App. A-6
// ------------------------------------------------------------------------------------------------
struct Dual {
  double v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, fma(a.v, b.d, a.d * b.v)}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }

// 1/sqrt(x) to ~1 ulp from v_rsq_f64 + two Newton steps (no IEEE sqrt/div sequences on the dependent chain)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

template <typename T> struct Num;
template <> struct Num<double> {
  static __device__ __forceinline__ double zero() { return 0.0; }
  static __device__ __forceinline__ double make(double v, double) { return v; }
  static __device__ __forceinline__ double val(double a) { return a; }
  static __device__ __forceinline__ double tan(double) { return 0.0; }
  static __device__ __forceinline__ double sqrt_(double a) { return sqrt(a); }
  static __device__ __forceinline__ double inv(double a) { return 1.0 / a; }
  // (sqrt(a), 1/sqrt(a)) with one rsqrt: s = a*y corrected by one Newton step on s
  static __device__ __forceinline__ void sqrt_inv(double a, double& s, double& inv) {
    double y = fast_rsqrt(a);
    double t = a * y;
    s = fma(0.5 * y, fma(-t, t, a), t);
    inv = y;
  }
  static __device__ __forceinline__ double rl(double a, int lane) { return readlane_f64(a, lane); }
  static __device__ __forceinline__ double sel(bool c, double a, double b) { return c ? a : b; }
  // a - b*c
  static __device__ __forceinline__ double nfma(double b, double c, double a) { return fma(-b, c, a); }
};
template <> struct Num<Dual> {
  static __device__ __forceinline__ Dual zero() { return {0.0, 0.0}; }
  static __device__ __forceinline__ Dual make(double v, double d) { return {v, d}; }
  static __device__ __forceinline__ double val(Dual a) { return a.v; }
  static __device__ __forceinline__ double tan(Dual a) { return a.d; }
  static __device__ __forceinline__ Dual sqrt_(Dual a) { double s = sqrt(a.v); return {s, a.d / (2.0 * s)}; }
  static __device__ __forceinline__ Dual inv(Dual a) { double r = 1.0 / a.v; return {r, -a.d * r * r}; }
  static __device__ __forceinline__ void sqrt_inv(Dual a, Dual& s, Dual& inv) {
    double sv, y;
    Num<double>::sqrt_inv(a.v, sv, y);
    double ds = 0.5 * a.d * y;          // d sqrt(a) = a' / (2 sqrt(a))
    s = {sv, ds};
    inv = {y, -ds * y * y};             // d (1/s) = -s' / s^2
  }
  static __device__ __forceinline__ Dual rl(Dual a, int lane) { return {readlane_f64(a.v, lane), readlane_f64(a.d, lane)}; }
  static __device__ __forceinline__ Dual sel(bool c, Dual a, Dual b) { return {c ? a.v : b.v, c ? a.d : b.d}; }
  static __device__ __forceinline__ Dual nfma(Dual b, Dual c, Dual a) {
    return {fma(-b.v, c.v, a.v), fma(-b.v, c.d, fma(-b.d, c.v, a.d))};
  }
};

}  // namespace asvgp
