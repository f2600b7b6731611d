// 2-D Kronecker (tensor-product) path: GPR_kron of gpr.py:239-359 without ever densifying.
//
// Replaces (reference): kronecker.make_kvs_sparse kronecker.py:7-33 + Kuf@y, Kuf@Kuf.T, todense gpr.py:266-272,
// the dense tf.linalg.cholesky / triangular_solve / cholesky_solve of gpr.py:293-307 and 319-326.
// Row index of basis pair (i1, i2) is i1*m2 + i2 (dim-0 major, = make_kvs_two_sparse).  A = Phi Phi^T is a
// "band of bands": A[(i1+d1)*m2 + i2+d2, i1*m2 + i2] != 0 only for |d1| <= k, |d2| <= k.  Only the lower triangle is
// kept: offsets (d1 = 0, d2 = 0..k) and (d1 = 1..k, d2 = -k..k), n_off = k(2k+1) + k + 1, stored as
// Ablk[off][col] (n_off x M_tot).  P = Kuu + A/s is then a plain band matrix of bandwidth bw = k*m2 + k which is
// factorised by a blocked right-looking band Cholesky on column-major band storage Pb[col*LD + (row-col)], LD = bw+1.
#include "asvgp_common.hpp"

namespace asvgp {

__host__ __device__ inline int kron_noff(int k) { return k * (2 * k + 1) + k + 1; }
__host__ __device__ inline int kron_off(int k, int d1, int d2) {
  return d1 == 0 ? d2 : (k + 1) + (d1 - 1) * (2 * k + 1) + (d2 + k);
}

// ---------------------------------------------------------------------------------------------------------
// Fused Khatri-Rao Phi pass: one thread per point, (k+1)^2 products in registers, fp64 global atomics into the
// L2-resident block band (3-8 MB) - first correct version; see DESIGN.md for the sorted/per-cell plan.
// ---------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void phi_kron2d_kernel(const double* __restrict__ X, const double* __restrict__ y,
                                                         long N, const double* __restrict__ mesh1, int n1, double id1,
                                                         int m1, const double* __restrict__ mesh2, int n2, double id2,
                                                         int m2, double* __restrict__ Ablk, double* __restrict__ rhs,
                                                         double* __restrict__ yy_out) {
  __shared__ double scratch[16];
  const long Mtot = (long)m1 * m2;
  double yy = 0.0;
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
    const double2 xv = *reinterpret_cast<const double2*>(X + 2 * n);
    const double yv = y[n];
    const int i1 = neighbour_index(xv.x, mesh1, n1, mesh1[0], id1);
    const int i2 = neighbour_index(xv.y, mesh2, n2, mesh2[0], id2);
    double v1[K + 1], v2[K + 1];
    bspline_pieces<K>((xv.x - mesh1[i1]) * id1, v1);
    bspline_pieces<K>((xv.y - mesh2[i2]) * id2, v2);
    yy = fma(yv, yv, yy);
#pragma unroll
    for (int a = 0; a <= K; ++a)
#pragma unroll
      for (int b = 0; b <= K; ++b) {
        const double w = v1[a] * v2[b];
        const long row = (long)(i1 + K - a) * m2 + (i2 + K - b);
        __hip_atomic_fetch_add(rhs + row, w * yv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int a2 = a; a2 <= K; ++a2)  // d1 = a2 - a >= 0  (row block >= col block)
#pragma unroll
          for (int b2 = 0; b2 <= K; ++b2) {
            const int d1 = a2 - a, d2 = b2 - b;
            if (d1 == 0 && d2 < 0) continue;
            const long col = (long)(i1 + K - a2) * m2 + (i2 + K - b2);
            __hip_atomic_fetch_add(Ablk + (long)kron_off(K, d1, d2) * Mtot + col, w * v1[a2] * v2[b2], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
          }
      }
  }
  double tot = block_sum(yy, scratch);
  if (threadIdx.x == 0 && tot != 0.0) __hip_atomic_fetch_add(yy_out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------
// Cell-sorted Khatri-Rao Phi pass.  The caller sorts the points by 2-D cell id c = i1 * (n2 - 1) + i2 (asvgp_kron_cell_index
// + a device radix sort) and passes the cell offsets.  All points of a cell touch the SAME (k+1)^2 (k+1)^2 / ~2 block-band
// entries, so one workgroup per cell stages, per point, the products v1[a] v1[a2] (a <= a2), v2[b] v2[b2] and v1[a], v2[b] y
// in the LDS, and thread t owns output entry t: it walks the staged points with two LDS reads and one FMA each and issues
// ONE global atomic per entry and cell - 136 atomics per cell instead of per point (k = 3).
// ---------------------------------------------------------------------------------------------------------
template <int K> struct KronOut {
  static constexpr int NP1 = (K + 1) * (K + 2) / 2;      // (a, a2), a <= a2
  static constexpr int NP2 = (K + 1) * (K + 1);          // (b, b2)
  static constexpr int NREC = NP1 + NP2 + 2 * (K + 1);   // doubles staged per point
  static constexpr int NBAND = NP1 * NP2 - (K + 1) * (K * (K + 1) / 2);   // minus (d1 = 0, d2 < 0)
  static constexpr int NOUT = NBAND + NP2;               // + the (k+1)^2 rhs entries
};
template <int K> constexpr int kron_chunk() { return K <= 4 ? 128 : 64; }   // points staged per pass (static LDS <= 64 KB)

template <int K>
__global__ __launch_bounds__(256) void phi_kron2d_cells_kernel(
    const double* __restrict__ X, const double* __restrict__ y, const long long* __restrict__ cell_start, int ncell,
    const double* __restrict__ mesh1, double id1, int m1, const double* __restrict__ mesh2, int n2, double id2, int m2,
    double* __restrict__ Ablk, double* __restrict__ rhs, double* __restrict__ yy_out) {
  using KO = KronOut<K>;
  constexpr int CH = kron_chunk<K>();
  constexpr int NPT = (KO::NOUT + 255) / 256;   // output entries per thread
  __shared__ double stage[CH * KO::NREC];
  __shared__ double scratch[16];
  const int tid = threadIdx.x;
  const long Mtot = (long)m1 * m2;
  // decode this thread's output entries once: entry o = tid + 256 j; band entries first, then the (k+1)^2 rhs entries
  int f1[NPT], f2[NPT];      // staged fields to multiply
  long tgt[NPT];             // offset into Ablk (band) or rhs, without the cell's i1 * m2 + i2
  int kind[NPT];             // 0 none, 1 band, 2 rhs
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int o = tid + 256 * j;
    kind[j] = 0; f1[j] = 0; f2[j] = 0; tgt[j] = 0;
    if (o < KO::NBAND) {
      int e = 0, p1 = 0;
      for (int a = 0; a <= K; ++a)
        for (int a2 = a; a2 <= K; ++a2, ++p1)
          for (int b = 0; b <= K; ++b)
            for (int b2 = 0; b2 <= K; ++b2) {
              if (a2 == a && b2 < b) continue;
              if (e == o) {
                kind[j] = 1; f1[j] = p1; f2[j] = KO::NP1 + b * (K + 1) + b2;
                tgt[j] = (long)kron_off(K, a2 - a, b2 - b) * Mtot + (long)(K - a2) * m2 + (K - b2);
              }
              ++e;
            }
    } else if (o < KO::NOUT) {
      const int a = (o - KO::NBAND) / (K + 1), b = (o - KO::NBAND) % (K + 1);
      kind[j] = 2; f1[j] = KO::NP1 + KO::NP2 + a; f2[j] = KO::NP1 + KO::NP2 + K + 1 + b;
      tgt[j] = (long)(K - a) * m2 + (K - b);
    }
  }
  double yy = 0.0;
  for (int c = blockIdx.x; c < ncell; c += gridDim.x) {
    const long long p0 = cell_start[c], p1e = cell_start[c + 1];
    if (p1e <= p0) continue;   // workgroup-uniform
    const int i1 = c / (n2 - 1), i2 = c - i1 * (n2 - 1);
    const double u1 = mesh1[i1], u2 = mesh2[i2];
    double acc[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) acc[j] = 0.0;
    for (long long base = p0; base < p1e; base += CH) {
      const int np = (int)((p1e - base < CH) ? (p1e - base) : CH);
      __syncthreads();
      if (tid < np) {
        const double2 xv = *reinterpret_cast<const double2*>(X + 2 * (base + tid));
        const double yv = y[base + tid];
        double v1[K + 1], v2[K + 1];
        bspline_pieces<K>((xv.x - u1) * id1, v1);
        bspline_pieces<K>((xv.y - u2) * id2, v2);
        double* rec = stage + tid;   // field-major: field f of point t at stage[f * CH + t] (conflict-free writes)
        int f = 0;
#pragma unroll
        for (int a = 0; a <= K; ++a)
#pragma unroll
          for (int a2 = a; a2 <= K; ++a2) rec[(f++) * CH] = v1[a] * v1[a2];
#pragma unroll
        for (int b = 0; b <= K; ++b)
#pragma unroll
          for (int b2 = 0; b2 <= K; ++b2) rec[(f++) * CH] = v2[b] * v2[b2];
#pragma unroll
        for (int a = 0; a <= K; ++a) rec[(f++) * CH] = v1[a];
#pragma unroll
        for (int b = 0; b <= K; ++b) rec[(f++) * CH] = v2[b] * yv;
        yy = fma(yv, yv, yy);
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NPT; ++j)
        if (kind[j]) {
          const double* g1 = stage + f1[j] * CH;
          const double* g2 = stage + f2[j] * CH;
          double t0 = 0.0, t1 = 0.0;
          int t = 0;
          for (; t + 1 < np; t += 2) { t0 = fma(g1[t], g2[t], t0); t1 = fma(g1[t + 1], g2[t + 1], t1); }
          if (t < np) t0 = fma(g1[t], g2[t], t0);
          acc[j] += t0 + t1;
        }
    }
    const long cell_off = (long)i1 * m2 + i2;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      if (kind[j] == 1 && acc[j] != 0.0) __hip_atomic_fetch_add(Ablk + tgt[j] + cell_off, acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (kind[j] == 2 && acc[j] != 0.0) __hip_atomic_fetch_add(rhs + tgt[j] + cell_off, acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  double tot = block_sum(yy, scratch);
  if (tid == 0 && tot != 0.0) __hip_atomic_fetch_add(yy_out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------
// The same statistics on the fp64 MATRIX CORE, without statistic atomics (round 4; the default).  Measured on the per-cell kernel
// above: 177 us for 1M points at 128 x 128 WITHOUT its atomics (rocprofv3; 164 us with them) - it is not the 2.45 M atomics that
// cost, it is one workgroup per cell: 62 points keep 62 of 256 threads busy in the staging phase, two barriers per chunk, two LDS
// reads per FMA in the walk (2.4 GB of LDS traffic), and the cell's loads are not overlapped with anything.
// A cell's contribution is a small Gram matrix: with phi_m = v1[a] v2[b] (m = a (k+1) + b, (k+1)^2 values per point) the entry
// (a, a2, b, b2) is sum_points phi_m phi_n, i.e. G = Phi_c^T Phi_c with Phi_c (points x (k+1)^2) - a rank-4 update per
// v_mfma_f64_16x16x4 (16 basis functions per tile: one tile at k <= 3, 2 x 2 at k = 4, upper-triangular tile pairs only).  ONE WAVE
// owns a cell: lane (ln, lg) evaluates ITS basis function ln of point lg from its two piece polynomials (2k FMAs, coefficients in
// registers) - the same register is the A operand (A[m][k] = phi_m(point k)) and the B operand - and the wave walks the cell four
// points per step; the rhs sums ride along on the VALU.  No barrier inside a cell.  A workgroup of eight waves takes a STRIP of 8
// consecutive cells (one per wave), parks the cells' results in the LDS and writes them entry-major -
// cellsum[entry][cell], 64-byte pieces - into a staging buffer; phi_kron2d_gather_kernel then forms every output (block-band offset,
// column) from the <= (k+1)^2 cells that touch it, in a fixed order: deterministic sums.
// ---------------------------------------------------------------------------------------------------------
constexpr int KRON_STRIP = 8;
typedef double kron_d4 __attribute__((ext_vector_type(4)));
// TIN: storage type of the (cell-sorted) points - double, or float for BASELINE config 4's fp32 data (12 B per point streamed; the
// values are widened exactly in registers, every operation after that is the fp64 one: the statistics equal those of the upcast data).
template <typename TIN> struct KronIn;
template <> struct KronIn<double> {
  static __device__ __forceinline__ double2 xy(const double* X, long long p) { return *reinterpret_cast<const double2*>(X + 2 * p); }
};
template <> struct KronIn<float> {
  static __device__ __forceinline__ double2 xy(const float* X, long long p) { const float2 v = *reinterpret_cast<const float2*>(X + 2 * p); return make_double2((double)v.x, (double)v.y); }
};
template <int K, typename TIN>
__global__ __launch_bounds__(64 * KRON_STRIP) void phi_kron2d_mfma_kernel(
    const TIN* __restrict__ X, const TIN* __restrict__ y, const long long* __restrict__ cell_start, int ncell, int ncell_pad,
    const double* __restrict__ mesh1, double id1, const double* __restrict__ mesh2, int n2, double id2,
    double* __restrict__ cellsum, double* __restrict__ yy_out) {
  using KO = KronOut<K>;
  constexpr int NB = (K + 1) * (K + 1), NT = (NB + 15) / 16, NTT = NT * (NT + 1) / 2;
  __shared__ double res[KO::NOUT * KRON_STRIP];
  __shared__ double2 stage_x[64 * KRON_STRIP];
  __shared__ double stage_y[64 * KRON_STRIP];
  __shared__ double scratch[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, lg = lane >> 4;
  // this lane's basis function per tile: m = 16 t + ln -> (a, b), the two piece polynomials (zeros past the last basis function)
  double ca[NT][K + 1], cb[NT][K + 1];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int m = 16 * t + ln, a = m / (K + 1), b = m % (K + 1);
#pragma unroll
    for (int q = 0; q <= K; ++q) { ca[t][q] = 0.0; cb[t][q] = 0.0; }
#pragma unroll
    for (int i = 0; i <= K; ++i) {
      if (m < NB && i == a) {
#pragma unroll
        for (int q = 0; q <= K; ++q) ca[t][q] = piece_coef<K, 0>(i, q);
      }
      if (m < NB && i == b) {
#pragma unroll
        for (int q = 0; q <= K; ++q) cb[t][q] = piece_coef<K, 0>(i, q);
      }
    }
  }
  // where this lane's accumulator values go: tile pair (tm <= tn), value i: row m = 16 tm + lg + 4 i, column n = 16 tn + ln
  short tgt[NTT][4];
  {
    int qi = 0;
#pragma unroll
    for (int tm = 0; tm < NT; ++tm)
#pragma unroll
      for (int tn = tm; tn < NT; ++tn, ++qi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = 16 * tm + lg + 4 * i, n = 16 * tn + ln;
          int e = -1;
          if (m < NB && n < NB) {
            const int a = m / (K + 1), b = m % (K + 1), a2 = n / (K + 1), b2 = n % (K + 1);
            if (a < a2 || (a == a2 && b <= b2)) {
              e = 0;                                            // entries before (a, a2, b, b2) in the list order of KronOut
              for (int x = 0; x < a; ++x) e += (K + 1) * (K + 2) / 2 + (K - x) * (K + 1) * (K + 1);
              if (a2 > a) e += (K + 1) * (K + 2) / 2 + (a2 - a - 1) * (K + 1) * (K + 1) + b * (K + 1) + b2;
              else { for (int x = 0; x < b; ++x) e += K + 1 - x; e += b2 - b; }
            }
          }
          tgt[qi][i] = (short)e;
        }
  }
  // The walk.  Wave w of the workgroup owns cell c0 + w of every strip the workgroup takes.  Its points come in chunks of 64: ONE coalesced
  // load per lane (its own point: 16 B + 8 B), parked in a wave-private 1.5 KB LDS buffer, read back four points per step as broadcasts;
  // the loads of the NEXT chunk (of this cell, or the first of the wave's next cell) are issued before the 16 steps of this one.
  double2* xs = stage_x + wv * 64;
  double* ys = stage_y + wv * 64;
  double yy = 0.0;
  int strip = blockIdx.x;
  auto cell_range = [&](int st, long long& a0, long long& a1) __attribute__((always_inline)) {
    const int c = st * KRON_STRIP + wv;
    const bool in = st * KRON_STRIP < ncell_pad && c < ncell;
    a0 = in ? cell_start[c] : 0;
    a1 = in ? cell_start[c + 1] : 0;
  };
  long long p0, p1e;
  cell_range(strip, p0, p1e);
  double2 xn = make_double2(0.0, 0.0);
  double yn = 0.0;
  if (p0 + lane < p1e) { xn = KronIn<TIN>::xy(X, p0 + lane); yn = (double)y[p0 + lane]; }
  for (; strip * KRON_STRIP < ncell_pad; strip += gridDim.x) {
    const int c0 = strip * KRON_STRIP, c = c0 + wv;
    kron_d4 acc[NTT];
    double r[NT];
#pragma unroll
    for (int q = 0; q < NTT; ++q) acc[q] = kron_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NT; ++t) r[t] = 0.0;
    long long q0, q1e;                                            // the wave's next cell
    cell_range(strip + gridDim.x, q0, q1e);
    if (p1e > p0) {
      const int i1 = c / (n2 - 1), i2 = c - i1 * (n2 - 1);
      const double u1 = mesh1[i1], u2 = mesh2[i2];
      for (long long base = p0; base < p1e; base += 64) {
        xs[lane] = xn; ys[lane] = yn;                             // (lanes past the cell's end park zeros: masked below)
        {
          const long long nb = base + 64 < p1e ? base + 64 + lane : q0 + lane;      // next chunk of this cell, else the next cell's first
          const long long ne = base + 64 < p1e ? p1e : q1e;
          if (nb < ne) { xn = KronIn<TIN>::xy(X, nb); yn = (double)y[nb]; } else { xn = make_double2(0.0, 0.0); yn = 0.0; }
        }
        const int np = (int)(p1e - base < 64 ? p1e - base : 64);
        auto step = [&](int st4, bool masked) __attribute__((always_inline)) {
          const int pi = 4 * st4 + lg;
          const bool ok = !masked || pi < np;
          const double2 xv = xs[pi];
          const double yv = ok ? ys[pi] : 0.0;
          const double t1 = (xv.x - u1) * id1, t2 = (xv.y - u2) * id2;
          double phi[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            double va = ca[t][K], vb = cb[t][K];
#pragma unroll
            for (int q = K - 1; q >= 0; --q) { va = fma(va, t1, ca[t][q]); vb = fma(vb, t2, cb[t][q]); }
            phi[t] = ok ? va * vb : 0.0;
            r[t] = fma(phi[t], yv, r[t]);
          }
          yy = fma(yv, yv, yy);                                   // (every one of the 16 lanes of a point: divided out at the end)
          int qi = 0;
#pragma unroll
          for (int tm = 0; tm < NT; ++tm)
#pragma unroll
            for (int tn = tm; tn < NT; ++tn, ++qi) acc[qi] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[tm], phi[tn], acc[qi], 0, 0, 0);
        };
        const int nfull = np >> 2;                                  // steps whose four points all exist: no masks
        for (int st4 = 0; st4 < nfull; ++st4) step(st4, false);
        if (np & 3) step(nfull, true);
      }
    } else if (q1e > q0) {
      // (an empty cell: the prefetch of the next cell was never issued - nothing was walked)
      if (q0 + lane < q1e) { xn = KronIn<TIN>::xy(X, q0 + lane); yn = (double)y[q0 + lane]; } else { xn = make_double2(0.0, 0.0); yn = 0.0; }
    }
    p0 = q0; p1e = q1e;
#pragma unroll
    for (int q = 0; q < NTT; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (tgt[q][i] >= 0) res[(int)tgt[q][i] * KRON_STRIP + wv] = acc[q][i];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double v = r[t];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lg == 0 && 16 * t + ln < NB) res[(KO::NBAND + 16 * t + ln) * KRON_STRIP + wv] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < KO::NOUT * KRON_STRIP; idx += 64 * KRON_STRIP) {
      const int o = idx / KRON_STRIP, cc = idx % KRON_STRIP;
      cellsum[(size_t)o * ncell_pad + c0 + cc] = res[idx];
    }
    __syncthreads();
  }
  double tot = block_sum(yy, scratch) * 0.0625;
  if (tid == 0 && tot != 0.0) __hip_atomic_fetch_add(yy_out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// gather: one thread per output (plane, column); plane < noff: block-band offset (d1, d2); plane == noff: the rhs
template <int K>
__global__ __launch_bounds__(256) void phi_kron2d_gather_kernel(const double* __restrict__ cellsum, int ncell_pad, int nc1, int nc2, int m1, int m2,
                                                                double* __restrict__ Ablk, double* __restrict__ rhs) {
  using KO = KronOut<K>;
  __shared__ short etab[K + 1][K + 1][K + 1][K + 1];      // (a, a2 >= a, b, b2) -> entry index of the cell's list
  if (threadIdx.x == 0) {
    int e = 0;
    for (int a = 0; a <= K; ++a)
      for (int a2 = a; a2 <= K; ++a2)
        for (int b = 0; b <= K; ++b)
          for (int b2 = 0; b2 <= K; ++b2) {
            if (a2 == a && b2 < b) { etab[a][a2][b][b2] = -1; continue; }
            etab[a][a2][b][b2] = (short)(e++);
          }
  }
  __syncthreads();
  const long Mtot = (long)m1 * m2;
  const int noff = kron_noff(K);
  const long total = (long)(noff + 1) * Mtot;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int o = (int)(t / Mtot);
    const long col = t - (long)o * Mtot;
    const int j1 = (int)(col / m2), j2 = (int)(col - (long)j1 * m2);
    double sum = 0.0;
    if (o < noff) {
      int d1, d2;
      if (o <= K) { d1 = 0; d2 = o; } else { const int q = o - (K + 1); d1 = 1 + q / (2 * K + 1); d2 = q % (2 * K + 1) - K; }
      for (int a2 = d1; a2 <= K; ++a2) {
        const int a = a2 - d1, i1 = j1 - K + a2;
        if (i1 < 0 || i1 >= nc1) continue;
        for (int b2 = (d2 > 0 ? d2 : 0); b2 <= K && b2 - d2 <= K; ++b2) {
          const int b = b2 - d2, i2 = j2 - K + b2;
          if (i2 < 0 || i2 >= nc2) continue;
          sum += cellsum[(size_t)etab[a][a2][b][b2] * ncell_pad + (size_t)i1 * nc2 + i2];
        }
      }
      Ablk[t] = sum;
    } else {
      for (int a = 0; a <= K; ++a) {
        const int i1 = j1 - K + a;
        if (i1 < 0 || i1 >= nc1) continue;
        for (int b = 0; b <= K; ++b) {
          const int i2 = j2 - K + b;
          if (i2 < 0 || i2 >= nc2) continue;
          sum += cellsum[(size_t)(KO::NBAND + a * (K + 1) + b) * ncell_pad + (size_t)i1 * nc2 + i2];
        }
      }
      rhs[col] = sum;
    }
  }
}

// 2-D cell id per point: i1 * (n2 - 1) + i2  (basis.py:58-59 index rule per dimension)
__global__ void kron_cell_index_kernel(const double* __restrict__ X, long N, const double* __restrict__ mesh1, int n1,
                                       double id1, const double* __restrict__ mesh2, int n2, double id2,
                                       int* __restrict__ cell) {
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const double2 xv = *reinterpret_cast<const double2*>(X + 2 * n);
  const int i1 = neighbour_index(xv.x, mesh1, n1, mesh1[0], id1);
  const int i2 = neighbour_index(xv.y, mesh2, n2, mesh2[0], id2);
  cell[n] = i1 * (n2 - 1) + i2;
}

// Khatri-Rao COO triplets (kronecker.make_kvs_sparse): for point n, entry e = a*(K+1)+b: row, value
template <int K>
__global__ void kron_evaluate_kernel(const double* __restrict__ X, long N, const double* __restrict__ mesh1, int n1,
                                     double id1, const double* __restrict__ mesh2, int n2, double id2, int m2,
                                     long long* __restrict__ rows, double* __restrict__ data) {
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const double2 xv = *reinterpret_cast<const double2*>(X + 2 * n);
  const int i1 = neighbour_index(xv.x, mesh1, n1, mesh1[0], id1);
  const int i2 = neighbour_index(xv.y, mesh2, n2, mesh2[0], id2);
  double v1[K + 1], v2[K + 1];
  bspline_pieces<K>((xv.x - mesh1[i1]) * id1, v1);
  bspline_pieces<K>((xv.y - mesh2[i2]) * id2, v2);
#pragma unroll
  for (int a = 0; a <= K; ++a)
#pragma unroll
    for (int b = 0; b <= K; ++b) {
      const long e = (long)(a * (K + 1) + b) * N + n;
      rows[e] = (long long)(i1 + K - a) * m2 + (i2 + K - b);
      data[e] = v1[a] * v2[b];
    }
}

// symmetric read of a 1-D lower band S (k+1, m): Sigma[i + d, i], d may be negative
__device__ __forceinline__ double band_sym(const double* S, int m, int i, int d) {
  return d >= 0 ? S[(long)d * m + i] : S[(long)(-d) * m + i + d];
}

// P (column-major band, LD = bw+1, zero-filled beforehand) = K1 (x) K2 + A / s ;  also tr( (S1 (x) S2) A )
__global__ void kron_assemble_kernel(const double* __restrict__ K1, const double* __restrict__ K2,
                                     const double* __restrict__ S1, const double* __restrict__ S2,
                                     const double* __restrict__ Ablk, int k, int m1, int m2, double s, long LD,
                                     double* __restrict__ Pb, double* __restrict__ trace_out) {
  __shared__ double scratch[16];
  const long Mtot = (long)m1 * m2;
  const int noff = kron_noff(k);
  double tr = 0.0;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < Mtot * noff; t += (long)gridDim.x * blockDim.x) {
    const int o = (int)(t / Mtot);
    const long c = t - (long)o * Mtot;
    int d1, d2;
    if (o <= k) { d1 = 0; d2 = o; } else { int q = o - (k + 1); d1 = 1 + q / (2 * k + 1); d2 = q % (2 * k + 1) - k; }
    const int i1 = (int)(c / m2), i2 = (int)(c - (long)i1 * m2);
    if (i1 + d1 >= m1 || i2 + d2 < 0 || i2 + d2 >= m2) continue;
    const double a = Ablk[t];
    const double kv = K1[(long)d1 * m1 + i1] * band_sym(K2, m2, i2, d2);
    if (Pb) Pb[c * LD + (long)d1 * m2 + d2] = kv + a / s;
    if (S1) {
      const double sv = S1[(long)d1 * m1 + i1] * band_sym(S2, m2, i2, d2);
      tr = fma((o == 0) ? 1.0 : 2.0, sv * a, tr);
    }
  }
  if (trace_out) {
    double tot = block_sum(tr, scratch);
    if (threadIdx.x == 0 && tot != 0.0) __hip_atomic_fetch_add(trace_out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The same P written as the TWO systems of the two-sided ("twisted") factorisation.  The chain of the band Cholesky is strictly
// sequential in the block columns (512 of them at 128 x 128), so P is split at a separator [h, h + Bb) at least a bandwidth wide:
//   top system    Pt: `padt` identity columns, then original columns 0 .. h + Bb - 1               (nb Bb columns)
//   bottom system Pr: `padb` identity columns, then original columns M - 1 down to h (REVERSED)     (nb Bb columns)
// Each is an ordinary lower band with the separator as its last Bb columns; the two factorisations are independent (they run
// concurrently, half the chain each) and what couples them is the separator's Schur complement
//   S = L_ss L_ss^T + J L'_ss L'_ss^T J - P_ss        (L_ss, L'_ss: the last diagonal super-blocks of the two factors, J: reversal),
// a dense Bb x Bb block factored last.  Entry (r, c), r >= c, of the band goes to Pt[(c + padt) LD + r - c] when r < h + Bb and to
// Pr[(off - r) LD + r - c] (off = padb + M - 1; reversal swaps the triangle) when c >= h; separator entries go to both.
__global__ void kron_assemble_twisted_kernel(const double* __restrict__ K1, const double* __restrict__ K2,
                                             const double* __restrict__ S1, const double* __restrict__ S2,
                                             const double* __restrict__ Ablk, int k, int m1, int m2, double s, long LD,
                                             long h, long top_end, long padt, long padb,
                                             double* __restrict__ Pt, double* __restrict__ Pr, double* __restrict__ trace_out) {
  __shared__ double scratch[16];
  const long Mtot = (long)m1 * m2;
  const int noff = kron_noff(k);
  const long off = padb + Mtot - 1;
  double tr = 0.0;
  const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x, gstride = (long)gridDim.x * blockDim.x;
  for (long u = gtid; u < padt; u += gstride) Pt[u * LD] = 1.0;
  for (long u = gtid; u < padb; u += gstride) Pr[u * LD] = 1.0;
  for (long t = gtid; t < Mtot * noff; t += gstride) {
    const int o = (int)(t / Mtot);
    const long c = t - (long)o * Mtot;
    int d1, d2;
    if (o <= k) { d1 = 0; d2 = o; } else { int q = o - (k + 1); d1 = 1 + q / (2 * k + 1); d2 = q % (2 * k + 1) - k; }
    const int i1 = (int)(c / m2), i2 = (int)(c - (long)i1 * m2);
    if (i1 + d1 >= m1 || i2 + d2 < 0 || i2 + d2 >= m2) continue;
    const double a = Ablk[t];
    const double kv = K1[(long)d1 * m1 + i1] * band_sym(K2, m2, i2, d2);
    const long d = (long)d1 * m2 + d2, r = c + d;
    const double pv = kv + a / s;
    if (r < top_end) Pt[(c + padt) * LD + d] = pv;
    if (c >= h) Pr[(off - r) * LD + d] = pv;
    if (S1) {
      const double sv = S1[(long)d1 * m1 + i1] * band_sym(S2, m2, i2, d2);
      tr = fma((o == 0) ? 1.0 : 2.0, sv * a, tr);
    }
  }
  if (trace_out) {
    double tot = block_sum(tr, scratch);
    if (threadIdx.x == 0 && tot != 0.0) __hip_atomic_fetch_add(trace_out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Blocked band Cholesky, column-major lower band Pb[c*LD + (r-c)], bandwidth bw, block NB.
// panel kernel (one workgroup): factor the NB x NB diagonal block in LDS, solve the rows below (one thread per row),
// carry one right-hand side along (forward substitution fused, like the 1-D sweep).
// ---------------------------------------------------------------------------------------------------------
constexpr int BB_NB = 32;

__global__ __launch_bounds__(512) void bb_panel_kernel(double* __restrict__ Pb, long M, int bw, long LD, long j0,
                                                        double* __restrict__ rhs, int* __restrict__ info) {
  __shared__ double Ls[BB_NB][BB_NB + 1];
  __shared__ double cs[BB_NB], invd[BB_NB];
  const int tid = threadIdx.x;
  const int nbk = (int)((M - j0 < BB_NB) ? (M - j0) : BB_NB);
  for (int idx = tid; idx < BB_NB * BB_NB; idx += blockDim.x) {
    int r = idx / BB_NB, c = idx % BB_NB;
    double v = (r == c) ? 1.0 : 0.0;
    if (r >= c && r < nbk && c < nbk) v = (r - c <= bw) ? Pb[(j0 + c) * LD + (r - c)] : 0.0;
    Ls[r][c] = v;
  }
  if (tid < BB_NB) cs[tid] = (rhs && tid < nbk) ? rhs[j0 + tid] : 0.0;
  __syncthreads();
  // Diagonal block: one wavefront, rows in registers (lane r holds row r; fully unrolled, so every register index and every
  // v_readlane lane is static) - no workgroup barriers inside the 32 column steps; the rhs block rides along (y = L^-1 b).
  if (tid < 64) {
    const int r = tid & 31;
    double a[BB_NB];
#pragma unroll
    for (int c = 0; c < BB_NB; ++c) a[c] = Ls[r][c];
    double t = cs[r];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < BB_NB; ++j) {
      const double piv = readlane_f64(a[j], j);
      if (!(piv > 0.0) && !bad) bad = j + 1;
      const double ljj = sqrt(piv), inv = 1.0 / ljj;
      a[j] = (r == j) ? ljj : a[j] * inv;              // (rows above j hold zeros in column j)
      const double yj = readlane_f64(t, j) * inv;       // forward substitution rides along
      t = (r == j) ? yj : fma(-a[j], yj, t);
#pragma unroll
      for (int c = j + 1; c < BB_NB; ++c) {
        const double lcj = readlane_f64(a[j], c);
        a[c] = (r >= c) ? fma(-a[j], lcj, a[c]) : a[c];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid < BB_NB) {
#pragma unroll
      for (int c = 0; c < BB_NB; ++c) Ls[r][c] = (c < r) ? a[c] : 0.0;
#pragma unroll
      for (int c = 0; c < BB_NB; ++c)
        if (c == r) { Ls[r][r] = a[c]; invd[r] = 1.0 / a[c]; }
      cs[r] = (r < nbk) ? t : 0.0;
    }
    if (tid == 0 && bad) atomicCAS(info, 0, (int)(j0 + bad));
  }
  __syncthreads();
  for (int idx = tid; idx < BB_NB * BB_NB; idx += blockDim.x) {
    int r = idx / BB_NB, c = idx % BB_NB;
    if (r >= c && r < nbk && c < nbk && r - c <= bw) Pb[(j0 + c) * LD + (r - c)] = Ls[r][c];
  }
  if (rhs && tid < nbk) rhs[j0 + tid] = cs[tid];
  const long r_lo = j0 + nbk;
  long r_hi = j0 + nbk - 1 + bw;  // last row touching this block column
  if (r_hi > M - 1) r_hi = M - 1;
  for (long r = r_lo + tid; r <= r_hi; r += blockDim.x) {
    double xr[BB_NB];
    double t = rhs ? rhs[r] : 0.0;
#pragma unroll
    for (int c = 0; c < BB_NB; ++c) {
      const long d = r - (j0 + c);
      double v = (c < nbk && d <= bw) ? Pb[(j0 + c) * LD + d] : 0.0;
#pragma unroll
      for (int p = 0; p < c; ++p) v = fma(-xr[p], Ls[c][p], v);
      xr[c] = v * invd[c];
      if (c < nbk && d <= bw) Pb[(j0 + c) * LD + d] = xr[c];
      t = fma(-xr[c], cs[c], t);
      __builtin_amdgcn_sched_barrier(0);   // keep the 528 broadcast reads of L from being hoisted into ~500 live registers
    }
    if (rhs) rhs[r] = t;
  }
}

// trailing update: P[r][r'] -= sum_c L[r][j0+c] L[r'][j0+c] for r >= r' in the rows below the panel, inside the band.
__global__ __launch_bounds__(256) void bb_update_kernel(double* __restrict__ Pb, long M, int bw, long LD, long j0) {
  __shared__ double Lr[32][BB_NB + 1], Lc[32][BB_NB + 1];
  const int nbk = (int)((M - j0 < BB_NB) ? (M - j0) : BB_NB);
  const long r_lo = j0 + nbk;
  // tile pair from the linear block index: ti >= tj
  int ti = 0, rem = blockIdx.x;
  while (rem > ti) { rem -= ti + 1; ++ti; }
  const int tj = rem;
  const long rbase = r_lo + (long)ti * 32, cbase = r_lo + (long)tj * 32;
  for (int idx = threadIdx.x; idx < 32 * BB_NB; idx += blockDim.x) {
    int rr = idx % 32, c = idx / 32;
    long r = rbase + rr, r2 = cbase + rr;
    long d = r - (j0 + c), d2 = r2 - (j0 + c);
    Lr[rr][c] = (c < nbk && r < M && d <= bw) ? Pb[(j0 + c) * LD + d] : 0.0;
    Lc[rr][c] = (c < nbk && r2 < M && d2 <= bw) ? Pb[(j0 + c) * LD + d2] : 0.0;
  }
  __syncthreads();
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int rr = tx + 16 * u, cc = ty + 16 * w;  // rr: row in tile ti (contiguous in memory), cc: row in tile tj
      const long r = rbase + rr, c = cbase + cc;
      if (r < M && c < M && r >= c && r - c <= bw) {
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < BB_NB; ++p) acc = fma(Lr[rr][p], Lc[cc][p], acc);
        Pb[c * LD + (r - c)] -= acc;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent band Cholesky (ONE launch; the launch-per-panel form above is the fallback for bw > 416).
// Left-looking over block columns of NB = 32: workgroup g owns block columns g, g + G, ...  For column c it keeps the whole panel
// (rows j0 .. j0+31+bw, 32 columns) in MFMA accumulators - 27 row tiles of 16 at bw = 387, 7 per wave, 112 VGPRs - applies the
// updates of the <= 13 finished block columns p that reach it (v_mfma_f64_16x16x4: C -= L[rows, p] L[c rows, p]^T, operands loaded
// straight from the band in the instruction's lane map, band mask applied on the load), then factors the 32 x 32 diagonal block with
// one wavefront (rows in registers, v_readlane broadcasts), inverts it (one column per lane), forms the rows below as
// Pan * Linv^T on the matrix cores again, writes the block column and publishes `done = c + 1` (release).  A consumer acquires
// `done >= p + 1` before it reads block column p.  Columns finish in increasing order (c needs c-1), so monotonic counters are
// the whole protocol (a second one, `early`, lets the next diagonal block start before the column is completely stored - see
// the kernel), and the lowest unfinished column never waits on anything unfinished: progress needs only that its owner
// is resident (G <= 16 workgroups).  The right-hand side rides along deterministically: t = b - sum_p L[c rows, p] y_p from the
// same B operands, y = Ldiag^-1 t by the factoring wavefront.
// ---------------------------------------------------------------------------------------------------------
typedef double bb_d4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long bbp_stamps[32];   // diagnostic: s_memtime stamps of block column ASVGP_BB_STAMP_COL (tools/bbp_stamps.py)
#define BBP_STAMP(k) do { if (stamp_col == c && tid == 0) bbp_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
constexpr int BBP_THREADS = 512;
// Eight waves: wave 0 owns the diagonal block and keeps its SIMD to itself - wave 4, which the hardware places on the same SIMD, only
// loads the panel and takes the barriers (a second busy wave there stretches the single-wave elimination from 17 K to 23 K cycles) -
// and the six others (1, 2, 3, 5, 6, 7) are the workers: worker index widx = 0..5, row tile rt belongs to worker (rt - 2) % BBP_NWK.
constexpr int BBP_NWK = 6;
__host__ __device__ inline int bbp_row_tiles(int bw) { return (BB_NB + bw + 15) / 16; }               // 16-row tiles of a panel
__host__ __device__ inline int bbp_rs(int bw) { return bbp_row_tiles(bw) * 16 + 1; }                  // LDS column stride (odd: bank spread)
__host__ inline size_t bbp_lds_bytes(int bw) { return sizeof(double) * ((size_t)BB_NB * bbp_rs(bw) + 2 * BB_NB * (BB_NB + 1) + 2 * BB_NB + 16 * 17 + 8); }

// 1/sqrt(x) and sqrt(x) in fp64 from v_rsq_f64 + two Newton steps (+ one correction of the root): ~16 VALU instructions instead of
// the ~80 of sqrt() followed by a division; relative error of both results <= 2e-16 (not correctly rounded: the factor is compared
// against a CPU band Cholesky at 1e-9 - 1e-12, never bit for bit).  Non-positive / NaN input gives NaN (reported through `info`).
__device__ __forceinline__ void bb_rsqrt(double x, double& inv, double& root) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  double g = x * y;
  g = fma(fma(-g, g, x), 0.5 * y, g);
  inv = y;
  root = g;
}

__device__ __forceinline__ double bb_rsqrt1(double x) {           // 1/sqrt(x) alone (two Newton steps on v_rsq_f64)
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  return y;
}

// one update C -= L[rows, p] L[c rows, p]^T on the NT row tiles rt = R0 + ST * t of the calling wave
template <int NT, int ST, bool FAST, bool RHS_ROW>
__device__ __forceinline__ void bbp_update(bb_d4 (&acc)[NT][2], double& tacc0, double& tacc1, const double* __restrict__ Pb,
                                           const double* __restrict__ rhs, long M, int bw, long LD, long j0, long p0, int nbk, int R16,
                                           int R0, int ln, int lg) {
  // element (r, p0 + q) of the band sits at Pp[q * (LD - 1) + r]
  const double* __restrict__ Pp = Pb + p0 * (LD - 1);
  const int dj = (int)(j0 - p0);                      // 32 .. 32 * PMAX
  // Every load of the B operand (and of y) is ISSUED before the first value is used: written as one loop, hipcc waited for each pair of
  // loads before issuing the next (s_waitcnt vmcnt(0) eight times in a row) - eight round trips to memory another CU has just written,
  // 19 K cycles of the diagonal chain's update by block column c-1 (tools/bbp_stamps.py).
  double breg[2][8], yreg[8];
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {
    const int q = 4 * s8 + lg;
    const double* __restrict__ bq = Pp + (long)q * (LD - 1) + j0 + ln;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int n = 16 * ct + ln;
      const bool ok = n < nbk && dj + n - q <= bw;
      breg[ct][s8] = FAST ? bq[16 * ct] : ((ok) ? bq[16 * ct] : 0.0);
    }
    yreg[s8] = (RHS_ROW && rhs) ? rhs[p0 + 4 * s8 + lg] : 0.0;
  }
  // ... and so is the first group of A operands (the only group of the diagonal wave: one round trip for the whole update)
  constexpr int GT = (NT < 4) ? NT : 3;                 // (groups of three row tiles: four need 16 more registers than 256 leave)
  constexpr bool HOIST = NT <= 2;                       // (the worker waves' nine tiles: 24 more live registers spill)
  double araw[HOIST ? GT : 1][8];
  if (HOIST) {
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int q = 4 * s8 + lg;
      const double* __restrict__ aq = Pp + (long)q * (LD - 1) + j0 + 16 * R0 + ln;
#pragma unroll
      for (int u = 0; u < (HOIST ? GT : 1); ++u) {
        const int rl = 16 * (R0 + ST * u) + ln;
        const bool ok = (FAST || j0 + rl < M) && dj + rl - q <= bw;
        araw[u][s8] = FAST ? aq[16 * ST * u] : (ok ? aq[16 * ST * u] : 0.0);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {
    const int q = 4 * s8 + lg;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int n = 16 * ct + ln;
      const bool ok = n < nbk && dj + n - q <= bw;
      breg[ct][s8] = ok ? breg[ct][s8] : 0.0;
    }
  }
  if (RHS_ROW && rhs) {
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      tacc0 = fma(breg[0][s8], yreg[s8], tacc0);
      tacc1 = fma(breg[1][s8], yreg[s8], tacc1);
    }
  }
  // A operands in groups of up to 4 row tiles: the group's 32 loads are in flight before its first MFMA waits (with a branch per
  // tile the loads of tile t+1 were issued after the MFMAs of tile t; all tiles at once spill)
  const long r_last = p0 + BB_NB - 1 + bw;            // last row of block column p
#pragma unroll
  for (int g0 = 0; g0 < NT; g0 += GT) {
    double areg[GT][8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int q = 4 * s8 + lg;
      const double* __restrict__ aq = Pp + (long)q * (LD - 1) + j0 + 16 * R0 + ln;
#pragma unroll
      for (int u = 0; u < GT; ++u) {
        const int t = g0 + u;
        if (t < NT) {
          const int rl = 16 * (R0 + ST * t) + ln;     // row inside the panel window
          const bool ok = (FAST || j0 + rl < M) && dj + rl - q <= bw;
          const double v = (HOIST && g0 == 0) ? araw[HOIST ? u : 0][s8] : (FAST ? aq[16 * ST * t] : (ok ? aq[16 * ST * t] : 0.0));
          areg[u][s8] = ok ? -v : 0.0;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < GT; ++u) {
      const int t = g0 + u;
      if (t < NT) {
        const int rt = R0 + ST * t;
        const long rbase = j0 + 16 * rt;
        if (rt < R16 && rbase <= r_last && rbase < M) {    // wave-uniform: tiles past the end of block column p get no MFMAs
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) {
            acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[u][s8], breg[0][s8], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[u][s8], breg[1][s8], acc[t][1], 0, 0, 0);
          }
        }
      }
    }
  }
}

// wait (this wavefront only) until counter[0] >= target; `seen` is the wave's last observed value.  The spin is bounded: a
// wavefront that has polled for seconds (the owner of the awaited column never became resident, or died) raises the abort word,
// which every other waiter sees, and the kernel drains with `info` = -1 instead of hanging the device.
__device__ __forceinline__ bool bbp_wait(unsigned& seen, unsigned target, const unsigned* __restrict__ counter, unsigned* __restrict__ abort_word,
                                         int lane) {
  if (seen >= target) return true;
  unsigned v = 0;
  int ok = 1;
  if (lane == 0) {
    long spins = 0;
    while ((v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {   // (ONE acquire fence, behind the loop)
      __builtin_amdgcn_s_sleep(8);
      if ((++spins & 1023) == 0 && (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > (1L << 25))) {
        __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
  }
  ok = __builtin_amdgcn_readfirstlane(ok);
  seen = __builtin_amdgcn_readfirstlane(v);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok != 0;
}

// The rows of one wave's tiles: panel window (LDS) <-> accumulators in the MFMA C layout (col = lane & 15, row = (lane >> 4) + 4 i)
template <int NT, int ST>
__device__ __forceinline__ void bbp_acc_load(bb_d4 (&acc)[NT][2], const double* Pan, int RS, int R16, int R0, int ln, int lg) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int rt = (R0 + ST * t < R16) ? R0 + ST * t : 0;      // (tiles past the window: any valid address, never stored back)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][ct][i] = Pan[(size_t)(16 * ct + ln) * RS + 16 * rt + lg + 4 * i];
  }
}
template <int NT, int ST>
__device__ __forceinline__ void bbp_acc_store(const bb_d4 (&acc)[NT][2], double* Pan, int RS, int R16, int R0, int ln, int lg) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int rt = R0 + ST * t;
    if (rt < R16) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) Pan[(size_t)(16 * ct + ln) * RS + 16 * rt + lg + 4 * i] = acc[t][ct][i];
    }
  }
}

// Work split inside the workgroup (eight waves, see BBP_NWK): wave 0 owns the two row tiles of the DIAGONAL block, the six workers
// (waves 1, 2, 3, 5, 6, 7) the tiles below it (worker widx: rt = 2 + widx, 2 + widx + 6, ...: NTW each), wave 4 - wave 0's SIMD neighbour -
// only loads the panel.  After the update by block column c-1 - the one a column has to wait for - wave 0 goes straight to the
// factorisation of the diagonal block while the others are still on the matrix cores.  (Four waves, three workers of nine tiles: the
// diagonal wave waited 11 K cycles per column at the barrier before the solve; eight waves with wave 4 working: the single-wave
// elimination stretched from 17 K to 23 K cycles.)
template <int NTW>
__global__ __launch_bounds__(BBP_THREADS) void bb_chol_persistent_kernel(double* __restrict__ Pb, long M, int bw, long LD,
                                                                         double* __restrict__ rhs, int* __restrict__ info,
                                                                         unsigned* __restrict__ done, int nbc, int stamp_col) {
  extern __shared__ double lds[];
  const int R16 = bbp_row_tiles(bw), RS = bbp_rs(bw);
  double* Pan = lds;                                  // [32 cols][RS rows]
  double* Ld = Pan + (size_t)BB_NB * RS;              // [32][33] diagonal factor
  double* Li = Ld + BB_NB * (BB_NB + 1);              // [32][33] its inverse
  double* ys = Li + BB_NB * (BB_NB + 1);              // [32] t = b - sum L y of this block / then y
  double* Tt = ys + 2 * BB_NB;                        // [16][17] product staging of the block inverse
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ln = lane & 15, lg = lane >> 4;
  const int PMAX = (BB_NB - 1 + bw) / BB_NB;          // block columns to the left that reach a panel
  const int widx = wv < 4 ? wv - 1 : wv - 2;          // worker index of waves 1, 2, 3, 5, 6, 7 (wave 0: the diagonal block; wave 4: idle)
  const bool worker = wv != 0 && wv != 4;
  const int R0 = 2 + widx;                            // first row tile of a worker (stride BBP_NWK)
  // done[0]: block columns completely finished.  done[16]: "early" arrivals, two per block column - the first row tile of wave 1
  // (rt = 2) and of wave 2 (rt = 3), i.e. rows j0+32 .. j0+63: all that the DIAGONAL block of the next column needs of this one.
  // Wave 0 of column c therefore starts its last update, the factorisation and the inverse as soon as column c-1 has published
  // those two tiles, while column c-1 is still solving and storing its other 23 tiles (which waves 1..3 of column c wait for).
  unsigned* early = done + 16;
  unsigned seen = 0, seen_early = 0;                  // per wavefront
  for (int c = blockIdx.x; c < nbc; c += gridDim.x) {
    const long j0 = (long)c * BB_NB;
    const int nbk = (int)((M - j0 < BB_NB) ? (M - j0) : BB_NB);
    const bool interior = j0 + 16 * R16 <= M;         // every row of the window exists: unmasked operand loads (value-masked)
    BBP_STAMP(0);
    // ---- panel of P: band -> LDS window (row-contiguous = coalesced; zeros outside the band, identity on the padding columns of a
    // short last block) -> accumulators in the MFMA C layout.  (Masked loads straight into the C layout needed 112 live exec masks
    // per lane: 700 spilled SGPRs and 200 spilled VGPRs.)
    const int RW = 16 * R16;
    for (int col = wv; col < BB_NB; col += BBP_THREADS / 64) {       // one column per wave and pass, lanes along the rows
      const double* __restrict__ src = Pb + (j0 + col) * LD - col;      // element (j0 + rl, j0 + col) at src[rl]
#pragma unroll 4
      for (int rl = lane; rl < RW; rl += 64) {
        const int d = rl - col;
        double v = (col >= nbk && rl == col) ? 1.0 : 0.0;
        if (col < nbk && j0 + rl < M && d >= 0 && d <= bw) v = src[rl];
        Pan[(size_t)col * RS + rl] = v;
      }
    }
    __syncthreads();
    bb_d4 acc[NTW][2];                                // waves 1..3: NTW tiles; wave 0: the diagonal block in the first two
    bb_d4 (&accd)[2][2] = reinterpret_cast<bb_d4 (&)[2][2]>(acc);
    {
      int lno = ln, lgo = lg;
      asm volatile("" : "+v"(lno), "+v"(lgo));
      if (wv == 0) bbp_acc_load<2, 1>(accd, Pan, RS, R16, 0, lno, lgo);
      else if (worker) bbp_acc_load<NTW, BBP_NWK>(acc, Pan, RS, R16, R0, lno, lgo);
    }
    __syncthreads();                                  // (the window is rewritten after the updates)
    double tacc0 = 0.0, tacc1 = 0.0;
    bool gave_up = false;
    BBP_STAMP(1);
    // ---- updates from the finished block columns that reach this one
    for (int p = (c > PMAX) ? c - PMAX : 0; p < c && wv != 4; ++p) {
      const bool arrived = (wv == 0 && p == c - 1) ? bbp_wait(seen_early, 2u * (unsigned)c, early, done + 24, lane)
                                                   : bbp_wait(seen, (unsigned)(p + 1), done, done + 24, lane);
      if (!arrived) gave_up = true;                   // (wave-uniform; the column is finished with whatever is there and flagged)
      if (p == c - 1) BBP_STAMP(2);
      const long p0 = (long)p * BB_NB;
      if (wv == 0) {
        if (interior) bbp_update<2, 1, true, true>(accd, tacc0, tacc1, Pb, rhs, M, bw, LD, j0, p0, nbk, R16, 0, ln, lg);
        else bbp_update<2, 1, false, true>(accd, tacc0, tacc1, Pb, rhs, M, bw, LD, j0, p0, nbk, R16, 0, ln, lg);
      } else {
        if (interior) bbp_update<NTW, BBP_NWK, true, false>(acc, tacc0, tacc1, Pb, rhs, M, bw, LD, j0, p0, nbk, R16, R0, ln, lg);
        else bbp_update<NTW, BBP_NWK, false, false>(acc, tacc0, tacc1, Pb, rhs, M, bw, LD, j0, p0, nbk, R16, R0, ln, lg);
      }
    }
    BBP_STAMP(3);
    if (gave_up && lane == 0) atomicExch(info, -1);   // a waited-for block column never arrived (see bbp_wait): results are invalid
    if (wv == 0) {
      // ---- diagonal block (one wavefront; single-wave code issues one VALU instruction per >= 4 cycles, so the instruction count
      // is the cost - the plain 32-step form with sqrt and division was 25 K cycles, its column-per-lane inverse another 16 K):
      //   1. columns 0..15 eliminated on all 32 rows (lane r = row r, v_readlane broadcasts, reciprocal square root by Newton
      //      steps on v_rsq_f64): L11 and L21 at once, the rhs block rides along;
      //   2. D22 -= L21 L21^T: one 16x16x16 product on the matrix cores;   3. columns 16..31 on rows 16..31;
      //   4. L11^-1 and L22^-1 together (lanes 0..15 / 16..31, one column each, right-looking: independent FMAs);
      //   5. the off-diagonal block of the inverse, -L22^-1 (L21 L11^-1): two more small MFMA products through the LDS.
      bbp_acc_store<2, 1>(accd, Pan, RS, R16, 0, ln, lg);
      {
        double t0 = tacc0, t1 = tacc1;                // t of this block: lane r (< 32) ends up with its row's sum
        t0 += __shfl_xor(t0, 16, 64); t0 += __shfl_xor(t0, 32, 64);
        t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
        if (lane < BB_NB) {
          const double bsum = (lane < 16) ? t0 : t1;
          ys[lane] = (rhs && lane < nbk) ? rhs[j0 + lane] - bsum : 0.0;
        }
      }
      int r = lane & 31;
      asm volatile("" : "+v"(r));   // opaque per block column: the lane predicates (r == j, r >= cc) below are otherwise hoisted out
                                    // of the column loop as loop invariants - 700 spilled SGPRs, reloaded through v_readlane
      double t = ys[r];
      double dinv = 1.0;                                // lane j < 32: 1 / L_jj
      int bad = 0;
      double a[16];
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) a[cc] = (cc <= r) ? Pan[(size_t)cc * RS + r] : 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double piv = readlane_f64(a[j], j);
        if (!(piv > 0.0) && !bad) bad = j + 1;
        const double inv = bb_rsqrt1(piv);
        if (lane == j) dinv = inv;                        // 1 / L_jj for the inverse below
        a[j] *= inv;                                      // (lane j: piv / sqrt(piv) = L_jj to an ulp; rows above j carry values nobody reads)
        const double yj = readlane_f64(t, j) * inv;       // forward substitution rides along
        t = (r > j) ? fma(-a[j], yj, t) : ((r == j) ? yj : t);
#pragma unroll
        for (int cc = j + 1; cc < 16; ++cc) {             // no row predicate: the entries above the diagonal are never read (2 v_readlane + 1 FMA)
          const double lcj = readlane_f64(a[j], cc);
          a[cc] = fma(-a[j], lcj, a[cc]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < BB_NB) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) Ld[r * (BB_NB + 1) + cc] = (cc <= r) ? a[cc] : 0.0;
      }
      BBP_STAMP(13);
      {   // D22 (C layout of the 16 x 16 tile) -= L21 L21^T
        bb_d4 c22;
#pragma unroll
        for (int i = 0; i < 4; ++i) c22[i] = Pan[(size_t)(16 + ln) * RS + 16 + lg + 4 * i];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double av = Ld[(16 + ln) * (BB_NB + 1) + 4 * s4 + lg];      // A[m][k] = L21[m][k] = B[k][n = m]
          c22 = __builtin_amdgcn_mfma_f64_16x16x4f64(-av, av, c22, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) Pan[(size_t)(16 + ln) * RS + 16 + lg + 4 * i] = c22[i];
      }
      BBP_STAMP(14);
      double b2[16];
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) b2[cc] = (r >= 16 && 16 + cc <= r) ? Pan[(size_t)(16 + cc) * RS + r] : 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double piv = readlane_f64(b2[j], 16 + j);
        if (!(piv > 0.0) && !bad) bad = 16 + j + 1;
        const double inv = bb_rsqrt1(piv);
        if (lane == 16 + j) dinv = inv;
        b2[j] *= inv;
        const double yj = readlane_f64(t, 16 + j) * inv;
        t = (r > 16 + j) ? fma(-b2[j], yj, t) : ((r == 16 + j) ? yj : t);
#pragma unroll
        for (int cc = j + 1; cc < 16; ++cc) {
          const double lcj = readlane_f64(b2[j], 16 + cc);
          b2[cc] = fma(-b2[j], lcj, b2[cc]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < BB_NB) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) Ld[r * (BB_NB + 1) + 16 + cc] = (r >= 16 && 16 + cc <= r) ? b2[cc] : 0.0;
        if (rhs && r < nbk) rhs[j0 + r] = t;
        ys[BB_NB + lane] = dinv;
      }
      if (lane == 0 && bad) atomicCAS(info, 0, (int)(j0 + bad));
      BBP_STAMP(5);
      {   // inverses of the two 16 x 16 triangles: lane (16 h + j) solves L_hh x = e_j
        int jj = lane & 15, o = lane & 16;
        asm volatile("" : "+v"(jj));
        double x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = (q == jj) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          x[q] *= ys[BB_NB + o + q];                      // (the reciprocal pivots of the factorisation: no divisions)
#pragma unroll
          for (int rr = q + 1; rr < 16; ++rr) x[rr] = fma(-Ld[(o + rr) * (BB_NB + 1) + o + q], x[q], x[rr]);
        }
        if (lane < BB_NB) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) Li[(o + rr) * (BB_NB + 1) + o + jj] = x[rr];
          if (o == 0) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) Li[rr * (BB_NB + 1) + 16 + jj] = 0.0;      // upper-right block of the inverse
          }
        }
      }
      BBP_STAMP(10);
      {   // lower-left block: -L22^-1 (L21 L11^-1)
        bb_d4 T = {0.0, 0.0, 0.0, 0.0}, U = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          T = __builtin_amdgcn_mfma_f64_16x16x4f64(Ld[(16 + ln) * (BB_NB + 1) + 4 * s4 + lg], Li[(4 * s4 + lg) * (BB_NB + 1) + ln], T, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) Tt[(lg + 4 * i) * 17 + ln] = T[i];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          U = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[(16 + ln) * (BB_NB + 1) + 16 + 4 * s4 + lg], Tt[(4 * s4 + lg) * 17 + ln], U, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) Li[(16 + lg + 4 * i) * (BB_NB + 1) + ln] = -U[i];
      }
      BBP_STAMP(11);
    } else if (worker) {
      int lno = ln, lgo = lg;
      asm volatile("" : "+v"(lno), "+v"(lgo));
      bbp_acc_store<NTW, BBP_NWK>(acc, Pan, RS, R16, R0, lno, lgo);    // (the A operands of the solve are read back in another lane map)
    }
    BBP_STAMP(12);
    __syncthreads();
    BBP_STAMP(6);
    // ---- rows below the diagonal block (waves 1..3): X = Pan Linv^T on the matrix cores, stored straight from the C layout
    // (16 lanes = 16 columns of the band per row: scattered 32-byte pieces, but fire-and-forget; a second pass through the LDS
    // for row-contiguous stores cost a barrier and 13 K cycles of write-completion wait on the critical path)
    if (wv == 0) {
      // the diagonal block itself back to the band - AFTER the barrier that releases the other waves into the solve: nothing on this
      // workgroup's chain reads it from the band (4.6 K cycles that sat between the inverse and the early tiles)
      for (int idx = lane; idx < BB_NB * BB_NB; idx += 64) {
        const int rr = idx % BB_NB, cc = idx / BB_NB;
        if (rr >= cc && rr < nbk && cc < nbk && rr - cc <= bw) Pb[(j0 + cc) * LD + (rr - cc)] = Ld[rr * (BB_NB + 1) + cc];
      }
    } else if (worker) {
      int ln = lane & 15, lg = lane >> 4;             // re-derived per block column and made opaque: the LDS / band addresses below are
      asm volatile("" : "+v"(ln), "+v"(lg));          // loop invariants otherwise, get hoisted out of the column loop, spilled, and come back
                                                      // through scratch loads each followed by s_waitcnt vmcnt(0) (43 K cycles for this phase)
      double bl[2][8];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) bl[ct][s8] = Li[(16 * ct + ln) * (BB_NB + 1) + 4 * s8 + lg];    // B[k][n] = Linv[n][k]
      double* __restrict__ xb[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) xb[ct] = Pb + (j0 + 16 * ct + ln) * (LD - 1) + j0 + 16 * R0 + lg;
      constexpr int GS = 3;
#pragma unroll
      for (int g0 = 0; g0 < NTW; g0 += GS) {
        double av[GS][8];                             // a group's A operands first (LDS latency once per group)
#pragma unroll
        for (int u = 0; u < GS; ++u) {
          const int rt = (g0 + u < NTW && R0 + BBP_NWK * (g0 + u) < R16) ? R0 + BBP_NWK * (g0 + u) : 2;
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) av[u][s8] = Pan[(size_t)(4 * s8 + lg) * RS + 16 * rt + ln];
        }
#pragma unroll
        for (int u = 0; u < GS; ++u) {
          const int rt = R0 + BBP_NWK * (g0 + u);
          if (g0 + u < NTW && rt < R16 && j0 + 16 * rt < M) {
            bb_d4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
              x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][s8], bl[0][s8], x0, 0, 0, 0);
              x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][s8], bl[1][s8], x1, 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int col = 16 * ct + ln, rl = 16 * rt + lg + 4 * i;
                // element (j0 + rl, j0 + col) = xb[ct][16 * BBP_NWK * (g0 + u) + 4 i]: two base pointers per lane, immediate offsets
                if (col < nbk && j0 + rl < M && rl - col <= bw) xb[ct][16 * BBP_NWK * (g0 + u) + 4 * i] = (ct == 0) ? x0[i] : x1[i];
              }
          }
          if (g0 + u == 0 && widx <= 1) {             // rows j0+32 .. j0+63 are in the band (or do not exist): early arrival
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");             // (the wave's stores; the counter itself then needs no second write-back)
            if (lane == 0) __hip_atomic_fetch_add(early, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    __syncthreads();
    BBP_STAMP(8);
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(done, (unsigned)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    BBP_STAMP(9);
  }
}

// backward substitution x = L^-T c for one block column (called for j0 descending): one workgroup.
__global__ __launch_bounds__(1024) void bb_backsolve_kernel(const double* __restrict__ Pb, long M, int bw, long LD, long j0,
                                                            double* __restrict__ x) {
  __shared__ double Ls[BB_NB][BB_NB + 1];
  __shared__ double ts[BB_NB];
  const int tid = threadIdx.x;
  const int nbk = (int)((M - j0 < BB_NB) ? (M - j0) : BB_NB);
  for (int idx = tid; idx < BB_NB * BB_NB; idx += blockDim.x) {
    int r = idx / BB_NB, c = idx % BB_NB;
    double v = (r == c) ? 1.0 : 0.0;
    if (r >= c && r < nbk && c < nbk) v = (r - c <= bw) ? Pb[(j0 + c) * LD + (r - c)] : 0.0;
    Ls[r][c] = v;
  }
  const long r_lo = j0 + nbk;
  long r_hi = j0 + nbk - 1 + bw;
  if (r_hi > M - 1) r_hi = M - 1;
  const int c = tid / 32, part = tid % 32;  // 32 columns x 32 row-lanes (a half wave per column)
  double acc = 0.0;
  if (c < nbk) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four independent load / FMA streams
    long r = r_lo + part;
    for (; r + 96 <= r_hi; r += 128) {
      const long d = r - (j0 + c);
      const double p0 = (d <= bw) ? Pb[(j0 + c) * LD + d] : 0.0, p1 = (d + 32 <= bw) ? Pb[(j0 + c) * LD + d + 32] : 0.0;
      const double p2 = (d + 64 <= bw) ? Pb[(j0 + c) * LD + d + 64] : 0.0, p3 = (d + 96 <= bw) ? Pb[(j0 + c) * LD + d + 96] : 0.0;
      a0 = fma(p0, x[r], a0); a1 = fma(p1, x[r + 32], a1); a2 = fma(p2, x[r + 64], a2); a3 = fma(p3, x[r + 96], a3);
    }
    for (; r <= r_hi; r += 32) {
      const long d = r - (j0 + c);
      if (d <= bw) a0 = fma(Pb[(j0 + c) * LD + d], x[r], a0);
    }
    acc = (a0 + a1) + (a2 + a3);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (part == 0) ts[c] = (c < nbk) ? x[j0 + c] - acc : 0.0;
  __syncthreads();
  if (tid == 0) {
    for (int cc = nbk - 1; cc >= 0; --cc) {
      double t = ts[cc];
      for (int p = cc + 1; p < nbk; ++p) t = fma(-Ls[p][cc], ts[p], t);
      ts[cc] = t / Ls[cc][cc];
      x[j0 + cc] = ts[cc];
    }
  }
}

// the whole backward substitution in ONE launch: the block columns are strictly sequential (x of block j0 needs every x below it), so
// one 1024-thread workgroup walks them with a barrier per step instead of 512 launches of the kernel above (19 us each, 10 ms at
// 128 x 128).  x written in earlier steps is read with device-scope loads (bypassing this CU's L1, which may hold the old line).
__global__ __launch_bounds__(1024) void bb_backsolve_persistent_kernel(const double* __restrict__ Pb, long M, int bw, long LD,
                                                                       double* __restrict__ x) {
  __shared__ double Ls[BB_NB][BB_NB + 1];
  __shared__ double ts[BB_NB];
  const int tid = threadIdx.x;
  const int c = tid / 32, part = tid % 32;            // 32 columns x 32 row-lanes (a half wave per column)
  for (long j0 = ((M - 1) / BB_NB) * BB_NB; j0 >= 0; j0 -= BB_NB) {
    const int nbk = (int)((M - j0 < BB_NB) ? (M - j0) : BB_NB);
    for (int idx = tid; idx < BB_NB * BB_NB; idx += blockDim.x) {
      int r = idx / BB_NB, cc = idx % BB_NB;
      double v = (r == cc) ? 1.0 : 0.0;
      if (r >= cc && r < nbk && cc < nbk) v = (r - cc <= bw) ? Pb[(j0 + cc) * LD + (r - cc)] : 0.0;
      Ls[r][cc] = v;
    }
    const long r_lo = j0 + nbk;
    long r_hi = j0 + nbk - 1 + bw;
    if (r_hi > M - 1) r_hi = M - 1;
    double acc = 0.0;
    if (c < nbk) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four independent load / FMA streams
      long r = r_lo + part;
      for (; r + 96 <= r_hi; r += 128) {
        const long d = r - (j0 + c);
        const double p0 = (d <= bw) ? Pb[(j0 + c) * LD + d] : 0.0, p1 = (d + 32 <= bw) ? Pb[(j0 + c) * LD + d + 32] : 0.0;
        const double p2 = (d + 64 <= bw) ? Pb[(j0 + c) * LD + d + 64] : 0.0, p3 = (d + 96 <= bw) ? Pb[(j0 + c) * LD + d + 96] : 0.0;
        a0 = fma(p0, __hip_atomic_load(x + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a0);
        a1 = fma(p1, __hip_atomic_load(x + r + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1);
        a2 = fma(p2, __hip_atomic_load(x + r + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a2);
        a3 = fma(p3, __hip_atomic_load(x + r + 96, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a3);
      }
      for (; r <= r_hi; r += 32) {
        const long d = r - (j0 + c);
        if (d <= bw) a0 = fma(Pb[(j0 + c) * LD + d], __hip_atomic_load(x + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a0);
      }
      acc = (a0 + a1) + (a2 + a3);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (part == 0) ts[c] = (c < nbk) ? __hip_atomic_load(x + j0 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - acc : 0.0;
    __syncthreads();
    if (tid < 64) {                                   // the 32 x 32 triangle: one wavefront, lane = row, columns descending
      const int r = tid & 31;
      double t = ts[r];
      for (int cc = nbk - 1; cc >= 0; --cc) {
        const double xc = readlane_f64(t, cc) / Ls[cc][cc];      // (t of row cc is final when its turn comes)
        t = (r == cc) ? xc : ((r < cc) ? fma(-Ls[cc][r], xc, t) : t);
      }
      if (tid < nbk) __hip_atomic_store(x + j0 + r, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

// sum of 2 log diag over the band factor
__global__ __launch_bounds__(1024) void bb_logdet_kernel(const double* __restrict__ Pb, long M, long LD, double* __restrict__ out) {
  __shared__ double scratch[16];
  double acc = 0.0;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += (long)gridDim.x * blockDim.x) acc += 2.0 * log(Pb[j * LD]);
  double tot = block_sum(acc, scratch);
  if (threadIdx.x == 0) __hip_atomic_fetch_add(out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// posterior mean / "Kuu part" of the variance per test point:  mean = phi*^T alpha,
// q = (phi1^T S1 phi1)(phi2^T S2 phi2) = phi*^T Kuu^-1 phi*   (Kuu^-1 = K1^-1 (x) K2^-1, only band entries needed)
template <int K>
__global__ void predict_kron2d_kernel(const double* __restrict__ X, long n, const double* __restrict__ mesh1, int n1,
                                      double id1, int m1, const double* __restrict__ mesh2, int n2, double id2, int m2,
                                      const double* __restrict__ alpha, const double* __restrict__ S1,
                                      const double* __restrict__ S2, double* __restrict__ mean, double* __restrict__ qk) {
  long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double2 xv = *reinterpret_cast<const double2*>(X + 2 * p);
  const int i1 = neighbour_index(xv.x, mesh1, n1, mesh1[0], id1);
  const int i2 = neighbour_index(xv.y, mesh2, n2, mesh2[0], id2);
  double v1[K + 1], v2[K + 1];
  bspline_pieces<K>((xv.x - mesh1[i1]) * id1, v1);
  bspline_pieces<K>((xv.y - mesh2[i2]) * id2, v2);
  double mu = 0.0;
#pragma unroll
  for (int a = 0; a <= K; ++a)
#pragma unroll
    for (int b = 0; b <= K; ++b) mu = fma(v1[a] * v2[b], alpha[(long)(i1 + K - a) * m2 + (i2 + K - b)], mu);
  mean[p] = mu;
  if (qk) {
    double q1 = 0.0, q2 = 0.0;
#pragma unroll
    for (int a = 0; a <= K; ++a)
#pragma unroll
      for (int a2 = 0; a2 <= K; ++a2) {  // rows i+K-a, i+K-a2
        int lo1 = i1 + K - (a > a2 ? a : a2), d = a > a2 ? a - a2 : a2 - a;
        q1 = fma(v1[a] * v1[a2], S1[(long)d * m1 + lo1], q1);
        int lo2 = i2 + K - (a > a2 ? a : a2);
        q2 = fma(v2[a] * v2[a2], S2[(long)d * m2 + lo2], q2);
      }
    qk[p] = q1 * q2;
  }
}


// ---------------------------------------------------------------------------------------------------------
// Selected inverse of P on the band, through dense super-blocks.
// With block size Bb >= bw (a multiple of 32) the band factor is block BIDIAGONAL: diagonal blocks L_ii (lower
// triangular) and sub-diagonal blocks L_{i+1,i}.  bb_blocks_kernel unpacks the column-major band factor into those two
// dense batches; the caller runs the block recursion
//     G_i = L_{i+1,i} L_ii^-1,   Sigma_{i+1,i} = -Sigma_{i+1,i+1} G_i,   Sigma_ii = (L_ii L_ii^T)^-1 - G_i^T Sigma_{i+1,i}
// with one batched triangular solve and 2(n-1) dense fp64 GEMMs (plain library calls), and the kernels below read
// Sigma[row, col] (|row - col| <= bw <= Bb) out of the two block batches.
// ---------------------------------------------------------------------------------------------------------
__global__ void bb_blocks_kernel(const double* __restrict__ Lb, long M, int bw, long LD, int Bb, long nblk,
                                 double* __restrict__ diag, double* __restrict__ sub) {
  const long per = (long)Bb * Bb;
  const long total = (2 * nblk - 1) * per;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const bool is_sub = t >= nblk * per;
    const long u = is_sub ? t - nblk * per : t;
    const long blk = u / per;
    const int r = (int)((u - blk * per) / Bb), c = (int)(u % Bb);
    const long row = (blk + (is_sub ? 1 : 0)) * Bb + r, col = blk * Bb + c;
    double v = 0.0;
    if (row < M && col < M && row >= col && row - col <= bw) v = Lb[col * LD + (row - col)];
    else if (!is_sub && r == c && row >= M) v = 1.0;   // identity padding of the last block
    (is_sub ? sub : diag)[u] = v;
  }
}

// Sigma[row, col] for row >= col, row - col <= Bb
__device__ __forceinline__ double sig_read(const double* __restrict__ SigD, const double* __restrict__ SigS, int Bb, long row, long col) {
  const long bi = row / Bb, bj = col / Bb;
  const long per = (long)Bb * Bb;
  const int r = (int)(row - bi * Bb), c = (int)(col - bj * Bb);
  return (bi == bj) ? SigD[bi * per + (long)r * Bb + c] : SigS[bj * per + (long)r * Bb + c];
}
// The band-restricted inverse as the selected inverse leaves it.  Plain layout (nb = 0): super-blocks of Bb anchored at column 0.
// Twisted layout (two-sided factorisation, asvgp_kron_assemble_twisted): two stacks of nb super-blocks, SigD [2][nb][Bb][Bb] and
// SigS [2][nb-1][Bb][Bb].  Stack 0 holds the TOP system - original columns [0, top_end) shifted down by `padt` identity columns, the
// separator [top_end - Bb, top_end) its last block; stack 1 the BOTTOM system in reversed order - system index u <-> original column
// off - u - whose last block is the same separator.  An entry with row < top_end is read from stack 0, any other from stack 1 (both
// of its indices are then >= top_end - Bb, i.e. inside the bottom system: the separator is at least a bandwidth wide).
struct SigView {
  const double* D; const double* S; int Bb; long nb, top_end, padt, off;
  __device__ __forceinline__ double at(long row, long col) const {
    if (nb == 0) return sig_read(D, S, Bb, row, col);
    const long per = (long)Bb * Bb;
    if (row < top_end) return sig_read(D, S, Bb, row + padt, col + padt);
    return sig_read(D + nb * per, S + (nb - 1) * per, Bb, off - col, off - row);      // reversal swaps the triangle: (row, col) -> (off - col, off - row)
  }
};

// 11 contractions over the block band (full symmetric sums: off-diagonal entries count twice):
//  0 tr(Sig A)  1 a^T A a  2 tr(Sig X1) 3 a^T X1 a  4 tr(Sig X2) 5 a^T X2 a  6 tr(Sig Kuu) 7 a^T Kuu a
//  8 tr((Z1 (x) S2) A)  9 tr((S1 (x) Z2) A)  10 tr((S1 (x) S2) A)      X1 = dK1 (x) K2,  X2 = K1 (x) dK2
__global__ void kron_grad_terms_kernel(SigView sig,
                                       const double* __restrict__ alpha, const double* __restrict__ Ablk,
                                       const double* __restrict__ K1, const double* __restrict__ K2,
                                       const double* __restrict__ dK1, const double* __restrict__ dK2,
                                       const double* __restrict__ S1, const double* __restrict__ S2,
                                       const double* __restrict__ Z1, const double* __restrict__ Z2, int k, int m1, int m2,
                                       double* __restrict__ out) {
  __shared__ double scratch[16];
  const long Mtot = (long)m1 * m2;
  const int noff = kron_noff(k);
  double acc[11];
#pragma unroll
  for (int q = 0; q < 11; ++q) acc[q] = 0.0;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < Mtot * noff; t += (long)gridDim.x * blockDim.x) {
    const int o = (int)(t / Mtot);
    const long c = t - (long)o * Mtot;
    int d1, d2;
    if (o <= k) { d1 = 0; d2 = o; } else { int q = o - (k + 1); d1 = 1 + q / (2 * k + 1); d2 = q % (2 * k + 1) - k; }
    const int i1 = (int)(c / m2), i2 = (int)(c - (long)i1 * m2);
    if (i1 + d1 >= m1 || i2 + d2 < 0 || i2 + d2 >= m2) continue;
    const long row = c + (long)d1 * m2 + d2;
    const double w = (o == 0) ? 1.0 : 2.0;
    const double a = Ablk[t];
    const double sg = sig.at(row, c);
    const double aa = alpha[row] * alpha[c];
    const double k1 = K1[(long)d1 * m1 + i1], k2 = band_sym(K2, m2, i2, d2);
    const double x1 = dK1[(long)d1 * m1 + i1] * k2, x2 = k1 * band_sym(dK2, m2, i2, d2), ku = k1 * k2;
    const double s1 = S1[(long)d1 * m1 + i1], s2 = band_sym(S2, m2, i2, d2);
    acc[0] = fma(w * sg, a, acc[0]);
    acc[1] = fma(w * aa, a, acc[1]);
    acc[2] = fma(w * sg, x1, acc[2]);
    acc[3] = fma(w * aa, x1, acc[3]);
    acc[4] = fma(w * sg, x2, acc[4]);
    acc[5] = fma(w * aa, x2, acc[5]);
    acc[6] = fma(w * sg, ku, acc[6]);
    acc[7] = fma(w * aa, ku, acc[7]);
    acc[8] = fma(w * a, Z1[(long)d1 * m1 + i1] * s2, acc[8]);
    acc[9] = fma(w * a, s1 * band_sym(Z2, m2, i2, d2), acc[9]);
    acc[10] = fma(w * a, s1 * s2, acc[10]);
  }
#pragma unroll
  for (int q = 0; q < 11; ++q) {
    double tot = block_sum(acc[q], scratch);
    __syncthreads();
    if (threadIdx.x == 0 && tot != 0.0) __hip_atomic_fetch_add(out + q, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// q_P = phi*^T P^-1 phi* per test point from the block batches: (k+1)^2 basis pairs, every index pair inside the band
template <int K>
__global__ void predict_kron2d_var_kernel(const double* __restrict__ X, long n, const double* __restrict__ mesh1, int n1,
                                          double id1, const double* __restrict__ mesh2, int n2, double id2, int m2,
                                          SigView sig, double* __restrict__ qp) {
  long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double2 xv = *reinterpret_cast<const double2*>(X + 2 * p);
  const int i1 = neighbour_index(xv.x, mesh1, n1, mesh1[0], id1);
  const int i2 = neighbour_index(xv.y, mesh2, n2, mesh2[0], id2);
  double v1[K + 1], v2[K + 1];
  bspline_pieces<K>((xv.x - mesh1[i1]) * id1, v1);
  bspline_pieces<K>((xv.y - mesh2[i2]) * id2, v2);
  double q = 0.0;
#pragma unroll
  for (int a = 0; a <= K; ++a)
#pragma unroll
    for (int b = 0; b <= K; ++b) {
      const long ra = (long)(i1 + K - a) * m2 + (i2 + K - b);
      const double wa = v1[a] * v2[b];
#pragma unroll
      for (int a2 = 0; a2 <= K; ++a2)
#pragma unroll
        for (int b2 = 0; b2 <= K; ++b2) {
          const long rb = (long)(i1 + K - a2) * m2 + (i2 + K - b2);
          if (rb > ra) continue;   // lower triangle, doubled below
          const double sg = sig.at(ra, rb);
          q = fma((rb == ra ? 1.0 : 2.0) * wa * v1[a2] * v2[b2], sg, q);
        }
    }
  qp[p] = q;
}

}  // namespace asvgp

using namespace asvgp;

extern "C" size_t asvgp_kron_stats_doubles(int64_t m1, int64_t m2, int k) {
  if (m1 < 1 || m2 < 1 || k < 1 || k > ASVGP_MAX_ORDER) return 0;
  return (size_t)kron_noff(k) * m1 * m2 + (size_t)m1 * m2 + 1;
}

#define KRON_DISPATCH(KK, CALL)                                  \
  switch (KK) {                                                  \
    case 1: { constexpr int K = 1; CALL; } break;                \
    case 2: { constexpr int K = 2; CALL; } break;                \
    case 3: { constexpr int K = 3; CALL; } break;                \
    case 4: { constexpr int K = 4; CALL; } break;                \
    case 5: { constexpr int K = 5; CALL; } break;                \
    default: { constexpr int K = 6; CALL; } break;               \
  }

extern "C" int asvgp_phi_accumulate_kron2d(const double* X, const double* y, int64_t N, const double* mesh1,
                                           int64_t n_mesh1, double delta1, int64_t m1, const double* mesh2,
                                           int64_t n_mesh2, double delta2, int64_t m2, int order, double* stats,
                                           asvgp_stream_t stream) {
  if ((N > 0 && (!X || !y)) || !mesh1 || !mesh2 || !stats || N < 0 || !(delta1 > 0) || !(delta2 > 0) ||
      n_mesh1 != m1 - order + 1 || n_mesh2 != m2 - order + 1) {
    set_error("phi_accumulate_kron2d: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  if (order < 1 || order > ASVGP_MAX_ORDER) { set_error("phi_accumulate_kron2d: order %d unsupported", order); return ASVGP_ERR_UNSUPPORTED; }
  if ((reinterpret_cast<uintptr_t>(X) & 15) != 0) { set_error("phi_accumulate_kron2d: X must be 16-byte aligned (N,2) row-major"); return ASVGP_ERR_BAD_ARG; }
  hipStream_t st = as_stream(stream);
  const size_t nd = asvgp_kron_stats_doubles(m1, m2, order);
  hipError_t e = hipMemsetAsync(stats, 0, nd * sizeof(double), st);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
  if (N == 0) return ASVGP_OK;
  const long Mtot = (long)m1 * m2;
  double* Ablk = stats;
  double* rhs = stats + (size_t)kron_noff(order) * Mtot;
  double* yy = rhs + Mtot;
  long blocks = (N + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  KRON_DISPATCH(order, hipLaunchKernelGGL(phi_kron2d_kernel<K>, dim3((unsigned)blocks), dim3(256), 0, st, X, y, (long)N, mesh1,
                                          (int)n_mesh1, 1.0 / delta1, (int)m1, mesh2, (int)n_mesh2, 1.0 / delta2, (int)m2,
                                          Ablk, rhs, yy));
  return check_launch("phi_accumulate_kron2d");
}

extern "C" int asvgp_kron_evaluate_2d(const double* X, int64_t N, const double* mesh1, int64_t n_mesh1, double delta1,
                                      const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                      int64_t* rows, double* data, asvgp_stream_t stream) {
  if (!X || !mesh1 || !mesh2 || !rows || !data || N < 0 || order < 1 || order > ASVGP_MAX_ORDER) {
    set_error("kron_evaluate_2d: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  if (N == 0) return ASVGP_OK;
  KRON_DISPATCH(order, hipLaunchKernelGGL(kron_evaluate_kernel<K>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0,
                                          as_stream(stream), X, (long)N, mesh1, (int)n_mesh1, 1.0 / delta1, mesh2,
                                          (int)n_mesh2, 1.0 / delta2, (int)m2, reinterpret_cast<long long*>(rows), data));
  return check_launch("kron_evaluate_2d");
}

extern "C" int asvgp_kron_assemble(const double* K1, const double* K2, const double* S1, const double* S2,
                                   const double* Ablk, int k, int64_t m1, int64_t m2, double noise_variance, double* Pb,
                                   double* trace_out, asvgp_stream_t stream) {
  if (!K1 || !K2 || !Ablk || k < 1 || k > ASVGP_MAX_ORDER || m1 < 1 || m2 < 1 || !(noise_variance > 0) || (!Pb && !trace_out) ||
      (trace_out && (!S1 || !S2))) {
    set_error("kron_assemble: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  hipStream_t st = as_stream(stream);
  const long Mtot = (long)m1 * m2, LD = (long)k * m2 + k + 1;
  hipError_t e = hipSuccess;
  if (Pb) e = hipMemsetAsync(Pb, 0, sizeof(double) * Mtot * LD, st);
  if (e == hipSuccess && trace_out) e = hipMemsetAsync(trace_out, 0, sizeof(double), st);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
  long total = Mtot * kron_noff(k);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(kron_assemble_kernel, dim3((unsigned)blocks), dim3(256), 0, st, K1, K2, trace_out ? S1 : nullptr, S2,
                     Ablk, k, (int)m1, (int)m2, noise_variance, LD, Pb, trace_out);
  return check_launch("kron_assemble");
}

static bool twist_ok(int64_t M, int64_t Bb, int64_t nb, int64_t top_end, int64_t padt, int64_t padb);

// P as the two systems of the two-sided factorisation (kron_assemble_twisted_kernel): Pt and Pr, each (nb Bb) x (bw + 1) doubles.
// top_end = h + Bb (end of the separator), padt = nb Bb - top_end, padb = nb Bb - (M - h).
extern "C" int asvgp_kron_assemble_twisted(const double* K1, const double* K2, const double* S1, const double* S2,
                                           const double* Ablk, int k, int64_t m1, int64_t m2, double noise_variance, int64_t Bb, int64_t nb,
                                           int64_t top_end, int64_t padt, int64_t padb, double* Pt, double* Pr, double* trace_out,
                                           asvgp_stream_t stream) {
  if (!K1 || !K2 || !Ablk || k < 1 || k > ASVGP_MAX_ORDER || m1 < 1 || m2 < 1 || !(noise_variance > 0) || !Pt || !Pr ||
      (trace_out && (!S1 || !S2)) || Bb < (int64_t)k * m2 + k || !twist_ok(m1 * m2, Bb, nb, top_end, padt, padb)) {
    set_error("kron_assemble_twisted: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  hipStream_t st = as_stream(stream);
  const long LD = (long)k * m2 + k + 1, Ms = (long)nb * Bb;
  hipError_t e = hipMemsetAsync(Pt, 0, sizeof(double) * Ms * LD, st);
  if (e == hipSuccess) e = hipMemsetAsync(Pr, 0, sizeof(double) * Ms * LD, st);
  if (e == hipSuccess && trace_out) e = hipMemsetAsync(trace_out, 0, sizeof(double), st);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
  long total = (long)m1 * m2 * kron_noff(k);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(kron_assemble_twisted_kernel, dim3((unsigned)blocks), dim3(256), 0, st, K1, K2, trace_out ? S1 : nullptr, S2,
                     Ablk, k, (int)m1, (int)m2, noise_variance, LD, (long)(top_end - Bb), (long)top_end, (long)padt, (long)padb, Pt, Pr, trace_out);
  return check_launch("kron_assemble_twisted");
}

extern "C" int asvgp_blockband_cholesky(double* Pb, int64_t M, int64_t bw, double* rhs, double* logdet, int* info,
                                        asvgp_stream_t stream) {
  if (!Pb || !info || M < 1 || bw < 0) { set_error("blockband_cholesky: bad argument"); return ASVGP_ERR_BAD_ARG; }
  hipStream_t st = as_stream(stream);
  const long LD = bw + 1;
  hipError_t e = hipMemsetAsync(info, 0, sizeof(int), st);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
  const int R16 = bbp_row_tiles((int)bw);
  static const bool persistent_off = getenv("ASVGP_BB_PERSISTENT") && atoi(getenv("ASVGP_BB_PERSISTENT")) == 0;   // (diagnostic: the per-panel launches)
  if (R16 <= 29 && !persistent_off) {
    // one launch: dataflow over block columns.  The arrival counter lives in a stream-ordered allocation (re-entrant across streams).
    unsigned* done = nullptr;
    e = hipMallocAsync(reinterpret_cast<void**>(&done), 128, st);
    if (e != hipSuccess) { set_error("hipMallocAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
    (void)hipMemsetAsync(done, 0, 128, st);
    const int nbc = (int)((M + BB_NB - 1) / BB_NB);
    const int G = nbc < 16 ? nbc : 16;
    const size_t lds_bytes = bbp_lds_bytes((int)bw);
    const int ntw = (R16 - 2 + BBP_NWK - 1) / BBP_NWK;  // row tiles below the diagonal block per wave (waves 1..BBP_NWK)
    static const int stamp_col = getenv("ASVGP_BB_STAMP_COL") ? atoi(getenv("ASVGP_BB_STAMP_COL")) : -1;
#define BBP_LAUNCH(NTW)                                                                                                              \
    {                                                                                                                                \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(bb_chol_persistent_kernel<NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds_bytes);                                                                                       \
      if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); (void)hipFreeAsync(done, st); return ASVGP_ERR_LDS_CAPACITY; } \
      hipLaunchKernelGGL(bb_chol_persistent_kernel<NTW>, dim3(G), dim3(BBP_THREADS), lds_bytes, st, Pb, (long)M, (int)bw, LD, rhs, info, done, nbc, stamp_col); \
    }
    if (ntw <= 3) BBP_LAUNCH(3) else BBP_LAUNCH(5)      // (R16 <= 29: at most 27 tiles below the diagonal block, 5 per worker wave)
#undef BBP_LAUNCH
    (void)hipFreeAsync(done, st);
  } else
  for (long j0 = 0; j0 < M; j0 += BB_NB) {
    hipLaunchKernelGGL(bb_panel_kernel, dim3(1), dim3(512), 0, st, Pb, (long)M, (int)bw, LD, j0, rhs, info);
    long nbk = (M - j0 < BB_NB) ? (M - j0) : BB_NB;
    long r_lo = j0 + nbk, r_hi = j0 + nbk - 1 + bw;
    if (r_hi > M - 1) r_hi = M - 1;
    long nrows = r_hi - r_lo + 1;
    if (nrows > 0) {
      long nt = (nrows + 31) / 32;
      hipLaunchKernelGGL(bb_update_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, Pb, (long)M, (int)bw, LD, j0);
    }
  }
  if (logdet) {
    e = hipMemsetAsync(logdet, 0, sizeof(double), st);
    if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
    hipLaunchKernelGGL(bb_logdet_kernel, dim3(16), dim3(1024), 0, st, Pb, (long)M, LD, logdet);
  }
  return check_launch("blockband_cholesky");
}

extern "C" int asvgp_debug_bbp_stamps(unsigned long long* out32) {   // diagnostic (not in the public header)
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(bbp_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? ASVGP_OK : ASVGP_ERR_HIP;
}

extern "C" int asvgp_blockband_backsolve(const double* Lb, int64_t M, int64_t bw, double* x, asvgp_stream_t stream) {
  if (!Lb || !x || M < 1 || bw < 0) { set_error("blockband_backsolve: bad argument"); return ASVGP_ERR_BAD_ARG; }
  hipStream_t st = as_stream(stream);
  const long LD = bw + 1;
  static const bool persistent_off = getenv("ASVGP_BB_PERSISTENT") && atoi(getenv("ASVGP_BB_PERSISTENT")) == 0;   // (diagnostic: one launch per block column)
  if (!persistent_off) {
    hipLaunchKernelGGL(bb_backsolve_persistent_kernel, dim3(1), dim3(1024), 0, st, Lb, (long)M, (int)bw, LD, x);
    return check_launch("blockband_backsolve");
  }
  for (long j0 = ((M - 1) / BB_NB) * BB_NB; j0 >= 0; j0 -= BB_NB)
    hipLaunchKernelGGL(bb_backsolve_kernel, dim3(1), dim3(1024), 0, st, Lb, (long)M, (int)bw, LD, j0, x);
  return check_launch("blockband_backsolve");
}

extern "C" int asvgp_predict_kron2d(const double* Xnew, int64_t n, const double* mesh1, int64_t n_mesh1, double delta1,
                                    int64_t m1, const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                    const double* alpha, const double* S1, const double* S2, double* mean, double* qk,
                                    asvgp_stream_t stream) {
  if (!Xnew || !mesh1 || !mesh2 || !alpha || !mean || n < 0 || order < 1 || order > ASVGP_MAX_ORDER || (qk && (!S1 || !S2))) {
    set_error("predict_kron2d: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  if (n == 0) return ASVGP_OK;
  KRON_DISPATCH(order, hipLaunchKernelGGL(predict_kron2d_kernel<K>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                          as_stream(stream), Xnew, (long)n, mesh1, (int)n_mesh1, 1.0 / delta1, (int)m1, mesh2,
                                          (int)n_mesh2, 1.0 / delta2, (int)m2, alpha, S1, S2, mean, qk));
  return check_launch("predict_kron2d");
}

extern "C" int asvgp_blockband_to_blocks(const double* Lb, int64_t M, int64_t bw, int64_t Bb, double* diag, double* sub,
                                         asvgp_stream_t stream) {
  if (!Lb || !diag || M < 1 || bw < 0 || Bb < bw || Bb < 1 || (Bb % 32) != 0) { set_error("blockband_to_blocks: bad argument (Bb must be a multiple of 32 and >= bw)"); return ASVGP_ERR_BAD_ARG; }
  const long nblk = (M + Bb - 1) / Bb;
  if (nblk > 1 && !sub) { set_error("blockband_to_blocks: sub is null"); return ASVGP_ERR_BAD_ARG; }
  const long total = (2 * nblk - 1) * Bb * Bb;
  long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(bb_blocks_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), Lb, (long)M, (int)bw, (long)(bw + 1),
                     (int)Bb, nblk, diag, sub);
  return check_launch("blockband_to_blocks");
}

static int kron_grad_terms_entry(SigView sig, const double* alpha,
                                 const double* Ablk, const double* K1, const double* K2, const double* dK1,
                                 const double* dK2, const double* S1, const double* S2, const double* Z1,
                                 const double* Z2, int k, int64_t m1, int64_t m2, double* out11, asvgp_stream_t stream) {
  if (!sig.D || !alpha || !Ablk || !K1 || !K2 || !dK1 || !dK2 || !S1 || !S2 || !Z1 || !Z2 || !out11 || k < 1 || k > ASVGP_MAX_ORDER ||
      m1 < 1 || m2 < 1 || sig.Bb < (int64_t)k * m2 + k) { set_error("kron_grad_terms: bad argument"); return ASVGP_ERR_BAD_ARG; }
  if ((sig.nb > 1 || ((m1 * m2 + sig.Bb - 1) / sig.Bb) > 1) && !sig.S) { set_error("kron_grad_terms: SigS is null"); return ASVGP_ERR_BAD_ARG; }
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(out11, 0, 11 * sizeof(double), st);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
  const long total = (long)m1 * m2 * kron_noff(k);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(kron_grad_terms_kernel, dim3((unsigned)blocks), dim3(256), 0, st, sig, alpha, Ablk, K1, K2,
                     dK1, dK2, S1, S2, Z1, Z2, k, (int)m1, (int)m2, out11);
  return check_launch("kron_grad_terms");
}

extern "C" int asvgp_kron_grad_terms(const double* SigD, const double* SigS, int64_t Bb, const double* alpha,
                                     const double* Ablk, const double* K1, const double* K2, const double* dK1,
                                     const double* dK2, const double* S1, const double* S2, const double* Z1,
                                     const double* Z2, int k, int64_t m1, int64_t m2, double* out11, asvgp_stream_t stream) {
  return kron_grad_terms_entry(SigView{SigD, SigS, (int)Bb, 0, 0, 0, 0}, alpha, Ablk, K1, K2, dK1, dK2, S1, S2, Z1, Z2, k, m1, m2, out11, stream);
}

static bool twist_ok(int64_t M, int64_t Bb, int64_t nb, int64_t top_end, int64_t padt, int64_t padb) {
  return nb >= 2 && top_end >= Bb && top_end <= M && padt >= 0 && padb >= 0 && top_end + padt == nb * Bb && (M - (top_end - Bb)) + padb == nb * Bb;
}

extern "C" int asvgp_kron_grad_terms_twisted(const double* SigD, const double* SigS, int64_t Bb, int64_t nb, int64_t top_end, int64_t padt,
                                             int64_t padb, const double* alpha,
                                             const double* Ablk, const double* K1, const double* K2, const double* dK1,
                                             const double* dK2, const double* S1, const double* S2, const double* Z1,
                                             const double* Z2, int k, int64_t m1, int64_t m2, double* out11, asvgp_stream_t stream) {
  if (!twist_ok(m1 * m2, Bb, nb, top_end, padt, padb)) { set_error("kron_grad_terms_twisted: inconsistent layout"); return ASVGP_ERR_BAD_ARG; }
  return kron_grad_terms_entry(SigView{SigD, SigS, (int)Bb, (long)nb, (long)top_end, (long)padt, (long)(padb + m1 * m2 - 1)}, alpha, Ablk, K1, K2, dK1, dK2,
                               S1, S2, Z1, Z2, k, m1, m2, out11, stream);
}

static int predict_kron2d_var_entry(const double* Xnew, int64_t n, const double* mesh1, int64_t n_mesh1, double delta1,
                                    const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                    SigView sig, double* qp, asvgp_stream_t stream) {
  if ((n > 0 && !Xnew) || !mesh1 || !mesh2 || !sig.D || !qp || n < 0 || order < 1 || order > ASVGP_MAX_ORDER ||
      sig.Bb < (int64_t)order * m2 + order) { set_error("predict_kron2d_var: bad argument"); return ASVGP_ERR_BAD_ARG; }
  if (n == 0) return ASVGP_OK;
  if ((reinterpret_cast<uintptr_t>(Xnew) & 15) != 0) { set_error("predict_kron2d_var: Xnew must be 16-byte aligned (n,2) row-major"); return ASVGP_ERR_BAD_ARG; }
  hipStream_t st = as_stream(stream);
  KRON_DISPATCH(order, hipLaunchKernelGGL(predict_kron2d_var_kernel<K>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, Xnew,
                                          (long)n, mesh1, (int)n_mesh1, 1.0 / delta1, mesh2, (int)n_mesh2, 1.0 / delta2, (int)m2,
                                          sig, qp));
  return check_launch("predict_kron2d_var");
}

extern "C" int asvgp_predict_kron2d_var(const double* Xnew, int64_t n, const double* mesh1, int64_t n_mesh1, double delta1,
                                        const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                        const double* SigD, const double* SigS, int64_t Bb, double* qp,
                                        asvgp_stream_t stream) {
  return predict_kron2d_var_entry(Xnew, n, mesh1, n_mesh1, delta1, mesh2, n_mesh2, delta2, m2, order, SigView{SigD, SigS, (int)Bb, 0, 0, 0, 0}, qp, stream);
}

extern "C" int asvgp_predict_kron2d_var_twisted(const double* Xnew, int64_t n, const double* mesh1, int64_t n_mesh1, double delta1,
                                                const double* mesh2, int64_t n_mesh2, double delta2, int64_t m1, int64_t m2, int order,
                                                const double* SigD, const double* SigS, int64_t Bb, int64_t nb, int64_t top_end,
                                                int64_t padt, int64_t padb, double* qp, asvgp_stream_t stream) {
  if (!twist_ok(m1 * m2, Bb, nb, top_end, padt, padb)) { set_error("predict_kron2d_var_twisted: inconsistent layout"); return ASVGP_ERR_BAD_ARG; }
  return predict_kron2d_var_entry(Xnew, n, mesh1, n_mesh1, delta1, mesh2, n_mesh2, delta2, m2, order,
                                  SigView{SigD, SigS, (int)Bb, (long)nb, (long)top_end, (long)padt, (long)(padb + m1 * m2 - 1)}, qp, stream);
}

extern "C" int asvgp_kron_cell_index(const double* X, int64_t N, const double* mesh1, int64_t n_mesh1, double delta1,
                                     const double* mesh2, int64_t n_mesh2, double delta2, int* cell, asvgp_stream_t stream) {
  if ((N > 0 && (!X || !cell)) || !mesh1 || !mesh2 || N < 0 || n_mesh1 < 2 || n_mesh2 < 2 || !(delta1 > 0) || !(delta2 > 0) ||
      (n_mesh1 - 1) * (n_mesh2 - 1) > 0x7fffffff) { set_error("kron_cell_index: bad argument"); return ASVGP_ERR_BAD_ARG; }
  if (N == 0) return ASVGP_OK;
  if ((reinterpret_cast<uintptr_t>(X) & 15) != 0) { set_error("kron_cell_index: X must be 16-byte aligned (N,2) row-major"); return ASVGP_ERR_BAD_ARG; }
  hipLaunchKernelGGL(kron_cell_index_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), X, (long)N, mesh1,
                     (int)n_mesh1, 1.0 / delta1, mesh2, (int)n_mesh2, 1.0 / delta2, cell);
  return check_launch("kron_cell_index");
}

template <typename TIN>
static int phi_accumulate_kron2d_sorted_entry(const TIN* Xs, const TIN* ys, int64_t N, const int64_t* cell_start,
                                                  const double* mesh1, int64_t n_mesh1, double delta1, int64_t m1,
                                                  const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                                  double* stats, asvgp_stream_t stream) {
  if ((N > 0 && (!Xs || !ys)) || !cell_start || !mesh1 || !mesh2 || !stats || N < 0 || !(delta1 > 0) || !(delta2 > 0) ||
      n_mesh1 != m1 - order + 1 || n_mesh2 != m2 - order + 1 || n_mesh1 < 2 || n_mesh2 < 2) {
    set_error("phi_accumulate_kron2d_sorted: bad argument");
    return ASVGP_ERR_BAD_ARG;
  }
  if (order < 1 || order > ASVGP_MAX_ORDER) { set_error("phi_accumulate_kron2d_sorted: order %d unsupported", order); return ASVGP_ERR_UNSUPPORTED; }
  if ((reinterpret_cast<uintptr_t>(Xs) & (2 * sizeof(TIN) - 1)) != 0) { set_error("phi_accumulate_kron2d_sorted: Xs must be aligned to one (x1, x2) pair, (N,2) row-major"); return ASVGP_ERR_BAD_ARG; }
  hipStream_t st = as_stream(stream);
  const size_t nd = asvgp_kron_stats_doubles(m1, m2, order);
  hipError_t e = hipSuccess;
  if (N == 0) {
    e = hipMemsetAsync(stats, 0, nd * sizeof(double), st);
    if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return ASVGP_ERR_HIP; }
    return ASVGP_OK;
  }
  const long Mtot = (long)m1 * m2;
  double* Ablk = stats;
  double* rhs = stats + (size_t)kron_noff(order) * Mtot;
  double* yy = rhs + Mtot;
  const long ncell = (long)(n_mesh1 - 1) * (n_mesh2 - 1);
  static const bool use_atomics_env = getenv("ASVGP_KRON_PHI_ATOMICS") && atoi(getenv("ASVGP_KRON_PHI_ATOMICS")) != 0;   // (the round-1..3 kernel, for comparison)
  const bool use_atomics = use_atomics_env && sizeof(TIN) == sizeof(double);
  const long ncell_pad = (ncell + KRON_STRIP - 1) / KRON_STRIP * KRON_STRIP;
  double* cellsum = nullptr;
  if (!use_atomics) {
    size_t nout = 0;
    KRON_DISPATCH(order, { nout = (size_t)KronOut<K>::NOUT; });
    if (hipMallocAsync(reinterpret_cast<void**>(&cellsum), nout * (size_t)ncell_pad * sizeof(double), st) != hipSuccess) {
      (void)hipGetLastError();
      cellsum = nullptr;                              // (no room for the staging buffer: the atomic kernel needs none)
    }
  }
  e = cellsum ? hipMemsetAsync(yy, 0, sizeof(double), st)            // (the gather overwrites every band and rhs entry)
              : hipMemsetAsync(stats, 0, nd * sizeof(double), st);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); if (cellsum) (void)hipFreeAsync(cellsum, st); return ASVGP_ERR_HIP; }
  if (cellsum) {
    const long strips = ncell_pad / KRON_STRIP;
    static const long kgrid = getenv("ASVGP_KRON_PHI_GRID") ? atol(getenv("ASVGP_KRON_PHI_GRID")) : 512;   // resident workgroups: each walks several strips, the next cell's points in flight
    const long total = (long)(kron_noff(order) + 1) * Mtot;
    long gblocks = (total + 255) / 256;
    if (gblocks > 8192) gblocks = 8192;
    KRON_DISPATCH(order, {
      hipLaunchKernelGGL((phi_kron2d_mfma_kernel<K, TIN>), dim3((unsigned)(strips < kgrid ? strips : kgrid)), dim3(64 * KRON_STRIP), 0, st, Xs, ys,
                         reinterpret_cast<const long long*>(cell_start), (int)ncell, (int)ncell_pad, mesh1, 1.0 / delta1, mesh2,
                         (int)n_mesh2, 1.0 / delta2, cellsum, yy);
      hipLaunchKernelGGL(phi_kron2d_gather_kernel<K>, dim3((unsigned)gblocks), dim3(256), 0, st, cellsum, (int)ncell_pad, (int)(n_mesh1 - 1),
                         (int)(n_mesh2 - 1), (int)m1, (int)m2, Ablk, rhs);
    });
    (void)hipFreeAsync(cellsum, st);
    return check_launch("phi_accumulate_kron2d_sorted (matrix-core cell sums + gather)");
  }
  if constexpr (sizeof(TIN) != sizeof(double)) {
    set_error("phi_accumulate_kron2d_sorted_f32: no room for the staging buffer (%ld cells)", ncell);
    return ASVGP_ERR_HIP;
  } else {
  long blocks = ncell < 4096 ? ncell : 4096;
  KRON_DISPATCH(order, {
    hipLaunchKernelGGL(phi_kron2d_cells_kernel<K>, dim3((unsigned)blocks), dim3(256), 0, st, Xs, ys,
                       reinterpret_cast<const long long*>(cell_start), (int)ncell, mesh1, 1.0 / delta1, (int)m1, mesh2,
                       (int)n_mesh2, 1.0 / delta2, (int)m2, Ablk, rhs, yy);
  });
  return check_launch("phi_accumulate_kron2d_sorted");
  }
}

extern "C" int asvgp_phi_accumulate_kron2d_sorted(const double* Xs, const double* ys, int64_t N, const int64_t* cell_start,
                                                  const double* mesh1, int64_t n_mesh1, double delta1, int64_t m1,
                                                  const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                                  double* stats, asvgp_stream_t stream) {
  return phi_accumulate_kron2d_sorted_entry<double>(Xs, ys, N, cell_start, mesh1, n_mesh1, delta1, m1, mesh2, n_mesh2, delta2, m2, order, stats, stream);
}

// fp32 STORAGE of the cell-sorted points (BASELINE config 4): 12 B per point streamed, widened exactly, fp64 arithmetic
extern "C" int asvgp_phi_accumulate_kron2d_sorted_f32(const float* Xs, const float* ys, int64_t N, const int64_t* cell_start,
                                                      const double* mesh1, int64_t n_mesh1, double delta1, int64_t m1,
                                                      const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                                      double* stats, asvgp_stream_t stream) {
  return phi_accumulate_kron2d_sorted_entry<float>(Xs, ys, N, cell_start, mesh1, n_mesh1, delta1, m1, mesh2, n_mesh2, delta2, m2, order, stats, stream);
}
