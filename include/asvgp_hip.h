/* asvgp_hip.h  --  C-ABI of the MI355X-native ASVGP hot path (libasvgp_hip.so, gfx950).
 *
 * Drop-in boundary: these entry points replace, for the asvgp.gpr ELBO/posterior
 * path only, (i) the scipy/TF work of asvgp/basis.py + gpr.py:39-44 (the N-dependent Phi pass) and
 * (ii) the `banded_matrices.banded` TensorFlow custom ops called at gpr.py:56-75 / utils.py:7-9.
 * Paths below are relative to the reference checkout (HJakeCunningham/ASVGP).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`; the caller owns every buffer
 *     (e.g. torch tensors' data_ptr()); the library allocates nothing outside an asvgp_handle_t (asvgp_create / _destroy),
 *     except one 128-byte stream-ordered scratch (hipMallocAsync / hipFreeAsync) inside asvgp_blockband_cholesky.
 *   - state lives in the handle: algorithm choices, the Phi-pass workgroup count, chain-ordering events, the kernel-timing
 *     ring and the prior-chain plan.  Entry points that take a handle are re-entrant ACROSS handles (one handle per model /
 *     stream / host thread); a NULL handle means the process-wide default handle (single-threaded convenience).
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and is
 *     asynchronous; no entry point synchronises the device.
 *   - return value: ASVGP_OK or a negative asvgp_status; asvgp_last_error_string() describes the last failure
 *     on the calling thread.  Numerical failure (band not positive definite) is reported asynchronously
 *     in the device slot `info`: 0 = ok, j+1 = first failing column j.
 *   - band layout = banded_matrices layout: an n x n matrix with lower bandwidth l and upper bandwidth u is a
 *     dense row-major (l+u+1) x n array with band[(u + i - j) * n + j] = A[i][j] (row u = main diagonal,
 *     sub-diagonal d right-padded with d zeros).  A "lower band" has u = 0 and k = l rows below the diagonal.
 *   - fp64 throughout ("f64" is the arithmetic type of the reference path).
 */
#ifndef ASVGP_HIP_H
#define ASVGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* asvgp_stream_t; /* hipStream_t */
typedef struct asvgp_handle_s* asvgp_handle_t;

typedef enum {
  ASVGP_OK = 0,
  ASVGP_ERR_BAD_ARG = -1,      /* null pointer, negative size, inconsistent shapes */
  ASVGP_ERR_UNSUPPORTED = -2,  /* spline order / bandwidth / kernel kind outside the built range */
  ASVGP_ERR_LDS_CAPACITY = -3, /* problem does not fit the 160 KiB LDS even after column chunking */
  ASVGP_ERR_WORKSPACE = -4,    /* workspace pointer null or too small */
  ASVGP_ERR_HIP = -5           /* a HIP runtime call failed (launch error) */
} asvgp_status;

enum { ASVGP_MATERN12 = 0, ASVGP_MATERN32 = 1, ASVGP_MATERN52 = 2 };
enum { ASVGP_MAX_ORDER = 6, ASVGP_MAX_BANDWIDTH = 8, ASVGP_MAX_KUU_TERMS = 9 };

int asvgp_version(void);
const char* asvgp_last_error_string(void);
const char* asvgp_status_name(int status);

/* Handle: bound to the HIP device current at creation.  asvgp_destroy releases the handle's events, its pinned factor-table
 * ring and device tables (the caller makes sure no work of the handle is still in flight). */
int asvgp_create(asvgp_handle_t* handle_out);
int asvgp_destroy(asvgp_handle_t handle);

/* ------------------------------------------------------------------------------------------------
 * Phi pass: B-spline design matrix + sufficient statistics, fused
 * replaces  basis.py:51-76 (SplineBasis.evaluate_basis -> CSR Phi), inducing_features.py:47-48 (make_Kuf),
 *           gpr.py:41-44 (Kuf@y, Kuf@Kuf.T, utils.sparse_to_band utils.py:24-30, sum y^2)
 * x: (N) inputs, y: (N, D) row-major targets, mesh: (n_mesh = M - order + 1) knots exactly as
 * basis.py:17 makes them (including its float32 rounding), delta = mesh[1]-mesh[0] (basis.py:18).
 * stats (output, overwritten): packed [ (order+1)*M lower band of Phi Phi^T | M*D  Phi y | 1  sum y^2 ].
 * This packed buffer is what one RCCL all-reduce(sum) combines across N-shards.
 * ---------------------------------------------------------------------------------------------- */
size_t asvgp_phi_workspace_bytes(int64_t M, int order, int64_t D);
int asvgp_phi_accumulate_1d(asvgp_handle_t handle, const double* x, const double* y, int64_t N, int64_t D,
                            const double* mesh, int64_t n_mesh, double delta, int order, int64_t M,
                            double* stats, void* workspace, size_t workspace_bytes, asvgp_stream_t stream);

/* Phi-pass algorithm of the handle.  0 = auto: 6 where it applies, else 5, else 3.
 *   1 = per-point LDS fp64 atomic scatter of the (order+1)(order+2)/2 products into the band;
 *   3 = as 1 with the products accumulated as 64-bit fixed-point integers (ds_add_u64 runs at twice the ds_add_f64 rate; per-diagonal
 *       power-of-two scales, error per addend <= 2^-43 of the diagonal's largest product, order-independent sums);
 *   5 = per-cell centred power sums S_p = sum s^p (p <= 2 order) in fixed point + direct fixed-point scatter of Phi y: 3 order + 2 LDS
 *       atomics per point; D = 1, 16-byte aligned x / y, image within the LDS (M <= 2048 at order 4);
 *   6 = tile sort: points counting-sorted by cell inside the LDS, the 3 order + 1 moments of a cell accumulated in the REGISTERS of the
 *       thread that owns the cell (no statistic atomics); D = 1, N >= 2, M <= 2048, 16-byte aligned x / y, mesh an exact
 *       numpy.linspace.  Sums follow arrival order: reproducible to rounding, not bit for bit (5 and 3 are).
 * Same statistics to <= 1e-12 of the band's largest entry. */
int asvgp_set_phi_algorithm(asvgp_handle_t handle, int algo);
/* Host-only (tests): Kuu for one theta in closed form, as the matrix-core launch of asvgp_elbo_grad_1d receives it instead of waiting for
 * the assembled band: kuu_diag8[d] = diagonal d on the Toeplitz interior, columns [*lo, *hi); bnd256 = the boundary columns' entries,
 * left part [d * 16 + col] (col < lo), right part [(8 + d) * 16 + (col - hi)] (col >= hi).  *hi <= *lo: no usable interior. */
int asvgp_prior_interior_kuu_host(const double* static_bands_host, int n_terms, int64_t M, int k, const double* coef_host,
                                  double* kuu_diag8, int64_t* lo, int64_t* hi, double* bnd256);
/* Mantissa width of the arithmetic the host forward pass of the planned prior chain runs in: 64 (x87 extended; the build refuses any other). */
int asvgp_host_mantissa_bits(void);
/* The library reads its debug / measurement switches (ASVGP_SPIN_LIMIT, ASVGP_DEBUG_NO_ASSEMBLY, ASVGP_CHAIN_STAMPS, ASVGP_BCR_STAMPS,
 * ASVGP_HOST_TIMES, ASVGP_PLAN_FIRST, ASVGP_NO_SPLIT) from the environment once; call this after changing them in a running process. */
int asvgp_debug_reload_env(void);
/* Deferred forward pass.  With on = 1 the matrix-core launch of asvgp_elbo_grad_1d returns right after the kernel launch; the host's
 * forward pass of the prior chain for that launch (~19 us, long double) runs in asvgp_prior_publish - which the caller issues after
 * enqueueing whatever should not wait behind it (bench.py: the theta-free Phi pass of the next step).  The launch's Kuu workgroup waits
 * (bounded) for the table; the next ELBO call, asvgp_set_deferred_forward_pass(h, 0) and asvgp_destroy publish a forgotten one.
 * on = 2: the pass runs on a worker thread the handle owns, posted BEFORE the launch call - it overlaps the launch path and whatever the
 * caller enqueues next, and the table is published ~20 us after the call was entered (on = 1 in bench.py's order: ~38 us).  The worker
 * spins while jobs keep coming and sleeps on a condition variable after ~2 ms without one (the next post wakes it);
 * asvgp_prior_publish then only waits for it. */
int asvgp_set_deferred_forward_pass(asvgp_handle_t handle, int on);
int asvgp_prior_publish(asvgp_handle_t handle);
/* Result mirror: 16 pinned host doubles owned by the handle.  While enabled, the fused ELBO + gradient launch (band algorithm 0 / 4 where
 * the matrix-core chains apply) also writes [out[0..7], info[0], info[1], sequence, sum of the first ten] there, the sequence number
 * last (after the other stores have been acknowledged; check the sum after seeing it), so a host that has to read every result - an optimiser: the next theta depends on it, the reference's example.py:31-32 -
 * polls 8 bytes of its own memory instead of paying a device-to-host copy and a stream synchronisation per step.
 * asvgp_result_mirror_pending: the sequence number the mirror will show when the handle's LAST ELBO launch has finished, or 0 when
 * that launch does not write the mirror (other algorithms, D > 1, ...): then read `out` / `info` as usual.  A launch that gave up
 * waiting (info[1] < 0, see asvgp_set_band_algorithm) never writes it: bound the poll and fall back to the stream. */
int asvgp_result_mirror(asvgp_handle_t handle, int enable, const double** host_ptr);
uint64_t asvgp_result_mirror_pending(asvgp_handle_t handle);
/* The checksum (double 11) is bound to the launch: sequence number + the ten values, added left to right in fp64.
 * asvgp_result_mirror_read: poll the mirror for `token` (asvgp_result_mirror_pending) from C and copy the ten values out once sequence
 * word and checksum agree; returns ASVGP_OK, or 1 when the token is 0 / nothing valid arrives within timeout_seconds (read through the
 * stream then).  asvgp_elbo_grad_host_1d = asvgp_elbo_grad_1d with the mirror armed + that read: ONE call per optimiser evaluation
 * (example.py:31-32: the next theta depends on this result). */
int asvgp_result_mirror_read(asvgp_handle_t handle, uint64_t token, double* result10, double timeout_seconds);
/* Launch-ahead (matrix-core launch only: k = 4, D = 1, M <= 2048, planned prior chain with the host forward pass; else
 * ASVGP_ERR_UNSUPPORTED and nothing is launched).  An optimiser's evaluations are dependent: theta of the next one comes from this one's
 * result (example.py:31-32).  asvgp_elbo_grad_ahead_1d enqueues the NEXT evaluation's launch before its theta exists - the kernel becomes
 * resident and waits (bounded, ASVGP_SPIN_LIMIT) on the handle's pinned theta box - and asvgp_elbo_publish_theta hands it theta (coefficients,
 * closed-form Kuu tables, the scalars of the bound) and then runs the host forward pass: the launch path and dispatch latency (~8-10 us)
 * are spent while the host still reads the previous result.  Result: `out` / the result mirror (asvgp_result_mirror_pending after the
 * ahead call).  One launch may wait per handle; asvgp_destroy withdraws it; a launch whose theta never came gives up (info[1] < 0). */
int asvgp_elbo_grad_ahead_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, int64_t N, int64_t M, int k,
                             int64_t D, double* out, int* info, void* workspace, size_t workspace_bytes, asvgp_stream_t stream);
int asvgp_elbo_publish_theta(asvgp_handle_t handle, double variance, double lengthscale, double noise_variance);
int asvgp_elbo_grad_host_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                            double lengthscale, double noise_variance, int64_t N, int64_t M, int k, int64_t D,
                            double* out, int* info, void* workspace, size_t workspace_bytes, asvgp_stream_t stream,
                            double* result10, double timeout_seconds);
/* the algorithm the handle's last asvgp_phi_accumulate_1d call actually ran (1, 3, 5 or 6; 0 before the first call) */
int asvgp_phi_last_algorithm(asvgp_handle_t handle);
/* Input order of the tile-sort Phi pass.  A time series (sorted / locally sorted x - the order of the reference's own large 1-D data,
 * experiments/large_regression/electricity.py:31-32) needs no sort: the instantiation with the time-series front loop sums such
 * an input lane-parallel into per-wave run sums and leaves that loop, at a tile boundary, for the general sort loop the first time a
 * 128-point row spans more than two cells.  Both instantiations return the same statistics for ANY input; the order only selects
 * the faster one.  order: 0 = probe once per (x pointer, N): 512 sampled rows, one 4-byte device-to-host copy and ONE stream
 * synchronisation at the first pass over that buffer (default); 1 = unsorted; 2 = time series.
 * asvgp_phi_last_input_order: what the last tile-sort launch on the handle ran as (1 or 2). */
int asvgp_set_phi_input_order(asvgp_handle_t handle, int order);
int asvgp_phi_last_input_order(asvgp_handle_t handle);
/* Measurement aid: a read-only pass over x and y (16 B/point) with the Phi pass's launch shape - the stream ceiling the
 * Phi kernel's achieved bandwidth is reported beside.  sink: device buffer of >= 8 bytes (never written for finite data). */
int asvgp_stream_probe(const double* x, const double* y, int64_t N, double* sink, asvgp_stream_t stream);
/* workgroups of the Phi-pass kernel: 0 = default (256, one per CU); a smaller number leaves CUs free so that a
 * concurrently enqueued asvgp_elbo_prior_chain_1d (second stream) is resident at the same time. */
int asvgp_set_phi_workgroups(asvgp_handle_t handle, int n);
/* Deferred reduce.  asvgp_phi_accumulate_1d is two launches: the streaming kernel (per-workgroup partial statistics into the
 * workspace) and a small cross-workgroup reduce into `stats`.  With on = 1 the accumulate call enqueues only the first (when the
 * default moment kernel applies; otherwise it reduces as usual) and asvgp_phi_reduce_1d enqueues the second on ITS stream argument -
 * the stream that consumes the statistics - after the caller has ordered it behind the accumulate call (an event).  A pipelined
 * caller thereby keeps its N-side stream to the streaming kernels alone.  asvgp_phi_reduce_1d without a pending reduce is a no-op. */
int asvgp_set_phi_deferred_reduce(asvgp_handle_t handle, int on);
int asvgp_phi_reduce_1d(asvgp_handle_t handle, asvgp_stream_t stream);

/* basis.py:58-59  neighbour_index = relu(searchsorted_left(mesh, x) - 1)  (integer work, bit-exact) */
int asvgp_phi_index_1d(const double* x, int64_t N, const double* mesh, int64_t n_mesh, double delta,
                       int64_t* idx, asvgp_stream_t stream);

/* basis.py:62,72-73  the COO triplets of Phi in the reference's concat order:
 * rows[i*N + n] = idx[n] + order - i, data[i*N + n] = piece i of point n (deriv-th x-derivative, 0..3). */
int asvgp_phi_evaluate_1d(const double* x, int64_t N, const double* mesh, int64_t n_mesh, double delta,
                          int order, int deriv, int64_t* rows, double* data, asvgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Kuu assembly   replaces inducing_features.py:12-44 (SplineFeatures1D.make_Kuu)
 * asvgp_matern_coeffs fills (host) the coefficient of each static band and its d/d lengthscale in the
 * fixed term order  A, B, C, D, BC, BC_grad, BC_ggrad, BC_ggrad_none, BC_none_ggrad  restricted to the
 * terms the kernel uses (Matern12: A,B,BC; Matern32: A,B,C,BC,BC_grad; Matern52: all nine).
 * static_bands: (n_terms, k+1, M) device array in that order.  dKuu_dl may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int asvgp_matern_coeffs(int kind, double variance, double lengthscale, double* coef_host, double* dcoef_dl_host,
                        int* n_terms_host);
int asvgp_kuu_assemble(const double* static_bands, int n_terms, const double* coef_host,
                       const double* dcoef_dl_host, int64_t M, int k, double* Kuu, double* dKuu_dl,
                       asvgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * banded_matrices.banded operator replacements (reference call sites in brackets)
 * ---------------------------------------------------------------------------------------------- */
/* cholesky_band(K) [gpr.py:56,73]: lower band (k+1, M) -> lower band of L. */
int asvgp_cholesky_band(const double* K, double* L, int64_t M, int k, int* info, asvgp_stream_t stream);
/* inverse_from_cholesky_band(L) [gpr.py:59]: lower band of (L L^T)^-1 restricted to the band. */
int asvgp_inverse_from_cholesky_band(const double* L, double* S, int64_t M, int k, asvgp_stream_t stream);
/* solve_triang_mat(L, B) [gpr.py:75]: X = L^-1 B (transpose_left=0) or L^-T B (1); B, X: (M, D) row-major. */
int asvgp_solve_triang_mat(const double* L, const double* B, double* X, int64_t M, int k, int64_t D,
                           int transpose_left, asvgp_stream_t stream);
/* product_band_band(left, right, ...) [gpr.py:60-69]: banded x banded cropped to the result band. */
int asvgp_product_band_band(const double* left, const double* right, double* out, int64_t M,
                            int left_lower, int left_upper, int right_lower, int right_upper,
                            int result_lower, int result_upper, asvgp_stream_t stream);
/* transpose_band(B, l, u) [utils.py:8]: (l,u) band of A -> (u,l) band of A^T. */
int asvgp_transpose_band(const double* in, double* out, int64_t M, int lower, int upper, asvgp_stream_t stream);
/* symmetrise_band(B, l) [gpr.py:62; utils.py:7-9]: lower band (l+1, M) -> symmetric band (2l+1, M). */
int asvgp_symmetrise_band(const double* in, double* out, int64_t M, int lower, asvgp_stream_t stream);
/* unpack_banded_matrix_to_dense / pack_dense_matrix_to_banded [utils.py:40-55] (test helpers; the hot path never densifies). */
int asvgp_unpack_banded_matrix_to_dense(const double* band, double* dense, int64_t M, int lower, int upper,
                                        asvgp_stream_t stream);
int asvgp_pack_dense_matrix_to_banded(const double* dense, double* band, int64_t M, int lower, int upper,
                                      asvgp_stream_t stream);
/* fused gpr.py:59-70: out[0] = trace(sym(S) sym(A)) for two lower bands (the only use of product_band_band on the path). */
int asvgp_band_trace_sym(const double* S, const double* A, int64_t M, int k, double* out, asvgp_stream_t stream);

/* Reverse mode (vector-Jacobian products) of the operators above - what banded_matrices registers as the gradients of its TF ops, so
 * that a per-op binding (INTEGRATION.md Level 2) can back-propagate through gpr.py:56-75.  Bands are lower bands (k+1, M); `work` is
 * (k+1) * M doubles of scratch (reciprocal diagonals of the inverse's adjoint; the sequential fallbacks' running adjoint).  The two recurrences run with their state in registers and
 * their inputs staged through the LDS a segment of columns at a time (any M; 0.22 / 0.29 ms at M = 2048, k = 4; the inverse's adjoint
 * for k <= 6, beyond that a wave-parallel form while 2 (k+1) M doubles fit the LDS, else a sequential sweep).  The training path of this
 * library is the fused asvgp_elbo_grad_1d.
 *   cholesky_band_vjp:              Kbar = d<Lbar, cholesky_band(K)> / dK          (over the stored lower-band entries)
 *   inverse_from_cholesky_band_vjp: Lbar = d<Sbar, inverse_from_cholesky_band(L)> / dL   (S = the forward result)
 *   solve_triang_mat:  X = L^-1 B:  Bbar = asvgp_solve_triang_mat(L, Xbar, transpose_left = 1),  Lbar = asvgp_band_outer_product(Bbar, X, sign = -1)
 *                      X = L^-T B:  Bbar = asvgp_solve_triang_mat(L, Xbar, transpose_left = 0),  Lbar = asvgp_band_outer_product(X, Bbar, sign = -1)
 *   product_band_band: Leftbar = product_band_band(Obar, transpose(Right)) cropped to Left's band, Rightbar = product_band_band(
 *                      transpose(Left), Obar) cropped to Right's band (asvgp_transpose_band + asvgp_product_band_band).
 * asvgp_band_outer_product: out[d, j] = sign * sum_c U[j + d, c] V[j, c], U, V dense (M, D) row-major. */
int asvgp_cholesky_band_vjp(const double* L, const double* Lbar, double* Kbar, double* work, int64_t M, int k, asvgp_stream_t stream);
int asvgp_inverse_from_cholesky_band_vjp(const double* L, const double* S, const double* Sbar, double* Lbar, double* work, int64_t M,
                                         int k, asvgp_stream_t stream);
int asvgp_band_outer_product(const double* U, const double* V, int64_t M, int64_t D, int k, double sign, double* out, asvgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused ELBO + gradient   replaces GPR_1d.elbo gpr.py:49-89 and the TF reverse-mode pass through the
 * banded_matrices op gradients that opt.minimize(training_loss) triggers (example.py:31-32).
 * stats: packed buffer of asvgp_phi_accumulate_1d (after the cross-rank sum); N = GLOBAL number of rows.
 * out (device, 8 doubles): [elbo, d/d variance, d/d lengthscale, d/d noise variance, log|Kuu|, log|P|,
 *                           trace(Kuu^-1 PhiPhi^T), |c|^2].
 * ---------------------------------------------------------------------------------------------- */
size_t asvgp_elbo_workspace_bytes(int64_t M, int k, int64_t D);   /* the workspace must be zero-filled ONCE by the caller
                                                                     (cross-workgroup arrival slots are re-armed by every call) */
/* band algorithm of the fused drivers below: 0 = auto, 1 = sequential single-wave sweeps (the reference's elimination order),
 * 2 = block cyclic reduction (O(log M) dependent levels), both chains on the GPU, 3 = block cyclic reduction with the PLANNED
 * prior chain (asvgp_prior_plan_1d): forward pass of the Kuu chain on the host in long double over the O(log M) distinct
 * nodes, backward (selected inverse) pass on the GPU, 4 = the planned chains with every 4 x 4 block product on the matrix cores
 * (v_mfma_f64_4x4x4f64; k = 4, D = 1, M <= 2048, asvgp_elbo_grad_1d).  Auto = 4 where it applies, else 3 when the handle holds a
 * matching plan, else 2 when both chains fit the 160 KiB LDS and D == 1, else 1.  2 / 3 return ASVGP_ERR_LDS_CAPACITY when the chains
 * do not fit.  info[1] < 0 after a fused launch: the launch gave up waiting for its helper workgroups (they never became resident);
 * its results are to be discarded, the workspace zero-filled again and the step re-issued (asvgp_amd.GPR_1d does that through
 * algorithm 1). */
int asvgp_set_band_algorithm(asvgp_handle_t handle, int algo);
/* Plan of the prior chain for one (basis, kernel kind).  static_bands_host: HOST copy of the (n_terms, k+1, M) array the ELBO
 * entry points receive on the device (inducing_features.py:16-44 order).  Kuu = sum_t c_t(theta) S_t is Toeplitz away from the
 * boundaries (basis.py:31-45), so every level of the odd-even elimination has only a handful of DISTINCT nodes; the plan
 * classifies them once, and each later ELBO call eliminates one representative per class on the host in long double
 * (~20 us at M = 2048) - the forward pass is where fp64 cyclic reduction loses up to three digits against the reference's
 * sequential Cholesky (gpr.py:56-59) when cond(Kuu) is large.  *planned = 0 (and ASVGP_OK) when the bands have no such
 * structure; static_bands_host = NULL drops the plan.  The handle owns the plan, a device node->record map and a pinned
 * factor-table ring. */
int asvgp_prior_plan_1d(asvgp_handle_t handle, const double* static_bands_host, int n_terms, int64_t M, int k, int* planned);
/* The planner alone, host memory only (no device): factor table for one theta - header [log|Kuu|, d log|Kuu|/dl, first failing
 * column + 1, n_rec], then the value plane and the tangent plane of n_rec records [L (k x k) | 1/diag L (k) | U_a | U_b] - and
 * the node -> record map of the ceil(M / k) block nodes.  coef / dcoef_dl as filled by asvgp_matern_coeffs. */
size_t asvgp_prior_table_doubles(const double* static_bands_host, int n_terms, int64_t M, int k);
int asvgp_prior_forward_host(const double* static_bands_host, int n_terms, int64_t M, int k, const double* coef_host,
                             const double* dcoef_dl_host, double* table_host, size_t table_doubles, int* node_rec_host);
/* Where the forward (elimination) half of the planned Kuu chain runs.  mode 0 (default): on the host in x87 long double, handed over
 * through the pinned table ring.  mode 1: on the GPU in double-double arithmetic (two-fp64 error-free transforms, ~104 bits; one
 * small launch - on the handle's own stream beside the matrix-core chains' launch, whose Kuu workgroup waits for it, else in front of
 * the consumer on the same stream -, table in device memory) - no host stage, no PCIe-mapped table, no
 * dependence on the host's long double format.  Replaces the factorisation half of gpr.py:56-59 (banded.cholesky_band(Kuu)) either
 * way; same table (asvgp_prior_forward_host), same consumers.  Needs a plan (asvgp_prior_plan_1d) when an ELBO entry point runs. */
int asvgp_set_prior_forward(asvgp_handle_t handle, int mode);
/* The GPU forward pass alone (tests, parity with asvgp_prior_forward_host): the table of the handle's plan for one theta, computed on
 * `stream` and copied to table_host (asvgp_prior_table_doubles entries); synchronises the stream. */
int asvgp_prior_forward_device(asvgp_handle_t handle, const double* coef_host, const double* dcoef_dl_host, double* table_host,
                               size_t table_doubles, void* stream);
/* Host-only: the plan's device image (class maps as ints, level-0 static-band entries as doubles: csrc/prior_plan.cpp) - what the GPU
 * forward pass walks.  ints / doubles NULL: sizes only.  For the CPU tests. */
int asvgp_prior_plan_image_host(const double* static_bands_host, int n_terms, int64_t M, int k, int* ints, size_t* n_ints,
                                double* doubles, size_t* n_doubles);
/* Diagnostics: s_memtime stamps of thread 0 along one GPU forward pass (out64: 64 values, [63] = how many; tools/prior_dd_probe.py). */
int asvgp_prior_forward_stamps(asvgp_handle_t handle, const double* coef_host, const double* dcoef_dl_host, uint64_t* out64, void* stream);
int asvgp_elbo_grad_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                       double lengthscale, double noise_variance, int64_t N, int64_t M, int k, int64_t D,
                       double* out, int* info, void* workspace, size_t workspace_bytes, asvgp_stream_t stream);

/* band(Kuu^-1) and its d/d-lengthscale tangent as an operator of its own: Kuu / dKuu_dl (assembled, inducing_features.py:12-44), S =
 * inverse_from_cholesky_band(cholesky_band(Kuu)) (gpr.py:56-59) and dS / dl, logdet2 (device) = [log|Kuu|, d log|Kuu| / dl].
 * What the Kronecker model needs of each 1-D factor (gpr.py:286-307); uses the handle's prior plan when there is one (host
 * forward pass in long double), else the all-GPU chains.  workspace: asvgp_elbo_workspace_bytes(M, k, 1). */
int asvgp_kuu_inverse_band_1d(asvgp_handle_t handle, const double* static_bands, int kind, double variance, double lengthscale,
                              int64_t M, int k, double* Kuu, double* dKuu_dl, double* S, double* dS_dl, double* logdet2, int* info,
                              void* workspace, size_t workspace_bytes, asvgp_stream_t stream);

/* The same computation split for scheduling: the PRIOR chain touches only theta (Kuu, dKuu/dl, band(Kuu^-1) and its
 * tangent, log|Kuu|) and may be enqueued on another stream concurrently with the Phi pass; the DATA chain
 * (P = Kuu + A/s factor/solve/inverse + the finalize) needs `stats` and must be ordered after the prior chain of the same
 * theta and workspace (stream order or an event).  elbo_prior_chain + elbo_data_chain == elbo_grad_1d. */
/* asvgp_elbo_chain_sync(h, 1): the handle orders the two calls itself with its own events - the prior chain records
 * "Kuu assembled" and "prior chain complete" on its stream, the data chain waits for the first before factorising P and
 * for the second before the finalize - so the caller needs no event of its own and the P chain overlaps the rest of the
 * prior chain.  One prior/data pair in flight per handle.  With the planned prior chain (band algorithm 3) the prior call
 * is a no-op and the data call runs both chains in one launch. */
int asvgp_elbo_chain_sync(asvgp_handle_t handle, int enable);
int asvgp_elbo_prior_chain_1d(asvgp_handle_t handle, const double* static_bands, int kind, double variance, double lengthscale,
                              double noise_variance, int64_t M, int k, int64_t D, int* info, void* workspace,
                              size_t workspace_bytes, asvgp_stream_t stream);
int asvgp_elbo_data_chain_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                             double lengthscale, double noise_variance, int64_t N, int64_t M, int k, int64_t D,
                             double* out, int* info, void* workspace, size_t workspace_bytes, asvgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Posterior   replaces GPR_1d.predict_f gpr.py:91-120 (CHOLMOD natural-ordering solves)
 * prepare: alpha = P^-1 Phi y / s  (M, D)  and  W = band(P^-1) - band(Kuu^-1)  (k+1, M), once per theta.
 * predict: per test point mean = phi*^T alpha, var = variance + phi*^T W phi*  (full_cov=False only).
 * ---------------------------------------------------------------------------------------------- */
int asvgp_posterior_prepare_1d(asvgp_handle_t handle, const double* stats, const double* static_bands, int kind, double variance,
                               double lengthscale, double noise_variance, int64_t M, int k, int64_t D,
                               double* alpha, double* W, int* info, void* workspace, size_t workspace_bytes,
                               asvgp_stream_t stream);
int asvgp_predict_1d(const double* xnew, int64_t n, const double* mesh, int64_t n_mesh, double delta, int order,
                     int64_t M, const double* alpha, const double* W, double variance, int64_t D, double* mean,
                     double* var, asvgp_stream_t stream);
/* The same on a handle (NULL: the process default).  The handle remembers whether `mesh` is an exact numpy.linspace (one device-to-host
 * copy the first time a mesh pointer is seen); if it is, D == 1 and n >= 262 144, the batch takes the cell-polynomial kernel: the variance
 * inside a cell is a polynomial of degree 2k in the local coordinate, its 2k+1 coefficients are built per workgroup into the LDS and a
 * point reads them contiguously (8 LDS instructions instead of 23 scattered reads).  Means agree with asvgp_predict_1d to an ulp (the same sum
 * from the same t), variances to rounding (<= 1e-12 of the prior variance). */
int asvgp_predict_1d_h(asvgp_handle_t handle, const double* xnew, int64_t n, const double* mesh, int64_t n_mesh, double delta, int order,
                       int64_t M, const double* alpha, const double* W, double variance, int64_t D, double* mean,
                       double* var, asvgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 2-D Kronecker (tensor-product) path   replaces kronecker.make_kvs_sparse kronecker.py:7-33 and the dense
 * linear algebra of GPR_kron gpr.py:239-359 (KufKfu.todense(), tf.linalg.cholesky / triangular_solve / cholesky_solve).
 * Basis pair (i1, i2) has row index i1*m2 + i2 (dim-0 major, as make_kvs_two_sparse).  Both bases share `order` = k
 * (gpr.py:261 takes bases[0].order).  X: (N, 2) row-major, 16-byte aligned.
 * kron stats (output of the Phi pass; the all-reduce payload): [ n_off*M_tot block band | M_tot rhs | 1 yy ],
 * M_tot = m1*m2, n_off = k(2k+1)+k+1 lower offsets: (d1=0, d2=0..k) then (d1=1..k, d2=-k..k);
 * entry  A[(i1+d1)*m2 + i2+d2, i1*m2 + i2]  at  stats[off*M_tot + i1*m2 + i2].
 * Wide-band storage for P = K1 (x) K2 + A/s: column-major lower band Pb[col*(bw+1) + (row-col)], bw = k*m2 + k.
 * ---------------------------------------------------------------------------------------------- */
size_t asvgp_kron_stats_doubles(int64_t m1, int64_t m2, int k);
int asvgp_phi_accumulate_kron2d(const double* X, const double* y, int64_t N, const double* mesh1, int64_t n_mesh1,
                                double delta1, int64_t m1, const double* mesh2, int64_t n_mesh2, double delta2,
                                int64_t m2, int order, double* stats, asvgp_stream_t stream);
/* Khatri-Rao COO triplets: rows[e*N + n], data[e*N + n], e = a*(k+1)+b <-> basis pair (idx1+k-a, idx2+k-b) of point n */
int asvgp_kron_evaluate_2d(const double* X, int64_t N, const double* mesh1, int64_t n_mesh1, double delta1,
                           const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order, int64_t* rows,
                           double* data, asvgp_stream_t stream);
/* Pb (may be NULL) = K1 (x) K2 + Ablk/s in wide-band storage; trace_out (may be NULL) = tr((K1 (x) K2)^-1 A) from the
 * 1-D inverse bands S1, S2 (gpr.py:307 trace(cholesky_solve(L_Kuu, KufKfu_dense))). */
int asvgp_kron_assemble(const double* K1, const double* K2, const double* S1, const double* S2, const double* Ablk, int k,
                        int64_t m1, int64_t m2, double noise_variance, double* Pb, double* trace_out,
                        asvgp_stream_t stream);
/* in-place blocked band Cholesky (gpr.py:293 tf.linalg.cholesky(P)); rhs (may be NULL, length M) is overwritten with
 * L^-1 rhs (gpr.py:295 triangular_solve); logdet (may be NULL) = 2 sum log diag L; info = first bad column + 1.
 * bw <= 432: ONE persistent launch (16 workgroups, left-looking dataflow over block columns of 32 through an arrival counter in a
 * 128-byte stream-ordered allocation; fp64 MFMA updates; bounded spins: info = -1 if a block column never arrives); wider bands:
 * one panel + one update launch per block column. */
int asvgp_blockband_cholesky(double* Pb, int64_t M, int64_t bw, double* rhs, double* logdet, int* info,
                             asvgp_stream_t stream);
/* x <- L^-T x */
int asvgp_blockband_backsolve(const double* Lb, int64_t M, int64_t bw, double* x, asvgp_stream_t stream);
/* posterior mean phi*^T alpha and (qk may be NULL) phi*^T Kuu^-1 phi* = (phi1^T S1 phi1)(phi2^T S2 phi2) per test point */
int asvgp_predict_kron2d(const double* Xnew, int64_t n, const double* mesh1, int64_t n_mesh1, double delta1, int64_t m1,
                         const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order, const double* alpha,
                         const double* S1, const double* S2, double* mean, double* qk, asvgp_stream_t stream);

/* The same statistics from CELL-SORTED points: Xs, ys are the rows of X, y permuted so that the 2-D cell id
 * c = i1 * (n_mesh2 - 1) + i2 (asvgp_kron_cell_index; basis.py:58-59 per dimension) is non-decreasing, cell_start[c] (int64,
 * n_cells + 1 entries) the first row of cell c.  A cell's statistics are a small Gram matrix: one wavefront per cell forms it on
 * the fp64 matrix core (v_mfma_f64_16x16x4 over the (k+1)^2 basis functions, four points per step), the per-cell results go
 * entry-major into a stream-ordered staging buffer (152 doubles per cell at k = 3, 350 at k = 4; hipMallocAsync on `stream`)
 * and a gather kernel forms every output from the <= (k+1)^2 cells that touch it: no statistic atomics, deterministic sums.
 * (ASVGP_KRON_PHI_ATOMICS=1 in the environment, or no room for the staging buffer: the per-cell kernel with one global atomic per
 * block-band entry and cell.) */
int asvgp_kron_cell_index(const double* X, int64_t N, const double* mesh1, int64_t n_mesh1, double delta1,
                          const double* mesh2, int64_t n_mesh2, double delta2, int* cell, asvgp_stream_t stream);
int asvgp_phi_accumulate_kron2d_sorted(const double* Xs, const double* ys, int64_t N, const int64_t* cell_start,
                                       const double* mesh1, int64_t n_mesh1, double delta1, int64_t m1,
                                       const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                       double* stats, asvgp_stream_t stream);
/* The same pass over points STORED in fp32 (BASELINE.json configs[3] names fp32 data): Xs (N, 2) and ys (N) are float arrays in cell
 * order, 12 bytes per point streamed; every value is widened exactly in registers and all arithmetic is the fp64 arithmetic of the entry
 * above - block band and rhs are bit for bit those of the upcast data streamed in the same order (the reference itself casts to fp64,
 * basis.py:54). */
int asvgp_phi_accumulate_kron2d_sorted_f32(const float* Xs, const float* ys, int64_t N, const int64_t* cell_start,
                                           const double* mesh1, int64_t n_mesh1, double delta1, int64_t m1,
                                           const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                                           double* stats, asvgp_stream_t stream);

/* Selected inverse of P = Kuu + KufKfu/sigma2 on the band (what the gradient of gpr.py:282-308 and the predictive variance
 * of gpr.py:319-330 need of P^-1), through dense super-blocks of size Bb (a multiple of 32, >= bw): the band factor is
 * block bidiagonal.  asvgp_blockband_to_blocks unpacks it into diag[nblk][Bb][Bb] (lower triangular, identity-padded) and
 * sub[nblk-1][Bb][Bb] (L[(i+1)Bb + r, i Bb + c]); the caller runs the block recursion
 *   G_i = sub_i diag_i^-1,  SigS_i = -SigD_{i+1} G_i,  SigD_i = (diag_i diag_i^T)^-1 - G_i^T SigS_i
 * with library GEMM / TRSM calls and hands SigD / SigS (same shapes) to the two kernels below. */
int asvgp_blockband_to_blocks(const double* Lb, int64_t M, int64_t bw, int64_t Bb, double* diag, double* sub,
                              asvgp_stream_t stream);
/* out11 (overwritten) = [tr(Sig A), a^T A a, tr(Sig X1), a^T X1 a, tr(Sig X2), a^T X2 a, tr(Sig Kuu), a^T Kuu a,
 * tr((Z1 (x) S2) A), tr((S1 (x) Z2) A), tr((S1 (x) S2) A)] with X1 = dK1 (x) K2, X2 = K1 (x) dK2, a = alpha; every band
 * argument is a 1-D lower band (k+1, m_i); Ablk as written by asvgp_phi_accumulate_kron2d. */
int asvgp_kron_grad_terms(const double* SigD, const double* SigS, int64_t Bb, const double* alpha, const double* Ablk,
                          const double* K1, const double* K2, const double* dK1, const double* dK2, const double* S1,
                          const double* S2, const double* Z1, const double* Z2, int k, int64_t m1, int64_t m2,
                          double* out11, asvgp_stream_t stream);
/* qp[i] = phi*(x_i)^T P^-1 phi*(x_i): (k+1)^4 reads of the block batches per test point instead of a triangular solve */
int asvgp_predict_kron2d_var(const double* Xnew, int64_t n, const double* mesh1, int64_t n_mesh1, double delta1,
                             const double* mesh2, int64_t n_mesh2, double delta2, int64_t m2, int order,
                             const double* SigD, const double* SigS, int64_t Bb, double* qp, asvgp_stream_t stream);
/* Two-sided ("twisted") factorisation of P: the chain of the band Cholesky is sequential in its M/32 block columns, so P is split at a
 * separator [h, h + Bb) one super-block wide (Bb >= bw) into a TOP system (padt identity columns, then original columns 0 .. h+Bb-1)
 * and a BOTTOM system in REVERSED order (padb identity columns, then original columns M-1 .. h), both nb Bb columns long with the
 * separator as their last super-block.  asvgp_kron_assemble_twisted writes the two bands (Pt, Pr: (nb Bb)(bw+1) doubles each, zeroed
 * here), asvgp_blockband_cholesky factors them independently (concurrently on two streams: half the chain each), and the separator's
 * Schur complement  S = L_ss L_ss^T + J L'_ss L'_ss^T J - P_ss  (J: reversal) is factored last.  top_end = h + Bb,
 * padt = nb Bb - top_end, padb = nb Bb - (M - h).  The selected inverse then runs outwards from S^-1 in both systems at once
 * (SigD [2][nb][Bb][Bb], SigS [2][nb-1][Bb][Bb]: stack 0 = top, stack 1 = bottom, reversed) and the _twisted forms of the two
 * consumers read that layout. */
int asvgp_kron_assemble_twisted(const double* K1, const double* K2, const double* S1, const double* S2, const double* Ablk, int k,
                                int64_t m1, int64_t m2, double noise_variance, int64_t Bb, int64_t nb, int64_t top_end, int64_t padt,
                                int64_t padb, double* Pt, double* Pr, double* trace_out, asvgp_stream_t stream);
int asvgp_kron_grad_terms_twisted(const double* SigD, const double* SigS, int64_t Bb, int64_t nb, int64_t top_end, int64_t padt,
                                  int64_t padb, const double* alpha, const double* Ablk, const double* K1, const double* K2,
                                  const double* dK1, const double* dK2, const double* S1, const double* S2, const double* Z1,
                                  const double* Z2, int k, int64_t m1, int64_t m2, double* out11, asvgp_stream_t stream);
int asvgp_predict_kron2d_var_twisted(const double* Xnew, int64_t n, const double* mesh1, int64_t n_mesh1, double delta1,
                                     const double* mesh2, int64_t n_mesh2, double delta2, int64_t m1, int64_t m2, int order,
                                     const double* SigD, const double* SigS, int64_t Bb, int64_t nb, int64_t top_end, int64_t padt,
                                     int64_t padb, double* qp, asvgp_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * Additive model (GPR_additive, gpr.py:139-236): Kuf = vstack(Kuf_1 .. Kuf_d), so Kuf Kuf^T (gpr.py:170-171) has the
 * 1-D banded blocks of asvgp_phi_accumulate_1d on its diagonal and dense cross blocks
 *   out[r, c] = sum_n phi_i(x_i[n])[r] * phi_j(x_j[n])[c]        (m_i x m_j, row-major, overwritten)
 * off it.  x_i, x_j: contiguous columns of X (N).  workspace: asvgp_phi_cross_workspace_bytes(m_i, m_j) bytes (per-
 * workgroup partial images; unused when the block exceeds the LDS and is accumulated with L2 atomics instead).
 * ---------------------------------------------------------------------------------------------- */
size_t asvgp_phi_cross_workspace_bytes(int64_t m_i, int64_t m_j);
int asvgp_phi_cross_2d(const double* x_i, const double* x_j, int64_t N, const double* mesh_i, int64_t n_mesh_i,
                       double delta_i, int64_t m_i, const double* mesh_j, int64_t n_mesh_j, double delta_j, int64_t m_j,
                       int order, double* out, void* workspace, size_t workspace_bytes, asvgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): when enabled, HIP events are recorded on the launch stream immediately around
 * every Phi-pass kernel launch (up to 1024 launches); asvgp_profile_read synchronises on them and returns the
 * summed kernel time in milliseconds and the number of launches, then resets the ring.  Host-side, not stream-ordered.
 * ---------------------------------------------------------------------------------------------- */
int asvgp_profile_enable(asvgp_handle_t handle, int on);   /* 0 off, 1 every launch, n > 1 every n-th launch (events perturb the stream) */
int asvgp_profile_read(asvgp_handle_t handle, double* phi_kernel_ms_sum_host, int64_t* launches_host);

#ifdef __cplusplus
}
#endif
#endif /* ASVGP_HIP_H */
