"""Drop-in for the subset of `banded_matrices.banded` the reference calls (gpr.py:56-75, utils.py:7-57),
on ROCm torch fp64 tensors, backed by the HIP C-ABI (include/asvgp_hip.h).  Same function names and keyword
names as the TF ops; band layout band[u+i-j, j] = A[i, j]."""
import torch

from . import _lib
from ._lib import check, f64c, get_lib, require_cuda, stream_ptr


class NotPositiveDefiniteError(_lib.AsvgpError):
    """cholesky_band met a non-positive pivot (the TF op raises InvalidArgument here)."""


def _ptr(t):
    return t.data_ptr()


def _cholesky_band_raw(band, check_pd=True):
    """banded.cholesky_band(K)  [gpr.py:56,73]: (k+1, M) lower band -> lower band of the Cholesky factor."""
    band = f64c(band)
    require_cuda(band)
    k, M = band.shape[0] - 1, band.shape[1]
    out = torch.empty_like(band)
    info = torch.zeros(1, dtype=torch.int32, device=band.device)
    check(get_lib().asvgp_cholesky_band(_ptr(band), _ptr(out), M, k, _ptr(info), stream_ptr()), "cholesky_band")
    if check_pd:
        col = int(info.item())
        if col != 0:
            raise NotPositiveDefiniteError("cholesky_band: band not positive definite at column %d" % (col - 1))
    return out


def _inverse_from_cholesky_band_raw(L):
    """banded.inverse_from_cholesky_band(L)  [gpr.py:59]: lower band of (L L^T)^-1 restricted to the band."""
    L = f64c(L)
    require_cuda(L)
    k, M = L.shape[0] - 1, L.shape[1]
    out = torch.empty_like(L)
    check(get_lib().asvgp_inverse_from_cholesky_band(_ptr(L), _ptr(out), M, k, stream_ptr()),
          "inverse_from_cholesky_band")
    return out


def _solve_triang_mat_raw(left, right, transpose_left=False):
    """banded.solve_triang_mat(L, B)  [gpr.py:75]: L^-1 B (or L^-T B) for a lower band L and dense B (M, D)."""
    left, right = f64c(left), f64c(right)
    require_cuda(left, right)
    k, M = left.shape[0] - 1, left.shape[1]
    squeeze = right.dim() == 1
    B = right.reshape(M, -1)
    out = torch.empty_like(B)
    check(get_lib().asvgp_solve_triang_mat(_ptr(left), _ptr(B), _ptr(out), M, k, B.shape[1], int(bool(transpose_left)),
                                           stream_ptr()), "solve_triang_mat")
    return out.reshape(-1) if squeeze else out


def _product_band_band_raw(left, right, left_lower_bandwidth, left_upper_bandwidth, right_lower_bandwidth,
                           right_upper_bandwidth, result_lower_bandwidth, result_upper_bandwidth):
    """banded.product_band_band(...)  [gpr.py:60-69] with the TF op's keyword names."""
    left, right = f64c(left), f64c(right)
    require_cuda(left, right)
    M = left.shape[1]
    assert left.shape[0] == left_lower_bandwidth + left_upper_bandwidth + 1
    assert right.shape == (right_lower_bandwidth + right_upper_bandwidth + 1, M)
    out = torch.empty((result_lower_bandwidth + result_upper_bandwidth + 1, M), dtype=torch.float64, device=left.device)
    check(get_lib().asvgp_product_band_band(_ptr(left), _ptr(right), _ptr(out), M, int(left_lower_bandwidth),
                                            int(left_upper_bandwidth), int(right_lower_bandwidth),
                                            int(right_upper_bandwidth), int(result_lower_bandwidth),
                                            int(result_upper_bandwidth), stream_ptr()), "product_band_band")
    return out


def transpose_band(band, lower_bandwidth, upper_bandwidth):
    """banded.transpose_band(B, l, u)  [utils.py:8]."""
    band = f64c(band)
    require_cuda(band)
    assert band.shape[0] == lower_bandwidth + upper_bandwidth + 1
    out = torch.empty_like(band)
    check(get_lib().asvgp_transpose_band(_ptr(band), _ptr(out), band.shape[1], int(lower_bandwidth),
                                         int(upper_bandwidth), stream_ptr()), "transpose_band")
    return out


def symmetrise_band(band, lower_bandwidth):
    """banded.symmetrise_band(B, l)  [gpr.py:62]: lower band (l+1, M) -> symmetric (2l+1, M)."""
    l = int(lower_bandwidth)
    if _needs_grad(band):        # a pure index shuffle: differentiable through torch (row l - d holds A[c - d, c] = lower[d, c - d])
        M = band.shape[1]
        upper = [torch.cat([band.new_zeros(d), band[d, :M - d]]) for d in range(l, 0, -1)]
        return torch.cat([torch.stack(upper), band]) if l > 0 else band
    band = f64c(band)
    require_cuda(band)
    assert band.shape[0] == l + 1
    out = torch.empty((2 * l + 1, band.shape[1]), dtype=torch.float64, device=band.device)
    check(get_lib().asvgp_symmetrise_band(_ptr(band), _ptr(out), band.shape[1], l, stream_ptr()), "symmetrise_band")
    return out


def unpack_banded_matrix_to_dense(band, lower_bandwidth, upper_bandwidth):
    """banded.unpack_banded_matrix_to_dense  [utils.py:40,47,55] - test helper, never on the hot path."""
    band = f64c(band)
    require_cuda(band)
    M = band.shape[1]
    out = torch.empty((M, M), dtype=torch.float64, device=band.device)
    check(get_lib().asvgp_unpack_banded_matrix_to_dense(_ptr(band), _ptr(out), M, int(lower_bandwidth),
                                                        int(upper_bandwidth), stream_ptr()), "unpack_banded")
    return out


def pack_dense_matrix_to_banded(dense, lower_bandwidth, upper_bandwidth):
    """banded.pack_dense_matrix_to_banded  [utils.py:43]."""
    dense = f64c(dense)
    require_cuda(dense)
    M = dense.shape[0]
    out = torch.empty((lower_bandwidth + upper_bandwidth + 1, M), dtype=torch.float64, device=dense.device)
    check(get_lib().asvgp_pack_dense_matrix_to_banded(_ptr(dense), _ptr(out), M, int(lower_bandwidth),
                                                      int(upper_bandwidth), stream_ptr()), "pack_dense")
    return out


def band_trace_sym(S_lower, A_lower):
    """trace(sym(S) sym(A)) for two lower bands: the fused form of gpr.py:59-70 (symmetrise x2 + product + sum)."""
    S_lower, A_lower = f64c(S_lower), f64c(A_lower)
    require_cuda(S_lower, A_lower)
    out = torch.empty(1, dtype=torch.float64, device=S_lower.device)
    check(get_lib().asvgp_band_trace_sym(_ptr(S_lower), _ptr(A_lower), S_lower.shape[1], S_lower.shape[0] - 1, _ptr(out),
                                         stream_ptr()), "band_trace_sym")
    return out[0]


# ---------------------------------------------------------------------------------------------------------------------------
# Differentiable front ends.  banded_matrices registers gradients for these four TF ops and the reference's optimiser differentiates
# GPR_1d.elbo through them (example.py:31-32); here the same functions are torch.autograd Functions over the C-ABI VJP entry points
# (asvgp_cholesky_band_vjp, asvgp_inverse_from_cholesky_band_vjp, asvgp_band_outer_product + the forward operators), so a bound
# written with these ops on tensors that require grad back-propagates.  Without grad-requiring inputs they are the plain launches.
# ---------------------------------------------------------------------------------------------------------------------------
def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


class _CholeskyBand(torch.autograd.Function):
    @staticmethod
    def forward(ctx, band, check_pd):
        L = _cholesky_band_raw(band.detach(), check_pd)
        ctx.save_for_backward(L)
        return L

    @staticmethod
    def backward(ctx, Lbar):
        (L,) = ctx.saved_tensors
        Lbar = f64c(Lbar)
        k, M = L.shape[0] - 1, L.shape[1]
        Kbar, work = torch.empty_like(L), torch.empty_like(L)
        check(get_lib().asvgp_cholesky_band_vjp(_ptr(L), _ptr(Lbar), _ptr(Kbar), _ptr(work), M, k, stream_ptr()), "cholesky_band_vjp")
        return Kbar, None


class _InverseFromCholeskyBand(torch.autograd.Function):
    @staticmethod
    def forward(ctx, L):
        Ld = f64c(L.detach())
        S = _inverse_from_cholesky_band_raw(Ld)
        ctx.save_for_backward(Ld, S)
        return S

    @staticmethod
    def backward(ctx, Sbar):
        L, S = ctx.saved_tensors
        Sbar = f64c(Sbar)
        k, M = L.shape[0] - 1, L.shape[1]
        Lbar, work = torch.empty_like(L), torch.empty_like(L)
        check(get_lib().asvgp_inverse_from_cholesky_band_vjp(_ptr(L), _ptr(S), _ptr(Sbar), _ptr(Lbar), _ptr(work), M, k, stream_ptr()),
              "inverse_from_cholesky_band_vjp")
        return Lbar


class _SolveTriangMat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, left, right, transpose_left):
        Ld, Bd = f64c(left.detach()), f64c(right.detach())
        X = _solve_triang_mat_raw(Ld, Bd, transpose_left)
        ctx.save_for_backward(Ld, X)
        ctx.transpose_left = bool(transpose_left)
        ctx.right_shape = right.shape
        return X

    @staticmethod
    def backward(ctx, Xbar):
        L, X = ctx.saved_tensors
        k, M = L.shape[0] - 1, L.shape[1]
        Xb = f64c(Xbar).reshape(M, -1)
        X2 = X.reshape(M, -1).contiguous()
        Bbar = _solve_triang_mat_raw(L, Xb, not ctx.transpose_left).reshape(M, -1).contiguous()
        Lbar = torch.empty_like(L)
        U, V = (Bbar, X2) if not ctx.transpose_left else (X2, Bbar)
        check(get_lib().asvgp_band_outer_product(_ptr(U), _ptr(V), M, X2.shape[1], k, -1.0, _ptr(Lbar), stream_ptr()), "band_outer_product")
        return Lbar, Bbar.reshape(ctx.right_shape), None


class _ProductBandBand(torch.autograd.Function):
    @staticmethod
    def forward(ctx, left, right, ll, lu, rl, ru, ol, ou):
        Ld, Rd = f64c(left.detach()), f64c(right.detach())
        ctx.save_for_backward(Ld, Rd)
        ctx.bw = (ll, lu, rl, ru, ol, ou)
        return _product_band_band_raw(Ld, Rd, ll, lu, rl, ru, ol, ou)

    @staticmethod
    def backward(ctx, Obar):
        left, right = ctx.saved_tensors
        ll, lu, rl, ru, ol, ou = ctx.bw
        Obar = f64c(Obar)
        Rt = transpose_band(right, rl, ru)                       # (lower ru, upper rl)
        Lt = transpose_band(left, ll, lu)                        # (lower lu, upper ll)
        Lbar = _product_band_band_raw(Obar, Rt, ol, ou, ru, rl, ll, lu)
        Rbar = _product_band_band_raw(Lt, Obar, lu, ll, ol, ou, rl, ru)
        return Lbar, Rbar, None, None, None, None, None, None


def cholesky_band(band, check_pd=True):
    """banded.cholesky_band(K)  [gpr.py:56,73]: (k+1, M) lower band -> lower band of the Cholesky factor (differentiable)."""
    if _needs_grad(band):
        return _CholeskyBand.apply(band, check_pd)
    return _cholesky_band_raw(band, check_pd)


def inverse_from_cholesky_band(L):
    """banded.inverse_from_cholesky_band(L)  [gpr.py:59]: lower band of (L L^T)^-1 restricted to the band (differentiable)."""
    if _needs_grad(L):
        return _InverseFromCholeskyBand.apply(L)
    return _inverse_from_cholesky_band_raw(L)


def solve_triang_mat(left, right, transpose_left=False):
    """banded.solve_triang_mat(L, B)  [gpr.py:75]: L^-1 B (or L^-T B) for a lower band L and dense B (M, D) (differentiable)."""
    if _needs_grad(left, right):
        return _SolveTriangMat.apply(left, right, transpose_left)
    return _solve_triang_mat_raw(left, right, transpose_left)


def product_band_band(left, right, left_lower_bandwidth, left_upper_bandwidth, right_lower_bandwidth,
                      right_upper_bandwidth, result_lower_bandwidth, result_upper_bandwidth):
    """banded.product_band_band(...)  [gpr.py:60-69] with the TF op's keyword names (differentiable)."""
    if _needs_grad(left, right):
        return _ProductBandBand.apply(left, right, int(left_lower_bandwidth), int(left_upper_bandwidth), int(right_lower_bandwidth),
                                      int(right_upper_bandwidth), int(result_lower_bandwidth), int(result_upper_bandwidth))
    return _product_band_band_raw(left, right, left_lower_bandwidth, left_upper_bandwidth, right_lower_bandwidth,
                                  right_upper_bandwidth, result_lower_bandwidth, result_upper_bandwidth)
