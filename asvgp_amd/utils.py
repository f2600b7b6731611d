"""Band <-> sparse helpers: mirror of asvgp/utils.py (the functions the hot path uses) on torch device tensors."""
import torch

from . import banded


def symmetrise_banded(K_lower):
    """utils.py:7-9: lower band (k+1, M) -> symmetric band (2k+1, M) via transpose_band + concat."""
    K_upper = banded.transpose_band(K_lower, K_lower.shape[0] - 1, 0)
    return torch.cat([K_upper[:-1, :], K_lower], dim=0)


def sparse_to_band(K_sparse, bandwidth):
    """utils.py:24-30: main + `bandwidth` sub-diagonals of a (torch sparse or dense) M x M matrix, right-padded."""
    K = K_sparse.to_dense() if K_sparse.layout != torch.strided else K_sparse
    M = K.shape[0]
    rows = [torch.diagonal(K, 0)]
    for i in range(1, bandwidth + 1):
        rows.append(torch.cat([torch.diagonal(K, -i), torch.zeros(i, dtype=K.dtype, device=K.device)]))
    return torch.stack(rows, dim=0).to(torch.float64)


def band_to_sparse(K_lower):
    """utils.py:32-33: lower-triangular sparse matrix (torch sparse COO) from a lower band; explicit zeros dropped."""
    k1, M = K_lower.shape
    dev = K_lower.device
    r, c, v = [], [], []
    for d in range(k1):
        j = torch.arange(0, M - d, device=dev)
        r.append(j + d)
        c.append(j)
        v.append(K_lower[d, :M - d])
    return torch.sparse_coo_tensor(torch.stack([torch.cat(r), torch.cat(c)]), torch.cat(v), (M, M)).coalesce()


def band_to_dense_sym(K_lower):
    k = K_lower.shape[0] - 1
    return banded.unpack_banded_matrix_to_dense(symmetrise_banded(K_lower), k, k)


def _kron_dense(mats):
    out = mats[0]
    for m in mats[1:]:
        out = torch.kron(out, m)
    return out


def bands_to_kron_cholesky(K_sparse, mat_bandwidth):
    """utils.py:45-51: dense kron(K_i) and kron(chol(K_i)) - O(M_tot^2) memory; the HIP GPR_kron never calls it
    (log|Kuu| and the trace are taken factor-wise), it exists for drop-in completeness and for the tests."""
    K_dense = [band_to_dense_sym(k) for k in K_sparse]
    Ls = [torch.linalg.cholesky(K) for K in K_dense]
    return _kron_dense(K_dense), _kron_dense(Ls)


def bands_to_sparse(K_bands, mat_bandwidth):
    """utils.py:53-57: sparse Kronecker product of the symmetric factors (torch sparse COO; d = 2 like the reference)."""
    K_dense = [band_to_dense_sym(k) for k in K_bands]
    return _kron_dense(K_dense[:2]).to_sparse()
