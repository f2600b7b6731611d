"""Column-wise Kronecker (Khatri-Rao) helpers: mirror of asvgp/kronecker.py:7-33 on the HIP library (d = 2).
Row index of the pair (i1, i2) is i1*m2 + i2, exactly make_kvs_two_sparse's `sparse_repeats` x `sparse_tile`."""
import torch

from ._lib import check, f64c, get_lib, require_cuda, stream_ptr


def make_kvs_coo(bases, X):
    """(rows, cols, data) of the (m1*m2, N) Khatri-Rao design matrix for X (N, 2); (k+1)^2 entries per point."""
    b1, b2 = bases
    assert b1.order == b2.order, "both bases must share the spline order (gpr.py:261)"
    X = f64c(torch.as_tensor(X, device=b1.device))
    require_cuda(X)
    n, k = X.shape[0], b1.order
    ne = (k + 1) ** 2
    rows = torch.empty(ne * n, dtype=torch.int64, device=X.device)
    data = torch.empty(ne * n, dtype=torch.float64, device=X.device)
    check(get_lib().asvgp_kron_evaluate_2d(X.data_ptr(), n, b1.mesh.data_ptr(), b1.mesh.shape[0], b1.delta_np,
                                           b2.mesh.data_ptr(), b2.mesh.shape[0], b2.delta_np, b2.m, k, rows.data_ptr(),
                                           data.data_ptr(), stream_ptr()), "kron_evaluate_2d")
    cols = torch.arange(n, dtype=torch.int64, device=X.device).repeat(ne)
    return rows, cols, data


def make_kvs_sparse(bases, X=None):
    """kronecker.py:32-33.  Two call forms:
      make_kvs_sparse(A_list)   - the reference's signature: a list of sparse per-dimension design matrices (m_i, N), reduced with
                                  make_kvs_two_sparse (any d);
      make_kvs_sparse(bases, X) - the fused form the model uses for d = 2: one kernel evaluates both bases at X (N, 2).
    Returns a torch sparse CSR (prod m_i, N)."""
    if X is None:
        from functools import reduce
        return reduce(make_kvs_two_sparse, list(bases))
    rows, cols, data = make_kvs_coo(bases, X)
    n = cols.shape[0] // (bases[0].order + 1) ** 2
    coo = torch.sparse_coo_tensor(torch.stack([rows, cols]), data, (bases[0].m * bases[1].m, n)).coalesce()
    return coo.to_sparse_csr()


def _coo(A):
    A = A.to_sparse_coo() if A.layout != torch.sparse_coo else A
    return A.coalesce()


def sparse_repeats(A, repeats):
    """kronecker.py:7-15: every row r of A becomes rows r*repeats + i, i = 0..repeats-1 (torch sparse in / CSR out)."""
    A = _coo(A)
    r, c = A.indices()
    i = torch.arange(repeats, device=r.device).repeat_interleave(r.shape[0])
    rows = i + r.repeat(repeats) * repeats
    out = torch.sparse_coo_tensor(torch.stack([rows, c.repeat(repeats)]), A.values().repeat(repeats),
                                  (repeats * A.shape[0], A.shape[1]))
    return out.coalesce().to_sparse_csr()


def sparse_tile(A, repeats):
    """kronecker.py:17-25: A stacked `repeats` times."""
    A = _coo(A)
    r, c = A.indices()
    i = torch.arange(repeats, device=r.device).repeat_interleave(r.shape[0])
    rows = r.repeat(repeats) + i * A.shape[0]
    out = torch.sparse_coo_tensor(torch.stack([rows, c.repeat(repeats)]), A.values().repeat(repeats),
                                  (repeats * A.shape[0], A.shape[1]))
    return out.coalesce().to_sparse_csr()


def make_kvs_two_sparse(A, B):
    """kronecker.py:27-30: column-wise Kronecker product of two sparse design matrices (generic route; the fused kernel
    behind make_kvs_sparse(bases, X) is what the model uses)."""
    M1 = _coo(sparse_repeats(A, B.shape[0]))
    M2 = _coo(sparse_tile(B, A.shape[0]))
    return _coo(M1 * M2).to_sparse_csr()                       # (sparse .multiply, as the reference: nothing is densified)


def twisted_layout(M, bw, force=None):
    """Separator and padding of the two-sided factorisation of a band matrix with M columns and bandwidth bw
    (include/asvgp_hip.h, asvgp_kron_assemble_twisted), or None when the one-sided band Cholesky is used.
    Super-block size Bb = bw rounded up to a multiple of 32; both systems have nb super-blocks, the separator [h, h + Bb) being the
    last one of each: top system = padt identity columns + original columns [0, top_end), bottom system (reversed) = padb identity
    columns + original columns [h, M).  force: None = by size (at least 6 super-blocks), True = wherever nb >= 3, False = never."""
    Bb = ((max(bw, 1) + 31) // 32) * 32
    nb = -(-(M + Bb) // (2 * Bb))
    if force is False or nb < 3 or (force is None and -(-M // Bb) < 6):
        return None
    padt = (2 * nb * Bb - Bb - M) // 2
    top_end = nb * Bb - padt
    h = top_end - Bb
    padb = nb * Bb - (M - h)
    return dict(Bb=Bb, nb=nb, top_end=top_end, h=h, padt=padt, padb=padb, bw=bw)

