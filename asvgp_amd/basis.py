"""B-spline bases: host-side mirror of asvgp/basis.py (SplineBasis, B1Spline ... B6Spline).

Same constructor signature and attributes as the reference (a, b, m, order, mesh, delta, A, B, [C, D], BC,
[BC_grad, BC_ggrad, BC_ggrad_none, BC_none_ggrad]; basis.py:13-18,126-131,...,658-666) with torch fp64 tensors on the
ROCm device instead of TF tensors.  The one-off O(M k) static bands are built on the host from exact rational
per-interval constants; the N-dependent work (evaluate_basis) runs in the HIP library.
"""
from fractions import Fraction
from functools import lru_cache
from math import comb, factorial

import numpy as np
import torch

from ._lib import check, f64c, get_lib, require_cuda, stream_ptr


def _device():
    if not torch.cuda.is_available():
        from ._lib import AsvgpError
        raise AsvgpError("asvgp_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


@lru_cache(maxsize=None)
def _pieces(order, deriv):
    """Monomial coefficients (Fractions, in t) of d^deriv/dt^deriv of N_k(t+i), i=0..k  (cf. basis.py:274-280 etc.)."""
    k = order
    polys = []
    for i in range(k + 1):
        c = [sum(Fraction((-1) ** j * comb(k + 1, j) * comb(k, p) * (i - j) ** (k - p), factorial(k))
                 for j in range(i + 1)) for p in range(k + 1)]
        for _ in range(deriv):
            c = [c[p] * p for p in range(1, len(c))] + [Fraction(0)]
        polys.append(tuple(c))
    return tuple(polys)


@lru_cache(maxsize=None)
def _gram_lists(order, deriv):
    """int_0^1 N^(p)(t+i) N^(p)(t+i+d) dt as Fractions: the d0..dk lists of basis.py l2_*_inner_product."""
    C = _pieces(order, deriv)
    out = []
    for d in range(order + 1):
        out.append(tuple(sum(a * b / (p + q + 1) for p, a in enumerate(C[i]) for q, b in enumerate(C[i + d]) if a and b)
                         for i in range(order + 1 - d)))
    return tuple(out)


def make_mesh(a, b, m, order):
    """basis.py:17-18: tf.linspace(a, b, m-(order-1)) cast to f64.  Python-float endpoints make TF compute the
    linspace in float32 - reproduced bit for bit; int / numpy-f64 endpoints give fp64."""
    n = m - (order - 1)
    if type(a) is float or type(b) is float:
        f = np.float32
        s, e = f(a), f(b)
        step = f((e - s) / f(n - 1))
        mesh = (s + step * np.arange(n, dtype=f)).astype(f)
        mesh[-1] = e
        mesh = mesh.astype(np.float64)
    else:
        mesh = np.linspace(np.float64(a), np.float64(b), n)
    return mesh, np.float64(mesh[1] - mesh[0])


class SplineBasis:
    """asvgp/basis.py:8-114."""
    order = None
    _static = ()

    def __init__(self, a, b, m, device=None):
        self.a, self.b, self.m = a, b, m
        self.device = device if device is not None else _device()
        self.mesh_np, delta = make_mesh(a, b, m, self.order)
        self.delta_np = float(delta)
        self.mesh = torch.from_numpy(self.mesh_np).to(self.device)
        self.delta = torch.tensor(self.delta_np, dtype=torch.float64, device=self.device)
        self.static_np = {}
        for name in self._static:
            band = self._build_static(name)
            self.static_np[name] = band
            setattr(self, name, torch.from_numpy(band).to(self.device))

    # -- static bands -------------------------------------------------------------------------------------
    def _make_banded_matrix(self, diags):
        """basis.py:31-45 (pad='right')."""
        rows = []
        for i, diag in enumerate(diags):
            diag = np.asarray(diag, dtype=np.float64)
            lhs = np.cumsum(diag)
            mid = np.repeat(np.sum(diag), self.m - 2 * diag.shape[0] - i)
            rows.append(np.concatenate([lhs, mid, lhs[::-1], np.zeros(i)]))
        return np.stack(rows, axis=0)

    def _gram(self, deriv):
        d = self.delta_np
        lists = _gram_lists(self.order, deriv)
        if deriv == 0:
            diags = [[float(c) * d for c in lst] for lst in lists]
        else:
            diags = [[float(c) / d ** (2 * deriv - 1) for c in lst] for lst in lists]
        return self._make_banded_matrix(diags)

    def make_boundary_conditions(self, dx=0):
        """basis.py:82-114.  dx=3,4 are identically zero as written in the reference."""
        k, m = self.order, self.m
        band = np.zeros((k + 1, m))
        if dx in (3, 4):
            return band
        vals = [float(c[0]) / self.delta_np ** dx for c in _pieces(k, dx)]   # phi^(dx)(a): t = 0
        lhs = np.array([vals[k - r] for r in range(k)])                      # row r <- piece k-r
        mat = np.outer(lhs, lhs)
        for i in range(k):
            l = np.diagonal(mat, offset=i)
            band[i, :l.shape[0]] = l
            band[i, m - i - l.shape[0]:m - i] = l
        return band

    def _build_static(self, name):
        if name in ("A", "B", "C", "D"):
            return self._gram("ABCD".index(name))
        return self.make_boundary_conditions({"BC": 0, "BC_grad": 1, "BC_ggrad": 2, "BC_ggrad_none": 3,
                                              "BC_none_ggrad": 4}[name])

    # -- N-dependent work: HIP ------------------------------------------------------------------------------
    def neighbour_index(self, X):
        """basis.py:58: relu(searchsorted(mesh, X) - 1) as int64 (device)."""
        x = f64c(torch.as_tensor(X, device=self.device)).reshape(-1)
        require_cuda(x)
        idx = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
        check(get_lib().asvgp_phi_index_1d(x.data_ptr(), x.shape[0], self.mesh.data_ptr(), self.mesh.shape[0],
                                           self.delta_np, idx.data_ptr(), stream_ptr()), "phi_index_1d")
        return idx

    def evaluate_basis_coo(self, X, dx=0):
        """(rows, cols, data) in the reference's concat order (basis.py:62,72-73)."""
        x = f64c(torch.as_tensor(X, device=self.device)).reshape(-1)
        require_cuda(x)
        n = x.shape[0]
        k = self.order
        rows = torch.empty((k + 1) * n, dtype=torch.int64, device=x.device)
        data = torch.empty((k + 1) * n, dtype=torch.float64, device=x.device)
        if int(dx) > 3:
            raise NotImplementedError
        check(get_lib().asvgp_phi_evaluate_1d(x.data_ptr(), n, self.mesh.data_ptr(), self.mesh.shape[0], self.delta_np,
                                              k, int(dx), rows.data_ptr(), data.data_ptr(), stream_ptr()),
              "phi_evaluate_1d")
        cols = torch.arange(n, dtype=torch.int64, device=x.device).repeat(k + 1)
        return rows, cols, data

    def evaluate_basis(self, X, dx=0, sparse=True):
        """basis.py:51-80: (m, n) design matrix; sparse=True -> torch sparse CSR (the reference returns scipy CSR),
        sparse=False -> dense scatter."""
        rows, cols, data = self.evaluate_basis_coo(X, dx)
        n = cols.shape[0] // (self.order + 1)
        coo = torch.sparse_coo_tensor(torch.stack([rows, cols]), data, (self.m, n)).coalesce()
        return coo.to_sparse_csr() if sparse else coo.to_dense()


def _mk(order_, static_, min_m=None):
    class _B(SplineBasis):
        order = order_
        _static = static_

        def __init__(self, a, b, m, device=None):
            if min_m is not None and m < min_m:
                raise NameError("Not enough basis functions m >= %d" % min_m)   # basis.py:379-380
            super().__init__(a, b, m, device)
    _B.__name__ = _B.__qualname__ = "B%dSpline" % order_
    return _B


_ALL = ("A", "B", "C", "D", "BC", "BC_grad", "BC_ggrad", "BC_ggrad_none", "BC_none_ggrad")
B1Spline = _mk(1, ("A", "B", "BC"))                                   # basis.py:117-131
B2Spline = _mk(2, ("A", "B", "C", "BC", "BC_grad"))                  # basis.py:170-186
B3Spline = _mk(3, _ALL)                                                # basis.py:252-272
B4Spline = _mk(4, _ALL, min_m=12)                                      # basis.py:372-395
B5Spline = _mk(5, _ALL)                                                # basis.py:506-526
B6Spline = _mk(6, ("A", "B", "C", "D", "BC", "BC_grad"))             # basis.py:649-666
