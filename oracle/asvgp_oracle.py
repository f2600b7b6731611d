"""CPU oracle for the ASVGP hot path  --  TEST INFRASTRUCTURE, NOT THE PRODUCT.

A plain numpy/scipy fp64 restatement of the reference algorithm for the path named by
BASELINE.json:north_star.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product package (asvgp_amd/) never does.

Parity pinning: this oracle is checked (tests/test_oracle_golden.py) against fixtures
produced by the reference's own basis.py / inducing_features.py / utils.py / kronecker.py
(tests/golden/make_golden.py) and against the only numeric golden in the reference,
the notebook printout ELBO = -60.8356263428725 (experiments/snelson/example.ipynb:78).
The banded_matrices ops (C++/TF custom ops, wheel banded_matrices-0.0.6, branch
awav/fix-banded-hashable-tensor; README.md:38,49) are NOT in /root/reference: their
published semantics are restated here and anchored on the reference's call sites
(gpr.py:56-75) - every op is uniquely defined linear algebra (Cholesky, inverse, product).

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from fractions import Fraction
from math import comb, factorial

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------------------
# Cardinal B-spline pieces (basis.py:133-136,188-192,274-280,397-403,528-535,668-676)
# --------------------------------------------------------------------------------------

def piece_coeffs(order, deriv=0):
    """Exact monomial coefficients (in t) of the `deriv`-th t-derivative of the order+1 pieces
    of the cardinal B-spline N_k on knots 0..k+1:  piece i (i=0..k) is N_k(t+i), t in [0,1].
    The reference's b1..b_{k+1} are these pieces written in (X, u, delta) with t=(X-u)/delta
    (e.g. basis.py:276-279: b1 = t^3/6 ... b4 = (1-t)^3/6).  Returns list[k+1] of list[k+1] Fractions,
    coefficient c[i][p] multiplying t^p."""
    k = order
    out = []
    for i in range(k + 1):
        c = [Fraction(0)] * (k + 1)
        # N_k(s) = 1/k! sum_j (-1)^j C(k+1,j) (s-j)_+^k ; on piece i: s=t+i, j=0..i
        for j in range(i + 1):
            sgn = -1 if j % 2 else 1
            for p in range(k + 1):  # (t + (i-j))^k binomial expansion
                c[p] += Fraction(sgn * comb(k + 1, j) * comb(k, p) * (i - j) ** (k - p), factorial(k))
        for _ in range(deriv):
            c = [c[p] * p for p in range(1, len(c))] + [Fraction(0)]
        out.append(c)
    return out


def piece_values(order, t, deriv=0):
    """(order+1, n) array: N_k^(deriv)(t+i) by Horner in t (t-derivative; divide by delta^deriv for d/dx)."""
    t = np.asarray(t, dtype=np.float64)
    C = piece_coeffs(order, deriv)
    out = np.empty((order + 1,) + t.shape)
    for i, c in enumerate(C):
        acc = np.zeros_like(t) + float(c[order])
        for p in range(order - 1, -1, -1):
            acc = acc * t + float(c[p])
        out[i] = acc
    return out


def gram_constants(order, deriv):
    """Per-interval rational constants int_0^1 N^(p)(t+i) N^(p)(t+i+d) dt  (the lists d0..dk of
    basis.py:303-369 (B3), 429-503 (B4), ... without the delta^(1-2p) factor).
    Returns list over d=0..k of list over i=0..k-d of Fractions."""
    C = piece_coeffs(order, deriv)
    k = order
    lists = []
    for d in range(k + 1):
        lst = []
        for i in range(k + 1 - d):
            a, b = C[i], C[i + d]
            tot = Fraction(0)
            for p, ap in enumerate(a):
                for q, bq in enumerate(b):
                    if ap and bq:
                        tot += ap * bq / (p + q + 1)
            lst.append(tot)
        lists.append(lst)
    return lists

# --------------------------------------------------------------------------------------
# Mesh, index rule, design matrix  (basis.py:13-18, 51-80)
# --------------------------------------------------------------------------------------

def make_mesh(a, b, m, order):
    """basis.py:17-18.  tf.linspace infers float32 from python-float endpoints and is only then cast
    to f64; int / numpy-f64 endpoints give an exact fp64 linspace."""
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


def neighbour_index(mesh, x):
    """basis.py:58: relu(searchsorted_left(mesh, x) - 1)  -- a table search, not arithmetic."""
    return np.maximum(np.searchsorted(mesh, x, side="left") - 1, 0).astype(np.int64)


def evaluate_basis_coo(mesh, delta, order, m, x, deriv=0):
    """basis.py:51-76.  Returns (rows, cols, data) in the reference's concat order:
    piece i (i=0..order) block-concatenated, rows = idx+order-i, cols = point index."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    n = x.shape[0]
    idx = neighbour_index(mesh, x)
    u = mesh[idx]
    t = (x - u) / delta
    vals = piece_values(order, t, deriv) / (delta ** deriv)
    rows = np.concatenate([idx + order - i for i in range(order + 1)])
    cols = np.tile(np.arange(n, dtype=np.int64), order + 1)
    return rows, cols, vals.reshape(-1)


def evaluate_basis(mesh, delta, order, m, x, deriv=0, sparse=True):
    """basis.py:51-80: CSR (m, n) design matrix Phi (sparse=True) or dense scatter (sparse=False)."""
    rows, cols, data = evaluate_basis_coo(mesh, delta, order, m, x, deriv)
    n = np.asarray(x).reshape(-1).shape[0]
    if sparse:
        return sp.csr_matrix((data, (rows, cols)), shape=(m, n))
    out = np.zeros((m, n))
    np.add.at(out, (rows, cols), data)
    return out

# --------------------------------------------------------------------------------------
# Static Gram bands and boundary bands  (basis.py:31-45, 82-114)
# --------------------------------------------------------------------------------------

def make_banded_matrix(diags, m):
    """basis.py:31-45 (pad='right'): row i = [cumsum(d_i), sum(d_i) x (m-2 len-i), reversed cumsum, i zeros]."""
    bands = []
    for i, diag in enumerate(diags):
        diag = np.asarray(diag, dtype=np.float64)
        lhs = np.cumsum(diag)
        mid = np.repeat(np.sum(diag), m - 2 * diag.shape[0] - i)
        bands.append(np.concatenate([lhs, mid, lhs[::-1], np.zeros(i)]))
    return np.stack(bands, axis=0)


def gram_band(order, deriv, delta, m):
    """A (deriv=0), B (1), C (2), D (3): basis.py l2_*_inner_product -> _make_banded_matrix.
    constant * delta^(1-2p), formed the way the reference forms it (c*delta, c/delta^(2p-1))."""
    consts = gram_constants(order, deriv)
    diags = []
    for lst in consts:
        if deriv == 0:
            diags.append([float(c) * delta for c in lst])
        else:
            diags.append([float(c) / delta ** (2 * deriv - 1) for c in lst])
    return make_banded_matrix(diags, m)


def boundary_band(order, dx, delta, m):
    """basis.py:82-114 (pad='right').  dx=0,1,2: outer product of phi^(dx)(a)[:order] with itself, placed
    at the top-left and (the same diagonals) bottom-right corners, plus a zero last row.
    dx=3,4 ('ggrad_none'/'none_ggrad'): rhs is evaluated at b whose first `order` rows are zero, so the
    band is identically zero for m > 2*order+1."""
    k = order
    band = np.zeros((k + 1, m))
    if dx in (3, 4):
        return band
    # phi^(dx)(a): idx=0, u=a, t=0; row r (r<order) holds piece i=order-r
    v = piece_values(k, np.array([0.0]), dx)[:, 0] / (delta ** dx)
    lhs = np.array([v[k - r] for r in range(k)])
    mat = np.outer(lhs, lhs)
    for i in range(k):
        l = np.diagonal(mat, offset=i)
        band[i, :l.shape[0]] = l
        start = m - i - l.shape[0]
        band[i, start:start + l.shape[0]] = l
    return band


STATIC_AVAILABLE = {  # basis.py:126-131,179-186,261-272,384-395,515-526,658-666
    1: ("A", "B", "BC"),
    2: ("A", "B", "C", "BC", "BC_grad"),
    3: ("A", "B", "C", "D", "BC", "BC_grad", "BC_ggrad", "BC_ggrad_none", "BC_none_ggrad"),
    4: ("A", "B", "C", "D", "BC", "BC_grad", "BC_ggrad", "BC_ggrad_none", "BC_none_ggrad"),
    5: ("A", "B", "C", "D", "BC", "BC_grad", "BC_ggrad", "BC_ggrad_none", "BC_none_ggrad"),
    6: ("A", "B", "C", "D", "BC", "BC_grad"),
}


class Basis:
    """B{order}Spline(a, b, m) (basis.py:117,170,252,372,506,649)."""

    def __init__(self, order, a, b, m):
        if order == 4 and m < 12:
            raise NameError("Not enough basis functions m >= 12")  # basis.py:379-380
        self.order, self.a, self.b, self.m = order, a, b, m
        self.mesh, self.delta = make_mesh(a, b, m, order)
        names = STATIC_AVAILABLE[order]
        for nm, p in (("A", 0), ("B", 1), ("C", 2), ("D", 3)):
            if nm in names:
                setattr(self, nm, gram_band(order, p, self.delta, m))
        for nm, dx in (("BC", 0), ("BC_grad", 1), ("BC_ggrad", 2), ("BC_ggrad_none", 3), ("BC_none_ggrad", 4)):
            if nm in names:
                setattr(self, nm, boundary_band(order, dx, self.delta, m))

    def evaluate_basis(self, X, dx=0, sparse=True):
        return evaluate_basis(self.mesh, self.delta, self.order, self.m, X, int(dx), sparse)

# --------------------------------------------------------------------------------------
# Kuu assembly  (inducing_features.py:12-44)
# --------------------------------------------------------------------------------------

MATERN12, MATERN32, MATERN52 = 0, 1, 2
_S3, _S5 = np.sqrt(3.0), np.sqrt(5.0)


def kuu_terms(kind, v, l):
    """[(static band name, coefficient, d coefficient / d lengthscale)]  inducing_features.py:16-44."""
    if kind == MATERN12:
        return [("A", 1 / (2 * l * v), -1 / (2 * l * l * v)), ("B", l / (2 * v), 1 / (2 * v)),
                ("BC", 1 / (2 * v), 0.0)]
    if kind == MATERN32:
        return [("A", _S3 / (4 * l * v), -_S3 / (4 * l * l * v)),
                ("B", l / (2 * _S3 * v), 1 / (2 * _S3 * v)),
                ("C", l ** 3 / (12 * _S3 * v), 3 * l * l / (12 * _S3 * v)),
                ("BC", 1 / (2 * v), 0.0), ("BC_grad", l ** 2 / (2 * v), l / v)]
    if kind == MATERN52:
        return [("A", (3 * _S5) / (16 * l * v), -(3 * _S5) / (16 * l * l * v)),
                ("B", (9 * l) / (16 * _S5 * v), 9 / (16 * _S5 * v)),
                ("C", (9 * l ** 3) / (80 * _S5 * v), (27 * l * l) / (80 * _S5 * v)),
                ("D", (3 * l ** 5) / (400 * _S5 * v), (15 * l ** 4) / (400 * _S5 * v)),
                ("BC", 9 / (16 * v), 0.0), ("BC_grad", (3 * l ** 2) / (10 * v), (6 * l) / (10 * v)),
                ("BC_ggrad", (9 * l ** 4) / (400 * v), (36 * l ** 3) / (400 * v)),
                ("BC_ggrad_none", (3 * l ** 2) / (80 * v), (6 * l) / (80 * v)),
                ("BC_none_ggrad", (3 * l ** 2) / (80 * v), (6 * l) / (80 * v))]
    raise ValueError(kind)


def make_Kuu(basis, kind, v, l, want_dl=False):
    """inducing_features.py:12-44: lower band (k+1, m).  Optionally also dKuu/dl (same static bands)."""
    K = np.zeros((basis.order + 1, basis.m))
    dK = np.zeros_like(K)
    for nm, c, dc in kuu_terms(kind, v, l):
        S = getattr(basis, nm)  # AttributeError if the basis lacks the band - as in the reference
        K = K + c * S
        dK = dK + dc * S
    return (K, dK) if want_dl else K

# --------------------------------------------------------------------------------------
# banded_matrices.banded op semantics (call sites gpr.py:56-75; utils.py:7-9,36-57)
# band layout: (l+u+1, n), band[u+i-j, j] = M[i, j]
# --------------------------------------------------------------------------------------

def unpack_banded_matrix_to_dense(band, l, u):
    band = np.asarray(band)
    n = band.shape[1]
    out = np.zeros((n, n), dtype=band.dtype)
    for r in range(l + u + 1):
        d = r - u  # i - j
        for j in range(n):
            i = j + d
            if 0 <= i < n:
                out[i, j] = band[r, j]
    return out


def pack_dense_matrix_to_banded(dense, l, u):
    dense = np.asarray(dense)
    n = dense.shape[0]
    band = np.zeros((l + u + 1, n), dtype=dense.dtype)
    for r in range(l + u + 1):
        d = r - u
        for j in range(n):
            i = j + d
            if 0 <= i < n:
                band[r, j] = dense[i, j]
    return band


def transpose_band(band, l, u):
    """(l,u) band of M -> (u,l) band of M^T (utils.py:8).  out[l + j - i, i] = M[i, j]."""
    band = np.asarray(band)
    n = band.shape[1]
    out = np.zeros_like(band)
    for r in range(l + u + 1):
        d = r - u  # i-j of the source entry
        # source entry M[i,j], i=j+d  ->  M^T[j,i]: row index in out = l + (j - i) = l - d, column i
        j = np.arange(max(0, -d), min(n, n - d))
        out[l - d, j + d] = band[r, j]
    return out


def symmetrise_band(lower, l):
    """gpr.py:62 / utils.py:7-9: lower band (l+1, n) -> symmetric band (2l+1, n)."""
    upper = transpose_band(lower, l, 0)
    return np.concatenate([upper[:-1, :], lower], axis=0)


def cholesky_band(K):
    """gpr.py:56,73: lower-band Cholesky, column recurrence."""
    K = np.asarray(K, dtype=np.float64)
    k, M = K.shape[0] - 1, K.shape[1]
    L = np.zeros_like(K)
    for j in range(M):
        for i in range(j, min(j + k, M - 1) + 1):
            s = K[i - j, j]
            for p in range(max(0, i - k), j):
                s -= L[i - p, p] * L[j - p, p]
            if i == j:
                if not s > 0:
                    raise np.linalg.LinAlgError("band not positive definite at column %d" % j)
                L[0, j] = np.sqrt(s)
            else:
                L[i - j, j] = s / L[0, j]
    return L


def cholesky_band_jvp(K, dK):
    """Forward-mode tangent of cholesky_band along dK (needed for d/dl tr(Kuu^-1 A))."""
    k, M = K.shape[0] - 1, K.shape[1]
    L = np.zeros_like(K)
    dL = np.zeros_like(K)
    for j in range(M):
        for i in range(j, min(j + k, M - 1) + 1):
            s, ds = K[i - j, j], dK[i - j, j]
            for p in range(max(0, i - k), j):
                s -= L[i - p, p] * L[j - p, p]
                ds -= dL[i - p, p] * L[j - p, p] + L[i - p, p] * dL[j - p, p]
            if i == j:
                L[0, j] = np.sqrt(s)
                dL[0, j] = ds / (2 * L[0, j])
            else:
                L[i - j, j] = s / L[0, j]
                dL[i - j, j] = (ds - L[i - j, j] * dL[0, j]) / L[0, j]
    return L, dL


def inverse_from_cholesky_band(L, dL=None):
    """gpr.py:59: lower band of (L L^T)^-1 restricted to the band (Takahashi / sparse inverse subset).
    With dL also returns the tangent of that band."""
    L = np.asarray(L, dtype=np.float64)
    k, M = L.shape[0] - 1, L.shape[1]
    S = np.zeros_like(L)
    dS = np.zeros_like(L) if dL is not None else None

    def sget(X, p, i):  # symmetric in-band read X[p,i]
        return X[p - i, i] if p >= i else X[i - p, p]

    for j in range(M - 1, -1, -1):
        hi = min(j + k, M - 1)
        ljj = L[0, j]
        for i in range(hi, j - 1, -1):
            acc = (1.0 / ljj) if i == j else 0.0
            for p in range(j + 1, hi + 1):
                acc -= L[p - j, j] * sget(S, p, i)
            S[i - j, j] = acc / ljj
            if dL is not None:
                dacc = (-dL[0, j] / (ljj * ljj)) if i == j else 0.0
                for p in range(j + 1, hi + 1):
                    dacc -= dL[p - j, j] * sget(S, p, i) + L[p - j, j] * sget(dS, p, i)
                dS[i - j, j] = (dacc - S[i - j, j] * dL[0, j]) / ljj
    return S if dL is None else (S, dS)


def cholesky_band_vjp(L, Lbar):
    """Reverse mode of cholesky_band (what banded_matrices registers for the op called at gpr.py:56,73): Kbar[d, j] = d<Lbar, L>/dK[d, j]
    for the stored lower-band entries of K (each K[i, j], i > j, is read once by the recurrence).  The adjoint of the column loop."""
    L = np.asarray(L, dtype=np.float64)
    k, M = L.shape[0] - 1, L.shape[1]
    Lb = np.array(Lbar, dtype=np.float64, copy=True)
    Kb = np.zeros_like(L)
    for j in range(M - 1, -1, -1):
        for i in range(min(j + k, M - 1), j - 1, -1):
            if i == j:
                sb = Lb[0, j] / (2.0 * L[0, j])
            else:
                sb = Lb[i - j, j] / L[0, j]
                Lb[0, j] -= Lb[i - j, j] * L[i - j, j] / L[0, j]
            Kb[i - j, j] = sb
            for p in range(max(0, i - k), j):
                Lb[i - p, p] -= sb * L[j - p, p]
                Lb[j - p, p] -= sb * L[i - p, p]
    return Kb


def inverse_from_cholesky_band_vjp(L, S, Sbar):
    """Reverse mode of inverse_from_cholesky_band (gpr.py:59): Lbar = d<Sbar, S>/dL over the stored lower band of L, S the forward
    result.  The adjoint of the Takahashi recurrence above, columns ascending."""
    L = np.asarray(L, dtype=np.float64)
    k, M = L.shape[0] - 1, L.shape[1]
    Sb = np.array(Sbar, dtype=np.float64, copy=True)
    Lb = np.zeros_like(L)

    def sget(X, p, i):
        return X[p - i, i] if p >= i else X[i - p, p]

    for j in range(M):
        hi = min(j + k, M - 1)
        ljj = L[0, j]
        for i in range(j, hi + 1):
            sb = Sb[i - j, j]
            accb = sb / ljj
            Lb[0, j] -= sb * S[i - j, j] / ljj
            if i == j:
                Lb[0, j] -= accb / (ljj * ljj)
            for p in range(j + 1, hi + 1):
                Lb[p - j, j] -= accb * sget(S, p, i)
                if p >= i:
                    Sb[p - i, i] -= accb * L[p - j, j]
                else:
                    Sb[i - p, p] -= accb * L[p - j, j]
    return Lb


def solve_triang_mat_vjp(L, X, Xbar, transpose_left=False):
    """Reverse mode of solve_triang_mat (gpr.py:75): X = L^-1 B gives Bbar = L^-T Xbar, Lbar = -band(Bbar X^T); the transposed solve
    X = L^-T B gives Bbar = L^-1 Xbar, Lbar = -band(X Bbar^T).  Returns (Lbar (k+1, M), Bbar (M, D))."""
    L = np.asarray(L, dtype=np.float64)
    k, M = L.shape[0] - 1, L.shape[1]
    Bb = solve_triang_mat(L, Xbar, transpose_left=not transpose_left)
    Lb = np.zeros_like(L)
    for d in range(k + 1):
        if not transpose_left:
            Lb[d, :M - d] = -np.sum(Bb[d:] * X[:M - d], axis=1)
        else:
            Lb[d, :M - d] = -np.sum(X[d:] * Bb[:M - d], axis=1)
    return Lb, Bb


def product_band_band_vjp(left, right, out_bar, ll, lu, rl, ru, ol, ou):
    """Reverse mode of product_band_band (gpr.py:60-69): Leftbar = band_(ll,lu)(Obar Right^T), Rightbar = band_(rl,ru)(Left^T Obar)."""
    Ld = unpack_banded_matrix_to_dense(left, ll, lu)
    Rd = unpack_banded_matrix_to_dense(right, rl, ru)
    Od = unpack_banded_matrix_to_dense(out_bar, ol, ou)
    return pack_dense_matrix_to_banded(Od @ Rd.T, ll, lu), pack_dense_matrix_to_banded(Ld.T @ Od, rl, ru)


def solve_triang_mat(L, B, transpose_left=False):
    """gpr.py:75: L^-1 B (or L^-T B) with L a lower band (k+1, M) and B dense (M, D)."""
    L = np.asarray(L, dtype=np.float64)
    k, M = L.shape[0] - 1, L.shape[1]
    X = np.array(B, dtype=np.float64, copy=True)
    if not transpose_left:
        for i in range(M):
            for p in range(max(0, i - k), i):
                X[i] -= L[i - p, p] * X[p]
            X[i] /= L[0, i]
    else:
        for i in range(M - 1, -1, -1):
            for p in range(i + 1, min(i + k, M - 1) + 1):
                X[i] -= L[p - i, i] * X[p]
            X[i] /= L[0, i]
    return X


def product_band_band(left, right, left_lower_bandwidth, left_upper_bandwidth, right_lower_bandwidth,
                      right_upper_bandwidth, result_lower_bandwidth, result_upper_bandwidth):
    """gpr.py:60-69: banded x banded cropped to the result band (dense reference implementation)."""
    Ld = unpack_banded_matrix_to_dense(left, left_lower_bandwidth, left_upper_bandwidth)
    Rd = unpack_banded_matrix_to_dense(right, right_lower_bandwidth, right_upper_bandwidth)
    return pack_dense_matrix_to_banded(Ld @ Rd, result_lower_bandwidth, result_upper_bandwidth)


def band_sym_dot(S, A):
    """<sym(S), sym(A)> for two lower bands = sum_j S0 A0 + 2 sum_{d>=1} S_d A_d  (= the trace gpr.py:60-70)."""
    return float(np.sum(S[0] * A[0]) + 2.0 * np.sum(S[1:] * A[1:]))


def band_sym_matvec(A, x):
    """sym(A) @ x for a lower band A (k+1, M), x (M, D)."""
    k, M = A.shape[0] - 1, A.shape[1]
    out = A[0][:, None] * x
    for d in range(1, k + 1):
        out[d:] += A[d, :M - d][:, None] * x[:M - d]
        out[:M - d] += A[d, :M - d][:, None] * x[d:]
    return out

# --------------------------------------------------------------------------------------
# utils.py helpers
# --------------------------------------------------------------------------------------

def sparse_to_band(K_sparse, bandwidth):
    """utils.py:24-30: main + `bandwidth` sub-diagonals, right-padded."""
    M = K_sparse.shape[0]
    rows = [np.asarray(K_sparse.diagonal()).reshape(-1)]
    for i in range(1, bandwidth + 1):
        rows.append(np.concatenate([np.asarray(K_sparse.diagonal(k=-i)).reshape(-1), np.zeros(i)]))
    return np.stack(rows, axis=0).astype(np.float64)


def band_to_sparse(K_lower):
    """utils.py:32-33 (lower-triangular sparse from a lower band)."""
    k1, M = K_lower.shape
    return sp.spdiags(K_lower, np.arange(0, -k1, -1), M, M)

# --------------------------------------------------------------------------------------
# GPR_1d  (gpr.py:18-136)
# --------------------------------------------------------------------------------------

def sufficient_stats(basis, X, y):
    """gpr.py:39-44, through the same steps the reference takes: CSR Phi, Phi@y, Phi@Phi.T, ->band, sum y^2."""
    Kuf = basis.evaluate_basis(X, dx=0, sparse=True)
    Kuf_y = np.asarray(Kuf @ y)
    KK = Kuf @ Kuf.T
    band = sparse_to_band(KK, basis.order)
    return band, Kuf_y, float(np.sum(np.square(y)))


def sufficient_stats_direct(basis, X, y):
    """Same result by direct accumulation (no CSR) - the reduction order the GPU kernel mirrors."""
    k, M = basis.order, basis.m
    x = np.asarray(X, dtype=np.float64).reshape(-1)
    y = np.asarray(y, dtype=np.float64).reshape(x.shape[0], -1)
    idx = neighbour_index(basis.mesh, x)
    t = (x - basis.mesh[idx]) / basis.delta
    vals = piece_values(k, t)  # piece i -> row idx + k - i
    band = np.zeros((k + 1, M))
    rhs = np.zeros((M, y.shape[1]))
    for i in range(k + 1):
        np.add.at(rhs, idx + k - i, vals[i][:, None] * y)
        for j in range(i, k + 1):  # row_i = idx+k-i >= row_j = idx+k-j ; d = j-i
            np.add.at(band[j - i], idx + k - j, vals[i] * vals[j])
    return band, rhs, float(np.sum(np.square(y)))


def elbo_1d(Kuu, A, b, yy, N, v, s, D=1):
    """gpr.py:49-89 from the sufficient statistics.  Returns (elbo, parts dict)."""
    L_Kuu = cholesky_band(Kuu)
    logdet_K = float(np.sum(np.log(np.square(L_Kuu[0]))))
    Kinv = inverse_from_cholesky_band(L_Kuu)
    trace_term = band_sym_dot(Kinv, A)
    P = A / s + Kuu
    L_P = cholesky_band(P)
    logdet_P = float(np.sum(np.log(np.square(L_P[0]))))
    c = solve_triang_mat(L_P, b) / s
    ND = N * D
    elbo = -0.5 * ND * np.log(2 * np.pi * s)
    elbo -= 0.5 * D * logdet_P
    elbo += 0.5 * D * logdet_K
    elbo -= 0.5 * yy / s
    elbo += 0.5 * float(np.sum(np.square(c)))
    elbo -= 0.5 * N * v / s
    elbo += 0.5 * trace_term / s
    return float(elbo), dict(L_Kuu=L_Kuu, L_P=L_P, Kinv=Kinv, c=c, logdet_K=logdet_K, logdet_P=logdet_P,
                             trace_term=trace_term)


def elbo_grad_1d(basis, kind, A, b, yy, N, v, l, s):
    """ELBO and its gradient w.r.t. (v, l, s) using only banded quantities.
    The reference obtains the same numbers by TF reverse-mode through the banded_matrices op gradients."""
    D = b.shape[1]
    Kuu, dKl = make_Kuu(basis, kind, v, l, want_dl=True)
    LK, dLK = cholesky_band_jvp(Kuu, dKl)
    SK, dSK = inverse_from_cholesky_band(LK, dLK)
    P = A / s + Kuu
    LP = cholesky_band(P)
    SP = inverse_from_cholesky_band(LP)
    c = solve_triang_mat(LP, b) / s
    alpha = solve_triang_mat(LP, c, transpose_left=True)  # P^-1 b / s
    logdet_K = float(np.sum(np.log(np.square(LK[0]))))
    logdet_P = float(np.sum(np.log(np.square(LP[0]))))
    trKA = band_sym_dot(SK, A)
    elbo = (-0.5 * N * D * np.log(2 * np.pi * s) - 0.5 * D * logdet_P + 0.5 * D * logdet_K - 0.5 * yy / s
            + 0.5 * float(np.sum(c * c)) - 0.5 * N * v / s + 0.5 * trKA / s)

    def G_dot(Kdot, dtrKA_along):
        # <G, Kdot>, G = 1/2 (D Kuu^-1 - D P^-1 - alpha alpha^T - Kuu^-1 A Kuu^-1 / s);
        # <Kuu^-1 A Kuu^-1, Kdot> = - d/dKdot tr(Kuu^-1 A)
        aKa = float(np.sum(alpha * band_sym_matvec(Kdot, alpha)))
        return 0.5 * (D * band_sym_dot(SK, Kdot) - D * band_sym_dot(SP, Kdot) - aKa + dtrKA_along / s)

    d_l = G_dot(dKl, band_sym_dot(dSK, A))
    # dKuu/dv = -Kuu/v  =>  d tr(Kuu^-1 A)/dv = + tr(Kuu^-1 A)/v
    d_v = G_dot(-Kuu / v, trKA / v) - 0.5 * N / s
    aAa = float(np.sum(alpha * band_sym_matvec(A, alpha)))
    d_s = (-0.5 * N * D / s + 0.5 * D * band_sym_dot(SP, A) / s ** 2 + 0.5 * yy / s ** 2 + 0.5 * aAa / s ** 2
           - float(np.sum(b * alpha)) / s ** 2 + 0.5 * N * v / s ** 2 - 0.5 * trKA / s ** 2)
    return float(elbo), np.array([d_v, d_l, d_s]), dict(alpha=alpha, SK=SK, SP=SP, LK=LK, LP=LP, c=c)


def elbo_grad_1d_extended(basis, kind, A, b, yy, N, v, l, s):
    """The same bound and gradient as elbo_grad_1d (gpr.py:49-89) with every banded recurrence
    carried in numpy long double (x86: 80-bit, 64-bit mantissa) from the fp64 inputs Kuu, dKuu/dl, A, b.  This is the
    'truth' the fp64 evaluation orders are measured against when cond(Kuu) eats digits (tests at the BASELINE
    size): it shows how much of a difference is rounding of the reference's own fp64 order.  D = 1 only."""
    ld = np.longdouble
    Kuu, dKl = make_Kuu(basis, kind, v, l, want_dl=True)
    k, M = Kuu.shape[0] - 1, Kuu.shape[1]
    assert b.shape[1] == 1

    def chol(Kb, dKb=None):
        L = np.zeros((k + 1, M), ld)
        dL = np.zeros((k + 1, M), ld) if dKb is not None else None
        for j in range(M):
            for i in range(j, min(j + k, M - 1) + 1):
                lo = max(0, i - k)
                p = np.arange(lo, j)
                sacc = ld(Kb[i - j, j]) - np.sum(L[i - p, p] * L[j - p, p])
                if dKb is not None:
                    dacc = ld(dKb[i - j, j]) - np.sum(dL[i - p, p] * L[j - p, p] + L[i - p, p] * dL[j - p, p])
                if i == j:
                    L[0, j] = np.sqrt(sacc)
                    if dKb is not None:
                        dL[0, j] = dacc / (2 * L[0, j])
                else:
                    L[i - j, j] = sacc / L[0, j]
                    if dKb is not None:
                        dL[i - j, j] = (dacc - L[i - j, j] * dL[0, j]) / L[0, j]
        return L, dL

    def takahashi(L, dL=None):
        S = np.zeros((k + 1, M), ld)
        dS = np.zeros((k + 1, M), ld) if dL is not None else None

        def sget(X, p, i):
            return X[p - i, i] if p >= i else X[i - p, p]
        for j in range(M - 1, -1, -1):
            hi = min(j + k, M - 1)
            ljj = L[0, j]
            for i in range(hi, j - 1, -1):
                acc = (ld(1) / ljj) if i == j else ld(0)
                dacc = (-dL[0, j] / (ljj * ljj)) if (dL is not None and i == j) else ld(0)
                for p in range(j + 1, hi + 1):
                    acc -= L[p - j, j] * sget(S, p, i)
                    if dL is not None:
                        dacc -= dL[p - j, j] * sget(S, p, i) + L[p - j, j] * sget(dS, p, i)
                S[i - j, j] = acc / ljj
                if dL is not None:
                    dS[i - j, j] = (dacc - S[i - j, j] * dL[0, j]) / ljj
        return S, dS

    def trsv(L, x, trans):
        x = x.astype(ld).copy()
        if not trans:
            for i in range(M):
                p = np.arange(max(0, i - k), i)
                x[i] = (x[i] - np.sum(L[i - p, p] * x[p])) / L[0, i]
        else:
            for i in range(M - 1, -1, -1):
                p = np.arange(i + 1, min(i + k, M - 1) + 1)
                x[i] = (x[i] - np.sum(L[p - i, i] * x[p])) / L[0, i]
        return x

    def symdot(S, B):
        B = B.astype(ld)
        return np.sum(S[0] * B[0]) + 2 * np.sum(S[1:] * B[1:])

    def symmv(B, x):
        B = B.astype(ld)
        out = B[0] * x
        for d in range(1, k + 1):
            out[d:] += B[d, :M - d] * x[:M - d]
            out[:M - d] += B[d, :M - d] * x[d:]
        return out

    Al, bl = A.astype(ld), b.reshape(-1).astype(ld)
    v_, l_, s_, N_, yy_ = ld(v), ld(l), ld(s), ld(N), ld(yy)
    LK, dLK = chol(Kuu, dKl)
    SK, dSK = takahashi(LK, dLK)
    P = Al / s_ + Kuu.astype(ld)
    LP, _ = chol(P)
    SP, _ = takahashi(LP)
    c = trsv(LP, bl, False) / s_
    alpha = trsv(LP, c, True)
    logdet_K = 2 * np.sum(np.log(LK[0]))
    logdet_P = 2 * np.sum(np.log(LP[0]))
    trKA = symdot(SK, A)
    two_pi = 2 * np.arctan(ld(1)) * 4
    elbo = (-N_ / 2 * np.log(two_pi * s_) - logdet_P / 2 + logdet_K / 2 - yy_ / (2 * s_) + np.sum(c * c) / 2
            - N_ * v_ / (2 * s_) + trKA / (2 * s_))

    def G_dot(Kdot, dtr):
        aKa = np.sum(alpha * symmv(Kdot, alpha))
        return (symdot(SK, Kdot) - symdot(SP, Kdot) - aKa + dtr / s_) / 2
    d_l = G_dot(dKl, symdot(dSK, A))
    d_v = G_dot(-Kuu / v, trKA / v_) - N_ / (2 * s_)
    aAa = np.sum(alpha * symmv(A, alpha))
    d_s = (-N_ / (2 * s_) + symdot(SP, A) / (2 * s_ ** 2) + yy_ / (2 * s_ ** 2) + aAa / (2 * s_ ** 2)
           - np.sum(bl * alpha) / s_ ** 2 + N_ * v_ / (2 * s_ ** 2) - trKA / (2 * s_ ** 2))
    return float(elbo), np.array([float(d_v), float(d_l), float(d_s)])


def predict_f_1d(basis, kind, A, b, v, l, s, Xnew):
    """gpr.py:94-120 (full_cov=False): mean = Phi*^T P^-1 b / s ; var = v + |L_P^-1 Phi*|^2 - Phi*^T Kuu^-1 Phi*.
    Dense textbook evaluation (the reference uses CHOLMOD with natural ordering == band Cholesky)."""
    M, k = basis.m, basis.order
    Kuu = make_Kuu(basis, kind, v, l)
    Kd = unpack_banded_matrix_to_dense(symmetrise_band(Kuu, k), k, k)
    Ad = unpack_banded_matrix_to_dense(symmetrise_band(A, k), k, k)
    P = Ad / s + Kd
    LP = np.linalg.cholesky(P)
    c = np.linalg.solve(LP, b) / s
    Kus = basis.evaluate_basis(Xnew, dx=0, sparse=False)
    tmp = np.linalg.solve(LP, Kus)
    mean = tmp.T @ c
    KiKus = np.linalg.solve(Kd, Kus)
    var = v + np.sum(tmp * tmp, axis=0) - np.sum(Kus * KiKus, axis=0)
    return mean, var.reshape(-1, 1)


def predict_f_1d_banded(basis, kind, A, b, v, l, s, Xnew):
    """Same posterior through band quantities only: alpha = P^-1 b/s and the
    (k+1)x(k+1) window of band(P^-1) - band(Kuu^-1) per test point."""
    k = basis.order
    Kuu = make_Kuu(basis, kind, v, l)
    LK = cholesky_band(Kuu)
    SK = inverse_from_cholesky_band(LK)
    LP = cholesky_band(A / s + Kuu)
    SP = inverse_from_cholesky_band(LP)
    alpha = solve_triang_mat(LP, solve_triang_mat(LP, b) / s, transpose_left=True)
    x = np.asarray(Xnew, dtype=np.float64).reshape(-1)
    idx = neighbour_index(basis.mesh, x)
    t = (x - basis.mesh[idx]) / basis.delta
    vals = piece_values(k, t)  # piece i -> row idx+k-i
    W = SP - SK
    mean = np.zeros((x.shape[0], b.shape[1]))
    var = np.full(x.shape[0], float(v))
    for i in range(k + 1):
        ri = idx + k - i
        mean += vals[i][:, None] * alpha[ri]
        for j in range(k + 1):
            rj = idx + k - j
            hi, lo = np.maximum(ri, rj), np.minimum(ri, rj)
            var += vals[i] * vals[j] * W[hi - lo, lo]
    return mean, var.reshape(-1, 1)

# --------------------------------------------------------------------------------------
# Kronecker (kronecker.py:7-33; gpr.py:239-359)
# --------------------------------------------------------------------------------------

def make_kvs_two_sparse(Pa, Pb):
    """kronecker.py:27-30: column-wise Kronecker (Khatri-Rao) product, row = i_a * m_b + i_b."""
    Pa, Pb = sp.csc_matrix(Pa), sp.csc_matrix(Pb)
    n = Pa.shape[1]
    ma, mb = Pa.shape[0], Pb.shape[0]
    rows, cols, data = [], [], []
    for c in range(n):
        ra = Pa.indices[Pa.indptr[c]:Pa.indptr[c + 1]]
        va = Pa.data[Pa.indptr[c]:Pa.indptr[c + 1]]
        rb = Pb.indices[Pb.indptr[c]:Pb.indptr[c + 1]]
        vb = Pb.data[Pb.indptr[c]:Pb.indptr[c + 1]]
        rows.append((ra[:, None] * mb + rb[None, :]).reshape(-1))
        data.append((va[:, None] * vb[None, :]).reshape(-1))
        cols.append(np.full(ra.shape[0] * rb.shape[0], c))
    return sp.csr_matrix((np.concatenate(data), (np.concatenate(rows), np.concatenate(cols))), shape=(ma * mb, n))


def make_kvs_sparse(P_list):
    """kronecker.py:32-33: left fold."""
    out = P_list[0]
    for P in P_list[1:]:
        out = make_kvs_two_sparse(out, P)
    return out


def band_to_dense_sym(lower):
    k = lower.shape[0] - 1
    return unpack_banded_matrix_to_dense(symmetrise_band(lower, k), k, k)


def elbo_kron(bases, kinds, thetas, s, X, y):
    """gpr.py:260-308 (dense, as the reference does it): Kuu = kron(K_i), L = kron(L_i) (utils.py:45-51),
    sum K_diag = N prod v_i (gpr.py:284).  thetas = [(v_i, l_i)]."""
    N = X.shape[0]
    Phis = [bs.evaluate_basis(X[:, i:i + 1]) for i, bs in enumerate(bases)]
    Kuf = make_kvs_sparse(Phis)
    A = (Kuf @ Kuf.T).toarray()
    b = np.asarray(Kuf @ y)
    Ks = [band_to_dense_sym(make_Kuu(bs, kd, v, l)) for bs, kd, (v, l) in zip(bases, kinds, thetas)]
    Kuu = Ks[0]
    for K in Ks[1:]:
        Kuu = np.kron(Kuu, K)
    LK = np.linalg.cholesky(Kuu)
    P = A / s + Kuu
    LP = np.linalg.cholesky(P)
    c = np.linalg.solve(LP, b) / s
    D = y.shape[1]
    vprod = float(np.prod([v for v, _ in thetas]))
    tr = np.trace(np.linalg.solve(Kuu, A))
    elbo = (-0.5 * N * D * np.log(2 * np.pi * s) - 0.5 * D * 2 * np.sum(np.log(np.diag(LP)))
            + 0.5 * D * 2 * np.sum(np.log(np.diag(LK))) - 0.5 * np.sum(y * y) / s + 0.5 * np.sum(c * c)
            - 0.5 * N * vprod / s + 0.5 * tr / s)
    return float(elbo), dict(A=A, b=b, Kuu=Kuu, P=P)


def _chol_extended(Ain):
    """Dense Cholesky in np.longdouble (80-bit on x86): the yardstick for ill-conditioned Kronecker cases, small grids only."""
    A = np.array(Ain, dtype=np.longdouble)
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        d = A[j, j] - np.dot(L[j, :j], L[j, :j])
        L[j, j] = np.sqrt(d)
        if j + 1 < n:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def _solve_lower_extended(L, B):
    X = np.array(B, dtype=np.longdouble)
    for j in range(L.shape[0]):
        X[j] = (X[j] - L[j, :j] @ X[:j]) / L[j, j]
    return X


def elbo_kron_extended(bases, kinds, thetas, s, X, y):
    """elbo_kron (gpr.py:260-308) with the two dense factorisations, the solves and the sums in np.longdouble; the inputs (Phi,
    Kuu factors) are the same fp64 numbers.  D = 1."""
    N = X.shape[0]
    Phis = [bs.evaluate_basis(X[:, i:i + 1]) for i, bs in enumerate(bases)]
    Kuf = make_kvs_sparse(Phis)
    A = (Kuf @ Kuf.T).toarray()
    b = np.asarray(Kuf @ y)
    Ks = [band_to_dense_sym(make_Kuu(bs, kd, v, l)) for bs, kd, (v, l) in zip(bases, kinds, thetas)]
    Kuu = np.array(Ks[0], dtype=np.longdouble)
    for K in Ks[1:]:
        Kuu = np.kron(Kuu, np.array(K, dtype=np.longdouble))
    ld = np.longdouble
    LK = _chol_extended(Kuu)
    P = np.array(A, dtype=ld) / ld(s) + Kuu
    LP = _chol_extended(P)
    c = _solve_lower_extended(LP, b) / ld(s)
    W = _solve_lower_extended(LK, A)                     # tr(Kuu^-1 A) = tr(LK^-1 A LK^-T) = sum over columns of |LK^-1 A^(1/2)|: use LK^-1 A LK^-T
    W = _solve_lower_extended(LK, W.T)
    tr = np.trace(W)
    vprod = ld(np.prod([v for v, _ in thetas]))
    elbo = (-ld(0.5) * N * np.log(2 * ld(np.pi) * ld(s)) - np.sum(np.log(np.diag(LP))) + np.sum(np.log(np.diag(LK)))
            - ld(0.5) * np.sum(np.array(y, dtype=ld) ** 2) / ld(s) + ld(0.5) * np.sum(c * c) - ld(0.5) * N * vprod / ld(s) + ld(0.5) * tr / ld(s))
    return float(elbo)


def elbo_grad_kron(bases, kinds, thetas, s, X, y):
    """The bound of elbo_kron and its analytic gradient w.r.t. [v_1, l_1, ..., v_d, l_d, s] (dense; small grids only):
    G = 1/2 (Kuu^-1 - P^-1 - alpha alpha^T - Kuu^-1 A Kuu^-1 / s) (SURVEY App. A-6 with Kuu = kron(K_i), gpr.py:282-308),
    d/dl_i = <G, K_1 x .. dK_i/dl_i .. x K_d>, d/dv_i = <G, -Kuu / v_i> - N prod(v) / (2 s v_i), d/ds as in the 1-D case.
    This is what TF autodiff returns for the reference's dense expression."""
    e, parts = elbo_kron(bases, kinds, thetas, s, X, y)
    A, b, Kuu, P = parts["A"], parts["b"], parts["Kuu"], parts["P"]
    N = X.shape[0]
    Kinv = np.linalg.inv(Kuu)
    Pinv = np.linalg.inv(P)
    alpha = Pinv @ b / s
    G = 0.5 * (Kinv - Pinv - alpha @ alpha.T - Kinv @ A @ Kinv / s)
    Ks, dKs = [], []
    for bs, kd, (v, l) in zip(bases, kinds, thetas):
        K, dK = make_Kuu(bs, kd, v, l, want_dl=True)
        Ks.append(band_to_dense_sym(K))
        dKs.append(band_to_dense_sym(dK))
    vprod = float(np.prod([v for v, _ in thetas]))
    g = []
    for i, (v, l) in enumerate(thetas):
        Kd = np.ones((1, 1))
        for j in range(len(bases)):
            Kd = np.kron(Kd, dKs[j] if j == i else Ks[j])
        g.append(float(np.sum(G * (-Kuu / v))) - 0.5 * N * vprod / (v * s))
        g.append(float(np.sum(G * Kd)))
    trKA = float(np.trace(Kinv @ A))
    yy = float(np.sum(y * y))
    g.append(float(-0.5 * N / s + 0.5 * np.sum(Pinv * A) / s ** 2 + 0.5 * yy / s ** 2 + 0.5 * (alpha.T @ A @ alpha).item() / s ** 2
                   - (b.T @ alpha).item() / s ** 2 + 0.5 * N * vprod / s ** 2 - 0.5 * trKA / s ** 2))
    return e, np.array(g)


def elbo_kron_banded(bases, kinds, thetas, s, X, y):
    """The same bound (gpr.py:282-308) without ever forming a dense M_tot x M_tot matrix, for grids of the BASELINE size
    (128 x 128): Kuf Kuf^T as a scipy sparse product, P = kron(K_1, K_2) + A / s as a LAPACK band (bandwidth k (m_2 + 1),
    scipy.linalg.cholesky_banded), log|Kuu| = m_2 log|K_1| + m_1 log|K_2|, tr(Kuu^-1 A) = <K_1^-1 x K_2^-1, A> over the
    non-zeros of A.  d = 2 only (utils.py:57 handles exactly that too)."""
    import scipy.linalg as sla
    assert len(bases) == 2 and y.shape[1] == 1
    N = X.shape[0]
    k = bases[0].order
    m1, m2 = bases[0].m, bases[1].m
    M = m1 * m2
    Phis = [bs.evaluate_basis(X[:, i:i + 1]) for i, bs in enumerate(bases)]
    Kuf = make_kvs_sparse(Phis).tocsr()
    A = (Kuf @ Kuf.T).tocoo()
    b = np.asarray(Kuf @ y).reshape(-1)
    Kd = [band_to_dense_sym(make_Kuu(bs, kd, v, l)) for bs, kd, (v, l) in zip(bases, kinds, thetas)]
    Kinv = [np.linalg.inv(K) for K in Kd]
    logdet_K = m2 * np.linalg.slogdet(Kd[0])[1] + m1 * np.linalg.slogdet(Kd[1])[1]
    i1, i2, j1, j2 = A.row // m2, A.row % m2, A.col // m2, A.col % m2
    trKA = float(np.sum(A.data * Kinv[0][i1, j1] * Kinv[1][i2, j2]))
    bw = k * (m2 + 1)
    Pb = np.zeros((bw + 1, M))                              # LAPACK lower band: Pb[i - j, j] = P[i, j]
    low = A.row >= A.col
    np.add.at(Pb, (A.row[low] - A.col[low], A.col[low]), A.data[low] / s)
    K1b, K2b = make_Kuu(bases[0], kinds[0], *thetas[0]), make_Kuu(bases[1], kinds[1], *thetas[1])
    for d1 in range(k + 1):                                 # kron(K_1, K_2)[(a + d1) m2 + (c + d2), a m2 + c] = K_1[a + d1, a] K_2[c + d2, c]
        for d2 in range(-k, k + 1):
            if d1 == 0 and d2 < 0:
                continue
            a = np.arange(m1 - d1)
            c = np.arange(max(0, -d2), min(m2, m2 - d2))
            v2 = K2b[abs(d2), np.minimum(c, c + d2)]
            col = (a[:, None] * m2 + c[None, :]).reshape(-1)
            val = (K1b[d1, a][:, None] * v2[None, :]).reshape(-1)
            Pb[d1 * m2 + d2, col] += val
    Lb = sla.cholesky_banded(Pb, lower=True)
    logdet_P = 2.0 * float(np.sum(np.log(Lb[0])))
    c = sla.solve_banded((bw, 0), Lb, b) / s
    vprod = float(np.prod([v for v, _ in thetas]))
    elbo = (-0.5 * N * np.log(2 * np.pi * s) - 0.5 * logdet_P + 0.5 * logdet_K - 0.5 * float(np.sum(y * y)) / s
            + 0.5 * float(np.sum(c * c)) - 0.5 * N * vprod / s + 0.5 * trKA / s)
    return float(elbo)


def predict_f_kron(bases, kinds, thetas, s, X, y, Xnew):
    """gpr.py:310-334: dense posterior (mean, var) of GPR_kron; var is tiled over D columns (gpr.py:331-332)."""
    _, parts = elbo_kron(bases, kinds, thetas, s, X, y)
    Kuu, P, b = parts["Kuu"], parts["P"], parts["b"]
    Kus = make_kvs_sparse([bs.evaluate_basis(Xnew[:, i:i + 1]) for i, bs in enumerate(bases)]).toarray()
    LP = np.linalg.cholesky(P)
    tmp = np.linalg.solve(LP, Kus)
    mean = tmp.T @ (np.linalg.solve(LP, b) / s)
    vprod = float(np.prod([v for v, _ in thetas]))
    var = vprod + np.sum(tmp * tmp, axis=0) - np.sum(Kus * np.linalg.solve(Kuu, Kus), axis=0)
    return mean, np.tile(var.reshape(-1, 1), (1, y.shape[1]))

# --------------------------------------------------------------------------------------
# Additive model (gpr.py:139-236): Kuf = vstack(Kuf_i), Kuu = blockdiag(Kuu_i), dense algebra as the reference
# --------------------------------------------------------------------------------------

def additive_stats(bases, X, y):
    """gpr.py:167-172: Kuf = sparse.vstack(Kuf_i); KufKfu = (Kuf @ Kuf.T).todense(); Kuf_y = Kuf @ y."""
    import scipy.sparse as sp
    Kuf = sp.vstack([bs.evaluate_basis(X[:, i:i + 1]) for i, bs in enumerate(bases)]).tocsr()
    return (Kuf @ Kuf.T).toarray(), np.asarray(Kuf @ y), float(np.sum(np.square(y)))


def _blockdiag(mats):
    n = sum(m.shape[0] for m in mats)
    out = np.zeros((n, n))
    o = 0
    for m in mats:
        out[o:o + m.shape[0], o:o + m.shape[0]] = m
        o += m.shape[0]
    return out


def elbo_additive(bases, kinds, thetas, s, X, y):
    """gpr.py:177-209: sum K_diag = N * sum v_i (gpr.py:181,207); log|Kuu| of the block-diagonal operator (gpr.py:187);
    dense Cholesky of P (gpr.py:191-194); tr(Kuu^-1 KufKfu) (gpr.py:208)."""
    N, D = X.shape[0], y.shape[1]
    A, b, yy = additive_stats(bases, X, y)
    Kuu = _blockdiag([band_to_dense_sym(make_Kuu(bs, kd, v, l)) for bs, kd, (v, l) in zip(bases, kinds, thetas)])
    sign, logdetK = np.linalg.slogdet(Kuu)
    P = Kuu + A / s
    LP = np.linalg.cholesky(P)
    c = np.linalg.solve(LP, b) / s
    vsum = float(np.sum([v for v, _ in thetas]))
    elbo = (-0.5 * N * D * np.log(2 * np.pi * s) - 0.5 * D * np.sum(np.log(np.square(np.diag(LP)))) + 0.5 * D * logdetK
            - 0.5 * yy / s + 0.5 * np.sum(np.square(c)) - 0.5 * N * D * vsum / s
            + 0.5 * D * np.trace(np.linalg.solve(Kuu, A)) / s)
    return float(elbo), dict(A=A, b=b, Kuu=Kuu, P=P, LP=LP, c=c)


def predict_f_additive(bases, kinds, thetas, s, X, y, Xnew):
    """gpr.py:211-236."""
    import scipy.sparse as sp
    _, parts = elbo_additive(bases, kinds, thetas, s, X, y)
    Kus = sp.vstack([bs.evaluate_basis(Xnew[:, i:i + 1]) for i, bs in enumerate(bases)]).toarray()
    tmp = np.linalg.solve(parts["LP"], Kus)
    mean = tmp.T @ parts["c"]
    var = float(np.sum([v for v, _ in thetas])) + np.sum(np.square(tmp), 0) - np.sum(np.linalg.solve(parts["Kuu"], Kus) * Kus, 0)
    return mean, np.tile(var.reshape(-1, 1), (1, y.shape[1]))


# --------------------------------------------------------------------------------------
# Parameter transforms + L-BFGS-B driver (example.py:28-33; GPflow Scipy optimiser / softplus / 1e-6 shift)
# --------------------------------------------------------------------------------------

def softplus(u):
    return np.logaddexp(0.0, u)


def softplus_inv(x):
    return x + np.log(-np.expm1(-x))


def sigmoid(u):
    return 1.0 / (1.0 + np.exp(-u))


LIK_SHIFT = 1e-6  # gpflow.likelihoods.Gaussian variance lower bound


def fit_1d(basis, kind, X, y, v0=1.0, l0=1.0, s0=1.0, maxiter=15000):
    """example.py:31-32: minimise -ELBO over unconstrained (v, l, s) with scipy L-BFGS-B (what
    gpflow.optimizers.Scipy calls), from GPflow's defaults."""
    from scipy.optimize import minimize
    A, b, yy = sufficient_stats(basis, X, y)
    N = X.shape[0]

    def fun(u):
        v, l, s = softplus(u[0]), softplus(u[1]), softplus(u[2]) + LIK_SHIFT
        e, g, _ = elbo_grad_1d(basis, kind, A, b, yy, N, v, l, s)
        return -e, -g * sigmoid(u)

    u0 = np.array([softplus_inv(v0), softplus_inv(l0), softplus_inv(s0 - LIK_SHIFT)])
    res = minimize(fun, u0, jac=True, method="L-BFGS-B", options=dict(maxiter=maxiter))
    theta = np.array([softplus(res.x[0]), softplus(res.x[1]), softplus(res.x[2]) + LIK_SHIFT])
    return -res.fun, theta, res
