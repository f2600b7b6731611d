#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *reference itself*.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  The reference's pure-Python modules (asvgp/basis.py,
inducing_features.py, utils.py, kronecker.py) use ~25 elementwise `tf.*` symbols;
they import unchanged when a throw-away numpy module named `tensorflow` (plus
empty `banded_matrices.banded`, a 3-class `gpflow.kernels` stub) is first on
sys.path.  gpr.py cannot be imported (real GPflow/CHOLMOD/
banded_matrices needed), so ELBO-level goldens are (a) the notebook printouts
and (b) a dense-fp64 textbook evaluation driven by the reference's own Phi/Kuu.

Outputs (data only - inputs and expected outputs, no reference source text):
    basis_fixtures.npz      mesh/delta, Phi (dense + CSR), static bands B1..B6
    kuu_fixtures.npz        make_Kuu bands for Matern12/32/52
    kron_fixtures.npz       Khatri-Rao Phi for a 12x14 grid
    snelson_fixtures.npz    Snelson data + PhiPhi^T band, Phi y, y^T y + dense ELBO table
    snelson/{train_inputs,train_outputs,test_inputs}   (data files of the reference's example)

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import shutil
import tempfile
import textwrap

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

TF_STANDIN = textwrap.dedent('''
    """numpy stand-in for the handful of tf symbols the reference's basis code uses."""
    import numpy as _np
    float64 = _np.float64
    int64 = _np.int64
    Tensor = _np.ndarray

    def cast(x, dtype=None):
        return _np.asarray(x, dtype=dtype)

    def linspace(start, stop, num):
        # tf.linspace infers float32 from python-float endpoints (SURVEY App. B-1)
        if type(start) is float or type(stop) is float:
            f = _np.float32
            s, e = f(start), f(stop)
            delta = f((e - s) / f(num - 1))
            out = (s + delta * _np.arange(num, dtype=f)).astype(f)
            out[-1] = e
            return out
        return _np.linspace(_np.float64(start), _np.float64(stop), int(num))

    def repeat(x, n):
        return _np.repeat(_np.asarray(x), int(n))

    def zeros(shape, dtype=_np.float64):
        return _np.zeros(shape, dtype=dtype)

    def ones(shape, dtype=_np.float64):
        return _np.ones(shape, dtype=dtype)

    def concat(xs, axis=0):
        return _np.concatenate([_np.atleast_1d(_np.asarray(x)) for x in xs], axis=axis)

    def stack(xs, axis=0):
        return _np.stack([_np.asarray(x) for x in xs], axis=axis)

    def reshape(x, shape):
        return _np.reshape(_np.asarray(x), shape)

    def searchsorted(a, v, side="left"):
        return _np.searchsorted(_np.asarray(a), _np.asarray(v), side=side)

    def gather(a, idx):
        return _np.asarray(a)[_np.asarray(idx)]

    def tile(x, reps):
        return _np.tile(_np.asarray(x), _np.asarray(reps))

    def range(n):
        return _np.arange(n)

    def constant(x, dtype=None):
        return _np.asarray(x, dtype=dtype)

    def transpose(x):
        return _np.asarray(x).T

    def scatter_nd(indices, updates, shape):
        out = _np.zeros(tuple(int(s) for s in shape), dtype=_np.asarray(updates).dtype)
        _np.add.at(out, tuple(_np.asarray(indices).T), _np.asarray(updates))
        return out

    def reverse(x, axis):
        return _np.flip(_np.asarray(x), axis=tuple(axis))

    class nn:
        @staticmethod
        def relu(x):
            return _np.maximum(_np.asarray(x), 0)

    class math:
        @staticmethod
        def cumsum(x):
            return _np.cumsum(_np.asarray(x))
        @staticmethod
        def reduce_sum(x):
            return _np.sum(_np.asarray(x))

    class linalg:
        @staticmethod
        def diag(v, k=0):
            return _np.diag(_np.asarray(v), k=k)
        @staticmethod
        def diag_part(m, k=0):
            return _np.diagonal(_np.asarray(m), offset=k).copy()
''')

GPFLOW_STANDIN = textwrap.dedent('''
    class _K:
        def __init__(self, variance=1.0, lengthscales=1.0):
            self.variance = variance
            self.lengthscales = lengthscales
    class kernels:
        class Matern12(_K): pass
        class Matern32(_K): pass
        class Matern52(_K): pass
''')


def _import_reference():
    tmp = tempfile.mkdtemp(prefix="asvgp_standin_")
    with open(os.path.join(tmp, "tensorflow.py"), "w") as f:
        f.write(TF_STANDIN)
    with open(os.path.join(tmp, "gpflow.py"), "w") as f:
        f.write(GPFLOW_STANDIN)
    os.makedirs(os.path.join(tmp, "banded_matrices"))
    open(os.path.join(tmp, "banded_matrices", "__init__.py"), "w").close()
    open(os.path.join(tmp, "banded_matrices", "banded.py"), "w").close()
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, tmp)
    import asvgp.basis as rbasis
    import asvgp.inducing_features as rfeat
    import asvgp.utils as rutils
    import asvgp.kronecker as rkron
    import gpflow as gstub
    return tmp, rbasis, rfeat, rutils, rkron, gstub


def _probe_points(basis, n_random, rng):
    """random + adversarial points: a, b, every 7th knot, +-1 ulp around knots."""
    mesh = np.asarray(basis.mesh, dtype=np.float64)
    a, b = float(mesh[0]), float(mesh[-1])
    pts = [rng.uniform(a, b, size=n_random)]
    knots = mesh[::7]
    pts.append(knots)
    pts.append(np.nextafter(knots, np.inf))
    pts.append(np.nextafter(knots, -np.inf))
    pts.append(np.array([a, b, np.nextafter(a, np.inf), np.nextafter(b, -np.inf)]))
    x = np.concatenate(pts)
    x = np.clip(x, a, b)
    return x


def _dense_elbo(Phi, y, Kuu_band, theta_v, theta_s):
    """Textbook dense evaluation of the collapsed bound (gpr.py:78-87) from the
    reference's own Phi (dense M x N) and Kuu lower band."""
    M, N = Phi.shape
    k = Kuu_band.shape[0] - 1
    Kuu = np.zeros((M, M))
    for d in range(k + 1):
        v = Kuu_band[d, :M - d]
        Kuu += np.diag(v, -d)
        if d:
            Kuu += np.diag(v, d)
    A = Phi @ Phi.T
    b = Phi @ y
    s = theta_s
    P = Kuu + A / s
    LP = np.linalg.cholesky(P)
    LK = np.linalg.cholesky(Kuu)
    c = np.linalg.solve(LP, b) / s
    logdetP = 2 * np.sum(np.log(np.diag(LP)))
    logdetK = 2 * np.sum(np.log(np.diag(LK)))
    tr = np.trace(np.linalg.solve(Kuu, A))
    D = y.shape[1]
    elbo = (-0.5 * N * D * np.log(2 * np.pi * s) - 0.5 * D * logdetP + 0.5 * D * logdetK
            - 0.5 * np.sum(y * y) / s + 0.5 * np.sum(c * c) - 0.5 * N * theta_v / s + 0.5 * tr / s)
    return float(elbo)


def main():
    tmp, rbasis, rfeat, rutils, rkron, gstub = _import_reference()
    rng = np.random.default_rng(20261003)
    try:
        # ------------------------------------------------------------------ basis
        out = {}
        specs = [  # (tag, order, a, b, m)
            ("B1_f", 1, -3.5, 10.5, 20), ("B2_f", 2, -3.5, 10.5, 20), ("B3_f", 3, -3.5, 10.5, 30),
            ("B3_f100", 3, -3.5, 10.5, 100), ("B4_f", 4, -3.5, 10.5, 30), ("B5_f", 5, -1.25, 2.5, 24),
            ("B6_f", 6, 0.0, 1.0, 28),
            ("B1_i", 1, 0, 1, 16), ("B2_i", 2, 0, 1, 17), ("B3_i", 3, 0, 1, 24), ("B4_i", 4, 0, 1, 64),
            ("B5_i", 5, -2, 3, 33), ("B6_i", 6, 0, 1, 40), ("B4_i1024", 4, 0, 1, 1024),
        ]
        tags = []
        for tag, order, a, b, m in specs:
            cls = getattr(rbasis, "B%dSpline" % order)
            bs = cls(a, b, m)
            tags.append(tag)
            out[tag + "/spec"] = np.array([order, float(a), float(b), m, 1.0 if type(a) is float else 0.0])
            out[tag + "/mesh"] = np.asarray(bs.mesh, dtype=np.float64)
            out[tag + "/delta"] = np.float64(bs.delta)
            for nm in ("A", "B", "C", "D", "BC", "BC_grad", "BC_ggrad", "BC_ggrad_none", "BC_none_ggrad"):
                if hasattr(bs, nm) and m <= 100:
                    out[tag + "/" + nm] = np.asarray(getattr(bs, nm), dtype=np.float64)
            x = _probe_points(bs, 200 if m <= 100 else 2000, rng)
            Phi = bs.evaluate_basis(x.reshape(-1, 1), dx=0, sparse=True)
            Phi.sort_indices()
            out[tag + "/x"] = x
            out[tag + "/csr_data"] = Phi.data.astype(np.float64)
            out[tag + "/csr_indices"] = Phi.indices.astype(np.int64)
            out[tag + "/csr_indptr"] = Phi.indptr.astype(np.int64)
            # neighbour index exactly as basis.py:58 computes it
            out[tag + "/idx"] = np.maximum(np.searchsorted(np.asarray(bs.mesh), x, side="left") - 1, 0).astype(np.int64)
        out["tags"] = np.array(tags)
        np.savez_compressed(os.path.join(HERE, "basis_fixtures.npz"), **out)

        # ------------------------------------------------------------------ Kuu
        kout = {}
        thetas = [(1.0, 1.0), (0.8, 1.03), (2.5, 0.05)]
        for tag, order, a, b, m in [("B3_f", 3, -3.5, 10.5, 30), ("B4_i", 4, 0, 1, 64), ("B2_i", 2, 0, 1, 17),
                                    ("B5_i", 5, -2, 3, 33), ("B1_i", 1, 0, 1, 16), ("B6_i", 6, 0, 1, 40)]:
            bs = getattr(rbasis, "B%dSpline" % order)(a, b, m)
            for kname in ("Matern12", "Matern32", "Matern52"):
                if kname == "Matern32" and order < 2:
                    continue
                if kname == "Matern52" and order not in (3, 4, 5):
                    continue
                for ti, (v, l) in enumerate(thetas):
                    kern = getattr(gstub.kernels, kname)(variance=v, lengthscales=l)
                    feat = rfeat.SplineFeatures1D(kern, bs)
                    kout["%s/%s/%d" % (tag, kname, ti)] = np.asarray(feat.make_Kuu(kern), dtype=np.float64)
        kout["thetas"] = np.array(thetas)
        np.savez_compressed(os.path.join(HERE, "kuu_fixtures.npz"), **kout)

        # ------------------------------------------------------------------ Khatri-Rao
        b1 = rbasis.B3Spline(0, 1, 12)
        b2 = rbasis.B3Spline(-1, 2, 14)
        X2 = np.stack([rng.uniform(0, 1, 64), rng.uniform(-1, 2, 64)], axis=1)
        P1 = b1.evaluate_basis(X2[:, :1])
        P2 = b2.evaluate_basis(X2[:, 1:])
        KR = rkron.make_kvs_sparse([P1, P2]).tocsr()
        KR.sort_indices()
        KR.eliminate_zeros()
        np.savez_compressed(os.path.join(HERE, "kron_fixtures.npz"), X=X2, m=np.array([12, 14]),
                            ab=np.array([[0, 1], [-1, 2]], dtype=np.float64), order=np.array([3, 3]),
                            dense=KR.toarray(), KKt=(KR @ KR.T).toarray())

        # ------------------------------------------------------------------ Snelson
        sdir = os.path.join(HERE, "snelson")
        os.makedirs(sdir, exist_ok=True)
        for fn in ("train_inputs", "train_outputs", "test_inputs"):
            shutil.copyfile(os.path.join(REF, "experiments/snelson/data", fn), os.path.join(sdir, fn))
        X = np.loadtxt(os.path.join(sdir, "train_inputs")).reshape(-1, 1)
        Y = np.loadtxt(os.path.join(sdir, "train_outputs")).reshape(-1, 1)
        sout = {"X": X, "Y": Y, "golden_elbo_asvgp": np.float64(-60.8356263428725),
                "golden_elbo_gp": np.float64(-60.573988814770104)}
        rows = []
        for tag, order, m in [("B3_30", 3, 30), ("B3_100", 3, 100), ("B4_30", 4, 30)]:
            bs = getattr(rbasis, "B%dSpline" % order)(-3.5, 10.5, m)
            Kuf = bs.evaluate_basis(X, dx=0, sparse=True)
            sout[tag + "/Kuf_y"] = np.asarray(Kuf @ Y)
            KK = Kuf @ Kuf.T
            sout[tag + "/KufKfu"] = np.asarray(rutils.sparse_to_band(KK, order), dtype=np.float64)
            sout[tag + "/tr_yTy"] = np.float64(np.sum(np.square(Y)))
            Phi = Kuf.toarray()
            for kname in ("Matern12", "Matern32", "Matern52"):
                for (v, l, s) in [(1.0, 1.0, 1.0), (0.8, 1.03, 0.08)]:
                    kern = getattr(gstub.kernels, kname)(variance=v, lengthscales=l)
                    Kb = np.asarray(rfeat.SplineFeatures1D(kern, bs).make_Kuu(kern), dtype=np.float64)
                    e = _dense_elbo(Phi, Y, Kb, v, s)
                    rows.append((order, m, {"Matern12": 0, "Matern32": 1, "Matern52": 2}[kname], v, l, s, e))
        sout["elbo_table"] = np.array(rows)  # order, m, kernel(0/1/2), v, l, s, dense ELBO
        np.savez_compressed(os.path.join(HERE, "snelson_fixtures.npz"), **sout)
        print("wrote fixtures to", HERE)
        for r in rows[:6]:
            print(r)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
