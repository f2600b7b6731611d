"""GPU parity: the HIP path (through the C-ABI, via asvgp_amd) against the CPU oracle and the golden fixtures.
Tolerances: indices/structure bit-exact; band/rhs/yy rel 1e-12; ELBO rel 1e-9; gradient rel 1e-6;
posterior abs 1e-8."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import asvgp_oracle as O

pytestmark = pytest.mark.gpu

KINDS = {0: "Matern12", 1: "Matern32", 2: "Matern52"}


def elbo_tol(e, N, v, s, yy, bcr=False):
    """fp64 tolerance for the bound: rel 1e-9 of the value plus a fraction of the two large cancelling
    terms N v/(2s), y^T y/(2s) the bound is a difference of (gpr.py:81-87): 2e-11 for the sequential column sweeps
    (the reference's elimination order), 5e-10 for block cyclic reduction (odd-even order: same factorisation,
    different rounding).  Calibration (DESIGN.md "Numerics"): on the Matern-5/2 / B4 / M=256 case below,
    cond(Kuu) = 1.1e7, against an 80-bit long-double evaluation the oracle is off by 1e-6, the sequential HIP
    sweep by ~1e-6 and BCR by 3e-5, on terms of size 1e5; two fp64 orders of the oracle itself (banded vs
    dense) differ by 9e-10 relative."""
    return 1e-9 * abs(e) + (5e-10 if bcr else 2e-11) * (0.5 * N * v / s + 0.5 * yy / s)


@pytest.fixture(scope="module")
def A():
    import asvgp_amd
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from asvgp_amd import _lib
    _lib.get_lib()  # must load: no fallback
    return asvgp_amd


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _reload_env():
    from asvgp_amd import _lib
    _lib.get_lib().asvgp_debug_reload_env()       # the library reads its debug switches once; tests that flip them say so


def _spec(F, tag):
    order, a, b, m, isf = F[tag + "/spec"]
    order, m = int(order), int(m)
    a, b = (float(a), float(b)) if isf else (int(a), int(b))
    return order, a, b, m


def _mk_basis(A, order, a, b, m):
    return getattr(A, "B%dSpline" % order)(a, b, m)


def _kernel(A, kind, v, l):
    return getattr(A, KINDS[kind])(variance=v, lengthscales=l)


# ------------------------------------------------------------------------------------------------ basis / Phi
def test_index_and_design_matrix_vs_reference_fixtures(A, golden_dir):
    F = np.load(os.path.join(golden_dir, "basis_fixtures.npz"))
    for tag in F["tags"]:
        order, a, b, m = _spec(F, tag)
        bs = _mk_basis(A, order, a, b, m)
        assert np.array_equal(bs.mesh.cpu().numpy(), F[tag + "/mesh"]), tag
        x = F[tag + "/x"]
        idx = bs.neighbour_index(dev(x)).cpu().numpy()
        assert np.array_equal(idx, F[tag + "/idx"]), tag                      # integer work: bit exact
        rows, cols, data = bs.evaluate_basis_coo(dev(x))
        orow, ocol, odata = O.evaluate_basis_coo(F[tag + "/mesh"], float(F[tag + "/delta"]), order, m, x)
        assert np.array_equal(rows.cpu().numpy(), orow), tag
        assert np.array_equal(cols.cpu().numpy(), ocol), tag
        np.testing.assert_allclose(data.cpu().numpy(), odata, rtol=0, atol=2e-14, err_msg=tag)
        csr = bs.evaluate_basis(dev(x).reshape(-1, 1))
        dense = csr.to_dense().cpu().numpy()
        import scipy.sparse as sp
        ref = sp.csr_matrix((F[tag + "/csr_data"], F[tag + "/csr_indices"], F[tag + "/csr_indptr"]), shape=(m, x.shape[0]))
        np.testing.assert_allclose(dense, ref.toarray(), rtol=0, atol=2e-14, err_msg=tag)


def test_basis_derivatives_vs_oracle(A):
    rng = np.random.default_rng(5)
    for order in range(1, 7):
        bs = _mk_basis(A, order, 0, 2, 30)
        ob = O.Basis(order, 0, 2, 30)
        x = rng.uniform(0, 2, 257)
        for dx in range(0, min(order, 3) + 1):
            rows, cols, data = bs.evaluate_basis_coo(dev(x), dx=dx)
            orow, ocol, odata = O.evaluate_basis_coo(ob.mesh, ob.delta, order, 30, x, dx)
            assert np.array_equal(rows.cpu().numpy(), orow)
            np.testing.assert_allclose(data.cpu().numpy(), odata, rtol=1e-12, atol=1e-12 * np.max(np.abs(odata)))


def test_static_bands_and_kuu_bit_level(A, golden_dir):
    F = np.load(os.path.join(golden_dir, "basis_fixtures.npz"))
    for tag in F["tags"]:
        order, a, b, m = _spec(F, tag)
        bs = _mk_basis(A, order, a, b, m)
        for nm in ("A", "B", "C", "D", "BC", "BC_grad", "BC_ggrad", "BC_ggrad_none", "BC_none_ggrad"):
            if tag + "/" + nm in F:
                ref = F[tag + "/" + nm]
                mine = getattr(bs, nm).cpu().numpy()
                assert np.max(np.abs(ref - mine)) <= 4e-15 * max(np.max(np.abs(ref)), 1e-300), (tag, nm)
    K = np.load(os.path.join(golden_dir, "kuu_fixtures.npz"))
    specs = {"B3_f": (3, -3.5, 10.5, 30), "B4_i": (4, 0, 1, 64), "B2_i": (2, 0, 1, 17), "B5_i": (5, -2, 3, 33),
             "B1_i": (1, 0, 1, 16), "B6_i": (6, 0, 1, 40)}
    kid = {"Matern12": 0, "Matern32": 1, "Matern52": 2}
    for key in K.files:
        if key == "thetas":
            continue
        tag, kn, ti = key.split("/")
        v, l = K["thetas"][int(ti)]
        bs = _mk_basis(A, *specs[tag])
        kern = _kernel(A, kid[kn], v, l)
        feat = A.SplineFeatures1D(kern, bs)
        Kuu, dK = feat.make_Kuu(kern, with_dl=True)
        ref = K[key]
        assert np.max(np.abs(Kuu.cpu().numpy() - ref)) <= 4e-15 * np.max(np.abs(ref)), key
        _, odK = O.make_Kuu(O.Basis(*specs[tag]), kid[kn], v, l, want_dl=True)
        np.testing.assert_allclose(dK.cpu().numpy(), odK, rtol=1e-13, atol=1e-13 * np.max(np.abs(odK)))
    with pytest.raises(AttributeError):
        b1 = _mk_basis(A, 1, 0, 1, 16)
        A.SplineFeatures1D(A.Matern32(), b1).make_Kuu(A.Matern32())


@pytest.mark.parametrize("order,M,N", [(1, 16, 1000), (2, 17, 999), (3, 30, 4097), (4, 64, 20000), (5, 33, 3001),
                                        (6, 40, 2048), (4, 1024, 50001)])
def test_phi_accumulate_vs_oracle(A, order, M, N):
    rng = np.random.default_rng(order * 1000 + M)
    a, b = (0, 1) if order != 5 else (-2, 3)
    bs = _mk_basis(A, order, a, b, M)
    ob = O.Basis(order, a, b, M)
    x = rng.uniform(a + 1e-9, b - 1e-9, N)
    x[:8] = ob.mesh[1:9]                      # points exactly on knots
    y = np.sin(20 * x) + 0.1 * rng.normal(size=N)
    model = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern12(), bs)
    band, rhs, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
    got = model.KufKfu.cpu().numpy()
    sc = np.max(np.abs(band))
    assert np.max(np.abs(got - band)) <= 1e-12 * sc
    assert np.array_equal(got == 0, band == 0)            # structural zeros of the right-padded band
    np.testing.assert_allclose(model.Kuf_y.cpu().numpy(), rhs, rtol=0, atol=1e-12 * np.max(np.abs(rhs)))
    assert abs(model.tr_yTy.item() - yy) <= 1e-12 * yy
    if N <= 5000:  # the CSR/SpGEMM route the reference takes gives the same numbers
        band2, rhs2, _ = O.sufficient_stats(ob, x.reshape(-1, 1), y.reshape(-1, 1))
        assert np.max(np.abs(got - band2)) <= 1e-12 * sc


def test_phi_edge_cases(A):
    bs = _mk_basis(A, 3, 0, 1, 24)
    ob = O.Basis(3, 0, 1, 24)
    # empty input
    m0 = A.GPR_1d((np.zeros((0, 1)), np.zeros((0, 1))), A.Matern12(), bs)
    assert float(m0._stats.abs().sum()) == 0.0
    # single point, odd counts, unaligned (sliced) inputs, two output columns
    rng = np.random.default_rng(1)
    for N in (1, 2, 3, 2047, 2049):
        x = rng.uniform(0.001, 0.999, N + 1)
        y = rng.normal(size=(N + 1, 2))
        xs = dev(x)[1:].reshape(-1, 1)            # 8-byte-aligned only -> scalar load path
        m = A.GPR_1d((xs, dev(y)[1:]), A.Matern12(), bs)
        band, rhs, yy = O.sufficient_stats_direct(ob, x[1:], y[1:])
        assert np.max(np.abs(m.KufKfu.cpu().numpy() - band)) <= 1e-12 * np.max(np.abs(band))
        np.testing.assert_allclose(m.Kuf_y.cpu().numpy(), rhs, rtol=0, atol=1e-12 * max(np.max(np.abs(rhs)), 1e-300))
        assert abs(m.tr_yTy.item() - yy) <= 1e-12 * yy
    with pytest.raises(AssertionError):
        A.GPR_1d((np.array([[0.0], [0.5]]), np.zeros((2, 1))), A.Matern12(), bs)   # gpr.py:25: X > a strictly
    with pytest.raises(AssertionError):
        A.GPR_1d((np.array([[0.5, 0.5]]), np.zeros((1, 1))), A.Matern12(), bs)     # gpr.py:23


def test_phi_full_size_properties(A):
    """BASELINE config 2 size (N=1M, M=1024, k=4): size-independent properties instead of an O(N) oracle run."""
    N, M = 1_000_000, 1024
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * (1 - 2e-9) + 1e-9
    y = torch.sin(20 * x) + 0.1 * torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
    bs = A.B4Spline(0, 1, M)
    m = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern32(), bs)
    band = m.KufKfu
    # partition of unity: 1^T Phi Phi^T 1 = N and 1^T Phi y = sum y
    tot = band[0].sum() + 2 * band[1:].sum()
    assert abs(tot.item() - N) <= 1e-9 * N
    assert abs(m.Kuf_y.sum().item() - y.sum().item()) <= 1e-9 * y.abs().sum().item()
    assert abs(m.tr_yTy.item() - (y * y).sum().item()) <= 1e-12 * (y * y).sum().item()
    # diagonal dominance of Phi Phi^T entries: band[0] >= 0, |band[d,j]| <= sqrt(band[0,j] band[0,j+d])
    assert (band[0] >= 0).all()
    for d in range(1, 5):
        assert (band[d, :M - d].abs() <= torch.sqrt(band[0, :M - d] * band[0, d:]) * (1 + 1e-12)).all()
        assert (band[d, M - d:] == 0).all()
    # linearity over shards: stats(all) = stats(first half) + stats(second half)
    h = N // 2
    m1 = A.GPR_1d((x[:h].reshape(-1, 1), y[:h].reshape(-1, 1)), A.Matern32(), bs)
    m2 = A.GPR_1d((x[h:].reshape(-1, 1), y[h:].reshape(-1, 1)), A.Matern32(), bs)
    s12 = m1._stats + m2._stats
    assert (s12 - m._stats).abs().max().item() <= 1e-11 * m._stats.abs().max().item()
    # sorted input (time-series order) gives the same statistics
    xs, order = torch.sort(x)
    m3 = A.GPR_1d((xs.reshape(-1, 1), y[order].reshape(-1, 1)), A.Matern32(), bs)
    assert (m3._stats - m._stats).abs().max().item() <= 1e-11 * m._stats.abs().max().item()
    # and on a 20k subsample the oracle agrees
    ob = O.Basis(4, 0, 1, M)
    sub = slice(0, 20000)
    ms = A.GPR_1d((x[sub].reshape(-1, 1), y[sub].reshape(-1, 1)), A.Matern32(), bs)
    ob_band, ob_rhs, _ = O.sufficient_stats_direct(ob, x[sub].cpu().numpy(), y[sub].cpu().numpy().reshape(-1, 1))
    assert np.max(np.abs(ms.KufKfu.cpu().numpy() - ob_band)) <= 1e-12 * np.max(np.abs(ob_band))


def test_phi_large_M_column_chunks(A):
    """M = 4096, k = 4 (BASELINE config 3 shape) exceeds one LDS image: column-chunked passes must agree."""
    rng = np.random.default_rng(11)
    N, M = 30000, 4096
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = rng.normal(size=(N, 1))
    bs = A.B4Spline(0, 1, M)
    m = A.GPR_1d((x.reshape(-1, 1), y), A.Matern52(), bs)
    band, rhs, yy = O.sufficient_stats_direct(O.Basis(4, 0, 1, M), x, y)
    assert np.max(np.abs(m.KufKfu.cpu().numpy() - band)) <= 1e-12 * np.max(np.abs(band))
    np.testing.assert_allclose(m.Kuf_y.cpu().numpy(), rhs, rtol=0, atol=1e-12 * np.max(np.abs(rhs)))
    assert abs(m.tr_yTy.item() - yy) <= 1e-12 * yy


# ------------------------------------------------------------------------------------------------ banded ops
def _spd_band(rng, k, M):
    Bm = rng.normal(size=(M, M))
    dense = Bm @ Bm.T
    dense = np.triu(np.tril(dense, k), -k)
    dense += np.eye(M) * (np.abs(dense).sum(1).max() + 1.0)
    return dense, O.pack_dense_matrix_to_banded(dense, k, 0)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8])
def test_banded_operator_api_vs_oracle(A, k):
    from asvgp_amd import banded
    rng = np.random.default_rng(k)
    for M in (k + 1, 9, 63, 64, 65, 130, 257, 1000):
        if M <= k:
            continue
        dense, lower = _spd_band(rng, k, M)
        L = banded.cholesky_band(dev(lower))
        oL = O.cholesky_band(lower)
        np.testing.assert_allclose(L.cpu().numpy(), oL, rtol=0, atol=1e-12 * np.max(np.abs(oL)), err_msg="M=%d" % M)
        assert np.array_equal(L.cpu().numpy() == 0, oL == 0)
        S = banded.inverse_from_cholesky_band(L)
        oS = O.inverse_from_cholesky_band(oL)
        np.testing.assert_allclose(S.cpu().numpy(), oS, rtol=0, atol=1e-12 * np.max(np.abs(oS)), err_msg="M=%d" % M)
        for D in (1, 3):
            rhs = rng.normal(size=(M, D))
            np.testing.assert_allclose(banded.solve_triang_mat(L, dev(rhs)).cpu().numpy(), O.solve_triang_mat(oL, rhs),
                                       rtol=0, atol=1e-11)
            np.testing.assert_allclose(banded.solve_triang_mat(L, dev(rhs), transpose_left=True).cpu().numpy(),
                                       O.solve_triang_mat(oL, rhs, True), rtol=0, atol=1e-11)
        sym = banded.symmetrise_band(dev(lower), k)
        assert np.array_equal(sym.cpu().numpy(), O.symmetrise_band(lower, k))            # index shuffles: bit exact
        assert np.array_equal(A.utils.symmetrise_banded(dev(lower)).cpu().numpy(), O.symmetrise_band(lower, k))
        assert np.array_equal(banded.unpack_banded_matrix_to_dense(sym, k, k).cpu().numpy(), dense)
        assert np.array_equal(banded.pack_dense_matrix_to_banded(dev(dense), k, 0).cpu().numpy(), lower)
        assert np.array_equal(banded.transpose_band(dev(lower), k, 0).cpu().numpy(), O.transpose_band(lower, k, 0))
        if M <= 130:
            prod = banded.product_band_band(sym, sym, left_lower_bandwidth=k, left_upper_bandwidth=k,
                                            right_lower_bandwidth=k, right_upper_bandwidth=k,
                                            result_lower_bandwidth=0, result_upper_bandwidth=0)
            np.testing.assert_allclose(prod.cpu().numpy(), O.product_band_band(O.symmetrise_band(lower, k),
                                       O.symmetrise_band(lower, k), k, k, k, k, 0, 0), rtol=1e-13)
            g = np.triu(np.tril(rng.normal(size=(M, M)), 2), -1)   # lower bw 1, upper bw 2
            gb = O.pack_dense_matrix_to_banded(g, 1, 2)
            assert np.array_equal(banded.transpose_band(dev(gb), 1, 2).cpu().numpy(), O.transpose_band(gb, 1, 2))
            pr = banded.product_band_band(dev(gb), sym, 1, 2, k, k, 2, 1)
            np.testing.assert_allclose(pr.cpu().numpy(), O.product_band_band(gb, O.symmetrise_band(lower, k), 1, 2, k, k, 2, 1),
                                       rtol=1e-12, atol=1e-12)
        tr = banded.band_trace_sym(S, dev(lower)).item()
        assert abs(tr - O.band_sym_dot(oS, lower)) <= 1e-10 * abs(tr)


def test_cholesky_not_positive_definite_reports_column(A):
    from asvgp_amd import banded
    bad = np.array([[4.0, 1.0, -3.0, 2.0], [1.0, 0.5, 0.2, 0.0]])
    with pytest.raises(banded.NotPositiveDefiniteError) as ei:
        banded.cholesky_band(dev(bad))
    assert "column 2" in str(ei.value)
    with pytest.raises(np.linalg.LinAlgError):
        O.cholesky_band(bad)


@pytest.mark.parametrize("M,k,col", [(500, 4, 137), (500, 4, 0), (500, 4, 499), (5000, 4, 4321), (3000, 2, 2999), (1300, 8, 650)])
def test_cholesky_not_positive_definite_column_of_the_streamed_kernel(A, M, k, col):
    """The lane-uniform kernel carries no pivot test in its sweep: a pivot that is not > 0 turns the rest of the diagonal into NaN and the
    first such column is found when the segments are stored.  Same report as the reference's op (LinAlgError at the first failing pivot):
    a negative pivot, a NaN entry, first / interior / last column, one segment and several."""
    from asvgp_amd import banded
    rng = np.random.default_rng(M + k + col)
    lower = np.zeros((k + 1, M))
    for d in range(k + 1):
        lower[d, :M - d] = rng.normal(size=M - d) * (0.3 ** d)
    dom = np.sum(np.abs(lower[1:]), axis=0)
    for d in range(1, k + 1):
        dom[d:] += np.abs(lower[d, :M - d])
    lower[0] = np.abs(lower[0]) + dom + 0.5
    for poison in (-1.0, float("nan")):
        bad = lower.copy()
        bad[0, col] = poison
        with pytest.raises(banded.NotPositiveDefiniteError) as ei:
            banded.cholesky_band(dev(bad))
        assert "column %d" % col in str(ei.value), str(ei.value)
    L = banded.cholesky_band(dev(lower)).cpu().numpy()           # and the untouched matrix factors
    assert np.max(np.abs(L - O.cholesky_band(lower))) <= 1e-13 * np.max(np.abs(L))


# ------------------------------------------------------------------------------------------------ ELBO / gradient
@pytest.fixture(scope="module")
def S(golden_dir):
    return np.load(os.path.join(golden_dir, "snelson_fixtures.npz"))


def test_snelson_statistics_vs_reference_fixture(A, S):
    for tag, (o, m) in {"B3_30": (3, 30), "B3_100": (3, 100), "B4_30": (4, 30)}.items():
        bs = _mk_basis(A, o, -3.5, 10.5, m)
        model = A.GPR_1d((S["X"], S["Y"]), A.Matern32(), bs)
        np.testing.assert_allclose(model.KufKfu.cpu().numpy(), S[tag + "/KufKfu"], rtol=0, atol=1e-12 * np.max(S[tag + "/KufKfu"]))
        np.testing.assert_allclose(model.Kuf_y.cpu().numpy(), S[tag + "/Kuf_y"], rtol=0, atol=1e-12 * np.max(np.abs(S[tag + "/Kuf_y"])))
        assert abs(model.tr_yTy.item() - S[tag + "/tr_yTy"]) < 1e-12 * S[tag + "/tr_yTy"]
        assert model.bandwidth == o
        dense = model.KufKfu_sparse.to_dense().cpu().numpy()
        np.testing.assert_allclose(dense, O.band_to_dense_sym(S[tag + "/KufKfu"]), atol=1e-12)


def test_elbo_table_and_gradient(A, S):
    X, Y = S["X"], S["Y"]
    for o, m, kd, v, l, s, e in S["elbo_table"]:
        o, m, kd = int(o), int(m), int(kd)
        bs = _mk_basis(A, o, -3.5, 10.5, m)
        kern = _kernel(A, kd, v, l)
        model = A.GPR_1d((X, Y), kern, bs)
        model.likelihood.variance.assign(s)
        got = model.elbo().item()
        assert abs(got - e) <= 1e-9 * abs(e), (o, m, kd, v, l, s)
        assert abs(model.maximum_log_likelihood_objective().item() - e) <= 1e-9 * abs(e)
        assert abs(model.training_loss().item() + e) <= 1e-9 * abs(e)
        ob = O.Basis(o, -3.5, 10.5, m)
        Ab, b, yy = O.sufficient_stats(ob, X, Y)
        oe, og, _ = O.elbo_grad_1d(ob, kd, Ab, b, yy, 200, v, l, s)
        r = model.elbo_and_grad().cpu().numpy()
        assert abs(r[0] - oe) <= 1e-9 * abs(oe)
        np.testing.assert_allclose(r[1:4], og, rtol=1e-6, atol=1e-6 * np.max(np.abs(og)))


def test_elbo_synthetic_medium_vs_oracle(A):
    rng = np.random.default_rng(3)
    N, M = 20000, 256
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    for kd, order in ((1, 4), (2, 4), (0, 2), (2, 5), (1, 6)):
        bs = _mk_basis(A, order, 0, 1, M)
        model = A.GPR_1d((x.reshape(-1, 1), y), _kernel(A, kd, 1.0, 0.05), bs)
        model.likelihood.variance.assign(0.01)
        ob = O.Basis(order, 0, 1, M)
        Ab, b, yy = O.sufficient_stats_direct(ob, x, y)
        ee, ge = O.elbo_grad_1d_extended(ob, kd, Ab, b, yy, N, 1.0, 0.05, 0.01)   # the same recurrences in long double: the yardstick
        r = model.elbo_and_grad().cpu().numpy()
        # SURVEY 8d gates, no allowance for the cancelling N v / s terms (VERDICT r3 weak #1a): measured <= 5e-12 |ELBO|, gradient <= 4e-9
        assert abs(r[0] - ee) <= 1e-9 * abs(ee), (kd, order, r[0], ee)
        np.testing.assert_allclose(r[1:4], ge, rtol=1e-6, err_msg=str((kd, order)))


@pytest.mark.parametrize("M,order,D,N", [(40, 3, 2, 3000), (1024, 4, 3, 30000), (2048, 4, 7, 20000), (257, 2, 2, 5000)])
def test_multi_output_columns_on_every_band_algorithm(A, M, order, D, N):
    """y with D > 1 columns (gpr.py:75,80-82: solve_triang_mat with an M x D rhs; the bound sums over the columns): the first
    column rides through the cyclic-reduction levels, the others replay the stored factors (bcr_solve_more) - on the planned
    chain (auto), the all-GPU chains (2) and the sequential sweeps (1)."""
    rng = np.random.default_rng(4 + M)
    x = rng.uniform(0.01, 0.99, N)
    y = np.stack([np.sin((5 + 2 * d) * x + d) for d in range(D)], 1) + 0.1 * rng.normal(size=(N, D))
    bs = _mk_basis(A, order, 0, 1, M)
    l = 0.2 if M < 100 else 0.02
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=0.9, lengthscales=l), bs)
    model.likelihood.variance.assign(0.05)
    ob = O.Basis(order, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y)
    oe, og, _ = O.elbo_grad_1d(ob, 1, Ab, b, yy, N, 0.9, l, 0.05)
    xs = rng.uniform(0.01, 0.99, 100)
    om, ov = O.predict_f_1d_banded(ob, 1, Ab, b, 0.9, l, 0.05, xs)
    try:
        for algo in (0, 2, 1):
            A.set_band_algorithm(algo)
            model._post = None
            r = model.elbo_and_grad().cpu().numpy()
            assert abs(r[0] - oe) <= elbo_tol(oe, N, 0.9, 0.05, yy, bcr=(algo == 2)), (algo, r[0], oe)
            np.testing.assert_allclose(r[1:4], og, rtol=1e-6, err_msg="algo %d" % algo)
            mean, var = model.predict_f(xs.reshape(-1, 1))
            assert mean.shape == (100, D)
            np.testing.assert_allclose(mean, om, atol=1e-8)
            np.testing.assert_allclose(var, ov, atol=1e-8)
    finally:
        A.set_band_algorithm(0)


def test_notebook_golden_end_to_end(A, S):
    """experiments/snelson/example.py:25-33 on the HIP path -> example.ipynb:78  ASVGP: ELBO = -60.8356263428725"""
    bs = A.B3Spline(-3.5, 10.5, 100)
    model = A.GPR_1d((S["X"], S["Y"]), A.Matern32(), bs)
    res = model.fit()
    e = model.elbo().item()
    assert abs(e - float(S["golden_elbo_asvgp"])) < 1e-7
    assert e < float(S["golden_elbo_gp"])
    np.testing.assert_allclose(model.theta(), [0.798145059, 1.026880136, 0.080066643], rtol=5e-5)


def test_notebook_golden_by_backpropagation_through_the_operators(A, S):
    """INTEGRATION Level 2: the reference's training loop (example.py:31-32) differentiates GPR_1d.elbo through the banded ops.  The
    same bound written op by op (gpr.py:49-89) on softplus-constrained torch scalars, its gradient from the operator VJPs, L-BFGS-B
    as gpflow.optimizers.Scipy runs it: reaches the notebook's ASVGP: ELBO = -60.8356263428725 (example.ipynb:78)."""
    from scipy.optimize import minimize
    from asvgp_amd import banded as Bd
    bs = A.B3Spline(-3.5, 10.5, 100)
    model = A.GPR_1d((S["X"], S["Y"]), A.Matern32(), bs)
    k, N = 3, float(model.num_data)
    St_ = model.inducing_features.static_stack(1)
    Aband, bvec, yy = model.KufKfu, model.Kuf_y, model.tr_yTy
    s3 = np.sqrt(3.0)

    def neg_elbo(u):
        ut = torch.tensor(u, dtype=torch.float64, device="cuda", requires_grad=True)
        sp = torch.nn.functional.softplus(ut)
        v, l, sg = sp[0], sp[1], sp[2] + 1e-6                             # gpflow: positive(), Gaussian variance lower bound 1e-6
        cs = [s3 / (4 * l * v), l / (2 * s3 * v), l ** 3 / (12 * s3 * v), 1 / (2 * v), l ** 2 / (2 * v)]
        Kuu = sum(c * St_[t] for t, c in enumerate(cs))
        Lk = Bd.cholesky_band(Kuu)
        Kinv = Bd.inverse_from_cholesky_band(Lk)
        trace = Bd.product_band_band(Bd.symmetrise_band(Kinv, k), Bd.symmetrise_band(Aband, k), k, k, k, k, 0, 0).sum()
        Lp = Bd.cholesky_band(Aband / sg + Kuu)
        c = Bd.solve_triang_mat(Lp, bvec) / sg
        elbo = (-0.5 * N * torch.log(2 * np.pi * sg) - 0.5 * torch.log(Lp[0] ** 2).sum() + 0.5 * torch.log(Lk[0] ** 2).sum()
                - 0.5 * yy / sg + 0.5 * (c ** 2).sum() - 0.5 * N * v / sg + 0.5 * trace / sg)
        (-elbo).backward()
        return -elbo.item(), ut.grad.cpu().numpy()

    inv_sp = lambda x: float(np.log(np.expm1(x)))
    u0 = np.array([inv_sp(1.0), inv_sp(1.0), inv_sp(1.0 - 1e-6)])        # GPflow defaults (all 1.0)
    res = minimize(neg_elbo, u0, jac=True, method="L-BFGS-B", options=dict(maxiter=15000))
    assert abs(-res.fun - float(S["golden_elbo_asvgp"])) < 1e-6, (-res.fun, S["golden_elbo_asvgp"])


def test_predict_vs_oracle_and_survey_values(A, S, golden_dir):
    Xs = np.loadtxt(os.path.join(golden_dir, "snelson", "test_inputs")).reshape(-1, 1)
    bs = A.B3Spline(-3.5, 10.5, 100)
    v, l, s = 0.798145059, 1.026880136, 0.080066643
    model = A.GPR_1d((S["X"], S["Y"]), A.Matern32(variance=v, lengthscales=l), bs)
    model.likelihood.variance.assign(s)
    mean, var = model.predict_f(Xs)
    assert isinstance(mean, np.ndarray) and mean.shape == (301, 1) and var.shape == (301, 1)
    ob = O.Basis(3, -3.5, 10.5, 100)
    Ab, b, yy = O.sufficient_stats(ob, S["X"], S["Y"])
    om, ov = O.predict_f_1d(ob, 1, Ab, b, v, l, s, Xs)
    np.testing.assert_allclose(mean, om, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var, ov, rtol=0, atol=1e-8)
    np.testing.assert_allclose(mean[[0, 150, 300], 0], [0.01123846, -0.20213625, -0.00027056], atol=2e-7)
    np.testing.assert_allclose(var[[0, 150, 300], 0], [0.79721169, 0.00731127, 0.79809196], atol=2e-7)
    with pytest.raises(NotImplementedError):
        model.predict_f(Xs, full_cov=True)
    # batch=True reproduces the reference's dropped remainder (gpr.py:125-136)
    big = np.linspace(-3, 10, 25_000).reshape(-1, 1)
    mb, vb = model.predict_f(big, batch=True)
    m1, v1 = model.predict_f(big)
    np.testing.assert_allclose(mb[:20_000], m1[:20_000], atol=1e-12)
    assert not mb[20_000:].any() and not vb[20_000:].any()
    # large streaming predict (LDS-staged path) equals the small path
    big2 = np.linspace(-3.4, 10.4, 200_000).reshape(-1, 1)
    m2, v2 = model.predict_f(big2)
    sel = np.arange(0, 200_000, 997)
    m3, v3 = model.predict_f(big2[sel])
    np.testing.assert_allclose(m2[sel], m3, atol=1e-13)
    np.testing.assert_allclose(v2[sel], v3, atol=1e-13)


def test_kron_and_dense_models_refuse_cpu_tensors(A):
    """GPR_kron (d = 2 and the dense d = 3 route) must raise the library's error for host tensors, not pass raw pointers."""
    from asvgp_amd._lib import AsvgpError
    rng = np.random.default_rng(0)
    for d in (2, 3):
        X = torch.from_numpy(rng.uniform(0.05, 0.95, (100, d)))
        y = torch.from_numpy(rng.normal(size=(100, 1)))
        bases = [A.B3Spline(0, 1, 8) for _ in range(d)]
        assert A.GPR_kron((X, y), [A.Matern32() for _ in range(d)], bases).X.is_cuda      # host tensors are moved to the basis' device
        bases[0].device = torch.device("cpu")                                              # a basis that lives on the host: refuse
        with pytest.raises(AsvgpError):
            A.GPR_kron((X, y), [A.Matern32() for _ in range(d)], bases)


def test_cpu_tensors_are_refused(A):
    from asvgp_amd import banded, _lib
    with pytest.raises(_lib.AsvgpError):
        banded.cholesky_band(torch.ones(2, 4, dtype=torch.float64))


def kuu_cond(ob, kd, v, l):
    K = O.band_to_dense_sym(O.make_Kuu(ob, kd, v, l))
    w = np.linalg.eigvalsh(K)
    return float(w[-1] / w[0])


@pytest.mark.parametrize("order,M,kd,l", [(1, 37, 0, 0.08), (2, 64, 1, 0.08), (3, 100, 1, 0.08), (4, 256, 2, 0.08),
                                          (4, 255, 1, 0.08), (4, 13, 0, 0.3), (5, 129, 2, 0.05), (6, 90, 1, 0.08),
                                          (4, 2048, 1, 0.003), (3, 1000, 2, 0.006),     # well conditioned (l ~ 6 delta)
                                          (4, 2048, 1, 0.05), (3, 1000, 2, 0.08),       # BASELINE-like, cond(Kuu) >> 1e8
                                          (4, 4096, 2, 0.002), (4, 4096, 1, 0.004), (3, 3000, 0, 0.01),   # BIG BCR layout (config 3 shape)
                                          (5, 2500, 2, 0.006)])
def test_block_cyclic_reduction_vs_sequential_sweeps_and_oracle(A, order, M, kd, l):
    """Three evaluation orders of the same factorisation - sequential column sweeps (1, the reference's order), all-GPU block
    cyclic reduction (2) and block cyclic reduction with the planned prior chain (3, the default: forward pass of the Kuu chain
    in long double on the host) - against the oracle (fp64, reference order) AND its long-double evaluation.
    Gate for the default path and for the sweeps: 1e-9 |ELBO| + max(5 |oracle - long double|, 2e-11 (N v/2s + y'y/2s)), i.e.
    within 5x of what the reference's own fp64 order loses (no eps*cond slack).  The all-GPU cyclic reduction is the fast
    fallback for bands without Toeplitz structure and is only required to stay within eps*cond of the cancelling terms."""
    rng = np.random.default_rng(order * 7 + M)
    N = 4000 if M < 1500 else 40000
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    v, s = 0.9, 0.02
    bs = _mk_basis(A, order, 0, 1, M)
    model = A.GPR_1d((x.reshape(-1, 1), y), _kernel(A, kd, v, l), bs)
    model.likelihood.variance.assign(s)
    xs = rng.uniform(0.001, 0.999, 500).reshape(-1, 1)
    res = {}
    try:
        for algo in (1, 2, 3):
            A.set_band_algorithm(algo)
            model._post = None
            r = model.elbo_and_grad().cpu().numpy()
            mean, var = model.predict_f(xs)
            res[algo] = (r, mean, var)
    finally:
        A.set_band_algorithm(0)
    model._post = None
    r0 = model.elbo_and_grad().cpu().numpy()
    np.testing.assert_allclose(r0, res[3][0], rtol=1e-9, atol=1e-9 * float(np.max(np.abs(r0))))   # auto == planned prior chain (the finalize sums with atomics: last bits vary)
    ob = O.Basis(order, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y)
    oe, og, _ = O.elbo_grad_1d(ob, kd, Ab, b, yy, N, v, l, s)
    ee, ge = O.elbo_grad_1d_extended(ob, kd, Ab, b, yy, N, v, l, s)
    om, ov = O.predict_f_1d_banded(ob, kd, Ab, b, v, l, s, xs)
    cond = kuu_cond(ob, kd, v, l)
    big = 0.5 * N * v / s + 0.5 * yy / s
    eps_c = 2.2e-16 * cond                       # what fp64 cyclic reduction can lose of the cancelling O(N v / s) terms
    gate = 1e-9 * abs(ee) + max(5 * abs(oe - ee), 2e-11 * big)
    for algo in (1, 3):
        r, mean, var = res[algo]
        assert abs(r[0] - ee) <= gate, (algo, r[0], ee, oe, gate, cond)
        gt = max(1e-6, 5 * float(np.max(np.abs((og - ge) / ge))))
        np.testing.assert_allclose(r[1:4], ge, rtol=gt, atol=gt * np.max(np.abs(ge)), err_msg="algo %d cond %.1e" % (algo, cond))
    r, mean, var = res[2]
    assert abs(r[0] - ee) <= elbo_tol(ee, N, v, s, yy, bcr=True) + eps_c * big, (2, r[0], ee, cond)
    gt = max(1e-6, 50 * eps_c)
    np.testing.assert_allclose(r[1:4], ge, rtol=gt, atol=gt * np.max(np.abs(ge)), err_msg="algo 2 cond %.1e" % cond)
    for algo in (1, 2, 3):
        r, mean, var = res[algo]
        pt = max(1e-8, 10 * eps_c) if algo == 2 else max(1e-8, 5 * float(np.max(np.abs(res[1][2] - ov))))
        np.testing.assert_allclose(mean, om, rtol=0, atol=pt)
        np.testing.assert_allclose(var, ov, rtol=0, atol=pt)


@pytest.mark.parametrize("order,M,N,sort", [(1, 16, 5000, False), (2, 33, 7001, False), (3, 100, 20000, True), (4, 64, 20000, False),
                                             (4, 1024, 100000, False), (4, 2048, 150001, True), (5, 200, 30000, False),
                                             (6, 90, 10000, True), (4, 4096, 60000, False), (3, 3000, 50000, True)])
def test_phi_algorithms_agree_with_oracle(A, order, M, N, sort):
    """Algorithm 1 (per-point fp64 LDS atomic band scatter), 3 (the same in 64-bit fixed point), 5 (fixed-point centred
    moments per cell for Phi Phi^T + fixed-point scatter of Phi y) and 6 (tile sort + register moments - the default where it
    applies) give the same statistics; sorted (time-series) inputs exercise the wave-uniform run mode / the heavy-cell path,
    repeated points the same-address paths."""
    rng = np.random.default_rng(M + N)
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    if sort:
        x = np.sort(x)
        x[N // 2:N // 2 + 3000] = x[N // 2]          # a few thousand identical points in one cell
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    bs = _mk_basis(A, order, 0, 1, M)
    ob = O.Basis(order, 0, 1, M)
    band, rhs, yy = O.sufficient_stats_direct(ob, x, y)
    got = {}
    try:
        for algo in (1, 3, 5, 6):
            A.set_phi_algorithm(algo)
            try:
                m = A.GPR_1d((x.reshape(-1, 1), y), A.Matern12(), bs)
            except RuntimeError:
                assert algo in (5, 6) and M > 2048  # the moment image / the owner map exceed one workgroup: auto falls back to 3
                continue
            got[algo] = m._stats.cpu().numpy().copy()
            gotb = m.KufKfu.cpu().numpy()
            assert np.max(np.abs(gotb - band)) <= 1e-12 * np.max(np.abs(band)), algo
            pad = np.zeros_like(band, dtype=bool)
            for d in range(1, order + 1):
                pad[d, M - d:] = True
            assert (gotb[pad] == 0).all(), algo       # structural zeros (right padding) stay exact zeros
            if algo == 1:
                assert np.array_equal(gotb == 0, band == 0), algo
            np.testing.assert_allclose(m.Kuf_y.cpu().numpy(), rhs, rtol=0, atol=1e-12 * np.max(np.abs(rhs)))
            assert abs(m.tr_yTy.item() - yy) <= 1e-12 * yy
    finally:
        A.set_phi_algorithm(0)
    for algo in got:
        assert np.max(np.abs(got[1] - got[algo])) <= 1e-12 * np.max(np.abs(got[1]))
    m = A.GPR_1d((x.reshape(-1, 1), y), A.Matern12(), bs)          # auto
    assert np.max(np.abs(m._stats.cpu().numpy() - got[1])) <= 1e-12 * np.max(np.abs(got[1]))


@pytest.mark.parametrize("order,M,N,shape", [(4, 2048, 400000, "sorted"), (4, 2048, 400001, "descending"), (4, 512, 300000, "blocks"),
                                              (4, 2048, 300000, "half"), (3, 100, 200000, "sorted"), (2, 33, 100000, "knots"),
                                              (5, 200, 150000, "sorted"), (6, 90, 150000, "blocks"), (1, 16, 100000, "sorted"),
                                              (4, 2048, 60000, "thin"), (4, 1024, 200000, "unsorted"), (4, 2048, 300000, "outlier")])
def test_phi_time_series_front_loop_gives_the_statistics_of_the_sort_loop_and_the_oracle(A, order, M, N, shape):
    """The tile-sort instantiation with the time-series front loop (asvgp_set_phi_input_order 2; run sums per wave, no sort) against
    the oracle and against the plain instantiation, on inputs that stay in the front loop (sorted, descending), leave it part of the
    way (half sorted, an out-of-order point), never enter it (unsorted), cross knots exactly, or have cells thinner than a row; and
    the probe's verdict on them (order 0)."""
    rng = np.random.default_rng(N + M)
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    if shape in ("sorted", "outlier", "knots", "thin"):
        x = np.sort(x)
    if shape == "descending":
        x = np.sort(x)[::-1].copy()
    if shape == "blocks":                                  # sorted inside blocks of 20 000 points, blocks in random order
        x = np.concatenate([np.sort(b) for b in np.array_split(x, max(1, N // 20000))])
    if shape == "half":
        x[:N // 2] = np.sort(x[:N // 2])
    if shape == "outlier":
        x[N // 3] = 0.999
        x[5] = 0.5
    ob = O.Basis(order, 0, 1, M)
    if shape == "knots":                                   # runs of points exactly ON knots
        x[1000:1200] = ob.mesh[7]
        x[50000:50100] = ob.mesh[-2]
        x = np.sort(x)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    bs = _mk_basis(A, order, 0, 1, M)
    band, rhs, yy = O.sufficient_stats_direct(ob, x, y)
    got = {}
    try:
        for mode in (1, 2, 0):
            A.set_phi_input_order(mode)
            m = A.GPR_1d((x.reshape(-1, 1), y), A.Matern12(), bs)
            assert m._h.phi_last_algorithm() == 6
            ran = m._h.phi_last_input_order()
            if mode:
                assert ran == mode
            elif shape in ("sorted", "descending", "knots", "outlier"):
                assert ran == 2, (shape, ran)                       # the probe: 512 sampled rows, nine in ten inside two cells
            elif shape in ("unsorted", "thin"):
                assert ran == 1, (shape, ran)                       # ("thin": cells of ~30 points - a 128-point row spans four)
            got[mode] = m._stats.cpu().numpy().copy()
            gotb = m.KufKfu.cpu().numpy()
            assert np.max(np.abs(gotb - band)) <= 1e-12 * np.max(np.abs(band)), mode
            np.testing.assert_allclose(m.Kuf_y.cpu().numpy(), rhs, rtol=0, atol=1e-12 * np.max(np.abs(rhs)))
            assert abs(m.tr_yTy.item() - yy) <= 1e-12 * yy
    finally:
        A.set_phi_input_order(0)
    assert np.max(np.abs(got[1] - got[2])) <= 1e-12 * np.max(np.abs(got[1]))


@pytest.mark.parametrize("bad", [1.5, -0.25, float("nan")])
def test_time_series_front_loop_reports_a_point_outside_the_mesh(A, bad):
    """A point outside [a, b] (or NaN) inside a sorted input sent through the time-series instantiation: the tile leaves the front loop and
    the sort loop reports it as NaN y^T y, exactly as the plain instantiation does (the model itself refuses such X first, gpr.py:22-26)."""
    rng = np.random.default_rng(5)
    N, M = 200000, 256
    x = np.sort(rng.uniform(1e-9, 1 - 1e-9, N))
    y = rng.normal(size=N)
    try:
        A.set_phi_input_order(2)
        m = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern12(), A.B4Spline(0, 1, M))
        assert m._h.phi_last_input_order() == 2 and torch.isfinite(m.tr_yTy)
        m.X[N // 2] = bad                            # behind the constructor's check
        m.phi_pass()
        assert m._h.phi_last_input_order() == 2 and torch.isnan(m.tr_yTy)
    finally:
        A.set_phi_input_order(0)


@pytest.mark.parametrize("order,M,N,dist", [(3, 1442, 45014, "ends"), (2, 1642, 242047, "clustered"), (4, 2048, 100000, "uniform")])
def test_phi_pass_on_float32_linspace_mesh_cells_that_are_not_exactly_delta_wide(A, order, M, N, dist):
    """basis.py:17 builds the knots of (-3.5, 10.5) in float32 (tf.linspace of Python floats): the cells are up to ulp32(10.5)/delta
    = 1e-4 wider than delta, so t = (x - knot)/delta reaches 1.0001 and the reference simply evaluates its pieces there.  (Found
    by tests/sweeps/fuzz_phi.py: the moment kernel's first 'outside the mesh' test was |t - 1/2| <= 0.50001 and reported such points.)"""
    rng = np.random.default_rng(order * M)
    a, b = -3.5, 10.5
    lo, hi = a + 1e-9 * (b - a), b - 1e-9 * (b - a)
    if dist == "ends":
        x = rng.choice([lo, a + 0.3 * (b - a), hi], N)
    elif dist == "clustered":
        x = np.clip(a + (b - a) * (0.5 + 0.03 * rng.standard_normal(N)), lo, hi)
    else:
        x = rng.uniform(lo, hi, N)
    y = 100.0 * rng.standard_normal((N, 1))
    ob = O.Basis(order, a, b, M)
    band, rhs, yy = O.sufficient_stats_direct(ob, x, y)
    try:
        for algo in (0, 3, 5):            # (6 needs an exact fp64 linspace: auto takes 5 on these meshes)
            A.set_phi_algorithm(algo)
            m = A.GPR_1d((x.reshape(-1, 1), y), A.Matern12(), _mk_basis(A, order, a, b, M))
            tol = 1e-12 + 4e-16 * N          # (the oracle's sequential sums lose ~N eps on heavily repeated x)
            assert np.max(np.abs(m.KufKfu.cpu().numpy() - band)) <= tol * np.max(np.abs(band)), algo
            assert np.max(np.abs(m.Kuf_y.cpu().numpy() - rhs)) <= tol * np.max(np.abs(rhs)), algo
            assert abs(m.tr_yTy.item() - yy) <= tol * yy, algo
    finally:
        A.set_phi_algorithm(0)


# ------------------------------------------------------------------------------------------------ Kronecker 2-D
def _blockband_to_dense(blk, k, m1, m2):
    """unpack the lower block band [off][col] into a dense symmetric (m1*m2)^2 matrix (test helper)."""
    M = m1 * m2
    out = np.zeros((M, M))
    offs = [(0, d2) for d2 in range(k + 1)] + [(d1, d2) for d1 in range(1, k + 1) for d2 in range(-k, k + 1)]
    for o, (d1, d2) in enumerate(offs):
        for i1 in range(m1 - d1):
            for i2 in range(max(0, -d2), min(m2, m2 - d2)):
                c = i1 * m2 + i2
                r = (i1 + d1) * m2 + i2 + d2
                out[r, c] = blk[o, c]
                out[c, r] = blk[o, c]
    return out


def test_khatri_rao_vs_reference_fixture(A, golden_dir):
    Kf = np.load(os.path.join(golden_dir, "kron_fixtures.npz"))
    X = Kf["X"]
    bases = [A.B3Spline(0, 1, 12), A.B3Spline(-1, 2, 14)]
    KR = A.kronecker.make_kvs_sparse(bases, dev(X)).to_dense().cpu().numpy()
    ref = Kf["dense"]
    assert np.array_equal(np.abs(KR) > 1e-300, np.abs(ref) > 1e-300)       # row ids (dim-0 major): bit exact
    np.testing.assert_allclose(KR, ref, rtol=0, atol=1e-15)
    rows, cols, data = A.kronecker.make_kvs_coo(bases, dev(X))
    assert int(rows.max()) < 12 * 14 and cols.shape[0] == 16 * X.shape[0]


@pytest.mark.parametrize("order,m1,m2,N", [(3, 8, 9, 300), (4, 12, 13, 2000), (2, 10, 7, 500), (3, 20, 24, 5000)])
def test_kron_statistics_elbo_predict_vs_oracle(A, order, m1, m2, N):
    rng = np.random.default_rng(m1 * 100 + m2)
    X = np.stack([rng.uniform(0.001, 0.999, N), rng.uniform(-0.999, 1.999, N)], axis=1)
    y = (np.sin(12 * X[:, :1]) * np.cos(3 * X[:, 1:]) + 0.1 * rng.normal(size=(N, 1)))
    B = getattr(A, "B%dSpline" % order)
    bases = [B(0, 1, m1), B(-1, 2, m2)]
    kerns = [A.Matern32(variance=1.1, lengthscales=0.3), A.Matern32(variance=0.7, lengthscales=0.6)]
    model = A.GPR_kron((X, y), kerns, bases)
    model.likelihood.variance.assign(0.05)
    obases = [O.Basis(order, 0, 1, m1), O.Basis(order, -1, 2, m2)]
    oe, parts = O.elbo_kron(obases, [1, 1], [(1.1, 0.3), (0.7, 0.6)], 0.05, X, y)
    Ad = _blockband_to_dense(model.KufKfu_blockband.cpu().numpy(), order, m1, m2)
    np.testing.assert_allclose(Ad, parts["A"], rtol=0, atol=1e-12 * np.max(np.abs(parts["A"])))
    np.testing.assert_allclose(model.Kuf_y.cpu().numpy(), parts["b"], rtol=0, atol=1e-12 * np.max(np.abs(parts["b"])))
    assert abs(model.tr_yTy.item() - np.sum(y * y)) <= 1e-12 * np.sum(y * y)
    assert model.true_bandwidth == order * (m2 + 1)
    e = model.elbo().item()
    yy = float(np.sum(y * y))
    assert abs(e - oe) <= elbo_tol(oe, N, 1.1 * 0.7, 0.05, yy, bcr=True), (e, oe)
    Xs = np.stack([rng.uniform(0.01, 0.99, 200), rng.uniform(-0.99, 1.99, 200)], axis=1)
    om, ov = O.predict_f_kron(obases, [1, 1], [(1.1, 0.3), (0.7, 0.6)], 0.05, X, y, Xs)
    mean, var = model.predict_f(Xs)
    np.testing.assert_allclose(mean, om, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var, ov, rtol=0, atol=1e-8)
    with pytest.raises(AssertionError):
        A.GPR_kron((X, y), kerns[:1], bases)                               # gpr.py:247


# ------------------------------------------------------------------------------------------------ N-shards + band all-reduce
def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                       # both ranks share the one GPU of the test box; gloo moves the band
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import asvgp_amd as A
    from asvgp_amd.dist import shard_bounds
    rng = np.random.default_rng(2024)
    N, M = 200_001, 512
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    lo, hi = shard_bounds(N, world, rank)
    model = A.GPR_1d((x[lo:hi].reshape(-1, 1), y[lo:hi]), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M),
                     process_group=dist.group.WORLD)
    model.likelihood.variance.assign(0.01)
    r = model.elbo_and_grad().cpu().numpy()
    # re-running the Phi pass on a sharded model all-reduces again (it used to leave the local shard behind)
    model.phi_pass()
    r_again = model.elbo_and_grad().cpu().numpy()
    assert np.allclose(r_again, r, rtol=1e-9, atol=0), (r_again, r)
    local = model.phi_pass(allreduce=False).clone()
    assert float(local[-1]) < 0.75 * float(model.phi_pass()[-1])      # y^T y of one shard vs the global sum
    # the 2-D models shard the same way (block band / flat additive buffer through the same all-reduce)
    rng2 = np.random.default_rng(77)
    N2 = 6001
    X2 = rng2.uniform(0.001, 0.999, (N2, 2))
    y2 = np.sin(5 * X2[:, :1]) + X2[:, 1:] ** 2 + 0.1 * rng2.normal(size=(N2, 1))
    lo2, hi2 = shard_bounds(N2, world, rank)
    bases2 = [A.B3Spline(0, 1, 11), A.B3Spline(0, 1, 10)]
    mk = A.GPR_kron((X2[lo2:hi2], y2[lo2:hi2]), [A.Matern32(), A.Matern32()], bases2, process_group=dist.group.WORLD)
    ma = A.GPR_additive((X2[lo2:hi2], y2[lo2:hi2]), [A.Matern32(), A.Matern32()], bases2, process_group=dist.group.WORLD)
    extra = (mk.num_data, mk.elbo().item(), ma.num_data, ma.elbo().item())
    q.put((rank, model.num_data, model._stats.cpu().numpy(), r, extra))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_model_matches_single_rank(A):
    """contiguous N-shards, local Phi pass, ONE sum of the packed band buffer, global N; here with 2 ranks on
    one GPU over gloo (the 8-GPU RCCL run is the driver's); compared within the fp64 tolerance, never bit-exactly."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(2024)
    N, M = 200_001, 512
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    single = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    single.likelihood.variance.assign(0.01)
    r1 = single.elbo_and_grad().cpu().numpy()
    s1 = single._stats.cpu().numpy()
    rng2 = np.random.default_rng(77)
    N2 = 6001
    X2 = rng2.uniform(0.001, 0.999, (N2, 2))
    y2 = np.sin(5 * X2[:, :1]) + X2[:, 1:] ** 2 + 0.1 * rng2.normal(size=(N2, 1))
    bases2 = [A.B3Spline(0, 1, 11), A.B3Spline(0, 1, 10)]
    ek = A.GPR_kron((X2, y2), [A.Matern32(), A.Matern32()], bases2).elbo().item()
    ea = A.GPR_additive((X2, y2), [A.Matern32(), A.Matern32()], bases2).elbo().item()
    for rank, n_glob, stats, r, extra in res:
        assert n_glob == N
        assert np.max(np.abs(stats - s1)) <= 1e-12 * np.max(np.abs(s1))
        assert abs(r[0] - r1[0]) <= elbo_tol(r1[0], N, 1.0, 0.01, float(s1[-1]), bcr=True)
        np.testing.assert_allclose(r[1:4], r1[1:4], rtol=1e-6)
        assert extra[0] == N2 and extra[2] == N2
        assert abs(extra[1] - ek) <= 1e-9 * abs(ek) and abs(extra[3] - ea) <= 1e-9 * abs(ea)


def test_split_prior_and_data_chain_equals_fused_call(A):
    """asvgp_elbo_prior_chain_1d (side stream) + asvgp_elbo_data_chain_1d == asvgp_elbo_grad_1d, for the all-GPU chains
    (band algorithm 2: two-stream schedule) and for the planned prior chain (3: the prior call is a no-op, the data call runs
    both chains in one launch) and the sequential sweeps (1)."""
    rng = np.random.default_rng(8)
    N, M = 30000, 1024
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.01), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(0.01)
    ref = model.elbo_and_grad().cpu().numpy()          # auto = planned prior chain
    side = torch.cuda.Stream()
    try:
        for algo in (2, 3, 1):
            A.set_band_algorithm(algo)
            fused = model.elbo_and_grad().cpu().numpy()
            np.testing.assert_allclose(fused, ref, rtol=1e-7)
            model._out.zero_()
            done = torch.cuda.Event()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model.launch_prior_chain()
                done.record(side)
            model.phi_pass()
            torch.cuda.current_stream().wait_event(done)
            split = model.launch_data_chain()[:4].cpu().numpy()
            model._check_pd()
            np.testing.assert_allclose(split, fused, rtol=1e-12)
        # handle-internal ordering (asvgp_elbo_chain_sync): no caller-side event
        A.set_band_algorithm(2)
        fused = model.elbo_and_grad().cpu().numpy()
        model._h.chain_sync(1)
        try:
            model._out.zero_()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model.launch_prior_chain()
            model.phi_pass()
            split2 = model.launch_data_chain()[:4].cpu().numpy()
            torch.cuda.current_stream().wait_stream(side)
        finally:
            model._h.chain_sync(0)
        np.testing.assert_allclose(split2, fused, rtol=1e-12)
    finally:
        A.set_band_algorithm(0)


def test_two_models_step_concurrently_on_two_streams_and_threads(A):
    """SURVEY 8 b3 / VERDICT r1 #6: library state lives in per-model handles, so two models with different theta, algorithms
    and sizes may step from two host threads on two streams; each must get exactly what it gets alone."""
    import threading
    rng = np.random.default_rng(31)
    specs = [(60000, 1024, 0.03, 0), (45001, 512, 0.08, 2)]
    models, alone = [], []
    for N, M, l, algo in specs:
        x = rng.uniform(1e-9, 1 - 1e-9, N)
        y = (np.sin(15 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
        m = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.1, lengthscales=l), A.B4Spline(0, 1, M))
        m.likelihood.variance.assign(0.02)
        if algo:
            m._h.set_band_algorithm(algo)
        models.append(m)
        alone.append(m.elbo_and_grad().cpu().numpy())
    torch.cuda.synchronize()
    got = [None, None]
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(40):
                    models[i].phi_pass()
                    r = models[i].elbo_and_grad(check_pd=False)
                st.synchronize()
                got[i] = r.cpu().numpy()
        except Exception as exc:      # noqa: BLE001
            errs.append(exc)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(2):
        np.testing.assert_allclose(got[i], alone[i], rtol=1e-12)


def test_kron_fit_improves_bound(A):
    rng = np.random.default_rng(12)
    N = 1500
    X = np.stack([rng.uniform(0.001, 0.999, N), rng.uniform(0.001, 0.999, N)], axis=1)
    y = (np.sin(6 * X[:, :1]) * np.cos(4 * X[:, 1:]) + 0.1 * rng.normal(size=(N, 1)))
    model = A.GPR_kron((X, y), [A.Matern32(), A.Matern32()], [A.B3Spline(0, 1, 10), A.B3Spline(0, 1, 11)])
    e0 = model.elbo().item()
    res = model.fit(maxiter=15)
    e1 = model.elbo().item()
    assert e1 > e0 + 1.0 and np.isfinite(e1)
    assert float(model.likelihood.variance) < 0.5      # started at 1.0, the data noise is 0.01


def test_config4_float32_data_is_accepted_and_upcast(A):
    """BASELINE config 4 says fp32 (the reference computes in fp64 throughout): float32 X, y are accepted and upcast once on the device -
    storage precision fp32, arithmetic fp64.  The bound equals the oracle's on the float32-rounded data and sits within the
    configuration's 1e-3 gate of the fp64-data value."""
    rng = np.random.default_rng(40)
    N, ms = 30000, [20, 18]
    X = rng.uniform(0.001, 0.999, (N, 2))
    y = np.sin(6 * X[:, :1]) * np.cos(4 * X[:, 1:]) + 0.1 * rng.standard_normal((N, 1))
    X32, y32 = X.astype(np.float32), y.astype(np.float32)
    kinds, th, s = [1, 1], [(1.0, 0.3), (0.9, 0.4)], 0.05
    obases = [O.Basis(3, 0, 1, m) for m in ms]

    def build(Xa, ya):
        mdl = A.GPR_kron((torch.from_numpy(Xa).cuda(), torch.from_numpy(ya).cuda()), [_kernel(A, kd, v, l) for kd, (v, l) in zip(kinds, th)],
                         [_mk_basis(A, 3, 0, 1, m) for m in ms])
        mdl.likelihood.variance.assign(s)
        return mdl

    m32 = build(X32, y32)
    assert m32.X.dtype == torch.float64
    e32 = float(m32.elbo())
    o32 = O.elbo_kron(obases, kinds, th, s, X32.astype(np.float64), y32.astype(np.float64))[0]
    o64 = O.elbo_kron(obases, kinds, th, s, X, y)[0]
    assert abs(e32 - o32) <= 1e-9 * abs(o32) + 1e-8 * (0.5 * N / s)
    assert abs(e32 - o64) <= 1e-3 * abs(o64)


def test_kron_three_dimensions_dense_route_vs_oracle(A):
    """kronecker.py:32-33 folds over any number of dimensions and gpr.py:260-308 is dense for all of them; d = 2 has the banded HIP
    path, any other d takes the same dense route on the device: bound, analytic gradient, posterior and one optimiser step for
    d = 3 against the dense oracle."""
    rng = np.random.default_rng(33)
    N = 4000
    X = rng.uniform(0.001, 0.999, (N, 3))
    y = (np.sin(5 * X[:, :1]) * np.cos(3 * X[:, 1:2]) + X[:, 2:] ** 2 + 0.1 * rng.standard_normal((N, 1)))
    ms, kinds, th, s = [7, 8, 9], [1, 0, 1], [(1.0, 0.5), (0.8, 0.6), (1.2, 0.4)], 0.05
    bases = [_mk_basis(A, 2, 0, 1, m) for m in ms]
    obases = [O.Basis(2, 0, 1, m) for m in ms]
    kerns = [_kernel(A, kd, v, l) for kd, (v, l) in zip(kinds, th)]
    model = A.GPR_kron((X, y), kerns, bases)
    model.likelihood.variance.assign(s)
    oe, og = O.elbo_grad_kron(obases, kinds, th, s, X, y)
    e, g = model.elbo_and_grad()
    assert abs(e - oe) <= 1e-9 * abs(oe) + 1e-8 * (0.5 * N / s)
    assert abs(model.elbo().item() - e) <= 1e-10 * abs(e)
    np.testing.assert_allclose(g, og, rtol=1e-6, atol=1e-6 * np.max(np.abs(og)))
    Xs = rng.uniform(0.01, 0.99, (200, 3))
    om, ov = O.predict_f_kron(obases, kinds, th, s, X, y, Xs)
    mean, var = model.predict_f(Xs)
    np.testing.assert_allclose(mean, om, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var[:, :1], ov[:, :1] if ov.ndim == 2 else ov.reshape(-1, 1), rtol=0, atol=1e-8)
    res = model.fit(maxiter=2)
    assert np.isfinite(res.fun)
    # round 4: P is factored as a BAND matrix (bandwidth k (m2 m3 + m3 + 1)) by the band Cholesky kernel, the gradient and the posterior come
    # from the band-restricted inverse; the dense factorisations remain for bands that do not fit - both routes, same numbers
    model = A.GPR_kron((X, y), [_kernel(A, kd, v, l) for kd, (v, l) in zip(kinds, th)], bases)
    model.likelihood.variance.assign(s)
    lay = model._nd_band_layout()
    assert lay is not None and lay["bw"] == 2 * (8 * 9 + 9 + 1)
    dense = A.GPR_kron((X, y), [_kernel(A, kd, v, l) for kd, (v, l) in zip(kinds, th)], bases)
    dense.likelihood.variance.assign(s)
    dense.nd_banded = False
    assert dense._nd_band_layout() is None
    eb, gb = model.elbo_and_grad()
    ed, gd = dense.elbo_and_grad()
    assert abs(eb - ed) <= 1e-10 * abs(ed)
    np.testing.assert_allclose(gb, gd, rtol=1e-7, atol=1e-7 * np.max(np.abs(gd)))
    mb, vb = model.predict_f(Xs)
    md, vd = dense.predict_f(Xs)
    np.testing.assert_allclose(mb, md, rtol=0, atol=1e-9)
    np.testing.assert_allclose(vb, vd, rtol=0, atol=1e-9)


def test_kron_four_dimensions_band_route_vs_dense_route(A):
    """d = 4 (kronecker.py:32-33 folds over any d): the band route (bandwidth k (m2 m3 m4 + m3 m4 + m4 + 1)) against the dense route and the oracle."""
    rng = np.random.default_rng(34)
    N = 3000
    X = rng.uniform(0.001, 0.999, (N, 4))
    y = (np.sin(4 * X[:, :1]) + X[:, 1:2] * X[:, 2:3] - np.cos(3 * X[:, 3:]) + 0.1 * rng.standard_normal((N, 1)))
    ms, kinds, th, s = [5, 6, 4, 5], [0, 0, 0, 0], [(1.0, 0.5), (0.8, 0.6), (1.2, 0.4), (0.9, 0.7)], 0.08
    bases = [_mk_basis(A, 1, 0, 1, m) for m in ms]
    mk = lambda: [_kernel(A, kd, v, l) for kd, (v, l) in zip(kinds, th)]
    model = A.GPR_kron((X, y), mk(), bases)
    model.likelihood.variance.assign(s)
    lay = model._nd_band_layout()
    assert lay is not None and lay["bw"] == 1 * (6 * 4 * 5 + 4 * 5 + 5 + 1)
    dense = A.GPR_kron((X, y), mk(), bases)
    dense.likelihood.variance.assign(s)
    dense.nd_banded = False
    eb, gb = model.elbo_and_grad()
    ed, gd = dense.elbo_and_grad()
    assert abs(eb - ed) <= 1e-10 * abs(ed)
    np.testing.assert_allclose(gb, gd, rtol=1e-7, atol=1e-7 * np.max(np.abs(gd)))
    oe, og = O.elbo_grad_kron([O.Basis(1, 0, 1, m) for m in ms], kinds, th, s, X, y)
    assert abs(eb - oe) <= 1e-9 * abs(oe) + 1e-8 * (0.5 * N / s)
    np.testing.assert_allclose(gb, og, rtol=1e-6, atol=1e-6 * np.max(np.abs(og)))
    Xs = rng.uniform(0.01, 0.99, (150, 4))
    mb, vb = model.predict_f(Xs)
    md, vd = dense.predict_f(Xs)
    np.testing.assert_allclose(mb, md, rtol=0, atol=1e-9)
    np.testing.assert_allclose(vb, vd, rtol=0, atol=1e-9)


# ------------------------------------------------------------------------------------------------ additive model
def _additive_case(A, rng, N, specs):
    d = len(specs)
    X = rng.uniform(0.001, 0.999, (N, d))
    y = (np.sin(6 * X[:, 0]) + (X[:, 1] ** 2 if d > 1 else 0) - (np.cos(5 * X[:, 2]) if d > 2 else 0)
         + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    bases = [_mk_basis(A, o, 0, 1, m) for (o, m, _, _, _) in specs]
    obases = [O.Basis(o, 0, 1, m) for (o, m, _, _, _) in specs]
    kerns = [_kernel(A, kd, v, l) for (_, _, kd, v, l) in specs]
    kinds = [kd for (_, _, kd, _, _) in specs]
    thetas = [(v, l) for (_, _, _, v, l) in specs]
    return X, y, bases, obases, kerns, kinds, thetas


@pytest.mark.parametrize("specs,N", [([(3, 20, 1, 1.0, 0.3), (3, 24, 1, 0.8, 0.5), (3, 16, 0, 1.2, 0.4)], 3001),
                                     ([(4, 30, 2, 0.9, 0.4), (4, 17, 1, 1.1, 0.6)], 5000),
                                     ([(2, 12, 0, 1.0, 0.5), (2, 9, 1, 0.7, 0.3), (2, 10, 1, 1.0, 0.4), (2, 11, 0, 0.5, 0.8)], 800)])
def test_additive_statistics_elbo_predict_vs_oracle(A, specs, N):
    """GPR_additive (gpr.py:139-236): vstack'ed design matrix -> banded diagonal + dense cross blocks; bound and
    posterior against the dense oracle."""
    rng = np.random.default_rng(N)
    X, y, bases, obases, kerns, kinds, thetas = _additive_case(A, rng, N, specs)
    s = 0.05
    model = A.GPR_additive((X, y), kerns, bases)
    model.likelihood.variance.assign(s)
    Aref, bref, yy = O.additive_stats(obases, X, y)
    got = model.KufKfu.cpu().numpy()
    assert got.shape == Aref.shape
    assert np.max(np.abs(got - Aref)) <= 1e-12 * np.max(np.abs(Aref))
    assert (got[Aref == 0] == 0).all()                      # outside the bands of the diagonal blocks
    np.testing.assert_allclose(model.Kuf_y.cpu().numpy(), bref, rtol=0, atol=1e-12 * np.max(np.abs(bref)))
    assert abs(model.tr_yTy.item() - yy) <= 1e-12 * yy
    oe, _ = O.elbo_additive(obases, kinds, thetas, s, X, y)
    e = model.elbo().item()
    vs = sum(v for v, _ in thetas)
    assert abs(e - oe) <= elbo_tol(oe, N, vs, s, yy), (e, oe)
    assert abs(model.maximum_log_likelihood_objective().item() - e) == 0 and model.training_loss().item() == -e
    Xs = rng.uniform(0.01, 0.99, (300, len(specs)))
    om, ov = O.predict_f_additive(obases, kinds, thetas, s, X, y, Xs)
    mean, var = model.predict_f(Xs)
    assert mean.shape == (300, 1) and var.shape == (300, 1)
    np.testing.assert_allclose(mean, om, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var, ov, rtol=0, atol=1e-8)


@pytest.mark.parametrize("specs,N", [([(3, 20, 1, 1.0, 0.3), (3, 24, 2, 0.8, 0.5), (3, 16, 0, 1.2, 0.4)], 3001),
                                     ([(4, 30, 2, 0.9, 0.4), (4, 17, 1, 1.1, 0.6)], 5000)])
def test_additive_analytic_gradient_vs_oracle_differences(A, specs, N):
    """GPR_additive.elbo_and_grad: the bound equals elbo(), and the analytic gradient (what TF autodiff through gpr.py:177-209
    delivers) matches central differences of the dense oracle bound to 1e-6 relative to the largest component."""
    rng = np.random.default_rng(N + 1)
    X, y, bases, obases, kerns, kinds, thetas = _additive_case(A, rng, N, specs)
    s = 0.05
    model = A.GPR_additive((X, y), kerns, bases)
    model.likelihood.variance.assign(s)
    e, g = model.elbo_and_grad()
    assert abs(e.item() - model.elbo().item()) <= 1e-10 * abs(e.item())
    g = g.cpu().numpy()
    flat = [t for th in thetas for t in th] + [s]
    ref = np.zeros(len(flat))
    for i in range(len(flat)):
        h = 1e-5 * flat[i]
        vals = []
        for sgn in (+1, -1):
            f2 = list(flat)
            f2[i] += sgn * h
            th2 = [(f2[2 * j], f2[2 * j + 1]) for j in range(len(specs))]
            vals.append(O.elbo_additive(obases, kinds, th2, f2[-1], X, y)[0])
        ref[i] = (vals[0] - vals[1]) / (2 * h)
    assert np.max(np.abs(g - ref)) <= 1e-6 * np.max(np.abs(ref)), (g, ref)
    res = model.fit(maxiter=3)                                  # the optimiser runs on the analytic gradient
    assert np.isfinite(res.fun)


def test_additive_cross_block_beyond_lds_and_errors(A):
    """m_i * m_j above the LDS image limit takes the L2-atomic path; the block equals Phi_i Phi_j^T either way."""
    rng = np.random.default_rng(5)
    N = 20000
    X = rng.uniform(0.001, 0.999, (N, 2))
    y = rng.normal(size=(N, 1))
    bases = [A.B3Spline(0, 1, 150), A.B3Spline(0, 1, 140)]          # 21000 doubles > 160 KB
    model = A.GPR_additive((X, y), [A.Matern32(), A.Matern32()], bases)
    P0 = bases[0].evaluate_basis(dev(X[:, :1].copy()), sparse=False)
    P1 = bases[1].evaluate_basis(dev(X[:, 1:].copy()), sparse=False)
    C = (P0 @ P1.t()).cpu().numpy()
    got = model.KufKfu.cpu().numpy()[:150, 150:]
    assert np.max(np.abs(got - C)) <= 1e-12 * np.max(np.abs(C))
    small = A.GPR_additive((X, y), [A.Matern32(), A.Matern32()], [A.B3Spline(0, 1, 50), A.B3Spline(0, 1, 40)])
    Q0 = small.bases[0].evaluate_basis(dev(X[:, :1].copy()), sparse=False)
    Q1 = small.bases[1].evaluate_basis(dev(X[:, 1:].copy()), sparse=False)
    C2 = (Q0 @ Q1.t()).cpu().numpy()
    assert np.max(np.abs(small.KufKfu.cpu().numpy()[:50, 50:] - C2)) <= 1e-12 * np.max(np.abs(C2))
    with pytest.raises(AssertionError):
        A.GPR_additive((X, y), [A.Matern32()], bases)                              # gpr.py:147
    with pytest.raises(AssertionError):
        A.GPR_additive((X, y), [A.Matern32(), A.Matern32()], [A.B3Spline(0, 1, 20), A.B2Spline(0, 1, 20)])   # gpr.py:165


def test_additive_fit_improves_bound(A):
    rng = np.random.default_rng(3)
    N = 1200
    X = rng.uniform(0.001, 0.999, (N, 2))
    y = (np.sin(6 * X[:, :1]) + np.cos(4 * X[:, 1:]) + 0.1 * rng.normal(size=(N, 1)))
    model = A.GPR_additive((X, y), [A.Matern32(), A.Matern32()], [A.B3Spline(0, 1, 14), A.B3Spline(0, 1, 15)])
    e0 = model.elbo().item()
    model.fit(maxiter=12)
    e1 = model.elbo().item()
    assert e1 > e0 + 1.0 and np.isfinite(e1)
    assert float(model.likelihood.variance) < 0.5


# ------------------------------------------------------------------------------------------------ experiment adapters
def test_experiment_adapters_metrics_and_timing_rows(A):
    """electricity.py:128-141 / eNATL60.py:82-123 scaffolding on synthetic stand-ins: metric definitions, timing keys."""
    from asvgp_amd import experiments as E
    from scipy.stats import norm
    rng = np.random.default_rng(0)
    t, m, v = rng.normal(size=50), rng.normal(size=50), rng.uniform(0.1, 2, 50)
    assert abs(E.NLL(t, m, v) - float(np.mean(-norm.logpdf(t, loc=m, scale=np.sqrt(v))))) < 1e-12   # eNATL60.py:33-36
    assert abs(E.MSE(t, m) - float(np.mean((t - m) ** 2))) < 1e-15
    X = rng.uniform(0.001, 0.999, (6000, 1))
    y = np.sin(12 * X) + 0.1 * rng.normal(size=X.shape)
    Xtr, Xte, ytr, yte = E.train_test_split(X, y, test_size=0.05, random_state=1)
    assert Xte.shape[0] == 300 and Xtr.shape[0] == 5700
    r = E.run_band_gpr_1d(Xtr, ytr, Xte, yte, A.Matern52(), A.B3Spline(0, 1, 60), maxiter=60)
    assert r["mse"] < 0.02 and r["nlpd"] < 0.0 and r["total_time"] >= r["opt_time"] > 0
    Xk, yk = E.synthetic_ssh(5000)
    rk = E.run_kron(Xk[:4500], yk[:4500], Xk[4500:], yk[4500:], [A.Matern32(), A.Matern32()],
                    [A.B3Spline(-80, -25, 12), A.B3Spline(15, 55, 12)], maxiter=8, predict_chunk=200)
    for key in ("num_train", "num_test", "spline_order", "time_precomp", "time_opt", "time_total", "nll", "mse", "GP"):
        assert key in rk
    assert rk["num_test"] == 500 and rk["spline_order"] == 3 and rk["mse"] < 0.05 and np.isfinite(rk["nll"])


def test_utils_and_kronecker_helper_mirrors(A):
    """utils.py:45-57 and kronecker.py:7-33 helper functions the reference calls (its unused band_to_tfband, band_to_kron_band and
    kron_log_determinant are out of scope): dense / sparse Kronecker forms."""
    from asvgp_amd import kronecker as KR, utils as U
    b1, b2 = A.B3Spline(0, 1, 9), A.B3Spline(0, 1, 9)
    k1, k2 = A.Matern32(variance=0.9, lengthscales=0.4), A.Matern32(variance=1.1, lengthscales=0.3)
    K1 = A.SplineFeatures1D(k1, b1).make_Kuu(k1)
    K2 = A.SplineFeatures1D(k2, b2).make_Kuu(k2)
    o1, o2 = O.Basis(3, 0, 1, 9), O.Basis(3, 0, 1, 9)
    D1 = O.band_to_dense_sym(O.make_Kuu(o1, 1, 0.9, 0.4))
    D2 = O.band_to_dense_sym(O.make_Kuu(o2, 1, 1.1, 0.3))
    Kd, Ld = U.bands_to_kron_cholesky([K1, K2], 3)
    ref = np.kron(D1, D2)
    np.testing.assert_allclose(Kd.cpu().numpy(), ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())
    refL = np.kron(np.linalg.cholesky(D1), np.linalg.cholesky(D2))
    np.testing.assert_allclose(Ld.cpu().numpy(), refL, rtol=1e-9, atol=1e-9 * np.abs(refL).max())
    np.testing.assert_allclose(U.bands_to_sparse([K1, K2], 3).to_dense().cpu().numpy(), ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())
    # generic sparse Khatri-Rao route == fused kernel == reference fixture semantics (row i1*m2 + i2)
    rng = np.random.default_rng(4)
    X = rng.uniform(0.01, 0.99, (40, 2))
    P1 = b1.evaluate_basis(dev(X[:, :1].copy()))
    P2 = b2.evaluate_basis(dev(X[:, 1:].copy()))
    generic = KR.make_kvs_two_sparse(P1, P2).to_dense().cpu().numpy()
    fused = KR.make_kvs_sparse([b1, b2], dev(X)).to_dense().cpu().numpy()
    np.testing.assert_allclose(generic, fused, rtol=1e-14, atol=1e-16)
    rep = KR.sparse_repeats(P1, 3).to_dense().cpu().numpy()
    np.testing.assert_array_equal(rep, np.repeat(P1.to_dense().cpu().numpy(), 3, axis=0))
    til = KR.sparse_tile(P1, 3).to_dense().cpu().numpy()
    np.testing.assert_array_equal(til, np.tile(P1.to_dense().cpu().numpy(), (3, 1)))
    model = A.GPR_kron((X, np.sin(5 * X[:, :1])), [k1, k2], [b1, b2])
    m1, v1 = model.predict_f(X[:7])
    m2, v2 = model.predict_f_sparse(X[:7])
    np.testing.assert_array_equal(m1, m2)
    assert v2.shape == (7, 1) and np.array_equal(v1[:, :1], v2)


def test_phi_fixed_point_scale_fallback(A):
    """The fixed-point Phi y scale is guessed from each workgroup's first tile; values beyond it (here 1e6 times larger,
    in the later part of every shard, and an all-zero first tile) must take the fp64 path and give the same sums."""
    rng = np.random.default_rng(21)
    N, M = 300000, 200
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    ob = O.Basis(4, 0, 1, M)
    bs = A.B4Spline(0, 1, M)
    for kind in ("late_large", "zero_head", "zero_head_tiny", "tiny"):
        y = 1e-3 * rng.normal(size=N)
        if kind == "zero_head_tiny":          # nothing to scale by in the first tile, small values afterwards (found by tests/sweeps/fuzz_phi.py)
            y[:4096] = 0.0
            y[4096:] = 1e-7 * rng.normal(size=N - 4096)
        elif kind == "late_large":
            y[N // 2:] *= 1e6
            y[5000::7] = -3e4
        elif kind == "zero_head":
            y[:4096] = 0.0
            y[4096:] = rng.normal(size=N - 4096) * 50
        else:
            y *= 1e-200
        m = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern12(), bs)
        band, rhs, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
        np.testing.assert_allclose(m.Kuf_y.cpu().numpy(), rhs, rtol=0, atol=1e-12 * np.max(np.abs(rhs)), err_msg=kind)
        assert np.max(np.abs(m.KufKfu.cpu().numpy() - band)) <= 1e-12 * np.max(np.abs(band))
        assert abs(m.tr_yTy.item() - yy) <= 1e-12 * yy
    for n1, yv in ((1, 1e-7), (3, -3e-5)):   # the odd tail point alone must set the scale
        m = A.GPR_1d((x[:n1].reshape(-1, 1), np.full((n1, 1), yv)), A.Matern12(), bs)
        band, rhs, yy = O.sufficient_stats_direct(ob, x[:n1], np.full((n1, 1), yv))
        np.testing.assert_allclose(m.Kuf_y.cpu().numpy(), rhs, rtol=0, atol=1e-12 * np.max(np.abs(rhs)))
    y = rng.normal(size=N)
    y[123456] = np.nan
    m = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern12(), bs)
    assert torch.isnan(m.Kuf_y).any() and torch.isnan(m.tr_yTy)      # a NaN observation is not silently dropped


@pytest.mark.parametrize("algo", [1, 3, 5, 6])
@pytest.mark.parametrize("bad", [1.5, -0.25, float("nan")])
def test_point_outside_the_mesh_is_reported_not_wrapped(A, algo, bad):
    """C-ABI level (the model classes refuse such X first, gpr.py:25-26): an x outside (a, b), or NaN, must not wrap the
    fixed-point image into finite garbage - every Phi algorithm reports it as NaN y^T y."""
    rng = np.random.default_rng(5)
    N, M = 50_000, 256
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = rng.normal(size=N)
    try:
        A.set_phi_algorithm(algo)
        m = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern12(), A.B4Spline(0, 1, M))
        assert torch.isfinite(m.tr_yTy)
        m.X[N // 3] = bad                            # behind the constructor's check
        m.phi_pass()
        assert torch.isnan(m.tr_yTy)
    finally:
        A.set_phi_algorithm(0)


def test_kron_full_size_properties_config4_and_config5_shapes(A):
    """BASELINE config 4 (N=1M, 128x128, k=3) and the eNATL60 stand-in shape (B4Spline(-80,-25,100) x B4Spline(15,55,100),
    eNATL60.py:84) at full basis size: size-independent properties instead of an O(M_tot^3) oracle."""
    from asvgp_amd import experiments as E
    g = torch.Generator(device="cuda").manual_seed(5)
    N = 1_000_000
    X = torch.rand((N, 2), dtype=torch.float64, device="cuda", generator=g) * (1 - 2e-6) + 1e-6
    y = (torch.sin(12 * X[:, :1]) * torch.cos(9 * X[:, 1:]) + 0.1 * torch.randn((N, 1), dtype=torch.float64, device="cuda", generator=g))
    bases = [A.B3Spline(0, 1, 128), A.B3Spline(0, 1, 128)]
    model = A.GPR_kron((X, y), [A.Matern32(variance=1.0, lengthscales=0.2), A.Matern32(variance=1.0, lengthscales=0.2)], bases)
    model.likelihood.variance.assign(0.01)
    k, m1, m2 = 3, 128, 128
    blk = model.KufKfu_blockband
    # partition of unity in both dimensions: 1^T A 1 = N, 1^T Kuf y = sum y  (diagonal offset (0,0) once, all others twice)
    tot = blk[0].sum() + 2 * blk[1:].sum()
    assert abs(tot.item() - N) <= 1e-9 * N
    assert abs(model.Kuf_y.sum().item() - y.sum().item()) <= 1e-9 * y.abs().sum().item()
    assert abs(model.tr_yTy.item() - (y * y).sum().item()) <= 1e-12 * (y * y).sum().item()
    assert (blk[0] >= 0).all()
    # linearity over N-shards (what the multi-GPU all-reduce relies on)
    h = N // 2
    kerns = [A.Matern32(), A.Matern32()]
    s1 = A.GPR_kron((X[:h], y[:h]), kerns, bases)._stats
    s2 = A.GPR_kron((X[h:], y[h:]), kerns, bases)._stats
    assert ((s1 + s2) - model._stats).abs().max().item() <= 1e-11 * model._stats.abs().max().item()
    # the bound is finite, below the exact-GP upper limit trivially implied by its terms, and invariant to a point shuffle
    e = model.elbo().item()
    assert np.isfinite(e)
    perm = torch.randperm(N, device="cuda", generator=g)
    m2_ = A.GPR_kron((X[perm].contiguous(), y[perm].contiguous()), model.kernels, bases)
    m2_.likelihood.variance.assign(0.01)
    assert abs(m2_.elbo().item() - e) <= 1e-9 * abs(e) + 5e-10 * (0.5 * N / 0.01)
    # the bound itself against the CPU banded oracle at the full 128 x 128 basis (LAPACK band Cholesky, bandwidth 387): N = 200k
    Ns = 200_000
    ms = A.GPR_kron((X[:Ns], y[:Ns]), [A.Matern32(variance=1.0, lengthscales=0.2), A.Matern32(variance=1.0, lengthscales=0.2)], bases)
    ms.likelihood.variance.assign(0.05)
    ob = [O.Basis(3, 0, 1, 128), O.Basis(3, 0, 1, 128)]
    Xh, yh = X[:Ns].cpu().numpy(), y[:Ns].cpu().numpy()
    oe = O.elbo_kron_banded(ob, [1, 1], [(1.0, 0.2), (1.0, 0.2)], 0.05, Xh, yh)
    es, gs = ms.elbo_and_grad()
    assert abs(es - oe) <= elbo_tol(oe, Ns, 1.0, 0.05, float((y[:Ns] * y[:Ns]).sum().item())), (es, oe)
    assert abs(ms.elbo().item() - oe) <= elbo_tol(oe, Ns, 1.0, 0.05, float((y[:Ns] * y[:Ns]).sum().item()))
    hfd = 1e-5 * 0.05                                            # d / d sigma^2 by central differences of the banded oracle (the others: small-grid test)
    fd_s = (O.elbo_kron_banded(ob, [1, 1], [(1.0, 0.2), (1.0, 0.2)], 0.05 + hfd, Xh, yh) -
            O.elbo_kron_banded(ob, [1, 1], [(1.0, 0.2), (1.0, 0.2)], 0.05 - hfd, Xh, yh)) / (2 * hfd)
    assert abs(gs[4] - fd_s) <= 1e-5 * abs(fd_s)
    # posterior at 2000 test points: mean close to the noise-free function, variance within (0, prior]
    Xs = torch.rand((2000, 2), dtype=torch.float64, device="cuda", generator=g) * 0.9 + 0.05
    mean, var = model.predict_f(Xs)
    f = (torch.sin(12 * Xs[:, :1]) * torch.cos(9 * Xs[:, 1:])).cpu().numpy()
    assert np.sqrt(np.mean((mean - f) ** 2)) < 0.05
    assert (var > 0).all() and (var < 1.0).all()
    # eNATL60 stand-in shape at reduced N
    Xk, yk = E.synthetic_ssh(200_000)
    mk = A.GPR_kron((Xk, yk), [A.Matern32(variance=0.1, lengthscales=8.0), A.Matern32(variance=1.0, lengthscales=8.0)],
                    [A.B4Spline(-80, -25, 100), A.B4Spline(15, 55, 100)])
    mk.likelihood.variance.assign(1e-3)
    assert mk.order == 4 and mk.Mtot == 10_000 and mk.true_bandwidth == 4 * 101
    blk = mk.KufKfu_blockband
    assert abs((blk[0].sum() + 2 * blk[1:].sum()).item() - 200_000) <= 1e-9 * 200_000
    ek = mk.elbo().item()
    assert np.isfinite(ek)
    mean, var = mk.predict_f(Xk[:3000])
    assert E.MSE(yk[:3000], mean) < 5e-3 and (var > 0).all()


@pytest.mark.parametrize("order,m1,m2,N", [(3, 10, 11, 1500), (2, 30, 8, 2500), (4, 14, 13, 3000)])
def test_kron_selected_inverse_and_analytic_gradient(A, order, m1, m2, N):
    """Band-restricted inverse of P through dense super-blocks vs the dense inverse; analytic gradient of the bound
    (TF autodiff in the reference, eNATL60.py:89) vs the analytic dense gradient of the oracle, gate 1e-6."""
    rng = np.random.default_rng(m1 * 31 + m2)
    X = np.stack([rng.uniform(0.001, 0.999, N), rng.uniform(-0.999, 1.999, N)], axis=1)
    y = (np.sin(6 * X[:, :1]) * np.cos(2 * X[:, 1:]) + 0.1 * rng.normal(size=(N, 1)))
    B = getattr(A, "B%dSpline" % order)
    bases = [B(0, 1, m1), B(-1, 2, m2)]
    th = [(1.1, 0.3), (0.7, 0.6)]
    s = 0.05
    kerns = [A.Matern32(variance=th[0][0], lengthscales=th[0][1]), A.Matern32(variance=th[1][0], lengthscales=th[1][1])]
    model = A.GPR_kron((X, y), kerns, bases)
    model.likelihood.variance.assign(s)
    model.twisted = False                                      # (the one-sided factorisation and its block layout; the two-sided one has its own test)
    obases = [O.Basis(order, 0, 1, m1), O.Basis(order, -1, 2, m2)]
    oe, parts = O.elbo_kron(obases, [1, 1], th, s, X, y)
    # selected inverse
    f = model._factor(want_alpha=True)
    SigD, SigS, Bb = model._selinv(f)
    Pinv = np.linalg.inv(parts["P"])
    M = m1 * m2
    nblk = (M + Bb - 1) // Bb
    assert nblk >= 2
    pad = nblk * Bb
    Pi = np.eye(pad)
    Pi[:M, :M] = Pinv
    sc = np.max(np.abs(Pinv))
    for i in range(nblk):
        np.testing.assert_allclose(SigD[i].cpu().numpy(), Pi[i * Bb:(i + 1) * Bb, i * Bb:(i + 1) * Bb], rtol=0, atol=1e-9 * sc)
        if i + 1 < nblk:
            np.testing.assert_allclose(SigS[i].cpu().numpy(), Pi[(i + 1) * Bb:(i + 2) * Bb, i * Bb:(i + 1) * Bb], rtol=0, atol=1e-9 * sc)
    # bound and gradient
    e, g = model.elbo_and_grad()
    yy = float(np.sum(y * y))
    assert abs(e - oe) <= elbo_tol(oe, N, th[0][0] * th[1][0], s, yy, bcr=True)
    assert abs(e - model.elbo().item()) <= 1e-9 * abs(e)

    _, og = O.elbo_grad_kron(obases, [1, 1], th, s, X, y)     # the analytic dense gradient (what TF autodiff gives the reference)
    np.testing.assert_allclose(g, og, rtol=1e-6, atol=1e-6 * np.max(np.abs(og)))


def test_kron_fp32_storage_gives_the_statistics_of_the_upcast_data(A):
    """BASELINE configs[3] names fp32 data: a GPR_kron built from float32 (X, y) streams a float32 cell-sorted copy through the Phi pass
    (asvgp_phi_accumulate_kron2d_sorted_f32, 12 B per point) - widened exactly in registers, fp64 arithmetic: the statistics are bit
    for bit those of the model built from the upcast float64 data (bound and gradient: the same to rounding), and the oracle's bound
    within the suite's gate."""
    rng = np.random.default_rng(77)
    for order, m1, m2, N in ((3, 20, 17, 30_000), (4, 14, 15, 9_001), (1, 9, 8, 700)):
        X32 = np.stack([rng.uniform(0.001, 0.999, N), rng.uniform(-0.999, 1.999, N)], axis=1).astype(np.float32)
        y32 = (np.sin(6 * X32[:, :1]) * np.cos(2 * X32[:, 1:]) + 0.1 * rng.normal(size=(N, 1))).astype(np.float32)
        B = getattr(A, "B%dSpline" % order)
        Kern, kind = (A.Matern12, 0) if order == 1 else (A.Matern32, 1)
        mk = lambda: [Kern(variance=1.1, lengthscales=0.3), Kern(variance=0.7, lengthscales=0.6)]
        m32 = A.GPR_kron((torch.from_numpy(X32), torch.from_numpy(y32)), mk(), [B(0, 1, m1), B(-1, 2, m2)])
        m64 = A.GPR_kron((torch.from_numpy(X32.astype(np.float64)), torch.from_numpy(y32.astype(np.float64))), mk(), [B(0, 1, m1), B(-1, 2, m2)])
        assert m32._fp32_storage and not m64._fp32_storage
        assert m32._sorted[0].dtype == torch.float32 and m64._sorted[0].dtype == torch.float64
        assert torch.equal(m32._stats[:-1], m64._stats[:-1])                  # block band and rhs: the same numbers in the same order
        assert abs(m32._stats[-1].item() - m64._stats[-1].item()) <= 1e-14 * m64._stats[-1].item()   # (y^T y: per-workgroup sums added with atomics)
        for m in (m32, m64):
            m.likelihood.variance.assign(0.05)
        # (the evaluation itself sums its trace / log-det terms with atomics: the same statistics give the bound to rounding, not to the bit)
        assert abs(m32.elbo().item() - m64.elbo().item()) <= 1e-12 * abs(m64.elbo().item())
        e32, g32 = m32.elbo_and_grad()
        e64, g64 = m64.elbo_and_grad()
        assert abs(e32 - e64) <= 1e-12 * abs(e64)
        np.testing.assert_allclose(g32, g64, rtol=1e-10, atol=1e-10 * np.max(np.abs(g64)))
        Xh, yh = X32.astype(np.float64), y32.astype(np.float64)
        oe, _ = O.elbo_kron([O.Basis(order, 0, 1, m1), O.Basis(order, -1, 2, m2)], [kind, kind], [(1.1, 0.3), (0.7, 0.6)], 0.05, Xh, yh)
        assert abs(e32 - oe) <= elbo_tol(oe, N, 1.1 * 0.7, 0.05, float(np.sum(yh * yh)), bcr=True)


@pytest.mark.parametrize("order,m1,m2,N", [(2, 40, 12, 6000), (3, 30, 9, 5000), (4, 26, 14, 7000), (1, 50, 31, 8000)])
def test_kron_two_sided_factorisation_equals_the_one_sided_and_the_dense_oracle(A, order, m1, m2, N):
    """The two-sided ("twisted") band Cholesky of P - top system and reversed bottom system factored concurrently, joined by the
    separator's Schur complement - against (a) the dense oracle: bound (the suite's gate) and analytic gradient (1e-6), the
    band-restricted inverse entry by entry from the two block stacks (1e-9 of the largest entry of P^-1), alpha; (b) the
    one-sided factorisation of the same model: bound 1e-11, gradient 1e-8, posterior mean / variance 1e-10."""
    rng = np.random.default_rng(m1 * 131 + m2)
    X = np.stack([rng.uniform(0.001, 0.999, N), rng.uniform(-0.999, 1.999, N)], axis=1)
    y = (np.sin(6 * X[:, :1]) * np.cos(2 * X[:, 1:]) + 0.1 * rng.normal(size=(N, 1)))
    B = getattr(A, "B%dSpline" % order)
    bases = [B(0, 1, m1), B(-1, 2, m2)]
    th = [(1.1, 0.3), (0.7, 0.6)]
    s = 0.05
    Kern, kind = (A.Matern12, 0) if order == 1 else (A.Matern32, 1)
    mk = lambda: [Kern(variance=th[0][0], lengthscales=th[0][1]), Kern(variance=th[1][0], lengthscales=th[1][1])]
    model = A.GPR_kron((X, y), mk(), bases)
    model.likelihood.variance.assign(s)
    lay = model._twist_layout()
    assert lay is not None and lay["nb"] >= 3, lay
    plain = A.GPR_kron((X, y), mk(), bases)
    plain.likelihood.variance.assign(s)
    plain.twisted = False
    assert plain._twist_layout() is None
    obases = [O.Basis(order, 0, 1, m1), O.Basis(order, -1, 2, m2)]
    oe, parts = O.elbo_kron(obases, [kind, kind], th, s, X, y)
    e, g = model.elbo_and_grad()
    ep, gp = plain.elbo_and_grad()
    yy = float(np.sum(y * y))
    assert abs(e - oe) <= elbo_tol(oe, N, th[0][0] * th[1][0], s, yy, bcr=True), (e, oe)
    assert abs(model.elbo().item() - e) <= 1e-11 * abs(e)
    assert abs(e - ep) <= 1e-11 * abs(ep), (e, ep)
    np.testing.assert_allclose(g, gp, rtol=1e-8, atol=1e-8 * np.max(np.abs(gp)))
    _, og = O.elbo_grad_kron(obases, [kind, kind], th, s, X, y)
    np.testing.assert_allclose(g, og, rtol=1e-6, atol=1e-6 * np.max(np.abs(og)))
    # the two block stacks against the dense inverse
    f = model._factor(want_alpha=False)
    SigD, SigS, Bb = model._selinv(f)
    nb, top_end, padt, padb = lay["nb"], lay["top_end"], lay["padt"], lay["padb"]
    M = m1 * m2
    Pinv = np.linalg.inv(parts["P"])
    sc = np.max(np.abs(Pinv))
    top = np.eye(nb * Bb); top[padt:, padt:] = Pinv[:top_end, :top_end]
    h = top_end - Bb
    rev = np.eye(nb * Bb); rev[padb:, padb:] = Pinv[h:, h:][::-1, ::-1]
    for st, ref in enumerate((top, rev)):
        for i in range(nb):
            np.testing.assert_allclose(SigD[st, i].cpu().numpy(), ref[i * Bb:(i + 1) * Bb, i * Bb:(i + 1) * Bb], rtol=0, atol=1e-9 * sc)
            if i + 1 < nb:
                np.testing.assert_allclose(SigS[st, i].cpu().numpy(), ref[(i + 1) * Bb:(i + 2) * Bb, i * Bb:(i + 1) * Bb], rtol=0, atol=1e-9 * sc)
    a_ref = Pinv @ np.asarray(parts["b"]).reshape(-1)
    np.testing.assert_allclose(f["alpha"].cpu().numpy() * s, a_ref, rtol=0, atol=1e-9 * np.max(np.abs(a_ref)))
    # posterior: both factorisations, 3000 points (the variance reads the stacks through the twisted accessor)
    Xs = np.stack([rng.uniform(0.01, 0.99, 3000), rng.uniform(-0.99, 1.99, 3000)], axis=1)
    m_t, v_t = model.predict_f(Xs)
    m_p, v_p = plain.predict_f(Xs)
    np.testing.assert_allclose(m_t, m_p, rtol=0, atol=1e-10 * max(1.0, np.max(np.abs(m_p))))
    np.testing.assert_allclose(v_t, v_p, rtol=0, atol=1e-10)
    # a P that is not positive definite is reported, not factored
    bad = A.GPR_kron((X, y), mk(), bases)
    bad.likelihood.variance.assign(s)
    bad._stats[:bad.noff * bad.Mtot].mul_(-1.0)
    with pytest.raises(A.NotPositiveDefiniteError if hasattr(A, "NotPositiveDefiniteError") else Exception):
        bad.elbo()


@pytest.mark.parametrize("order,m1,m2,N", [(1, 7, 9, 3000), (2, 9, 8, 4001), (3, 12, 10, 20000), (4, 14, 15, 30000), (5, 16, 15, 8000),
                                           (6, 17, 18, 6000)])
def test_kron_cell_sorted_phi_pass_equals_per_point_pass(A, order, m1, m2, N):
    """The cell-sorted Khatri-Rao accumulation (default) and the per-point atomic kernel give the same block band, Kuf y
    and y^T y; the cell ids follow the per-dimension index rule (basis.py:58-59) bit for bit."""
    rng = np.random.default_rng(order + m1)
    X = np.stack([rng.uniform(0.001, 0.999, N), rng.uniform(-0.999, 1.999, N)], axis=1)
    X[:50, 0] = 0.5                                   # a heavy cell column
    y = rng.normal(size=(N, 1))
    B = getattr(A, "B%dSpline" % order)
    bases = [B(0, 1, m1), B(-1, 2, m2)]
    model = A.GPR_kron((X, y), [A.Matern12(), A.Matern12()], bases)
    s_sorted = model._stats.clone()
    model.phi_pass(sorted_cells=False)
    s_point = model._stats.clone()
    sc = s_point.abs().max().item()
    assert (s_sorted - s_point).abs().max().item() <= 1e-12 * sc
    assert torch.equal(s_sorted == 0, s_point == 0)
    # cell ids vs the oracle's index rule
    Xs, ys, start = model._sorted
    o1, o2 = O.Basis(order, 0, 1, m1), O.Basis(order, -1, 2, m2)
    i1 = O.neighbour_index(o1.mesh, Xs[:, 0].cpu().numpy())
    i2 = O.neighbour_index(o2.mesh, Xs[:, 1].cpu().numpy())
    cid = i1 * (len(o2.mesh) - 1) + i2
    assert np.all(np.diff(cid) >= 0)
    st = start.cpu().numpy()
    assert st[-1] == N and np.array_equal(np.bincount(cid, minlength=len(st) - 1), np.diff(st))


def test_gpmodel_surface_on_every_model_class(A):
    """trainable_variables / predict_y / predict_log_density (the GPflow GPModel methods the scripts call) exist on all three
    model classes and agree with the posterior moments."""
    from scipy.stats import norm
    rng = np.random.default_rng(8)
    N = 800
    X = rng.uniform(0.01, 0.99, (N, 2))
    y = np.sin(5 * X[:, :1]) + X[:, 1:] + 0.1 * rng.normal(size=(N, 1))
    bases = [A.B3Spline(0, 1, 12), A.B3Spline(0, 1, 11)]
    models = [A.GPR_1d((X[:, :1], y), A.Matern32(), bases[0]),
              A.GPR_kron((X, y), [A.Matern32(), A.Matern32()], bases),
              A.GPR_additive((X, y), [A.Matern32(), A.Matern32()], bases)]
    for m in models:
        d = 1 if isinstance(m, A.GPR_1d) else 2
        m.likelihood.variance.assign(0.05)
        Xs, ys = X[:50, :d], y[:50]
        assert len(m.trainable_variables) == len(m.trainable_parameters) == (3 if d == 1 else 5)
        mf, vf = m.predict_f(Xs)
        my, vy = m.predict_y(Xs)
        np.testing.assert_allclose(my, mf)
        np.testing.assert_allclose(vy, vf + 0.05, rtol=1e-12)
        ld = m.predict_log_density((Xs, ys))
        assert ld.shape == (50,)                                   # gpflow's Gaussian likelihood sums over the output dimension
        np.testing.assert_allclose(ld, norm.logpdf(ys, loc=my, scale=np.sqrt(vy)).sum(-1), rtol=1e-10, atol=1e-12)
        assert isinstance(ld.numpy(), np.ndarray) and isinstance(my.numpy(), np.ndarray)   # electricity.py:132-138 calls .numpy() on both


def test_predict_paths_odd_counts_unaligned_and_staged(A, S):
    """predict_1d: the vector path (two points per lane, 16-B loads), its odd tail, the scalar path for an unaligned input
    slice, and the LDS-staged form (n* >= 65 536) all give the posterior of the banded oracle."""
    X, y = S["X"], S["Y"]
    bs = A.B3Spline(-3.5, 10.5, 40)
    model = A.GPR_1d((X, y), A.Matern32(variance=0.8, lengthscales=1.0), bs)
    model.likelihood.variance.assign(0.08)
    ob = O.Basis(3, -3.5, 10.5, 40)
    Ab, b, yy = O.sufficient_stats_direct(ob, X[:, 0], y)
    rng = np.random.default_rng(9)
    for n in (1, 2, 3, 7, 4096, 65_537, 70_001):
        xs = rng.uniform(-3.4, 10.4, n + 1)
        om, ov = O.predict_f_1d_banded(ob, 1, Ab, b, 0.8, 1.0, 0.08, xs[1:].reshape(-1, 1)) if n <= 4096 else (None, None)
        for view in (dev(xs[1:].copy()), dev(xs)[1:]):              # aligned copy / 8-byte-offset slice
            mean, var = model.predict_f_device(view.reshape(-1, 1))
            assert mean.shape == (n, 1) and var.shape == (n, 1)
            if om is not None:
                np.testing.assert_allclose(mean.cpu().numpy(), om, rtol=0, atol=1e-8)
                np.testing.assert_allclose(var.cpu().numpy(), ov, rtol=0, atol=1e-8)
            else:                                                    # large n: the two load paths must agree bit for bit
                ref = model.predict_f_device(dev(xs[1:].copy()).reshape(-1, 1))
                assert torch.equal(mean, ref[0]) and torch.equal(var, ref[1])
                sub = slice(0, 500)
                om2, ov2 = O.predict_f_1d_banded(ob, 1, Ab, b, 0.8, 1.0, 0.08, xs[1:][sub].reshape(-1, 1))
                np.testing.assert_allclose(mean.cpu().numpy()[sub], om2, rtol=0, atol=1e-8)
                np.testing.assert_allclose(var.cpu().numpy()[sub], ov2, rtol=0, atol=1e-8)


def test_predict_cell_polynomial_kernel_equals_the_table_kernel_and_the_oracle(A):
    """predict_poly_kernel (n* >= 262 144, D = 1, exact-linspace mesh): per-cell variance polynomials built in the LDS.  Against the
    table-read kernel (the same points in chunks below its threshold): means to an ulp, variances to 1e-12 of the prior variance;
    against the banded oracle on a sample: 1e-8 absolute (the gate of every predict test).  Planted points: on knots, on both ends of
    the mesh, an ulp either side of a knot.  A float32-linspace mesh (Python-float end points, basis.py:17) must keep to the table
    kernel and give the same numbers as its own small batches."""
    rng = np.random.default_rng(21)
    for order, m, n_data, nstar in ((4, 2048, 200_000, 400_001), (3, 300, 20_000, 262_144), (1, 50, 5_000, 300_000), (5, 128, 20_000, 262_145)):
        Bs = {1: A.B1Spline, 3: A.B3Spline, 4: A.B4Spline, 5: A.B5Spline}[order]
        x = rng.uniform(1e-9, 1 - 1e-9, n_data); y = np.sin(20 * x) + 0.1 * rng.standard_normal(n_data)
        bs = Bs(0, 1, m)
        kind, Kern = (0, A.Matern12) if order == 1 else (1, A.Matern32)      # (B1 carries the Matern-1/2 bands only, as in the reference)
        model = A.GPR_1d((dev(x).reshape(-1, 1), dev(y).reshape(-1, 1)), Kern(variance=1.3, lengthscales=0.1), bs)
        model.likelihood.variance.assign(0.02)
        xs = rng.uniform(0.0, 1.0, nstar)
        mesh = bs.mesh_np
        planted = np.concatenate([mesh[[0, 1, 2, len(mesh) // 2, -2, -1]], np.nextafter(mesh[3:6], 0.0), np.nextafter(mesh[3:6], 1.0)])
        xs[: len(planted)] = planted
        xd = dev(xs)
        mean, var = model.predict_f_device(xd.reshape(-1, 1))
        assert mean.shape == (nstar, 1) and var.shape == (nstar, 1)
        # the table kernel on the same points, 50 000 at a time (below both thresholds)
        mt, vt = [], []
        for lo in range(0, nstar, 50_000):
            a_, b_ = model.predict_f_device(xd[lo:lo + 50_000].clone().reshape(-1, 1))
            mt.append(a_); vt.append(b_)
        mt, vt = torch.cat(mt), torch.cat(vt)
        dm = (mean - mt).abs().max().item()                                # (the same sum from the same t; hipcc contracts the two kernels' k = 1 code differently: 1 ulp)
        assert dm <= 4e-16 * max(1.0, mt.abs().max().item()), "order %d: means differ from the table kernel by %.3e" % (order, dm)
        dv = (var - vt).abs().max().item()
        assert dv <= 1e-12 * 1.3, "order %d: variances differ from the table kernel by %.3e" % (order, dv)
        ob = O.Basis(order, 0, 1, m)
        Ab, b, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
        sel = np.concatenate([np.arange(len(planted)), rng.integers(0, nstar, 300)])
        om, ov = O.predict_f_1d_banded(ob, kind, Ab, b, 1.3, 0.1, 0.02, xs[sel].reshape(-1, 1))
        np.testing.assert_allclose(mean.cpu().numpy()[sel], om, rtol=0, atol=1e-8)
        np.testing.assert_allclose(var.cpu().numpy()[sel], ov, rtol=0, atol=1e-8)
        model.close()
    # float32-linspace mesh: not an exact fp64 linspace - the handle keeps such a mesh on the table kernel
    x = rng.uniform(-3.4, 10.4, 20_000); y = np.sin(x) + 0.1 * rng.standard_normal(20_000)
    bs = A.B3Spline(-3.5, 10.5, 100)
    model = A.GPR_1d((dev(x).reshape(-1, 1), dev(y).reshape(-1, 1)), A.Matern32(variance=0.8, lengthscales=1.0), bs)
    model.likelihood.variance.assign(0.08)
    xs = dev(rng.uniform(-3.5, 10.5, 300_000))
    mean, var = model.predict_f_device(xs.reshape(-1, 1))
    m2, v2 = model.predict_f_device(xs[:50_000].clone().reshape(-1, 1))
    assert torch.equal(mean[:50_000], m2) and torch.equal(var[:50_000], v2)


# ------------------------------------------------------------------------------------------------ BASELINE headline size
def test_headline_config_parity_n10m(A):
    """BASELINE.json north star, exactly as bench.py runs it: N = 10M, M = 2048, B4, Matern-3/2, theta = (1, 0.05, 0.01),
    default_rng(1234).  Statistics <= 1e-12 of the largest entry against the oracle's direct accumulation; ELBO and gradient
    against the oracle (fp64, reference elimination order) AND the oracle's long-double evaluation of the same recurrences.
    Gate for the bound: |GPU - long double| <= 1e-9 |ELBO|, no allowance for the fp64 oracle's own distance from the
    long-double value (0.018 here, cond(Kuu) = 3.5e7), which is printed as information only."""
    import bench
    N, M = 10_000_000, 2048
    v, l, s = 1.0, 0.05, 0.01
    x, y = bench.synth(N)
    xd, yd = dev(x).reshape(-1, 1), dev(y).reshape(-1, 1)
    model = A.GPR_1d((xd, yd), A.Matern32(variance=v, lengthscales=l), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(s)
    got = model._stats.cpu().numpy()
    r = model.elbo_and_grad().cpu().numpy()
    ob = O.Basis(4, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
    ref = np.concatenate([Ab.reshape(-1), b.reshape(-1), [yy]])
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    oe, og, _ = O.elbo_grad_1d(ob, O.MATERN32, Ab, b, yy, N, v, l, s)
    ee, ge = O.elbo_grad_1d_extended(ob, O.MATERN32, Ab, b, yy, N, v, l, s)
    assert abs(oe - ee) <= 0.05                     # the reference order itself: 0.018 (6e-9 relative)
    gate = 1e-9 * abs(ee)                           # VERDICT r2 #1b: against the long-double value, no allowance (measured: 3e-11 |ELBO|)
    assert abs(r[0] - ee) <= gate, ("ELBO vs long double", r[0], ee, oe, gate)
    print("headline: |GPU - long double| = %.3g (%.2g rel), |fp64 oracle - long double| = %.3g (information)" % (
        abs(r[0] - ee), abs(r[0] - ee) / abs(ee), abs(oe - ee)))
    assert abs(r[0] - oe) <= gate + abs(oe - ee), ("ELBO vs oracle", r[0], oe)
    np.testing.assert_allclose(r[1:4], ge, rtol=1e-6)
    np.testing.assert_allclose(r[1:4], og, rtol=1e-6)
    # sorted (time-series) order of the same points: same statistics (run mode of the Phi pass), same bound
    o = np.argsort(x)
    ms = A.GPR_1d((dev(x[o]).reshape(-1, 1), dev(y[o]).reshape(-1, 1)), A.Matern32(variance=v, lengthscales=l), A.B4Spline(0, 1, M))
    ms.likelihood.variance.assign(s)
    assert np.max(np.abs(ms._stats.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))
    assert abs(ms.elbo_and_grad().cpu().numpy()[0] - ee) <= gate


def test_config3_n10m_m4096_matern52_single_gpu(A):
    """BASELINE config 3 (N = 10M, M = 4096, Matern-5/2, B4) on ONE GPU: the statistics of all 10M points (band-scatter kernel in
    two column chunks - the moment image does not fit M = 4096) to 1e-12, bound and gradient against the oracle and its long-double
    evaluation with the headline gate.  BASELINE names no theta; lengthscale 0.005 (20 cells, cond(Kuu) ~ 1e9) - at the north
    star's 0.05 cond(Kuu) ~ 1e17 and even the oracle's fp64 and long-double values differ by 6e3."""
    N, M = 10_000_000, 4096
    v, l, s = 1.0, 0.005, 0.01
    rng = np.random.default_rng(1)
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
    model = A.GPR_1d((dev(x).reshape(-1, 1), dev(y).reshape(-1, 1)), A.Matern52(variance=v, lengthscales=l), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(s)
    ob = O.Basis(4, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
    ref = np.concatenate([Ab.reshape(-1), b.reshape(-1), [yy]])
    assert np.max(np.abs(model._stats.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))
    r = model.elbo_and_grad().cpu().numpy()
    oe, og, _ = O.elbo_grad_1d(ob, O.MATERN52, Ab, b, yy, N, v, l, s)
    ee, ge = O.elbo_grad_1d_extended(ob, O.MATERN52, Ab, b, yy, N, v, l, s)
    gate = 1e-9 * abs(ee)                           # VERDICT r2 #1b: against the long-double value, no allowance
    assert abs(r[0] - ee) <= gate, (r[0], ee, oe, gate)
    print("config 3: |GPU - long double| = %.3g (%.2g rel), |fp64 oracle - long double| = %.3g (information)" % (
        abs(r[0] - ee), abs(r[0] - ee) / abs(ee), abs(oe - ee)))
    np.testing.assert_allclose(r[1:4], ge, rtol=5e-6)       # (1.6e-6 for every elimination order, the sequential one included: cond 1e9)


@pytest.mark.parametrize("M,k", [(5000, 4), (4099, 4), (2100, 6), (6001, 2), (1500, 8)])
def test_operator_vjps_beyond_the_lds_capacity_walk_the_band_in_segments(A, M, k):
    """The two adjoint recurrences (gradients banded_matrices registers for cholesky_band / inverse_from_cholesky_band, gpr.py:56-59) keep
    their state in registers and stage their inputs through the LDS a segment of columns at a time, so M is not limited by the 160 KB
    (BASELINE config 3: M = 4096): C-ABI VJPs against the oracle's adjoint sweeps for sizes of 2-4 segments and k = 8 (wave-parallel
    / sequential fallbacks of the inverse's adjoint)."""
    from asvgp_amd import banded as Bd
    rng = np.random.default_rng(M + k)
    lower = np.zeros((k + 1, M))
    for d in range(k + 1):
        lower[d, :M - d] = rng.normal(size=M - d) * (0.3 ** d)
    dom = np.sum(np.abs(lower[1:]), axis=0)
    for d in range(1, k + 1):
        dom[d:] += np.abs(lower[d, :M - d])
    lower[0] = np.abs(lower[0]) + dom + 0.5                  # strictly diagonally dominant: positive definite
    L = O.cholesky_band(lower)
    S = O.inverse_from_cholesky_band(L)
    Lbar, Sbar = np.zeros((k + 1, M)), np.zeros((k + 1, M))
    for d in range(k + 1):
        Lbar[d, :M - d] = rng.normal(size=M - d)
        Sbar[d, :M - d] = rng.normal(size=M - d)
    Kt = dev(lower).requires_grad_(True)
    (Bd.cholesky_band(Kt) * dev(Lbar)).sum().backward()
    ref = O.cholesky_band_vjp(L, Lbar)
    assert np.max(np.abs(Kt.grad.cpu().numpy() - ref)) <= 1e-11 * np.max(np.abs(ref))
    Lt = dev(L).requires_grad_(True)
    (Bd.inverse_from_cholesky_band(Lt) * dev(Sbar)).sum().backward()
    ref = O.inverse_from_cholesky_band_vjp(L, S, Sbar)
    assert np.max(np.abs(Lt.grad.cpu().numpy() - ref)) <= 1e-11 * np.max(np.abs(ref))


def test_operator_vjps_vs_oracle_and_reference_style_bound_backpropagates(A):
    """banded_matrices registers gradients for cholesky_band, inverse_from_cholesky_band, solve_triang_mat and product_band_band; the
    reference's optimiser differentiates GPR_1d.elbo through them (gpr.py:56-87, example.py:31-32).  (i) each C-ABI VJP against the
    oracle's adjoint sweeps (themselves checked against dense autograd on CPU); (ii) the bound written op by op exactly as
    gpr.py:49-89 does, on torch scalars that require grad: back-propagation reproduces the fused analytic gradient."""
    from asvgp_amd import banded as Bd
    rng = np.random.default_rng(2)
    M, k = 300, 4
    ob = O.Basis(4, 0, 1, M)
    K = O.make_Kuu(ob, 1, 1.0, 0.05)
    L = O.cholesky_band(K)
    S = O.inverse_from_cholesky_band(L)
    Lbar = rng.standard_normal(L.shape)
    Sbar = rng.standard_normal(L.shape)
    for d in range(1, k + 1):
        Lbar[d, M - d:] = 0
        Sbar[d, M - d:] = 0
    Kt = dev(K).requires_grad_(True)
    Lt = Bd.cholesky_band(Kt)
    (Lt * dev(Lbar)).sum().backward()
    ref = O.cholesky_band_vjp(L, Lbar)
    assert np.max(np.abs(Kt.grad.cpu().numpy() - ref)) <= 1e-9 * np.max(np.abs(ref))
    Lt = dev(L).requires_grad_(True)
    (Bd.inverse_from_cholesky_band(Lt) * dev(Sbar)).sum().backward()
    ref = O.inverse_from_cholesky_band_vjp(L, S, Sbar)
    assert np.max(np.abs(Lt.grad.cpu().numpy() - ref)) <= 1e-9 * np.max(np.abs(ref))
    Bm, Xbar = rng.standard_normal((M, 2)), rng.standard_normal((M, 2))
    for tr in (False, True):
        Lt, Bt = dev(L).requires_grad_(True), dev(Bm).requires_grad_(True)
        (Bd.solve_triang_mat(Lt, Bt, transpose_left=tr) * dev(Xbar)).sum().backward()
        X = O.solve_triang_mat(L, Bm, transpose_left=tr)
        rl, rb = O.solve_triang_mat_vjp(L, X, Xbar, transpose_left=tr)
        assert np.max(np.abs(Lt.grad.cpu().numpy() - rl)) <= 1e-10 * np.max(np.abs(rl))
        assert np.max(np.abs(Bt.grad.cpu().numpy() - rb)) <= 1e-10 * np.max(np.abs(rb))
    Ssym, Asym = O.symmetrise_band(S, k), O.symmetrise_band(K, k)
    Obar = rng.standard_normal(Ssym.shape)
    St, At = dev(Ssym).requires_grad_(True), dev(Asym).requires_grad_(True)
    (Bd.product_band_band(St, At, k, k, k, k, k, k) * dev(Obar)).sum().backward()
    rl, rr = O.product_band_band_vjp(Ssym, Asym, Obar, k, k, k, k, k, k)
    assert np.max(np.abs(St.grad.cpu().numpy() - rl)) <= 1e-12 * np.max(np.abs(rl))
    assert np.max(np.abs(At.grad.cpu().numpy() - rr)) <= 1e-12 * np.max(np.abs(rr))

    # (ii) gpr.py:49-89 op by op
    N = 20000
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.standard_normal(N)).reshape(-1, 1)
    v0, l0, s0 = 1.3, 0.04, 0.02
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=v0, lengthscales=l0), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(s0)
    fused = model.elbo_and_grad().cpu().numpy()
    v = torch.tensor(v0, dtype=torch.float64, device="cuda", requires_grad=True)
    l = torch.tensor(l0, dtype=torch.float64, device="cuda", requires_grad=True)
    sg = torch.tensor(s0, dtype=torch.float64, device="cuda", requires_grad=True)
    St_ = model.inducing_features.static_stack(1)                    # A, B, C, BC, BC_grad (inducing_features.py:25-33)
    s3 = np.sqrt(3.0)
    cs = [s3 / (4 * l * v), l / (2 * s3 * v), l ** 3 / (12 * s3 * v), 1 / (2 * v), l ** 2 / (2 * v)]
    Kuu = sum(c * St_[t] for t, c in enumerate(cs))
    Aband, bvec, yy = model.KufKfu, model.Kuf_y, model.tr_yTy
    Lk = Bd.cholesky_band(Kuu)                                        # gpr.py:56
    logdet_K = torch.log(Lk[0] ** 2).sum()                            # gpr.py:57
    Kinv = Bd.inverse_from_cholesky_band(Lk)                          # gpr.py:59
    prod = Bd.product_band_band(Bd.symmetrise_band(Kinv, k), Bd.symmetrise_band(Aband, k), left_lower_bandwidth=k,
                                left_upper_bandwidth=k, right_lower_bandwidth=k, right_upper_bandwidth=k,
                                result_lower_bandwidth=0, result_upper_bandwidth=0)     # gpr.py:60-70: only the diagonal is summed
    trace = prod.sum()
    P = Aband / sg + Kuu                                              # gpr.py:72
    Lp = Bd.cholesky_band(P)                                          # gpr.py:73
    logdet_P = torch.log(Lp[0] ** 2).sum()
    c = Bd.solve_triang_mat(Lp, bvec) / sg                            # gpr.py:75
    elbo = (-0.5 * N * torch.log(2 * np.pi * sg) - 0.5 * logdet_P + 0.5 * logdet_K - 0.5 * yy / sg + 0.5 * (c ** 2).sum()
            - 0.5 * N * v / sg + 0.5 * trace / sg)                    # gpr.py:78-87
    elbo.backward()
    assert abs(elbo.item() - fused[0]) <= 1e-9 * abs(fused[0])
    got = np.array([v.grad.item(), l.grad.item(), sg.grad.item()])
    np.testing.assert_allclose(got, fused[1:4], rtol=1e-7)


def test_deferred_reduce_gives_the_same_statistics(A):
    """asvgp_set_phi_deferred_reduce / asvgp_phi_reduce_1d: the streaming kernel and the cross-workgroup reduce as two calls (the
    pipelined bench keeps the reduce off its N-side stream); same statistics as the fused call, and a reduce with nothing pending
    is a no-op."""
    rng = np.random.default_rng(11)
    N, M = 300_000, 1024
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
    m = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern32(), A.B4Spline(0, 1, M))
    ref = m._stats.clone()
    m._h.set_phi_deferred_reduce(1)
    m.phi_pass()
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        m.phi_reduce()
        m.phi_reduce()                                   # nothing pending any more: no-op
    torch.cuda.synchronize()
    assert torch.max(torch.abs(m._stats - ref)).item() <= 1e-12 * torch.max(torch.abs(ref)).item()
    m._h.set_phi_deferred_reduce(0)
    m.phi_pass()
    torch.cuda.synchronize()
    assert torch.max(torch.abs(m._stats - ref)).item() <= 1e-12 * torch.max(torch.abs(ref)).item()


def test_bench_self_launch_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no torchrun environment must launch its own workers (the driver's SCALE command);
    rehearsed with gloo and both ranks on the one GPU of the test box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ASVGP_BENCH_BACKEND"] = "gloo"
    env["ASVGP_BENCH_NOWEAK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--points", "400000", "--features", "512"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "roofline" in d


def test_fused_launch_that_gives_up_waiting_falls_back_to_the_multi_launch_path(A):
    """ADVICE r2 / VERDICT r2 #6: a fused ELBO launch whose helper workgroups never report (forced through the library's test hook, with a
    short spin limit) must not return a bound built from incomplete bands: info stays negative (sticky), the model re-arms its
    workspace, re-issues the step through the multi-launch sweeps and counts the event; the next ordinary call works again."""
    rng = np.random.default_rng(3)
    N, M = 20000, 2048
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(0.01)
    good = model.elbo_and_grad().cpu().numpy()
    os.environ["ASVGP_DEBUG_NO_ASSEMBLY"] = "1"
    os.environ["ASVGP_SPIN_LIMIT"] = "20000"
    _reload_env()
    try:
        model._launch_elbo()
        torch.cuda.synchronize()
        assert model._info.tolist()[1] < 0                      # the abort reached the host
        got = model.elbo_and_grad().cpu().numpy()               # aborts again, falls back inside the call
    finally:
        del os.environ["ASVGP_DEBUG_NO_ASSEMBLY"], os.environ["ASVGP_SPIN_LIMIT"]
        _reload_env()
    assert getattr(model, "fused_launch_fallbacks", 0) >= 1
    ob = O.Basis(4, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y)
    oe, og, _ = O.elbo_grad_1d(ob, O.MATERN32, Ab, b, yy, N, 1.0, 0.05, 0.01)
    assert abs(got[0] - oe) <= elbo_tol(oe, N, 1.0, 0.01, yy)
    np.testing.assert_allclose(got[1:4], og, rtol=1e-6)
    again = model.elbo_and_grad().cpu().numpy()                 # the fused path, re-armed
    np.testing.assert_allclose(again, good, rtol=1e-9)
    # VERDICT r3 weak #1d: a give-up that was transient is answered by the SAME fused launch, re-issued once the device is idle - the
    # step then has the fused path's numbers (the ones the headline gate holds), not band algorithm 1's
    os.environ["ASVGP_DEBUG_NO_ASSEMBLY"] = "1"
    os.environ["ASVGP_SPIN_LIMIT"] = "20000"
    _reload_env()
    try:
        model._launch_elbo()
        torch.cuda.synchronize()
        assert model._info.tolist()[1] < 0
    finally:
        del os.environ["ASVGP_DEBUG_NO_ASSEMBLY"], os.environ["ASVGP_SPIN_LIMIT"]
        _reload_env()
    before = model.fused_launch_fallbacks
    model._check_pd(model._launch_elbo)                         # the hook is off again: the retry is the fused launch itself
    assert model.fused_launch_fallbacks == before + 1 and model._h.band_algorithm == 0
    np.testing.assert_allclose(model._out[:4].cpu().numpy(), good, rtol=1e-12)


def test_launch_ahead_of_theta_gives_the_ordinary_launch_numbers(A):
    """asvgp_elbo_grad_ahead_1d / asvgp_elbo_publish_theta (an optimiser's dependent evaluations, example.py:31-32): the launch goes out
    before its theta exists, waits resident on the handle's pinned theta box and computes the SAME bound and gradient as the ordinary
    launch once it has it - for several thetas in a row, on the time-series instantiation's statistics too; a model the matrix-core
    launch does not apply to reports None and launches nothing; a launch left waiting is withdrawn when the handle is closed."""
    rng = np.random.default_rng(11)
    N, M = 60000, 2048
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(0.01)
    for th in ((1.0, 0.05, 0.01), (0.8, 0.03, 0.02), (1.3, 0.08, 0.005)):
        ref = model.read_elbo_host(model.launch_elbo_host(th))
        tok = model.launch_elbo_ahead()
        assert tok                                              # the matrix-core launch applies: a token for the mirror
        time.sleep(0.002)                                       # the kernel is resident and waiting by now
        model.publish_theta(th)
        got = model.read_elbo_host(tok)
        np.testing.assert_allclose(got, ref, rtol=1e-12)
    ob = O.Basis(4, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y)
    ee, ge = O.elbo_grad_1d_extended(ob, O.MATERN32, Ab, b, yy, N, 1.3, 0.08, 0.005)
    assert abs(got[0] - ee) <= 1e-9 * abs(ee)
    np.testing.assert_allclose(got[1:4], ge, rtol=1e-6)
    # a second launch while one waits is refused; closing the handle withdraws the waiting launch (no hang, nothing left on the device)
    tok = model.launch_elbo_ahead()
    with pytest.raises(A._lib.AsvgpError):
        model.launch_elbo_ahead()
    model.close()
    torch.cuda.synchronize()
    # bandwidth 3: no matrix-core launch - nothing is launched ahead
    m3 = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.05), A.B3Spline(0, 1, 200))
    assert m3.launch_elbo_ahead() is None
    r3 = m3.elbo_and_grad_host()
    assert np.isfinite(r3).all()


def test_host_result_mirror_gives_the_stream_path_numbers(A):
    """asvgp_result_mirror (include/asvgp_hip.h): the fused launch writes [out, info, sequence] into pinned host memory and the host polls
    the sequence word; the numbers are the ones the stream path returns, a launch that cannot write the mirror (band algorithm 1)
    reports token 0 and is read through the stream, and an aborted launch (no mirror write) ends in the same fallback as before."""
    rng = np.random.default_rng(5)
    N, M = 30000, 1024
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(0.01)
    dev = model.elbo_and_grad().tolist()
    for rep in range(3):                                            # sequence numbers advance; every read is this launch's
        tok = model.launch_elbo_host()
        assert tok == rep + 1
        host = model.read_elbo_host(tok)
        assert host == dev
    model.kernel.lengthscales.assign(0.07)
    host2 = model.elbo_and_grad_host()
    dev2 = model.elbo_and_grad().tolist()
    assert host2 == dev2 and host2 != dev
    model._h.set_band_algorithm(1)
    try:
        tok = model.launch_elbo_host()
        assert tok == 0
        np.testing.assert_allclose(model.read_elbo_host(tok), dev2, rtol=1e-7)
    finally:
        model._h.set_band_algorithm(0)
    os.environ["ASVGP_DEBUG_NO_ASSEMBLY"] = "1"
    os.environ["ASVGP_SPIN_LIMIT"] = "20000"
    _reload_env()
    try:
        got = model.read_elbo_host(model.launch_elbo_host(), poll_seconds=0.02)
    finally:
        del os.environ["ASVGP_DEBUG_NO_ASSEMBLY"], os.environ["ASVGP_SPIN_LIMIT"]
        _reload_env()
    assert getattr(model, "fused_launch_fallbacks", 0) >= 1
    np.testing.assert_allclose(got, dev2, rtol=1e-7)
    assert model.elbo_and_grad_host() == dev2                       # re-armed


def test_non_positive_definite_data_chain_is_reported_through_the_mirror_path(A):
    """A band buffer that makes P = Kuu + KufKfu / sigma2 indefinite must raise from the host-read path exactly as from elbo_and_grad()."""
    from asvgp_amd import banded
    rng = np.random.default_rng(6)
    N, M = 5000, 512
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = np.sin(20 * x).reshape(-1, 1)
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(0.01)
    assert np.isfinite(model.elbo_and_grad_host()).all()
    model._stats[100] = -1.0e9                                      # diagonal entry 100 of the KufKfu band
    for fn in (model.elbo_and_grad, model.elbo_and_grad_host):
        with pytest.raises(banded.NotPositiveDefiniteError) as ei:
            fn()
        assert "P = Kuu" in str(ei.value)


def test_config2_n1m_m1024_bound_and_gradient_vs_oracle(A):
    """BASELINE config 2 (1D synthetic N = 1M, Matern-3/2, M = 1024, band k = 4, fp64, one GPU): statistics, bound and gradient
    against the oracle and its long-double evaluation (VERDICT r2 #1c)."""
    N, M = 1_000_000, 1024
    v, l, s = 1.0, 0.05, 0.01
    rng = np.random.default_rng(2)
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
    model = A.GPR_1d((dev(x).reshape(-1, 1), dev(y).reshape(-1, 1)), A.Matern32(variance=v, lengthscales=l), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(s)
    ob = O.Basis(4, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
    ref = np.concatenate([Ab.reshape(-1), b.reshape(-1), [yy]])
    assert np.max(np.abs(model._stats.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))
    r = model.elbo_and_grad().cpu().numpy()
    oe, og, _ = O.elbo_grad_1d(ob, O.MATERN32, Ab, b, yy, N, v, l, s)
    ee, ge = O.elbo_grad_1d_extended(ob, O.MATERN32, Ab, b, yy, N, v, l, s)
    assert abs(r[0] - ee) <= 1e-9 * abs(ee), (r[0], ee, oe)
    np.testing.assert_allclose(r[1:4], ge, rtol=1e-6)
    np.testing.assert_allclose(r[1:4], og, rtol=1e-6)
    assert model.elbo_and_grad_host() == r[:4].tolist()
    mean, var = model.predict_f(np.linspace(0.01, 0.99, 500).reshape(-1, 1))
    assert np.sqrt(np.mean((mean[:, 0] - np.sin(20 * np.linspace(0.01, 0.99, 500))) ** 2)) < 0.02 and (var > 0).all()


def test_config5_enatl60_stand_in_at_full_size_on_one_gpu(A):
    """BASELINE config 5 at its full size on ONE GPU (VERDICT r2 #1a): GPR_kron with B4Spline(-80,-25,100) x B4Spline(15,55,100)
    (eNATL60.py:84) over N = 14M synthetic sea-surface points.  Size-independent properties: partition of unity, linearity over
    eight contiguous N-shards (exactly what the 8-GPU all-reduce relies on) to 1e-11, invariance of the bound under a point
    shuffle; the 200k-point oracle comparison lives in test_kron_full_size_properties_config4_and_config5_shapes."""
    from asvgp_amd import experiments as E
    N = 14_000_000
    Xh, yh = E.synthetic_ssh(N)
    X, y = dev(Xh), dev(yh)
    bases = [A.B4Spline(-80, -25, 100), A.B4Spline(15, 55, 100)]
    kerns = lambda: [A.Matern32(variance=0.1, lengthscales=8.0), A.Matern32(variance=1.0, lengthscales=8.0)]
    model = A.GPR_kron((X, y), kerns(), bases)
    model.likelihood.variance.assign(1e-3)
    assert model.order == 4 and model.Mtot == 10_000 and model.true_bandwidth == 4 * 101
    blk = model.KufKfu_blockband
    assert abs((blk[0].sum() + 2 * blk[1:].sum()).item() - N) <= 1e-9 * N
    assert abs(model.Kuf_y.sum().item() - y.sum().item()) <= 1e-9 * y.abs().sum().item()
    assert abs(model.tr_yTy.item() - (y * y).sum().item()) <= 1e-12 * (y * y).sum().item()
    full = model._stats.clone()
    acc = torch.zeros_like(full)
    for r in range(8):
        lo, hi = r * N // 8, (r + 1) * N // 8
        acc += A.GPR_kron((X[lo:hi], y[lo:hi]), kerns(), bases)._stats
    assert (acc - full).abs().max().item() <= 1e-11 * full.abs().max().item()
    e, g = model.elbo_and_grad()
    assert np.isfinite(e) and np.isfinite(np.asarray(g)).all()
    perm = torch.randperm(N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    shuffled = A.GPR_kron((X[perm].contiguous(), y[perm].contiguous()), kerns(), bases)
    shuffled.likelihood.variance.assign(1e-3)
    assert (shuffled._stats - full).abs().max().item() <= 1e-11 * full.abs().max().item()
    e2 = shuffled.elbo().item()
    assert abs(e2 - e) <= 1e-9 * abs(e) + 5e-10 * (0.5 * N * 0.1 / 1e-3), (e, e2)
    mean, var = model.predict_f(X[:5000])
    assert E.MSE(y[:5000], mean) < 2e-3 and (var > 0).all()


def test_kron_boundary_attributes_and_reference_make_kvs_signature(A):
    """GPR_kron exposes Kuf / KufKfu_sparse / KufKfu_dense / KufKfu_band (gpr.py:269-273) as lazy views built from the block
    band, and kronecker.make_kvs_sparse accepts the reference's own signature - a list of per-dimension sparse design matrices."""
    rng = np.random.default_rng(21)
    N, k, m1, m2 = 700, 3, 9, 11
    X = rng.uniform(0.02, 0.98, (N, 2))
    y = np.sin(4 * X[:, :1]) * X[:, 1:] + 0.05 * rng.normal(size=(N, 1))
    bases = [A.B3Spline(0, 1, m1), A.B3Spline(0, 1, m2)]
    model = A.GPR_kron((X, y), [A.Matern32(), A.Matern32()], bases)
    Kuf = model.Kuf.to_dense()
    ref = (Kuf @ Kuf.T).cpu().numpy()
    dense = model.KufKfu_dense.cpu().numpy()
    np.testing.assert_allclose(dense, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_allclose(dense, _blockband_to_dense(model.KufKfu_blockband.cpu().numpy(), k, m1, m2), rtol=0, atol=0)
    sp = model.KufKfu_sparse
    assert sp.layout == torch.sparse_coo and sp.shape == (m1 * m2, m1 * m2)
    np.testing.assert_array_equal(sp.to_dense().cpu().numpy(), dense)
    band = model.KufKfu_band.cpu().numpy()
    assert band.shape == (model.bandwidth + 1, m1 * m2)
    for d in (0, 1, k, m2, m2 + k, model.bandwidth):
        np.testing.assert_array_equal(band[d, :m1 * m2 - d], np.diagonal(dense, -d))
    # the reference's call form: make_kvs_sparse([Kuf_1, Kuf_2]) (gpr.py:268-269)
    A_list = [model.inducing_features[i].make_Kuf(dev(X[:, i:i + 1])) for i in range(2)]
    KR = A.kronecker.make_kvs_sparse(A_list).to_dense().cpu().numpy()
    np.testing.assert_allclose(KR, Kuf.cpu().numpy(), rtol=0, atol=1e-15)
    # three factors through the same reduce
    B3 = A.B3Spline(0, 1, 8)
    A3 = A.SplineFeatures1D(A.Matern32(), B3).make_Kuf(dev(rng.uniform(0.02, 0.98, (N, 1))))
    KR3 = A.kronecker.make_kvs_sparse(A_list + [A3]).to_dense().cpu().numpy()
    a, b, c = (t.to_dense().cpu().numpy() for t in A_list + [A3])
    np.testing.assert_allclose(KR3, np.einsum("in,jn,kn->ijkn", a, b, c).reshape(-1, N), rtol=0, atol=1e-15)


def test_deferred_reduce_is_flushed_by_every_consumer_of_the_statistics(A):
    """with the reduce deferred, a bound must never be built from the zeroed statistics buffer and an additive model's
    per-dimension passes (one handle, one partials workspace) must not overwrite each other's parked partials: the ELBO / posterior
    entry points flush a reduce that targets their buffer, the next accumulate on the handle flushes the previous one."""
    rng = np.random.default_rng(12)
    N, M = 200_000, 512
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
    m = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    m.likelihood.variance.assign(0.01)
    ref_stats = m._stats.clone()
    ref = m.elbo_and_grad().tolist()
    m._h.set_phi_deferred_reduce(1)
    m.phi_pass()                                          # partials parked, statistics buffer zeroed
    np.testing.assert_allclose(m.elbo_and_grad().tolist(), ref, rtol=1e-9)   # the ELBO entry point flushed the reduce first
    assert (m._stats - ref_stats).abs().max().item() <= 1e-12 * ref_stats.abs().max().item()   # (tile sort: reproducible to rounding)
    m.phi_pass()
    mean, var = m.predict_f(np.array([[0.3], [0.6]]))     # posterior_prepare flushes too
    m._h.set_phi_deferred_reduce(0)
    m.phi_pass()
    mean2, var2 = m.predict_f(np.array([[0.3], [0.6]]))
    np.testing.assert_allclose(mean, mean2, rtol=1e-10)
    np.testing.assert_allclose(var, var2, rtol=1e-8)
    # additive model: d accumulate calls on one handle
    Xa = rng.uniform(0.01, 0.99, (20_000, 2))
    ya = np.sin(5 * Xa[:, :1]) + Xa[:, 1:] + 0.1 * rng.normal(size=(20_000, 1))
    add = A.GPR_additive((Xa, ya), [A.Matern32(), A.Matern32()], [A.B3Spline(0, 1, 24), A.B3Spline(0, 1, 20)])
    s_ref = add._stats.clone()
    add._h.set_phi_deferred_reduce(1)
    add.phi_pass()                                        # every per-dimension reduce is out before the cross blocks
    torch.cuda.synchronize()
    assert (add._stats - s_ref).abs().max().item() <= 1e-12 * s_ref.abs().max().item()


def test_deferred_forward_pass_gives_the_same_numbers(A):
    """asvgp_set_deferred_forward_pass / asvgp_prior_publish: the matrix-core launch returns before the host's forward pass of the prior chain;
    the caller publishes it later (bench.py: after enqueueing the next Phi pass).  Same bits as the ordinary call; a forgotten publish is
    made up by the read / the next launch / the teardown, never left to the kernel's bounded wait."""
    rng = np.random.default_rng(9)
    N, M = 40000, 2048
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(0.01)
    ref = model.elbo_and_grad_host()
    model._h.set_deferred_forward_pass(1)
    tok = model.launch_elbo_host()
    model.phi_pass()                                   # other work between launch and publish
    model._h.publish_forward()
    np.testing.assert_allclose(model.read_elbo_host(tok), ref, rtol=1e-9)      # (the Phi pass in between re-sums the statistics: rounding)
    tok = model.launch_elbo_host()                     # publish forgotten: the read makes up for it
    got = model.read_elbo_host(tok)
    assert got == model.elbo_and_grad_host()           # and so does an ordinary call in deferred mode
    tok = model.launch_elbo_host()
    tok2 = model.launch_elbo_host()                    # ... and the next launch
    assert model.read_elbo_host(tok2) == got
    model._h.set_deferred_forward_pass(0)
    assert model.elbo_and_grad_host() == got
    assert getattr(model, "fused_launch_fallbacks", 0) == 0
    model.kernel.lengthscales.assign(0.06)
    model._h.set_deferred_forward_pass(1)
    model.launch_elbo_host()
    model.close()                                      # teardown with a launch still waiting for its table


def test_matrix_core_chains_forced_where_they_do_not_apply_are_refused_cleanly(A):
    """Band algorithm 4 outside its domain (k != 4, M > 2048) is ASVGP_ERR_UNSUPPORTED with a message that says why - decided before a
    factor-table slot is taken, so the handle keeps working (17 more launches: every slot of the ring is reused)."""
    from asvgp_amd._lib import AsvgpError
    rng = np.random.default_rng(4)
    N = 20000
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    for basis in (A.B3Spline(0, 1, 300), A.B4Spline(0, 1, 3000)):
        model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.02), basis)
        model.likelihood.variance.assign(0.05)
        ref = model.elbo_and_grad().cpu().numpy()
        model._h.set_band_algorithm(4)
        with pytest.raises(AsvgpError) as ei:
            model.elbo_and_grad()
        assert "UNSUPPORTED" in str(ei.value) and "matrix-core" in str(ei.value)
        model._h.set_band_algorithm(0)
        for _ in range(17):
            got = model.elbo_and_grad().cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-9)


# ------------------------------------------------------------------------------------------------ Kuu forward pass on the GPU (double-double)
@pytest.mark.parametrize("order,M,kind,l", [(4, 2048, 1, 0.05), (4, 2047, 1, 0.05), (4, 1024, 0, 0.1), (3, 333, 2, 0.03), (5, 129, 2, 0.1),
                                            (6, 90, 1, 0.08), (1, 37, 0, 0.08), (2, 64, 1, 0.08), (4, 13, 0, 0.3), (4, 4096, 2, 0.005)])
def test_prior_forward_pass_on_the_gpu_double_double_vs_host_long_double(A, order, M, kind, l):
    """asvgp_set_prior_forward(h, 1) (VERDICT r2 missing #1): the forward (elimination) half of the planned Kuu chain - the factorisation
    half of gpr.py:56-59 - on the GPU in double-double arithmetic.  Its table against the host's long-double table of the same plan
    (asvgp_prior_forward_host, itself pinned against an all-nodes long-double elimination in the CPU tests): values to a few ulp of the
    record's largest entry times the elimination's growth, tangents (plain doubles on both sides, different summation orders) 1e-9."""
    from asvgp_amd import _lib
    lib = _lib.get_lib()
    bs = O.Basis(order, 0, 1, M)
    terms = O.kuu_terms(kind, 0.9, l)
    S = np.ascontiguousarray(np.stack([getattr(bs, nm) for nm, _, _ in terms]))
    c = np.zeros(16); dc = np.zeros(16)
    c[:len(terms)] = [t[1] for t in terms]
    dc[:len(terms)] = [t[2] for t in terms]
    n = lib.asvgp_prior_table_doubles(S.ctypes.data, len(terms), M, order)
    assert n > 0
    host = np.zeros(n)
    rec = np.zeros((M + order - 1) // order, dtype=np.int32)
    assert lib.asvgp_prior_forward_host(S.ctypes.data, len(terms), M, order, c.ctypes.data, dc.ctypes.data, host.ctypes.data, n, rec.ctypes.data) == 0
    h = _lib.Handle()
    assert h.prior_plan(S, len(terms), M, order)
    got = h.prior_forward_device(c, dc, n)
    B, R = order, int(host[3])
    W = 6 * B * B + B
    assert got[3] == R and got[2] == host[2] == 0
    hv, ht = host[8:8 + R * W].reshape(R, W), host[8 + R * W:8 + 2 * R * W].reshape(R, W)
    gv, gt = got[8:8 + R * W].reshape(R, W), got[8 + R * W:8 + 2 * R * W].reshape(R, W)
    root = R - 1
    worst_v = worst_t = 0.0
    for r_ in range(R):
        cols = slice(0, 3 * B * B + B) if r_ == root else slice(0, W)        # (the root record defines L, 1/diag, Sigma_00 and a zero U_b only)
        sv = max(1.0, float(np.max(np.abs(hv[r_, cols]))))
        st = max(1.0, float(np.max(np.abs(ht[r_, cols]))))
        worst_v = max(worst_v, float(np.max(np.abs(gv[r_, cols] - hv[r_, cols]))) / sv)
        worst_t = max(worst_t, float(np.max(np.abs(gt[r_, cols] - ht[r_, cols]))) / st)
    # the host rounds 64-bit-mantissa results to fp64, the device ~106-bit ones: both sides are within an ulp of their own exact
    # result, and the two exact results differ by the 2^-64 rounding errors of the host times the growth along the elimination
    slack = 1e-9 if kind == 2 else 1e-12
    assert worst_v <= slack, worst_v
    assert worst_t <= 1e3 * slack, worst_t
    assert abs(got[0] - host[0]) <= 1e-13 * abs(host[0]) and abs(got[1] - host[1]) <= 1e-9 * abs(host[1])
    print("dd forward pass k=%d M=%d kind=%d: values %.2g, tangents %.2g of the record scale" % (order, M, kind, worst_v, worst_t))
    h.close()


def test_prior_forward_on_the_gpu_reports_a_non_positive_pivot(A):
    """An indefinite 'Kuu' (negative variance): both forward passes report the same first failing column."""
    from asvgp_amd import _lib
    lib = _lib.get_lib()
    M, order = 64, 3
    bs = O.Basis(order, 0, 1, M)
    terms = O.kuu_terms(1, 0.9, 0.1)
    S = np.ascontiguousarray(np.stack([getattr(bs, nm) for nm, _, _ in terms]))
    c = np.zeros(16); dc = np.zeros(16)
    c[:len(terms)] = [-t[1] for t in terms]
    dc[:len(terms)] = [t[2] for t in terms]
    n = lib.asvgp_prior_table_doubles(S.ctypes.data, len(terms), M, order)
    host = np.zeros(n)
    rec = np.zeros((M + order - 1) // order, dtype=np.int32)
    assert lib.asvgp_prior_forward_host(S.ctypes.data, len(terms), M, order, c.ctypes.data, dc.ctypes.data, host.ctypes.data, n, rec.ctypes.data) == 0
    h = _lib.Handle()
    assert h.prior_plan(S, len(terms), M, order)
    got = h.prior_forward_device(c, dc, n)
    assert host[2] > 0 and got[2] == host[2]
    h.close()


def test_headline_bound_with_the_all_gpu_double_double_prior_chain(A):
    """The headline configuration (N = 10M, M = 2048, Matern-3/2; cond(Kuu) = 3.5e7) with NO host arithmetic in the chain:
    asvgp_amd.set_prior_forward(1).  Same gate as the default path: |GPU - long double oracle| <= 1e-9 |ELBO|, gradient 1e-6 - the gate the
    all-fp64 GPU chains (band algorithms 1, 2) miss by two orders of magnitude.  Also M = 4096 / Matern-5/2 (config 3's
    chain, thread-per-node kernels) and the posterior's operator, and a switch back to the host pass on the same handle."""
    import bench
    N, M = 10_000_000, 2048
    v, l, s = 1.0, 0.05, 0.01
    x, y = bench.synth(N)
    xd, yd = dev(x).reshape(-1, 1), dev(y).reshape(-1, 1)
    ob = O.Basis(4, 0, 1, M)
    try:
        A.set_prior_forward(1)
        model = A.GPR_1d((xd, yd), A.Matern32(variance=v, lengthscales=l), A.B4Spline(0, 1, M))
        model.likelihood.variance.assign(s)
        r = model.elbo_and_grad().cpu().numpy()
        stats = model._stats.cpu().numpy()
        E = 5 * M
        Ab, b, yy = stats[:E].reshape(5, M), stats[E:E + M].reshape(M, 1), float(stats[-1])
        ee, ge = O.elbo_grad_1d_extended(ob, O.MATERN32, Ab, b, yy, N, v, l, s)
        gate = 1e-9 * abs(ee)
        print("headline, double-double forward pass on the GPU: |GPU - long double| = %.3g (%.2g rel)" % (abs(r[0] - ee), abs(r[0] - ee) / abs(ee)))
        assert abs(r[0] - ee) <= gate, (r[0], ee, gate)
        np.testing.assert_allclose(r[1:4], ge, rtol=1e-6)
        # the host-read path (result mirror) and repeated steps at other theta reuse the device table ring
        for ll in (0.04, 0.06, 0.05):
            model.kernel.lengthscales.assign(ll)
            r2 = model.elbo_and_grad_host()
        assert abs(r2[0] - ee) <= gate
        # same handle, host pass again: the two agree far inside the gate
        model._h.set_prior_forward(0)
        r3 = model.elbo_and_grad().cpu().numpy()
        assert abs(r3[0] - r[0]) <= 0.1 * gate and np.allclose(r3[1:4], r[1:4], rtol=1e-7)
        model._h.set_prior_forward(1)
        mean, var = model.predict_f(dev(np.linspace(0.01, 0.99, 1000)).reshape(-1, 1))
        model._h.set_prior_forward(0)
        model._post = None                               # (drop the cached posterior operator: recompute with the host pass)
        mean0, var0 = model.predict_f(dev(np.linspace(0.01, 0.99, 1000)).reshape(-1, 1))
        assert np.allclose(np.asarray(mean), np.asarray(mean0), atol=1e-9) and np.allclose(np.asarray(var), np.asarray(var0), atol=1e-9)
    finally:
        A.set_prior_forward(0)


@pytest.mark.parametrize("order,M,kind", [(1, 40, 0), (2, 77, 1), (3, 100, 2), (4, 256, 2), (5, 200, 1), (6, 150, 1), (4, 4096, 2)])
def test_bound_with_gpu_forward_pass_equals_host_forward_pass_all_orders(A, order, M, kind):
    """Every bandwidth / kernel family through the planned chain with the forward pass on the GPU: bound and gradient equal the host-pass
    result to 1e-11 |ELBO| / 1e-8 (1e-6 at cond 1e9) (both carry > 64 mantissa bits through the elimination; what differs is fp64 rounding of the table)."""
    rng = np.random.default_rng(order * 100 + kind)
    N = 20000
    x = rng.uniform(0, 1, N)
    y = np.sin(12 * x) + 0.1 * rng.standard_normal(N)
    l = 0.005 if M == 4096 else 0.08
    res = []
    for mode in (0, 1):
        model = A.GPR_1d((dev(x).reshape(-1, 1), dev(y).reshape(-1, 1)), _kernel(A, kind, 0.9, l), _mk_basis(A, order, 0, 1, M))
        model.likelihood.variance.assign(0.05)
        model._h.set_prior_forward(mode)
        res.append(model.elbo_and_grad().cpu().numpy())
        model.close()
    assert abs(res[0][0] - res[1][0]) <= 1e-11 * abs(res[0][0]) + 1e-9, (res[0][0], res[1][0])
    # (M = 4096 / Matern-5/2: cond(Kuu) ~ 1e9 and the tangents are fp64 on both sides - 2.4e-8 measured, the stated gradient gate is 1e-6)
    np.testing.assert_allclose(res[1][1:4], res[0][1:4], rtol=1e-6 if M == 4096 else 1e-8, atol=1e-8)


def test_forward_pass_on_the_handles_worker_thread_gives_the_inline_numbers(A):
    """asvgp_set_deferred_forward_pass(h, 2): the host's long-double forward pass of the Kuu chain runs on a worker thread the handle owns,
    posted before the launch call.  Same table, same kernel: the results are bit-identical to the inline pass over a run of dependent steps
    (theta changes every step), switching modes back and forth on one handle works, and teardown joins the thread."""
    rng = np.random.default_rng(21)
    N, M = 50000, 2048
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(0.01)
    ls = [0.05 * (1.0 + 0.01 * i) for i in range(12)]
    ref = []
    for l in ls:
        model.kernel.lengthscales.assign(l)
        ref.append(model.elbo_and_grad_host())
    for mode in (2, 1, 2, 0):
        model._h.set_deferred_forward_pass(mode)
        got = []
        for l in ls:
            model.kernel.lengthscales.assign(l)
            got.append(model.elbo_and_grad_host())       # (mode 1: read_elbo_host publishes the pass itself)
        assert got == ref, mode
    model._h.set_deferred_forward_pass(2)
    model.kernel.lengthscales.assign(0.05)
    assert model.elbo_and_grad().tolist()[:4] == ref[0]       # the stream path waits for the same table
    model.close()                                             # joins the worker


@pytest.mark.parametrize("M", [512, 513, 516, 517, 1020, 1024, 1025, 1028, 1029, 1500, 2044, 2047, 2048])
def test_p_chain_on_two_workgroups_at_awkward_sizes(A, M):
    """The matrix-core P chain is split over two workgroups at the top of its elimination tree when nb = ceil(M / 4) >= 128 (bcr_mfma.hpp
    BmSplit): sizes around the powers of two - a right half of one node, of none (nb = 2^j + 1: the separator is the last node), an odd
    last block, a matrix that ends inside a block - against the oracle (statistics from the oracle, so only the chains differ), against the
    single-workgroup chain (ASVGP_NO_SPLIT: equal to rounding) and through the host-read path."""
    rng = np.random.default_rng(M)
    N = 30000
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    v, l, s = 1.1, 6.0 / M, 0.02
    ob = O.Basis(4, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y)
    ee, ge = O.elbo_grad_1d_extended(ob, O.MATERN32, Ab, b, yy, N, v, l, s)   # long double: the yardstick of the SURVEY 8d gates
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=v, lengthscales=l), A.B4Spline(0, 1, M))
    model.likelihood.variance.assign(s)
    r = model.elbo_and_grad().cpu().numpy()
    assert abs(r[0] - ee) <= 1e-9 * abs(ee), (M, r[0], ee)             # no allowance (VERDICT r3 weak #1a)
    np.testing.assert_allclose(r[1:4], ge, rtol=1e-6)
    rh = model.elbo_and_grad_host()
    assert rh == r[:4].tolist()
    os.environ["ASVGP_NO_SPLIT"] = "1"
    _reload_env()
    try:
        r1 = model.elbo_and_grad().cpu().numpy()
    finally:
        del os.environ["ASVGP_NO_SPLIT"]
        _reload_env()
    np.testing.assert_allclose(r1[:4], r[:4], rtol=1e-11)
    mean, var = model.predict_f(np.linspace(0.01, 0.99, 50).reshape(-1, 1))
    om, ov = O.predict_f_1d_banded(ob, O.MATERN32, Ab, b, v, l, s, np.linspace(0.01, 0.99, 50).reshape(-1, 1))
    np.testing.assert_allclose(np.asarray(mean), om, atol=1e-8)
    np.testing.assert_allclose(np.asarray(var), ov, atol=1e-8)
