"""CPU (-m "not gpu"): the C-ABI library builds, loads and exports every symbol include/asvgp_hip.h declares
(no compute calls), host-side argument checking, host logic of the package (mesh, static bands, parameter
transforms), and the N>1 sharding + collective path on gloo with world_size 2."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import asvgp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def lib():
    from asvgp_amd import build, _lib
    build.build(verbose=False)
    return _lib.get_lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "asvgp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(asvgp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 24
    from asvgp_amd import _lib
    for n in names:
        assert hasattr(lib, n), "libasvgp_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "no ctypes prototype for %s" % n
    assert lib.asvgp_version() >= 100
    assert lib.asvgp_status_name(0) == b"ASVGP_OK"
    assert lib.asvgp_status_name(-3) == b"ASVGP_ERR_LDS_CAPACITY"


def test_host_side_argument_checks_fail_loudly(lib):
    from asvgp_amd import _lib
    assert lib.asvgp_cholesky_band(None, None, 10, 4, None, None) == -1
    assert b"cholesky_band" in lib.asvgp_last_error_string()
    assert lib.asvgp_cholesky_band(ctypes.c_void_p(8), ctypes.c_void_p(8), 10, 9, None, None) == -2   # bandwidth > 8
    assert lib.asvgp_phi_accumulate_1d(None, ctypes.c_void_p(8), ctypes.c_void_p(8), 10, 1, ctypes.c_void_p(8), 7, 0.1, 4, 10,
                                       ctypes.c_void_p(8), None, 0, None) == -4                         # no workspace
    assert lib.asvgp_phi_accumulate_1d(None, ctypes.c_void_p(8), ctypes.c_void_p(8), 10, 1, ctypes.c_void_p(8), 5, 0.1, 4, 10,
                                       ctypes.c_void_p(8), ctypes.c_void_p(8), 1 << 30, None) == -1     # n_mesh != M-k+1
    assert lib.asvgp_phi_accumulate_1d(None, ctypes.c_void_p(8), ctypes.c_void_p(8), 10, 1, ctypes.c_void_p(8), 7, 0.1, 7, 13,
                                       ctypes.c_void_p(8), ctypes.c_void_p(8), 1 << 30, None) == -2     # order 7
    assert lib.asvgp_set_band_algorithm(None, 7) == -1 and lib.asvgp_set_phi_algorithm(None, 2) == -1
    assert lib.asvgp_destroy(None) == 0
    assert lib.asvgp_phi_workspace_bytes(2048, 4, 1) == 8 * 256 * (7 * 2048 + 1)
    assert lib.asvgp_elbo_workspace_bytes(2048, 4, 1) >= 8 * (9 * 5 * 2048 + 2 * 2048)
    with pytest.raises(_lib.AsvgpError):
        _lib.check(-2, "x")
    with pytest.raises(_lib.AsvgpError):
        _lib.require_cuda(torch.zeros(3))


def test_matern_coefficient_table(lib):
    for kind in (0, 1, 2):
        for v, l in ((1.0, 1.0), (0.8, 1.03), (2.5, 0.05)):
            c = (ctypes.c_double * 9)()
            dc = (ctypes.c_double * 9)()
            n = ctypes.c_int(0)
            assert lib.asvgp_matern_coeffs(kind, v, l, c, dc, ctypes.byref(n)) == 0
            terms = O.kuu_terms(kind, v, l)
            assert n.value == len(terms)
            for t, (nm, cc, dcc) in enumerate(terms):
                assert c[t] == cc, (kind, nm)                      # same rounding sequence as inducing_features.py
                assert abs(dc[t] - dcc) <= 1e-15 * abs(dcc) if dcc else dc[t] == 0
    n = ctypes.c_int(0)
    c = (ctypes.c_double * 9)()
    assert lib.asvgp_matern_coeffs(7, 1.0, 1.0, c, c, ctypes.byref(n)) == -2
    assert lib.asvgp_matern_coeffs(1, -1.0, 1.0, c, c, ctypes.byref(n)) == -1


def test_package_host_logic_mesh_and_static_bands(golden_dir):
    import asvgp_amd as A
    F = np.load(os.path.join(golden_dir, "basis_fixtures.npz"))
    cpu = torch.device("cpu")
    for tag in F["tags"]:
        order, a, b, m, isf = F[tag + "/spec"]
        order, m = int(order), int(m)
        a, b = (float(a), float(b)) if isf else (int(a), int(b))
        bs = getattr(A, "B%dSpline" % order)(a, b, m, device=cpu)
        assert bs.order == order and bs.m == m
        assert np.array_equal(bs.mesh_np, F[tag + "/mesh"]) and bs.delta_np == float(F[tag + "/delta"])
        for nm in ("A", "B", "C", "D", "BC", "BC_grad", "BC_ggrad", "BC_ggrad_none", "BC_none_ggrad"):
            if tag + "/" + nm in F:
                ref = F[tag + "/" + nm]
                assert np.max(np.abs(ref - bs.static_np[nm])) <= 4e-15 * max(np.max(np.abs(ref)), 1e-300), (tag, nm)
            else:
                assert not hasattr(bs, nm) or m > 100
    with pytest.raises(NameError):
        A.B4Spline(0, 1, 11, device=cpu)
    # compute entry points refuse to run without the GPU
    from asvgp_amd import _lib
    bs = A.B3Spline(0, 1, 20, device=cpu)
    with pytest.raises(_lib.AsvgpError):
        bs.neighbour_index(np.array([0.5]))


def test_parameter_transforms():
    from asvgp_amd import kernels as K
    p = K.Parameter(1.0)
    assert abs(float(p) - 1.0) < 1e-15 and abs(p.unconstrained - O.softplus_inv(1.0)) < 1e-15
    g = K.Gaussian()
    assert abs(float(g.variance) - 1.0) < 1e-15 and abs(g.variance.unconstrained - O.softplus_inv(1.0 - 1e-6)) < 1e-15
    for u in (-30.0, -1.0, 0.0, 2.0, 40.0):
        p.unconstrained = u
        assert abs(float(p) - O.softplus(u)) <= 1e-15 * max(1.0, abs(u))
        assert abs(p.dtheta_du() - O.sigmoid(u)) < 1e-15
    k = K.Matern52(variance=2.0, lengthscales=0.3)
    assert k.kind == 2 and abs(2 * k.variance - 4.0) < 1e-14 and abs(k.lengthscales ** 2 - 0.09) < 1e-15


def test_shard_bounds_cover_and_align():
    from asvgp_amd.dist import shard_bounds
    for N in (0, 1, 7, 1000, 10_000_000, 10_000_001):
        for W in (1, 2, 3, 4, 8):
            spans = [shard_bounds(N, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            for (l0, h0), (l1, h1) in zip(spans, spans[1:]):
                assert h0 == l1 and l0 <= h0
            assert all(lo % 2 == 0 for lo, hi in spans if hi > lo)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asvgp_amd.dist import allreduce_stats, shard_bounds
    rng = np.random.default_rng(99)
    N, M = 5001, 40
    x = rng.uniform(0.001, 0.999, N)
    y = rng.normal(size=(N, 1))
    bs = O.Basis(4, 0, 1, M)
    lo, hi = shard_bounds(N, world, rank)
    band, rhs, yy = O.sufficient_stats_direct(bs, x[lo:hi], y[lo:hi])     # the oracle stands in for the HIP Phi pass
    packed = torch.from_numpy(np.concatenate([band.reshape(-1), rhs.reshape(-1), [yy]]))
    n_glob = allreduce_stats(packed, hi - lo, None)
    q.put((rank, n_glob, packed.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_band_allreduce_matches_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(99)
    N, M = 5001, 40
    x = rng.uniform(0.001, 0.999, N)
    y = rng.normal(size=(N, 1))
    band, rhs, yy = O.sufficient_stats_direct(O.Basis(4, 0, 1, M), x, y)
    full = np.concatenate([band.reshape(-1), rhs.reshape(-1), [yy]])
    for rank, n_glob, packed in res:
        assert n_glob == N
        np.testing.assert_allclose(packed, full, rtol=0, atol=1e-12 * np.max(np.abs(full)))   # sum order differs: tolerance
    # the ELBO from the all-reduced statistics equals the single-rank one
    e1, _ = O.elbo_1d(O.make_Kuu(O.Basis(4, 0, 1, M), 1, 1.0, 0.1), band, rhs, yy, N, 1.0, 0.05)
    p = res[0][2]
    e2, _ = O.elbo_1d(O.make_Kuu(O.Basis(4, 0, 1, M), 1, 1.0, 0.1), p[:5 * M].reshape(5, M), p[5 * M:6 * M].reshape(M, 1),
                      p[-1], N, 1.0, 0.05)
    assert abs(e1 - e2) <= 1e-10 * abs(e1)


def test_bench_self_launch_dry_two_ranks():
    """bench.py --gpus 2 without a torchrun environment spawns its own workers before any GPU call;
    ASVGP_BENCH_DRY=1 stops after the rendezvous + one all-reduce so that the launcher can be rehearsed without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ASVGP_BENCH_DRY"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry": True, "n_gpus": 2, "ranks_seen": 2}


# ------------------------------------------------------------------------------------------------ planned prior chain (host part)
def _bcr_forward_long_double(K, B):
    """Forward pass of block cyclic reduction (odd-even elimination of the B x B block-tridiagonal view of the lower band K),
    every node, in numpy long double: what asvgp_prior_forward_host must reproduce on its de-duplicated nodes."""
    ld = np.longdouble
    k, M = K.shape[0] - 1, K.shape[1]
    nb = (M + B - 1) // B
    D = [np.zeros((B, B), ld) for _ in range(nb)]
    E = [np.zeros((B, B), ld) for _ in range(nb)]
    for j in range(M):
        for d in range(k + 1):
            i = j + d
            if i < M:
                if i // B == j // B:
                    D[i // B][i % B, j % B] = D[i // B][j % B, i % B] = K[d, j]
                else:
                    E[j // B][i % B, j % B] = K[d, j]
    for r in range(M, nb * B):
        D[r // B][r % B, r % B] = 1

    def chol(A):
        L = np.zeros_like(A)
        for j in range(B):
            L[j, j] = np.sqrt(A[j, j] - np.sum(L[j, :j] ** 2))
            for i in range(j + 1, B):
                L[i, j] = (A[i, j] - np.sum(L[i, :j] * L[j, :j])) / L[j, j]
        return L

    def solve(L, X):
        X = X.copy()
        for i in range(B):
            X[i] = (X[i] - L[i, :i] @ X[:i]) / L[i, i]
        return X
    fac, logdet, h = {}, ld(0), 1
    while h < nb:
        for i in range(h, nb, 2 * h):
            a, b = i - h, i + h
            L = chol(D[i])
            logdet += 2 * np.sum(np.log(np.diag(L)))
            Ua = solve(L, E[a])
            Ub = solve(L, E[i].T) if b < nb else np.zeros((B, B), ld)
            fac[i] = (L, Ua, Ub)
            D[a] = D[a] - Ua.T @ Ua
            if b < nb:
                D[b] = D[b] - Ub.T @ Ub
                E[a] = -(Ub.T @ Ua)
        h *= 2
    L0 = chol(D[0])
    logdet += 2 * np.sum(np.log(np.diag(L0)))
    return fac, L0, logdet


@pytest.mark.parametrize("order,M,kind,l", [(4, 2048, 1, 0.05), (4, 2047, 1, 0.05), (4, 1024, 0, 0.1), (4, 512, 2, 0.03), (3, 333, 2, 0.03), (4, 64, 1, 0.3)])
def test_closed_form_kuu_equals_the_assembled_band_bit_for_bit(lib, order, M, kind, l):
    """The matrix-core P chain does not wait for the assembled Kuu: it receives diagonal values on the Toeplitz interior and a table of
    the boundary columns (prior_plan_interior_kuu).  Every entry must be the very double inducing_features.py:12-44 produces."""
    import ctypes
    bs = O.Basis(order, 0, 1, M)
    terms = O.kuu_terms(kind, 0.9, l)
    S = np.ascontiguousarray(np.stack([getattr(bs, nm) for nm, _, _ in terms]))
    c = np.array([t[1] for t in terms])
    kd, bnd = np.zeros(8), np.zeros(256)
    lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.asvgp_prior_interior_kuu_host(S.ctypes.data, len(terms), M, order, c.ctypes.data, kd.ctypes.data, ctypes.byref(lo),
                                             ctypes.byref(hi), bnd.ctypes.data) == 0
    lo, hi = lo.value, hi.value
    assert 0 < lo <= 16 and M - 16 <= hi <= M - order and hi - lo > M // 2
    K = O.make_Kuu(bs, kind, 0.9, l)                       # (order + 1, M) lower band
    for d in range(order + 1):
        assert np.array_equal(K[d, lo:hi], np.full(hi - lo, kd[d])), d
        assert np.array_equal(K[d, :lo], bnd[d * 16:d * 16 + lo]), d
        n_right = M - d - hi                                # entries of diagonal d that exist to the right of the interior
        assert np.array_equal(K[d, hi:M - d], bnd[(8 + d) * 16:(8 + d) * 16 + n_right]), d


def test_host_forward_pass_runs_in_x87_extended_precision(lib):
    """VERDICT r2 #1d: prior_plan.cpp static_asserts LDBL_MANT_DIG == 64; the library reports the width it was built with, and numpy's
    longdouble on this host - the oracle's extended evaluation - is the same format."""
    assert lib.asvgp_host_mantissa_bits() == 64
    assert np.finfo(np.longdouble).nmant == 63          # (numpy counts the stored fraction bits: 64-bit mantissa with its explicit leading bit)


@pytest.mark.parametrize("order,M,kind,l", [(4, 2048, 1, 0.05), (4, 2047, 1, 0.05), (4, 1024, 0, 0.1), (3, 333, 2, 0.03), (5, 129, 2, 0.1),
                                            (6, 90, 1, 0.08), (1, 37, 0, 0.08), (2, 64, 1, 0.08), (4, 13, 0, 0.3)])
def test_prior_plan_host_forward_vs_long_double_bcr(lib, order, M, kind, l):
    """asvgp_prior_forward_host eliminates ONE representative per class of bit-identical nodes; every node's factors
    (through the node -> record map), the log-determinant and the l-tangents must equal the all-nodes long-double forward pass."""
    bs = O.Basis(order, 0, 1, M)
    terms = O.kuu_terms(kind, 0.9, l)
    S = np.ascontiguousarray(np.stack([getattr(bs, nm) for nm, _, _ in terms]))
    c = np.array([t[1] for t in terms])
    dc = np.array([t[2] for t in terms])
    n = lib.asvgp_prior_table_doubles(S.ctypes.data, len(terms), M, order)
    assert n > 0
    tab = np.zeros(n)
    nb = (M + order - 1) // order
    rec = np.zeros(nb, dtype=np.int32)
    assert lib.asvgp_prior_forward_host(S.ctypes.data, len(terms), M, order, c.ctypes.data, dc.ctypes.data, tab.ctypes.data, n,
                                        rec.ctypes.data) == 0
    B, R = order, int(tab[3])
    W = 6 * B * B + B          # L, 1/diag, U_a, U_b | G_a^T, G_b^T, D^-1 (the matrix-core backward pass's operands)
    assert n == 8 + 2 * R * W and tab[2] == 0 and R <= 3 * 12 and rec.min() >= 0 and rec.max() == R - 1 == rec[0]
    val, tan = tab[8:8 + R * W].reshape(R, W), tab[8 + R * W:].reshape(R, W)
    K, dK = O.make_Kuu(bs, kind, 0.9, l, want_dl=True)
    fac, L0, logdet = _bcr_forward_long_double(K, B)
    cond_slack = 1e-9 if kind == 2 else 1e-11      # two 64-bit-mantissa evaluations in different operation orders, times cond(Kuu)
    for i, (L, Ua, Ub) in fac.items():
        for off, X in ((0, L), (B * B + B, Ua), (2 * B * B + B, Ub)):
            got = val[rec[i], off:off + B * B].reshape(B, B)
            assert np.max(np.abs(got - X.astype(np.float64))) <= cond_slack * max(1.0, float(np.max(np.abs(X)))), (i, off)
        np.testing.assert_allclose(val[rec[i], B * B:B * B + B] * np.diag(L).astype(np.float64), 1.0, rtol=cond_slack)
        Li = np.linalg.inv(L.astype(np.float64)).astype(np.longdouble)      # (4 x 4 .. 6 x 6 triangular: well conditioned)
        Li = Li + Li @ (np.eye(B, dtype=np.longdouble) - L @ Li)             # one refinement step in long double
        for off, X in ((3 * B * B + B, Ua.T @ Li), (4 * B * B + B, Ub.T @ Li), (5 * B * B + B, Li.T @ Li)):
            got = val[rec[i], off:off + B * B].reshape(B, B)
            assert np.max(np.abs(got - X.astype(np.float64))) <= 10 * cond_slack * max(1.0, float(np.max(np.abs(X)))), (i, off)
    np.testing.assert_allclose(val[rec[0], :B * B].reshape(B, B), L0.astype(np.float64), rtol=0, atol=cond_slack * float(np.max(L0)))
    assert abs(tab[0] - float(logdet)) <= max(1e-12, cond_slack * 1e-2) * abs(float(logdet))
    # tangents against central differences of the long-double pass (well-conditioned cases only resolve this)
    h = 1e-6 * l
    fp, _, ldp = _bcr_forward_long_double(O.make_Kuu(bs, kind, 0.9, l + h), B)
    fm, _, ldm = _bcr_forward_long_double(O.make_Kuu(bs, kind, 0.9, l - h), B)
    assert abs(tab[1] - float((ldp - ldm) / (2 * h))) <= 2e-5 * abs(tab[1])
    if M <= 129:
        for i in fac:
            for off, q in ((0, 0), (B * B + B, 1), (2 * B * B + B, 2)):
                fd = ((fp[i][q] - fm[i][q]) / (2 * h)).astype(np.float64)
                got = tan[rec[i], off:off + B * B].reshape(B, B)
                assert np.max(np.abs(got - fd)) <= 1e-5 * max(1e-12, float(np.max(np.abs(fd)))), (i, off)

            def gmats(f):
                L_, Ua_, Ub_ = f
                Li_ = np.linalg.inv(L_.astype(np.float64))
                return Ua_.astype(np.float64).T @ Li_, Ub_.astype(np.float64).T @ Li_, Li_.T @ Li_
            for off, Xp, Xm in zip((3 * B * B + B, 4 * B * B + B, 5 * B * B + B), gmats(fp[i]), gmats(fm[i])):
                fd = (Xp - Xm) / (2 * h)
                got = tan[rec[i], off:off + B * B].reshape(B, B)
                assert np.max(np.abs(got - fd)) <= 1e-4 * max(1e-12, float(np.max(np.abs(fd)))), (i, off)
    # a band without Toeplitz structure has no plan
    rng = np.random.default_rng(0)
    Sr = np.ascontiguousarray(S + rng.uniform(0, 1e-3, S.shape) * (np.abs(S) > 0))
    if nb > 64:
        assert lib.asvgp_prior_table_doubles(Sr.ctypes.data, len(terms), M, order) == 0


@pytest.mark.parametrize("order,M,kind", [(4, 2048, 1), (4, 2047, 1), (3, 333, 2), (6, 90, 1), (1, 37, 0), (4, 13, 0), (4, 4096, 2)])
def test_prior_plan_device_image_is_consistent_with_the_host_plan(lib, order, M, kind):
    """The GPU forward pass (csrc/prior_dd.hpp, asvgp_set_prior_forward(h, 1)) walks a flat image of the plan: class maps per level as ints,
    the static-band entries of the level-0 representative blocks as doubles.  Host-only checks: header fields, level offsets inside the
    image, class indices inside the previous level's class counts, node counts summing to the number of nodes a level eliminates, record
    numbering equal to the host table's, and the level-0 entries reproducing Kuu's blocks exactly (same rounding sequence as make_Kuu)."""
    bs = O.Basis(order, 0, 1, M)
    terms = O.kuu_terms(kind, 0.9, 0.07)
    S = np.ascontiguousarray(np.stack([getattr(bs, nm) for nm, _, _ in terms]))
    ni, nd = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.asvgp_prior_plan_image_host(S.ctypes.data, len(terms), M, order, None, ctypes.byref(ni), None, ctypes.byref(nd)) == 0
    ints = np.zeros(ni.value, dtype=np.int32)
    dbl = np.zeros(max(nd.value, 1))
    assert lib.asvgp_prior_plan_image_host(S.ctypes.data, len(terms), M, order, ints.ctypes.data, ctypes.byref(ni), dbl.ctypes.data, ctypes.byref(nd)) == 0
    B, nt, levels, n_rec, nd0, ne0, off_lv, total, off_dc, off_ec = [int(v) for v in ints[:10]]
    nb = (M + B - 1) // B
    assert B == order and nt == len(terms) and total == ni.value and off_lv == 16 and (1 << levels) >= nb > (1 << levels) // 2
    n = lib.asvgp_prior_table_doubles(S.ctypes.data, len(terms), M, order)
    assert n == 8 + 2 * n_rec * (6 * B * B + B)
    prev_d, prev_e, rec, left = nd0, ne0, 0, nb
    for l in range(levels):
        nq, ndn, o_node, o_rep, o_dn, rec0 = [int(v) for v in ints[off_lv + 8 * l: off_lv + 8 * l + 6]]
        assert rec0 == rec and 0 < nq <= 24 and 0 < ndn <= 24 and 16 + 8 * levels <= o_node < o_rep <= o_dn < off_dc
        node = ints[o_node:o_node + 4 * nq].reshape(nq, 4)
        assert node[:, 0].min() >= 0 and node[:, 0].max() < prev_d and node[:, 1].min() >= 0 and node[:, 1].max() < prev_e
        assert node[:, 2].min() >= -1 and node[:, 2].max() < prev_e
        assert int(node[:, 3].sum()) == left // 2                    # every second node of the level is eliminated
        reps = ints[o_rep:o_rep + nq]
        assert reps.min() >= 0 and reps.max() < nb
        dn = ints[o_dn:o_dn + 3 * ndn].reshape(ndn, 3)
        assert dn[:, 0].min() >= 0 and dn[:, 0].max() < prev_d and dn[:, 1:].min() >= -1 and dn[:, 1:].max() < nq
        rec += nq
        left -= left // 2
        prev_d, prev_e = ndn, nq                                     # next level: D classes; couplings = the fills of the eliminated nodes
    assert rec == n_rec - 1 and left == 1
    # level-0 entries: sum_t c_t * entry_t equals the Kuu block entries of a representative (class 0 = block 0: the top-left corner)
    c = np.array([t[1] for t in terms])
    K = O.make_Kuu(bs, kind, 0.9, 0.07)
    codes = ints[off_dc:off_dc + B * B].reshape(B, B)
    ent = dbl[:B * B * nt].reshape(B, B, nt)
    for r in range(B):
        for cc in range(r + 1):
            if r < M:
                acc = c[0] * ent[r, cc, 0]
                for t in range(1, nt):
                    acc = acc + c[t] * ent[r, cc, t]
                assert codes[r, cc] == 0 and acc == K[r - cc, cc], (r, cc)


def test_two_sided_factorisation_layout_invariants():
    """asvgp_amd.kronecker.twisted_layout (host logic of GPR_kron's two-sided band Cholesky): for every size both systems have nb super-blocks,
    the separator is their last block, paddings are non-negative, the interiors do not overlap and together with the separator cover
    [0, M); small matrices keep the one-sided factorisation unless forced."""
    from asvgp_amd.kronecker import twisted_layout
    seen = 0
    for M in list(range(40, 3000, 37)) + [16384, 10000, 128 * 128, 100 * 100]:
        for bw in (3, 10, 26, 36, 60, 147, 387, 404):
            for force in (None, True, False):
                lay = twisted_layout(M, bw, force)
                Bb = ((max(bw, 1) + 31) // 32) * 32
                if force is False:
                    assert lay is None
                    continue
                if lay is None:
                    nb = -(-(M + Bb) // (2 * Bb))
                    assert nb < 3 or (force is None and -(-M // Bb) < 6)
                    continue
                seen += 1
                nb, top_end, h, padt, padb = lay["nb"], lay["top_end"], lay["h"], lay["padt"], lay["padb"]
                assert lay["Bb"] == Bb and Bb >= bw and Bb % 32 == 0 and nb >= 3
                assert padt >= 0 and padb >= 0 and h == top_end - Bb and 0 <= h and top_end <= M
                assert padt + top_end == nb * Bb                      # top system: padding + columns [0, top_end)
                assert padb + (M - h) == nb * Bb                      # bottom system: padding + columns [h, M)
                assert abs(padt - padb) <= 1                          # balanced chains
                assert (nb - 1) * Bb >= padt + h and (nb - 1) * Bb >= padb + (M - top_end)   # interiors fit in front of the separator block
    assert seen > 100
    lay = twisted_layout(128 * 128, 3 * 128 + 3)
    assert lay == dict(Bb=416, nb=21, top_end=8400, h=7984, padt=336, padb=336, bw=387)

