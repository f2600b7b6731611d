"""Randomised parity sweep of GPR_kron and GPR_additive against the dense oracle (run on the GPU box).
usage: python tests/sweeps/fuzz_kron_additive.py [n_cases] [seed]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
from oracle import asvgp_oracle as O
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KIND = ["Matern12", "Matern32", "Matern52"]
orders_for = {0: [1, 2, 3, 4], 1: [2, 3, 4], 2: [3, 4]}
min_m = {1: 4, 2: 7, 3: 9, 4: 12}
fails = 0
for case in range(n_cases):
    kron = rng.random() < 0.6
    d = (2 if rng.random() < 0.75 else 3) if kron else int(rng.integers(1, 5))     # (d = 3: the band route for P, round 4)
    kds = [int(rng.integers(0, 3)) for _ in range(d)]
    order = int(rng.choice(sorted(set.intersection(*[set(orders_for[k]) for k in kds]))))
    if kron and d == 3 and order > 2:                             # (the dense long-double oracle of a d = 3 case: M_tot <= 1000)
        kds = [0, 0, int(rng.integers(0, 2))] if order > 2 else kds
        order = int(rng.choice(sorted(set.intersection(*[set(orders_for[k]) for k in kds]) & {1, 2})))
    ms = [int(rng.integers(min_m[order], 26 if not (kron and d == 3) else min_m[order] + 3)) for _ in range(d)]
    N = int(rng.integers(200, 20000))
    X = rng.uniform(0.001, 0.999, (N, d))
    if rng.random() < 0.3: X[:, 0] = np.sort(X[:, 0])
    y = (np.sin(5 * X[:, :1]) + (X[:, 1:2] ** 2 if d > 1 else 0) + 0.1 * rng.standard_normal((N, 1)))
    th = [(float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.15, 0.8))) for _ in range(d)]
    s = float(10 ** rng.uniform(-2.5, -0.5))
    try:
        bases = [getattr(A, "B%dSpline" % order)(0, 1, m) for m in ms]
        obases = [O.Basis(order, 0, 1, m) for m in ms]
        kerns = [getattr(A, KIND[k])(variance=v, lengthscales=l) for k, (v, l) in zip(kds, th)]
        Xs = rng.uniform(0.01, 0.99, (100, d))
        yy = float(np.sum(y * y))
        if kron:
            model = A.GPR_kron((X, y), kerns, bases); model.likelihood.variance.assign(s)
            model.twisted = [None, True, False][int(rng.integers(0, 3))]      # two-sided factorisation: by size / wherever it fits / never
            if d == 3 and rng.random() < 0.3:
                model.nd_banded = False                                       # (the dense route of d != 2)
            oe, parts = O.elbo_kron(obases, kds, th, s, X, y)
            e, g = model.elbo_and_grad()
            om, ov = O.predict_f_kron(obases, kds, th, s, X, y, Xs)
            fd = O.elbo_grad_kron(obases, kds, th, s, X, y)[1]      # the dense analytic gradient (differences of the bound lose ~1e-4 here)
            eg = np.max(np.abs(g - fd) / (np.abs(fd) + np.max(np.abs(fd))))
            vs = float(np.prod([v for v, _ in th]))
            xe = O.elbo_kron_extended(obases, kds, th, s, X, y)       # the dense bound in long double: how far is the fp64 oracle itself?
            kron_slack = 5 * abs(oe - xe)
            e_ref = xe
        else:
            model = A.GPR_additive((X, y), kerns, bases); model.likelihood.variance.assign(s)
            oe, _ = O.elbo_additive(obases, kds, th, s, X, y)
            e = model.elbo().item()
            om, ov = O.predict_f_additive(obases, kds, th, s, X, y, Xs)
            eg = 0.0
            vs = sum(v for v, _ in th)
            kron_slack, e_ref, oracle_limited = 0.0, oe, False
        mean, var = model.predict_f(Xs)
        tol_e = 1e-9 * abs(oe) + max(5e-9 * (0.5 * N * vs / s + 0.5 * yy / s), kron_slack)
        ee = abs(float(e) - e_ref)
        # gradient / posterior references are the fp64 DENSE oracle (central differences of it for the gradient): where that oracle's
        # own bound is off its long-double evaluation by more than 2e-9 of the large terms (Kuu = K1 (x) K2 with two Matern-5/2
        # factors: cond ~ 1e10+), it is no yardstick for them and only the bound is checked, against the long-double value
        oracle_limited = kron and abs(oe - xe) > 2e-9 * (0.5 * N * vs / s + 0.5 * yy / s)
        ep = max(np.max(np.abs(mean - om)), np.max(np.abs(var - ov)))
        ok = ee <= tol_e and ((kron and oracle_limited) or (eg <= 5e-5 and ep <= 1e-7))
    except Exception as ex:  # noqa
        ok, ee, eg, ep, tol_e = False, -1, -1, -1, 0
        print("EXC", repr(ex)[:300])
    if not ok:
        fails += 1
        print("FAIL case %d: %s d %d order %d m %s N %d kinds %s s %.3g | elbo err %.2e (tol %.2e) grad %.2e post %.2e" % (
            case, "kron" if kron else "additive", d, order, ms, N, kds, s, ee, tol_e, eg, ep), flush=True)
print("cases %d, failures %d" % (n_cases, fails))
