"""d = 3 GPR_kron: the band route for P (round 4) against the dense route, bound + gradient and posterior."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
rng = np.random.default_rng(5)
for order, m, N in ((2, 12, 200_000), (3, 11, 200_000), (1, 16, 200_000)):
    X = rng.uniform(0.001, 0.999, (N, 3)); y = (np.sin(5 * X[:, :1]) * np.cos(3 * X[:, 1:2]) + X[:, 2:] ** 2 + 0.1 * rng.standard_normal((N, 1)))
    B = getattr(A, "B%dSpline" % order)
    Kern = A.Matern12 if order == 1 else A.Matern32
    out = []
    for banded in (None, False):
        model = A.GPR_kron((X, y), [Kern(variance=1.0, lengthscales=0.4) for _ in range(3)], [B(0, 1, m) for _ in range(3)])
        model.likelihood.variance.assign(0.05)
        model.nd_banded = banded
        lay = model._nd_band_layout()
        e, g = model.elbo_and_grad(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); e, g = model.elbo_and_grad(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out.append((sorted(ts)[1] * 1e3, float(e), lay["bw"] if lay else None))
    print("order %d, %d^3 = %d basis functions, N = %d: bound + gradient band route %.1f ms (bandwidth %s) | dense route %.1f ms | rel diff of the bound %.1e" % (
        order, m, m ** 3, N, out[0][0], out[0][2], out[1][0], abs(out[0][1] - out[1][1]) / abs(out[1][1])), flush=True)
