"""BASELINE config 3 (N = 10M, M = 4096, Matern-5/2, B4Spline): timings per band algorithm and parity of the statistics, the
bound and the gradient against the oracle (fp64 reference order and long double).  usage: python tests/sweeps/c3_probe.py [N] [lengthscale]
(N = 1_250_000 is one rank's share of the 8-GPU run).  BASELINE names no theta for this configuration: at the north star's
lengthscale 0.05 (205 cells) Matern-5/2 gives cond(Kuu) ~ 1e17 - beyond fp64 for ANY elimination order, the oracle's own fp64 and
long-double values differ by 6e3 - so the parity figure is taken at lengthscale 0.005 (20 cells, cond ~ 1e9)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
from oracle import asvgp_oracle as O
N, M = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 4096
v, l, s = 1.0, float(sys.argv[2]) if len(sys.argv) > 2 else 0.005, 0.01
rng = np.random.default_rng(1)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern52(variance=v, lengthscales=l), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(s)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("phi pass  %.1f us" % t(model.phi_pass), flush=True)
ob = O.Basis(4, 0, 1, M)
Ab, b, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
ref = np.concatenate([Ab.reshape(-1), b.reshape(-1), [yy]])
print("statistics: max |diff| / max |ref| = %.2e" % (np.max(np.abs(model._stats.cpu().numpy() - ref)) / np.max(np.abs(ref))), flush=True)
oe, og, _ = O.elbo_grad_1d(ob, O.MATERN52, Ab, b, yy, N, v, l, s)
ee, ge = O.elbo_grad_1d_extended(ob, O.MATERN52, Ab, b, yy, N, v, l, s)
print("oracle fp64 %.6f  long double %.6f  (|diff| %.3g)" % (oe, ee, abs(oe - ee)))
names = ("d/dv", "d/dl", "d/ds2")
print("gradient, per component, relative to the long-double gradient %s:" % np.array2string(np.asarray(ge, dtype=np.float64), precision=6))
print("  fp64 oracle (reference elimination order): " + "  ".join("%s %.2e" % (nm, abs(a - b) / abs(b)) for nm, a, b in zip(names, og, ge)))
for algo in (0, 1, 2, 3):
    A.set_band_algorithm(algo)
    r = model.elbo_and_grad().cpu().numpy()
    print("band algorithm %d: elbo+grad %.1f us | ELBO - long double %+.3e | gradient rel %.2e" % (
        algo, t(lambda: model.elbo_and_grad(check_pd=False)), r[0] - ee, np.max(np.abs(r[1:4] - ge) / np.abs(ge))), flush=True)
    print("  per component: " + "  ".join("%s %.2e" % (nm, abs(a - b) / abs(b)) for nm, a, b in zip(names, r[1:4], ge)), flush=True)
A.set_band_algorithm(0)
