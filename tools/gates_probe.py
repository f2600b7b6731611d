import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import asvgp_amd as A
from oracle import asvgp_oracle as O
KINDS = {0: "Matern12", 1: "Matern32", 2: "Matern52"}
def kern(kd, v, l): return getattr(A, KINDS[kd])(variance=v, lengthscales=l)
def basis(order, a, b, m): return getattr(A, "B%dSpline" % order)(a, b, m)
rng = np.random.default_rng(3)
N, M = 20000, 256
x = rng.uniform(1e-9, 1 - 1e-9, N); y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
for kd, order in ((1, 4), (2, 4), (0, 2), (2, 5), (1, 6)):
    bs = basis(order, 0, 1, M)
    model = A.GPR_1d((x.reshape(-1, 1), y), kern(kd, 1.0, 0.05), bs); model.likelihood.variance.assign(0.01)
    ob = O.Basis(order, 0, 1, M)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y)
    oe, og, _ = O.elbo_grad_1d(ob, kd, Ab, b, yy, N, 1.0, 0.05, 0.01)
    ee, ge = O.elbo_grad_1d_extended(ob, kd, Ab, b, yy, N, 1.0, 0.05, 0.01)
    r = model.elbo_and_grad().cpu().numpy()
    print("medium", kd, order, "rel vs ld %.2e  rel vs f64 %.2e  oracle f64 vs ld %.2e  grad rel vs ld %.2e" % (abs(r[0]-ee)/abs(ee), abs(r[0]-oe)/abs(oe), abs(oe-ee)/abs(ee), np.max(np.abs((r[1:4]-ge)/ge))))
# awkward sizes
for Mx in (512, 513, 1020, 1029, 1500, 2044, 2047, 2048):
    rng = np.random.default_rng(Mx)
    N = 30000
    x = rng.uniform(1e-9, 1 - 1e-9, N); y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
    v, l, s = 1.0, 0.05, 0.02
    model = A.GPR_1d((x.reshape(-1, 1), y), A.Matern32(variance=v, lengthscales=l), A.B4Spline(0, 1, Mx)); model.likelihood.variance.assign(s)
    ob = O.Basis(4, 0, 1, Mx)
    Ab, b, yy = O.sufficient_stats_direct(ob, x, y)
    ee, ge = O.elbo_grad_1d_extended(ob, 1, Ab, b, yy, N, v, l, s)
    r = model.elbo_and_grad().cpu().numpy()
    print("awkward", Mx, "rel vs ld %.2e grad %.2e" % (abs(r[0]-ee)/abs(ee), np.max(np.abs((r[1:4]-ge)/ge))))
