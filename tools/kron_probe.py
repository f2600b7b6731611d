"""Diagnostic: time the 2-D Kronecker path at the BASELINE config-4 shape (N=1M, 128 x 128, B3)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
N, m = int(os.environ.get("KN", 1_000_000)), int(os.environ.get("KM", 128))
rng = np.random.default_rng(1234)
X = rng.uniform(1e-9, 1 - 1e-9, size=(N, 2)); y = (np.sin(12 * X[:, :1]) * np.cos(9 * X[:, 1:]) + 0.1 * rng.standard_normal((N, 1)))
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
bases = [A.B3Spline(0, 1, m), A.B3Spline(0, 1, m)]
kern = [A.Matern32(variance=1.0, lengthscales=0.2), A.Matern32(variance=1.0, lengthscales=0.2)]
torch.cuda.synchronize(); t0 = time.perf_counter()
model = A.GPR_kron((Xd, yd), kern, bases); model.likelihood.variance.assign(0.01)
if os.environ.get("KTWIST") is not None:
    model.twisted = bool(int(os.environ["KTWIST"]))            # KTWIST=0: the one-sided band Cholesky
print("factorisation:", "two-sided %s" % model._twist_layout() if model._twist_layout() else "one-sided")
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(2): model.phi_pass()
torch.cuda.synchronize(); t2 = time.perf_counter()
model.phi_pass(); torch.cuda.synchronize(); t3 = time.perf_counter()
e = model.elbo().item(); torch.cuda.synchronize(); t4 = time.perf_counter()
e = model.elbo().item(); torch.cuda.synchronize(); t5 = time.perf_counter()
Xs = torch.from_numpy(rng.uniform(0.01, 0.99, size=(10000, 2))).cuda()
mean, var = model.predict_f_device(Xs); torch.cuda.synchronize(); t6 = time.perf_counter()
print("N=%d M_tot=%d bw=%d | construct %.1f ms | phi pass %.2f ms (%.0f Mpoints/s) | elbo %.1f ms | predict 10k (incl. factor) %.1f ms | elbo=%.6f" % (
    N, m * m, model.true_bandwidth, (t1 - t0) * 1e3, (t3 - t2) * 1e3, N / (t3 - t2) / 1e6, (t5 - t4) * 1e3, (t6 - t5) * 1e3, e))
mean, var = model.predict_f_device(Xs); torch.cuda.synchronize(); t7 = time.perf_counter()
Xl = torch.from_numpy(rng.uniform(0.01, 0.99, size=(1_000_000, 2))).cuda()
torch.cuda.synchronize(); t8 = time.perf_counter()
mean, var = model.predict_f_device(Xl); torch.cuda.synchronize(); t9 = time.perf_counter()
eg = model.elbo_and_grad(); torch.cuda.synchronize(); t10 = time.perf_counter()
eg = model.elbo_and_grad(); torch.cuda.synchronize(); t11 = time.perf_counter()
reps = []
for _ in range(5):
    ta = time.perf_counter(); model.elbo().item(); tb = time.perf_counter(); model.elbo_and_grad(); tc = time.perf_counter()
    reps.append((tb - ta, tc - tb))
print("5 repeats: elbo %s ms | elbo+grad %s ms" % ([round(a * 1e3, 2) for a, _ in reps], [round(b * 1e3, 2) for _, b in reps]))
f = model._factor(want_alpha=True); torch.cuda.synchronize(); t12 = time.perf_counter()
model._selinv(f); torch.cuda.synchronize(); t13 = time.perf_counter()
print("predict 10k (cached factor) %.2f ms | predict 1M %.1f ms | elbo+grad %.1f ms | factor %.1f ms | selected inverse %.1f ms | grad %s" % (
    (t7 - t6) * 1e3, (t9 - t8) * 1e3, (t11 - t10) * 1e3, (t12 - t11) * 1e3, (t13 - t12) * 1e3, np.round(eg[1], 3)))
