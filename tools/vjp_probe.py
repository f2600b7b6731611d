"""Timing of the operator VJPs (single-thread adjoint sweeps) at M = 2048, k = 4 against their forward operators."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
from asvgp_amd import banded as B
M = 2048
feat = A.SplineFeatures1D(A.Matern32(lengthscales=0.05), A.B4Spline(0, 1, M))
K = feat.make_Kuu(A.Matern32(lengthscales=0.05))
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
L = B.cholesky_band(K); S = B.inverse_from_cholesky_band(L)
print("forward: cholesky_band %.2f ms, inverse_from_cholesky_band %.2f ms" % (t(lambda: B.cholesky_band(K)), t(lambda: B.inverse_from_cholesky_band(L))))
def chol_bwd():
    Kt = K.clone().requires_grad_(True); B.cholesky_band(Kt).sum().backward()
def inv_bwd():
    Lt = L.clone().requires_grad_(True); B.inverse_from_cholesky_band(Lt).sum().backward()
print("forward + backward: cholesky_band %.2f ms, inverse_from_cholesky_band %.2f ms" % (t(chol_bwd), t(inv_bwd)))
