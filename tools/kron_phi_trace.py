"""Kernel-trace workload: the Kronecker Phi pass alone (config-4 shape), 10 passes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
N, m = 1_000_000, 128
rng = np.random.default_rng(1234)
X = rng.uniform(1e-9, 1 - 1e-9, size=(N, 2)); y = (np.sin(12 * X[:, :1]) * np.cos(9 * X[:, 1:]) + 0.1 * rng.standard_normal((N, 1)))
model = A.GPR_kron((torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()), [A.Matern32(), A.Matern32()], [A.B3Spline(0, 1, m), A.B3Spline(0, 1, m)])
for _ in range(10):
    model.phi_pass()
torch.cuda.synchronize()
