"""Kernel trace workload for the Kronecker path (config-4 shape): 5 x elbo_and_grad under rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
N, m = 1_000_000, 128
rng = np.random.default_rng(1234)
X = rng.uniform(1e-9, 1 - 1e-9, size=(N, 2)); y = (np.sin(12 * X[:, :1]) * np.cos(9 * X[:, 1:]) + 0.1 * rng.standard_normal((N, 1)))
model = A.GPR_kron((torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()), [A.Matern32(variance=1.0, lengthscales=0.2), A.Matern32(variance=1.0, lengthscales=0.2)],
                   [A.B3Spline(0, 1, m), A.B3Spline(0, 1, m)])
model.likelihood.variance.assign(0.01)
if os.environ.get("KTWIST") is not None:
    model.twisted = bool(int(os.environ["KTWIST"]))
for _ in range(int(os.environ.get("KREPS", 6))):
    model.elbo_and_grad()
torch.cuda.synchronize()
