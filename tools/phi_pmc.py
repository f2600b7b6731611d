"""Run only the Phi pass a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
N, M = 10_000_000, 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
if os.environ.get("PHI_SORTED"):
    o = np.argsort(x); x, y = x[o], y[o]
A.set_phi_algorithm(int(os.environ.get("PHI_ALGO", "0")))
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(), A.B4Spline(0, 1, M))
for _ in range(5):
    model.phi_pass()
torch.cuda.synchronize()
