"""Kronecker Phi pass alone (config-4 shape): HIP-event time per pass."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
N, m = int(os.environ.get("KN", 1_000_000)), int(os.environ.get("KM", 128))
order = int(os.environ.get("KORDER", 3))
rng = np.random.default_rng(1234)
X = rng.uniform(1e-9, 1 - 1e-9, size=(N, 2)); y = (np.sin(12 * X[:, :1]) * np.cos(9 * X[:, 1:]) + 0.1 * rng.standard_normal((N, 1)))
B = getattr(A, "B%dSpline" % order)
model = A.GPR_kron((torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()), [A.Matern32(), A.Matern32()], [B(0, 1, m), B(0, 1, m)])
for _ in range(3): model.phi_pass()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
ev[0].record()
for i in range(20):
    model.phi_pass(); ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(20))
print("grid %s order %d N %d %dx%d: phi pass median %.1f us  min %.1f us (events around the whole pass: memset + kernels)" % (os.environ.get("ASVGP_KRON_PHI_GRID", "default"), order, N, m, m, ts[10], ts[0]))
