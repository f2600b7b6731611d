"""Diagnostic: capture one bench step (prior chain on a side stream + Phi pass + data chain) in a HIP graph and replay."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
from asvgp_amd import _lib
lib = _lib.get_lib(); lib.asvgp_set_phi_workgroups(248); lib.asvgp_elbo_chain_sync(1)
N, M = 10_000_000, 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(0.01)
side = torch.cuda.Stream(priority=-1)
def step():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        model.launch_prior_chain()
    model.phi_pass()
    model.launch_data_chain()      # ordered against the prior chain by the library's own events (asvgp_elbo_chain_sync)
for _ in range(3): step()
torch.cuda.synchronize()
ref = model._out.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 100
for _ in range(K): g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("graph replay: %.1f us/step  (%.0f Mpoints/s)   result matches eager: %s" % (dt * 1e6, N / dt / 1e6, torch.allclose(model._out, ref, rtol=1e-9)))
t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("eager:        %.1f us/step  (%.0f Mpoints/s)" % (dt * 1e6, N / dt / 1e6))
