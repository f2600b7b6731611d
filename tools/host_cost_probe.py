"""Host-side cost of enqueueing one step (no waiting for the GPU): Phi pass call, ELBO call, and their parts."""
import sys, time, ctypes, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
from asvgp_amd import _lib
N, M = 10_000_000, 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
m = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
m.likelihood.variance.assign(0.01)
for _ in range(5): m.phi_pass(); m._launch_elbo()
torch.cuda.synchronize()
def host(f, n=20):
    tot = 0.0
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); tot += time.perf_counter() - t0
    return tot / n * 1e6
print("phi_pass() host  %.1f us" % host(m.phi_pass))
print("_launch_elbo() host %.1f us" % host(m._launch_elbo))
print("theta() %.1f us, _statics() %.1f us" % (host(m.theta), host(m._statics)))
lib = _lib.get_lib()
print("stream_ptr %.1f us" % host(_lib.stream_ptr if hasattr(_lib, "stream_ptr") else (lambda: torch.cuda.current_stream().cuda_stream)))
