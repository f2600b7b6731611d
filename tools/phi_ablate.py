"""Diagnostic: time the Phi-pass kernel stage by stage (ASVGP_PHI_ABLATE) with the in-library HIP events."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
from asvgp_amd import _lib
lib = _lib.get_lib()
N, M = 10_000_000, 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
xd = torch.from_numpy(x).cuda().reshape(-1, 1); yd = torch.from_numpy(y).cuda().reshape(-1, 1)
bs = A.B4Spline(0, 1, M)
model = A.GPR_1d((xd, yd), A.Matern32(), bs)
def timeit(tag):
    for _ in range(3): model.phi_pass()
    torch.cuda.synchronize(); lib.asvgp_profile_enable(1)
    for _ in range(10): model.phi_pass()
    torch.cuda.synchronize()
    ms, n = ctypes.c_double(0), ctypes.c_int64(0); lib.asvgp_profile_read(ctypes.byref(ms), ctypes.byref(n)); lib.asvgp_profile_enable(0)
    us = ms.value / n.value * 1e3
    print("%-40s %8.1f us  %6.2f TB/s-equivalent" % (tag, us, 16 * N / us / 1e6), flush=True)
os.environ["ASVGP_PHI_ABLATE"] = "9"; A.set_phi_algorithm(2)
model.phi_pass(); torch.cuda.synchronize()
E1 = 6 * 2048 + 1
w = model._phi_ws.cpu().numpy()
st = np.array([w[b * E1: b * E1 + 6] for b in range(256)])
print("per-phase cycles (thread 0; mean over 256 blocks; 7 tiles): P1 cell+rank | P2 scan (incl. prefetch issue) | P3 scatter | P4 owner | end barrier")
print(np.round(st.mean(0)[:5]), " total", st.mean(0)[:5].sum())
print("max over blocks", np.round(st.max(0)[:5]))
for algo, modes in ((2, (1, 2, 3, 4, 5, 0)), (3, (0,))):
    for ab in modes:
        os.environ["ASVGP_PHI_ABLATE"] = str(ab)
        A.set_phi_algorithm(algo)
        timeit("algo %d ablate %d" % (algo, ab))
