"""Phi-pass kernel time per algorithm on unsorted / sorted / clustered inputs (N = 10M, M = 2048)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
N, M = 10_000_000, 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
xc = np.clip(0.5 + 0.02 * rng.standard_normal(N), 1e-9, 1 - 1e-9)       # clustered: most points in ~80 cells
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, xs in (("unsorted", x), ("sorted", np.sort(x)), ("clustered", xc)):
    xd = torch.from_numpy(xs).cuda().reshape(-1, 1); yd = torch.from_numpy(y).cuda().reshape(-1, 1)
    for algo in (1, 3, 5):
        A.set_phi_algorithm(algo)
        m = A.GPR_1d((xd, yd), A.Matern32(), A.B4Spline(0, 1, M))
        print("%-10s algo %d  %.1f us" % (name, algo, t(m.phi_pass)), flush=True)
A.set_phi_algorithm(0)
