"""BASELINE config 3 shape on one rank: N/8 = 1.25M points, M = 4096, Matern-5/2, B4Spline: Phi pass and ELBO+grad timings."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
N, M = 1_250_000, 4096
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * (1 - 2e-9) + 1e-9
y = torch.sin(20 * x) + 0.1 * torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
model = A.GPR_1d((x.reshape(-1, 1), y.reshape(-1, 1)), A.Matern52(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(0.01)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("phi pass  %.1f us" % t(model.phi_pass))
for algo in (0, 1):
    A.set_band_algorithm(algo)
    print("algo", algo, "elbo+grad %.1f us" % t(lambda: model.elbo_and_grad(check_pd=False)), model.elbo_and_grad().cpu().numpy()[:4])
