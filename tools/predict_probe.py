"""Streaming posterior throughput: predict_f_device on n* test points (8 B in + 16 B out per point)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
N, M = 1_000_000, 2048
rng = np.random.default_rng(0)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(0.01)
for ns in (100_000, 10_000_000, 50_000_000):
    xs = torch.rand(ns, dtype=torch.float64, device="cuda") * 0.998 + 0.001
    for srt in (False, True):
        xq = torch.sort(xs)[0] if srt else xs
        for _ in range(3):                                    # (allocator warm-up: the outputs of the first calls at a new size are cudaMalloc'ed)
            m, v = model.predict_f_device(xq.reshape(-1, 1))
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):                                    # median of three groups of five back-to-back calls
            t0 = time.perf_counter()
            for _ in range(5): m, v = model.predict_f_device(xq.reshape(-1, 1))
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5)
        dt = sorted(ts)[1]
        print("n* %9d %s: %8.1f us  %7.1f Gpoints/s  %6.2f TB/s (24 B/point)" % (ns, "sorted  " if srt else "unsorted", dt * 1e6, ns / dt / 1e9, 24 * ns / dt / 1e12), flush=True)
