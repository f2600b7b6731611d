"""Per-term accuracy of the bound at the BASELINE size: [log|Kuu|, log|P|, tr(Kuu^-1 A), |c|^2] of every band algorithm against the
oracle's fp64 and long-double evaluations (gpurun: python tests/sweeps/elbo_terms.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import asvgp_amd as A
import bench
from oracle import asvgp_oracle as O
N, M = int(os.environ.get("N", 10_000_000)), int(os.environ.get("M", 2048))
v, l, s = 1.0, float(os.environ.get("L", 0.05)), 0.01
x, y = bench.synth(N)
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=v, lengthscales=l), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(s)
ob = O.Basis(4, 0, 1, M)
Ab, b, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
ld = np.longdouble
# long-double terms through the oracle's extended evaluator internals
Kuu = O.make_Kuu(ob, O.MATERN32, v, l)
oe, parts = O.elbo_1d(Kuu, Ab, b, yy, N, v, s)
print("oracle f64: logK %.12f logP %.12f trKA %.9f cc %.6f elbo %.6f" % (parts["logdet_K"], parts["logdet_P"], parts["trace_term"], float(np.sum(parts["c"] ** 2)), oe))
ee, ge = O.elbo_grad_1d_extended(ob, O.MATERN32, Ab, b, yy, N, v, l, s)
print("long double elbo %.6f grad %s" % (ee, ge))
for algo in (1, 2, 3):
    A.set_band_algorithm(algo)
    r = model._launch_elbo().cpu().numpy()
    model._check_pd()
    print("algo %d: logK %.12f logP %.12f trKA %.9f cc %.6f elbo %.6f (err %.4f) grad rel %s" % (algo, r[4], r[5], r[6], r[7], r[0], r[0] - ee, (r[1:4] - ge) / ge))
    print("   term errors vs oracle f64: dlogK %.3e dlogP %.3e dtrKA %.3e (-> elbo %.4f) dcc %.3e" % (r[4] - parts["logdet_K"], r[5] - parts["logdet_P"], r[6] - parts["trace_term"], 0.5 * (r[6] - parts["trace_term"]) / s, r[7] - float(np.sum(parts["c"] ** 2))))
A.set_band_algorithm(0)
