"""Config 3 (M = 4096, Matern-5/2, lengthscale 0.005, cond(Kuu) ~ 1e9): band(Kuu^-1) and its lengthscale tangent from the GPU
(asvgp_kuu_inverse_band_1d, band algorithms 0 and 1) against the fp64 oracle, entry by entry and contracted with A."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
from asvgp_amd.inducing_features import SplineFeatures1D
from oracle import asvgp_oracle as O
N, M = 1_000_000, 4096
v, l, s = 1.0, 0.005, 0.01
rng = np.random.default_rng(1)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
ob = O.Basis(4, 0, 1, M)
Ab, b, yy = O.sufficient_stats_direct(ob, x, y.reshape(-1, 1))
Kuu, dKl = O.make_Kuu(ob, O.MATERN52, v, l, want_dl=True)
LK, dLK = O.cholesky_band_jvp(Kuu, dKl)
SK, dSK = O.inverse_from_cholesky_band(LK, dLK)
kern = A.Matern52(variance=v, lengthscales=l)
bs = A.B4Spline(0, 1, M)
feat = SplineFeatures1D(kern, bs)
for algo in (0, 1):
    A.set_band_algorithm(algo)
    K, dK, S, dS, ld2, info = feat.inverse_band(kern)
    K, dK, S, dS = [t.cpu().numpy() for t in (K, dK, S, dS)]
    rel = lambda a, r: np.max(np.abs(a - r)) / np.max(np.abs(r))
    print("band algorithm %d: Kuu %.2e  dKuu/dl %.2e  band(Kuu^-1) %.2e  d band(Kuu^-1)/dl %.2e (max abs diff / max abs, against the fp64 oracle)" % (algo, rel(K, Kuu), rel(dK, dKl), rel(S, SK), rel(dS, dSK)))
    print("   <dS, A>: gpu %.10e oracle fp64 %.10e  rel diff %.2e;  <S, A>: rel diff %.2e" % (O.band_sym_dot(dS, Ab), O.band_sym_dot(dSK, Ab), abs(O.band_sym_dot(dS, Ab) - O.band_sym_dot(dSK, Ab)) / abs(O.band_sym_dot(dSK, Ab)),
          abs(O.band_sym_dot(S, Ab) - O.band_sym_dot(SK, Ab)) / abs(O.band_sym_dot(SK, Ab))))
