"""Randomised check of the operator kernels cholesky_band / inverse_from_cholesky_band (LDS-resident forms for M (k+1) doubles <= 156 KB, the
register-window sweeps beyond) against the oracle's sweeps.  usage: python tests/sweeps/fuzz_band_ops.py [n_cases] [seed]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
from asvgp_amd import banded
from oracle import asvgp_oracle as O
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
for case in range(n_cases):
    k = int(rng.integers(1, 9))
    M = int(rng.choice([k + 2, 2 * k + 1, 2 * k + 2, 2 * k + 3, int(rng.integers(2 * k + 2, 400)), int(rng.integers(400, 5000)), int(rng.integers(3000, 9000))]))
    lower = np.zeros((k + 1, M))
    for d in range(k + 1):
        lower[d, :M - d] = rng.normal(size=M - d) * (0.3 ** d)
    lower[0] = np.abs(lower[0]) + 1.5 * np.sum(np.abs(lower[1:]), axis=0) + 0.5       # diagonally dominant: positive definite
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    L = banded.cholesky_band(dev(lower))
    oL = O.cholesky_band(lower)
    S = banded.inverse_from_cholesky_band(L)
    oS = O.inverse_from_cholesky_band(oL)
    eL = np.max(np.abs(L.cpu().numpy() - oL)) / np.max(np.abs(oL))
    eS = np.max(np.abs(S.cpu().numpy() - oS)) / np.max(np.abs(oS))
    zl = np.array_equal(L.cpu().numpy() == 0, oL == 0)
    if not (eL <= 1e-13 and eS <= 1e-12 and zl):
        fails += 1
        print("FAIL case %d: k %d M %d  L %.2e S %.2e zeros %s" % (case, k, M, eL, eS, zl), flush=True)
print("cases %d, failures %d" % (n_cases, fails))
