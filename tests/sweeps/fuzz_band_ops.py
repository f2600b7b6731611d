"""Randomised check of the operator kernels cholesky_band / inverse_from_cholesky_band and of their adjoints (lane-uniform register-window
kernels, the band streamed through the LDS in segments; ASVGP_BAND_OPS_SEG_BLOCKS=1..3 forces short segments) against the oracle's sweeps.
usage: python tests/sweeps/fuzz_band_ops.py [n_cases] [seed]"""
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
    dom = np.sum(np.abs(lower[1:]), axis=0)                                              # sub-diagonal entries of the column ...
    for d in range(1, k + 1):
        dom[d:] += np.abs(lower[d, :M - d])                                               # ... and of the row (the same entries, mirrored)
    lower[0] = np.abs(lower[0]) + dom + 0.5                                               # strictly diagonally dominant: positive definite
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    L = banded.cholesky_band(dev(lower))
    oL = O.cholesky_band(lower)
    S = banded.inverse_from_cholesky_band(L)
    oS = O.inverse_from_cholesky_band(oL)
    eL = np.max(np.abs(L.cpu().numpy() - oL)) / np.max(np.abs(oL))
    eS = np.max(np.abs(S.cpu().numpy() - oS)) / np.max(np.abs(oS))
    zl = np.array_equal(L.cpu().numpy() == 0, oL == 0)
    # the adjoints through torch.autograd (asvgp_cholesky_band_vjp / asvgp_inverse_from_cholesky_band_vjp) against the oracle's adjoint sweeps
    eV = eW = 0.0
    if M <= 9000:
        Lbar = np.zeros((k + 1, M)); Sbar = np.zeros((k + 1, M))
        for d in range(k + 1):
            Lbar[d, :M - d] = rng.normal(size=M - d); Sbar[d, :M - d] = rng.normal(size=M - d)
        bt = dev(lower).requires_grad_(True)
        Lt = banded.cholesky_band(bt)
        (Lt * dev(Lbar)).sum().backward()
        oK = O.cholesky_band_vjp(oL, Lbar)
        eV = np.max(np.abs(bt.grad.cpu().numpy() - oK)) / np.max(np.abs(oK))
        Lr = dev(oL).requires_grad_(True)
        St = banded.inverse_from_cholesky_band(Lr)
        (St * dev(Sbar)).sum().backward()
        oLb = O.inverse_from_cholesky_band_vjp(oL, oS, Sbar)
        eW = np.max(np.abs(Lr.grad.cpu().numpy() - oLb)) / np.max(np.abs(oLb))
    if not (eL <= 1e-13 and eS <= 1e-12 and zl and eV <= 1e-11 and eW <= 1e-11):
        fails += 1
        print("FAIL case %d: k %d M %d  L %.2e S %.2e zeros %s  vjp(chol) %.2e vjp(inverse) %.2e" % (case, k, M, eL, eS, zl, eV, eW), flush=True)
print("cases %d, failures %d" % (n_cases, fails))
