"""Operator VJP timings at the headline band size (M = 2048, k = 4): the C-ABI adjoints of cholesky_band and inverse_from_cholesky_band
(wave-parallel since round 3) against the forward operators.  usage: python tools/vjp_probe.py [M=2048] [k=4]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
from asvgp_amd import banded, _lib
lib = _lib.get_lib()
def chol_vjp(L, Lbar):
    Kbar, work = torch.empty_like(L), torch.empty_like(L)
    _lib.check(lib.asvgp_cholesky_band_vjp(L.data_ptr(), Lbar.data_ptr(), Kbar.data_ptr(), work.data_ptr(), L.shape[1], L.shape[0] - 1, _lib.stream_ptr()), "vjp")
    return Kbar
def inv_vjp(L, S, Sbar):
    Lbar, work = torch.empty_like(L), torch.empty_like(L)
    _lib.check(lib.asvgp_inverse_from_cholesky_band_vjp(L.data_ptr(), S.data_ptr(), Sbar.data_ptr(), Lbar.data_ptr(), work.data_ptr(), L.shape[1], L.shape[0] - 1, _lib.stream_ptr()), "vjp")
    return Lbar
from oracle import asvgp_oracle as O
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(0)
L0 = rng.uniform(-0.3, 0.3, (k + 1, M)); L0[0] = rng.uniform(1.0, 2.0, M)
for d in range(1, k + 1):
    L0[d, M - d:] = 0.0
Ld = O.unpack_banded_matrix_to_dense(L0, k, 0)
K = torch.from_numpy(np.ascontiguousarray(O.pack_dense_matrix_to_banded(Ld @ Ld.T, k, 0))).cuda()
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
L = banded.cholesky_band(K)
S = banded.inverse_from_cholesky_band(L)
Lbar = torch.from_numpy(rng.normal(size=(k + 1, M))).cuda()
Sbar = torch.from_numpy(rng.normal(size=(k + 1, M))).cuda()
print("M = %d, k = %d" % (M, k))
print("cholesky_band                      %8.1f us" % timed(lambda: banded.cholesky_band(K)))
print("inverse_from_cholesky_band         %8.1f us" % timed(lambda: banded.inverse_from_cholesky_band(L)))
print("cholesky_band_vjp                  %8.1f us" % timed(lambda: chol_vjp(L, Lbar)))
print("inverse_from_cholesky_band_vjp     %8.1f us" % timed(lambda: inv_vjp(L, S, Sbar)))
kb = chol_vjp(L, Lbar).cpu().numpy()
lb = inv_vjp(L, S, Sbar).cpu().numpy()
ok = O.cholesky_band_vjp(L.cpu().numpy(), Lbar.cpu().numpy())
ol = O.inverse_from_cholesky_band_vjp(L.cpu().numpy(), S.cpu().numpy(), Sbar.cpu().numpy())
print("max rel diff vs the oracle's adjoint sweeps: cholesky %.2e, inverse %.2e" % (np.abs(kb - ok).max() / np.abs(ok).max(), np.abs(lb - ol).max() / np.abs(ol).max()))
