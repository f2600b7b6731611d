"""Diagnostic: s_memtime stamps of one block column of the persistent band Cholesky (ASVGP_BB_STAMP_COL, 128 x 128 B3 Kronecker model)."""
import os, sys, ctypes
os.environ.setdefault("ASVGP_BB_STAMP_COL", "256")
import numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
from asvgp_amd import _lib
N, m = 200_000, 128
rng = np.random.default_rng(1)
X = rng.uniform(1e-9, 1 - 1e-9, size=(N, 2)); y = np.sin(12 * X[:, :1]) * np.cos(9 * X[:, 1:]) + 0.1 * rng.standard_normal((N, 1))
model = A.GPR_kron((torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()), [A.Matern32(lengthscales=0.2), A.Matern32(lengthscales=0.2)], [A.B3Spline(0, 1, m), A.B3Spline(0, 1, m)])
model.likelihood.variance.assign(0.01)
model.twisted = False        # one system, one launch: the two systems of the two-sided form run concurrently and would both write the stamps
for _ in range(3): e = model.elbo().item()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib = ctypes.CDLL(os.path.join(os.path.dirname(A.__file__), "libasvgp_hip.so"))
lib.asvgp_debug_bbp_stamps(buf)
st = np.array(list(buf)[:16], dtype=np.float64)
idx = [0, 1, 2, 3, 5, 6, 8, 9]
names = ["init (band -> LDS window -> accumulators)", "updates by block columns <= c-2 (incl. waiting for them)", "update by block column c-1 (wave 0: diagonal tiles)",
         "diagonal block: 16 columns, MFMA, 16 columns", "block inverse + diagonal block to the band", "barrier, MFMA solve of the rows below, stores", "fence + flag"]
for n, a, b in zip(names, idx[:-1], idx[1:]):
    print("%-62s %8.0f cycles" % (n, st[b] - st[a]))
print("flag of c-1 seen -> flag of c published: %.0f cycles" % (st[9] - st[2]))

print("inside the diagonal block: first 16 columns %.0f | 16x16x16 Schur update %.0f | last 16 columns %.0f" % (st[13] - st[3], st[14] - st[13], st[5] - st[14]))
print("inside the inverse phase: two triangles + L^-1 to the LDS %.0f | off-diagonal block (two MFMA products) %.0f | diagonal block to the band %.0f | barrier %.0f" % (
    st[10] - st[5], st[11] - st[10], st[12] - st[11], st[6] - st[12]))
