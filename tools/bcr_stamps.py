"""Diagnostic: per-level cycle stamps of the two BCR chains (ASVGP_BCR_STAMPS=1)."""
import os, sys
os.environ["ASVGP_BCR_STAMPS"] = sys.argv[1] if len(sys.argv) > 1 else "1"   # 2 = stamp a second, warm pass
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
N, M = 1_000_000, 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(0.01)
for _ in range(3): model.elbo_and_grad()
torch.cuda.synchronize()
ws = model._elbo_ws.cpu().numpy()
k, D = 4, 1
off = 9 * (k + 1) * M + 2 * M * D
st = ws[off + 8: off + 8 + 48]
for name, s in (("Kuu chain (Dual)", st[:24]), ("P chain (double+rhs)", st[24:48])):
    s = s[s > 0]
    print(name, "total cycles %.0f  (%.1f us @2.4GHz)" % (s.sum(), s.sum() / 2400))
    print("   prepass %.0f | forward levels %s | root %.0f | backward levels %s | output+logdet %.0f" % (
        s[0], np.round(s[1:10]).astype(int).tolist(), s[10], np.round(s[11:20]).astype(int).tolist(), s[20] if len(s) > 20 else -1))
