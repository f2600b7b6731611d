"""Diagnostic: per-level cycle stamps of the two all-GPU BCR chains (band algorithm 2; ASVGP_BCR_STAMPS=1).  The default path (planned
prior chain, fused launch) carries no stamps: its P chain is the same code as the "P chain" printed here."""
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
A.set_band_algorithm(2)
split = len(sys.argv) > 2 and sys.argv[2] == "split"   # the bench's path: prior kernel + data kernel instead of the two-chain kernel
for _ in range(3):
    if split:
        model.launch_prior_chain(); model.launch_data_chain()
    else:
        model.elbo_and_grad()
torch.cuda.synchronize()
ws = model._elbo_ws.cpu().numpy()
k, D = 4, 1
off = 9 * (k + 1) * M + 2 * M * D
st = ws[off + 8: off + 8 + 48]
nb = (M + k - 1) // k
levels = int(np.ceil(np.log2(nb)))
for name, s in (("Kuu chain (Dual)", st[:24]), ("P chain (double+rhs)", st[24:48])):
    s = s[s > 0]
    print(name, "total cycles %.0f  (%.1f us @2.4GHz)" % (s.sum(), s.sum() / 2400))
    print("   gathers + forward level 0 %.0f | forward levels 1.. %s | root %.0f | backward levels %d..0 %s | x, log-det, info %.0f" % (
        s[0], np.round(s[1:levels]).astype(int).tolist(), s[levels], levels - 1,
        np.round(s[levels + 1:2 * levels + 1]).astype(int).tolist(), s[2 * levels + 1] if len(s) > 2 * levels + 1 else -1))
