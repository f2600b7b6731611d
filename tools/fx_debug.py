import sys, numpy as np, torch
sys.path.insert(0, ".")
import asvgp_amd as A
rng = np.random.default_rng(64 + 20000)
N, M, order = 20000, 64, 4
x = rng.uniform(1e-9, 1 - 1e-9, N); y = (np.sin(20 * x) + 0.1 * rng.normal(size=N)).reshape(-1, 1)
bs = A.B4Spline(0, 1, M)
A.set_phi_algorithm(3)
outs = []
for r in range(4):
    m = A.GPR_1d((x.reshape(-1, 1), y), A.Matern12(), bs)
    outs.append(m._stats.cpu().numpy().copy())
nb = (order + 1) * M
for r in range(1, 4):
    d = np.nonzero(outs[r][:nb] != outs[0][:nb])[0]
    print("run", r, "differs at", d[:20], "n", len(d))
    for e in d[:5]:
        print("   ", e, outs[0][e].hex(), outs[r][e].hex(), (outs[r][e] - outs[0][e]) / outs[0][e])
A.set_phi_algorithm(1)
m = A.GPR_1d((x.reshape(-1, 1), y), A.Matern12(), bs)
ref = m._stats.cpu().numpy()
print("max rel-to-max diff vs fp64 atomics:", np.max(np.abs(ref[:nb] - outs[0][:nb])) / np.max(np.abs(ref[:nb])))
