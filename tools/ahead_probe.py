"""M side alone, dependent evaluations (theta of the next from the host-read result of this one): the ordinary launch against launch-ahead
(asvgp_elbo_grad_ahead_1d + asvgp_elbo_publish_theta).  usage: python tools/ahead_probe.py [M=2048]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = 1_000_000
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(0.01)
th0 = (1.0, 0.05, 0.01)

def nxt(r):
    d = (r[0] * 1e3) % 1.0 - 0.5
    return tuple(t * (1.0 + 1e-6 * d) for t in th0)

def ordinary(n):
    th = th0
    for _ in range(n):
        r = model.read_elbo_host(model.launch_elbo_host(th))
        th = nxt(r)
    return r

def ahead(n):
    th = th0
    tok = model.launch_elbo_ahead()
    for i in range(n):
        model.publish_theta(th)
        tok_next = model.launch_elbo_ahead() if i + 1 < n else None
        r = model.read_elbo_host(tok)
        th = nxt(r)
        tok = tok_next
    return r

for name, fn in (("ordinary launch (launch_elbo_host + read_elbo_host)", ordinary), ("launch-ahead (publish_theta, next launch_elbo_ahead, read_elbo_host)", ahead)):
    fn(50)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        r = fn(200)
        ts.append((time.perf_counter() - t0) / 200 * 1e6)
    print("%-75s %.1f us per dependent evaluation (median of 7 x 200; min %.1f)  elbo %.6f" % (name, float(np.median(ts)), min(ts), r[0]))
