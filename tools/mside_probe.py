"""M-side timing: the fused ELBO + gradient launch against its parts (Kuu backward pass alone through asvgp_kuu_inverse_band_1d).
usage: python tools/mside_probe.py [M=2048]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
from asvgp_amd import _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = 1_000_000
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(0.01)
def timed(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, t_enq * 1e6 / reps
for algo in (0, 2):
    A.set_band_algorithm(algo)
    us, enq = timed(model._launch_elbo)
    print("band algorithm %d: ELBO + gradient launch %.1f us per call (host enqueue %.1f us)  -> %s" % (algo, us, enq, model._out[:4].tolist()))
    if algo == 0:
        us0 = us
A.set_band_algorithm(0)
os.environ["ASVGP_CHAIN_STAMPS"] = "1"
_lib.get_lib().asvgp_debug_reload_env()
for _ in range(3):
    model._launch_elbo()
torch.cuda.synchronize()
del os.environ["ASVGP_CHAIN_STAMPS"]
_lib.get_lib().asvgp_debug_reload_env()
k, D = 4, 1
ws = model._elbo_ws.cpu().numpy()
off = 9 * (k + 1) * M + 2 * M * D
st = ws[off + 8: off + 8 + 12].reshape(3, 4)
t0 = st[:, 0].min()
for name, row in zip(("P chain (workgroup 0: left half / whole)", "Kuu backward (workgroup 1)", "workgroup 2 (P chain, right half; or helper 0)"), st):
    print("%-38s start +%.2f us | phase mark +%.2f us | end +%.2f us (100 MHz wall clock, relative to the earliest start)%s" % (
        name, (row[0] - t0) / 100, (row[0] - t0 + row[1]) / 100, (row[0] - t0 + row[2]) / 100,
        (" | solve done +%.2f us" % ((row[0] - t0 + row[3]) / 100)) if row[3] > 0 else ""))
ex = ws[off + 24 + 4: off + 24 + 7]
print("P workgroup tail (relative to its start): barrier behind the solve +%.2f us | trace / quadratic-form loop done +%.2f us | workgroup sums done +%.2f us" % tuple(
    (st[0][0] - t0 + v) / 100 for v in ex))
lv = ws[off + 32: off + 64]
print("P chain (left / whole) per-phase cycles (s_memtime deltas): " + " ".join("%.1fK" % (v / 1e3) for v in lv if v > 0))
E = (k + 1) * M
lp = ws[5 * E + 128: 5 * E + 160]
print("P chain, right workgroup, per-phase cycles:                 " + " ".join("%.1fK" % (v / 1e3) for v in lp if v > 0))
print("  [pre-pass | forward levels 0.. (the top level preceded by the right -> left hand-over) | root | backward levels top..0 (the top one includes the left -> right hand-over) | outputs]")
feat = model.inducing_features
us, enq = timed(lambda: feat.inverse_band(model.kernel))
print("Kuu chain alone (asvgp_kuu_inverse_band_1d: assemble + planned backward pass): %.1f us per call (host enqueue %.1f us)" % (us, enq))

# a dependent loop without any Phi pass: launch, poll the pinned result mirror, repeat (what an optimiser does per evaluation)
model.elbo_and_grad_host()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    r = model.elbo_and_grad_host()
dt = (time.perf_counter() - t0) / 300 * 1e6
print("elbo_and_grad_host() back to back: %.1f us per evaluation (kernel %.1f us -> %.1f us of launch path, dispatch latency and mirror round trip)" % (dt, us0, dt - us0))
t0 = time.perf_counter()
for _ in range(300):
    r = model.elbo_and_grad().tolist()
dt2 = (time.perf_counter() - t0) / 300 * 1e6
print("elbo_and_grad().tolist() back to back (stream path: two device-to-host copies): %.1f us per evaluation" % dt2)
