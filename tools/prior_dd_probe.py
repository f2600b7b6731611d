"""Kuu forward pass: host (x87 long double) against GPU (double-double).  Times the double-double kernel alone, the ELBO + gradient
evaluation back to back (what an optimiser sees) with either pass, and a small shard's whole step (N = 1.25M: one rank's share of 8).
usage: python tools/prior_dd_probe.py [M=2048]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
from asvgp_amd import _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(1234)


def mk(N):
    x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
    m = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    m.likelihood.variance.assign(0.01)
    return m


def timed(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps, t_enq * 1e6 / reps


model = mk(1_000_000)
lib = _lib.get_lib()
import ctypes
c = (ctypes.c_double * 16)(); dc = (ctypes.c_double * 16)(); n = ctypes.c_int(0)
lib.asvgp_matern_coeffs(1, 1.0, 0.05, c, dc, ctypes.byref(n))
cn, dcn = np.array(list(c)), np.array(list(dc))
h = model._h
model._launch_elbo(); torch.cuda.synchronize()          # (plans the prior chain)
t0 = time.perf_counter()
for _ in range(200):
    h.prior_forward_device(cn, dcn, 20000)
dt = (time.perf_counter() - t0) / 200 * 1e6
print("asvgp_prior_forward_device (kernel + 13 KB copy back + stream synchronise): %.1f us per call" % dt)
st = np.zeros(64, dtype=np.uint64)
lib.asvgp_prior_forward_stamps(h.ptr, cn.ctypes.data, dcn.ctypes.data, st.ctypes.data, _lib.stream_ptr())
lib.asvgp_prior_forward_stamps(h.ptr, cn.ctypes.data, dcn.ctypes.data, st.ctypes.data, _lib.stream_ptr())
ns = int(st[63]); t = (st[:ns] - st[0]).astype(np.int64)
print("stamps (100 MHz ticks -> us) of thread 0, %d of them: %s" % (ns, " ".join("%.2f" % (v / 100) for v in t)))
print("  [0 start, 1 class maps in the LDS, 2 level-0 blocks, then per level: classes done, barrier, next-level blocks; level 1 adds 4 inner stamps (loaded, factored, solved, products); last two: before the log-det reduce, end]")
for mode in (0, 1):
    h.set_prior_forward(mode)
    us, enq = timed(model._launch_elbo)
    model.elbo_and_grad_host(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        r = model.elbo_and_grad_host()
    dt = (time.perf_counter() - t0) / 300 * 1e6
    print("forward pass %s: ELBO + gradient launch(es) %.1f us per call enqueued ahead (host enqueue %.1f us); elbo_and_grad_host() back to back %.1f us; ELBO %.10g"
          % ("on the GPU (double-double)" if mode else "on the host (long double)", us, enq, dt, r[0]))
h.set_prior_forward(0)
