"""Where the dependent step's time goes: the same host loop as bench.py's schedule C with parts removed.  usage: python tools/dep_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
from asvgp_amd import _lib
N, M = 10_000_000, 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
xd, yd = torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)
theta0 = (1.0, 0.05, 0.01)


def model(defer, wgs=240):
    m = A.GPR_1d((xd, yd), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
    m.likelihood.variance.assign(0.01)
    m._h.set_phi_workgroups(wgs)
    if defer:
        m._h.set_phi_deferred_reduce(1)
    return m


def run(name, phi, defer, event, prio=-1, steps=200, wgs=240, phi_first=False, nl=2, split=False, worker=False):
    lanes = [model(defer or split, wgs) for _ in range(nl)]
    if worker:                                                     # the host forward pass on the handle's worker thread (asvgp_set_deferred_forward_pass(h, 2))
        for ln in lanes:
            ln._h.set_deferred_forward_pass(2)
    s_n, s_m = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)
    evs = [torch.cuda.Event() for _ in range(nl)]
    th = theta0
    def phi_into(k):
        _lib.set_stream(s_n)
        lanes[k].phi_pass(allreduce=False)
        if event and not split:
            evs[k].record(s_n)
    ahead = nl - 1
    if phi:
        for k in range(ahead):
            phi_into(k)
            if split:
                lanes[k].phi_reduce(); evs[k].record(s_n)
    torch.cuda.synchronize()
    t_all = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ln = lanes[i % nl]
            nxt = (i + ahead) % nl
            if phi and phi_first:
                phi_into(nxt)                                      # (split: the streaming kernel only, reduce parked)
            ln.kernel.variance.assign(th[0]); ln.kernel.lengthscales.assign(th[1]); ln.likelihood.variance.assign(th[2])
            if event and not evs[i % nl].query():
                s_m.wait_event(evs[i % nl])
            _lib.set_stream(s_m)
            tok = ln.launch_elbo_host()
            if phi and not phi_first:
                phi_into(nxt)
            if split:
                _lib.set_stream(s_n)
                lanes[nxt].phi_reduce(); evs[nxt].record(s_n)
            r = ln.read_elbo_host(tok, check_pd=False)       # (the variants without the event race on the statistics: timing only)
            d = ((r[0] * 1e3) % 1.0 - 0.5) if np.isfinite(r[0]) else 0.0
            th = tuple(t * (1.0 + 1e-6 * d) for t in theta0)
        torch.cuda.synchronize()
        t_all.append((time.perf_counter() - t0) / steps * 1e6)
    _lib.set_stream(None)
    print("%-70s %6.1f us per step (%s)" % (name, np.median(t_all), ", ".join("%.1f" % v for v in t_all)), flush=True)


run("M side only (theta, launch, poll)", False, False, False)
run("M side only, forward pass on the worker thread", False, False, False, worker=True)
run("bench.py round-3 first form: ELBO launch, Phi + reduce + event (two buffer sets)", True, False, True)
run("Phi KERNEL first, ELBO launch, then reduce + event (two buffer sets) = bench.py's `value`", True, False, True, phi_first=True, split=True)
run("same, forward pass on the worker thread (bench.py --worker-forward)", True, False, True, phi_first=True, split=True, worker=True)
run("same, both streams at default priority", True, False, True, phi_first=True, split=True, prio=0)
run("same, Phi grid 248 workgroups", True, False, True, phi_first=True, split=True, wgs=248)
run("three buffer sets: ELBO launch first, Phi of step i+2 behind it", True, False, True, nl=3)
run("same, both streams at default priority", True, False, True, nl=3, prio=0)
run("three buffer sets, split Phi (kernel, ELBO launch, reduce + event)", True, False, True, nl=3, phi_first=True, split=True)
run("three buffer sets, forward pass on the worker thread (bench.py --worker-forward)", True, False, True, nl=3, worker=True)
