"""Phi pass timing by algorithm, input order and shard size (kernel + cross-workgroup reduce, torch events around 20 passes).
usage: python tools/phi_probe.py [N=10000000] [M=2048]"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N)
y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
out = {}
for order_name, xs in (("unsorted", x), ("sorted", np.sort(x)), ("clustered", np.clip(0.5 + 0.08 * rng.standard_normal(N), 1e-9, 1 - 1e-9))):
    xd = torch.from_numpy(xs).cuda().reshape(-1, 1)
    yd = torch.from_numpy(y).cuda().reshape(-1, 1)
    ref = None
    for algo in (5, 6):
        A.set_phi_algorithm(algo)
        m = A.GPR_1d((xd, yd), A.Matern32(), A.B4Spline(0, 1, M))
        st = m._stats.clone()
        if ref is None:
            ref = st
        err = ((st - ref).abs().max() / ref.abs().max()).item()
        for _ in range(3):
            m.phi_pass()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m.phi_pass()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        ts = (m._h.phi_last_input_order() == 2) if algo == 6 else False
        out["%s/algo%d" % (order_name, algo)] = {"us_per_pass": round(us, 1), "max_rel_diff_vs_algo5": err, "time_series_instantiation": ts}
        print("%-10s algo %d%s  %7.1f us per pass (kernel + reduce)   diff vs algo 5: %.2e" % (order_name, algo, " (time-series front loop)" if ts else "", us, err), flush=True)
A.set_phi_algorithm(0)
print(json.dumps({"N": N, "M": M, "results": out}))
