"""Diagnostic: when does the prior chain (side stream) start/end relative to the Phi pass (main stream)?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asvgp_amd as A
from asvgp_amd import _lib
lib = _lib.get_lib()
nwg = int(os.environ.get("PHI_WG", "248")); prio = int(os.environ.get("SIDE_PRIO", "-1"))
lib.asvgp_set_phi_workgroups(nwg)
N, M = 10_000_000, 2048
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
model = A.GPR_1d((torch.from_numpy(x).cuda().reshape(-1, 1), torch.from_numpy(y).cuda().reshape(-1, 1)), A.Matern32(variance=1.0, lengthscales=0.05), A.B4Spline(0, 1, M))
model.likelihood.variance.assign(0.01)
main = torch.cuda.current_stream(); side = torch.cuda.Stream(priority=prio)
E = lambda: torch.cuda.Event(enable_timing=True)
rows = []
evs = []
SYNC = int(os.environ.get("SYNC", "1"))
for it in range(12):
    t0, ps, pe, f1, d0, d1 = E(), E(), E(), E(), E(), E()
    t0.record(main)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ps.record(side); model.launch_prior_chain(); pe.record(side)
    model.phi_pass(); f1.record(main)
    main.wait_event(pe)
    d0.record(main); model.launch_data_chain(); d1.record(main)
    if SYNC: torch.cuda.synchronize()
    evs.append((t0, ps, pe, f1, d0, d1))
torch.cuda.synchronize()
for (t0, ps, pe, f1, d0, d1) in evs:
    rows.append([t0.elapsed_time(e) * 1e3 for e in (ps, pe, f1, d0, d1)])
if not SYNC:
    for i in range(2, 12): print("step %2d: " % i + "  ".join("%6.0f" % v for v in rows[i]), "  t0 since prev t0: %.0f" % (evs[i-1][0].elapsed_time(evs[i][0]) * 1e3))
r = np.array(rows[2:])
print("PHI_WG=%d SIDE_PRIO=%d  (us after step start; median of 10)" % (nwg, prio))
print("prior start %.0f  prior end %.0f | phi end %.0f | data start %.0f  data end %.0f" % tuple(np.median(r, 0)))
