import sys, numpy as np, torch, ctypes
sys.path.insert(0, "/root/repo")
import asvgp_amd as A
from asvgp_amd import _lib
lib = _lib.get_lib()
rng = np.random.default_rng(1234)
N = 10_000_000
x = rng.uniform(1e-9, 1 - 1e-9, N); y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
xd = torch.from_numpy(x).cuda().reshape(-1, 1); yd = torch.from_numpy(y).cuda().reshape(-1, 1)
for n in (5_000_000, 3_000_000, 2_500_000, 2_400_000, 2_000_000, 1_250_000):
    for wg in (0, 240):
        m = A.GPR_1d((xd[:n], yd[:n]), A.Matern32(), A.B4Spline(0, 1, 2048))
        if wg: m._h.set_phi_workgroups(wg)
        for _ in range(3): m.phi_pass()
        torch.cuda.synchronize()
        lib.asvgp_profile_enable(m._h.ptr, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.phi_pass()
        e1.record(); torch.cuda.synchronize()
        ms, ln = ctypes.c_double(0), ctypes.c_int64(0)
        lib.asvgp_profile_read(m._h.ptr, ctypes.byref(ms), ctypes.byref(ln))
        print("n=%d wg=%d: pass %.1f us, kernel %.1f us (order %d)" % (n, wg, e0.elapsed_time(e1) * 50, ms.value / max(ln.value, 1) * 1e3, m._h.phi_last_input_order()))
