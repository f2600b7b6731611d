"""SplineFeatures1D: mirror of asvgp/inducing_features.py:6-48 on the HIP library."""
import ctypes

import torch

from . import kernels
from ._lib import Handle, check, get_lib, stream_ptr

TERM_ORDER = ("A", "B", "C", "D", "BC", "BC_grad", "BC_ggrad", "BC_ggrad_none", "BC_none_ggrad")
KIND_TERMS = {
    0: ("A", "B", "BC"),                                   # inducing_features.py:16-20
    1: ("A", "B", "C", "BC", "BC_grad"),                  # inducing_features.py:22-30
    2: TERM_ORDER,                                         # inducing_features.py:32-44
}


class SplineFeatures1D:
    def __init__(self, kernel, basis):
        self.kernel = kernel
        self.basis = basis
        self._stacked = {}

    def static_stack(self, kind):
        """(n_terms, k+1, M) device array of the static bands this kernel kind combines, in C-ABI term order."""
        if kind not in self._stacked:
            missing = [nm for nm in KIND_TERMS[kind] if not hasattr(self.basis, nm)]
            if missing:  # the reference fails the same way: AttributeError on self.basis.<band>
                raise AttributeError("'%s' object has no attribute '%s'" % (type(self.basis).__name__, missing[0]))
            self._stacked[kind] = torch.stack([getattr(self.basis, nm) for nm in KIND_TERMS[kind]]).contiguous()
        return self._stacked[kind]

    def make_Kuu(self, kernel, with_dl=False):
        """inducing_features.py:12-44: banded Kuu (k+1, M) for a Matern-1/2, 3/2 or 5/2 kernel."""
        assert isinstance(kernel, (kernels.Matern12, kernels.Matern32, kernels.Matern52))
        lib = get_lib()
        S = self.static_stack(kernel.kind)
        c = (ctypes.c_double * 9)()
        dc = (ctypes.c_double * 9)()
        n = ctypes.c_int(0)
        check(lib.asvgp_matern_coeffs(kernel.kind, float(kernel.variance), float(kernel.lengthscales), c, dc,
                                      ctypes.byref(n)), "matern_coeffs")
        k, M = self.basis.order, self.basis.m
        Kuu = torch.empty((k + 1, M), dtype=torch.float64, device=S.device)
        dK = torch.empty_like(Kuu) if with_dl else None
        check(lib.asvgp_kuu_assemble(S.data_ptr(), n.value, c, dc, M, k, Kuu.data_ptr(),
                                     dK.data_ptr() if with_dl else None, stream_ptr()), "kuu_assemble")
        return (Kuu, dK) if with_dl else Kuu

    def inverse_band(self, kernel):
        """(Kuu, dKuu/dl, S, dS/dl, logdet2) with S = band(Kuu^-1) = inverse_from_cholesky_band(cholesky_band(Kuu)) (gpr.py:56-59),
        its exact lengthscale tangent (forward mode through the factorisation: band(Kuu^-1 dKuu Kuu^-1) = -dS/dl) and
        logdet2 = [log|Kuu|, d log|Kuu|/dl] on the device - asvgp_kuu_inverse_band_1d, with the handle's prior plan (forward pass
        on the host in long double over the distinct nodes) when the bands have the Toeplitz structure."""
        import numpy as np
        lib = get_lib()
        S = self.static_stack(kernel.kind)
        k, M = self.basis.order, self.basis.m
        if getattr(self, "_h", None) is None:
            self._h = Handle()
            self._h_kind = None
            self._ws = torch.zeros(lib.asvgp_elbo_workspace_bytes(M, k, 1) // 8, dtype=torch.float64, device=S.device)
            self._info = torch.zeros(2, dtype=torch.int32, device=S.device)
        if self._h_kind != kernel.kind:
            self._h.prior_plan(np.ascontiguousarray(S.cpu().numpy()), S.shape[0], M, k)
            self._h_kind = kernel.kind
        out = [torch.empty((k + 1, M), dtype=torch.float64, device=S.device) for _ in range(4)]
        logdet2 = torch.empty(2, dtype=torch.float64, device=S.device)
        check(lib.asvgp_kuu_inverse_band_1d(self._h.ptr, S.data_ptr(), kernel.kind, float(kernel.variance), float(kernel.lengthscales),
                                            M, k, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                            logdet2.data_ptr(), self._info.data_ptr(), self._ws.data_ptr(), self._ws.numel() * 8,
                                            stream_ptr()), "kuu_inverse_band_1d")
        return out[0], out[1], out[2], out[3], logdet2, self._info

    def make_Kuf(self, X, sparse=True):
        """inducing_features.py:47-48 (the `sparse` argument is ignored there too)."""
        return self.basis.evaluate_basis(X, dx=0, sparse=True)
