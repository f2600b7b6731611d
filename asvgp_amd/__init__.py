"""asvgp_amd: MI355X-native (gfx950 HIP) implementation of the ASVGP ELBO/posterior hot path, behind the reference's
own model/operator surface.  See DESIGN.md / INTEGRATION.md.  There is no CPU fallback: without the HIP library and a
ROCm device every compute entry point raises."""
from . import banded, basis, dist, inducing_features, kernels, utils  # noqa: F401
from .basis import B1Spline, B2Spline, B3Spline, B4Spline, B5Spline, B6Spline  # noqa: F401
from .gpr import GPR_1d, GPR_additive, GPR_kron  # noqa: F401
from . import kronecker  # noqa: F401
from .inducing_features import SplineFeatures1D  # noqa: F401
from .kernels import Gaussian, Matern12, Matern32, Matern52  # noqa: F401


def set_band_algorithm(algo):
    """0 = auto, 1 = sequential single-wave sweeps, 2 = block cyclic reduction (asvgp_set_band_algorithm)."""
    from ._lib import check, get_lib
    check(get_lib().asvgp_set_band_algorithm(int(algo)), "set_band_algorithm")


def set_phi_algorithm(algo):
    """0 = auto, 1 = fp64 LDS atomic scatter, 2 = counting sort + per-cell moments, 3 = fixed-point band scatter,
    4 = per-cell buckets + moments (asvgp_set_phi_algorithm)."""
    from ._lib import check, get_lib
    check(get_lib().asvgp_set_phi_algorithm(int(algo)), "set_phi_algorithm")
