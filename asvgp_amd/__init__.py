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
    """0 = auto, 1 = sequential single-wave sweeps, 2 = block cyclic reduction on the GPU, 3 = block cyclic reduction with the
    planned (host, long double) prior forward pass, 4 = the planned chains on the matrix cores (asvgp_set_band_algorithm) - applied to every live model handle and to
    models created later."""
    from ._lib import AsvgpError, get_lib, set_default_algorithms
    get_lib()
    if int(algo) not in (0, 1, 2, 3, 4):
        raise AsvgpError("set_band_algorithm: 0..4")
    set_default_algorithms(band=int(algo))


def set_phi_algorithm(algo):
    """0 = auto, 1 = fp64 LDS atomic band scatter, 3 = fixed-point band scatter, 5 = fixed-point centred-moment scatter,
    6 = tile sort + register moments (asvgp_set_phi_algorithm) - applied to every live model handle and to models created later."""
    from ._lib import AsvgpError, get_lib, set_default_algorithms
    get_lib()
    if int(algo) not in (0, 1, 3, 5, 6):
        raise AsvgpError("set_phi_algorithm: 0, 1, 3, 5 or 6")
    set_default_algorithms(phi=int(algo))


def set_phi_input_order(order):
    """Input order of the tile-sort Phi pass (asvgp_set_phi_input_order): 0 = probe once per data buffer (default), 1 = unsorted,
    2 = time series (sorted / locally sorted x: the instantiation with the time-series front loop).  The statistics are the same
    either way; applied to every live model handle and to models created later."""
    from ._lib import AsvgpError, get_lib, set_default_algorithms
    get_lib()
    if int(order) not in (0, 1, 2):
        raise AsvgpError("set_phi_input_order: 0 (probe), 1 (unsorted) or 2 (time series)")
    set_default_algorithms(phi_order=int(order))


def set_prior_forward(mode):
    """Where the forward (elimination) half of the Kuu chain runs: 0 = on the host in x87 long double (default; prior_plan.cpp), 1 = on the
    GPU in double-double arithmetic (prior_dd.hip: no host stage, no dependence on the host's long double) - asvgp_set_prior_forward,
    applied to every live model handle and to models created later."""
    from ._lib import AsvgpError, get_lib, set_default_algorithms
    get_lib()
    if int(mode) not in (0, 1):
        raise AsvgpError("set_prior_forward: 0 (host, long double) or 1 (GPU, double-double)")
    set_default_algorithms(prior_forward=int(mode))
