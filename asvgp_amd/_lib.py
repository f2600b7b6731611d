"""ctypes loader for the C-ABI in include/asvgp_hip.h.  No CPU fallback: a missing library is a loud error."""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libasvgp_hip.so")

_c = ctypes
_P, _I, _L, _D, _Z = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_double, _c.c_size_t

# name -> (restype, argtypes): exactly the prototypes of include/asvgp_hip.h
SIGNATURES = {
    "asvgp_version": (_I, []),
    "asvgp_last_error_string": (_c.c_char_p, []),
    "asvgp_status_name": (_c.c_char_p, [_I]),
    "asvgp_phi_workspace_bytes": (_Z, [_L, _I, _L]),
    "asvgp_create": (_I, [_c.POINTER(_P)]),
    "asvgp_destroy": (_I, [_P]),
    "asvgp_phi_accumulate_1d": (_I, [_P, _P, _P, _L, _L, _P, _L, _D, _I, _L, _P, _P, _Z, _P]),
    "asvgp_set_phi_algorithm": (_I, [_P, _I]),
    "asvgp_phi_last_algorithm": (_I, [_P]),
    "asvgp_set_phi_input_order": (_I, [_P, _I]),
    "asvgp_phi_last_input_order": (_I, [_P]),
    "asvgp_stream_probe": (_I, [_P, _P, _L, _P, _P]),
    "asvgp_debug_reload_env": (_I, []),
    "asvgp_set_deferred_forward_pass": (_I, [_P, _I]),
    "asvgp_prior_publish": (_I, [_P]),
    "asvgp_host_mantissa_bits": (_I, []),
    "asvgp_prior_interior_kuu_host": (_I, [_P, _I, _L, _I, _P, _P, _c.POINTER(_L), _c.POINTER(_L), _P]),
    "asvgp_result_mirror": (_I, [_P, _I, _c.POINTER(_P)]),
    "asvgp_result_mirror_read": (_I, [_P, _c.c_uint64, _P, _D]),
    "asvgp_elbo_grad_ahead_1d": (_I, [_P, _P, _P, _I, _L, _L, _I, _L, _P, _P, _P, _Z, _P]),
    "asvgp_elbo_publish_theta": (_I, [_P, _D, _D, _D]),
    "asvgp_elbo_grad_host_1d": (_I, [_P, _P, _P, _I, _D, _D, _D, _L, _L, _I, _L, _P, _P, _P, _Z, _P, _P, _D]),
    "asvgp_result_mirror_pending": (_c.c_uint64, [_P]),
    "asvgp_set_phi_workgroups": (_I, [_P, _I]),
    "asvgp_set_phi_deferred_reduce": (_I, [_P, _I]),
    "asvgp_phi_reduce_1d": (_I, [_P, _P]),
    "asvgp_phi_index_1d": (_I, [_P, _L, _P, _L, _D, _P, _P]),
    "asvgp_phi_evaluate_1d": (_I, [_P, _L, _P, _L, _D, _I, _I, _P, _P, _P]),
    "asvgp_matern_coeffs": (_I, [_I, _D, _D, _c.POINTER(_D), _c.POINTER(_D), _c.POINTER(_I)]),
    "asvgp_kuu_assemble": (_I, [_P, _I, _c.POINTER(_D), _c.POINTER(_D), _L, _I, _P, _P, _P]),
    "asvgp_cholesky_band": (_I, [_P, _P, _L, _I, _P, _P]),
    "asvgp_inverse_from_cholesky_band": (_I, [_P, _P, _L, _I, _P]),
    "asvgp_solve_triang_mat": (_I, [_P, _P, _P, _L, _I, _L, _I, _P]),
    "asvgp_product_band_band": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "asvgp_transpose_band": (_I, [_P, _P, _L, _I, _I, _P]),
    "asvgp_symmetrise_band": (_I, [_P, _P, _L, _I, _P]),
    "asvgp_unpack_banded_matrix_to_dense": (_I, [_P, _P, _L, _I, _I, _P]),
    "asvgp_pack_dense_matrix_to_banded": (_I, [_P, _P, _L, _I, _I, _P]),
    "asvgp_band_trace_sym": (_I, [_P, _P, _L, _I, _P, _P]),
    "asvgp_cholesky_band_vjp": (_I, [_P, _P, _P, _P, _L, _I, _P]),
    "asvgp_inverse_from_cholesky_band_vjp": (_I, [_P, _P, _P, _P, _P, _L, _I, _P]),
    "asvgp_band_outer_product": (_I, [_P, _P, _L, _L, _I, _D, _P, _P]),
    "asvgp_elbo_workspace_bytes": (_Z, [_L, _I, _L]),
    "asvgp_set_band_algorithm": (_I, [_P, _I]),
    "asvgp_prior_plan_1d": (_I, [_P, _P, _I, _L, _I, _c.POINTER(_I)]),
    "asvgp_prior_table_doubles": (_Z, [_P, _I, _L, _I]),
    "asvgp_prior_forward_host": (_I, [_P, _I, _L, _I, _P, _P, _P, _Z, _P]),
    "asvgp_set_prior_forward": (_I, [_P, _I]),
    "asvgp_prior_forward_device": (_I, [_P, _P, _P, _P, _Z, _P]),
    "asvgp_prior_forward_stamps": (_I, [_P, _P, _P, _P, _P]),
    "asvgp_prior_plan_image_host": (_I, [_P, _I, _L, _I, _P, _P, _P, _P]),
    "asvgp_elbo_grad_1d": (_I, [_P, _P, _P, _I, _D, _D, _D, _L, _L, _I, _L, _P, _P, _P, _Z, _P]),
    "asvgp_kuu_inverse_band_1d": (_I, [_P, _P, _I, _D, _D, _L, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "asvgp_elbo_chain_sync": (_I, [_P, _I]),
    "asvgp_elbo_prior_chain_1d": (_I, [_P, _P, _I, _D, _D, _D, _L, _I, _L, _P, _P, _Z, _P]),
    "asvgp_elbo_data_chain_1d": (_I, [_P, _P, _P, _I, _D, _D, _D, _L, _L, _I, _L, _P, _P, _P, _Z, _P]),
    "asvgp_posterior_prepare_1d": (_I, [_P, _P, _P, _I, _D, _D, _D, _L, _I, _L, _P, _P, _P, _P, _Z, _P]),
    "asvgp_predict_1d": (_I, [_P, _L, _P, _L, _D, _I, _L, _P, _P, _D, _L, _P, _P, _P]),
    "asvgp_predict_1d_h": (_I, [_P, _P, _L, _P, _L, _D, _I, _L, _P, _P, _D, _L, _P, _P, _P]),
    "asvgp_profile_enable": (_I, [_P, _I]),
    "asvgp_profile_read": (_I, [_P, _c.POINTER(_D), _c.POINTER(_L)]),
    "asvgp_kron_stats_doubles": (_Z, [_L, _L, _I]),
    "asvgp_phi_accumulate_kron2d": (_I, [_P, _P, _L, _P, _L, _D, _L, _P, _L, _D, _L, _I, _P, _P]),
    "asvgp_kron_evaluate_2d": (_I, [_P, _L, _P, _L, _D, _P, _L, _D, _L, _I, _P, _P, _P]),
    "asvgp_kron_assemble": (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _D, _P, _P, _P]),
    "asvgp_blockband_cholesky": (_I, [_P, _L, _L, _P, _P, _P, _P]),
    "asvgp_blockband_backsolve": (_I, [_P, _L, _L, _P, _P]),
    "asvgp_predict_kron2d": (_I, [_P, _L, _P, _L, _D, _L, _P, _L, _D, _L, _I, _P, _P, _P, _P, _P, _P]),
    "asvgp_kron_cell_index": (_I, [_P, _L, _P, _L, _D, _P, _L, _D, _P, _P]),
    "asvgp_phi_accumulate_kron2d_sorted": (_I, [_P, _P, _L, _P, _P, _L, _D, _L, _P, _L, _D, _L, _I, _P, _P]),
    "asvgp_phi_accumulate_kron2d_sorted_f32": (_I, [_P, _P, _L, _P, _P, _L, _D, _L, _P, _L, _D, _L, _I, _P, _P]),
    "asvgp_blockband_to_blocks": (_I, [_P, _L, _L, _L, _P, _P, _P]),
    "asvgp_kron_grad_terms": (_I, [_P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _P, _P]),
    "asvgp_predict_kron2d_var": (_I, [_P, _L, _P, _L, _D, _P, _L, _D, _L, _I, _P, _P, _L, _P, _P]),
    "asvgp_kron_assemble_twisted": (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _D, _L, _L, _L, _L, _L, _P, _P, _P, _P]),
    "asvgp_kron_grad_terms_twisted": (_I, [_P, _P, _L, _L, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _P, _P]),
    "asvgp_predict_kron2d_var_twisted": (_I, [_P, _L, _P, _L, _D, _P, _L, _D, _L, _L, _I, _P, _P, _L, _L, _L, _L, _L, _P, _P]),
    "asvgp_phi_cross_workspace_bytes": (_Z, [_L, _L]),
    "asvgp_phi_cross_2d": (_I, [_P, _P, _L, _P, _L, _D, _L, _P, _L, _D, _L, _I, _P, _P, _Z, _P]),
}


class AsvgpError(RuntimeError):
    pass


_lib = None


def get_lib():
    """Load libasvgp_hip.so (built in-tree by asvgp_amd.build).  Raises if it is missing - there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsvgpError("HIP extension %s is missing: run `python -m asvgp_amd.build` (hipcc, gfx950). "
                         "asvgp_amd has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise AsvgpError("libasvgp_hip.so does not export %s" % name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        lib = get_lib()
        raise AsvgpError("%s failed: %s (%s)" % (what or "asvgp call", lib.asvgp_status_name(status).decode(),
                                                  lib.asvgp_last_error_string().decode()))


class Handle:
    """asvgp_create / asvgp_destroy: the library state of ONE model (algorithm choices, Phi workgroup count, chain events,
    timing ring, prior-chain plan).  Models own one each, so two models may step on two streams / host threads."""
    _defaults = {"band": 0, "phi": 0, "prior_forward": 0, "phi_order": 0}
    _live = None

    def __init__(self):
        import weakref
        lib = get_lib()
        h = _P()
        check(lib.asvgp_create(ctypes.byref(h)), "asvgp_create")
        self.ptr = h
        self._lib = lib
        self.band_algorithm = 0
        if Handle._live is None:
            Handle._live = weakref.WeakSet()
        Handle._live.add(self)
        if Handle._defaults["band"]:
            self.set_band_algorithm(Handle._defaults["band"])
        if Handle._defaults["phi"]:
            self.set_phi_algorithm(Handle._defaults["phi"])
        if Handle._defaults["prior_forward"]:
            self.set_prior_forward(Handle._defaults["prior_forward"])
        if Handle._defaults["phi_order"]:
            self.set_phi_input_order(Handle._defaults["phi_order"])

    def set_prior_forward(self, mode):
        """asvgp_set_prior_forward: 0 = the Kuu chain's forward pass on the host (x87 long double), 1 = on the GPU (double-double)."""
        check(self._lib.asvgp_set_prior_forward(self.ptr, int(mode)), "set_prior_forward")

    def prior_forward_device(self, coef, dcoef, n_doubles):
        """asvgp_prior_forward_device: the GPU forward pass alone for one theta; returns the factor table as a numpy array."""
        import numpy as np
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        dcoef = np.ascontiguousarray(dcoef, dtype=np.float64)
        tab = np.zeros(int(n_doubles), dtype=np.float64)
        check(self._lib.asvgp_prior_forward_device(self.ptr, coef.ctypes.data, dcoef.ctypes.data, tab.ctypes.data, tab.size, stream_ptr()),
              "prior_forward_device")
        return tab

    def set_band_algorithm(self, algo):
        check(self._lib.asvgp_set_band_algorithm(self.ptr, int(algo)), "set_band_algorithm")
        self.band_algorithm = int(algo)        # (what THIS handle runs: restored after a fallback step)

    def set_phi_algorithm(self, algo):
        check(self._lib.asvgp_set_phi_algorithm(self.ptr, int(algo)), "set_phi_algorithm")

    def phi_last_algorithm(self):
        return int(self._lib.asvgp_phi_last_algorithm(self.ptr))

    def set_phi_input_order(self, order):
        """asvgp_set_phi_input_order: 0 probe once per (x, N), 1 unsorted, 2 time series (selects the tile-sort instantiation)."""
        check(self._lib.asvgp_set_phi_input_order(self.ptr, int(order)), "set_phi_input_order")

    def phi_last_input_order(self):
        return int(self._lib.asvgp_phi_last_input_order(self.ptr))

    def result_mirror(self, enable=True):
        """asvgp_result_mirror: the handle's 16 pinned doubles [out[0..7], info[0], info[1], sequence] as a numpy view (None when off)."""
        import numpy as np
        ptr = _P()
        check(self._lib.asvgp_result_mirror(self.ptr, int(bool(enable)), ctypes.byref(ptr)), "result_mirror")
        if not enable or not ptr.value:
            self._mirror = None
            return None
        self._mirror = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(_D)), shape=(16,))
        return self._mirror

    def set_deferred_forward_pass(self, on):
        """0: the ELBO call runs the host forward pass itself; 1: the caller does (publish_forward()); 2: the handle's worker thread does,
        started before the launch call (the pass overlaps the launch path; one spinning host thread per handle)."""
        check(self._lib.asvgp_set_deferred_forward_pass(self.ptr, int(on)), "set_deferred_forward_pass")

    def publish_forward(self):
        """asvgp_prior_publish: run the host forward pass a deferred launch is waiting for (a no-op when none is pending)."""
        self._lib.asvgp_prior_publish(self.ptr)

    def result_mirror_pending(self):
        return int(self._lib.asvgp_result_mirror_pending(self.ptr))

    def set_phi_workgroups(self, n):
        check(self._lib.asvgp_set_phi_workgroups(self.ptr, int(n)), "set_phi_workgroups")

    deferred = False

    def set_phi_deferred_reduce(self, on):
        check(self._lib.asvgp_set_phi_deferred_reduce(self.ptr, int(on)), "set_phi_deferred_reduce")
        self.deferred = bool(on)

    def chain_sync(self, on):
        check(self._lib.asvgp_elbo_chain_sync(self.ptr, int(on)), "elbo_chain_sync")

    def prior_plan(self, static_stack_host, n_terms, M, k):
        """asvgp_prior_plan_1d from a host (numpy, C-contiguous fp64) copy of the static-band stack; returns True when planned."""
        ok = _I(0)
        ptr = None if static_stack_host is None else static_stack_host.ctypes.data
        check(self._lib.asvgp_prior_plan_1d(self.ptr, ptr, int(n_terms), int(M), int(k), ctypes.byref(ok)), "prior_plan_1d")
        return bool(ok.value)

    def close(self):
        """asvgp_destroy: waits for the handle's OWN last launch to have consumed its pinned table / written its mirror (no device-wide
        synchronisation), then frees.  Called by model.close() and, best effort, from __del__."""
        if getattr(self, "ptr", None) is not None and self.ptr:
            try:
                self._lib.asvgp_destroy(self.ptr)
            except Exception:
                pass
            self.ptr = None
            self._mirror = None

    def __del__(self):
        self.close()


def set_default_algorithms(band=None, phi=None, prior_forward=None, phi_order=None):
    """Algorithm choice for every live handle and for handles created later (asvgp_amd.set_band_algorithm / set_phi_algorithm /
    set_prior_forward)."""
    if prior_forward is not None:
        Handle._defaults["prior_forward"] = int(prior_forward)
        for h in list(Handle._live or ()):
            h.set_prior_forward(prior_forward)
    if phi_order is not None:
        Handle._defaults["phi_order"] = int(phi_order)
        for h in list(Handle._live or ()):
            h.set_phi_input_order(phi_order)
    if band is not None:
        Handle._defaults["band"] = int(band)
    if phi is not None:
        Handle._defaults["phi"] = int(phi)
    for h in list(Handle._live or ()):
        if band is not None:
            h.set_band_algorithm(band)
        if phi is not None:
            h.set_phi_algorithm(phi)


_stream_override = None


def set_stream(stream=None):
    """Launch every following library call on `stream` (a torch.cuda.Stream) instead of torch's current stream; None restores the
    default.  Saves the `with torch.cuda.stream(...)` enter / exit (several microseconds each) in latency-bound host loops."""
    global _stream_override
    _stream_override = None if stream is None else ctypes.c_void_p(stream.cuda_stream)


def stream_ptr():
    if _stream_override is not None:
        return _stream_override
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise AsvgpError("asvgp_amd operates on ROCm device tensors only (got a %s tensor); there is no CPU path"
                             % t.device)


def f64c(t):
    """contiguous fp64 view/copy"""
    if t.dtype != torch.float64:
        t = t.to(torch.float64)
    return t.contiguous()
