"""GPR_1d: drop-in for asvgp/gpr.py:18-136 hosted on the MI355X HIP library.

Same surface: GPR_1d(data=(X[N,1], y[N,D]), kernel, basis) with .elbo(), .maximum_log_likelihood_objective(),
.training_loss(), .predict_f(Xnew, full_cov=False, full_output_cov=False, batch=False) and the attributes
X, y, bandwidth, Kuf_y, KufKfu, KufKfu_sparse, tr_yTy, kernel, likelihood.variance, inducing_features.
Additions (not in the reference): elbo_and_grad() (analytic gradient, replacing TF autodiff through the
banded_matrices op gradients), fit() (L-BFGS-B driver, replacing gpflow.optimizers.Scipy), process_group= for
N-sharded construction over RCCL.
"""
import numpy as np
import torch

from . import kernels, utils
from ._lib import AsvgpError, check, f64c, get_lib, require_cuda, stream_ptr
from .banded import NotPositiveDefiniteError
from .dist import allreduce_stats
from .inducing_features import SplineFeatures1D


def _to_device(a, device):
    t = torch.as_tensor(a)
    return f64c(t.to(device))


class GPR_1d:
    def __init__(self, data, kernel, basis, process_group=None, distributed=None):
        # Check inputs (gpr.py:22-26)
        assert isinstance(kernel, (kernels.Matern12, kernels.Matern32, kernels.Matern52))
        assert data[0].shape[1] == 1
        dev = basis.device
        self.X, self.y = _to_device(data[0], dev), _to_device(data[1], dev)
        if self.y.dim() == 1:
            self.y = self.y.reshape(-1, 1)
        require_cuda(self.X, self.y)
        if self.X.shape[0] > 0:
            lo, hi = torch.aminmax(self.X)
            assert lo.item() > basis.a
            assert hi.item() < basis.b
        # Init model (gpr.py:29-34)
        self.kernel = kernel
        self.likelihood = kernels.Gaussian()
        self.basis = basis
        self.inducing_features = SplineFeatures1D(kernel, basis)
        self.bandwidth = self.basis.order  # gpr.py:37
        self.num_data_local = self.X.shape[0]
        self.D = self.y.shape[1]
        k, M, D = self.bandwidth, basis.m, self.D
        # Precompute static quantities (gpr.py:39-44): one fused Phi pass
        lib = get_lib()
        self._stats = torch.empty((k + 1) * M + M * D + 1, dtype=torch.float64, device=dev)
        wsb = lib.asvgp_phi_workspace_bytes(M, k, D)
        self._phi_ws = torch.empty(wsb // 8, dtype=torch.float64, device=dev)
        self._wsb = wsb
        self.phi_pass()
        if distributed is None:
            distributed = process_group is not None
        self.num_data = allreduce_stats(self._stats, self.num_data_local, process_group) if distributed \
            else self.num_data_local
        self.KufKfu = self._stats[:(k + 1) * M].view(k + 1, M)
        self.Kuf_y = self._stats[(k + 1) * M:(k + 1) * M + M * D].view(M, D)
        self.tr_yTy = self._stats[-1]
        self._elbo_ws = torch.empty(lib.asvgp_elbo_workspace_bytes(M, k, D) // 8, dtype=torch.float64, device=dev)
        self._out = torch.zeros(8, dtype=torch.float64, device=dev)
        self._info = torch.zeros(2, dtype=torch.int32, device=dev)
        self._post = None

    # ------------------------------------------------------------------------------------------------------
    def phi_pass(self):
        """(Re)run the fused N-dependent pass: asvgp_phi_accumulate_1d -> packed [band | Phi y | y^T y]."""
        b = self.basis
        check(get_lib().asvgp_phi_accumulate_1d(self.X.data_ptr(), self.y.data_ptr(), self.X.shape[0], self.D,
                                                b.mesh.data_ptr(), b.mesh.shape[0], b.delta_np, b.order, b.m,
                                                self._stats.data_ptr(), self._phi_ws.data_ptr(), self._wsb,
                                                stream_ptr()), "phi_accumulate_1d")
        return self._stats

    @property
    def KufKfu_sparse(self):
        """gpr.py:42: Kuf @ Kuf.T as a sparse M x M matrix (here: symmetric torch sparse COO rebuilt from the band)."""
        low = utils.band_to_sparse(self.KufKfu)
        strict = utils.band_to_sparse(torch.cat([torch.zeros_like(self.KufKfu[:1]), self.KufKfu[1:]], 0))
        return (low + strict.t()).coalesce()

    def theta(self):
        return float(self.kernel.variance), float(self.kernel.lengthscales), float(self.likelihood.variance)

    def _launch_elbo(self):
        v, l, s = self.theta()
        b = self.basis
        S = self.inducing_features.static_stack(self.kernel.kind)
        check(get_lib().asvgp_elbo_grad_1d(self._stats.data_ptr(), S.data_ptr(), self.kernel.kind, v, l, s,
                                           self.num_data, b.m, self.bandwidth, self.D, self._out.data_ptr(),
                                           self._info.data_ptr(), self._elbo_ws.data_ptr(),
                                           self._elbo_ws.numel() * 8, stream_ptr()), "elbo_grad_1d")
        return self._out

    def _check_pd(self):
        info = self._info.tolist()
        if info[0]:
            raise NotPositiveDefiniteError("Kuu band not positive definite at column %d" % (info[0] - 1))
        if info[1]:
            raise NotPositiveDefiniteError("P = Kuu + KufKfu/sigma2 not positive definite at column %d" % (info[1] - 1))

    def elbo(self, check_pd=True):
        """Variational bound on the log marginal likelihood (gpr.py:49-89); 0-d device tensor."""
        out = self._launch_elbo()
        if check_pd:
            self._check_pd()
        return out[0].clone()

    def maximum_log_likelihood_objective(self):
        return self.elbo()  # gpr.py:46-47 (reduce_sum of a scalar)

    def training_loss(self):
        """gpflow InternalDataTrainingLossMixin: -(objective + log_prior); no priors on this path."""
        return -self.elbo()

    def elbo_and_grad(self, check_pd=True):
        """(elbo, d elbo / d (variance, lengthscale, noise variance)) as a 4-vector device tensor [e, dv, dl, ds]."""
        out = self._launch_elbo()
        if check_pd:
            self._check_pd()
        return out[:4].clone()

    # -- optimiser (example.py:28-33: gpflow.optimizers.Scipy = scipy L-BFGS-B on unconstrained variables) ---
    @property
    def trainable_parameters(self):
        return [self.kernel.variance, self.kernel.lengthscales, self.likelihood.variance]

    def fit(self, maxiter=15000):
        from scipy.optimize import minimize
        params = self.trainable_parameters

        def fun(u):
            for p, ui in zip(params, u):
                p.unconstrained = float(ui)
            try:
                r = self.elbo_and_grad().tolist()
            except NotPositiveDefiniteError:
                return np.inf, np.zeros(3)
            g = np.array(r[1:4]) * np.array([p.dtheta_du() for p in params])
            return -r[0], -g

        u0 = np.array([p.unconstrained for p in params])
        res = minimize(fun, u0, jac=True, method="L-BFGS-B", options=dict(maxiter=maxiter))
        for p, ui in zip(params, res.x):
            p.unconstrained = float(ui)
        return res

    # -- posterior (gpr.py:91-136) -------------------------------------------------------------------------
    def _posterior(self):
        v, l, s = self.theta()
        key = (v, l, s)
        if self._post is not None and self._post[0] == key:
            return self._post[1], self._post[2]
        b = self.basis
        k, M, D = self.bandwidth, b.m, self.D
        alpha = torch.empty((M, D), dtype=torch.float64, device=self._stats.device)
        W = torch.empty((k + 1, M), dtype=torch.float64, device=self._stats.device)
        S = self.inducing_features.static_stack(self.kernel.kind)
        check(get_lib().asvgp_posterior_prepare_1d(self._stats.data_ptr(), S.data_ptr(), self.kernel.kind, v, l, s, M, k,
                                                   D, alpha.data_ptr(), W.data_ptr(), self._info.data_ptr(),
                                                   self._elbo_ws.data_ptr(), self._elbo_ws.numel() * 8, stream_ptr()),
              "posterior_prepare_1d")
        self._check_pd()
        self._post = (key, alpha, W)
        return alpha, W

    def predict_f_device(self, Xnew):
        """Posterior mean (n, D) and variance (n, 1) as device tensors - one streaming kernel (8 B in, 16 B out)."""
        alpha, W = self._posterior()
        b = self.basis
        x = _to_device(Xnew, self._stats.device).reshape(-1)
        n = x.shape[0]
        mean = torch.empty((n, self.D), dtype=torch.float64, device=x.device)
        var = torch.empty((n, 1), dtype=torch.float64, device=x.device)
        check(get_lib().asvgp_predict_1d(x.data_ptr(), n, b.mesh.data_ptr(), b.mesh.shape[0], b.delta_np, b.order, b.m,
                                         alpha.data_ptr(), W.data_ptr(), float(self.kernel.variance), self.D,
                                         mean.data_ptr(), var.data_ptr(), stream_ptr()), "predict_1d")
        return mean, var

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False, batch=False):
        """gpr.py:91-136.  Returns numpy (mean, var) like the reference.  batch=True reproduces the reference's
        10 000-row chunking including its dropped remainder (rows beyond the last full chunk stay zero, gpr.py:125-136)."""
        assert not full_output_cov
        if full_cov:
            raise NotImplementedError  # gpr.py:113
        if not batch:
            mean, var = self.predict_f_device(Xnew)
            return mean.cpu().numpy(), var.cpu().numpy()
        num_test = Xnew.shape[0]
        mean = np.zeros((num_test, 1))
        var = np.zeros((num_test, 1))
        nfull = int(num_test / 10_000) * 10_000
        if nfull:
            m_, v_ = self.predict_f_device(Xnew[:nfull])
            mean[:nfull] = m_.cpu().numpy()[:, :1]
            var[:nfull] = v_.cpu().numpy()
        return mean, var

    def predict_y(self, Xnew):
        mean, var = self.predict_f(Xnew)
        return mean, var + float(self.likelihood.variance)

    def predict_log_density(self, data):
        """gpflow GPModel.predict_log_density for the Gaussian likelihood (used by large_regression/electricity.py:138)."""
        Xnew, Ynew = data
        mean, var = self.predict_y(Xnew)
        Ynew = np.asarray(Ynew, dtype=np.float64).reshape(mean.shape)
        return -0.5 * (np.log(2 * np.pi * var) + (Ynew - mean) ** 2 / var)
