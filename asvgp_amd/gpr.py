"""GPR_1d: drop-in for asvgp/gpr.py:18-136 hosted on the MI355X HIP library.

Same surface: GPR_1d(data=(X[N,1], y[N,D]), kernel, basis) with .elbo(), .maximum_log_likelihood_objective(),
.training_loss(), .predict_f(Xnew, full_cov=False, full_output_cov=False, batch=False) and the attributes
X, y, bandwidth, Kuf_y, KufKfu, KufKfu_sparse, tr_yTy, kernel, likelihood.variance, inducing_features.
Additions (not in the reference): elbo_and_grad() (analytic gradient, replacing TF autodiff through the
banded_matrices op gradients), fit() (L-BFGS-B driver, replacing gpflow.optimizers.Scipy), process_group= for
N-sharded construction over RCCL.
"""
import math

import numpy as np
import torch

from . import kernels, utils
from . import kernels as kernels_mod
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
        self._elbo_ws = torch.zeros(lib.asvgp_elbo_workspace_bytes(M, k, D) // 8, dtype=torch.float64, device=dev)  # zero-init: finalize's arrival slots
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

    # -- the same computation split for scheduling (asvgp_elbo_prior_chain_1d / asvgp_elbo_data_chain_1d) ------------
    def launch_prior_chain(self):
        """Enqueue the theta-only half (Kuu, its l-tangent, band(Kuu^-1), log|Kuu|) on the CURRENT stream.  It does not
        read the statistics, so it may run on a side stream while the Phi pass / all-reduce are in flight."""
        v, l, s = self.theta()
        S = self.inducing_features.static_stack(self.kernel.kind)
        check(get_lib().asvgp_elbo_prior_chain_1d(S.data_ptr(), self.kernel.kind, v, l, s, self.basis.m, self.bandwidth,
                                                  self.D, self._info.data_ptr(), self._elbo_ws.data_ptr(),
                                                  self._elbo_ws.numel() * 8, stream_ptr()), "elbo_prior_chain_1d")

    def launch_data_chain(self):
        """Enqueue the data half (P chain + finalize) on the current stream; it must be ordered after launch_prior_chain
        for the same theta (same stream, or current_stream().wait_event(...))."""
        v, l, s = self.theta()
        S = self.inducing_features.static_stack(self.kernel.kind)
        check(get_lib().asvgp_elbo_data_chain_1d(self._stats.data_ptr(), S.data_ptr(), self.kernel.kind, v, l, s,
                                                 self.num_data, self.basis.m, self.bandwidth, self.D, self._out.data_ptr(),
                                                 self._info.data_ptr(), self._elbo_ws.data_ptr(),
                                                 self._elbo_ws.numel() * 8, stream_ptr()), "elbo_data_chain_1d")
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


class GPR_kron:
    """Drop-in for asvgp/gpr.py:239-359 (d = 2): GPR_kron((X[N,2], y[N,1]), kernels, bases) with elbo(),
    maximum_log_likelihood_objective(), training_loss(), predict_f(Xnew).  Never densifies: KufKfu is a block band
    (asvgp_phi_accumulate_kron2d), Kuu = K1 (x) K2 is handled factor-wise (log|Kuu| = m2 log|K1| + m1 log|K2|, the trace
    needs only band(K1^-1) (x) band(K2^-1)), and P = Kuu + KufKfu/sigma2 is a bandwidth k(m2+1) band matrix factorised by a
    blocked band Cholesky - the reference runs dense O(M_tot^3) tf.linalg.cholesky (gpr.py:293)."""

    def __init__(self, data, kernels, bases, process_group=None, distributed=None):
        dev = bases[0].device
        self.X, self.y = _to_device(data[0], dev), _to_device(data[1], dev)
        self.n, self.d = self.X.shape[0], self.X.shape[1]
        assert len(kernels) == len(bases) == self.d          # gpr.py:247
        assert self.y.shape[1] == 1                          # gpr.py:248
        if self.d != 2:
            raise NotImplementedError("asvgp_amd.GPR_kron implements the d = 2 tensor product (all reference configs)")
        for kern in kernels:
            assert isinstance(kern, (kernels_mod.Matern12, kernels_mod.Matern32, kernels_mod.Matern52))
        assert bases[0].order == bases[1].order
        for i, bs in enumerate(bases):
            if self.n:
                lo, hi = torch.aminmax(self.X[:, i])
                assert lo.item() >= bs.a and hi.item() <= bs.b
        self.kernels, self.bases = kernels, bases
        self.kernel = kernels[-1]                            # gpr.py:254 passes the leaked loop variable
        self.likelihood = kernels_mod.Gaussian()
        self.m, self.order = bases[0].m, bases[0].order      # gpr.py:260-261
        self.bandwidth = int((self.m ** self.d - 1) * self.order / (self.m - 1))   # gpr.py:262 (as written)
        self.true_bandwidth = self.order * (bases[1].m + 1)
        self.inducing_features = [SplineFeatures1D(kernels[i], bases[i]) for i in range(self.d)]
        lib = get_lib()
        m1, m2, k = bases[0].m, bases[1].m, self.order
        self.Mtot = m1 * m2
        self.noff = k * (2 * k + 1) + k + 1
        self._stats = torch.empty(lib.asvgp_kron_stats_doubles(m1, m2, k), dtype=torch.float64, device=dev)
        self.phi_pass()
        if distributed is None:
            distributed = process_group is not None
        self.num_data = allreduce_stats(self._stats, self.n, process_group) if distributed else self.n
        self.KufKfu_blockband = self._stats[:self.noff * self.Mtot].view(self.noff, self.Mtot)
        self.Kuf_y = self._stats[self.noff * self.Mtot:self.noff * self.Mtot + self.Mtot].view(self.Mtot, 1)
        self.tr_yTy = self._stats[-1]
        self._info = torch.zeros(1, dtype=torch.int32, device=dev)
        self._post = None

    def phi_pass(self):
        b1, b2 = self.bases
        check(get_lib().asvgp_phi_accumulate_kron2d(self.X.data_ptr(), self.y.data_ptr(), self.n, b1.mesh.data_ptr(),
                                                    b1.mesh.shape[0], b1.delta_np, b1.m, b2.mesh.data_ptr(),
                                                    b2.mesh.shape[0], b2.delta_np, b2.m, self.order,
                                                    self._stats.data_ptr(), stream_ptr()), "phi_accumulate_kron2d")
        return self._stats

    @property
    def Kuf(self):
        """gpr.py:269 kron.make_kvs_sparse(Kuf): sparse (m1*m2, N) Khatri-Rao design matrix."""
        from . import kronecker
        return kronecker.make_kvs_sparse(self.bases, self.X)

    def theta(self):
        return [(float(k.variance), float(k.lengthscales)) for k in self.kernels], float(self.likelihood.variance)

    def _factor(self, want_alpha):
        """Kuu factors per dimension, trace term, wide-band Cholesky of P with the rhs riding along."""
        from . import banded
        lib = get_lib()
        b1, b2 = self.bases
        m1, m2, k = b1.m, b2.m, self.order
        s = float(self.likelihood.variance)
        Ks = [f.make_Kuu(kern) for f, kern in zip(self.inducing_features, self.kernels)]
        Ls = [banded.cholesky_band(K) for K in Ks]
        Ss = [banded.inverse_from_cholesky_band(L) for L in Ls]
        logdet_K = m2 * torch.log(Ls[0][0] ** 2).sum() + m1 * torch.log(Ls[1][0] ** 2).sum()
        bw = k * m2 + k
        dev = self._stats.device
        Pb = torch.empty(self.Mtot * (bw + 1), dtype=torch.float64, device=dev)
        tr = torch.zeros(1, dtype=torch.float64, device=dev)
        check(lib.asvgp_kron_assemble(Ks[0].data_ptr(), Ks[1].data_ptr(), Ss[0].data_ptr(), Ss[1].data_ptr(),
                                      self.KufKfu_blockband.data_ptr(), k, m1, m2, s, Pb.data_ptr(), tr.data_ptr(),
                                      stream_ptr()), "kron_assemble")
        c = self.Kuf_y.reshape(-1).clone()
        logdet_P = torch.zeros(1, dtype=torch.float64, device=dev)
        check(lib.asvgp_blockband_cholesky(Pb.data_ptr(), self.Mtot, bw, c.data_ptr(), logdet_P.data_ptr(),
                                           self._info.data_ptr(), stream_ptr()), "blockband_cholesky")
        col = int(self._info.item())
        if col:
            raise NotPositiveDefiniteError("P = Kuu + KufKfu/sigma2 not positive definite at column %d" % (col - 1))
        alpha = None
        if want_alpha:
            alpha = c.clone()
            check(lib.asvgp_blockband_backsolve(Pb.data_ptr(), self.Mtot, bw, alpha.data_ptr(), stream_ptr()),
                  "blockband_backsolve")
            alpha = alpha / s
        return dict(Ks=Ks, Ls=Ls, Ss=Ss, logdet_K=logdet_K, logdet_P=logdet_P[0], trace=tr[0], c=c / s, Lb=Pb, bw=bw,
                    alpha=alpha, s=s)

    def elbo(self):
        """gpr.py:282-308."""
        f = self._factor(want_alpha=False)
        s = f["s"]
        N = float(self.num_data)
        vprod = 1.0
        for kern in self.kernels:
            vprod *= float(kern.variance)                    # gpr.py:284: prod of K_diag
        elbo = -0.5 * N * math.log(2 * math.pi * s)
        elbo = elbo - 0.5 * f["logdet_P"] + 0.5 * f["logdet_K"] - 0.5 * self.tr_yTy / s
        elbo = elbo + 0.5 * (f["c"] ** 2).sum() - 0.5 * N * vprod / s + 0.5 * f["trace"] / s
        return elbo

    def maximum_log_likelihood_objective(self):
        return self.elbo()

    def training_loss(self):
        return -self.elbo()

    @property
    def trainable_parameters(self):
        ps = []
        for kern in self.kernels:
            ps += [kern.variance, kern.lengthscales]
        return ps + [self.likelihood.variance]

    def fit(self, maxiter=200):
        """eNATL60.py:88-89 opt.minimize(model_kron.training_loss, ...): L-BFGS-B on the unconstrained parameters.  Interim:
        the gradient is a 2-point finite difference of elbo() (the analytic one needs the band-restricted inverse of P,
        DESIGN.md 4.4), i.e. 2 d + 2 bound evaluations per iteration."""
        from scipy.optimize import minimize
        params = self.trainable_parameters

        def fun(u):
            for p, ui in zip(params, u):
                p.unconstrained = float(ui)
            try:
                return -float(self.elbo().item())
            except NotPositiveDefiniteError:
                return np.inf

        u0 = np.array([p.unconstrained for p in params])
        res = minimize(fun, u0, jac="2-point", method="L-BFGS-B", options=dict(maxiter=maxiter, eps=1e-6))
        for p, ui in zip(params, res.x):
            p.unconstrained = float(ui)
        return res

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        """gpr.py:310-334: (mean, var) as numpy (n,1); var = prod v + |L_P^-1 phi*|^2 - phi*^T Kuu^-1 phi*."""
        assert not full_output_cov
        if full_cov:
            raise NotImplementedError
        mean, var = self.predict_f_device(Xnew)
        return mean.cpu().numpy(), var.cpu().numpy()

    predict_f_sparse = predict_f                              # gpr.py:336-359 computes the same moments with CHOLMOD

    def predict_f_device(self, Xnew, chunk=4096):
        lib = get_lib()
        key = self.theta()
        if self._post is None or self._post[0] != key:
            self._post = (key, self._factor(want_alpha=True))
        f = self._post[1]
        b1, b2 = self.bases
        X = _to_device(Xnew, self._stats.device)
        n = X.shape[0]
        mean = torch.empty(n, dtype=torch.float64, device=X.device)
        qk = torch.empty(n, dtype=torch.float64, device=X.device)
        check(lib.asvgp_predict_kron2d(X.data_ptr(), n, b1.mesh.data_ptr(), b1.mesh.shape[0], b1.delta_np, b1.m,
                                       b2.mesh.data_ptr(), b2.mesh.shape[0], b2.delta_np, b2.m, self.order,
                                       f["alpha"].data_ptr(), f["Ss"][0].data_ptr(), f["Ss"][1].data_ptr(), mean.data_ptr(),
                                       qk.data_ptr(), stream_ptr()), "predict_kron2d")
        vprod = 1.0
        for kern in self.kernels:
            vprod *= float(kern.variance)
        qp = self._quad_P(X, f, chunk)
        var = vprod + qp - qk
        return mean.reshape(-1, 1), var.reshape(-1, 1)

    def _quad_P(self, X, f, chunk):
        """phi*^T P^-1 phi* = |L_P^-1 phi*|^2 (gpr.py:320-330), by multi-right-hand-side forward substitution on the
        wide-band factor, a chunk of test points at a time (first version: blocked solves through torch on unpacked
        diagonal blocks; the planned replacement is the band-restricted selected inverse)."""
        from . import kronecker
        Lb, bw, M = f["Lb"].view(self.Mtot, f["bw"] + 1), f["bw"], self.Mtot
        out = torch.empty(X.shape[0], dtype=torch.float64, device=X.device)
        NB = 256
        for c0 in range(0, X.shape[0], chunk):
            Xc = X[c0:c0 + chunk]
            rows, cols, data = kronecker.make_kvs_coo(self.bases, Xc)
            Z = torch.zeros((M, Xc.shape[0]), dtype=torch.float64, device=X.device)
            Z.index_put_((rows, cols), data, accumulate=True)
            r_first = int(rows.min().item())
            for j0 in range((r_first // NB) * NB, M, NB):
                j1 = min(j0 + NB, M)
                nb = j1 - j0
                # unpack the nb x nb diagonal block and the (<= bw + nb) x nb panel below it from band storage
                hi = min(j1 + bw, M)
                r = torch.arange(j0, hi, device=X.device).reshape(-1, 1)
                cc = torch.arange(j0, j1, device=X.device).reshape(1, -1)
                d = r - cc
                ok = (d >= 0) & (d <= bw)
                blk = torch.where(ok, Lb[cc.expand_as(d), d.clamp(0, bw)], torch.zeros((), dtype=torch.float64, device=X.device))
                Z[j0:j1] = torch.linalg.solve_triangular(blk[:nb], Z[j0:j1], upper=False)
                if hi > j1:
                    Z[j1:hi] -= blk[nb:] @ Z[j0:j1]
            out[c0:c0 + chunk] = (Z * Z).sum(0)
        return out
