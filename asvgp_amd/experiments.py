"""Experiment adapters: the metric and timing scaffolding of the reference's experiment scripts, on the HIP models.

  * MSE / NLL                       experiments/eNATL60/eNATL60.py:28-36
  * train_test_split                the sklearn call of experiments/large_regression/electricity.py:99
  * run_band_gpr_1d                 the "Band - GPR" arm of electricity.py:128-141 (fit, predict, NLPD, MSE, timings)
  * run_kron                        eNATL60.py:82-123 (t_precomp / t_opt / t_total, 10 000-row predict chunks, metrics row)
  * synthetic_ssh                   stand-in for the unavailable eNATL60 file

The datasets themselves are not shipped with the reference; every function takes arrays."""
import math
import time

import numpy as np
import torch


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def MSE(truth, prediction):
    """eNATL60.py:28-31."""
    se = (_np(truth) - _np(prediction)) ** 2
    return float(se.mean())


def NLL(truth, pred_mean, pred_var):
    """eNATL60.py:33-36: mean negative Gaussian log density with std = sqrt(pred_var)."""
    truth, mean, var = _np(truth), _np(pred_mean), _np(pred_var)
    return float((0.5 * np.log(2 * math.pi * var) + 0.5 * (truth - mean) ** 2 / var).mean())


def train_test_split(X, y, test_size=0.05, random_state=0):
    """Shuffled split with a fixed seed (electricity.py:99 uses sklearn's with random_state=i)."""
    X, y = _np(X), _np(y)
    n = X.shape[0]
    n_test = int(math.ceil(n * test_size)) if test_size < 1 else int(test_size)
    perm = np.random.RandomState(random_state).permutation(n)
    te, tr = perm[:n_test], perm[n_test:]
    return X[tr], X[te], y[tr], y[te]


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def run_band_gpr_1d(X, y, X_test, y_test, kernel, basis, maxiter=15000):
    """electricity.py:128-141: build, optimise, predict; returns the quantities the script appends per repetition."""
    from .gpr import GPR_1d
    _sync()
    ta = time.time()
    model = GPR_1d((X, y), kernel, basis)
    model.fit(maxiter=maxiter)
    _sync()
    tb = time.time()
    y_pred, _ = model.predict_f(X_test)
    _sync()
    tc = time.time()
    nlpd = -float(np.mean(_np(model.predict_log_density((X_test, y_test)))))
    return dict(model=model, nlpd=nlpd, mse=MSE(y_test, y_pred), opt_time=tb - ta, pred_time=tc - tb, total_time=tc - ta)


def run_kron(X_train, y_train, X_test, y_test, kernels, bases, maxiter=50, predict_chunk=10_000):
    """eNATL60.py:82-123: time the precompute (constructor) and the optimisation separately, predict in 10 000-row
    chunks, report MSE / NLL and the metrics row of the script (as a dict)."""
    from .gpr import GPR_kron
    _sync()
    t0 = time.time()
    model = GPR_kron((X_train, y_train), kernels, bases)
    _sync()
    t_precomp = time.time() - t0
    t1 = time.time()
    model.fit(maxiter=maxiter)
    _sync()
    t_opt = time.time() - t1
    t_total = time.time() - t0
    Xt = _np(X_test)
    n = Xt.shape[0]
    mean, var = np.zeros((n, 1)), np.zeros((n, 1))
    for lo in range(0, n, predict_chunk):           # eNATL60.py:96-102 (the script drops a ragged tail; this keeps it)
        m_, v_ = model.predict_f(Xt[lo:lo + predict_chunk])
        mean[lo:lo + predict_chunk], var[lo:lo + predict_chunk] = m_, v_
    return dict(num_train=_np(X_train).shape[0], num_test=n, spline_order=model.order, time_precomp=t_precomp,
                time_opt=t_opt, time_total=t_total, nll=NLL(y_test, mean, var), mse=MSE(y_test, mean), GP=model)


def synthetic_ssh(n, seed=1997):
    """Stand-in for the sea-surface-height field of eNATL60.py:40-58 (file not available): lon ~ U(-75,-30),
    lat ~ U(20,50), a smooth multi-scale field plus 0.02 noise."""
    rng = np.random.default_rng(seed)
    lon = rng.uniform(-75 + 1e-6, -30 - 1e-6, n)
    lat = rng.uniform(20 + 1e-6, 50 - 1e-6, n)
    ssh = (0.4 * np.sin(0.25 * lon) * np.cos(0.3 * lat) + 0.15 * np.sin(0.9 * lon + 0.5 * lat)
           + 0.02 * (lat - 35) + 0.02 * rng.normal(size=n))
    return np.stack([lon, lat], axis=1), ssh.reshape(-1, 1)
