"""Minimal stand-ins for the GPflow objects the reference's hot path touches: Matern12/32/52 kernels
with `.variance` / `.lengthscales`, a Gaussian likelihood with `.variance`, and softplus-constrained Parameters
(GPflow defaults: value 1.0, positive(lower=1e-6) on the likelihood variance)."""
import math


def softplus(u):
    return math.log1p(math.exp(-abs(u))) + max(u, 0.0)


def softplus_inv(x):
    return x + math.log(-math.expm1(-x))


def sigmoid(u):
    return 1.0 / (1.0 + math.exp(-u)) if u >= 0 else math.exp(u) / (1.0 + math.exp(u))


class Parameter:
    """positive parameter theta = softplus(u) + lower  (gpflow.Parameter with the default positive bijector)."""

    def __init__(self, value, lower=0.0, trainable=True):
        self.lower = float(lower)
        self.trainable = trainable
        self.assign(value)

    def assign(self, value):
        """Set the constrained value.  The value is kept EXACTLY (gpflow would return softplus(softplus_inv(value)), an ulp or
        two away): with cond(Kuu) ~ 1e7 the fp64 rounding pattern of the Kuu band alone moves the bound at the 1e-9 level, so
        parity against a CPU evaluation at theta needs the very same doubles."""
        value = float(value)
        assert value > self.lower
        self._u = softplus_inv(value - self.lower)
        self._value = value

    @property
    def unconstrained(self):
        return self._u

    @unconstrained.setter
    def unconstrained(self, u):
        self._u = float(u)
        self._value = softplus(self._u) + self.lower

    def numpy(self):
        return self._value

    __float__ = numpy

    def dtheta_du(self):
        return sigmoid(self.unconstrained)

    def __repr__(self):
        return "Parameter(%r)" % self.numpy()

    # arithmetic like a float so expressions such as `2 * kernel.variance` work (inducing_features.py:17)
    def __mul__(self, o): return float(self) * o
    __rmul__ = __mul__
    def __truediv__(self, o): return float(self) / o
    def __rtruediv__(self, o): return o / float(self)
    def __pow__(self, o): return float(self) ** o
    def __add__(self, o): return float(self) + o
    __radd__ = __add__


class _Matern:
    kind = None

    def __init__(self, variance=1.0, lengthscales=1.0):
        self.variance = Parameter(variance)
        self.lengthscales = Parameter(lengthscales)

    def K_diag(self, X):
        """kernel.K_diag(X) (gpr.py:52): the variance repeated N times; only its sum N*variance is used (gpr.py:86)."""
        import torch
        return torch.full((X.shape[0],), float(self.variance), dtype=torch.float64, device=getattr(X, "device", None))


class Matern12(_Matern):
    kind = 0


class Matern32(_Matern):
    kind = 1


class Matern52(_Matern):
    kind = 2


class Gaussian:
    """gpflow.likelihoods.Gaussian: variance default 1.0 with lower bound 1e-6."""

    def __init__(self, variance=1.0):
        self.variance = Parameter(variance, lower=1e-6)
