"""Stein variational gradient descent for batch acquisition: the API of bore/optimizers/svgd
(``SVGD``, ``RadialBasis``, ``DistortionConstant``, ``DistortionExpDecay``, ``rank``;
base.py:11-131, kernels.py:4-28) for the requests the DEVICE kernels do not take.

``BatchMaximizableMixin.argmax_batch`` (bore_amd/mixins.py) runs all SVGD iterations of all
particles in one launch (``bore_svgd_optimize``).  What it refuses -- more particles than fit a
compute unit's LDS, a callable transform, a user's own kernel object or callback -- comes here: the value
and input gradient of all particles are still ONE HIP launch per iteration (``func``), the
particle interaction is a short float64 numpy driver.  The step-for-step restatement of the
reference (bit-equal to its recorded trajectories) is the checker, ``oracle/svgd_oracle.py``;
this driver is held to it at 1e-10 (tests/test_svgd.py).
"""
from __future__ import annotations

import numpy as np
from sklearn.utils import check_random_state

from .utils import from_bounds


def rank(a):
    """Fraction of entries <= each entry (the "weak" empirical CDF the reference's doctest shows:
    [0.453, 0.859, 0.379, 0.379, 0.762] -> [0.6, 1.0, 0.4, 0.4, 0.8])."""
    a = np.asarray(a)
    assert a.ndim == 1, "only support 1d arrays!"
    return (a[None, :] <= a[:, None]).sum(axis=1) / a.size


class DistortionConstant:
    def __init__(self, c=1.):
        self.c = c

    def __call__(self, beta):
        return self.c


class DistortionExpDecay:
    def __init__(self, lambd=1.):
        self.lambd = lambd

    def __call__(self, beta):
        return beta ** (-self.lambd)


class RadialBasis:
    """k(x, x') = exp(-|x - x'|^2 / 2h^2); ``length_scale=None``: the median heuristic
    h^2 = median |x - x'|^2 / (2 log(n + 1)); h >= 1e-6."""

    def __init__(self, length_scale=1.0):
        self.length_scale = length_scale

    def value_and_grad(self, X):
        """(K [n, n], sum_j d k(x_j, x_i) / d x_j [n, d]) -- the repulsion term of the update."""
        delta = X[:, None, :] - X[None, :, :]
        dist2 = np.einsum("ijd,ijd->ij", delta, delta)
        h = self.length_scale
        if h is None:
            h = np.sqrt(np.median(dist2) / (2.0 * np.log(len(X) + 1)))
        h = max(float(h), 1e-6)
        K = np.exp(dist2 * (-0.5 / h ** 2))
        return K, np.einsum("ij,ijd->id", K, delta) / h ** 2


class SVGD:
    """n_iter updates x += step * phi / (eps + sqrt(hist)) with phi = (K (zeta * grad f) + tau * repulsion) / n
    and hist the exponential average of phi^2 (its first value phi^2 itself)."""

    def __init__(self, kernel=None, n_iter=1000, step_size=1e-3, alpha=.9, eps=1e-6, tau=1.,
                 distortion=None):
        self.kernel = kernel if kernel is not None else RadialBasis()
        self.distortion = distortion if distortion is not None else DistortionConstant()
        self.n_iter, self.step_size, self.alpha, self.eps, self.tau = n_iter, step_size, alpha, eps, tau

    def optimize_from_init(self, func, x_init, bounds=None, callback=None):
        """``func(X [n, d]) -> (f [n], grad [n, d])``; the particles after n_iter updates, clipped to
        ``bounds`` after each."""
        box = None if bounds is None else from_bounds(bounds)[0]
        x = np.array(x_init, dtype=np.float64)
        hist = None
        for _ in range(self.n_iter):
            K, repulsion = self.kernel.value_and_grad(x)
            f, grad = func(x)
            weight = np.asarray(self.distortion(rank(f)), dtype=np.float64).reshape(-1, 1) * np.ones((len(x), 1))
            phi = (K @ (weight * grad) + self.tau * repulsion) / len(x)
            hist = phi * phi if hist is None else self.alpha * hist + (1.0 - self.alpha) * (phi * phi)
            x = x + self.step_size * phi / (self.eps + np.sqrt(hist))
            if box is not None:
                x = np.clip(x, box[0], box[1])
            if callback is not None:
                callback(x)
        return x

    def optimize(self, func, batch_size, bounds=None, callback=None, random_state=None):
        """From ``batch_size`` uniform draws in the box (the draw bore/optimizers/svgd/base.py:121-129 makes)."""
        (low, high), dims = from_bounds(bounds)
        x_init = check_random_state(random_state).uniform(low=low, high=high, size=(batch_size, dims))
        return self.optimize_from_init(func, x_init, bounds, callback)
