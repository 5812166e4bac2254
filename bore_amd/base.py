"""bore.base on MI355X: ``convert`` (bore/base.py:7-42).

The reference stacks four decorators around ``transform(model(x))``
(bore/decorators.py:24-79: unbatch, squeeze(-1), tf.function value_and_gradient,
numpy_io) to get ``x (D,) float64 -> [val (), grad (D,) float64]`` for
``scipy.optimize.minimize(jac=True)``.  Here the whole composite is ONE kernel,
``bore_mlp_value_and_input_grad``; because Dense acts on the last axis the same
callable also takes ``(R, D)`` and returns ``(R,)``, ``(R, D)`` -- rows are
independent, which is what the batched restarts use.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.stats import truncnorm

from . import ops
from .transforms import CallableTransform, resolve


def convert(model, transform=None):
    """Build ``fn(x) -> [val, grad]`` for ``model`` (a bore_amd Sequential).

    x: float64 array ``(D,)`` or ``(R, D)``.  val: float32 scalar / ``(R,)`` (the network
    dtype); grad: float64 with the shape of x (dtype of the watched input,
    bore/decorators.py:54-61).  Returns a 2-element list like ``numpy_io`` does
    (bore/decorators.py:75-77)."""
    tr = resolve(transform)

    def fn(x):
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        X = np.ascontiguousarray(np.atleast_2d(x))
        model._ensure_built(X)
        Xd = torch.from_numpy(X).to(model.theta.device).reshape(1, X.shape[0], X.shape[1])
        if isinstance(tr, CallableTransform):
            # f and d f / d x from the kernel; the callable and its derivative on the host (float32
            # value, float64 gradient, like numpy_io / value_and_gradient: bore/decorators.py:51-77)
            f, gf = ops.mlp_value_and_input_grad(model._desc, model.theta, Xd, "identity", False)
            val, dT = tr.value_and_derivative(f[0].cpu().numpy())
            grad = dT[:, None] * gf[0].cpu().numpy()
        else:
            val, grad = ops.mlp_value_and_input_grad(model._desc, model.theta, Xd, tr.name, tr.negate)
            val = val[0].cpu().numpy()
            grad = grad[0].cpu().numpy()
        if single:
            return [val[0], grad[0]]
        return [val, grad]

    fn.transform = tr
    return fn


def truncated_normal(loc, scale, lower, upper):
    """Frozen normal(loc, scale) truncated to [lower, upper] (bore/base.py:45-48: scipy's
    truncnorm takes its limits in standard-deviation units)."""
    a = (lower - loc) / scale
    b = (upper - loc) / scale
    return truncnorm(a=a, b=b, loc=loc, scale=scale)


def maybe_distort(loc, distortion=None, bounds=None, random_state=None, print_fn=print):
    """Candidate post-processing of the plugin (bore/base.py:51-64): with ``distortion`` set,
    replace the maximiser by one draw of a normal centred on it (scale = distortion) truncated
    to the box; otherwise return it untouched.  ``bounds`` is a ``scipy.optimize.Bounds``."""
    if distortion is None:
        return loc
    assert bounds is not None, "must specify bounds!"
    ret = truncated_normal(loc=loc, scale=distortion, lower=bounds.lb,
                           upper=bounds.ub).rvs(random_state=random_state)
    print_fn(f"Suggesting x={ret} (after applying distortion={distortion:.3E})")
    return ret
