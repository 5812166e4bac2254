"""Output transforms of the acquisition objective.

The reference passes TensorFlow callables (``tf.identity``, ``tf.sigmoid``, ``tf.exp``:
TRANSFORMS, bore/plugins/hpbandster/base.py:18) and composes the minimisation form
as ``lambda u: transform(-u)`` (bore/mixins.py:20).  A Python callable cannot run
inside a HIP kernel, so the three transforms the reference names are records the kernels
know (``Transform``); ``negated()`` is the composition with ``-u``.

ANY other elementwise callable (the reference takes any TF callable, bore/mixins.py:16) is a
``CallableTransform``: it must take and return a torch tensor and be differentiable by
torch.autograd (``lambda u: torch.nn.functional.softplus(u)``).  The network's value and input
gradient still come from the HIP kernel; the callable and its derivative are applied to them
on the host (float32, like the reference's TF graph), which the host-driven restart modes
("lockstep", "sequential") and the host SVGD use -- the one-launch device modes need a named
transform and fall back with a warning.
"""
from __future__ import annotations

import numpy as np


class Transform:
    __slots__ = ("name", "negate")

    def __init__(self, name, negate=False):
        if name not in ("identity", "sigmoid", "exp"):
            raise ValueError(f"transform must be one of ('identity', 'sigmoid', 'exp'), got {name!r}")
        self.name = name
        self.negate = bool(negate)

    def negated(self):
        return Transform(self.name, not self.negate)

    def __call__(self, u):
        u = np.asarray(u)
        if self.negate:
            u = -u
        if self.name == "identity":
            return u
        if self.name == "sigmoid":
            return 1.0 / (1.0 + np.exp(-u))
        return np.exp(u)

    def __repr__(self):
        return f"Transform({self.name!r}, negate={self.negate})"


class CallableTransform:
    """``transform`` given as a torch-differentiable elementwise callable (see the module docstring)."""
    __slots__ = ("fn", "negate")
    name = None          # (no kernel-side name: the device-only modes check for this)

    def __init__(self, fn, negate=False):
        if not callable(fn):
            raise TypeError(f"transform {fn!r} is not callable")
        self.fn = fn
        self.negate = bool(negate)

    def negated(self):
        return CallableTransform(self.fn, not self.negate)

    def value_and_derivative(self, f):
        """(T(s f), d T(s f) / d f) for the float32 network outputs ``f``, s = -1 when negated."""
        import torch
        sign = -1.0 if self.negate else 1.0
        u = torch.tensor(sign * np.asarray(f, dtype=np.float32), dtype=torch.float32, requires_grad=True)
        what = ("a callable transform must map a torch tensor to a torch tensor of the same shape "
                "(elementwise, differentiable by torch); named transforms: 'identity', 'sigmoid', 'exp'")
        try:                       # (a numpy function such as np.tanh fails inside with torch's own message)
            t = self.fn(u)
        except Exception as err:
            raise TypeError(f"{what} -- {self.fn!r} raised {type(err).__name__}: {err}") from err
        if not isinstance(t, torch.Tensor) or t.shape != u.shape:
            raise TypeError(what)
        try:
            (dT,) = torch.autograd.grad(t.sum(), u)
        except RuntimeError as err:
            raise TypeError(f"{what} -- the result of {self.fn!r} does not depend differentiably on its "
                            f"input ({err})") from err
        return t.detach().numpy().astype(np.float32), sign * dT.numpy().astype(np.float64)

    def __call__(self, u):
        return self.value_and_derivative(np.asarray(u))[0]

    def __repr__(self):
        return f"CallableTransform({self.fn!r}, negate={self.negate})"


identity = Transform("identity")
sigmoid = Transform("sigmoid")
exp = Transform("exp")

TRANSFORMS = dict(identity=identity, sigmoid=sigmoid, exp=exp)


def _KNOWN_OBJECTS():
    """Functions that ARE one of the named transforms: numpy's / torch's exp and sigmoid, scipy's expit."""
    known = [np.exp]
    try:
        import torch
        known += [torch.exp, torch.sigmoid]
    except Exception:                # pragma: no cover
        pass
    try:
        from scipy.special import expit
        known.append(expit)
    except Exception:                # pragma: no cover
        pass
    return known


def resolve(t):
    if t is None:
        return identity
    if isinstance(t, (Transform, CallableTransform)):
        return t
    if isinstance(t, str):
        if t not in TRANSFORMS:
            raise ValueError(f"`transform` must be one of {tuple(TRANSFORMS)}")
        return TRANSFORMS[t]
    # Known objects only: a user function that merely happens to be CALLED exp / sigmoid / identity is
    # applied as the callable it is, not silently replaced by the kernel transform of that name.
    if t in _KNOWN_OBJECTS():
        return TRANSFORMS[t.__name__ if t.__name__ != "expit" else "sigmoid"]
    # The reference's own idiom: transform=tf.identity (bore/mixins.py:16, 94), and the plugin's table of
    # tf.identity / tf.sigmoid / tf.exp (bore/plugins/hpbandster/base.py:128-131).  Functions that COME FROM
    # tensorflow / jax / numpy under one of the three names are those functions (code ported from the reference
    # keeps working); a user's own function of that name stays the callable it is.
    # (by (package, name) PAIRS that are the elementwise function -- numpy.identity / jax.numpy.identity
    # build identity MATRICES and stay the callables they are)
    mod = (getattr(t, "__module__", None) or "").split(".")[0]
    name = getattr(t, "__name__", None)
    elementwise = {"tensorflow": ("identity", "sigmoid", "exp"), "keras": ("identity", "sigmoid", "exp"),
                   "numpy": ("exp",), "jax": ("sigmoid", "exp")}
    if name in TRANSFORMS and name in elementwise.get(mod, ()):
        return TRANSFORMS[name]
    if callable(t):                 # any other elementwise, torch-differentiable callable
        return CallableTransform(t)
    raise TypeError(f"transform {t!r}: pass 'identity', 'sigmoid', 'exp', a bore_amd.transforms "
                    "object or a torch-differentiable elementwise callable")
