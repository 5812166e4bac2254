"""Output transforms of the acquisition objective.

The reference passes TensorFlow callables (``tf.identity``, ``tf.sigmoid``, ``tf.exp``:
TRANSFORMS, bore/plugins/hpbandster/base.py:18) and composes the minimisation form
as ``lambda u: transform(-u)`` (bore/mixins.py:20).  A Python callable cannot run
inside a HIP kernel, so transforms are named records; ``negated()`` is the
composition with ``-u``.
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


identity = Transform("identity")
sigmoid = Transform("sigmoid")
exp = Transform("exp")

TRANSFORMS = dict(identity=identity, sigmoid=sigmoid, exp=exp)


def resolve(t):
    if t is None:
        return identity
    if isinstance(t, Transform):
        return t
    if isinstance(t, str):
        if t not in TRANSFORMS:
            raise ValueError(f"`transform` must be one of {tuple(TRANSFORMS)}")
        return TRANSFORMS[t]
    name = getattr(t, "__name__", None)
    if name in TRANSFORMS:          # e.g. np.exp, a function called sigmoid/identity
        return TRANSFORMS[name]
    raise TypeError(f"transform {t!r}: pass 'identity', 'sigmoid', 'exp' or a bore_amd.transforms "
                    "object; arbitrary callables cannot run inside the HIP kernel")
