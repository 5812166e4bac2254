"""Bounds handling for the acquisition optimiser (behaviour of bore/optimizers/utils.py:4-16)."""
from scipy.optimize import Bounds


def from_bounds(bounds):
    """Normalise ``bounds`` to ``((low, high), dim)``.

    Accepts what the reference accepts: a ``scipy.optimize.Bounds`` (its ``lb``/``ub``
    arrays are returned as they are) or a sequence of ``(low, high)`` pairs (returned as
    two tuples)."""
    if isinstance(bounds, Bounds):
        lo, hi = bounds.lb, bounds.ub
        assert len(lo) == len(hi), "lower and upper bounds sizes do not match!"
        return (lo, hi), len(lo)
    pairs = list(bounds)
    lo = tuple(p[0] for p in pairs)
    hi = tuple(p[1] for p in pairs)
    return (lo, hi), len(pairs)
