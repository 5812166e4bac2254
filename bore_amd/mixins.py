"""bore.mixins on MI355X: ``MaximizableMixin`` (behaviour of bore/mixins.py:14-89).

Control flow kept from the reference -- uniform samples, screening ``predict``,
``argpartition`` for the ``num_starts`` best, one bound-constrained L-BFGS-B per start
on ``transform(-f(x))``, best acceptable result -- with the two TensorFlow hot spots
replaced by HIP kernels (screening forward; value + input-gradient) and the restarts
advanced together instead of one after another.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import OptimizeResult, minimize
from sklearn.utils import check_random_state

from .base import convert
from .optimizers import lockstep
from .optimizers.utils import from_bounds
from .transforms import identity, resolve

#: how ``maxima`` runs its restarts.  "lockstep": all restarts share one batched f/g
#: launch per round (results identical to "sequential").  "sequential": the reference's
#: loop (bore/mixins.py:57-60), also used for any method other than L-BFGS-B.
RESTART_MODES = ("lockstep", "sequential")

_DEFAULT_OPTIONS = dict(maxiter=1000, ftol=1e-9)   # bore/mixins.py:23


def _check_counts(num_starts, num_samples):
    # the reference guards its arguments with asserts (bore/mixins.py:35-43): same errors
    assert num_samples is not None, "`num_samples` must be specified!"
    assert num_samples > 0, "`num_samples` must be positive integer!"
    assert num_starts is not None, "`num_starts` must be specified!"
    assert num_starts >= 0, "`num_starts` must be nonnegative integer!"
    assert num_samples >= num_starts, \
        "number of random samples (`num_samples`) must be " \
        "greater than number of starting points (`num_starts`)"


class MaximizableMixin:

    restart_mode = "lockstep"

    def __init__(self, transform=identity, *args, **kwargs):
        # first positional argument is `transform`, as in the reference (bore/mixins.py:16)
        super(MaximizableMixin, self).__init__(*args, **kwargs)
        self.transform = resolve(transform)
        # scipy minimises, so the objective is transform(-f(x)) (bore/mixins.py:18-20)
        self._func_min = convert(self, transform=self.transform.negated())

    # -- pieces of maxima -----------------------------------------------------
    def _screen(self, bounds, num_samples, random_state):
        """Uniform candidates and their NEGATED raw classifier output (no transform:
        bore/mixins.py:45-52)."""
        (low, high), dim = from_bounds(bounds)
        X_init = random_state.uniform(low=low, high=high, size=(num_samples, dim))
        return X_init, -self.predict(X_init).squeeze(axis=-1)

    def _minimize_from(self, X0, bounds, method, options):
        if method == "L-BFGS-B" and self.restart_mode == "lockstep" and lockstep.available():
            return lockstep.minimize_lockstep(self._func_min, X0, bounds=bounds,
                                              **dict(options or {}))
        return [minimize(self._func_min, x0=x0, method=method, jac=True, bounds=bounds,
                         options=options) for x0 in X0]

    # -- reference surface ------------------------------------------------------
    def maxima(self, bounds, num_starts=5, num_samples=1024, method="L-BFGS-B",
               options=_DEFAULT_OPTIONS, print_fn=print, random_state=None):
        """All local optima found from the ``num_starts`` best of ``num_samples`` uniform
        samples, as a list of ``OptimizeResult`` (bore/mixins.py:22-72)."""
        random_state = check_random_state(random_state)   # mutated, shared with the caller
        _check_counts(num_starts, num_samples)
        X_init, f_init = self._screen(bounds, num_samples, random_state)

        if num_starts == 0:
            # legal: the best random sample, wrapped (bore/mixins.py:67-70)
            best = np.argmin(f_init, axis=None)
            return [OptimizeResult(x=X_init[best], fun=f_init[best], success=True)]

        order = np.argpartition(f_init, kth=num_starts - 1, axis=None)[:num_starts]
        results = self._minimize_from(X_init[order], bounds, method, options)
        for k, res in enumerate(results, start=1):
            print_fn(f"[Maximum {k:02d}: value={res.fun:.3f}] "
                     f"success: {res.success}, "
                     f"iterations: {res.nit:02d}, "
                     f"status: {res.status} ({res.message})")
        return results

    def argmax(self, bounds, filter_fn=lambda res: True, *args, **kwargs):
        """Lowest-``fun`` result that converged or hit maxiter (``status == 1`` is not a
        failure) and passes ``filter_fn``; ``None`` when nothing qualifies
        (bore/mixins.py:74-89; ties keep the earliest)."""
        chosen = None
        for res in self.maxima(bounds, *args, **kwargs):
            if not (res.success or res.status == 1):
                continue
            if not filter_fn(res):
                continue
            if chosen is None or res.fun < chosen.fun:
                chosen = res
        return chosen


class BatchMaximizableMixin(MaximizableMixin):
    """bore/mixins.py:92-116.  ``_func_max`` (the SVGD objective: value + input gradient of
    ``transform(f(x))``) runs on the same kernel; the SVGD driver itself is SURVEY.md §8
    row f-2 ("next") and is not part of this build yet."""

    def __init__(self, transform=identity, *args, **kwargs):
        super(BatchMaximizableMixin, self).__init__(transform, *args, **kwargs)
        self._func_max = convert(self, transform=self.transform)

    def argmax_batch(self, batch_size, bounds, length_scale=None, n_iter=1000,
                     step_size=1e-3, alpha=.9, eps=1e-6, tau=1.0, lambd=None,
                     random_state=None):
        raise NotImplementedError("argmax_batch (SVGD batch acquisition) is outside the hot "
                                  "path built so far (SURVEY.md §8 row f-2)")
