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

#: how ``maxima`` runs its restarts.
#: "device" (default): every restart's L-BFGS-B runs inside ONE kernel (bore_lbfgsb_minimize;
#:     same algorithm, constants and stopping rules as SciPy's, fp64; on the trained classifiers
#:     of every BASELINE config its pick is SciPy's, tests/test_gpu_agreement.py).  A request
#:     the kernel does not take (more than 64 input dimensions, ...) falls back to "lockstep"
#:     with a warning.
#: "lockstep": SciPy's own L-BFGS-B state machines on the host, all restarts sharing one
#:     batched f/g launch per round (results bit-identical to "sequential"): the reference mode
#:     the device optimiser is checked against.
#: "sequential": the reference's loop (bore/mixins.py:57-60), one scipy.optimize.minimize
#:     per start; also used for any method other than L-BFGS-B.
RESTART_MODES = ("device", "lockstep", "sequential")

_LBFGSB_MESSAGES = None


def _message(task, msg):
    global _LBFGSB_MESSAGES
    if _LBFGSB_MESSAGES is None:
        try:
            from scipy.optimize._lbfgsb_py import status_messages, task_messages
        except Exception:  # pragma: no cover
            status_messages = {0: "START", 1: "NEW_X", 3: "FG", 4: "CONVERGENCE", 5: "STOP",
                               7: "ERROR", 8: "ABNORMAL"}
            task_messages = {0: "", 401: "NORM OF PROJECTED GRADIENT <= PGTOL",
                             402: "RELATIVE REDUCTION OF F <= FACTR*EPSMCH",
                             502: "TOTAL NO. OF F,G EVALUATIONS EXCEEDS LIMIT",
                             504: "TOTAL NO. OF ITERATIONS REACHED LIMIT"}
        _LBFGSB_MESSAGES = (status_messages, task_messages)
    st, tk = _LBFGSB_MESSAGES
    return st.get(task, str(task)) + ": " + tk.get(msg, str(msg))

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

    restart_mode = "device"
    #: "device" (default): the screening kernel picks the starts and the restarts follow it on the stream, the results
    #: handed out in the reference's (np.argpartition's) order -- same list as "host", one wait per call instead of
    #: two (_maxima_on_device).  "host": predict, argpartition on the host, then the restarts: the literal form.
    screen_mode = "device"

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

    def _minimize_on_device(self, X0, bounds, options):
        import torch
        from . import ops
        (low, high), dim = from_bounds(bounds)
        low = [-np.inf if v is None else v for v in low]
        high = [np.inf if v is None else v for v in high]
        unknown = set(options) - {"maxcor", "ftol", "gtol", "maxfun", "maxiter", "maxls"}
        if unknown:
            raise TypeError(f"unknown L-BFGS-B options: {sorted(unknown)}")
        self._ensure_built(X0)
        x0 = torch.from_numpy(np.ascontiguousarray(X0, dtype=np.float64)[None]).to(self.theta.device)
        tr = self._func_min.transform
        x, fun, jac, info = ops.lbfgsb_minimize(self._desc, self.theta, x0, low, high, tr.name,
                                                tr.negate, **options)
        x, fun, jac, info = (t[0] for t in ops.lbfgsb_results_to_host(x, fun, jac, info))
        return self._results(x, fun, jac, info, range(len(X0)))

    def _results(self, x, fun, jac, info, rows):
        return [OptimizeResult(x=x[r], fun=float(fun[r]), jac=jac[r], nit=int(info[r, 0]),
                               nfev=int(info[r, 1]), njev=int(info[r, 1]),
                               status=int(info[r, 2]), success=bool(info[r, 2] == 0),
                               message=_message(int(info[r, 3]), int(info[r, 4])))
                for r in rows]

    def _maxima_on_device(self, X_init, bounds, num_starts, options):
        """``maxima`` without the stop in the middle (round 5).  The reference predicts on the candidates, picks the
        starts on the host and runs the restarts (bore/mixins.py:45-60); done literally that is two waits for the
        device per call -- the predictions, then the restarts -- with the host's argpartition and an upload between
        them.  Here the screening kernel picks the starts on the device and the restarts follow it on the stream;
        the predictions come back meanwhile and the host computes THE REFERENCE'S order of the starts
        (np.argpartition's) from them while the restarts run, then hands the results out in that order.  The same
        list as the literal form, bit for bit (tested); if the device's set of starts ever differs from
        argpartition's -- a tie across the cut -- the literal form runs (returns None here)."""
        import torch
        from . import ops
        (low, high), dim = from_bounds(bounds)
        num_samples = len(X_init)
        lo = [-np.inf if v is None else v for v in low]
        hi = [np.inf if v is None else v for v in high]
        unknown = set(options) - {"maxcor", "ftol", "gtol", "maxfun", "maxiter", "maxls"}
        if unknown:
            raise TypeError(f"unknown L-BFGS-B options: {sorted(unknown)}")
        self._ensure_built(X_init)
        dev = self.theta.device
        st = getattr(self, "_screen_stage", None)
        if st is None or st[0].numel() < num_samples or st[1].numel() < num_starts or st[3].numel() < num_samples * dim:
            st = self._screen_stage = (torch.empty(num_samples, dtype=torch.float32).pin_memory(),
                                       torch.empty(num_starts, dtype=torch.int32).pin_memory(), torch.cuda.Event(),
                                       torch.empty(num_samples * dim, dtype=torch.float64).pin_memory(), torch.cuda.Event())
            st[4].record()
        # (the candidates go up through pinned memory: a pageable upload is stream-ordered on the HOST as well -- the
        # caller would stand here until the fit in front of it on the stream is done, and the screening, the copies and
        # the restarts below would be enqueued one by one onto an idle device instead of behind a running fit)
        st[4].synchronize()                        # (the last call's upload has left the buffer)
        xs = st[3][:num_samples * dim].view(num_samples, dim)
        xs.copy_(torch.from_numpy(np.ascontiguousarray(X_init, dtype=np.float64)))
        Xd = torch.empty((num_samples, dim), dtype=torch.float64, device=dev)
        Xd.copy_(xs, non_blocking=True)
        st[4].record()
        x0, idx, pred = ops.screen_topk(self._desc, self.theta, Xd, num_starts, want_pred=True)
        st[0][:num_samples].copy_(pred[0], non_blocking=True)
        st[1][:num_starts].copy_(idx[0], non_blocking=True)
        st[2].record()
        tr = self._func_min.transform
        x, fun, jac, info = ops.lbfgsb_minimize(self._desc, self.theta, x0, lo, hi, tr.name, tr.negate, **options)
        st[2].synchronize()                       # (fit + screening are done; the restarts are running)
        f_init = -st[0][:num_samples].numpy()
        picks = st[1][:num_starts].numpy()
        order = np.argpartition(f_init, kth=num_starts - 1, axis=None)[:num_starts]
        where = {int(r): k for k, r in enumerate(picks)}
        if len(where) != num_starts or any(int(r) not in where for r in order):
            return f_init, None                   # (a tie across the cut: the literal form decides)
        x, fun, jac, info = (t[0] for t in ops.lbfgsb_results_to_host(x, fun, jac, info))
        return f_init, self._results(x, fun, jac, info, [where[int(r)] for r in order])

    def _minimize_from(self, X0, bounds, method, options):
        assert self.restart_mode in RESTART_MODES, self.restart_mode
        if method == "L-BFGS-B" and self.restart_mode == "device":
            from ._lib import UnsupportedError
            try:
                if self._func_min.transform.name is None:
                    raise UnsupportedError("a callable transform runs on the host")
                return self._minimize_on_device(X0, bounds, dict(options or {}))
            except UnsupportedError as e:
                import warnings
                warnings.warn(f"restart_mode='device' cannot take this request ({e}); running "
                              "SciPy's L-BFGS-B on the host around the f/g kernel instead",
                              RuntimeWarning, stacklevel=3)
        if (method == "L-BFGS-B" and self.restart_mode in ("device", "lockstep")
                and lockstep.available()):
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
        (low, high), dim = from_bounds(bounds)
        X_init = random_state.uniform(low=low, high=high, size=(num_samples, dim))    # (ONE draw, bore/mixins.py:45-47)
        f_init = results = None
        if (self.screen_mode == "device" and self.restart_mode == "device" and method == "L-BFGS-B" and num_starts > 0
                and self._func_min.transform.name is not None):
            from ._lib import UnsupportedError
            try:
                f_init, results = self._maxima_on_device(X_init, bounds, num_starts, dict(options or {}))
            except UnsupportedError:
                f_init = None                          # (the literal form below says why, or takes another route)
        if f_init is None:
            f_init = -self.predict(X_init).squeeze(axis=-1)

        if num_starts == 0:
            # legal: the best random sample, wrapped (bore/mixins.py:67-70)
            best = np.argmin(f_init, axis=None)
            return [OptimizeResult(x=X_init[best], fun=f_init[best], success=True)]

        if results is None:
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
    """bore/mixins.py:92-116: batch acquisition by Stein variational gradient descent.

    ``svgd_mode`` chooses where the particle interaction runs:
      "device"  (default) all iterations in ONE launch (``bore_svgd_optimize``: particles, kernel
                matrix and history in LDS; beyond 64 particles the matrix entries are formed on the fly);
                float32 networks, and bfloat16 ones of the wide static shapes; what does not fit a
                compute unit's LDS (32 n D bytes of particle state beside the network), a callable
                transform or a user's kernel object goes to "host" with a warning;
      "host"    for what the device refuses: ``_func_max`` -- value + input gradient of
                ``transform(f(x))`` for all particles -- is one HIP launch per SVGD iteration;
                kernel matrix, repulsion and the Adagrad step are a short float64 numpy driver
                (bore_amd/optimizers/svgd.py).  The checker of both is oracle/svgd_oracle.py, the
                reference's SVGD step for step, bit-equal to trajectories recorded from the reference;
                the device kernel equals it to rounding (1e-9 after 200 iterations,
                tests/test_svgd.py) and is ~an order of magnitude faster than the host driver."""

    svgd_mode = "device"

    def __init__(self, transform=identity, *args, **kwargs):
        super(BatchMaximizableMixin, self).__init__(transform, *args, **kwargs)
        self._func_max = convert(self, transform=self.transform)

    def argmax_batch(self, batch_size, bounds, length_scale=None, n_iter=1000,
                     step_size=1e-3, alpha=.9, eps=1e-6, tau=1.0, lambd=None,
                     random_state=None):
        from .optimizers.svgd import SVGD, DistortionConstant, DistortionExpDecay, RadialBasis
        assert self.svgd_mode in ("device", "host"), self.svgd_mode
        x_init = None
        if self.svgd_mode == "device":
            import torch
            from . import ops
            from ._lib import UnsupportedError
            random_state = check_random_state(random_state)
            (low, high), dims = from_bounds(bounds)
            # (the same draw SVGD.optimize makes: bore/optimizers/svgd/base.py:121-129)
            x_init = random_state.uniform(low=low, high=high, size=(batch_size, dims))
            self._ensure_built(x_init)
            try:
                if self.transform.name is None:
                    raise UnsupportedError("a callable transform runs on the host")
                out = ops.svgd_optimize(
                    self._desc, self.theta,
                    torch.from_numpy(np.ascontiguousarray(x_init[None])).to(self.theta.device), low,
                    high, self.transform.name, length_scale=length_scale, n_iter=n_iter,
                    step_size=step_size, alpha=alpha, eps=eps, tau=tau, lambd=lambd)
                return out[0].cpu().numpy()
            except UnsupportedError as e:
                import warnings
                warnings.warn(f"svgd_mode='device' cannot take this request ({e}); running the "
                              "particle interaction on the host instead", RuntimeWarning,
                              stacklevel=2)
        distortion = DistortionConstant() if lambd is None else DistortionExpDecay(lambd=lambd)
        svgd = SVGD(kernel=RadialBasis(length_scale=length_scale), n_iter=n_iter,
                    step_size=step_size, alpha=alpha, eps=eps, tau=tau, distortion=distortion)
        if x_init is not None:      # (fallback: the particles already drawn above)
            return svgd.optimize_from_init(self._func_max, x_init, bounds=bounds)
        return svgd.optimize(self._func_max, batch_size, bounds=bounds, random_state=random_state)
