"""R independent L-BFGS-B minimisations advanced in lock-step.

The reference runs its restarts one after another, each f/g request being one
single-point TensorFlow call (bore/mixins.py:57-60).  SciPy's L-BFGS-B core is a
reverse-communication routine (``_lbfgsb.setulb``): it returns whenever it needs
f and g at a point.  Here all R state machines are stepped until each either
finishes or asks for an evaluation, and all pending points go through ONE batched
``bore_mlp_value_and_input_grad`` launch per round.  Each state machine is the
same third-party code ``scipy.optimize.minimize(method="L-BFGS-B")`` drives, fed
the same numbers, so results are identical to the sequential loop (x, fun, nit,
nfev, status bit for bit; tests/test_lockstep.py).

``setulb`` is private SciPy API whose signature changed between the reference's
pinned 1.7 (Fortran) and 1.15 (C); ``available()`` gates on the signature and
callers fall back to sequential ``minimize`` when it does not match.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import OptimizeResult

try:  # private modules: guarded
    from scipy.optimize import _lbfgsb
    from scipy.optimize._lbfgsb_py import LbfgsInvHessProduct, status_messages, task_messages
    from scipy.optimize._constraints import old_bound_to_new
except Exception:  # pragma: no cover
    _lbfgsb = None

_SIG = "setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,lsave,isave,dsave,maxls,ln_task)"


def available():
    return _lbfgsb is not None and (_lbfgsb.setulb.__doc__ or "").strip().startswith(_SIG)


class _Problem:
    __slots__ = ("x", "f", "g", "wa", "iwa", "task", "ln_task", "lsave", "isave", "dsave",
                 "nit", "nfev", "last_x", "last_f", "last_g", "done")


def minimize_lockstep(fg_batch, X0, bounds=None, maxcor=10, ftol=2.2204460492503131e-09,
                      gtol=1e-5, maxfun=15000, maxiter=15000, maxls=20, with_index=False,
                      **unknown):
    """Minimise R problems sharing one objective.  ``fg_batch(X (k, D) f64) -> (val (k,),
    grad (k, D))``; with ``with_index`` it is called as ``fg_batch(X, idx)`` where ``idx`` are
    the problem numbers of the rows (callers that keep a fixed [R, D] device buffer).
    ``bounds``: anything ``minimize`` accepts.  Returns a list of R ``OptimizeResult`` with
    the fields ``_minimize_lbfgsb`` fills."""
    if unknown:
        raise TypeError(f"unknown L-BFGS-B options: {sorted(unknown)}")
    if not available():
        raise RuntimeError("scipy.optimize._lbfgsb.setulb has an unexpected signature")
    X0 = np.atleast_2d(np.asarray(X0, dtype=np.float64))
    R, n = X0.shape
    m = maxcor
    factr = ftol / np.finfo(float).eps
    nbd = np.zeros(n, np.int32)
    low = np.zeros(n, np.float64)
    upp = np.zeros(n, np.float64)
    if bounds is not None:
        from scipy.optimize import Bounds
        if isinstance(bounds, Bounds):
            lb, ub = np.broadcast_to(bounds.lb, n).astype(float), np.broadcast_to(bounds.ub, n).astype(float)
        else:
            if len(bounds) != n:
                raise ValueError("length of x0 != length of bounds")
            lb, ub = old_bound_to_new(bounds)
        if (lb > ub).any():
            raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
        X0 = np.clip(X0, lb, ub)
        for i in range(n):
            lo_inf, up_inf = np.isinf(lb[i]), np.isinf(ub[i])
            if not lo_inf:
                low[i] = lb[i]
            if not up_inf:
                upp[i] = ub[i]
            nbd[i] = {(True, True): 0, (False, True): 1, (False, False): 2, (True, False): 3}[
                (bool(lo_inf), bool(up_inf))]
    if not maxls > 0:
        raise ValueError("maxls must be positive.")

    probs = []
    for r in range(R):
        p = _Problem()
        p.x = np.array(X0[r], dtype=np.float64)
        p.f = np.array(0.0, dtype=np.float64)
        p.g = np.zeros(n, dtype=np.float64)
        p.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        p.iwa = np.zeros(3 * n, dtype=np.int32)
        p.task = np.zeros(2, dtype=np.int32)
        p.ln_task = np.zeros(2, dtype=np.int32)
        p.lsave = np.zeros(4, dtype=np.int32)
        p.isave = np.zeros(44, dtype=np.int32)
        p.dsave = np.zeros(29, dtype=np.float64)
        p.nit = 0
        p.nfev = 0
        p.last_x = None
        p.done = False
        probs.append(p)

    # ScalarFunction evaluates x0 when it is constructed (nfev = 1) and then serves the
    # first FG request at the same point from its cache.
    val, grad = fg_batch(X0, np.arange(R)) if with_index else fg_batch(X0)
    for r, p in enumerate(probs):
        p.last_x = p.x.copy()
        p.last_f = float(val[r])
        p.last_g = np.array(grad[r], dtype=np.float64)
        p.nfev = 1

    active = list(range(R))
    while active:
        pending = []
        for r in active:
            p = probs[r]
            while True:
                _lbfgsb.setulb(m, p.x, low, upp, nbd, p.f, p.g, factr, gtol, p.wa, p.iwa, p.task,
                               p.lsave, p.isave, p.dsave, maxls, p.ln_task)
                if p.task[0] == 3:
                    if np.array_equal(p.x, p.last_x):     # ScalarFunction's cache hit
                        p.f = np.array(p.last_f, dtype=np.float64)
                        p.g = p.last_g.copy()
                        continue
                    pending.append(r)
                    break
                elif p.task[0] == 1:
                    p.nit += 1
                    if p.nit >= maxiter:
                        p.task[0], p.task[1] = 5, 504
                    elif p.nfev > maxfun:
                        p.task[0], p.task[1] = 5, 502
                else:
                    p.done = True
                    break
        if pending:
            Xp = np.stack([probs[r].x for r in pending])
            val, grad = fg_batch(Xp, np.asarray(pending)) if with_index else fg_batch(Xp)
            for i, r in enumerate(pending):
                p = probs[r]
                p.nfev += 1
                p.last_x = p.x.copy()
                p.last_f = float(val[i])
                p.last_g = np.array(grad[i], dtype=np.float64)
                p.f = np.array(p.last_f, dtype=np.float64)
                p.g = p.last_g.copy()
        active = pending

    results = []
    for p in probs:
        if p.task[0] == 4:
            warnflag = 0
        elif p.nfev > maxfun or p.nit >= maxiter:
            warnflag = 1
        else:
            warnflag = 2
        s = p.wa[0:m * n].reshape(m, n)
        y = p.wa[m * n:2 * m * n].reshape(m, n)
        n_corrs = min(int(p.isave[30]), maxcor)
        msg = status_messages[p.task[0]] + ": " + task_messages[p.task[1]]
        results.append(OptimizeResult(fun=float(p.f), jac=p.g, nfev=p.nfev, njev=p.nfev,
                                      nit=p.nit, status=warnflag, message=msg, x=p.x,
                                      success=(warnflag == 0),
                                      hess_inv=LbfgsInvHessProduct(s[:n_corrs], y[:n_corrs])))
    return results
