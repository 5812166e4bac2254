"""The lock-step restart driver against the reference's sequential loop
(scipy.optimize.minimize per start, bore/mixins.py:57-60): identical results when fed
identical f/g values.  CPU: the f/g is the oracle evaluated row by row."""
import numpy as np
import pytest
from scipy.optimize import Bounds, minimize

from bore_amd.optimizers import lockstep
from oracle import bore_oracle as O

CASES = [(6, [32, 32, 1], ["relu", "relu", "sigmoid"], "identity"),
         (2, [16, 16, 1], ["relu", "relu", "sigmoid"], "identity"),
         (3, [32, 32, 32, 1], ["elu"] * 3 + ["linear"], "sigmoid"),
         (4, [8, 1], ["tanh", "linear"], "exp")]


def same(a, b):
    return (np.array_equal(a.x, b.x) and a.fun == b.fun and a.nit == b.nit and a.nfev == b.nfev
            and a.status == b.status and a.message == b.message and np.array_equal(a.jac, b.jac)
            and a.success == b.success)


@pytest.mark.skipif(not lockstep.available(), reason="scipy setulb signature differs")
@pytest.mark.parametrize("D,units,acts,tr", CASES)
def test_lockstep_equals_sequential_minimize(D, units, acts, tr):
    rs = np.random.RandomState(0)
    p = O.glorot_uniform_params(D, units, rs)
    for i in range(1, len(p), 2):
        p[i] = rs.normal(scale=.1, size=p[i].shape).astype(np.float32)
    fg1 = lambda x: O.value_and_input_grad(p, acts, x, tr)

    def fgb(X):
        out = [fg1(x) for x in X]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])

    X0 = rs.uniform(-0.2, 1.2, size=(12, D))          # some starts outside the box: clipped
    bounds = Bounds(np.zeros(D), np.ones(D))
    for opts in (dict(maxiter=1000, ftol=1e-9), dict(maxiter=3), dict(maxfun=5)):
        seq = [minimize(fg1, x0=x0, method="L-BFGS-B", jac=True, bounds=bounds, options=opts)
               for x0 in X0]
        lk = lockstep.minimize_lockstep(fgb, X0, bounds=bounds, **opts)
        assert all(same(a, b) for a, b in zip(seq, lk)), opts
    # list-of-tuples bounds and half-open boxes
    tb = [(0.0, None)] * D
    seq = [minimize(fg1, x0=x0, method="L-BFGS-B", jac=True, bounds=tb, options=dict(maxiter=5))
           for x0 in X0[:3]]
    lk = lockstep.minimize_lockstep(fgb, X0[:3], bounds=tb, maxiter=5)
    assert all(same(a, b) for a, b in zip(seq, lk))


def test_lockstep_rejects_unknown_options():
    with pytest.raises(TypeError):
        lockstep.minimize_lockstep(lambda X: (X[:, 0], X), np.zeros((1, 2)), foo=1)
