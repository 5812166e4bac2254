"""SVGD batch acquisition against trajectories recorded from the
REFERENCE's own SVGD (tests/golden/ref_svgd.npz, made by importing bore.optimizers.svgd),
and against sklearn's RBF kernel as the reference's tests do (tests/test_optimizers.py:76-100)."""
import os

import numpy as np
import pytest
from sklearn.metrics.pairwise import rbf_kernel

from bore_amd.optimizers.svgd import (SVGD, DistortionConstant, DistortionExpDecay, RadialBasis,
                                      rank)
from conftest import GOLDEN
from oracle import bore_oracle as O


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "ref_svgd.npz"))


def test_rank_doctest_values():
    a = np.array([0.4532752, 0.858725, 0.3792093, 0.6631048, 0.7619765])
    np.testing.assert_array_equal(rank(a), [0.4, 1.0, 0.2, 0.6, 0.8])
    a = np.array([0.4532752, 0.858725, 0.3792093, 0.3792093, 0.7619765])
    np.testing.assert_array_equal(rank(a), [0.6, 1.0, 0.4, 0.4, 0.8])


@pytest.mark.parametrize("n_samples,n_features", [(1, 1), (4, 2), (16, 64)])
@pytest.mark.parametrize("length_scale", [1e-3, 0.5, 2.0])
def test_kernel_against_sklearn(n_samples, n_features, length_scale):
    X = np.random.RandomState(42).rand(n_samples, n_features)
    K, K_grad = RadialBasis(length_scale=length_scale).value_and_grad(X)
    assert K.shape == (n_samples, n_samples) and K_grad.shape == (n_samples, n_features)
    np.testing.assert_array_almost_equal(K, rbf_kernel(X, gamma=.5 / length_scale ** 2), decimal=12)


def test_trajectories_equal_the_reference_bit_for_bit(g):
    params = [g[f"p_{i}"] for i in range(6)]
    acts = ["tanh", "relu", "linear"]

    def func(X):   # T(f), T = sigmoid, through the oracle's T(-f) with the output layer negated
        q = [p.copy() for p in params]
        q[-2], q[-1] = -q[-2], -q[-1]
        return O.value_and_input_grad(q, acts, X, "sigmoid", dtype=np.float64)

    for k in range(4):
        n, ls, lambd, n_iter = g[f"cfg_{k}"]
        ls = None if ls < 0 else float(ls)
        dist = DistortionConstant() if lambd < 0 else DistortionExpDecay(lambd=float(lambd))
        K, Kg = RadialBasis(length_scale=ls).value_and_grad(g[f"x0_{k}"])
        assert np.array_equal(K, g[f"K_{k}"]) and np.array_equal(Kg, g[f"Kg_{k}"])
        assert np.array_equal(rank(func(g[f"x0_{k}"])[0]), g[f"rank_{k}"])
        svgd = SVGD(kernel=RadialBasis(length_scale=ls), n_iter=int(n_iter), step_size=1e-2,
                    distortion=dist)
        x = svgd.optimize_from_init(func, g[f"x0_{k}"], bounds=[(0.0, 1.0)] * 3)
        assert np.array_equal(x, g[f"x_{k}"]), k
        assert ((x >= 0) & (x <= 1)).all()


@pytest.mark.gpu
def test_argmax_batch_on_the_gpu(gpu):
    """BatchMaximizableSequential.argmax_batch (bore/mixins.py:100-116): particles stay in the
    box, are reproducible under a seed and climb the classifier output."""
    from scipy.optimize import Bounds
    from bore_amd.layers import Dense
    from bore_amd.models import BatchMaximizableSequential
    rs = np.random.RandomState(0)
    model = BatchMaximizableSequential("sigmoid", seed=2)
    model.add(Dense(16, activation="relu", input_dim=2))
    model.add(Dense(16, activation="relu"))
    model.add(Dense(1))
    model.compile(optimizer="adam", loss=__import__("bore_amd").BinaryCrossentropy(from_logits=True))
    X = rs.uniform(size=(64, 2))
    y = np.sum((X - 0.3) ** 2, 1)
    model.fit(X, y < np.quantile(y, 0.25), epochs=300, batch_size=64)
    bounds = Bounds(np.zeros(2), np.ones(2))
    x0 = np.random.RandomState(5).uniform(size=(8, 2))
    xa = model.argmax_batch(8, bounds, n_iter=60, step_size=1e-2, random_state=5)
    xb = model.argmax_batch(8, bounds, n_iter=60, step_size=1e-2, random_state=5)
    assert xa.shape == (8, 2) and np.array_equal(xa, xb)
    assert ((xa >= 0) & (xa <= 1)).all()
    # without the repulsion term the particles follow the (kernel-smoothed) gradient uphill
    xc = model.argmax_batch(8, bounds, n_iter=60, step_size=1e-2, tau=0.0, random_state=5)
    assert model._func_max(xc)[0].mean() > model._func_max(x0)[0].mean()
    # _func_max is transform(f), no negation
    v, grad = model._func_max(xa)
    assert v.shape == (8,) and grad.shape == (8, 2) and ((v > 0) & (v < 1)).all()
