"""SVGD batch acquisition against trajectories recorded from the
REFERENCE's own SVGD (tests/golden/ref_svgd.npz, made by importing bore.optimizers.svgd),
and against sklearn's RBF kernel as the reference's tests do (tests/test_optimizers.py:76-100)."""
import os

import numpy as np
import pytest
from sklearn.metrics.pairwise import rbf_kernel

# The CHECKER: the reference's SVGD restated step for step (bit-equal to its recorded trajectories below);
# the device kernels and the product's short host driver (bore_amd/optimizers/svgd.py) are held to it.
from oracle.svgd_oracle import (SVGD, DistortionConstant, DistortionExpDecay, RadialBasis, rank)
from bore_amd.optimizers import svgd as product
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


def test_product_host_driver_tracks_the_checker(g):
    """bore_amd/optimizers/svgd.py -- the API of bore.optimizers.svgd for what the device kernels refuse --
    is its own short statement of the update (einsum distances, no step-for-step mirror of the reference):
    the same doctest ranks, kernel values against sklearn, and the recorded reference trajectories to 1e-10
    (another summation order; with a distortion, rank ties can flip a weight: 1e-6)."""
    a = np.array([0.4532752, 0.858725, 0.3792093, 0.3792093, 0.7619765])
    np.testing.assert_array_equal(product.rank(a), [0.6, 1.0, 0.4, 0.4, 0.8])
    X = np.random.RandomState(42).rand(16, 64)
    K, Kg = product.RadialBasis(length_scale=0.5).value_and_grad(X)
    np.testing.assert_array_almost_equal(K, rbf_kernel(X, gamma=.5 / 0.5 ** 2), decimal=12)
    Ko, Kgo = RadialBasis(length_scale=None).value_and_grad(X)
    Kp, Kgp = product.RadialBasis(length_scale=None).value_and_grad(X)
    np.testing.assert_allclose(Kp, Ko, rtol=1e-12, atol=0)
    np.testing.assert_allclose(Kgp, Kgo, rtol=1e-10, atol=1e-13)
    params = [g[f"p_{i}"] for i in range(6)]
    acts = ["tanh", "relu", "linear"]

    def func(X):
        q = [p.copy() for p in params]
        q[-2], q[-1] = -q[-2], -q[-1]
        return O.value_and_input_grad(q, acts, X, "sigmoid", dtype=np.float64)

    for k in range(4):
        n, ls, lambd, n_iter = g[f"cfg_{k}"]
        ls = None if ls < 0 else float(ls)
        dist = product.DistortionConstant() if lambd < 0 else product.DistortionExpDecay(lambd=float(lambd))
        x = product.SVGD(kernel=product.RadialBasis(length_scale=ls), n_iter=int(n_iter), step_size=1e-2,
                         distortion=dist).optimize_from_init(func, g[f"x0_{k}"], bounds=[(0.0, 1.0)] * 3)
        np.testing.assert_allclose(x, g[f"x_{k}"], rtol=0, atol=1e-10 if lambd < 0 else 1e-6)
    seen = []
    x = product.SVGD(n_iter=3).optimize(func, 5, bounds=[(0.0, 1.0)] * 3, callback=seen.append, random_state=1)
    assert x.shape == (5, 3) and len(seen) == 3 and np.array_equal(seen[-1], x)


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


@pytest.mark.gpu
@pytest.mark.parametrize("n,D,units,ls,lambd,tr", [
    (8, 2, [16, 16, 1], None, None, "sigmoid"),
    (5, 3, [32, 32, 1], 0.3, None, "identity"),
    (16, 6, [32, 32, 1], None, 0.5, "sigmoid"),
    (33, 2, [16, 16, 1], None, None, "exp"),
    (64, 6, [32, 32, 1], 0.7, None, "sigmoid"),
    (24, 16, [64, 64, 64, 1], None, None, "sigmoid"),
    (1, 2, [16, 16, 1], None, None, "identity")])
def test_device_svgd_tracks_the_host_statement(gpu, n, D, units, ls, lambd, tr):
    """bore_svgd_optimize (all iterations in one launch) against SVGD.optimize_from_init of
    oracle/svgd_oracle.py (bit-equal to the reference, test above) driven by the same
    device f/g operator: same particles to rounding -- sums run in another order and exp() is
    the device's -- after 200 iterations, with clipping active on some coordinates."""
    import torch
    from bore_amd import _lib, ops
    from test_gpu_parity import dev, pack, rand_model
    rs = np.random.RandomState(n + D)
    acts = ["tanh"] + ["relu"] * (len(units) - 2) + ["linear"]
    desc = _lib.make_desc(D, units, acts)
    L = 2
    th = dev(np.stack([pack(rand_model(rs, D, units)) for _ in range(L)]))
    x0 = rs.uniform(size=(L, n, D))
    kw = dict(n_iter=200, step_size=1e-2, alpha=.9, eps=1e-6, tau=1.)
    out = ops.svgd_optimize(desc, th, dev(x0), np.zeros(D), np.ones(D), tr, length_scale=ls,
                            lambd=lambd, **kw).cpu().numpy()
    assert ((out >= 0) & (out <= 1)).all()
    for l in range(L):
        def func(X):
            v, g = ops.mlp_value_and_input_grad(desc, th[l:l + 1], dev(X[None]), tr, False)
            return v.cpu().numpy()[0].astype(np.float64), g.cpu().numpy()[0]
        dist = DistortionConstant() if lambd is None else DistortionExpDecay(lambd=lambd)
        ref = SVGD(kernel=RadialBasis(length_scale=ls), distortion=dist, **kw).optimize_from_init(
            func, x0[l], bounds=[(0.0, 1.0)] * D)
        # rank ties/near-ties under DistortionExpDecay can flip a weight: looser there
        np.testing.assert_allclose(out[l], ref, rtol=0, atol=1e-9 if lambd is None else 1e-6)
    assert (np.abs(out - x0) > 1e-4).any()          # the particles moved
    # no box: same arithmetic without the clip
    free = ops.svgd_optimize(desc, th, dev(x0), None, None, tr, length_scale=ls, lambd=lambd,
                             n_iter=3, step_size=1e-2).cpu().numpy()
    assert np.isfinite(free).all()
    # what does not fit the LDS beside the network is refused, loudly
    big = _lib.make_desc(16, [64, 64, 64, 1], ["relu"] * 3 + ["linear"])
    thb = dev(pack(rand_model(rs, 16, [64, 64, 64, 1]))[None])
    with pytest.raises(RuntimeError, match="LDS"):
        ops.svgd_optimize(big, thb, dev(rs.uniform(size=(1, 64, 16))), n_iter=1)
    with pytest.raises(RuntimeError, match="particles"):
        ops.svgd_optimize(desc, th, dev(rs.uniform(size=(L, 4097, D))), n_iter=1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,D,ls,lambd", [(65, 2, None, None), (100, 6, None, 0.5), (130, 3, 0.2, None),
                                          (256, 2, None, None), (96, 11, None, None), (300, 2, None, None),
                                          (600, 3, 0.25, 0.5), (1024, 2, None, None)])
def test_device_svgd_with_more_than_64_particles(gpu, n, D, ls, lambd):
    """Beyond 64 particles the n x n kernel matrix does not fit in LDS: svgd_big_kernel forms its
    entries on the fly (one thread per particle, in turns beyond 256), the median heuristic's radix select
    recomputes the distances per pass, the network sees the particles 64 rows at a time.  Same
    sums in the same order as the small kernel: against the host statement (bit-equal to the
    reference's SVGD) driven by the same device f/g operator, to rounding."""
    import torch
    from bore_amd import _lib, ops
    from test_gpu_parity import dev, pack, rand_model
    rs = np.random.RandomState(n * 7 + D)
    units = [16, 16, 1]
    acts = ["tanh", "relu", "linear"]
    desc = _lib.make_desc(D, units, acts)
    th = dev(pack(rand_model(rs, D, units))[None])
    x0 = rs.uniform(size=(1, n, D))
    kw = dict(n_iter=60, step_size=1e-2, alpha=.9, eps=1e-6, tau=1.)
    out = ops.svgd_optimize(desc, th, dev(x0), np.zeros(D), np.ones(D), "sigmoid", length_scale=ls,
                            lambd=lambd, **kw).cpu().numpy()
    assert ((out >= 0) & (out <= 1)).all() and (np.abs(out - x0) > 1e-4).any()

    def func(X):
        v, g = ops.mlp_value_and_input_grad(desc, th, dev(X[None]), "sigmoid", False)
        return v.cpu().numpy()[0].astype(np.float64), g.cpu().numpy()[0]
    dist = DistortionConstant() if lambd is None else DistortionExpDecay(lambd=lambd)
    ref = SVGD(kernel=RadialBasis(length_scale=ls), distortion=dist, **kw).optimize_from_init(
        func, x0[0], bounds=[(0.0, 1.0)] * D)
    np.testing.assert_allclose(out[0], ref, rtol=0, atol=1e-9 if lambd is None else 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,D,units,ls,lambd", [(24, 16, [64, 64, 64, 1], None, None), (40, 32, [128, 128, 1], 0.8, None),
                                                (70, 16, [64, 64, 64, 1], None, None), (16, 32, [128, 128, 1], None, 0.5)])
def test_device_svgd_of_a_bfloat16_network(gpu, n, D, units, ls, lambd):
    """A mixed_bfloat16 model (wide static shapes, the only ones with bfloat16 kernels): the particles' value and
    input gradient run on the bf16 matrix cores inside the SVGD launch (arg_bf16_mfma.h) -- the arithmetic
    bore_mlp_value_and_input_grad gives the host statement for such a model, so the two track each other to
    rounding as for float32 networks (a bfloat16 network is a step function of x, but its steps sit at the
    bfloat16 roundings of x itself: both sides see the same inputs unless a particle lands within 1e-16 of one)."""
    import torch
    from bore_amd import _lib, ops
    from test_gpu_parity import dev, pack, rand_model
    rs = np.random.RandomState(3 * n + D)
    acts = ["relu"] * (len(units) - 1) + ["linear"]
    desc = _lib.make_desc(D, units, acts, compute="bfloat16")
    th = dev(pack(rand_model(rs, D, units))[None])
    x0 = rs.uniform(size=(1, n, D))
    kw = dict(n_iter=60, step_size=1e-2, alpha=.9, eps=1e-6, tau=1.)
    out = ops.svgd_optimize(desc, th, dev(x0), np.zeros(D), np.ones(D), "sigmoid", length_scale=ls,
                            lambd=lambd, **kw).cpu().numpy()
    assert ((out >= 0) & (out <= 1)).all() and (np.abs(out - x0) > 1e-4).any()

    def func(X):
        v, g = ops.mlp_value_and_input_grad(desc, th, dev(X[None]), "sigmoid", False)
        return v.cpu().numpy()[0].astype(np.float64), g.cpu().numpy()[0]
    dist = DistortionConstant() if lambd is None else DistortionExpDecay(lambd=lambd)
    ref = SVGD(kernel=RadialBasis(length_scale=ls), distortion=dist, **kw).optimize_from_init(
        func, x0[0], bounds=[(0.0, 1.0)] * D)
    np.testing.assert_allclose(out[0], ref, rtol=0, atol=1e-9 if lambd is None else 1e-6)
    # bfloat16 outside the wide static shapes has no kernels: refused, as everywhere in the library
    small = _lib.make_desc(2, [16, 16, 1], ["relu", "relu", "linear"], compute="bfloat16")
    with pytest.raises(RuntimeError, match="bfloat16"):
        ops.svgd_optimize(small, dev(pack(rand_model(rs, 2, [16, 16, 1]))[None]), dev(rs.uniform(size=(1, 8, 2))), n_iter=1)


@pytest.mark.gpu
def test_argmax_batch_device_mode(gpu):
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
    model.svgd_mode = "host"
    host = model.argmax_batch(8, bounds, n_iter=100, step_size=1e-2, random_state=5)
    model.svgd_mode = "device"
    devp = model.argmax_batch(8, bounds, n_iter=100, step_size=1e-2, random_state=5)
    np.testing.assert_allclose(devp, host, rtol=0, atol=1e-9)
    full = model.argmax_batch(8, bounds, random_state=5)          # the reference's 1000 iterations
    assert full.shape == (8, 2) and ((full >= 0) & (full <= 1)).all()


@pytest.mark.gpu
def test_argmax_batch_of_a_mixed_bfloat16_model_runs_on_the_device(gpu):
    """BatchMaximizableSequential with dtype_policy="mixed_bfloat16" (BASELINE config 5's network): the device mode
    takes the request (no fall-back warning) and agrees with the host statement around the same bfloat16 f/g kernel."""
    import warnings
    from scipy.optimize import Bounds
    from bore_amd.layers import Dense
    from bore_amd.models import BatchMaximizableSequential
    rs = np.random.RandomState(3)
    D = 32
    model = BatchMaximizableSequential("sigmoid", seed=2, dtype_policy="mixed_bfloat16")
    model.add(Dense(128, activation="relu", input_dim=D))
    model.add(Dense(128, activation="relu"))
    model.add(Dense(1))
    model.compile(optimizer="adam", loss=__import__("bore_amd").BinaryCrossentropy(from_logits=True))
    X = rs.uniform(size=(256, D))
    y = np.sum((X - 0.4) ** 2, 1)
    model.fit(X, y < np.quantile(y, 0.25), epochs=30, batch_size=64)
    bounds = Bounds(np.zeros(D), np.ones(D))
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # a fall-back to the host would warn
        devp = model.argmax_batch(12, bounds, n_iter=50, step_size=1e-2, random_state=5)
    model.svgd_mode = "host"
    host = model.argmax_batch(12, bounds, n_iter=50, step_size=1e-2, random_state=5)
    assert devp.shape == (12, D) and ((devp >= 0) & (devp <= 1)).all()
    np.testing.assert_allclose(devp, host, rtol=0, atol=1e-9)
