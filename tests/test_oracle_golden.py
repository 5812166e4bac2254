"""The oracle against (a) vectors produced by the REFERENCE's own importable modules,
(b) closed-form known answers, (c) an independent autodiff (torch, CPU), (d) its own
committed float64 regression vectors."""
import numpy as np
import pytest

from oracle import bore_oracle as O
from conftest import golden_params


def test_labels_match_reference_record(golden_labels, golden_misc):
    g = golden_labels
    for k, case in enumerate(golden_misc["label_cases"]):
        z, tau = O.labels(g[f"y{k}"], case["gamma"])
        assert z.dtype == np.bool_
        assert np.array_equal(z, g[f"z{k}"]), case
        assert np.array_equal(g[f"Xo{k}"], g[f"X{k}"])


def test_positive_class_size_rule():
    # ceil(gamma*(N-1)) positives for distinct y at gamma = 0.25
    rs = np.random.RandomState(3)
    for n, npos in [(9, 2), (10, 3), (64, 16), (65, 16), (110, 28)]:
        z, _ = O.labels(rs.normal(size=n), 0.25)
        assert z.sum() == npos


def test_steps_per_epoch_matches_reference(golden_misc):
    for n, b, s in golden_misc["steps_per_epoch"]:
        assert O.steps_per_epoch(n, b) == s
    # the reference's doctest table, bore/math.py:17-27
    assert [O.steps_per_epoch(n, 64) for n in (32, 64, 100, 1000)] == [1, 1, 2, 16]
    assert O.epochs_from_steps(1000, 100, 64) == 500


def test_from_bounds_matches_reference(golden_misc):
    from scipy.optimize import Bounds
    for c in golden_misc["from_bounds"]:
        (lo, hi), dim = O.from_bounds([tuple(b) for b in c["bounds"]])
        assert list(lo) == c["low"] and list(hi) == c["high"] and dim == c["dim"]
        (lo, hi), dim = O.from_bounds(Bounds(np.array(c["low"]), np.array(c["high"])))
        assert list(lo) == c["low_b"] and list(hi) == c["high_b"] and dim == c["dim_b"]


def test_dense_sequential_off_by_one():
    d, units, acts = O.dense_sequential_layout(2, 1, num_layers=2, num_units=32)
    assert (d, units) == (2, [32, 32, 32, 1])      # bore/models.py:16-19 fall-through
    assert O.dense_sequential_layout(5, 1, 0, 8)[1] == [1]


def test_adam_first_step_closed_form():
    # from zero state, delta = -1e-3 * g / (|g| + 1e-7/sqrt(1e-3))
    for g, want in [(0.3, -9.99989e-4), (-2e-5, 8.63473e-4), (1e-9, -3.16128e-7)]:
        p = [np.zeros(1, np.float64)]
        st = O.AdamState(p)
        O.adam_step(p, [np.array([g])], st)
        assert st.t == 1
        assert p[0][0] == pytest.approx(want, rel=2e-5)
        assert p[0][0] == pytest.approx(-1e-3 * g / (abs(g) + 1e-7 / np.sqrt(1e-3)), rel=1e-6)
    # torch-style Adam (eps inside the bias-corrected denominator) would give -1e-3*g/(|g|+1e-8)
    assert abs(-1e-3 * 1e-9 / (1e-9 + 1e-8) - (-3.16128e-7)) > 1e-5 * 3e-7


def test_adam_state_persists_across_fits():
    rs = np.random.RandomState(0)
    p = O.glorot_uniform_params(2, [4, 1], rs, np.float64)
    q = [a.copy() for a in p]
    X, z = rs.uniform(size=(10, 2)), rs.uniform(size=10) < 0.4
    perms = np.stack([rs.permutation(10) for _ in range(4)])
    acts = ["relu", "sigmoid"]
    s1, s2 = O.AdamState(p), O.AdamState(q)
    O.fit(p, acts, s1, X, z, perms, dtype=np.float64)
    O.fit(q, acts, s2, X, z, perms[:2], dtype=np.float64)
    O.fit(q, acts, s2, X, z, perms[2:], dtype=np.float64)
    assert s1.t == s2.t == 4
    for a, b in zip(p, q):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["branin", "hartmann", "plugin", "hpo16"])
def test_float64_regression_vectors(golden_mlp, name):
    meta, g = golden_mlp
    c = meta[name]
    n = len(c["units"])
    p = golden_params(g, name, "p0", n)
    l2 = None if c["l2"] is None else [c["l2"]] * (2 * n - 2) + [0.0, 0.0]
    idx = g[f"{name}_perms"][0][:c["batch"]]
    loss, grads = O.loss_and_grads(p, c["acts"], g[f"{name}_X"][idx],
                                   g[f"{name}_z"][idx].astype(np.float64), l2=l2)
    assert loss == pytest.approx(float(g[f"{name}_loss0"]), rel=1e-12)
    for i, gr in enumerate(grads):
        np.testing.assert_allclose(gr, g[f"{name}_g0_{i}"], rtol=1e-10, atol=1e-14)
    st = O.AdamState(p)
    hist = O.fit(p, c["acts"], st, g[f"{name}_X"], g[f"{name}_z"], g[f"{name}_perms"],
                 batch_size=c["batch"], l2=l2, dtype=np.float64)
    np.testing.assert_allclose(hist, g[f"{name}_hist"], rtol=1e-10)
    for i in range(2 * n):
        np.testing.assert_allclose(p[i], g[f"{name}_p1_{i}"], rtol=1e-9, atol=1e-13)
    val, grad = O.value_and_input_grad(p, c["acts"], g[f"{name}_Xq"], c["transform"], np.float64)
    np.testing.assert_allclose(val, g[f"{name}_val"], rtol=1e-9)
    np.testing.assert_allclose(grad, g[f"{name}_grad"], rtol=1e-8, atol=1e-13)
    # float32 mode of the oracle stays within float32 distance of the float64 vectors
    p32 = [a.astype(np.float32) for a in golden_params(g, name, "p1", n)]
    v32, g32 = O.value_and_input_grad(p32, c["acts"], g[f"{name}_Xq"], c["transform"])
    assert v32.dtype == np.float32 and g32.dtype == np.float64
    np.testing.assert_allclose(v32, g[f"{name}_val"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(g32, g[f"{name}_grad"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("acts,transform", [(["relu", "relu", "sigmoid"], "identity"),
                                            (["elu", "tanh", "linear"], "sigmoid"),
                                            (["sigmoid", "elu", "linear"], "exp")])
def test_gradients_against_torch_autograd(acts, transform):
    """Independent autodiff of the same formulas (torch CPU, float64)."""
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(1)
    D, units = 3, [8, 5, 1]
    p = O.glorot_uniform_params(D, units, rs, np.float64)
    for i in range(1, 6, 2):
        p[i] = rs.normal(scale=0.2, size=p[i].shape)
    X, z = rs.normal(size=(11, D)), (rs.uniform(size=11) < 0.5).astype(np.float64)
    tp = [torch.tensor(a, requires_grad=True) for a in p]
    tx = torch.tensor(X, requires_grad=True)
    f = {"relu": torch.relu, "elu": F.elu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
         "linear": lambda a: a}

    def net(h, logits):
        for l in range(3):
            a = h @ tp[2 * l] + tp[2 * l + 1]
            h = a if (l == 2 and logits) else f[acts[l]](a)
        return h

    if acts[-1] in ("sigmoid", "linear"):
        loss_t = F.binary_cross_entropy_with_logits(net(tx, True)[:, 0], torch.tensor(z))
        loss_t.backward()
        loss, grads = O.loss_and_grads(p, acts, X, z)
        assert loss == pytest.approx(loss_t.item(), rel=1e-12)
        for a, b in zip(grads, tp):
            np.testing.assert_allclose(a, b.grad.numpy(), rtol=1e-9, atol=1e-13)
        tx.grad = None
    u = -net(tx, False)[:, 0]
    T = {"identity": u, "sigmoid": torch.sigmoid(u), "exp": torch.exp(u)}[transform]
    T.sum().backward()
    val, grad = O.value_and_input_grad(p, acts, X, transform, np.float64)
    np.testing.assert_allclose(val, T.detach().numpy(), rtol=1e-12)
    np.testing.assert_allclose(grad, tx.grad.numpy(), rtol=1e-9, atol=1e-13)


def test_keras_adam_against_adjusted_torch_adam():
    """Keras Adam == torch Adam with eps_torch = eps_keras / sqrt(1 - beta2^t) per step
    (the two differ only in where eps sits); checks the recurrence over several steps."""
    rs = np.random.RandomState(2)
    p = [rs.normal(size=(4, 3))]
    st = O.AdamState(p)
    w = p[0].copy()
    m = np.zeros_like(w)
    v = np.zeros_like(w)
    for t in range(1, 8):
        g = rs.normal(size=w.shape) * 10.0 ** rs.randint(-6, 1)
        O.adam_step(p, [g], st)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        mhat, vhat = m / (1 - 0.9 ** t), v / (1 - 0.999 ** t)
        w = w - 1e-3 * mhat / (np.sqrt(vhat) + 1e-7 / np.sqrt(1 - 0.999 ** t))
        np.testing.assert_allclose(p[0], w, rtol=1e-9, atol=1e-15)


def test_partial_last_batch_takes_a_step_with_its_own_mean():
    rs = np.random.RandomState(5)
    p = O.glorot_uniform_params(2, [3, 1], rs, np.float64)
    X, z = rs.uniform(size=(65, 2)), rs.uniform(size=65) < 0.3
    st = O.AdamState(p)
    hist = O.fit(p, ["relu", "sigmoid"], st, X, z, np.arange(65)[None], batch_size=64,
                 dtype=np.float64)
    assert st.t == 2 and hist.shape == (1,)


def test_linear_network_corner_known_answer():
    """tests/test_models.py:12-50 builds an activation-free net; its maximiser over a box
    is the corner picked by the sign of W1.W2.W3.W4."""
    from scipy.optimize import Bounds
    for seed in (0, 42, 8888):
        rs = np.random.RandomState(seed)
        D, units, acts = O.dense_sequential_layout(2, 1, 2, 32)
        p = O.glorot_uniform_params(D, units, rs)
        w_eff = np.linalg.multi_dot([a.astype(np.float64) for a in p[0::2]])[:, 0]
        bounds = Bounds(lb=np.zeros(2), ub=np.ones(2))
        X_test = rs.uniform(size=(1024, 2))
        y_test = O.predict(p, acts, X_test)
        assert y_test.shape == (1024, 1)
        opt = O.argmax(p, acts, bounds, num_starts=5, num_samples=1024, random_state=rs)
        assert opt is not None and opt.x.shape == (2,)
        np.testing.assert_allclose(opt.x, (w_eff > 0).astype(float), atol=1e-6)
        # the reference's own assertion (tests/test_models.py:50)
        assert np.greater_equal(O.predict(p, acts, opt.x[None]), y_test).all()


def test_argmax_filter_and_zero_starts():
    from scipy.optimize import Bounds
    rs = np.random.RandomState(0)
    p = O.glorot_uniform_params(2, [16, 16, 1], rs)
    acts = ["relu", "relu", "sigmoid"]
    b = Bounds(np.zeros(2), np.ones(2))
    assert O.argmax(p, acts, b, filter_fn=lambda r: False, num_starts=2, random_state=0) is None
    res = O.maxima(p, acts, b, num_starts=0, num_samples=64, random_state=0)
    assert len(res) == 1 and res[0].success and res[0].x.shape == (2,)
    with pytest.raises(AssertionError):
        O.maxima(p, acts, b, num_starts=5, num_samples=4)


def test_public_tf_keras_docstring_vectors_third_party_known_answers():
    """THIRD-PARTY known answers, not reference-held fixtures: the worked examples printed in the
    public tf.keras API documentation (TF 2.x docstrings of ``tf.keras.losses.BinaryCrossentropy``
    and ``tf.keras.optimizers.Adam``).  The reference holds no vector for the Keras half of the path
    (parity unpinned) and TensorFlow cannot run here; these narrow what "unpinned" can
    hide: the BCE-from-logits formula, the probability form it must agree with, and the Adam update
    (bias correction, epsilon placement) that bore/plugins/hpbandster/base.py:156-157,184 and
    README.rst:60-66,93 rely on."""
    # BinaryCrossentropy(from_logits=True)([0, 1, 0, 0], [-18.6, 0.51, 2.94, -12.8]) -> 0.865
    a = np.array([[-18.6], [0.51], [2.94], [-12.8]], dtype=np.float32)
    z = np.array([[0.0], [1.0], [0.0], [0.0]], dtype=np.float32)
    assert float(O.bce_with_logits(a, z).mean()) == pytest.approx(0.865, abs=5e-4)
    # the same through the model path: Dense(1) logit output, loss_and_grads' mean over the batch
    W, b = np.ones((1, 1), np.float32), np.zeros(1, np.float32)
    loss, grads = O.loss_and_grads([W, b], ["linear"], a, z[:, 0])
    assert float(loss) == pytest.approx(0.865, abs=5e-4)
    loss_s, grads_s = O.loss_and_grads([W, b], ["sigmoid"], a, z[:, 0])   # sigmoid output: cached logits
    assert float(loss_s) == float(loss) and np.array_equal(grads_s[0], grads[0])
    ev, _ = O.evaluate([W, b], ["sigmoid"], a, z[:, 0])
    assert ev == pytest.approx(0.865, abs=5e-4)
    # BinaryCrossentropy()([[0, 1], [0, 0]], [[0.6, 0.4], [0.4, 0.6]]) -> 0.815 (probabilities)
    p = np.array([0.6, 0.4, 0.4, 0.6], dtype=np.float64)
    zt = np.array([0.0, 1.0, 0.0, 0.0])
    logit = np.log(p / (1 - p)).reshape(-1, 1)
    assert float(O.bce_with_logits(logit, zt.reshape(-1, 1)).mean()) == pytest.approx(0.815, abs=5e-4)
    assert float(-(zt * np.log(p) + (1 - zt) * np.log(1 - p)).mean()) == pytest.approx(0.815, abs=5e-4)
    # Adam(learning_rate=0.1): var1 = 10.0, loss = var1 ** 2 / 2 (gradient = var1); one step -> 9.9
    for dt in (np.float32, np.float64):
        var = [np.array([10.0], dtype=dt)]
        st = O.AdamState(var)
        O.adam_step(var, [var[0].copy()], st, lr=0.1)
        assert st.t == 1 and float(var[0][0]) == pytest.approx(9.9, abs=1e-6)
        # (a second step moves by almost the full learning rate again: m / sqrt(v) stays ~ 1)
        O.adam_step(var, [var[0].copy()], st, lr=0.1)
        assert float(var[0][0]) == pytest.approx(9.8, abs=1e-4)
