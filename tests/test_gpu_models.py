"""The reference-facing API (bore_amd.models) on the GPU: the reference's own property
test (tests/test_models.py:12-50), the README loop (README.rst:54-103), parity of
predict / convert with the oracle, and lock-step == sequential restarts."""
import numpy as np
import pytest
from scipy.optimize import Bounds

from bore_amd.base import convert
from bore_amd.layers import BinaryCrossentropy, Dense, l2
from bore_amd.models import MaximizableDenseSequential, MaximizableSequential
from oracle import bore_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 42, 8888])
def test_maximizable_dense_sequential(gpu, seed):
    """Same statements as the reference's test, same seeds."""
    random_state = np.random.RandomState(seed)
    input_dim, output_dim, n_layers, n_units, n_starts, n_samples = 2, 1, 2, 32, 5, 1024
    bounds = Bounds(lb=np.zeros(input_dim), ub=np.ones(input_dim))
    model = MaximizableDenseSequential(input_dim=input_dim, output_dim=output_dim,
                                       num_layers=n_layers, num_units=n_units, seed=seed)
    X_test = random_state.uniform(low=bounds.lb, high=bounds.ub, size=(n_samples, input_dim))
    y_test = model.predict(X_test)
    assert y_test.shape == (n_samples, output_dim) and y_test.dtype == np.float32
    opt = model.argmax(bounds=bounds, num_starts=n_starts, num_samples=n_samples,
                       method="L-BFGS-B", options=dict(maxiter=1000, ftol=1e-9),
                       print_fn=lambda x: None, random_state=random_state)
    assert opt.x.shape == (input_dim,)
    X_opt = np.expand_dims(opt.x, axis=0)
    assert np.greater_equal(model.predict(X_opt), y_test).all()
    # analytic known answer for the activation-free net: the corner sign(W1 W2 W3 W4) picks
    w_eff = np.linalg.multi_dot([w.astype(np.float64) for w in model.get_weights()[0::2]])[:, 0]
    np.testing.assert_allclose(opt.x, (w_eff > 0).astype(float), atol=1e-6)
    # same weights through the oracle give the same predictions
    ref = O.predict(model.get_weights(), [l.activation for l in model.layers], X_test)
    np.testing.assert_allclose(y_test, ref, rtol=2e-5, atol=2e-6)


def branin01(X):
    x1, x2 = 15.0 * X[..., 0] - 5.0, 15.0 * X[..., 1]
    return ((x2 - 5.1 / (4 * np.pi ** 2) * x1 ** 2 + 5 / np.pi * x1 - 6) ** 2
            + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10)


def test_readme_loop_on_branin(gpu):
    """README.rst:54-103 (with the argument names the CODE accepts)."""
    rs = np.random.RandomState(0)
    classifier = MaximizableSequential(seed=0)
    classifier.add(Dense(16, activation="relu"))
    classifier.add(Dense(16, activation="relu"))
    classifier.add(Dense(1, activation="sigmoid"))
    classifier.compile(optimizer="adam", loss="binary_crossentropy")
    bounds = Bounds(lb=np.zeros(2), ub=np.ones(2))
    features = list(rs.uniform(size=(10, 2)))
    targets = list(branin01(np.array(features)))
    for i in range(12):
        X, y = np.vstack(features), np.hstack(targets)
        tau = np.quantile(y, q=0.25)
        z = np.less(y, tau)
        hist = classifier.fit(X, z, epochs=200, batch_size=64)
        assert len(hist.history["loss"]) == 200 and np.isfinite(hist.history["loss"]).all()
        res = classifier.argmax(bounds, num_starts=3, num_samples=1024, method="L-BFGS-B",
                                print_fn=lambda s: None, random_state=rs)
        x_next = res.x if res is not None else rs.uniform(size=2)
        assert x_next.shape == (2,) and (x_next >= 0).all() and (x_next <= 1).all()
        features.append(x_next)
        targets.append(branin01(x_next))
    m, v, t = classifier.get_optimizer_state()
    assert t == 12 * 200                     # Adam's counter ran across all fit calls
    assert classifier.evaluate(X, z) < 0.69  # better than chance
    assert min(targets[10:]) < np.median(targets[:10])


def test_plugin_style_model_fit_evaluate_argmax(gpu):
    """What ClassifierConfigGenerator builds (plugins/hpbandster/base.py:145-194,246-254):
    elu, l2 regularisers, from_logits loss + accuracy metric, sigmoid transform."""
    rs = np.random.RandomState(1)
    D = 4
    net = MaximizableDenseSequential(transform="sigmoid", input_dim=D, output_dim=1, num_layers=2,
                                     num_units=32, seed=3,
                                     layer_kws=dict(activation="elu", kernel_regularizer=l2(1e-4),
                                                    bias_regularizer=l2(1e-4)))
    net.compile(optimizer="adam", metrics=["accuracy"], loss=BinaryCrossentropy(from_logits=True))
    X = rs.uniform(size=(40, D))
    z = np.sum(X, axis=1) < 1.6
    w0 = net.get_weights() if net.built else None
    net.fit(X, z, epochs=1000 // 1, batch_size=64, verbose=False)
    loss, accuracy = net.evaluate(X, z, verbose=False)
    acts = [l.activation for l in net.layers]
    n = len(acts)
    ref_loss, ref_acc = O.evaluate(net.get_weights(), acts, X, z,
                                   l2=[1e-4] * (2 * n - 2) + [0.0, 0.0])
    assert loss == pytest.approx(ref_loss, rel=5e-5) and accuracy == pytest.approx(ref_acc)
    assert accuracy > 0.8
    seen = []
    opt = net.argmax(Bounds(np.zeros(D), np.ones(D)), num_starts=5, num_samples=1024,
                     method="L-BFGS-B", options=dict(maxiter=1000, ftol=1e-9),
                     print_fn=seen.append, filter_fn=lambda res: True, random_state=rs)
    assert len(seen) == 5 and all(s.startswith("[Maximum 0") for s in seen)
    assert opt is None or (0 <= opt.fun <= 1)          # sigmoid(-f) lies in (0, 1)
    assert net.argmax(Bounds(np.zeros(D), np.ones(D)), filter_fn=lambda r: False,
                      print_fn=lambda s: None, random_state=rs) is None
    del w0


@pytest.mark.parametrize("transform", ["identity", "sigmoid", "exp"])
def test_convert_matches_oracle_bridge(gpu, transform):
    rs = np.random.RandomState(2)
    D = 3
    model = MaximizableSequential(transform, seed=1)
    model.add(Dense(8, activation="tanh", input_dim=D))
    model.add(Dense(1))
    model.build()
    acts = ["tanh", "linear"]
    x = rs.uniform(size=D)
    out = model._func_min(x)
    assert isinstance(out, list) and len(out) == 2          # numpy_io returns a list
    val, grad = out
    assert val.dtype == np.float32 and val.shape == () and grad.dtype == np.float64 and grad.shape == (D,)
    rv, rg = O.value_and_input_grad(model.get_weights(), acts, x, transform)
    assert val == pytest.approx(rv, rel=2e-5) and np.allclose(grad, rg, rtol=2e-4, atol=2e-6)
    Xb = rs.uniform(size=(9, D))
    vb, gb = model._func_min(Xb)
    assert vb.shape == (9,) and gb.shape == (9, D)
    fmax = convert(model, transform)                      # T(f), no negation
    v2, g2 = fmax(x)
    h = O.predict(model.get_weights(), acts, x[None], dtype=np.float64)[0, 0]
    want = {"identity": h, "sigmoid": 1 / (1 + np.exp(-h)), "exp": np.exp(h)}[transform]
    assert v2 == pytest.approx(want, rel=2e-5)


def test_lockstep_restarts_equal_sequential_on_the_gpu(gpu):
    """Same HIP f/g values -> the batched restarts reproduce the reference's sequential
    scipy loop bit for bit (x, fun, nit, nfev, status)."""
    rs = np.random.RandomState(3)
    model = MaximizableSequential(seed=5)
    model.add(Dense(32, activation="relu", input_dim=6))
    model.add(Dense(32, activation="relu"))
    model.add(Dense(1, activation="sigmoid"))
    model.compile(optimizer="adam", loss="binary_crossentropy")
    X = rs.uniform(size=(60, 6))
    model.fit(X, np.sum(X, 1) < 2.7, epochs=50, batch_size=64)
    b = Bounds(np.zeros(6), np.ones(6))
    out = {}
    for mode in ("lockstep", "sequential"):
        model.restart_mode = mode
        out[mode] = model.maxima(b, num_starts=8, num_samples=256, print_fn=lambda s: None,
                                 random_state=np.random.RandomState(7))
    for a, c in zip(out["lockstep"], out["sequential"]):
        assert np.array_equal(a.x, c.x) and a.fun == c.fun and a.nit == c.nit
        assert a.nfev == c.nfev and a.status == c.status and a.message == c.message


def test_get_set_weights_roundtrip_keras_order(gpu):
    model = MaximizableSequential(seed=0)
    model.add(Dense(5, activation="relu", input_dim=3))
    model.add(Dense(1, activation="sigmoid"))
    model.build()
    w = model.get_weights()
    assert [a.shape for a in w] == [(3, 5), (5,), (5, 1), (1,)]
    assert all(a.dtype == np.float32 for a in w) and not w[1].any()      # zero biases
    lim = np.sqrt(6 / 8)
    assert np.abs(w[0]).max() <= lim                                      # glorot_uniform
    w2 = [a + 1 for a in w]
    model.set_weights(w2)
    assert all(np.array_equal(a, b) for a, b in zip(model.get_weights(), w2))
    with pytest.raises(ValueError):
        model.set_weights(w2[:2])


def test_mixed_bfloat16_policy_runs_config5(gpu):
    """BASELINE config 5 through the model API: 32-D, 128-128-1, bf16 compute, fp32 masters;
    fit + argmax with many restarts on the device."""
    from bore_amd.models import MaximizableSequential
    rs = np.random.RandomState(2)
    D = 32
    X = rs.uniform(size=(256, D))
    y = ((X - 0.35) ** 2).sum(axis=1)
    z = y < np.quantile(y, 0.25)
    model = MaximizableSequential(seed=4, dtype_policy="mixed_bfloat16", transform="sigmoid")
    model.add(Dense(128, activation="relu", input_dim=D))
    model.add(Dense(128, activation="relu"))
    model.add(Dense(1))
    model.compile(optimizer="adam", loss=BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    h = model.fit(X, z, epochs=40, batch_size=64)
    assert h.history["loss"][-1] < 0.5 * h.history["loss"][0]
    loss, acc = model.evaluate(X, z)
    assert acc > 0.9
    model.restart_mode = "device"
    res = model.argmax([(0.0, 1.0)] * D, num_starts=64, num_samples=1024, random_state=0)
    assert res is not None and res.x.shape == (D,)
    assert model.predict(res.x[None])[0, 0] >= np.quantile(model.predict(X), 0.99)
    with pytest.raises(ValueError):
        MaximizableSequential(dtype_policy="float16")


def test_fit_callbacks_and_large_batches(gpu):
    """Keras' callback protocol at epoch boundaries (incl. model.stop_training) and batch sizes
    above 64: same trajectory as the single-launch fit."""
    from bore_amd.layers import Dense
    from bore_amd.models import MaximizableSequential
    rs = np.random.RandomState(3)
    X = rs.uniform(size=(150, 2))
    z = X[:, 0] + X[:, 1] < 0.8

    def make():
        m = MaximizableSequential(seed=5)
        m.add(Dense(16, activation="relu"))
        m.add(Dense(16, activation="relu"))
        m.add(Dense(1, activation="sigmoid"))
        m.compile(optimizer="adam", loss="binary_crossentropy")
        return m

    class Log:
        def __init__(self, stop_at=None):
            self.events, self.stop_at = [], stop_at

        def set_model(self, model):
            self.model = model

        def on_train_begin(self, logs):
            self.events.append("begin")

        def on_epoch_end(self, epoch, logs):
            self.events.append((epoch, logs["loss"]))
            if self.stop_at is not None and epoch == self.stop_at:
                self.model.stop_training = True

        def on_train_end(self, logs):
            self.events.append("end")

    a, b, c = make(), make(), make()
    ha = a.fit(X, z, epochs=6, batch_size=100)
    log = Log()
    hb = b.fit(X, z, epochs=6, batch_size=100, callbacks=[log])
    np.testing.assert_allclose(ha.history["loss"], hb.history["loss"], rtol=1e-6)
    for u, v in zip(a.get_weights(), b.get_weights()):
        np.testing.assert_allclose(u, v, rtol=1e-5, atol=1e-7)
    assert log.events[0] == "begin" and log.events[-1] == "end" and len(log.events) == 8
    assert [e[0] for e in log.events[1:-1]] == list(range(6))
    stop = Log(stop_at=2)
    hc = c.fit(X, z, epochs=6, batch_size=100, callbacks=[stop])
    assert len(hc.history["loss"]) == 3
    np.testing.assert_allclose(hc.history["loss"], ha.history["loss"][:3], rtol=1e-6)


def test_callable_transform_drives_the_host_restart_modes(gpu):
    """`transform` as a callable (bore/mixins.py:16 takes any TF callable; here any
    torch-differentiable elementwise one): f and df/dx come from the HIP kernel, the callable and
    its derivative are applied on the host.  A callable that spells a named transform gives the
    named transform's objective; a custom one (softplus) passes a finite-difference check against
    the float64 oracle forward; argmax runs SciPy around the kernel (the device-only mode falls
    back with a warning) and returns a maximiser of the transformed classifier output."""
    import torch
    import warnings
    rs = np.random.RandomState(4)
    D = 3
    weights = None
    vals = {}
    for name, tr in [("named", "sigmoid"), ("callable", lambda u: torch.sigmoid(u)),
                     ("softplus", lambda u: torch.nn.functional.softplus(u))]:
        model = MaximizableSequential(tr, seed=1)
        model.add(Dense(8, activation="tanh", input_dim=D))
        model.add(Dense(1))
        model.build()
        if weights is None:
            weights = model.get_weights()
        model.set_weights(weights)
        Xb = np.random.RandomState(1).uniform(size=(7, D))
        v, g = model._func_min(Xb)
        assert v.dtype == np.float32 and v.shape == (7,) and g.dtype == np.float64 and g.shape == (7, D)
        vals[name] = (v, g, model)
    np.testing.assert_allclose(vals["callable"][0], vals["named"][0], rtol=1e-6)
    np.testing.assert_allclose(vals["callable"][1], vals["named"][1], rtol=1e-5, atol=1e-8)
    # softplus(-f): against the float64 oracle forward and central differences
    acts = ["tanh", "linear"]
    Xb = np.random.RandomState(1).uniform(size=(7, D))
    f64 = lambda X: np.log1p(np.exp(-O.predict(weights, acts, X, dtype=np.float64)[:, 0]))
    v, g, model = vals["softplus"]
    np.testing.assert_allclose(v, f64(Xb), rtol=2e-5)
    for d in range(D):
        e = np.zeros(D); e[d] = 1e-5
        np.testing.assert_allclose(g[:, d], (f64(Xb + e) - f64(Xb - e)) / 2e-5, rtol=2e-3, atol=1e-6)
    b = Bounds(np.zeros(D), np.ones(D))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = model.argmax(b, num_starts=4, num_samples=128, print_fn=lambda s: None,
                           random_state=np.random.RandomState(2))
    assert any("callable transform" in str(x.message) for x in w)        # device mode said why it stepped aside
    assert res is not None and np.all(res.x >= 0) and np.all(res.x <= 1)
    grid = np.random.RandomState(3).uniform(size=(2048, D))
    assert res.fun <= f64(grid).min() + 1e-4                              # a minimiser of softplus(-f) over the box
    model.restart_mode = "lockstep"
    res2 = model.argmax(b, num_starts=4, num_samples=128, print_fn=lambda s: None,
                        random_state=np.random.RandomState(2))
    assert np.array_equal(res.x, res2.x) and res.fun == res2.fun


@pytest.mark.parametrize("seed", [0, 3, 11])
@pytest.mark.parametrize("D,units,acts,tr,R", [(2, [16, 16, 1], ["relu", "relu", "sigmoid"], "identity", 3),
                                               (6, [32, 32, 32, 1], ["elu", "elu", "elu", "linear"], "sigmoid", 5),
                                               (5, [24, 7, 1], ["tanh", "relu", "sigmoid"], "exp", 8)])
def test_maxima_with_device_screening_hands_out_the_literal_forms_list(gpu, seed, D, units, acts, tr, R):
    """``maxima`` picks its starts with the screening kernel and lets the restarts follow on the stream
    (screen_mode "device", round 5: one wait per call instead of two); the results come back in the reference's
    order -- np.argpartition's over the predictions (bore/mixins.py:53-57) -- so the list is the literal form's
    (screen_mode "host": predict, argpartition on the host, restarts) bit for bit, and so is the caller's random
    state afterwards."""
    from scipy.optimize import Bounds
    from bore_amd.layers import Dense
    from bore_amd.models import MaximizableSequential
    bounds = Bounds(np.zeros(D), np.ones(D))
    outs = []
    for mode in ("host", "device"):
        model = MaximizableSequential(tr, seed=seed)
        for u, a in zip(units, acts):
            model.add(Dense(u, activation=a))
        model.compile(optimizer="adam", loss="binary_crossentropy" if acts[-1] == "sigmoid" else None)
        model.build(D)
        model.screen_mode = mode
        rs = np.random.RandomState(100 + seed)
        lines = []
        res = model.maxima(bounds, num_starts=R, num_samples=256, print_fn=lines.append, random_state=rs)
        outs.append((res, lines, rs.uniform()))
    (a, la, ua), (b, lb, ub) = outs
    assert ua == ub and la == lb and len(a) == len(b) == R
    for ra, rb in zip(a, b):
        assert np.array_equal(ra.x, rb.x) and ra.fun == rb.fun and np.array_equal(ra.jac, rb.jac)
        assert (ra.nit, ra.nfev, ra.status, ra.success, ra.message) == (rb.nit, rb.nfev, rb.status, rb.success, rb.message)


@pytest.mark.gpu
def test_build_then_smoke_in_one_interpreter(gpu):
    """The driver's two hooks back to back in ONE process: build() must leave the process able to compute (it once
    opened the library ahead of torch -- two HIP runtimes, no device in the second: tests/test_cabi.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert "smoke ok" in r.stdout
