"""Host-side mirror of the reference interface: everything that runs without a GPU."""
import numpy as np
import pytest

import bore_amd
from bore_amd import shuffle, transforms
from bore_amd.data import Record, classification_labels
from bore_amd.layers import Adam, BinaryCrossentropy, Dense, l2, resolve_loss, resolve_optimizer
from bore_amd.math import ceil_divide, epochs_per_iteration, steps_per_epoch
from bore_amd.models import (DenseSequential, MaximizableDenseSequential, MaximizableSequential,
                             Sequential)
from bore_amd.optimizers.utils import from_bounds


def test_record_matches_reference_vectors(golden_labels, golden_misc):
    g = golden_labels
    for k, case in enumerate(golden_misc["label_cases"]):
        rec = Record()
        for xi, yi in zip(g[f"X{k}"], g[f"y{k}"]):
            rec.append(x=xi, y=yi)
        assert rec.size() == case["n"]
        X, z = rec.load_classification_data(case["gamma"])
        assert X.dtype == np.float64 and z.dtype == np.bool_
        assert np.array_equal(X, g[f"Xo{k}"]) and np.array_equal(z, g[f"z{k}"])
        assert np.array_equal(classification_labels(g[f"y{k}"], case["gamma"]), g[f"z{k}"])
        dup = np.array([rec.is_duplicate(p) for p in g[f"probe{k}"]])
        assert np.array_equal(dup, g[f"dup{k}"])


def test_math_matches_reference_vectors(golden_misc):
    for n, b, s in golden_misc["steps_per_epoch"]:
        assert steps_per_epoch(n, b) == s
    assert ceil_divide(7, 2) == 4 and ceil_divide(8, 2) == 4
    assert epochs_per_iteration(1000, 10, 64) == 1000 and epochs_per_iteration(1000, 100, 64) == 500


def test_from_bounds_matches_reference_vectors(golden_misc):
    from scipy.optimize import Bounds
    for c in golden_misc["from_bounds"]:
        (lo, hi), dim = from_bounds([tuple(b) for b in c["bounds"]])
        assert list(lo) == c["low"] and list(hi) == c["high"] and dim == c["dim"]
        assert isinstance(lo, tuple)
        (lo, hi), dim = from_bounds(Bounds(np.array(c["low"]), np.array(c["high"])))
        assert list(lo) == c["low_b"] and dim == c["dim_b"]


def test_transforms_registry():
    assert set(bore_amd.TRANSFORMS) == {"identity", "sigmoid", "exp"}
    t = transforms.resolve("sigmoid")
    assert t.name == "sigmoid" and not t.negate and t.negated().negate
    assert transforms.resolve(None) is transforms.identity
    assert transforms.resolve(np.exp).name == "exp"
    np.testing.assert_allclose(transforms.exp.negated()(1.0), np.exp(-1.0))
    with pytest.raises(ValueError):
        transforms.resolve("softplus")
    doubled = transforms.resolve(lambda u: u * 2)          # any other callable: applied on the host (round 3)
    assert isinstance(doubled, transforms.CallableTransform) and doubled.name is None
    with pytest.raises(TypeError):
        transforms.resolve(3)


def test_dense_descriptor_and_compile_arguments():
    d = Dense(16, activation="relu", input_dim=2, kernel_regularizer=l2(1e-3),
              bias_regularizer=l2(1e-4))
    assert (d.units, d.activation, d.input_dim) == (16, "relu", 2)
    assert d.l2_kernel == pytest.approx(1e-3) and d.l2_bias == pytest.approx(1e-4)
    assert Dense(3).activation == "linear"
    with pytest.raises(ValueError):
        Dense(4, activation="gelu")
    assert resolve_optimizer("adam") == Adam()
    assert Adam().epsilon == 1e-7                     # Keras, not torch (1e-8)
    assert resolve_loss("binary_crossentropy").from_logits is False
    assert resolve_loss(BinaryCrossentropy(from_logits=True)).from_logits is True
    with pytest.raises(NotImplementedError):
        resolve_optimizer("sgd")
    with pytest.raises(NotImplementedError):
        resolve_loss("mse")


def test_dense_sequential_reproduces_the_reference_off_by_one():
    m = DenseSequential(input_dim=2, output_dim=1, num_layers=2, num_units=32)
    assert [l.units for l in m.layers] == [32, 32, 32, 1]       # bore/models.py:16-19
    assert m.count_params() == 2 * 32 + 32 + 2 * (32 * 32 + 32) + 33
    m = MaximizableDenseSequential(transform="exp", input_dim=3, output_dim=1, num_layers=1,
                                   num_units=8, layer_kws=dict(activation="elu"))
    assert [l.units for l in m.layers] == [8, 8, 1] and m.transform.name == "exp"
    assert m.layers[0].activation == "elu" and m.layers[-1].activation == "linear"
    assert m._func_min.transform.negate and m._func_min.transform.name == "exp"


def test_first_positional_argument_is_transform():
    # bore/mixins.py:16: MaximizableMixin.__init__(self, transform=..., *args, **kwargs)
    m = MaximizableSequential("sigmoid")
    assert m.transform.name == "sigmoid"
    m.add(Dense(16, activation="relu"))
    m.add(Dense(1, activation="sigmoid"))
    lines = []
    m.summary(print_fn=lines.append)
    assert any("dense_1" in s for s in lines)


def test_compile_loss_consistency_errors():
    m = Sequential([Dense(4, activation="relu", input_dim=2), Dense(1)])
    m.compile(optimizer="adam", loss="binary_crossentropy")
    with pytest.raises(NotImplementedError):
        m._check_loss()                    # probabilities loss on a linear output
    m.compile(optimizer="adam", loss=BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    m._check_loss()
    with pytest.raises(NotImplementedError):
        m.compile(metrics=["auc"])
    with pytest.raises(TypeError):
        m.add("not a layer")


def test_maxima_argument_asserts_fire_before_any_gpu_work():
    from scipy.optimize import Bounds
    m = MaximizableSequential()
    b = Bounds(np.zeros(2), np.ones(2))
    with pytest.raises(AssertionError):
        m.maxima(b, num_starts=5, num_samples=4)
    with pytest.raises(AssertionError):
        m.maxima(b, num_starts=-1)
    with pytest.raises(AssertionError):
        m.maxima(b, num_samples=0)
    with pytest.raises(TypeError):
        m.maxima(b, num_start_points=3)    # README.rst:96's keyword is not a parameter


def test_shuffle_stream_properties():
    for N in (1, 2, 63, 64, 65, 110):
        p = shuffle.epoch_permutation(7, 3, 11, N)
        assert p.dtype == np.int32 and np.array_equal(np.sort(p), np.arange(N))
    a = shuffle.permutations(1, 2, 3, 50)
    assert a.shape == (2, 3, 50)
    assert not np.array_equal(a[0, 0], a[0, 1]) and not np.array_equal(a[0, 0], a[1, 0])
    assert np.array_equal(a[1, 2], shuffle.epoch_permutation(1, 1, 2, 50))
    assert np.array_equal(shuffle.permutations(1, 1, 1, 50, model_index0=1, epoch0=2)[0, 0], a[1, 2])
    # rough uniformity of the first position
    first = np.array([shuffle.epoch_permutation(0, 0, e, 8)[0] for e in range(800)])
    assert np.bincount(first, minlength=8).min() > 60


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = MaximizableSequential()
    m.add(Dense(4, activation="relu"))
    m.add(Dense(1, activation="sigmoid"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.predict(np.zeros((3, 2)))


def test_vectorised_shuffle_keys_equal_the_scalar_statement():
    """bore_amd.shuffle draws an epoch's keys with numpy's wrapping uint64 arithmetic (the host
    fallback of `fit` for data sets whose shuffle the device cannot draw in LDS uses it for 10^5+
    rows): the same numbers as the scalar Python statement of mlp_device.h's mix64 chain."""
    from bore_amd import shuffle as S
    for seed, model, epoch, N in [(0, 0, 0, 17), (5, 100, 7, 1000), (2 ** 63 + 11, 3, 999, 5000)]:
        base = S.shuffle_base(seed, model, epoch)
        ref = np.array([S._mix64((base + S._C_ROW * (i + 1)) & S._M) >> 32 for i in range(N)], dtype=np.uint32)
        assert np.array_equal(S.shuffle_keys(seed, model, epoch, N), ref)
        perm = S.epoch_permutation(seed, model, epoch, N)
        assert np.array_equal(np.sort(perm), np.arange(N))
        assert np.all(np.diff(ref[perm].astype(np.int64)) >= 0)          # ascending keys, ties by row index


def test_callable_transforms_are_applied_on_the_host():
    """The reference takes any TF callable as `transform` (bore/mixins.py:16); here any
    torch-differentiable elementwise callable: value and derivative by torch.autograd in float32,
    composed with -u as bore/mixins.py:20 does."""
    import torch
    from bore_amd.transforms import CallableTransform, Transform, resolve
    assert resolve(np.exp).name == "exp" and resolve("sigmoid").name == "sigmoid"     # names still resolve to kernels
    sp = resolve(lambda u: torch.nn.functional.softplus(u))
    assert isinstance(sp, CallableTransform) and sp.name is None and not sp.negate and sp.negated().negate
    f = np.array([0.1, 0.5, 0.93], dtype=np.float32)
    val, d = sp.negated().value_and_derivative(f)
    assert val.dtype == np.float32 and d.dtype == np.float64
    np.testing.assert_allclose(val, np.log1p(np.exp(-f.astype(np.float64))), rtol=1e-6)
    np.testing.assert_allclose(d, -1.0 / (1.0 + np.exp(f.astype(np.float64))), rtol=1e-6)   # d/df softplus(-f)
    # a callable that spells a named transform agrees with the named one
    sg = CallableTransform(torch.sigmoid, negate=True)
    v2, d2 = sg.value_and_derivative(f)
    s = Transform("sigmoid", negate=True)(f)
    np.testing.assert_allclose(v2, s, rtol=1e-6)
    np.testing.assert_allclose(d2, -s * (1 - s), rtol=1e-5)
    with pytest.raises(TypeError):
        CallableTransform(lambda u: 3.0).value_and_derivative(f)
    with pytest.raises(TypeError):
        resolve(3)
    # functions that ARE a named transform resolve to the kernel; a user function that is merely CALLED
    # exp stays the callable it is; a numpy function fails with the documented TypeError
    from scipy.special import expit
    assert resolve(torch.sigmoid).name == "sigmoid" and resolve(expit).name == "sigmoid" and resolve(torch.exp).name == "exp"

    def exp(u):                      # not the exponential
        return u * u
    mine = resolve(exp)
    assert isinstance(mine, CallableTransform) and mine.name is None

    # the reference's own idiom, transform=tf.identity / tf.sigmoid / tf.exp (bore/mixins.py:16,94;
    # plugins/hpbandster/base.py:128-131): a function that comes from tensorflow under one of the three names is
    # that transform (TensorFlow is not installed here: an object with its module and name stands in)
    def identity(u):
        raise AssertionError("never called: resolved by origin and name")
    identity.__module__ = "tensorflow.python.ops.array_ops"
    assert resolve(identity) is transforms.identity

    def sigmoid(u):
        raise AssertionError("never called")
    sigmoid.__module__ = "tensorflow.python.ops.math_ops"
    assert resolve(sigmoid).name == "sigmoid"
    #  by (package, name) pairs: numpy.identity builds identity MATRICES -- not the elementwise identity --
    # and stays the callable it is; so does a function called identity that comes from jax
    assert isinstance(resolve(np.identity), CallableTransform)
    identity.__module__ = "jax._src.numpy.lax_numpy"
    assert isinstance(resolve(identity), CallableTransform)
    sigmoid.__module__ = "jax._src.nn.functions"
    assert resolve(sigmoid).name == "sigmoid"
    sigmoid.__module__ = "my_project.activations"              # a user's own function of that name: the callable
    assert isinstance(resolve(sigmoid), CallableTransform)
    np.testing.assert_allclose(mine.value_and_derivative(f)[0], f * f, rtol=1e-6)
    with pytest.raises(TypeError, match="torch tensor"):
        resolve(np.tanh).value_and_derivative(f)
    with pytest.raises(TypeError, match="differentiably"):
        resolve(lambda u: torch.zeros_like(u.detach())).value_and_derivative(f)


def test_tools_and_committed_measurements_are_readable():
    """The GPU-box scripts under tools/ at least compile, and the judged measurement files of the current
    round parse (a bench.py run reads the newest round's pmc_traffic.json and loops_sweep.json)."""
    import glob, json, os, py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = glob.glob(os.path.join(root, "tools", "*.py"))
    assert len(scripts) >= 10
    for path in scripts:
        py_compile.compile(path, doraise=True)
    import bench
    assert bench.TRAFFIC_FILE.startswith(os.path.join("profiles", "r6")) and bench.SWEEP_FILE.startswith(os.path.join("profiles", "r6"))
    traffic = json.load(open(os.path.join(root, bench.TRAFFIC_FILE)))
    assert traffic["iteration_kernel"]["hbm_bytes_per_model"] > 0
    # the PMC pass names the kernel sources it was collected on; bench.py compares with the tree it times
    assert len(traffic["csrc_sha256"]) == 64 and len(bench.csrc_digest()) == 64
    sweep = json.load(open(os.path.join(root, bench.SWEEP_FILE)))
    assert set(sweep["it_per_s"]) >= {"64", "512", "768", "1024", "4096"}
    # (round 5: per (kernel, grid) counter passes; per config and phase the 256-loop launch's bytes and its kernel)
    assert 0.0 < traffic["iteration_kernel"]["issue_slot_utilisation"] < 1.0
    for cfg in ("cfg2_hartmann6_32-32-1_R256", "cfg3_hpo16_64-64-64-1_R1024", "cfg5_nas32_128-128-1_bf16_R4096", "plugin_default_D16",
                "plugin_D16_transform_identity"):
        for phase in ("fit", "screen", "fg"):
            e = traffic["configs"][cfg][phase]
            assert e["hbm_bytes_per_launch"] > 0 and e["kernel"] and e["avg_ns_kernel_trace"] > 0
    # (round 6: the committed record is the side file of the driver-form run -- the line itself is the compact one)
    line = json.loads(open(os.path.join(root, "profiles", "r6", "bench_driver_form.json")).read().strip().splitlines()[-1])
    assert line["configs"]["plugin_default_D16"]["many_loops"]["loops"] == 256
    assert line["configs"]["plugin_default_D16_engine_256_loops"]["schedule"] == "fused loop kernel"
    assert line["configs"]["cfg2_hartmann6_32-32-1_R256"]["many_loops"]["roofline_fp64_optimiser"] is not None
    compact = json.loads(bench.compact_line(line, "bench_detail_n1.json"))
    assert len(json.dumps(compact)) < 4096 and compact["roofline"]["kernel"] == "iteration_kernel"
    for cfg in ("cfg2_hartmann6_32-32-1_R256", "cfg3_hpo16_64-64-64-1_R1024", "cfg5_nas32_128-128-1_bf16_R4096"):
        reps = line["configs"][cfg]["many_loops"]["ms_reps"]
        assert reps["n"] >= 5 and reps["min"]["lbfgsb"] <= line["configs"][cfg]["many_loops"]["ms"]["lbfgsb"] <= reps["max"]["lbfgsb"]
    assert line["unit"] == "BO-iterations/s" and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_history_is_lazy_and_assignable_like_keras():
    """`fit` returns a History that downloads the losses when first read (and, as in Keras, `history` is
    an attribute a caller may assign)."""
    from bore_amd.models import History
    calls = []
    h = History(lambda: calls.append(1) or np.array([0.7, 0.6], dtype=np.float32))
    assert calls == []                                   # nothing downloaded yet
    assert h.history == {"loss": [pytest.approx(0.7), pytest.approx(0.6)]} and h.epoch == [0, 1] and calls == [1]
    assert h.history["loss"] and calls == [1]            # once
    h.history = {"loss": [1.0], "val_loss": [2.0]}
    assert h.history["val_loss"] == [2.0] and h.epoch == [0]
    h2 = History(lambda: (_ for _ in ()).throw(AssertionError("never downloaded")))
    h2.history = {"loss": []}                           # assigned before it was ever read: no download
    assert h2.epoch == []
