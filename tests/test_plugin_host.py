"""Host-side callers of the hot path: candidate post-processing and the plugin's
suggest/observe control flow (no GPU needed for these parts)."""
import os

import numpy as np
import pytest
from scipy.optimize import Bounds
from scipy.stats import truncnorm

from bore_amd.base import maybe_distort, truncated_normal
from bore_amd.plugins import ClassifierSuggester

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_maybe_distort_follows_bore_base():
    """bore/base.py:45-64: None -> untouched; else one truncated-normal draw inside the box,
    from the caller's RandomState (consumed)."""
    b = Bounds(lb=np.array([0.0, -1.0]), ub=np.array([1.0, 2.0]))
    loc = np.array([0.98, -0.9])
    assert maybe_distort(loc, None, b) is loc
    with pytest.raises(AssertionError):
        maybe_distort(loc, 0.1)
    msgs = []
    rs = np.random.RandomState(5)
    x = maybe_distort(loc, 0.3, b, rs, print_fn=msgs.append)
    assert np.all(x >= b.lb) and np.all(x <= b.ub) and len(msgs) == 1 and "3.000E-01" in msgs[0]
    want = truncnorm(a=(b.lb - loc) / 0.3, b=(b.ub - loc) / 0.3, loc=loc,
                     scale=0.3).rvs(random_state=np.random.RandomState(5))
    assert np.array_equal(x, want)
    assert not np.array_equal(maybe_distort(loc, 0.3, b, rs, print_fn=msgs.append), x)
    d = truncated_normal(0.5, 0.1, 0.0, 1.0)
    assert abs(d.cdf(1.0) - 1.0) < 1e-12 and d.cdf(0.0) == 0.0


def test_suggester_random_phases_and_streams():
    """get_config's control flow before any model exists (bore/plugins/hpbandster/base.py:
    216-236): epsilon-greedy draw from random_state first, then the initial design; random
    candidates come from the space's own stream."""
    s = ClassifierSuggester([(0.0, 1.0), (-5.0, 5.0)], seed=3, random_rate=0.5, num_random_init=4)
    space = np.random.RandomState(3)
    coin = np.random.RandomState(3)
    seen = set()
    for i in range(4):
        x, info = s.suggest()
        assert np.array_equal(x, space.uniform([0.0, -5.0], [1.0, 5.0]))
        assert info["source"] == ("random:rate" if coin.binomial(p=0.5, n=1) else "random:init")
        seen.add(info["source"])
        s.observe(x, float(i), budget=1.0)
    assert s.record.size() == 4 and s.record.budgets == [1.0] * 4 and s.logit is None
    with pytest.raises(AssertionError):
        ClassifierSuggester([(0, 1)], gamma=1.5)
    with pytest.raises(AssertionError):
        ClassifierSuggester([(0, 1)], transform="tanh")


@pytest.mark.gpu
def test_suggester_runs_the_plugin_loop(gpu):
    """The plugin's defaults end to end (elu x3 hidden layers of 32, logit output, sigmoid
    transform, 1000 steps per iteration, 5 restarts, duplicate filter) on Branin."""
    from bore_amd.engine import branin01
    s = ClassifierSuggester([(0.0, 1.0)] * 2, seed=0, random_rate=None, num_random_init=8,
                            num_steps_per_iter=200)
    sources = []
    for i in range(20):
        x, info = s.suggest()
        assert x.shape == (2,) and np.all(x >= 0) and np.all(x <= 1)
        assert not s.record.is_duplicate(x) or info["source"] != "model"
        sources.append(info["source"])
        s.observe(x, float(branin01(x)))
    assert sources[:8] == ["random:init"] * 8 and sources.count("model") >= 10
    assert len(s.logit.layers) == 4                    # num_layers + 1 hidden + output
    loss, acc = s.last_fit
    assert np.isfinite(loss) and 0.5 <= acc <= 1.0
    y = np.array(s.record.targets)
    assert y[8:].min() < np.median(y[:8])              # the model-guided points find better values
    # retrain=True drops the model after every suggestion; distortion perturbs inside the box
    s2 = ClassifierSuggester([(0.0, 1.0)] * 2, seed=1, random_rate=None, num_random_init=3,
                             num_steps_per_iter=100, retrain=True, distortion=0.05)
    for i in range(5):
        x, info = s2.suggest()
        s2.observe(x, float(branin01(x)))
    assert s2.logit is None and info["source"] == "model"
    # a seeded suggester is reproducible end to end (weights, shuffles, candidates)
    runs = []
    for _ in range(2):
        s3 = ClassifierSuggester([(0.0, 1.0)] * 2, seed=7, num_random_init=4, num_steps_per_iter=100)
        xs = []
        for i in range(8):
            x, _ = s3.suggest()
            xs.append(x)
            s3.observe(x, float(branin01(x)))
        runs.append(np.array(xs))
    assert np.array_equal(runs[0], runs[1])


def _reference_space(seed=8888):
    """The search space of the reference's tests/test_types.py:17-33."""
    from bore_amd.plugins.types import Categorical, DenseSpace, UniformFloat, UniformInteger
    return DenseSpace([
        UniformInteger("n_units_1", lower=0, upper=5), UniformInteger("n_units_2", lower=0, upper=5),
        UniformFloat("dropout_1", lower=0, upper=0.9), UniformFloat("dropout_2", lower=0, upper=0.9),
        Categorical("activation_fn_1", ["tanh", "relu"]), Categorical("activation_fn_2", ["tanh", "relu"]),
        UniformInteger("init_lr", lower=0, upper=5), Categorical("lr_schedule", ["cosine", "const"]),
        UniformInteger("batch_size", lower=0, upper=3)], seed=seed)


def test_dense_encoding_reproduces_the_reference_test_vector():
    """tests/test_types.py:36-104 of the reference, values and all: shapes, bounds, the dense
    array of a given configuration, exact round trip, argmax decoding of soft one-hots."""
    cs = _reference_space()
    assert cs.get_dimensions(sparse=True) == 9 and cs.get_dimensions(sparse=False) == 12
    b = cs.get_bounds()
    np.testing.assert_array_equal(b.lb, np.zeros(12))
    np.testing.assert_array_equal(b.ub, np.ones(12))
    assert cs.get_hyperparameter_by_idx(0) == "activation_fn_1"
    dct = {'activation_fn_1': 'relu', 'activation_fn_2': 'tanh', 'batch_size': 2,
           'dropout_1': 0.39803953082292726, 'dropout_2': 0.022039062686389176, 'init_lr': 0,
           'lr_schedule': 'cosine', 'n_units_1': 5, 'n_units_2': 1}
    array = cs.to_array(dct)
    assert np.less_equal(0., array).all() and np.less_equal(array, 1.).all()
    np.testing.assert_array_almost_equal(array, [0., 1., 1., 0., 0.62500063, 0.44226615, 0.02448785,
                                                 0.08333194, 1., 0., 0.91666806, 0.24999917])
    assert cs.from_array(array) == dct                      # "recover original dictionary exactly"
    array[0], array[1] = 0.8, 0.6                           # soft one-hot: argmax
    assert cs.from_array(array)["activation_fn_1"] == "tanh"
    for cfg in cs.sample_configuration(size=5):
        a = cs.to_array(cfg)
        assert a.shape == (12,) and cs.from_array(a) == cfg
    assert isinstance(cs.sample_configuration(), dict)


class _FakeHp:
    """A ConfigSpace hyperparameter as far as the converter looks at one: class name + attributes."""
    def __init__(self, name, **kw):
        self.name = name
        self.__dict__.update(kw)


def _fake_configspace_classes():
    mk = lambda cls: type(cls, (_FakeHp,), {})
    return (mk("UniformIntegerHyperparameter"), mk("UniformFloatHyperparameter"),
            mk("CategoricalHyperparameter"), mk("NormalFloatHyperparameter"))


def test_configspace_bridge_takes_a_configuration_space_by_duck_typing():
    """DenseConfigurationSpace(other, seed) (bore/plugins/hpbandster/types.py:17-35) wraps a real
    ConfigSpace.ConfigurationSpace; ConfigSpace is not in this image, so the bridge goes by class name
    and attributes.  A fake space with the reference's own test hyperparameters
    (tests/test_types.py:17-33, insertion order as there) must give the dense encoding of the
    reference's test vector; old (get_hyperparameters) and new (Mapping) ConfigSpace surfaces;
    unsupported types raise the reference's NotImplementedError."""
    from bore_amd.plugins.types import DenseSpace, dense_space_from
    from bore_amd.plugins.hpbandster import ClassifierConfigGenerator
    UI, UF, Cat, Normal = _fake_configspace_classes()
    hps = [UI("n_units_1", lower=0, upper=5, log=False), UI("n_units_2", lower=0, upper=5, log=False),
           UF("dropout_1", lower=0, upper=0.9, log=False), UF("dropout_2", lower=0, upper=0.9, log=False),
           Cat("activation_fn_1", choices=("tanh", "relu")), Cat("activation_fn_2", choices=("tanh", "relu")),
           UI("init_lr", lower=0, upper=5, log=False), Cat("lr_schedule", choices=("cosine", "const")),
           UI("batch_size", lower=0, upper=3, log=False)]

    class OldSpace:                       # ConfigSpace < 0.7
        def get_hyperparameters(self):
            return list(hps)

    class NewSpace(dict):                 # ConfigSpace >= 0.7: Mapping name -> hyperparameter
        pass

    want = _reference_space()
    dct = {'activation_fn_1': 'relu', 'activation_fn_2': 'tanh', 'batch_size': 2,
           'dropout_1': 0.39803953082292726, 'dropout_2': 0.022039062686389176, 'init_lr': 0,
           'lr_schedule': 'cosine', 'n_units_1': 5, 'n_units_2': 1}
    for space in (OldSpace(), NewSpace((hp.name, hp) for hp in hps)):
        cs = dense_space_from(space, seed=8888)
        assert isinstance(cs, DenseSpace)
        assert cs.get_dimensions(sparse=True) == 9 and cs.get_dimensions(sparse=False) == 12
        assert cs.get_hyperparameter_by_idx(0) == "activation_fn_1"
        assert np.array_equal(cs.to_array(dct), want.to_array(dct))
        assert cs.from_array(cs.to_array(dct)) == dct
        assert cs.sample_configuration(size=3) == _reference_space().sample_configuration(size=3)
    assert dense_space_from(want) is want
    with pytest.raises(NotImplementedError, match="Only hyperparameters of types"):
        dense_space_from(NewSpace(x=Normal("x", mu=0.0, sigma=1.0)))
    with pytest.raises(TypeError, match="not a configuration space"):
        dense_space_from(object())
    # the HpBandSter generator accepts the ConfigSpace object itself (reference: base.py:100)
    gen = ClassifierConfigGenerator(OldSpace(), gamma=0.25, num_random_init=3, random_rate=None, retrain=False,
                                    classifier_kws={}, fit_kws={}, optimizer_kws={}, seed=8888)
    cfg, info = gen.get_config(budget=1.0)          # fewer than num_random_init observations: a random draw
    assert set(cfg) == set(dct) and info == {}


def test_dense_encoding_log_scales_and_integer_edges():
    from bore_amd.plugins.types import DenseSpace, UniformFloat, UniformInteger, array_from_dict, dict_from_array
    cs = DenseSpace([UniformFloat("lr", 1e-5, 1e-1, log=True), UniformInteger("width", 16, 1024, log=True),
                     UniformInteger("depth", 1, 4)])
    for cfg in [dict(lr=1e-5, width=16, depth=1), dict(lr=1e-1, width=1024, depth=4),
                dict(lr=3e-3, width=100, depth=2)]:
        a = array_from_dict(cs, cfg)
        assert ((a >= 0) & (a <= 1)).all()
        back = dict_from_array(cs, a)
        assert back["width"] == cfg["width"] and back["depth"] == cfg["depth"]
        assert back["lr"] == pytest.approx(cfg["lr"], rel=1e-12)
    # every integer owns an equal share of [0, 1]
    u = np.linspace(0, 1, 4001)
    assert cs.get_hyperparameter_by_idx(0) == "depth"       # (sorted by name, like ConfigSpace)
    counts = np.bincount([dict_from_array(cs, np.array([t, 0.5, 0.5]))["depth"] for t in u])[1:]
    assert counts.min() >= 999 and counts.max() <= 1002


@pytest.mark.gpu
def test_suggester_over_a_mixed_space(gpu):
    """get_config / new_result with configuration dictionaries (bore/plugins/hpbandster/base.py:
    250-265, 274-278): a categorical + integer + float space through the dense encoding."""
    from bore_amd.plugins.types import Categorical, DenseSpace, UniformFloat, UniformInteger
    space = DenseSpace([Categorical("act", ["tanh", "relu", "elu"]), UniformInteger("units", 1, 8),
                        UniformFloat("lr", 1e-4, 1e-1, log=True)], seed=3)
    good = dict(act="relu", units=6, lr=1e-2)

    def loss(cfg):      # a made-up validation loss with a known best configuration
        return (0.0 if cfg["act"] == good["act"] else 1.0) + 0.1 * abs(cfg["units"] - good["units"]) \
            + abs(np.log10(cfg["lr"]) - np.log10(good["lr"]))

    s = ClassifierSuggester(space=space, seed=0, random_rate=None, num_random_init=10,
                            num_steps_per_iter=200)
    assert s.input_dim == 5
    sources = []
    for i in range(30):
        cfg, info = s.suggest()
        assert set(cfg) == {"act", "units", "lr"} and cfg["act"] in ("tanh", "relu", "elu")
        assert isinstance(cfg["units"], int) and 1 <= cfg["units"] <= 8 and 1e-4 <= cfg["lr"] <= 1e-1
        sources.append(info["source"])
        s.observe(cfg, loss(cfg))
    assert sources.count("model") >= 15
    X = np.vstack(s.record.features)
    assert X.shape == (30, 5) and ((X >= 0) & (X <= 1)).all()
    assert np.all(np.sort(X[:, :3], axis=1)[:, :2] == 0) and np.all(X[:, :3].max(axis=1) == 1)   # one-hot
    y = np.array(s.record.targets)
    assert y[10:].min() <= y[:10].min()                  # the classifier finds at least as good a config


class _FakeJob:
    """What hpbandster hands to new_result: kwargs (config, budget), result, exception, id."""

    def __init__(self, config, budget, loss, exception=None, id=(0, 0, 0)):
        self.kwargs = dict(config=config, budget=budget)
        self.result = dict(loss=loss, info={})
        self.exception, self.id = exception, id


def _adapter_space(seed=None):
    from bore_amd.plugins.types import Categorical, DenseSpace, UniformFloat, UniformInteger
    return DenseSpace([Categorical("act", ["tanh", "relu", "elu"]), UniformInteger("units", 1, 8),
                       UniformFloat("lr", 1e-4, 1e-1, log=True)], seed=seed)


def test_hpbandster_adapter_surface_before_any_model():
    """bore/plugins/hpbandster/base.py:216-236, 267-288: get_config(budget) -> (dict, {}) from the
    space's own seeded stream while the record is short; new_result(job) encodes
    job.kwargs["config"] with the dense one-hot map and records job.result["loss"] and the budget."""
    from bore_amd.plugins.hpbandster import BORE, ClassifierConfigGenerator
    cg = ClassifierConfigGenerator(
        config_space=_adapter_space(), gamma=0.25, num_random_init=5, random_rate=None,
        retrain=False, classifier_kws={}, fit_kws={}, optimizer_kws=dict(num_starts=3), seed=11)
    mirror = _adapter_space(seed=11)                    # DenseConfigurationSpace(config_space, seed=seed)
    for i in range(5):
        cfg, info = cg.get_config(budget=1.0 / 3 ** (i % 3))
        assert info == {} and cfg == mirror.sample_configuration()
        cg.new_result(_FakeJob(cfg, 1.0 / 3 ** (i % 3), loss=float(i)))
    assert cg.record.size() == 5 and cg.record.targets == [0.0, 1.0, 2.0, 3.0, 4.0]
    assert cg.record.budgets[:3] == [1.0, 1.0 / 3, 1.0 / 9] and cg.logit is None
    assert np.array_equal(cg.record.features[-1], mirror.to_array(cfg))
    assert cg.input_dim == 5 and np.array_equal(cg.bounds.ub, np.ones(5))
    # a failed job is logged by the base class and still indexed like the reference does (:269-288)
    msgs = []
    cg.logger = type("L", (), {"warning": lambda self, m: msgs.append(m)})()
    cg.new_result(_FakeJob(cfg, 1.0, loss=9.0, exception="Traceback ...", id=(1, 0, 2)))
    assert cg.record.size() == 6 and len(msgs) == 1 and "(1, 0, 2)" in msgs[0]
    # the reference's argument checks (:91-96, :127-128)
    for bad in (dict(gamma=1.0), dict(num_random_init=0), dict(random_rate=1.0)):
        kw = dict(config_space=_adapter_space(), gamma=0.25, num_random_init=5, random_rate=0.1,
                  retrain=False, classifier_kws={}, fit_kws={}, optimizer_kws={}, seed=0)
        kw.update(bad)
        with pytest.raises(AssertionError):
            ClassifierConfigGenerator(**kw)
    with pytest.raises(AssertionError):
        ClassifierConfigGenerator(config_space=_adapter_space(), gamma=0.25, num_random_init=5,
                                  random_rate=0.1, retrain=False, classifier_kws={}, fit_kws={},
                                  optimizer_kws=dict(transform="tanh"), seed=0)
    # BORE(HyperBand): gamma defaults to 1 / eta, HyperBand's ladder of budgets (:33-34, :64-82)
    opt = BORE(_adapter_space(), eta=3, min_budget=1, max_budget=81, seed=2)
    assert opt.config_generator.gamma == pytest.approx(1 / 3) and opt.max_SH_iter == 5
    np.testing.assert_allclose(opt.budgets, [1, 3, 9, 27, 81])
    assert opt.config["gamma"] == pytest.approx(1 / 3) and opt.config["num_random_init"] == 10
    assert opt.config_generator.num_random_init == 10
    assert opt.config_generator._suggester.num_starts == 5
    assert opt.config_generator._suggester.num_steps_per_iter == 1000


@pytest.mark.gpu
def test_hpbandster_adapter_drives_the_hot_path(gpu):
    """get_config / new_result as a HpBandSter master calls them, with the model phases on the
    GPU: after num_random_init results every get_config fits and maximises the classifier."""
    from bore_amd.plugins.hpbandster import BORE
    good = dict(act="relu", units=6, lr=1e-2)

    def loss(cfg):
        return (0.0 if cfg["act"] == good["act"] else 1.0) + 0.1 * abs(cfg["units"] - good["units"]) \
            + abs(np.log10(cfg["lr"]) - np.log10(good["lr"]))

    opt = BORE(_adapter_space(), eta=3, min_budget=1, max_budget=9, num_random_init=10,
               random_rate=None, num_steps_per_iter=200, num_starts=5, seed=0)
    cg = opt.config_generator
    sources = []
    for i in range(30):
        budget = float(opt.budgets[i % len(opt.budgets)])
        cfg, info = cg.get_config(budget)
        assert info == {} and set(cfg) == {"act", "units", "lr"}
        assert isinstance(cfg["units"], int) and 1 <= cfg["units"] <= 8 and 1e-4 <= cfg["lr"] <= 1e-1
        sources.append(cg.last_info["source"])
        cg.new_result(_FakeJob(cfg, budget, loss(cfg), id=(i, 0, 0)))
    assert sources[:10] == ["random:init"] * 10 and sources.count("model") >= 15
    assert cg.record.size() == 30 and len(cg.logit.layers) == 4
    y = np.array(cg.record.targets)
    assert y[10:].min() <= y[:10].min()
    with pytest.raises(ImportError):
        opt.run(n_iterations=1)
