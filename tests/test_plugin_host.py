"""Host-side callers of the hot path: candidate post-processing, batch
de-duplication and the plugin's suggest/observe control flow (no GPU needed for these parts)."""
import os

import numpy as np
import pytest
from scipy.optimize import Bounds
from scipy.stats import truncnorm

from bore_amd.base import maybe_distort, truncated_normal
from bore_amd.plugins import ClassifierSuggester
from bore_amd.utils.deduplicate import pad_unique_random, set_diff_2d

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_deduplicate_matches_reference_goldens():
    """tests/golden/ref_dedup.npz holds outputs of the reference's bore.utils.deduplicate."""
    g = np.load(os.path.join(GOLDEN, "ref_dedup.npz"))
    for k in range(4):
        A, size = g[f"A{k}"], int(g[f"size{k}"])
        B = g[f"B{k}"] if f"B{k}" in g.files else None
        d = A.shape[1]
        if B is not None:
            assert np.array_equal(set_diff_2d(np.unique(A, axis=0), B), g[f"diff{k}"])
        out = pad_unique_random(A, size=size, bounds=[(0.0, 1.0)] * d, B=B, random_state=11 + k)
        assert np.array_equal(out, g[f"out{k}"]), k
        assert out.shape == (size, d) and len(np.unique(out, axis=0)) == size


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
