"""The plugin's BO driver without HpBandSter / ConfigSpace.

``ClassifierSuggester`` is ``ClassifierConfigGenerator`` (bore/plugins/hpbandster/base.py:84-288)
without HyperBand around it.  Candidates are plain arrays in a box, or -- with ``space=`` a
``bore_amd.plugins.types.DenseSpace`` (the ConfigSpace-free statement of the reference's dense
one-hot encoding, types.py) -- configuration dictionaries, encoded and decoded as
``array_from_dict`` / ``dict_from_array`` do in the reference (:250-265, :274-278).  Everything between --
what is fitted when, with which defaults, how a suggestion is chosen, when a random point is
returned instead -- follows the reference:

    get_config  (:216-265)  ->  suggest()
    new_result  (:267-288)  ->  observe(x, loss, budget)

Defaults are those of ``BORE.__init__`` (:23-30).
"""
import logging

import numpy as np
from scipy.optimize import Bounds

from ..base import maybe_distort
from ..data import Record
from ..layers import BinaryCrossentropy, l2
from ..math import steps_per_epoch
from ..models import MaximizableDenseSequential
from ..optimizers.utils import from_bounds
from ..transforms import TRANSFORMS


class ClassifierSuggester:
    def __init__(self, bounds=None, gamma=1 / 3, num_random_init=10, random_rate=0.1, retrain=False,
                 num_starts=5, num_samples=1024, batch_size=64, num_steps_per_iter=1000,
                 num_epochs_per_iter=None, optimizer="adam", num_layers=2, num_units=32,
                 activation="elu", l2_factor=None, transform="sigmoid", method="L-BFGS-B",
                 max_iter=1000, ftol=1e-9, distortion=None, seed=None, logger=None, space=None):
        """space: a ``DenseSpace``; then ``bounds`` is its unit box, ``suggest()`` returns a
        configuration dictionary and ``observe()`` takes one."""
        assert (bounds is None) != (space is None), "give either `bounds` or `space`"
        self.space = space
        if space is not None:
            bounds = space.get_bounds()
        assert 0. < gamma < 1., "`gamma` must be in (0, 1)"
        assert num_random_init > 0, "number of initial random designs must be non-zero!"
        assert random_rate is None or 0. <= random_rate < 1., "`random_rate` must be in [0, 1)"
        assert transform in TRANSFORMS, f"`transform` must be one of {tuple(TRANSFORMS)}"
        (low, high), dim = from_bounds(bounds)
        self.low, self.high = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)
        self.bounds = Bounds(lb=self.low, ub=self.high)
        self.input_dim = dim
        self.gamma, self.num_random_init, self.random_rate = gamma, num_random_init, random_rate
        self.retrain = retrain
        self.num_layers, self.num_units, self.activation = num_layers, num_units, activation
        self.optimizer = optimizer
        self.kernel_regularizer = None if l2_factor is None else l2(l2_factor)
        self.bias_regularizer = None if l2_factor is None else l2(l2_factor)
        self.batch_size = batch_size
        self.num_steps_per_iter, self.num_epochs_per_iter = num_steps_per_iter, num_epochs_per_iter
        self.transform = transform
        self.num_starts, self.num_samples, self.method = num_starts, num_samples, method
        self.ftol, self.max_iter, self.distortion = ftol, max_iter, distortion
        self.logit = None
        self.record = Record()
        self.seed = seed
        self.random_state = np.random.RandomState(seed)
        # the reference draws its random configurations from the ConfigSpace's own seeded
        # generator (DenseConfigurationSpace(config_space, seed=seed), :100), not from
        # random_state: a separate stream here too
        self._space_rng = np.random.RandomState(seed)
        self._net_rng = np.random.RandomState(seed)
        self.logger = logger or logging.getLogger("bore_amd.plugins")
        self.last_fit = None  # (loss, accuracy) of the most recent update

    # -- model -------------------------------------------------------------------------------
    def _build_compile_network(self):
        """:145-159 (note DenseSequential's num_layers + 1 hidden layers, bore/models.py:16-19)."""
        # (the reference leaves weight init and shuffling to TF's global seed; here a seeded
        # suggester is reproducible end to end: every network it builds draws from one stream)
        net_seed = None if self.seed is None else int(self._net_rng.randint(0, 2**31 - 1))
        network = MaximizableDenseSequential(
            transform=self.transform, input_dim=self.input_dim, output_dim=1,
            num_layers=self.num_layers, num_units=self.num_units,
            layer_kws=dict(activation=self.activation,
                           kernel_regularizer=self.kernel_regularizer,
                           bias_regularizer=self.bias_regularizer), seed=net_seed)
        network.compile(optimizer=self.optimizer, metrics=["accuracy"],
                        loss=BinaryCrossentropy(from_logits=True))
        network.summary(print_fn=self.logger.debug)
        return network

    def _update_classifier(self):
        """:161-194: label, fit (epochs from the step budget unless given), evaluate."""
        X, z = self.record.load_classification_data(self.gamma)
        dataset_size = self.record.size()
        num_steps = steps_per_epoch(dataset_size, self.batch_size)
        epochs = self.num_epochs_per_iter
        if epochs is None:
            epochs = self.num_steps_per_iter // num_steps
        self.logit.fit(X, z, epochs=epochs, batch_size=self.batch_size, callbacks=[],
                       verbose=False)
        loss, accuracy = self.logit.evaluate(X, z, verbose=False)
        self.last_fit = (loss, accuracy)
        self.logger.info(f"[Model fit: loss={loss:.3f}, accuracy={accuracy:.3f}] "
                         f"dataset size: {dataset_size}, batch size: {self.batch_size}, "
                         f"steps per epoch: {num_steps}, num epochs: {epochs}")

    def _is_unique(self, res):
        """:210-214."""
        dup = self.record.is_duplicate(res.x)
        if dup:
            self.logger.warning("Duplicate detected! Skipping...")
        return not dup

    # -- the two entry points ------------------------------------------------------------------
    def suggest(self):
        """One candidate, as ``get_config`` (:216-265) picks it; returns ``(x, info)`` with
        ``info["source"]`` in {"random:rate", "random:init", "random:failed", "model"}.  With a
        ``space`` the candidate is a configuration dictionary (dict_from_array, :264)."""
        x, info = self._suggest_array()
        if self.space is not None:
            return (x if isinstance(x, dict) else self.space.from_array(x)), info
        return x, info

    def _suggest_array(self):
        if self.space is not None:   # config_space.sample_configuration() (:220-221)
            x_random = self.space.sample_configuration()
        else:
            x_random = self._space_rng.uniform(self.low, self.high)
        if self.random_rate is not None and self.random_state.binomial(p=self.random_rate, n=1):
            return x_random, dict(source="random:rate")
        if self.record.size() < self.num_random_init:
            return x_random, dict(source="random:init")
        if self.logit is None:
            self.logit = self._build_compile_network()
        self._update_classifier()
        opt = self.logit.argmax(self.bounds, num_starts=self.num_starts,
                                num_samples=self.num_samples, method=self.method,
                                options=dict(maxiter=self.max_iter, ftol=self.ftol),
                                print_fn=self.logger.debug, filter_fn=self._is_unique,
                                random_state=self.random_state)
        if opt is None:
            self.logger.warning("[Glob. maximum: not found!] Suggesting random candidate...")
            return x_random, dict(source="random:failed")
        self.logger.info(f"[Glob. maximum: value={-opt.fun:.3f} x={opt.x}]")
        x = maybe_distort(opt.x, self.distortion, self.bounds, self.random_state,
                          print_fn=self.logger.info)
        if self.retrain:  # :201-208: drop the model, the next suggestion trains from scratch
            self.logit = None
        return np.asarray(x, dtype=np.float64), dict(source="model", value=-opt.fun)

    def observe(self, x, loss, budget=None):
        """``new_result`` (:267-288): log the evaluated candidate (a dictionary is encoded with
        array_from_dict, :276)."""
        if isinstance(x, dict):
            assert self.space is not None, "a configuration dictionary needs `space`"
            x = self.space.to_array(x)
        self.record.append(x=np.asarray(x, dtype=np.float64), y=loss, b=budget)
