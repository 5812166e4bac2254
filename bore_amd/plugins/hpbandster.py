"""The HpBandSter-facing surface of the plugin (bore/plugins/hpbandster/base.py:21-82, 84-143,
216-288): what a HpBandSter ``Master`` calls.

    ClassifierConfigGenerator(config_space, gamma, num_random_init, random_rate, retrain,
                              classifier_kws, fit_kws, optimizer_kws, seed)
        .get_config(budget)                 -> (config_dict, {})            (:216-265)
        .new_result(job, update_model=True)    reads job.kwargs["config" | "budget"],
                                               job.result["loss"]           (:267-288)
    BORE(config_space, eta=3, min_budget=0.01, max_budget=1, gamma=None, ...)   (:21-82)

Neither ``hpbandster`` nor ``ConfigSpace`` is in this image.  The generator is therefore
duck-typed: it derives from ``hpbandster.core.base_config_generator`` when that imports and from a
stand-in with the same two members (``logger``, ``new_result`` logging a job's exception)
otherwise, and ``config_space`` is a ``bore_amd.plugins.types.DenseSpace`` (the ConfigSpace-free
dense one-hot encoding) or a ``ConfigSpace.ConfigurationSpace``, whose hyperparameters are taken over
by class name (``types.dense_space_from``; float / integer / categorical, as types.py:76-88).  Everything between the two calls -- when a random configuration is
returned, what is fitted with which defaults, how the suggestion is picked, filtered and distorted
-- is ``ClassifierSuggester`` (classifier.py), i.e. the kernels of the hot path.

``BORE`` is HyperBand with this generator in place of the random one.  With hpbandster present it
IS a ``HyperBand`` (grandparent initialiser, as the reference does, :55-57); without it the class
still builds the generator and HyperBand's budget ladder (``eta``, ``budgets``, ``max_SH_iter``,
``config``: :59-82) so that the pieces can be driven and tested, and ``run()`` says what is missing.
"""
import logging

import numpy as np

from .classifier import ClassifierSuggester
from .types import DenseSpace, dense_space_from

try:                                                    # pragma: no cover  (not in this image)
    from hpbandster.core.base_config_generator import base_config_generator as _GeneratorBase
    from hpbandster.optimizers.hyperband import HyperBand as _HyperBand
    HAVE_HPBANDSTER = True
except Exception:
    HAVE_HPBANDSTER = False
    _HyperBand = object

    class _GeneratorBase:
        """The two members of hpbandster's base_config_generator this plugin relies on."""

        def __init__(self, logger=None):
            self.logger = logger if logger is not None else logging.getLogger("hpbandster")

        def new_result(self, job, update_model=True):
            if job.exception is not None:
                self.logger.warning("job {} failed with exception\n{}".format(job.id, job.exception))


class ClassifierConfigGenerator(_GeneratorBase):

    def __init__(self, config_space, gamma, num_random_init, random_rate, retrain, classifier_kws,
                 fit_kws, optimizer_kws, seed, **kwargs):
        super().__init__(**kwargs)
        # DenseConfigurationSpace(config_space, seed=seed) (:100): the hyperparameters of a DenseSpace or
        # of a ConfigSpace.ConfigurationSpace (types.dense_space_from), the space's own seeded stream
        self.config_space = DenseSpace(dense_space_from(config_space).hyperparameters, seed=seed)
        # same defaults as the reference reads out of the three dictionaries (:105-138)
        self._suggester = ClassifierSuggester(
            space=self.config_space, gamma=gamma, num_random_init=num_random_init,
            random_rate=random_rate, retrain=retrain,
            num_layers=classifier_kws.get("num_layers", 2),
            num_units=classifier_kws.get("num_units", 32),
            activation=classifier_kws.get("activation", "elu"),
            optimizer=classifier_kws.get("optimizer", "adam"),
            l2_factor=classifier_kws.get("l2_factor"),
            batch_size=fit_kws.get("batch_size", 64),
            num_steps_per_iter=fit_kws.get("num_steps_per_iter", 100),
            num_epochs_per_iter=fit_kws.get("num_epochs_per_iter"),
            transform=optimizer_kws.get("transform", "sigmoid"),
            num_starts=optimizer_kws.get("num_starts"),
            num_samples=optimizer_kws.get("num_samples", 1024),
            method=optimizer_kws.get("method", "L-BFGS-B"),
            ftol=optimizer_kws.get("ftol", 1e-9), max_iter=optimizer_kws.get("max_iter", 1000),
            distortion=optimizer_kws.get("distortion"), seed=seed, logger=self.logger)
        s = self._suggester
        self.gamma, self.num_random_init, self.random_rate = s.gamma, s.num_random_init, s.random_rate
        self.input_dim, self.bounds = s.input_dim, s.bounds
        self.retrain, self.seed = retrain, seed

    # the reference's attributes, live views of the suggester's state
    record = property(lambda self: self._suggester.record)
    logit = property(lambda self: self._suggester.logit)
    random_state = property(lambda self: self._suggester.random_state)

    def get_config(self, budget):
        """:216-265.  The budget plays no part in the suggestion (as in the reference)."""
        config_dict, info = self._suggester.suggest()
        self.last_info = info
        return (config_dict, {})

    def new_result(self, job, update_model=True):
        """:267-288."""
        super().new_result(job)
        budget = job.kwargs["budget"]     # (recorded; "we do not actually do anything with the budget")
        config_dict = job.kwargs["config"]
        loss = job.result["loss"]
        self._suggester.observe(config_dict, loss, budget=budget)


class BORE(_HyperBand):

    def __init__(self, config_space, eta=3, min_budget=0.01, max_budget=1, gamma=None,
                 num_random_init=10, random_rate=0.1, retrain=False, num_starts=5, num_samples=1024,
                 batch_size=64, num_steps_per_iter=1000, num_epochs_per_iter=None, optimizer="adam",
                 num_layers=2, num_units=32, activation="elu", l2_factor=None, transform="sigmoid",
                 method="L-BFGS-B", max_iter=1000, ftol=1e-9, distortion=None, seed=None, **kwargs):
        if gamma is None:
            gamma = 1 / eta
        cg = ClassifierConfigGenerator(
            config_space=config_space, gamma=gamma, num_random_init=num_random_init,
            random_rate=random_rate, retrain=retrain,
            classifier_kws=dict(num_layers=num_layers, num_units=num_units, l2_factor=l2_factor,
                                activation=activation, optimizer=optimizer),
            fit_kws=dict(batch_size=batch_size, num_steps_per_iter=num_steps_per_iter,
                         num_epochs_per_iter=num_epochs_per_iter),
            optimizer_kws=dict(transform=transform, method=method, max_iter=max_iter, ftol=ftol,
                               distortion=distortion, num_starts=num_starts, num_samples=num_samples),
            seed=seed)
        if HAVE_HPBANDSTER:                             # pragma: no cover
            super(_HyperBand, self).__init__(config_generator=cg, **kwargs)   # grandparent: Master
        else:
            self.config_generator, self.config = cg, {}
        # HyperBand's own set-up, which replacing the generator skips (:59-82)
        self.eta, self.min_budget, self.max_budget = eta, min_budget, max_budget
        self.max_SH_iter = -int(np.log(min_budget / max_budget) / np.log(eta)) + 1
        self.budgets = max_budget * np.power(eta, -np.linspace(self.max_SH_iter - 1, 0, self.max_SH_iter))
        self.config.update({'eta': eta, 'min_budget': min_budget, 'max_budget': max_budget,
                            'budgets': self.budgets, 'max_SH_iter': self.max_SH_iter, 'gamma': gamma,
                            'num_random_init': num_random_init, 'seed': seed})

    if not HAVE_HPBANDSTER:
        def run(self, *args, **kwargs):
            raise ImportError("BORE.run needs the hpbandster package (its Master / nameserver / "
                              "workers); drive `config_generator.get_config(budget)` / "
                              "`.new_result(job)` directly instead")
