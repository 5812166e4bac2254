"""Dense (one-hot) encoding of mixed search spaces, ConfigSpace-free.

What it stands in for: ``DenseConfigurationSpace`` / ``DenseConfiguration``
(bore/plugins/hpbandster/types.py:17-136) -- the map between a configuration dictionary and
the point in [0, 1]^d the classifier sees: numerical hyperparameters keep one coordinate,
categorical ones get one coordinate per choice (encode: one-hot; decode: argmax).  The
reference builds this on ConfigSpace, which supplies the hyperparameter classes, their
alphabetical ordering and the scaling of numerical values to [0, 1]; ConfigSpace is not in this
image, so those three things are restated here for the three hyperparameter types the
reference supports (types.py:76-88):

* ``UniformFloat(name, lower, upper, log=False)``:  (v - lower) / (upper - lower), in log space
  when ``log``;
* ``UniformInteger(name, lower, upper, log=False)``: the float rule on the widened interval
  [lower - 0.49999, upper + 0.49999] (so that rounding back gives every integer equal mass),
  decode rounds to the nearest integer;
* ``Categorical(name, choices)``: index of the choice.

Pinned by the reference's own test vector (tests/test_types.py:62-90, in
tests/test_plugin_host.py).
"""
import numpy as np
from scipy.optimize import Bounds


class UniformFloat:
    def __init__(self, name, lower, upper, log=False):
        assert upper > lower, f"{name}: upper must exceed lower"
        self.name, self.lower, self.upper, self.log = name, float(lower), float(upper), bool(log)
        self._lo, self._hi = (np.log(self.lower), np.log(self.upper)) if log else (self.lower, self.upper)

    size = 1

    def to_unit(self, value):
        v = np.log(value) if self.log else float(value)
        return (v - self._lo) / (self._hi - self._lo)

    def from_unit(self, u):
        v = u * (self._hi - self._lo) + self._lo
        v = float(np.exp(v)) if self.log else float(v)
        return min(max(v, self.lower), self.upper)

    def sample(self, rng):
        return self.from_unit(rng.uniform())


class UniformInteger(UniformFloat):
    def __init__(self, name, lower, upper, log=False):
        self.ilower, self.iupper = int(lower), int(upper)
        super().__init__(name, self.ilower - 0.49999, self.iupper + 0.49999, log)

    def to_unit(self, value):
        return super().to_unit(int(value))

    def from_unit(self, u):
        v = u * (self._hi - self._lo) + self._lo
        v = float(np.exp(v)) if self.log else float(v)
        return int(min(max(int(np.rint(v)), self.ilower), self.iupper))


class Categorical:
    def __init__(self, name, choices):
        self.name, self.choices = name, list(choices)
        assert len(self.choices) >= 1

    @property
    def size(self):
        return len(self.choices)

    def sample(self, rng):
        return self.choices[rng.randint(len(self.choices))]


class DenseSpace:
    """The dense view of a list of hyperparameters (sorted by name, as ConfigSpace stores them)."""

    def __init__(self, hyperparameters, seed=None):
        self.hyperparameters = sorted(hyperparameters, key=lambda hp: hp.name)
        names = [hp.name for hp in self.hyperparameters]
        assert len(set(names)) == len(names), "duplicate hyperparameter names"
        # types.py:66-91 (_get_mappings): source index -> target index (, size)
        self.nums, self.cats = [], []
        trg = 0
        for src, hp in enumerate(self.hyperparameters):
            if isinstance(hp, Categorical):
                self.cats.append((src, trg, hp.size))
                trg += hp.size
            elif isinstance(hp, UniformFloat):
                self.nums.append((src, trg))
                trg += 1
            else:
                raise NotImplementedError("Only hyperparameters of types `Categorical`, "
                                          "`UniformInteger`, `UniformFloat` are supported!")
        self.size_sparse, self.size_dense = len(self.hyperparameters), trg
        self.random_state = np.random.RandomState(seed)

    def get_dimensions(self, sparse=False):
        return self.size_sparse if sparse else self.size_dense

    def get_bounds(self):
        return Bounds(np.zeros(self.size_dense), np.ones(self.size_dense))

    def get_hyperparameter_by_idx(self, idx):
        return self.hyperparameters[idx].name

    # -- dictionary <-> sparse vector (what ConfigSpace's Configuration does) ----------------------
    def _sparse_from_dict(self, dct):
        missing = [hp.name for hp in self.hyperparameters if hp.name not in dct]
        assert not missing and len(dct) == self.size_sparse, f"configuration must name exactly {self.size_sparse} hyperparameters"
        v = np.empty(self.size_sparse)
        for i, hp in enumerate(self.hyperparameters):
            v[i] = hp.choices.index(dct[hp.name]) if isinstance(hp, Categorical) else hp.to_unit(dct[hp.name])
        return v

    def _dict_from_sparse(self, v):
        return {hp.name: (hp.choices[int(v[i])] if isinstance(hp, Categorical) else hp.from_unit(v[i]))
                for i, hp in enumerate(self.hyperparameters)}

    # -- DenseConfiguration.to_array / from_array (types.py:102-136) -------------------------------
    def to_array(self, dct, dtype="float64"):
        sparse = self._sparse_from_dict(dct)
        dense = np.zeros(self.size_dense, dtype=dtype)
        for src, trg in self.nums:
            dense[trg] = sparse[src]
        for src, trg, _ in self.cats:
            dense[trg + int(sparse[src])] = 1
        return dense

    def from_array(self, array_dense):
        array_dense = np.asarray(array_dense)
        assert array_dense.shape == (self.size_dense,)
        sparse = np.empty(self.size_sparse)
        for src, trg in self.nums:
            sparse[src] = array_dense[trg]
        for src, trg, size in self.cats:
            sparse[src] = np.argmax(array_dense[trg:trg + size])
        return self._dict_from_sparse(sparse)

    def sample_configuration(self, size=1):
        draw = lambda: {hp.name: hp.sample(self.random_state) for hp in self.hyperparameters}
        return draw() if size == 1 else [draw() for _ in range(size)]


def array_from_dict(space, dct):
    """bore/plugins/hpbandster/types.py:7-9."""
    return space.to_array(dct)


def dict_from_array(space, array):
    """bore/plugins/hpbandster/types.py:12-14."""
    return space.from_array(array)


def dense_space_from(configspace, seed=None):
    """A ``DenseSpace`` from a ``ConfigSpace.ConfigurationSpace`` (or anything shaped like one), as
    ``DenseConfigurationSpace(other, seed=...)`` takes one (bore/plugins/hpbandster/types.py:17-35):
    only the hyperparameters are carried over -- "conditions, clauses, seed, and other metadata
    ignored" (types.py:22-23).  Duck-typed, because ConfigSpace is not in this image: the space must
    offer ``get_hyperparameters()`` (or ``values()``), and every hyperparameter is recognised by its
    CLASS NAME -- ``UniformFloatHyperparameter`` (``lower``, ``upper``, ``log``),
    ``UniformIntegerHyperparameter`` (same), ``CategoricalHyperparameter`` (``choices``) -- anything
    else raises the reference's NotImplementedError (types.py:83-88).  A ``DenseSpace`` passes through."""
    if isinstance(configspace, DenseSpace):
        return configspace
    if hasattr(configspace, "get_hyperparameters"):
        hps = list(configspace.get_hyperparameters())
    elif hasattr(configspace, "values"):          # ConfigSpace >= 0.7: a Mapping name -> hyperparameter
        hps = list(configspace.values())
    else:
        raise TypeError(f"not a configuration space: {type(configspace).__name__} has neither "
                        "get_hyperparameters() nor values()")
    out = []
    for hp in hps:
        if isinstance(hp, (UniformFloat, Categorical)):       # already ours (UniformInteger is a UniformFloat)
            out.append(hp)
            continue
        kind = type(hp).__name__
        if kind == "CategoricalHyperparameter":
            out.append(Categorical(hp.name, list(hp.choices)))
        elif kind == "UniformIntegerHyperparameter":
            out.append(UniformInteger(hp.name, hp.lower, hp.upper, bool(getattr(hp, "log", False))))
        elif kind == "UniformFloatHyperparameter":
            out.append(UniformFloat(hp.name, hp.lower, hp.upper, bool(getattr(hp, "log", False))))
        else:
            raise NotImplementedError("Only hyperparameters of types `CategoricalHyperparameter`, "
                                      "`UniformIntegerHyperparameter`, `UniformFloatHyperparameter` "
                                      f"are supported! (got {kind} for {getattr(hp, 'name', '?')!r})")
    return DenseSpace(out, seed=seed)
