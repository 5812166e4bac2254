"""Callers of the hot path (bore/plugins)."""
from .classifier import ClassifierSuggester  # noqa: F401
from .types import (Categorical, DenseSpace, UniformFloat, UniformInteger,  # noqa: F401
                    array_from_dict, dense_space_from, dict_from_array)
from .hpbandster import BORE, ClassifierConfigGenerator  # noqa: F401
