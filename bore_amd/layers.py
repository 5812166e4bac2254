"""Descriptors standing in for the Keras objects the reference's call sites pass.

The reference builds its classifier from ``tensorflow.keras.layers.Dense`` and
compiles it with ``optimizer="adam"`` / ``loss="binary_crossentropy"`` or
``BinaryCrossentropy(from_logits=True)`` (README.rst:60-66;
bore/plugins/hpbandster/base.py:147-157).  These are plain records of the same
names and keyword arguments; the arithmetic happens in libbore_hip.so.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

ACTIVATIONS = ("linear", "relu", "elu", "sigmoid", "tanh")


@dataclass(frozen=True)
class L2:
    """``tensorflow.keras.regularizers.l2(l2)``: penalty l2 * sum(w**2)."""
    l2: float = 0.01


def l2(l2=0.01):
    return L2(float(l2))


def _reg_factor(reg):
    if reg is None:
        return 0.0
    if isinstance(reg, L2):
        return reg.l2
    if isinstance(reg, (int, float)):
        return float(reg)
    raise TypeError(f"unsupported regularizer {reg!r}: only l2(factor) runs on the HIP path")


class Dense:
    """``Dense(units, activation=None, input_dim=None, kernel_regularizer=None,
    bias_regularizer=None)``.  Kernel initialiser glorot_uniform, bias zeros, use_bias
    always True (the Keras defaults the reference never overrides)."""

    def __init__(self, units, activation=None, input_dim=None, input_shape=None,
                 kernel_regularizer=None, bias_regularizer=None, name=None):
        units = int(units)
        if units < 1:
            raise ValueError("units must be a positive integer")
        if callable(activation) and hasattr(activation, "__name__"):
            activation = activation.__name__
        if activation is None:
            activation = "linear"
        if activation not in ACTIVATIONS:
            raise ValueError(f"activation {activation!r} is not supported on the HIP path "
                             f"(supported: {ACTIVATIONS})")
        if input_dim is None and input_shape is not None:
            input_dim = input_shape[-1]
        self.units = units
        self.activation = activation
        self.input_dim = None if input_dim is None else int(input_dim)
        self.l2_kernel = _reg_factor(kernel_regularizer)
        self.l2_bias = _reg_factor(bias_regularizer)
        self.name = name

    def __repr__(self):
        return f"Dense({self.units}, activation={self.activation!r})"


@dataclass
class BinaryCrossentropy:
    """``tensorflow.keras.losses.BinaryCrossentropy(from_logits=...)``."""
    from_logits: bool = False


@dataclass
class Adam:
    """``tensorflow.keras.optimizers.Adam`` (TF 2.5 defaults; epsilon is 1e-7, outside the
    bias correction -- NOT torch.optim.Adam's form)."""
    learning_rate: float = 1e-3
    beta_1: float = 0.9
    beta_2: float = 0.999
    epsilon: float = 1e-7


def resolve_optimizer(opt) -> Adam:
    if isinstance(opt, Adam):
        return opt
    if isinstance(opt, str) and opt.lower() == "adam":
        return Adam()
    raise NotImplementedError(f"optimizer {opt!r}: the HIP fit kernel implements Keras Adam only")


def resolve_loss(loss) -> Optional[BinaryCrossentropy]:
    if loss is None:
        return None
    if isinstance(loss, BinaryCrossentropy):
        return loss
    if isinstance(loss, str) and loss in ("binary_crossentropy", "bce"):
        return BinaryCrossentropy(from_logits=False)
    raise NotImplementedError(f"loss {loss!r}: the HIP fit kernel implements binary "
                              "cross-entropy only (the BORE classifier loss)")
