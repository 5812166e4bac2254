"""bore_amd -- the BORE density-ratio classifier hot path (MLP fit + multi-start argmax)
on AMD Instinct MI355X (gfx950): hand-written HIP kernels behind the reference's
``bore.models`` / ``MaximizableSequential`` API surface (ltiao/bore v1.5.0).

Importing the package does not need a GPU; computing anything does (no CPU fallback).
"""
__version__ = "0.1.0"

from . import transforms  # noqa: F401
from .layers import Adam, BinaryCrossentropy, Dense, l2  # noqa: F401
from .transforms import TRANSFORMS  # noqa: F401
