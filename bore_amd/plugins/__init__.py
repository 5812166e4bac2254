"""Callers of the hot path (bore/plugins)."""
from .classifier import ClassifierSuggester  # noqa: F401
