"""Helpers around the hot path (bore/utils)."""
