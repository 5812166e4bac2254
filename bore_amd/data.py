"""Observation store + label step of the BO loop (behaviour of bore/data.py:4-48).

``load_classification_data`` is the per-iteration label step that precedes ``fit``
(README.rst:89-90): the gamma-quantile of the targets (numpy's default linear
interpolation) splits the observations, with STRICT ``<`` so ties at the threshold are
negatives.
"""
import numpy as np


class Record:
    """Append-only (x, y[, budget]) log."""

    def __init__(self):
        self.features, self.targets, self.budgets = [], [], []

    def size(self):
        return len(self.targets)

    def append(self, x, y, b=None):
        self.features.append(x)
        self.targets.append(y)
        if b is not None:
            self.budgets.append(b)

    def load_feature_matrix(self):
        return np.vstack(self.features)

    def load_target_vector(self):
        return np.hstack(self.targets)

    def load_regression_data(self):
        return self.load_feature_matrix(), self.load_target_vector()

    def load_classification_data(self, gamma):
        """-> X (N, D) float64, z (N,) bool with z = y < quantile(y, gamma)."""
        X, y = self.load_regression_data()
        return X, classification_labels(y, gamma)

    def is_duplicate(self, x, rtol=1e-5, atol=1e-8):
        """True if ``x`` is allclose to any stored feature vector (the ``filter_fn`` hook of
        ``argmax``: bore/plugins/hpbandster/base.py:210-214)."""
        for prev in self.features:
            if np.allclose(prev, x, rtol=rtol, atol=atol):
                return True
        return False


def classification_labels(y, gamma):
    """z = y < np.quantile(y, gamma)   (bore/data.py:33-34)."""
    y = np.asarray(y)
    return np.less(y, np.quantile(y, q=gamma))
