"""Batch de-duplication helpers (behaviour of bore/utils/deduplicate.py:8-45), used with
``argmax_batch`` to turn a batch of maximisers into ``size`` distinct, not yet evaluated
candidates."""
import numpy as np
from scipy.spatial.distance import cdist
from sklearn.utils import check_random_state

from ..optimizers.utils import from_bounds


def set_diff_2d(A, B, metric="euclidean", tol=1e-8):
    """Rows of A farther than ``tol`` from every row of B (bore/utils/deduplicate.py:8-16)."""
    far = np.greater(cdist(A, B, metric=metric), tol)
    return A[np.all(far, axis=-1)]


def pad_unique_random(A, size, bounds, B=None, metric="euclidean", tol=1e-8, random_state=None):
    """Unique rows of A that are not in B, topped up with uniform draws from the box until
    there are ``size`` of them (bore/utils/deduplicate.py:19-45).  The draws come from
    ``random_state`` in the reference's order: one ``uniform(size=(missing, dim))`` call per
    round, rounds repeating while a draw collides."""
    random_state = check_random_state(random_state)
    (low, high), dim = from_bounds(bounds)
    while True:
        A = np.unique(A, axis=0)
        if B is not None:
            A = set_diff_2d(A, B, metric=metric, tol=tol)
        missing = size - A.shape[0]
        if missing == 0:
            return A
        A = np.vstack((A, random_state.uniform(low=low, high=high, size=(missing, dim))))
