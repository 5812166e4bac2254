"""Step-count arithmetic of the fit loop (behaviour of bore/math.py:4-29)."""
import numpy as np


def ceil_divide(a, b, *args, **kwargs):
    """Ceiling division through floor division of the negation (array-friendly)."""
    return -np.floor_divide(-a, b, *args, **kwargs)


def steps_per_epoch(dataset_size, batch_size):
    """Gradient steps in one pass over the data; a trailing partial batch still counts.

    >>> [steps_per_epoch(n, 64) for n in (32, 64, 100, 1000)]
    [1, 1, 2, 16]
    """
    return int(ceil_divide(dataset_size, batch_size))


def epochs_per_iteration(num_steps_per_iter, dataset_size, batch_size):
    """The plugin's rule when no epoch count is given: as many whole epochs as fit in
    ``num_steps_per_iter`` steps (bore/plugins/hpbandster/base.py:166-170)."""
    return num_steps_per_iter // steps_per_epoch(dataset_size, batch_size)
