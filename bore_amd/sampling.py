"""Host statement of the device candidate stream (bore_amd/csrc/bore_argmax.hip:
candidate_base, candidates_kernel).  ``maxima`` samples its candidates from numpy's
``RandomState`` on the host (bore/mixins.py:33,49) -- that is what the drop-in API keeps
doing.  Replica runs that never leave the device draw them from this counter-based
stream instead: element i of draw k of model m is

    u = (mix64(base(seed, m, k) + C*(i+1)) >> 11) * 2**-53,   x = low + (high - low) * u
"""
import numpy as np

from .shuffle import _M, _mix64

_TAG, _C_MODEL, _C_DRAW, _C_ELEM = (0xA0761D6478BD642F, 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03,
                                    0x8CB92BA72F3D8DD7)


def candidate_base(seed, model, draw):
    h = _mix64((seed ^ _TAG) & _M)
    h = _mix64((h + _C_MODEL * (model + 1)) & _M)
    return _mix64((h + _C_DRAW * (draw + 1)) & _M)


def uniform_candidates(seed, n_models, n_samples, low, high, model_index0=0, draw_index=0):
    low = np.atleast_1d(np.asarray(low, dtype=np.float64))
    high = np.atleast_1d(np.asarray(high, dtype=np.float64))
    D = low.size
    out = np.empty((n_models, n_samples, D))
    for m in range(n_models):
        base = candidate_base(seed, model_index0 + m, draw_index)
        r = np.array([_mix64((base + _C_ELEM * (i + 1)) & _M) >> 11 for i in range(n_samples * D)],
                     dtype=np.float64)
        u = (r * (1.0 / 9007199254740992.0)).reshape(n_samples, D)
        out[m] = low + (high - low) * u
    return out
