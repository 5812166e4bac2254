"""Host statement of the in-kernel shuffle stream (bore_amd/csrc/mlp_device.h: mix64,
shuffle_base, shuffle_key, make_perm).  Keras reshuffles the rows every epoch from an
RNG stream that cannot be reproduced outside TensorFlow; this build draws each epoch's
permutation from a counter-based stream keyed by (seed, model index, epoch index) so
that any epoch of any model can be regenerated independently, on the device or here.
"""
import numpy as np

_M = (1 << 64) - 1
_C_MODEL, _C_EPOCH, _C_ROW = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03, 0x8CB92BA72F3D8DD7


def _mix64(z):
    z &= _M
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & _M
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & _M
    z ^= z >> 31
    return z


def shuffle_base(seed, model, epoch):
    h = _mix64((seed + _C_MODEL * (model + 1)) & _M)
    return _mix64((h + _C_EPOCH * (epoch + 1)) & _M)


def _mix64_array(z):
    """_mix64 on a uint64 array (numpy's unsigned arithmetic wraps modulo 2**64)."""
    z = z ^ (z >> np.uint64(30))
    z = z * np.uint64(0xBF58476D1CE4E5B9)
    z = z ^ (z >> np.uint64(27))
    z = z * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def shuffle_keys(seed, model, epoch, N):
    base = shuffle_base(seed, model, epoch)
    i = np.arange(1, N + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(base) + np.uint64(_C_ROW) * i
        return (_mix64_array(z) >> np.uint64(32)).astype(np.uint32)


def epoch_permutation(seed, model, epoch, N):
    """Rows listed by ascending (key, row index)."""
    return np.argsort(shuffle_keys(seed, model, epoch, N), kind="stable").astype(np.int32)


def permutations(seed, n_models, epochs, N, model_index0=0, epoch0=0):
    """[n_models, epochs, N] int32 -- what ``ops.shuffle_perm`` returns."""
    out = np.empty((n_models, epochs, N), dtype=np.int32)
    for m in range(n_models):
        for e in range(epochs):
            out[m, e] = epoch_permutation(seed, model_index0 + m, epoch0 + e, N)
    return out
