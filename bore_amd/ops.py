"""Tensor-level operators over libbore_hip.so.

Every function takes CUDA (ROCm) torch tensors, checks shapes/dtypes on the host
before anything is launched (a kernel trusts its operands), enqueues ONE kernel
on the current torch stream and returns device tensors without synchronising.
A leading ``n_models`` dimension batches independent models (BO loops).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _chk(t, dtype, shape, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")


def param_count(desc):
    p = _lib.lib().bore_param_count(C.byref(desc))
    if p < 0:
        _lib.check(int(p))
    return int(p)


def mlp_forward(desc, theta, X, out=None):
    """theta [L,P] f32; X [L,N,D] f32 or [N,D] f32 shared by all models -> [L,N] f32."""
    L, P = theta.shape
    D = desc.input_dim
    shared = X.dim() == 2
    N = X.shape[-2]
    _chk(theta, torch.float32, (L, param_count(desc)), "theta")
    _chk(X, torch.float32, (N, D) if shared else (L, N, D), "X")
    if out is None:
        out = torch.empty((L, N), dtype=torch.float32, device=theta.device)
    else:
        _chk(out, torch.float32, (L, N), "out")
    if N == 0:
        return out
    _lib.check(_lib.lib().bore_mlp_forward(C.byref(desc), L, _lib.ptr(theta), _lib.ptr(X), N,
                                           int(shared), _lib.ptr(out), _lib.stream_ptr()))
    return out


def mlp_value_and_input_grad(desc, theta, X, transform="identity", negate=True, val=None,
                             grad=None):
    """theta [L,P] f32; X [L,R,D] f64 -> (val [L,R] f32, grad [L,R,D] f64) of T(+-f(x))."""
    L, P = theta.shape
    D = desc.input_dim
    _chk(theta, torch.float32, (L, param_count(desc)), "theta")
    if X.dim() != 3:
        raise ValueError("X: expected [n_models, n_rows, D]")
    R = X.shape[1]
    _chk(X, torch.float64, (L, R, D), "X")
    if val is None:
        val = torch.empty((L, R), dtype=torch.float32, device=theta.device)
    else:
        _chk(val, torch.float32, (L, R), "val")
    if grad is None:
        grad = torch.empty((L, R, D), dtype=torch.float64, device=theta.device)
    else:
        _chk(grad, torch.float64, (L, R, D), "grad")
    if transform not in _lib.TRANSFORM:
        raise ValueError(f"unknown transform {transform!r}")
    if R == 0:
        return val, grad
    _lib.check(_lib.lib().bore_mlp_value_and_input_grad(
        C.byref(desc), L, _lib.ptr(theta), _lib.ptr(X), R, _lib.TRANSFORM[transform],
        int(bool(negate)), _lib.ptr(val), _lib.ptr(grad), _lib.stream_ptr()))
    return val, grad


def mlp_fit(desc, theta, m, v, t, X, z, epochs, batch_size, perm=None, seed=0, model_index0=0,
            epoch0=0, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, want_loss=True):
    """In-place Keras-form fit of L models.  Returns epoch_loss [L,epochs] f32 (or None)."""
    L, P = theta.shape
    D = desc.input_dim
    _chk(theta, torch.float32, (L, param_count(desc)), "theta")
    _chk(m, torch.float32, (L, P), "adam_m")
    _chk(v, torch.float32, (L, P), "adam_v")
    _chk(t, torch.int64, (L,), "adam_t")
    if X.dim() != 3:
        raise ValueError("X: expected [n_models, N, D]")
    N = X.shape[1]
    _chk(X, torch.float32, (L, N, D), "X")
    _chk(z, torch.float32, (L, N), "z")
    epochs = int(epochs)
    if perm is not None:
        _chk(perm, torch.int32, (L, epochs, N), "perm")
        # an out-of-range row index would read outside X: validate before launching
        if epochs and N and (int(perm.min()) < 0 or int(perm.max()) >= N):
            raise ValueError("perm: entries must lie in [0, N)")
    loss = torch.empty((L, epochs), dtype=torch.float32, device=theta.device) if want_loss else None
    cfg = _lib.AdamCfg(lr, beta1, beta2, eps)
    _lib.check(_lib.lib().bore_mlp_fit(
        C.byref(desc), L, _lib.ptr(theta), _lib.ptr(m), _lib.ptr(v), _lib.ptr(t), _lib.ptr(X),
        _lib.ptr(z), N, epochs, int(batch_size), _lib.ptr(perm), C.c_uint64(seed & (2**64 - 1)),
        int(model_index0), int(epoch0), C.byref(cfg), _lib.ptr(loss), _lib.stream_ptr()))
    return loss


def mlp_evaluate(desc, theta, X, z):
    """-> (loss [L] f32, accuracy [L] f32)."""
    L, P = theta.shape
    D = desc.input_dim
    _chk(theta, torch.float32, (L, param_count(desc)), "theta")
    N = X.shape[1]
    _chk(X, torch.float32, (L, N, D), "X")
    _chk(z, torch.float32, (L, N), "z")
    loss = torch.empty(L, dtype=torch.float32, device=theta.device)
    acc = torch.empty(L, dtype=torch.float32, device=theta.device)
    _lib.check(_lib.lib().bore_mlp_evaluate(C.byref(desc), L, _lib.ptr(theta), _lib.ptr(X),
                                            _lib.ptr(z), N, _lib.ptr(loss), _lib.ptr(acc),
                                            _lib.stream_ptr()))
    return loss, acc


def shuffle_perm(seed, n_models, epochs, N, model_index0=0, epoch0=0, device=None):
    """The in-kernel shuffle stream of mlp_fit(perm=None), materialised: [L,epochs,N] int32."""
    device = device or _lib.require_gpu()
    perm = torch.empty((n_models, epochs, N), dtype=torch.int32, device=device)
    _lib.check(_lib.lib().bore_shuffle_perm(C.c_uint64(seed & (2**64 - 1)), int(model_index0),
                                            n_models, int(epoch0), epochs, N, _lib.ptr(perm),
                                            _lib.stream_ptr()))
    return perm
