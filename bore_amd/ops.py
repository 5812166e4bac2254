"""Tensor-level operators over libbore_hip.so.

Every function takes CUDA (ROCm) torch tensors, checks shapes/dtypes on the host
before anything is launched (a kernel trusts its operands), enqueues ONE kernel
on the current torch stream and returns device tensors without synchronising.
A leading ``n_models`` dimension batches independent models (BO loops).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

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
    """In-place Keras-form fit of L models.  Returns epoch_loss [L,epochs] f32 (or None).
    (desc.compute = bfloat16: mixed precision, include/bore_hip.h enum bore_compute.)"""
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


def labels(y, gamma, want_tau=False):
    """y [L,N] f64 -> z [L,N] f32 in {0,1} (z = y < np.quantile(y, gamma)); optionally tau [L]."""
    if y.dim() != 2:
        raise ValueError("y: expected [n_models, N]")
    L, N = y.shape
    _chk(y, torch.float64, (L, N), "y")
    z = torch.empty((L, N), dtype=torch.float32, device=y.device)
    tau = torch.empty(L, dtype=torch.float64, device=y.device) if want_tau else None
    _lib.check(_lib.lib().bore_labels(L, _lib.ptr(y), N, float(gamma), _lib.ptr(z), _lib.ptr(tau),
                                      _lib.stream_ptr()))
    return (z, tau) if want_tau else z


def _host_f64(a, D, name):
    import numpy as np
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (D,)))
    if a.shape != (D,):
        raise ValueError(f"{name}: expected {D} values")
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def uniform_candidates(seed, n_models, n_samples, low, high, model_index0=0, draw_index=0,
                       device=None):
    """X ~ U(low, high): [n_models, n_samples, D] f64 from the counter-based device stream
    (numpy statement: bore_amd.sampling.uniform_candidates)."""
    import numpy as np
    device = device or _lib.require_gpu()
    D = len(np.atleast_1d(low))
    lo, lo_p = _host_f64(low, D, "low")
    hi, hi_p = _host_f64(high, D, "high")
    X = torch.empty((n_models, n_samples, D), dtype=torch.float64, device=device)
    _lib.check(_lib.lib().bore_uniform_candidates(C.c_uint64(seed & (2**64 - 1)), int(model_index0),
                                                  n_models, int(draw_index), n_samples, D, lo_p,
                                                  hi_p, _lib.ptr(X), _lib.stream_ptr()))
    return X


def screen_topk(desc, theta, X_init, num_starts, want_pred=False):
    """Screening of ``maxima``: predict on X_init ([L,Ns,D] or shared [Ns,D], f64) and return
    (x0 [L,R,D] f64, idx [L,R] int32[, pred [L,Ns] f32]) for the R = num_starts best rows."""
    L, P = theta.shape
    D = desc.input_dim
    shared = X_init.dim() == 2
    Ns = X_init.shape[-2]
    _chk(theta, torch.float32, (L, param_count(desc)), "theta")
    _chk(X_init, torch.float64, (Ns, D) if shared else (L, Ns, D), "X_init")
    R = int(num_starts)
    x0 = torch.empty((L, R, D), dtype=torch.float64, device=theta.device)
    idx = torch.empty((L, R), dtype=torch.int32, device=theta.device)
    pred = torch.empty((L, Ns), dtype=torch.float32, device=theta.device) if (want_pred or Ns >= 1024) else None
    _lib.check(_lib.lib().bore_screen_topk(C.byref(desc), L, _lib.ptr(theta), _lib.ptr(X_init), Ns,
                                           int(shared), R, _lib.ptr(x0), _lib.ptr(idx),
                                           _lib.ptr(pred), _lib.stream_ptr()))
    return (x0, idx, pred) if want_pred else (x0, idx)


def sample_screen_topk(desc, theta, seed, n_samples, low, high, num_starts, model_index0=0,
                       draw_index=0, want_pred=False):
    """``uniform_candidates`` + ``screen_topk`` in one launch; the candidates are never written to
    memory (``bore_sample_screen_topk``).  Returns (x0 [L,R,D] f64, idx [L,R] int32[, pred])."""
    L = theta.shape[0]
    D = desc.input_dim
    _chk(theta, torch.float32, (L, param_count(desc)), "theta")
    lo, lo_p = _host_f64(low, D, "low")
    hi, hi_p = _host_f64(high, D, "high")
    R, Ns = int(num_starts), int(n_samples)
    x0 = torch.empty((L, R, D), dtype=torch.float64, device=theta.device)
    idx = torch.empty((L, R), dtype=torch.int32, device=theta.device)
    # (a prediction buffer lets the library spread the predictions of a few wide models over the
    # device: bore_argmax.hip, screen_body MODE 1 / 2; same results)
    pred = torch.empty((L, Ns), dtype=torch.float32, device=theta.device) if (want_pred or Ns >= 1024) else None
    _lib.check(_lib.lib().bore_sample_screen_topk(
        C.byref(desc), L, _lib.ptr(theta), C.c_uint64(seed & (2**64 - 1)), int(model_index0),
        int(draw_index), Ns, lo_p, hi_p, R, _lib.ptr(x0), _lib.ptr(idx), _lib.ptr(pred),
        _lib.stream_ptr()))
    return (x0, idx, pred) if want_pred else (x0, idx)


def lbfgsb_minimize(desc, theta, x0, low, high, transform="identity", negate=True, maxcor=10,
                    ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000, maxiter=15000,
                    maxls=20):
    """R bound-constrained L-BFGS-B minimisations of T(+-f(x)) per model, on the device.

    x0 [L,R,D] f64; low/high: length-D host sequences (+-inf for open sides).
    Returns (x [L,R,D] f64, fun [L,R] f64, jac [L,R,D] f64, info [L,R,5] int32) with
    info = (nit, nfev, status, task, message) as in scipy's OptimizeResult / task tables."""
    L, P = theta.shape
    D = desc.input_dim
    _chk(theta, torch.float32, (L, param_count(desc)), "theta")
    if x0.dim() != 3:
        raise ValueError("x0: expected [n_models, num_starts, D]")
    R = x0.shape[1]
    _chk(x0, torch.float64, (L, R, D), "x0")
    if transform not in _lib.TRANSFORM:
        raise ValueError(f"unknown transform {transform!r}")
    lo, lo_p = _host_f64(low, D, "low")
    hi, hi_p = _host_f64(high, D, "high")
    opts = _lib.LbfgsbOpts(int(maxcor), int(maxiter), int(maxfun), int(maxls), float(ftol),
                           float(gtol))
    # one allocation behind the four results (``lbfgsb_results_to_host`` brings them home in one copy)
    n_xd, n_i = L * R * D, (L * R * 5 + 1) // 2
    buf = torch.empty(2 * n_xd + L * R + n_i, dtype=torch.float64, device=theta.device)
    x = buf[:n_xd].view(L, R, D)
    jac = buf[n_xd:2 * n_xd].view(L, R, D)
    fun = buf[2 * n_xd:2 * n_xd + L * R].view(L, R)
    info = buf[2 * n_xd + L * R:].view(torch.int32)[:L * R * 5].view(L, R, 5)
    if R == 0:
        return x, fun, jac, info
    _lib.check(_lib.lib().bore_lbfgsb_minimize(
        C.byref(desc), L, _lib.ptr(theta), _lib.TRANSFORM[transform], int(bool(negate)),
        _lib.ptr(x0), R, lo_p, hi_p, C.byref(opts), _lib.ptr(x), _lib.ptr(fun), _lib.ptr(jac),
        _lib.ptr(info), _lib.stream_ptr()))
    return x, fun, jac, info


def lbfgsb_results_to_host(x, fun, jac, info):
    """The four results of ``lbfgsb_minimize`` as numpy arrays through ONE device-to-host copy (they are views
    of one allocation; four separate ``.cpu()`` calls are four synchronisations)."""
    L, R, D = x.shape
    n_xd = L * R * D
    total = 2 * n_xd + L * R + (L * R * 5 + 1) // 2
    packed = (x.dtype == torch.float64 and x.is_contiguous() and x.storage_offset() == 0
              and x.untyped_storage().nbytes() == 8 * total
              and all(t.untyped_storage().data_ptr() == x.untyped_storage().data_ptr() for t in (fun, jac, info)))
    if not packed:      # (not the views lbfgsb_minimize returns: one copy each)
        return tuple(t.cpu().numpy() for t in (x, fun, jac, info))
    h = torch.as_strided(x, (total,), (1,), 0).cpu().numpy()
    return (h[:n_xd].reshape(L, R, D), h[2 * n_xd:2 * n_xd + L * R].reshape(L, R),
            h[n_xd:2 * n_xd].reshape(L, R, D),
            h[2 * n_xd + L * R:].view(np.int32)[:L * R * 5].reshape(L, R, 5))


class ObservationStore:
    """Device-resident ``Record`` of n_models BO loops (bore/data.py:4-29): fp64 features and
    targets with room for ``cap`` rows per loop, appended to on the device
    (``bore_append_observations``), plus the dense views the label step and the fit read."""

    def __init__(self, n_models, D, cap, device=None):
        device = device or _lib.require_gpu()
        self.L, self.D, self.cap, self.n = int(n_models), int(D), int(cap), 0
        self.X = torch.empty((self.L, self.cap, self.D), dtype=torch.float64, device=device)
        self.y = torch.empty((self.L, self.cap), dtype=torch.float64, device=device)
        self.X32 = torch.empty(self.L * self.cap * self.D, dtype=torch.float32, device=device)
        self.y_dense = torch.empty(self.L * self.cap, dtype=torch.float64, device=device)

    def load(self, X, y):
        """Replace the contents with X [L, n, D], y [L, n] (host arrays or tensors)."""
        X = torch.as_tensor(X, dtype=torch.float64)
        y = torch.as_tensor(y, dtype=torch.float64)
        n = X.shape[1]
        if tuple(X.shape) != (self.L, n, self.D) or tuple(y.shape) != (self.L, n) or n > self.cap:
            raise ValueError("ObservationStore.load: expected X [L, n <= cap, D] and y [L, n]")
        self.X[:, :n].copy_(X)
        self.y[:, :n].copy_(y)
        self.n = n
        self.append(None, None)

    def grow(self, cap):
        """Reallocate for ``cap`` rows per loop, keeping the contents (synchronises)."""
        old = self
        new = ObservationStore(self.L, self.D, cap, device=self.X.device)
        new.X[:, :old.n].copy_(old.X[:, :old.n])
        new.y[:, :old.n].copy_(old.y[:, :old.n])
        new.n = old.n
        torch.cuda.synchronize()
        self.__dict__.update(new.__dict__)
        if self.n:
            # the dense views live in the new buffers.  The refresh runs on the CURRENT stream and
            # the caller may go on to use another one (ReplicaEngine appends on its group's
            # non-blocking stream): finish it before returning, as the docstring promises.
            self.append(None, None)
            torch.cuda.synchronize()

    def append(self, x_new, y_new):
        """Append one row per loop (x_new [L, D], y_new [L], device fp64; None = just refresh the
        dense views) on the current stream."""
        if x_new is not None:
            _chk(x_new, torch.float64, (self.L, self.D), "x_new")
            _chk(y_new, torch.float64, (self.L,), "y_new")
            if self.n + 1 > self.cap:
                raise ValueError("ObservationStore is full: grow() it first")
        _lib.check(_lib.lib().bore_append_observations(
            self.L, self.D, _lib.ptr(self.X), _lib.ptr(self.y), self.n, self.cap,
            _lib.ptr(x_new), _lib.ptr(y_new), _lib.ptr(self.X32), _lib.ptr(self.y_dense),
            _lib.stream_ptr()))
        if x_new is not None:
            self.n += 1

    def views(self):
        """(X32 [L, n, D] fp32, y [L, n] fp64): dense, as bore_labels / bore_mlp_fit take them."""
        n = self.n
        return (self.X32[:self.L * n * self.D].view(self.L, n, self.D),
                self.y_dense[:self.L * n].view(self.L, n))


def select_best(x, fun, info, store=None, rtol=1e-5, atol=1e-8):
    """``MaximizableMixin.argmax``'s pick over device L-BFGS-B results (bore/mixins.py:74-89), with
    ``Record.is_duplicate`` as filter_fn when ``store`` (an ObservationStore) is given.
    Returns (x_best [L, D] f64, best [L] int32: restart index or -1 = None)."""
    L, R, D = x.shape
    _chk(x, torch.float64, (L, R, D), "x")
    _chk(fun, torch.float64, (L, R), "fun")
    _chk(info, torch.int32, (L, R, 5), "info")
    x_best = torch.zeros((L, D), dtype=torch.float64, device=x.device)
    best = torch.empty(L, dtype=torch.int32, device=x.device)
    if store is not None and (store.L, store.D) != (L, D):
        raise ValueError("select_best: store does not match the results")
    _lib.check(_lib.lib().bore_select_best(
        L, R, D, _lib.ptr(x), _lib.ptr(fun), _lib.ptr(info),
        _lib.ptr(store.X) if store is not None else _lib.ptr(None),
        store.n if store is not None else 0, store.cap if store is not None else 0,
        float(rtol), float(atol), _lib.ptr(x_best), _lib.ptr(best), _lib.stream_ptr()))
    return x_best, best


def svgd_optimize(desc, theta, x_init, low=None, high=None, transform="identity", length_scale=None,
                  n_iter=1000, step_size=1e-3, alpha=.9, eps=1e-6, tau=1., c=1., lambd=None):
    """SVGD on transform(f(x)) for L models x n particles, all iterations in one launch
    (``bore_svgd_optimize``).  x_init [L, n, D] f64 -> particles [L, n, D] f64.  ``lambd`` selects
    DistortionExpDecay, otherwise DistortionConstant(c); low/high None = no clipping."""
    L = theta.shape[0]
    D = desc.input_dim
    _chk(theta, torch.float32, (L, param_count(desc)), "theta")
    if x_init.dim() != 3:
        raise ValueError("x_init: expected [n_models, n_particles, D]")
    n = x_init.shape[1]
    _chk(x_init, torch.float64, (L, n, D), "x_init")
    if transform not in _lib.TRANSFORM:
        raise ValueError(f"unknown transform {transform!r}")
    if (low is None) != (high is None):
        raise ValueError("low and high go together")
    null = C.POINTER(C.c_double)()
    lo_p = hi_p = null
    if low is not None:
        lo, lo_p = _host_f64(low, D, "low")
        hi, hi_p = _host_f64(high, D, "high")
    opts = _lib.SvgdOpts(int(n_iter), 0 if lambd is None else 1, float(step_size), float(alpha),
                         float(eps), float(tau), -1.0 if length_scale is None else float(length_scale),
                         float(c) if lambd is None else float(lambd))
    out = torch.empty_like(x_init)
    _lib.check(_lib.lib().bore_svgd_optimize(C.byref(desc), L, _lib.ptr(theta),
                                             _lib.TRANSFORM[transform], _lib.ptr(x_init), n, lo_p, hi_p,
                                             C.byref(opts), _lib.ptr(out), _lib.stream_ptr()))
    return out
