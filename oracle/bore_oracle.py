"""CPU oracle for the BORE classifier hot path (fit + argmax).

TEST INFRASTRUCTURE ONLY.  Nothing under ``bore_amd/`` imports this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may.  It is the checker, never the thing shipped or measured.

What it restates (all citations relative to the reference checkout,
``/root/reference`` = ltiao/bore v1.5.0):

* model container            bore/models.py:9-33, README.rst:60-63
* objective ``T(-f(x))``      bore/mixins.py:16-20
* value-and-gradient bridge  bore/base.py:7-42, bore/decorators.py:24-79
* screening + restarts       bore/mixins.py:22-72
* argmax filter              bore/mixins.py:74-89
* bounds handling            bore/optimizers/utils.py:4-16
* label step                 bore/data.py:31-35, README.rst:89-90
* steps per epoch            bore/math.py:4-29
* epochs-from-steps rule     bore/plugins/hpbandster/base.py:166-170

The floating-point work itself lives in third-party dependencies that are NOT
under /root/reference: ``tensorflow==2.5.0`` (setup.py:42; Dense, activations,
``sigmoid_cross_entropy_with_logits``, ``ResourceApplyAdam``, ``Model.fit``)
and ``scipy==1.7.0`` (setup.py:15; L-BFGS-B).  TensorFlow cannot be imported in
the build container or on the GPU box, so the Keras half is a restatement of the
published TF 2.5 semantics:

  forward   a_l = act_l(a_{l-1} @ W_l + b_l),  W_l stored (in, out)   [Keras Dense]
  loss      mean_b( max(a,0) - a*z + log1p(exp(-|a|)) ) (+ sum l2*theta^2)
            [tf.nn.sigmoid_cross_entropy_with_logits; Keras routes a
             ``sigmoid`` output + "binary_crossentropy" through the cached
             logits, so the README form and the plugin's from_logits=True form
             share this formula]
  Adam      t += 1; alpha = lr*sqrt(1-b2^t)/(1-b1^t);
            m += (g-m)(1-b1); v += (g*g-v)(1-b2);
            theta -= alpha*m/(sqrt(v)+eps)    with eps=1e-7 OUTSIDE the bias
            correction; t, m, v persist across fit() calls  [ResourceApplyAdam]
  shuffle   one permutation of the N rows per epoch, consecutive slices of
            ``batch_size``; the last batch may be partial and still takes a
            step (bore/math.py:12-13).  TF's shuffle stream cannot be
            reproduced, so the permutation is an explicit input here.

PARITY UNPINNED for the Keras numerics: the reference holds no golden vector,
known-answer test or fixture for fit / predict / gradients; its
only test of this path, tests/test_models.py:12-50, is a property test which
this repository re-runs against this oracle (tests/test_oracle_golden.py, the
linear-net maximiser test) and against the HIP path (tests/test_gpu_models.py,
tests/test_gpu_argmax.py::test_reference_property_test_with_device_restarts).  What IS pinned against the reference itself (imported in the build
container, vectors committed under tests/golden/): the label step
(bore.data.Record), steps_per_epoch (bore.math), from_bounds
(bore.optimizers.utils).  The SciPy half is pinned by calling the real
third-party code (scipy 1.15.3 here vs 1.7.0 pinned by the reference).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import Bounds, OptimizeResult, minimize

ACTIVATIONS = ("linear", "relu", "elu", "sigmoid", "tanh")
TRANSFORMS = ("identity", "sigmoid", "exp")  # bore/plugins/hpbandster/base.py:18


# --------------------------------------------------------------------------
# side helpers restated from the reference's pure-numpy modules
# --------------------------------------------------------------------------
def ceil_divide(a, b):
    """bore/math.py:4-5."""
    return -np.floor_divide(-a, b)


def steps_per_epoch(dataset_size, batch_size):
    """bore/math.py:8-29: the last partial batch still takes a gradient step."""
    return int(ceil_divide(dataset_size, batch_size))


def epochs_from_steps(num_steps_per_iter, dataset_size, batch_size):
    """bore/plugins/hpbandster/base.py:166-170."""
    return num_steps_per_iter // steps_per_epoch(dataset_size, batch_size)


def labels(y, gamma):
    """bore/data.py:31-35 == README.rst:89-90: tau = quantile(y, gamma); z = y < tau."""
    y = np.asarray(y, dtype=np.float64)
    tau = np.quantile(y, q=gamma)
    return np.less(y, tau), tau


def from_bounds(bounds):
    """bore/optimizers/utils.py:4-16."""
    if isinstance(bounds, Bounds):
        low, high = bounds.lb, bounds.ub
        dim = len(low)
        assert dim == len(high), "lower and upper bounds sizes do not match!"
    else:
        low, high = zip(*bounds)
        dim = len(bounds)
    return (low, high), dim


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------
def dense_sequential_layout(input_dim, output_dim, num_layers, num_units,
                            activation="linear", final_activation="linear"):
    """Layer list produced by DenseSequential (bore/models.py:11-21).

    The reference's loop adds an input Dense on i == 0 and then FALLS THROUGH
    to the unconditional add, so ``num_layers`` hidden layers become
    ``num_layers + 1`` (bore/models.py:16-19).  Reproduced on purpose.
    """
    units, acts = [], []
    for i in range(num_layers):
        if not i:
            units.append(num_units)
            acts.append(activation)
        units.append(num_units)
        acts.append(activation)
    units.append(output_dim)
    acts.append(final_activation)
    return input_dim, units, acts


def glorot_uniform_params(input_dim, units, rs, dtype=np.float32):
    """Keras defaults: glorot_uniform kernels (limit = sqrt(6/(fan_in+fan_out))),
    zero biases.  Keras order [W1 (in,out), b1, W2, b2, ...].  The random stream
    is numpy's, not TF's (cannot be reproduced offline)."""
    params = []
    fan_in = input_dim
    for u in units:
        limit = np.sqrt(6.0 / (fan_in + u))
        params.append(rs.uniform(-limit, limit, size=(fan_in, u)).astype(dtype))
        params.append(np.zeros(u, dtype=dtype))
        fan_in = u
    return params


def _act(name, a):
    if name == "linear" or name is None:
        return a
    if name == "relu":
        return np.maximum(a, a.dtype.type(0))
    if name == "elu":
        return np.where(a > 0, a, np.expm1(np.minimum(a, a.dtype.type(0))))
    if name == "sigmoid":
        return _sigmoid(a)
    if name == "tanh":
        return np.tanh(a)
    raise ValueError(name)


def _act_grad_from_output(name, h):
    """d act / d pre-activation, written in terms of the activation OUTPUT h."""
    one = h.dtype.type(1)
    if name == "linear" or name is None:
        return np.ones_like(h)
    if name == "relu":
        return (h > 0).astype(h.dtype)
    if name == "elu":
        return np.where(h > 0, one, h + one)
    if name == "sigmoid":
        return h * (one - h)
    if name == "tanh":
        return one - h * h
    raise ValueError(name)


def _sigmoid(a):
    # numerically stable, dtype preserving
    e = np.exp(-np.abs(a))
    return np.where(a >= 0, 1 / (1 + e), e / (1 + e)).astype(a.dtype)


def forward(params, acts, X, return_all=False, logits=False):
    """Dense stack forward.  ``X`` (n, D) in the compute dtype.

    ``logits=True`` skips the final activation (used by the loss when the last
    layer is ``sigmoid``: Keras computes BCE from the cached logits)."""
    h = X
    hs = [h]
    n_layers = len(acts)
    for l in range(n_layers):
        W, b = params[2 * l], params[2 * l + 1]
        a = h @ W + b
        if l == n_layers - 1 and logits:
            h = a
        else:
            h = _act(acts[l], a)
        hs.append(h)
    return hs if return_all else h


def predict(params, acts, X, dtype=np.float32):
    """Keras ``predict``: float64 input is cast to the layer dtype; output (n, 1)."""
    p = [np.asarray(q, dtype=dtype) for q in params]
    return forward(p, acts, np.asarray(X, dtype=dtype))


def bce_with_logits(a, z):
    """tf.nn.sigmoid_cross_entropy_with_logits, per element."""
    zero = a.dtype.type(0)
    return np.maximum(a, zero) - a * z + np.log1p(np.exp(-np.abs(a)))


def loss_and_grads(params, acts, Xb, zb, l2=None):
    """Mean BCE (from logits) over the batch (+ l2 penalties) and d loss/d params.

    ``l2``: optional list of per-tensor factors aligned with ``params``
    (kernel_regularizer / bias_regularizer = l2(f): penalty f*sum(theta^2),
    bore/plugins/hpbandster/base.py:113-116)."""
    dt = Xb.dtype
    nb = Xb.shape[0]
    n_layers = len(acts)
    final_is_sigmoid = acts[-1] == "sigmoid"
    assert acts[-1] in ("sigmoid", "linear", None), \
        "BCE-from-logits needs a sigmoid or linear final layer"
    hs = forward(params, acts, Xb, return_all=True, logits=True)
    a = hs[-1]                                  # logits (nb, 1)
    zcol = zb.reshape(nb, 1).astype(dt)
    loss = bce_with_logits(a, zcol).mean(dtype=dt)
    delta = (_sigmoid(a) - zcol) / dt.type(nb)  # d loss / d logits
    grads = [None] * len(params)
    for l in range(n_layers - 1, -1, -1):
        W = params[2 * l]
        grads[2 * l] = hs[l].T @ delta
        grads[2 * l + 1] = delta.sum(axis=0)
        if l > 0:
            delta = (delta @ W.T) * _act_grad_from_output(acts[l - 1], hs[l])
    if l2 is not None:
        for i, f in enumerate(l2):
            if f:
                f = dt.type(f)
                loss = loss + f * np.sum(params[i] * params[i], dtype=dt)
                grads[i] = grads[i] + dt.type(2) * f * params[i]
    del final_is_sigmoid
    return loss, grads


class AdamState:
    """Slots of tf.keras.optimizers.Adam: iterations t, m, v (persist across fits)."""

    def __init__(self, params):
        self.t = 0
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]


def adam_alpha(t, lr, beta1, beta2, dtype=np.float32):
    """lr_t of Keras Adam._prepare_local, evaluated in ``dtype``.  The powers are
    taken in float64 on the float32-rounded betas and rounded once (what a
    correctly-rounded float32 pow returns)."""
    dt = np.dtype(dtype).type
    b1p = dt(np.power(np.float64(dt(beta1)), t))
    b2p = dt(np.power(np.float64(dt(beta2)), t))
    return dt(dt(lr) * np.sqrt(dt(1) - b2p) / (dt(1) - b1p))


def adam_step(params, grads, st, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
    """ResourceApplyAdam (non-nesterov), in place."""
    dt = params[0].dtype.type
    st.t += 1
    alpha = adam_alpha(st.t, lr, beta1, beta2, params[0].dtype)
    omb1 = dt(1) - dt(beta1)
    omb2 = dt(1) - dt(beta2)
    for p, g, m, v in zip(params, grads, st.m, st.v):
        m += (g - m) * omb1
        v += (g * g - v) * omb2
        p -= (m * alpha) / (np.sqrt(v) + dt(eps))


def fit(params, acts, st, X, z, perms, batch_size=64, lr=1e-3, beta1=0.9,
        beta2=0.999, eps=1e-7, l2=None, dtype=np.float32):
    """Keras ``fit(X, z, epochs=len(perms), batch_size)`` with explicit shuffles.

    ``perms``: (epochs, N) int array, one permutation of range(N) per epoch.
    Updates ``params`` / ``st`` in place; returns the per-epoch loss Keras
    would log (batch losses averaged with batch-size weights)."""
    Xc = np.asarray(X, dtype=dtype)
    zc = np.asarray(z).astype(dtype)
    N = Xc.shape[0]
    hist = []
    for perm in np.asarray(perms):
        tot = 0.0
        for s in range(0, N, batch_size):
            idx = perm[s:s + batch_size]
            loss, grads = loss_and_grads(params, acts, Xc[idx], zc[idx], l2=l2)
            adam_step(params, grads, st, lr, beta1, beta2, eps)
            tot += float(loss) * len(idx)
        hist.append(tot / N)
    return np.asarray(hist)


def bf16_round(x):
    """Round float32 values to the nearest bfloat16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32).reshape(np.shape(x))


def loss_and_grads_bf16(params, acts, Xb, zb):
    """The mixed-precision step bore_mlp_fit_bf16 defines (include/bore_hip.h): weights, biases,
    inputs, every layer output, the logits and every delta rounded to bfloat16; products and
    sums in float32; loss and d loss / d logit in float32.  Returns (loss, float32 grads)."""
    f32 = np.float32
    nb = Xb.shape[0]
    n_layers = len(acts)
    Wb = [bf16_round(params[2 * l]) for l in range(n_layers)]
    bb = [bf16_round(params[2 * l + 1]) for l in range(n_layers)]
    hs = [bf16_round(np.asarray(Xb, dtype=f32))]
    for l in range(n_layers):
        pre = hs[-1] @ Wb[l] + bb[l]
        a = pre if l == n_layers - 1 else _act(acts[l], pre)
        hs.append(bf16_round(a.astype(f32)))
    a = hs[-1]
    zcol = zb.reshape(nb, 1).astype(f32)
    loss = bce_with_logits(a, zcol).mean(dtype=f32)
    delta = bf16_round(((_sigmoid(a) - zcol) / f32(nb)).astype(f32))
    grads = [None] * len(params)
    for l in range(n_layers - 1, -1, -1):
        grads[2 * l] = (hs[l].T @ delta).astype(f32)
        grads[2 * l + 1] = delta.sum(axis=0, dtype=f32)
        if l > 0:
            delta = bf16_round(((delta @ Wb[l].T) * _act_grad_from_output(acts[l - 1], hs[l])).astype(f32))
    return loss, grads


def forward_bf16(params, acts, X, return_all=False):
    """Model output with bfloat16 weights / activations and float32 sums (enum bore_compute,
    BORE_COMPUTE_BF16): inputs and every layer output (the last one included) rounded."""
    f32 = np.float32
    hs = [bf16_round(np.asarray(X, dtype=f32))]
    for l, act in enumerate(acts):
        pre = hs[-1] @ bf16_round(params[2 * l]) + bf16_round(params[2 * l + 1])
        hs.append(bf16_round(_act(act, pre).astype(f32)))
    return hs if return_all else hs[-1]


def value_and_input_grad_bf16(params, acts, X, transform="identity", negate=True):
    """``x -> [T(+-f(x)), grad]`` in the mixed-precision arithmetic: the deltas of every layer
    are rounded to bfloat16, the input gradient itself is the float32 sum."""
    f32 = np.float32
    X = np.atleast_2d(np.asarray(X))
    hs = forward_bf16(params, acts, X.astype(f32), return_all=True)
    f = hs[-1]
    sign = f32(-1.0 if negate else 1.0)
    u = sign * f
    if transform == "sigmoid":
        T = _sigmoid(u); dT = T * (f32(1) - T)
    elif transform == "exp":
        T = np.exp(u); dT = T
    else:
        T = u; dT = np.ones_like(u)
    delta = bf16_round((sign * dT * _act_grad_from_output(acts[-1], f)).astype(f32))
    for l in range(len(acts) - 1, -1, -1):
        back = delta @ bf16_round(params[2 * l]).T
        delta = back if l == 0 else bf16_round((back * _act_grad_from_output(acts[l - 1], hs[l])).astype(f32))
    return T[:, 0].astype(f32), delta.astype(np.float64)


def fit_bf16(params, acts, st, X, z, perms, batch_size=64, lr=1e-3, beta1=0.9, beta2=0.999,
             eps=1e-7):
    """``fit`` with the mixed-precision step above; ``params`` are the float32 master weights."""
    Xc = np.asarray(X, dtype=np.float32)
    zc = np.asarray(z).astype(np.float32)
    N = Xc.shape[0]
    hist = []
    for perm in np.asarray(perms):
        tot = 0.0
        for s in range(0, N, batch_size):
            idx = perm[s:s + batch_size]
            loss, grads = loss_and_grads_bf16(params, acts, Xc[idx], zc[idx])
            adam_step(params, grads, st, lr, beta1, beta2, eps)
            tot += float(loss) * len(idx)
        hist.append(tot / N)
    return np.asarray(hist)


def evaluate(params, acts, X, z, dtype=np.float32, l2=None):
    """Keras ``evaluate``: mean BCE over all rows and ``metrics=["accuracy"]``.

    Keras resolves "accuracy" to binary_accuracy(threshold=0.5) applied to the
    model OUTPUT as is (so to raw logits under from_logits=True,
    bore/plugins/hpbandster/base.py:156-157)."""
    Xc = np.asarray(X, dtype=dtype)
    zc = np.asarray(z).astype(dtype).reshape(-1, 1)
    p = [np.asarray(q, dtype=dtype) for q in params]
    a = forward(p, acts, Xc, logits=True)
    out = _sigmoid(a) if acts[-1] == "sigmoid" else a
    loss = bce_with_logits(a, zc).mean(dtype=np.dtype(dtype).type)
    if l2 is not None:
        for q, f in zip(p, l2):
            if f:
                loss = loss + np.dtype(dtype).type(f) * np.sum(q * q)
    acc = np.mean((out > 0.5) == (zc > 0.5))
    return float(loss), float(acc)


# --------------------------------------------------------------------------
# acquisition
# --------------------------------------------------------------------------
def _transform(name, u):
    if name == "identity":
        return u, np.ones_like(u)
    if name == "sigmoid":
        s = _sigmoid(u)
        return s, s * (1 - s)
    if name == "exp":
        e = np.exp(u)
        return e, e
    raise ValueError(name)


def value_and_input_grad(params, acts, X, transform="identity", dtype=np.float32):
    """``convert(model, lambda u: transform(-u))`` (bore/mixins.py:20, bore/base.py:35-40).

    X: (D,) or (R, D) float64.  Returns [val, grad]: val in the network dtype
    (shape () or (R,)), grad float64 with the shape of X -- the dtype of the
    watched input (bore/decorators.py:54-61).  Rows are independent, so the
    (R, D) form is R stacked single-point evaluations."""
    X = np.asarray(X, dtype=np.float64)
    single = X.ndim == 1
    Xc = np.atleast_2d(X).astype(dtype)
    p = [np.asarray(q, dtype=dtype) for q in params]
    hs = forward(p, acts, Xc, return_all=True)
    f = hs[-1]                                   # (R, 1) model output
    val, dT = _transform(transform, -f)          # T(-f), T'(-f)
    delta = -dT * _act_grad_from_output(acts[-1], f)
    for l in range(len(acts) - 1, -1, -1):
        delta = delta @ p[2 * l].T
        if l > 0:
            delta = delta * _act_grad_from_output(acts[l - 1], hs[l])
    val = val[:, 0]
    grad = delta.astype(np.float64)
    if single:
        return [val[0], grad[0]]
    return [val, grad]


def maxima(params, acts, bounds, num_starts=5, num_samples=1024,
           method="L-BFGS-B", options=None, transform="identity",
           random_state=None, print_fn=lambda s: None, dtype=np.float32,
           X_init=None, compute="float32"):
    """bore/mixins.py:22-72, restated line for line in behaviour:
    uniform samples -> predict -> argpartition -> SEQUENTIAL scipy minimize per
    start with one single-point f/g call per evaluation.
    compute="bfloat16": the same loop over the mixed-precision statement of the network
    (forward_bf16 / value_and_input_grad_bf16), the arithmetic of a mixed_bfloat16 Keras model."""
    if options is None:
        options = dict(maxiter=1000, ftol=1e-9)
    if random_state is None or isinstance(random_state, (int, np.integer)):
        random_state = np.random.RandomState(random_state)
    assert num_samples is not None and num_samples > 0
    assert num_starts is not None and num_starts >= 0
    assert num_samples >= num_starts
    (low, high), dim = from_bounds(bounds)
    if X_init is None:
        X_init = random_state.uniform(low=low, high=high, size=(num_samples, dim))
    if compute == "bfloat16":
        z_init = forward_bf16(params, acts, X_init).squeeze(axis=-1)

        def func_min(x):
            v, g = value_and_input_grad_bf16(params, acts, x[None, :], transform, True)
            return [v[0], g[0]]
    else:
        assert compute == "float32", compute
        z_init = predict(params, acts, X_init, dtype=dtype).squeeze(axis=-1)
        func_min = lambda x: value_and_input_grad(params, acts, x, transform, dtype)
    f_init = -z_init
    results = []
    if num_starts > 0:
        ind = np.argpartition(f_init, kth=num_starts - 1, axis=None)
        for i in range(num_starts):
            res = minimize(func_min, x0=X_init[ind[i]], method=method, jac=True,
                           bounds=bounds, options=options)
            results.append(res)
            print_fn(f"[Maximum {i+1:02d}: value={res.fun:.3f}] success: {res.success}, "
                     f"iterations: {res.nit:02d}, status: {res.status} ({res.message})")
    else:
        i = np.argmin(f_init, axis=None)
        results.append(OptimizeResult(x=X_init[i], fun=f_init[i], success=True))
    return results


def argmax(params, acts, bounds, filter_fn=lambda res: True, **kwargs):
    """bore/mixins.py:74-89."""
    best = None
    for res in maxima(params, acts, bounds, **kwargs):
        if (res.success or res.status == 1) and filter_fn(res):
            if best is None or res.fun < best.fun:
                best = res
    return best
