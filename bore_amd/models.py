"""bore.models on MI355X: the classifier containers of the reference
(bore/models.py:9-45) with the Keras surface its call sites use
(README.rst:60-96; bore/plugins/hpbandster/base.py:145-194), backed by
libbore_hip.so.  ``Sequential`` stands in for ``tensorflow.keras.Sequential``:
same method names, argument meaning and defaults for the subset on the hot path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from . import shuffle as _shuffle
from .layers import Adam, BinaryCrossentropy, Dense, resolve_loss, resolve_optimizer
from .mixins import BatchMaximizableMixin, MaximizableMixin


class History:
    """What Keras ``fit`` returns: ``.history['loss']`` is the per-epoch logged loss.

    The losses are computed by the fit kernel either way; they are DOWNLOADED when first looked at.  ``fit`` itself
    no longer waits for its kernel (round 5): the caller's next lines -- README.rst:93-95 goes straight on to
    ``argmax`` -- run while the device fits, and a loop that never reads the history never pays for the copy."""

    def __init__(self, loss):
        self._loss, self._history = loss, None

    @property
    def history(self):
        if self._history is None:
            loss = self._loss() if callable(self._loss) else self._loss
            self._history = {"loss": [float(v) for v in loss]}
            self._loss = None
        return self._history

    @history.setter
    def history(self, value):        # (Keras' History.history is a plain attribute: callers may assign it)
        self._history, self._loss = value, None

    @property
    def epoch(self):
        return list(range(len(self.history["loss"])))


class Sequential:
    """Ordered stack of Dense layers living on one MI355X.

    State owned by the model across calls (reference: the live Keras object,
    bore/plugins/hpbandster/base.py:196-208): packed parameters ``theta`` in Keras
    ``get_weights()`` order, Adam slots ``m``, ``v`` and the iteration counter
    ``t`` -- so consecutive ``fit`` calls warm-start exactly like Keras.
    """

    def __init__(self, layers=None, name=None, seed=None, dtype_policy="float32"):
        """``dtype_policy``: "float32", or "mixed_bfloat16" (what
        ``tf.keras.mixed_precision.set_global_policy("mixed_bfloat16")`` does to a Keras model:
        float32 variables, bfloat16 compute; wide static shapes only, include/bore_hip.h enum
        bore_compute): fit, predict and the argmax objective all round like bfloat16 math;
        ``evaluate`` reads the float32 master weights."""
        if dtype_policy not in ("float32", "mixed_bfloat16"):
            raise ValueError(f"dtype_policy must be 'float32' or 'mixed_bfloat16', got {dtype_policy!r}")
        self.dtype_policy = dtype_policy
        self.name = name or "sequential"
        self.layers = []
        self._rs = np.random.RandomState(seed)
        self._shuffle_seed = int(self._rs.randint(0, 2**31 - 1)) if seed is None else int(seed)
        self._epochs_seen = 0
        self._desc = None
        self._input_dim = None
        self.theta = self.adam_m = self.adam_v = self.adam_t = None
        self._optimizer = Adam()
        self._loss = None
        self._metrics = []
        self._compiled = False
        for layer in layers or []:
            self.add(layer)

    # -- construction ----------------------------------------------------
    def add(self, layer):
        if not isinstance(layer, Dense):
            raise TypeError("only bore_amd.layers.Dense layers run on the HIP path")
        if self.theta is not None:
            raise RuntimeError("cannot add layers after the model has been built")
        if not self.layers and layer.input_dim is not None:
            self._input_dim = layer.input_dim
        self.layers.append(layer)
        if len(self.layers) > _lib.MAX_LAYERS:
            raise ValueError(f"at most {_lib.MAX_LAYERS} Dense layers are supported")

    @property
    def built(self):
        return self.theta is not None

    def build(self, input_dim=None):
        """Allocate + initialise the parameters (glorot_uniform kernels, zero biases)."""
        if self.built:
            return
        if not self.layers:
            raise RuntimeError("model has no layers")
        if input_dim is None:
            input_dim = self._input_dim
        if input_dim is None:
            raise RuntimeError("input dimension unknown: pass input_dim to the first Dense "
                               "layer or call the model on data first")
        dev = _lib.require_gpu()
        self._input_dim = int(input_dim)
        self._desc = _lib.make_desc(self._input_dim, [l.units for l in self.layers],
                                    [l.activation for l in self.layers],
                                    [l.l2_kernel for l in self.layers],
                                    [l.l2_bias for l in self.layers],
                                    compute="bfloat16" if self.dtype_policy == "mixed_bfloat16"
                                    else "float32")
        ws = []
        fan_in = self._input_dim
        for l in self.layers:
            limit = np.sqrt(6.0 / (fan_in + l.units))
            ws.append(self._rs.uniform(-limit, limit, size=(fan_in, l.units)).astype(np.float32))
            ws.append(np.zeros(l.units, dtype=np.float32))
            fan_in = l.units
        flat = np.concatenate([w.reshape(-1) for w in ws])
        P = ops.param_count(self._desc)
        assert flat.size == P
        self.theta = torch.from_numpy(flat).to(dev).reshape(1, P).contiguous()
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_t = torch.zeros(1, dtype=torch.int64, device=dev)

    def _ensure_built(self, x):
        if not self.built:
            self.build(np.shape(x)[-1])
        elif np.shape(x)[-1] != self._input_dim:
            raise ValueError(f"expected input dimension {self._input_dim}, got {np.shape(x)[-1]}")

    def compile(self, optimizer="adam", loss=None, metrics=None, **kwargs):
        self._optimizer = resolve_optimizer(optimizer)
        self._loss = resolve_loss(loss)
        self._metrics = list(metrics or [])
        for m in self._metrics:
            if m not in ("accuracy", "acc", "binary_accuracy"):
                raise NotImplementedError(f"metric {m!r} is not available on the HIP path")
        self._compiled = True

    def _check_loss(self):
        if not self._compiled or self._loss is None:
            raise RuntimeError("compile(optimizer=..., loss=...) the model before fit/evaluate")
        final = self.layers[-1]
        if final.units != 1:
            raise ValueError("the BORE classifier needs a single output unit")
        if self._loss.from_logits and final.activation != "linear":
            raise ValueError("BinaryCrossentropy(from_logits=True) needs a linear output layer")
        if not self._loss.from_logits and final.activation != "sigmoid":
            raise NotImplementedError(
                "binary_crossentropy on probabilities needs a sigmoid output layer (Keras then "
                "uses its logits); other output activations are not implemented")

    # -- data plumbing -----------------------------------------------------
    def _to_dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.theta.device, dtype=dtype).contiguous()
        a = np.ascontiguousarray(np.asarray(a), dtype={torch.float32: np.float32,
                                                      torch.float64: np.float64}[dtype])
        return torch.from_numpy(a).to(self.theta.device)

    # -- Keras surface -------------------------------------------------------
    def fit(self, x, y, batch_size=None, epochs=1, verbose=False, callbacks=None, shuffle=True,
            perm=None, **kwargs):
        """Keras ``fit`` (README.rst:93; bore/plugins/hpbandster/base.py:184).  Any ``batch_size``
        (more than 64 rows: 64-row sub-tiles inside one Adam step, same sums).  ``perm``
        (epochs, N) is an extension: explicit per-epoch shuffles (used by the parity tests);
        default draws them on the device.

        ``callbacks``: objects with any of Keras' ``set_model``, ``on_train_begin``,
        ``on_epoch_begin``, ``on_epoch_end(epoch, logs)`` (logs = {"loss": ...}),
        ``on_train_end``.  Without callbacks the whole fit is ONE kernel; with them it is one
        launch per epoch, so that ``model.stop_training = True`` (early stopping) takes effect at
        the epoch boundary as in Keras -- same arithmetic either way (Adam state and the shuffle
        stream carry over between launches).

        ASYNCHRONOUS without callbacks: the call returns once the kernel is enqueued on the model's stream (launch
        errors -- bad arguments, an unsupported shape -- are raised here; the library checks ``hipGetLastError`` after
        every launch).  The returned ``History`` keeps the device tensor of the losses and downloads it when
        ``.history`` is first read; an error the device reports while the kernel RUNS surfaces at the next call that
        waits for the stream (``.history``, ``predict``, ``argmax``, ``get_weights``), as with any stream-ordered API.
        ``verbose=True`` or callbacks make the call wait."""
        x = np.asarray(x) if not isinstance(x, torch.Tensor) else x
        self._ensure_built(x)
        self._check_loss()
        batch_size = 32 if batch_size is None else int(batch_size)
        if batch_size < 1:
            raise ValueError(f"batch_size must be positive, got {batch_size}")
        if not isinstance(x, torch.Tensor) and not isinstance(y, torch.Tensor):
            # host arrays (the reference's callers): features and labels cross PCIe in ONE copy
            xh = np.asarray(x, dtype=np.float32).reshape(-1, self._input_dim)
            N = xh.shape[0]
            # (through a pinned staging buffer the model keeps: the copy is asynchronous and half the cost of one
            # from pageable memory, tools/xfer_probe.py; the buffer is reused once its last copy has been seen done)
            n_both = N * (self._input_dim + 1)
            st = getattr(self, "_stage", None)
            if st is None or st[0].numel() < n_both:
                st = self._stage = (torch.empty(max(2 * n_both, 1024), dtype=torch.float32).pin_memory(), torch.cuda.Event())
            else:
                st[1].synchronize()
            both = st[0].numpy()[:n_both]
            both[:N * self._input_dim] = xh.ravel()
            both[N * self._input_dim:] = np.asarray(y).reshape(-1)     # (raises on a length mismatch)
            both = st[0][:n_both].to(self.theta.device, non_blocking=True)
            st[1].record()
            X = both[:N * self._input_dim].reshape(1, N, self._input_dim)
            z = both[N * self._input_dim:].reshape(1, N)
        else:
            X = self._to_dev(x, torch.float32).reshape(1, -1, self._input_dim)
            N = X.shape[1]
            z = self._to_dev(np.asarray(y).reshape(-1) if not isinstance(y, torch.Tensor)
                             else y.reshape(-1), torch.float32).reshape(1, N)
        epochs = int(epochs)
        if perm is None and not shuffle:
            perm = np.tile(np.arange(N, dtype=np.int32), (epochs, 1))
        if perm is not None:
            perm = np.asarray(perm)
            if perm.shape != (epochs, N) or not np.array_equal(np.sort(perm, axis=1),
                                                               np.tile(np.arange(N), (epochs, 1))):
                raise ValueError("perm must hold one permutation of range(N) per epoch")
            perm = torch.from_numpy(np.ascontiguousarray(perm, dtype=np.int32)) \
                .to(self.theta.device).reshape(1, epochs, N)
        o = self._optimizer

        def launch(e0, n_epochs):
            kw = dict(seed=self._shuffle_seed, epoch0=self._epochs_seen, lr=o.learning_rate,
                      beta1=o.beta_1, beta2=o.beta_2, eps=o.epsilon)
            try:
                loss = ops.mlp_fit(self._desc, self.theta, self.adam_m, self.adam_v, self.adam_t, X, z,
                                   n_epochs, batch_size,
                                   perm=None if perm is None else perm[:, e0:e0 + n_epochs].contiguous(), **kw)
            except _lib.NeedsPermError:         # (BORE_E_NEEDS_PERM: a code, not a message)
                if perm is not None:
                    raise
                # A data set too long for the device to draw its shuffles in LDS (the reference's fit
                # takes whatever the record holds, README.rst:93): the SAME stream from its host
                # statement (bore_amd.shuffle), a few epochs per launch, read from memory by the kernel.
                chunk = max(1, min(n_epochs, (64 << 20) // (4 * N)))
                parts = []
                for c0 in range(0, n_epochs, chunk):
                    ne = min(chunk, n_epochs - c0)
                    pc = _shuffle.permutations(self._shuffle_seed, 1, ne, N, epoch0=self._epochs_seen + c0)
                    pc = torch.from_numpy(pc).to(self.theta.device)
                    parts.append(ops.mlp_fit(self._desc, self.theta, self.adam_m, self.adam_v, self.adam_t,
                                             X, z, ne, batch_size, perm=pc, **kw))
                loss = torch.cat(parts, dim=1)
            self._epochs_seen += n_epochs
            return loss

        callbacks = list(callbacks or [])
        if not callbacks:
            loss = launch(0, epochs)                 # (enqueued: nothing waits for the kernel here)
            if verbose:
                for e, v in enumerate(loss[0].cpu().numpy()):
                    print(f"Epoch {e + 1}/{epochs} - loss: {float(v):.4f}")
            return History(lambda: loss[0].cpu().numpy())

        def call(name, *args):
            for cb in callbacks:
                fn = getattr(cb, name, None)
                if fn is not None:
                    fn(*args)

        self.stop_training = False
        call("set_model", self)
        call("on_train_begin", {})
        losses = []
        for e in range(epochs):
            call("on_epoch_begin", e, {})
            losses.append(float(launch(e, 1)[0].cpu().numpy()[0]))
            call("on_epoch_end", e, {"loss": losses[-1]})
            if self.stop_training:
                break
        call("on_train_end", {})
        return History(np.asarray(losses, dtype=np.float32))

    def evaluate(self, x, y, batch_size=None, verbose=False, **kwargs):
        """Keras ``evaluate``: loss, or [loss, accuracy] when compiled with metrics."""
        self._ensure_built(x)
        self._check_loss()
        X = self._to_dev(x, torch.float32).reshape(1, -1, self._input_dim)
        z = self._to_dev(np.asarray(y).reshape(-1), torch.float32).reshape(1, X.shape[1])
        loss, acc = ops.mlp_evaluate(self._desc, self.theta, X, z)
        if self._metrics:
            return [float(loss[0]), float(acc[0])]
        return float(loss[0])

    def predict(self, x, batch_size=None, verbose=0, **kwargs):
        """Keras ``predict``: (N, D) array -> (N, 1) float32 array."""
        self._ensure_built(x)
        X = self._to_dev(x, torch.float32).reshape(-1, self._input_dim)
        out = ops.mlp_forward(self._desc, self.theta, X)
        return out[0].cpu().numpy().reshape(-1, 1)

    def __call__(self, x, training=False):
        return self.predict(x)

    def get_weights(self):
        """Keras order [W1 (in,out), b1, W2, b2, ...] as float32 numpy arrays."""
        if not self.built:
            raise RuntimeError("model is not built yet")
        flat = self.theta[0].cpu().numpy()
        out, off, fan_in = [], 0, self._input_dim
        for l in self.layers:
            out.append(flat[off:off + fan_in * l.units].reshape(fan_in, l.units).copy())
            off += fan_in * l.units
            out.append(flat[off:off + l.units].copy())
            off += l.units
            fan_in = l.units
        return out

    def set_weights(self, weights):
        if not self.built:
            self.build(np.shape(weights[0])[0])
        cur = self.get_weights()
        if len(weights) != len(cur) or any(np.shape(a) != b.shape for a, b in zip(weights, cur)):
            raise ValueError("weight shapes do not match the model")
        flat = np.concatenate([np.asarray(w, dtype=np.float32).reshape(-1) for w in weights])
        self.theta.copy_(torch.from_numpy(flat).reshape(1, -1))

    def get_optimizer_state(self):
        """(m, v, t): Adam slots in Keras weight order + iteration counter (checkpointing)."""
        return (self.adam_m[0].cpu().numpy(), self.adam_v[0].cpu().numpy(), int(self.adam_t[0]))

    def set_optimizer_state(self, m, v, t):
        self.adam_m.copy_(torch.from_numpy(np.asarray(m, dtype=np.float32)).reshape(1, -1))
        self.adam_v.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(1, -1))
        self.adam_t.fill_(int(t))

    def count_params(self):
        fan_in, n = self._input_dim, 0
        for l in self.layers:
            n += (fan_in or 0) * l.units + l.units
            fan_in = l.units
        return n

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"')
        fan_in = self._input_dim
        for i, l in enumerate(self.layers):
            n = "?" if fan_in is None else fan_in * l.units + l.units
            print_fn(f" dense_{i} (Dense)  output (None, {l.units})  activation {l.activation}  "
                     f"params {n}")
            fan_in = l.units
        print_fn(f"Total params: {self.count_params() if self._input_dim else '?'}")


class DenseSequential(Sequential):
    """bore/models.py:9-21.  The reference's loop adds an input layer on ``i == 0`` and
    then falls through to the unconditional add, so ``num_layers`` yields
    ``num_layers + 1`` hidden layers; kept, as the reference's only
    in-repo caller (plugins/hpbandster/base.py:147-155) relies on whatever this builds."""

    def __init__(self, input_dim, output_dim, num_layers, num_units, layer_kws={},
                 final_layer_kws={}, **kwargs):
        super(DenseSequential, self).__init__(**kwargs)
        for i in range(num_layers):
            if not i:
                self.add(Dense(num_units, input_dim=input_dim, **layer_kws))
            self.add(Dense(num_units, **layer_kws))
        self.add(Dense(output_dim, **final_layer_kws))
        if self._input_dim is None:       # num_layers == 0
            self._input_dim = int(input_dim)


class Model(Sequential):
    """Stand-in for ``tensorflow.keras.Model`` restricted to a Dense stack (``layers=[...]``);
    the functional API (used only by the out-of-scope LSTM factory) is not provided."""


class MaximizableModel(MaximizableMixin, Model):
    pass


class MaximizableSequential(MaximizableMixin, Sequential):
    pass


class MaximizableDenseSequential(MaximizableMixin, DenseSequential):
    pass


class BatchMaximizableModel(BatchMaximizableMixin, Model):
    pass


class BatchMaximizableSequential(BatchMaximizableMixin, Sequential):
    pass


class BatchMaximizableDenseSequential(BatchMaximizableMixin, DenseSequential):
    pass


__all__ = ["Sequential", "DenseSequential", "Model", "MaximizableModel", "MaximizableSequential",
           "MaximizableDenseSequential", "BatchMaximizableModel", "BatchMaximizableSequential",
           "BatchMaximizableDenseSequential", "Dense", "BinaryCrossentropy", "Adam", "History"]
