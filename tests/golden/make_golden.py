#!/usr/bin/env python
"""Regenerate the committed golden vectors.  Runs ONLY in the build container
(needs /root/reference); the GPU box and the test-suite read the .npz/.json
files, never the reference.

Two kinds of vector:

* ``ref_*.npz / ref_*.json`` -- outputs of the REFERENCE ITSELF, obtained by
  importing the modules of ltiao/bore that import without TensorFlow
  (bore.data.Record, bore.math.steps_per_epoch, bore.optimizers.utils.from_bounds).
  These pin the oracle's restatement of the label step / step count / bounds.
* ``mlp_*.npz`` -- float64 outputs of oracle/bore_oracle.py on fixed inputs.
  They are regression vectors for the Keras half (TensorFlow cannot be run, so
  there is nothing of the reference's to record: parity unpinned, see the
  oracle header); their gradients are cross-checked against central finite
  differences before being written.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")


def ref_vectors():
    from scipy.optimize import Bounds
    from bore.data import Record                      # reference
    from bore.math import steps_per_epoch             # reference
    from bore.optimizers.utils import from_bounds     # reference

    out = {}
    cases = []
    rs = np.random.RandomState(0)
    for n, gamma, d, ties in [(9, 0.25, 2, False), (10, 0.25, 2, False), (64, 0.25, 2, False),
                              (65, 0.25, 2, False), (110, 0.25, 2, False), (30, 1 / 3, 6, False),
                              (17, 0.5, 3, True), (256, 0.25, 6, False), (4, 0.25, 1, True),
                              (1, 0.25, 2, False), (2, 0.25, 2, False), (100, 0.1, 16, True)]:
        X = rs.uniform(size=(n, d))
        y = rs.normal(size=n)
        if ties:
            y = np.round(y, 0)
        rec = Record()
        for xi, yi in zip(X, y):
            rec.append(x=xi, y=yi)
        Xo, z = rec.load_classification_data(gamma)
        k = len(cases)
        out[f"X{k}"] = X
        out[f"y{k}"] = y
        out[f"z{k}"] = z
        out[f"Xo{k}"] = Xo
        # duplicates: one exact row, one within rtol, one outside
        probes = np.stack([X[0], X[-1] * (1 + 5e-6), X[0] + 1e-3])
        out[f"probe{k}"] = probes
        out[f"dup{k}"] = np.array([rec.is_duplicate(p) for p in probes])
        cases.append(dict(n=n, gamma=gamma, d=d))
    np.savez(os.path.join(HERE, "ref_labels.npz"), **out)

    spe = [(n, b, steps_per_epoch(n, b)) for n in (1, 10, 32, 63, 64, 65, 100, 110, 128, 129, 1000, 1024)
           for b in (1, 32, 64, 100)]
    fb = []
    for bounds in ([(0.0, 1.0)] * 2, [(-5.0, 10.0), (0.0, 15.0)], [(0.0, 1.0)] * 6):
        (lo, hi), dim = from_bounds(bounds)
        (lo2, hi2), dim2 = from_bounds(Bounds(lb=np.array(lo), ub=np.array(hi)))
        fb.append(dict(bounds=bounds, low=list(lo), high=list(hi), dim=dim,
                       low_b=list(map(float, lo2)), high_b=list(map(float, hi2)), dim_b=dim2))
    with open(os.path.join(HERE, "ref_misc.json"), "w") as f:
        json.dump(dict(label_cases=cases, steps_per_epoch=spe, from_bounds=fb), f, indent=1)


def mlp_vectors():
    from oracle import bore_oracle as O

    f64 = np.float64
    specs = [
        ("branin", 2, [16, 16, 1], ["relu", "relu", "sigmoid"], "identity", None),
        ("hartmann", 6, [32, 32, 1], ["relu", "relu", "linear"], "sigmoid", None),
        ("plugin", 3, [32, 32, 32, 1], ["elu", "elu", "elu", "linear"], "exp", 1e-4),
        ("hpo16", 16, [64, 64, 64, 1], ["tanh", "relu", "elu", "linear"], "sigmoid", None),
    ]
    out = {}
    meta = {}
    for name, D, units, acts, transform, l2f in specs:
        rs = np.random.RandomState(42)
        params = O.glorot_uniform_params(D, units, rs, dtype=f64)
        for i in range(1, len(params), 2):       # non-zero biases exercise more paths
            params[i] = rs.normal(scale=0.1, size=params[i].shape)
        N, B, E = 100, 64, 3
        X = rs.uniform(size=(N, D))
        z = rs.uniform(size=N) < 0.3
        perms = np.stack([rs.permutation(N) for _ in range(E)])
        l2 = None if l2f is None else [l2f] * (len(params) - 2) + [0.0, 0.0]
        p0 = [p.copy() for p in params]
        # loss / grads on the first batch + finite differences
        idx = perms[0][:B]
        loss, grads = O.loss_and_grads(params, acts, X[idx], z[idx].astype(f64), l2=l2)
        h = 1e-6
        for ti in range(len(params)):
            flat = params[ti].reshape(-1)
            for j in rs.choice(flat.size, size=min(5, flat.size), replace=False):
                old = flat[j]
                flat[j] = old + h
                lp, _ = O.loss_and_grads(params, acts, X[idx], z[idx].astype(f64), l2=l2)
                flat[j] = old - h
                lm, _ = O.loss_and_grads(params, acts, X[idx], z[idx].astype(f64), l2=l2)
                flat[j] = old
                fd = (lp - lm) / (2 * h)
                assert abs(fd - grads[ti].reshape(-1)[j]) < 1e-7 * max(1, abs(fd)), (name, ti, j)
        # fit trajectory
        st = O.AdamState(params)
        hist = O.fit(params, acts, st, X, z, perms, batch_size=B, l2=l2, dtype=f64)
        # value + input gradient at the trained weights
        Xq = rs.uniform(size=(37, D))
        val, grad = O.value_and_input_grad(params, acts, Xq, transform, dtype=f64)
        for r in range(3):
            for j in range(D):
                e = np.zeros(D)
                e[j] = h
                vp, _ = O.value_and_input_grad(params, acts, Xq[r] + e, transform, dtype=f64)
                vm, _ = O.value_and_input_grad(params, acts, Xq[r] - e, transform, dtype=f64)
                fd = (vp - vm) / (2 * h)
                assert abs(fd - grad[r, j]) < 1e-6 * max(1, abs(fd)), (name, r, j)
        pred = O.predict(params, acts, Xq, dtype=f64)
        ev = O.evaluate(params, acts, X, z, dtype=f64, l2=l2)
        out.update({f"{name}_X": X, f"{name}_z": z, f"{name}_perms": perms,
                    f"{name}_loss0": np.array(loss), f"{name}_hist": hist,
                    f"{name}_Xq": Xq, f"{name}_val": val, f"{name}_grad": grad,
                    f"{name}_pred": pred, f"{name}_eval": np.array(ev),
                    f"{name}_t": np.array(st.t)})
        for i, (a, b, g, m, v) in enumerate(zip(p0, params, grads, st.m, st.v)):
            out[f"{name}_p0_{i}"] = a
            out[f"{name}_p1_{i}"] = b
            out[f"{name}_g0_{i}"] = g
            out[f"{name}_m_{i}"] = m
            out[f"{name}_v_{i}"] = v
        meta[name] = dict(D=D, units=units, acts=acts, transform=transform, l2=l2f,
                          N=N, batch=B, epochs=E)
    np.savez_compressed(os.path.join(HERE, "mlp_golden.npz"), **out)
    with open(os.path.join(HERE, "mlp_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)


def svgd_vectors():
    """Trajectories of the REFERENCE's SVGD (bore.optimizers.svgd imports without TensorFlow)
    on a fixed float64 objective: the oracle MLP's T(f) value/gradient."""
    from bore.optimizers.svgd.base import SVGD, DistortionConstant, DistortionExpDecay, rank
    from bore.optimizers.svgd.kernels import RadialBasis
    from oracle import bore_oracle as O
    out = {}
    rs = np.random.RandomState(7)
    D, units, acts = 3, [16, 16, 1], ["tanh", "relu", "linear"]
    params = O.glorot_uniform_params(D, units, rs, dtype=np.float64)

    def func(X):      # T(f) with T = sigmoid: negate the oracle's T(-f) convention
        q = [p.copy() for p in params]
        q[-2] = -q[-2]; q[-1] = -q[-1]                 # f -> -f on a linear output layer
        v, g = O.value_and_input_grad(q, acts, X, "sigmoid", dtype=np.float64)
        return v, g

    for k, (n, ls, lambd, n_iter) in enumerate([(4, 0.5, None, 50), (16, None, None, 50),
                                                 (16, 1.0, 1.0, 30), (64, None, 0.5, 20)]):
        x0 = rs.uniform(size=(n, D))
        dist = DistortionConstant() if lambd is None else DistortionExpDecay(lambd=lambd)
        svgd = SVGD(kernel=RadialBasis(length_scale=ls), n_iter=n_iter, step_size=1e-2,
                    distortion=dist)
        out[f"x0_{k}"] = x0
        out[f"x_{k}"] = svgd.optimize_from_init(func, x0, bounds=[(0.0, 1.0)] * D)
        K, Kg = RadialBasis(length_scale=ls).value_and_grad(x0)
        out[f"K_{k}"], out[f"Kg_{k}"] = K, Kg
        out[f"rank_{k}"] = rank(func(x0)[0])
        out[f"cfg_{k}"] = np.array([n, -1 if ls is None else ls, -1 if lambd is None else lambd, n_iter])
    for i, p in enumerate(params):
        out[f"p_{i}"] = p
    np.savez_compressed(os.path.join(HERE, "ref_svgd.npz"), **out)



if __name__ == "__main__":
    ref_vectors()
    mlp_vectors()
    svgd_vectors()
    print("golden vectors written to", HERE)
