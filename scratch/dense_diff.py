import sys, os
import numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import lbfgsb_host as H
from oracle import bore_oracle as O
OPTS = dict(maxiter=1000, ftol=1e-9)
def branin01(X):
    x1, x2 = 15.0 * X[..., 0] - 5.0, 15.0 * X[..., 1]
    return ((x2 - 5.1 / (4 * np.pi ** 2) * x1 ** 2 + 5 / np.pi * x1 - 6) ** 2
            + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10)
acts = ["relu", "relu", "sigmoid"]
N = 20
shown = 0
for seed in range(12):
    rs = np.random.RandomState(seed)
    p = O.glorot_uniform_params(2, [16, 16, 1], rs)
    st = O.AdamState(p)
    X = rs.uniform(size=(N, 2)); y = branin01(X)
    z, _ = O.labels(y, 0.25)
    for rep in range(3):
        perms = np.stack([rs.permutation(N) for _ in range(200)])
        O.fit(p, acts, st, X, z, perms, batch_size=64)
    Xs = rs.uniform(size=(1024, 2))
    pred = O.forward(p, acts, Xs).reshape(-1)
    top = Xs[np.argsort(-pred)[:3]]
    for x0 in list(top) + list(rs.uniform(size=(3, 2))):
        tr = {}
        for form in (0, 1):
            log = []
            def fg(x):
                f, g = O.value_and_input_grad(p, acts, x, "identity")
                log.append((x.copy(), float(f), np.array(g, dtype=float)))
                return f, g
            r = H.minimize(fg, x0, (np.zeros(2), np.ones(2)), dense=form, **OPTS)
            tr[form] = (r, log)
        a, b = tr[0], tr[1]
        if (a[0].nit, a[0].nfev, a[0].status) != (b[0].nit, b[0].nfev, b[0].status) and shown < 4:
            shown += 1
            print("seed", seed, "x0", x0, "compact", a[0].nit, a[0].nfev, a[0].status, "dense", b[0].nit, b[0].nfev, b[0].status)
            for k in range(max(len(a[1]), len(b[1]))):
                xa = a[1][k] if k < len(a[1]) else None
                xb = b[1][k] if k < len(b[1]) else None
                same = xa is not None and xb is not None and np.array_equal(xa[0], xb[0])
                print(k, "same" if same else "DIFF", None if xa is None else (xa[0], xa[1], xa[2]), None if xb is None else (xb[0], xb[1]))
                if not same and k > 0:
                    break
