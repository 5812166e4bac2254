"""Experiment: how often does an L-BFGS-B evaluation request a point whose fp32 image equals that of
the last evaluated point or of the line search's base point?  (oracle model, host build of lbfgsb.h)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import bore_oracle as O
import lbfgsb_host as H

def branin01(X):
    x1 = 15 * X[:, 0] - 5; x2 = 15 * X[:, 1]
    return (x2 - 5.1 / (4 * np.pi ** 2) * x1 ** 2 + 5 / np.pi * x1 - 6) ** 2 + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10

tot = hit_last = hit_any = 0
for seed in range(6):
    rs = np.random.RandomState(seed)
    acts = ["relu", "relu", "sigmoid"]
    params = O.glorot_uniform_params(2, [16, 16, 1], rs)
    st = O.AdamState(params)
    X = rs.uniform(size=(10, 2)); y = branin01(X)
    for it in range(25):
        z = O.labels(y, 0.25)[0].astype(np.float32)
        N = len(y)
        perms = [rs.permutation(N) for _ in range(200)]
        O.fit(params, acts, st, X.astype(np.float32), z, perms)
        # three restarts from the best of 1024 samples
        Xs = rs.uniform(size=(1024, 2))
        v = O.predict(params, acts, Xs).ravel()
        idx = np.argsort(-v)[:3]
        best = None
        for x0 in Xs[idx]:
            seen = []
            def fun(x):
                f, g = O.value_and_input_grad(params, acts, x, transform="identity")
                seen.append(x.astype(np.float32).copy())
                return float(f), np.asarray(g, dtype=np.float64).ravel()
            r = H.minimize(fun, x0, (np.zeros(2), np.ones(2)), form=0)
            for k in range(1, len(seen)):
                tot += 1
                if np.array_equal(seen[k], seen[k - 1]): hit_last += 1
                if any(np.array_equal(seen[k], s) for s in seen[:k]): hit_any += 1
            if best is None or r.fun < best.fun: best = r
        xn = np.clip(best.x, 0, 1)
        X = np.vstack([X, xn]); y = np.append(y, branin01(xn[None, :]))
    print(seed, tot, hit_last, hit_any, flush=True)
print(f"evaluations {tot}: fp32 point equals the previous one {hit_last / tot:.3f}, equals any earlier one {hit_any / tot:.3f}")
