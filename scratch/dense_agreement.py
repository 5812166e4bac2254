"""(needs scratch/dense_form.patch applied for dense=1; with the committed header only form 0 runs)
Agreement of the two forms of lbfgsb.h (compact 2m x 2m / explicit B) with scipy on
(a) Rosenbrock and (b) the fp32 classifier objective of a TRAINED 16-16-1 network on Branin data
(config 1), restarts chosen as the reference does (top-3 of 1024 uniform samples) plus random ones."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scipy.optimize import Bounds, minimize, rosen, rosen_der
import lbfgsb_host as H
from oracle import bore_oracle as O

OPTS = dict(maxiter=1000, ftol=1e-9)
FORMS = (0, 1) if os.environ.get('DENSE') else (0,)

def stats(name, res):
    for form in FORMS:
        same = np.mean([(a.nit, a.nfev, a.status) == (b[form].nit, b[form].nfev, b[form].status)
                        and np.allclose(a.x, b[form].x, atol=1e-7) for a, b in res])
        df = np.array([abs(a.fun - b[form].fun) for a, b in res])
        dx = np.array([np.abs(a.x - b[form].x).max() for a, b in res])
        oka = np.mean([a.success or a.status == 1 for a, _ in res])
        okb = np.mean([b[form].success or b[form].status == 1 for _, b in res])
        print(f"{name:28s} {'dense  ' if form else 'compact'}: identical {same:5.1%}  med|dfun| {np.median(df):.1e} "
              f"p90 {np.percentile(df,90):.1e}  |dx|<1e-5 {np.mean(dx<1e-5):5.1%} <1e-3 {np.mean(dx<1e-3):5.1%}  ok scipy {oka:.2f} ours {okb:.2f}  "
              f"nit {np.mean([a.nit for a,_ in res]):.1f}/{np.mean([b[form].nit for _,b in res]):.1f} "
              f"nfev {np.mean([a.nfev for a,_ in res]):.1f}/{np.mean([b[form].nfev for _,b in res]):.1f}", flush=True)

def run(fun, x0, lb, ub):
    a = minimize(fun, x0, jac=True, method="L-BFGS-B", bounds=Bounds(lb, ub), options=OPTS)
    return a, tuple(H.minimize(fun, x0, (lb, ub), **(dict(dense=f) if os.environ.get('DENSE') else {}), **OPTS) for f in FORMS)

rs = np.random.RandomState(0)
for n in (2, 5, 8):
    res = []
    for _ in range(30):
        x0 = rs.uniform(-2, 2, size=n)
        res.append(run(lambda x: (rosen(x), rosen_der(x)), x0, np.full(n, -1.5), np.full(n, 2.0)))
    stats(f"rosen n={n}", res)

def branin01(X):
    x1, x2 = 15.0 * X[..., 0] - 5.0, 15.0 * X[..., 1]
    return ((x2 - 5.1 / (4 * np.pi ** 2) * x1 ** 2 + 5 / np.pi * x1 - 6) ** 2
            + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10)

acts = ["relu", "relu", "sigmoid"]
for N in (20, 60, 110):
    res = []
    for seed in range(12):
        rs = np.random.RandomState(seed)
        p = O.glorot_uniform_params(2, [16, 16, 1], rs)
        st = O.AdamState(p)
        X = rs.uniform(size=(N, 2)); y = branin01(X)
        z, _ = O.labels(y, 0.25)
        for rep in range(3):        # warm-started fits as in the BO loop
            perms = np.stack([rs.permutation(N) for _ in range(200)])
            O.fit(p, acts, st, X, z, perms, batch_size=64)
        fg = lambda x: tuple(O.value_and_input_grad(p, acts, x, "identity"))
        Xs = rs.uniform(size=(1024, 2))
        pred = O.forward(p, acts, Xs).reshape(-1)
        top = Xs[np.argsort(-pred)[:3]]
        for x0 in list(top) + list(rs.uniform(size=(3, 2))):
            res.append(run(fg, x0, np.zeros(2), np.ones(2)))
    stats(f"classifier 2D N={N}", res)
