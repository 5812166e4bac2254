import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from bore_amd import _lib, ops
from test_gpu_parity import dev, pack, rand_model
import lbfgsb_host as H
D, units, acts, tr, R = 6, [32, 32, 1], ["relu", "relu", "linear"], "sigmoid", 40
rs = np.random.RandomState(D)
desc = _lib.make_desc(D, units, acts)
L = 2
params = [rand_model(rs, D, units) for _ in range(L)]
th = dev(np.stack([pack(p) for p in params]))
X0 = rs.uniform(-0.1, 1.1, size=(L, R, D))
lo, hi = np.zeros(D), np.ones(D)
opts = dict(maxiter=1000, ftol=1e-9)
x, fun, jac, info = (t.cpu().numpy() for t in ops.lbfgsb_minimize(desc, th, dev(X0), lo, hi, tr, True, **opts))
bad = ~((x >= 0) & (x <= 1))
print("bad entries", bad.sum(), "nan", np.isnan(x).sum())
for l in range(L):
    for r in range(R):
        if bad[l, r].any():
            print(l, r, x[l, r], fun[l, r], info[l, r])
print(info[0, :, :3].T)
def fg_gpu(l, xx):
    v, g = ops.mlp_value_and_input_grad(desc, th[l:l+1], dev(np.atleast_2d(xx)[None]), tr, True)
    return v.cpu().numpy()[0, 0], g.cpu().numpy()[0, 0]
for r in range(R):
    h = H.minimize(lambda xx: fg_gpu(0, xx), X0[0, r], (lo, hi), **opts)
    ok = np.array_equal(h.x, x[0, r])
    if not ok: print("mismatch", r, h.x, x[0, r], (h.nit, h.nfev, h.status), info[0, r])
