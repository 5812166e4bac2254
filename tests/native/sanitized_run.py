"""Driven by tests/test_lbfgsb_host.py::test_host_build_is_clean_under_asan_and_ubsan in a subprocess with
libasan/libubsan preloaded: many L-BFGS-B runs of the ASan+UBSan build of bore_amd/csrc/lbfgsb.h."""
import sys, os, ctypes
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import lbfgsb_host as H
H.SO = sys.argv[1]
H._lib = ctypes.CDLL(H.SO); H._lib.lbfgsb_host_minimize.restype = ctypes.c_int
import numpy as np
from scipy.optimize import rosen, rosen_der
from oracle import bore_oracle as O
rs = np.random.RandomState(0)
n_runs = 0
for n in (1, 2, 3, 5, 10, 31):
    for box in ("wide", "active", "free"):
        for _ in range(3):
            x0 = rs.uniform(-2, 2, size=n)
            lb, ub = np.full(n, -1.5), np.full(n, 2.0)
            if box == "active": ub[:] = 0.8
            if box == "free": lb[:], ub[:] = -np.inf, np.inf
            f = (lambda x: (rosen(x), rosen_der(x))) if n > 1 else (lambda x: (float((x[0]-0.3)**2), 2*(x-0.3)))
            for mc in (1, 3, 10, 17):
                r = H.minimize(f, x0, (lb, ub), maxiter=200, ftol=1e-9, maxcor=mc)
                n_runs += 1
p = O.glorot_uniform_params(2, [16, 16, 1], rs)
fg = lambda x: tuple(O.value_and_input_grad(p, ["relu", "relu", "sigmoid"], x, "identity"))
for _ in range(40):
    H.minimize(fg, rs.uniform(-0.1, 1.1, size=2), (np.zeros(2), np.ones(2)), maxiter=1000, ftol=1e-9); n_runs += 1
print("sanitized runs ok:", n_runs)
