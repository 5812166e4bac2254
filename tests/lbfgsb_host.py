"""ctypes driver for the host build of bore_amd/csrc/lbfgsb.h (test scaffolding)."""
import ctypes as C
import os
import subprocess

import numpy as np
from scipy.optimize import OptimizeResult

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "lbfgsb_host.cpp")
HDR = os.path.join(os.path.dirname(HERE), "bore_amd", "csrc", "lbfgsb.h")
SO = os.path.join(HERE, "native", "liblbfgsb_host.so")
_CB = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
_lib = None


def lib():
    global _lib
    if _lib is None:
        if (not os.path.exists(SO)
                or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR))):
            subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                            SRC, "-o", SO], check=True)
        _lib = C.CDLL(SO)
        _lib.lbfgsb_host_minimize.restype = C.c_int
        _lib.lbfgsb_host_minimize_form.restype = C.c_int
    return _lib


def minimize(fun, x0, bounds, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000,
             maxiter=15000, maxls=20, form=0):
    """fun(x) -> (f, g).  bounds: (lb, ub) arrays with +-inf for open sides.
    form: 0 reverse communication, 1 the routine calls the evaluation (DIRECT), 2 DIRECT with the
    two-variable line search in registers (lbfgsb.h)."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    n = x0.size
    lb, ub = (np.broadcast_to(np.asarray(b, dtype=np.float64), n).copy() for b in bounds)
    nbd = np.zeros(n, dtype=np.int32)
    for i in range(n):
        lo, up = np.isfinite(lb[i]), np.isfinite(ub[i])
        nbd[i] = 2 if lo and up else 1 if lo else 3 if up else 0
    l = np.where(np.isfinite(lb), lb, 0.0)
    u = np.where(np.isfinite(ub), ub, 0.0)

    def cb(n_, xp, fp, gp):
        x = np.ctypeslib.as_array(xp, shape=(n_,)).copy()
        f, g = fun(x)
        fp[0] = float(f)
        g = np.asarray(g, dtype=np.float64)
        for i in range(n_):
            gp[i] = g[i]

    xo, go = np.empty(n), np.empty(n)
    fo = C.c_double()
    oi = np.zeros(5, dtype=np.int32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    lib().lbfgsb_host_minimize_form(int(form), n, maxcor, dp(x0), dp(l), dp(u), nbd.ctypes.data_as(C.POINTER(C.c_int)),
                               C.c_double(ftol / np.finfo(float).eps), C.c_double(gtol), maxiter,
                               maxfun, maxls, _CB(cb), dp(xo), C.byref(fo), dp(go),
                               oi.ctypes.data_as(C.POINTER(C.c_int)))
    return OptimizeResult(x=xo, fun=fo.value, jac=go, nit=int(oi[0]), nfev=int(oi[1]),
                          status=int(oi[2]), success=oi[2] == 0, task=(int(oi[3]), int(oi[4])))
