"""bore_amd/csrc/lbfgsb.h (the in-kernel L-BFGS-B state machine), compiled for the host
with g++, against the third-party code the reference calls:
scipy.optimize.minimize(method="L-BFGS-B") (bore/mixins.py:59-60)."""
import numpy as np
import pytest
from scipy.optimize import Bounds, minimize, rosen, rosen_der

import lbfgsb_host as H
from oracle import bore_oracle as O

OPTS = dict(maxiter=1000, ftol=1e-9)       # bore/mixins.py:23


def both(fun, x0, lb, ub, **kw):
    a = minimize(fun, x0, jac=True, method="L-BFGS-B", bounds=Bounds(lb, ub), options=kw)
    b = H.minimize(fun, x0, (lb, ub), **kw)
    return a, b


@pytest.mark.parametrize("n", [2, 5, 10])
@pytest.mark.parametrize("box", ["wide", "active", "free", "lower", "upper"])
def test_same_trajectory_as_scipy_on_rosenbrock(n, box):
    rs = np.random.RandomState(n)
    for _ in range(4):
        x0 = rs.uniform(-2, 2, size=n)
        lb, ub = np.full(n, -1.5), np.full(n, 2.0)
        if box == "active":
            ub[:] = 0.8
        elif box == "free":
            lb[:], ub[:] = -np.inf, np.inf
        elif box == "lower":
            ub[:] = np.inf
        elif box == "upper":
            lb[:] = -np.inf
        a, b = both(lambda x: (rosen(x), rosen_der(x)), x0, lb, ub, **OPTS)
        assert (a.nit, a.nfev, a.status) == (b.nit, b.nfev, b.status)
        np.testing.assert_allclose(b.x, a.x, rtol=0, atol=1e-6)    # rounding over <= 80 iterations
        assert b.fun == pytest.approx(a.fun, abs=1e-10)
        np.testing.assert_allclose(b.jac, a.jac, atol=1e-4)


def test_limits_and_status_codes_match_scipy():
    f = lambda x: (rosen(x), rosen_der(x))
    x0 = np.array([-1.2, 1.0, -0.5, 0.7])
    lb, ub = np.full(4, -2.0), np.full(4, 2.0)
    for kw in (dict(maxiter=3), dict(maxiter=1), dict(maxfun=4), dict(maxfun=1), dict(maxls=1),
               dict(maxls=2), dict(maxcor=3), dict(maxcor=1), dict(gtol=1e-1), dict(ftol=1e-2)):
        a, b = both(f, x0, lb, ub, **kw)
        assert (a.nit, a.nfev, a.status) == (b.nit, b.nfev, b.status), kw
        np.testing.assert_allclose(b.x, a.x, atol=1e-4 if kw.get("maxcor") == 1 else 1e-9,
                                   err_msg=str(kw))
    # message codes
    a, b = both(f, x0, lb, ub, maxiter=2)
    assert b.task == (5, 504) and "ITERATIONS" in a.message
    a, b = both(f, x0, lb, ub)
    assert b.task[0] == 4 and a.status == 0


def test_start_on_bounds_outside_box_and_stationary_points():
    q = lambda x: (0.5 * np.sum((x - 0.3) ** 2), x - 0.3)
    lb, ub = np.zeros(3), np.ones(3)
    for x0 in (np.zeros(3), np.ones(3), np.array([-5.0, 0.5, 7.0]), np.full(3, 0.3)):
        a, b = both(q, x0, lb, ub, **OPTS)
        assert (a.nit, a.nfev, a.status) == (b.nit, b.nfev, b.status)
        np.testing.assert_allclose(b.x, a.x, atol=1e-12)
    # minimiser outside the box: every variable ends on a bound
    q2 = lambda x: (0.5 * np.sum((x - 2.0) ** 2), x - 2.0)
    a, b = both(q2, np.full(3, 0.5), lb, ub, **OPTS)
    assert (a.nit, a.nfev, a.status) == (b.nit, b.nfev, b.status)
    np.testing.assert_array_equal(b.x, np.ones(3))
    # fixed variable (l == u)
    lb2, ub2 = np.array([0.0, 0.4, 0.0]), np.array([1.0, 0.4, 1.0])
    a, b = both(q, np.array([0.9, 0.4, 0.1]), lb2, ub2, **OPTS)
    assert (a.nit, a.nfev, a.status) == (b.nit, b.nfev, b.status)
    np.testing.assert_allclose(b.x, a.x, atol=1e-12)


def test_ill_conditioned_quadratic_with_many_active_bounds():
    rs = np.random.RandomState(0)
    n = 30
    A = rs.normal(size=(n, n))
    A = A @ A.T + 1e-3 * np.eye(n)
    c = rs.normal(size=n) * 3
    f = lambda x: (0.5 * x @ A @ x - c @ x, A @ x - c)
    a, b = both(f, rs.uniform(size=n), np.zeros(n), np.ones(n), maxiter=1000, ftol=1e-12)
    assert a.status == b.status == 0
    assert b.fun == pytest.approx(a.fun, rel=1e-9)
    assert abs(a.nit - b.nit) <= max(3, a.nit // 5)
    np.testing.assert_allclose(b.x, a.x, atol=1e-5)
    assert ((b.x == 0) | (b.x == 1)).sum() >= 5


@pytest.mark.parametrize("D,units,acts,tr", [(2, [16, 16, 1], ["relu", "relu", "sigmoid"], "identity"),
                                             (6, [32, 32, 1], ["relu", "relu", "linear"], "sigmoid"),
                                             (3, [32, 32, 32, 1], ["elu"] * 3 + ["linear"], "exp")])
def test_fp32_classifier_objective_statistics_match_scipy(D, units, acts, tr):
    """The real objective: fp32 network output, fp64 optimiser (bore/decorators.py:54-61).
    The line search works at the fp32 noise floor here (ftol=1e-9 on values with 6e-8
    resolution), yet almost every run reproduces scipy's iteration count, evaluation count
    (including ScalarFunction's cache hits), status and iterate."""
    rs = np.random.RandomState(1)
    p = O.glorot_uniform_params(D, units, rs)
    for i in range(1, len(p), 2):
        p[i] = rs.normal(scale=.1, size=p[i].shape).astype(np.float32)
    fg = lambda x: tuple(O.value_and_input_grad(p, acts, x, tr))
    lb, ub = np.zeros(D), np.ones(D)
    R = 40
    res = [both(fg, rs.uniform(size=D), lb, ub, **OPTS) for _ in range(R)]
    same = sum((a.nit, a.nfev, a.status) == (b.nit, b.nfev, b.status)
               and np.allclose(a.x, b.x, atol=1e-7) for a, b in res)
    assert same >= 0.8 * R
    fa = np.array([a.fun for a, _ in res])
    fb = np.array([b.fun for _, b in res])
    assert np.median(np.abs(fa - fb)) <= 1e-6
    assert np.mean(np.abs(fa - fb) < 1e-4) >= 0.9           # same local optimum, with few exceptions
    ok_a = np.mean([a.success or a.status == 1 for a, _ in res])
    ok_b = np.mean([b.success or b.status == 1 for _, b in res])
    assert abs(ok_a - ok_b) <= 0.15                         # same share of accepted restarts
    na = np.mean([a.nfev for a, _ in res])
    nb = np.mean([b.nfev for _, b in res])
    assert abs(na - nb) <= 0.25 * na


def test_host_build_is_clean_under_asan_and_ubsan(tmp_path):
    """The optimiser header compiled with -fsanitize=address,undefined (GPU sanitizers are not
    available on the pool; the host build shares every line of the state machine): 256 runs over
    n = 1..31, maxcor = 1..17, all bound patterns, plus the fp32 classifier objective."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    so = str(tmp_path / "liblbfgsb_host_san.so")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                         "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                         os.path.join(here, "native", "lbfgsb_host.cpp"), "-o", so],
                        capture_output=True, text=True)
    if cc.returncode != 0:
        pytest.skip("no sanitizer runtime in this toolchain: " + cc.stderr[-200:])
    pre = [subprocess.run(["g++", f"-print-file-name={n}"], capture_output=True, text=True).stdout.strip()
           for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(p) for p in pre):
        pytest.skip("sanitizer runtimes not found")
    env = dict(os.environ, LD_PRELOAD=":".join(pre), ASAN_OPTIONS="detect_leaks=0")
    run = subprocess.run([sys.executable, os.path.join(here, "native", "sanitized_run.py"), so],
                         capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0 and "sanitized runs ok: 256" in run.stdout, run.stderr[-2000:]


def test_the_three_forms_of_the_routine_give_the_same_bits():
    """lbfgsb_advance as a reverse-communication state machine (form 0), calling the evaluation
    itself (form 1, what a wave that owns one problem runs on the device) and with the two-variable
    line search in registers (form 2, the fused iteration kernel): same operations in the same
    order -- x, fun, jac, nit, nfev, status, task bit for bit, on smooth problems and on the float32
    classifier objective whose line searches collapse, hit the evaluation cache and end abnormally."""
    def same(a, b):
        return (np.array_equal(a.x, b.x) and a.fun == b.fun and np.array_equal(a.jac, b.jac)
                and (a.nit, a.nfev, a.status, a.task) == (b.nit, b.nfev, b.status, b.task))
    rs = np.random.RandomState(11)
    n_cases = n_abnormal = 0
    for n in (2, 5):
        for box in range(3):
            x0 = rs.uniform(-2, 2, size=n)
            lb = np.full(n, -1.5) if box != 2 else np.full(n, -np.inf)
            ub = np.full(n, 0.8 if box == 1 else 2.0) if box != 2 else np.full(n, np.inf)
            f = lambda x: (rosen(x), rosen_der(x))
            for kw in (OPTS, dict(maxls=2), dict(maxiter=3), dict(maxfun=7)):
                r0 = H.minimize(f, x0, (lb, ub), form=0, **kw)
                for form in (1, 2):
                    assert same(r0, H.minimize(f, x0, (lb, ub), form=form, **kw)), (n, box, kw, form)
                n_cases += 1
    # the classifier objective in float32 (bore/mixins.py:20: transform(-f(x))), two variables
    acts = ["relu", "relu", "sigmoid"]
    for seed in range(12):
        rs = np.random.RandomState(100 + seed)
        p = O.glorot_uniform_params(2, [16, 16, 1], rs)
        for q in p:
            q *= 3.0          # saturating nets: flat regions, abnormal line searches
        def f(x):
            v, g = O.value_and_input_grad(p, acts, x, "identity")
            return float(v), np.asarray(g, dtype=np.float64)
        for _ in range(4):
            x0 = rs.uniform(size=2)
            r0 = H.minimize(f, x0, (np.zeros(2), np.ones(2)), form=0, **OPTS)
            for form in (1, 2):
                assert same(r0, H.minimize(f, x0, (np.zeros(2), np.ones(2)), form=form, **OPTS)), (seed, form)
            n_cases += 1
            n_abnormal += r0.status == 2
    assert n_cases >= 70 and n_abnormal >= 3      # (the abnormal endings are exercised)


def test_counting_build_runs_the_same_optimisation(tmp_path):
    """tests/native/lbfgsb_flops.cpp -- lbfgsb.h with `double` replaced by an operation-counting wrapper, the build
    behind bench.py's `roofline_fp64_optimiser` (tools/lbfgsb_flops.py) -- walks the same path as the plain host build:
    same point bit for bit, same counts of iterations and evaluations; and it does count."""
    import ctypes as C
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so = str(tmp_path / "liblbfgsb_flops.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                    os.path.join(here, "native", "lbfgsb_flops.cpp"), "-o", so], check=True)
    lib = C.CDLL(so)
    CB = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
    n = 6
    rs = np.random.RandomState(4)
    lb, ub = np.full(n, -1.5), np.full(n, 0.9)

    def fun(x):
        return rosen(x), rosen_der(x)

    def cb(n_, xp, fp, gp):
        x = np.ctypeslib.as_array(xp, shape=(n_,)).copy()
        f, g = fun(x)
        fp[0] = float(f)
        for i in range(n_):
            gp[i] = g[i]

    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for _ in range(3):
        x0 = rs.uniform(-1.4, 0.8, size=n)
        ref = H.minimize(fun, x0, (lb, ub), **OPTS)
        xo, fo = np.empty(n), C.c_double()
        oi, oc = np.zeros(3, dtype=np.int32), np.zeros(2, dtype=np.uint64)
        nbd = np.full(n, 2, dtype=np.int32)
        lib.lbfgsb_flops_minimize(n, 10, dp(np.ascontiguousarray(x0)), dp(lb), dp(ub), nbd.ctypes.data_as(C.POINTER(C.c_int)),
                                  C.c_double(OPTS["ftol"] / np.finfo(float).eps), C.c_double(1e-5), OPTS["maxiter"], 15000, 20,
                                  CB(cb), dp(xo), C.byref(fo), oi.ctypes.data_as(C.POINTER(C.c_int)),
                                  oc.ctypes.data_as(C.POINTER(C.c_ulonglong)))
        assert (int(oi[0]), int(oi[1]), int(oi[2])) == (ref.nit, ref.nfev, ref.status)
        assert np.array_equal(xo, ref.x) and fo.value == ref.fun
        assert oc[0] > 100 * ref.nit and 0 < oc[1] < oc[0]
