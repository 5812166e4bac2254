"""Where a workgroup of the restart kernel spends its time (diagnostic build libbore_hip_stamps.so =
-DBORE_STAMPS; workgroup 0, thread 0; GPU box): begin_kernel | stage theta | optimiser init |
advance | f/g | total, in shader clocks."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_stamps.so"))
import numpy as np, torch
from bore_amd import _lib, ops
lib = _lib.lib()
for name, D, units, compute, L, R in [("cfg5_bf16", 32, [128, 128, 1], "bfloat16", 1, 4096), ("cfg3", 16, [64, 64, 64, 1], "float32", 1, 1024),
                                      ("cfg2", 6, [32, 32, 1], "float32", 1, 256), ("cfg1", 2, [16, 16, 1], "float32", 8, 3)]:
    rs = np.random.RandomState(3)
    acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
    desc = _lib.make_desc(D, units, acts, compute=compute)
    P = ops.param_count(desc)
    th = torch.from_numpy(rs.normal(scale=0.3, size=(L, P)).astype(np.float32)).cuda()
    x0 = torch.from_numpy(rs.uniform(size=(L, R, D))).cuda()
    lo, hi = np.zeros(D), np.ones(D)
    for maxiter in (0, 1000):
        for rep in range(2):
            lib.bore_debug_lphases_reset()
            x, fun, jac, info = ops.lbfgsb_minimize(desc, th, x0, lo, hi, "identity", True, maxiter=maxiter, ftol=1e-9)
            torch.cuda.synchronize()
        out = (C.c_longlong * 8)(); lib.bore_debug_lstamps(out)
        o = list(out)
        ph = (C.c_longlong * 16)(); lib.bore_debug_lphases(ph); ph = list(ph)
        names = os.environ.get("LB_NAMES", "cauchy,formk,cmprlb,subsm,lnsrlb,matupd,formt,-").split(",")
        print("   phases (workgroup 0 / wave 0; cycles per call x calls): " + "  ".join(
            f"{nm} {ph[i] // max(ph[8 + i], 1)} x {ph[8 + i]}" for i, nm in enumerate(names) if ph[8 + i]))
        print(f"{name} maxiter {maxiter}: begin {o[4]} stage {o[5]} init {o[6]} | advance {o[0]} fg {o[1]} rounds {o[2]} nit {o[3]} | total {o[7]}", flush=True)
