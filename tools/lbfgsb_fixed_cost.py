"""Fixed cost per workgroup of the restart kernel: time of bore_lbfgsb_minimize with maxiter 0 / 1 / 1000
for the BASELINE shapes (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bore_amd import _lib, ops

for name, D, units, compute, L, R in [("cfg5_bf16", 32, [128, 128, 1], "bfloat16", 1, 4096), ("cfg5_bf16", 32, [128, 128, 1], "bfloat16", 16, 4096),
                                      ("cfg3", 16, [64, 64, 64, 1], "float32", 1, 1024), ("cfg3", 16, [64, 64, 64, 1], "float32", 64, 1024),
                                      ("cfg2", 6, [32, 32, 1], "float32", 64, 256), ("cfg1", 2, [16, 16, 1], "float32", 512, 3)]:
    rs = np.random.RandomState(3)
    acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
    desc = _lib.make_desc(D, units, acts, compute=compute)
    P = ops.param_count(desc)
    th = torch.from_numpy(rs.normal(scale=0.3, size=(L, P)).astype(np.float32)).cuda()
    x0 = torch.from_numpy(rs.uniform(size=(L, R, D))).cuda()
    lo, hi = np.zeros(D), np.ones(D)
    row = []
    for maxiter in (0, 1, 1000):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            x, fun, jac, info = ops.lbfgsb_minimize(desc, th, x0, lo, hi, "identity", True, maxiter=maxiter, ftol=1e-9)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        inf = info.cpu().numpy()
        row.append(f"maxiter {maxiter}: {1e3 * dt:8.3f} ms (nit {inf[..., 0].mean():.1f}, nfev {inf[..., 1].mean():.1f})")
    print(f"{name} L={L} R={R}: " + " | ".join(row), flush=True)
