"""f/g of the restart kernel's one-point-per-wave evaluation (PointNet) against the row kernel's (matrix
pipeline) at the same points: must be bit-identical (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bore_amd import _lib, ops
for D, units, compute, tr in [(16, [64, 64, 64, 1], "float32", "sigmoid"), (16, [64, 64, 64, 1], "bfloat16", "sigmoid"),
                              (32, [128, 128, 1], "bfloat16", "identity"), (6, [32, 32, 1], "float32", "identity")]:
    rs = np.random.RandomState(5)
    acts = ["relu"] * (len(units) - 1) + ["linear"]
    desc = _lib.make_desc(D, units, acts, compute=compute)
    P = ops.param_count(desc)
    th = torch.from_numpy(rs.normal(scale=0.3, size=(1, P)).astype(np.float32)).cuda()
    x0 = torch.from_numpy(rs.uniform(size=(1, 8, D))).cuda()
    lo, hi = np.zeros(D), np.ones(D)
    x, fun, jac, info = ops.lbfgsb_minimize(desc, th, x0, lo, hi, tr, True, maxiter=0, ftol=1e-9)
    v, g = ops.mlp_value_and_input_grad(desc, th, x, tr, True)
    dv = (v - fun).abs().max().item(); dg = (g - jac).abs().max().item()
    print(f"D={D} {units} {compute} {tr}: max |dval| {dv:.3e}  max |dgrad| {dg:.3e}  (val {fun[0,:3].cpu().numpy()}, rows {v[0,:3].cpu().numpy()})")
