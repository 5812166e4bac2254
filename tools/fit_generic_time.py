"""Fit launch of nets OUTSIDE the static shapes (the generic kernel flavours: what a user's own architecture
runs), four against eight waves per workgroup (GPU box).  usage: python tools/fit_generic_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bore_amd import _lib, ops
rs = np.random.RandomState(0)
for D, units, N in ((10, [32, 32, 1], 256), (4, [32, 32, 1], 100), (8, [64, 64, 1], 256), (3, [16, 16, 16, 1], 256), (12, [48, 1], 300)):
    acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
    desc = _lib.make_desc(D, units, acts)
    P = ops.param_count(desc)
    X = torch.from_numpy(rs.uniform(size=(1, N, D)).astype(np.float32)).cuda()
    z = torch.from_numpy((rs.uniform(size=(1, N)) < 0.25).astype(np.float32)).cuda()
    out = {}
    for w8 in ("0", "1"):
        os.environ["BORE_FIT_W8"] = w8
        th = torch.from_numpy(rs.normal(scale=0.2, size=(1, P)).astype(np.float32)).cuda()
        m, v = torch.zeros_like(th), torch.zeros_like(th)
        t = torch.zeros(1, dtype=torch.int64, device="cuda")
        ops.mlp_fit(desc, th, m, v, t, X, z, 5, 64, seed=1, want_loss=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.mlp_fit(desc, th, m, v, t, X, z, 200, 64, seed=1, epoch0=5, want_loss=False)
        e1.record()
        torch.cuda.synchronize()
        out[w8] = e0.elapsed_time(e1)
    print(f"{D}->{'-'.join(map(str, units))}, N {N}, 200 epochs: four waves {out['0']:.2f} ms, eight {out['1']:.2f} ms", flush=True)
