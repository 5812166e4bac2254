"""The plugin's real default network, D -> 32-32-32-1 (elu x3, linear output; static shape 5): fit, screening and
restarts of ONE model against the generic flavour (BORE_FIT_PAD=0 for the fit; the acquisition kernels have no switch:
their generic time is taken on a net of other widths, 32-32-31-1).  GPU box.  usage: python tools/plugin_default_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bore_amd import _lib, ops
rs = np.random.RandomState(0)


def timed(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for D, N, E in ((10, 100, 200), (16, 100, 200), (6, 100, 500), (16, 100, 500)):
    out = {}
    for units in ([32, 32, 32, 1], [32, 32, 31, 1]):
        acts = ["elu", "elu", "elu", "linear"]
        desc = _lib.make_desc(D, units, acts)
        P = ops.param_count(desc)
        X = torch.from_numpy(rs.uniform(size=(1, N, D)).astype(np.float32)).cuda()
        z = torch.from_numpy((rs.uniform(size=(1, N)) < 1 / 3).astype(np.float32)).cuda()
        th = torch.from_numpy(rs.normal(scale=0.2, size=(1, P)).astype(np.float32)).cuda()
        m, v = torch.zeros_like(th), torch.zeros_like(th)
        t = torch.zeros(1, dtype=torch.int64, device="cuda")
        lo, hi = np.zeros(D), np.ones(D)
        fit = timed(lambda: ops.mlp_fit(desc, th, m, v, t, X, z, E, 64, seed=1, want_loss=False))
        scr = timed(lambda: ops.sample_screen_topk(desc, th, 0, 1024, lo, hi, 5))
        x0, _ = ops.sample_screen_topk(desc, th, 0, 1024, lo, hi, 5)
        rst = timed(lambda: ops.lbfgsb_minimize(desc, th, x0, lo, hi, "sigmoid", True, maxiter=1000, ftol=1e-9))
        out["-".join(map(str, units))] = (fit, scr, rst)
    (f5, s5, r5), (fg, sg, rg) = out["32-32-32-1"], out["32-32-31-1"]
    print(f"{D}->32-32-32-1, N {N}, {E} epochs: fit {f5:.2f} ms, screen {s5:.3f} ms, 5 restarts {r5:.2f} ms   "
          f"(generic flavour on {D}->32-32-31-1: fit {fg:.2f}, screen {sg:.3f}, restarts {rg:.2f})", flush=True)
