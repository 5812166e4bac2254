"""How long the slowest restarts of a wide config's launch are against the mean (GPU box): iterations and
evaluations per restart over `loops` loops.  usage: python tools/restart_tail.py <cfg2|cfg3|cfg5> [loops]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bore_amd import _lib, ops
key = {"cfg2": "cfg2_hartmann6_32-32-1_R256", "cfg3": "cfg3_hpo16_64-64-64-1_R1024", "cfg5": "cfg5_nas32_128-128-1_bf16_R4096"}[sys.argv[1]]
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 256
c = bench.WIDE_CONFIGS[key]
D, units, R, Ns, N = c["D"], c["units"], c["R"], c["Ns"], c["N"]
acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
desc = _lib.make_desc(D, units, acts, compute=c["compute"])
P = ops.param_count(desc)
rs = np.random.RandomState(0)
th = np.zeros((loops, P), dtype=np.float32)
for l in range(loops):
    off, fan = 0, D
    for u in units:
        lim = np.sqrt(6.0 / (fan + u))
        th[l, off:off + fan * u] = rs.uniform(-lim, lim, size=fan * u)
        off += fan * u + u
        fan = u
th = torch.from_numpy(th).cuda()
m, v = torch.zeros_like(th), torch.zeros_like(th)
t = torch.zeros(loops, dtype=torch.int64, device="cuda")
X, y = bench._synthetic(rs, loops, N, D)
z = (y < np.quantile(y, 0.25, axis=1)[:, None]).astype(np.float32)
Xd, zd = torch.from_numpy(X.astype(np.float32)).cuda(), torch.from_numpy(z).cuda()
lo, hi = np.zeros(D), np.ones(D)
for k in range(3):
    ops.mlp_fit(desc, th, m, v, t, Xd, zd, 200, 64, seed=0, epoch0=k * 200, want_loss=False)
    x0, _ = ops.sample_screen_topk(desc, th, 0, Ns, lo, hi, R, draw_index=k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    x, fun, jac, info = ops.lbfgsb_minimize(desc, th, x0, lo, hi, "identity", True, maxiter=1000, ftol=1e-9)
    e1.record()
    torch.cuda.synchronize()
    inf = info.cpu().numpy()
    nit, nfev = inf[:, :, 0].ravel(), inf[:, :, 1].ravel()
    q = lambda a: " ".join(f"{np.percentile(a, p):.0f}" for p in (50, 90, 99, 99.9, 100))
    print(f"{sys.argv[1]} x {loops} loops, fit {k + 1}: restarts {e0.elapsed_time(e1):.2f} ms; nit mean {nit.mean():.1f} p50/90/99/99.9/max {q(nit)}; "
          f"nfev mean {nfev.mean():.1f} p50/90/99/99.9/max {q(nfev)}; restarts with nfev > 10 x mean: {(nfev > 10 * nfev.mean()).sum()}", flush=True)
