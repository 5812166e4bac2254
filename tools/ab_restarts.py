"""A/B of the restart launch of a BASELINE wide config on FIXED inputs (GPU box): the models are fitted once, the starts
screened once, then the same bore_lbfgsb_minimize call is repeated under each setting of an environment switch in turn
(the library reads its switches per launch), interleaved, and the median / min per setting is printed with the
evaluation requests served -- identical across settings, asserted, as are the results bit for bit.
usage: python tools/ab_restarts.py <cfg2|cfg3|cfg5|plugin> <ENV_NAME> <v1,v2,...> [loops] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bench import _counts, _synthetic
from bore_amd import _lib, ops

key = {"cfg2": "cfg2_hartmann6_32-32-1_R256", "cfg3": "cfg3_hpo16_64-64-64-1_R1024",
       "cfg5": "cfg5_nas32_128-128-1_bf16_R4096", "plugin": "plugin_D16_transform_identity"}[sys.argv[1]]
c = dict(bench.WIDE_CONFIGS, **bench.PLUGIN_CONFIGS)[key]
env, values = sys.argv[2], sys.argv[3].split(",")
loops = int(sys.argv[4]) if len(sys.argv) > 4 else 256
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 7
D, units, R, Ns, N = c["D"], c["units"], c["R"], c["Ns"], c["N"]
acts = c.get("acts") or ["relu"] * (len(units) - 1) + ["sigmoid"]
transform, gamma, epochs = c.get("transform", "identity"), c.get("gamma", 0.25), c.get("epochs", 200)
desc = _lib.make_desc(D, units, acts, compute=c["compute"])
M, P = _counts(D, units)
rs = np.random.RandomState(0)
th = np.zeros((loops, P), dtype=np.float32)
for l in range(loops):
    off, fan = 0, D
    for u in units:
        lim = np.sqrt(6.0 / (fan + u))
        th[l, off:off + fan * u] = rs.uniform(-lim, lim, size=fan * u)
        off += fan * u + u
        fan = u
X, y = _synthetic(rs, loops, N, D)
z = (y < np.quantile(y, gamma, axis=1)[:, None]).astype(np.float32)
dev = torch.device("cuda", 0)
theta = torch.from_numpy(th).to(dev)
m, v = torch.zeros_like(theta), torch.zeros_like(theta)
t = torch.zeros(loops, dtype=torch.int64, device=dev)
Xd, zd = torch.from_numpy(X.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev)
lo, hi = np.zeros(D), np.ones(D)
for k in range(2):          # (two fits: the surface the bench's second repetition sees)
    ops.mlp_fit(desc, theta, m, v, t, Xd, zd, epochs, 64, seed=0, epoch0=k * epochs, want_loss=False)
x0, _ = ops.sample_screen_topk(desc, theta, 0, Ns, lo, hi, R, draw_index=1)
torch.cuda.synchronize()
ms = {val: [] for val in values}
ref = None
for r in range(rounds + 1):              # (round 0 warms up)
    for val in values:
        os.environ[env] = val
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.lbfgsb_minimize(desc, theta, x0, lo, hi, transform, True, maxiter=1000, ftol=1e-9)
        e1.record()
        torch.cuda.synchronize()
        got = [o.cpu().numpy() for o in out]
        if ref is None:
            ref = got
        assert all(np.array_equal(a, b) for a, b in zip(ref, got)), f"{env}={val}: results differ"
        if r:
            ms[val].append(e0.elapsed_time(e1))
nfev = float(ref[3][:, :, 1].sum())
print(f"{sys.argv[1]} x {loops} loops, {R} restarts each: {nfev:.0f} evaluation requests, nit mean {ref[3][:, :, 0].mean():.1f}, "
      f"status<=1 {np.mean(ref[3][:, :, 2] <= 1):.3f}; {rounds} interleaved rounds, same bits under every setting")
for val in values:
    a = np.array(ms[val])
    print(f"  {env}={val:>4s}: median {np.median(a):7.3f} ms  min {a.min():7.3f}  max {a.max():7.3f}  "
          f"({np.median(a) / nfev * 1e6:.2f} ms per million requests)")
