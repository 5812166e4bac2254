"""Phase cycles of the restart launch of a wide BASELINE config (default config 2: 6-D, 32-32-1, 256 restarts per
loop) per PROBLEM: the stamps of tools/lbfgsb_phases_standalone.py summed over every wave's problems (diagnostic
build libbore_hip_stamps.so = -DBORE_STAMPS; GPU box).  usage: python tools/lbfgsb_phases_cfg.py [cfg2|cfg3|cfg5] [loops]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_stamps.so"))
import numpy as np, torch
import bench
from bore_amd import _lib, ops
lib = _lib.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
name, c = next((k, v) for k, v in bench.WIDE_CONFIGS.items() if k.startswith(which))
D, units, R, Ns, N = c["D"], c["units"], c["R"], c["Ns"], c["N"]
acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
desc = _lib.make_desc(D, units, acts, compute=c["compute"])
rs = np.random.RandomState(0)
M, P = bench._counts(D, units)
th = np.zeros((L, P), dtype=np.float32)
for l in range(L):
    off, fan = 0, D
    for u in units:
        lim = np.sqrt(6.0 / (fan + u))
        th[l, off:off + fan * u] = rs.uniform(-lim, lim, size=fan * u)
        off += fan * u + u
        fan = u
X, y = bench._synthetic(rs, L, N, D)
z = (y < np.quantile(y, 0.25, axis=1)[:, None]).astype(np.float32)
theta = torch.from_numpy(th).cuda()
m, v = torch.zeros_like(theta), torch.zeros_like(theta)
t = torch.zeros(L, dtype=torch.int64, device="cuda")
ops.mlp_fit(desc, theta, m, v, t, torch.from_numpy(X.astype(np.float32)).cuda(), torch.from_numpy(z).cuda(), 200, 64, seed=1,
            want_loss=False)
lo, hi = np.zeros(D), np.ones(D)
x0, _ = ops.sample_screen_topk(desc, theta, 0, Ns, lo, hi, R, draw_index=1)
names = ["cauchy", "formk", "cmprlb", "subsm", "lnsrlb", "matupd", "formt", "head", "freev", "accept", "cachechk", "bfgspair", "d=z-x"]
for rep in range(2):
    lib.bore_debug_lphases_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    x, fun, jac, info = ops.lbfgsb_minimize(desc, theta, x0, lo, hi, "sigmoid", True, maxiter=1000, ftol=1e-9)
    e1.record()
    torch.cuda.synchronize()
pp = (C.c_ulonglong * (4096 * 64))()
lib.bore_debug_lpp(pp)
pp = np.array(pp, dtype=np.float64).reshape(4096, 64).sum(axis=0)
n_prob = L * R
inf = info.cpu().numpy()
print(f"{name}: {L} loops x {R} restarts, launch {e0.elapsed_time(e1):.2f} ms (stamps build); per problem: nit {inf[..., 0].mean():.1f} "
      f"nfev {inf[..., 1].mean():.1f}; advance {pp[13] / n_prob:.0f} cycles, f/g {pp[14] / n_prob:.0f} ({pp[45] / n_prob:.1f} rounds)")
acc = 0.0
for i, nm in enumerate(names):
    acc += pp[i]
    print(f"  {nm:8s}: {pp[i] / n_prob:9.0f} cycles per problem, {pp[32 + i] / n_prob:6.2f} calls, {pp[i] / max(pp[32 + i], 1):7.0f} per call")
if os.environ.get("FORMK"):      # (build with -DBORE_STAMPS_FORMK: buckets 19..27 = the stages of formk)
    for g, nm in zip(range(19, 28), ["new rows", "old parts", "assembly", "first factorisation", "triangular solves", "(2,2) block",
                                     "second factorisation", "-", "-"]):
        print(f"  formk / {nm:22s}: {pp[g] / n_prob:9.0f} cycles per problem, {pp[g] / max(pp[32 + g], 1):7.0f} per call")
gaps = sum(pp[g] for g in range(15, 32))
print(f"  gaps between phases: {gaps / n_prob:9.0f} cycles per problem; unaccounted {(pp[13] - acc - gaps) / n_prob:.0f}")
