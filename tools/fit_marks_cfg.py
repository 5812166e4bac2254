"""Where an Adam step of the 6->32-32-1 fit (BASELINE config 2, N = 256) spends its cycles: the marks of
tools/fit_marks.py around a stand-alone fit launch (diagnostic build libbore_hip_fitmarks.so =
-DBORE_FIT_MARKS; GPU box).  usage: python tools/fit_marks_cfg.py [N] [loops] [cfg2|plugin]
(plugin: the plugin's default network 16->32-32-32-1, elu x3 + linear; BORE_FIT_W8=0 for the four-wave body the marks
are kept per wave for)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_fitmarks.so"))
import numpy as np, torch
from bore_amd import _lib, ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1
which = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
D, units, E = (6, [32, 32, 1], 40) if which == "cfg2" else (16, [32, 32, 32, 1], 40)
rs = np.random.RandomState(7)
desc = _lib.make_desc(D, units, ["relu", "relu", "sigmoid"] if which == "cfg2" else ["elu", "elu", "elu", "linear"])
P = ops.param_count(desc)
th = torch.from_numpy(rs.normal(scale=0.2, size=(L, P)).astype(np.float32)).cuda()
m, v = torch.zeros_like(th), torch.zeros_like(th)
t = torch.zeros(L, dtype=torch.int64, device="cuda")
X = torch.from_numpy(rs.uniform(size=(L, N, D)).astype(np.float32)).cuda()
z = torch.from_numpy((rs.uniform(size=(L, N)) < 0.25).astype(np.float32)).cuda()
ops.mlp_fit(desc, th, m, v, t, X, z, 2, 64, seed=3, want_loss=False)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
_lib.lib().bore_debug_fit_marks(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.mlp_fit(desc, th, m, v, t, X, z, E, 64, seed=3, epoch0=2, want_loss=False)
e1.record()
torch.cuda.synchronize()
_lib.lib().bore_debug_fit_marks(buf, 0)
a = np.array(buf, dtype=np.float64).reshape(8, 32)
names = ["gather+requests", "forward", "loss+delta", "backward+copies", "wait mid barrier", "dW+Adam phase", "wait end barrier",
         "step loop top -> step", "  task: requests", "  task: matrix chain", "  task: Adam+stores", "(a mark itself)", "(end barrier -> epoch top)",
         "(epoch top -> shuffle chosen)", "(-> step loop top)", "(mid barrier -> own task done)"]
n_steps = a[:, 16 + 5].max()
print(f"{D}->{'-'.join(map(str, units))}, N {N}, {L} loops, {E} epochs: {1e3 * e0.elapsed_time(e1) / (E * ((N + 63) // 64)):.2f} us per Adam step (marks build); {n_steps:.0f} steps marked")
for i, nm in enumerate(names):
    print(f"  {nm:30s} " + "  ".join(f"{a[w, i] / max(a[w, 16 + i], 1):7.0f} ({a[w, 16 + i] / max(n_steps, 1):4.2f})" for w in range(8) if a[w, 16 + 7] > 0))
print("  sum of 0..7 per step           " + "  ".join(f"{sum(a[w, i] for i in range(8)) / max(n_steps, 1):7.0f}       " for w in range(8) if a[w, 16 + 7] > 0))
print("  12..14 per step                " + "  ".join(f"{sum(a[w, i] for i in (12, 13, 14)) / max(n_steps, 1):7.0f}       " for w in range(8) if a[w, 16 + 7] > 0))
