"""Cycle stamps of ONE Adam step of the static-shape float32 fit (diagnostic build
bore_amd/csrc/libbore_hip_stamps.so = the library compiled with -DBORE_STAMPS; GPU box).
Per wave: gather | forward | loss + delta | backward (+ stores) | wait at the barrier | dW task:
operand requests, MFMA chain, Adam + stores | wait at the step's last barrier."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_stamps.so"))
import numpy as np, torch
from bore_amd import _lib, ops
lib = _lib.lib()


def run(N, units=(16, 16, 1), D=2, acts=("relu", "relu", "sigmoid")):
    desc = _lib.make_desc(D, list(units), list(acts)); P = ops.param_count(desc); L = 1
    rs = np.random.RandomState(0)
    th = torch.from_numpy(rs.normal(scale=.3, size=(L, P)).astype(np.float32)).cuda()
    m = torch.zeros_like(th); v = torch.zeros_like(th); t = torch.zeros(L, dtype=torch.int64, device='cuda')
    X = torch.from_numpy(rs.uniform(size=(L, N, D)).astype(np.float32)).cuda(); z = (torch.rand(L, N, device='cuda') < 0.25).float()
    for _ in range(3):
        ops.mlp_fit(desc, th, m, v, t, X, z, 5, 64, want_loss=False)
    torch.cuda.synchronize()
    out = (C.c_longlong * 64)(); lib.bore_debug_stamps(out)
    for w in range(4):
        a = np.array(out[16 * w:16 * w + 12])
        print(f"N={N} wave {w}: gather {a[1]-a[0]} fwd {a[2]-a[1]} loss {a[3]-a[2]} bwd {a[4]-a[3]} barrier-wait {a[5]-a[4]} | "
              f"task: pre {a[8]-a[5]} loads {a[9]-a[8]} mfma {a[10]-a[9]} adam {a[11]-a[10]} | to end {a[6]-a[11]} last barrier {a[7]-a[6]} | step {a[7]-a[0]}")


for N in (12, 24, 40, 64):
    run(N)
