"""Cycles per phase of an Adam step of the wide mixed-precision fit (needs the diagnostic build
bore_amd/csrc/libbore_hip_dbg.so = the library compiled with -DBORE_WIDE_STAMPS; GPU box)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_dbg.so"))
import numpy as np, torch
from bore_amd import _lib, ops
NAMES = ["gather (+A_0 image)", "forward", "loss + delta", "backward", "store A^T / D^T images", "barrier",
         "grads (MFMA)", "-", "scatter + packed Adam", "barrier"]
for name, D, units, compute in [("shape4_bf16", 32, [128, 128, 1], "bfloat16"), ("shape3_bf16", 16, [64, 64, 64, 1], "bfloat16"),
                                ("shape3_f32 (phases: gather, forward, store_A+loss, backward, store_D, barrier, grads, barrier+scatter+barrier, packed Adam)",
                                 16, [64, 64, 64, 1], "float32")]:
    rs = np.random.RandomState(7)
    acts = ["relu"] * (len(units) - 1) + ["sigmoid"]
    desc = _lib.make_desc(D, units, acts, compute=compute)
    P = ops.param_count(desc)
    L, N, E = 1, 256, 20
    th = torch.from_numpy(rs.normal(scale=0.2, size=(L, P)).astype(np.float32)).cuda()
    m, v = torch.zeros_like(th), torch.zeros_like(th)
    t = torch.zeros(L, dtype=torch.int64, device="cuda")
    X = torch.from_numpy(rs.uniform(size=(L, N, D)).astype(np.float32)).cuda()
    z = torch.from_numpy((rs.uniform(size=(L, N)) < 0.25).astype(np.float32)).cuda()
    ops.mlp_fit(desc, th, m, v, t, X, z, 2, 64, seed=3, want_loss=False)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    _lib.lib().bore_debug_wide_stamps(buf, 1)
    ops.mlp_fit(desc, th, m, v, t, X, z, E, 64, seed=3, epoch0=2, want_loss=False)
    torch.cuda.synchronize()
    _lib.lib().bore_debug_wide_stamps(buf, 1)
    a = np.array(buf[:]).reshape(4, 16) / (E * 4)
    print(f"{name}: cycles per Adam step, per wave (sum wave 0: {a[0].sum():.0f})")
    for i, nm in enumerate(NAMES):
        print(f"  {nm:28s} " + " ".join(f"{a[w, i]:9.0f}" for w in range(4)))
