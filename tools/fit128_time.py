import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bore_amd import _lib, ops
rs = np.random.RandomState(1)
D, units = 32, [128, 128, 1]
desc = _lib.make_desc(D, units, ["relu", "relu", "sigmoid"])
P = ops.param_count(desc); L, N, E = 3, 256, 50
th = torch.from_numpy(rs.normal(scale=0.2, size=(L, P)).astype(np.float32)).cuda()
m, v = torch.zeros_like(th), torch.zeros_like(th); t = torch.zeros(L, dtype=torch.int64, device="cuda")
X = torch.from_numpy(rs.uniform(size=(L, N, D)).astype(np.float32)).cuda(); z = (torch.rand(L, N, device="cuda") < 0.25).float()
ops.mlp_fit(desc, th, m, v, t, X, z, 5, 64, seed=3, want_loss=False); torch.cuda.synchronize()
t0 = time.perf_counter(); ops.mlp_fit(desc, th, m, v, t, X, z, E, 64, seed=3, epoch0=5, want_loss=False); torch.cuda.synchronize()
print("fp32 32->128-128-1: %.2f us per Adam step" % (1e6 * (time.perf_counter() - t0) / (E * 4)))
