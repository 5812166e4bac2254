import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bore_amd import _lib, ops
rs = np.random.RandomState(0)
for D in (10, 16):
    desc = _lib.make_desc(D, [32, 32, 32, 1], ["elu", "elu", "elu", "linear"])
    P = ops.param_count(desc)
    X = torch.from_numpy(rs.uniform(size=(1, 100, D)).astype(np.float32)).cuda()
    z = torch.from_numpy((rs.uniform(size=(1, 100)) < 1 / 3).astype(np.float32)).cuda()
    for w8 in ("0", "1"):
        os.environ["BORE_FIT_W8"] = w8
        th = torch.from_numpy(rs.normal(scale=0.2, size=(1, P)).astype(np.float32)).cuda()
        m, v = torch.zeros_like(th), torch.zeros_like(th)
        t = torch.zeros(1, dtype=torch.int64, device="cuda")
        ops.mlp_fit(desc, th, m, v, t, X, z, 5, 64, seed=1, want_loss=False)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.mlp_fit(desc, th, m, v, t, X, z, 200, 64, seed=1, epoch0=5, want_loss=False)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(f"{D}->32-32-32-1 N 100, 200 epochs, BORE_FIT_W8={w8}: {np.median(ts):.2f} ms (min {min(ts):.2f})", flush=True)
