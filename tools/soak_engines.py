"""Soak: the asynchronous engine with resident workgroups (fused kernel) against the lock-step engine (separate
kernels) over 150 BO iterations -- N 10 -> 160 crosses every form of the fit (shuffles by the idle wave, by the
workgroup, four-epoch groups, one epoch at a time); trajectories and weights must be bit-equal.  GPU box:
python tools/soak_engines.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bore_amd.engine import NativeEngine
kw = dict(epochs=40)
a = NativeEngine(np.arange(100, 148), async_loops=True, **kw)
b = NativeEngine(np.arange(100, 148), groups=3, **kw)
t0 = time.time()
for n in (30, 40, 50, 30):       # N: 10 -> 160 (crosses 48, 64, 112, 128)
    a.run(n)
b.run(150)
Xa, ya = a.observations(); Xb, yb = b.observations()
print("shape", Xa.shape, "equal X", np.array_equal(Xa, Xb), "equal y", np.array_equal(ya, yb),
      "state", all(np.array_equal(u, v) for u, v in zip(a.state(), b.state())), "%.1f s" % (time.time() - t0))
