import sys; sys.path.insert(0, '.')
import numpy as np, torch
from bore_amd.engine import ReplicaEngine
kw = dict(epochs=20, mode="device", deduplicate=True, num_samples=64)
for grow in (False, True):
    a = ReplicaEngine(np.arange(6), select="device", groups=2, **kw)
    b = ReplicaEngine(np.arange(6), select="host", **kw)
    if grow:
        for g in a.groups:
            g.store.grow(g.store.n + 2)
    for step in range(12):
        xa, ya = a.step(); xb, yb = b.step()
        if not np.array_equal(xa, xb):
            print("grow", grow, "step", step, "differs in loops", np.nonzero((xa != xb).any(axis=1))[0], a.stats["none_results"], b.stats["none_results"])
            for g in a.groups:
                print(" best", g.best_pin.numpy(), "info status", g.info_pin.numpy()[:, :, 2].tolist())
            break
    else:
        print("grow", grow, "equal", a.X.shape, b.X.shape)
