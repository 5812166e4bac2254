"""Soak: asynchronous fused schedule against lock-step groups over a long run (record growth, N >> 64)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from bore_amd.engine import NativeEngine
for L, steps, kw in [(64, 600, {}), (512, 150, dict(deduplicate=True)), (7, 300, dict(num_samples=32, epochs=10))]:
    t0 = time.perf_counter(); a = NativeEngine(np.arange(L), async_loops=True, **kw); a.run(steps); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); b = NativeEngine(np.arange(L), groups=4, **kw); b.run(steps); tb = time.perf_counter() - t0
    same = np.array_equal(a.X, b.X) and np.array_equal(a.y, b.y) and all(np.array_equal(u, v) for u, v in zip(a.state(), b.state()))
    print(f"L={L} steps={steps} {kw}: identical={same}  async {ta:.2f} s, groups {tb:.2f} s; none {a.take_stats()['none_results']}", flush=True)
    assert same
