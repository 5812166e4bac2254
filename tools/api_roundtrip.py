"""The drop-in model API with HOST arrays in and out (the PCIe-inclusive figure of DESIGN.md 5): one BO loop
through MaximizableSequential.fit / .argmax exactly as README.rst:83-103 drives it -- numpy X, z uploaded by
every fit, the suggestion downloaded by every argmax -- against the device-resident replica engine.
usage: python tools/api_roundtrip.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.optimize import Bounds
from bore_amd.engine import NativeEngine, branin01
from bore_amd.layers import Dense
from bore_amd.models import MaximizableSequential

T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(0)
model = MaximizableSequential(seed=0)
model.add(Dense(16, activation="relu"))
model.add(Dense(16, activation="relu"))
model.add(Dense(1, activation="sigmoid"))
model.compile(optimizer="adam", loss="binary_crossentropy")
bounds = Bounds(np.zeros(2), np.ones(2))
X = rs.uniform(size=(10, 2))
y = branin01(X)


def step():
    global X, y
    z = y < np.quantile(y, 0.25)
    model.fit(X, z, epochs=200, batch_size=64)
    res = model.argmax(bounds, num_starts=3, num_samples=1024, print_fn=lambda s: None, random_state=rs)
    x = rs.uniform(size=2) if res is None else res.x
    X, y = np.vstack([X, x]), np.append(y, branin01(x))


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(T):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
eng = NativeEngine(np.arange(1), async_loops=True, objective="branin01")
eng.run(3)
t1 = time.perf_counter()
eng.run(T)
de = time.perf_counter() - t1
print(f"model API, host arrays every call (PCIe-inclusive): {T / dt:.0f} BO-iterations/s ({1e3 * dt / T:.2f} ms per iteration, N 13..{12 + T}); "
      f"replica engine, one loop, data resident: {T / de:.0f} it/s ({1e3 * de / T:.2f} ms)")
