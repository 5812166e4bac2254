"""Where a BO iteration through the model API spends its host time (GPU box): per-call wall time of fit / argmax and of
the caller's own lines, and the GPU's busy time per iteration (kernel durations from HIP events).
usage: python tools/api_timeline.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.optimize import Bounds
from bore_amd.engine import branin01
from bore_amd.layers import Dense
from bore_amd.models import MaximizableSequential

T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(0)
model = MaximizableSequential(seed=0)
for u, a in ((16, "relu"), (16, "relu"), (1, "sigmoid")):
    model.add(Dense(u, activation=a))
model.compile(optimizer="adam", loss="binary_crossentropy")
bounds = Bounds(np.zeros(2), np.ones(2))
X = rs.uniform(size=(10, 2))
y = branin01(X)
acc = dict(user=0.0, fit=0.0, argmax=0.0, gpu=0.0)
pc = time.perf_counter


def step(timed):
    global X, y
    t0 = pc()
    z = y < np.quantile(y, 0.25)
    t1 = pc()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.fit(X, z, epochs=200, batch_size=64)
    t2 = pc()
    res = model.argmax(bounds, num_starts=3, num_samples=1024, print_fn=lambda s: None, random_state=rs)
    e1.record()
    t3 = pc()
    x = rs.uniform(size=2) if res is None else res.x
    X, y = np.vstack([X, x]), np.append(y, branin01(x))
    t4 = pc()
    if timed:
        e1.synchronize()
        acc["user"] += (t1 - t0) + (t4 - t3); acc["fit"] += t2 - t1; acc["argmax"] += t3 - t2
        acc["gpu"] += 1e-3 * e0.elapsed_time(e1)


for _ in range(3):
    step(False)
torch.cuda.synchronize()
t0 = pc()
for _ in range(T):
    step(True)
torch.cuda.synchronize()
dt = pc() - t0
print(f"{1e3 * dt / T:.3f} ms per iteration: caller's own lines {1e3 * acc['user'] / T:.3f}, fit() returns after {1e3 * acc['fit'] / T:.3f}, "
      f"argmax() {1e3 * acc['argmax'] / T:.3f}; first kernel enqueued -> last result on the stream {1e3 * acc['gpu'] / T:.3f} ms")
