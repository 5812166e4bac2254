"""argmax_batch (SVGD, 1000 iterations): host particle interaction vs one device launch."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from scipy.optimize import Bounds
import bore_amd
from bore_amd.layers import Dense
from bore_amd.models import BatchMaximizableSequential
for D, units, n in [(2, (16, 16), 8), (2, (16, 16), 64), (6, (32, 32), 16), (16, (64, 64, 64), 24)]:
    rs = np.random.RandomState(0)
    model = BatchMaximizableSequential("sigmoid", seed=2)
    for i, u in enumerate(units):
        model.add(Dense(u, activation="relu", **(dict(input_dim=D) if i == 0 else {})))
    model.add(Dense(1))
    model.compile(optimizer="adam", loss=bore_amd.BinaryCrossentropy(from_logits=True))
    X = rs.uniform(size=(128, D)); y = np.sum((X - 0.3) ** 2, 1)
    model.fit(X, y < np.quantile(y, 0.25), epochs=100, batch_size=64)
    b = Bounds(np.zeros(D), np.ones(D))
    out = {}
    for mode in ("host", "device"):
        model.svgd_mode = mode
        model.argmax_batch(n, b, n_iter=10, random_state=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out[mode] = model.argmax_batch(n, b, random_state=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"D={D} units={units} particles={n} {mode}: {dt*1e3:.1f} ms for 1000 iterations", flush=True)
    print("   max |host - device| =", np.abs(out["host"] - out["device"]).max())
