"""The drop-in model API on a net OUTSIDE the BASELINE shapes' input dimensions (the plugin's default 32-32-1 on a
D-dimensional space): seconds per call of fit (200 epochs) and argmax (GPU box).  usage: python tools/api_generic_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.optimize import Bounds
from bore_amd.layers import Dense
from bore_amd.models import MaximizableSequential
for D, N in ((4, 100), (6, 100), (10, 100)):
    rs = np.random.RandomState(0)
    model = MaximizableSequential(seed=0, transform="sigmoid")
    model.add(Dense(32, activation="relu")); model.add(Dense(32, activation="relu")); model.add(Dense(1, activation="sigmoid"))
    model.compile(optimizer="adam", loss="binary_crossentropy")
    X = rs.uniform(size=(N, D)); y = ((X - 0.3) ** 2).sum(1); z = y < np.quantile(y, 0.25)
    bounds = Bounds(np.zeros(D), np.ones(D))
    for _ in range(3):
        model.fit(X, z, epochs=200, batch_size=64); model.argmax(bounds, num_starts=5, num_samples=1024, print_fn=lambda s: None, random_state=rs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): model.fit(X, z, epochs=200, batch_size=64)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20): model.argmax(bounds, num_starts=5, num_samples=1024, print_fn=lambda s: None, random_state=rs)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{D}->32-32-1, N {N}: fit {1e3 * (t1 - t0) / 20:.2f} ms, argmax (5 starts of 1024 samples) {1e3 * (t2 - t1) / 20:.2f} ms", flush=True)
