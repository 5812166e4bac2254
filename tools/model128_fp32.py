import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from scipy.optimize import Bounds
from bore_amd.layers import Dense
from bore_amd.models import MaximizableSequential
rs = np.random.RandomState(0)
m = MaximizableSequential(seed=0)
m.add(Dense(128, activation="relu")); m.add(Dense(128, activation="relu")); m.add(Dense(1, activation="sigmoid"))
m.compile(optimizer="adam", loss="binary_crossentropy")
X = rs.uniform(size=(256, 32)); y = ((X - 0.4) ** 2).sum(1); z = y < np.quantile(y, 0.25)
h = m.fit(X, z, epochs=30, batch_size=64)
print("loss", h.history["loss"][0], "->", h.history["loss"][-1])
res = m.argmax(Bounds(np.zeros(32), np.ones(32)), num_starts=8, num_samples=1024, print_fn=lambda s: None, random_state=np.random.RandomState(1))
print("argmax", None if res is None else (res.fun, res.nit))
