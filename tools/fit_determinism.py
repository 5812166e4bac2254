"""Run-to-run reproducibility of the wide fits and bit-equality of the eight-wave 6->32-32-1 fit with the four-wave
one: the same launch 80 times, theta / m / v compared bit for bit (GPU box).  usage: python tools/fit_determinism.py"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bore_amd import _lib, ops
from test_gpu_parity import dev, pack, rand_model
rs = np.random.RandomState(11)
def check(D, units, acts, w8_ref, w8_test, compute="float32", runs=80, N=64, E=1):
    desc = _lib.make_desc(D, units, acts, compute=compute)
    th0 = np.stack([pack(rand_model(rs, D, units))])
    X = dev(rs.uniform(size=(1, N, D)), torch.float32)
    z = dev((rs.uniform(size=(1, N)) < 0.25).astype(np.float32))
    def run(w8):
        os.environ["BORE_FIT_W8"] = w8
        th = dev(th0); m, v = torch.zeros_like(th), torch.zeros_like(th)
        t = torch.zeros(1, dtype=torch.int64, device="cuda")
        ops.mlp_fit(desc, th, m, v, t, X, z, E, 64, seed=5)
        return [a.cpu().numpy().ravel() for a in (th, m, v)]
    ref = run(w8_ref)
    bad = 0
    for k in range(runs):
        got = run(w8_test)
        bad += any((a != b).any() for a, b in zip(ref, got))
    print(D, units, compute, "ref w8=%s test w8=%s N=%d E=%d: %d of %d runs differ" % (w8_ref, w8_test, N, E, bad, runs), flush=True)
A3 = ["relu", "elu", "tanh", "sigmoid"]
check(16, [64, 64, 64, 1], A3, "0", "0")
check(16, [64, 64, 64, 1], A3, "0", "0", N=256, E=3)
check(16, [64, 64, 64, 1], A3, "0", "0", compute="bfloat16")
check(32, [128, 128, 1], ["relu", "relu", "sigmoid"], "0", "0", compute="bfloat16")
check(32, [128, 128, 1], ["relu", "relu", "sigmoid"], "0", "0", compute="bfloat16", N=256, E=2)
check(6, [32, 32, 1], ["relu", "relu", "sigmoid"], "0", "1")
check(6, [32, 32, 1], ["relu", "relu", "sigmoid"], "0", "1", N=256, E=3)
