"""Fit launch of BASELINE configs 3 / 5 (and 2) at 1 / 64 / 256 loops, HIP events (GPU box).
usage: [BORE_LIB_PATH=...] python tools/cfg_fit.py [cfg-substring ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
want = sys.argv[1:]
for name, c in bench.WIDE_CONFIGS.items():
    if want and not any(w in name for w in want):
        continue
    for loops in (1, 64, 256):
        c1 = dict(c, R=min(c["R"], 64), Ns=min(c["Ns"], 256))     # (the restart phase is not the subject)
        r = bench.config_gpu(name, c1, loops=loops, reps=3)
        print(f"{name}: {loops} loops: fit {r['ms']['fit']:.2f} ms", flush=True)
