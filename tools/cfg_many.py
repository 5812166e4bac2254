"""BASELINE configs 2 / 3 / 5 with many loops: time of the restart launch under the two problem-to-
lane mappings of lbfgsb_kernel (GPU box).  usage: BORE_LBFGSB_COOP_GRID=<n> python tools/cfg_many.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
for name, c in bench.WIDE_CONFIGS.items():
    loops = 64 if c["R"] >= 4096 else 256
    r = bench.config_gpu(name, c, loops=loops, reps=2)
    print(f"coop grid max {os.environ.get('BORE_LBFGSB_COOP_GRID', '8192'):>8s} | {name}: {loops} loops: fit {r['ms']['fit']:.1f} ms, "
          f"screen {r['ms']['screen']:.2f}, restarts {r['ms']['lbfgsb']:.1f} ms ({r['fg_rows_per_iteration']:.0f} f/g rows), "
          f"{r['it_per_s']:.0f} it/s", flush=True)
