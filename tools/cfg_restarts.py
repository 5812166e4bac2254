"""Restart-phase time of a BASELINE wide config with many loops (GPU box): fit / screen / restarts in
ms through bench.config_gpu.  usage: python tools/cfg_restarts.py <cfg2|cfg3|cfg5> [loops] [reps]
(BORE_LIB_PATH selects an experiment build.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
key = {"cfg2": "cfg2_hartmann6_32-32-1_R256", "cfg3": "cfg3_hpo16_64-64-64-1_R1024", "cfg5": "cfg5_nas32_128-128-1_bf16_R4096"}[sys.argv[1]]
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
r = bench.config_gpu(key, bench.WIDE_CONFIGS[key], loops=loops, reps=reps)
print(os.path.basename(os.environ.get("BORE_LIB_PATH", "default")), sys.argv[1], loops, {k: round(v, 2) for k, v in r["ms"].items()},
      "fg rows/iteration %.0f" % r["fg_rows_per_iteration"], "ok frac %.3f" % r["restarts_ok_frac"], flush=True)
