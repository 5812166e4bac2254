"""Diagnostics (a -DBORE_QUEUE_STAMP_DRAW build): when and where the workgroups of the work-queue launch started.
usage: BORE_LIB_PATH=.../libbore_hip_faststamp.so python tools/wg_starts.py LOOPS"""
import sys, os, ctypes, collections
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bore_amd.engine import NativeEngine
from bore_amd import _lib
L = int(sys.argv[1])
eng = NativeEngine(np.arange(L), async_loops=True, objective="branin01")
eng.run(3)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["BORE_LIB_PATH"])
out = np.zeros((4096, 2), dtype=np.int64)
assert lib.bore_debug_wg_starts(out.ctypes.data_as(ctypes.c_void_p)) == 0
t = out[:, 0]
live = t > 0
n = int(live.sum())
t0 = t[live].min()
us = (t[live] - t0) / 100.0
print(f"{L} loops: {n} workgroups started; start times (us after the first): p50 {np.median(us):.0f} p90 {np.percentile(us, 90):.0f} max {us.max():.0f}")
late = us > 500
print(f"  started within 500 us: {int((~late).sum())}; later: {int(late.sum())}")
hw = out[live, 1] & 0xffffffff
xcc = (out[live, 1] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = [(int(x), int(s), int(a), int(c)) for x, s, a, c in zip(xcc[~late], se[~late], sh[~late], cu[~late])]
cnt = collections.Counter(key)
print(f"  distinct (xcc, se, sh, cu) among the early ones: {len(cnt)}; workgroups per CU histogram: {sorted(collections.Counter(cnt.values()).items())}")
