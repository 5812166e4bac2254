import sys, os, subprocess, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from bore_amd import _lib
so = os.path.abspath('scratch/libbore_acc.so')
subprocess.run(['hipcc','-O3','--offload-arch=gfx950','-std=c++17','-shared','-fPIC','-ffp-contract=off','-Ibore_amd/csrc','-I.',
                'bore_amd/csrc/bore_hip.hip','scratch/bore_argmax_stamp.hip','-o',so],check=True, stderr=subprocess.DEVNULL)
_lib.LIB_PATH = so
from bore_amd.engine import ReplicaEngine
lib=_lib.lib()
eng = ReplicaEngine(np.arange(512), mode="device", groups=1)
for _ in range(12): eng.step()
torch.cuda.synchronize(); lib.bore_debug_acc_reset()
nsteps=5
for _ in range(nsteps): eng.step()
torch.cuda.synchronize()
out=(C.c_ulonglong*16)(); lib.bore_debug_acc(out); o=np.array(list(out),dtype=np.float64)
names=["entry/loop","cauchy","freev","formk","subsm","d=z-x","lnsrlb","ls-tail(cache chk)","newx+conv tests","matupd","formt","FG(wave)","total(thread)","nit","nfev","cmprlb"]
tot=o[12]
for n_,v in zip(names,o): print(f"{n_:22s} {v:14.0f}  {100*v/tot if n_ not in ('nit','nfev') else 0:6.1f}%")
print("problems", 512*3*nsteps, "cycles/problem", tot/(512*3*nsteps), "nit/problem", o[13]/(512*3*nsteps), "nfev/problem", o[14]/(512*3*nsteps))
