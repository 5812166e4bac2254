import sys, os, subprocess, ctypes as C; sys.path.insert(0,'.')
import numpy as np, torch
from bore_amd import _lib
so = os.path.abspath('scratch/libbore_stamp.so')
subprocess.run(['hipcc','-O3','--offload-arch=gfx950','-std=c++17','-shared','-fPIC','-ffp-contract=off','-DBORE_STAMPS',
                'bore_amd/csrc/bore_hip.hip','bore_amd/csrc/bore_argmax.hip','-o',so],check=True, stderr=subprocess.DEVNULL)
_lib.LIB_PATH = so
from bore_amd.engine import ReplicaEngine
lib=_lib.lib()
eng = ReplicaEngine(np.arange(64), groups=1)
names=['cauchy','formk','cmprlb','subsm','lnsrlb','matupd','formt']
for step in range(60):
    lib.bore_debug_lphases_reset()
    eng.step()
    if step % 10 == 9:
        g = eng.groups[0]; info = g.info_pin.numpy()
        out=(C.c_longlong*8)(); lib.bore_debug_lstamps(out); o=list(out)
        ph=(C.c_longlong*16)(); lib.bore_debug_lphases(ph); ph=list(ph)
        print(f"step {step}: N={eng.N} mean nit {info[:,:,0].mean():.1f} max {info[:,:,0].max()} mean nfev {info[:,:,1].mean():.1f} max {info[:,:,1].max()}; prob0: adv {o[0]} fg {o[1]} rounds {o[2]} nit {o[3]}")
        print('   ', ', '.join(f"{nm} {ph[i]}/{ph[8+i]}={ph[i]/max(ph[8+i],1):.0f}" for i,nm in enumerate(names)))
