"""Per-problem cycle breakdown of lbfgsb_kernel (diagnostic -DBORE_STAMPS build): which phases make
the SLOWEST problems of a launch slow.  Run on the GPU box: python scratch/lprob.py"""
import sys, os, subprocess, ctypes as C; sys.path.insert(0, '.')
import numpy as np, torch
from bore_amd import _lib
so = os.path.abspath('scratch/libbore_stamp.so')
subprocess.run(['hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off',
                '-DBORE_STAMPS', *['bore_amd/csrc/' + f for f in _lib.SOURCES], '-o', so],
               check=True, stderr=subprocess.DEVNULL)
_lib.LIB_PATH = so
from bore_amd.engine import ReplicaEngine
lib = _lib.lib()
L = 128
eng = ReplicaEngine(np.arange(L), groups=1)
names = ['cauchy', 'formk', 'cmprlb', 'subsm', 'lnsrlb', 'matupd', 'formt']
buf = np.zeros((4096, 16), dtype=np.uint64)
rows = []
for step in range(100):
    lib.bore_debug_lphases_reset()
    eng.step()
    if step % 10 == 9:
        info = eng.groups[0].info_pin.numpy().copy()           # [L, 3, 5]
        lib.bore_debug_lpp(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
        pp = buf[:4 * L].reshape(L, 4, 16)[:, :3].astype(np.float64)    # [L, 3, 16]
        tot = pp[..., 7] + pp[..., 15]
        q = np.unravel_index(np.argsort(-tot.ravel())[:5], tot.shape)
        print(f"step {step} N={eng.N}: mean total {tot.mean()/1e3:.0f}k cycles, max {tot.max()/1e3:.0f}k; mean nit {info[...,0].mean():.1f} nfev {info[...,1].mean():.1f}")
        ph = pp[..., :7].sum(axis=(0, 1)); cnt = pp[..., 8:15].sum(axis=(0, 1))
        print("   all problems: " + ", ".join(f"{nm} {ph[i]/tot.sum():.0%} ({ph[i]/max(cnt[i],1):.0f}/call x{cnt[i]/tot.size:.1f})" for i, nm in enumerate(names))
              + f", f/g {pp[...,15].sum()/tot.sum():.0%}, other-advance {(pp[...,7].sum()-ph.sum())/tot.sum():.0%}")
        for l, r in zip(*q):
            p = pp[l, r]
            print(f"   slow: {tot[l,r]/1e3:.0f}k nit {info[l,r,0]} nfev {info[l,r,1]} status {info[l,r,2]}: "
                  + ", ".join(f"{nm} {p[i]/1e3:.0f}k/{int(p[8+i])}" for i, nm in enumerate(names))
                  + f", fg {p[15]/1e3:.0f}k, adv-other {(p[7]-p[:7].sum())/1e3:.0f}k")
        rows.append(np.concatenate([pp.reshape(-1, 16), info.reshape(-1, 5)[:, :3]], axis=1))
np.save('gpurun_out/lprob.npy', np.concatenate(rows))
