import sys, os, subprocess, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
flag = sys.argv[1]
from bore_amd import _lib
so = os.path.abspath(f'scratch/libbore_ab_{flag}.so')
subprocess.run(['hipcc','-O3','--offload-arch=gfx950','-std=c++17','-shared','-fPIC','-ffp-contract=off','-Ibore_amd/csrc'] + ([f'-D{flag}'] if flag!='NONE' else []) +
               ['bore_amd/csrc/bore_hip.hip','bore_amd/csrc/bore_argmax.hip','-o',so],check=True, stderr=subprocess.DEVNULL)
_lib.LIB_PATH = so
import lbfgsb_host as H
H.SO = os.path.abspath(f'scratch/liblbfgsb_host_{flag}.so')
subprocess.run(["g++","-O2","-std=c++17","-shared","-fPIC","-ffp-contract=off"] + ([f'-D{flag}'] if flag!='NONE' else []) + [H.SRC,"-o",H.SO],check=True)
H._lib = C.CDLL(H.SO); H._lib.lbfgsb_host_minimize.restype = C.c_int
from bore_amd import ops
from test_gpu_parity import dev, pack, rand_model
D, units, acts, tr, R = 3, [32, 32, 32, 1], ["elu"] * 3 + ["linear"], "exp", 9
rs = np.random.RandomState(D); desc = _lib.make_desc(D, units, acts); L = 2
params = [rand_model(rs, D, units) for _ in range(L)]
th = dev(np.stack([pack(p) for p in params]))
X0 = rs.uniform(-0.1, 1.1, size=(L, R, D)); lo, hi = np.zeros(D), np.ones(D)
opts = dict(maxiter=1000, ftol=1e-9)
x, fun, jac, info = (t.cpu().numpy() for t in ops.lbfgsb_minimize(desc, th, dev(X0), lo, hi, tr, True, **opts))
bad=0
for l in range(L):
    def fg(xx):
        v, g = ops.mlp_value_and_input_grad(desc, th[l:l+1], dev(np.atleast_2d(xx)[None]), tr, True)
        return v.cpu().numpy()[0,0], g.cpu().numpy()[0,0]
    for r in range(R):
        h = H.minimize(fg, X0[l, r], (lo, hi), **opts)
        if not np.array_equal(h.x, x[l, r]): bad+=1; print(flag,'mismatch',l,r,h.x,x[l,r],(h.nit,h.nfev),info[l,r,:2])
print(flag,'mismatches',bad)
