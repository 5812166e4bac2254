import sys, os, subprocess, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from bore_amd import _lib
so = os.path.abspath('scratch/libbore_lstamp.so')
subprocess.run(['hipcc','-O3','--offload-arch=gfx950','-std=c++17','-shared','-fPIC','-ffp-contract=off','-Ibore_amd/csrc',
                'bore_amd/csrc/bore_hip.hip','scratch/bore_argmax_stamp.hip','-o',so],check=True)
_lib.LIB_PATH = so
from bore_amd import ops
from test_gpu_parity import dev, pack, rand_model
lib=_lib.lib()
def run(D,units,acts,tr,R,L=1):
    rs=np.random.RandomState(1)
    desc=_lib.make_desc(D,units,acts)
    th=dev(np.stack([pack(rand_model(rs,D,units)) for _ in range(L)]))
    X0=rs.uniform(size=(L,R,D))
    for _ in range(2):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); x,fun,jac,info=ops.lbfgsb_minimize(desc,th,dev(X0),np.zeros(D),np.ones(D),tr,True,maxiter=1000,ftol=1e-9); e1.record(); torch.cuda.synchronize()
    o2=(C.c_longlong*16)(); lib.bore_debug_ls(o2); o2=list(o2); print('   LS round stamps: deltas [entry,copy->lnsrlb,ddot,dcsrch,xupd,cachechk,tail]', [o2[i+1]-o2[i] for i in range(7)])
    out=(C.c_longlong*8)(); lib.bore_debug_lstamps(out)
    o=list(out)
    info=info.cpu().numpy()
    print(f"D={D} {units} R={R} L={L}: kernel {e0.elapsed_time(e1)*1e3:.0f} us; wave0/prob0: advance {o[0]} cyc, fg {o[1]} cyc, rounds {o[2]}, nit {o[3]}, first-adv {o[4]}; per round adv {o[0]/max(o[2],1):.0f} fg {o[1]/max(o[2],1):.0f}; mean nfev {info[:,:,1].mean():.1f} max {info[:,:,1].max()} mean nit {info[:,:,0].mean():.1f}")
run(2,[16,16,1],["relu","relu","sigmoid"],"identity",3)
run(2,[16,16,1],["relu","relu","sigmoid"],"identity",3,L=64)
run(6,[32,32,1],["relu","relu","linear"],"sigmoid",3)
run(6,[32,32,1],["relu","relu","linear"],"sigmoid",64)
run(16,[64,64,64,1],["relu"]*3+["linear"],"sigmoid",8)
