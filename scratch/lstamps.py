import sys, os, subprocess, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from bore_amd import _lib
so = os.path.abspath('scratch/libbore_stamp.so')
subprocess.run(['hipcc','-O3','--offload-arch=gfx950','-std=c++17','-shared','-fPIC','-ffp-contract=off','-DBORE_STAMPS',
                'bore_amd/csrc/bore_hip.hip','bore_amd/csrc/bore_argmax.hip','-o',so],check=True)
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
        lib.bore_debug_lphases_reset()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); x,fun,jac,info=ops.lbfgsb_minimize(desc,th,dev(X0),np.zeros(D),np.ones(D),tr,True,maxiter=1000,ftol=1e-9); e1.record(); torch.cuda.synchronize()
    out=(C.c_longlong*8)(); lib.bore_debug_lstamps(out)
    o=list(out)
    info=info.cpu().numpy()
    ph=(C.c_longlong*16)(); lib.bore_debug_lphases(ph); ph=list(ph)
    names=['cauchy','formk','cmprlb','subsm','lnsrlb','matupd','formt']
    print('   phases (total cyc / calls = per call):', ', '.join(f"{nm} {ph[i]}/{ph[8+i]}={ph[i]/max(ph[8+i],1):.0f}" for i,nm in enumerate(names)))
    print(f"D={D} {units} R={R} L={L}: kernel {e0.elapsed_time(e1)*1e3:.0f} us; wave0/prob0: advance {o[0]} cyc, fg {o[1]} cyc, rounds {o[2]}, nit {o[3]}; per round adv {o[0]/max(o[2],1):.0f} fg {o[1]/max(o[2],1):.0f}; per nit {(o[0]+o[1])/max(o[3],1):.0f}; mean nfev {info[:,:,1].mean():.1f} max {info[:,:,1].max()} mean nit {info[:,:,0].mean():.1f}")
run(2,[16,16,1],["relu","relu","sigmoid"],"identity",3)
run(2,[16,16,1],["relu","relu","sigmoid"],"sigmoid",3)
run(2,[16,16,1],["relu","relu","linear"],"sigmoid",3)
run(2,[16,16,1],["relu","relu","sigmoid"],"identity",3,L=64)
run(6,[32,32,1],["relu","relu","linear"],"sigmoid",3)
run(6,[32,32,1],["relu","relu","linear"],"sigmoid",64)
run(16,[64,64,64,1],["relu"]*3+["linear"],"sigmoid",8)
