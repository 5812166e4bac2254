import sys, os, subprocess, ctypes as C; sys.path.insert(0,'.')
import numpy as np, torch
from bore_amd import _lib
so = os.path.abspath('scratch/libbore_stamp.so')
subprocess.run(['hipcc','-O3','--offload-arch=gfx950','-std=c++17','-shared','-fPIC','-ffp-contract=off','-DBORE_STAMPS',
                'bore_amd/csrc/bore_hip.hip','bore_amd/csrc/bore_argmax.hip','-o',so],check=True)
_lib.LIB_PATH = so
from bore_amd import ops
lib=_lib.lib()
def run(N,units=(16,16,1),D=2,acts=("relu","relu","sigmoid")):
    desc=_lib.make_desc(D,list(units),list(acts)); P=ops.param_count(desc); L=1
    rs=np.random.RandomState(0)
    th=torch.from_numpy(rs.normal(scale=.3,size=(L,P)).astype(np.float32)).cuda()
    m=torch.zeros_like(th); v=torch.zeros_like(th); t=torch.zeros(L,dtype=torch.int64,device='cuda')
    X=torch.from_numpy(rs.uniform(size=(L,N,D)).astype(np.float32)).cuda(); z=(torch.rand(L,N,device='cuda')<0.25).float()
    for _ in range(3): ops.mlp_fit(desc,th,m,v,t,X,z,5,64,want_loss=False)
    torch.cuda.synchronize()
    out=(C.c_longlong*64)(); lib.bore_debug_stamps(out)
    for w in range(4):
        a=np.array(out[16*w:16*w+12]); print(N,units,'wave',w,'deltas',np.diff(a[:8]),'task: pre',a[8]-a[5],'loads',a[9]-a[8],'mfma',a[10]-a[9],'adam',a[11]-a[10],'post',a[6]-a[11])
run(64,(64,64,64,1),16,("relu","relu","relu","linear"))
