import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
from bore_amd import _lib, ops
def run(L,N,E,B=64,D=2,units=(16,16,1),acts=("relu","relu","sigmoid"),reps=5, explicit_perm=False):
    desc=_lib.make_desc(D,list(units),list(acts)); P=ops.param_count(desc)
    rs=np.random.RandomState(0)
    th=torch.from_numpy(rs.normal(scale=.3,size=(L,P)).astype(np.float32)).cuda()
    m=torch.zeros_like(th); v=torch.zeros_like(th); t=torch.zeros(L,dtype=torch.int64,device='cuda')
    X=torch.from_numpy(rs.uniform(size=(L,N,D)).astype(np.float32)).cuda(); z=(torch.rand(L,N,device='cuda')<0.25).float()
    perm=None
    if explicit_perm:
        perm=torch.from_numpy(np.stack([[rs.permutation(N) for _ in range(E)] for _ in range(L)]).astype(np.int32)).cuda()
    ops.mlp_fit(desc,th,m,v,t,X,z,E,B,perm=perm,want_loss=False); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(reps):
        e0.record(); ops.mlp_fit(desc,th,m,v,t,X,z,E,B,perm=perm,want_loss=False); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    steps=E*(-(-N//B))
    print(f"L={L:5d} N={N:4d} E={E:4d} B={B:3d} units={units} perm={'x' if explicit_perm else 'k'}: {min(ts)*1e3:9.1f} us  -> {min(ts)*1e3/steps:7.2f} us/step")
run(1,16,1); run(1,16,200); run(1,16,400); run(1,64,200); run(1,128,200); run(64,16,200); run(64,64,200); run(256,64,200); run(1024,64,200)
run(1,16,200,explicit_perm=True); run(1,64,200,explicit_perm=True)
run(1,64,200,units=(32,32,1),D=6,acts=("relu","relu","linear")); run(1,64,200,units=(64,64,64,1),D=16,acts=("relu","relu","relu","linear"))
run(1,64,200,B=32); run(1,64,200,B=16)
