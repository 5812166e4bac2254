import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, time
from bore_amd import _lib, ops
from test_gpu_parity import dev, pack, rand_model
def run(name, D, units, acts, tr, R, Ns, N=256, fit=True, compute='float32'):
    rs=np.random.RandomState(0); desc=_lib.make_desc(D,units,acts,compute=compute); P=ops.param_count(desc)
    th=dev(pack(rand_model(rs,D,units))).reshape(1,-1); m=torch.zeros_like(th); v=torch.zeros_like(th); t=torch.zeros(1,dtype=torch.int64,device='cuda')
    X=rs.uniform(size=(1,N,D)); y=np.sum((X-0.4)**2,axis=2); z=(y<np.quantile(y,0.25)).astype(np.float32)
    torch.cuda.synchronize(); t0=time.perf_counter()
    if fit:
        try:
            ops.mlp_fit(desc,th,m,v,t,dev(X,torch.float32),dev(z),200,64,want_loss=False); torch.cuda.synchronize()
            t0=time.perf_counter(); ops.mlp_fit(desc,th,m,v,t,dev(X,torch.float32),dev(z),200,64,want_loss=False); torch.cuda.synchronize()
            tf=time.perf_counter()-t0
        except RuntimeError as e:
            tf=float('nan'); print("  fit:",str(e)[:100])
    else: tf=float('nan')
    t0=time.perf_counter()
    Xc=ops.uniform_candidates(1,1,Ns,np.zeros(D),np.ones(D)); x0,idx=ops.screen_topk(desc,th,Xc,R); torch.cuda.synchronize(); ts=time.perf_counter()-t0
    t0=time.perf_counter()
    x,fun,jac,info=ops.lbfgsb_minimize(desc,th,x0,np.zeros(D),np.ones(D),tr,True,maxiter=1000,ftol=1e-9); torch.cuda.synchronize(); tl=time.perf_counter()-t0
    info=info.cpu().numpy()[0]
    print(f"{name}: fit(200ep,N={N}) {tf*1e3:.1f} ms; screen+topk {ts*1e3:.2f} ms; lbfgsb R={R}: {tl*1e3:.1f} ms; nit mean {info[:,0].mean():.1f} max {info[:,0].max()}, nfev mean {info[:,1].mean():.1f} max {info[:,1].max()}, status {np.bincount(info[:,2],minlength=3)}")
run("cfg1", 2,[16,16,1],["relu","relu","sigmoid"],"identity",3,1024,N=64)
run("cfg2", 6,[32,32,1],["relu","relu","sigmoid"],"identity",256,1024)
run("cfg3",16,[64,64,64,1],["relu"]*3+["sigmoid"],"identity",1024,1024)
run("cfg3-bf16",16,[64,64,64,1],["relu"]*3+["sigmoid"],"identity",8,1024,compute="bfloat16")
run("cfg5-bf16",32,[128,128,1],["relu","relu","sigmoid"],"identity",4096,4096,compute="bfloat16")
