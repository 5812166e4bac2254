import sys; sys.path.insert(0,'.')
import numpy as np, torch
from bore_amd.engine import ReplicaEngine
eng = ReplicaEngine(np.arange(512), groups=1)
rec=[]
for step in range(100):
    eng.step()
    info = eng.groups[0].info_pin.numpy().copy()
    rec.append(info[:,:,:2])
rec=np.array(rec)  # [steps, 512, 3, 2]
np.save('gpurun_out/nit_nfev.npy', rec)
cost = 52e3*rec[...,0] + 6e3*rec[...,1]     # cycles per problem
print("mean problem", cost.mean()/2.4e3, "us;  mean of per-loop max", cost.max(axis=2).mean()/2.4e3, "us")
for G in (1,4,8,16,32,64,128,512):
    per = cost.reshape(100, G, 512//G*3).max(axis=2)   # [steps, G]
    print(f"G={G:4d}: mean group tail {per.mean()/2.4e3:8.1f} us, max {per.max()/2.4e3:8.1f}")
