import numpy as np, time, sys
sys.path.insert(0,'/root/repo')
from roargraph_amd import build
from oracle import gt_numpy
rng=np.random.default_rng(1)
nb,d,r=200000,200,32
A=(rng.standard_normal((r,d))/np.sqrt(r)).astype(np.float32)
base=(rng.standard_normal((nb,r)).astype(np.float32)@A+0.05*rng.standard_normal((nb,d)).astype(np.float32))
nt=40000
tq=((0.3+0.5*rng.standard_normal((nt,r))).astype(np.float32)@A+0.05*rng.standard_normal((nt,d)).astype(np.float32))
q=((0.3+0.5*rng.standard_normal((200,r))).astype(np.float32)@A+0.05*rng.standard_normal((200,d)).astype(np.float32))
t=time.time(); ids=np.concatenate([gt_numpy.groundtruth_blocked(base,tq[i:i+4096],"ip",100)[0] for i in range(0,nt,4096)]); print("gt",time.time()-t,flush=True)
t=time.time(); off,nbrs,ep=build.build_roargraph(base,ids.astype(np.uint32),"ip",100,35,500,num_threads=8); print("build",time.time()-t,nbrs.size/nb,flush=True)
np.savez("/tmp/rescore_hub_ix.npz",off=off,nbrs=nbrs,ep=ep,base=base,q=q)
