import numpy as np, time, sys, hashlib
sys.path.insert(0,'/root/repo')
from roargraph_amd import build
from oracle import gt_numpy
out=sys.argv[1]
res={}
for (nb,d,metric,seed) in [(20000,200,"ip",1),(8000,512,"l2",2),(12000,200,"l2",3)]:
    rng=np.random.default_rng(seed)
    r=12
    A=(rng.standard_normal((r,d))/np.sqrt(r)).astype(np.float32)
    base=(rng.standard_normal((nb,r)).astype(np.float32)@A+0.05*rng.standard_normal((nb,d)).astype(np.float32))
    tq=((0.3+0.5*rng.standard_normal((nb//4,r))).astype(np.float32)@A+0.05*rng.standard_normal((nb//4,d)).astype(np.float32))
    ids,_=gt_numpy.groundtruth_blocked(base,tq,metric,100)
    t=time.time()
    off,nbrs,ep=build.build_roargraph(base,ids.astype(np.uint32),metric,100,35,500,num_threads=1)
    h=hashlib.md5(off.tobytes()+nbrs.tobytes()+bytes([ep%256])).hexdigest()
    print(nb,d,metric,"T=1 build %.1fs avgdeg %.1f"%(time.time()-t,nbrs.size/nb),h,flush=True)
    res[(nb,d,metric)]=h
open(out,'w').write(repr(res))
