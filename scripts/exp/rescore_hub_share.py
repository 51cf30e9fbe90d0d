import numpy as np, sys
z=np.load('/tmp/rescore_hub_ix.npz'); off=z['off'].astype(np.int64); nbrs=z['nbrs']; ep=int(z['ep']); base=z['base']; Q=z['q']
nb=base.shape[0]
indeg=np.bincount(nbrs,minlength=nb)
rank=np.empty(nb,np.int64); rank[np.argsort(-indeg,kind='stable')]=np.arange(nb)
print("avg deg",nbrs.size/nb,"max indeg",indeg.max(),"indeg share of top 256/1024/4096/16384:",[round(float(np.sort(indeg)[::-1][:h].sum()/nbrs.size),3) for h in (256,1024,4096,16384)])
id_bits=int(np.ceil(np.log2(nb)))
def run(L,F,nq=60):
    tot_tests=tot_fresh=tot_forgot=0; forgot_rank=[]; tests_rank=[]
    mask=(1<<id_bits)-1; rem_bits=id_bits-F
    for qi in range(nq):
        q=Q[qi]
        filt=np.full(1<<F,-1,np.int64)
        scored=set()
        beam=[(float(-(base[ep]@q)),ep,False)]
        scored_ep=False
        while True:
            # closest unexpanded
            idx=next((i for i,e in enumerate(beam) if not e[2]),None)
            if idx is None: break
            dcur,u,_=beam[idx]; beam[idx]=(dcur,u,True)
            ns=nbrs[off[u]:off[u+1]]
            cand=[]
            for v in ns:
                v=int(v); tot_tests+=1
                x=(v*0x9E3779B1)&mask; slot=x>>rem_bits
                hit = filt[slot]==v
                if hit: continue
                filt[slot]=v
                if v in scored:
                    tot_forgot+=1; forgot_rank.append(rank[v])
                else:
                    scored.add(v); tot_fresh+=1
                cand.append(v)
            if cand:
                dd=-(base[cand]@q)
                for v,dv in zip(cand,dd):
                    dv=float(dv)
                    if len(beam)>=L and (dv,v)>=(beam[-1][0],beam[-1][1]): continue
                    if any(e[1]==v for e in beam): continue
                    beam.append((dv,v,False)); beam.sort(key=lambda e:(e[0],e[1])); 
                    if len(beam)>L: beam.pop()
    fr=np.array(forgot_rank)
    print("L",L,"filter 2^%d"%F,"tests/q",tot_tests/nq,"fresh/q",tot_fresh/nq,"rescored/q",tot_forgot/nq,"ratio %.3f"%(1+tot_forgot/max(1,tot_fresh)),
          "share of re-scores on top-H hubs:",{h:round(float((fr<h).mean()),3) for h in (256,1024,4096,16384,65536)},flush=True)
for L,F in ((200,12),(500,12)):
    run(L,F,nq=40)
