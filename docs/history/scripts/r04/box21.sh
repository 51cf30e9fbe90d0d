#!/bin/bash
# round 4, box 21: how far up the exact LDS set pays when queries may outgrow it (forced form) -- L_pq 100 ... 700
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box21
mkdir -p $OUT
cd $R
RG_TRACE_ALLOC=1 timeout 1500 python scripts/exp/k1_ab.py --L 100,150,200,300,500,700 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "forced:visited=2,lset=100000;forced_r8:visited=2,lset=100000,rows_per_pass=32;nolset:visited=2,lset=0;look:visited=0;filter:visited=1;auto:visited=2" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep "rg_mem\] [0-9d]" $OUT/k1_ab.err | cut -c1-150
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-10s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-10s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows), all(r['same_cmps'] in (None,True) for r in rows if r['config']!='filter'))"
