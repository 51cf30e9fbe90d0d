#!/bin/bash
# round 4, box 46: narrow beams after the scratch fix: rows in flight (16 / 32) and resident queries of the exact-LDS-set form
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box46
mkdir -p $OUT
cd $R
timeout 600 python scripts/exp/k1_ab.py --L 30,50,100,200 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "auto:visited=2;rpp32:visited=2,rows_per_pass=32;w10:visited=2,waves_per_cu=10;w12:visited=2,waves_per_cu=12;w14:visited=2,waves_per_cu=14;w16:visited=2,waves_per_cu=16" > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-10s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-10s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows), all(r['same_cmps'] in (None,True) for r in rows))"
