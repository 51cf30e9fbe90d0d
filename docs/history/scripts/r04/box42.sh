#!/bin/bash
# round 4, box 42: at L_pq 220 - 300, the exact LDS set with the tags behind it (default there) against the look-ahead tags with the set in front
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box42
mkdir -p $OUT
cd $R
timeout 900 python scripts/exp/k1_ab.py --L 220,240,260,280,300 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "auto:visited=2;front_only:visited=2,lset_tags=0;look_front:visited=0;lset_tags_always:visited=2,lset_tags=2" > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-22s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-22s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows), all(r['same_cmps'] in (None,True) for r in rows))"
tail -1 $OUT/k1_ab.err
