#!/bin/bash
# round 4, box 13: the rest of the -m gpu suite; resident queries per CU at narrow beams with the exact LDS set (the L_pq 50 / 60 dip of box 12)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box13
mkdir -p $OUT
cd $R
( time timeout 3300 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
timeout 1500 python scripts/exp/k1_ab.py --L 40,50,60,80 --index-cache /tmp/ix.npz --pipelined --nbatch 6 \
  --configs "auto:visited=2;w15:visited=2,waves_per_cu=15;w14:visited=2,waves_per_cu=14;w13:visited=2,waves_per_cu=13;w12:visited=2,waves_per_cu=12;w11:visited=2,waves_per_cu=11;w10:visited=2,waves_per_cu=10;w9:visited=2,waves_per_cu=9;w8:visited=2,waves_per_cu=8;nolset:visited=2,lset=0;filter:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-8s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-8s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows))"
