#!/bin/bash
# round 4, box 26: the exact LDS set at L_pq 250 - 500 with fewer residents (more LDS per query) against the exact tags
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box26
mkdir -p $OUT
cd $R
# K2 at full fill with 1 / 2 / 4 / 6 / 8 row segments per query block (512 or 510 items for 512 workgroups): what a segment costs
for K in 100 10; do
  GT_K=$K GT_FORMS=default timeout 300 python scripts/exp/gt_small_batch.py 200 10000000 8192,10880,16384,32768,65536 > $OUT/gt_fill_K$K.jsonl 2> $OUT/gt_fill_K$K.err
  cat $OUT/gt_fill_K$K.jsonl
done
timeout 1200 python scripts/exp/k1_ab.py --L 300,400,500 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "auto:visited=2;forced_w8:visited=2,lset=100000;forced_r8_w6:visited=2,lset=100000,rows_per_pass=32,waves_per_cu=6;forced_r4_w7:visited=2,lset=100000,waves_per_cu=7;forced_r4_w6:visited=2,lset=100000,waves_per_cu=6;look:visited=0;filter:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-14s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-14s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows), all(r['same_cmps'] in (None,True) for r in rows if r['config']!='filter'))"
tail -2 $OUT/k1_ab.err
