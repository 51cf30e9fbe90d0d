#!/bin/bash
# round 4, box 38: look-ahead byte-tag form with an exact LDS set in front of the tags (knob front_set): parity, then A/B at L_pq 300 - 1000
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box38
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "exact_set or exact_words or byte_tags" > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -5
RG_TRACE_ADAPTIVE=1 timeout 1500 python scripts/exp/k1_ab.py --L 300,400,500,700,1000 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "look:visited=0;front70:visited=0,front_set=70;front50:visited=0,front_set=50;front85:visited=0,front_set=85;auto:visited=2" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-14s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-14s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows), all(r['same_cmps'] in (None,True) for r in rows))"
tail -2 $OUT/k1_ab.err
