#!/bin/bash
# round 4, box 20: the exact LDS set with the bucketed side table -- parity, then where the capacity cliff is now (forced form, residents swept)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box20
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_concurrency.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -n "passed\|failed" $OUT/pytest.log | tail -2
RG_TRACE_ADAPTIVE=1 timeout 1200 python scripts/exp/k1_ab.py --L 50,60,80,100,120,150 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "auto:visited=2;f13:visited=2,lset=100000,waves_per_cu=13;f12:visited=2,lset=100000,waves_per_cu=12;f11:visited=2,lset=100000,waves_per_cu=11;f10:visited=2,lset=100000,waves_per_cu=10;f9:visited=2,lset=100000,waves_per_cu=9;f8:visited=2,lset=100000,waves_per_cu=8;nolset:visited=2,lset=0;filter:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep "exact LDS set at" $OUT/k1_ab.err | sed 's/.*at L=/L=/' | sort | uniq -c | sort -k2,2 -k1,1rn | head -60
grep "outgrew" $OUT/k1_ab.err | sed 's/.*batch //' | sort | uniq -c | sort -rn | head -40
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-8s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-8s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows), all(r['same_cmps'] in (None,True) for r in rows if r['config']!='filter'))"
