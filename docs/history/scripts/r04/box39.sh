#!/bin/bash
# round 4, box 39: the front set in the default rule: GPU suite, smoke, A/B against front_set = 0, then the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box39
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
RG_TRACE_ADAPTIVE=1 timeout 1200 python scripts/exp/k1_ab.py --L 240,280,300,350,400,500 --index-cache /tmp/ix.npz --pipelined --nbatch 4 \
  --configs "auto:visited=2;no_front:visited=2,front_set=0;no_front_no_lset_tags:visited=2,front_set=0,lset_tags=0" > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
Ls=sorted({r['L'] for r in rows}); cf=[]
for r in rows:
    if r['config'] not in cf: cf.append(r['config'])
print('%-22s'%'cfg'+''.join('%8d'%L for L in Ls))
for c in cf: print('%-22s'%c+''.join('%8.1f'%next((r['pct_of_8TBs'] for r in rows if r['config']==c and r['L']==L),0) for L in Ls))
print('exact', all(r['same_ids_hops'] in (None,True) for r in rows), all(r['same_cmps'] in (None,True) for r in rows))"
RG_TRACE_ALLOC=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python scripts/show_bench.py $OUT/bench.json | cut -c1-220
