#!/bin/bash
# round 4, box 18: the allocator after an in-process build (box 17, run 1: two classes found, plain fallbacks) -- step trace, three fresh builds
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box18
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  RG_TRACE_ALLOC=2 timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --sweep 1000,2000 --L 50 --no-worstcase --no-fast --no-two-streams --gt-nq 0 --cpu-seconds 0 --config1-nb 0 --configs '' > $OUT/bench_run$i.json 2> $OUT/bench_run$i.err
  python -c "
import json
l=[x for x in open('$OUT/bench_run$i.json') if x.startswith('{')]
r=json.loads(l[-1]); print('run $i', round(r['roofline']['frac'],4), [(p['L_pq'], round(p['pct_of_8000'],1)) for p in r['L_pq_sweep']], r['device_memory'])"
  grep "rg_mem\] [0-9d]" $OUT/bench_run$i.err | cut -c1-200
  grep -c "class 0" $OUT/bench_run$i.err; grep -c "class 1" $OUT/bench_run$i.err; grep -c "class 2" $OUT/bench_run$i.err
done
