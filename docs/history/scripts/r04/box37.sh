#!/bin/bash
# round 4, box 37: the driver's command three times in a row (fresh processes) at the round's final code
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box37
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  RG_TRACE_ALLOC=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_run$i.json 2> $OUT/bench_run$i.err
  python scripts/show_bench.py $OUT/bench_run$i.json | grep -E "^value|L +(50|100|200|300|500|1000|2000) |^worst|^gt|rank128|webvid|laion" | cut -c1-200
done
