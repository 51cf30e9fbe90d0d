#!/bin/bash
# the default bench (10M, genuine index) as the driver runs it, output kept under gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG:-r02_bench}
mkdir -p $OUT
cd $R
( time python bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS} ) > $OUT/bench.log 2> $OUT/bench.err
grep -h '^{' $OUT/bench.log > $OUT/bench.json
tail -5 $OUT/bench.err
