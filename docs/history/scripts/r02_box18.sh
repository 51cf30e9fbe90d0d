#!/bin/bash
# BASELINE config 4 shape: LAION-10M d=512 L2 top-100, index built in the run
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_box18
mkdir -p $OUT
cd $R
( time python bench.py --dim 512 --nb 10000000 --metric l2 --k 100 --steps 10 --warmup 3 --sweep 100,150,200,300,500,700,1000,2000 --config1-nb 0 ) > $OUT/bench_laion.log 2> $OUT/bench_laion.err
grep -h '^{' $OUT/bench_laion.log > $OUT/bench_laion_shape.json
