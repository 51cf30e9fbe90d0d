#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_box16
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
python scripts/exp/k1_phases.py --nb 2000000 --save /tmp/ix --Ls 30,50,100,200 --modes 2,1 --out $OUT/small_L.json > $OUT/small_L.log 2>&1
# BASELINE config 5 shape: webvid-2.5M d=512 IP, index built in the run
( time python bench.py --dim 512 --nb 2500000 --metric ip --steps 10 --warmup 3 ) > $OUT/bench_webvid.log 2> $OUT/bench_webvid.err
grep -h '^{' $OUT/bench_webvid.log > $OUT/bench_webvid_shape.json
