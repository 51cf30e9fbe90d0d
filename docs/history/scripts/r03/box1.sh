#!/bin/bash
# round 3, first box: the -m gpu suite (new: cosine goldens, look-ahead form of the exact words, entry point vs oracle), then the
# A/B of the K1 forms on the genuine 10M index in one process
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_box1
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.log 2>&1; echo "gpu rc=$?" >> $OUT/tests_gpu.log
tail -5 $OUT/tests_gpu.log
timeout 1500 python scripts/exp/k1_ab.py --L 300,500,700,1000,2000 --index-cache /tmp/ix.npz \
  --configs "atomics:visited=0,lookahead=0;look:visited=0,lookahead=1;look_nofilter:visited=0,lookahead=1,exact_filter=0;look_r4:visited=0,lookahead=1,rows_per_pass=16;look_r4_nofilter:visited=0,lookahead=1,rows_per_pass=16,exact_filter=0;atomics_r4:visited=0,lookahead=0,rows_per_pass=16;filter:visited=1;default:visited=2" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
tail -50 $OUT/k1_ab.jsonl; tail -5 $OUT/k1_ab.err
