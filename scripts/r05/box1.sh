#!/bin/bash
# round 5, box 1: new parity tests (hub bits, ADVICE repeat-edge case), the bench's compact line on a small run, and the first A/B of
# the hub bitmap in front of the look-ahead tags on the 10M bench index
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box1
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "hub or repeats or look_ahead or exact_words or byte_tags" > $OUT/pytest_hub.log 2>&1
tail -5 $OUT/pytest_hub.log
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -k "bench_one_gpu or bench_multi_rank" > $OUT/pytest_bench.log 2>&1
tail -5 $OUT/pytest_bench.log
timeout 1500 python scripts/exp/k1_ab.py --L 200,300,400,500,700,1000,1500,2000 --nbatch 3 --reps 2 --index-cache /tmp/ix.npz \
  --configs "off:visited=0,lookahead=1,hub_bits=0;auto60:visited=0,lookahead=1;pct40:visited=0,lookahead=1,hub_pct=40;pct80:visited=0,lookahead=1,hub_pct=80;auto_nofs:visited=0,lookahead=1,front_set=0;off_nofs:visited=0,lookahead=1,hub_bits=0,front_set=0" \
  > $OUT/k1_ab_hub.jsonl 2> $OUT/k1_ab_hub.err
cat $OUT/k1_ab_hub.jsonl
tail -3 $OUT/k1_ab_hub.err
