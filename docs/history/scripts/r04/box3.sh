#!/bin/bash
# round 4, box 3: (a) memory classes at 2-GiB granularity + compositions; (b) parity tests with the counts in the tail of
# the launch; (c) A/B: counts in the tail (default) vs K4 / end-of-query counts (count_tail=0) vs filter alone
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box3
mkdir -p $OUT
cd $R
timeout 900 scripts/exp/bin/alloc_map2 2 132 > $OUT/alloc_map2.jsonl 2> $OUT/alloc_map2.err
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_concurrency.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 1500 python scripts/exp/k1_ab.py --L 20,50,100,200,500,1000 --index-cache /tmp/ix.npz --pipelined \
  --configs "tail:visited=2;k4:visited=2,count_tail=0;filter:visited=1" \
  > $OUT/k1_ab.jsonl 2> $OUT/k1_ab.err
tail -3 $OUT/k1_ab.err
cat $OUT/alloc_map2.jsonl | cut -c1-400
grep '^{"config' $OUT/k1_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['config'], r['L'], r['pct_of_8TBs'], r['same_ids_hops'], r['same_cmps'])"
