#!/bin/bash
# round 4, box 16: K2 with thresholds shared between the pieces of a query -- parity tests, then the A/B at 10,000 ... 100,000 queries
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_box16
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_groundtruth.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "groundtruth or gt or balanced or shard or streamed or rank or config4_gt or ties" > $OUT/pytest_gt.log 2>&1
tail -5 $OUT/pytest_gt.log
timeout 900 python scripts/exp/gt_small_batch.py 200 10000000 2000,10000,30000,65536,100000 > $OUT/gt_small_batch_200.jsonl 2> $OUT/gt_small.err
timeout 900 python scripts/exp/gt_small_batch.py 512 4000000 10000,65536 l2 > $OUT/gt_small_batch_512.jsonl 2>> $OUT/gt_small.err
cat $OUT/gt_small_batch_200.jsonl $OUT/gt_small_batch_512.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['d'], r['nq'], '%-26s' % r['form'], r['seconds'], r['frac_of_157.3'], r['ids_equal_first_form'])"
tail -3 $OUT/gt_small.err
