#!/bin/bash
# round 6, box 7: K2's compaction event with the next query's keys requested ahead (gt_select_event) against the selection query by query (RG_GT_DIAG=32)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_box7
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_groundtruth.py tests/test_gpu_baseline_shapes.py -m gpu -q -x > $OUT/pytest_gt.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gt.log
GT_FORMS="event:;per_query:RG_GT_DIAG=32;event_again:;per_query_again:RG_GT_DIAG=32" timeout 600 python scripts/exp/gt_small_batch.py 200 10000000 10000,30000,65536 ip > $OUT/gt_d200_event.jsonl 2> $OUT/gt_d200_event.err; cat $OUT/gt_d200_event.jsonl
for M in ip l2; do
  GT_FORMS="event:;per_query:RG_GT_DIAG=32;event_again:;per_query_again:RG_GT_DIAG=32" timeout 600 python scripts/exp/gt_small_batch.py 512 3000000 10000,30000,65536 $M > $OUT/gt_d512_event_$M.jsonl 2> $OUT/gt_d512_event_$M.err; cat $OUT/gt_d512_event_$M.jsonl
done
