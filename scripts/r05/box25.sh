#!/bin/bash
# round 5, box 25: K2 quick reject in the tile filter (one maximum tree + one compare per tile) against the row-by-row test (RG_GT_DIAG=32)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box25
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_groundtruth.py -x -q > $OUT/pytest_gt.log 2>&1; tail -2 $OUT/pytest_gt.log
GT_FORMS="quick:;rows:RG_GT_DIAG=32;quick2:;rows2:RG_GT_DIAG=32" timeout 900 python scripts/exp/gt_small_batch.py 200 10000000 10000,30000,65536,100000 > $OUT/gt_ab_quick.jsonl 2> $OUT/gt.err
cut -c1-190 $OUT/gt_ab_quick.jsonl
GT_FORMS="quick:;rows:RG_GT_DIAG=32" timeout 900 python scripts/exp/gt_small_batch.py 512 3000000 10000,65536 l2 > $OUT/gt_ab_quick_512.jsonl 2>> $OUT/gt.err
cut -c1-190 $OUT/gt_ab_quick_512.jsonl
tail -2 $OUT/gt.err
