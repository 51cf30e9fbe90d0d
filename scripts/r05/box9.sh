#!/bin/bash
# round 5, box 9: K2 quota thresholds between the pieces of a query block -- GT tests, the full-size parity test, A/B against RG_GT_NOSHARE=1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box9
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_groundtruth.py tests/test_gpu_baseline_shapes.py -x -q -k "not config2 and not config4_10m and not config5_2p5m" > $OUT/pytest_gt.log 2>&1; tail -3 $OUT/pytest_gt.log
GT_FORMS="quota:;own:RG_GT_NOSHARE=1;quota2:;own2:RG_GT_NOSHARE=1" timeout 900 python scripts/exp/gt_small_batch.py 200 10000000 2048,8192,10000,16384,30000,65536,100000 > $OUT/gt_ab_quota.jsonl 2> $OUT/gt_ab.err
cut -c1-190 $OUT/gt_ab_quota.jsonl
GT_FORMS="quota:;own:RG_GT_NOSHARE=1" timeout 900 python scripts/exp/gt_small_batch.py 512 3000000 10000,30000,65536 l2 > $OUT/gt_ab_quota_512.jsonl 2>> $OUT/gt_ab.err
cut -c1-190 $OUT/gt_ab_quota_512.jsonl
tail -2 $OUT/gt_ab.err
