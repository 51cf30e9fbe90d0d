#!/bin/bash
# round 5, box 11: the balanced split forced where equal items fit one round (RG_GT_BALANCE_ONE=1), with the quota thresholds
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_box11
mkdir -p $OUT
cd $R
GT_FORMS="quota:;quota_bal:RG_GT_BALANCE_ONE=1;own_bal:RG_GT_BALANCE_ONE=1,RG_GT_NOSHARE=1;quota2:;quota_bal2:RG_GT_BALANCE_ONE=1" timeout 900 python scripts/exp/gt_small_batch.py 200 10000000 4096,8192,10000,16384,30000,50000 > $OUT/gt_ab_balance_one.jsonl 2> $OUT/gt.err
cut -c1-190 $OUT/gt_ab_balance_one.jsonl
tail -2 $OUT/gt.err
